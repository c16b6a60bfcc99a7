"""gym.Env / gym.Wrapper protocol (old 4-tuple step API, ``reset() -> obs``, ``seed(int)``)."""


class Env:
    metadata = {}
    reward_range = (-float("inf"), float("inf"))
    spec = None
    action_space = None
    observation_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def seed(self, seed=None):
        return []

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    """Forwards everything to ``self.env`` (attribute lookups included)."""

    def __init__(self, env):
        self.env = env
        self._action_space = None
        self._observation_space = None

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def spec(self):
        return self.env.spec

    @property
    def action_space(self):
        return self.env.action_space if self._action_space is None else self._action_space

    @action_space.setter
    def action_space(self, space):
        self._action_space = space

    @property
    def observation_space(self):
        return self.env.observation_space if self._observation_space is None else self._observation_space

    @observation_space.setter
    def observation_space(self, space):
        self._observation_space = space

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def seed(self, seed=None):
        return self.env.seed(seed)

    def close(self):
        return self.env.close()

    @property
    def unwrapped(self):
        return self.env.unwrapped
