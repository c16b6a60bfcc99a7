"""ActorCritic with the reference's constructor, initialisation and method surface (ppo.py:25-59), whose
parameters are views into ONE flat fp32 device buffer — the layout the HIP kernels consume."""
import numpy as np
import torch
from torch import nn
from torch.distributions import Categorical

from . import _native as N


def layer_init(layer, std=np.sqrt(2), bias_const=0.0):
    """ppo.py:25-28."""
    torch.nn.init.orthogonal_(layer.weight, std)
    torch.nn.init.constant_(layer.bias, bias_const)
    return layer


class ActorCritic(nn.Module):
    """actor 4->64->64->n_actions, critic 4->64->64->1, tanh (ppo.py:31-47).

    Construction happens on the CPU in the reference's layer order so that ``torch.manual_seed(s)`` yields the
    reference's initial weights; the 12 parameter tensors are then re-pointed at slices of ``self.flat``
    (order of ``agent.parameters()`` == include/mi_rl.h "Parameter layout").
    """

    def __init__(self, env, device=None):
        super().__init__()
        obs_dim = int(np.array(env.observation_space.shape).prod())
        n_act = env.action_space.n
        if obs_dim != 4 or n_act != 2:
            raise N.MiError("the HIP kernels are specialised for CartPole (obs 4, actions 2); got %d/%d" % (obs_dim, n_act))
        self.actor = nn.Sequential(
            layer_init(nn.Linear(obs_dim, 64)), nn.Tanh(),
            layer_init(nn.Linear(64, 64)), nn.Tanh(),
            layer_init(nn.Linear(64, n_act), std=0.01),
        )
        self.critic = nn.Sequential(
            layer_init(nn.Linear(obs_dim, 64)), nn.Tanh(),
            layer_init(nn.Linear(64, 64)), nn.Tanh(),
            layer_init(nn.Linear(64, 1), std=1.0),
        )
        dev = torch.device(device if device is not None else getattr(env, "device", "cuda"))
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in self.parameters()]).to(dev, torch.float32).contiguous()
        assert flat.numel() == N.NPARAMS
        self.flat = flat
        off = 0
        for p in self.parameters():
            n = p.numel()
            p.data = flat[off:off + n].view(p.shape)
            off += n
        self.device = dev

    # -- helpers -------------------------------------------------------------------------------
    def load_flat(self, vec):
        """Overwrite all parameters from a flat vector (numpy or tensor) in parameters() order."""
        self.flat.copy_(torch.as_tensor(vec, dtype=torch.float32).reshape(-1).to(self.device))

    def _forward(self, observation, want_logits, want_value):
        obs = observation.to(self.device, torch.float32)
        lead = obs.shape[:-1]
        obs = obs.reshape(-1, 4).contiguous()
        n = obs.shape[0]
        logits = torch.empty((n, 2), dtype=torch.float32, device=self.device) if want_logits else None
        value = torch.empty(n, dtype=torch.float32, device=self.device) if want_value else None
        N.check(N.lib().mi_ppo_forward(N.ptr(self.flat), N.ptr(obs), n, N.ptr(logits), N.ptr(value), N.stream_ptr(self.device)),
                "mi_ppo_forward")
        return (logits.reshape(*lead, 2) if want_logits else None), (value.reshape(lead) if want_value else None)

    # -- reference surface (inference; training gradients come from mi_ppo_minibatch_grad) ----------------
    def get_value(self, observation):
        """ppo.py:49-50."""
        return self._forward(observation, False, True)[1]

    def get_action_distribution(self, observation):
        """ppo.py:52-54."""
        return Categorical(logits=self._forward(observation, True, False)[0])

    def get_action(self, observation):
        """ppo.py:56-59."""
        distribution = self.get_action_distribution(observation)
        action = distribution.sample()
        return action, distribution.log_prob(action)


class QNetwork(nn.Module):
    """QNetwork of the reference dqn.py:24-36 (4 -> 120 -> 84 -> n_actions, ReLU, torch default init) whose 6 parameter
    tensors are views into one flat fp32 device buffer (include/mi_rl.h "DQN")."""

    def __init__(self, env, device=None):
        super().__init__()
        obs_dim = int(np.prod(env.observation_space.shape))
        if obs_dim != 4 or env.action_space.n != 2:
            raise N.MiError("the HIP kernels are specialised for CartPole (obs 4, actions 2)")
        self.network = nn.Sequential(nn.Linear(obs_dim, 120), nn.ReLU(), nn.Linear(120, 84), nn.ReLU(), nn.Linear(84, env.action_space.n))
        dev = torch.device(device if device is not None else getattr(env, "device", "cuda"))
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in self.parameters()]).to(dev, torch.float32).contiguous()
        assert flat.numel() == N.DQN_NPARAMS
        self.flat = flat
        off = 0
        for p in self.parameters():
            n = p.numel()
            p.data = flat[off:off + n].view(p.shape)
            off += n
        self.device = dev

    def load_flat(self, vec):
        self.flat.copy_(torch.as_tensor(vec, dtype=torch.float32).reshape(-1).to(self.device))

    def load_state_dict(self, state_dict, *a, **kw):
        """target_network.load_state_dict(q_network.state_dict()) (dqn.py:70,137): keeps the flat-buffer views intact."""
        own = dict(self.named_parameters())
        with torch.no_grad():
            for k, v in state_dict.items():
                own[k].copy_(v)

    def forward(self, observation):
        """dqn.py:35-36."""
        obs = observation.to(self.device, torch.float32)
        lead = obs.shape[:-1]
        obs = obs.reshape(-1, 4).contiguous()
        q = torch.empty((obs.shape[0], 2), dtype=torch.float32, device=self.device)
        N.check(N.lib().mi_dqn_forward(N.ptr(self.flat), N.ptr(obs), obs.shape[0], N.ptr(q), N.stream_ptr(self.device)), "mi_dqn_forward")
        return q.reshape(*lead, 2)


def _bind_flat(module, flat):
    """Re-point every parameter of `module` at its slice of `flat` (parameters() order)."""
    off = 0
    for p in module.parameters():
        n = p.numel()
        p.data = flat[off:off + n].view(p.shape)
        off += n
    assert off == flat.numel()
    module.flat = flat


class _FlatModule(nn.Module):
    def _finish(self, env, device, nparams):
        dev = torch.device(device if device is not None else getattr(env, "device", "cuda"))
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in self.parameters()]).to(dev, torch.float32).contiguous()
        assert flat.numel() == nparams
        _bind_flat(self, flat)
        self.device = dev

    def load_flat(self, vec):
        self.flat.copy_(torch.as_tensor(vec, dtype=torch.float32).reshape(-1).to(self.device))

    def load_state_dict(self, state_dict, *a, **kw):
        """qf1_target.load_state_dict(qf1.state_dict()) (sac.py:115-116): keeps the flat-buffer views intact."""
        own = dict(self.named_parameters())
        with torch.no_grad():
            for k, v in state_dict.items():
                if k in own:
                    own[k].copy_(v)


def pack(*modules):
    """Lay the flat buffers of several modules back to back in ONE buffer (the twin critics share one Adam, sac.py:117) and
    re-point their parameters; returns the joint buffer."""
    joint = torch.cat([m.flat for m in modules]).contiguous()
    off = 0
    for m in modules:
        n = m.flat.numel()
        _bind_flat(m, joint[off:off + n])
        off += n
    return joint


class SoftQNetwork(_FlatModule):
    """SoftQNetwork of the reference sac.py:29-43 (cat(obs, action) -> 256 -> 256 -> 1, ReLU, torch default init)."""

    def __init__(self, env, device=None):
        super().__init__()
        obs_dim, act_dim = int(np.prod(env.observation_space.shape)), int(np.prod(env.action_space.shape))
        if obs_dim != 3 or act_dim != 1:
            raise N.MiError("the SAC kernels are specialised for Pendulum (obs 3, action 1); got %d/%d" % (obs_dim, act_dim))
        self.network = nn.Sequential(nn.Linear(obs_dim + act_dim, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 1))
        self._finish(env, device, N.SAC_Q_NPARAMS)

    def forward(self, observation, action):
        """sac.py:40-43."""
        obs = observation.to(self.device, torch.float32).reshape(-1, 3).contiguous()
        act = action.to(self.device, torch.float32).reshape(-1).contiguous()
        out = torch.empty(obs.shape[0], dtype=torch.float32, device=self.device)
        N.check(N.lib().mi_sac_q_forward(N.ptr(self.flat), N.ptr(obs), N.ptr(act), obs.shape[0], N.ptr(out), N.stream_ptr(self.device)), "mi_sac_q_forward")
        return out


class Actor(_FlatModule):
    """Actor of the reference sac.py:46-78 (3 -> 256 -> 256 ReLU, mean head, tanh-bounded log-std head, tanh-squashed action)."""

    def __init__(self, env, device=None):
        super().__init__()
        obs_dim, act_dim = int(np.prod(env.observation_space.shape)), int(np.prod(env.action_space.shape))
        if obs_dim != 3 or act_dim != 1:
            raise N.MiError("the SAC kernels are specialised for Pendulum (obs 3, action 1); got %d/%d" % (obs_dim, act_dim))
        self.shared_net = nn.Sequential(nn.Linear(obs_dim, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU())
        self.mean_net = nn.Linear(256, act_dim)
        self.log_std_net = nn.Sequential(nn.Linear(256, act_dim), nn.Tanh())
        hi, lo = np.asarray(env.action_space.high, np.float32), np.asarray(env.action_space.low, np.float32)
        self.register_buffer("action_scale", torch.tensor((hi - lo) / 2.0, dtype=torch.float32))
        self.register_buffer("action_bias", torch.tensor((hi + lo) / 2.0, dtype=torch.float32))
        if float(self.action_scale) != 2.0 or float(self.action_bias) != 0.0:
            raise N.MiError("the SAC kernels are specialised for Pendulum's action range [-2, 2]")
        self._finish(env, device, N.SAC_ACTOR_NPARAMS)

    def get_action(self, observation, eps=None):
        """sac.py:65-78.  eps: the standard-normal draws of rsample() (default: torch.randn on the device)."""
        obs = observation.to(self.device, torch.float32)
        lead = obs.shape[:-1]
        obs = obs.reshape(-1, 3).contiguous()
        n = obs.shape[0]
        e = torch.randn(n, device=self.device) if eps is None else eps.to(self.device, torch.float32).reshape(-1).contiguous()
        action = torch.empty(n, dtype=torch.float32, device=self.device)
        logp = torch.empty(n, dtype=torch.float32, device=self.device)
        N.check(N.lib().mi_sac_actor_sample(N.ptr(self.flat), N.ptr(obs), N.ptr(e), n, N.ptr(action), N.ptr(logp), N.stream_ptr(self.device)),
                "mi_sac_actor_sample")
        return action.reshape(*lead, 1), logp.reshape(lead)


class DuelingQNetwork(_FlatModule):
    """QNetwork of the reference dueling_dqn.py:24-40 (features 4 -> 120 -> 84 ReLU, value stream 84 -> 1, advantage stream 84 -> n,
    values + (advantages - mean advantages)); attribute names as in the reference (incl. its `feauture_layer` spelling).
    `flat` holds the dueling parameters; `eff` the equivalent plain-DQN vector the kernels consume (refresh with `repack()`)."""

    def __init__(self, env, device=None):
        super().__init__()
        obs_dim = int(np.prod(env.observation_space.shape))
        if obs_dim != 4 or env.action_space.n != 2:
            raise N.MiError("the HIP kernels are specialised for CartPole (obs 4, actions 2)")
        self.feauture_layer = nn.Sequential(nn.Linear(obs_dim, 120), nn.ReLU(), nn.Linear(120, 84), nn.ReLU())
        self.value_stream = nn.Linear(84, 1)
        self.advantage_stream = nn.Linear(84, env.action_space.n)
        self._finish(env, device, N.DUELING_NPARAMS)
        self.eff = torch.empty(N.DQN_NPARAMS, dtype=torch.float32, device=self.device)
        self.repack()

    def repack(self):
        """refresh the plain-DQN image of the parameters (after an optimizer step / load_state_dict)"""
        N.check(N.lib().mi_dueling_pack(N.ptr(self.flat), N.ptr(self.eff), N.stream_ptr(self.device)), "mi_dueling_pack")

    def load_flat(self, vec):
        super().load_flat(vec)
        self.repack()

    def load_state_dict(self, state_dict, *a, **kw):
        super().load_state_dict(state_dict, *a, **kw)
        self.repack()

    def forward(self, observation):
        """dueling_dqn.py:36-40."""
        obs = observation.to(self.device, torch.float32)
        lead = obs.shape[:-1]
        obs = obs.reshape(-1, 4).contiguous()
        q = torch.empty((obs.shape[0], 2), dtype=torch.float32, device=self.device)
        N.check(N.lib().mi_dqn_forward(N.ptr(self.eff), N.ptr(obs), obs.shape[0], N.ptr(q), N.stream_ptr(self.device)), "mi_dqn_forward")
        return q.reshape(*lead, 2)
