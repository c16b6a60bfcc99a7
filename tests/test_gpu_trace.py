"""f4: a device run dumped in the golden fixtures' key layout (deep_rl_amd/trace.py) and diffed KEY BY KEY against the fixture captured from the
unmodified reference (oracle/capture_ppo_trace.py:126-157).  PPO: the whole reference run (156 updates, 2,496 optimizer steps, 19,968 env steps) at N = 1 with the drop-in's
hyper-parameters under golden-forced randomness (the reference's actions, reset noise and minibatch indices).  DQN / SAC: the reference's first acting steps teacher-forced,
then production training, dumped and checked for layout + content."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = 128


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _golden(name):
    with np.load(os.path.join(ROOT, "tests", "golden", name)) as z:
        return {k: z[k] for k in z.files}


def _ulp_close(a, b, max_ulp=1):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (np.abs(a - b) <= max_ulp * np.spacing(np.abs(np.asarray(b, np.float32))).astype(np.float64)).all()


def test_ppo_trace_dump_matches_the_reference_fixture_key_by_key(dev, tmp_path):
    import deep_rl_amd as D
    from deep_rl_amd.trace import PPOTrace

    g = _golden("ppo_ref_trace.npz")
    # the drop-in's own setup (deep_rl_amd/ppo.py, reference ppo.py:62-90) at num_envs = 1
    total_timesteps, num_steps, learning_rate, seed = 20_000, 128, 2.5e-4, 1
    num_updates = total_timesteps // num_steps
    env = D.make("CartPole-v1", num_envs=1, device=dev)
    env.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    agent = D.ActorCritic(env)
    agent.load_flat(g["init_params"])      # (the CPU QR of the orthogonal init rounds differently per host: a9 is pinned separately in test_gpu_init.py)
    optimizer = D.ClipAdam(agent, lr=learning_rate, eps=1e-5, max_grad_norm=0.5)
    engine = D.PPOEngine(env, agent, optimizer, num_steps=num_steps, n_minibatch=4, update_epochs=4, gamma=0.99, gae_lambda=0.95, clip_coef=0.2, ent_coef=0.01,
                         vf_coef=0.5, max_episodes_logged=4 * num_steps)
    tr = PPOTrace(engine, {"total_timesteps": total_timesteps, "num_updates": num_updates, "learning_rate": learning_rate, "seed": seed})
    tr.reset(torch.from_numpy(g["reset_states"][:1]))
    acts = torch.from_numpy(g["actions_all"].astype(np.int64)).reshape(-1, T, 1)
    ar, resets = g["after_reset_all"], g["reset_states"]
    ri = 1
    for u in range(num_updates):
        fr = np.zeros((T, 1, 4))
        for t in range(T):
            s = u * T + t
            if s + 1 < len(ar) and ar[s + 1]:
                fr[t, 0] = resets[ri]; ri += 1
        optimizer.param_groups[0]["lr"] = (1.0 - u / num_updates) * learning_rate
        tr.update(forced_actions=acts[u], forced_resets=torch.from_numpy(fr), mb_inds=g["mb_inds"][16 * u:16 * (u + 1)])
    path = tr.save(tmp_path / "device_ppo_trace")
    d = _golden_path(path)

    # ---- the diff, key by key ----
    assert set(d) == set(g), set(d) ^ set(g)
    for k in g:
        assert d[k].shape == g[k].shape and d[k].dtype == g[k].dtype, (k, d[k].shape, g[k].shape, d[k].dtype, g[k].dtype)
    exact = ["hparams", "init_params", "actions_all", "terminated_all", "after_reset_all", "mb_inds", "episode_global_step", "episode_return", "final_global_step"]
    for k in exact:
        assert np.array_equal(d[k], g[k]), k
    assert np.array_equal(d["reset_states"], g["reset_states"])                # the forced reset noise comes back as given, one per episode + the initial one
    # float64 dynamics: the device evaluates fdlibm's sin / cos kernels, gym calls libm — the states agree to a few 1e-16 relative, the float32 observations to 1 ulp
    assert np.abs(d["state_first"] - g["state_first"]).max() < 1e-12
    assert _ulp_close(d["obs_all"], g["obs_all"]) and (d["obs_all"] != g["obs_all"]).sum() <= 40
    # the tolerances of test_whole_reference_run_replayed_on_device (2,496 chained Adam steps)
    ot, og = d["opt_terms"], g["opt_terms"]
    assert (np.abs(ot[:, :4] - og[:, :4]) / np.maximum(np.abs(og[:, :4]), 1e-2)).max() < 2e-3
    assert np.array_equal(ot[:, 4], og[:, 4])                                  # the annealed learning rate, to the bit
    assert np.abs(ot[:, 5:] - og[:, 5:]).max() < 2e-2                          # sum(params), sum|params| after every step (9,155 terms, each within 2e-4 at the end)
    ne = np.abs(d["clip_norm"] - g["clip_norm"]) / g["clip_norm"]
    assert np.median(ne) < 2e-5 and (ne > 2e-3).sum() <= 8 and ne.max() < 0.2
    assert np.abs(d["full_grads"] - g["full_grads"]).max() <= 3e-6 * np.abs(g["full_grads"]).max()      # the first 16 pre-clip gradients vs torch autograd
    assert np.abs(d["full_params"] - g["full_params"]).max() < 1e-6
    assert np.abs(d["final_params"] - g["final_params"]).max() < 2e-4
    assert abs(d["final_explained_var"][0] - g["final_explained_var"][0]) < 2e-2 * abs(g["final_explained_var"][0])
    us_d, us_g = d["update_sums"], g["update_sums"]
    assert np.array_equal(us_d[:, [2, 4, 5]], us_g[:, [2, 4, 5]])              # sums of actions, rewards, dones: integers
    assert np.abs(us_d - us_g).max() < 2e-2 * max(1.0, np.abs(us_g).max())
    for i in range(3):
        for nm in ("actions", "rewards", "dones"):
            assert np.array_equal(d["upd%d_%s" % (i, nm)], g["upd%d_%s" % (i, nm)]), (i, nm)
        assert _ulp_close(d["upd%d_observations" % i], g["upd%d_observations" % i])
        tol = 3e-6 if i == 0 else 2e-4                                          # update 0 runs on identical parameters; 1 and 2 after 16 / 32 chained steps
        for nm in ("values", "log_probs", "advantages", "returns"):
            assert np.abs(d["upd%d_%s" % (i, nm)] - g["upd%d_%s" % (i, nm)]).max() < tol * max(1.0, np.abs(g["upd%d_%s" % (i, nm)]).max()), (i, nm)
        assert np.abs(d["upd%d_params_before" % i] - g["upd%d_params_before" % i]).max() < 1e-5


def _golden_path(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def test_dqn_trace_dump_layout_and_teacher_forced_prefix(dev, tmp_path):
    """The reference dqn.py run's first 2,000 steps teacher-forced (its actions and reset noise), then 40 production training steps, dumped: every key the recorder
    writes exists in the fixture with the same dtype / rank, the env-side keys equal the fixture's prefix, the shadow walk's self-check held on every launch."""
    import deep_rl_amd as D
    from deep_rl_amd.trace import DQNTrace

    g = _golden("dqn_ref_trace.npz")
    env = D.make("CartPole-v1", num_envs=1, device=dev)
    env.seed(1); torch.manual_seed(1)
    q = D.QNetwork(env); t = D.QNetwork(env)
    q.load_flat(g["init_params"]); t.load_state_dict(q.state_dict())
    eng = D.DQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=100_001, batch_size=128, learning_starts=10_000, total_timesteps=100_000, max_episodes_logged=64)
    tr = DQNTrace(eng, {"train_frequency": 10, "learning_rate": 2.5e-4, "target_network_frequency": 500, "seed": 1}, checkpoints=(3,))
    tr.reset(torch.from_numpy(g["reset_states"][:1]))
    acts, ar, resets = g["actions_all"].astype(np.int64), g["after_reset_all"], g["reset_states"]
    ri, gs, n_steps = 1, 0, 2000
    while gs < n_steps:
        fr = np.zeros((10, 1, 4))
        for s in range(10):
            if ar[gs + s + 1]:
                fr[s, 0] = resets[ri]; ri += 1
        tr.act(10, forced_actions=torch.from_numpy(acts[gs:gs + 10].reshape(10, 1)), forced_resets=torch.from_numpy(fr))
        gs += 10
    for _ in range(40):
        tr.train_step()
    d = _golden_path(tr.save(tmp_path / "device_dqn_trace"))
    assert set(d) <= set(g), set(d) - set(g)
    for k in d:
        assert d[k].dtype == g[k].dtype and d[k].ndim == g[k].ndim, (k, d[k].dtype, g[k].dtype)
    assert np.array_equal(d["hparams"], g["hparams"]) and np.array_equal(d["init_params"], g["init_params"])
    for k in ("actions_all", "terminated_all", "after_reset_all"):
        assert np.array_equal(d[k], g[k][:n_steps]), k
    assert np.array_equal(d["reset_states"], g["reset_states"][:len(d["reset_states"])]) and len(d["reset_states"]) == ri
    assert _ulp_close(d["obs_first"], g["obs_first"][:n_steps])
    assert np.allclose(d["obs_block_sums"], g["obs_block_sums"][:2], rtol=1e-5, atol=1e-4)
    n_ep = int((g["episode_global_step"] <= n_steps).sum())
    assert np.array_equal(d["episode_global_step"], g["episode_global_step"][:n_ep]) and np.array_equal(d["episode_return"], g["episode_return"][:n_ep])
    assert len(d["loss_all"]) == 40 and np.isfinite(d["loss_all"]).all() and d["batch_inds_first"].shape == (40, 128) and d["full_grads"].shape == (8, 10934)
    assert d["ck_update"].tolist() == [3] and np.array_equal(d["ck_params"][0], d["full_params"][2])     # parameters BEFORE update 3 = after update 2


def test_sac_trace_dump_layout(dev, tmp_path):
    """The reference sac.py run's first 300 Pendulum steps teacher-forced, then 30 production iterations, dumped: layout vs the fixture, env-side keys vs its prefix."""
    import deep_rl_amd as D
    from deep_rl_amd.trace import SACTrace

    g = _golden("sac_ref_trace.npz")
    env = D.make("Pendulum-v1", num_envs=1, device=dev)
    env.seed(1); torch.manual_seed(1)
    actor = D.Actor(env)
    qs = [D.SoftQNetwork(env) for _ in range(4)]
    actor.load_flat(g["init_actor"])
    eng = D.SACEngine(env, actor, *qs, slots=30_001, batch_size=256, learning_starts=300, max_episodes_logged=8)
    eng.q_flat.copy_(torch.from_numpy(g["init_q"]).to(dev)); eng.qt_flat.copy_(eng.q_flat)
    tr = SACTrace(eng, {"total_timesteps": 30_000, "policy_frequency": 2, "target_network_frequency": 1, "policy_lr": 3e-4, "q_lr": 1e-3, "seed": 1})
    tr.reset(torch.from_numpy(g["reset_states"][:1]))
    ar, resets = g["after_reset_all"], g["reset_states"]
    ri = 1
    for gs in range(300):
        fr = None
        if ar[gs + 1]:
            fr = torch.from_numpy(resets[ri:ri + 1]); ri += 1
        tr.act(forced_actions=torch.from_numpy(g["actions_all"][gs:gs + 1]), forced_resets=fr)
    for _ in range(30):
        tr.act()
        tr.train_step()
    d = _golden_path(tr.save(tmp_path / "device_sac_trace"))
    assert set(d) <= set(g), set(d) - set(g)
    for k in d:
        assert d[k].dtype == g[k].dtype and d[k].ndim == g[k].ndim, (k, d[k].dtype, g[k].dtype)
    assert np.array_equal(d["init_actor"], g["init_actor"]) and np.array_equal(d["init_q"], g["init_q"])
    assert np.array_equal(d["actions_all"][:300], g["actions_all"][:300]) and np.array_equal(d["after_reset_all"][:300], g["after_reset_all"][:300])
    assert _ulp_close(d["obs_first"][:300], g["obs_first"][:300], max_ulp=2)            # numpy's SIMD sin / cos vs the fdlibm kernels: 2 float32 ulps
    assert np.abs(d["rewards_all"][:300] - g["rewards_all"][:300]).max() < 2e-6 * np.abs(g["rewards_all"][:300]).max()
    assert d["q_losses"].shape == (30, 4) and d["actor_losses"].shape[1] == 4 and d["alpha_steps"].shape[1] == 5 and np.isfinite(d["q_losses"]).all()
    assert d["actor_losses"].shape[0] == d["alpha_steps"].shape[0] == 30            # policy_frequency 2: two actor + alpha updates every second step
