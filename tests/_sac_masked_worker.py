"""Worker of tests/test_gpu_sac_robust.py: a fresh process (its CU mask, if any, was put in the environment BEFORE it initialises the GPU) trains SAC for a few
iterations and prints a digest of the final state plus what the library decided.  Every form of the update kernels (single workgroup, sibling roles, owed alpha
step) is bit-identical, so the digests of differently masked processes must agree."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import deep_rl_amd as D  # noqa: E402
from deep_rl_amd import _native as N  # noqa: E402

batch = int(os.environ.get("MIRL_TEST_BATCH", "256"))
dev = torch.device("cuda", 0)
env = D.make("Pendulum-v1", num_envs=64, device=dev, seed=3)
torch.manual_seed(3)
actor = D.Actor(env)
qs = [D.SoftQNetwork(env) for _ in range(4)]
qs[2].load_state_dict(qs[0].state_dict()); qs[3].load_state_dict(qs[1].state_dict())
eng = D.SACEngine(env, actor, *qs, slots=64, batch_size=batch, learning_starts=4)
eng.reset()
owed = 0
for _ in range(24):
    eng.act()
    if eng.global_step > 6:
        eng.train_step()
        owed += eng._owed is not None
eng.check(wait=True)
h = hashlib.sha256()
for t in (eng.actor.flat, eng.q_flat, eng.qt_flat, eng.log_alpha, eng._alpha_m, eng._alpha_v, eng.actor_optimizer.exp_avg, eng.q_optimizer.exp_avg_sq):
    h.update(t.cpu().numpy().tobytes())
finite = bool(torch.isfinite(eng.q_flat).all() and torch.isfinite(eng.actor.flat).all())
print("SAC_WORKER " + json.dumps({"digest": h.hexdigest(), "usable_cus": N.lib().mi_sac_usable_cus(), "owed_fits": bool(eng._owed_fits), "owed_seen": owed, "finite": finite,
                                  "device_cus": torch.cuda.get_device_properties(0).multi_processor_count}))
