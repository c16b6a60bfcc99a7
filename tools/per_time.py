import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
eng = D.PERDQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=256, batch_size=128, learning_starts=100, total_timesteps=100000)
eng.reset()
for _ in range(40): eng.act(10); eng.train_step()
stored = min(eng.global_step, eng.slots) * eng.N
def run(batch, sample):
    tm = N.Timer(); s = N.stream_ptr(dev)
    for _ in range(5):
        N.check(N.lib().mi_per_sample_current(1, 7, N.ptr(eng.priorities), stored, eng.slots * eng.N, float(stored), 0.6, 0.5, batch, sample, N.ptr(eng._per_ws), N.ptr(eng.batch_inds), N.ptr(eng.weights), s))
    torch.cuda.synchronize(); tm.start(s)
    for _ in range(50):
        N.check(N.lib().mi_per_sample_current(1, 7, N.ptr(eng.priorities), stored, eng.slots * eng.N, float(stored), 0.6, 0.5, batch, sample, N.ptr(eng._per_ws), N.ptr(eng.batch_inds), N.ptr(eng.weights), s))
    tm.stop(s); return 1e3 * tm.elapsed_ms() / 50
for b, smp in ((1, 0), (1, 1), (64, 1), (128, 1), (128, 0)):
    print("batch %d sample %d: %.1f us per call (back to back)" % (b, smp, run(b, smp)))
