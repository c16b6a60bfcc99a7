#!/usr/bin/env python3
"""Phase durations of one SoftQNetwork forward (libmirl built with -DSAC_STAMPS): MIRL_SO=deep_rl_amd/libmirl_stamps.so python tools/sac_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D
dev = torch.device("cuda", 0)
env = D.make("Pendulum-v1", num_envs=1, device=dev)
torch.manual_seed(1)
q = D.SoftQNetwork(env)
for n in (256, 4096):
    obs = torch.randn(n, 3, device=dev); act = torch.randn(n, 1, device=dev)
    for _ in range(5):
        out = q(obs, act)
    torch.cuda.synchronize()
    print("n=%d" % n, "load x / layer1 / mfma pass (first touch) / bias+head / combine / mfma pass again (us):", [round(v / 100, 2) for v in out[:6].tolist()])
