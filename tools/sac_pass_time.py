#!/usr/bin/env python3
"""Launches SoftQNetwork forwards (one 256x256 MFMA pass + thin layers) at ONE batch size (argv[1]) — run under rocprofv3 (tools/profile_ab.sh)
with MIRL_SO pointing at experimental builds (-DSAC_EXP=1: weights loaded once; -DSAC_EXP=2: loads only, no MFMA)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D
dev = torch.device("cuda", 0)
env = D.make("Pendulum-v1", num_envs=1, device=dev)
torch.manual_seed(1)
q = D.SoftQNetwork(env)
for n in [int(a) for a in sys.argv[1:]] or [256]:
    obs = torch.randn(n, 3, device=dev); act = torch.randn(n, 1, device=dev)
    for _ in range(30): q(obs, act)
    torch.cuda.synchronize()
