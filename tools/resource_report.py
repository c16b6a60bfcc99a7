#!/usr/bin/env python3
"""Print per-kernel register / LDS / occupancy usage of the HIP sources (hipcc -Rpass-analysis)."""
import re, subprocess, sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "deep_rl_amd", "csrc")
files = sys.argv[1:] or ["mi_env.hip", "mi_rollout.hip", "mi_update.hip", "mi_dqn.hip", "mi_sac.hip"]
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
                          "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(SRC, f), "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    cur = None
    for ln in out.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", ln)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()[:60]}
            continue
        m = re.search(r"remark: +([A-Za-z /\[\]]+): (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
            if m.group(1).startswith("LDS Size"):
                print("%-62s VGPR %3d AGPR %3d SGPR %3d spillV %3d spillS %3d scratch %4d occ %d LDS %6d" % (
                    cur["name"], cur.get("VGPRs", -1), cur.get("AGPRs", -1), cur.get("TotalSGPRs", -1), cur.get("VGPRs Spill", -1),
                    cur.get("SGPRs Spill", -1), cur.get("ScratchSize [bytes/lane]", -1), cur.get("Occupancy [waves/SIMD]", -1),
                    cur["LDS Size [bytes/block]"]))
                cur = None
