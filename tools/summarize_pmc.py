#!/usr/bin/env python3
"""Condense rocprofv3 CSVs (kernel trace + --pmc counter_collection passes) into per-kernel averages (JSON)."""
import csv, json, sys, collections

def short(name):
    for k in ("sac_dw2_adam_kernel", "sac_dw2_gemm_kernel", "sac_grad_reduce_kernel", "sac_critic_kernel", "sac_actor_kernel", "sac_act_kernel",   # (before "grad_reduce_kernel": substring)
              "grad_kernel_bx", "grad_kernel_f32", "grad_kernel", "grad_reduce_kernel", "clip_adam_kernel", "rollout_q4_kernel", "rollout_mfma_kernel", "rollout_kernel", "gae_kernel",
              "adv_stats_kernel", "perm_stats_kernel", "perm_kernel", "dqn_act4_kernel", "dqn_act_kernel", "dqn_td_kernel", "dqn_reduce2_kernel", "dqn_reduce_kernel"):
        if k in name:
            return k
    return name[:40]

out = {}
trace = sys.argv[1]
dur = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(trace)):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta = out.setdefault(short(r["Kernel_Name"]), {})
        for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
            if k in r:
                meta[k] = r[k]
except Exception as e:
    print("trace: %r" % e, file=sys.stderr)
for k, v in dur.items():
    out.setdefault(k, {}).update({"launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3})
for f in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as e:
        print("%s: %r" % (f, e), file=sys.stderr)
    for k, ctrs in acc.items():
        for c, v in ctrs.items():
            out.setdefault(k, {})["pmc_" + c + "_avg"] = sum(v) / len(v)
for k, m in out.items():
    # HBM bytes per launch, gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB units;
    # FETCH_SIZE reports exactly half of a wide coalesced read stream -> doubled value given beside the raw one.
    if "pmc_FETCH_SIZE_avg" in m:
        m["hbm_read_bytes_raw"] = m["pmc_FETCH_SIZE_avg"] * 1024
        m["hbm_read_bytes_x2_wide_stream_correction"] = 2 * m["pmc_FETCH_SIZE_avg"] * 1024
    if "pmc_WRITE_SIZE_avg" in m:
        m["hbm_write_bytes"] = m["pmc_WRITE_SIZE_avg"] * 1024
print(json.dumps(out, indent=1, sort_keys=True))
