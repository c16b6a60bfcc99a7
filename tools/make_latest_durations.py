#!/usr/bin/env python3
"""profiles/latest_kernel_durations.json from the rocprofv3 kernel TRACE of the headline command:
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o stats -- python3 bench.py --headline-only --no-cpu-baseline --steps 20 --warmup 5
usage: make_latest_durations.py <build tag> <..._kernel_trace.csv>   (prints the JSON)

One PPO outer update = the launches from one rollout_q4_kernel to the next (mi_ppo_update's fixed sequence: rollout, permutations / statistics, 16 x {gradient, slab
sum}, clip + Adam).  Over the steady-state updates of the trace (the first 10 and the last one dropped): per kernel the DEVICE time per update (sum of End - Start of its
launches), the update's span (rollout start -> next rollout start) and launch_gaps = span - sum, all from the SAME profiled run, so that they add up — which HIP-event
brackets cannot give (each pair costs the loop ~3 us).  bench.py quotes this file as `kernel_device_ms_per_update` beside its own live window (profiling itself slows the
loop a little: compare span_ms with the bench line's ms_per_step)."""
import csv, json, os, sys, collections

tag, path = sys.argv[1], sys.argv[2]
names = ("rollout_q4_kernel", "perm_stats_kernel", "grad_kernel_f32", "grad_kernel_bx", "grad_reduce_kernel", "clip_adam_kernel")


def short(n):
    for k in names:
        if k in n.split("(")[0]:
            return k
    return None


def source_id():
    """mi_source_id() of the library that was profiled (this script runs right behind the profiled command, on the same box, with the same libmirl.so)."""
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from deep_rl_amd import _native as N
        return N.lib().mi_source_id().decode()
    except Exception as ex:  # noqa: BLE001
        return "unknown (%s)" % type(ex).__name__


rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(path))), key=lambda x: x[0])
starts = [i for i, r in enumerate(rows) if r[2] == "rollout_q4_kernel"]
updates = []
for a, b in zip(starts[:-1], starts[1:]):
    seq = rows[a:b]
    kinds = collections.Counter(r[2] for r in seq)
    if kinds.get("grad_kernel_f32") != 16 or kinds.get("grad_reduce_kernel") != 16 or None in kinds or len(seq) != 35:   # only plain f32 updates with nothing else in between
        continue
    per = collections.defaultdict(float)
    for s, e, k in seq:
        per[k] += (e - s) / 1e6
    updates.append((rows[b][0] - rows[a][0], per))
updates = updates[10:-1]
n = len(updates)
res = {}
for k in names:
    v = [u[1].get(k, 0.0) for u in updates]
    if n and sum(v) > 0:
        cnt = {"rollout_q4_kernel": 1, "perm_stats_kernel": 1, "clip_adam_kernel": 1}.get(k, 16)
        res[k] = {"ms_per_update": round(sum(v) / n, 5), "launches_per_update": cnt, "avg_launch_us": round(1e3 * sum(v) / n / cnt, 3)}
span = sum(u[0] for u in updates) / max(n, 1) / 1e6
total = sum(v["ms_per_update"] for v in res.values())
print(json.dumps({"build": tag, "source_id": source_id(), "source": "profiles/%s_kernel_durations.json (rocprofv3 --kernel-trace of `bench.py --headline-only`)" % tag, "updates_averaged": n, "ppo_update": res,
                  "span_ms": round(span, 5), "sum_ms": round(total, 5), "launch_gaps_ms": round(span - total, 5), "launches_per_update": 35}, indent=1, sort_keys=True))
