#!/bin/bash
# same-box A/B of the round-5 tree (ab_r05/, its own bench.py + library) against the current tree: headline + sharded-route legs
for r in 1 2 3 4 5; do
  for t in ab_r05 .; do
    b=$(cd $t && MIRL_BENCH_SHARDED_LEG=1 PYTHONPATH=$(pwd) timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['sharded_route']
print('%s ms/upd  grad %.2f us | base %.4f  assume %+.2f  p2p %s  rccl1 %+.2f us/step (grad %.2f)' % (d['timed_windows']['ms_per_step'], d['roofline']['avg_launch_us'], s['baseline']['ms_per_step'],
      s['assume_sharded']['delta_us_per_optimizer_step'], [s['p2p_synthetic']['world%d' % w]['delta_us_per_optimizer_step'] for w in (2, 4, 8)], s['rccl_world1']['delta_us_per_optimizer_step'], s['rccl_world1']['grad_kernel_avg_launch_us']))")
    echo "round $r | $t | $b"
  done
done
