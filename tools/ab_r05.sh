#!/bin/bash
# same-box A/B of the ROUND-5 TREE (ab_r05/, its own bench.py + library: another ABI, so MIRL_SO cannot do it) against the current tree: headline + sharded-route legs.
# In the build container first (ab_r05/ is git-ignored, but travels with gpurun):
#   mkdir ab_r05 && git archive adaa180 | tar -x -C ab_r05 && rm -rf ab_r05/profiles ab_r05/tests/golden ab_r05/docs && make -C ab_r05/deep_rl_amd/csrc
# then on the GPU box: tools/ab_r05.sh   (profiles/r06d_ab_against_r05_tree.txt)
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
