#!/bin/bash
# A/B several builds of libmirl (same ABI) with the split-bf16 gradient variant switched on: short bench.py per build, the variant's numbers only.
# usage: tools/ab_variant.sh <steps> <so1> <so2> ...   (paths relative to the repo root)
STEPS=$1; shift
for so in "$@"; do
  export MIRL_SO=$(pwd)/$so
  b=$(timeout 300 python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); v = d['variant_bf16x3']
print('f32 %.1fM steps/s %.3f ms grad %.1f us | bf16x3 %.1fM steps/s %.3f ms grad %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_us'], v['value']/1e6, v['ms_per_step'], v['grad_kernel_avg_launch_us']))")
  echo "$so | $b"
done
