#!/bin/bash
# A/B several builds of libmirl (same ABI) in ONE process-per-variant loop on the GPU box: parity smoke + short bench.
# usage: tools/ab_bench.sh <steps> <so1> <so2> ...   (paths relative to repo root)
STEPS=$1; shift
for so in "$@"; do
  export MIRL_SO=$(pwd)/$so
  r=$(timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1)
  b=$(timeout 300 python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%.1fM steps/s  %.2f ms/upd  grad %.1f us (%.1f%% mfma)  %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_us'], 100*d['roofline']['frac'], {k: v for k, v in d['kernel_ms_per_update_bracketed'].items() if k != 'note'}))")
  echo "$so | $r | $b"
done
