#!/bin/bash
# A/B several builds of libmirl (same ABI) on the off-policy loops: tools/bench_dqn.py per build and variant, REPS rounds interleaved.
# usage: tools/ab_dqn.sh <reps> "<variants: dqn dueling per>" <so1> <so2> ...   (paths relative to the repo root)
REPS=$1; VARS=$2; shift; shift
for r in $(seq 1 $REPS); do
  for v in $VARS; do
    for so in "$@"; do
      export MIRL_SO=$(pwd)/$so
      b=$(timeout 300 python tools/bench_dqn.py --variant $v 2>/dev/null | grep '^{"variant"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%.2f us/iteration  (enqueue %.1f)  loss %.10g' % (d['us_per_iteration'], d['host_enqueue_us_per_iteration'], d['loss']))")
      echo "round $r | $v | $so | $b"
    done
  done
done
