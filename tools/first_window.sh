#!/bin/bash
# How often is the FIRST timed window of `bench.py --steps 20 --warmup 5` (what the driver measures) slower than the two behind it?  N runs, one line each.
# usage: tools/first_window.sh <runs> [extra bench.py flags]
N=$1; shift
for r in $(seq 1 $N); do
  timeout 300 python ${BENCH:-bench.py} --steps 20 --warmup 5 --no-cpu-baseline --headline-only "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
w = d['timed_windows']['ms_per_step']
print('run $r: %s  first / median of the others = %.4f' % (w, w[0] / sorted(w[1:])[len(w[1:]) // 2]))"
done
