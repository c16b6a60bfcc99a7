#!/bin/bash
# A/B of an engine switch (environment variable) on the config-4 loop with ONE build: tools/bench_sac.py per setting and batch, REPS rounds interleaved.
# usage: tools/ab_sac_env.sh <reps> <VAR> <value1> <value2> ...     e.g.  tools/ab_sac_env.sh 3 MIRL_SAC_TRANSPOSED 0 1
REPS=$1; VAR=$2; shift; shift
for r in $(seq 1 $REPS); do
  for b in 256 4096; do
    for v in "$@"; do
      l=$(env $VAR=$v python tools/bench_sac.py --batch $b --iters 400 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('%.1f us per iteration, q_losses %s' % (d['us_per_iteration'], d['q_losses']))")
      echo "round $r | batch $b | $VAR=$v | $l"
    done
  done
done
