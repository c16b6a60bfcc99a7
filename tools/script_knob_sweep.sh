#!/bin/bash
# Runs on the GPU box: the five drop-in scripts under a sweep of their environment knobs (env counts that are no multiple of anything, tiny and wrapped rings, odd batches,
# short budgets): every run must end with rc 0 and finite returns / losses.  Host-logic smoke, not a parity test.  usage: tools/script_knob_sweep.sh > gpurun_out/knobs.txt
run() { # name, env assignments...
  name=$1; shift
  out=$(env "$@" PRINT_EPISODES=0 timeout 120 python -m deep_rl_amd.$name 2>&1); rc=$?
  last=$(echo "$out" | tail -1 | cut -c1-160)
  nan=$(echo "$out" | grep -ci "nan\|Traceback\|Error")
  echo "$name [$*] rc=$rc bad_lines=$nan | $last"
}
for n in 1 3 7 16 33 250 1000; do run ppo NUM_ENVS=$n TOTAL_TIMESTEPS=$((n * 128 * 5)); done
run ppo NUM_ENVS=4096 TOTAL_TIMESTEPS=$((4096 * 128 * 3))
for cfg in "1 2000 64 200" "3 3000 7 100" "17 4000 128 2000" "100 20000 1 50" "250 30000 333 30" "4096 200000 128 16" "5 2500 1000 2501"; do
  set -- $cfg
  for s in dqn dueling_dqn per; do run $s NUM_ENVS=$1 TOTAL_TIMESTEPS=$2 BATCH_SIZE=$3 MEMORY_SIZE=$4 LEARNING_STARTS=$(( $2 / 10 )); done
done
for cfg in "1 1200 64 300" "3 900 7 50" "33 2000 256 1000" "200 3000 100 20" "2048 12288 256 8"; do
  set -- $cfg
  run sac NUM_ENVS=$1 TOTAL_TIMESTEPS=$2 BATCH_SIZE=$3 MEMORY_SIZE=$4 LEARNING_STARTS=$(( $2 / 6 ))
done
