#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the SAME bench command,
# and condenses them into small CSV/JSON summaries under gpurun_out/prof_<tag>/ (copy what you want judged to profiles/).
TAG=${1:-r01}
STEPS=${2:-10}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -o stats -- $CMD > $OUT/stats_run.log 2>&1
find /tmp/rp_stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find /tmp/rp_stats -name '*kernel_trace.csv' -exec cp {} /tmp/kernel_trace.csv \;
PCMD="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only"
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/rp_pmc$i -o pmc -- $PCMD > $OUT/pmc${i}_run.log 2>&1
  find /tmp/rp_pmc$i -name '*counter_collection.csv' -exec cp {} /tmp/pmc$i.csv \;
done
python3 $REPO/tools/summarize_pmc.py /tmp/kernel_trace.csv /tmp/pmc*.csv > $OUT/pmc_summary.json 2> $OUT/pmc_summary.err
# the same counter passes with the split-bf16 gradient variant switched on (grad_kernel_bx; experiment)
export MIRL_PPO_CONTRACTION=bf16x3
i=0
for CTRS in "FETCH_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/rp_bx$i -o pmc -- $PCMD > $OUT/bx_pmc${i}_run.log 2>&1
  find /tmp/rp_bx$i -name '*counter_collection.csv' -exec cp {} /tmp/bx_pmc$i.csv \;
  find /tmp/rp_bx$i -name '*kernel_trace.csv' -exec cp {} /tmp/bx_trace.csv \;
done
unset MIRL_PPO_CONTRACTION
python3 $REPO/tools/summarize_pmc.py /tmp/bx_trace.csv /tmp/bx_pmc*.csv > $OUT/pmc_summary_bf16x3.json 2>> $OUT/pmc_summary.err
# BASELINE configs 3 / 4: kernel stats + counter passes of tools/bench_dqn.py / bench_sac.py at the reference's batch and at the scaled batch 4,096.
# The program stands directly behind `--`, counters in passes of their own (never together with a trace domain beyond --kernel-trace).
for W in "dqn_b128 bench_dqn.py --batch 128 --iters 120" "dqn_b4096 bench_dqn.py --batch 4096 --iters 120" "sac_b256 bench_sac.py --batch 256 --iters 120" "sac_b4096 bench_sac.py --batch 4096 --iters 120"; do
  set -- $W
  NAME=$1; shift
  WCMD="python3 $REPO/tools/$*"
  rm -rf /tmp/rp_${NAME}_*
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_${NAME}_stats -o stats -- $WCMD > $OUT/${NAME}_stats_run.log 2>&1
  find /tmp/rp_${NAME}_stats -name '*kernel_stats.csv' -exec cp {} $OUT/${NAME}_kernel_stats.csv \;
  find /tmp/rp_${NAME}_stats -name '*kernel_trace.csv' -exec cp {} /tmp/${NAME}_trace.csv \;
  i=0
  for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/rp_${NAME}_pmc$i -o pmc -- $WCMD > $OUT/${NAME}_pmc${i}_run.log 2>&1
    find /tmp/rp_${NAME}_pmc$i -name '*counter_collection.csv' -exec cp {} /tmp/${NAME}_pmc$i.csv \;
  done
  python3 $REPO/tools/summarize_pmc.py /tmp/${NAME}_trace.csv /tmp/${NAME}_pmc*.csv > $OUT/pmc_summary_${NAME}.json 2>> $OUT/pmc_summary.err
done
python3 $REPO/tools/make_latest_pmc.py $TAG $OUT > $OUT/latest_pmc.json 2>> $OUT/pmc_summary.err
ls -la $OUT
