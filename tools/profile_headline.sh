#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the HEADLINE command only, condensed into gpurun_out/<tag>_kernel_stats.csv and
# gpurun_out/<tag>_kernel_durations.json (tools/make_latest_durations.py: per-kernel device time, span and launch gaps of one PPO update, from ONE profiled run).
# Copy the two into profiles/ (the JSON also as profiles/latest_kernel_durations.json: bench.py quotes it as `kernel_device_ms_per_update`).
TAG=${1:-r05}
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$TAG
export MIRL_BENCH_SHARDED_LEG=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o stats -- python3 $REPO/bench.py --headline-only --no-cpu-baseline --single-window --steps 40 --warmup 5 > $REPO/gpurun_out/${TAG}_headline_run.log 2>&1
find /tmp/rp_$TAG -name '*kernel_stats.csv' -exec cp {} $REPO/gpurun_out/${TAG}_kernel_stats.csv \;
find /tmp/rp_$TAG -name '*kernel_trace.csv' -exec cp {} /tmp/${TAG}_kernel_trace.csv \;
python3 $REPO/tools/make_latest_durations.py $TAG /tmp/${TAG}_kernel_trace.csv > $REPO/gpurun_out/${TAG}_kernel_durations.json
cat $REPO/gpurun_out/${TAG}_kernel_durations.json
tail -2 $REPO/gpurun_out/${TAG}_headline_run.log | cut -c1-600
