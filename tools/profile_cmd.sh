#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of an arbitrary python tool, CSV summary to gpurun_out/<tag>_kernel_stats.csv
# usage: tools/profile_cmd.sh <tag> <script relative to repo> [args...]
TAG=$1; shift
REPO=$(pwd)
SCRIPT=$REPO/$1; shift
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o stats -- python3 $SCRIPT "$@" > $REPO/gpurun_out/${TAG}_run.log 2>&1
find /tmp/rp_$TAG -name '*kernel_stats.csv' -exec cp {} $REPO/gpurun_out/${TAG}_kernel_stats.csv \;
cat $REPO/gpurun_out/${TAG}_kernel_stats.csv | head -${LINES_SHOWN:-30}
