#!/bin/bash
# usage: tools/profile_ab.sh <regex> <script> -- <so1> <so2> ...   : per-variant rocprofv3 kernel stats of the kernels matching the regex
PAT=$1; SCRIPT=$2; shift 3
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  export MIRL_SO=$REPO/$so
  rm -rf /tmp/rp_ab
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_ab -o stats -- python3 $REPO/$SCRIPT > /dev/null 2>&1
  echo "== $so"
  find /tmp/rp_ab -name '*kernel_stats.csv' -exec cat {} \; | python3 -c "
import csv, sys, re
for row in csv.reader(sys.stdin):
    if re.search(r'$PAT', row[0]): print('   %-36s calls %5s avg %8.2f us  min %8.2f  max %8.2f' % (row[0][:36], row[1], float(row[3]) / 1e3, float(row[5]) / 1e3, float(row[6]) / 1e3))
"
done
