#!/bin/bash
# Usage (on the GPU box): tools/prof_bench.sh <name> [bench args...] -> gpurun_out/prof_<name>/ + the top kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
OUT=$R/gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $NAME -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv under $OUT"); raise SystemExit(0)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:26]:
    print("%-78s %5s %10s %6s" % (r["Name"][:78], r["Calls"], r["AverageNs"], r["Percentage"]))
PY
tail -c 300 $OUT/bench.log
