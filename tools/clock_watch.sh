#!/bin/bash
# tools/clock_watch.sh <out> <cmd...>: sample rocm-smi clocks/power every 0.2 s while <cmd> runs (is the run power-capped?)
out=$1; shift
( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.2; done ) > $out &
W=$!
"$@"
rc=$?
kill $W
exit $rc
