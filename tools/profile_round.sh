#!/bin/bash
# Round profile: the default bench (its JSON line and the counters its own rocprofv3 --pmc passes collected) and the
# rocprofv3 kernel trace + stats of the same command.
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>      -> gpurun_out/profile_<tag>/
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd $R && timeout -k 10 500 python3 -u bench.py --dump-pmc $OUT/pmc.json > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- python3 $R/bench.py --no-pmc --no-cpu-baseline > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
cp $OUT/trace/k_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
echo profile done
