#!/bin/bash
# Counter passes of one bench frame, one rocprofv3 run per pass, summarised per kernel.
# usage (on the GPU box, from the repo root): tools/pmc_pass.sh <tag> "<bench.py arguments>" "<counters of pass 1>" ["<counters of pass 2>" ...]
# (limits per pass: 8 SQ_, 4 TCC_, 4 TCP_, 2 TA_, 2 TD_, 2 GRBM_ counters)
set -u
TAG=$1; ARGS="$2 --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-pmc"; shift 2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -o pmc -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
done
cd $R && python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
echo "pmc passes done: $OUT/pmc_summary.txt"
