#!/bin/bash
# Kernel trace and HBM-traffic counters of the stress scene (55.8 M triangles, tree of 3.3 GB: does not fit the caches).
# usage (on the GPU box, from the repo root): tools/profile_stress.sh <tag>
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/stress_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--spp 8 ${STRESS_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- python3 $R/tools/stress.py $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -3 $OUT/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/fetch -o pmc -- python3 $R/tools/stress.py $ARGS > $OUT/fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/write -o pmc -- python3 $R/tools/stress.py $ARGS > $OUT/write.log 2>&1 || { echo write failed; exit 1; }
cd $R && python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
tail -1 $OUT/trace.log | cut -c1-600
cut -c1-140 $OUT/trace/k_kernel_stats.csv | head -8
grep -A5 "k_trace<false>\|k_shadow<false>\|^k_surface" $OUT/pmc_summary.txt
