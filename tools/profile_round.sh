#!/bin/bash
# Round profile: default bench, rocprofv3 kernel trace + stats of the same command, HBM-traffic PMC passes.
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd $R && timeout -k 10 500 python3 -u bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || { echo trace failed; exit 1; }
ARGS="--steps 1 --warmup 0 --no-roofline --no-cpu-baseline"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/fetch -o pmc -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/write -o pmc -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1 || { echo write failed; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o pmc -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1 || { echo sq failed; exit 1; }
cd $R && python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
python3 tools/make_traffic.py $OUT
echo profile done
