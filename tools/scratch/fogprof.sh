#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fogprof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--scene fog --spp 1 --steps 1 --warmup 0 --no-roofline --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -3 $OUT/trace.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o pmc -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1 || { echo sq failed; exit 1; }
cd $R && python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/trace/*kernel_stats.csv | cut -c1-150 | head -12
grep -A9 "^k_volume" $OUT/pmc_summary.txt
