#!/bin/bash
# Kernel trace and counters (HBM traffic, SQ activity) of one bench scene at reduced spp.
# usage (on the GPU box, from the repo root): tools/profile_scene.sh <tag> <bench.py arguments, e.g. --scene fog --spp 2>
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/scene_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$* --steps 1 --warmup 0 --no-roofline --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -3 $OUT/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/fetch -o pmc -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/write -o pmc -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1 || { echo write failed; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o pmc -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1 || { echo sq failed; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_FLAT --output-format csv -d $OUT/sq2 -o pmc -- python3 $R/bench.py $ARGS > $OUT/sq2.log 2>&1 || { echo sq2 failed; exit 1; }
cd $R && python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
cut -c1-140 $OUT/trace/k_kernel_stats.csv | head -10
echo profile done
