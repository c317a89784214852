#!/bin/bash
# PMC passes over a short bench run (one counter group per process, as rocprofv3 does not split passes).
# usage: tools/pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--spp 16 --steps 1 --warmup 0 --no-roofline --no-cpu-baseline $*"
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/sq -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/sq.log 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/fetch -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/write -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/write.log 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/sq2 -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/sq2.log 2>&1 || exit 1
echo pmc done
