#!/bin/bash
# memory-pipe counters (TA/TCP/TD/GRBM) over a short bench run
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--spp 16 --steps 1 --warmup 0 --no-roofline --no-cpu-baseline $*"
timeout -s KILL 180 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT/mem -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/mem.log 2>&1 || { tail -5 $OUT/mem.log; exit 1; }
echo pmc mem done
