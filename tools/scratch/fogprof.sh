#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fogprof2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--scene fog --spp 2 --steps 1 --warmup 0 --no-roofline --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o pmc -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1 || { echo sq failed; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/fetch -o pmc -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/write -o pmc -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1 || { echo write failed; exit 1; }
cd $R && python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
python3 - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/fogprof2/pmc_summary.json"))
for k,v in d.items():
    if not k.startswith(("k_","void k_")) or "FETCH_SIZE" not in v: continue
    print(k[:28], "launches", v["SQ_INSTS_VALU"]["launches"], "valu_total %.3g"%v["SQ_INSTS_VALU"]["sum"], "fetch_x2_GB %.1f"%(2*v["FETCH_SIZE"]["sum"]*1024/1e9), "write_GB %.1f"%(v["WRITE_SIZE"]["sum"]*1024/1e9), "gui_cycles %.3g"%(v["GRBM_GUI_ACTIVE"]["sum"]/8))
PY
