#!/bin/bash
# HBM traffic and MFMA utilisation per kernel family: separate rocprofv3 --pmc passes over the serial bench (every kernel alone).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--serial --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-train"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/bench.py $ARGS > $O/f.json 2> $O/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/bench.py $ARGS > $O/w.json 2> $O/w.err
# MFMA utilisation (SQ counters + GRBM in their own pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/m -o m -- python3 $R/bench.py $ARGS > $O/m.json 2> $O/m.err
cd $R && python tools/pmc_mfma.py $(find $O/m -name "*counter_collection.csv" | head -1) $O/pmc_mfma.json
cd $R && python tools/pmc_traffic.py $(find $O/f -name "*counter_collection.csv" | head -1) $(find $O/w -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json
find $O -name "*counter_collection.csv" -delete
