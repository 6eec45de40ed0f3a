#!/bin/bash
# HBM traffic per kernel family: two separate rocprofv3 --pmc passes over the serial bench (every kernel alone).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $O/f.json 2> $O/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $O/w.json 2> $O/w.err
cd $R && python tools/pmc_traffic.py $(find $O/f -name "*counter_collection.csv" | head -1) $(find $O/w -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json
find $O -name "*counter_collection.csv" -delete
