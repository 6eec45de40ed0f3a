#!/bin/bash
# Ordered kernel sequence of one training step (run through gpurun from the repo root); result in gpurun_out/train_seq.txt
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tseq
rocprofv3 --kernel-trace --output-format csv -d /tmp/tseq -o t -- python3 $R/tools/train_bench.py s 64 640 3 > $R/gpurun_out/train_seq.log 2>&1
python3 $R/tools/trace_seq.py $(find /tmp/tseq -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/train_seq.txt
tail -3 $R/gpurun_out/train_seq.txt
