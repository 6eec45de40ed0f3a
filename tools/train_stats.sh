#!/bin/bash
# Per-kernel totals + ordered sequence of the training step under rocprofv3, then the m-seg (config 4, one GPU's share) step time
# (run through gpurun from the repo root); results in gpurun_out/train_stats.csv, train_seq.txt, tb_m.log
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tstat
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tstat -o t -- python3 $R/tools/train_bench.py s 64 640 4 > $R/gpurun_out/train_stats.log 2>&1
cp $(find /tmp/tstat -name "*kernel_stats.csv" | head -1) $R/gpurun_out/train_stats.csv
python3 $R/tools/trace_seq.py $(find /tmp/tstat -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/train_seq.txt
tail -1 $R/gpurun_out/train_seq.txt
cd $R
python3 tools/train_bench.py m 64 640 4 > gpurun_out/tb_m.log 2>&1
tail -2 gpurun_out/tb_m.log
