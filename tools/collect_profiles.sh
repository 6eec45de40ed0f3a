#!/bin/bash
# Copy the outputs of tools/profile_refresh.sh, tools/pmc_refresh.sh, tools/op_table.py and tools/train_stats.sh from gpurun_out/ (scratch)
# into profiles/ under the round's prefix:  tools/collect_profiles.sh r04 [op_table_file]
set -e
P=${1:?round prefix}; T=${2:-gpurun_out/op_table.txt}; O=gpurun_out
cp $O/refresh/bench_default.json profiles/${P}_bench_default.json
cp $O/refresh/bench_under_rocprof.json profiles/${P}_bench_under_rocprof.json
cp $O/refresh/bench_serial_under_rocprof.json profiles/${P}_bench_serial_under_rocprof.json
cp $O/refresh/prof_default/d_kernel_stats.csv profiles/${P}_bench_kernel_stats.csv
cp $O/refresh/prof_serial/s_kernel_stats.csv profiles/${P}_bench_serial_kernel_stats.csv
cp $O/pmc/pmc_traffic.json profiles/${P}_pmc_traffic.json
cp $O/pmc/pmc_mfma.json profiles/${P}_pmc_mfma.json
[ -f $T ] && cp $T profiles/${P}_op_table.txt
[ -f $O/train_stats.csv ] && cp $O/train_stats.csv profiles/${P}_train_kernel_stats.csv
[ -f $O/train_seq.txt ] && cp $O/train_seq.txt profiles/${P}_train_step_sequence.txt
ls -la profiles/${P}_*
