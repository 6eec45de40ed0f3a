#!/bin/bash
# Round profile refresh (run through gpurun from the repo root): default bench line, rocprofv3 kernel stats of the same
# command, and of the --serial command (no overlap: per-kernel averages comparable with the live event samples).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; mkdir -p $O
cd $R && python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o d -- python3 $R/bench.py --no-cpu-baseline --no-train > $O/bench_under_rocprof.json 2> $O/rocprof_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o s -- python3 $R/bench.py --serial --no-cpu-baseline --no-train > $O/bench_serial_under_rocprof.json 2> $O/rocprof_serial.err
find $O -name "*kernel_trace.csv" -delete   # tens of MB; the stats summaries are what is kept
ls -R $O | head -30
