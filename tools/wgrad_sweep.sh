#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wgrad_sweep; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/tools/wgrad_sweep.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 $R/tools/wgrad_sweep.py --parse $(find $O -name "t_kernel_trace.csv") > $R/gpurun_out/wgrad_sweep.txt
find $O -name "*.csv" -delete
cat $R/gpurun_out/wgrad_sweep.txt
