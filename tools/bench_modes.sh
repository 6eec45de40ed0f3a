#!/bin/bash
# bench.py in its pipelining modes on one box (engines in flight x lanes), no profiling / baseline / training: images/s each.
R=${GRAFT_REPO_ROOT:-.}
for mode in "--engines 1 --lanes on" "--engines 1 --lanes off" "--engines 2 --lanes off" "--engines 2 --lanes on" "--engines 3 --lanes off" "--serial"; do
  python $R/bench.py --no-cpu-baseline --no-train --no-profile $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', d['value'], d['ms_per_step'])"
done
