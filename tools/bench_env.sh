#!/bin/bash
# usage: tools/bench_env.sh VAR v1 v2 ...   -> runs bench.py (no profiling) once per value of the env var
var=$1; shift
for v in "$@"; do
  export $var=$v
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], d['ms_per_step'])"
done
