#!/bin/bash
# bench.py under one experiment switch at a time (two engines in flight, 200 steps)
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"; }
for k in "X=0" "M355_SMALLM=0" "M355_SMALLM=150" "M355_SMALLM=600" "M355_NO_SLAB=1" "M355_NO_CVFUSE=1" "M355_LEAN=1" "M355_NO_PERSIST=1" "M355_PERSIST=1" "M355_PERSIST=2" "M355_HALO_VARIANT=1" "M355_NO_WIDE=1" "M355_NO_UPFUSE=1" "X=1"; do run "$k"; done
