for plan in 1220 1100 0000 1210 1221 1220; do
  M355_LANE_PLAN=$plan python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$plan', d['value'], d['ms_per_step'])" >> gpurun_out/lanes.log
done
