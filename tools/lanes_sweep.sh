for plan in 1220 1222 1221 1230 1231 1233 2110 1220; do
  M355_LANE_PLAN=$plan python bench.py --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$plan', d['value'], d['ms_per_step'])" >> gpurun_out/lanes.log
done
