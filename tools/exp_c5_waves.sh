#!/bin/bash
# config 5 pair call against the resident waves per CU (MH_WAVES_PER_CU: fewer waves = more LDS per wave for the depth stack)
cd "$(dirname "$0")/.."
for B in 131072 1048576; do
 for w in 8 6 5 4 3; do
  MH_WAVES_PER_CU=$w MH_BENCH_NO_PMC=1 python bench.py --config 5 --batch $B --steps 5 --warmup 2 --regions 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config 5 B=$B MH_WAVES_PER_CU=$w', '%.1f M/s' % (l['value']/1e6), '%.3f ms/step' % (l['ms_per_step']), l['check']['ok'])"
 done
done
