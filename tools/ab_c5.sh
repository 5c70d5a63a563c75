#!/bin/bash
# config 5 (bench.py --config 5) with the fused depth-first pair walk on and off, at the per-GPU shard and at the full batch
cd "$(dirname "$0")/.."
for B in 131072 1048576; do
 for v in 0 1 0 1; do
  MH_DFS_PAIR=$v MH_BENCH_NO_PMC=1 python bench.py --config 5 --batch $B --steps 5 --warmup 2 --regions 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config 5 B=$B MH_DFS_PAIR=$v', '%.1f M/s' % (l['value']/1e6), '%.3f ms/step' % (l['ms_per_step']), l['check'])"
 done
done
