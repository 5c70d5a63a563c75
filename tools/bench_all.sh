#!/bin/bash
# The metric and the other BASELINE configurations with bench.py (no CPU baseline, no PMC passes): one JSON summary line each.
cd "$(dirname "$0")/.."
for cfg in 0 3 4; do
  MH_BENCH_NO_PMC=1 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config $cfg', '%.1f M/s' % (l['value']/1e6), '%.3f us/step' % (l['ms_per_step']*1e3), 'kernels', {k: round(v*1e3,3) for k,v in l['kernels_ms'].items()}, 'frac %.4f' % l['roofline']['frac'], l['check'].get('ok'))"
done
