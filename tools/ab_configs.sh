#!/bin/bash
# A/B of bench.py --config C between code-object directories: tools/ab_configs.sh C dirA dirB ... rounds
cd "$(dirname "$0")/.."
C=$1; shift
DIRS="${@:1:$#-1}"; N=${@: -1}
for i in $(seq $N); do
  for d in $DIRS; do
    MH_SPEC_DIR=$PWD/$d MH_BENCH_NO_PMC=1 python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config $C', '$d', '%.1f M/s' % (l['value']/1e6), '%.3f us/step' % (l['ms_per_step']*1e3), l['check'].get('ok'))"
  done
done
