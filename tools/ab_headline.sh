#!/bin/bash
# A/B of the headline (bench.py, driver's flags) between code objects in two directories: tools/ab_headline.sh dirA dirB ... rounds
# (alternating, so that both see the same box and clocks); prints value and ms_per_step of every run
cd "$(dirname "$0")/.."
DIRS="${@:1:$#-1}"; N=${@: -1}
for i in $(seq $N); do
  for d in $DIRS; do
    MH_SPEC_DIR=$PWD/$d MH_BENCH_NO_PMC=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$d', '%.1f M/s' % (l['value']/1e6), '%.3f us/step' % (l['ms_per_step']*1e3), 'kernel %.3f us' % (l['kernels_ms']['rnea_aba']*1e3), l['check']['ok'], l['config']['kernel_variant'][:24])"
  done
done
