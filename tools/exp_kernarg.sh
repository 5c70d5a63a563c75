cd /root/repo
for v in 0 1 0 1; do
  HIP_FORCE_DEV_KERNARG=$v MH_SPEC_DIR=$PWD/build/exp_new MH_BENCH_NO_PMC=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('HIP_FORCE_DEV_KERNARG=$v', '%.1f M/s' % (l['value']/1e6), '%.3f us/step' % (l['ms_per_step']*1e3), 'kernel %.3f us' % (l['kernels_ms']['rnea_aba']*1e3))"
done
