#!/bin/bash
# usage (GPU box, repo root): tools/r05_profiles.sh [part ...]   -- the round-5 measurements that go under profiles/ (each step bounded by its own timeout)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
set -o pipefail
parts=${@:-bench trace pair counters c2 all}
has() { [[ " $parts " == *" $1 "* ]]; }
if has bench; then
  timeout -k 10 400 python bench.py > $out/r05_bench_line_final.json 2> $out/r05_bench_line.err || exit 1
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/r05_bench_line_driver_flags.json 2>> $out/r05_bench_line.err || exit 1
  timeout -k 10 400 python bench.py --config 3 > $out/r05_bench_config3.json 2>> $out/r05_bench_line.err || exit 1
  timeout -k 10 400 python bench.py --config 4 > $out/r05_bench_config4.json 2>> $out/r05_bench_line.err || exit 1
  timeout -k 10 600 python bench.py --config 5 > $out/r05_bench_config5.json 2>> $out/r05_bench_line.err || exit 1
  echo "bench lines done"
fi
if has trace; then
  for c in hl:"" c3:"--config 3" c4:"--config 4" c5:"--config 5"; do
    tag=${c%%:*}; flags=${c#*:}
    ( cd /tmp && export TMPDIR=/tmp MH_BENCH_NO_PMC=1 && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/r05_${tag}_trace -o $tag --output-format csv -- python3 $root/bench.py $flags --no-cpu-baseline > /dev/null 2>&1 ) || exit 1
    cp $(find $out/r05_${tag}_trace -name '*kernel_stats.csv' | head -1) $out/r05_trace_${tag}_kernel_stats.csv
    echo "trace $tag done"
  done
fi
if has pair; then
  timeout -k 10 300 python tools/exp_pair_big.py 4096 8192 16384 20480 24576 32768 65536 262144 2>&1 | grep "pair B=" > $out/r05_pair_call_rates.txt || exit 1
  echo "pair rates done"
fi
if has counters; then
  timeout -k 10 500 tools/pmc_zvf.sh 262144 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES FETCH_SIZE WRITE_SIZE > $out/r05_sq_counters_b262144.txt 2>&1 || exit 1
  echo "counters done"
fi
if has c2; then
  timeout -k 10 200 python tools/exp_c2_floor.py 2>&1 | grep -v "amdgpu.ids" > $out/r05_c2_floor.txt || exit 1
  echo "c2 done"
fi
if has all; then
  timeout -k 10 400 python tools/bench_configs.py 2>&1 | grep -v "amdgpu.ids" > $out/r05_all_configs_rates_final.txt || exit 1
  echo "all configs done"
fi
