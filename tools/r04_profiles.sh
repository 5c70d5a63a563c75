#!/bin/bash
# usage (GPU box, repo root): tools/r04_profiles.sh [part ...]   -- the round-4 measurements that go under profiles/ (each step bounded by its own timeout)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
set -o pipefail
parts=${@:-rates bench stamps c2 counters trace all}
has() { [[ " $parts " == *" $1 "* ]]; }
if has rates; then
  { MH_ZVF=0 MH_ZVB=0 timeout -k 10 200 python tools/exp_zvb2.py 8192 16384 24576 32768 49152 65536 131072 262144 2>&1 | grep "B=" || exit 1
    MH_ZVF=0 timeout -k 10 200 python tools/exp_zvb2.py 24576 32768 49152 65536 131072 262144 2>&1 | grep "B=" || exit 1
    timeout -k 10 200 python tools/exp_zvb2.py 8192 16384 24576 32768 49152 65536 131072 262144 2>&1 | grep "B=" || exit 1; } > $out/r04_zvf_vs_others.txt || exit 1
fi
if has bench; then
  timeout -k 10 400 python bench.py --config 4 > $out/r04_bench_config4.json 2> $out/r04_bench_config4.err || exit 1
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/r04_bench_line_driver_flags.json 2> $out/r04_bench_line.err || exit 1
  timeout -k 10 400 python bench.py > $out/r04_bench_line.json 2>> $out/r04_bench_line.err || exit 1
  timeout -k 10 400 python bench.py --config 3 > $out/r04_bench_config3.json 2>> $out/r04_bench_line.err || exit 1
fi
if has stamps; then
  MH_SPEC_DIR=$root/exp_probe MH_ZVF=2 MH_ZV=0 timeout -k 10 120 python tools/exp_zvf_probe.py 262144 2>&1 | grep -v "self-check\|amdgpu.ids" > $out/r04_zvf_phase_stamps.txt || exit 1
fi
if has c2; then
  timeout -k 10 200 python tools/exp_c2_floor.py 2>&1 | grep -v "amdgpu.ids" > $out/r04_c2_floor.txt || exit 1
  MH_SPEC_DIR=$root/exp_probe_arm timeout -k 10 120 python tools/exp_c2_floor.py stamps 1024 2>&1 | grep -v "amdgpu.ids\|self-check" >> $out/r04_c2_floor.txt || exit 1
fi
if has counters; then
  timeout -k 10 500 tools/pmc_zvf.sh 262144 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE > $out/r04_sq_counters_b262144.txt 2>&1 || exit 1
fi
if has trace; then
  ( cd /tmp && export TMPDIR=/tmp MH_BENCH_NO_PMC=1 && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/r04_c4_trace -o c4 --output-format csv -- python3 $root/bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1 ) || exit 1
  cp $(find $out/r04_c4_trace -name '*kernel_stats.csv' | head -1) $out/r04_bench_config4_kernel_stats.csv
  ( cd /tmp && export TMPDIR=/tmp MH_BENCH_NO_PMC=1 && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/r04_hl_trace -o hl --output-format csv -- python3 $root/bench.py --no-cpu-baseline > /dev/null 2>&1 ) || exit 1
  cp $(find $out/r04_hl_trace -name '*kernel_stats.csv' | head -1) $out/r04_final_bench_b4096_kernel_stats.csv
  ( cd /tmp && export TMPDIR=/tmp MH_BENCH_NO_PMC=1 && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/r04_c3_trace -o c3 --output-format csv -- python3 $root/bench.py --config 3 --no-cpu-baseline > /dev/null 2>&1 ) || exit 1
  cp $(find $out/r04_c3_trace -name '*kernel_stats.csv' | head -1) $out/r04_bench_config3_kernel_stats.csv
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/r04_c2_trace -o c2 --output-format csv -- python3 $root/tools/exp_c2_floor.py 1024 > /dev/null 2>&1 ) || exit 1
  cp $(find $out/r04_c2_trace -name '*kernel_stats.csv' | head -1) $out/r04_c2_arm7_b1024_kernel_stats.csv
fi
if has all; then
  timeout -k 10 400 python tools/bench_configs.py 2>&1 | grep -v "amdgpu.ids" > $out/r04_all_configs_rates.txt || exit 1
fi
echo done
