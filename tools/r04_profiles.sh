#!/bin/bash
# usage (GPU box, repo root): tools/r04_profiles.sh   -- the round-4 measurements that go under profiles/ (each step bounded by its own timeout)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
set -o pipefail
step() { echo "== $1" >&2; }
step "two launches against the one-job kernel"
{ for z in 0 1; do MH_ZVB=$z timeout -k 10 200 python tools/exp_zvb2.py 8192 16384 24576 32768 49152 65536 131072 262144 2>&1 | grep "B=" || exit 1; done; } > $out/r04_zvb_vs_tree_split.txt || exit 1
step "bench lines"
timeout -k 10 280 python bench.py --config 4 > $out/r04_bench_config4.json 2> $out/r04_bench_config4.err || exit 1
timeout -k 10 280 python bench.py --steps 20 --warmup 5 > $out/r04_bench_line_driver_flags.json 2> $out/r04_bench_line.err || exit 1
step "phase stamps of the two launches"
MH_SPEC_DIR=$root/exp_probe MH_ZVB=2 timeout -k 10 120 python tools/exp_zvb_probe.py 262144 2>&1 | grep -v "self-check\|amdgpu.ids" > $out/r04_zvb_phase_stamps.txt || exit 1
step "C2 floor"
timeout -k 10 200 python tools/exp_c2_floor.py 2>&1 | grep -v "amdgpu.ids" > $out/r04_c2_floor.txt || exit 1
MH_SPEC_DIR=$root/exp_probe_arm timeout -k 10 120 python tools/exp_c2_floor.py stamps 1024 2>&1 | grep -v "amdgpu.ids\|self-check" >> $out/r04_c2_floor.txt || exit 1
step "all configurations"
timeout -k 10 400 python tools/bench_configs.py 2>&1 | grep -v "amdgpu.ids" > $out/r04_all_configs_rates.txt || exit 1
echo done
