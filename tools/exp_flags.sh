#!/bin/bash
# Experiment: compiler-flag variants of the humanoid's tree-split kernels (MINIMAL build), for tools/exp_flags_run.sh on the GPU box.
# usage: tools/exp_flags.sh   (builds exp_build/flags/<name>/libmecano_hip_topo_<key>.so in parallel)
cd "$(dirname "$0")/.."
b() { name=$1; shift; mkdir -p exp_build/flags/$name; python tools/isa.py --so "$@" > exp_build/flags/$name.log 2>&1 && mv exp_build/libmecano_hip_topo_b5c1e26c784c54fa.so exp_build/flags/$name/ || echo "$name FAILED"; }
# tools/isa.py writes to one fixed path: build one at a time
b base
b maxilp -mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1
b nomisched -mllvm -enable-misched=0
b nopostmisched -mllvm -enable-post-misched=0
b O2 -O2
b vgpr_basic -mllvm -vgpr-regalloc=basic
b sgpr_basic -mllvm -sgpr-regalloc=basic
b no_agpr_spill -mllvm -amdgpu-spill-vgpr-to-agpr=0
b no_rp_resched -mllvm -amdgpu-disable-unclustered-high-rp-reschedule=1
b no_licm -mllvm -disable-machine-licm
b no_sink -mllvm -disable-machine-sink
ls exp_build/flags
