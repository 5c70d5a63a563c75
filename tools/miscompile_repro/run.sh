#!/bin/bash
# Reproduces the wrong-result whole-tree ABA of round 1 and bisects it over compiler flags.
#   1. here (no GPU needed):   tools/miscompile_repro/run.sh build     -> exp_build/old = commit 786995f + the patch, library + flag variants
#   2. on a GPU box (gpurun):  tools/miscompile_repro/run.sh check     -> one line per variant and batch size
# The patch forces kWholeTreeAba on and reduces the code object to ONE kernel, spec_kernel<TP, double, ABA, rows in LDS, identity maps,
# hand-over in the global workspace> (-DMH_DIAG_ONLY -DMH_DIAG_FLAGS=3), so a variant compiles in ten seconds.
set -e
root=$(cd "$(dirname "$0")/../.." && pwd); old=$root/exp_build/old
if [ "$1" = build ]; then
   rm -rf $old && mkdir -p $old && git -C $root archive 786995f | tar -x -C $old
   (cd $old && patch -p0 mecano_amd/csrc/mh_spec.hip < $root/tools/miscompile_repro/mh_spec_786995f_diag.patch)
   cp $root/tools/miscompile_repro/build_diag.py $root/tools/miscompile_repro/diag_old3.py $old/
   cd $old && python -c "import sys; sys.path.insert(0, '.'); from mecano_amd import build; build.build_lib(force=True)"
   b() { name=$1; shift; DIAG_OUT=v_$name python build_diag.py "$@" > v_$name.log 2>&1 || echo "build $name FAILED"; }
   b base & b O1 -O1 & b O2 -O2 & b no_slp -fno-slp-vectorize & b signed_zeros -fsigned-zeros & b no_finite_math -fno-finite-math-only &
   b no_misched -mllvm -enable-misched=0 & b no_post_misched -mllvm -enable-post-misched=0 & wait
   b no_machine_licm -mllvm -disable-machine-licm & b sgpr_spill_to_memory -mllvm -amdgpu-spill-sgpr-to-vgpr=0 & b no_agpr_spill -mllvm -amdgpu-spill-vgpr-to-agpr=0 &
   b wwm_regalloc_basic -mllvm -wwm-regalloc=basic & b sgpr_regalloc_basic -mllvm -sgpr-regalloc=basic & b sgpr_regalloc_fast -mllvm -sgpr-regalloc=fast &
   b vgpr_regalloc_basic -mllvm -vgpr-regalloc=basic & b passing_plan_rows_direct -DMH_DIAG_FLAGS=2 & wait
else
   cd $old
   for v in v_*/; do
      io=1; [ "$v" = v_passing_plan_rows_direct/ ] && io=0
      DIAG_IO=$io MH_SPEC_DIR=$v python diag_old3.py ${v%/} 2>&1 | grep -v amdgpu.ids
   done
fi
