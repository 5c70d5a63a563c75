"""The compile-time partition of the humanoid for the tree-split kernels (mh_spec_kernels.h: Split<TP>), read back on the host: a small
program that includes the kernels' headers prints the plan (hipcc compiles it here without a GPU).  Pins what DESIGN.md §6.2 says about the
fused forward dynamics of device-filling batches: the head -- a one-body limb on the chest -- is owned by a leg's wave in the forward
dynamics and walked by an arm's wave in the inverse-dynamics phase (Split::mail_one_limb), its three values travel through LDS slots
behind everything else, and two workgroups still fit a CU."""
import os
import shutil
import subprocess

import pytest

from mecano_amd import build as mbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
#include "mh_spec_kernels.h"
#include "mh_zv_kernels.h"
struct TP
{
   static constexpr int N = MH_TOPO_N;
   static constexpr int parent[N] = {MH_TOPO_PARENTS};
   static constexpr int type[N] = {MH_TOPO_TYPES};
};
int main()
{
   using S = mh::Split<TP>;
   using P = mh::ZvfPlan<TP>;
   printf("limbs %d staged %d mailed %d use_mail %d mail_base %d lds_slots %d usable %d\n", S::n_limbs(), (int)S::staged(), S::mailed_limb(), (int)P::use_mail(),
          P::mail_base(), P::lds_slots(), (int)P::usable());
   for (int k = 0; k < S::n_limbs(); k++)
      printf("limb %d root %d owner %d walker %d plain %d\n", k, S::limb_root(k), S::owner(k), S::template owner_sel<2>(k), S::owner_plain(k));
   for (int j = 0; j < TP::N; j++)
      if (S::is_trunk(j))
         printf("trunk %d wrench %d\n", j, S::template owner_sel<2>(S::template f_limb_sel<2>(j)));
   return 0;
}
"""


@pytest.mark.timeout(300)
def test_humanoid_split_plan_mails_the_head(tmp_path):
    hipcc = mbuild.hipcc()
    if not (hipcc and (shutil.which(hipcc) or os.path.exists(hipcc))):
        pytest.skip("no hipcc")
    key, parents, kinds = mbuild.topology_of(mbuild.registered_models()["humanoid30"])
    src, exe = tmp_path / "plan.hip", tmp_path / "plan"
    src.write_text(SRC)
    defs = [f"-DMH_TOPO_N={len(parents)}", "-DMH_TOPO_PARENTS=" + ",".join(str(p) for p in parents), "-DMH_TOPO_TYPES=" + ",".join(str(k) for k in kinds)]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-w", "-I", os.path.join(ROOT, "mecano_amd", "csrc")] + defs + [str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).splitlines()
    head = dict(zip(out[0].split()[0::2], (int(v) for v in out[0].split()[1::2])))
    assert head["limbs"] == 5 and head["staged"] == 1 and head["usable"] == 1
    limbs = [dict(zip(l.split()[0::2], (int(v) for v in l.split()[1::2]))) for l in out if l.startswith("limb ")]
    sizes = {l["limb"]: sum(1 for j in range(len(parents)) if _root_of(parents, j, {m["root"] for m in limbs}) == l["root"]) for l in limbs}
    one_body = [l for l in limbs if sizes[l["limb"]] == 1]
    assert len(one_body) == 1  # the head
    h = one_body[0]
    assert head["mailed"] == h["limb"] and head["use_mail"] == 1
    # forward dynamics: beside a leg (a six-body limb); inverse-dynamics phase: on the wave of an arm (which hangs on the same trunk body)
    leg_waves = {l["owner"] for l in limbs if sizes[l["limb"]] == 6}
    arm_walkers = {l["walker"] for l in limbs if sizes[l["limb"]] == 4 and parents[l["root"]] == parents[h["root"]]}
    assert h["owner"] in leg_waves and h["walker"] in arm_walkers and h["walker"] != h["owner"]
    # nobody else moves, and the plain inverse dynamics had the head beside an arm all along
    assert all(l["walker"] == l["owner"] for l in limbs if l is not h)
    assert h["plain"] in {l["plain"] for l in limbs if sizes[l["limb"]] == 4}
    # three mail slots behind both phases' LDS, two workgroups per CU
    assert head["lds_slots"] == head["mail_base"] + 3 and head["lds_slots"] * 64 * 8 * 2 <= 160 * 1024
    # the trunk's wrenches: the pelvis on a leg's wave, the torso bodies on the arms' waves (the only ones that walk them)
    trunk = [dict(zip(l.split()[0::2], (int(v) for v in l.split()[1::2]))) for l in out if l.startswith("trunk ")]
    for t in trunk:
        if parents[t["trunk"]] < 0:
            assert t["wrench"] in leg_waves
        else:
            assert t["wrench"] in arm_walkers


def _root_of(parents, j, roots):
    while j >= 0 and j not in roots:
        j = parents[j]
    return j
