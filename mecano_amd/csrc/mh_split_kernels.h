// mh_split_kernels.h -- run-time-topology RNEA / ABA / CRBA with the tree split over the FOUR waves of a workgroup (small batches).
//
// A batch of a few thousand configurations puts one wave per 64 configurations on the device: 64 waves on 1024 SIMDs at B = 4096, each
// walking all n bodies one after the other.  The topology-specialised code objects split the tree over four waves at compile time
// (mh_spec_kernels.h, Split<TP>); these kernels do the same for ANY model from a plan the host makes when the model is created
// (mh_api.hip: split_rt_plan): the TRUNK (bodies the split goes through: the root side of every branching that is used) and the LIMBS
// (whole subtrees hanging off trunk bodies; joints are stored depth-first, so a limb is a contiguous index range), dealt to the waves
// largest first.
//
//   outward sweeps   every wave walks the trunk (redundantly: it needs the trunk's velocities / accelerations for its limbs and would
//                    idle otherwise; all waves store the same values), then its own limbs;
//   inward sweep     every wave its own limbs; a limb root hands its contribution (RNEA: wrench, 6; ABA: articulated inertia + bias
//                    wrench, 27) to an EXCHANGE record in the workgroup's workspace block instead of its parent's slots; barrier; wave 0
//                    folds the trunk, reading the records of the limbs attached to each trunk body; (ABA) barrier.
//
// Same per-body arithmetic, workspace slots and flags as the sweep kernels of mh_kernels.h (rnea_kernel / aba_kernel; the body records
// are a copy with the flags of the limb roots and trunk bodies adapted), per-body accelerations / twists and joint wrenches included; no
// acceleration-source joints.  InverseDynamicsCalculator.java:873-966, ForwardDynamicsCalculator.java:1085-1310.
#pragma once
#include <type_traits>
#include "mh_dfs_kernels.h"

namespace mh
{
constexpr int SPLIT_WAVES = 4;
constexpr int SPLIT_MAX_SEG = 16; // limbs per wave
// The workgroup's workspace block: the first slots live in LDS (a small batch gives a workgroup the CU to itself: 160 KB that would sit
// idle, and a walk's chain of dependent workspace round trips is what these kernels wait for), the rest in the global block.  The slot
// numbers in the adapted body records carry their home like the depth-first kernels' (DFS_LDS bit, mh_dfs_kernels.h: DStack, st_*);
// a group of slots (a 6-vector, a 21 + 6 record, ...) is homed as a whole by its first slot.
constexpr int SPLIT_LDS_MARGIN = 48; // slots past the boundary a group that starts below it may reach
// MODE: 0 = every slot in LDS, 1 = every slot in the global block (no branch in either), 2 = as the slot code says
template <typename T, class SK>
MH_DEV T sw_ld(const SK &S, int code)
{
   if (SK::mode == 0 || (SK::mode == 2 && (code & DFS_LDS)))
      return S.lds[(code & DFS_SLOT) * 64];
   return S.glb[(long)code * 64];
}
template <typename T, class SK>
MH_DEV void sw_st(const SK &S, int code, T v)
{
   if (SK::mode == 0 || (SK::mode == 2 && (code & DFS_LDS)))
      S.lds[(code & DFS_SLOT) * 64] = v;
   else
      S.glb[(long)code * 64] = v;
}
template <typename T, class SK>
MH_DEV void sw_store_ri(const SK &S, int code, const RI<T> &r)
{
   const T v[10] = {r.m, r.h.x, r.h.y, r.h.z, r.I.xx, r.I.xy, r.I.xz, r.I.yy, r.I.yz, r.I.zz};
   if (SK::mode == 0 || (SK::mode == 2 && (code & DFS_LDS)))
   {
      const dfs_lds_ptr<T> sp = S.lds + (code & DFS_SLOT) * 64;
#pragma unroll
      for (int k = 0; k < 10; k++)
         sp[k * 64] = v[k];
   }
   else
   {
      T *const sp = S.glb + (long)code * 64;
#pragma unroll
      for (int k = 0; k < 10; k++)
         sp[k * 64] = v[k];
   }
}
template <typename T, class SK>
MH_DEV RI<T> sw_load_ri(const SK &S, int code)
{
   T v[10];
   if (SK::mode == 0 || (SK::mode == 2 && (code & DFS_LDS)))
   {
      const dfs_lds_ptr<T> sp = S.lds + (code & DFS_SLOT) * 64;
#pragma unroll
      for (int k = 0; k < 10; k++)
         v[k] = sp[k * 64];
   }
   else
   {
      const T *const sp = S.glb + (long)code * 64;
#pragma unroll
      for (int k = 0; k < 10; k++)
         v[k] = sp[k * 64];
   }
   RI<T> r;
   r.m = v[0], r.h = V3<T>{v[1], v[2], v[3]}, r.I = S3<T>{v[4], v[5], v[6], v[7], v[8], v[9]};
   return r;
}
// joint transform on the first / a later visit of a body: (cos, sin) of revolute joints go through the workspace, the rest is re-read from q
template <typename T, class SK>
MH_DEV JX<T> sw_joint_from_q(const SK &S, int type, ciptr cfg_map, int cfg_ofs, const T *qrow, long q_es, int jp)
{
   const JX<T> jx = joint_from_q<T>(type, cfg_map, cfg_ofs, qrow, q_es, (T *)nullptr, 0, 0, false);
   if (type == JT_REVOLUTE)
      sw_st<T>(S, jp, jx.c), sw_st<T>(S, jp + 1, jx.s);
   return jx;
}
template <typename T, class SK>
MH_DEV JX<T> sw_joint_again(const SK &S, int type, ciptr cfg_map, int cfg_ofs, const T *qrow, long q_es, int jp)
{
   if (type == JT_REVOLUTE)
   {
      JX<T> jx;
      jx.c = sw_ld<T>(S, jp), jx.s = sw_ld<T>(S, jp + 1), jx.d = T(0);
      return jx;
   }
   return joint_again<T>(type, cfg_map, cfg_ofs, qrow, q_es, (const T *)nullptr, 0, 0);
}
struct SplitDev
{
   const int *meta;    // [n][MI_STRIDE]: MI_FLAGS adapted, MI_HAND = exchange slot of a limb root (-1 otherwise)
   const int *trunk;   // trunk bodies, ascending
   const int *seg;     // [SPLIT_WAVES][SPLIT_MAX_SEG][2]: (first, one past last) body of the wave's limbs, ascending
   const int *xl_ofs;  // [n_trunk + 1]: per trunk body (position in `trunk`), its attached limbs' exchange slots in xl
   const int *xl;
   int n_trunk, slots; // workspace slots of one workgroup block: the sweep kernels' + the exchange records
   int n_seg[SPLIT_WAVES];
   int roles;          // pair_split_kernel: 0 = first half of the grid inverse dynamics, second half forward dynamics; 2 = forward dynamics only
};

// ============================================================================================ RNEA
// (a device function so that one launch can run it next to the other algorithm's: pair_split_kernel below.  blk / nblk: this
// workgroup's first group of 64 configurations and the stride to its next; ws_blk: its block of the workspace)
template <typename T, int MODE>
MH_DEV void rnea_split_body(const Args<T> &A, const SplitDev &P, long blk, long nblk, long ws_blk)
{
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   const ciptr meta = as_const(P.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map);
   const ciptr trunk = as_const(P.trunk), seg = as_const(P.seg), xl_ofs = as_const(P.xl_ofs), xl = as_const(P.xl);
   const int tid = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   extern __shared__ double lds_raw[];
   const DStack<T, MODE> S{(dfs_lds_ptr<T>)lds_raw + tid, A.ws + ws_blk * ((long)P.slots * 64) + tid};
   const V3<T> Z{T(0), T(0), T(0)};
   const int n_seg = P.n_seg[wave];
   const long groups = (A.B + 63) / 64;

   for (long grp = blk; grp < groups; grp += nblk)
   {
      const long cfg0 = grp * 64 + tid;
      const bool active = cfg0 < A.B;
      const long cfg = active ? cfg0 : A.B - 1; // the lanes of a ragged last group repeat its last configuration (no store)
      const T *qrow = A.q + cfg * A.q_bs;
      const T *qdrow = A.qd + cfg * A.v_bs;
      const T *qddrow = A.in3 + cfg * A.v_bs;
      const T *frow = A.fext ? A.fext + cfg * A.f_bs : nullptr;
      T *trow = A.out + cfg * A.v_bs;

      // ---- outward sweep: velocities, accelerations, Newton-Euler wrench (InverseDynamicsCalculator.java:873-917)
      SV<T> v_prev{Z, Z}, a_prev{Z, Z};
      auto outward = [&](int j, bool writes) {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type_rt = mi[MI_TYPE], flags = mi[MI_FLAGS];
         auto kbody = [&](auto kind) {
         const int type = kind;
         const CRef<T, false> c{CB + j * MC_STRIDE};
         SV<T> vp, ap;
         if (parent < 0)
         {
            vp = SV<T>{Z, Z};
            ap = root_acceleration(A); // :343-348
         }
         else if (flags & MF_PARENT_ADJ)
            vp = v_prev, ap = a_prev;
         else
         {
            const int sp = meta[parent * MI_STRIDE + MI_SLOT_VA];
            vp = st_load6<T>(S, sp);
            ap = st_load6<T>(S, sp + 6);
         }
         const XF<T> Xb = load_xb<T>(c);
         const JX<T> jx = sw_joint_from_q<T>(S, type, cfg_map, mi[MI_CFG], qrow, A.q_es, mi[MI_SLOT_JP]);
         const SV<T> vJ = joint_vec<T>(type, dof_map, mi[MI_DOF], qdrow, A.v_es, A.coriolis != 0);
         const SV<T> aJ = joint_vec<T>(type, dof_map, mi[MI_DOF], qddrow, A.v_es, A.accel != 0);
         SV<T> v = motion_down(type, jx, Xb, vp) + vJ;
         const SV<T> a = motion_down(type, jx, Xb, ap) + aJ + crm(v, vJ);
         if (!A.coriolis)
            v = SV<T>{Z, Z};
         if (writes)
         { // optional per-body outputs (RigidBodyAccelerationProvider; as rnea_kernel<.., BODIES>)
            if (A.body_acc)
               store_body_motion<T>(c, A.body_acc + cfg * A.f_bs, A.f_es, mi[MI_EXT], a);
            if (A.body_twist)
               store_body_motion<T>(c, A.body_twist + cfg * A.f_bs, A.f_es, mi[MI_EXT], v);
         }
         const RI<T> I = load_inertia<T>(c);
         SV<T> f = mul(I, a) + crf(v, mul(I, v));
         if (frow)
            f = f - load_fext<T>(c, frow, A.f_es, mi[MI_EXT]);
         st_store6<T>(S, mi[MI_SLOT_F], f);
         if (flags & MF_STORE_VA)
         {
            st_store6<T>(S, mi[MI_SLOT_VA], v);
            st_store6<T>(S, mi[MI_SLOT_VA] + 6, a);
         }
         v_prev = v, a_prev = a;
         }; // kbody
         switch (type_rt)
         { // one dispatch on the joint kind per body step, straight-line code per kind (mh_dfs_kernels.h)
            case JT_REVOLUTE: kbody(std::integral_constant<int, JT_REVOLUTE>{}); break;
            case JT_PRISMATIC: kbody(std::integral_constant<int, JT_PRISMATIC>{}); break;
            case JT_SIXDOF: kbody(std::integral_constant<int, JT_SIXDOF>{}); break;
            case JT_PLANAR: kbody(std::integral_constant<int, JT_PLANAR>{}); break;
            case JT_SPHERICAL: kbody(std::integral_constant<int, JT_SPHERICAL>{}); break;
            default: kbody(std::integral_constant<int, JT_FIXED>{}); break;
         }
      };
      for (int k = 0; k < P.n_trunk; k++)
         outward(trunk[k], active && wave == 0);
      for (int s = 0; s < n_seg; s++)
      {
         const int j0 = seg[(wave * SPLIT_MAX_SEG + s) * 2], j1 = seg[(wave * SPLIT_MAX_SEG + s) * 2 + 1];
         for (int j = j0; j < j1; j++)
            outward(j, active);
      }
      // ---- inward sweep (:930-966)
      SV<T> carry{Z, Z};
      bool have_carry = false;
      auto inward = [&](int j, int xk0, int xk1) {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type_rt = mi[MI_TYPE], flags = mi[MI_FLAGS], xs = mi[MI_HAND];
         auto kbody = [&](auto kind) {
         const int type = kind;
         const CRef<T, false> c{CB + j * MC_STRIDE};
         SV<T> f = st_load6<T>(S, mi[MI_SLOT_F]);
         if (have_carry)
            f = f + carry;
         for (int k = xk0; k < xk1; k++) // trunk bodies: what the attached limbs handed up
            f = f + st_load6<T>(S, xl[k]);
         if (active)
         {
            write_joint_rows<T>(type, dof_map + mi[MI_DOF], trow, A.v_es, f);
            if (A.joint_wrench) // InverseDynamicsCalculator.getComputedJointWrench (:578-585)
               store_joint_wrench<T>(c, A.joint_wrench + cfg * A.f_bs, A.f_es, mi[MI_EXT], f);
         }
         have_carry = false;
         if (parent >= 0)
         {
            const XF<T> Xb = load_xb<T>(c);
            const JX<T> jx = sw_joint_again<T>(S, type, cfg_map, mi[MI_CFG], qrow, A.q_es, mi[MI_SLOT_JP]);
            const SV<T> fp = force_up(type, jx, Xb, f);
            if (xs >= 0)
               st_store6<T>(S, xs, fp); // a limb root: through the exchange record
            else if (flags & MF_PARENT_ADJ)
               carry = fp, have_carry = true;
            else
               st_add6<T>(S, meta[parent * MI_STRIDE + MI_SLOT_F], fp);
         }
         }; // kbody
         switch (type_rt)
         { // one dispatch on the joint kind per body step, straight-line code per kind (mh_dfs_kernels.h)
            case JT_REVOLUTE: kbody(std::integral_constant<int, JT_REVOLUTE>{}); break;
            case JT_PRISMATIC: kbody(std::integral_constant<int, JT_PRISMATIC>{}); break;
            case JT_SIXDOF: kbody(std::integral_constant<int, JT_SIXDOF>{}); break;
            case JT_PLANAR: kbody(std::integral_constant<int, JT_PLANAR>{}); break;
            case JT_SPHERICAL: kbody(std::integral_constant<int, JT_SPHERICAL>{}); break;
            default: kbody(std::integral_constant<int, JT_FIXED>{}); break;
         }
      };
      for (int s = n_seg - 1; s >= 0; s--)
      {
         const int j0 = seg[(wave * SPLIT_MAX_SEG + s) * 2], j1 = seg[(wave * SPLIT_MAX_SEG + s) * 2 + 1];
         have_carry = false;
         for (int j = j1 - 1; j >= j0; j--)
            inward(j, 0, 0);
      }
      __syncthreads(); // the limbs' records are in the workgroup's block
      if (wave == 0)
      {
         have_carry = false;
         for (int k = P.n_trunk - 1; k >= 0; k--)
            inward(trunk[k], xl_ofs[k], xl_ofs[k + 1]);
      }
      __syncthreads(); // the next group of configurations re-uses the block
   }
}

// ============================================================================================ ABA
// (a device function so that one launch can run it next to the other algorithm's: pair_split_kernel below.  blk / nblk: this
// workgroup's first group of 64 configurations and the stride to its next; ws_blk: its block of the workspace)
template <typename T, int MODE>
MH_DEV void aba_split_body(const Args<T> &A, const SplitDev &P, long blk, long nblk, long ws_blk)
{
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   const ciptr meta = as_const(P.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map);
   const ciptr trunk = as_const(P.trunk), seg = as_const(P.seg), xl_ofs = as_const(P.xl_ofs), xl = as_const(P.xl);
   const int tid = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   extern __shared__ double lds_raw[];
   const DStack<T, MODE> S{(dfs_lds_ptr<T>)lds_raw + tid, A.ws + ws_blk * ((long)P.slots * 64) + tid};
   const V3<T> Z{T(0), T(0), T(0)};
   const int n_seg = P.n_seg[wave];
   const long groups = (A.B + 63) / 64;

   for (long grp = blk; grp < groups; grp += nblk)
   {
      const long cfg0 = grp * 64 + tid;
      const bool active = cfg0 < A.B;
      const long cfg = active ? cfg0 : A.B - 1;
      const T *qrow = A.q + cfg * A.q_bs;
      const T *qdrow = A.qd + cfg * A.v_bs;
      const T *taurow = A.in3 + cfg * A.v_bs;
      const T *frow = A.fext ? A.fext + cfg * A.f_bs : nullptr;
      T *orow = A.out + cfg * A.v_bs;

      // ---- pass one (ForwardDynamicsCalculator.java:1085-1127): velocities, bias wrench p, bias acceleration c
      SV<T> v_prev{Z, Z};
      auto pass1 = [&](int j, bool writes) {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type_rt = mi[MI_TYPE], flags = mi[MI_FLAGS];
         auto kbody = [&](auto kind) {
         const int type = kind;
         const CRef<T, false> c{CB + j * MC_STRIDE};
         SV<T> vp;
         if (parent < 0)
            vp = SV<T>{Z, Z};
         else if (flags & MF_PARENT_ADJ)
            vp = v_prev;
         else
            vp = st_load6<T>(S, meta[parent * MI_STRIDE + MI_SLOT_VA]);
         const XF<T> Xb = load_xb<T>(c);
         const JX<T> jx = sw_joint_from_q<T>(S, type, cfg_map, mi[MI_CFG], qrow, A.q_es, mi[MI_SLOT_JP]);
         const SV<T> vJ = joint_vec<T>(type, dof_map, mi[MI_DOF], qdrow, A.v_es, true);
         const SV<T> v = motion_down(type, jx, Xb, vp) + vJ;
         if (writes && A.body_twist)
            store_body_motion<T>(c, A.body_twist + cfg * A.f_bs, A.f_es, mi[MI_EXT], v);
         const RI<T> I = load_inertia<T>(c);
         SV<T> p = crf(v, mul(I, v));
         if (frow)
            p = p - load_fext<T>(c, frow, A.f_es, mi[MI_EXT]);
         st_store6<T>(S, mi[MI_SLOT_F], p);
         st_store6<T>(S, mi[MI_SLOT_C], crm(v, vJ));
         if (flags & MF_STORE_VA)
            st_store6<T>(S, mi[MI_SLOT_VA], v);
         v_prev = v;
         }; // kbody
         switch (type_rt)
         { // one dispatch on the joint kind per body step, straight-line code per kind (mh_dfs_kernels.h)
            case JT_REVOLUTE: kbody(std::integral_constant<int, JT_REVOLUTE>{}); break;
            case JT_PRISMATIC: kbody(std::integral_constant<int, JT_PRISMATIC>{}); break;
            case JT_SIXDOF: kbody(std::integral_constant<int, JT_SIXDOF>{}); break;
            case JT_PLANAR: kbody(std::integral_constant<int, JT_PLANAR>{}); break;
            case JT_SPHERICAL: kbody(std::integral_constant<int, JT_SPHERICAL>{}); break;
            default: kbody(std::integral_constant<int, JT_FIXED>{}); break;
         }
      };
      for (int k = 0; k < P.n_trunk; k++)
         pass1(trunk[k], active && wave == 0);
      for (int s = 0; s < n_seg; s++)
      {
         const int j0 = seg[(wave * SPLIT_MAX_SEG + s) * 2], j1 = seg[(wave * SPLIT_MAX_SEG + s) * 2 + 1];
         for (int j = j0; j < j1; j++)
            pass1(j, active);
      }
      // ---- pass two (:1136-1254): articulated inertias and bias wrenches, leaves to root
      ABI<T> Icarry;
      SV<T> pcarry{Z, Z};
      bool have_carry = false;
      auto pass2 = [&](int j, int xk0, int xk1) {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type_rt = mi[MI_TYPE], flags = mi[MI_FLAGS], xs = mi[MI_HAND];
         auto kbody = [&](auto kind) {
         const int type = kind;
         const CRef<T, false> c{CB + j * MC_STRIDE};
         ABI<T> IA = abi_from_rigid(load_inertia<T>(c));
         SV<T> pA = st_load6<T>(S, mi[MI_SLOT_F]);
         if (have_carry)
         {
            add(IA, Icarry);
            pA = pA + pcarry;
         }
         if (flags & MF_HAS_ACC)
            add(IA, st_load_abi<T>(S, mi[MI_SLOT_IA])); // (the bias wrenches of those children were added to slot F)
         for (int k = xk0; k < xk1; k++)
         { // trunk bodies: articulated inertia and bias wrench of the attached limbs
            add(IA, st_load_abi<T>(S, xl[k]));
            pA = pA + st_load6<T>(S, xl[k] + 21);
         }
         have_carry = false;
         const int sf = mi[MI_SLOT_F];
         ciptr di = dof_map + mi[MI_DOF];
         ABI<T> Ia = IA;
         SV<T> pa = pA;
         bool handed_up = false;
         if (type == JT_REVOLUTE || type == JT_PRISMATIC)
         {
            V3<T> ua, ul;
            T D, pz;
            if (type == JT_REVOLUTE)
               ua = V3<T>{IA.A.xz, IA.A.yz, IA.A.zz}, ul = V3<T>{IA.C.zx, IA.C.zy, IA.C.zz}, D = IA.A.zz, pz = pA.a.z;
            else
               ua = V3<T>{IA.C.xz, IA.C.yz, IA.C.zz}, ul = V3<T>{IA.L.xz, IA.L.yz, IA.L.zz}, D = IA.L.zz, pz = pA.l.z;
            const T dinv = T(1) / D;                 // :1183
            const T u = taurow[di[0] * A.v_es] - pz; // :1200-1215
            st_store6<T>(S, sf, SV<T>{ua, ul});
            sw_st<T>(S, sf + 6, dinv);
            sw_st<T>(S, sf + 7, u);
            if (parent >= 0)
            {
               const SV<T> cj = st_load6<T>(S, mi[MI_SLOT_C]);
               const T ud = u * dinv;
               if (type == JT_REVOLUTE)
               {
                  rank1_down_revolute(Ia, ua, ul, dinv);           // :1220-1226
                  pa = pA + mul(Ia, cj) + SV<T>{ud * ua, ud * ul}; // :1229-1234
                  JX<T> jx;
                  jx.c = sw_ld<T>(S, mi[MI_SLOT_JP]), jx.s = sw_ld<T>(S, mi[MI_SLOT_JP] + 1), jx.d = T(0);
                  revolute_up(jx, load_xb<T>(c), Ia, pa); // :1156-1166; pa is now expressed in the parent's frame
                  handed_up = true;
               }
               else
               {
                  rank1_down(Ia, ua, ul, dinv);
                  pa = pA + mul(Ia, cj) + SV<T>{ud * ua, ud * ul};
               }
            }
         }
         else if (type == JT_PLANAR || type == JT_SPHERICAL)
         { // 3-DoF joint: U = IA S (6 x 3), D = S^T U (3 x 3), u = tau - S^T pA   (:1177-1215 with N = 3)
            const SV<T> U0 = mul(IA, unit_twist<T>(type, 0)), U1 = mul(IA, unit_twist<T>(type, 1)), U2 = mul(IA, unit_twist<T>(type, 2));
            const V3<T> d0 = comp3(type, U0), d1 = comp3(type, U1), d2 = comp3(type, U2);
            const S3<T> Di = spd3_inverse(S3<T>{d0.x, d0.y, d0.z, d1.y, d1.z, d2.z});
            const V3<T> tau3{taurow[di[0] * A.v_es], taurow[di[1] * A.v_es], taurow[di[2] * A.v_es]};
            const V3<T> u3 = tau3 - comp3(type, pA);
            const int sl = mi[MI_SLOT_LK];
            st_store6<T>(S, sl, U0), st_store6<T>(S, sl + 6, U1), st_store6<T>(S, sl + 12, U2);
            st_store6<T>(S, sl + 18, SV<T>{V3<T>{Di.xx, Di.xy, Di.xz}, V3<T>{Di.yy, Di.yz, Di.zz}});
            sw_st<T>(S, sl + 24, u3.x), sw_st<T>(S, sl + 25, u3.y), sw_st<T>(S, sl + 26, u3.z);
            if (parent >= 0)
            {
               const SV<T> W0 = Di.xx * U0 + Di.xy * U1 + Di.xz * U2, W1 = Di.xy * U0 + Di.yy * U1 + Di.yz * U2, W2 = Di.xz * U0 + Di.yz * U1 + Di.zz * U2;
               rank1_pair_down(Ia, W0, U0), rank1_pair_down(Ia, W1, U1), rank1_pair_down(Ia, W2, U2);
               const SV<T> cj = st_load6<T>(S, mi[MI_SLOT_C]);
               pa = pA + mul(Ia, cj) + u3.x * W0 + u3.y * W1 + u3.z * W2;
            }
         }
         else if (type == JT_SIXDOF)
         { // S = 1_6: pass three needs only x = IA^-1 u; for the parent Ia = 0 and pa = tau
            const SV<T> tau{V3<T>{taurow[di[0] * A.v_es], taurow[di[1] * A.v_es], taurow[di[2] * A.v_es]},
                            V3<T>{taurow[di[3] * A.v_es], taurow[di[4] * A.v_es], taurow[di[5] * A.v_es]}};
            const SV<T> x = spd6_solve(IA, tau - pA);
            st_store6<T>(S, sf, x);
            if (parent >= 0)
            {
               Ia.A = S3<T>{T(0), T(0), T(0), T(0), T(0), T(0)};
               Ia.L = Ia.A;
               Ia.C = M3<T>{T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
               pa = tau;
            }
         }
         // fixed joint: the whole articulated body is handed over unchanged (c = 0)
         if (parent >= 0)
         {
            SV<T> pp = pa;
            if (!handed_up)
            {
               const XF<T> Xb = load_xb<T>(c);
               const JX<T> jx = sw_joint_again<T>(S, type, cfg_map, mi[MI_CFG], qrow, A.q_es, mi[MI_SLOT_JP]);
               if (type != JT_SIXDOF) // (a floating joint transmits no inertia: Ia = 0 stays 0)
                  abi_up(type, jx, Xb, Ia); // :1156-1166
               pp = force_up(type, jx, Xb, pa);
            }
            if (xs >= 0)
            { // a limb root: through the exchange record
               st_store_abi<T>(S, xs, Ia);
               st_store6<T>(S, xs + 21, pp);
            }
            else if (flags & MF_PARENT_ADJ)
               Icarry = Ia, pcarry = pp, have_carry = true;
            else
            {
               ciptr pmi = meta + parent * MI_STRIDE;
               if (flags & MF_ACC_FIRST)
                  st_store_abi<T>(S, pmi[MI_SLOT_IA], Ia);
               else
               {
                  ABI<T> acc = st_load_abi<T>(S, pmi[MI_SLOT_IA]);
                  add(acc, Ia);
                  st_store_abi<T>(S, pmi[MI_SLOT_IA], acc);
               }
               st_add6<T>(S, pmi[MI_SLOT_F], pp);
            }
         }
         }; // kbody
         switch (type_rt)
         { // one dispatch on the joint kind per body step, straight-line code per kind (mh_dfs_kernels.h)
            case JT_REVOLUTE: kbody(std::integral_constant<int, JT_REVOLUTE>{}); break;
            case JT_PRISMATIC: kbody(std::integral_constant<int, JT_PRISMATIC>{}); break;
            case JT_SIXDOF: kbody(std::integral_constant<int, JT_SIXDOF>{}); break;
            case JT_PLANAR: kbody(std::integral_constant<int, JT_PLANAR>{}); break;
            case JT_SPHERICAL: kbody(std::integral_constant<int, JT_SPHERICAL>{}); break;
            default: kbody(std::integral_constant<int, JT_FIXED>{}); break;
         }
      };
      for (int s = n_seg - 1; s >= 0; s--)
      {
         const int j0 = seg[(wave * SPLIT_MAX_SEG + s) * 2], j1 = seg[(wave * SPLIT_MAX_SEG + s) * 2 + 1];
         have_carry = false;
         for (int j = j1 - 1; j >= j0; j--)
            pass2(j, 0, 0);
      }
      __syncthreads(); // the limbs' records are in the workgroup's block
      if (wave == 0)
      {
         have_carry = false;
         for (int k = P.n_trunk - 1; k >= 0; k--)
            pass2(trunk[k], xl_ofs[k], xl_ofs[k + 1]);
      }
      __syncthreads(); // the trunk's U, 1/D, u are in the block
      // ---- pass three (:1259-1310): joint accelerations, root to leaves
      SV<T> a_prev{Z, Z};
      auto pass3 = [&](int j, bool writes) {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type_rt = mi[MI_TYPE], flags = mi[MI_FLAGS];
         auto kbody = [&](auto kind) {
         const int type = kind;
         const CRef<T, false> c{CB + j * MC_STRIDE};
         SV<T> ap;
         if (parent < 0)
            ap = root_acceleration(A); // :259-264
         else if (flags & MF_PARENT_ADJ)
            ap = a_prev;
         else
            ap = st_load6<T>(S, meta[parent * MI_STRIDE + MI_SLOT_VA]);
         const XF<T> Xb = load_xb<T>(c);
         const JX<T> jx = sw_joint_again<T>(S, type, cfg_map, mi[MI_CFG], qrow, A.q_es, mi[MI_SLOT_JP]);
         SV<T> a = motion_down(type, jx, Xb, ap) + st_load6<T>(S, mi[MI_SLOT_C]); // :1270-1273
         const int sf = mi[MI_SLOT_F];
         ciptr di = dof_map + mi[MI_DOF];
         if (type == JT_REVOLUTE || type == JT_PRISMATIC)
         {
            const SV<T> U = st_load6<T>(S, sf);
            const T dinv = sw_ld<T>(S, sf + 6), u = sw_ld<T>(S, sf + 7);
            const T qdd = dinv * (u - (dot(U.a, a.a) + dot(U.l, a.l))); // :1280-1282
            if (writes)
               orow[di[0] * A.v_es] = qdd;
            if (type == JT_REVOLUTE)
               a.a.z += qdd;
            else
               a.l.z += qdd;
         }
         else if (type == JT_PLANAR || type == JT_SPHERICAL)
         {
            const int sl = mi[MI_SLOT_LK];
            const SV<T> U0 = st_load6<T>(S, sl), U1 = st_load6<T>(S, sl + 6), U2 = st_load6<T>(S, sl + 12);
            const SV<T> dv = st_load6<T>(S, sl + 18);
            const S3<T> Di{dv.a.x, dv.a.y, dv.a.z, dv.l.x, dv.l.y, dv.l.z};
            const V3<T> r{sw_ld<T>(S, sl + 24) - (dot(U0.a, a.a) + dot(U0.l, a.l)), sw_ld<T>(S, sl + 25) - (dot(U1.a, a.a) + dot(U1.l, a.l)),
                          sw_ld<T>(S, sl + 26) - (dot(U2.a, a.a) + dot(U2.l, a.l))};
            const V3<T> qdd = mul(Di, r);
            if (writes)
               orow[di[0] * A.v_es] = qdd.x, orow[di[1] * A.v_es] = qdd.y, orow[di[2] * A.v_es] = qdd.z;
            a = a + from_comp3(type, qdd);
         }
         else if (type == JT_SIXDOF)
         {
            const SV<T> x = st_load6<T>(S, sf);
            const SV<T> qdd = x - a;
            if (writes)
            {
               orow[di[0] * A.v_es] = qdd.a.x, orow[di[1] * A.v_es] = qdd.a.y, orow[di[2] * A.v_es] = qdd.a.z;
               orow[di[3] * A.v_es] = qdd.l.x, orow[di[4] * A.v_es] = qdd.l.y, orow[di[5] * A.v_es] = qdd.l.z;
            }
            a = x;
         }
         if (flags & MF_STORE_VA)
            st_store6<T>(S, mi[MI_SLOT_VA], a);
         if (writes && A.body_acc)
            store_body_motion<T>(c, A.body_acc + cfg * A.f_bs, A.f_es, mi[MI_EXT], a);
         a_prev = a;
         }; // kbody
         switch (type_rt)
         { // one dispatch on the joint kind per body step, straight-line code per kind (mh_dfs_kernels.h)
            case JT_REVOLUTE: kbody(std::integral_constant<int, JT_REVOLUTE>{}); break;
            case JT_PRISMATIC: kbody(std::integral_constant<int, JT_PRISMATIC>{}); break;
            case JT_SIXDOF: kbody(std::integral_constant<int, JT_SIXDOF>{}); break;
            case JT_PLANAR: kbody(std::integral_constant<int, JT_PLANAR>{}); break;
            case JT_SPHERICAL: kbody(std::integral_constant<int, JT_SPHERICAL>{}); break;
            default: kbody(std::integral_constant<int, JT_FIXED>{}); break;
         }
      };
      for (int k = 0; k < P.n_trunk; k++)
         pass3(trunk[k], active && wave == 0);
      for (int s = 0; s < n_seg; s++)
      {
         const int j0 = seg[(wave * SPLIT_MAX_SEG + s) * 2], j1 = seg[(wave * SPLIT_MAX_SEG + s) * 2 + 1];
         for (int j = j0; j < j1; j++)
            pass3(j, active);
      }
      __syncthreads(); // the next group of configurations re-uses the block
   }
}
template <typename T, int MODE>
__global__ void __launch_bounds__(256) rnea_split_kernel(Args<T> A, SplitDev P)
{
   rnea_split_body<T, MODE>(A, P, blockIdx.x, gridDim.x, blockIdx.x);
}
template <typename T, int MODE>
__global__ void __launch_bounds__(256) aba_split_kernel(Args<T> A, SplitDev P)
{
   aba_split_body<T, MODE>(A, P, blockIdx.x, gridDim.x, blockIdx.x);
}
// tau = RNEA(q, qd, A.in3) -> A.out and qdd = ABA(q, qd, A.in3b) -> A.outb of the same configurations in ONE launch: the first half of
// the grid runs the inverse dynamics, the second half the forward dynamics, each workgroup on a CU and a workspace block of its own.
// For small batches of a model without a code object (2 * groups <= CUs): two launches side by side on two streams cost their
// fork / join events (~15 us against kernels of 30 and 48 us on the humanoid at B = 4096).
// P.roles == 2: forward dynamics alone (A.in3 -> A.out) on the whole grid -- the fp64 mh_aba_f64 runs through THIS kernel too, so that
// the pair call and the single call execute the same machine code (two instantiations of the body differ in which products the
// compiler contracts into FMAs: last-bit differences between the two calls otherwise).
template <typename T, int MODE>
__global__ void __launch_bounds__(256) pair_split_kernel(Args<T> A, SplitDev P)
{
   const long half = P.roles == 2 ? 0 : gridDim.x / 2;
   if ((long)blockIdx.x < half)
      rnea_split_body<T, MODE>(A, P, blockIdx.x, half, blockIdx.x);
   else
   {
      Args<T> A2 = A;
      if (P.roles != 2)
         A2.in3 = A.in3b, A2.out = A.outb;
      aba_split_body<T, MODE>(A2, P, (long)blockIdx.x - half, (long)gridDim.x - half, blockIdx.x);
   }
}

// ============================================================================================ CRBA
// CompositeRigidBodyMassMatrixCalculator.java:588-707, 770-798.  The columns of a limb body need the joint transforms of its ancestors,
// not their composite inertias: every wave finishes ALL columns of its limbs' bodies (walking up through the trunk with the trunk's
// transforms, which every wave forms itself) before the barrier; a limb root leaves its composite inertia (10 values) in its exchange
// record; after the barrier wave 0 folds the trunk's composite inertias and writes the trunk bodies' columns.  H was zeroed by the caller.
template <typename T, int MODE>
__global__ void __launch_bounds__(256) crba_split_kernel(Args<T> A, SplitDev P)
{
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   const ciptr meta = as_const(P.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map);
   const ciptr trunk = as_const(P.trunk), seg = as_const(P.seg), xl_ofs = as_const(P.xl_ofs), xl = as_const(P.xl);
   const int tid = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   extern __shared__ double lds_raw[];
   const DStack<T, MODE> S{(dfs_lds_ptr<T>)lds_raw + tid, A.ws + (long)blockIdx.x * ((long)P.slots * 64) + tid};
   const int n_seg = P.n_seg[wave], nv = m.nv;
   const long groups = (A.B + 63) / 64;

   for (long grp = blockIdx.x; grp < groups; grp += gridDim.x)
   {
      const long cfg0 = grp * 64 + tid;
      const bool active = cfg0 < A.B;
      const long cfg = active ? cfg0 : A.B - 1;
      const T *qrow = A.q + cfg * A.q_bs;
      T *H = A.out + cfg * A.v_bs; // v_bs / v_es carry the per-configuration / per-entry strides of H here
      const long h_es = A.v_es;
      auto transform = [&](int j) {
         ciptr mi = meta + j * MI_STRIDE;
         (void)sw_joint_from_q<T>(S, mi[MI_TYPE], cfg_map, mi[MI_CFG], qrow, A.q_es, mi[MI_SLOT_JP]);
      };
      for (int k = 0; k < P.n_trunk; k++)
         transform(trunk[k]);
      for (int s = 0; s < n_seg; s++)
      {
         const int j0 = seg[(wave * SPLIT_MAX_SEG + s) * 2], j1 = seg[(wave * SPLIT_MAX_SEG + s) * 2 + 1];
         for (int j = j0; j < j1; j++)
            transform(j);
      }
      RI<T> rcarry;
      bool have_carry = false;
      auto body = [&](int j, int xk0, int xk1) {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type = mi[MI_TYPE], flags = mi[MI_FLAGS], xs = mi[MI_HAND];
         const CRef<T, false> c{CB + j * MC_STRIDE};
         RI<T> Ic = load_inertia<T>(c);
         if (have_carry)
            add(Ic, rcarry);
         if (flags & MF_HAS_ACC)
            add(Ic, sw_load_ri<T>(S, mi[MI_SLOT_IA]));
         for (int k = xk0; k < xk1; k++)
            add(Ic, sw_load_ri<T>(S, xl[k]));
         have_carry = false;
         const int nd = dof_count(type);
         ciptr dj = dof_map + mi[MI_DOF];
         const XF<T> Xb = load_xb<T>(c);
         const JX<T> jx = sw_joint_again<T>(S, type, cfg_map, mi[MI_CFG], qrow, A.q_es, mi[MI_SLOT_JP]);
         for (int k = 0; k < nd; k++)
         {
            SV<T> F = mul(Ic, unit_twist<T>(type, k)); // :663-667
            const int col = dj[k];
            if (active)
            { // diagonal block (:700-707)
               if (type == JT_REVOLUTE)
                  H[((long)col * nv + col) * h_es] = F.a.z;
               else if (type == JT_PRISMATIC)
                  H[((long)col * nv + col) * h_es] = F.l.z;
               else
                  for (int r = 0; r < nd; r++)
                     H[((long)dj[r] * nv + col) * h_es] = comp(F, dof_comp(type, r));
            }
            int prev = j, anc = parent; // ancestors (:783-792)
            XF<T> Xp = Xb;
            JX<T> jp = jx;
            int tp = type;
            while (anc >= 0)
            {
               F = force_up(tp, jp, Xp, F);
               ciptr ma = meta + anc * MI_STRIDE;
               const int ta = ma[MI_TYPE];
               ciptr da = dof_map + ma[MI_DOF];
               if (active)
               {
                  if (ta == JT_REVOLUTE)
                     H[((long)da[0] * nv + col) * h_es] = F.a.z, H[((long)col * nv + da[0]) * h_es] = F.a.z;
                  else if (ta == JT_PRISMATIC)
                     H[((long)da[0] * nv + col) * h_es] = F.l.z, H[((long)col * nv + da[0]) * h_es] = F.l.z;
                  else
                     for (int r = 0; r < dof_count(ta); r++)
                     {
                        const T hv = comp(F, dof_comp(ta, r));
                        H[((long)da[r] * nv + col) * h_es] = hv;
                        H[((long)col * nv + da[r]) * h_es] = hv;
                     }
               }
               prev = anc;
               anc = ma[MI_PARENT];
               if (anc >= 0)
               {
                  Xp = load_xb<T>(CRef<T, false>{CB + prev * MC_STRIDE});
                  jp = sw_joint_again<T>(S, ta, cfg_map, ma[MI_CFG], qrow, A.q_es, ma[MI_SLOT_JP]);
                  tp = ta;
               }
            }
         }
         if (parent >= 0)
         {
            rigid_up(type, jx, Xb, Ic); // :651-661
            if (xs >= 0)
               sw_store_ri<T>(S, xs, Ic); // a limb root: through the exchange record
            else if (flags & MF_PARENT_ADJ)
               rcarry = Ic, have_carry = true;
            else
            {
               const int sp = meta[parent * MI_STRIDE + MI_SLOT_IA];
               if (flags & MF_ACC_FIRST)
                  sw_store_ri<T>(S, sp, Ic);
               else
               {
                  RI<T> acc = sw_load_ri<T>(S, sp);
                  add(acc, Ic);
                  sw_store_ri<T>(S, sp, acc);
               }
            }
         }
      };
      for (int s = n_seg - 1; s >= 0; s--)
      {
         const int j0 = seg[(wave * SPLIT_MAX_SEG + s) * 2], j1 = seg[(wave * SPLIT_MAX_SEG + s) * 2 + 1];
         have_carry = false;
         for (int j = j1 - 1; j >= j0; j--)
            body(j, 0, 0);
      }
      __syncthreads(); // the limbs' composite inertias are in the workgroup's block
      if (wave == 0)
      {
         have_carry = false;
         for (int k = P.n_trunk - 1; k >= 0; k--)
            body(trunk[k], xl_ofs[k], xl_ofs[k + 1]);
      }
      __syncthreads(); // the next group of configurations re-uses the block
   }
}
} // namespace mh
