// mh_dfs_kernels.h -- run-time-topology RNEA / ABA as ONE depth-first walk per sweep, with a per-lane stack indexed by tree DEPTH.
//
// The first run-time-topology kernels (mh_kernels.h: rnea_kernel, aba_kernel) sweep the joint list forwards and backwards and keep what
// must survive between the sweeps in a per-lane workspace indexed by BODY: on the 128-body tree of BASELINE.json's configs[4] that is
// 1.3 GB written and read back once per launch (3.7x the algorithmic traffic), and every model without a specialised code object pays it.
// Here the walk follows the tree the way the recursion does (algorithms/InverseDynamicsCalculator.java:873-966,
// algorithms/ForwardDynamicsCalculator.java:1085-1310): a body is VISITed on the way down and POPped when its subtree is finished, so
// at any moment only the bodies on the current root-to-leaf path hold live state.
//
//   * the host compiles the tree into an EVENT PROGRAM (2 n words, scalar loads): VISIT(j) / POP(j) in depth-first order, with flags that
//     say where the operands are: in registers (the previous event left them there: a chain never touches memory) or in the stack
//     frame of the body / its parent;
//   * stack frames exist for non-leaf bodies only; their offsets are a function of the tree (sum of the ancestors' frame sizes), so the
//     stack a lane needs is (deepest path) x (frame), not (bodies) x (record): 146 slots instead of ~2500 on the 128-body tree.  It lives
//     in LDS (slot-major, [slot][64 lanes]: conflict-free) when a few waves per CU fit, else in a global slot-major workspace;
//   * RNEA needs nothing else.  ABA's inward sweep (passes one and two fused into the walk) hands U, 1/D, u' = u - U.c (and cos, sin, c
//     for bodies with children) per body to the outward sweep: 8..16 values per body instead of 30+, in LDS for small models, in the
//     global workspace otherwise.
//
// lane = configuration; state matrices are read per lane with the caller's strides (AoS rows or SoA columns alike: consecutive bodies
// of the walk read consecutive row entries, so every fetched line is consumed before it leaves L2 -- no transposed scratch copies).
// Same arithmetic primitives as the other kernels (mh_device.h, mh_kernels.h); per-body outputs, joint wrenches and
// acceleration-source joints stay on the sweep kernels of mh_kernels.h.
#pragma once
#include "mh_kernels.h"

namespace mh
{
// ---- event program
enum : int
{
   EV_POP = 1,          // else VISIT
   EV_PARENT_REGS = 2,  // VISIT: the previous event was VISIT(parent): its state is in registers
   EV_LEAF = 2,         // POP: the previous event was VISIT(this body): its state is in registers
   EV_LAST_CHILD = 4,   // POP: next event is POP(parent): hand the contribution over in registers
   EV_ACC_FIRST = 8,    // POP (ABA): first contribution to the parent's accumulator: store, do not add
   EV_BODY_SHIFT = 8
};
// stack-frame slots a joint transform takes (revolute: cos, sin; prismatic: q; fixed: nothing)
__host__ __device__ constexpr int jx_slots(int type) { return type == JT_REVOLUTE ? 2 : (type == JT_PRISMATIC ? 1 : (general_x(type) ? 12 : 0)); }
__host__ __device__ constexpr int rnea_frame_slots(int type, int n_children)
{ // [f 6][jx][v, a 12 when later children re-read them]
   return n_children == 0 ? 0 : 6 + jx_slots(type) + (n_children >= 2 ? 12 : 0);
}
__host__ __device__ constexpr int aba_frame_slots(int type, int n_children)
{ // [p 6][c 6][jx][v 6, articulated-inertia accumulator 21 + bias accumulator 6 when several children contribute]
   return n_children == 0 ? 0 : 12 + jx_slots(type) + (n_children >= 2 ? 33 : 0);
}
__host__ __device__ constexpr int aba_hand_slots(int type, int n_children)
{ // inward -> outward hand-over; + the bias acceleration c (or the body acceleration of a 6-DoF joint) when children need this body's a
   const int own = type == JT_REVOLUTE ? 10 : (type == JT_PRISMATIC ? 8 : (type == JT_SIXDOF ? 6 : (type == JT_FIXED ? 0 : 27)));
   return own + ((n_children > 0 && type != JT_FIXED) ? 6 : 0);
}

#define MH_ST(slot) st[(long)(slot)*ss]
template <typename T, class P>
MH_DEV void st_store6(P st, long ss, int slot, const SV<T> &v)
{
   MH_ST(slot + 0) = v.a.x, MH_ST(slot + 1) = v.a.y, MH_ST(slot + 2) = v.a.z, MH_ST(slot + 3) = v.l.x, MH_ST(slot + 4) = v.l.y, MH_ST(slot + 5) = v.l.z;
}
template <typename T, class P>
MH_DEV SV<T> st_load6(P st, long ss, int slot)
{
   return SV<T>{V3<T>{MH_ST(slot + 0), MH_ST(slot + 1), MH_ST(slot + 2)}, V3<T>{MH_ST(slot + 3), MH_ST(slot + 4), MH_ST(slot + 5)}};
}
template <typename T, class P>
MH_DEV void st_store_jx(P st, long ss, int slot, int type, const JX<T> &jx)
{
   if (type == JT_REVOLUTE)
      MH_ST(slot) = jx.c, MH_ST(slot + 1) = jx.s;
   else if (type == JT_PRISMATIC)
      MH_ST(slot) = jx.d;
   else if (general_x(type))
   {
      MH_ST(slot + 0) = jx.X.R.xx, MH_ST(slot + 1) = jx.X.R.xy, MH_ST(slot + 2) = jx.X.R.xz, MH_ST(slot + 3) = jx.X.R.yx, MH_ST(slot + 4) = jx.X.R.yy;
      MH_ST(slot + 5) = jx.X.R.yz, MH_ST(slot + 6) = jx.X.R.zx, MH_ST(slot + 7) = jx.X.R.zy, MH_ST(slot + 8) = jx.X.R.zz;
      MH_ST(slot + 9) = jx.X.p.x, MH_ST(slot + 10) = jx.X.p.y, MH_ST(slot + 11) = jx.X.p.z;
   }
}
template <typename T, class P>
MH_DEV JX<T> st_load_jx(P st, long ss, int slot, int type)
{
   JX<T> jx;
   jx.c = T(1), jx.s = T(0), jx.d = T(0);
   if (type == JT_REVOLUTE)
      jx.c = MH_ST(slot), jx.s = MH_ST(slot + 1);
   else if (type == JT_PRISMATIC)
      jx.d = MH_ST(slot);
   else if (general_x(type))
   {
      jx.X.R = M3<T>{MH_ST(slot + 0), MH_ST(slot + 1), MH_ST(slot + 2), MH_ST(slot + 3), MH_ST(slot + 4), MH_ST(slot + 5), MH_ST(slot + 6), MH_ST(slot + 7), MH_ST(slot + 8)};
      jx.X.p = V3<T>{MH_ST(slot + 9), MH_ST(slot + 10), MH_ST(slot + 11)};
   }
   return jx;
}
template <typename T, class P>
MH_DEV void st_store_abi(P st, long ss, int s, const ABI<T> &I)
{
   MH_ST(s + 0) = I.A.xx, MH_ST(s + 1) = I.A.xy, MH_ST(s + 2) = I.A.xz, MH_ST(s + 3) = I.A.yy, MH_ST(s + 4) = I.A.yz, MH_ST(s + 5) = I.A.zz;
   MH_ST(s + 6) = I.L.xx, MH_ST(s + 7) = I.L.xy, MH_ST(s + 8) = I.L.xz, MH_ST(s + 9) = I.L.yy, MH_ST(s + 10) = I.L.yz, MH_ST(s + 11) = I.L.zz;
   MH_ST(s + 12) = I.C.xx, MH_ST(s + 13) = I.C.xy, MH_ST(s + 14) = I.C.xz, MH_ST(s + 15) = I.C.yx, MH_ST(s + 16) = I.C.yy, MH_ST(s + 17) = I.C.yz;
   MH_ST(s + 18) = I.C.zx, MH_ST(s + 19) = I.C.zy, MH_ST(s + 20) = I.C.zz;
}
template <typename T, class P>
MH_DEV ABI<T> st_load_abi(P st, long ss, int s)
{
   ABI<T> I;
   I.A = S3<T>{MH_ST(s + 0), MH_ST(s + 1), MH_ST(s + 2), MH_ST(s + 3), MH_ST(s + 4), MH_ST(s + 5)};
   I.L = S3<T>{MH_ST(s + 6), MH_ST(s + 7), MH_ST(s + 8), MH_ST(s + 9), MH_ST(s + 10), MH_ST(s + 11)};
   I.C = M3<T>{MH_ST(s + 12), MH_ST(s + 13), MH_ST(s + 14), MH_ST(s + 15), MH_ST(s + 16), MH_ST(s + 17), MH_ST(s + 18), MH_ST(s + 19), MH_ST(s + 20)};
   return I;
}

// ---- inputs of one body, fetched ONE EVENT AHEAD of their use.  The walk is a chain of dependent latencies otherwise -- scalar load of
// the body's record, scalar load of its matrix rows, per-lane global load, only then arithmetic -- and with one wave per SIMD (B = 4096)
// or a global stack nothing else hides them: measured 2.5-5 us per body before, the arithmetic itself is ~0.5 us.
template <typename T>
struct In
{ // the entries of a 1-DoF joint (all but a few joints of any robot): 3 registers per buffer.  Multi-DoF joints (a floating base, the
  // odd spherical joint) read theirs when they are reached -- prefetching their 7 + 6 + 6 entries as well cost 100 fp64 registers and
  // 82 spills in the ABA kernel.
   T q, v, x;
};
__host__ __device__ constexpr bool one_dof(int type) { return type == JT_REVOLUTE || type == JT_PRISMATIC; }
// what: 1 = configuration entry, 2 = velocity entry, 4 = third matrix (qdd | tau)
template <typename T>
MH_DEV void fetch_inputs(In<T> &o, int what, int type, ciptr ci, ciptr di, const T *qrow, long q_es, const T *vrow, const T *xrow, long v_es)
{
   if (!one_dof(type))
      return;
   if (what & 1)
      o.q = qrow[ci[0] * q_es];
   if (what & 6)
   {
      const long r = di[0] * v_es;
      if (what & 2)
         o.v = vrow[r];
      if (what & 4)
         o.x = xrow[r];
   }
}
// joint transform of one lane: 1-DoF joints from the fetched entry, the others from q (same arithmetic as joint_from_q)
template <typename T>
MH_DEV JX<T> joint_of_in(int type, const In<T> &in, ciptr cfg_map, int cfg_ofs, const T *qrow, long q_es)
{
   if (!one_dof(type))
      return joint_from_q<T>(type, cfg_map, cfg_ofs, qrow, q_es, (T *)nullptr, 0, 0, false);
   JX<T> jx;
   jx.c = T(1), jx.s = T(0), jx.d = T(0);
   if (type == JT_REVOLUTE)
      sincos_t(in.q, jx.s, jx.c);
   else
      jx.d = in.q;
   return jx;
}
// S x : 1-DoF joints from the fetched entry, the others from the matrix row
template <typename T>
MH_DEV SV<T> joint_vec_in(int type, T x1, ciptr dof_map, int dof_ofs, const T *row, long es, bool enabled)
{
   if (!one_dof(type))
      return joint_vec<T>(type, dof_map, dof_ofs, row, es, enabled);
   SV<T> o{V3<T>{T(0), T(0), T(0)}, V3<T>{T(0), T(0), T(0)}};
   if (enabled)
   {
      if (type == JT_REVOLUTE)
         o.a.z = x1;
      else
         o.l.z = x1;
   }
   return o;
}
template <typename T>
MH_DEV void write_joint_rows(int type, ciptr di, T *row, long es, const SV<T> &f)
{ // tau = S^T f : component picks in the canonical joint frames
   if (type == JT_REVOLUTE)
      row[di[0] * es] = f.a.z;
   else if (type == JT_PRISMATIC)
      row[di[0] * es] = f.l.z;
   else if (type == JT_SIXDOF)
   {
      row[di[0] * es] = f.a.x, row[di[1] * es] = f.a.y, row[di[2] * es] = f.a.z;
      row[di[3] * es] = f.l.x, row[di[4] * es] = f.l.y, row[di[5] * es] = f.l.z;
   }
   else if (type == JT_PLANAR)
      row[di[0] * es] = f.a.y, row[di[1] * es] = f.l.x, row[di[2] * es] = f.l.z;
   else if (type == JT_SPHERICAL)
      row[di[0] * es] = f.a.x, row[di[1] * es] = f.a.y, row[di[2] * es] = f.a.z;
}

// ============================================================================================ RNEA
// STK_LDS: the depth stack lives in LDS ([slot][64]), else in the global workspace A.ws ([slot][A.ws_stride]).
template <typename T, bool STK_LDS>
__global__ void __launch_bounds__(64) rnea_dfs_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   const ciptr meta = as_const(m.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map), prog = as_const(m.prog);
   const long lane = (long)blockIdx.x * 64 + threadIdx.x;
   const long nlanes = (long)gridDim.x * 64;
   const V3<T> Z{T(0), T(0), T(0)};
   auto walk = [&](auto st, const long ss) {
      for (long cfg = lane; cfg < A.B; cfg += nlanes)
      {
         const T *qrow = A.q + cfg * A.q_bs;
         const T *qdrow = A.qd + cfg * A.v_bs;
         const T *qddrow = A.in3 + cfg * A.v_bs;
         const T *frow = A.fext ? A.fext + cfg * A.f_bs : nullptr;
         T *trow = A.out + cfg * A.v_bs;
         SV<T> v_reg{Z, Z}, a_reg{Z, Z}, f_reg{Z, Z}, carry{Z, Z};
         JX<T> jx_reg;
         jx_reg.c = T(1), jx_reg.s = T(0), jx_reg.d = T(0);
         In<T> nxt;
         auto prefetch = [&](int e1) { // the inputs event e1 will consume (a VISIT of a 1-DoF joint: q, qd, qdd; a POP: nothing)
            if (e1 >= m.n_events)
               return;
            const int ev1 = prog[e1];
            if (ev1 & EV_POP)
               return;
            ciptr m1 = meta + (ev1 >> EV_BODY_SHIFT) * MI_STRIDE;
            fetch_inputs<T>(nxt, 1 | (A.coriolis ? 2 : 0) | (A.accel ? 4 : 0), m1[MI_TYPE], cfg_map + m1[MI_CFG], dof_map + m1[MI_DOF], qrow, A.q_es,
                            qdrow, qddrow, A.v_es);
         };
         prefetch(0);
         for (int e = 0; e < m.n_events; e++)
         {
            const int ev = prog[e];
            const int j = ev >> EV_BODY_SHIFT;
            ciptr mi = meta + j * MI_STRIDE;
            const int parent = mi[MI_PARENT], type = mi[MI_TYPE], nch = mi[MI_NCH], fr = mi[MI_DFS_R];
            const CRef<T, false> c{CB + j * MC_STRIDE};
            const XF<T> Xb = load_xb<T>(c);
            const In<T> in = nxt;
            prefetch(e + 1);
            if (!(ev & EV_POP))
            { // ---- VISIT: velocity, acceleration, Newton-Euler wrench of the body (InverseDynamicsCalculator.java:873-917)
               SV<T> vp, ap;
               if (parent < 0)
               {
                  vp = SV<T>{Z, Z};
                  ap = SV<T>{Z, V3<T>{-A.gx, -A.gy, -A.gz}}; // :343-348
               }
               else if (ev & EV_PARENT_REGS)
                  vp = v_reg, ap = a_reg;
               else
               {
                  ciptr mp = meta + parent * MI_STRIDE;
                  const int va = mp[MI_DFS_R] + 6 + jx_slots(mp[MI_TYPE]);
                  vp = st_load6<T>(st, ss, va), ap = st_load6<T>(st, ss, va + 6);
               }
               const JX<T> jx = joint_of_in<T>(type, in, cfg_map, mi[MI_CFG], qrow, A.q_es);
               const SV<T> vJ = joint_vec_in<T>(type, in.v, dof_map, mi[MI_DOF], qdrow, A.v_es, A.coriolis != 0);
               const SV<T> aJ = joint_vec_in<T>(type, in.x, dof_map, mi[MI_DOF], qddrow, A.v_es, A.accel != 0);
               SV<T> v = motion_down(type, jx, Xb, vp) + vJ;
               const SV<T> a = motion_down(type, jx, Xb, ap) + aJ + crm(v, vJ);
               if (!A.coriolis)
                  v = SV<T>{Z, Z};
               const RI<T> I = load_inertia<T>(c);
               SV<T> f = mul(I, a) + crf(v, mul(I, v));
               if (frow)
                  f = f - load_fext<T>(c, frow, A.f_es, mi[MI_EXT]);
               if (nch >= 1)
               { // children follow: park what POP needs
                  st_store6<T>(st, ss, fr, f);
                  st_store_jx<T>(st, ss, fr + 6, type, jx);
                  if (nch >= 2)
                     st_store6<T>(st, ss, fr + 6 + jx_slots(type), v), st_store6<T>(st, ss, fr + 12 + jx_slots(type), a);
               }
               v_reg = v, a_reg = a, f_reg = f, jx_reg = jx;
            }
            else
            { // ---- POP: the subtree is complete: joint effort, wrench handed to the parent (:930-966)
               SV<T> f;
               JX<T> jx;
               if (ev & EV_LEAF)
                  f = f_reg, jx = jx_reg;
               else
               {
                  f = st_load6<T>(st, ss, fr) + carry;
                  jx = st_load_jx<T>(st, ss, fr + 6, type);
               }
               write_joint_rows<T>(type, dof_map + mi[MI_DOF], trow, A.v_es, f);
               if (parent >= 0)
               {
                  const SV<T> fp = force_up(type, jx, Xb, f);
                  if (ev & EV_LAST_CHILD)
                     carry = fp;
                  else
                  {
                     const int pf = meta[parent * MI_STRIDE + MI_DFS_R];
                     st_store6<T>(st, ss, pf, st_load6<T>(st, ss, pf) + fp);
                  }
               }
            }
         }
      }
   };
   if constexpr (STK_LDS)
      walk((T *)lds_raw + threadIdx.x, 64L);
   else
      walk(A.ws + lane, A.ws_stride);
}

// ============================================================================================ ABA
// STK_LDS / HND_LDS: where the depth stack and the inward -> outward hand-over live (LDS, stack first; or the global workspace, stack first).
template <typename T, bool STK_LDS, bool HND_LDS>
__global__ void __launch_bounds__(64) aba_dfs_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   const ciptr meta = as_const(m.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map), prog = as_const(m.prog);
   const long lane = (long)blockIdx.x * 64 + threadIdx.x;
   const long nlanes = (long)gridDim.x * 64;
   const V3<T> Z{T(0), T(0), T(0)};
   auto walk = [&](auto st, const long ss, auto hd, const long hs) {
#define MH_HD(slot) hd[(long)(slot)*hs]
      for (long cfg = lane; cfg < A.B; cfg += nlanes)
      {
         const T *qrow = A.q + cfg * A.q_bs;
         const T *qdrow = A.qd + cfg * A.v_bs;
         const T *taurow = A.in3 + cfg * A.v_bs;
         const T *frow = A.fext ? A.fext + cfg * A.f_bs : nullptr;
         T *orow = A.out + cfg * A.v_bs;
         // ---- inward part: passes one and two (ForwardDynamicsCalculator.java:1085-1254) fused into one depth-first walk
         SV<T> v_reg{Z, Z}, p_reg{Z, Z}, c_reg{Z, Z}, pcarry{Z, Z};
         JX<T> jx_reg;
         jx_reg.c = T(1), jx_reg.s = T(0), jx_reg.d = T(0);
         ABI<T> Icarry = abi_from_rigid(RI<T>{T(0), Z, S3<T>{T(0), T(0), T(0), T(0), T(0), T(0)}});
         In<T> nxt;
         auto prefetch = [&](int e1) { // the inputs event e1 will consume (1-DoF joints): a VISIT q and qd, a POP the joint's effort
            if (e1 >= m.n_events)
               return;
            const int ev1 = prog[e1];
            ciptr m1 = meta + (ev1 >> EV_BODY_SHIFT) * MI_STRIDE;
            fetch_inputs<T>(nxt, (ev1 & EV_POP) ? 4 : (1 | 2), m1[MI_TYPE], cfg_map + m1[MI_CFG], dof_map + m1[MI_DOF], qrow, A.q_es, qdrow, taurow, A.v_es);
         };
         prefetch(0);
         for (int e = 0; e < m.n_events; e++)
         {
            const int ev = prog[e];
            const int j = ev >> EV_BODY_SHIFT;
            ciptr mi = meta + j * MI_STRIDE;
            const int parent = mi[MI_PARENT], type = mi[MI_TYPE], nch = mi[MI_NCH], fr = mi[MI_DFS_A], hf = mi[MI_HAND];
            const int jxs = jx_slots(type);
            const CRef<T, false> c{CB + j * MC_STRIDE};
            const XF<T> Xb = load_xb<T>(c);
            const In<T> in = nxt;
            prefetch(e + 1);
            if (!(ev & EV_POP))
            { // ---- VISIT (:1085-1127): velocity, bias wrench p, bias acceleration c
               SV<T> vp;
               if (parent < 0)
                  vp = SV<T>{Z, Z};
               else if (ev & EV_PARENT_REGS)
                  vp = v_reg;
               else
               {
                  ciptr mp = meta + parent * MI_STRIDE;
                  vp = st_load6<T>(st, ss, mp[MI_DFS_A] + 12 + jx_slots(mp[MI_TYPE]));
               }
               const JX<T> jx = joint_of_in<T>(type, in, cfg_map, mi[MI_CFG], qrow, A.q_es);
               const SV<T> vJ = joint_vec_in<T>(type, in.v, dof_map, mi[MI_DOF], qdrow, A.v_es, true);
               const SV<T> v = motion_down(type, jx, Xb, vp) + vJ;
               const RI<T> I = load_inertia<T>(c);
               SV<T> p = crf(v, mul(I, v));
               if (frow)
                  p = p - load_fext<T>(c, frow, A.f_es, mi[MI_EXT]);
               const SV<T> cj = crm(v, vJ);
               if (nch >= 1)
               {
                  st_store6<T>(st, ss, fr, p), st_store6<T>(st, ss, fr + 6, cj);
                  st_store_jx<T>(st, ss, fr + 12, type, jx);
                  if (nch >= 2)
                     st_store6<T>(st, ss, fr + 12 + jxs, v);
               }
               v_reg = v, p_reg = p, c_reg = cj, jx_reg = jx;
            }
            else
            { // ---- POP (:1136-1254): articulated inertia and bias wrench of the finished subtree, joint-space quantities, hand-up
               ABI<T> IA = abi_from_rigid(load_inertia<T>(c));
               SV<T> pA, cj;
               JX<T> jx;
               if (ev & EV_LEAF)
                  pA = p_reg, cj = c_reg, jx = jx_reg;
               else
               {
                  pA = st_load6<T>(st, ss, fr) + pcarry, cj = st_load6<T>(st, ss, fr + 6);
                  jx = st_load_jx<T>(st, ss, fr + 12, type);
                  add(IA, Icarry);
                  if (nch >= 2)
                  {
                     add(IA, st_load_abi<T>(st, ss, fr + 18 + jxs));
                     pA = pA + st_load6<T>(st, ss, fr + 39 + jxs);
                  }
               }
               ABI<T> Ia = IA;
               SV<T> pa = pA;
               if (type == JT_REVOLUTE || type == JT_PRISMATIC)
               {
                  V3<T> ua, ul;
                  T D, pz;
                  if (type == JT_REVOLUTE)
                     ua = V3<T>{IA.A.xz, IA.A.yz, IA.A.zz}, ul = V3<T>{IA.C.zx, IA.C.zy, IA.C.zz}, D = IA.A.zz, pz = pA.a.z;
                  else
                     ua = V3<T>{IA.C.xz, IA.C.yz, IA.C.zz}, ul = V3<T>{IA.L.xz, IA.L.yz, IA.L.zz}, D = IA.L.zz, pz = pA.l.z;
                  const T dinv = T(1) / D;                 // :1183
                  const T u = in.x - pz;                   // :1200-1215
                  MH_HD(hf + 0) = ua.x, MH_HD(hf + 1) = ua.y, MH_HD(hf + 2) = ua.z, MH_HD(hf + 3) = ul.x, MH_HD(hf + 4) = ul.y, MH_HD(hf + 5) = ul.z;
                  MH_HD(hf + 6) = dinv;
                  MH_HD(hf + 7) = u - (dot(ua, cj.a) + dot(ul, cj.l)); // u' = u - U.c: the outward sweep then needs U.(X a_parent) only
                  int hn = hf + 8;
                  if (type == JT_REVOLUTE)
                     MH_HD(hf + 8) = jx.c, MH_HD(hf + 9) = jx.s, hn = hf + 10;
                  if (nch >= 1)
                     MH_HD(hn + 0) = cj.a.x, MH_HD(hn + 1) = cj.a.y, MH_HD(hn + 2) = cj.a.z, MH_HD(hn + 3) = cj.l.x, MH_HD(hn + 4) = cj.l.y, MH_HD(hn + 5) = cj.l.z;
                  if (parent >= 0)
                  {
                     if (type == JT_REVOLUTE)
                        rank1_down_revolute(Ia, ua, ul, dinv); // :1220-1226 (Ia S = 0: exact structural zeros)
                     else
                        rank1_down(Ia, ua, ul, dinv);
                     const T ud = u * dinv;
                     pa = pA + mul(Ia, cj) + SV<T>{ud * ua, ud * ul}; // :1229-1234
                  }
               }
               else if (type == JT_PLANAR || type == JT_SPHERICAL)
               { // 3-DoF joint (:1177-1234 with N = 3)
                  const SV<T> U0 = mul(IA, unit_twist<T>(type, 0)), U1 = mul(IA, unit_twist<T>(type, 1)), U2 = mul(IA, unit_twist<T>(type, 2));
                  const V3<T> d0 = comp3(type, U0), d1 = comp3(type, U1), d2 = comp3(type, U2);
                  const S3<T> Di = spd3_inverse(S3<T>{d0.x, d0.y, d0.z, d1.y, d1.z, d2.z});
                  ciptr di = dof_map + mi[MI_DOF];
                  const V3<T> tau3{taurow[di[0] * A.v_es], taurow[di[1] * A.v_es], taurow[di[2] * A.v_es]};
                  const V3<T> u3 = tau3 - comp3(type, pA);
                  const SV<T> Us[3] = {U0, U1, U2};
#pragma unroll
                  for (int k = 0; k < 3; k++)
                  {
                     MH_HD(hf + 6 * k + 0) = Us[k].a.x, MH_HD(hf + 6 * k + 1) = Us[k].a.y, MH_HD(hf + 6 * k + 2) = Us[k].a.z;
                     MH_HD(hf + 6 * k + 3) = Us[k].l.x, MH_HD(hf + 6 * k + 4) = Us[k].l.y, MH_HD(hf + 6 * k + 5) = Us[k].l.z;
                  }
                  MH_HD(hf + 18) = Di.xx, MH_HD(hf + 19) = Di.xy, MH_HD(hf + 20) = Di.xz, MH_HD(hf + 21) = Di.yy, MH_HD(hf + 22) = Di.yz, MH_HD(hf + 23) = Di.zz;
                  MH_HD(hf + 24) = u3.x - (dot(U0.a, cj.a) + dot(U0.l, cj.l));
                  MH_HD(hf + 25) = u3.y - (dot(U1.a, cj.a) + dot(U1.l, cj.l));
                  MH_HD(hf + 26) = u3.z - (dot(U2.a, cj.a) + dot(U2.l, cj.l));
                  if (nch >= 1)
                     MH_HD(hf + 27) = cj.a.x, MH_HD(hf + 28) = cj.a.y, MH_HD(hf + 29) = cj.a.z, MH_HD(hf + 30) = cj.l.x, MH_HD(hf + 31) = cj.l.y, MH_HD(hf + 32) = cj.l.z;
                  if (parent >= 0)
                  {
                     const SV<T> W0 = Di.xx * U0 + Di.xy * U1 + Di.xz * U2, W1 = Di.xy * U0 + Di.yy * U1 + Di.yz * U2, W2 = Di.xz * U0 + Di.yz * U1 + Di.zz * U2;
                     rank1_pair_down(Ia, W0, U0), rank1_pair_down(Ia, W1, U1), rank1_pair_down(Ia, W2, U2);
                     pa = pA + mul(Ia, cj) + u3.x * W0 + u3.y * W1 + u3.z * W2;
                  }
               }
               else if (type == JT_SIXDOF)
               { // S = 1_6: x = IA^-1 (tau - pA) is the body acceleration; for the parent Ia = 0, pa = tau
                  const SV<T> tau = joint_vec<T>(type, dof_map, mi[MI_DOF], taurow, A.v_es, true);
                  const SV<T> x = spd6_solve(IA, tau - pA);
                  const SV<T> xc = x - cj; // qdd = x - (X a_parent + c)
                  MH_HD(hf + 0) = xc.a.x, MH_HD(hf + 1) = xc.a.y, MH_HD(hf + 2) = xc.a.z, MH_HD(hf + 3) = xc.l.x, MH_HD(hf + 4) = xc.l.y, MH_HD(hf + 5) = xc.l.z;
                  if (nch >= 1)
                     MH_HD(hf + 6) = x.a.x, MH_HD(hf + 7) = x.a.y, MH_HD(hf + 8) = x.a.z, MH_HD(hf + 9) = x.l.x, MH_HD(hf + 10) = x.l.y, MH_HD(hf + 11) = x.l.z;
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
                  if (type == JT_REVOLUTE)
                     revolute_up(jx, Xb, Ia, pp);
                  else
                  {
                     abi_up(type, jx, Xb, Ia); // :1156-1166
                     pp = force_up(type, jx, Xb, pa);
                  }
                  if (ev & EV_LAST_CHILD)
                     Icarry = Ia, pcarry = pp;
                  else
                  {
                     ciptr mp = meta + parent * MI_STRIDE;
                     const int acc = mp[MI_DFS_A] + 18 + jx_slots(mp[MI_TYPE]);
                     if (ev & EV_ACC_FIRST)
                        st_store_abi<T>(st, ss, acc, Ia), st_store6<T>(st, ss, acc + 21, pp);
                     else
                     {
                        ABI<T> s = st_load_abi<T>(st, ss, acc);
                        add(s, Ia);
                        st_store_abi<T>(st, ss, acc, s);
                        st_store6<T>(st, ss, acc + 21, st_load6<T>(st, ss, acc + 21) + pp);
                     }
                  }
               }
            }
         }
         // ---- outward part: pass three (:1259-1310), joint accelerations root to leaves
         SV<T> a_reg{Z, Z};
         auto prefetch_q = [&](int j1) { // revolute joints take (cos, sin) from the hand-over; prismatic ones re-read q (fetched ahead)
            if (j1 >= m.n)
               return;
            ciptr m1 = meta + j1 * MI_STRIDE;
            if (m1[MI_TYPE] == JT_PRISMATIC)
               fetch_inputs<T>(nxt, 1, m1[MI_TYPE], cfg_map + m1[MI_CFG], dof_map + m1[MI_DOF], qrow, A.q_es, qdrow, taurow, A.v_es);
         };
         prefetch_q(0);
         for (int j = 0; j < m.n; j++)
         {
            ciptr mi = meta + j * MI_STRIDE;
            const int parent = mi[MI_PARENT], type = mi[MI_TYPE], nch = mi[MI_NCH], fr = mi[MI_DFS_A], hf = mi[MI_HAND];
            const CRef<T, false> c{CB + j * MC_STRIDE};
            const In<T> in = nxt;
            prefetch_q(j + 1);
            SV<T> ap;
            if (parent < 0)
               ap = SV<T>{Z, V3<T>{-A.gx, -A.gy, -A.gz}}; // :259-264
            else if (parent == j - 1)
               ap = a_reg;
            else
               ap = st_load6<T>(st, ss, meta[parent * MI_STRIDE + MI_DFS_A]);
            JX<T> jx;
            if (type == JT_REVOLUTE)
               jx.c = MH_HD(hf + 8), jx.s = MH_HD(hf + 9), jx.d = T(0);
            else
               jx = joint_of_in<T>(type, in, cfg_map, mi[MI_CFG], qrow, A.q_es);
            const SV<T> apx = motion_down(type, jx, load_xb<T>(c), ap);
            ciptr di = dof_map + mi[MI_DOF];
            SV<T> a = apx;
            if (type == JT_REVOLUTE || type == JT_PRISMATIC)
            {
               const V3<T> ua{MH_HD(hf + 0), MH_HD(hf + 1), MH_HD(hf + 2)}, ul{MH_HD(hf + 3), MH_HD(hf + 4), MH_HD(hf + 5)};
               const T qdd = MH_HD(hf + 6) * (MH_HD(hf + 7) - (dot(ua, apx.a) + dot(ul, apx.l))); // :1280-1282 with u' = u - U.c
               orow[di[0] * A.v_es] = qdd;
               if (nch >= 1)
               {
                  const int hn = hf + (type == JT_REVOLUTE ? 10 : 8);
                  a = apx + SV<T>{V3<T>{MH_HD(hn + 0), MH_HD(hn + 1), MH_HD(hn + 2)}, V3<T>{MH_HD(hn + 3), MH_HD(hn + 4), MH_HD(hn + 5)}};
                  if (type == JT_REVOLUTE)
                     a.a.z += qdd;
                  else
                     a.l.z += qdd;
               }
            }
            else if (type == JT_PLANAR || type == JT_SPHERICAL)
            {
               V3<T> r;
               T rr[3];
#pragma unroll
               for (int k = 0; k < 3; k++)
               {
                  const V3<T> ua{MH_HD(hf + 6 * k + 0), MH_HD(hf + 6 * k + 1), MH_HD(hf + 6 * k + 2)}, ul{MH_HD(hf + 6 * k + 3), MH_HD(hf + 6 * k + 4), MH_HD(hf + 6 * k + 5)};
                  rr[k] = MH_HD(hf + 24 + k) - (dot(ua, apx.a) + dot(ul, apx.l));
               }
               r = V3<T>{rr[0], rr[1], rr[2]};
               const S3<T> Di{MH_HD(hf + 18), MH_HD(hf + 19), MH_HD(hf + 20), MH_HD(hf + 21), MH_HD(hf + 22), MH_HD(hf + 23)};
               const V3<T> qdd = mul(Di, r);
               orow[di[0] * A.v_es] = qdd.x, orow[di[1] * A.v_es] = qdd.y, orow[di[2] * A.v_es] = qdd.z;
               if (nch >= 1)
                  a = apx + SV<T>{V3<T>{MH_HD(hf + 27), MH_HD(hf + 28), MH_HD(hf + 29)}, V3<T>{MH_HD(hf + 30), MH_HD(hf + 31), MH_HD(hf + 32)}} + from_comp3(type, qdd);
            }
            else if (type == JT_SIXDOF)
            {
               const SV<T> xc{V3<T>{MH_HD(hf + 0), MH_HD(hf + 1), MH_HD(hf + 2)}, V3<T>{MH_HD(hf + 3), MH_HD(hf + 4), MH_HD(hf + 5)}};
               const SV<T> qdd = xc - apx;
               orow[di[0] * A.v_es] = qdd.a.x, orow[di[1] * A.v_es] = qdd.a.y, orow[di[2] * A.v_es] = qdd.a.z;
               orow[di[3] * A.v_es] = qdd.l.x, orow[di[4] * A.v_es] = qdd.l.y, orow[di[5] * A.v_es] = qdd.l.z;
               if (nch >= 1)
                  a = SV<T>{V3<T>{MH_HD(hf + 6), MH_HD(hf + 7), MH_HD(hf + 8)}, V3<T>{MH_HD(hf + 9), MH_HD(hf + 10), MH_HD(hf + 11)}};
            }
            if (nch >= 2)
               st_store6<T>(st, ss, fr, a); // later children re-read it
            a_reg = a;
         }
      }
#undef MH_HD
   };
   T *const lds = (T *)lds_raw + threadIdx.x;
   T *const glb = A.ws + lane;
   if constexpr (STK_LDS && HND_LDS)
      walk(lds, 64L, lds + (long)m.aba_stack * 64, 64L);
   else if constexpr (STK_LDS)
      walk(lds, 64L, glb, A.ws_stride);
   else
      walk(glb, A.ws_stride, glb + (long)m.aba_stack * A.ws_stride, A.ws_stride);
}
#undef MH_ST
} // namespace mh
