// mh_zv_kernels.h -- bias-split forward dynamics for the tree-split code objects (gfx950).
//
// ForwardDynamicsCalculator.java:1085-1310 in one wave per limb is a chain of heavy body steps, and every step carries the velocity-
// dependent terms (v, c = v x vJ, p = v x* I v, Ia c) next to the 6 x 6 congruence of the articulated inertia.  The accelerations are
//     qdd = H(q)^-1 (tau - h(q, qd)),        h = inverse dynamics at zero acceleration (gravity and external wrenches included),
// and the articulated inertias depend on q alone.  So the work is cut into two jobs that run SIDE BY SIDE on different workgroups:
//
//   bias job  (4 waves)  tau' = tau - RNEA(q, qd, 0, g, f_ext): the tree-split inverse dynamics of mh_spec_kernels.h with the
//                        accelerations switched off, written to a scratch matrix and published with a release flag per 64 configurations;
//   inertia job (4 waves) pass two of the reference WITHOUT bias terms (IA, U, D^-1 per body; :1146-1235 with p = c = 0), then -- once
//                        the flag is up -- the bias fold (pA' = sum X* pa'_c, u = tau' - S^T pA', pa' = pA' + U D^-1 u; 6-vectors only)
//                        and pass three from a zero root acceleration (:1263-1305 with c = 0: gravity already sits in h).
//
// The serial chain of the inertia job loses the velocity terms (a third of its fp64 instructions and ~100 registers of live state); the
// bias job is as long as an inverse dynamics call and finishes first, so the flag is normally up when the inertia job asks for it.
// A third job of the same launch computes the inverse dynamics output of mh_rnea_aba_f64.
//
// Deadlock freedom: a consumer workgroup spins only on the flag of a producer with a LOWER workgroup id (dispatched earlier, never
// waiting for anything itself); the spin is bounded by a wall-clock limit after which the launch reports an error instead of hanging.
#pragma once
#include "mh_spec_kernels.h"

namespace mh
{
template <class TP>
struct ZvStore
{
   static constexpr int REG_SLOTS = Split<TP>::zv_reg_slots();
   static constexpr int kind(int j) { return Split<TP>::is_trunk(j) ? ST_LDS_KIND : ST_REG_KIND; }
   static constexpr int index(int j) { return Split<TP>::is_trunk(j) ? Split<TP>::zv_trunk_slot(j) : Split<TP>::zv_reg_slot(j); }
   // one wave owns the body's slots: what the bias fold leaves for the outward sweep overwrites 1/D / the factor (Tree<TP>::zv_result_slot)
   static constexpr bool shared(int j) { return !Split<TP>::is_trunk(j); }
};
// The same for the two-launch form of device-filling batches (spec_zvb_kernel): with a staged trunk EVERY wave folds the root body for
// itself (ZvIn / ZvFold MODE 2), so its slots -- 21 for the factor of a 6-DoF root -- are registers of each wave instead of LDS; with
// them out of the way two workgroups fit the LDS of a CU.
template <class TP>
struct ZvbStore
{
   using S = Split<TP>;
   static constexpr bool root_in_regs(int j) { return S::staged() && j == S::root(); }
   static constexpr int REG_SLOTS = S::zv_reg_slots() + (S::staged() ? Tree<TP>::zv_slots_of(S::root() < 0 ? 0 : S::root(), true) : 0);
   static constexpr int trunk_slot(int j)
   { // LDS slots of the trunk bodies before j, the root left out when it lives in registers
      int s = 0;
      for (int i = 0; i < j; i++)
         s += S::is_trunk(i) && !root_in_regs(i) ? Tree<TP>::zv_slots_of(i, false) : 0;
      return s;
   }
   static constexpr int TRUNK_SLOTS = trunk_slot(TP::N);
   static constexpr int kind(int j) { return S::is_trunk(j) && !root_in_regs(j) ? ST_LDS_KIND : ST_REG_KIND; }
   static constexpr int index(int j) { return root_in_regs(j) ? S::zv_reg_slots() : (S::is_trunk(j) ? trunk_slot(j) : S::zv_reg_slot(j)); }
   static constexpr bool shared(int j) { return kind(j) == ST_REG_KIND; }
};
// The store of the FUSED bias + inertia kernel (spec_zvf_kernel): as ZvbStore, plus room for tau - h of every joint -- the inverse dynamics
// of the same workgroup leaves it there instead of in rows of a scratch matrix.  A limb body: ndof more registers of its owner behind its
// bias-split slots.  A trunk body in LDS: ndof more slots (of their own: a plain split folds the trunk on every wave at its own pace, so
// the result may not overwrite the effort it was formed from).  A root body kept in registers: its tau - h in ndof LDS slots behind the
// trunk's (every wave folds the root, one wave computed the efforts).
// Round 5: the MAILED limb (Split::mail_one_limb: a one-body revolute limb whose inverse dynamics another wave walks than the one that owns
// it in the forward dynamics).  Its pair and tau - h -- slots 7, 8, 9 of the body -- live in three LDS slots behind everything either
// phase keeps there (ZvfPlan::mail_base), written by the walking wave, read by the owner behind the barrier between the phases; its other
// slots stay registers of the owner.
template <class TP>
struct ZvfPlan;
template <class TP>
struct ZvfStore
{
   using S = Split<TP>;
   using TR = Tree<TP>;
   static constexpr bool slot_homes = true;
   static constexpr bool mailed(int j) { return ZvfPlan<TP>::use_mail() && !S::is_trunk(j) && S::limb_index_of_body(j) == S::mailed_limb(); }
   static constexpr int slot_kind(int j, int k) { return mailed(j) && k >= 7 ? ST_LDS_KIND : kind(j); }
   static constexpr int slot_index(int j, int k) { return mailed(j) && k >= 7 ? ZvfPlan<TP>::mail_base() + (k - 7) : index(j) + k; }
   static constexpr bool root_in_regs(int j) { return S::staged() && j == S::root(); }
   static constexpr int limb_slots(int j) { return TR::zv_slots_of(j, true) + TR::ndof(j); }
   static constexpr int reg_slot(int j)
   { // registers of the limb bodies before j that the same wave owns
      int s = 0;
      for (int i = 0; i < j; i++)
         s += !S::is_trunk(i) && S::owner(S::limb_index_of_body(i)) == S::owner(S::limb_index_of_body(j)) ? limb_slots(i) : 0;
      return s;
   }
   static constexpr int limb_reg_slots()
   {
      int best = 0;
      for (int w = 0; w < 4; w++)
      {
         int n = 0;
         for (int j = 0; j < TP::N; j++)
            n += !S::is_trunk(j) && S::owner(S::limb_index_of_body(j)) == w ? limb_slots(j) : 0;
         best = n > best ? n : best;
      }
      return best;
   }
   static constexpr int REG_SLOTS = limb_reg_slots() + (S::staged() ? TR::zv_slots_of(S::root() < 0 ? 0 : S::root(), true) : 0);
   static constexpr int trunk_slot(int j)
   {
      int s = 0;
      for (int i = 0; i < j; i++)
         s += S::is_trunk(i) && !root_in_regs(i) ? TR::zv_slots_of(i, false) + TR::ndof(i) : 0;
      return s;
   }
   static constexpr int root_tau_slot() { return trunk_slot(TP::N); }
   static constexpr int ROOT_TAU_SLOTS = S::staged() ? TR::ndof(S::root() < 0 ? 0 : S::root()) : 0;
   // tau - h of the bodies of a LATE limb (the long ones: the legs) leaves its owner's registers before the inertia walk of that limb --
   // the walk that sets the kernel's register count -- and waits for the fold in LDS slots behind the trunk's (zvf_park_late_tau)
   // (and of every other limb of a wave that owns a late one -- the neck beside a leg: the same wave's registers)
   static constexpr bool owns_late(int w)
   {
      for (int k = 0; k < S::n_limbs(); k++)
         if (S::is_late(k) && S::owner(k) == w)
            return true;
      return false;
   }
   static constexpr bool late_body(int j) { return S::staged() && !S::is_trunk(j) && owns_late(S::owner(S::limb_index_of_body(j))); }
   static constexpr int late_tau_slot(int j)
   {
      int s = trunk_slot(TP::N) + ROOT_TAU_SLOTS;
      for (int i = 0; i < j; i++)
         s += late_body(i) ? TR::ndof(i) : 0;
      return s;
   }
   static constexpr int TRUNK_SLOTS = late_tau_slot(TP::N); // trunk bodies | the root's tau - h | the late limbs' tau - h
   static constexpr int tau_slot(int j) { return TR::zv_slots_of(j, !S::is_trunk(j)); } // (not for a root kept in registers: root_tau_slot)
   static constexpr int kind(int j) { return S::is_trunk(j) && !root_in_regs(j) ? ST_LDS_KIND : ST_REG_KIND; }
   static constexpr int index(int j) { return root_in_regs(j) ? limb_reg_slots() : (S::is_trunk(j) ? trunk_slot(j) : reg_slot(j)); }
   static constexpr bool shared(int j) { return kind(j) == ST_REG_KIND; }
};
#ifdef MH_ZV_PROBE // experiment builds: 100 MHz real-time stamps per group, job, wave and phase (tools/exp_zv_probe.py)
__device__ unsigned long long zv_probe[4096 * 3 * 4 * 16];
#define ZV_STAMP(job, ph)                                                                                                                  \
   do                                                                                                                                      \
   {                                                                                                                                       \
      const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                                                      \
      if ((threadIdx.x & 63) == 0 && k < 4096)                                                                                             \
         zv_probe[((k * 3 + (job)) * 4 + (threadIdx.x >> 6)) * 16 + (ph)] = t_;                                                            \
   } while (0)
// ... and inside the inward walk of the inertia job: per group, body and point of the body step (0 children done, 1 constants in and
// inertia summed, 2 division and downdate done, 3 handed up), stamped by the wave that walks the body (the root body: wave 0)
// (-DMH_ZV_PROBE_BODY on top of -DMH_ZV_PROBE: these stamps' stores are waited for at every __syncthreads() and stretch the phases around
// the barriers; read the per-body durations from such a build, the phase times from a build without them)
__device__ unsigned long long zv_probe_body[4096 * 32 * 4];
#ifdef MH_ZV_PROBE_BODY
#define ZV_STAMP_BODY(body, ph)                                                                                                            \
   do                                                                                                                                      \
   {                                                                                                                                       \
      const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                                                      \
      if ((threadIdx.x & 63) == 0 && cx.own < 4096 && ((body) != 0 || cx.wave == 0))                                                      \
         zv_probe_body[(cx.own * 32 + (body)) * 4 + (ph)] = t_;                                                                            \
   } while (0)
#else
#define ZV_STAMP_BODY(body, ph)
#endif
#else
#define ZV_STAMP(job, ph)
#define ZV_STAMP_BODY(body, ph)
#endif
// A workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release fence, which on gfx950 waits for
// EVERY outstanding vector-memory operation of the wave (s_waitcnt vmcnt(0)): loads requested ahead of time would be waited for at the
// first barrier behind them.  Safe where no wave of the workgroup reads global memory another wave of it wrote.
MH_DEV void zv_lds_barrier()
{
   asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// the barriers of the bias fold: loads are in flight across them (the fused kernel's next rows, the small-batch kernel's trunk columns), and
// what the waves exchange there goes through LDS
template <class CX>
MH_DEV void zv_fold_barrier()
{
   zv_lds_barrier();
}
constexpr int ZV_XW = 21; // limb -> trunk exchange record: the articulated inertia (A 6, L 6, C 9); the bias fold reuses its first 6 slots

template <int LIMB, class CX, typename T>
MH_DEV void x_put_ia(const CX &cx, const ABI<T> &I)
{
   constexpr int XW = ZV_XW;
   x_put<LIMB, XW, 0, CX, T>(cx, I.A.xx), x_put<LIMB, XW, 1, CX, T>(cx, I.A.xy), x_put<LIMB, XW, 2, CX, T>(cx, I.A.xz);
   x_put<LIMB, XW, 3, CX, T>(cx, I.A.yy), x_put<LIMB, XW, 4, CX, T>(cx, I.A.yz), x_put<LIMB, XW, 5, CX, T>(cx, I.A.zz);
   x_put<LIMB, XW, 6, CX, T>(cx, I.L.xx), x_put<LIMB, XW, 7, CX, T>(cx, I.L.xy), x_put<LIMB, XW, 8, CX, T>(cx, I.L.xz);
   x_put<LIMB, XW, 9, CX, T>(cx, I.L.yy), x_put<LIMB, XW, 10, CX, T>(cx, I.L.yz), x_put<LIMB, XW, 11, CX, T>(cx, I.L.zz);
   x_put<LIMB, XW, 12, CX, T>(cx, I.C.xx), x_put<LIMB, XW, 13, CX, T>(cx, I.C.xy), x_put<LIMB, XW, 14, CX, T>(cx, I.C.xz);
   x_put<LIMB, XW, 15, CX, T>(cx, I.C.yx), x_put<LIMB, XW, 16, CX, T>(cx, I.C.yy), x_put<LIMB, XW, 17, CX, T>(cx, I.C.yz);
   x_put<LIMB, XW, 18, CX, T>(cx, I.C.zx), x_put<LIMB, XW, 19, CX, T>(cx, I.C.zy), x_put<LIMB, XW, 20, CX, T>(cx, I.C.zz);
}
template <int LIMB, class CX, typename T>
MH_DEV ABI<T> x_get_ia(const CX &cx)
{
   constexpr int XW = ZV_XW;
   ABI<T> I;
   I.A = S3<T>{x_get<LIMB, XW, 0, CX, T>(cx), x_get<LIMB, XW, 1, CX, T>(cx), x_get<LIMB, XW, 2, CX, T>(cx),
               x_get<LIMB, XW, 3, CX, T>(cx), x_get<LIMB, XW, 4, CX, T>(cx), x_get<LIMB, XW, 5, CX, T>(cx)};
   I.L = S3<T>{x_get<LIMB, XW, 6, CX, T>(cx), x_get<LIMB, XW, 7, CX, T>(cx), x_get<LIMB, XW, 8, CX, T>(cx),
               x_get<LIMB, XW, 9, CX, T>(cx), x_get<LIMB, XW, 10, CX, T>(cx), x_get<LIMB, XW, 11, CX, T>(cx)};
   I.C = M3<T>{x_get<LIMB, XW, 12, CX, T>(cx), x_get<LIMB, XW, 13, CX, T>(cx), x_get<LIMB, XW, 14, CX, T>(cx),
               x_get<LIMB, XW, 15, CX, T>(cx), x_get<LIMB, XW, 16, CX, T>(cx), x_get<LIMB, XW, 17, CX, T>(cx),
               x_get<LIMB, XW, 18, CX, T>(cx), x_get<LIMB, XW, 19, CX, T>(cx), x_get<LIMB, XW, 20, CX, T>(cx)};
   return I;
}
template <typename T>
MH_DEV ABI<T> abi_zero()
{
   ABI<T> z;
   z.A = S3<T>{T(0), T(0), T(0), T(0), T(0), T(0)};
   z.L = z.A;
   z.C = M3<T>{T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
   return z;
}
template <int J, int K0, class CX, typename T>
MH_DEV void st_put_ldl(const CX &cx, const LDL6<T> &F)
{ // 21 hand-over slots of a 6-DoF body
   cx.st.template put<J, 0>(F.f[0]), cx.st.template put<J, 1>(F.f[1]), cx.st.template put<J, 2>(F.f[2]), cx.st.template put<J, 3>(F.f[3]);
   cx.st.template put<J, 4>(F.f[4]), cx.st.template put<J, 5>(F.f[5]), cx.st.template put<J, 6>(F.f[6]), cx.st.template put<J, 7>(F.f[7]);
   cx.st.template put<J, 8>(F.f[8]), cx.st.template put<J, 9>(F.f[9]), cx.st.template put<J, 10>(F.f[10]), cx.st.template put<J, 11>(F.f[11]);
   cx.st.template put<J, 12>(F.f[12]), cx.st.template put<J, 13>(F.f[13]), cx.st.template put<J, 14>(F.f[14]), cx.st.template put<J, 15>(F.f[15]);
   cx.st.template put<J, 16>(F.f[16]), cx.st.template put<J, 17>(F.f[17]), cx.st.template put<J, 18>(F.f[18]), cx.st.template put<J, 19>(F.f[19]);
   cx.st.template put<J, 20>(F.f[20]);
}
template <int J, class CX, typename T>
MH_DEV LDL6<T> st_get_ldl(const CX &cx)
{
   LDL6<T> F;
   F.f[0] = cx.st.template get<J, 0>(), F.f[1] = cx.st.template get<J, 1>(), F.f[2] = cx.st.template get<J, 2>(), F.f[3] = cx.st.template get<J, 3>();
   F.f[4] = cx.st.template get<J, 4>(), F.f[5] = cx.st.template get<J, 5>(), F.f[6] = cx.st.template get<J, 6>(), F.f[7] = cx.st.template get<J, 7>();
   F.f[8] = cx.st.template get<J, 8>(), F.f[9] = cx.st.template get<J, 9>(), F.f[10] = cx.st.template get<J, 10>(), F.f[11] = cx.st.template get<J, 11>();
   F.f[12] = cx.st.template get<J, 12>(), F.f[13] = cx.st.template get<J, 13>(), F.f[14] = cx.st.template get<J, 14>(), F.f[15] = cx.st.template get<J, 15>();
   F.f[16] = cx.st.template get<J, 16>(), F.f[17] = cx.st.template get<J, 17>(), F.f[18] = cx.st.template get<J, 18>(), F.f[19] = cx.st.template get<J, 19>();
   F.f[20] = cx.st.template get<J, 20>();
   return F;
}

// ---- inward sweep without bias terms (ForwardDynamicsCalculator.java:1146-1235 with p = c = 0): returns the articulated inertia the
//      subtree hands to its parent, in the parent's frame.  MODE as in AbaIn: 0 whole subtree, 1 trunk pass (limb roots come from the
//      exchange area), 2 the root body alone of a staged trunk, 3 a late limb carrying the workgroup's first barrier.
// (cos, sin) of every revolute joint below and including J, into the store slots ZvIn leaves them in anyway.  A body step of the
// inward walk begins with the sincos of its joint angle -- a long dependent chain with ONE wave per SIMD and nothing to fill its gaps --
// while the joints of a limb need nothing from each other: formed together, up front, the evaluations interleave (leg of the humanoid:
// inward walk 7.6 -> 5.45 us, profiles/r03_presincos_experiment.txt).
// (cos, sin) of revolute joint J: formed from q, or -- CSMODE 2 -- read from the scratch matrix the bias launch left them in
template <class TP, int J, class CX, typename T>
MH_DEV JX<T> zv_revolute_joint(const CX &cx)
{
   if constexpr (CX::csmode == 2)
   {
      constexpr int R = Tree<TP>::rev_index(J);
      JX<T> jx;
      jx.c = cx.cs[(2 * R) * cx.cs_stride], jx.s = cx.cs[(2 * R + 1) * cx.cs_stride], jx.d = T(0);
      return jx;
   }
   else
      return spec_joint<JT_REVOLUTE, Tree<TP>::cfg_ofs(J), CX, T>(cx);
}
// FAST: every pair by the straight-line fast path of the sincos (mh_device.h: sincos_fast), `bad` collects the angles outside its range;
// the caller repeats the walk with FAST = false behind one branch when any is (zv_pre).  Round 5: with the range test INSIDE every
// evaluation (sincos_t) each pair was a basic block of its own and the six of a leg ran one after the other -- 1.5 us on the per-body
// stamps (profiles/r05_zv_body_stamps_before.txt) against 0.55 for six interleaved chains.
template <class TP, int J, typename T, class CX, bool FAST = false>
struct ZvPre
{
   template <int K>
   static MH_DEV void children(const CX &cx, bool &bad)
   {
      if constexpr (K < Tree<TP>::n_children(J))
      {
         ZvPre<TP, Tree<TP>::child(J, K), T, CX, FAST>::run(cx, bad);
         children<K + 1>(cx, bad);
      }
   }
   static MH_DEV void run(const CX &cx, bool &bad)
   {
      children<0>(cx, bad);
      if constexpr (TP::type[J] == JT_REVOLUTE)
      {
         if constexpr (FAST && CX::csmode != 2)
         {
            const T x = cx.q(Tree<TP>::cfg_ofs(J));
            T s, c;
            sincos_fast(x, s, c);
            bad = bad || !sincos_in_fast_range(x);
            cx.st.template put<J, 7>(c);
            cx.st.template put<J, 8>(s);
         }
         else
         {
            const JX<T> jx = zv_revolute_joint<TP, J, CX, T>(cx);
            cx.st.template put<J, 7>(jx.c);
            cx.st.template put<J, 8>(jx.s);
         }
      }
   }
};
#ifndef MH_ZV_PRE_FAST
#define MH_ZV_PRE_FAST 1 // 0: every pair through sincos_t, as before round 5 (A/B measurements)
#endif
template <class TP, int J, typename T, class CX>
MH_DEV void zv_pre(const CX &cx)
{
   bool bad = !MH_ZV_PRE_FAST;
   ZvPre<TP, J, T, CX, MH_ZV_PRE_FAST != 0>::run(cx, bad);
   if (__builtin_expect(bad, 0)) // an angle of 2^19 rad or more somewhere in this limb: once more, every pair through the full sincos
      ZvPre<TP, J, T, CX, false>::run(cx, bad);
}
// The same for the revolute TRUNK bodies of the sub-trunk rooted at J (the wave that will fold the sub-trunk forms them -- or, CSMODE 2,
// requests them from memory -- before it starts on its limbs and finds them in the trunk's LDS slots when it gets there)
template <class TP, int J, typename T, class CX, bool FAST = false>
struct ZvPreTrunk
{
   template <int K>
   static MH_DEV void children(const CX &cx, bool &bad)
   {
      if constexpr (K < Tree<TP>::n_children(J))
      {
         if constexpr (Split<TP>::is_trunk(Tree<TP>::child(J, K)))
            ZvPreTrunk<TP, Tree<TP>::child(J, K), T, CX, FAST>::run(cx, bad);
         children<K + 1>(cx, bad);
      }
   }
   static MH_DEV void run(const CX &cx, bool &bad)
   {
      children<0>(cx, bad);
      if constexpr (TP::type[J] == JT_REVOLUTE)
      {
         if constexpr (FAST && CX::csmode != 2)
         {
            const T x = cx.q(Tree<TP>::cfg_ofs(J));
            T s, c;
            sincos_fast(x, s, c);
            bad = bad || !sincos_in_fast_range(x);
            cx.st.template put<J, 7>(c);
            cx.st.template put<J, 8>(s);
         }
         else
         {
            const JX<T> jx = zv_revolute_joint<TP, J, CX, T>(cx);
            cx.st.template put<J, 7>(jx.c);
            cx.st.template put<J, 8>(jx.s);
         }
      }
   }
};
template <class TP, int W, int I, typename T, class CX, bool FAST = false>
MH_DEV void zv_pre_subtrunks_of(const CX &cx, bool &bad)
{
   using S = Split<TP>;
   if constexpr (I < S::n_sub())
   {
      if constexpr (S::sub_owner(S::sub_top(I)) == W)
         ZvPreTrunk<TP, S::sub_top(I), T, CX, FAST>::run(cx, bad);
      zv_pre_subtrunks_of<TP, W, I + 1, T, CX, FAST>(cx, bad);
   }
}
// every pair wave W needs for its inward walks -- its limbs, early and late, and the sub-trunks it folds -- formed in ONE straight-line
// block in front of them (round 5: seven to nine independent sincos chains interleave; the pairs of the sub-trunk's bodies used to be
// evaluated inside their body steps, 0.2 us each on the wave with the longest path)
#ifndef MH_ZV_PRE_WAVE
#define MH_ZV_PRE_WAVE 1 // 0: one pre-pass per limb, the sub-trunk's pairs inside its body steps, as before (A/B measurements)
#endif
template <class TP, int W, int K, typename T, class CX, bool FAST>
MH_DEV void zv_pre_limbs_of(const CX &cx, bool &bad)
{
   using S = Split<TP>;
   if constexpr (K < S::n_limbs())
   {
      if constexpr (S::owner(K) == W)
         ZvPre<TP, S::limb_root(K), T, CX, FAST>::run(cx, bad);
      zv_pre_limbs_of<TP, W, K + 1, T, CX, FAST>(cx, bad);
   }
}
template <class TP, int W, typename T, class CX>
MH_DEV void zv_pre_wave(const CX &cx)
{
   using S = Split<TP>;
   bool bad = false;
   zv_pre_limbs_of<TP, W, 0, T, CX, true>(cx, bad);
   if constexpr (S::staged())
      zv_pre_subtrunks_of<TP, W, 0, T, CX, true>(cx, bad);
   if (__builtin_expect(bad, 0)) // an angle of 2^19 rad or more among them: once more, every pair through the full sincos
   {
      zv_pre_limbs_of<TP, W, 0, T, CX, false>(cx, bad);
      if constexpr (S::staged())
         zv_pre_subtrunks_of<TP, W, 0, T, CX, false>(cx, bad);
   }
}
template <class TP, int J, typename T, class CX, int MODE = 0>
struct ZvIn
{
   template <int K>
   static MH_DEV void children(const CX &cx, ABI<T> &acc)
   {
      if constexpr (K < Tree<TP>::n_children(J))
      {
         constexpr int C = Tree<TP>::child(J, K);
         ABI<T> c;
         if constexpr ((MODE == 1 || MODE == 2) && !Split<TP>::is_trunk(C))
            c = x_get_ia<Split<TP>::limb_index(C), CX, T>(cx);
         else if constexpr (MODE == 2)
            c = x_get_ia<Split<TP>::sub_slot(C), CX, T>(cx);
         else
            c = ZvIn<TP, C, T, CX, MODE>::run(cx);
         if constexpr (K == 0)
            acc = c;
         else
            add(acc, c);
         children<K + 1>(cx, acc);
      }
   }
   static MH_DEV ABI<T> run(const CX &cx)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J];
      constexpr bool HAS_PARENT = TP::parent[J] >= 0;
      constexpr bool LEAF = Tree<TP>::n_children(J) == 0;
      constexpr int CO = Tree<TP>::cfg_ofs(J);
      ABI<T> up = abi_zero<T>();
      if constexpr (!LEAF)
         children<0>(cx, up);
      if constexpr (MODE == 3 && Split<TP>::is_cut(J))
         __syncthreads(); // barrier 1 of the staged trunk: the early limbs of every wave are in the exchange area
      MH_BODY_FENCE();
      ZV_STAMP_BODY(J, 0);
      const T *cp = cx.C + J * MC_STRIDE;
      asm volatile("" : "+s"(cp)); // the constants are read where they are used, never kept across a subtree
      const CRef<T, false> c{cp};
      // (cos, sin) already in the body's slots: limbs (ZvPre ran, zv_limbs_in_of); CSMODE 2: the bodies of a staged sub-trunk too (ZvPreTrunk); CSMODE 3: every body (left there by the inverse dynamics of the same workgroup)
      constexpr bool PRE = TYPE == JT_REVOLUTE && ((!Split<TP>::is_trunk(J) && (MODE == 0 || MODE == 3)) || ((CX::csmode == 2 || (CX::csmode == 0 && MH_ZV_PRE_WAVE)) && Split<TP>::staged() && Split<TP>::is_trunk(J) && MODE == 1) || (CX::csmode == 3 && Split<TP>::is_trunk(J)));
      JQ<T> jq;
      JX<T> jx;
      if constexpr (TYPE == JT_REVOLUTE && CX::csmode == 2 && !PRE)
         jx = zv_revolute_joint<TP, J, CX, T>(cx);
      else if constexpr (!PRE)
         jq = spec_joint_read<TYPE, CO, CX, T>(cx);
      const RI<T> I = load_inertia<T>(c);
      MH_BODY_FENCE();
      ABI<T> IA = abi_from_rigid(I);
      if constexpr (!LEAF)
         add(IA, up);
      MH_BODY_FENCE();
      ZV_STAMP_BODY(J, 1);
      // the pose is requested only now, when the ten inertia constants have been consumed: requested together they hold 44 SGPRs next to
      // the kernel's arguments, more than the file has, and every use became a v_readlane from a spill lane (15 % of the instructions of
      // this chain); the sincos and the rank-1 downdate in front of its first use cover the latency
      const XF<T> Xb = load_xb_j<TP, J, T>(c);
      if constexpr (PRE)
         jx.c = cx.st.template get<J, 7>(), jx.s = cx.st.template get<J, 8>(), jx.d = T(0);
      else if constexpr (!(TYPE == JT_REVOLUTE && CX::csmode == 2))
         jx = spec_joint_from<TYPE, T>(jq);
      ABI<T> out = abi_zero<T>();
      if constexpr (TYPE == JT_REVOLUTE || TYPE == JT_PRISMATIC)
      {
         V3<T> ua, ul;
         T D;
         if constexpr (TYPE == JT_REVOLUTE)
            ua = V3<T>{IA.A.xz, IA.A.yz, IA.A.zz}, ul = V3<T>{IA.C.zx, IA.C.zy, IA.C.zz}, D = IA.A.zz;
         else
            ua = V3<T>{IA.C.xz, IA.C.yz, IA.C.zz}, ul = V3<T>{IA.L.xz, IA.L.yz, IA.L.zz}, D = IA.L.zz;
         const T dinv = rcp_fast(D); // (:1183 has 1.0 / D: a correctly rounded quotient; this one is within an ulp of it)
         const V3<T> sa = dinv * ua, sl = dinv * ul;
         cx.st.template put<J, 0>(sa.x), cx.st.template put<J, 1>(sa.y), cx.st.template put<J, 2>(sa.z);
         cx.st.template put<J, 3>(sl.x), cx.st.template put<J, 4>(sl.y), cx.st.template put<J, 5>(sl.z);
         cx.st.template put<J, 6>(dinv);
         if constexpr (TYPE == JT_REVOLUTE)
         {
            cx.st.template put<J, 7>(jx.c);
            cx.st.template put<J, 8>(jx.s);
         }
         if constexpr (HAS_PARENT)
         {
            if constexpr (TYPE == JT_REVOLUTE)
               rank1_down_revolute(IA, ua, ul, dinv);
            else
               rank1_down(IA, ua, ul, dinv);
#ifdef MH_ZV_PROBE_BODY
            MH_BODY_FENCE();
            ZV_STAMP_BODY(J, 2);
#endif
            abi_up(TYPE, jx, Xb, IA);
            out = IA;
         }
      }
      else if constexpr (TYPE == JT_SIXDOF)
         st_put_ldl<J, 0, CX, T>(cx, spd6_factor(IA)); // S = 1: U = D = IA, nothing is left for the parent (Ia = 0)
      else if constexpr (HAS_PARENT)
      { // fixed joint
         abi_up(TYPE, jx, Xb, IA);
         out = IA;
      }
      MH_BODY_FENCE();
      ZV_STAMP_BODY(J, 3);
      return out;
   }
};

// ---- the bias fold: pA' = sum over the children of X* pa'_c;  u = tau' - S^T pA';  pa' = pA' + U D^-1 u  (:1224-1235 with c = 0).
//      Returns pa' in the parent's frame.  MODE 0: whole subtree; 1: trunk pass, limb roots come from the exchange area; 2: the root
//      body alone of a staged fold (sub-trunks come from the exchange area too).
template <class TP, int J, typename T, class CX, int MODE = 0>
struct ZvFold
{
   template <int K>
   static MH_DEV void children(const CX &cx, SV<T> &pA)
   {
      if constexpr (K < Tree<TP>::n_children(J))
      {
         constexpr int C = Tree<TP>::child(J, K);
         SV<T> c;
         if constexpr ((MODE == 1 || MODE == 2) && !Split<TP>::is_trunk(C))
            c = x_get6<Split<TP>::limb_index(C), CX::fold_xw, 0, CX, T>(cx);
         else if constexpr (MODE == 2) // staged fold: the sub-trunk below the root was folded by one wave (zv_subtrunks_fold_of)
            c = x_get6<Split<TP>::sub_slot(C), CX::fold_xw, 6, CX, T>(cx);
         else
            c = ZvFold<TP, C, T, CX, MODE>::run(cx);
         if constexpr (K == 0)
            pA = c;
         else
            pA = pA + c;
         children<K + 1>(cx, pA);
      }
   }
   static MH_DEV SV<T> run(const CX &cx)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J];
      constexpr bool HAS_PARENT = TP::parent[J] >= 0;
      constexpr bool LEAF = Tree<TP>::n_children(J) == 0;
      constexpr int DO = Tree<TP>::dof_ofs(J), CO = Tree<TP>::cfg_ofs(J);
      constexpr int RS = Tree<TP>::zv_result_slot(J, CX::SPolicy::shared(J));
      const V3<T> Z{T(0), T(0), T(0)};
      SV<T> pA{Z, Z};
      if constexpr (!LEAF)
         children<0>(cx, pA);
      MH_BODY_FENCE();
      const T *cp = cx.C + J * MC_STRIDE;
      asm volatile("" : "+s"(cp));
      const CRef<T, false> c{cp};
      SV<T> up{Z, Z};
      if constexpr (TYPE == JT_REVOLUTE || TYPE == JT_PRISMATIC)
      {
         T tau;
         if constexpr (CX::csmode == 3)
            tau = zvf_tau_get1<TP, J, 0, CX, T>(cx);
         else
            tau = cx.in3(DO);
         JX<T> jx;
         jx.c = T(1), jx.s = T(0), jx.d = T(0);
         if constexpr (TYPE == JT_REVOLUTE)
            jx.c = cx.st.template get<J, 7>(), jx.s = cx.st.template get<J, 8>();
         else if constexpr (HAS_PARENT)
            jx.d = cx.q(CO);
         const V3<T> sa{cx.st.template get<J, 0>(), cx.st.template get<J, 1>(), cx.st.template get<J, 2>()};
         const V3<T> sl{cx.st.template get<J, 3>(), cx.st.template get<J, 4>(), cx.st.template get<J, 5>()};
         const T dinv = cx.st.template get<J, 6>();
         const T u = tau - (TYPE == JT_REVOLUTE ? pA.a.z : pA.l.z);
         cx.st.template put<J, RS>(u * dinv);
         if constexpr (HAS_PARENT)
         {
            const XF<T> Xb = load_xb_j<TP, J, T>(c);
            const SV<T> pa = pA + SV<T>{u * sa, u * sl};
            up = force_up(TYPE, jx, Xb, pa);
         }
      }
      else if constexpr (TYPE == JT_SIXDOF)
      {
         SV<T> tau;
         if constexpr (CX::csmode == 3)
            tau = SV<T>{V3<T>{zvf_tau_get1<TP, J, 0, CX, T>(cx), zvf_tau_get1<TP, J, 1, CX, T>(cx), zvf_tau_get1<TP, J, 2, CX, T>(cx)},
                        V3<T>{zvf_tau_get1<TP, J, 3, CX, T>(cx), zvf_tau_get1<TP, J, 4, CX, T>(cx), zvf_tau_get1<TP, J, 5, CX, T>(cx)}};
         else
            tau = spec_vec<TYPE, DO, 1, CX, T>(cx, true);
         const SV<T> x = spd6_solve(st_get_ldl<J, CX, T>(cx), tau - pA);
         cx.st.template put<J, RS + 0>(x.a.x), cx.st.template put<J, RS + 1>(x.a.y), cx.st.template put<J, RS + 2>(x.a.z);
         cx.st.template put<J, RS + 3>(x.l.x), cx.st.template put<J, RS + 4>(x.l.y), cx.st.template put<J, RS + 5>(x.l.z);
         if constexpr (HAS_PARENT)
         {
            const JX<T> jx = spec_joint<TYPE, CO, CX, T>(cx);
            up = force_up(TYPE, jx, load_xb_j<TP, J, T>(c), tau); // Ia = 0, pa = tau
         }
      }
      else if constexpr (HAS_PARENT)
      {
         JX<T> jx;
         jx.c = T(1), jx.s = T(0), jx.d = T(0);
         up = force_up(TYPE, jx, load_xb_j<TP, J, T>(c), pA);
      }
      MH_BODY_FENCE();
      return up;
   }
};

// ---- outward sweep from a zero root acceleration (:1263-1305 with c = 0).  MODE 1: a limb hanging off this trunk body is continued
//      only by the wave that owns it; the trunk itself is walked by every wave, its outputs written by wave 0.
template <class TP, int J, typename T, class CX, int MODE = 0>
struct ZvOut
{
   template <int K>
   static MH_DEV void children(const CX &cx, const SV<T> &a)
   {
      if constexpr (K < Tree<TP>::n_children(J))
      {
         constexpr int C = Tree<TP>::child(J, K);
         if constexpr (MODE == 1 && !Split<TP>::is_trunk(C))
         {
            if (cx.wave == Split<TP>::owner(Split<TP>::limb_index(C)))
               ZvOut<TP, C, T, CX, 0>::run(cx, a);
         }
         else
            ZvOut<TP, C, T, CX, MODE>::run(cx, a);
         children<K + 1>(cx, a);
      }
   }
   static MH_DEV void run(const CX &cx, const SV<T> &ap)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J];
      constexpr bool HAS_PARENT = TP::parent[J] >= 0;
      constexpr bool LEAF = Tree<TP>::n_children(J) == 0;
      constexpr int DO = Tree<TP>::dof_ofs(J), CO = Tree<TP>::cfg_ofs(J);
      constexpr int RS = Tree<TP>::zv_result_slot(J, CX::SPolicy::shared(J));
      const T *cp = cx.C + J * MC_STRIDE;
      asm volatile("" : "+s"(cp));
      const CRef<T, false> c{cp};
      const V3<T> Z{T(0), T(0), T(0)};
      SV<T> a{Z, Z};
      if constexpr (HAS_PARENT)
      {
         JX<T> jx;
         jx.c = T(1), jx.s = T(0), jx.d = T(0);
         if constexpr (TYPE == JT_REVOLUTE)
            jx.c = cx.st.template get<J, 7>(), jx.s = cx.st.template get<J, 8>();
         else if constexpr (TYPE != JT_FIXED)
            jx = spec_joint<TYPE, CO, CX, T>(cx);
         a = motion_down(TYPE, jx, load_xb_j<TP, J, T>(c), ap);
      }
      if constexpr (TYPE == JT_REVOLUTE || TYPE == JT_PRISMATIC)
      {
         const V3<T> sa{cx.st.template get<J, 0>(), cx.st.template get<J, 1>(), cx.st.template get<J, 2>()};
         const V3<T> sl{cx.st.template get<J, 3>(), cx.st.template get<J, 4>(), cx.st.template get<J, 5>()};
         const T qdd = cx.st.template get<J, RS>() - (dot(sa, a.a) + dot(sl, a.l));
         if (MODE == 0 || cx.wave == 0)
            cx.out(DO, qdd);
         if constexpr (TYPE == JT_REVOLUTE)
            a.a.z += qdd;
         else
            a.l.z += qdd;
      }
      else if constexpr (TYPE == JT_SIXDOF)
      {
         const SV<T> x{V3<T>{cx.st.template get<J, RS + 0>(), cx.st.template get<J, RS + 1>(), cx.st.template get<J, RS + 2>()},
                       V3<T>{cx.st.template get<J, RS + 3>(), cx.st.template get<J, RS + 4>(), cx.st.template get<J, RS + 5>()}};
         if (MODE == 0 || cx.wave == 0)
            spec_write<TYPE, DO, CX, T>(cx, x - a);
         a = x;
      }
      MH_BODY_FENCE();
      if constexpr (!LEAF)
         children<0>(cx, a);
   }
};

template <class TP, typename T, class CX, int MODE, int K = 0>
MH_DEV void zv_roots_in(const CX &cx)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      (void)ZvIn<TP, Tree<TP>::child(-1, K), T, CX, MODE>::run(cx);
      zv_roots_in<TP, T, CX, MODE, K + 1>(cx);
   }
}
template <class TP, typename T, class CX, int K = 0>
MH_DEV void zv_roots_fold(const CX &cx)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      (void)ZvFold<TP, Tree<TP>::child(-1, K), T, CX, 1>::run(cx);
      zv_roots_fold<TP, T, CX, K + 1>(cx);
   }
}
template <class TP, typename T, class CX, int K = 0>
MH_DEV void zv_roots_out(const CX &cx)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      const V3<T> Z{T(0), T(0), T(0)};
      ZvOut<TP, Tree<TP>::child(-1, K), T, CX, 1>::run(cx, SV<T>{Z, Z});
      zv_roots_out<TP, T, CX, K + 1>(cx);
   }
}
// the limbs of wave W: LATE = -1 all of them (plain split), 0 / 1 the early / late ones of a staged trunk
template <class TP, int W, int K, int LATE, typename T, class CX>
MH_DEV void zv_limbs_in_of(const CX &cx)
{
   using S = Split<TP>;
   if constexpr (K < S::n_limbs())
   {
      if constexpr (S::owner(K) == W && (LATE < 0 || (S::is_late(K) ? 1 : 0) == LATE))
      {
         if constexpr (CX::csmode != 3 && !(CX::csmode == 0 && MH_ZV_PRE_WAVE)) // (CSMODE 3: the inverse dynamics of this workgroup left the pairs in the slots; CSMODE 0: zv_pre_wave)
            zv_pre<TP, S::limb_root(K), T, CX>(cx);
         x_put_ia<K, CX, T>(cx, ZvIn<TP, S::limb_root(K), T, CX, (LATE >= 0 && S::cut_limb(W) == K ? 3 : 0)>::run(cx));
      }
      zv_limbs_in_of<TP, W, K + 1, LATE, T, CX>(cx);
   }
}
template <class TP, int W, int I, typename T, class CX>
MH_DEV void zv_subtrunks_of(const CX &cx)
{
   using S = Split<TP>;
   if constexpr (I < S::n_sub())
   {
      constexpr int ST = S::sub_top(I);
      if constexpr (S::sub_owner(ST) == W)
         x_put_ia<S::sub_slot(ST), CX, T>(cx, ZvIn<TP, ST, T, CX, 1>::run(cx));
      zv_subtrunks_of<TP, W, I + 1, T, CX>(cx);
   }
}
// fused kernel: wave W moves tau - h of its late limbs' joints from its registers to their LDS slots (ZvfStore::late_tau_slot)
template <class TP, int W, int J, typename T, class CX>
MH_DEV void zvf_park_late_tau(const CX &cx)
{
   using SP = typename CX::SPolicy;
   if constexpr (J < TP::N)
   {
      if constexpr (SP::late_body(J) && Split<TP>::owner(Split<TP>::limb_index_of_body(J)) == W)
      {
         constexpr int TS = SP::tau_slot(J), L0 = SP::late_tau_slot(J);
         cx.st.lbase[L0 * 64] = cx.st.template get<J, TS>();
         if constexpr (Tree<TP>::ndof(J) == 6)
         {
            cx.st.lbase[(L0 + 1) * 64] = cx.st.template get<J, TS + 1>(), cx.st.lbase[(L0 + 2) * 64] = cx.st.template get<J, TS + 2>();
            cx.st.lbase[(L0 + 3) * 64] = cx.st.template get<J, TS + 3>(), cx.st.lbase[(L0 + 4) * 64] = cx.st.template get<J, TS + 4>();
            cx.st.lbase[(L0 + 5) * 64] = cx.st.template get<J, TS + 5>();
         }
      }
      zvf_park_late_tau<TP, W, J + 1, T, CX>(cx);
   }
}
// limb phases of the inward sweep, wave by wave; a staged trunk passes its first barrier in here exactly once per wave
template <class TP, int W, typename T, class CX>
MH_DEV void zv_limbs_in(const CX &cx)
{
   using S = Split<TP>;
   if constexpr (W < 4)
   {
      if (cx.wave == W)
      {
         if constexpr (CX::csmode == 3)
            zvf_park_late_tau<TP, W, 0, T, CX>(cx);
         if constexpr (CX::csmode == 0 && MH_ZV_PRE_WAVE)
            zv_pre_wave<TP, W, T, CX>(cx);
         if constexpr (S::staged())
         {
            if constexpr (CX::csmode == 2)
            {
               bool unused = false;
               zv_pre_subtrunks_of<TP, W, 0, T, CX>(cx, unused);
            }
            zv_limbs_in_of<TP, W, 0, 0, T, CX>(cx);
            if constexpr (S::cut_limb(W) < 0)
               __syncthreads();
            zv_limbs_in_of<TP, W, 0, 1, T, CX>(cx);
            zv_subtrunks_of<TP, W, 0, T, CX>(cx);
         }
         else
            zv_limbs_in_of<TP, W, 0, -1, T, CX>(cx);
      }
      else
         zv_limbs_in<TP, W + 1, T, CX>(cx);
   }
}
template <class TP, int W, int K, typename T, class CX>
MH_DEV void zv_limbs_fold_of(const CX &cx)
{
   using S = Split<TP>;
   if constexpr (K < S::n_limbs())
   {
      if constexpr (S::owner(K) == W)
         x_put6<K, CX::fold_xw, 0, CX, T>(cx, ZvFold<TP, S::limb_root(K), T, CX, 0>::run(cx));
      zv_limbs_fold_of<TP, W, K + 1, T, CX>(cx);
   }
}
template <class TP, int W, typename T, class CX>
MH_DEV void zv_limbs_fold(const CX &cx)
{
   if constexpr (W < 4)
   {
      if (cx.wave == W)
         zv_limbs_fold_of<TP, W, 0, T, CX>(cx);
      else
         zv_limbs_fold<TP, W + 1, T, CX>(cx);
   }
}

// ---- which trunk bodies a wave has to walk in the outward sweep: those above a limb it owns (it needs their accelerations), plus the
//      ones it WRITES the accelerations of -- for every trunk body the lowest-numbered wave that walks it anyway
template <class TP>
struct ZvWalk
{
   using S = Split<TP>;
   static constexpr bool below(int j, int top)
   { // is body j in the subtree of (or equal to) body top
      for (int a = j; a >= 0; a = TP::parent[a])
         if (a == top)
            return true;
      return false;
   }
   static constexpr bool needs(int W, int J)
   {
      for (int k = 0; k < S::n_limbs(); k++)
         if (S::owner(k) == W && below(S::limb_root(k), J))
            return true;
      return false;
   }
   static constexpr int writer(int J)
   {
      for (int w = 0; w < 4; w++)
         if (needs(w, J))
            return w;
      return 0;
   }
};
// outward sweep of wave W over the trunk (MODE 1 of ZvOut with the subtrees pruned in which W owns nothing), then W's limbs
template <class TP, int J, int W, typename T, class CX>
struct ZvOutW
{
   template <int K>
   static MH_DEV void children(const CX &cx, const SV<T> &a)
   {
      if constexpr (K < Tree<TP>::n_children(J))
      {
         constexpr int C = Tree<TP>::child(J, K);
         if constexpr (!Split<TP>::is_trunk(C))
         {
            if constexpr (Split<TP>::owner(Split<TP>::limb_index(C)) == W)
               ZvOut<TP, C, T, CX, 0>::run(cx, a);
         }
         else if constexpr (ZvWalk<TP>::needs(W, C))
            ZvOutW<TP, C, W, T, CX>::run(cx, a);
         children<K + 1>(cx, a);
      }
   }
   static MH_DEV void run(const CX &cx, const SV<T> &ap)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J];
      constexpr bool HAS_PARENT = TP::parent[J] >= 0;
      constexpr int DO = Tree<TP>::dof_ofs(J), CO = Tree<TP>::cfg_ofs(J);
      constexpr int RS = Tree<TP>::zv_result_slot(J, CX::SPolicy::shared(J));
      constexpr bool WRITES = ZvWalk<TP>::writer(J) == W;
      const T *cp = cx.C + J * MC_STRIDE;
      asm volatile("" : "+s"(cp));
      const CRef<T, false> c{cp};
      const V3<T> Z{T(0), T(0), T(0)};
      SV<T> a{Z, Z};
      if constexpr (HAS_PARENT)
      {
         JX<T> jx;
         jx.c = T(1), jx.s = T(0), jx.d = T(0);
         if constexpr (TYPE == JT_REVOLUTE)
            jx.c = cx.st.template get<J, 7>(), jx.s = cx.st.template get<J, 8>();
         else if constexpr (TYPE != JT_FIXED)
            jx = spec_joint<TYPE, CO, CX, T>(cx);
         a = motion_down(TYPE, jx, load_xb_j<TP, J, T>(c), ap);
      }
      if constexpr (TYPE == JT_REVOLUTE || TYPE == JT_PRISMATIC)
      {
         const V3<T> sa{cx.st.template get<J, 0>(), cx.st.template get<J, 1>(), cx.st.template get<J, 2>()};
         const V3<T> sl{cx.st.template get<J, 3>(), cx.st.template get<J, 4>(), cx.st.template get<J, 5>()};
         const T qdd = cx.st.template get<J, RS>() - (dot(sa, a.a) + dot(sl, a.l));
         if constexpr (WRITES)
            cx.out(DO, qdd);
         if constexpr (TYPE == JT_REVOLUTE)
            a.a.z += qdd;
         else
            a.l.z += qdd;
      }
      else if constexpr (TYPE == JT_SIXDOF)
      {
         const SV<T> x{V3<T>{cx.st.template get<J, RS + 0>(), cx.st.template get<J, RS + 1>(), cx.st.template get<J, RS + 2>()},
                       V3<T>{cx.st.template get<J, RS + 3>(), cx.st.template get<J, RS + 4>(), cx.st.template get<J, RS + 5>()}};
         if constexpr (WRITES)
            spec_write<TYPE, DO, CX, T>(cx, x - a);
         a = x;
      }
      MH_BODY_FENCE();
      children<0>(cx, a);
   }
};
template <class TP, int W, typename T, class CX, int K = 0>
MH_DEV void zv_roots_out_wave(const CX &cx)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      const V3<T> Z{T(0), T(0), T(0)};
      constexpr int R = Tree<TP>::child(-1, K);
      if constexpr (ZvWalk<TP>::needs(W, R) || ZvWalk<TP>::writer(R) == W)
         ZvOutW<TP, R, W, T, CX>::run(cx, SV<T>{Z, Z});
      zv_roots_out_wave<TP, W, T, CX, K + 1>(cx);
   }
}
// limbs of wave W in the bias fold: LATE = -1 all, 0 / 1 the early / late ones of a staged trunk
template <class TP, int W, int K, int LATE, typename T, class CX>
MH_DEV void zv_limbs_fold_sel(const CX &cx)
{
   using S = Split<TP>;
   if constexpr (K < S::n_limbs())
   {
      if constexpr (S::owner(K) == W && (LATE < 0 || (S::is_late(K) ? 1 : 0) == LATE))
         x_put6<K, CX::fold_xw, 0, CX, T>(cx, ZvFold<TP, S::limb_root(K), T, CX, 0>::run(cx));
      zv_limbs_fold_sel<TP, W, K + 1, LATE, T, CX>(cx);
   }
}
// the sub-trunks wave W folds between the fold's two barriers: bias wrench handed up in the SECOND record of the exchange slot the
// sub-trunk's inertia travelled in (its first record holds that early limb's own bias wrench, which this very fold consumes)
template <class TP, int W, int I, typename T, class CX>
MH_DEV void zv_subtrunks_fold_of(const CX &cx)
{
   using S = Split<TP>;
   if constexpr (I < S::n_sub())
   {
      constexpr int ST = S::sub_top(I);
      if constexpr (S::sub_owner(ST) == W)
         x_put6<S::sub_slot(ST), CX::fold_xw, 6, CX, T>(cx, ZvFold<TP, ST, T, CX, 1>::run(cx));
      zv_subtrunks_fold_of<TP, W, I + 1, T, CX>(cx);
   }
}
// Bias fold and outward sweep of wave W.  Staged trunk: early limbs | barrier | late limbs and, on the waves that own them, the sub-trunks
// (their results go to the trunk's LDS slots) | barrier | the root body alone on every wave | the outward sweep over the part of the
// trunk W needs.  The serial chain is max(early limbs + sub-trunk, late limbs) + root instead of (all limbs of a wave) + (whole trunk).
// Plain split: limbs | barrier | whole trunk on every wave | pruned outward sweep.  Every wave passes the same number of barriers.
struct ZvNoHook
{
   MH_DEV void after_early() const {}
   MH_DEV void after_late() const {}
   MH_DEV void before_out() const {}
};
// HOOK: called by every wave behind its early limbs' fold (plain split: behind its limbs' fold), behind its late limbs' fold (plain split:
// behind the barrier in front of the trunk) and between its fold and its outward sweep.  The two-stage hand-off of the small-batch kernel
// asks for the trunk's bias efforts at the first and stores them at the second; the fused kernel requests the next group's rows at the
// third (the root's factor and the fold's temporaries are dead, the request has the outward sweep and the copy-out to land).
template <class TP, int W, typename T, class CX, class HOOK = ZvNoHook>
MH_DEV void zv_fold_out(const CX &cx, const HOOK &hook = HOOK())
{
   using S = Split<TP>;
   if constexpr (W < 4)
   {
      if (cx.wave == W)
      {
         if constexpr (S::staged())
         {
#ifdef MH_ZV_PROBE
            const long k = (long)cx.own; // (the group, for the stamps: zv_aba_group leaves it there in probe builds)
#endif
            zv_limbs_fold_sel<TP, W, 0, 0, T, CX>(cx);
            hook.after_early();
            ZV_STAMP(1, 7);
            // (An arrival count instead of this barrier -- only the wave that folds a sub-trunk waiting for the early limbs, the legs' waves
            // going straight on -- was measured: the legs finish 0.3 us earlier, but arm -> torso -> pelvis is the fold's chain: no change.)
            zv_fold_barrier<CX>();
            zv_limbs_fold_sel<TP, W, 0, 1, T, CX>(cx);
            hook.after_late();
            zv_subtrunks_fold_of<TP, W, 0, T, CX>(cx);
            ZV_STAMP(1, 8);
            zv_fold_barrier<CX>();
            (void)ZvFold<TP, S::root(), T, CX, 2>::run(cx);
            ZV_STAMP(1, 9);
         }
         else
         {
            zv_limbs_fold_sel<TP, W, 0, -1, T, CX>(cx);
            hook.after_early();
            zv_fold_barrier<CX>();
            hook.after_late();
            zv_roots_fold<TP, T, CX>(cx);
         }
         // CSMODE 2 writes the accelerations IN PLACE over the bias rows (no LDS left for rows of their own at two workgroups per CU): no
         // wave may start writing while another still folds the trunk, whose efforts it reads from those rows (CSMODE 3: its result rows
         // lie over the fold's exchange records)
         if constexpr (CX::csmode >= 2)
            zv_fold_barrier<CX>();
         hook.before_out();
         asm volatile("" ::: "memory");
         zv_roots_out_wave<TP, W, T, CX>(cx);
      }
      else
         zv_fold_out<TP, W + 1, T, CX, HOOK>(cx, hook);
   }
}

// rows of one matrix, global -> LDS, all NT threads of the workgroup (N entries per configuration, `rows` configurations)
template <typename T, int N, int NT>
MH_DEV void zv_stage_rows(lds_ptr<T> dst, const T *src, int rows)
{
   constexpr int U = (64 * N + NT - 1) / NT;
   T r[U];
   const int n = rows * N, t = threadIdx.x;
#pragma unroll
   for (int u = 0; u < U; u++)
      r[u] = t + NT * u < n ? src[t + NT * u] : T(0);
#pragma unroll
   for (int u = 0; u < U; u++)
      if (t + NT * u < 64 * N)
         dst[t + NT * u] = r[u];
}

struct ZvSync
{
   int *flags;     // per 64 configurations ZV_SYNC_STRIDE ints (two 128-byte lines): [0] the flag -- the bias job stores `epoch` there once
                   // its rows can be read --, [32] the mailbox the inertia job leaves its XCD's id in
   int *error;     // set to 1 when an inertia job gave up waiting (wall-clock limit): the host turns it into MH_ERR_HIP
   int epoch;
   int jobs;       // 2: bias + inertia (mh_aba_f64); 3: + the inverse dynamics of mh_rnea_aba_f64
   int same_l2;    // 1: a bias job that finds its inertia job behind the SAME L2 (both read HW_REG_XCC_ID) leaves rows and flag in that L2
   unsigned wait_ticks; // how long an inertia job waits for its flag, in ticks of the 100 MHz real-time counter (MH_ZV_WAIT_MS; default 2 s)
};
// id of the XCD this wave runs on (HW_REG_XCC_ID, register 20, bits 3:0)
MH_DEV int zv_xcc_id() { return (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf); }
MH_DEV int zv_mail_key(int epoch, int xcc) { return (int)(((unsigned)epoch & 0x03ffffffu) << 5) | 16 | xcc; }

// The hand-off follows the one form MI355X_MICROARCH.md lists as valid without agent-scope fences (a release fence writes back the whole
// L2: 7-11 us measured here; an acquire fence invalidates it): every byte of the bias rows is stored sc1 (write-through) and loaded sc1
// (served past the L1), every storing wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup barrier behind which ONE lane
// stores the flag sc1, and the consumer's polling wave joins a workgroup barrier before any wave loads the rows.  Where the two jobs of
// a group have PROVED to run behind the same L2 (mailbox, zv_bias_group) the stores are workgroup-scope (sc0) instead: rows and flag
// stay in that L2, which is where the consumer's sc1 polls and loads are served from; everything else is the same.
// Returns true once the flag of group k holds this launch's epoch; false after the wall-clock limit, with the error word set (the caller
// then writes NaN rows instead of accelerations: zv_aba_group).  The limit is generous (seconds): the real-time counter keeps counting
// while a queue is preempted, and a producer that has not even been dispatched yet (HIP promises no dispatch order) needs a CU to free up.
MH_DEV bool zv_wait(const ZvSync &sy, long k)
{
   const int *f = sy.flags + k * ZV_SYNC_STRIDE;
   if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sy.epoch)
      return true;
   const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
   for (;;)
   {
      __builtin_amdgcn_s_sleep(1);
      if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sy.epoch)
         return true;
      if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)sy.wait_ticks)
      {
         if ((threadIdx.x & 63) == 0)
            __hip_atomic_store(sy.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
         return false;
      }
   }
}
// rows of one matrix, LDS -> global, all NT threads: write-through (sc1), or left in the L2 the reader shares (sc0)
template <typename T, int NT>
MH_DEV void zv_publish_rows(T *dst, lds_ptr<T> src, int n, bool same_l2)
{ // same_l2: the reader is known to sit behind this L2 -- workgroup-scope stores (sc0) leave the lines there, where its sc1 loads are served
   if (same_l2)
      for (int i = threadIdx.x; i < n; i += NT)
         __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
   else
      for (int i = threadIdx.x; i < n; i += NT)
         __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// rows of one matrix, global -> LDS, every load sc1 (past the L1), all NT threads
template <typename T, int N, int NT>
MH_DEV void zv_fetch_rows(lds_ptr<T> dst, const T *src, int rows)
{
   constexpr int U = (64 * N + NT - 1) / NT;
   T r[U];
   const int n = rows * N, t = threadIdx.x;
#pragma unroll
   for (int u = 0; u < U; u++)
      r[u] = t + NT * u < n ? __hip_atomic_load(src + t + NT * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : T(0);
#pragma unroll
   for (int u = 0; u < U; u++)
      if (t + NT * u < 64 * N)
         dst[t + NT * u] = r[u];
}

// The rows of the NEXT group a workgroup will work on, requested while the current one is being finished and held in registers until the
// LDS rows are free (both launches loop over groups with two workgroups per CU: a wave that sits waiting for its rows is a quarter of what
// the SIMD has to run).  Stamps at B = 262 144 (profiles/r04_zvb_phase_stamps.txt): staging q, qd, tau took 3.2 of the bias launch's 11.3 us
// per group, the (cos, sin) pairs and the bias rows ~2 of the inertia launch's 11.4.
template <typename T, int N, int NT>
struct RowRegs
{
   static constexpr int U = (64 * N + NT - 1) / NT;
   T r[U];
   // t: this thread's number among the NT that take part (threadIdx.x when all 256 do; threadIdx.x - 64 when waves 1-3 load for the group)
   MH_DEV void issue(const T *src, int rows, int t = threadIdx.x)
   { // rows >= 1.  Entries past the last row are clamped onto it, not selected away: a select on the loaded value makes the load a
     // synchronous one (the wave waits for it right here -- 1.4 us per group on the stamps), and nobody reads those LDS entries anyway
      const int last = rows * N - 1;
#pragma unroll
      for (int u = 0; u < U; u++)
         r[u] = src[t + NT * u < last ? t + NT * u : last];
   }
   MH_DEV void commit(lds_ptr<T> dst, int t = threadIdx.x) const
   {
#pragma unroll
      for (int u = 0; u < U; u++)
         if (t + NT * u < 64 * N)
            dst[t + NT * u] = r[u];
   }
};
// ---- The hand-off in two stages (identity index maps; MH_ZV_TWO_STAGE=0 builds the one-stage form below for every model).
// Phase stamps of the one-stage form (profiles/r04_zv_phase_stamps_b4096.txt): the inertia job is through with the root body at 9.2 us and
// holds the bias efforts at 11.5 -- the trunk pass of the bias job (done 9.15), the copy of all rows (9.6), the acknowledgements (9.9), the
// flag (10.0 -> seen 10.6), the fetch (11.5).  But the LIMBS' bias efforts exist at 8.3 (the barrier in front of the trunk pass), and the
// limbs' folds are the first 1.5 us of what the inertia job does with the rows; the trunk's entries are needed behind them.  So:
//   * the hand-off matrix of a group is column-major, [nv][64]: a wave publishes a column with one 512-byte store instruction, so
//     different waves publish different columns at different times without writing any line in pieces;
//   * waves 1-3 of the bias job publish the limb dofs' columns while wave 0 folds the trunk; each waits for its stores' acknowledgements
//     and counts itself in (LDS); the last of the three stores flag A.  Wave 0 publishes the trunk dofs' columns behind its pass, waits,
//     stores flag B;
//   * every wave of the inertia job polls flag A itself, loads the columns of ITS limbs' dofs into its lanes' LDS rows and folds its
//     limbs; it polls flag B and requests the trunk's columns behind its early limbs' fold and stores them to its rows behind its late
//     limbs' fold, in front of the first use (sub-trunk / root fold).
// Each of the two hand-offs is the form of MI355X_MICROARCH.md's table, first row: every byte stored sc1 and loaded sc1 (8 bytes per
// lane), every storing wave drains its stores (s_waitcnt vmcnt(0)) before it is counted, ONE lane signals for all the stores the flag
// covers (the last arrival at the LDS counter: condition (3) of that guide), the flag is an sc1 store polled by sc1 loads, and a wave
// loads only after ITS OWN poll has matched.
#ifndef MH_ZV_TWO_STAGE
#define MH_ZV_TWO_STAGE 1
#endif
template <class TP>
struct ZvCols
{
   static constexpr int NV = Tree<TP>::total_dofs();
   static constexpr int body_of(int d)
   {
      for (int j = 0; j < TP::N; j++)
         if (d >= Tree<TP>::dof_ofs(j) && d < Tree<TP>::dof_ofs(j) + Tree<TP>::ndof(j))
            return j;
      return 0;
   }
   static constexpr bool trunk(int d) { return Split<TP>::is_trunk(body_of(d)); }
   // the wave of the inertia job that folds the limb this dof belongs to (-1: a trunk dof)
   static constexpr int aba_owner(int d) { return trunk(d) ? -1 : Split<TP>::owner(Split<TP>::limb_index_of_body(body_of(d))); }
   static constexpr int limb_number(int d)
   { // how many limb columns come before column d
      int n = 0;
      for (int c = 0; c < d; c++)
         n += trunk(c) ? 0 : 1;
      return n;
   }
};
// the columns of wave W's limbs (OWNER = W) or of the trunk (OWNER = -1), global -> registers (sc1 loads) -> the lane's LDS row
template <class TP, typename T, int OWNER>
struct ZvColRegs
{
   static constexpr int count()
   {
      int n = 0;
      for (int d = 0; d < ZvCols<TP>::NV; d++)
         n += ZvCols<TP>::aba_owner(d) == OWNER ? 1 : 0;
      return n;
   }
   T r[count() > 0 ? count() : 1];
   template <int D = 0, int K = 0>
   MH_DEV void issue(const T *src, int lane)
   {
      if constexpr (D < ZvCols<TP>::NV)
      {
         if constexpr (ZvCols<TP>::aba_owner(D) == OWNER)
         {
            r[K] = __hip_atomic_load(src + D * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            issue<D + 1, K + 1>(src, lane);
         }
         else
            issue<D + 1, K>(src, lane);
      }
   }
   template <int D = 0, int K = 0>
   MH_DEV void commit(lds_ptr<T> row) const
   {
      if constexpr (D < ZvCols<TP>::NV)
      {
         if constexpr (ZvCols<TP>::aba_owner(D) == OWNER)
         {
            row[D] = r[K];
            commit<D + 1, K + 1>(row);
         }
         else
            commit<D + 1, K>(row);
      }
   }
   // the sentinel back into the columns this wave has consumed (self-signalling stage one, below)
   template <int D = 0>
   MH_DEV void reset(T *dst, int lane) const
   {
      if constexpr (D < ZvCols<TP>::NV && sizeof(T) == 8)
      {
         if constexpr (ZvCols<TP>::aba_owner(D) == OWNER)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst) + D * 64 + lane, 0x7ff4a5a57ff4a5a5ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         reset<D + 1>(dst, lane);
      }
   }
};
// ---- Stage one without a flag (round 5): the limb columns SIGNAL THEMSELVES.
// Stamps of the flag form (profiles/r05_zv_phase_stamps_prepass_rcp.txt): limb barrier of the bias job 7.00 us, columns stored 7.35, stores
// acknowledged 7.75, flag A stored 7.90, seen by the inertia job 8.30, columns fetched 8.95 -- two round trips on the producer's side (drain,
// flag) and two on the consumer's (poll, fetch) for 15 KB.  Each element of a column is ONE aligned 8-byte store and one 8-byte load: a
// reader sees either what was there before or the value, never a mixture.  So the hand-off matrix holds a SENTINEL between launches -- a
// signalling-NaN bit pattern, which no arithmetic instruction can produce (tau - h comes out of a v_add_f64: a NaN result is quieted) --,
// the producer just stores its columns (sc1, no drain, no count, no flag) and every consumer wave polls THE COLUMNS of its own limbs (sc1
// loads) until no lane that holds a configuration reads the sentinel any more: one round trip behind the store's arrival.  The wave then
// writes the sentinel back (sc1; ordered behind its own loads of the same addresses; the next launch on the stream starts behind this
// kernel's end).  The bit patterns are compared as integers (the build's -ffinite-math-only knows no NaN values).
// What the epoch in the flag used to give for free -- a producer that publishes AFTER its consumer gave up cannot be taken for the next
// launch's -- is kept by a poison word: a consumer that runs into the wall-clock limit sets it (device word flags[ZV_POISON_WORD] of the
// context, and error[1] in mapped host memory beside the error word), every later consumer of the context gives up at once (NaN rows,
// MH_ERR_HIP at the next synchronisation point) until the host has seen error[1], waited for the device, refilled the matrix with
// sentinels and cleared both (mh_api.hip: zv_launch).  Stage two (the trunk's columns under flag B) has a microsecond of slack and stays
// as it was.
#ifndef MH_ZV_SELF_SIGNAL
#define MH_ZV_SELF_SIGNAL 1 // 0: stage one under flag A, as in round 4 (A/B measurements)
#endif
constexpr unsigned long long ZV_SENTINEL = 0x7ff4a5a57ff4a5a5ull; // (both halves equal: the host fills with a 32-bit pattern)
constexpr int ZV_POISON_WORD = 3;                                  // index into the context's flag words (group 0's line)
template <typename T>
MH_DEV bool zv_is_sentinel(T v)
{
   if constexpr (sizeof(T) == 8)
      return __builtin_bit_cast(unsigned long long, v) == ZV_SENTINEL;
   else
      return false;
}
template <class TP, typename T, int OWNER>
MH_DEV bool zv_cols_pending(const ZvColRegs<TP, T, OWNER> &c)
{
   bool p = false;
#pragma unroll
   for (int i = 0; i < ZvColRegs<TP, T, OWNER>::count(); i++)
      p = p || zv_is_sentinel(c.r[i]);
   return p;
}
// the limb columns of wave OWNER into its lanes' LDS rows, polled until they are all there; false: gave up (wall-clock limit, or the
// context is poisoned)
template <class TP, typename T, int OWNER>
MH_DEV bool zv_take_cols(const ZvSync &sy, const T *src, T *reset, lds_ptr<T> row, int lane, bool active, int poison)
{
   ZvColRegs<TP, T, OWNER> c;
   bool ok = poison == 0;
   if (ok)
   {
      c.issue(src, lane);
      if (__builtin_amdgcn_ballot_w64(active && zv_cols_pending(c)) != 0)
      {
         const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
         for (;;)
         {
            __builtin_amdgcn_s_sleep(1);
            c.issue(src, lane);
            if (__builtin_amdgcn_ballot_w64(active && zv_cols_pending(c)) == 0)
               break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)sy.wait_ticks)
            {
               ok = false;
               break;
            }
         }
      }
   }
   if (ok)
   {
      c.commit(row);
      c.reset(reset, lane);
   }
   else if ((threadIdx.x & 63) == 0)
   { // (whatever arrives in this matrix from now on may be a late producer's: nothing of it is trusted until the host has refilled it)
      __hip_atomic_store(sy.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(sy.error + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(sy.flags + ZV_POISON_WORD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   return ok;
}
// columns of this group's hand-off matrix from the lane's LDS row, write-through.  SHARE = -1: the trunk dofs' columns; 0..2: the limb dofs'
// columns number % 3 == SHARE.  All LDS reads first, then all stores (read and stored one by one -- the columns picked by a run-time
// share -- the seven columns of a wave took 0.5 us: a chain of LDS round trips).
template <class TP, typename T, int SHARE>
struct ZvPublish
{
   static constexpr bool mine(int d) { return SHARE < 0 ? ZvCols<TP>::trunk(d) : (!ZvCols<TP>::trunk(d) && ZvCols<TP>::limb_number(d) % 3 == SHARE); }
   static constexpr int count()
   {
      int n = 0;
      for (int d = 0; d < ZvCols<TP>::NV; d++)
         n += mine(d) ? 1 : 0;
      return n;
   }
   T r[count() > 0 ? count() : 1];
   template <int D = 0, int K = 0>
   MH_DEV void read(lds_ptr<T> row)
   {
      if constexpr (D < ZvCols<TP>::NV)
      {
         if constexpr (mine(D))
         {
            r[K] = row[D];
            read<D + 1, K + 1>(row);
         }
         else
            read<D + 1, K>(row);
      }
   }
   // the rows hold h, the efforts arrived on their own (MH_ZV_TAU_LATE): tau - h is formed here
   template <int D = 0, int K = 0>
   MH_DEV void read_minus(lds_ptr<T> tau_row, lds_ptr<T> row)
   {
      if constexpr (D < ZvCols<TP>::NV)
      {
         if constexpr (mine(D))
         {
            r[K] = tau_row[D] - row[D];
            read_minus<D + 1, K + 1>(tau_row, row);
         }
         else
            read_minus<D + 1, K>(tau_row, row);
      }
   }
   template <int D = 0, int K = 0>
   MH_DEV void store(T *dst, int lane) const
   {
      if constexpr (D < ZvCols<TP>::NV)
      {
         if constexpr (mine(D))
         {
            __hip_atomic_store(dst + D * 64 + lane, r[K], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            store<D + 1, K + 1>(dst, lane);
         }
         else
            store<D + 1, K>(dst, lane);
      }
   }
   static MH_DEV void run(T *dst, lds_ptr<T> row, int lane)
   {
      ZvPublish p;
      p.read(row);
      p.store(dst, lane);
   }
   static MH_DEV void run_minus(T *dst, lds_ptr<T> tau_row, lds_ptr<T> row, int lane)
   {
      ZvPublish p;
      p.read_minus(tau_row, row);
      p.store(dst, lane);
   }
};
// The bias job's efforts are needed LAST (tau - h, when the columns are published) and are a third of what it stages: with MH_ZV_TAU_LATE
// they are requested with the other rows but stored to LDS rows of their own only in front of the limb barrier, the walks start as soon as
// q and qd are there (46 KB at the ~11 bytes per clock a CU takes in: 1.25 us; 31 KB: ~0.95), write h, and the publishing waves subtract.
// Measured (profiles/r05_ab_tau_late.txt): 14.15-14.26 us per step either way on one box -- the walks do not start sooner by what the
// staging saves.  Off.
#ifndef MH_ZV_TAU_LATE
#define MH_ZV_TAU_LATE 0 // 1: the efforts stored late, tau - h formed by the publishing waves (experiment)
#endif
template <class TP>
constexpr int zv_bias_extra_rows()
{
   return MH_ZV_TAU_LATE && MH_ZV_TWO_STAGE ? Tree<TP>::total_dofs() : 0;
}
// bias job of group k, two-stage hand-off: taup = this launch's hand-off matrices [groups][nv][64]
template <class TP, typename T>
MH_DEV void zv_bias_group2(const Args<T> &A, long k, lds_ptr<T> lds, T *taup, const ZvSync &sy)
{
   using S = Split<TP>;
   using CX = Ctx<T, true, true, std::conditional_t<MH_RNEA_PRE != 0, RneaPreStore<TP>, SplitStore<TP>>, false, MH_ZV_TAU_LATE ? 0 : 1>;
   const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const int lane = threadIdx.x & 63;
   constexpr int nq = Tree<TP>::total_cfgs(), nv = Tree<TP>::total_dofs(); // (dense index maps: the model's nq, nv -- known without the kernel-argument segment)
   const lds_ptr<T> lxc = lds, lst = lxc + S::n_limbs() * 6 * 64, lq = lst + S::RNEA_TRUNK_SLOTS * 64, lqd = lq + 64 * nq, lx = lqd + 64 * nv;
   const lds_ptr<T> lt = lx + 64 * nv; // MH_ZV_TAU_LATE: the efforts' rows
   ZV_STAMP(0, 14);
   const long cfg0 = k * 64;
   const int rows = (int)(A.B - cfg0 < 64 ? A.B - cfg0 : 64);
   const bool active = lane < rows;
   __shared__ int zv_limb_waves; // waves 1-3 that have published their columns and seen them acknowledged
   if (threadIdx.x == 0)
      zv_limb_waves = 0;
   ZV_STAMP(0, 0);
   RowRegs<T, MH_ZV_TAU_LATE ? Tree<TP>::total_dofs() : 1, 256> rtau;
   if constexpr (MH_ZV_TAU_LATE)
   {
      RowRegs<T, Tree<TP>::total_cfgs(), 256> rq;
      RowRegs<T, Tree<TP>::total_dofs(), 256> rd;
      rq.issue(A.q + cfg0 * nq, rows), rd.issue(A.qd + cfg0 * nv, rows), rtau.issue(A.in3 + cfg0 * nv, rows);
      rq.commit(lq), rd.commit(lqd);
      zv_lds_barrier(); // (LDS only: the efforts stay in flight)
   }
   else
   {
      wave_stage_in<T, Tree<TP>::total_cfgs(), Tree<TP>::total_dofs(), 256>(lq, lqd, lx, A.q + cfg0 * nq, A.qd + cfg0 * nv, A.in3 + cfg0 * nv, rows);
      __syncthreads();
   }
   CX cx;
   fill_ctx<T>(cx, A, active ? cfg0 + lane : cfg0);
   cx.coriolis = 1, cx.accel = 0;
   cx.lq = lq + lane * nq, cx.lqd = lqd + lane * nv, cx.lx = lx + lane * nv, cx.lo = cx.lx;
   cx.wave = wave;
   cx.xbase = lxc + lane;
   cx.st.lbase = lst + lane;
   cx.st.gbase = nullptr, cx.st.stride = 0, cx.st.lane = 0;
   ZV_STAMP(0, 1);
   if (active)
      split_rnea_limbs<TP, 0, T, CX>(cx);
   ZV_STAMP(0, 2);
   if constexpr (MH_ZV_TAU_LATE)
      rtau.commit(lt);
   __syncthreads(); // the limbs' entries of every row are final; the limbs' wrenches are parked for the trunk pass
   ZV_STAMP(0, 3);
   T *const dst = taup + k * 64 * nv;
   int *const flags = sy.flags + k * ZV_SYNC_STRIDE;
   // (Storing tau - h of a joint to its column the moment it is formed -- no publishing pass at all -- was measured: the stores slow the
   // limbs' walks by 0.45 us, flag A comes 0.3 us earlier, flag B 0.4 later, the step takes as long: profiles/r04_zv_two_stage.txt.)
   if (wave == 0)
   {
      if (active)
         rnea_trunk_roots<TP, T, CX>(cx);
      ZV_STAMP(0, 4);
      if constexpr (MH_ZV_TAU_LATE)
         ZvPublish<TP, T, -1>::run_minus(dst, lt + lane * nv, lx + lane * nv, lane);
      else
         ZvPublish<TP, T, -1>::run(dst, lx + lane * nv, lane);
      ZV_STAMP(0, 5);
#ifdef MH_ZV_TEST_FLAG_BEFORE_DRAIN // tests/test_handoff_isa.py compiles this ONCE, to ISA text only, to prove that its checks catch a flag
                                    // that can overtake its columns; it is never linked into anything
      if (lane == 0)
         __hip_atomic_store(flags, sy.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the columns have been acknowledged by memory ...
      ZV_STAMP(0, 6);
#if !defined(MH_ZV_TEST_FLAG_BEFORE_DRAIN) && !defined(MH_ZV_TEST_NO_FLAG)
      if (lane == 0) // ... so flag B, stored by the wave that stored them, is never seen ahead of them
         __hip_atomic_store(flags, sy.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      ZV_STAMP(0, 7);
   }
   else
   {
      ZV_STAMP(0, 4);
#if !(MH_ZV_SELF_SIGNAL && defined(MH_ZV_TEST_NO_FLAG)) // (the test build of a producer that never signals: here, one that never publishes)
      if constexpr (MH_ZV_TAU_LATE)
      {
         if (wave == 1)
            ZvPublish<TP, T, 0>::run_minus(dst, lt + lane * nv, lx + lane * nv, lane);
         else if (wave == 2)
            ZvPublish<TP, T, 1>::run_minus(dst, lt + lane * nv, lx + lane * nv, lane);
         else
            ZvPublish<TP, T, 2>::run_minus(dst, lt + lane * nv, lx + lane * nv, lane);
      }
      else if (wave == 1)
         ZvPublish<TP, T, 0>::run(dst, lx + lane * nv, lane);
      else if (wave == 2)
         ZvPublish<TP, T, 1>::run(dst, lx + lane * nv, lane);
      else
         ZvPublish<TP, T, 2>::run(dst, lx + lane * nv, lane);
#endif
      ZV_STAMP(0, 5);
#if !MH_ZV_SELF_SIGNAL
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's columns have been acknowledged ...
      ZV_STAMP(0, 6);
      int before = 0;
      if (lane == 0) // ... and the wave that learns it is the last of the three to say so stores flag A for all of them
         before = __hip_atomic_fetch_add(&zv_limb_waves, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifndef MH_ZV_TEST_NO_FLAG // (tests/test_gpu_edge_cases.py builds ONE code object with this macro into a scratch directory: a producer that never
                          // signals, to see the consumer give up, write NaN rows and raise the error word -- never shipped)
      if (lane == 0 && before == 2)
         __hip_atomic_store(flags + 1, sy.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
#endif // (self-signalling: the columns are their own flag -- nothing to wait for, nothing to count)
      ZV_STAMP(0, 7);
   }
}

// bias job of group k: taup rows = tau - RNEA(q, qd, 0); A.in3 = tau
template <class TP, typename T, bool IDENT>
MH_DEV void zv_bias_group(const Args<T> &A, long k, lds_ptr<T> lds, T *taup, const ZvSync &sy)
{
   using S = Split<TP>;
   using CX = Ctx<T, true, IDENT, std::conditional_t<MH_RNEA_PRE != 0, RneaPreStore<TP>, SplitStore<TP>>, false, 1>;
   const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const int lane = threadIdx.x & 63;
   const int nq = A.m.nq, nv = A.m.nv;
   const lds_ptr<T> lxc = lds, lst = lxc + S::n_limbs() * 6 * 64, lq = lst + S::RNEA_TRUNK_SLOTS * 64, lqd = lq + 64 * nq, lx = lqd + 64 * nv;
   ZV_STAMP(0, 14);
   const long cfg0 = k * 64;
   const int rows = (int)(A.B - cfg0 < 64 ? A.B - cfg0 : 64);
   const bool active = lane < rows;
   __shared__ int zv_same_l2;
   ZV_STAMP(0, 0);
   wave_stage_in<T, Tree<TP>::total_cfgs(), Tree<TP>::total_dofs(), 256>(lq, lqd, lx, A.q + cfg0 * nq, A.qd + cfg0 * nv, A.in3 + cfg0 * nv, rows);
   __syncthreads();
   ZV_STAMP(0, 1);
   CX cx;
   fill_ctx<T>(cx, A, active ? cfg0 + lane : cfg0);
   cx.coriolis = 1, cx.accel = 0;
   cx.lq = lq + lane * nq, cx.lqd = lqd + lane * nv, cx.lx = lx + lane * nv, cx.lo = cx.lx;
   cx.wave = wave;
   cx.xbase = lxc + lane;
   cx.st.lbase = lst + lane;
   cx.st.gbase = nullptr, cx.st.stride = 0, cx.st.lane = 0;
   if (active)
      split_rnea_limbs<TP, 0, T, CX>(cx);
   ZV_STAMP(0, 2);
   if (threadIdx.x == 64)
      zv_same_l2 = sy.same_l2 && __hip_atomic_load(sy.flags + k * ZV_SYNC_STRIDE + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == zv_mail_key(sy.epoch, zv_xcc_id());
   __syncthreads();
   ZV_STAMP(0, 3);
   const bool same = zv_same_l2 != 0;
   if (active && wave == 0)
      rnea_trunk_roots<TP, T, CX>(cx);
   // (Which L2 the rows will be read from was looked up before the barrier above.)  The inertia job of these configurations started
   // together with this one and left the id of its XCD in the mailbox.  The same id as this workgroup's own: one L2 serves both,
   // and rows and flag left there (stores that are NOT write-through) are what that job's sc1 polls and loads find: flag stored -> seen
   // 0.24 us and the rows fetched in 0.67 us, against 0.36 and 0.85 us through memory.  Anything else in the mailbox (not started,
   // another XCD, same_l2 off): write-through.  The inertia job empties its mailbox when it has consumed the rows, so that a replay
   // of a captured launch (same epoch) never finds the previous replay's.
   // (The flags of 32 groups shared a 128-byte line at first.  A consumer that arrives BEFORE the flag then saw it 1.6-1.9 us after its
   // store -- the line bounced between 32 pollers and 32 writers -- which is why speeding up the inertia job alone made the step
   // slower; with a line per group a waiting consumer sees it as fast as a late one: profiles/r03_presincos_experiment.txt.)
   ZV_STAMP(0, 4);
   __syncthreads();
   // (Publishing the limbs' entries from the idle waves while wave 0 runs the trunk was measured: the masked 8-byte stores write every
   // line in pieces -- 1.4 + 1.1 us for the two parts against 0.55 us for whole rows.)
   zv_publish_rows<T, 256>(taup + cfg0 * nv, lx, rows * nv, same);
   ZV_STAMP(0, 5);
#ifdef MH_ZV_TEST_FLAG_BEFORE_DRAIN // tests/test_handoff_isa.py compiles this ONCE, to ISA text only, to prove that its checks catch a flag
                                    // that can overtake its rows; it is never linked into anything
   if (threadIdx.x == 0)
      __hip_atomic_store(sy.flags + k * ZV_SYNC_STRIDE, sy.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave: its stores have been acknowledged by the L2 / by memory ...
   ZV_STAMP(0, 6);
   __syncthreads();
#if !defined(MH_ZV_TEST_FLAG_BEFORE_DRAIN) && !defined(MH_ZV_TEST_NO_FLAG)
   if (threadIdx.x == 0) // ... so the flag, stored behind the barrier (the same way as the rows), is never seen ahead of them
   {
      if (same)
         __hip_atomic_store(sy.flags + k * ZV_SYNC_STRIDE, sy.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else
         __hip_atomic_store(sy.flags + k * ZV_SYNC_STRIDE, sy.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
#endif
   ZV_STAMP(0, 7);
}

// inertia job of group k: qdd rows from q and the bias rows; A.out = qdd
template <class TP, typename T, bool IDENT>
MH_DEV void zv_aba_group(const Args<T> &A, long k, lds_ptr<T> lds, const T *taup, const ZvSync &sy)
{
   using S = Split<TP>;
#ifndef MH_ZV_DIRECT_OUT
#define MH_ZV_DIRECT_OUT 0 // experiment knob: 1 = accelerations stored per lane as they are formed instead of LDS rows + one coalesced copy
#endif
   using CX = Ctx<T, true, IDENT, ZvStore<TP>, false, MH_ZV_DIRECT_OUT ? 2 : 0>;
   const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const int lane = threadIdx.x & 63;
   const int nq = A.m.nq, nv = A.m.nv;
   // LDS map: exchange [n_limbs * 21][64] | trunk hand-over slots [ZV_TRUNK_SLOTS][64] | [64][nq] q | [64][nv] tau' | [64][nv] qdd.
   // The results get rows of their own: every wave folds the trunk at its own pace, reading the trunk's tau' entries, while wave 0 --
   // which writes the trunk's accelerations -- may already be in its outward sweep (written in place they raced: seen as wrong limb
   // accelerations of the other waves on a tree whose wave 0 is the first to finish its fold).
   const lds_ptr<T> lxc = lds, lst = lxc + S::n_limbs() * ZV_XW * 64, lq = lst + S::ZV_TRUNK_SLOTS * 64, lx = lq + 64 * nq, lres = lx + 64 * nv;
   ZV_STAMP(1, 14);
   const long cfg0 = k * 64;
   const int rows = (int)(A.B - cfg0 < 64 ? A.B - cfg0 : 64);
   const bool active = lane < rows;
   ZV_STAMP(1, 0);
   if (sy.same_l2 && threadIdx.x == 0) // where this job runs, for the bias job of the same configurations (see there)
      __hip_atomic_store(sy.flags + k * ZV_SYNC_STRIDE + 32, zv_mail_key(sy.epoch, zv_xcc_id()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   zv_stage_rows<T, Tree<TP>::total_cfgs(), 256>(lq, A.q + cfg0 * nq, rows);
   __syncthreads();
   ZV_STAMP(1, 1);
   CX cx;
   fill_ctx<T>(cx, A, active ? cfg0 + lane : cfg0);
   cx.lq = lq + lane * nq, cx.lqd = lq, cx.lx = lx + lane * nv, cx.lo = lres + lane * nv;
   cx.wave = wave;
   cx.xbase = lxc + lane;
   cx.st.lbase = lst + lane;
   cx.st.gbase = nullptr, cx.st.stride = 0, cx.st.lane = 0;
#ifdef MH_ZV_PROBE
   cx.own = (unsigned long long)k;
#endif
#ifdef MH_ZV_TWICE // experiment: the inward limb phase a second time, through a warm instruction cache (stamps 12 = first pass done)
#pragma unroll 1
   for (int rep = 0; rep < 2; rep++)
   {
      asm volatile("" ::: "memory");
      if (active)
         zv_limbs_in<TP, 0, T, CX>(cx);
      if (rep == 0)
      {
         ZV_STAMP(1, 12);
         __syncthreads();
         ZV_STAMP(1, 13);
      }
   }
#else
   if (active) // (lane 0 of every wave is active, so each wave reaches the barrier a staged trunk carries in here)
      zv_limbs_in<TP, 0, T, CX>(cx);
#endif
   ZV_STAMP(1, 2);
   __syncthreads(); // every limb's (and sub-trunk's) articulated inertia is in the exchange area
   ZV_STAMP(1, 3);
   if (active)
      zv_roots_in<TP, T, CX, (S::staged() ? 2 : 1)>(cx);
   ZV_STAMP(1, 4);
   // (Polling from a spare wave WHILE the others run the root body's step was measured: the flag is then seen ~0.7 us LATER -- a poll that
   // reaches memory before the flag store leaves the old line behind for the polls after it -- 17.6 against 16.1 us per step.  Publishing
   // the limbs' rows ahead of the trunk's entries, fetched by the least-loaded wave during the limb phase: 19.6 us, every extra
   // store -> flag -> poll -> load chain through memory costs ~3 us.  Rows and flag kept in the L2 by sc0 stores where both jobs prove to sit
   // behind the same one (HW_REG_XCC_ID through a mailbox; 64 of 64 groups did): fetch 0.67 instead of 0.85 us, step time unchanged -- the
   // flag is up before the root step ends.  The plain form below is the fastest of the four.)
   __shared__ int zv_gave_up;
   if (wave == 0)
   {
      const bool seen = zv_wait(sy, k);
      if (lane == 0)
         zv_gave_up = seen ? 0 : 1;
   }
   __syncthreads(); // the polling wave has seen the flag: now every wave may load the rows
   if (zv_gave_up)
   { // (the whole workgroup takes this branch) the bias rows never came: NaN instead of accelerations formed from whatever the scratch
     // matrix holds, and no fold; the error word is set, the host reports MH_ERR_HIP at its next synchronisation point (mh_api.hip)
      for (int i = threadIdx.x; i < rows * nv; i += 256) // (as a bit pattern: the build's -ffinite-math-only knows no NaN values)
      {
         if constexpr (sizeof(T) == 8)
            reinterpret_cast<unsigned long long *>(A.out)[cfg0 * nv + i] = 0x7ff8000000000000ull;
         else
            reinterpret_cast<unsigned *>(A.out)[cfg0 * nv + i] = 0x7fc00000u;
      }
      return;
   }
   ZV_STAMP(1, 5);
   zv_fetch_rows<T, Tree<TP>::total_dofs(), 256>(lx, taup + cfg0 * nv, rows);
   // The flag goes back to zero once its rows have been consumed: a captured launch is replayed with the SAME epoch (hipGraph), and a flag
   // left standing from the previous replay would let this job read the previous replay's rows.  (Stream order puts the reset before the
   // next launch's bias job; the rows' loads were issued above and this store cannot pass the flag poll it depends on.)
   if (threadIdx.x == 0)
   {
      __hip_atomic_store(sy.flags + k * ZV_SYNC_STRIDE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sy.same_l2)
         __hip_atomic_store(sy.flags + k * ZV_SYNC_STRIDE + 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   __syncthreads(); // bias rows staged; nobody reads the exchange area's inertias any more
   ZV_STAMP(1, 6);
   asm volatile("" ::: "memory");
   if (active) // (lane 0 of every wave is active: each wave reaches the barriers the fold carries in here)
      zv_fold_out<TP, 0, T, CX>(cx);
   ZV_STAMP(1, 10);
   if constexpr (!MH_ZV_DIRECT_OUT)
   {
      __syncthreads();
      wave_copy_out<T, 256>(A.out + cfg0 * nv, lres, rows * nv);
   }
   ZV_STAMP(1, 11);
}

// polls one word of the group's flag line until it holds this launch's epoch (see zv_wait): true when seen, false after the wall-clock limit
MH_DEV bool zv_wait_word(const ZvSync &sy, const int *f)
{
   if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sy.epoch)
      return true;
   const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
   for (;;)
   {
      __builtin_amdgcn_s_sleep(1);
      if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sy.epoch)
         return true;
      if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)sy.wait_ticks)
      {
         if ((threadIdx.x & 63) == 0)
            __hip_atomic_store(sy.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
         return false;
      }
   }
}
// inertia job of group k, two-stage hand-off (see zv_bias_group2)
template <class TP, typename T, bool STEP>
MH_DEV void zv_aba_group2(const Args<T> &A, long k, lds_ptr<T> lds, const T *taup, const ZvSync &sy)
{
   using S = Split<TP>;
   using CX = Ctx<T, true, true, ZvStore<TP>, false, 0>;
   const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const int lane = threadIdx.x & 63;
   constexpr int nq = Tree<TP>::total_cfgs(), nv = Tree<TP>::total_dofs(); // (dense index maps: the model's nq, nv -- known without the kernel-argument segment)
   // LDS map: as zv_aba_group
   const lds_ptr<T> lxc = lds, lst = lxc + S::n_limbs() * ZV_XW * 64, lq = lst + S::ZV_TRUNK_SLOTS * 64, lx = lq + 64 * nq, lres = lx + 64 * nv;
   ZV_STAMP(1, 14);
   const long cfg0 = k * 64;
   const int rows = (int)(A.B - cfg0 < 64 ? A.B - cfg0 : 64);
   const bool active = lane < rows;
   __shared__ int zv_gave_up;
   if (threadIdx.x == 0)
      zv_gave_up = 0;
   ZV_STAMP(1, 0);
#if MH_ZV_SELF_SIGNAL
   // the context's poison word (see zv_take_cols), requested with the rows and long there when it is looked at
   const int poison = __hip_atomic_load(sy.flags + ZV_POISON_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
   zv_stage_rows<T, Tree<TP>::total_cfgs(), 256>(lq, A.q + cfg0 * nq, rows);
   __syncthreads();
   ZV_STAMP(1, 1);
   CX cx;
   fill_ctx<T>(cx, A, active ? cfg0 + lane : cfg0);
   cx.lq = lq + lane * nq, cx.lqd = lq, cx.lx = lx + lane * nv, cx.lo = lres + lane * nv;
   cx.wave = wave;
   cx.xbase = lxc + lane;
   cx.st.lbase = lst + lane;
   cx.st.gbase = nullptr, cx.st.stride = 0, cx.st.lane = 0;
#ifdef MH_ZV_PROBE
   cx.own = (unsigned long long)k;
#endif
   if (active) // (lane 0 of every wave is active, so each wave reaches the barrier a staged trunk carries in here)
      zv_limbs_in<TP, 0, T, CX>(cx);
   ZV_STAMP(1, 2);
   __syncthreads(); // every limb's (and sub-trunk's) articulated inertia is in the exchange area
   ZV_STAMP(1, 3);
   if (active)
      zv_roots_in<TP, T, CX, (S::staged() ? 2 : 1)>(cx);
   ZV_STAMP(1, 4);
   // simulation step: the velocities of the group, requested by ALL threads (each its share of the rows, whether its lane holds a
   // configuration or not), in front of the first poll (tools/isa_handoff.py: every load behind it is a hand-off load) and in flight
   // until the outward sweep is through
   RowRegs<T, STEP ? Tree<TP>::total_dofs() : 1, 256> rqd;
   if constexpr (STEP)
      rqd.issue(A.qd + cfg0 * nv, rows);
   // ---- stage one of the hand-off: the bias efforts of this wave's limbs, straight to its lanes' rows (nobody else reads those entries)
   const T *const src = taup + k * 64 * nv;
   const int *const flags = sy.flags + k * ZV_SYNC_STRIDE;
   const lds_ptr<T> row = lx + lane * nv;
#if MH_ZV_SELF_SIGNAL
   bool seen;
   {
      T *const back = const_cast<T *>(src);
      if (wave == 0)
         seen = zv_take_cols<TP, T, 0>(sy, src, back, row, lane, active, poison);
      else if (wave == 1)
         seen = zv_take_cols<TP, T, 1>(sy, src, back, row, lane, active, poison);
      else if (wave == 2)
         seen = zv_take_cols<TP, T, 2>(sy, src, back, row, lane, active, poison);
      else
         seen = zv_take_cols<TP, T, 3>(sy, src, back, row, lane, active, poison);
   }
   ZV_STAMP(1, 5);
#else
   bool seen = zv_wait_word(sy, flags + 1);
   ZV_STAMP(1, 5);
   if (wave == 0)
   {
      ZvColRegs<TP, T, 0> c;
      c.issue(src, lane), c.commit(row);
   }
   else if (wave == 1)
   {
      ZvColRegs<TP, T, 1> c;
      c.issue(src, lane), c.commit(row);
   }
   else if (wave == 2)
   {
      ZvColRegs<TP, T, 2> c;
      c.issue(src, lane), c.commit(row);
   }
   else
   {
      ZvColRegs<TP, T, 3> c;
      c.issue(src, lane), c.commit(row);
   }
#endif
   // a first look at flag B, in flight during the early limbs' fold (a poll is a round trip to memory: 0.45 us on the stamps)
   const int b_first = __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   zv_lds_barrier(); // nobody reads the exchange area's inertias any more (the fold's records go over them)
   ZV_STAMP(1, 6);
   asm volatile("" ::: "memory");
   // ---- stage two rides in the fold: the trunk's bias efforts, requested behind the early limbs' fold, stored behind the late limbs'
   struct TrunkTau : ZvNoHook
   {
      mutable ZvColRegs<TP, T, -1> c;
      const ZvSync &sy;
      const int *flags;
      const T *src;
      lds_ptr<T> row;
      int lane, b_first;
      bool &seen;
      MH_DEV void after_early() const
      {
         if (b_first != sy.epoch)
            seen = zv_wait_word(sy, flags) && seen;
         c.issue(src, lane);
      }
      MH_DEV void after_late() const { c.commit(row); } // (every wave stores the trunk's entries of its own lanes' rows: the same values four times)
   };
   const TrunkTau trunk{{}, {}, sy, flags, src, row, lane, b_first, seen};
   if (active) // (lane 0 of every wave is active: each wave reaches the barriers the fold carries in here)
      zv_fold_out<TP, 0, T, CX>(cx, trunk);
   ZV_STAMP(1, 10);
   if (!seen && lane == 0)
      zv_gave_up = 1;
   __syncthreads();
   // The flags go back to zero once their columns have been consumed: a captured launch is replayed with the SAME epoch (hipGraph), and a flag
   // left standing from the previous replay would let this job read the previous replay's columns.  (Stream order puts the reset before the
   // next launch's bias job; every wave's polls and loads lie in front of the barrier above.)
   if (threadIdx.x == 0)
   {
      __hip_atomic_store(const_cast<int *>(flags), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(const_cast<int *>(flags) + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   if (zv_gave_up)
   { // (the whole workgroup takes this branch) columns that never came: NaN instead of accelerations formed from whatever the scratch
     // matrix holds; the error word is set, the host reports MH_ERR_HIP at its next synchronisation point (mh_api.hip)
      for (int i = threadIdx.x; i < rows * nv; i += 256) // (as a bit pattern: the build's -ffinite-math-only knows no NaN values)
         reinterpret_cast<unsigned long long *>(A.out)[cfg0 * nv + i] = 0x7ff8000000000000ull;
      if constexpr (STEP)
      { // a simulation step: the new state of those configurations is NaN too
         for (int i = threadIdx.x; i < rows * nq; i += 256)
            reinterpret_cast<unsigned long long *>(A.q_next)[cfg0 * nq + i] = 0x7ff8000000000000ull;
         for (int i = threadIdx.x; i < rows * nv; i += 256)
            reinterpret_cast<unsigned long long *>(A.qd_next)[cfg0 * nv + i] = 0x7ff8000000000000ull;
      }
      return;
   }
   if constexpr (STEP)
   { // fused simulation step (a kernel of its own, spec_zv_kernel<.., STEP = true>: the plain call's code stays as it is): q (lq) and the fresh accelerations (lres) of the 64 configurations sit in LDS, the velocities arrive in the
     // rows the bias efforts no longer need (lx) -- integrated in place (MultiBodySystemStateIntegrator.java:365-441, 503-575, 710-733)
     // and streamed out with the accelerations
      rqd.commit(lx);
      __syncthreads();
      if (active)
         integrate_rows<TP, 0, T>(wave, lq + lane * nq, lx + lane * nv, lres + lane * nv, A.dt, T(0.5) * A.dt * A.dt);
      __syncthreads();
      wave_copy_out<T, 256>(A.q_next + cfg0 * nq, lq, rows * nq);
      wave_copy_out<T, 256>(A.qd_next + cfg0 * nv, lx, rows * nv);
   }
   wave_copy_out<T, 256>(A.out + cfg0 * nv, lres, rows * nv);
   ZV_STAMP(1, 11);
}

// One launch, jobs * ceil(B / 64) workgroups (padded to blocks of eight): blocks of eight consecutive workgroup ids share a role, so
// the bias job and the inertia job of the same 64 configurations have ids that differ by a multiple of
// eight -- the dispatcher deals ids round-robin to the eight XCDs, which puts them behind the same L2 -- and the producer's id is lower.
#ifndef MH_ZV_KERNEL_ATTR
#define MH_ZV_KERNEL_ATTR
#endif
// The first five arguments repeat what the jobs need to REQUEST THEIR ROWS -- the three state matrices of the critical jobs, the batch size,
// the number of jobs (which job is this workgroup?) -- as plain leading arguments: the code objects are compiled with
// -amdgpu-kernarg-preload-count (mecano_amd/build.py), so the dispatcher hands those ten dwords over in SGPRs and the staging loads are issued
// without waiting for the kernel-argument segment (0.65 us from the first instruction to the end of the prologue on the stamps: one cold
// read of device memory that every wave of every launch used to sit out before it could ask for anything).
template <class TP, typename T, bool IDENT, bool STEP = false>
__global__ void __launch_bounds__(256) MH_ZV_KERNEL_ATTR spec_zv_kernel(const T *q0, const T *qd0, const T *tau0, long B0, int jobs0, Args<T> A_full, T *taup,
                                                                        ZvSync sy_full)
{
   extern __shared__ double lds_raw[];
   const int blk = (int)blockIdx.x;
   Args<T> A = A_full;
   A.q = q0, A.qd = qd0, A.in3b = tau0, A.B = B0;
   ZvSync sy = sy_full;
   sy.jobs = jobs0;
   // bias and inertia job alternate in blocks of eight; the inverse dynamics job of the pair call (five microseconds of slack) takes the ids
   // behind all of them, so that the two jobs on the critical path are dispatched first (15.55 -> 15.35 us per step)
   const int padded = (int)gridDim.x / sy.jobs;
   const bool tail = sy.jobs == 3 && blk >= 2 * padded;
   const int role = tail ? 2 : (blk >> 3) & 1;
   const long k = tail ? (long)(blk - 2 * padded) : (long)(blk / 16) * 8 + (blk & 7);
   if (k * 64 >= A.B)
      return;
   ZV_STAMP(role, 15);
   if (role == 0)
   {
      Args<T> A2 = A;
      A2.in3 = A.in3b;
      if constexpr (IDENT && MH_ZV_TWO_STAGE && sizeof(T) == 8)
         zv_bias_group2<TP, T>(A2, k, (lds_ptr<T>)lds_raw, taup, sy);
      else
         zv_bias_group<TP, T, IDENT>(A2, k, (lds_ptr<T>)lds_raw, taup, sy);
   }
   else if (role == 1)
   {
      Args<T> A2 = A;
      A2.out = A.outb;
      if constexpr (IDENT && MH_ZV_TWO_STAGE && sizeof(T) == 8)
         zv_aba_group2<TP, T, STEP>(A2, k, (lds_ptr<T>)lds_raw, taup, sy);
      else
         zv_aba_group<TP, T, IDENT>(A2, k, (lds_ptr<T>)lds_raw, taup, sy);
   }
   else
   {
      split_group<TP, T, 0, IDENT, true>(A, k, (A.B + 63) / 64, (lds_ptr<T>)lds_raw);
      ZV_STAMP(2, 1);
   }
}

// ============================================================================================ device-filling batches: two launches
// Beyond one workgroup per CU and job the two jobs stop running side by side, and the one-job forward dynamics of mh_spec_kernels.h holds
// a SIMD with ONE wave (446 registers).  The inertia job fits 256 registers; what kept it at one workgroup per CU was 129 KB of LDS.  Here
// the bias job is a launch of its own that runs first and leaves, for every configuration, the rows tau - h(q, qd) and (cos, sin) of every
// revolute joint; the inertia job then needs no q, no sincos and no flag, and its LDS shrinks to the limbs' exchange records plus the
// trunk's hand-over slots (ZvbStore: the root body in registers): the bias rows are staged into the exchange area once the inward sweep
// has consumed it, the fold's records sit behind them and the accelerations are written in place -- 69 KB for the humanoid, two
// workgroups per CU, two waves per SIMD.
template <class TP>
struct ZvbPlan
{
   using S = Split<TP>;
   static constexpr int NV = Tree<TP>::total_dofs();
   static constexpr int x_slots()
   { // exchange area: the limbs' inertia records during the inward sweep; afterwards [64][nv] bias rows | 12 slots per limb for the fold
      const int in = S::n_limbs() * ZV_XW, fold = NV + S::n_limbs() * 12;
      return in > fold ? in : fold;
   }
   static constexpr int lds_slots() { return x_slots() + ZvbStore<TP>::TRUNK_SLOTS; }
   static constexpr int bias_lds_slots(int nq, int nv) { return S::n_limbs() * 6 + S::RNEA_TRUNK_SLOTS + nq + 2 * nv; }
};


// first launch: rows tau - RNEA(q, qd, 0) (A.in3 = tau) to taup [B][nv], (cos, sin) of the revolute joints to cs [2 n_rev][cs_stride].
// BIAS = false: the plain inverse dynamics of device-filling batches (A.in3 = qdd, rows RNEA(q, qd, qdd) to taup = A.out, cs unused) --
// the tree-split walk of spec_split_kernel in this kernel's persistent loop, i.e. with the next group's rows in flight behind the trunk pass.
template <class TP, typename T, bool IDENT, bool BIAS = true>
__global__ void __launch_bounds__(256, 2) spec_zvb_bias_kernel(Args<T> A, T *taup, T *cs, long cs_stride)
{
   extern __shared__ double lds_raw[];
   using S = Split<TP>;
   using CX = Ctx<T, true, IDENT, std::conditional_t<MH_RNEA_PRE != 0, RneaPreStore<TP>, SplitStore<TP>>, false, BIAS ? 1 : 0, BIAS ? 1 : 0>;
   constexpr int NQ = Tree<TP>::total_cfgs(), NV = Tree<TP>::total_dofs();
   const lds_ptr<T> lds = (lds_ptr<T>)lds_raw;
   const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const int lane = threadIdx.x & 63;
   const int nq = A.m.nq, nv = A.m.nv;
   const lds_ptr<T> lxc = lds, lst = lxc + S::n_limbs() * 6 * 64, lq = lst + S::RNEA_TRUNK_SLOTS * 64, lqd = lq + 64 * nq, lx = lqd + 64 * nv;
   // waves 1-3 stage the rows (wave 0 folds the trunk while the next group's are requested: the workgroup waits for that pass)
   RowRegs<T, NQ, 192> rq;
   RowRegs<T, NV, 192> rd, rx;
   const int loader = (int)threadIdx.x - 64;
   auto request = [&](long k) {
      const long cfg0 = k * 64;
      const int rows = (int)(A.B - cfg0 < 64 ? A.B - cfg0 : 64); // (k is a group of the batch: at least one row)
      rq.issue(A.q + cfg0 * nq, rows, loader), rd.issue(A.qd + cfg0 * nv, rows, loader), rx.issue(A.in3 + cfg0 * nv, rows, loader);
   };
   const long ngroups = (A.B + 63) / 64;
   if (wave != 0)
      request(blockIdx.x);
   for (long k = blockIdx.x; k < ngroups; k += gridDim.x)
   {
      const long cfg0 = k * 64;
      const int rows = (int)(A.B - cfg0 < 64 ? A.B - cfg0 : 64);
      const bool active = lane < rows;
      ZV_STAMP(0, 0);
      if (wave != 0)
         rq.commit(lq, loader), rd.commit(lqd, loader), rx.commit(lx, loader);
      __syncthreads();
      ZV_STAMP(0, 1);
      CX cx;
      fill_ctx<T>(cx, A, active ? cfg0 + lane : cfg0);
      if constexpr (BIAS)
         cx.coriolis = 1, cx.accel = 0;
      cx.lq = lq + lane * nq, cx.lqd = lqd + lane * nv, cx.lx = lx + lane * nv, cx.lo = cx.lx;
      cx.wave = wave;
      cx.xbase = lxc + lane;
      cx.st.lbase = lst + lane;
      cx.st.gbase = nullptr, cx.st.stride = 0, cx.st.lane = 0;
      if constexpr (BIAS)
         cx.cs = cs + cfg0 + lane, cx.cs_stride = cs_stride; // (the scratch is padded to whole groups; inactive lanes never store)
      if (active)
         split_rnea_limbs<TP, 0, T, CX>(cx);
      ZV_STAMP(0, 2);
      __syncthreads();
      ZV_STAMP(0, 3);
      // the next group's rows: in flight during the trunk pass and the copy-out.  (Unconditionally -- the last turn asks for its own group
      // again: under a condition the registers would have to keep the OLD rows alive through the whole loop body, 48 more live registers.)
      if (wave != 0)
         request(k + gridDim.x < ngroups ? k + gridDim.x : k);
      else if (active)
         rnea_trunk_roots<TP, T, CX>(cx);
      ZV_STAMP(0, 4);
      zv_lds_barrier(); // (__syncthreads() would wait for the rows just requested: its release fence drains the vector-memory counter)
      wave_copy_out<T, 256>(taup + cfg0 * nv, lx, rows * nv);
      ZV_STAMP(0, 5);
      zv_lds_barrier(); // the LDS rows are free for the next group
      ZV_STAMP(0, 6);
   }
}

// second launch, group k: qdd rows (A.out) from the bias rows and the (cos, sin) pairs.
// (Requesting the NEXT group's pairs during the fold and carrying them in registers to the next turn was built and measured: the kernel
// sits at 231 of 256 registers, the carried pairs cost 156-184 bytes of scratch per lane and the step got slower, not faster.)
template <class TP, typename T, bool IDENT>
MH_DEV void zvb_aba_group(const Args<T> &A, long k, lds_ptr<T> lds, const T *taup, const T *cs, long cs_stride)
{
   using S = Split<TP>;
   using PL = ZvbPlan<TP>;
   using CX = Ctx<T, true, IDENT, ZvbStore<TP>, false, 0, 2>;
   const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const int lane = threadIdx.x & 63;
   const int nv = A.m.nv;
   // LDS map: exchange area [x_slots][64] | the trunk's hand-over slots [TRUNK_SLOTS][64].  Exchange area, inward sweep: one inertia record
   // of ZV_XW slots per limb; afterwards: [64][nv] bias rows, overwritten by the accelerations | CX::fold_xw slots per limb for the fold.
   const lds_ptr<T> lxc = lds, lst = lxc + PL::x_slots() * 64, lx = lxc, lfold = lxc + 64 * nv;
   const long cfg0 = k * 64;
   const int rows = (int)(A.B - cfg0 < 64 ? A.B - cfg0 : 64);
   const bool active = lane < rows;
   CX cx;
   fill_ctx<T>(cx, A, active ? cfg0 + lane : cfg0);
   cx.lq = lxc, cx.lqd = lxc; // (never read: CSMODE 2 takes q from the caller's matrix)
   cx.lx = lx + lane * nv, cx.lo = cx.lx;
   cx.wave = wave;
   cx.xbase = lxc + lane;
   cx.st.lbase = lst + lane;
   cx.st.gbase = nullptr, cx.st.stride = 0, cx.st.lane = 0;
   cx.cs = const_cast<T *>(cs) + cfg0 + lane, cx.cs_stride = cs_stride; // padded to whole groups: inactive lanes read what nobody uses
#ifdef MH_ZV_PROBE
   cx.own = (unsigned long long)k;
#endif
   ZV_STAMP(1, 0);
   if (active) // (lane 0 of every wave is active, so each wave reaches the barrier a staged trunk carries in here)
      zv_limbs_in<TP, 0, T, CX>(cx);
   ZV_STAMP(1, 2);
   __syncthreads(); // every limb's (and sub-trunk's) articulated inertia is in the exchange area
   ZV_STAMP(1, 3);
#ifndef MH_ZVB_ROWS_EARLY
#define MH_ZVB_ROWS_EARLY 1 // the bias rows are requested before the root body's step and held in registers across it (231 -> 247 registers, no scratch; 219.7 -> 214.5 us at B = 262 144); 0: requested when the exchange area is free
#endif
   RowRegs<T, Tree<TP>::total_dofs(), 256> rt;
   if constexpr (MH_ZVB_ROWS_EARLY)
      rt.issue(taup + cfg0 * nv, rows);
   if (active)
      zv_roots_in<TP, T, CX, (S::staged() ? 2 : 1)>(cx);
   ZV_STAMP(1, 4);
   __syncthreads(); // nobody reads the exchange area's inertias any more
   ZV_STAMP(1, 5);
   if constexpr (!MH_ZVB_ROWS_EARLY)
      rt.issue(taup + cfg0 * nv, rows);
   rt.commit(lx);
   __syncthreads();
   ZV_STAMP(1, 6);
   cx.xbase = lfold + lane;
   asm volatile("" ::: "memory");
   if (active) // (lane 0 of every wave is active: each wave reaches the barriers the fold carries in here)
      zv_fold_out<TP, 0, T, CX>(cx);
   ZV_STAMP(1, 10);
   __syncthreads();
   wave_copy_out<T, 256>(A.out + cfg0 * nv, lx, rows * nv);
   ZV_STAMP(1, 11);
   __syncthreads(); // the exchange area is written again by the next group of this workgroup
   ZV_STAMP(1, 12);
}
template <class TP, typename T, bool IDENT>
__global__ void __launch_bounds__(256, 2) spec_zvb_kernel(Args<T> A, const T *taup, const T *cs, long cs_stride)
{
   extern __shared__ double lds_raw[];
   for (long k = blockIdx.x; k * 64 < A.B; k += gridDim.x)
      zvb_aba_group<TP, T, IDENT>(A, k, (lds_ptr<T>)lds_raw, taup, cs, cs_stride);
}

// ============================================================================================ device-filling batches: ONE launch
// Both jobs in the same workgroup, one after the other, two workgroups per CU.  The two-launch form pays for its seam: the bias rows and 48
// (cos, sin) pairs per configuration travel through memory, each launch stages its inputs and copies its outputs, each loop turn has its
// own barriers and tail (profiles/r04_zvb_phase_stamps.txt: 9.4 + 11.4 us per group, of which ~5 are the seam).  Fused, the inverse dynamics
// of a group walks its limbs with the FORWARD dynamics' limb owners (Split<TP>::owner_sel<1>), so the pair of a limb joint and its
// tau - h stay in the registers of the wave that needs them next; the trunk's go to the trunk's LDS slots; nothing is written to memory but
// the accelerations.  LDS (slots of 64 doubles; humanoid): inverse dynamics [q 31 | qd 30 | tau 30 | limb wrenches 30 | parked trunk
// wrenches 32] = 153; forward dynamics [trunk slots 39, written by wave 0's trunk pass over the rows of q, which nobody reads any more |
// exchange area 105] = 144: 78 KB, two workgroups per CU.  Needs every joint below the root to be revolute or fixed (the later phases
// would read q otherwise, whose rows are gone by then): the benchmark humanoid; other trees keep the two launches.
template <class TP>
struct ZvfPlan
{
   using S = Split<TP>;
   using ST = ZvfStore<TP>;
   static constexpr int NQ = Tree<TP>::total_cfgs(), NV = Tree<TP>::total_dofs();
   static constexpr bool joints_ok()
   {
      for (int j = 0; j < TP::N; j++)
         if (TP::parent[j] >= 0 && TP::type[j] != JT_REVOLUTE && TP::type[j] != JT_FIXED)
            return false;
      return true;
   }
   static constexpr int x_slots()
   { // exchange area: the limbs' inertia records during the inward sweep; afterwards 12 slots per limb for the fold | [64][nv] result rows
      const int in = S::n_limbs() * ZV_XW, fold = S::n_limbs() * 12 + NV;
      return in > fold ? in : fold;
   }
   static constexpr int rnea_slots() { return NQ + 2 * NV + S::n_limbs() * 6 + S::RNEA_TRUNK_SLOTS; }
   static constexpr int aba_slots() { return ST::TRUNK_SLOTS + x_slots(); }
   // the mailed limb's three slots: behind what either phase keeps in LDS -- if two workgroups per CU still fit
   static constexpr int mail_base() { return rnea_slots() > aba_slots() ? rnea_slots() : aba_slots(); }
   static constexpr bool use_mail() { return MH_ZVF_MAIL != 0 && S::staged() && S::mailed_limb() >= 0 && (mail_base() + 3) * 64 * 8 * 2 <= 160 * 1024; }
   static constexpr int lds_slots() { return mail_base() + (use_mail() ? 3 : 0); }
   // the trunk slots are written while the inverse dynamics' exchange area and parking area are still being read: they must fit under
   // the rows of q and qd, which are dead by then
   static constexpr bool usable() { return S::usable() && joints_ok() && ST::TRUNK_SLOTS <= NQ + NV && lds_slots() * 64 * 8 * 2 <= 160 * 1024; }
   // the pair call's last phase: [qdd rows | tau rows | limb wrenches] behind the trunk's slots, in front of the mailed limb's
   static constexpr bool pair_usable() { return usable() && ST::TRUNK_SLOTS + 2 * NV + S::n_limbs() * 6 <= mail_base(); }
   // a simulation step re-stages the rows of q and qd behind the outward sweep, over the (dead) trunk slots and fold records: they must end
   // in front of the result rows
   static constexpr bool step_usable() { return usable() && NQ + NV <= ST::TRUNK_SLOTS + S::n_limbs() * 12; }
};
// ---- The pair call (mh_rnea_aba_f64) of device-filling batches in the SAME launch (round 5).  Inverse dynamics is linear in the joint
// accelerations: tau = h(q, qd, a_root, f_ext) + M(q) qdd, and the fused kernel has h -- its first phase leaves tau_in - h of every joint
// where the bias fold reads it, and they are still there when the group's accelerations have been written.  What is missing is M(q) qdd of
// the caller's qdd: one more walk WITHOUT velocities -- da = X da_parent + S qdd outwards, df = I da + sum X^T df_child inwards,
// tau = tau_in - (tau_in - h) + S^T df -- with the pairs (cos, sin) the other phases left in the store's slots: no sincos, no cross
// products, no staging of q and qd.  It runs behind the outward sweep, when the workgroup's LDS is free but for the trunk's slots:
// [qdd rows | tau_in rows, overwritten in place by tau | limb wrenches].  The limbs are walked by the waves that hold their pairs (the
// owners of the first phase), each wave walks da down the trunk to its limbs for itself, wave 0 folds the trunk (MODE 1: limb roots from the
// exchange area).  Against a launch of its own for the inverse dynamics (q, qd, qdd staged again, every sincos again, its own barriers
// and tail: 95 us at 262 144 configurations beside the forward dynamics' 166) this phase adds 8 us to a group of 17 -- the limbs' walk
// 3.7-4.6 us, wave 0's fold of the trunk 2.2, copy-out 0.4 (profiles/r05_zvf_phase_stamps_pair.txt); without the scheduling fences between
// a body's phases the same (profiles/r05_ab_delta_fences.txt).
template <class TP, int J, typename T, class CX>
MH_DEV JX<T> zvf_delta_joint(const CX &cx)
{
   JX<T> jx;
   jx.c = T(1), jx.s = T(0), jx.d = T(0);
   if constexpr (TP::type[J] == JT_REVOLUTE)
      jx.c = cx.st.template get<J, 7>(), jx.s = cx.st.template get<J, 8>();
   return jx;
}
// da of trunk body J, walked down from the root
template <class TP, int J, typename T, class CX>
MH_DEV SV<T> zvf_delta_trunk_a(const CX &cx)
{
   constexpr int TYPE = TP::type[J], P = TP::parent[J];
   const SV<T> aJ = spec_vec<TYPE, Tree<TP>::dof_ofs(J), 1, CX, T>(cx, true);
   if constexpr (P < 0)
      return aJ;
   else
   {
      const SV<T> ap = zvf_delta_trunk_a<TP, P, T, CX>(cx);
      const CRef<T, false> c{cx.C + J * MC_STRIDE};
      return motion_down(TYPE, zvf_delta_joint<TP, J, T, CX>(cx), load_xb_j<TP, J, T>(c), ap) + aJ;
   }
}
template <class TP, int J, int K, class CX, typename T>
MH_DEV void zvf_delta_tau1(const CX &cx, T v)
{ // the row holds tau_in: tau = tau_in - (tau_in - h) + S^T df
   constexpr int DO = Tree<TP>::dof_ofs(J) + K;
   cx.lo[cx.di(DO)] = cx.lo[cx.di(DO)] - zvf_tau_get1<TP, J, K, CX, T>(cx) + v;
}
template <class TP, int J, typename T, class CX, int MODE = 0>
struct ZvfDelta
{
   template <int K>
   static MH_DEV void children(const CX &cx, const SV<T> &a, SV<T> &f)
   {
      if constexpr (K < Tree<TP>::n_children(J))
      {
         constexpr int C = Tree<TP>::child(J, K);
         if constexpr (MODE == 1 && !Split<TP>::is_trunk(C))
            f = f + x_get6<Split<TP>::limb_index(C), 6, 0, CX, T>(cx);
         else
            f = f + ZvfDelta<TP, C, T, CX, MODE>::run(cx, a);
         children<K + 1>(cx, a, f);
      }
   }
   static MH_DEV SV<T> run(const CX &cx, const SV<T> &ap)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J];
      constexpr bool HAS_PARENT = TP::parent[J] >= 0;
      const T *cp = cx.C + J * MC_STRIDE;
      asm volatile("" : "+s"(cp));
      const CRef<T, false> c{cp};
      const SV<T> aJ = spec_vec<TYPE, Tree<TP>::dof_ofs(J), 1, CX, T>(cx, true);
      SV<T> a = aJ;
      if constexpr (HAS_PARENT)
         a = motion_down(TYPE, zvf_delta_joint<TP, J, T, CX>(cx), load_xb_j<TP, J, T>(c), ap) + aJ;
      SV<T> f = mul(load_inertia<T>(c), a);
      MH_BODY_FENCE();
      children<0>(cx, a, f);
      MH_BODY_FENCE();
      if constexpr (TYPE == JT_REVOLUTE)
         zvf_delta_tau1<TP, J, 0, CX, T>(cx, f.a.z);
      else if constexpr (TYPE == JT_PRISMATIC)
         zvf_delta_tau1<TP, J, 0, CX, T>(cx, f.l.z);
      else if constexpr (TYPE == JT_SIXDOF)
      {
         zvf_delta_tau1<TP, J, 0, CX, T>(cx, f.a.x), zvf_delta_tau1<TP, J, 1, CX, T>(cx, f.a.y), zvf_delta_tau1<TP, J, 2, CX, T>(cx, f.a.z);
         zvf_delta_tau1<TP, J, 3, CX, T>(cx, f.l.x), zvf_delta_tau1<TP, J, 4, CX, T>(cx, f.l.y), zvf_delta_tau1<TP, J, 5, CX, T>(cx, f.l.z);
      }
      const V3<T> Z{T(0), T(0), T(0)};
      SV<T> up{Z, Z};
      if constexpr (HAS_PARENT)
      { // (the pose is read again rather than kept across the subtree: see RneaSub)
         const T *c2p = cx.C + J * MC_STRIDE;
         asm volatile("" : "+s"(c2p));
         up = force_up(TYPE, zvf_delta_joint<TP, J, T, CX>(cx), load_xb_j<TP, J, T>(CRef<T, false>{c2p}), f);
      }
      MH_BODY_FENCE();
      return up;
   }
};
// the limbs of wave W -- under the owners of the kernel's first phase: the mailed limb's pair and tau - h are in LDS, so the wave that
// shares its trunk walk takes it here too --: da walked down the trunk to each, the limb's wrench into the exchange area
template <class TP, int W, int K, typename T, class CX>
MH_DEV void zvf_delta_limbs_of(const CX &cx)
{
   using S = Split<TP>;
   if constexpr (K < S::n_limbs())
   {
      if constexpr (S::template owner_sel<(ZvfPlan<TP>::use_mail() ? 2 : 1)>(K) == W)
      {
         constexpr int R = S::limb_root(K), P = TP::parent[R];
         const V3<T> Z{T(0), T(0), T(0)};
         SV<T> ap{Z, Z};
         if constexpr (P >= 0)
            ap = zvf_delta_trunk_a<TP, P, T, CX>(cx);
         x_put6<K, 6, 0, CX, T>(cx, ZvfDelta<TP, R, T, CX, 0>::run(cx, ap));
      }
      zvf_delta_limbs_of<TP, W, K + 1, T, CX>(cx);
   }
}
template <class TP, int W, typename T, class CX>
MH_DEV void zvf_delta_limbs(const CX &cx)
{
   if constexpr (W < 4)
   {
      if (cx.wave == W)
         zvf_delta_limbs_of<TP, W, 0, T, CX>(cx);
      else
         zvf_delta_limbs<TP, W + 1, T, CX>(cx);
   }
}
template <class TP, typename T, class CX, int K = 0>
MH_DEV void zvf_delta_roots(const CX &cx)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      const V3<T> Z{T(0), T(0), T(0)};
      (void)ZvfDelta<TP, Tree<TP>::child(-1, K), T, CX, 1>::run(cx, SV<T>{Z, Z});
      zvf_delta_roots<TP, T, CX, K + 1>(cx);
   }
}
template <typename T, int NQ, int NV>
struct ZvfRows
{ // the three input matrices' rows of the NEXT group (requested during the fold of the current one)
   RowRegs<T, NQ, 256> q;
   RowRegs<T, NV, 256> qd, x;
   MH_DEV void request(const Args<T> &A, long k)
   {
      const long cfg0 = k * 64;
      const int rows = (int)(A.B - cfg0 < 64 ? A.B - cfg0 : 64);
      q.issue(A.q + cfg0 * A.m.nq, rows), qd.issue(A.qd + cfg0 * A.m.nv, rows), x.issue(A.in3 + cfg0 * A.m.nv, rows);
   }
};
// one group of 64 configurations; `next`: the group whose rows are requested on the way (the same group again on the last turn)
template <class TP, typename T, bool IDENT, bool STEP = false, bool PAIR = false>
MH_DEV void zvf_group(const Args<T> &A, long k, long next, lds_ptr<T> lds, ZvfRows<T, ZvfPlan<TP>::NQ, ZvfPlan<TP>::NV> &rows_ahead)
{
   using S = Split<TP>;
   using CX = Ctx<T, true, IDENT, ZvfStore<TP>, false, 0, 3>;
   const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const int lane = threadIdx.x & 63;
   const int nq = A.m.nq, nv = A.m.nv;
   // inverse dynamics: rows | limb wrenches | parked trunk wrenches.  Forward dynamics: trunk slots (over the rows of q / qd) | exchange area.
   const lds_ptr<T> lq = lds, lqd = lq + 64 * nq, lx = lqd + 64 * nv, lxc1 = lx + 64 * nv, lpark = lxc1 + S::n_limbs() * 6 * 64;
   const lds_ptr<T> lst = lds, lxc2 = lst + ZvfStore<TP>::TRUNK_SLOTS * 64, lres = lxc2 + S::n_limbs() * 12 * 64;
   const long cfg0 = k * 64;
   const int rows = (int)(A.B - cfg0 < 64 ? A.B - cfg0 : 64);
   const bool active = lane < rows;
   ZV_STAMP(2, 0);
#ifndef MH_ZVF_AHEAD
#define MH_ZVF_AHEAD 3 // when the NEXT group's rows are requested: 3 = behind the outward sweep (nothing else alive: 240 registers, no scratch;
                       // in flight during the copy-out); 1 = before the fold and 2 = behind it (longer in flight, but 96-112 bytes of scratch per
                       // lane, and with scratch the step is 40 % slower); 0 = when the group's own turn starts (no registers, fully exposed)
#endif
   if constexpr (!MH_ZVF_AHEAD)
      rows_ahead.request(A, k);
   rows_ahead.q.commit(lq), rows_ahead.qd.commit(lqd), rows_ahead.x.commit(lx);
   __syncthreads();
   ZV_STAMP(2, 1);
   CX cx;
   fill_ctx<T>(cx, A, active ? cfg0 + lane : cfg0);
   cx.coriolis = 1, cx.accel = 0;
   cx.lq = lq + lane * nq, cx.lqd = lqd + lane * nv, cx.lx = lx + lane * nv, cx.lo = lres + lane * nv;
   cx.wave = wave;
   cx.xbase = lxc1 + lane;
   cx.park = lpark + lane;
   cx.st.lbase = lst + lane;
   cx.st.gbase = nullptr, cx.st.stride = 0, cx.st.lane = 0;
#ifdef MH_ZV_PROBE
   cx.own = (unsigned long long)k;
#endif
   // ---- inverse dynamics at zero acceleration, limbs by the forward dynamics' owners: pairs and tau - h of the limb joints to registers
   if (active)
      split_rnea_limbs<TP, 0, T, CX, (ZvfPlan<TP>::use_mail() ? 2 : 1)>(cx);
   ZV_STAMP(2, 2);
   __syncthreads(); // the limbs' wrenches are in the exchange area; nobody reads the rows of q and qd any more
   ZV_STAMP(2, 3);
   if (active && wave == 0)
      rnea_trunk_roots<TP, T, CX>(cx); // the trunk's pairs and tau - h: to the trunk's slots (over the rows of q)
   ZV_STAMP(2, 4);
   __syncthreads();
   // ---- articulated inertias
   cx.xbase = lxc2 + lane;
   asm volatile("" ::: "memory");
   if (active) // (lane 0 of every wave is active, so each wave reaches the barrier a staged trunk carries in here)
      zv_limbs_in<TP, 0, T, CX>(cx);
   ZV_STAMP(2, 5);
   __syncthreads(); // every limb's (and sub-trunk's) articulated inertia is in the exchange area
   ZV_STAMP(2, 6);
   if (active)
      zv_roots_in<TP, T, CX, (S::staged() ? 2 : 1)>(cx);
   ZV_STAMP(2, 7);
   __syncthreads(); // nobody reads the exchange area's inertias any more
   if constexpr (MH_ZVF_AHEAD == 1)
      rows_ahead.request(A, next); // in flight during the fold, the outward sweep and the copy-out (unconditionally: see spec_zvb_bias_kernel)
   ZV_STAMP(2, 8);
   // ---- bias fold and outward sweep
   asm volatile("" ::: "memory");
   struct Ahead : ZvNoHook
   {
      ZvfRows<T, ZvfPlan<TP>::NQ, ZvfPlan<TP>::NV> &rows;
      const Args<T> &A;
      long next;
      MH_DEV void before_out() const
      {
         if constexpr (MH_ZVF_AHEAD == 2)
            rows.request(A, next); // behind the fold: in flight during the outward sweep and the copy-out
      }
   };
   const Ahead ahead{{}, rows_ahead, A, next};
   if (active) // (lane 0 of every wave is active: each wave reaches the barriers the fold carries in here)
      zv_fold_out<TP, 0, T, CX>(cx, ahead); // (a ragged group is the last one: what it would request is never committed)
   ZV_STAMP(2, 10);
   if constexpr (STEP)
   { // fused simulation step (spec_zvf_kernel<.., STEP = true>): the rows of q and qd were given up to the trunk's slots long ago -- they are
     // requested again (nothing else is alive in the registers here), land over the dead trunk slots and fold records in front of the result
     // rows (ZvfPlan::step_usable), and the new state is integrated in place and streamed out beside the accelerations
     // (MultiBodySystemStateIntegrator.java:365-441, 503-575, 710-733)
      RowRegs<T, ZvfPlan<TP>::NQ, 256> rq;
      RowRegs<T, ZvfPlan<TP>::NV, 256> rd;
      rq.issue(A.q + cfg0 * nq, rows), rd.issue(A.qd + cfg0 * nv, rows);
      zv_lds_barrier(); // every wave is through with its outward sweep: the trunk's slots are dead
      rq.commit(lq), rd.commit(lqd);
      zv_lds_barrier();
      if (active)
         integrate_rows<TP, 0, T>(wave, lq + lane * nq, lqd + lane * nv, lres + lane * nv, A.dt, T(0.5) * A.dt * A.dt);
      zv_lds_barrier();
      wave_copy_out<T, 256>(A.q_next + cfg0 * nq, lq, rows * nq);
      wave_copy_out<T, 256>(A.qd_next + cfg0 * nv, lqd, rows * nv);
   }
   if constexpr (MH_ZVF_AHEAD == 3 && !PAIR)
      rows_ahead.request(A, next); // behind the outward sweep (nothing else is alive any more): in flight during the copy-out
   RowRegs<T, PAIR ? ZvfPlan<TP>::NV : 1, 256> ra, rt; // PAIR: the caller's accelerations and the efforts once more, in flight during the copy-out
   if constexpr (PAIR)
      ra.issue(A.in3b + cfg0 * nv, rows), rt.issue(A.in3 + cfg0 * nv, rows);
   zv_lds_barrier();
   wave_copy_out<T, 256>(A.out + cfg0 * nv, lres, rows * nv);
   ZV_STAMP(2, 11);
   if constexpr (PAIR)
   { // ---- tau = h + M(q) qdd of the caller's accelerations (A.in3b) into A.outb: see ZvfDelta.  The two matrices' rows land over the
     // fold's records (dead since the outward sweep began), in front of the result rows that are being copied out; the limbs' wrenches go
     // over the result rows, behind the barrier that ends the copy.
      const lds_ptr<T> lqdd = lxc2, ltau = lqdd + 64 * nv, lxd = ltau + 64 * nv;
      if constexpr (2 * ZvfPlan<TP>::NV > S::n_limbs() * 12)
         zv_lds_barrier(); // (the tau rows reach into the result rows: wait for the copy)
      ra.commit(lqdd), rt.commit(ltau);
      if constexpr (MH_ZVF_AHEAD == 3)
         rows_ahead.request(A, next); // (in flight during the whole phase)
      zv_lds_barrier();
      cx.lx = lqdd + lane * nv, cx.lo = ltau + lane * nv;
      cx.xbase = lxd + lane;
      asm volatile("" ::: "memory");
      if (active)
         zvf_delta_limbs<TP, 0, T, CX>(cx);
      ZV_STAMP(2, 13);
      zv_lds_barrier(); // the limbs' wrenches are in the exchange area
      if (active && wave == 0)
         zvf_delta_roots<TP, T, CX>(cx);
      ZV_STAMP(2, 14);
      zv_lds_barrier();
      wave_copy_out<T, 256>(A.outb + cfg0 * nv, ltau, rows * nv);
   }
   zv_lds_barrier(); // the rows are committed over the result rows by the next turn
   ZV_STAMP(2, 12);
}
template <class TP, typename T, bool IDENT, bool STEP = false, bool PAIR = false>
__global__ void __launch_bounds__(256, 2) spec_zvf_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   ZvfRows<T, ZvfPlan<TP>::NQ, ZvfPlan<TP>::NV> rows_ahead;
   const long ngroups = (A.B + 63) / 64;
   if constexpr (MH_ZVF_AHEAD)
      rows_ahead.request(A, blockIdx.x);
   for (long k = blockIdx.x; k < ngroups; k += gridDim.x)
      zvf_group<TP, T, IDENT, STEP, PAIR>(A, k, k + gridDim.x < ngroups ? k + gridDim.x : k, (lds_ptr<T>)lds_raw, rows_ahead);
}
} // namespace mh
