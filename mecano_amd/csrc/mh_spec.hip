// mh_spec.hip -- one topology-specialised code object.  Compiled once per kinematic-tree shape by mecano_amd/build.py:
//
//   hipcc --offload-arch=gfx950 -DMH_TOPO_N=25 "-DMH_TOPO_PARENTS=-1,0,1,..." "-DMH_TOPO_TYPES=2,0,0,..." \
//         -shared -o libmecano_hip_topo_<key>.so mh_spec.hip
//
// libmecano_hip.so dlopen()s it from its own directory when a model with this topology is created
// (mh_model_create) and routes mh_rnea_f64 / mh_aba_f64 to it; every other model and dtype runs on the generic kernels.
#include "mh_spec_kernels.h"
#include "mh_zv_kernels.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

// MH_SPEC_PART = 1 | 2: this file compiled as one of two translation units that are linked into one code object (mecano_amd/build.py: the
// device code of a unit is generated kernel by kernel on one core, and the humanoid's took five minutes); undefined / 0: everything.
#ifndef MH_SPEC_PART
#define MH_SPEC_PART 0
#endif
#ifndef MH_TOPO_N
#error "MH_TOPO_N / MH_TOPO_PARENTS / MH_TOPO_TYPES must be defined"
#endif

namespace
{
struct TP
{
   static constexpr int N = MH_TOPO_N;
   static constexpr int parent[N] = {MH_TOPO_PARENTS};
   static constexpr int type[N] = {MH_TOPO_TYPES};
};
constexpr int kParents[TP::N] = {MH_TOPO_PARENTS};
constexpr int kTypes[TP::N] = {MH_TOPO_TYPES};
using TR = mh::Tree<TP>;
constexpr bool kinds_supported()
{
   for (int j = 0; j < TP::N; j++)
      if (TP::type[j] < mh::JT_REVOLUTE || TP::type[j] > mh::JT_FIXED)
         return false;
   return true;
}
static_assert(kinds_supported(), "specialised code objects cover revolute, prismatic, 6-DoF and fixed joints; planar / spherical joints run on the generic kernels");

enum : int
{
   F_IO_LDS = 1, // state rows staged in LDS (AoS layout only)
   F_IDENT = 2,  // identity index maps
   F_ST_LDS = 4, // ABA hand-over store in LDS
   F_BODIES = 16, // tree-split RNEA / ABA that also write the per-body accelerations / twists (identity maps + LDS rows only)
   F_OCC3 = 32    // tree-split RNEA without LDS rows (SoA): the build with a three-waves-per-SIMD register budget (device-filling batches)
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: the "already raised to" cache is kept per device
// (models may live on different devices of one process, mh_set_device) and is atomic (two host threads with handles of the same topology).
struct LdsAttr
{
   std::atomic<size_t> bytes[16] = {};
};
hipError_t ensure_lds_attr(const void *kern, size_t lds, LdsAttr &cache)
{
   if (lds <= 64 * 1024)
      return hipSuccess;
   int dev = 0;
   if (hipGetDevice(&dev) != hipSuccess)
      dev = 0;
   std::atomic<size_t> &have = cache.bytes[dev & 15];
   if (lds <= have.load(std::memory_order_acquire))
      return hipSuccess;
   const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
   if (e == hipSuccess)
      have.store(lds, std::memory_order_release);
   return e;
}

long lds_bytes(int algo, int flags, int nq, int nv)
{
   long b = 0;
   if (flags & F_IO_LDS)
      b += (long)(nq + 2 * nv) * 64 * sizeof(double);
   if (algo == 1 && (flags & F_ST_LDS))
      b += (long)TR::aba_slot(TP::N) * 64 * sizeof(double);
   return b;
}

template <int ALGO, bool IO, bool ID, bool ST>
hipError_t go(const mh::Args<double> &A, int grid, size_t lds, hipStream_t stream)
{
   auto kern = &mh::spec_kernel<TP, double, ALGO, IO, ID, ST>;
   static LdsAttr attr; // per instantiation
   if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(kern), lds, attr); e != hipSuccess)
      return e;
   hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, stream, A);
   return hipGetLastError();
}
// Whole-tree (one wave per 64 configurations) ABA is built only for trees WITHOUT a tree-split form (chains): where the tree-split
// kernel exists the dispatcher prefers it at every batch size, and the whole-tree ABA of a 25-body tree needs the full 512-register
// budget plus 50-130 spills -- the one place where hipcc 7.2 handed back wrong results for some memory plans (open issues).
// Long chains are out as well: a whole-tree walk keeps the down-sweep state of every body on the path alive across the turn-around
// (18+ values per body), so a 30-body chain needs the full 512 registers plus ~600 spilled VGPRs and ~1000 spilled SGPRs (measured:
// 10 minutes of compile time, 1.9 KB of scratch per lane) -- the regime of the wrong-result build.  Chains of more than
// kWholeTreeMaxBodies bodies get no whole-tree kernels at all (RNEA included: 512 registers + 70..400 scalar spills at 30 bodies); they
// run on the run-time-topology kernels, whose per-lane stack is indexed by tree depth.  Whatever IS built is checked against those
// kernels when the object is loaded (mh_model_create).
constexpr int kWholeTreeMaxBodies = 12;
#ifdef MH_FORCE_WHOLE_TREE_ABA // diagnosis builds only (tools/diag_whole_tree.py): the regime this gate exists to keep out
constexpr bool kWholeTreeAba = true;
#else
constexpr bool kWholeTreeAba = !mh::Split<TP>::usable() && TP::N <= kWholeTreeMaxBodies;
#endif
constexpr bool kWholeTreeRnea = mh::Split<TP>::usable() || TP::N <= kWholeTreeMaxBodies;
template <int ALGO, bool IO, bool ID>
hipError_t go_st(int flags, const mh::Args<double> &A, int grid, size_t lds, hipStream_t s)
{
   if constexpr (ALGO == 1)
   {
      if constexpr (!kWholeTreeAba)
         return hipErrorNotSupported;
      else
      {
         if (flags & F_ST_LDS)
            return go<ALGO, IO, ID, true>(A, grid, lds, s);
         return go<ALGO, IO, ID, false>(A, grid, lds, s);
      }
   }
   else
   {
      if constexpr (!kWholeTreeRnea)
         return hipErrorNotSupported;
      else
         return go<ALGO, IO, ID, false>(A, grid, lds, s);
   }
}
template <int ALGO>
hipError_t go_flags(int flags, const mh::Args<double> &A, int grid, size_t lds, hipStream_t s)
{
   const bool io = flags & F_IO_LDS, id = flags & F_IDENT;
   if (io && id)
      return go_st<ALGO, true, true>(flags, A, grid, lds, s);
   if (io)
      return go_st<ALGO, true, false>(flags, A, grid, lds, s);
   if (id)
      return go_st<ALGO, false, true>(flags, A, grid, lds, s);
   return go_st<ALGO, false, false>(flags, A, grid, lds, s);
}
} // namespace

template <bool ID>
hipError_t go_fused(const mh::Args<double> &A, int waves, hipStream_t stream)
{
   if constexpr (!kWholeTreeAba)
      return hipErrorNotSupported;
   else
   {
   auto kern = &mh::spec_fused_kernel<TP, double, ID>;
   const size_t lds = (size_t)std::max(lds_bytes(0, F_IO_LDS, A.m.nq, A.m.nv), lds_bytes(1, F_ST_LDS, A.m.nq, A.m.nv));
   static LdsAttr attr;
   if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(kern), lds, attr); e != hipSuccess)
      return e;
   hipLaunchKernelGGL(kern, dim3(2 * waves), dim3(64), lds, stream, A);
   return hipGetLastError();
   }
}

using SPL = mh::Split<TP>;
static long split_lds_bytes(int algo, int flags, int nq, int nv)
{
   if (algo == 2)
      return std::max(split_lds_bytes(0, flags, nq, nv), split_lds_bytes(1, flags, nq, nv));
   long b = (long)(SPL::n_limbs() * (algo == 0 ? 6 : 27) + (algo == 1 ? SPL::TRUNK_SLOTS : SPL::RNEA_TRUNK_SLOTS)) * 64 * sizeof(double);
   if (flags & F_IO_LDS)
      b += (long)(nq + (algo == 1 ? 3 : 2) * nv) * 64 * sizeof(double); // forward dynamics: result rows of their own (split_group)
   return b;
}
template <class K>
hipError_t launch_lds(K kern, const mh::Args<double> &A, int grid, size_t lds, LdsAttr &attr, hipStream_t stream)
{
   if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(kern), lds, attr); e != hipSuccess)
      return e;
   hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, A);
   return hipGetLastError();
}
template <bool ID, bool IO>
hipError_t go_fused_split(const mh::Args<double> &A, int groups, hipStream_t stream)
{
   if constexpr (SPL::usable())
   {
      static LdsAttr attr_bytes;
      return launch_lds(&mh::spec_fused_split_kernel<TP, double, ID, IO>, A, 2 * groups,
                        (size_t)split_lds_bytes(2, IO ? F_IO_LDS : 0, A.m.nq, A.m.nv), attr_bytes, stream);
   }
   else
      return hipErrorNotSupported;
}
template <int ALGO, bool ID, bool IO>
hipError_t go_split(const mh::Args<double> &A, int groups, hipStream_t stream, bool occ3 = false)
{
   if constexpr (SPL::usable())
   {
      static LdsAttr attr_bytes, attr_occ3; // one cache per kernel
      if constexpr (ALGO == 0 && ID && !IO)
      {
         if (occ3)
            return launch_lds(&mh::spec_split_kernel_occ3<TP, double, ALGO, ID, IO>, A, groups,
                              (size_t)split_lds_bytes(ALGO, 0, A.m.nq, A.m.nv), attr_occ3, stream);
      }
      return launch_lds(&mh::spec_split_kernel<TP, double, ALGO, ID, IO>, A, groups,
                        (size_t)split_lds_bytes(ALGO, IO ? F_IO_LDS : 0, A.m.nq, A.m.nv), attr_bytes, stream);
   }
   else
      return hipErrorNotSupported;
}
template <bool ID, bool IO>
hipError_t go_split_algo(int algo, const mh::Args<double> &A, int groups, hipStream_t s, bool occ3)
{
   if (algo == 2)
      return go_fused_split<ID, IO>(A, groups, s);
   if (algo == 0)
      return go_split<0, ID, IO>(A, groups, s, occ3);
   if (algo == 1)
      return go_split<1, ID, IO>(A, groups, s);
   return hipErrorNotSupported;
}

extern "C" {
// tree-split kernels (4 waves per 64 configurations): available when the tree has a trunk with at least two limbs
#if MH_SPEC_PART != 2 // ---- part 1 of a two-part build: everything but the bias-split / fused / rows-ahead launchers
int mh_spec_split_usable(void) { return SPL::usable() ? 1 : 0; }
long mh_spec_split_lds_bytes(int algo, int flags, int nq, int nv) { return split_lds_bytes(algo, flags, nq, nv); }
// algo: 0 = RNEA, 1 = ABA, 2 = fused RNEA+ABA (2 * groups workgroups); groups = ceil(B / 64) or fewer (grid-stride).
// flags: F_IDENT, F_IO_LDS (AoS rows staged in LDS; needs dense index maps)
int mh_spec_launch_split(int algo, int flags, const void *args, int groups, void *stream)
{
   const mh::Args<double> &A = *(const mh::Args<double> *)args;
   const bool id = flags & F_IDENT, io = flags & F_IO_LDS;
   hipStream_t s = (hipStream_t)stream;
   const bool occ3 = (flags & F_OCC3) != 0;
   if (flags & F_BODIES)
   {
#ifdef MH_SPEC_MINIMAL
      return (int)hipErrorNotSupported;
#else
      if constexpr (SPL::usable())
      {
         if (id && io && (algo == 0 || algo == 1))
         {
            static LdsAttr attr0, attr1;
            const size_t lds = (size_t)split_lds_bytes(algo, F_IO_LDS, A.m.nq, A.m.nv);
            if (algo == 0)
               return (int)launch_lds(&mh::spec_split_kernel<TP, double, 0, true, true, true>, A, groups, lds, attr0, s);
            return (int)launch_lds(&mh::spec_split_kernel<TP, double, 1, true, true, true>, A, groups, lds, attr1, s);
         }
      }
      return (int)hipErrorNotSupported;
#endif
   }
   if (id && io)
      return (int)go_split_algo<true, true>(algo, A, groups, s, occ3);
#ifdef MH_SPEC_MINIMAL // experiment builds (tools/): only the variant bench.py runs, seconds instead of minutes to compile
   return (int)hipErrorNotSupported;
#else
   if (id)
      return (int)go_split_algo<true, false>(algo, A, groups, s, occ3);
   if (io)
      return (int)go_split_algo<false, true>(algo, A, groups, s, occ3);
   return (int)go_split_algo<false, false>(algo, A, groups, s, occ3);
#endif
}
#endif
#if MH_SPEC_PART != 1 // ---- part 2: the kernels of mh_zv_kernels.h
// ---- bias-split forward dynamics (mh_zv_kernels.h): AoS matrices with dense index maps, rows staged in LDS
long mh_spec_zv_lds_bytes(int nq, int nv)
{
   if constexpr (!SPL::usable())
      return 0;
   const long inertia = (long)(SPL::n_limbs() * mh::ZV_XW + SPL::ZV_TRUNK_SLOTS + nq + 2 * nv) * 64 * (long)sizeof(double);
   // (the bias job: the tree-split inverse dynamics' map + the efforts' rows of their own, mh_zv_kernels.h: MH_ZV_TAU_LATE)
   return std::max(inertia, split_lds_bytes(0, F_IO_LDS, nq, nv) + (long)mh::zv_bias_extra_rows<TP>() * 64 * (long)sizeof(double));
}
int mh_spec_zv_usable(void) { return SPL::usable() ? 1 : 0; }
// 1: launches with identity index maps (the two-stage hand-off) expect `taup` to hold the sentinel wherever no column has been published
// (mh_zv_kernels.h, "Stage one without a flag"); the library keeps a matrix of its own for them and refills it after a give-up
int mh_spec_zv_self_signal(void) { return MH_ZV_TWO_STAGE && MH_ZV_SELF_SIGNAL ? 1 : 0; }
#ifdef MH_ZV_PROBE
int mh_spec_zv_probe_read(void *dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mh::zv_probe), bytes); }
int mh_spec_zv_probe_body_read(void *dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mh::zv_probe_body), bytes); }
#endif
// jobs = 2: qdd = ABA(q, qd, tau) with args->in3b = tau, args->outb = qdd.  jobs = 3: additionally args->out = RNEA(q, qd, args->in3).
// taup: scratch [B][nv]; sync_flags: ceil(B / 64) * ZV_SYNC_STRIDE ints that never held `epoch` before; same_l2: rows may stay in a shared L2; error: one int
// (host-mapped), set when a wait ran into wait_ticks (100 MHz ticks): that group's accelerations are then NaN.
int mh_spec_launch_zv(int flags, const void *args, void *taup, int *sync_flags, int *error, int epoch, int jobs, int same_l2, unsigned wait_ticks, void *stream)
{
   if constexpr (SPL::usable())
   {
      const mh::Args<double> &A = *(const mh::Args<double> *)args;
      if (!(flags & F_IO_LDS) || (jobs != 2 && jobs != 3))
         return (int)hipErrorNotSupported;
      const size_t lds = (size_t)mh_spec_zv_lds_bytes(A.m.nq, A.m.nv);
      if (lds > 160 * 1024)
         return (int)hipErrorNotSupported;
      const long groups = (A.B + 63) / 64, padded = (groups + 7) / 8 * 8;
      const mh::ZvSync sy{sync_flags, error, epoch, jobs, same_l2 ? 1 : 0, wait_ticks};
      hipStream_t s = (hipStream_t)stream;
      if (A.q_next && (!(flags & F_IDENT) || jobs != 2 || !MH_ZV_TWO_STAGE))
         return (int)hipErrorNotSupported; // a simulation step rides in the two-stage form only (identity index maps, forward dynamics alone)
      if ((flags & F_IDENT) && A.q_next)
      {
         static LdsAttr attr_step;
         auto kern = &mh::spec_zv_kernel<TP, double, true, true>;
         if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(kern), lds, attr_step); e != hipSuccess)
            return (int)e;
         hipLaunchKernelGGL(kern, dim3((unsigned)(padded * jobs)), dim3(256), lds, s, A.q, A.qd, A.in3b, A.B, jobs, A, (double *)taup, sy);
         return (int)hipGetLastError();
      }
      if (flags & F_IDENT)
      {
         static LdsAttr attr;
         auto kern = &mh::spec_zv_kernel<TP, double, true>;
         if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(kern), lds, attr); e != hipSuccess)
            return (int)e;
         hipLaunchKernelGGL(kern, dim3((unsigned)(padded * jobs)), dim3(256), lds, s, A.q, A.qd, A.in3b, A.B, jobs, A, (double *)taup, sy);
         return (int)hipGetLastError();
      }
#ifdef MH_SPEC_MINIMAL
      return (int)hipErrorNotSupported;
#else
      static LdsAttr attr2;
      auto kern = &mh::spec_zv_kernel<TP, double, false>;
      if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(kern), lds, attr2); e != hipSuccess)
         return (int)e;
      hipLaunchKernelGGL(kern, dim3((unsigned)(padded * jobs)), dim3(256), lds, s, A.q, A.qd, A.in3b, A.B, jobs, A, (double *)taup, sy);
      return (int)hipGetLastError();
#endif
   }
   else
      return (int)hipErrorNotSupported;
}
// ---- forward dynamics of device-filling batches as two launches (mh_zv_kernels.h, spec_zvb_*): AoS matrices, dense index maps.
// cs: scratch [2 * revolute joints][cs_stride] with cs_stride >= B rounded up to whole groups of 64; taup: scratch [B][nv].
int mh_spec_zvb_usable(void)
{
   if constexpr (SPL::usable())
      return mh::ZvbPlan<TP>::lds_slots() * 64 * (long)sizeof(double) * 2 <= 160 * 1024 ? 1 : 0; // two workgroups per CU or not at all
   else
      return 0;
}
int mh_spec_zvb_cs_rows(void) { return 2 * TR::n_revolute(); }
long mh_spec_zvb_lds_bytes(int which, int nq, int nv)
{ // which: 0 = the bias launch, 1 = the inertia launch
   if constexpr (SPL::usable())
      return (long)(which == 0 ? mh::ZvbPlan<TP>::bias_lds_slots(nq, nv) : mh::ZvbPlan<TP>::lds_slots()) * 64 * (long)sizeof(double);
   else
      return 0;
}
// which: 1 = bias launch only, 2 = inertia launch only, 3 = both (measurements time them apart).  args->in3 = tau, args->out = qdd.
int mh_spec_launch_zvb(int flags, const void *args, void *taup, void *cs, long cs_stride, int groups, int which, void *stream)
{
   if constexpr (SPL::usable())
   {
      const mh::Args<double> &A = *(const mh::Args<double> *)args;
      if (!(flags & F_IO_LDS) || !mh_spec_zvb_usable() || cs_stride < (A.B + 63) / 64 * 64 || groups < 1)
         return (int)hipErrorNotSupported;
      const size_t lds0 = (size_t)mh_spec_zvb_lds_bytes(0, A.m.nq, A.m.nv), lds1 = (size_t)mh_spec_zvb_lds_bytes(1, A.m.nq, A.m.nv);
      if (lds0 > 160 * 1024)
         return (int)hipErrorNotSupported;
      hipStream_t s = (hipStream_t)stream;
      auto run = [&](auto ident) -> hipError_t {
         constexpr bool ID = decltype(ident)::value;
         static LdsAttr attr0, attr1;
         auto k0 = &mh::spec_zvb_bias_kernel<TP, double, ID>;
         auto k1 = &mh::spec_zvb_kernel<TP, double, ID>;
         if (which & 1)
         {
            if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(k0), lds0, attr0); e != hipSuccess)
               return e;
            hipLaunchKernelGGL(k0, dim3((unsigned)groups), dim3(256), lds0, s, A, (double *)taup, (double *)cs, cs_stride);
            if (const hipError_t e = hipGetLastError(); e != hipSuccess)
               return e;
         }
         if (which & 2)
         {
            if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(k1), lds1, attr1); e != hipSuccess)
               return e;
            hipLaunchKernelGGL(k1, dim3((unsigned)groups), dim3(256), lds1, s, A, (const double *)taup, (const double *)cs, cs_stride);
            if (const hipError_t e = hipGetLastError(); e != hipSuccess)
               return e;
         }
         return hipSuccess;
      };
      if (flags & F_IDENT)
         return (int)run(std::true_type{});
#ifdef MH_SPEC_MINIMAL
      return (int)hipErrorNotSupported;
#else
      return (int)run(std::false_type{});
#endif
   }
   else
      return (int)hipErrorNotSupported;
}
// ---- inverse dynamics of device-filling batches in the bias launch's persistent loop (spec_zvb_bias_kernel<.., BIAS = false>): AoS matrices,
// dense index maps, no per-body outputs.  args->in3 = qdd, args->out = tau; groups = workgroups of the launch.
int mh_spec_launch_rnea_ahead(int flags, const void *args, int groups, void *stream)
{
   if constexpr (SPL::usable())
   {
      const mh::Args<double> &A = *(const mh::Args<double> *)args;
      if (!(flags & F_IO_LDS) || !mh_spec_zvb_usable() || groups < 1 || A.body_acc || A.body_twist || A.m.nq != TR::total_cfgs() || A.m.nv != TR::total_dofs())
         return (int)hipErrorNotSupported;
      const size_t lds = (size_t)mh_spec_zvb_lds_bytes(0, A.m.nq, A.m.nv);
      if (lds * 2 > 160 * 1024)
         return (int)hipErrorNotSupported;
      hipStream_t s = (hipStream_t)stream;
      auto run = [&](auto ident) -> hipError_t {
         constexpr bool ID = decltype(ident)::value;
         static LdsAttr attr;
         auto k = &mh::spec_zvb_bias_kernel<TP, double, ID, false>;
         if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(k), lds, attr); e != hipSuccess)
            return e;
         hipLaunchKernelGGL(k, dim3((unsigned)groups), dim3(256), lds, s, A, A.out, (double *)nullptr, 0L);
         return hipGetLastError();
      };
      if (flags & F_IDENT)
         return (int)run(std::true_type{});
#ifdef MH_SPEC_MINIMAL
      return (int)hipErrorNotSupported;
#else
      return (int)run(std::false_type{});
#endif
   }
   else
      return (int)hipErrorNotSupported;
}
// ---- the same as ONE launch (spec_zvf_kernel: bias and inertia job fused in a workgroup); trees whose joints below the root are revolute / fixed
int mh_spec_zvf_usable(void)
{
   if constexpr (SPL::usable())
      return mh::ZvfPlan<TP>::usable() ? 1 : 0;
   else
      return 0;
}
long mh_spec_zvf_lds_bytes(void)
{
   if constexpr (SPL::usable())
      return (long)mh::ZvfPlan<TP>::lds_slots() * 64 * (long)sizeof(double);
   else
      return 0;
}
// args->in3 = tau, args->out = qdd; groups = workgroups of the launch (each loops over the batch's groups of 64 configurations)
int mh_spec_launch_zvf(int flags, const void *args, int groups, void *stream)
{
   if constexpr (SPL::usable())
   {
      const mh::Args<double> &A = *(const mh::Args<double> *)args;
      if (!(flags & F_IO_LDS) || !(flags & F_IDENT) || !mh_spec_zvf_usable() || groups < 1 || A.m.nq != TR::total_cfgs() || A.m.nv != TR::total_dofs())
         return (int)hipErrorNotSupported;
      if (A.q_next)
      { // a simulation step: the kernel that integrates the rows it holds and writes the new state too
         if constexpr (mh::ZvfPlan<TP>::step_usable())
         {
            static LdsAttr attr_step;
            return (int)launch_lds(&mh::spec_zvf_kernel<TP, double, true, true>, A, groups, (size_t)mh_spec_zvf_lds_bytes(), attr_step, (hipStream_t)stream);
         }
         else
            return (int)hipErrorNotSupported;
      }
      if (A.in3b && A.outb)
      { // the pair call: args->in3b = qdd of the caller, args->outb = tau (mh_zv_kernels.h: ZvfDelta)
         if constexpr (mh::ZvfPlan<TP>::pair_usable())
         {
            static LdsAttr attr_pair;
            return (int)launch_lds(&mh::spec_zvf_kernel<TP, double, true, false, true>, A, groups, (size_t)mh_spec_zvf_lds_bytes(), attr_pair, (hipStream_t)stream);
         }
         else
            return (int)hipErrorNotSupported;
      }
      static LdsAttr attr;
      return (int)launch_lds(&mh::spec_zvf_kernel<TP, double, true>, A, groups, (size_t)mh_spec_zvf_lds_bytes(), attr, (hipStream_t)stream);
   }
   else
      return (int)hipErrorNotSupported;
}
// 1: mh_spec_launch_zvf serves the pair call too (args->in3b, args->outb)
int mh_spec_zvf_pair_usable(void)
{
   if constexpr (SPL::usable())
      return mh::ZvfPlan<TP>::pair_usable() ? 1 : 0;
   else
      return 0;
}
#endif
#if MH_SPEC_PART != 2
// the tree-split plan of this topology, for tests and documentation: out[0] = usable, [1] = staged trunk, [2] = limbs, [3] = sub-trunks,
// [4] = root trunk body, then per limb (root body, bodies, ABA owner wave, RNEA / CRBA owner wave, late) and per wave the body after
// whose children its cut barrier sits (-1: explicit barrier).  Returns the number of ints written (<= cap).
int mh_spec_split_plan(int *out, int cap)
{
   int n = 0;
   auto put = [&](int v) {
      if (n < cap)
         out[n] = v;
      n++;
   };
   put(SPL::usable() ? 1 : 0), put(SPL::staged() ? 1 : 0), put(SPL::n_limbs()), put(SPL::n_sub()), put(SPL::root());
   for (int k = 0; k < SPL::n_limbs(); k++)
      put(SPL::limb_root(k)), put(SPL::P.size_of[k]), put(SPL::owner(k)), put(SPL::owner_plain(k)), put(SPL::is_late(k) ? 1 : 0);
   for (int w = 0; w < 4; w++)
      put(SPL::P.cut_body[w]);
   return n < cap ? n : cap;
}
// everything this object shares with libmecano_hip.so beyond its own entry points (argument structs, record strides, frame convention)
unsigned long long mh_spec_abi(void) { return mh::spec_abi_stamp(); }
// build provenance: the hash of the kernel sources and code-generation flags this object was compiled from (mecano_amd/build.py and
// mh_build_code_object compute it and pass it in; libmecano_hip.so refuses an object whose hash is not the one it was built beside),
// and, as a string any tool can find in the file without loading it, the whole build id: sources, tree, extra flags
#define MH_SPEC_STR2_(...) #__VA_ARGS__
#define MH_SPEC_STR_(...) MH_SPEC_STR2_(__VA_ARGS__)
#ifndef MH_SPEC_SOURCES_HASH
#define MH_SPEC_SOURCES_HASH unhashed
#endif
#ifndef MH_BUILD_EXTRA
#define MH_BUILD_EXTRA none
#endif
__attribute__((used)) const char mh_spec_build_id_string[] = "MH_BUILD_ID=" MH_SPEC_STR_(MH_SPEC_SOURCES_HASH) ";N=" MH_SPEC_STR_(MH_TOPO_N) ";P=" MH_SPEC_STR_(
   MH_TOPO_PARENTS) ";T=" MH_SPEC_STR_(MH_TOPO_TYPES) ";X=" MH_SPEC_STR_(MH_BUILD_EXTRA) ";";
const char *mh_spec_sources_hash(void) { return MH_SPEC_STR_(MH_SPEC_SOURCES_HASH); }
const char *mh_spec_build_id(void) { return mh_spec_build_id_string; }
// 1: built with -DMH_SPEC_MINIMAL (mh_build_code_object's fast form, tools/isa.py): only the tree-split RNEA / ABA / pair kernels
int mh_spec_minimal(void)
{
#ifdef MH_SPEC_MINIMAL
   return 1;
#else
   return 0;
#endif
}
int mh_spec_n(void) { return TP::N; }
const int *mh_spec_parents(void) { return kParents; }
const int *mh_spec_types(void) { return kTypes; }
int mh_spec_nq(void) { return TR::cfg_ofs(TP::N); }
int mh_spec_nv(void) { return TR::dof_ofs(TP::N); }
int mh_spec_aba_slots(void) { return TR::aba_slot(TP::N); }
// which (algo, flag) combinations this code object was built with
int mh_spec_supports(int algo, int flags)
{
   // ABA with BOTH the state rows and the hand-over store in LDS is built but not offered: on gfx950 / ROCm 7.2 that
   // variant of the 25-body kernel (512 registers, ~100 spills) returned wrong velocity-dependent terms although each of
   // the two LDS uses is exact on its own (tests/test_gpu_parity.py::test_every_specialised_variant); see DESIGN.md.
   if (algo == 1 && (flags & F_IO_LDS) && (flags & F_ST_LDS))
      return 0;
   if (algo == 1 && !kWholeTreeAba)
      return 0; // trees with a tree-split form, long chains: whole-tree ABA is not built (see kWholeTreeAba)
   if (algo == 0 && !kWholeTreeRnea)
      return 0;
   return algo == 0 || algo == 1;
}
// dynamic LDS one workgroup (one wave) needs for (algo, flags) with the model's matrix sizes
long mh_spec_lds_bytes(int algo, int flags, int nq, int nv) { return lds_bytes(algo, flags, nq, nv); }
// fused RNEA + ABA (small batches): `waves` workgroups per job, 2 * waves in the grid; needs dense index maps (rows staged as blocks)
long mh_spec_fused_lds_bytes(int nq, int nv) { return std::max(lds_bytes(0, F_IO_LDS, nq, nv), lds_bytes(1, F_ST_LDS, nq, nv)); }
int mh_spec_launch_fused(int flags, const void *args, int waves, void *stream)
{
#ifdef MH_SPEC_MINIMAL
   return (int)hipErrorNotSupported;
#else
   const mh::Args<double> &A = *(const mh::Args<double> *)args;
   return (int)((flags & F_IDENT) ? go_fused<true>(A, waves, (hipStream_t)stream) : go_fused<false>(A, waves, (hipStream_t)stream));
#endif
}
// CRBA (fp64).  Returns in *needs_zero_fill whether the caller must zero H first (direct-store kernel) or not (packed kernel).
long mh_spec_crba_lds_bytes(void) { return (long)mh::HMap<TP>::T.n_slots * 64 * sizeof(double); }
int mh_spec_crba_packed(int flags) { return (flags & F_IDENT) && mh_spec_crba_lds_bytes() <= 160 * 1024 ? 1 : 0; }
int mh_spec_launch_crba(int flags, const void *args, int grid, void *stream)
{
#ifdef MH_SPEC_MINIMAL
   return (int)hipErrorNotSupported;
#else
   const mh::Args<double> &A = *(const mh::Args<double> *)args;
   if (mh_spec_crba_packed(flags))
   {
      auto kern = &mh::spec_crba_packed_kernel<TP, double>;
      const size_t lds = (size_t)mh_spec_crba_lds_bytes();
      static LdsAttr attr;
      if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(kern), lds, attr); e != hipSuccess)
         return (int)e;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, (hipStream_t)stream, A);
   }
   else if (flags & F_IDENT)
      hipLaunchKernelGGL((mh::spec_crba_kernel<TP, double, true>), dim3(grid), dim3(64), 0, (hipStream_t)stream, A);
   else
      hipLaunchKernelGGL((mh::spec_crba_kernel<TP, double, false>), dim3(grid), dim3(64), 0, (hipStream_t)stream, A);
   return (int)hipGetLastError();
#endif
}
// mass + Coriolis matrix (fp64): one wave per 64 configurations, direct stores into zero-filled H and C (A.out, A.outb; strides f_bs, f_es)
// parts waves per group of 64 configurations, each writing the columns of every parts-th body (small batches; bodies are tracked in a
// 64-bit mask)
int mh_spec_launch_coriolis_parts(int flags, const void *args, int grid, int parts, void *stream)
{
#ifdef MH_SPEC_MINIMAL
   return (int)hipErrorNotSupported;
#else
   const mh::Args<double> &A = *(const mh::Args<double> *)args;
   const dim3 g(grid, TP::N <= 64 ? std::max(1, std::min(parts, (int)TP::N)) : 1);
   // the fast variant: identity index maps, AoS matrices (entry stride 1, nv = the tree's DoFs): compile-time entry offsets
   if ((flags & F_IDENT) && A.f_es == 1 && A.m.nv == TR::total_dofs())
      hipLaunchKernelGGL((mh::spec_coriolis_kernel<TP, double, true, true>), g, dim3(64), 0, (hipStream_t)stream, A);
   else if (flags & F_IDENT)
      hipLaunchKernelGGL((mh::spec_coriolis_kernel<TP, double, true, false>), g, dim3(64), 0, (hipStream_t)stream, A);
   else
      hipLaunchKernelGGL((mh::spec_coriolis_kernel<TP, double, false, false>), g, dim3(64), 0, (hipStream_t)stream, A);
   return (int)hipGetLastError();
#endif
}
int mh_spec_launch_coriolis(int flags, const void *args, int grid, void *stream) { return mh_spec_launch_coriolis_parts(flags, args, grid, 1, stream); }
// centroidal momentum matrix (+ convective term when args->b is not NULL) (fp64): one wave per 64 configurations, A zero-filled by the caller
int mh_spec_launch_centroidal_parts(int flags, const void *args, int grid, int parts, void *stream)
{
#ifdef MH_SPEC_MINIMAL
   return (int)hipErrorNotSupported;
#else
   const mh::CentArgs<double> &A = *(const mh::CentArgs<double> *)args;
   const bool id = flags & F_IDENT, wb = A.b != nullptr;
   const dim3 g(grid, TP::N <= 64 ? std::max(1, std::min(parts, (int)TP::N)) : 1);
   if (id && wb)
      hipLaunchKernelGGL((mh::spec_centroidal_kernel<TP, double, true, true>), g, dim3(64), 0, (hipStream_t)stream, A);
   else if (id)
      hipLaunchKernelGGL((mh::spec_centroidal_kernel<TP, double, true, false>), g, dim3(64), 0, (hipStream_t)stream, A);
   else if (wb)
      hipLaunchKernelGGL((mh::spec_centroidal_kernel<TP, double, false, true>), g, dim3(64), 0, (hipStream_t)stream, A);
   else
      hipLaunchKernelGGL((mh::spec_centroidal_kernel<TP, double, false, false>), g, dim3(64), 0, (hipStream_t)stream, A);
   return (int)hipGetLastError();
#endif
}
int mh_spec_launch_centroidal(int flags, const void *args, int grid, void *stream) { return mh_spec_launch_centroidal_parts(flags, args, grid, 1, stream); }
// tree-split CRBA: identity maps, AoS, packed image + limb exchange in LDS
static long crba_split_lds(int lanes_per_group)
{ // lane-major image (odd pitch) of lanes_per_group rows + limb exchange records of as many lanes (odd pitch) + entry -> slot table
   return ((long)(mh::HMap<TP>::T.n_slots | 1) + (long)((SPL::n_limbs() * 10) | 1)) * lanes_per_group * (long)sizeof(double)
          + (long)((mh::HMap<TP>::NV * mh::HMap<TP>::NV * 2 + 7) & ~7);
}
long mh_spec_crba_split_lds_bytes(void) { return crba_split_lds(64); }
// LDS of one workgroup of the fused RNEA + CRBA launch with CRBA groups of lanes_per_group configurations (the host sizes the grid by it)
long mh_spec_rnea_crba_lds_bytes(int lanes_per_group, int nq, int nv)
{
   return std::max(split_lds_bytes(0, F_IO_LDS, nq, nv), crba_split_lds(lanes_per_group));
}
int mh_spec_crba_split_usable(void) { return SPL::usable() && mh_spec_crba_split_lds_bytes() <= 160 * 1024 ? 1 : 0; }
int mh_spec_launch_crba_split(const void *args, int groups, int lanes_per_group, void *stream)
{
   if constexpr (SPL::usable())
   {
      const mh::Args<double> &A = *(const mh::Args<double> *)args;
      auto kern = &mh::spec_crba_split_kernel<TP, double>;
      const size_t lds = (size_t)crba_split_lds(std::max(1, std::min(64, lanes_per_group)));
      static LdsAttr attr;
      if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(kern), lds, attr); e != hipSuccess)
         return (int)e;
      hipLaunchKernelGGL(kern, dim3(groups), dim3(256), lds, (hipStream_t)stream, A, lanes_per_group);
      return (int)hipGetLastError();
   }
   else
      return (int)hipErrorNotSupported;
}
// RNEA and CRBA of the same configurations in one launch (identity maps, AoS, rows staged in LDS): rnea_groups workgroups of 64
// configurations for the RNEA + crba_groups workgroups of lanes_per_group for the CRBA; args->out = tau, args->outb = H
int mh_spec_launch_rnea_crba(const void *args, int rnea_groups, int crba_groups, int lanes_per_group, void *stream)
{
#ifdef MH_SPEC_MINIMAL
   return (int)hipErrorNotSupported;
#else
   if constexpr (SPL::usable())
   {
      const mh::Args<double> &A = *(const mh::Args<double> *)args;
      auto kern = &mh::spec_rnea_crba_split_kernel<TP, double>;
      const size_t lds = (size_t)mh_spec_rnea_crba_lds_bytes(std::max(1, std::min(64, lanes_per_group)), A.m.nq, A.m.nv);
      if (lds > 160 * 1024 || !mh_spec_crba_split_usable())
         return (int)hipErrorNotSupported;
      static LdsAttr attr;
      if (const hipError_t e = ensure_lds_attr(reinterpret_cast<const void *>(kern), lds, attr); e != hipSuccess)
         return (int)e;
      hipLaunchKernelGGL(kern, dim3(rnea_groups + crba_groups), dim3(256), lds, (hipStream_t)stream, A, lanes_per_group, rnea_groups);
      return (int)hipGetLastError();
   }
   else
      return (int)hipErrorNotSupported;
#endif
}
// algo: 0 = RNEA, 1 = ABA; fp64 only.  args points to mh::Args<double>.
int mh_spec_launch(int algo, int flags, const void *args, int grid, void *stream)
{
#ifdef MH_SPEC_MINIMAL
   return (int)hipErrorNotSupported;
#else
   const mh::Args<double> &A = *(const mh::Args<double> *)args;
   const size_t lds = (size_t)lds_bytes(algo, flags, A.m.nq, A.m.nv);
   if (algo == 0)
      return (int)go_flags<0>(flags, A, grid, lds, (hipStream_t)stream);
   if (algo == 1)
      return (int)go_flags<1>(flags, A, grid, lds, (hipStream_t)stream);
   return (int)hipErrorNotSupported;
#endif
}
#endif
}
