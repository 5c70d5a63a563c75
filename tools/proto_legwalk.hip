// proto_legwalk.hip -- gated prototype of sub-configuration parallelism.
//
// The inertia job of the headline kernel (mh_zv_kernels.h, ZvIn) walks a leg of the humanoid inwards as SIX dependent body steps on ONE
// wave: per step 1/D, the rank-1 downdate Ia = IA - U D^-1 U^T, the congruence with R_b Rz(q) of the three 3 x 3 blocks (A, L, C), the
// translation and the sum into the parent (ForwardDynamicsCalculator.java:1146-1235 with p = c = 0, ArticulatedBodyInertia.java:359-375).
// Here the same arithmetic (the library's own building blocks, mh_kernels.h) as
//   (S)  one wave per leg: the reference point, 4 waves = 4 legs per workgroup of 64 configurations, as in the kernel;
//   (P)  TWO waves per leg that split every body step and exchange through LDS behind workgroup barriers (8 waves per workgroup):
//          wave Q owns L, the z column of A and the z row of C (what the division and U need), wave P owns the xy block of A and the
//          x, y rows of C (what survives the downdate); per step  Q: 1/D, s = U/D -> LDS | P: downdate, congruence of A and C |
//          Q: downdate and congruence of L | both: L'.z* <-> C'.z* and the A'.z* partials through LDS | translation, sum;
//   (P2) the same with P's congruence moved in FRONT of the first exchange (the congruence is linear: R (A - s u^T) R^T =
//          R A R^T - (R s)(R u)^T), so that it overlaps Q's division;
//   (M)  the congruence's 3 x 3 products as v_mfma_f64_4x4x4_4b contractions (padded to 4 x 4, operands already in the MFMA layout: the
//          best case for the matrix pipe, no layout conversion charged) against the same products as v_fma_f64 with lane = configuration.
// Stamps: s_memtime (shader cycles) and s_memrealtime (100 MHz) around the walk of every wave; the host prints, per variant, the median
// over the workgroups of (last wave out - first wave in).  Build and run: tools/proto_legwalk.sh (GPU box).
#include "../mecano_amd/csrc/mh_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace mh;

#define CHECK(x)                                                                  \
   do                                                                             \
   {                                                                              \
      hipError_t e_ = (x);                                                        \
      if (e_ != hipSuccess)                                                       \
      {                                                                           \
         fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
         exit(1);                                                                 \
      }                                                                           \
   } while (0)

constexpr int NB = 6;       // bodies of the limb: a leg of the 30-DoF humanoid
constexpr int CS = 24;      // doubles per body: m, h(3), I(6), R_b(9), p(3), pad(2)
constexpr int NCHAIN = 4;   // limbs per group of 64 configurations (the kernel's four waves)
constexpr int NST = 9;      // values a body step leaves for the fold / outward sweeps: U/D (6), 1/D, cos, sin

#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the model's constants are read with scalar loads (constant address space), as in the library (mh_device.h: CRef)
typedef const double __attribute__((address_space(4))) *cptr;

struct Stamp
{
   long long c0, c1, r0, r1;
};

template <bool TOP>
__device__ __forceinline__ XF<double> load_xb(cptr c)
{
   XF<double> X;
   X.R = M3<double>{c[10], c[11], c[12], c[13], c[14], c[15], c[16], c[17], c[18]};
   if constexpr (TOP)
      X.p = V3<double>{c[19], c[20], c[21]}; // the limb's root hangs off a trunk body: general offset
   else
      X.p = V3<double>{c[19], 0.0, 0.0};     // first child of a 1-DoF joint: its origin lies on the parent's x axis (mh_api.hip, canonical frames)
   return X;
}
__device__ __forceinline__ RI<double> load_ri(cptr c)
{
   return RI<double>{c[0], V3<double>{c[1], c[2], c[3]}, S3<double>{c[4], c[5], c[6], c[7], c[8], c[9]}};
}

// ---------------------------------------------------------------------------------------------------- (S) one wave per leg
__global__ void __launch_bounds__(256) walk_single(const double *q, const double *consts, double *out, Stamp *stamps, int rounds, int cold)
{
   if (cold) // the headline kernel is 268 KB of straight-line code that every wave runs through ONCE: its instruction cache is always cold
      asm volatile("s_icache_inv\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0" ::: "memory");
   extern __shared__ double lds[];
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const cptr cbase = (cptr)consts + (size_t)wave * NB * CS;
   const long row = (long)blockIdx.x * 64 + lane;
   double qv[NB];
#pragma unroll
   for (int j = 0; j < NB; j++)
      qv[j] = q[row * (NCHAIN * NB) + wave * NB + j];
   double *st = lds + (size_t)wave * NB * NST * 64 + lane;
   __syncthreads();
   ABI<double> up = ABI<double>{};
   Stamp s;
   for (int r = 0; r < rounds; r++) // (round 0 warms the instruction cache; the stamps of the last round are kept)
   {
      s.c0 = __builtin_amdgcn_s_memtime(), s.r0 = __builtin_amdgcn_s_memrealtime();
      JX<double> jx[NB];
#pragma unroll
      for (int j = 0; j < NB; j++)
         sincos_t(qv[j], jx[j].s, jx[j].c), jx[j].d = 0.0;
#pragma unroll
      for (int j = NB - 1; j >= 0; j--)
      {
         FENCE();
         cptr c = cbase + j * CS;
         asm volatile("" : "+s"(c));
         ABI<double> IA = abi_from_rigid(load_ri(c));
         if (j != NB - 1)
            add(IA, up);
         FENCE();
         const XF<double> Xb = j == 0 ? load_xb<true>(c) : load_xb<false>(c);
         const V3<double> ua{IA.A.xz, IA.A.yz, IA.A.zz}, ul{IA.C.zx, IA.C.zy, IA.C.zz};
         const double dinv = 1.0 / IA.A.zz;
         const V3<double> sa = dinv * ua, sl = dinv * ul;
         double *sj = st + (size_t)j * NST * 64;
         sj[0] = sa.x, sj[64] = sa.y, sj[128] = sa.z, sj[192] = sl.x, sj[256] = sl.y, sj[320] = sl.z, sj[384] = dinv, sj[448] = jx[j].c, sj[512] = jx[j].s;
         rank1_down_revolute(IA, ua, ul, dinv);
         abi_up(JT_REVOLUTE, jx[j], Xb, IA);
         up = IA;
      }
      FENCE();
      s.c1 = __builtin_amdgcn_s_memtime(), s.r1 = __builtin_amdgcn_s_memrealtime();
      qv[0] += up.A.xx * 1e-300; // the rounds depend on each other: nothing is hoisted or dropped
   }
   double *o = out + ((size_t)blockIdx.x * NCHAIN + wave) * 21 * 64 + lane;
   o[0] = up.A.xx, o[64] = up.A.xy, o[128] = up.A.xz, o[192] = up.A.yy, o[256] = up.A.yz, o[320] = up.A.zz;
   o[384] = up.L.xx, o[448] = up.L.xy, o[512] = up.L.xz, o[576] = up.L.yy, o[640] = up.L.yz, o[704] = up.L.zz;
   o[768] = up.C.xx, o[832] = up.C.xy, o[896] = up.C.xz, o[960] = up.C.yx, o[1024] = up.C.yy, o[1088] = up.C.yz, o[1152] = up.C.zx, o[1216] = up.C.zy,
   o[1280] = up.C.zz;
   if (lane == 0)
      stamps[blockIdx.x * 8 + wave] = s;
}

// ---------------------------------------------------------------------------------------------------- (P) two waves per leg
// LDS per leg: E1 (Q -> P) 7 slots, E2a (Q -> P) 6 slots, E2b (P -> Q) 6 slots, sincos 2 * NB slots; all [slot][64 lanes]
constexpr int XSLOTS = 7 + 6 + 6 + 2 * NB;

template <int MODE> // 1: straightforward; 2: P's congruence in front of the first exchange (linearity)
__global__ void __launch_bounds__(512) walk_pair(const double *q, const double *consts, double *out, Stamp *stamps, int rounds, int same_simd)
{
   extern __shared__ double lds[];
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   // which two waves share a leg: (w, w + 4) sit on the same SIMD (the workgroup's waves go round the four SIMDs), (2 c, 2 c + 1) do not
   const int chain = same_simd ? (wave & 3) : (wave >> 1);
   const bool isQ = same_simd ? (wave >= 4) : (wave & 1);
   const cptr cbase = (cptr)consts + (size_t)chain * NB * CS;
   const long row = (long)blockIdx.x * 64 + lane;
   double qv[NB];
#pragma unroll
   for (int j = 0; j < NB; j++)
      qv[j] = q[row * (NCHAIN * NB) + chain * NB + j];
   constexpr int NSTP = 7; // (cos, sin) stay where the exchange left them
   double *st = lds + (size_t)chain * NB * NSTP * 64 + lane;
   double *xs = lds + (size_t)NCHAIN * NB * NSTP * 64 + (size_t)chain * XSLOTS * 64 + lane;
   double *e1 = xs, *e2a = xs + 7 * 64, *e2b = xs + 13 * 64, *ecs = xs + 19 * 64;
   __syncthreads();
   Stamp s;
   // P's state
   double axx = 0, axy = 0, ayy = 0, cxx = 0, cxy = 0, cxz = 0, cyx = 0, cyy = 0, cyz = 0;
   // Q's state
   S3<double> L{};
   double axz = 0, ayz = 0, azz = 0, czx = 0, czy = 0, czz = 0;
   ABI<double> fin = ABI<double>{};
   for (int r = 0; r < rounds; r++)
   {
      s.c0 = __builtin_amdgcn_s_memtime(), s.r0 = __builtin_amdgcn_s_memrealtime();
      // (cos, sin): each wave forms half of the leg's pairs and leaves them in LDS for its partner
      JX<double> jx[NB];
#pragma unroll
      for (int j = 0; j < NB; j++)
      {
         jx[j].d = 0.0;
         if ((j < NB / 2) != isQ)
         {
            sincos_t(qv[j], jx[j].s, jx[j].c);
            ecs[(2 * j) * 64] = jx[j].c, ecs[(2 * j + 1) * 64] = jx[j].s;
         }
      }
      lds_barrier();
#pragma unroll
      for (int j = 0; j < NB; j++)
         if ((j < NB / 2) == isQ)
            jx[j].c = ecs[(2 * j) * 64], jx[j].s = ecs[(2 * j + 1) * 64];
#pragma unroll
      for (int j = NB - 1; j >= 0; j--)
      {
         cptr c = cbase + j * CS;
         asm volatile("" : "+s"(c));
         const RI<double> I = load_ri(c);
         const XF<double> Xb = j == 0 ? load_xb<true>(c) : load_xb<false>(c);
         const double a = Xb.p.x;
         const M3<double> R = revolute_rotation(jx[j], Xb.R);
         if (isQ)
         {
            // ---- this body's share of the rigid inertia (ArticulatedBodyInertia.java:176-186: A = I, L = m 1, C = [h]x)
            if (j == NB - 1)
               L = S3<double>{I.m, 0.0, 0.0, I.m, 0.0, I.m}, axz = I.I.xz, ayz = I.I.yz, azz = I.I.zz, czx = -I.h.y, czy = I.h.x, czz = 0.0;
            else
               L.xx += I.m, L.yy += I.m, L.zz += I.m, axz += I.I.xz, ayz += I.I.yz, azz += I.I.zz, czx -= I.h.y, czy += I.h.x;
            const double dinv = 1.0 / azz;
            const double sx = dinv * axz, sy = dinv * ayz, sz = dinv * azz;
            const V3<double> sl{dinv * czx, dinv * czy, dinv * czz};
            e1[0] = sx, e1[64] = sy, e1[128] = axz, e1[192] = ayz, e1[256] = czx, e1[320] = czy, e1[384] = czz;
            double *sj = st + (size_t)j * NSTP * 64;
            sj[0] = sx, sj[64] = sy, sj[128] = sz, sj[192] = sl.x, sj[256] = sl.y, sj[320] = sl.z, sj[384] = dinv;
            lds_barrier(); // ---- exchange 1: s and U on their way to P
            L.xx -= sl.x * czx, L.xy -= sl.x * czy, L.xz -= sl.x * czz, L.yy -= sl.y * czy, L.yz -= sl.y * czz, L.zz -= sl.z * czz;
            const S3<double> Lr = conj(R, L);
            if (j == 0)
            {
               e2a[0] = Lr.xx, e2a[64] = Lr.xy, e2a[128] = Lr.xz, e2a[192] = Lr.yy, e2a[256] = Lr.yz, e2a[320] = Lr.zz;
               lds_barrier();
            }
            else
            {
               e2a[0] = Lr.xz, e2a[64] = Lr.yz, e2a[128] = Lr.zz;
               lds_barrier(); // ---- exchange 2
               const double fxz = e2b[0], pyz = e2b[64], pzz = e2b[128], rzx = e2b[192], rzy = e2b[256], rzz = e2b[320];
               const double a2 = a * a;
               L = Lr;
               czx = rzx + a * Lr.xy, czy = rzy + a * Lr.yy, czz = rzz + a * Lr.yz;
               axz = fxz, ayz = pyz - a2 * Lr.yz, azz = pzz + a2 * Lr.yy;
            }
         }
         else
         {
            if (j == NB - 1)
               axx = I.I.xx, axy = I.I.xy, ayy = I.I.yy, cxx = 0.0, cxy = -I.h.z, cxz = I.h.y, cyx = I.h.z, cyy = 0.0, cyz = -I.h.x;
            else
               axx += I.I.xx, axy += I.I.xy, ayy += I.I.yy, cxy -= I.h.z, cxz += I.h.y, cyx += I.h.z, cyz -= I.h.x;
            S3<double> Ar;
            M3<double> Cr;
            if constexpr (MODE == 2)
            {
               // the congruence of the un-downdated blocks while Q divides ...
               const double tax = R.xx * axx + R.xy * axy, tay = R.xx * axy + R.xy * ayy, tbx = R.yx * axx + R.yy * axy, tby = R.yx * axy + R.yy * ayy,
                            tcx = R.zx * axx + R.zy * axy, tcy = R.zx * axy + R.zy * ayy;
               Ar.xx = tax * R.xx + tay * R.xy, Ar.xy = tax * R.yx + tay * R.yy, Ar.xz = tax * R.zx + tay * R.zy;
               Ar.yy = tbx * R.yx + tby * R.yy, Ar.yz = tbx * R.zx + tby * R.zy, Ar.zz = tcx * R.zx + tcy * R.zy;
               const double txx = R.xx * cxx + R.xy * cyx, txy = R.xx * cxy + R.xy * cyy, txz = R.xx * cxz + R.xy * cyz;
               const double tyx = R.yx * cxx + R.yy * cyx, tyy = R.yx * cxy + R.yy * cyy, tyz = R.yx * cxz + R.yy * cyz;
               const double tzx = R.zx * cxx + R.zy * cyx, tzy = R.zx * cxy + R.zy * cyy, tzz = R.zx * cxz + R.zy * cyz;
               Cr.xx = txx * R.xx + txy * R.xy + txz * R.xz, Cr.xy = txx * R.yx + txy * R.yy + txz * R.yz, Cr.xz = txx * R.zx + txy * R.zy + txz * R.zz;
               Cr.yx = tyx * R.xx + tyy * R.xy + tyz * R.xz, Cr.yy = tyx * R.yx + tyy * R.yy + tyz * R.yz, Cr.yz = tyx * R.zx + tyy * R.zy + tyz * R.zz;
               Cr.zx = tzx * R.xx + tzy * R.xy + tzz * R.xz, Cr.zy = tzx * R.yx + tzy * R.yy + tzz * R.yz, Cr.zz = tzx * R.zx + tzy * R.zy + tzz * R.zz;
               lds_barrier(); // ---- exchange 1
               const double sx = e1[0], sy = e1[64], uax = e1[128], uay = e1[192], ulx = e1[256], uly = e1[320], ulz = e1[384];
               // ... and the rank-1 term in the rotated frame behind it: (R s)(R u)^T
               const V3<double> rs{R.xx * sx + R.xy * sy, R.yx * sx + R.yy * sy, R.zx * sx + R.zy * sy};
               const V3<double> ru{R.xx * uax + R.xy * uay, R.yx * uax + R.yy * uay, R.zx * uax + R.zy * uay};
               const V3<double> rl{R.xx * ulx + R.xy * uly + R.xz * ulz, R.yx * ulx + R.yy * uly + R.yz * ulz, R.zx * ulx + R.zy * uly + R.zz * ulz};
               Ar.xx -= rs.x * ru.x, Ar.xy -= rs.x * ru.y, Ar.xz -= rs.x * ru.z, Ar.yy -= rs.y * ru.y, Ar.yz -= rs.y * ru.z, Ar.zz -= rs.z * ru.z;
               Cr.xx -= rs.x * rl.x, Cr.xy -= rs.x * rl.y, Cr.xz -= rs.x * rl.z, Cr.yx -= rs.y * rl.x, Cr.yy -= rs.y * rl.y, Cr.yz -= rs.y * rl.z;
               Cr.zx -= rs.z * rl.x, Cr.zy -= rs.z * rl.y, Cr.zz -= rs.z * rl.z;
            }
            else
            {
               lds_barrier(); // ---- exchange 1
               const double sx = e1[0], sy = e1[64], uax = e1[128], uay = e1[192], ulx = e1[256], uly = e1[320], ulz = e1[384];
               axx -= sx * uax, axy -= sx * uay, ayy -= sy * uay;
               cxx -= sx * ulx, cxy -= sx * uly, cxz -= sx * ulz, cyx -= sy * ulx, cyy -= sy * uly, cyz -= sy * ulz;
               const ABI<double> Ia{S3<double>{axx, axy, 0.0, ayy, 0.0, 0.0}, S3<double>{}, M3<double>{cxx, cxy, cxz, cyx, cyy, cyz, 0.0, 0.0, 0.0}};
               Ar = conj(R, Ia.A); // (the structural zeros fold: -fno-signed-zeros -ffinite-math-only, as in the library)
               Cr = conj(R, Ia.C);
            }
            if (j == 0)
            {
               lds_barrier();
               ABI<double> T{Ar, S3<double>{e2a[0], e2a[64], e2a[128], e2a[192], e2a[256], e2a[320]}, Cr};
               translate(T, Xb.p);
               fin = T;
            }
            else
            {
               e2b[0] = Ar.xz + a * Cr.xy, e2b[64] = Ar.yz - a * Cr.zz + a * Cr.yy, e2b[128] = Ar.zz + a * Cr.zy + a * Cr.zy;
               e2b[192] = Cr.zx, e2b[256] = Cr.zy, e2b[320] = Cr.zz;
               lds_barrier(); // ---- exchange 2
               const double lxz = e2a[0], lyz = e2a[64], lzz = e2a[128];
               const double nyx = Cr.yx - a * lxz, nyy = Cr.yy - a * lyz, nyz = Cr.yz - a * lzz;
               axx = Ar.xx, axy = Ar.xy - a * Cr.xz, ayy = Ar.yy - a * Cr.yz - a * nyz;
               cxx = Cr.xx, cxy = Cr.xy, cxz = Cr.xz, cyx = nyx, cyy = nyy, cyz = nyz;
            }
         }
      }
      FENCE();
      s.c1 = __builtin_amdgcn_s_memtime(), s.r1 = __builtin_amdgcn_s_memrealtime();
      qv[0] += (fin.A.xx + L.xx) * 1e-300;
      lds_barrier(); // (the next round's first stores must not overtake the partner's last loads)
   }
   if (!isQ)
   {
      double *o = out + ((size_t)blockIdx.x * NCHAIN + chain) * 21 * 64 + lane;
      const ABI<double> &up = fin;
      o[0] = up.A.xx, o[64] = up.A.xy, o[128] = up.A.xz, o[192] = up.A.yy, o[256] = up.A.yz, o[320] = up.A.zz;
      o[384] = up.L.xx, o[448] = up.L.xy, o[512] = up.L.xz, o[576] = up.L.yy, o[640] = up.L.yz, o[704] = up.L.zz;
      o[768] = up.C.xx, o[832] = up.C.xy, o[896] = up.C.xz, o[960] = up.C.yx, o[1024] = up.C.yy, o[1088] = up.C.yz, o[1152] = up.C.zx, o[1216] = up.C.zy,
      o[1280] = up.C.zz;
   }
   if (lane == 0)
      stamps[blockIdx.x * 8 + wave] = s;
}

// ---------------------------------------------------------------------------------------------------- (M) the congruence on the matrix pipe
// One body step's congruence = R X R^T for X in {A, L, C}: six 3 x 3 x 3 products per configuration.  v_mfma_f64_4x4x4_4b computes FOUR
// independent 4 x 4 x 4 products per instruction (one element of A, B and D per lane): 64 configurations need 6 * 64 / 4 = 96 of them per
// body step, with 27 of every 64 multiply-adds useful (3 x 3 padded to 4 x 4).  Timed: those 96 MFMAs (dependent in pairs: T = R X, then
// T R^T) with every operand already in registers in the MFMA layout -- no conversion from lane = configuration charged -- against the 114
// v_fma / v_mul the library issues for the same congruence with lane = configuration (conj of A with its zero row and column, of L, of C
// with its zero row).
typedef double v1d;
__global__ void __launch_bounds__(256) congruence_mfma(double *out, Stamp *stamps, int steps, double seed)
{
   const int lane = threadIdx.x & 63;
   double r = seed + lane * 1e-3, x[3] = {1.0 + lane, 2.0 + lane, 3.0 + lane};
   Stamp s;
   s.c0 = __builtin_amdgcn_s_memtime(), s.r0 = __builtin_amdgcn_s_memrealtime();
   for (int it = 0; it < steps; it++)
   {
      // 16 groups of four configurations; per group three blocks, two dependent products each
#pragma unroll
      for (int g = 0; g < 16; g++)
#pragma unroll
         for (int b = 0; b < 3; b++)
         {
            double t = __builtin_amdgcn_mfma_f64_4x4x4f64(r, x[b], 0.0, 0, 0, 0);
            x[b] = __builtin_amdgcn_mfma_f64_4x4x4f64(t, r, 0.0, 0, 0, 0);
         }
   }
   s.c1 = __builtin_amdgcn_s_memtime(), s.r1 = __builtin_amdgcn_s_memrealtime();
   out[blockIdx.x * 256 + threadIdx.x] = x[0] + x[1] + x[2];
   if (lane == 0)
      stamps[blockIdx.x * 8 + (threadIdx.x >> 6)] = s;
}
__global__ void __launch_bounds__(256) congruence_valu(double *out, Stamp *stamps, int steps, double seed)
{
   const int lane = threadIdx.x & 63;
   M3<double> R{0.36, 0.48, -0.8, -0.8, 0.6, 0.0, 0.48, 0.64, 0.6};
   R.xx += seed * 1e-9 * lane;
   ABI<double> I{S3<double>{1.0 + lane, 0.1, 0.0, 2.0, 0.0, 0.0}, S3<double>{3.0, 0.2, 0.3, 4.0, 0.1, 5.0}, M3<double>{0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.0, 0.0, 0.0}};
   Stamp s;
   s.c0 = __builtin_amdgcn_s_memtime(), s.r0 = __builtin_amdgcn_s_memrealtime();
   for (int it = 0; it < steps; it++)
   {
      FENCE();
      rotate(I, R);
      I.A.xz = 0.0, I.A.yz = 0.0, I.A.zz = 0.0, I.C.zx = 0.0, I.C.zy = 0.0, I.C.zz = 0.0; // what the downdate of the next body leaves
   }
   s.c1 = __builtin_amdgcn_s_memtime(), s.r1 = __builtin_amdgcn_s_memrealtime();
   out[blockIdx.x * 256 + threadIdx.x] = I.A.xx + I.L.xx + I.C.xx + I.C.yz + I.L.yz + I.A.xy + I.A.yy;
   if (lane == 0)
      stamps[blockIdx.x * 8 + (threadIdx.x >> 6)] = s;
}

// ---------------------------------------------------------------------------------------------------- host
static double median(std::vector<double> v)
{
   std::sort(v.begin(), v.end());
   return v[v.size() / 2];
}
struct Result
{
   double cycles, us, slowest_wave_us;
};
static Result reduce(const std::vector<Stamp> &st, int groups, int waves)
{
   std::vector<double> cyc, us, own;
   for (int g = 0; g < groups; g++)
   {
      long long c0 = st[g * 8].c0, c1 = st[g * 8].c1, r0 = st[g * 8].r0, r1 = st[g * 8].r1;
      double slow = 0;
      for (int w = 0; w < waves; w++)
      {
         const Stamp &s = st[g * 8 + w];
         c0 = std::min(c0, s.c0), c1 = std::max(c1, s.c1), r0 = std::min(r0, s.r0), r1 = std::max(r1, s.r1);
         slow = std::max(slow, (double)(s.r1 - s.r0) * 0.01);
      }
      cyc.push_back((double)(c1 - c0)), us.push_back((double)(r1 - r0) * 0.01), own.push_back(slow);
   }
   return Result{median(cyc), median(us), median(own)};
}

int main(int argc, char **argv)
{
   const int groups = argc > 1 ? atoi(argv[1]) : 64; // the headline: 4096 configurations = 64 groups, one workgroup per CU
   const int rounds = 3, launches = 400;
   const long B = (long)groups * 64;
   srand(7);
   auto uni = [] { return rand() / (double)RAND_MAX; };
   std::vector<double> hq((size_t)B * NCHAIN * NB), hc((size_t)NCHAIN * NB * CS, 0.0);
   for (double &v : hq)
      v = (uni() * 2 - 1) * 3.14159;
   for (int k = 0; k < NCHAIN * NB; k++)
   {
      double *c = &hc[(size_t)k * CS];
      c[0] = 0.1 + uni();
      for (int i = 1; i <= 3; i++)
         c[i] = c[0] * (uni() * 2 - 1) * 0.3;
      // J = L L^T + parallel axis share: positive definite, as MecanoRandomTools.nextSymmetricPositiveDefiniteMatrix3D draws it
      double Lm[3][3] = {{1e-4 + 2 * uni(), 0, 0}, {uni() - 0.5, 1e-4 + 2 * uni(), 0}, {uni() - 0.5, uni() - 0.5, 1e-4 + 2 * uni()}};
      double J[3][3];
      for (int i = 0; i < 3; i++)
         for (int j = 0; j < 3; j++)
         {
            J[i][j] = 0;
            for (int l = 0; l < 3; l++)
               J[i][j] += Lm[i][l] * Lm[j][l];
         }
      const double hh = (c[1] * c[1] + c[2] * c[2] + c[3] * c[3]) / c[0];
      c[4] = J[0][0] + hh, c[5] = J[0][1], c[6] = J[0][2], c[7] = J[1][1] + hh, c[8] = J[1][2], c[9] = J[2][2] + hh;
      // a random rotation from a random unit quaternion
      double qx = uni() - 0.5, qy = uni() - 0.5, qz = uni() - 0.5, qs = uni() - 0.5;
      const double n = std::sqrt(qx * qx + qy * qy + qz * qz + qs * qs);
      qx /= n, qy /= n, qz /= n, qs /= n;
      c[10] = 1 - 2 * (qy * qy + qz * qz), c[11] = 2 * (qx * qy - qz * qs), c[12] = 2 * (qx * qz + qy * qs);
      c[13] = 2 * (qx * qy + qz * qs), c[14] = 1 - 2 * (qx * qx + qz * qz), c[15] = 2 * (qy * qz - qx * qs);
      c[16] = 2 * (qx * qz - qy * qs), c[17] = 2 * (qy * qz + qx * qs), c[18] = 1 - 2 * (qx * qx + qy * qy);
      c[19] = uni() * 2 - 1, c[20] = uni() * 2 - 1, c[21] = uni() * 2 - 1;
   }
   double *dq, *dc, *dout1, *dout2;
   Stamp *dst;
   const size_t out_bytes = (size_t)groups * NCHAIN * 21 * 64 * sizeof(double);
   CHECK(hipMalloc(&dq, hq.size() * 8));
   CHECK(hipMalloc(&dc, hc.size() * 8));
   CHECK(hipMalloc(&dout1, out_bytes));
   CHECK(hipMalloc(&dout2, out_bytes));
   CHECK(hipMalloc(&dst, (size_t)groups * 8 * sizeof(Stamp)));
   CHECK(hipMemcpy(dq, hq.data(), hq.size() * 8, hipMemcpyHostToDevice));
   CHECK(hipMemcpy(dc, hc.data(), hc.size() * 8, hipMemcpyHostToDevice));
   const size_t lds_single = (size_t)NCHAIN * NB * NST * 64 * 8, lds_pair = (size_t)NCHAIN * NB * 7 * 64 * 8 + (size_t)NCHAIN * XSLOTS * 64 * 8;
   CHECK(hipFuncSetAttribute((const void *)walk_single, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_single));
   CHECK(hipFuncSetAttribute((const void *)walk_pair<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair));
   CHECK(hipFuncSetAttribute((const void *)walk_pair<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair));
   std::vector<Stamp> st((size_t)groups * 8);
   std::vector<double> ref((size_t)groups * NCHAIN * 21 * 64), got(ref.size());
   auto timed = [&](const char *name, auto launch, int waves, bool check) {
      for (int k = 0; k < launches; k++)
         launch();
      CHECK(hipDeviceSynchronize());
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
   CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0));
      for (int k = 0; k < 200; k++)
         launch();
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(st.data(), dst, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
      const Result r = reduce(st, groups, waves);
      double err = -1;
      if (check)
      {
         CHECK(hipMemcpy(got.data(), dout2, out_bytes, hipMemcpyDeviceToHost));
         double worst = 0, scale = 0;
         for (size_t k = 0; k < ref.size(); k++)
            worst = std::max(worst, std::fabs(got[k] - ref[k])), scale = std::max(scale, std::fabs(ref[k]));
         err = worst / scale;
      }
      printf("%-58s walk %8.0f cycles = %6.2f us (slowest wave alone %6.2f us)   kernel %6.2f us per launch", name, r.cycles, r.us, r.slowest_wave_us, ms * 1e3 / 200 / 1.0);
      if (check)
         printf("   max |diff| / max |ref| vs (S) %.2e", err);
      printf("\n");
      return r;
   };
   printf("# leg walk of the inertia job (6 revolute bodies inwards), %d groups of 64 configurations, one workgroup per group; the walk is timed in its\n"
          "# last of %d rounds inside one launch (warm instruction cache); cycles = s_memtime, us = s_memrealtime (100 MHz)\n",
          groups, rounds);
   timed("(S0) one wave per leg, ONE round behind s_icache_inv (cold code)", [&] { hipLaunchKernelGGL(walk_single, dim3(groups), dim3(256), lds_single, 0, dq, dc, dout1, dst, 1, 1); }, 4, false);
   const Result rs = timed("(S)  one wave per leg, 4 waves per workgroup", [&] { hipLaunchKernelGGL(walk_single, dim3(groups), dim3(256), lds_single, 0, dq, dc, dout1, dst, rounds, 0); }, 4, false);
   CHECK(hipMemcpy(ref.data(), dout1, out_bytes, hipMemcpyDeviceToHost));
   for (int same = 0; same < 2; same++)
   {
      const Result r1 = timed(same ? "(P)  two waves per leg, partners on the SAME SIMD" : "(P)  two waves per leg, partners on different SIMDs",
                              [&] { hipLaunchKernelGGL(walk_pair<1>, dim3(groups), dim3(512), lds_pair, 0, dq, dc, dout2, dst, rounds, same); }, 8, true);
      const Result r2 = timed(same ? "(P2) ... congruence ahead of the exchange, SAME SIMD" : "(P2) ... congruence ahead of the exchange, different SIMDs",
                              [&] { hipLaunchKernelGGL(walk_pair<2>, dim3(groups), dim3(512), lds_pair, 0, dq, dc, dout2, dst, rounds, same); }, 8, true);
      printf("#    -> (P) %.1f %%, (P2) %.1f %% of (S)\n", 100.0 * r1.us / rs.us, 100.0 * r2.us / rs.us);
   }
   // ---- (M)
   double *dm;
   CHECK(hipMalloc(&dm, (size_t)groups * 256 * 8));
   const int steps = 64;
   const Result rm = timed("(M)  congruence of one body step as 96 v_mfma_f64_4x4x4_4b", [&] { hipLaunchKernelGGL(congruence_mfma, dim3(groups), dim3(256), 0, 0, dm, dst, steps, 0.5); }, 4, false);
   const Result rv = timed("(V)  the same congruence as v_fma_f64, lane = configuration", [&] { hipLaunchKernelGGL(congruence_valu, dim3(groups), dim3(256), 0, 0, dm, dst, steps, 0.5); }, 4, false);
   printf("#    per body step and wave of 64 configurations: MFMA %.0f cycles, VALU %.0f cycles (%d steps timed)\n", rm.cycles / steps, rv.cycles / steps, steps);
   return 0;
}
