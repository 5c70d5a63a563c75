// mh_device.h -- small spatial-algebra building blocks for the gfx950 kernels.
//
// Everything here works on plain scalars in registers (lane = one configuration), in the engine's
// *canonical joint frames*: the frame after every 1-DoF joint is rotated once, on the host, so that the
// joint axis is +z.  A revolute joint is then Rz(q), a prismatic joint a slide along z, S is a unit
// basis vector, tau = one component of the joint wrench, U = one column of the articulated inertia.
// Results (tau, qdd, H) are scalars per DoF and do not depend on that choice of frame.
//
// Spatial vectors are (angular, linear) like Mecano's (spatial/interfaces/SpatialVectorReadOnly.java).
#pragma once
#include <hip/hip_runtime.h>

#define MH_DEV __device__ __forceinline__

namespace mh
{
template <typename T>
struct V3
{
   T x, y, z;
};
template <typename T>
struct SV
{ // spatial vector
   V3<T> a, l;
};
template <typename T>
struct M3
{ // general 3x3, row-major
   T xx, xy, xz, yx, yy, yz, zx, zy, zz;
};
template <typename T>
struct S3
{ // symmetric 3x3
   T xx, xy, xz, yy, yz, zz;
};
template <typename T>
struct XF
{ // rigid transform child -> parent: x_p = R x_c + p
   M3<T> R;
   V3<T> p;
};
// rigid-body inertia about a frame origin: mass, first moment h = m c, rotational inertia I about the origin
template <typename T>
struct RI
{
   T m;
   V3<T> h;
   S3<T> I;
};
// articulated-body inertia [[A, C], [C^T, L]]  (algorithms/ArticulatedBodyInertia.java:42-55)
template <typename T>
struct ABI
{
   S3<T> A, L;
   M3<T> C;
};

template <typename T>
MH_DEV V3<T> v3(T x, T y, T z)
{
   return V3<T>{x, y, z};
}
template <typename T>
MH_DEV V3<T> operator+(V3<T> a, V3<T> b)
{
   return {a.x + b.x, a.y + b.y, a.z + b.z};
}
template <typename T>
MH_DEV V3<T> operator-(V3<T> a, V3<T> b)
{
   return {a.x - b.x, a.y - b.y, a.z - b.z};
}
template <typename T>
MH_DEV V3<T> operator*(T s, V3<T> a)
{
   return {s * a.x, s * a.y, s * a.z};
}
template <typename T>
MH_DEV V3<T> cross(V3<T> a, V3<T> b)
{
   return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T>
MH_DEV T dot(V3<T> a, V3<T> b)
{
   return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <typename T>
MH_DEV V3<T> mul(const M3<T> &R, V3<T> v)
{ // R v
   return {R.xx * v.x + R.xy * v.y + R.xz * v.z, R.yx * v.x + R.yy * v.y + R.yz * v.z, R.zx * v.x + R.zy * v.y + R.zz * v.z};
}
template <typename T>
MH_DEV V3<T> tmul(const M3<T> &R, V3<T> v)
{ // R^T v
   return {R.xx * v.x + R.yx * v.y + R.zx * v.z, R.xy * v.x + R.yy * v.y + R.zy * v.z, R.xz * v.x + R.yz * v.y + R.zz * v.z};
}
template <typename T>
MH_DEV V3<T> mul(const S3<T> &A, V3<T> v)
{
   return {A.xx * v.x + A.xy * v.y + A.xz * v.z, A.xy * v.x + A.yy * v.y + A.yz * v.z, A.xz * v.x + A.yz * v.y + A.zz * v.z};
}
// rotation about z by angle with (c, s):  Rz v  and  Rz^T v
template <typename T>
MH_DEV V3<T> rotz(T c, T s, V3<T> v)
{
   return {c * v.x - s * v.y, s * v.x + c * v.y, v.z};
}
template <typename T>
MH_DEV V3<T> rotzT(T c, T s, V3<T> v)
{
   return {c * v.x + s * v.y, c * v.y - s * v.x, v.z};
}

// ---- motion vectors parent -> child through X (child -> parent pose):  w' = R^T w ; v' = R^T (v + w x p)
template <typename T>
MH_DEV SV<T> motion_to_child(const XF<T> &X, SV<T> m)
{
   // every component is ONE flat sum of products, so that it contracts into a multiply followed by an unbroken chain of FMAs (a sum of two
   // separately parenthesised product sums costs an extra add per component: 13 % of the fp64 instructions of the ABA kernel were such adds)
   SV<T> o;
   o.a = tmul(X.R, m.a);
   const V3<T> &p = X.p;
   const T tx = m.a.y * p.z - m.a.z * p.y + m.l.x, ty = m.a.z * p.x - m.a.x * p.z + m.l.y, tz = m.a.x * p.y - m.a.y * p.x + m.l.z;
   o.l = tmul(X.R, V3<T>{tx, ty, tz});
   return o;
}
// ---- force vectors child -> parent:  f' = R f ; n' = R n + p x f'      (spatial/interfaces/FixedFrameSpatialForceBasics.java:249-259)
template <typename T>
MH_DEV SV<T> force_to_parent(const XF<T> &X, SV<T> w)
{
   SV<T> o;
   const M3<T> &R = X.R;
   const V3<T> &p = X.p;
   o.l = mul(R, w.l);
   o.a.x = R.xx * w.a.x + R.xy * w.a.y + R.xz * w.a.z + p.y * o.l.z - p.z * o.l.y;
   o.a.y = R.yx * w.a.x + R.yy * w.a.y + R.yz * w.a.z + p.z * o.l.x - p.x * o.l.z;
   o.a.z = R.zx * w.a.x + R.zy * w.a.y + R.zz * w.a.z + p.x * o.l.y - p.y * o.l.x;
   return o;
}

// ---- rigid inertia times motion vector: momentum = [I w + h x v ; m v - h x w]
template <typename T>
MH_DEV SV<T> mul(const RI<T> &I, SV<T> v)
{
   SV<T> o;
   const S3<T> &J = I.I;
   const V3<T> &h = I.h;
   o.a.x = J.xx * v.a.x + J.xy * v.a.y + J.xz * v.a.z + h.y * v.l.z - h.z * v.l.y;
   o.a.y = J.xy * v.a.x + J.yy * v.a.y + J.yz * v.a.z + h.z * v.l.x - h.x * v.l.z;
   o.a.z = J.xz * v.a.x + J.yz * v.a.y + J.zz * v.a.z + h.x * v.l.y - h.y * v.l.x;
   o.l.x = I.m * v.l.x - h.y * v.a.z + h.z * v.a.y;
   o.l.y = I.m * v.l.y - h.z * v.a.x + h.x * v.a.z;
   o.l.z = I.m * v.l.z - h.x * v.a.y + h.y * v.a.x;
   return o;
}
// v x* f  (spatial force cross product): [w x n + v x f ; w x f]
template <typename T>
MH_DEV SV<T> crf(SV<T> v, SV<T> f)
{
   SV<T> o;
   o.a.x = v.a.y * f.a.z - v.a.z * f.a.y + v.l.y * f.l.z - v.l.z * f.l.y;
   o.a.y = v.a.z * f.a.x - v.a.x * f.a.z + v.l.z * f.l.x - v.l.x * f.l.z;
   o.a.z = v.a.x * f.a.y - v.a.y * f.a.x + v.l.x * f.l.y - v.l.y * f.l.x;
   o.l = cross(v.a, f.l);
   return o;
}
// v x m (spatial motion cross product): [w x mw ; w x mv + v x mw]
template <typename T>
MH_DEV SV<T> crm(SV<T> v, SV<T> m)
{
   SV<T> o;
   o.a = cross(v.a, m.a);
   o.l.x = v.a.y * m.l.z - v.a.z * m.l.y + v.l.y * m.a.z - v.l.z * m.a.y;
   o.l.y = v.a.z * m.l.x - v.a.x * m.l.z + v.l.z * m.a.x - v.l.x * m.a.z;
   o.l.z = v.a.x * m.l.y - v.a.y * m.l.x + v.l.x * m.a.y - v.l.y * m.a.x;
   return o;
}
template <typename T>
MH_DEV SV<T> operator+(SV<T> a, SV<T> b)
{
   return {a.a + b.a, a.l + b.l};
}
template <typename T>
MH_DEV SV<T> operator*(T s, SV<T> a)
{
   return SV<T>{s * a.a, s * a.l};
}
template <typename T>
MH_DEV SV<T> operator-(SV<T> a, SV<T> b)
{
   return {a.a - b.a, a.l - b.l};
}

// ---- quaternion (x, y, z, s), normalised on input like Euclid's Quaternion.set, to a rotation matrix
template <typename T>
MH_DEV M3<T> quat_to_R(T x, T y, T z, T s)
{
   T inv = T(1) / sqrt(x * x + y * y + z * z + s * s);
   x *= inv, y *= inv, z *= inv, s *= inv;
   M3<T> R;
   R.xx = T(1) - T(2) * (y * y + z * z), R.xy = T(2) * (x * y - z * s), R.xz = T(2) * (x * z + y * s);
   R.yx = T(2) * (x * y + z * s), R.yy = T(1) - T(2) * (x * x + z * z), R.yz = T(2) * (y * z - x * s);
   R.zx = T(2) * (x * z - y * s), R.zy = T(2) * (y * z + x * s), R.zz = T(1) - T(2) * (x * x + y * y);
   return R;
}

// ---- 1 / d by v_rcp_f64 and two Newton steps: the reciprocal the compiler's own fp64 division starts from, without its scaling and fix-up
//      of denormal / overflowing quotients -- twelve dependent instructions on the critical path of every body step of the bias-split
//      forward dynamics against five here; error about one ulp.  For joint-space inertias (D = S^T IA S, sums of rigid inertias: far from
//      either end of the exponent range) and the pivots of the floating base's LDL^T (mh_zv_kernels.h: ZvIn, spd6_factor).
#ifndef MH_FAST_RCP
#define MH_FAST_RCP 1 // 0: IEEE division (A/B measurements)
#endif
MH_DEV double rcp_fast(double d)
{
#if MH_FAST_RCP
   double r = __builtin_amdgcn_rcp(d);
   double e = fma(-d, r, 1.0);
   r = fma(r, e, r);
   e = fma(-d, r, 1.0);
   return fma(r, e, r);
#else
   return 1.0 / d;
#endif
}
MH_DEV float rcp_fast(float d) { return 1.0f / d; }

// ---- R S R^T for symmetric S
template <typename T>
MH_DEV S3<T> conj(const M3<T> &R, const S3<T> &S)
{
   // T = R S
   T txx = R.xx * S.xx + R.xy * S.xy + R.xz * S.xz, txy = R.xx * S.xy + R.xy * S.yy + R.xz * S.yz, txz = R.xx * S.xz + R.xy * S.yz + R.xz * S.zz;
   T tyx = R.yx * S.xx + R.yy * S.xy + R.yz * S.xz, tyy = R.yx * S.xy + R.yy * S.yy + R.yz * S.yz, tyz = R.yx * S.xz + R.yy * S.yz + R.yz * S.zz;
   T tzx = R.zx * S.xx + R.zy * S.xy + R.zz * S.xz, tzy = R.zx * S.xy + R.zy * S.yy + R.zz * S.yz, tzz = R.zx * S.xz + R.zy * S.yz + R.zz * S.zz;
   S3<T> o;
   o.xx = txx * R.xx + txy * R.xy + txz * R.xz;
   o.xy = txx * R.yx + txy * R.yy + txz * R.yz;
   o.xz = txx * R.zx + txy * R.zy + txz * R.zz;
   o.yy = tyx * R.yx + tyy * R.yy + tyz * R.yz;
   o.yz = tyx * R.zx + tyy * R.zy + tyz * R.zz;
   o.zz = tzx * R.zx + tzy * R.zy + tzz * R.zz;
   return o;
}
// ---- R M R^T for a general M
template <typename T>
MH_DEV M3<T> conj(const M3<T> &R, const M3<T> &M)
{
   T txx = R.xx * M.xx + R.xy * M.yx + R.xz * M.zx, txy = R.xx * M.xy + R.xy * M.yy + R.xz * M.zy, txz = R.xx * M.xz + R.xy * M.yz + R.xz * M.zz;
   T tyx = R.yx * M.xx + R.yy * M.yx + R.yz * M.zx, tyy = R.yx * M.xy + R.yy * M.yy + R.yz * M.zy, tyz = R.yx * M.xz + R.yy * M.yz + R.yz * M.zz;
   T tzx = R.zx * M.xx + R.zy * M.yx + R.zz * M.zx, tzy = R.zx * M.xy + R.zy * M.yy + R.zz * M.zy, tzz = R.zx * M.xz + R.zy * M.yz + R.zz * M.zz;
   M3<T> o;
   o.xx = txx * R.xx + txy * R.xy + txz * R.xz, o.xy = txx * R.yx + txy * R.yy + txz * R.yz, o.xz = txx * R.zx + txy * R.zy + txz * R.zz;
   o.yx = tyx * R.xx + tyy * R.xy + tyz * R.xz, o.yy = tyx * R.yx + tyy * R.yy + tyz * R.yz, o.yz = tyx * R.zx + tyy * R.zy + tyz * R.zz;
   o.zx = tzx * R.xx + tzy * R.xy + tzz * R.xz, o.zy = tzx * R.yx + tzy * R.yy + tzz * R.yz, o.zz = tzx * R.zx + tzy * R.zy + tzz * R.zz;
   return o;
}
// ---- Rz(c,s) S Rz^T : planar rotation of a symmetric matrix
template <typename T>
MH_DEV S3<T> conj_z(T c, T s, const S3<T> &S)
{
   S3<T> o;
   T cc = c * c, ss = s * s, cs = c * s;
   o.xx = cc * S.xx - T(2) * cs * S.xy + ss * S.yy;
   o.yy = ss * S.xx + T(2) * cs * S.xy + cc * S.yy;
   o.xy = cs * (S.xx - S.yy) + (cc - ss) * S.xy;
   o.xz = c * S.xz - s * S.yz;
   o.yz = s * S.xz + c * S.yz;
   o.zz = S.zz;
   return o;
}
// ---- Rz M Rz^T for a general M
template <typename T>
MH_DEV M3<T> conj_z(T c, T s, const M3<T> &M)
{
   // rows first: T = Rz M
   T txx = c * M.xx - s * M.yx, txy = c * M.xy - s * M.yy, txz = c * M.xz - s * M.yz;
   T tyx = s * M.xx + c * M.yx, tyy = s * M.xy + c * M.yy, tyz = s * M.xz + c * M.yz;
   M3<T> o;
   o.xx = c * txx - s * txy, o.xy = s * txx + c * txy, o.xz = txz;
   o.yx = c * tyx - s * tyy, o.yy = s * tyx + c * tyy, o.yz = tyz;
   o.zx = c * M.zx - s * M.zy, o.zy = s * M.zx + c * M.zy, o.zz = M.zz;
   return o;
}

// ---- translate an articulated inertia by p (frame origin moves so that old origin sits at p in the new frame):
//      with P = [p]x :  C' = C + P L ;  A' = A + P C^T + (P C'^T)^T ;  L' = L
//      (algorithms/ArticulatedBodyInertiaAlorigthmTools.java:32-163 in matrix form)
template <typename T>
MH_DEV void translate(ABI<T> &I, V3<T> p)
{
   const S3<T> &L = I.L;
   const M3<T> C = I.C;
   // PL = P L, rows: (P L)_x* = -pz L_y* + py L_z*, (P L)_y* = pz L_x* - px L_z*, (P L)_z* = -py L_x* + px L_y*
   M3<T> N; // C' = C + P L
   N.xx = C.xx - p.z * L.xy + p.y * L.xz, N.xy = C.xy - p.z * L.yy + p.y * L.yz, N.xz = C.xz - p.z * L.yz + p.y * L.zz;
   N.yx = C.yx + p.z * L.xx - p.x * L.xz, N.yy = C.yy + p.z * L.xy - p.x * L.yz, N.yz = C.yz + p.z * L.xz - p.x * L.zz;
   N.zx = C.zx - p.y * L.xx + p.x * L.xy, N.zy = C.zy - p.y * L.xy + p.x * L.yy, N.zz = C.zz - p.y * L.xz + p.x * L.yz;
   // M1 = P C^T : (M1)_ij = sum_k P_ik C_jk ;  M2 = P N^T ;  A' = A + M1 + M2^T  (symmetric)
   // row x of P = (0, -pz, py), row y = (pz, 0, -px), row z = (-py, px, 0)
   // each entry one flat sum: A_ij + (M1)_ij + (M2)_ji as a single FMA chain
   I.A.xx = I.A.xx - p.z * C.xy + p.y * C.xz - p.z * N.xy + p.y * N.xz;
   I.A.xy = I.A.xy - p.z * C.yy + p.y * C.yz + p.z * N.xx - p.x * N.xz;
   I.A.xz = I.A.xz - p.z * C.zy + p.y * C.zz - p.y * N.xx + p.x * N.xy;
   I.A.yy = I.A.yy + p.z * C.yx - p.x * C.yz + p.z * N.yx - p.x * N.yz;
   I.A.yz = I.A.yz + p.z * C.zx - p.x * C.zz - p.y * N.yx + p.x * N.yy;
   I.A.zz = I.A.zz - p.y * C.zx + p.x * C.zy - p.y * N.zx + p.x * N.zy;
   I.C = N;
}
// ---- slide along z by d (prismatic joint): same as translate(I, {0,0,d})
template <typename T>
MH_DEV void translate_z(ABI<T> &I, T d)
{
   translate(I, V3<T>{T(0), T(0), d});
}
template <typename T>
MH_DEV void rotate(ABI<T> &I, const M3<T> &R)
{
   I.A = conj(R, I.A);
   I.L = conj(R, I.L);
   I.C = conj(R, I.C);
}
template <typename T>
MH_DEV void rotate_z(ABI<T> &I, T c, T s)
{
   I.A = conj_z(c, s, I.A);
   I.L = conj_z(c, s, I.L);
   I.C = conj_z(c, s, I.C);
}
// I * (a, l)
template <typename T>
MH_DEV SV<T> mul(const ABI<T> &I, SV<T> v)
{
   SV<T> o;
   const S3<T> &A = I.A, &L = I.L;
   const M3<T> &C = I.C;
   o.a.x = A.xx * v.a.x + A.xy * v.a.y + A.xz * v.a.z + C.xx * v.l.x + C.xy * v.l.y + C.xz * v.l.z;
   o.a.y = A.xy * v.a.x + A.yy * v.a.y + A.yz * v.a.z + C.yx * v.l.x + C.yy * v.l.y + C.yz * v.l.z;
   o.a.z = A.xz * v.a.x + A.yz * v.a.y + A.zz * v.a.z + C.zx * v.l.x + C.zy * v.l.y + C.zz * v.l.z;
   o.l.x = C.xx * v.a.x + C.yx * v.a.y + C.zx * v.a.z + L.xx * v.l.x + L.xy * v.l.y + L.xz * v.l.z;
   o.l.y = C.xy * v.a.x + C.yy * v.a.y + C.zy * v.a.z + L.xy * v.l.x + L.yy * v.l.y + L.yz * v.l.z;
   o.l.z = C.xz * v.a.x + C.yz * v.a.y + C.zz * v.a.z + L.xz * v.l.x + L.yz * v.l.y + L.zz * v.l.z;
   return o;
}
template <typename T>
MH_DEV ABI<T> abi_from_rigid(const RI<T> &r)
{ // algorithms/ArticulatedBodyInertia.java:176-186 : A = I, L = m 1, C = [h]x
   ABI<T> o;
   o.A = r.I;
   o.L = S3<T>{r.m, T(0), T(0), r.m, T(0), r.m};
   o.C = M3<T>{T(0), -r.h.z, r.h.y, r.h.z, T(0), -r.h.x, -r.h.y, r.h.x, T(0)};
   return o;
}

// ---- rigid inertia child -> parent through (R, p):  m' = m ; h' = R h + m p ; I' = R I R^T + shift
//      shift of the rotational inertia when the origin moves by p (old origin at p in the new frame), with
//      hR = R h :  I' = I_R + ( 2 (p.hR) + m p.p ) 1 - ( p hR^T + hR p^T + m p p^T )
//      (parallel axis, tools/MecanoTools.java:449-547 with c = h/m)
template <typename T>
MH_DEV void shift_origin(RI<T> &r, V3<T> p)
{
   V3<T> h = r.h;
   T d = T(2) * dot(p, h) + r.m * dot(p, p);
   r.I.xx += d - (T(2) * p.x * h.x + r.m * p.x * p.x);
   r.I.yy += d - (T(2) * p.y * h.y + r.m * p.y * p.y);
   r.I.zz += d - (T(2) * p.z * h.z + r.m * p.z * p.z);
   r.I.xy -= p.x * h.y + h.x * p.y + r.m * p.x * p.y;
   r.I.xz -= p.x * h.z + h.x * p.z + r.m * p.x * p.z;
   r.I.yz -= p.y * h.z + h.y * p.z + r.m * p.y * p.z;
   r.h = h + r.m * p;
}
template <typename T>
MH_DEV void add(RI<T> &a, const RI<T> &b)
{
   a.m += b.m;
   a.h = a.h + b.h;
   a.I.xx += b.I.xx, a.I.xy += b.I.xy, a.I.xz += b.I.xz, a.I.yy += b.I.yy, a.I.yz += b.I.yz, a.I.zz += b.I.zz;
}

// ---- factorised body inertia B = v x* I of the Coriolis-matrix recursion (algorithms/FactorizedBodyInertia.java:136-158): a general
//      6x6 [[A, TR], [BL, L]] whose bottom-right block stays skew under sums and rigid transforms, L = [l]x with l = sum m w
template <typename T>
struct FB
{
   M3<T> A, TR, BL;
   V3<T> l;
};
// a b^T - (a.b) 1  =  [b]x [a]x
template <typename T>
MH_DEV M3<T> outer_minus_dot(V3<T> a, V3<T> b)
{
   const T d = dot(a, b);
   return M3<T>{a.x * b.x - d, a.x * b.y, a.x * b.z, a.y * b.x, a.y * b.y - d, a.y * b.z, a.z * b.x, a.z * b.y, a.z * b.z - d};
}
template <typename T>
MH_DEV void add(M3<T> &a, const M3<T> &b)
{
   a.xx += b.xx, a.xy += b.xy, a.xz += b.xz, a.yx += b.yx, a.yy += b.yy, a.yz += b.yz, a.zx += b.zx, a.zy += b.zy, a.zz += b.zz;
}
template <typename T>
MH_DEV void sub(M3<T> &a, const M3<T> &b)
{
   a.xx -= b.xx, a.xy -= b.xy, a.xz -= b.xz, a.yx -= b.yx, a.yy -= b.yy, a.yz -= b.yz, a.zx -= b.zx, a.zy -= b.zy, a.zz -= b.zz;
}
// [p]x M
template <typename T>
MH_DEV M3<T> tilde_mul(V3<T> p, const M3<T> &M)
{
   return M3<T>{p.y * M.zx - p.z * M.yx, p.y * M.zy - p.z * M.yy, p.y * M.zz - p.z * M.yz,
                p.z * M.xx - p.x * M.zx, p.z * M.xy - p.x * M.zy, p.z * M.xz - p.x * M.zz,
                p.x * M.yx - p.y * M.xx, p.x * M.yy - p.y * M.xy, p.x * M.yz - p.y * M.xz};
}
// M [p]x
template <typename T>
MH_DEV M3<T> mul_tilde(const M3<T> &M, V3<T> p)
{
   return M3<T>{M.xy * p.z - M.xz * p.y, M.xz * p.x - M.xx * p.z, M.xx * p.y - M.xy * p.x,
                M.yy * p.z - M.yz * p.y, M.yz * p.x - M.yx * p.z, M.yx * p.y - M.yy * p.x,
                M.zy * p.z - M.zz * p.y, M.zz * p.x - M.zx * p.z, M.zx * p.y - M.zy * p.x};
}
// B = v x* I for the rigid inertia (m, h, I about the frame origin) of a body moving with v = (w, u):
//   A = [w]x I - [u]x [h]x ;  TR = [w]x [h]x + m [u]x ;  BL = -[w]x [h]x ;  L = m [w]x
template <typename T>
MH_DEV FB<T> fb_from_rigid(const RI<T> &r, SV<T> v)
{
   FB<T> B;
   const M3<T> I{r.I.xx, r.I.xy, r.I.xz, r.I.xy, r.I.yy, r.I.yz, r.I.xz, r.I.yz, r.I.zz};
   const M3<T> wh = outer_minus_dot(r.h, v.a); // [w]x [h]x
   B.A = tilde_mul(v.a, I);
   sub(B.A, outer_minus_dot(r.h, v.l));
   B.BL = M3<T>{-wh.xx, -wh.xy, -wh.xz, -wh.yx, -wh.yy, -wh.yz, -wh.zx, -wh.zy, -wh.zz};
   const V3<T> mu = r.m * v.l;
   B.TR = M3<T>{wh.xx, wh.xy - mu.z, wh.xz + mu.y, wh.yx + mu.z, wh.yy, wh.yz - mu.x, wh.zx - mu.y, wh.zy + mu.x, wh.zz};
   B.l = r.m * v.a;
   return B;
}
template <typename T>
MH_DEV void add(FB<T> &a, const FB<T> &b)
{
   add(a.A, b.A), add(a.TR, b.TR), add(a.BL, b.BL);
   a.l = a.l + b.l;
}
template <typename T>
MH_DEV void rotate(FB<T> &B, const M3<T> &R)
{
   B.A = conj(R, B.A), B.TR = conj(R, B.TR), B.BL = conj(R, B.BL);
   B.l = mul(R, B.l);
}
// the four in-place translation updates of FactorizedBodyInertia.java:323-330: TR += [p]x L ; A += [p]x BL ; A -= TR [p]x ; BL -= L [p]x
template <typename T>
MH_DEV void translate(FB<T> &B, V3<T> p)
{
   add(B.TR, outer_minus_dot(B.l, p)); // [p]x [l]x = l p^T - (p.l) 1
   add(B.A, tilde_mul(p, B.BL));
   sub(B.A, mul_tilde(B.TR, p));
   sub(B.BL, outer_minus_dot(p, B.l)); // [l]x [p]x = p l^T - (l.p) 1
}
// B s and B^T s
template <typename T>
MH_DEV SV<T> mul(const FB<T> &B, SV<T> s)
{
   return SV<T>{mul(B.A, s.a) + mul(B.TR, s.l), mul(B.BL, s.a) + cross(B.l, s.l)};
}
template <typename T>
MH_DEV SV<T> tmul(const FB<T> &B, SV<T> s)
{
   return SV<T>{tmul(B.A, s.a) + tmul(B.BL, s.l), tmul(B.TR, s.a) - cross(B.l, s.l)};
}

// ---- wave-uniform read-only data (model constants, index maps): read through the constant address space so that the
//      compiler emits scalar loads (s_load) into SGPRs instead of 64 identical vector loads
template <typename T>
MH_DEV T ldc(const T *p)
{
   typedef const T __attribute__((address_space(4))) * cptr;
   return *(cptr)(unsigned long long)p;
}
typedef const int __attribute__((address_space(4))) *ciptr;
MH_DEV ciptr as_const(const int *p) { return (ciptr)(unsigned long long)p; }
// accessor of one joint's constants: LDS copy (broadcast ds_read) or scalar loads from global memory
template <typename T, bool LDS>
struct CRef
{
   const T *p;
   MH_DEV T operator[](int k) const
   {
      if constexpr (LDS)
         return p[k];
      else
         return ldc(p + k);
   }
};

// ---- sincos.  fp64: Cody-Waite reduction by pi/2 (exact products through FMA) + the classic minimax kernels on
//      [-pi/4, pi/4]; about 30 instructions instead of the ~120 of the library routine with its Payne-Hanek path.
//      Error <= ~2 ulp for |x| < 2^19; larger arguments take the library path (out of line).
//      The out-of-line routine returns (sin, cos) BY VALUE, in registers: with pointer outputs the caller's s / c get stack slots and
//      every call site -- one per body -- carries an unconditional 16-byte scratch store on the hot path (seen in the ISA and as
//      1.4 MB of WRITE_SIZE per 4096-configuration launch on top of the 0.96 MB of results).
struct SinCos
{
   double s, c;
};
__device__ __attribute__((noinline)) SinCos sincos_slow(double x)
{
   SinCos r;
   sincos(x, &r.s, &r.c);
   return r;
}
// The fast path alone: straight-line code, valid for |x| < 2^19 (sincos_in_fast_range).  Several of these in one basic block interleave
// -- six joints of a limb formed together cost little more than one --, which the branch to the slow path inside sincos_t prevents: a
// caller that forms many pairs calls this for all of them, ORs the range tests, and repeats the lot with sincos_t behind ONE branch.
MH_DEV bool sincos_in_fast_range(double x) { return fabs(x) < 524288.0; }
MH_DEV bool sincos_in_fast_range(float) { return true; }
MH_DEV void sincos_fast(double x, double &s, double &c);
MH_DEV void sincos_t(double x, double &s, double &c)
{
   if (__builtin_expect(!sincos_in_fast_range(x), 0))
   {
      const SinCos r = sincos_slow(x);
      s = r.s, c = r.c;
      return;
   }
   sincos_fast(x, s, c);
}
MH_DEV void sincos_fast(double x, double &s, double &c)
{
   const double k = rint(x * 6.36619772367581382433e-01);
   double r = fma(-k, 1.57079632679489655800e+00, x);
   r = fma(-k, 6.12323399573676603587e-17, r);
   r = fma(-k, -1.49738490485916983084e-33, r); // third term: only matters for |k| ~ 2^19
   const double z = r * r;
   double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
   ps = fma(z, ps, 2.75573137070700676789e-06);
   ps = fma(z, ps, -1.98412698298579493134e-04);
   ps = fma(z, ps, 8.33333333332248946124e-03);
   ps = fma(z, ps, -1.66666666666666324348e-01);
   const double sr = fma(z * r, ps, r);
   double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
   pc = fma(z, pc, -2.75573143513906633035e-07);
   pc = fma(z, pc, 2.48015872894767294178e-05);
   pc = fma(z, pc, -1.38888888888741095749e-03);
   pc = fma(z, pc, 4.16666666666666019037e-02);
   const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
   const int n = (int)k;
   const double s1 = (n & 1) ? cr : sr, c1 = (n & 1) ? sr : cr;
   s = (n & 2) ? -s1 : s1;
   c = ((n + 1) & 2) ? -c1 : c1;
}
MH_DEV void sincos_t(float x, float &s, float &c) { sincosf(x, &s, &c); }
MH_DEV void sincos_fast(float x, float &s, float &c) { sincosf(x, &s, &c); }

} // namespace mh
