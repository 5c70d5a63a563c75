// mh_kernels.h -- generic (run-time topology) batched RNEA / ABA / CRBA kernels for gfx950.
//
// Mapping: one wavefront lane = one configuration (q, qd, qdd | tau); the 64 lanes of a wave stride the batch, so
// every workspace access and every SoA state access is one contiguous 512-byte (fp64) line per wave-instruction.
// The kinematic tree (parents, joint kinds, index maps) is wave-uniform and is read through scalar loads; the
// per-joint constants (pose of the joint in its parent, spatial inertia of the successor) are staged once per
// workgroup in LDS and read as broadcasts.  Per-lane intermediates that must survive between the outward and
// the inward sweep of the tree live in a per-lane-strided global workspace (slot-major: ws[slot][lane]).
//
// Replaces, per configuration (paths relative to /root/reference/src/main/java/us/ihmc/mecano/):
//   rnea_kernel  algorithms/InverseDynamicsCalculator.java:873-959 + the frame update it relies on
//   aba_kernel   algorithms/ForwardDynamicsCalculator.java:1085-1310
//   crba_kernel  algorithms/CompositeRigidBodyMassMatrixCalculator.java:588-707,770-798
// in the engine's canonical joint frames (see mh_device.h); outputs are frame-independent.
#pragma once
#include "mh_device.h"

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <type_traits>

namespace mh
{
enum : int
{
   JT_REVOLUTE = 0,
   JT_PRISMATIC = 1,
   JT_SIXDOF = 2,
   JT_FIXED = 3,
   JT_PLANAR = 4,   // q = (pitch, x, z); qd = (w_y, v_x, v_z)          multiBodySystem/interfaces/PlanarJointReadOnly.java:17-72
   JT_SPHERICAL = 5 // q = quaternion (x, y, z, s); qd = angular velocity   multiBodySystem/interfaces/SphericalJointReadOnly.java:18-104
};
// joints whose transform is a general (R, p) pair held in JX::X, and the layout of their DoFs in a spatial vector
__host__ __device__ constexpr bool general_x(int type) { return type == JT_SIXDOF || type == JT_PLANAR || type == JT_SPHERICAL; }
__host__ __device__ constexpr int dof_count(int type) { return type == JT_SIXDOF ? 6 : (type == JT_FIXED ? 0 : (type == JT_PLANAR || type == JT_SPHERICAL ? 3 : 1)); }
__host__ __device__ constexpr int cfg_count(int type) { return type == JT_SIXDOF ? 7 : (type == JT_FIXED ? 0 : (type == JT_PLANAR ? 3 : (type == JT_SPHERICAL ? 4 : 1))); }
// component (0..2 angular x y z, 3..5 linear x y z) of the canonical after-joint frame that DoF k of a joint moves along: the motion
// subspaces are unit vectors (JointReadOnly.java:201-207; planar w_y, v_x, v_z: tools/MecanoTools.java:920-952; spherical: :1002-1043)
__host__ __device__ constexpr int dof_comp(int type, int k)
{
   return type == JT_REVOLUTE ? 2 : (type == JT_PRISMATIC ? 5 : (type == JT_PLANAR ? (k == 0 ? 1 : (k == 1 ? 3 : 5)) : k));
}

// ---- per-joint integer record (wave-uniform, scalar loads)
enum : int
{
   MI_PARENT = 0,   // engine index of the parent joint, -1 = root body
   MI_TYPE = 1,
   MI_DOF = 2,      // offset into dof_map
   MI_CFG = 3,      // offset into cfg_map
   MI_EXT = 4,      // index of the joint in the caller's mh_model_desc order (f_ext rows)
   MI_FLAGS = 5,
   MI_SLOT_JP = 6,  // 2 slots (cos, sin) for revolute joints
   MI_SLOT_F = 7,   // 8 slots: RNEA wrench (6) | ABA bias wrench -> U (6) + Dinv, u
   MI_SLOT_VA = 8,  // 12 slots: velocity + acceleration for children that do not directly follow their parent
   MI_SLOT_C = 9,   // 6 slots: ABA bias acceleration
   MI_SLOT_IA = 10, // 21 slots: ABA articulated inertia / CRBA composite inertia accumulator
   MI_SLOT_LK = 11, // 27 slots (6-DoF joints only): articulated inertia + bias wrench of an ACCELERATION_SOURCE joint
   // depth-first kernels (mh_dfs_kernels.h)
   MI_NCH = 12,     // number of children
   MI_DFS_R = 13,   // offset of the body's frame in the RNEA depth stack (non-leaf bodies)
   MI_DFS_A = 14,   // ... in the ABA depth stack
   MI_HAND = 15,    // offset of the body's record in ABA's inward -> outward hand-over
   MI_PFR_R = 16,   // the PARENT's frame offset in the RNEA stack / its (v, a) slots -- no dependent load of the parent's record
   MI_PVA_R = 17,
   MI_PFR_A = 18,   // the parent's frame offset in the ABA stack, its v slots, its accumulator slots
   MI_PV_A = 19,
   MI_PACC_A = 20,
   MI_ROW_Q = 21,   // 1-DoF joints: the joint's row in configuration matrices / velocity-sized matrices (dof_map / cfg_map resolved)
   MI_ROW_V = 22,
   MI_STRIDE = 24
};
enum : int
{
   MF_PARENT_ADJ = 1, // parent == j - 1: hand values over in registers
   MF_STORE_VA = 2,   // some child c != j + 1 exists: it reloads (v, a) from the workspace
   MF_ACC_FIRST = 4,  // this body is the first (highest index) non-adjacent child of its parent: store, do not add
   MF_HAS_ACC = 8,    // some child c != j + 1 exists: its inertia contribution arrives through the workspace
   MF_LOCKED = 16     // JointSourceMode.ACCELERATION_SOURCE (ForwardDynamicsCalculator.java:45-57): qdd is an input, tau an output
};
// ---- per-joint real constants (LDS)
enum : int
{
   MC_RB = 0,  // 9: rotation of the canonical before-joint frame in the parent's canonical after-joint frame, row-major
   MC_PB = 9,  // 3: its position
   MC_M = 12,  // mass
   MC_H = 13,  // 3: first moment m c about the canonical after-joint origin
   MC_I = 16,  // 6: rotational inertia about that origin (xx, xy, xz, yy, yz, zz)
   MC_RF = 22, // 9: rotation body-fixed -> canonical after-joint (for external wrenches)
   MC_PF = 31, // 3: position of the body-fixed frame
   MC_QA = 34, // 9: rotation canonical after-joint -> Mecano's after-joint frame (joint wrench outputs)
   MC_OA = 43, // 3: origin of the canonical after-joint frame in Mecano's after-joint frame
   MC_STRIDE = 46
};
// Version of the canonical-frame construction of mh_model_create (axis -> +z, first child on the x axis, inertia about the joint origin).
// A topology-specialised code object folds parts of it at compile time (Tree<TP>::p_aligned): it must come from the same convention.
constexpr int MH_FRAME_CONVENTION = 2;

struct DevModel
{
   int n, nq, nv, n_slots;
   const int *meta;    // [n][MI_STRIDE]
   const int *dof_map; // concatenated getJointDoFIndices
   const int *cfg_map; // concatenated getJointConfigurationIndices
   const void *consts; // [n][MC_STRIDE] of T
   // depth-first kernels: event program (2 n words) and per-lane slot counts
   const int *prog;
   int n_events, rnea_stack, aba_stack, aba_hand;
};

template <typename T>
struct Args
{
   DevModel m;
   long B;
   const T *q, *qd, *in3; // in3 = qdd (RNEA) | tau (ABA)
   const T *fext;
   T *out;
   T *ws;
   long ws_stride; // lanes in the grid
   long q_bs, q_es; // batch stride / element stride of configuration matrices
   long v_bs, v_es; // ... of velocity-sized matrices
   long f_bs, f_es; // ... of the external wrench array (element = joint * 6 + component)
   T gx, gy, gz;    // minus the LINEAR part of the root acceleration (the gravity vector when it was given as one)
   T rax, ray, raz; // ANGULAR part of the root acceleration (InverseDynamicsCalculator.setRootAcceleration, java:413-427); 0 with a gravity vector
   int coriolis, accel;
   // second job of a fused RNEA+ABA launch (specialised kernels only): tau in, qdd out
   // aba_kernel<.., LOCKED>: in3b = given accelerations of the ACCELERATION_SOURCE joints, outb = tau of all joints (may be NULL)
   const T *in3b;
   T *outb;
   // optional per-body outputs (generic kernels with BODIES): spatial acceleration / twist of every successor body relative to the
   // inertial frame, in its body-fixed frame, [B][n_joints][6] laid out like fext (f_bs, f_es); either may be NULL
   T *body_acc, *body_twist;
   // optional (rnea_kernel with BODIES): the wrench every joint transmits (moment, force), in Mecano's frame after the joint, laid out like
   // fext -- InverseDynamicsCalculator.getComputedJointWrench (InverseDynamicsCalculator.java:578-585); may be NULL
   T *joint_wrench;
   // fused simulation step (tree-split ABA kernel): when q_next is not NULL the new state after one MultiBodySystemStateIntegrator
   // step of size dt is written as well (q_next [B][nq], qd_next [B][nv]; may alias q / qd)
   T dt;
   T *q_next, *qd_next;
};

// spatial acceleration of the root body (angular, linear), in root-body coordinates: (0, -g) for a gravity vector
// (InverseDynamicsCalculator.java:343-348, ForwardDynamicsCalculator.java:259-264) or what setRootAcceleration was given (:413-427 / :340)
template <class ARGS>
MH_DEV auto root_acceleration(const ARGS &A) -> SV<std::remove_cv_t<std::remove_reference_t<decltype(A.gx)>>>
{
   using T = std::remove_cv_t<std::remove_reference_t<decltype(A.gx)>>;
   return SV<T>{V3<T>{A.rax, A.ray, A.raz}, V3<T>{-A.gx, -A.gy, -A.gz}};
}
template <typename T, class CR>
MH_DEV XF<T> load_xb(const CR &c)
{
   XF<T> X;
   X.R = M3<T>{c[MC_RB + 0], c[MC_RB + 1], c[MC_RB + 2], c[MC_RB + 3], c[MC_RB + 4], c[MC_RB + 5], c[MC_RB + 6], c[MC_RB + 7], c[MC_RB + 8]};
   X.p = V3<T>{c[MC_PB + 0], c[MC_PB + 1], c[MC_PB + 2]};
   return X;
}
template <typename T, class CR>
MH_DEV RI<T> load_inertia(const CR &c)
{
   RI<T> r;
   r.m = c[MC_M];
   r.h = V3<T>{c[MC_H + 0], c[MC_H + 1], c[MC_H + 2]};
   r.I = S3<T>{c[MC_I + 0], c[MC_I + 1], c[MC_I + 2], c[MC_I + 3], c[MC_I + 4], c[MC_I + 5]};
   return r;
}

// joint transform of one lane: only the members of the joint's kind are meaningful
template <typename T>
struct JX
{
   T c, s;  // revolute: cos q, sin q
   T d;     // prismatic: q
   XF<T> X; // sixdof: after-joint -> before-joint
};

// motion vector from the parent's after-joint frame into this joint's after-joint frame (no joint velocity added)
template <typename T>
MH_DEV SV<T> motion_down(int type, const JX<T> &jx, const XF<T> &Xb, SV<T> m)
{
   SV<T> b = motion_to_child(Xb, m);
   if (type == JT_REVOLUTE)
      return SV<T>{rotzT(jx.c, jx.s, b.a), rotzT(jx.c, jx.s, b.l)};
   if (type == JT_PRISMATIC)
      return SV<T>{b.a, V3<T>{b.l.x + b.a.y * jx.d, b.l.y - b.a.x * jx.d, b.l.z}};
   if (general_x(type))
      return motion_to_child(jx.X, b);
   return b;
}
// force vector from this joint's after-joint frame up into the parent's after-joint frame
template <typename T>
MH_DEV SV<T> force_up(int type, const JX<T> &jx, const XF<T> &Xb, SV<T> w)
{
   SV<T> b;
   if (type == JT_REVOLUTE)
      b = SV<T>{rotz(jx.c, jx.s, w.a), rotz(jx.c, jx.s, w.l)};
   else if (type == JT_PRISMATIC)
      b = SV<T>{V3<T>{w.a.x - jx.d * w.l.y, w.a.y + jx.d * w.l.x, w.a.z}, w.l};
   else if (general_x(type))
      b = force_to_parent(jx.X, w);
   else
      b = w;
   return force_to_parent(Xb, b);
}
// R_b Rz(q): orientation of a revolute joint's after-joint frame in the parent's frame (12 flops)
template <typename T>
MH_DEV M3<T> revolute_rotation(const JX<T> &jx, const M3<T> &B)
{
   const T c = jx.c, s = jx.s;
   return M3<T>{c * B.xx + s * B.xy, c * B.xy - s * B.xx, B.xz, c * B.yx + s * B.yy, c * B.yy - s * B.yx, B.yz,
                c * B.zx + s * B.zy, c * B.zy - s * B.zx, B.zz};
}
// articulated inertia and bias wrench of a revolute body handed to the parent: ONE congruence / one rotation with R_b Rz(q) instead
// of a planar one (~48 + 8 instructions) followed by the constant one
template <typename T>
MH_DEV void revolute_up(const JX<T> &jx, const XF<T> &Xb, ABI<T> &I, SV<T> &w)
{
   const XF<T> X{revolute_rotation(jx, Xb.R), Xb.p};
   rotate(I, X.R);
   translate(I, X.p);
   w = force_to_parent(X, w);
}
template <typename T>
MH_DEV void abi_up(int type, const JX<T> &jx, const XF<T> &Xb, ABI<T> &I)
{
   if (type == JT_REVOLUTE)
   {
      rotate(I, revolute_rotation(jx, Xb.R));
      translate(I, Xb.p);
      return;
   }
   else if (type == JT_PRISMATIC)
      translate_z(I, jx.d);
   else if (general_x(type))
   {
      rotate(I, jx.X.R);
      translate(I, jx.X.p);
   }
   rotate(I, Xb.R);
   translate(I, Xb.p);
}
template <typename T>
MH_DEV void rigid_up(int type, const JX<T> &jx, const XF<T> &Xb, RI<T> &r)
{
   if (type == JT_REVOLUTE)
   {
      r.h = rotz(jx.c, jx.s, r.h);
      r.I = conj_z(jx.c, jx.s, r.I);
   }
   else if (type == JT_PRISMATIC)
      shift_origin(r, V3<T>{T(0), T(0), jx.d});
   else if (general_x(type))
   {
      r.h = mul(jx.X.R, r.h);
      r.I = conj(jx.X.R, r.I);
      shift_origin(r, jx.X.p);
   }
   r.h = mul(Xb.R, r.h);
   r.I = conj(Xb.R, r.I);
   shift_origin(r, Xb.p);
}

#define MH_WS(slot) ws[(long)(slot)*ws_stride]

// joint transform from the inputs (first visit of a body in a kernel); stores (cos, sin) of revolute joints
template <typename T>
MH_DEV JX<T> joint_from_q(int type, ciptr cfg_map, int cfg_ofs, const T *qrow, long q_es, T *ws, long ws_stride, int slot_jp, bool store)
{
   JX<T> jx;
   jx.c = T(1), jx.s = T(0), jx.d = T(0);
   if (type == JT_REVOLUTE)
   {
      T qv = qrow[cfg_map[cfg_ofs] * q_es];
      sincos_t(qv, jx.s, jx.c);
      if (store)
      {
         MH_WS(slot_jp) = jx.c;
         MH_WS(slot_jp + 1) = jx.s;
      }
   }
   else if (type == JT_PRISMATIC)
      jx.d = qrow[cfg_map[cfg_ofs] * q_es];
   else if (type == JT_SIXDOF)
   {
      ciptr ci = cfg_map + cfg_ofs;
      jx.X.R = quat_to_R(qrow[ci[0] * q_es], qrow[ci[1] * q_es], qrow[ci[2] * q_es], qrow[ci[3] * q_es]);
      jx.X.p = V3<T>{qrow[ci[4] * q_es], qrow[ci[5] * q_es], qrow[ci[6] * q_es]};
   }
   else if (type == JT_SPHERICAL)
   {
      ciptr ci = cfg_map + cfg_ofs;
      jx.X.R = quat_to_R(qrow[ci[0] * q_es], qrow[ci[1] * q_es], qrow[ci[2] * q_es], qrow[ci[3] * q_es]);
      jx.X.p = V3<T>{T(0), T(0), T(0)};
   }
   else if (type == JT_PLANAR)
   { // rotation about y by the pitch, translation (x, 0, z)
      ciptr ci = cfg_map + cfg_ofs;
      T sp, cp;
      sincos_t(qrow[ci[0] * q_es], sp, cp);
      jx.X.R = M3<T>{cp, T(0), sp, T(0), T(1), T(0), -sp, T(0), cp};
      jx.X.p = V3<T>{qrow[ci[1] * q_es], T(0), qrow[ci[2] * q_es]};
   }
   return jx;
}
// joint transform on a later visit: revolute (cos, sin) come back from the workspace, the rest is re-read from q
template <typename T>
MH_DEV JX<T> joint_again(int type, ciptr cfg_map, int cfg_ofs, const T *qrow, long q_es, const T *ws, long ws_stride, int slot_jp)
{
   JX<T> jx;
   jx.c = T(1), jx.s = T(0), jx.d = T(0);
   if (type == JT_REVOLUTE)
   {
      jx.c = MH_WS(slot_jp);
      jx.s = MH_WS(slot_jp + 1);
   }
   else if (type == JT_PRISMATIC)
      jx.d = qrow[cfg_map[cfg_ofs] * q_es];
   else if (type == JT_SIXDOF)
   {
      ciptr ci = cfg_map + cfg_ofs;
      jx.X.R = quat_to_R(qrow[ci[0] * q_es], qrow[ci[1] * q_es], qrow[ci[2] * q_es], qrow[ci[3] * q_es]);
      jx.X.p = V3<T>{qrow[ci[4] * q_es], qrow[ci[5] * q_es], qrow[ci[6] * q_es]};
   }
   else if (type == JT_SPHERICAL)
   {
      ciptr ci = cfg_map + cfg_ofs;
      jx.X.R = quat_to_R(qrow[ci[0] * q_es], qrow[ci[1] * q_es], qrow[ci[2] * q_es], qrow[ci[3] * q_es]);
      jx.X.p = V3<T>{T(0), T(0), T(0)};
   }
   else if (type == JT_PLANAR)
   { // rotation about y by the pitch, translation (x, 0, z)
      ciptr ci = cfg_map + cfg_ofs;
      T sp, cp;
      sincos_t(qrow[ci[0] * q_es], sp, cp);
      jx.X.R = M3<T>{cp, T(0), sp, T(0), T(1), T(0), -sp, T(0), cp};
      jx.X.p = V3<T>{qrow[ci[1] * q_es], T(0), qrow[ci[2] * q_es]};
   }
   return jx;
}

template <typename T>
MH_DEV void ws_store6(T *ws, long ws_stride, int slot, SV<T> v)
{
   MH_WS(slot + 0) = v.a.x, MH_WS(slot + 1) = v.a.y, MH_WS(slot + 2) = v.a.z;
   MH_WS(slot + 3) = v.l.x, MH_WS(slot + 4) = v.l.y, MH_WS(slot + 5) = v.l.z;
}
template <typename T>
MH_DEV SV<T> ws_load6(const T *ws, long ws_stride, int slot)
{
   SV<T> v;
   v.a = V3<T>{MH_WS(slot + 0), MH_WS(slot + 1), MH_WS(slot + 2)};
   v.l = V3<T>{MH_WS(slot + 3), MH_WS(slot + 4), MH_WS(slot + 5)};
   return v;
}
template <typename T>
MH_DEV void ws_add6(T *ws, long ws_stride, int slot, SV<T> v)
{
   MH_WS(slot + 0) += v.a.x, MH_WS(slot + 1) += v.a.y, MH_WS(slot + 2) += v.a.z;
   MH_WS(slot + 3) += v.l.x, MH_WS(slot + 4) += v.l.y, MH_WS(slot + 5) += v.l.z;
}

// velocity of the joint in its own (canonical) after-joint frame, and the matching slice of another DoF-sized vector
template <typename T>
MH_DEV SV<T> joint_vec(int type, ciptr dof_map, int dof_ofs, const T *row, long es, bool enabled)
{
   SV<T> o{V3<T>{T(0), T(0), T(0)}, V3<T>{T(0), T(0), T(0)}};
   if (!enabled)
      return o;
   if (type == JT_REVOLUTE)
      o.a.z = row[dof_map[dof_ofs] * es];
   else if (type == JT_PRISMATIC)
      o.l.z = row[dof_map[dof_ofs] * es];
   else if (type == JT_SIXDOF)
   {
      ciptr di = dof_map + dof_ofs;
      o.a = V3<T>{row[di[0] * es], row[di[1] * es], row[di[2] * es]};
      o.l = V3<T>{row[di[3] * es], row[di[4] * es], row[di[5] * es]};
   }
   else if (type == JT_SPHERICAL)
   {
      ciptr di = dof_map + dof_ofs;
      o.a = V3<T>{row[di[0] * es], row[di[1] * es], row[di[2] * es]};
   }
   else if (type == JT_PLANAR)
   {
      ciptr di = dof_map + dof_ofs;
      o.a.y = row[di[0] * es], o.l.x = row[di[1] * es], o.l.z = row[di[2] * es];
   }
   return o;
}
// unit motion vector of DoF k of a joint of the given kind, canonical frame
template <typename T>
MH_DEV SV<T> unit_twist(int type, int k)
{
   SV<T> s{V3<T>{T(0), T(0), T(0)}, V3<T>{T(0), T(0), T(0)}};
   if (type == JT_REVOLUTE)
      s.a.z = T(1);
   else if (type == JT_PRISMATIC)
      s.l.z = T(1);
   else
   {
      const int e = dof_comp(type, k);
      s.a.x = e == 0 ? T(1) : T(0), s.a.y = e == 1 ? T(1) : T(0), s.a.z = e == 2 ? T(1) : T(0);
      s.l.x = e == 3 ? T(1) : T(0), s.l.y = e == 4 ? T(1) : T(0), s.l.z = e == 5 ? T(1) : T(0);
   }
   return s;
}
// component k (0..2 angular, 3..5 linear) of a spatial vector; Sᵀ w of a joint = the components dof_comp(type, .) of w
template <typename T>
MH_DEV T comp(SV<T> w, int k)
{
   return k == 0 ? w.a.x : k == 1 ? w.a.y : k == 2 ? w.a.z : k == 3 ? w.l.x : k == 4 ? w.l.y : w.l.z;
}
// the three DoF components of a planar / spherical joint
template <typename T>
MH_DEV V3<T> comp3(int type, SV<T> w)
{
   return type == JT_PLANAR ? V3<T>{w.a.y, w.l.x, w.l.z} : w.a;
}
template <typename T>
MH_DEV SV<T> from_comp3(int type, V3<T> x)
{
   const V3<T> Z{T(0), T(0), T(0)};
   return type == JT_PLANAR ? SV<T>{V3<T>{T(0), x.x, T(0)}, V3<T>{x.y, T(0), x.z}} : SV<T>{x, Z};
}
// inverse of a symmetric positive-definite 3x3 (adjugate / determinant): the joint-space inertia block D of a 3-DoF joint
// (the reference inverts it with UnrolledInverseFromMinor, ForwardDynamicsCalculator.java:1183-1196)
template <typename T>
MH_DEV S3<T> spd3_inverse(const S3<T> &D)
{
   const T cxx = D.yy * D.zz - D.yz * D.yz, cxy = D.xz * D.yz - D.xy * D.zz, cxz = D.xy * D.yz - D.xz * D.yy;
   const T cyy = D.xx * D.zz - D.xz * D.xz, cyz = D.xy * D.xz - D.xx * D.yz, czz = D.xx * D.yy - D.xy * D.xy;
   const T inv = T(1) / (D.xx * cxx + D.xy * cxy + D.xz * cxz);
   return S3<T>{cxx * inv, cxy * inv, cxz * inv, cyy * inv, cyz * inv, czz * inv};
}
// external wrench of the body (body-fixed frame) brought to the canonical after-joint frame
template <typename T, class CR>
MH_DEV SV<T> load_fext(const CR &c, const T *frow, long f_es, int ext)
{
   XF<T> X;
   X.R = M3<T>{c[MC_RF + 0], c[MC_RF + 1], c[MC_RF + 2], c[MC_RF + 3], c[MC_RF + 4], c[MC_RF + 5], c[MC_RF + 6], c[MC_RF + 7], c[MC_RF + 8]};
   X.p = V3<T>{c[MC_PF + 0], c[MC_PF + 1], c[MC_PF + 2]};
   SV<T> w;
   const long e = (long)ext * 6;
   w.a = V3<T>{frow[(e + 0) * f_es], frow[(e + 1) * f_es], frow[(e + 2) * f_es]};
   w.l = V3<T>{frow[(e + 3) * f_es], frow[(e + 4) * f_es], frow[(e + 5) * f_es]};
   return force_to_parent(X, w);
}

// motion vector of a body from the engine's canonical after-joint frame into Mecano's body-fixed frame, written to row `ext`
// (RigidBodyAccelerationProvider: InverseDynamicsCalculator.java:242-250, ForwardDynamicsCalculator.java:170-180)
template <typename T, class CR>
MH_DEV void store_body_motion(const CR &c, T *row, long f_es, int ext, const SV<T> &m)
{
   XF<T> X;
   X.R = M3<T>{c[MC_RF + 0], c[MC_RF + 1], c[MC_RF + 2], c[MC_RF + 3], c[MC_RF + 4], c[MC_RF + 5], c[MC_RF + 6], c[MC_RF + 7], c[MC_RF + 8]};
   X.p = V3<T>{c[MC_PF + 0], c[MC_PF + 1], c[MC_PF + 2]};
   const SV<T> b = motion_to_child(X, m);
   const long e = (long)ext * 6;
   row[(e + 0) * f_es] = b.a.x, row[(e + 1) * f_es] = b.a.y, row[(e + 2) * f_es] = b.a.z;
   row[(e + 3) * f_es] = b.l.x, row[(e + 4) * f_es] = b.l.y, row[(e + 5) * f_es] = b.l.z;
}

// wrench a joint transmits, from the engine's canonical after-joint frame into Mecano's frame after the joint, written to row `ext`
// (InverseDynamicsCalculator.getComputedJointWrench, InverseDynamicsCalculator.java:578-585: jointWrench is left in frameAfterJoint, :947)
template <typename T, class CR>
MH_DEV void store_joint_wrench(const CR &c, T *row, long f_es, int ext, const SV<T> &w)
{
   XF<T> X;
   X.R = M3<T>{c[MC_QA + 0], c[MC_QA + 1], c[MC_QA + 2], c[MC_QA + 3], c[MC_QA + 4], c[MC_QA + 5], c[MC_QA + 6], c[MC_QA + 7], c[MC_QA + 8]};
   X.p = V3<T>{c[MC_OA + 0], c[MC_OA + 1], c[MC_OA + 2]};
   const SV<T> b = force_to_parent(X, w);
   const long e = (long)ext * 6;
   row[(e + 0) * f_es] = b.a.x, row[(e + 1) * f_es] = b.a.y, row[(e + 2) * f_es] = b.a.z;
   row[(e + 3) * f_es] = b.l.x, row[(e + 4) * f_es] = b.l.y, row[(e + 5) * f_es] = b.l.z;
}

template <typename T>
MH_DEV void stage_consts(const DevModel &m, T *lds)
{
   const T *g = (const T *)m.consts;
   for (int i = threadIdx.x; i < m.n * MC_STRIDE; i += blockDim.x)
      lds[i] = g[i];
   __syncthreads();
}

// ============================================================================================ layout staging
// src [rows][cols] -> dst [cols][rows], 64 x 64 tiles through LDS, coalesced on both sides.  The run-time-topology kernels read one
// matrix entry per lane: with AoS matrices ([B][n], lanes n * sizeof(T) bytes apart) every wave-load touches 64 cache lines and the
// lines are fetched again for the next entries (measured 2x the SoA time on the 128-body tree); for big batches of wide matrices the
// host side therefore transposes the inputs into a scratch copy, runs the kernel on SoA strides and transposes the result back.
template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(const T *__restrict__ src, T *__restrict__ dst, long rows, long cols)
{
   __shared__ T tile[64][65];
   const long nbr = (rows + 63) / 64; // 1-D grid: tile (blockIdx.x % nbr, blockIdx.x / nbr)
   const long r0 = ((long)blockIdx.x % nbr) * 64, c0 = ((long)blockIdx.x / nbr) * 64;
   const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
   for (int i = ty; i < 64; i += 4)
      if (r0 + i < rows && c0 + tx < cols)
         tile[i][tx] = src[(r0 + i) * cols + c0 + tx];
   __syncthreads();
   for (int i = ty; i < 64; i += 4)
      if (c0 + i < cols && r0 + tx < rows)
         dst[(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// Wide matrices (the 128-body tree: n = 323 / 362, not a multiple of anything): the AoS side of a 64 x 64 tile is 64 pieces of 256 bytes
// at odd alignments, and the tile kernel reaches 3.4 TB/s (read + write) there.  But R consecutive ROWS of an AoS matrix are one
// contiguous block of R * n entries: with R = 128 / sizeof(T) it starts on a 128-byte boundary whatever n is, so a workgroup moves it with
// aligned 16-byte accesses and the other side sees whole 128-byte lines (R entries of one column).  The block passes through LDS as it
// lies in memory ([R][n], 128 n bytes); the transposition is the LDS gather / scatter.
// Needs B % V == 0 (V = 16 / sizeof(T): every column segment is then 16-byte aligned), 16-byte aligned bases and 128 n bytes of LDS.
template <typename T>
struct RowBlock
{
   static constexpr int V = 16 / (int)sizeof(T), R = 128 / (int)sizeof(T), LPC = R / V; // LPC lanes per column segment (8)
   typedef T VT __attribute__((ext_vector_type(V)));
};
// Round 5: the workgroups are persistent (cus * workgroups-per-CU of them, block b, b + grid, ...) and a block's accesses are requested
// while the block before it is still passing through LDS -- one block per workgroup left the read stream idle during every gather /
// store phase (4.5-4.8 TB/s read + write at 3 workgroups per CU: profiles/r05_c5_pair_traffic_bias_absorbed.txt).  The barriers order LDS traffic only
// (__syncthreads() would drain the requests just issued).
MH_DEV void lds_only_barrier()
{
   asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
constexpr int ROW_BLOCK_MAX_N = 512; // 128 n bytes of LDS <= 64 KB; 16 vectors of 16 bytes per thread and block
// Non-temporal accesses (tools/proto_transpose.hip, profiles/r05_proto_transpose.txt: 131 072 x 323 floats, rows -> columns 5.2 -> 6.5 TB/s
// with nt stores, columns -> rows 4.7 -> 6.3; a float4 copy of the same bytes 5.2-5.8 plain, 5.7-6.3 nt)
template <bool NT, class VT, typename T>
MH_DEV VT row_block_load(const T *p)
{
   if constexpr (NT)
      return __builtin_nontemporal_load((const VT *)p);
   else
      return *(const VT *)p;
}
template <bool NT, class VT, typename T>
MH_DEV void row_block_store(T *p, VT v)
{
   if constexpr (NT)
      __builtin_nontemporal_store(v, (VT *)p);
   else
      *(VT *)p = v;
}
// src [B][n] (AoS) -> dst [n][B] (SoA)
template <typename T, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) rows_to_columns_kernel(const T *__restrict__ src, T *__restrict__ dst, long B, int n, long blocks)
{
   using RB = RowBlock<T>;
   using VT = typename RB::VT;
   constexpr int V = RB::V, R = RB::R, LPC = RB::LPC, NU = ROW_BLOCK_MAX_N * R / (256 * V);
   extern __shared__ double lds_raw[];
   T *const blk = (T *)lds_raw;
   const int rb = (threadIdx.x % LPC) * V, jl = threadIdx.x / LPC;
   VT reg[NU];
   auto request = [&](long b) { // the block as it lies in memory, 16 bytes per lane
      const long r0 = b * R;
      const int len = (int)(B - r0 < R ? B - r0 : R) * n; // rows % V == 0, hence len % V == 0
      const T *const flat = src + r0 * n;
#pragma unroll
      for (int u = 0; u < NU; u++)
         if ((threadIdx.x + 256 * u) * V + V <= len)
            reg[u] = row_block_load<NTL, VT>(flat + (threadIdx.x + 256 * u) * V);
   };
   long b = blockIdx.x;
   if (b < blocks)
      request(b);
   for (; b < blocks; b += gridDim.x)
   {
      const long r0 = b * R;
      const int rows = (int)(B - r0 < R ? B - r0 : R), len = rows * n;
#pragma unroll
      for (int u = 0; u < NU; u++)
         if ((threadIdx.x + 256 * u) * V + V <= len)
            *(VT *)(blk + (threadIdx.x + 256 * u) * V) = reg[u];
      lds_only_barrier();
      if (b + gridDim.x < blocks)
         request(b + gridDim.x);
      if (rb < rows)
         for (int j = jl; j < n; j += 256 / LPC)
         {
            VT w;
#pragma unroll
            for (int k = 0; k < V; k++)
               w[k] = blk[(rb + k) * n + j];
            row_block_store<NTS, VT>(dst + (long)j * B + r0 + rb, w);
         }
      lds_only_barrier(); // the block is free for the next one
   }
}
// src [n][B] (SoA) -> dst [B][n] (AoS)
template <typename T, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) columns_to_rows_kernel(const T *__restrict__ src, T *__restrict__ dst, long B, int n, long blocks)
{
   using RB = RowBlock<T>;
   using VT = typename RB::VT;
   constexpr int V = RB::V, R = RB::R, LPC = RB::LPC, NU = ROW_BLOCK_MAX_N / (256 / LPC);
   extern __shared__ double lds_raw[];
   T *const blk = (T *)lds_raw;
   const int rb = (threadIdx.x % LPC) * V, jl = threadIdx.x / LPC;
   VT reg[NU];
   auto request = [&](long b) { // R entries of every column: LPC lanes take a 128-byte line
      const long r0 = b * R;
      const int rows = (int)(B - r0 < R ? B - r0 : R);
      if (rb < rows)
      {
#pragma unroll
         for (int u = 0; u < NU; u++)
            if (jl + (256 / LPC) * u < n)
               reg[u] = row_block_load<NTL, VT>(src + (long)(jl + (256 / LPC) * u) * B + r0 + rb);
      }
   };
   long b = blockIdx.x;
   if (b < blocks)
      request(b);
   for (; b < blocks; b += gridDim.x)
   {
      const long r0 = b * R;
      const int rows = (int)(B - r0 < R ? B - r0 : R), len = rows * n;
      if (rb < rows)
      {
#pragma unroll
         for (int u = 0; u < NU; u++)
            if (jl + (256 / LPC) * u < n)
            {
#pragma unroll
               for (int k = 0; k < V; k++)
                  blk[(rb + k) * n + jl + (256 / LPC) * u] = reg[u][k];
            }
      }
      lds_only_barrier();
      if (b + gridDim.x < blocks)
         request(b + gridDim.x);
      T *const flat = dst + r0 * n;
      for (int i = threadIdx.x * V; i + V <= len; i += 256 * V)
         row_block_store<NTS, VT>(flat + i, *(const VT *)(blk + i));
      lds_only_barrier();
   }
}
// [B][n] -> [n][B] (to_columns) or [n][B] -> [B][n] on `stream`: row blocks for wide matrices, 64 x 64 tiles otherwise
template <typename T>
inline void transpose_rows(const T *src, T *dst, long B, long n, bool to_columns, hipStream_t stream)
{
   using RB = RowBlock<T>;
   const size_t lds = (size_t)128 * (size_t)n;
   if (B % RB::V == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && lds >= 16 * 1024 && n <= ROW_BLOCK_MAX_N)
   {
      static const long cus = [] {
         int dev = 0, v = 0;
         if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
         return (long)v;
      }();
      const long blocks = (B + RB::R - 1) / RB::R, per_cu = std::max<long>(1, (long)(160 * 1024 / lds));
      const dim3 grid((unsigned)std::min<long>(blocks, cus * per_cu));
      static const int nt = [] { // MH_TRANSPOSE_NT: bit 0 non-temporal stores, bit 1 non-temporal loads (measurements)
         const char *e = getenv("MH_TRANSPOSE_NT");
         return e ? atoi(e) : 3; // (in the pair call of the 128-body tree 3 is ahead of 1 and 0 by about 1 %: profiles/r05_c5_transpose_nt.txt)
      }();
      auto go = [&](auto ntl, auto nts) {
         if (to_columns)
            hipLaunchKernelGGL((rows_to_columns_kernel<T, decltype(ntl)::value, decltype(nts)::value>), grid, dim3(256), lds, stream, src, dst, B, (int)n, blocks);
         else
            hipLaunchKernelGGL((columns_to_rows_kernel<T, decltype(ntl)::value, decltype(nts)::value>), grid, dim3(256), lds, stream, src, dst, B, (int)n, blocks);
      };
      if (nt == 3)
         go(std::true_type{}, std::true_type{});
      else if (nt == 2)
         go(std::true_type{}, std::false_type{});
      else if (nt == 1)
         go(std::false_type{}, std::true_type{});
      else
         go(std::false_type{}, std::false_type{});
      return;
   }
   const dim3 grid((unsigned)(((B + 63) / 64) * ((n + 63) / 64)));
   if (to_columns)
      hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, stream, src, dst, B, n);
   else
      hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, stream, src, dst, n, B);
}

// ============================================================================================ RNEA
template <typename T, bool LDSC, bool BODIES = false>
__global__ void __launch_bounds__(256) rnea_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   if constexpr (LDSC)
   {
      T *C = (T *)lds_raw;
      stage_consts<T>(m, C);
      CB = C;
   }
   const ciptr meta = as_const(m.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map);
   const long lane = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const long nlanes = (long)gridDim.x * blockDim.x;
   // the workspace is a block of [slot][64 lanes] per wave: a slot offset is a scalar shifted by a constant, not a 64-bit multiply by a
   // run-time stride (which was four scalar instructions in front of every one of the kernel's ~350 workspace accesses)
   constexpr long ws_stride = 64;
   T *ws = A.ws + (lane >> 6) * ((long)m.n_slots * 64) + (lane & 63);
   const V3<T> Z{T(0), T(0), T(0)};

   for (long cfg = lane; cfg < A.B; cfg += nlanes)
   {
      const T *qrow = A.q + cfg * A.q_bs;
      const T *qdrow = A.qd + cfg * A.v_bs;
      const T *qddrow = A.in3 + cfg * A.v_bs;
      const T *frow = A.fext ? A.fext + cfg * A.f_bs : nullptr;
      T *trow = A.out + cfg * A.v_bs;

      // ---- outward sweep: velocities, accelerations, Newton-Euler wrench of every body
      SV<T> v_prev{Z, Z}, a_prev{Z, Z};
      for (int j = 0; j < m.n; j++)
      {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type = mi[MI_TYPE], flags = mi[MI_FLAGS];
         const CRef<T, LDSC> c{CB + j * MC_STRIDE};
         SV<T> vp, ap;
         if (parent < 0)
         {
            vp = SV<T>{Z, Z};
            ap = root_acceleration(A); // InverseDynamicsCalculator.java:343-348
         }
         else if (flags & MF_PARENT_ADJ)
         {
            vp = v_prev, ap = a_prev;
         }
         else
         {
            const int sp = meta[parent * MI_STRIDE + MI_SLOT_VA];
            vp = ws_load6(ws, ws_stride, sp);
            ap = ws_load6(ws, ws_stride, sp + 6);
         }
         const XF<T> Xb = load_xb<T>(c);
         const JX<T> jx = joint_from_q<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP], true);
         const SV<T> vJ = joint_vec<T>(type, dof_map, mi[MI_DOF], qdrow, A.v_es, A.coriolis != 0);
         const SV<T> aJ = joint_vec<T>(type, dof_map, mi[MI_DOF], qddrow, A.v_es, A.accel != 0);
         SV<T> v = motion_down(type, jx, Xb, vp) + vJ;
         SV<T> a = motion_down(type, jx, Xb, ap) + aJ + crm(v, vJ);
         if (!A.coriolis)
            v = SV<T>{Z, Z};
         if constexpr (BODIES)
         {
            if (A.body_acc)
               store_body_motion<T>(c, A.body_acc + cfg * A.f_bs, A.f_es, mi[MI_EXT], a);
            if (A.body_twist)
               store_body_motion<T>(c, A.body_twist + cfg * A.f_bs, A.f_es, mi[MI_EXT], v);
         }
         const RI<T> I = load_inertia<T>(c);
         SV<T> f = mul(I, a) + crf(v, mul(I, v));
         if (frow)
            f = f - load_fext<T>(c, frow, A.f_es, mi[MI_EXT]);
         ws_store6(ws, ws_stride, mi[MI_SLOT_F], f);
         if (flags & MF_STORE_VA)
         {
            ws_store6(ws, ws_stride, mi[MI_SLOT_VA], v);
            ws_store6(ws, ws_stride, mi[MI_SLOT_VA] + 6, a);
         }
         v_prev = v, a_prev = a;
      }
      // ---- inward sweep: joint efforts, wrenches handed to the parents
      SV<T> carry{Z, Z};
      bool have_carry = false;
      for (int j = m.n - 1; j >= 0; j--)
      {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type = mi[MI_TYPE], flags = mi[MI_FLAGS];
         const CRef<T, LDSC> c{CB + j * MC_STRIDE};
         SV<T> f = ws_load6(ws, ws_stride, mi[MI_SLOT_F]);
         if (have_carry)
            f = f + carry;
         if constexpr (BODIES)
         {
            if (A.joint_wrench)
               store_joint_wrench<T>(c, A.joint_wrench + cfg * A.f_bs, A.f_es, mi[MI_EXT], f);
         }
         ciptr di = dof_map + mi[MI_DOF];
         if (type == JT_REVOLUTE)
            trow[di[0] * A.v_es] = f.a.z;
         else if (type == JT_PRISMATIC)
            trow[di[0] * A.v_es] = f.l.z;
         else if (type == JT_SIXDOF)
         {
            trow[di[0] * A.v_es] = f.a.x, trow[di[1] * A.v_es] = f.a.y, trow[di[2] * A.v_es] = f.a.z;
            trow[di[3] * A.v_es] = f.l.x, trow[di[4] * A.v_es] = f.l.y, trow[di[5] * A.v_es] = f.l.z;
         }
         else if (type == JT_PLANAR || type == JT_SPHERICAL)
         {
            const V3<T> t3 = comp3(type, f);
            trow[di[0] * A.v_es] = t3.x, trow[di[1] * A.v_es] = t3.y, trow[di[2] * A.v_es] = t3.z;
         }
         have_carry = false;
         if (parent >= 0)
         {
            const XF<T> Xb = load_xb<T>(c);
            const JX<T> jx = joint_again<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP]);
            const SV<T> fp = force_up(type, jx, Xb, f);
            if (flags & MF_PARENT_ADJ)
            {
               carry = fp;
               have_carry = true;
            }
            else
               ws_add6(ws, ws_stride, meta[parent * MI_STRIDE + MI_SLOT_F], fp);
         }
      }
   }
}

// ============================================================================================ ABA
template <typename T>
MH_DEV void ws_store_abi(T *ws, long ws_stride, int s, const ABI<T> &I)
{
   MH_WS(s + 0) = I.A.xx, MH_WS(s + 1) = I.A.xy, MH_WS(s + 2) = I.A.xz, MH_WS(s + 3) = I.A.yy, MH_WS(s + 4) = I.A.yz, MH_WS(s + 5) = I.A.zz;
   MH_WS(s + 6) = I.L.xx, MH_WS(s + 7) = I.L.xy, MH_WS(s + 8) = I.L.xz, MH_WS(s + 9) = I.L.yy, MH_WS(s + 10) = I.L.yz, MH_WS(s + 11) = I.L.zz;
   MH_WS(s + 12) = I.C.xx, MH_WS(s + 13) = I.C.xy, MH_WS(s + 14) = I.C.xz, MH_WS(s + 15) = I.C.yx, MH_WS(s + 16) = I.C.yy, MH_WS(s + 17) = I.C.yz;
   MH_WS(s + 18) = I.C.zx, MH_WS(s + 19) = I.C.zy, MH_WS(s + 20) = I.C.zz;
}
template <typename T>
MH_DEV ABI<T> ws_load_abi(const T *ws, long ws_stride, int s)
{
   ABI<T> I;
   I.A = S3<T>{MH_WS(s + 0), MH_WS(s + 1), MH_WS(s + 2), MH_WS(s + 3), MH_WS(s + 4), MH_WS(s + 5)};
   I.L = S3<T>{MH_WS(s + 6), MH_WS(s + 7), MH_WS(s + 8), MH_WS(s + 9), MH_WS(s + 10), MH_WS(s + 11)};
   I.C = M3<T>{MH_WS(s + 12), MH_WS(s + 13), MH_WS(s + 14), MH_WS(s + 15), MH_WS(s + 16), MH_WS(s + 17), MH_WS(s + 18), MH_WS(s + 19), MH_WS(s + 20)};
   return I;
}
template <typename T>
MH_DEV void add(ABI<T> &a, const ABI<T> &b)
{
   a.A.xx += b.A.xx, a.A.xy += b.A.xy, a.A.xz += b.A.xz, a.A.yy += b.A.yy, a.A.yz += b.A.yz, a.A.zz += b.A.zz;
   a.L.xx += b.L.xx, a.L.xy += b.L.xy, a.L.xz += b.L.xz, a.L.yy += b.L.yy, a.L.yz += b.L.yz, a.L.zz += b.L.zz;
   a.C.xx += b.C.xx, a.C.xy += b.C.xy, a.C.xz += b.C.xz, a.C.yx += b.C.yx, a.C.yy += b.C.yy, a.C.yz += b.C.yz;
   a.C.zx += b.C.zx, a.C.zy += b.C.zy, a.C.zz += b.C.zz;
}
// Ia = IA - U U^T / D for a 1-DoF joint whose U = (ua, ul)
// I -= w u^T symmetrised over the (A, L, C) blocks, for w = U D^-1 column and u = U column of a multi-DoF joint: summed over the
// DoFs the updates are symmetric, each one alone is not -- only the entries the block layout stores are touched
template <typename T>
MH_DEV void rank1_pair_down(ABI<T> &I, const SV<T> &w, const SV<T> &u)
{
   I.A.xx -= w.a.x * u.a.x, I.A.xy -= T(0.5) * (w.a.x * u.a.y + w.a.y * u.a.x), I.A.xz -= T(0.5) * (w.a.x * u.a.z + w.a.z * u.a.x);
   I.A.yy -= w.a.y * u.a.y, I.A.yz -= T(0.5) * (w.a.y * u.a.z + w.a.z * u.a.y), I.A.zz -= w.a.z * u.a.z;
   I.L.xx -= w.l.x * u.l.x, I.L.xy -= T(0.5) * (w.l.x * u.l.y + w.l.y * u.l.x), I.L.xz -= T(0.5) * (w.l.x * u.l.z + w.l.z * u.l.x);
   I.L.yy -= w.l.y * u.l.y, I.L.yz -= T(0.5) * (w.l.y * u.l.z + w.l.z * u.l.y), I.L.zz -= w.l.z * u.l.z;
   // C couples angular rows with linear columns: (U D^-1 U^T)_al = sum_k w_k.a u_k.l^T, symmetrised with the transposed pairing
   I.C.xx -= T(0.5) * (w.a.x * u.l.x + u.a.x * w.l.x), I.C.xy -= T(0.5) * (w.a.x * u.l.y + u.a.x * w.l.y), I.C.xz -= T(0.5) * (w.a.x * u.l.z + u.a.x * w.l.z);
   I.C.yx -= T(0.5) * (w.a.y * u.l.x + u.a.y * w.l.x), I.C.yy -= T(0.5) * (w.a.y * u.l.y + u.a.y * w.l.y), I.C.yz -= T(0.5) * (w.a.y * u.l.z + u.a.y * w.l.z);
   I.C.zx -= T(0.5) * (w.a.z * u.l.x + u.a.z * w.l.x), I.C.zy -= T(0.5) * (w.a.z * u.l.y + u.a.z * w.l.y), I.C.zz -= T(0.5) * (w.a.z * u.l.z + u.a.z * w.l.z);
}
template <typename T>
MH_DEV void rank1_down(ABI<T> &I, V3<T> ua, V3<T> ul, T dinv)
{
   V3<T> sa = dinv * ua, sl = dinv * ul;
   I.A.xx -= sa.x * ua.x, I.A.xy -= sa.x * ua.y, I.A.xz -= sa.x * ua.z, I.A.yy -= sa.y * ua.y, I.A.yz -= sa.y * ua.z, I.A.zz -= sa.z * ua.z;
   I.L.xx -= sl.x * ul.x, I.L.xy -= sl.x * ul.y, I.L.xz -= sl.x * ul.z, I.L.yy -= sl.y * ul.y, I.L.yz -= sl.y * ul.z, I.L.zz -= sl.z * ul.z;
   I.C.xx -= sa.x * ul.x, I.C.xy -= sa.x * ul.y, I.C.xz -= sa.x * ul.z;
   I.C.yx -= sa.y * ul.x, I.C.yy -= sa.y * ul.y, I.C.yz -= sa.y * ul.z;
   I.C.zx -= sa.z * ul.x, I.C.zy -= sa.z * ul.y, I.C.zz -= sa.z * ul.z;
}
// The same for a revolute joint about z (ua = A e_z, ul = C^T e_z, D = A.zz): Ia S = 0, so the z row / column of A and the z row of C
// cancel exactly -- they are set to zero instead of being computed, and (with -fno-signed-zeros -ffinite-math-only) every product with
// them in the congruence that follows folds away: 15 live entries instead of 21.
template <typename T>
MH_DEV void rank1_down_revolute(ABI<T> &I, V3<T> ua, V3<T> ul, T dinv)
{
   const T sx = dinv * ua.x, sy = dinv * ua.y;
   const V3<T> sl = dinv * ul;
   I.A.xx -= sx * ua.x, I.A.xy -= sx * ua.y, I.A.yy -= sy * ua.y;
   I.A.xz = T(0), I.A.yz = T(0), I.A.zz = T(0);
   I.L.xx -= sl.x * ul.x, I.L.xy -= sl.x * ul.y, I.L.xz -= sl.x * ul.z, I.L.yy -= sl.y * ul.y, I.L.yz -= sl.y * ul.z, I.L.zz -= sl.z * ul.z;
   I.C.xx -= sx * ul.x, I.C.xy -= sx * ul.y, I.C.xz -= sx * ul.z;
   I.C.yx -= sy * ul.x, I.C.yy -= sy * ul.y, I.C.yz -= sy * ul.z;
   I.C.zx = T(0), I.C.zy = T(0), I.C.zz = T(0);
}
// solve IA x = b for a symmetric positive definite 6x6 (floating joint: ForwardDynamicsCalculator.java:1195-1196 uses a
// Cholesky inverse); LDL^T without square roots, fully unrolled so that M stays in registers
template <typename T>
MH_DEV SV<T> spd6_solve(const ABI<T> &I, SV<T> b)
{
   T M[6][6];
   M[0][0] = I.A.xx, M[0][1] = I.A.xy, M[0][2] = I.A.xz, M[1][1] = I.A.yy, M[1][2] = I.A.yz, M[2][2] = I.A.zz;
   M[0][3] = I.C.xx, M[0][4] = I.C.xy, M[0][5] = I.C.xz, M[1][3] = I.C.yx, M[1][4] = I.C.yy, M[1][5] = I.C.yz;
   M[2][3] = I.C.zx, M[2][4] = I.C.zy, M[2][5] = I.C.zz;
   M[3][3] = I.L.xx, M[3][4] = I.L.xy, M[3][5] = I.L.xz, M[4][4] = I.L.yy, M[4][5] = I.L.yz, M[5][5] = I.L.zz;
   T x[6] = {b.a.x, b.a.y, b.a.z, b.l.x, b.l.y, b.l.z};
   T dinv[6];
   // upper-triangular LDL^T in place: M[i][j] (i<j) becomes L_ji
#pragma unroll
   for (int k = 0; k < 6; k++)
   {
      dinv[k] = T(1) / M[k][k];
      T l[6];
#pragma unroll
      for (int j = k + 1; j < 6; j++)
      {
         l[j] = M[k][j] * dinv[k];
#pragma unroll
         for (int i = k + 1; i <= j; i++)
            M[i][j] -= M[k][i] * l[j]; // row k still holds the unscaled entries here
         x[j] -= l[j] * x[k];          // forward substitution fused
      }
#pragma unroll
      for (int j = k + 1; j < 6; j++)
         M[k][j] = l[j];
   }
#pragma unroll
   for (int k = 0; k < 6; k++)
      x[k] *= dinv[k];
#pragma unroll
   for (int k = 5; k >= 0; k--)
   {
#pragma unroll
      for (int j = k + 1; j < 6; j++)
         x[k] -= M[k][j] * x[j];
   }
   return SV<T>{V3<T>{x[0], x[1], x[2]}, V3<T>{x[3], x[4], x[5]}};
}

// The same in two steps (bias-split forward dynamics, mh_zv_kernels.h): the factor depends on the configuration only and is formed while
// the bias efforts are still being computed elsewhere; the substitution runs once they have arrived.  f[0..14] = L (row-wise, k < j),
// f[15..20] = 1 / D.
template <typename T>
struct LDL6
{
   T f[21];
};
__host__ __device__ constexpr int ldl6_index(int k, int j) { return k * 5 - k * (k - 1) / 2 + (j - k - 1); } // k < j
template <typename T>
MH_DEV LDL6<T> spd6_factor(const ABI<T> &I)
{
   T M[6][6];
   M[0][0] = I.A.xx, M[0][1] = I.A.xy, M[0][2] = I.A.xz, M[1][1] = I.A.yy, M[1][2] = I.A.yz, M[2][2] = I.A.zz;
   M[0][3] = I.C.xx, M[0][4] = I.C.xy, M[0][5] = I.C.xz, M[1][3] = I.C.yx, M[1][4] = I.C.yy, M[1][5] = I.C.yz;
   M[2][3] = I.C.zx, M[2][4] = I.C.zy, M[2][5] = I.C.zz;
   M[3][3] = I.L.xx, M[3][4] = I.L.xy, M[3][5] = I.L.xz, M[4][4] = I.L.yy, M[4][5] = I.L.yz, M[5][5] = I.L.zz;
   LDL6<T> F;
#pragma unroll
   for (int k = 0; k < 6; k++)
   {
      const T dinv = rcp_fast(M[k][k]);
      F.f[15 + k] = dinv;
#pragma unroll
      for (int j = k + 1; j < 6; j++)
      {
         const T l = M[k][j] * dinv;
#pragma unroll
         for (int i = k + 1; i <= j; i++)
            M[i][j] -= M[k][i] * l;
         F.f[ldl6_index(k, j)] = l;
      }
   }
   return F;
}
template <typename T>
MH_DEV SV<T> spd6_solve(const LDL6<T> &F, SV<T> b)
{
   T x[6] = {b.a.x, b.a.y, b.a.z, b.l.x, b.l.y, b.l.z};
#pragma unroll
   for (int k = 0; k < 6; k++)
#pragma unroll
      for (int j = k + 1; j < 6; j++)
         x[j] -= F.f[ldl6_index(k, j)] * x[k];
#pragma unroll
   for (int k = 0; k < 6; k++)
      x[k] *= F.f[15 + k];
#pragma unroll
   for (int k = 5; k >= 0; k--)
#pragma unroll
      for (int j = k + 1; j < 6; j++)
         x[k] -= F.f[ldl6_index(k, j)] * x[j];
   return SV<T>{V3<T>{x[0], x[1], x[2]}, V3<T>{x[3], x[4], x[5]}};
}

// LOCKED: some joints are ACCELERATION_SOURCE (:1237-1253, 1284-1297, 1315-1363).  Mecano's pass four re-runs a Newton-Euler sweep to
// get the efforts of those joints; here tau = S^T (IA a + pA) is read off the articulated quantities pass two already holds, which is
// the same wrench (the articulated-body equation of the subtree) without a fourth sweep.
#ifndef MH_SWEEP_AHEAD
#define MH_SWEEP_AHEAD 0
#endif
template <typename T, bool LDSC, bool LOCKED = false, bool BODIES = false>
__global__ void __launch_bounds__(256) aba_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   if constexpr (LDSC)
   {
      T *C = (T *)lds_raw;
      stage_consts<T>(m, C);
      CB = C;
   }
   const ciptr meta = as_const(m.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map);
   const long lane = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const long nlanes = (long)gridDim.x * blockDim.x;
   // the workspace is a block of [slot][64 lanes] per wave: a slot offset is a scalar shifted by a constant, not a 64-bit multiply by a
   // run-time stride (which was four scalar instructions in front of every one of the kernel's ~350 workspace accesses)
   constexpr long ws_stride = 64;
   T *ws = A.ws + (lane >> 6) * ((long)m.n_slots * 64) + (lane & 63);
   const V3<T> Z{T(0), T(0), T(0)};

   for (long cfg = lane; cfg < A.B; cfg += nlanes)
   {
      const T *qrow = A.q + cfg * A.q_bs;
      const T *qdrow = A.qd + cfg * A.v_bs;
      const T *taurow = A.in3 + cfg * A.v_bs;
      const T *frow = A.fext ? A.fext + cfg * A.f_bs : nullptr;
      T *orow = A.out + cfg * A.v_bs;

      // The per-lane operands of a body are gathered by pre1 / pre2 / pre3 (1-DoF joints: rows resolved in the body's record, no walk
      // through the index maps); MH_SWEEP_AHEAD = 1 requests them one body ahead.
      // ---- pass one (ForwardDynamicsCalculator.java:1085-1127): velocities, bias wrench p, bias acceleration c
      SV<T> v_prev{Z, Z};
      T nq_ = T(0), nv_ = T(0);
      auto pre1 = [&](int j1) {
         if (j1 < m.n)
         {
            ciptr m1 = meta + j1 * MI_STRIDE;
            const int t1 = m1[MI_TYPE];
            if (t1 == JT_REVOLUTE || t1 == JT_PRISMATIC)
               nq_ = qrow[m1[MI_ROW_Q] * A.q_es], nv_ = qdrow[m1[MI_ROW_V] * A.v_es];
         }
      };
      if (MH_SWEEP_AHEAD)
         pre1(0);
      for (int j = 0; j < m.n; j++)
      {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type_rt = mi[MI_TYPE], flags = mi[MI_FLAGS];
         auto body = [&](auto kind) { // one dispatch on the joint kind per body, straight-line code per kind (mh_dfs_kernels.h)
         const int type = kind;
         const CRef<T, LDSC> c{CB + j * MC_STRIDE};
         if (!MH_SWEEP_AHEAD)
            pre1(j);
         const T q_in = nq_, v_in = nv_;
         if (MH_SWEEP_AHEAD)
            pre1(j + 1);
         SV<T> vp;
         if (parent < 0)
            vp = SV<T>{Z, Z};
         else if (flags & MF_PARENT_ADJ)
            vp = v_prev;
         else
            vp = ws_load6(ws, ws_stride, meta[parent * MI_STRIDE + MI_SLOT_VA]);
         const XF<T> Xb = load_xb<T>(c);
         JX<T> jx;
         SV<T> vJ{Z, Z};
         if (type == JT_REVOLUTE)
         {
            jx.d = T(0);
            sincos_t(q_in, jx.s, jx.c);
            MH_WS(mi[MI_SLOT_JP]) = jx.c, MH_WS(mi[MI_SLOT_JP] + 1) = jx.s;
            vJ.a.z = v_in;
         }
         else if (type == JT_PRISMATIC)
            jx.c = T(1), jx.s = T(0), jx.d = q_in, vJ.l.z = v_in;
         else
         {
            jx = joint_from_q<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP], true);
            vJ = joint_vec<T>(type, dof_map, mi[MI_DOF], qdrow, A.v_es, true);
         }
         const SV<T> v = motion_down(type, jx, Xb, vp) + vJ;
         if constexpr (BODIES)
         {
            if (A.body_twist)
               store_body_motion<T>(c, A.body_twist + cfg * A.f_bs, A.f_es, mi[MI_EXT], v);
         }
         const RI<T> I = load_inertia<T>(c);
         SV<T> p = crf(v, mul(I, v));
         if (frow)
            p = p - load_fext<T>(c, frow, A.f_es, mi[MI_EXT]);
         ws_store6(ws, ws_stride, mi[MI_SLOT_F], p);
         ws_store6(ws, ws_stride, mi[MI_SLOT_C], crm(v, vJ));
         if (flags & MF_STORE_VA)
            ws_store6(ws, ws_stride, mi[MI_SLOT_VA], v);
         v_prev = v;
         }; // body
         switch (type_rt)
         {
            case JT_REVOLUTE: body(std::integral_constant<int, JT_REVOLUTE>{}); break;
            case JT_PRISMATIC: body(std::integral_constant<int, JT_PRISMATIC>{}); break;
            case JT_SIXDOF: body(std::integral_constant<int, JT_SIXDOF>{}); break;
            case JT_PLANAR: body(std::integral_constant<int, JT_PLANAR>{}); break;
            case JT_SPHERICAL: body(std::integral_constant<int, JT_SPHERICAL>{}); break;
            default: body(std::integral_constant<int, JT_FIXED>{}); break;
         }
      }
      // ---- pass two (:1136-1254): articulated inertias and bias wrenches, leaves to root
      ABI<T> Icarry;
      SV<T> pcarry{Z, Z};
      bool have_carry = false;
      // the next body of this sweep is j - 1: its bias wrench and bias acceleration (pass one wrote them; a body in between only adds to
      // the slots of a NON-adjacent parent, never to those of j - 1) and its effort
      SV<T> npA{Z, Z}, ncj{Z, Z};
      T ntau = T(0);
      auto pre2 = [&](int j1) {
         if (j1 >= 0)
         {
            ciptr m1 = meta + j1 * MI_STRIDE;
            const int t1 = m1[MI_TYPE];
            npA = ws_load6(ws, ws_stride, m1[MI_SLOT_F]);
            if (t1 == JT_REVOLUTE || t1 == JT_PRISMATIC)
               ncj = ws_load6(ws, ws_stride, m1[MI_SLOT_C]), ntau = taurow[m1[MI_ROW_V] * A.v_es];
         }
      };
      if (MH_SWEEP_AHEAD)
         pre2(m.n - 1);
      for (int j = m.n - 1; j >= 0; j--)
      {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type_rt = mi[MI_TYPE], flags = mi[MI_FLAGS];
         auto body = [&](auto kind) { // one dispatch on the joint kind per body, straight-line code per kind (mh_dfs_kernels.h)
         const int type = kind;
         const CRef<T, LDSC> c{CB + j * MC_STRIDE};
         ABI<T> IA = abi_from_rigid(load_inertia<T>(c));
         if (!MH_SWEEP_AHEAD)
            pre2(j);
         SV<T> pA = npA;
         const SV<T> cj_in = ncj;
         const T tau_in = ntau;
         if (MH_SWEEP_AHEAD)
            pre2(j - 1);
         if (have_carry)
         {
            add(IA, Icarry);
            pA = pA + pcarry;
         }
         if (flags & MF_HAS_ACC)
            add(IA, ws_load_abi(ws, ws_stride, mi[MI_SLOT_IA]));
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
            {
               ua = V3<T>{IA.A.xz, IA.A.yz, IA.A.zz}, ul = V3<T>{IA.C.zx, IA.C.zy, IA.C.zz};
               D = IA.A.zz, pz = pA.a.z;
            }
            else
            {
               ua = V3<T>{IA.C.xz, IA.C.yz, IA.C.zz}, ul = V3<T>{IA.L.xz, IA.L.yz, IA.L.zz};
               D = IA.L.zz, pz = pA.l.z;
            }
            if (LOCKED && (flags & MF_LOCKED))
            {
               ws_store6(ws, ws_stride, sf, SV<T>{ua, ul});
               MH_WS(sf + 7) = pz;
               if (parent >= 0)
               { // :1237-1253  Ia = IA ; pa = pA + IA (c + S qdd)
                  SV<T> cj = ws_load6(ws, ws_stride, mi[MI_SLOT_C]);
                  const T qg = (A.in3b + cfg * A.v_bs)[di[0] * A.v_es];
                  if (type == JT_REVOLUTE)
                     cj.a.z += qg;
                  else
                     cj.l.z += qg;
                  pa = pA + mul(IA, cj);
               }
            }
            else
            {
            const T dinv = T(1) / D;                          // :1183
            const T u = tau_in - pz;                          // :1200-1215
            ws_store6(ws, ws_stride, sf, SV<T>{ua, ul});
            MH_WS(sf + 6) = dinv;
            MH_WS(sf + 7) = u;
            if (parent >= 0)
            {
               const SV<T> cj = cj_in;
               const T ud = u * dinv;
               if (type == JT_REVOLUTE)
               { // Ia S = 0: structural zeros, and the hand-up in the same block so that the products with them fold (as in the
                 // depth-first and the specialised kernels)
                  rank1_down_revolute(Ia, ua, ul, dinv);       // :1220-1226
                  pa = pA + mul(Ia, cj) + SV<T>{ud * ua, ud * ul}; // :1229-1234
                  JX<T> jx;
                  jx.c = MH_WS(mi[MI_SLOT_JP]), jx.s = MH_WS(mi[MI_SLOT_JP] + 1), jx.d = T(0);
                  revolute_up(jx, load_xb<T>(c), Ia, pa);      // :1156-1166; pa is now expressed in the parent's frame
                  handed_up = true;
               }
               else
               {
                  rank1_down(Ia, ua, ul, dinv);
                  pa = pA + mul(Ia, cj) + SV<T>{ud * ua, ud * ul};
               }
            }
            }
         }
         else if (LOCKED && dof_count(type) >= 3 && (flags & MF_LOCKED))
         {
            ws_store_abi(ws, ws_stride, mi[MI_SLOT_LK], IA);
            ws_store6(ws, ws_stride, mi[MI_SLOT_LK] + 21, pA);
            if (parent >= 0)
            {
               const SV<T> qg = joint_vec<T>(type, dof_map, mi[MI_DOF], A.in3b + cfg * A.v_bs, A.v_es, true); // S qdd_given
               pa = pA + mul(IA, ws_load6(ws, ws_stride, mi[MI_SLOT_C]) + qg);
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
            ws_store6(ws, ws_stride, sl, U0), ws_store6(ws, ws_stride, sl + 6, U1), ws_store6(ws, ws_stride, sl + 12, U2);
            MH_WS(sl + 18) = Di.xx, MH_WS(sl + 19) = Di.xy, MH_WS(sl + 20) = Di.xz, MH_WS(sl + 21) = Di.yy, MH_WS(sl + 22) = Di.yz, MH_WS(sl + 23) = Di.zz;
            MH_WS(sl + 24) = u3.x, MH_WS(sl + 25) = u3.y, MH_WS(sl + 26) = u3.z;
            if (parent >= 0)
            { // Ia = IA - U D^-1 U^T ; pa = pA + Ia c + U D^-1 u   (:1220-1234)
               const SV<T> W0 = Di.xx * U0 + Di.xy * U1 + Di.xz * U2, W1 = Di.xy * U0 + Di.yy * U1 + Di.yz * U2, W2 = Di.xz * U0 + Di.yz * U1 + Di.zz * U2;
               rank1_pair_down(Ia, W0, U0), rank1_pair_down(Ia, W1, U1), rank1_pair_down(Ia, W2, U2);
               const SV<T> cj = ws_load6(ws, ws_stride, mi[MI_SLOT_C]);
               pa = pA + mul(Ia, cj) + u3.x * W0 + u3.y * W1 + u3.z * W2;
            }
         }
         else if (type == JT_SIXDOF)
         {
            // S = 1_6: U = IA, D = IA.  Pass three needs only x = IA^-1 u; for the parent Ia = 0 and pa = pA + u = tau.
            const SV<T> tau{V3<T>{taurow[di[0] * A.v_es], taurow[di[1] * A.v_es], taurow[di[2] * A.v_es]},
                            V3<T>{taurow[di[3] * A.v_es], taurow[di[4] * A.v_es], taurow[di[5] * A.v_es]}};
            const SV<T> x = spd6_solve(IA, tau - pA);
            ws_store6(ws, ws_stride, sf, x);
            if (parent >= 0)
            {
               Ia.A = S3<T>{T(0), T(0), T(0), T(0), T(0), T(0)};
               Ia.L = Ia.A;
               Ia.C = M3<T>{T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
               pa = tau;
            }
         }
         else if (parent >= 0)
         { // fixed joint: the whole articulated body is handed over unchanged (c = 0)
            pa = pA;
         }
         if (parent >= 0)
         {
            SV<T> pp = pa;
            if (!handed_up)
            {
               const XF<T> Xb = load_xb<T>(c);
               const JX<T> jx = joint_again<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP]);
               if (type == JT_REVOLUTE)
                  revolute_up(jx, Xb, Ia, pp);
               else
               {
                  if (type != JT_SIXDOF || (LOCKED && (flags & MF_LOCKED))) // an effort-source floating joint transmits no inertia: Ia = 0 stays 0
                     abi_up(type, jx, Xb, Ia); // :1156-1166
                  pp = force_up(type, jx, Xb, pa);
               }
            }
            if (flags & MF_PARENT_ADJ)
            {
               Icarry = Ia, pcarry = pp, have_carry = true;
            }
            else
            {
               ciptr pmi = meta + parent * MI_STRIDE;
               if (flags & MF_ACC_FIRST)
                  ws_store_abi(ws, ws_stride, pmi[MI_SLOT_IA], Ia);
               else
               {
                  ABI<T> acc = ws_load_abi(ws, ws_stride, pmi[MI_SLOT_IA]);
                  add(acc, Ia);
                  ws_store_abi(ws, ws_stride, pmi[MI_SLOT_IA], acc);
               }
               ws_add6(ws, ws_stride, pmi[MI_SLOT_F], pp);
            }
         }
         }; // body
         switch (type_rt)
         {
            case JT_REVOLUTE: body(std::integral_constant<int, JT_REVOLUTE>{}); break;
            case JT_PRISMATIC: body(std::integral_constant<int, JT_PRISMATIC>{}); break;
            case JT_SIXDOF: body(std::integral_constant<int, JT_SIXDOF>{}); break;
            case JT_PLANAR: body(std::integral_constant<int, JT_PLANAR>{}); break;
            case JT_SPHERICAL: body(std::integral_constant<int, JT_SPHERICAL>{}); break;
            default: body(std::integral_constant<int, JT_FIXED>{}); break;
         }
      }
      // ---- pass three (:1259-1310): joint accelerations, root to leaves
      SV<T> a_prev{Z, Z};
      // next body of this sweep: bias acceleration, and for 1-DoF joints U, (cos, sin) | q -- all final since pass two
      SV<T> ncj3{Z, Z}, nU{Z, Z};
      T nc_ = T(1), ns_ = T(0);
      auto pre3 = [&](int j1) {
         if (j1 < m.n)
         {
            ciptr m1 = meta + j1 * MI_STRIDE;
            const int t1 = m1[MI_TYPE];
            ncj3 = ws_load6(ws, ws_stride, m1[MI_SLOT_C]);
            if (t1 == JT_REVOLUTE)
               nU = ws_load6(ws, ws_stride, m1[MI_SLOT_F]), nc_ = MH_WS(m1[MI_SLOT_JP]), ns_ = MH_WS(m1[MI_SLOT_JP] + 1);
            else if (t1 == JT_PRISMATIC)
               nU = ws_load6(ws, ws_stride, m1[MI_SLOT_F]), nc_ = qrow[m1[MI_ROW_Q] * A.q_es];
         }
      };
      if (MH_SWEEP_AHEAD)
         pre3(0);
      for (int j = 0; j < m.n; j++)
      {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type_rt = mi[MI_TYPE], flags = mi[MI_FLAGS];
         auto body = [&](auto kind) { // one dispatch on the joint kind per body, straight-line code per kind (mh_dfs_kernels.h)
         const int type = kind;
         const CRef<T, LDSC> c{CB + j * MC_STRIDE};
         if (!MH_SWEEP_AHEAD)
            pre3(j);
         const SV<T> cj3 = ncj3, U_in = nU;
         const T c_in = nc_, s_in = ns_;
         if (MH_SWEEP_AHEAD)
            pre3(j + 1);
         SV<T> ap;
         if (parent < 0)
            ap = root_acceleration(A); // :259-264
         else if (flags & MF_PARENT_ADJ)
            ap = a_prev;
         else
            ap = ws_load6(ws, ws_stride, meta[parent * MI_STRIDE + MI_SLOT_VA]);
         const XF<T> Xb = load_xb<T>(c);
         JX<T> jx;
         if (type == JT_REVOLUTE)
            jx.c = c_in, jx.s = s_in, jx.d = T(0);
         else if (type == JT_PRISMATIC)
            jx.c = T(1), jx.s = T(0), jx.d = c_in;
         else
            jx = joint_again<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP]);
         SV<T> a = motion_down(type, jx, Xb, ap) + cj3; // :1270-1273
         const int sf = mi[MI_SLOT_F];
         ciptr di = dof_map + mi[MI_DOF];
         if (type == JT_REVOLUTE || type == JT_PRISMATIC)
         {
            const SV<T> U = U_in;
            T qdd;
            if (LOCKED && (flags & MF_LOCKED))
               qdd = (A.in3b + cfg * A.v_bs)[di[0] * A.v_es]; // :1284-1297
            else
            {
               const T dinv = MH_WS(sf + 6), u = MH_WS(sf + 7);
               qdd = dinv * (u - (dot(U.a, a.a) + dot(U.l, a.l))); // :1280-1282
            }
            orow[di[0] * A.v_es] = qdd;
            if (type == JT_REVOLUTE)
               a.a.z += qdd;
            else
               a.l.z += qdd;
            if (LOCKED && A.outb)
            { // effort of the joint: S^T (IA a + pA) for a locked joint (cf. :1315-1363), the input otherwise
               T *trow = A.outb + cfg * A.v_bs;
               trow[di[0] * A.v_es] = (flags & MF_LOCKED) ? dot(U.a, a.a) + dot(U.l, a.l) + MH_WS(sf + 7) : taurow[di[0] * A.v_es];
            }
         }
         else if (LOCKED && dof_count(type) >= 3 && (flags & MF_LOCKED))
         {
            const T *gr = A.in3b + cfg * A.v_bs;
            a = a + joint_vec<T>(type, dof_map, mi[MI_DOF], gr, A.v_es, true);
            for (int k = 0; k < dof_count(type); k++)
               orow[di[k] * A.v_es] = gr[di[k] * A.v_es];
            if (A.outb)
            {
               const SV<T> w = mul(ws_load_abi(ws, ws_stride, mi[MI_SLOT_LK]), a) + ws_load6(ws, ws_stride, mi[MI_SLOT_LK] + 21);
               T *trow = A.outb + cfg * A.v_bs;
               for (int k = 0; k < dof_count(type); k++)
                  trow[di[k] * A.v_es] = comp(w, dof_comp(type, k));
            }
         }
         else if (type == JT_PLANAR || type == JT_SPHERICAL)
         { // qdd = D^-1 (u - U^T a')   (:1280-1282)
            const int sl = mi[MI_SLOT_LK];
            const SV<T> U0 = ws_load6(ws, ws_stride, sl), U1 = ws_load6(ws, ws_stride, sl + 6), U2 = ws_load6(ws, ws_stride, sl + 12);
            const S3<T> Di{MH_WS(sl + 18), MH_WS(sl + 19), MH_WS(sl + 20), MH_WS(sl + 21), MH_WS(sl + 22), MH_WS(sl + 23)};
            const V3<T> r{MH_WS(sl + 24) - (dot(U0.a, a.a) + dot(U0.l, a.l)), MH_WS(sl + 25) - (dot(U1.a, a.a) + dot(U1.l, a.l)),
                          MH_WS(sl + 26) - (dot(U2.a, a.a) + dot(U2.l, a.l))};
            const V3<T> qdd = mul(Di, r);
            orow[di[0] * A.v_es] = qdd.x, orow[di[1] * A.v_es] = qdd.y, orow[di[2] * A.v_es] = qdd.z;
            a = a + from_comp3(type, qdd);
            if (LOCKED && A.outb)
            {
               T *trow = A.outb + cfg * A.v_bs;
               for (int k = 0; k < 3; k++)
                  trow[di[k] * A.v_es] = taurow[di[k] * A.v_es];
            }
         }
         else if (type == JT_SIXDOF)
         {
            const SV<T> x = ws_load6(ws, ws_stride, sf);
            const SV<T> qdd = x - a;
            orow[di[0] * A.v_es] = qdd.a.x, orow[di[1] * A.v_es] = qdd.a.y, orow[di[2] * A.v_es] = qdd.a.z;
            orow[di[3] * A.v_es] = qdd.l.x, orow[di[4] * A.v_es] = qdd.l.y, orow[di[5] * A.v_es] = qdd.l.z;
            a = x;
            if (LOCKED && A.outb)
            {
               T *trow = A.outb + cfg * A.v_bs;
               for (int k = 0; k < 6; k++)
                  trow[di[k] * A.v_es] = taurow[di[k] * A.v_es];
            }
         }
         if (flags & MF_STORE_VA)
            ws_store6(ws, ws_stride, mi[MI_SLOT_VA], a);
         if constexpr (BODIES)
         {
            if (A.body_acc)
               store_body_motion<T>(c, A.body_acc + cfg * A.f_bs, A.f_es, mi[MI_EXT], a);
         }
         a_prev = a;
         }; // body
         switch (type_rt)
         {
            case JT_REVOLUTE: body(std::integral_constant<int, JT_REVOLUTE>{}); break;
            case JT_PRISMATIC: body(std::integral_constant<int, JT_PRISMATIC>{}); break;
            case JT_SIXDOF: body(std::integral_constant<int, JT_SIXDOF>{}); break;
            case JT_PLANAR: body(std::integral_constant<int, JT_PLANAR>{}); break;
            case JT_SPHERICAL: body(std::integral_constant<int, JT_SPHERICAL>{}); break;
            default: body(std::integral_constant<int, JT_FIXED>{}); break;
         }
      }
   }
}

// ============================================================================================ state integration
// tools/MultiBodySystemStateIntegrator.java:365-733: one explicit constant-acceleration step of every
// joint state.  Pure streaming (reads q, qd, qdd once, writes q', qd' [, qdd'] once), no workspace.
// The arithmetic is in the joints' own (Mecano) frames: nothing here depends on the engine's canonical frames.
template <typename T>
struct IntArgs
{
   DevModel m;
   long B;
   T dt;
   const T *q, *qd, *qdd;
   T *q_out, *qd_out, *qdd_out; // qdd_out may be NULL
   long q_bs, q_es, v_bs, v_es;
};
// 6-DoF joint, MultiBodySystemStateIntegrator.java:503-575 on plain values: (quat x y z s, p) pose, (w, v) twist, (al, a) acceleration
// in the frame after the joint.  Returns the new pose and twist (and the re-expressed acceleration when an != nullptr).
template <typename T>
MH_DEV void integrate_sixdof(T dt, T hdd, T &qx, T &qy, T &qz, T &qs, V3<T> &p, V3<T> &w, V3<T> &v, const V3<T> &al, const V3<T> &a, V3<T> *an)
{
   const V3<T> a_o = a + cross(w, v); // linear acceleration at the body origin (SpatialAccelerationReadOnly.java:197-204)
   const V3<T> rv = dt * w + hdd * al;
   const V3<T> wn = w + dt * al;
   const V3<T> dp = dt * v + hdd * a_o;
   const T th = sqrt(dot(rv, rv));
   T dx = T(0), dy = T(0), dz = T(0), ds = T(1);
   if (th >= T(1.0e-12))
   {
      T sh, ch;
      sincos_t(T(0.5) * th, sh, ch);
      const T sc = sh / th;
      dx = rv.x * sc, dy = rv.y * sc, dz = rv.z * sc, ds = ch;
   }
   const M3<T> R0 = quat_to_R(qx, qy, qz, qs), Rd = quat_to_R(dx, dy, dz, ds);
   p = p + mul(R0, dp);
   v = tmul(Rd, v + dt * a_o);
   w = wn;
   const T nx = qs * dx + qx * ds + qy * dz - qz * dy; // q' = q * dq (Hamilton product)
   const T ny = qs * dy - qx * dz + qy * ds + qz * dx;
   const T nz = qs * dz + qx * dy - qy * dx + qz * ds;
   const T ns = qs * ds - qx * dx - qy * dy - qz * dz;
   qx = nx, qy = ny, qz = nz, qs = ns;
   if (an)
      *an = tmul(Rd, a_o) + cross(v, w); // :561-562, FixedFrameSpatialAccelerationBasics.java:81-90
}
// One joint of one configuration.  ci / di: the joint's entries of the configuration / DoF index maps.
template <typename T, class IP>
MH_DEV void integrate_joint(int type, IP ci, IP di, const T *qr, const T *vr, const T *ar, T *qo, T *vo, T *ao, const IntArgs<T> &A, T dt, T hdd)
{
   if (type == JT_REVOLUTE || type == JT_PRISMATIC)
   { // :433-441, 710-733
      const T q0 = qr[ci[0] * A.q_es], v0 = vr[di[0] * A.v_es], a0 = ar[di[0] * A.v_es];
      qo[ci[0] * A.q_es] = hdd * a0 + dt * v0 + q0;
      vo[di[0] * A.v_es] = dt * a0 + v0;
      if (ao)
         ao[di[0] * A.v_es] = a0;
   }
   else if (type == JT_SPHERICAL)
   { // :445-449, 578-625: q' = q * quat(dt w + dt^2/2 al), w' = w + dt al
      const T qx = qr[ci[0] * A.q_es], qy = qr[ci[1] * A.q_es], qz = qr[ci[2] * A.q_es], qs = qr[ci[3] * A.q_es];
      const V3<T> w{vr[di[0] * A.v_es], vr[di[1] * A.v_es], vr[di[2] * A.v_es]}, al{ar[di[0] * A.v_es], ar[di[1] * A.v_es], ar[di[2] * A.v_es]};
      const V3<T> rv = dt * w + hdd * al, wn = w + dt * al;
      const T th = sqrt(dot(rv, rv));
      T dx = T(0), dy = T(0), dz = T(0), ds = T(1);
      if (th >= T(1.0e-12))
      {
         T sh, ch;
         sincos_t(T(0.5) * th, sh, ch);
         const T sc = sh / th;
         dx = rv.x * sc, dy = rv.y * sc, dz = rv.z * sc, ds = ch;
      }
      qo[ci[0] * A.q_es] = qs * dx + qx * ds + qy * dz - qz * dy;
      qo[ci[1] * A.q_es] = qs * dy - qx * dz + qy * ds + qz * dx;
      qo[ci[2] * A.q_es] = qs * dz + qx * dy - qy * dx + qz * ds;
      qo[ci[3] * A.q_es] = qs * ds - qx * dx - qy * dy - qz * dz;
      vo[di[0] * A.v_es] = wn.x, vo[di[1] * A.v_es] = wn.y, vo[di[2] * A.v_es] = wn.z;
      if (ao)
         ao[di[0] * A.v_es] = al.x, ao[di[1] * A.v_es] = al.y, ao[di[2] * A.v_es] = al.z;
   }
   else if (type == JT_PLANAR)
   { // the 6-DoF scheme (:503-575) confined to the XZ plane: rotation vector (0, th, 0), a_o = a + w x v
      const T pitch = qr[ci[0] * A.q_es], px = qr[ci[1] * A.q_es], pz = qr[ci[2] * A.q_es];
      const T wy = vr[di[0] * A.v_es], vx = vr[di[1] * A.v_es], vz = vr[di[2] * A.v_es];
      const T aly = ar[di[0] * A.v_es], ax = ar[di[1] * A.v_es], az = ar[di[2] * A.v_es];
      const T aox = ax + wy * vz, aoz = az - wy * vx;
      const T th = dt * wy + hdd * aly;
      const T dpx = dt * vx + hdd * aox, dpz = dt * vz + hdd * aoz;
      T s0, c0, sd, cd;
      sincos_t(pitch, s0, c0);
      sincos_t(th, sd, cd);
      const T wn = wy + dt * aly, cx = vx + dt * aox, cz = vz + dt * aoz;
      const T vnx = cd * cx - sd * cz, vnz = sd * cx + cd * cz;
      qo[ci[0] * A.q_es] = pitch + th;
      qo[ci[1] * A.q_es] = px + c0 * dpx + s0 * dpz;
      qo[ci[2] * A.q_es] = pz - s0 * dpx + c0 * dpz;
      vo[di[0] * A.v_es] = wn, vo[di[1] * A.v_es] = vnx, vo[di[2] * A.v_es] = vnz;
      if (ao)
      {
         ao[di[0] * A.v_es] = aly;
         ao[di[1] * A.v_es] = cd * aox - sd * aoz - vnz * wn;
         ao[di[2] * A.v_es] = sd * aox + cd * aoz + vnx * wn;
      }
   }
   else if (type == JT_SIXDOF)
   { // :503-575
      T qx = qr[ci[0] * A.q_es], qy = qr[ci[1] * A.q_es], qz = qr[ci[2] * A.q_es], qs = qr[ci[3] * A.q_es];
      V3<T> p{qr[ci[4] * A.q_es], qr[ci[5] * A.q_es], qr[ci[6] * A.q_es]};
      V3<T> w{vr[di[0] * A.v_es], vr[di[1] * A.v_es], vr[di[2] * A.v_es]}, v{vr[di[3] * A.v_es], vr[di[4] * A.v_es], vr[di[5] * A.v_es]};
      const V3<T> al{ar[di[0] * A.v_es], ar[di[1] * A.v_es], ar[di[2] * A.v_es]}, a{ar[di[3] * A.v_es], ar[di[4] * A.v_es], ar[di[5] * A.v_es]};
      V3<T> an;
      integrate_sixdof<T>(dt, hdd, qx, qy, qz, qs, p, w, v, al, a, ao ? &an : nullptr);
      qo[ci[0] * A.q_es] = qx, qo[ci[1] * A.q_es] = qy, qo[ci[2] * A.q_es] = qz, qo[ci[3] * A.q_es] = qs;
      qo[ci[4] * A.q_es] = p.x, qo[ci[5] * A.q_es] = p.y, qo[ci[6] * A.q_es] = p.z;
      vo[di[0] * A.v_es] = w.x, vo[di[1] * A.v_es] = w.y, vo[di[2] * A.v_es] = w.z;
      vo[di[3] * A.v_es] = v.x, vo[di[4] * A.v_es] = v.y, vo[di[5] * A.v_es] = v.z;
      if (ao)
      {
         ao[di[0] * A.v_es] = al.x, ao[di[1] * A.v_es] = al.y, ao[di[2] * A.v_es] = al.z;
         ao[di[3] * A.v_es] = an.x, ao[di[4] * A.v_es] = an.y, ao[di[5] * A.v_es] = an.z;
      }
   }
}
// SoA matrices: lane = configuration (consecutive lanes read consecutive addresses of every matrix row), joints looped.
template <typename T>
__global__ void __launch_bounds__(256) integrate_soa_kernel(IntArgs<T> A)
{
   const DevModel &m = A.m;
   const ciptr meta = as_const(m.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map);
   const long nlanes = (long)gridDim.x * blockDim.x;
   const T dt = A.dt, hdd = T(0.5) * A.dt * A.dt;
   for (long cfg = (long)blockIdx.x * blockDim.x + threadIdx.x; cfg < A.B; cfg += nlanes)
   {
      const T *qr = A.q + cfg * A.q_bs, *vr = A.qd + cfg * A.v_bs, *ar = A.qdd + cfg * A.v_bs;
      T *qo = A.q_out + cfg * A.q_bs, *vo = A.qd_out + cfg * A.v_bs;
      T *ao = A.qdd_out ? A.qdd_out + cfg * A.v_bs : nullptr;
      for (int j = 0; j < m.n; j++)
      {
         ciptr mi = meta + j * MI_STRIDE;
         integrate_joint<T, ciptr>(mi[MI_TYPE], cfg_map + mi[MI_CFG], dof_map + mi[MI_DOF], qr, vr, ar, qo, vo, ao, A, dt, hdd);
      }
   }
}
// AoS matrices: a workgroup takes tiles of `tile` configurations and sweeps them twice.  Pass A: thread = (configuration of the
// tile, 1-DoF joint), joint running fastest, so that consecutive threads touch consecutive entries of a row and no lane idles behind
// a 6-DoF neighbour.  Pass B: thread = (configuration, multi-DoF joint).  The two joint lists are built once per workgroup in LDS.
template <typename T>
__global__ void __launch_bounds__(256) integrate_aos_kernel(IntArgs<T> A, int tile)
{
   extern __shared__ int lds_meta[]; // [n] q index | [n] v index of the 1-DoF joints ; [n] engine index of the others ; counts
   const DevModel &m = A.m;
   int *l1_q = lds_meta, *l1_v = lds_meta + m.n, *lm = lds_meta + 2 * m.n, *cnt = lds_meta + 3 * m.n;
   if (threadIdx.x == 0)
   {
      int n1 = 0, nm = 0;
      for (int j = 0; j < m.n; j++)
      {
         const int type = m.meta[j * MI_STRIDE + MI_TYPE];
         if (type == JT_REVOLUTE || type == JT_PRISMATIC)
         {
            l1_q[n1] = m.cfg_map[m.meta[j * MI_STRIDE + MI_CFG]];
            l1_v[n1] = m.dof_map[m.meta[j * MI_STRIDE + MI_DOF]];
            n1++;
         }
         else if (type != JT_FIXED)
            lm[nm++] = j;
      }
      cnt[0] = n1, cnt[1] = nm;
   }
   __syncthreads();
   const unsigned n1 = (unsigned)cnt[0], nm = (unsigned)cnt[1];
   const T dt = A.dt, hdd = T(0.5) * A.dt * A.dt;
   const long ntiles = (A.B + tile - 1) / tile;
   for (long t = blockIdx.x; t < ntiles; t += gridDim.x)
   {
      const long cfg0 = t * tile;
      const unsigned rows = (unsigned)(A.B - cfg0 < (long)tile ? A.B - cfg0 : (long)tile);
      for (unsigned u = threadIdx.x; u < rows * n1; u += blockDim.x)
      { // :433-441, 710-733
         const unsigned lc = u / n1, k = u - lc * n1;
         const long cfg = cfg0 + lc;
         const long qi = cfg * A.q_bs + l1_q[k], vi = cfg * A.v_bs + l1_v[k];
         const T q0 = A.q[qi], v0 = A.qd[vi], a0 = A.qdd[vi];
         A.q_out[qi] = hdd * a0 + dt * v0 + q0;
         A.qd_out[vi] = dt * a0 + v0;
         if (A.qdd_out)
            A.qdd_out[vi] = a0;
      }
      for (unsigned u = threadIdx.x; u < rows * nm; u += blockDim.x)
      {
         const unsigned lc = u / nm, j = (unsigned)lm[u - lc * nm];
         const long cfg = cfg0 + lc;
         const int *mi = m.meta + j * MI_STRIDE;
         integrate_joint<T, const int *>(mi[MI_TYPE], m.cfg_map + mi[MI_CFG], m.dof_map + mi[MI_DOF], A.q + cfg * A.q_bs, A.qd + cfg * A.v_bs,
                                         A.qdd + cfg * A.v_bs, A.q_out + cfg * A.q_bs, A.qd_out + cfg * A.v_bs,
                                         A.qdd_out ? A.qdd_out + cfg * A.v_bs : nullptr, A, dt, hdd);
      }
   }
}

// ============================================================================================ CRBA
template <typename T>
MH_DEV void ws_store_ri(T *ws, long ws_stride, int s, const RI<T> &r)
{
   MH_WS(s + 0) = r.m, MH_WS(s + 1) = r.h.x, MH_WS(s + 2) = r.h.y, MH_WS(s + 3) = r.h.z;
   MH_WS(s + 4) = r.I.xx, MH_WS(s + 5) = r.I.xy, MH_WS(s + 6) = r.I.xz, MH_WS(s + 7) = r.I.yy, MH_WS(s + 8) = r.I.yz, MH_WS(s + 9) = r.I.zz;
}
template <typename T>
MH_DEV RI<T> ws_load_ri(const T *ws, long ws_stride, int s)
{
   RI<T> r;
   r.m = MH_WS(s + 0);
   r.h = V3<T>{MH_WS(s + 1), MH_WS(s + 2), MH_WS(s + 3)};
   r.I = S3<T>{MH_WS(s + 4), MH_WS(s + 5), MH_WS(s + 6), MH_WS(s + 7), MH_WS(s + 8), MH_WS(s + 9)};
   return r;
}
// H is [B][nv][nv] row-major (h_bs = nv*nv, element (r,c) at r*nv + c) and must be zero-filled before the launch:
// the kernel writes only the entries of related joints (CompositeRigidBodyMassMatrixCalculator.java:298,841-845).
template <typename T, bool LDSC>
__global__ void __launch_bounds__(256) crba_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   if constexpr (LDSC)
   {
      T *C = (T *)lds_raw;
      stage_consts<T>(m, C);
      CB = C;
   }
   const ciptr meta = as_const(m.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map);
   const long lane = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const long nlanes = (long)gridDim.x * blockDim.x;
   // the workspace is a block of [slot][64 lanes] per wave: a slot offset is a scalar shifted by a constant, not a 64-bit multiply by a
   // run-time stride (which was four scalar instructions in front of every one of the kernel's ~350 workspace accesses)
   constexpr long ws_stride = 64;
   // gridDim.y waves may share a group of 64 configurations (small batches): each runs the sweeps that build the per-body state and
   // takes the columns of every gridDim.y-th body -- the columns of different bodies are independent
   const int part = blockIdx.y, parts = gridDim.y;
   T *ws = A.ws + ((long)part * gridDim.x * (blockDim.x >> 6) + (lane >> 6)) * ((long)m.n_slots * 64) + (lane & 63);
   const int nv = m.nv;

   for (long cfg = lane; cfg < A.B; cfg += nlanes)
   {
      const T *qrow = A.q + cfg * A.q_bs;
      T *H = A.out + cfg * A.v_bs; // v_bs / v_es carry the per-configuration / per-entry strides of H here
      const long h_es = A.v_es;
      for (int j = 0; j < m.n; j++)
      {
         ciptr mi = meta + j * MI_STRIDE;
         (void)joint_from_q<T>(mi[MI_TYPE], cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP], true);
      }
      RI<T> rcarry;
      bool have_carry = false;
      for (int j = m.n - 1; j >= 0; j--)
      {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type = mi[MI_TYPE], flags = mi[MI_FLAGS];
         const CRef<T, LDSC> c{CB + j * MC_STRIDE};
         RI<T> Ic = load_inertia<T>(c);
         if (have_carry)
            add(Ic, rcarry);
         if (flags & MF_HAS_ACC)
            add(Ic, ws_load_ri(ws, ws_stride, mi[MI_SLOT_IA]));
         have_carry = false;
         const int nd = dof_count(type);
         ciptr dj = dof_map + mi[MI_DOF];
         const XF<T> Xb = load_xb<T>(c);
         const JX<T> jx = joint_again<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP]);
         for (int k = 0; k < (j % parts == part ? nd : 0); k++)
         {
            SV<T> F = mul(Ic, unit_twist<T>(type, k)); // :663-667
            const int col = dj[k];
            // diagonal block (:700-707)
            if (type == JT_REVOLUTE)
               H[((long)col * nv + col) * h_es] = F.a.z;
            else if (type == JT_PRISMATIC)
               H[((long)col * nv + col) * h_es] = F.l.z;
            else
               for (int r = 0; r < nd; r++)
                  H[((long)dj[r] * nv + col) * h_es] = comp(F, dof_comp(type, r));
            // ancestors (:783-792)
            int prev = j, anc = parent;
            XF<T> Xp = Xb;
            JX<T> jp = jx;
            int tp = type;
            while (anc >= 0)
            {
               F = force_up(tp, jp, Xp, F);
               ciptr ma = meta + anc * MI_STRIDE;
               const int ta = ma[MI_TYPE];
               ciptr da = dof_map + ma[MI_DOF];
               if (ta == JT_REVOLUTE)
               {
                  H[((long)da[0] * nv + col) * h_es] = F.a.z;
                  H[((long)col * nv + da[0]) * h_es] = F.a.z;
               }
               else if (ta == JT_PRISMATIC)
               {
                  H[((long)da[0] * nv + col) * h_es] = F.l.z;
                  H[((long)col * nv + da[0]) * h_es] = F.l.z;
               }
               else
                  for (int r = 0; r < dof_count(ta); r++)
                  {
                     const T hv = comp(F, dof_comp(ta, r));
                     H[((long)da[r] * nv + col) * h_es] = hv;
                     H[((long)col * nv + da[r]) * h_es] = hv;
                  }
               prev = anc;
               anc = ma[MI_PARENT];
               if (anc >= 0)
               {
                  Xp = load_xb<T>(CRef<T, LDSC>{CB + prev * MC_STRIDE});
                  jp = joint_again<T>(ta, cfg_map, ma[MI_CFG], qrow, A.q_es, ws, ws_stride, ma[MI_SLOT_JP]);
                  tp = ta;
               }
            }
         }
         if (parent >= 0)
         {
            rigid_up(type, jx, Xb, Ic); // :651-661
            if (flags & MF_PARENT_ADJ)
            {
               rcarry = Ic, have_carry = true;
            }
            else
            {
               const int sp = meta[parent * MI_STRIDE + MI_SLOT_IA];
               if (flags & MF_ACC_FIRST)
                  ws_store_ri(ws, ws_stride, sp, Ic);
               else
               {
                  RI<T> acc = ws_load_ri(ws, ws_stride, sp);
                  add(acc, Ic);
                  ws_store_ri(ws, ws_stride, sp, acc);
               }
            }
         }
      }
   }
}


// ============================================================================================ Coriolis matrix
// Mass matrix H and Coriolis / centrifugal matrix C of CompositeRigidBodyMassMatrixCalculator with setEnableCoriolisMatrixCalculation(true)
// (CompositeRigidBodyMassMatrixCalculator.java:604-630, 669-692, 709-768; factorisation B = v x* I of FactorizedBodyInertia.java).
// One inward sweep carries the composite rigid inertia Ic (10 scalars) and the composite factorised inertia Bc (30); for DoF k of body j
//     F1 = Ic Sd + Bc S,  F2 = Ic S,  F3 = Bc^T S          (Sd = v_j x S, the derivative of the constant unit twist)
// climb to the root through force transforms; at an ancestor i:  H_ik = S_i.F2,  C_ik = S_i.F1,  C_ki = Sd_i.F2 + S_i.F3
// = S_i.(F3 - v_i x* F2)  since (v x m).f = -m.(v x* f): every entry is a component pick in the canonical joint frames.
// Outputs [B][nv][nv] row-major (strides as in crba_kernel), zero-filled before the launch.  A.out = H, A.outb = C.
template <typename T>
MH_DEV void ws_store_m3(T *ws, long ws_stride, int s, const M3<T> &M)
{
   MH_WS(s + 0) = M.xx, MH_WS(s + 1) = M.xy, MH_WS(s + 2) = M.xz, MH_WS(s + 3) = M.yx, MH_WS(s + 4) = M.yy, MH_WS(s + 5) = M.yz;
   MH_WS(s + 6) = M.zx, MH_WS(s + 7) = M.zy, MH_WS(s + 8) = M.zz;
}
template <typename T>
MH_DEV M3<T> ws_load_m3(const T *ws, long ws_stride, int s)
{
   return M3<T>{MH_WS(s + 0), MH_WS(s + 1), MH_WS(s + 2), MH_WS(s + 3), MH_WS(s + 4), MH_WS(s + 5), MH_WS(s + 6), MH_WS(s + 7), MH_WS(s + 8)};
}
template <typename T>
MH_DEV void ws_store_fb(T *ws, long ws_stride, int s, const FB<T> &B)
{
   ws_store_m3(ws, ws_stride, s, B.A), ws_store_m3(ws, ws_stride, s + 9, B.TR), ws_store_m3(ws, ws_stride, s + 18, B.BL);
   MH_WS(s + 27) = B.l.x, MH_WS(s + 28) = B.l.y, MH_WS(s + 29) = B.l.z;
}
template <typename T>
MH_DEV FB<T> ws_load_fb(const T *ws, long ws_stride, int s)
{
   FB<T> B;
   B.A = ws_load_m3(ws, ws_stride, s), B.TR = ws_load_m3(ws, ws_stride, s + 9), B.BL = ws_load_m3(ws, ws_stride, s + 18);
   B.l = V3<T>{MH_WS(s + 27), MH_WS(s + 28), MH_WS(s + 29)};
   return B;
}
template <typename T>
MH_DEV void fb_up(int type, const JX<T> &jx, const XF<T> &Xb, FB<T> &B)
{
   if (type == JT_REVOLUTE)
   {
      rotate(B, revolute_rotation(jx, Xb.R));
      translate(B, Xb.p);
      return;
   }
   else if (type == JT_PRISMATIC)
      translate(B, V3<T>{T(0), T(0), jx.d});
   else if (general_x(type))
   {
      rotate(B, jx.X.R);
      translate(B, jx.X.p);
   }
   rotate(B, Xb.R);
   translate(B, Xb.p);
}

template <typename T, bool LDSC>
__global__ void __launch_bounds__(256) coriolis_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   if constexpr (LDSC)
   {
      T *C = (T *)lds_raw;
      stage_consts<T>(m, C);
      CB = C;
   }
   const ciptr meta = as_const(m.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map);
   const long lane = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const long nlanes = (long)gridDim.x * blockDim.x;
   // the workspace is a block of [slot][64 lanes] per wave: a slot offset is a scalar shifted by a constant, not a 64-bit multiply by a
   // run-time stride (which was four scalar instructions in front of every one of the kernel's ~350 workspace accesses)
   constexpr long ws_stride = 64;
   // gridDim.y waves may share a group of 64 configurations (small batches): each runs the sweeps that build the per-body state and
   // takes the columns of every gridDim.y-th body -- the columns of different bodies are independent
   const int part = blockIdx.y, parts = gridDim.y;
   T *ws = A.ws + ((long)part * gridDim.x * (blockDim.x >> 6) + (lane >> 6)) * ((long)m.n_slots * 64) + (lane & 63);
   const int nv = m.nv;
   const V3<T> Z{T(0), T(0), T(0)};

   for (long cfg = lane; cfg < A.B; cfg += nlanes)
   {
      const T *qrow = A.q + cfg * A.q_bs;
      const T *qdrow = A.qd + cfg * A.v_bs;
      const long h_bs = A.f_bs, h_es = A.f_es; // per-configuration / per-entry strides of the two matrices
      T *H = A.out + cfg * h_bs;
      T *Cm = A.outb + cfg * h_bs;
      // ---- outward sweep: joint transforms and body velocities (kept for the inward sweep: Sd of every ancestor needs them)
      SV<T> v_prev{Z, Z};
      for (int j = 0; j < m.n; j++)
      {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type = mi[MI_TYPE], flags = mi[MI_FLAGS];
         const CRef<T, LDSC> c{CB + j * MC_STRIDE};
         SV<T> vp{Z, Z};
         if (parent >= 0)
            vp = (flags & MF_PARENT_ADJ) ? v_prev : ws_load6(ws, ws_stride, meta[parent * MI_STRIDE + MI_SLOT_C]);
         const JX<T> jx = joint_from_q<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP], true);
         const SV<T> v = motion_down(type, jx, load_xb<T>(c), vp) + joint_vec<T>(type, dof_map, mi[MI_DOF], qdrow, A.v_es, true);
         ws_store6(ws, ws_stride, mi[MI_SLOT_C], v);
         v_prev = v;
      }
      // ---- inward sweep
      RI<T> rcarry;
      FB<T> bcarry;
      bool have_carry = false;
      for (int j = m.n - 1; j >= 0; j--)
      {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type = mi[MI_TYPE], flags = mi[MI_FLAGS];
         const CRef<T, LDSC> c{CB + j * MC_STRIDE};
         const SV<T> vj = ws_load6(ws, ws_stride, mi[MI_SLOT_C]);
         RI<T> Ic = load_inertia<T>(c);
         FB<T> Bc = fb_from_rigid(Ic, vj); // :671-673
         if (have_carry)
         {
            add(Ic, rcarry);
            add(Bc, bcarry);
         }
         if (flags & MF_HAS_ACC)
         {
            add(Ic, ws_load_ri(ws, ws_stride, mi[MI_SLOT_IA]));
            add(Bc, ws_load_fb(ws, ws_stride, mi[MI_SLOT_IA] + 10));
         }
         have_carry = false;
         const int nd = dof_count(type);
         ciptr dj = dof_map + mi[MI_DOF];
         const XF<T> Xb = load_xb<T>(c);
         const JX<T> jx = joint_again<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP]);
         for (int k = 0; k < (j % parts == part ? nd : 0); k++)
         {
            const SV<T> S = unit_twist<T>(type, k);
            const SV<T> Sd = crm(vj, S);            // :620-626
            SV<T> F2 = mul(Ic, S);                  // :663-667
            SV<T> F1 = mul(Ic, Sd) + mul(Bc, S);    // :686-688
            SV<T> F3 = tmul(Bc, S);                 // :690-691
            const int col = dj[k];
            for (int r = 0; r < nd; r++)
            { // the joint's own block (:698-724)
               const int e = dof_comp(type, r);
               H[((long)dj[r] * nv + col) * h_es] = comp(F2, e);
               Cm[((long)dj[r] * nv + col) * h_es] = comp(F1, e);
            }
            int prev = j, anc = parent;
            XF<T> Xp = Xb;
            JX<T> jp = jx;
            int tp = type;
            while (anc >= 0)
            { // :729-768
               F1 = force_up(tp, jp, Xp, F1);
               F2 = force_up(tp, jp, Xp, F2);
               F3 = force_up(tp, jp, Xp, F3);
               ciptr ma = meta + anc * MI_STRIDE;
               const int ta = ma[MI_TYPE];
               ciptr da = dof_map + ma[MI_DOF];
               const SV<T> G = F3 - crf(ws_load6(ws, ws_stride, ma[MI_SLOT_C]), F2);
               for (int r = 0; r < dof_count(ta); r++)
               {
                  const int e = dof_comp(ta, r);
                  const T hv = comp(F2, e);
                  H[((long)da[r] * nv + col) * h_es] = hv;
                  H[((long)col * nv + da[r]) * h_es] = hv;
                  Cm[((long)da[r] * nv + col) * h_es] = comp(F1, e);
                  Cm[((long)col * nv + da[r]) * h_es] = comp(G, e);
               }
               prev = anc;
               anc = ma[MI_PARENT];
               if (anc >= 0)
               {
                  Xp = load_xb<T>(CRef<T, LDSC>{CB + prev * MC_STRIDE});
                  jp = joint_again<T>(ta, cfg_map, ma[MI_CFG], qrow, A.q_es, ws, ws_stride, ma[MI_SLOT_JP]);
                  tp = ta;
               }
            }
         }
         if (parent >= 0)
         {
            rigid_up(type, jx, Xb, Ic); // :651-661
            fb_up(type, jx, Xb, Bc);    // :675-683
            if (flags & MF_PARENT_ADJ)
            {
               rcarry = Ic, bcarry = Bc, have_carry = true;
            }
            else
            {
               const int sp = meta[parent * MI_STRIDE + MI_SLOT_IA];
               if (flags & MF_ACC_FIRST)
               {
                  ws_store_ri(ws, ws_stride, sp, Ic);
                  ws_store_fb(ws, ws_stride, sp + 10, Bc);
               }
               else
               {
                  RI<T> acc = ws_load_ri(ws, ws_stride, sp);
                  add(acc, Ic);
                  ws_store_ri(ws, ws_stride, sp, acc);
                  FB<T> bacc = ws_load_fb(ws, ws_stride, sp + 10);
                  add(bacc, Bc);
                  ws_store_fb(ws, ws_stride, sp + 10, bacc);
               }
            }
         }
      }
   }
}

// ============================================================================================ centroidal momentum
// Centroidal momentum matrix A (6 x nv, h = A qd) and convective term b (dh/dt = A qdd + b) of CompositeRigidBodyMassMatrixCalculator
// (CompositeRigidBodyMassMatrixCalculator.java:316-342, 801-839).  Column k of A is the unit momentum F2 = Ic S_k of the mass-matrix sweep,
// climbed to the root body frame and re-expressed in the centroidal momentum frame; b is the sum over the bodies of their dynamic
// wrenches under the Coriolis accelerations (zero root and joint accelerations), which the inward sweep of the wrenches delivers at the
// root.  The frame is a constant pose (fR, fp) in the root body frame, optionally re-centred on the centre of mass of the considered
// bodies (frames/CenterOfMassReferenceFrame.java) -- known only once the sweep has reached the root, so A is then fixed up in place.
template <typename T>
struct CentArgs
{
   DevModel m;
   long B;
   const T *q, *qd; // qd may be NULL when b is not asked for
   T *A, *b, *com;  // [B][6][nv], [B][6] or NULL, [B][3] or NULL
   T *ws;
   long ws_stride;
   long q_bs, q_es, v_bs, v_es;
   long a_bs, a_es, b_bs, b_es, c_bs, c_es; // batch / element strides of A, b, com
   T fR[9], fp[3];                          // pose of the (parent of the) centroidal momentum frame in the root body frame
   int at_com;
};

template <typename T, bool LDSC>
__global__ void __launch_bounds__(256) centroidal_kernel(CentArgs<T> A)
{
   extern __shared__ double lds_raw[];
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   if constexpr (LDSC)
   {
      T *C = (T *)lds_raw;
      stage_consts<T>(m, C);
      CB = C;
   }
   const ciptr meta = as_const(m.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map);
   const long lane = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const long nlanes = (long)gridDim.x * blockDim.x;
   // the workspace is a block of [slot][64 lanes] per wave: a slot offset is a scalar shifted by a constant, not a 64-bit multiply by a
   // run-time stride (which was four scalar instructions in front of every one of the kernel's ~350 workspace accesses)
   constexpr long ws_stride = 64;
   // gridDim.y waves may share a group of 64 configurations (small batches): each runs the sweeps that build the per-body state and
   // takes the columns of every gridDim.y-th body -- the columns of different bodies are independent
   const int part = blockIdx.y, parts = gridDim.y;
   T *ws = A.ws + ((long)part * gridDim.x * (blockDim.x >> 6) + (lane >> 6)) * ((long)m.n_slots * 64) + (lane & 63);
   const int nv = m.nv;
   const V3<T> Z{T(0), T(0), T(0)};
   XF<T> Xf; // centroidal frame -> root body frame
   Xf.R = M3<T>{A.fR[0], A.fR[1], A.fR[2], A.fR[3], A.fR[4], A.fR[5], A.fR[6], A.fR[7], A.fR[8]};
   Xf.p = V3<T>{A.fp[0], A.fp[1], A.fp[2]};
   const bool with_b = A.b != nullptr;

   for (long cfg = lane; cfg < A.B; cfg += nlanes)
   {
      const T *qrow = A.q + cfg * A.q_bs;
      const T *qdrow = with_b ? A.qd + cfg * A.v_bs : nullptr;
      T *Am = A.A + cfg * A.a_bs;
      // ---- outward sweep: joint transforms; with b also velocities, Coriolis accelerations and the bodies' dynamic wrenches
      SV<T> v_prev{Z, Z}, a_prev{Z, Z};
      for (int j = 0; j < m.n; j++)
      {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type = mi[MI_TYPE], flags = mi[MI_FLAGS];
         const CRef<T, LDSC> c{CB + j * MC_STRIDE};
         const JX<T> jx = joint_from_q<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP], true);
         if (!with_b)
            continue;
         SV<T> vp{Z, Z}, ap{Z, Z};
         if (parent >= 0 && (flags & MF_PARENT_ADJ))
            vp = v_prev, ap = a_prev;
         else if (parent >= 0)
         {
            const int sp = meta[parent * MI_STRIDE + MI_SLOT_VA];
            vp = ws_load6(ws, ws_stride, sp);
            ap = ws_load6(ws, ws_stride, sp + 6);
         }
         const XF<T> Xb = load_xb<T>(c);
         const SV<T> vJ = joint_vec<T>(type, dof_map, mi[MI_DOF], qdrow, A.v_es, true);
         const SV<T> v = motion_down(type, jx, Xb, vp) + vJ;
         const SV<T> a = motion_down(type, jx, Xb, ap) + crm(v, vJ); // :826-831
         const RI<T> I = load_inertia<T>(c);
         ws_store6(ws, ws_stride, mi[MI_SLOT_F], mul(I, a) + crf(v, mul(I, v))); // :833
         if (flags & MF_STORE_VA)
         {
            ws_store6(ws, ws_stride, mi[MI_SLOT_VA], v);
            ws_store6(ws, ws_stride, mi[MI_SLOT_VA] + 6, a);
         }
         v_prev = v, a_prev = a;
      }
      // ---- inward sweep: composite inertias, columns of A, wrenches
      RI<T> rcarry, root_I{T(0), Z, S3<T>{T(0), T(0), T(0), T(0), T(0), T(0)}};
      SV<T> fcarry{Z, Z}, root_f{Z, Z};
      bool have_carry = false;
      for (int j = m.n - 1; j >= 0; j--)
      {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type = mi[MI_TYPE], flags = mi[MI_FLAGS];
         const CRef<T, LDSC> c{CB + j * MC_STRIDE};
         RI<T> Ic = load_inertia<T>(c);
         SV<T> f{Z, Z};
         if (with_b)
            f = ws_load6(ws, ws_stride, mi[MI_SLOT_F]);
         if (have_carry)
         {
            add(Ic, rcarry);
            f = f + fcarry;
         }
         if (flags & MF_HAS_ACC)
            add(Ic, ws_load_ri(ws, ws_stride, mi[MI_SLOT_IA]));
         have_carry = false;
         const int nd = dof_count(type);
         ciptr dj = dof_map + mi[MI_DOF];
         const XF<T> Xb = load_xb<T>(c);
         const JX<T> jx = joint_again<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP]);
         for (int k = 0; k < (j % parts == part ? nd : 0); k++)
         {
            SV<T> F = mul(Ic, unit_twist<T>(type, k)); // :663-667
            int prev = j, anc = parent;
            XF<T> Xp = Xb;
            JX<T> jp = jx;
            int tp = type;
            for (;;)
            { // climb to the root body frame (the mass-matrix walk :783-792 plus the last step of changeFrame, :805)
               F = force_up(tp, jp, Xp, F);
               if (anc < 0)
                  break;
               ciptr ma = meta + anc * MI_STRIDE;
               tp = ma[MI_TYPE];
               Xp = load_xb<T>(CRef<T, LDSC>{CB + anc * MC_STRIDE});
               jp = joint_again<T>(tp, cfg_map, ma[MI_CFG], qrow, A.q_es, ws, ws_stride, ma[MI_SLOT_JP]);
               prev = anc;
               anc = ma[MI_PARENT];
            }
            (void)prev;
            // root body frame -> centroidal frame: f' = R^T f ; n' = R^T (n - p x f)
            const V3<T> fl = tmul(Xf.R, F.l), fa = tmul(Xf.R, F.a - cross(Xf.p, F.l));
            const long col = dj[k];
            Am[(0 * nv + col) * A.a_es] = fa.x, Am[(1 * nv + col) * A.a_es] = fa.y, Am[(2 * nv + col) * A.a_es] = fa.z;
            Am[(3 * nv + col) * A.a_es] = fl.x, Am[(4 * nv + col) * A.a_es] = fl.y, Am[(5 * nv + col) * A.a_es] = fl.z;
         }
         rigid_up(type, jx, Xb, Ic);
         const SV<T> fp = force_up(type, jx, Xb, f);
         if (parent < 0)
         {
            add(root_I, Ic);
            root_f = root_f + fp;
         }
         else if (flags & MF_PARENT_ADJ)
         {
            rcarry = Ic, fcarry = fp, have_carry = true;
         }
         else
         {
            const int sp = meta[parent * MI_STRIDE + MI_SLOT_IA];
            if (flags & MF_ACC_FIRST)
               ws_store_ri(ws, ws_stride, sp, Ic);
            else
            {
               RI<T> acc = ws_load_ri(ws, ws_stride, sp);
               add(acc, Ic);
               ws_store_ri(ws, ws_stride, sp, acc);
            }
            if (with_b)
               ws_add6(ws, ws_stride, meta[parent * MI_STRIDE + MI_SLOT_F], fp);
         }
      }
      // ---- the frame's origin: centre of mass of the considered bodies, given in the frame (CenterOfMassCalculator.java:70-91)
      V3<T> shift{T(0), T(0), T(0)}; // in frame coordinates
      if (A.at_com)
      {
         const T inv_m = T(1) / root_I.m;
         shift = tmul(Xf.R, inv_m * root_I.h - Xf.p);
         for (int j = part; j < m.n; j += parts) // (the columns this wave wrote)
         {
            ciptr mi = meta + j * MI_STRIDE;
            ciptr dj = dof_map + mi[MI_DOF];
            for (int k = 0; k < dof_count(mi[MI_TYPE]); k++)
            { // moving the origin by `shift`: n' = n - shift x f
               const long col = dj[k];
               const V3<T> fl{Am[(3 * nv + col) * A.a_es], Am[(4 * nv + col) * A.a_es], Am[(5 * nv + col) * A.a_es]};
               const V3<T> d = cross(shift, fl);
               Am[(0 * nv + col) * A.a_es] -= d.x, Am[(1 * nv + col) * A.a_es] -= d.y, Am[(2 * nv + col) * A.a_es] -= d.z;
            }
         }
      }
      if (A.com && part == 0)
      {
         T *crow = A.com + cfg * A.c_bs;
         crow[0] = shift.x, crow[A.c_es] = shift.y, crow[2 * A.c_es] = shift.z;
      }
      if (with_b && part == 0)
      {
         const V3<T> fl = tmul(Xf.R, root_f.l);
         const V3<T> fa = tmul(Xf.R, root_f.a - cross(Xf.p, root_f.l)) - cross(shift, fl);
         T *brow = A.b + cfg * A.b_bs;
         brow[0] = fa.x, brow[A.b_es] = fa.y, brow[2 * A.b_es] = fa.z;
         brow[3 * A.b_es] = fl.x, brow[4 * A.b_es] = fl.y, brow[5 * A.b_es] = fl.z;
      }
   }
}

// ============================================================================================ relative accelerations
// RigidBodyAccelerationProvider.getRelativeAcceleration(base, body) (algorithms/interfaces/RigidBodyAccelerationProvider.java:199-235): the
// acceleration of body's body-fixed frame with respect to base's, expressed in body's, from the per-body accelerations and twists a
// previous mh_rnea_bodies / mh_aba_bodies call produced (both relative to the inertial frame, in the body-fixed frames).  The base's
// acceleration is re-expressed in the body's frame with the velocity-dependent terms of SpatialAccelerationBasics.changeFrame(desiredFrame,
// deltaTwist, bodyTwist) (spatial/interfaces/SpatialAccelerationBasics.java:192-200): lin += v_d x w_b + w_d x v_b, ang += w_d x w_b with
// (w_d, v_d) the twist of the base frame relative to the body frame and (w_b, v_b) the base's own twist, both in the base frame.
// lane = configuration; the pose of a body-fixed frame in the root frame is composed on the fly up the tree (no workspace).
template <typename T>
struct RelArgs
{
   DevModel m;
   long B;
   const T *q;
   const T *body_acc, *body_twist; // [B][n_joints][6] laid out with (f_bs, f_es); body_twist NULL = velocities not considered
   T *out;                         // [B][n_pairs][6] with (o_bs, o_es)
   const int *pairs;               // device, [n_pairs][2]: ENGINE indices (base, body), -1 = the root body
   int n_pairs;
   long q_bs, q_es, f_bs, f_es, o_bs, o_es;
   T gx, gy, gz;
   T rax, ray, raz;
};
template <typename T>
MH_DEV XF<T> compose(const XF<T> &a, const XF<T> &b)
{ // a o b: first b, then a
   XF<T> o;
   o.R = M3<T>{a.R.xx * b.R.xx + a.R.xy * b.R.yx + a.R.xz * b.R.zx, a.R.xx * b.R.xy + a.R.xy * b.R.yy + a.R.xz * b.R.zy,
               a.R.xx * b.R.xz + a.R.xy * b.R.yz + a.R.xz * b.R.zz, a.R.yx * b.R.xx + a.R.yy * b.R.yx + a.R.yz * b.R.zx,
               a.R.yx * b.R.xy + a.R.yy * b.R.yy + a.R.yz * b.R.zy, a.R.yx * b.R.xz + a.R.yy * b.R.yz + a.R.yz * b.R.zz,
               a.R.zx * b.R.xx + a.R.zy * b.R.yx + a.R.zz * b.R.zx, a.R.zx * b.R.xy + a.R.zy * b.R.yy + a.R.zz * b.R.zy,
               a.R.zx * b.R.xz + a.R.zy * b.R.yz + a.R.zz * b.R.zz};
   o.p = mul(a.R, b.p) + a.p;
   return o;
}
// motion vector child -> parent through X (child -> parent pose): w' = R w ; v' = R v + p x w'
template <typename T>
MH_DEV SV<T> motion_to_parent(const XF<T> &X, SV<T> m)
{
   SV<T> o;
   o.a = mul(X.R, m.a);
   o.l = mul(X.R, m.l) + cross(X.p, o.a);
   return o;
}
// pose of the body-fixed frame of engine body e in the root body frame (identity for e < 0)
template <typename T>
MH_DEV XF<T> body_pose_in_root(const DevModel &m, ciptr meta, ciptr cfg_map, const T *CB, const T *qrow, long q_es, int e)
{
   XF<T> X{M3<T>{T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}, V3<T>{T(0), T(0), T(0)}};
   if (e < 0)
      return X;
   {
      const CRef<T, false> c{CB + e * MC_STRIDE};
      X.R = M3<T>{c[MC_RF + 0], c[MC_RF + 1], c[MC_RF + 2], c[MC_RF + 3], c[MC_RF + 4], c[MC_RF + 5], c[MC_RF + 6], c[MC_RF + 7], c[MC_RF + 8]};
      X.p = V3<T>{c[MC_PF + 0], c[MC_PF + 1], c[MC_PF + 2]};
   }
   for (int j = e; j >= 0;)
   {
      ciptr mi = meta + j * MI_STRIDE;
      const int type = mi[MI_TYPE];
      const CRef<T, false> c{CB + j * MC_STRIDE};
      const JX<T> jx = joint_from_q<T>(type, cfg_map, mi[MI_CFG], qrow, q_es, (T *)nullptr, 0, 0, false);
      XF<T> XJ;
      if (general_x(type))
         XJ = jx.X;
      else
      {
         XJ.R = M3<T>{jx.c, -jx.s, T(0), jx.s, jx.c, T(0), T(0), T(0), T(1)};
         XJ.p = V3<T>{T(0), T(0), jx.d};
      }
      X = compose(load_xb<T>(c), compose(XJ, X));
      j = mi[MI_PARENT];
   }
   return X;
}
template <typename T>
__global__ void __launch_bounds__(256) relative_acceleration_kernel(RelArgs<T> A)
{
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   const ciptr meta = as_const(m.meta), cfg_map = as_const(m.cfg_map), pairs = as_const(A.pairs);
   const long nlanes = (long)gridDim.x * blockDim.x;
   const V3<T> Z{T(0), T(0), T(0)};
   for (long cfg = (long)blockIdx.x * blockDim.x + threadIdx.x; cfg < A.B; cfg += nlanes)
   {
      const T *qrow = A.q + cfg * A.q_bs;
      const T *arow = A.body_acc + cfg * A.f_bs;
      const T *trow = A.body_twist ? A.body_twist + cfg * A.f_bs : nullptr;
      T *orow = A.out + cfg * A.o_bs;
      auto load6 = [&](const T *row, int e) {
         const long k = (long)meta[e * MI_STRIDE + MI_EXT] * 6;
         return SV<T>{V3<T>{row[(k + 0) * A.f_es], row[(k + 1) * A.f_es], row[(k + 2) * A.f_es]},
                      V3<T>{row[(k + 3) * A.f_es], row[(k + 4) * A.f_es], row[(k + 5) * A.f_es]}};
      };
      for (int k = 0; k < A.n_pairs; k++)
      {
         const int b1 = pairs[2 * k], b2 = pairs[2 * k + 1];
         const XF<T> X1 = body_pose_in_root<T>(m, meta, cfg_map, CB, qrow, A.q_es, b1);
         const XF<T> X2 = body_pose_in_root<T>(m, meta, cfg_map, CB, qrow, A.q_es, b2);
         // pose of frame 1 in frame 2: x_2 = R2^T (R1 x_1 + p1 - p2)
         XF<T> X12;
         X12.R = M3<T>{X2.R.xx * X1.R.xx + X2.R.yx * X1.R.yx + X2.R.zx * X1.R.zx, X2.R.xx * X1.R.xy + X2.R.yx * X1.R.yy + X2.R.zx * X1.R.zy,
                       X2.R.xx * X1.R.xz + X2.R.yx * X1.R.yz + X2.R.zx * X1.R.zz, X2.R.xy * X1.R.xx + X2.R.yy * X1.R.yx + X2.R.zy * X1.R.zx,
                       X2.R.xy * X1.R.xy + X2.R.yy * X1.R.yy + X2.R.zy * X1.R.zy, X2.R.xy * X1.R.xz + X2.R.yy * X1.R.yz + X2.R.zy * X1.R.zz,
                       X2.R.xz * X1.R.xx + X2.R.yz * X1.R.yx + X2.R.zz * X1.R.zx, X2.R.xz * X1.R.xy + X2.R.yz * X1.R.yy + X2.R.zz * X1.R.zy,
                       X2.R.xz * X1.R.xz + X2.R.yz * X1.R.yz + X2.R.zz * X1.R.zz};
         X12.p = tmul(X2.R, X1.p - X2.p);
         const SV<T> a_root = root_acceleration(A);
         SV<T> a1 = b1 < 0 ? a_root : load6(arow, b1);
         const SV<T> a2 = b2 < 0 ? a_root : load6(arow, b2);
         if (trow)
         {
            const SV<T> t1 = b1 < 0 ? SV<T>{Z, Z} : load6(trow, b1), t2 = b2 < 0 ? SV<T>{Z, Z} : load6(trow, b2);
            const SV<T> d = t1 - motion_to_child(X12, t2); // twist of the base frame relative to the body frame, in the base frame (:217)
            a1.l = a1.l + cross(d.l, t1.a) + cross(d.a, t1.l);
            a1.a = a1.a + cross(d.a, t1.a);
         }
         const SV<T> r = a2 - motion_to_parent(X12, a1); // :228
         const long o = (long)k * 6;
         orow[(o + 0) * A.o_es] = r.a.x, orow[(o + 1) * A.o_es] = r.a.y, orow[(o + 2) * A.o_es] = r.a.z;
         orow[(o + 3) * A.o_es] = r.l.x, orow[(o + 4) * A.o_es] = r.l.y, orow[(o + 5) * A.o_es] = r.l.z;
      }
   }
}

// ============================================================================================ joint torque regressor
// algorithms/JointTorqueRegressorCalculator.java:173-190, 749-833: tau = Y(q, qd, qdd) pi with pi = the ten inertial parameters of every
// body (mass, centre-of-mass offset, Ixx, Ixy, Ixz, Iyy, Iyz, Izz in its body-fixed frame, :877-889).  The reference zeroes every inertia,
// runs the first pass of the inverse dynamics once and its second pass once per (body, parameter) with that body's inertia set to the unit
// basis (:795-806, :574-590).  A body alone carrying inertia loads exactly the joints between it and the root, so here one outward sweep
// computes every body's twist and acceleration, forms the ten basis wrenches in the body-fixed frame (computeDynamicWrench,
// spatial/interfaces/SpatialInertiaReadOnly.java:229-296) and carries them up the ancestors together; entries of joints that do not
// support the body stay zero (memset by the host).  The bases MCOM_X/Y/Z set a centre-of-mass offset on a body of zero mass
// (:579-581): with a twist every term of that wrench is multiplied by the mass (tools/MecanoTools.java:632-702, 785-822), so the
// reference's columns 1..3 are zero -- MODE 0; without one (Coriolis and centrifugal terms switched off: the twist is null,
// InverseDynamicsCalculator.java:937) computeDynamicMoment leaves c x a unscaled (tools/MecanoTools.java:650-692: the scale sits inside
// the velocity branch), so the columns hold the moment e x a -- MODE 2.  MODE 1 writes the derivatives with respect to the first
// moment m c instead (what an identification wants).
// Y is [B][nv][10 n]: A.out, A.f_bs = nv * 10 n, A.f_es = 1 (MH_LAYOUT_SOA: [nv][10 n][B], f_bs = 1, f_es = B); the block of body `e`
// (caller's joint order) starts at column 10 e.
template <typename T, bool LDSC, int MODE>
__global__ void __launch_bounds__(256) regressor_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   if constexpr (LDSC)
   {
      T *C = (T *)lds_raw;
      stage_consts<T>(m, C);
      CB = C;
   }
   const ciptr meta = as_const(m.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map);
   const long lane = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const long nlanes = (long)gridDim.x * blockDim.x;
   constexpr long ws_stride = 64;
   // gridDim.y waves share a group of 64 configurations: each runs the whole first pass (a fifth of the work) and the wrenches and the
   // climb of every gridDim.y-th body -- small batches spread over the device that way (the bodies' columns are independent)
   const int part = blockIdx.y, parts = gridDim.y;
   T *ws = A.ws + ((long)part * gridDim.x * (blockDim.x >> 6) + (lane >> 6)) * ((long)m.n_slots * 64) + (lane & 63);
   const V3<T> Z{T(0), T(0), T(0)};
   const long ycols = (long)m.n * 10;
   constexpr bool FM = MODE != 0;
   constexpr int NW = FM ? 10 : 7; // wrenches carried: mass, (centre-of-mass columns,) six moments of inertia
   constexpr int I0 = FM ? 4 : 1;

   for (long cfg = lane; cfg < A.B; cfg += nlanes)
   {
      const T *qrow = A.q + cfg * A.q_bs;
      const T *qdrow = A.qd + cfg * A.v_bs;
      const T *qddrow = A.in3 + cfg * A.v_bs;
      T *Y = A.out + cfg * A.f_bs;
      SV<T> v_prev{Z, Z}, a_prev{Z, Z};
      for (int j = 0; j < m.n; j++)
      {
         ciptr mi = meta + j * MI_STRIDE;
         const int parent = mi[MI_PARENT], type = mi[MI_TYPE], flags = mi[MI_FLAGS];
         const CRef<T, LDSC> c{CB + j * MC_STRIDE};
         // ---- the first pass of the inverse dynamics (InverseDynamicsCalculator.java:873-917), as rnea_kernel runs it
         SV<T> vp, ap;
         if (parent < 0)
         {
            vp = SV<T>{Z, Z};
            ap = root_acceleration(A);
         }
         else if (flags & MF_PARENT_ADJ)
            vp = v_prev, ap = a_prev;
         else
         {
            const int sp = meta[parent * MI_STRIDE + MI_SLOT_VA];
            vp = ws_load6(ws, ws_stride, sp);
            ap = ws_load6(ws, ws_stride, sp + 6);
         }
         const XF<T> Xb = load_xb<T>(c);
         const JX<T> jx = joint_from_q<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, ws, ws_stride, mi[MI_SLOT_JP], true);
         const SV<T> vJ = joint_vec<T>(type, dof_map, mi[MI_DOF], qdrow, A.v_es, A.coriolis != 0);
         const SV<T> aJ = joint_vec<T>(type, dof_map, mi[MI_DOF], qddrow, A.v_es, A.accel != 0);
         SV<T> v = motion_down(type, jx, Xb, vp) + vJ;
         const SV<T> a = motion_down(type, jx, Xb, ap) + aJ + crm(v, vJ);
         if (!A.coriolis)
            v = SV<T>{Z, Z};
         if (flags & MF_STORE_VA)
         {
            ws_store6(ws, ws_stride, mi[MI_SLOT_VA], v);
            ws_store6(ws, ws_stride, mi[MI_SLOT_VA] + 6, a);
         }
         v_prev = v, a_prev = a;
         if (j % parts != part)
            continue;
         // ---- the basis wrenches of this body, in its body-fixed frame (:574-590 with computeDynamicWrench)
         XF<T> Xf;
         Xf.R = M3<T>{c[MC_RF + 0], c[MC_RF + 1], c[MC_RF + 2], c[MC_RF + 3], c[MC_RF + 4], c[MC_RF + 5], c[MC_RF + 6], c[MC_RF + 7], c[MC_RF + 8]};
         Xf.p = V3<T>{c[MC_PF + 0], c[MC_PF + 1], c[MC_PF + 2]};
         const SV<T> vb = motion_to_child(Xf, v), ab = motion_to_child(Xf, a);
         const V3<T> w = vb.a, wd = ab.a;
         SV<T> W[NW];
         W[0] = SV<T>{Z, ab.l + cross(w, vb.l)}; // M: f = a + w x v (computeDynamicForceFast, tools/MecanoTools.java:728-752)
         if constexpr (FM)
         { // d/d(m c): n = h x a + w (v.h) - v (w.h);  f = -(h x wd) - w x (h x w)   (tools/MecanoTools.java:632-702, 785-822)
            const V3<T> E[3] = {V3<T>{T(1), T(0), T(0)}, V3<T>{T(0), T(1), T(0)}, V3<T>{T(0), T(0), T(1)}};
#pragma unroll
            for (int e = 0; e < 3; e++)
            {
               const V3<T> h = E[e];
               if constexpr (MODE == 2)
                  W[1 + e] = SV<T>{cross(h, ab.l), Z}; // the reference without a twist: c x a, not scaled by the (zero) mass
               else
               {
                  const V3<T> n = cross(h, ab.l) + (dot(vb.l, h) * w - dot(w, h) * vb.l);
                  const V3<T> f = Z - (cross(h, wd) + cross(w, cross(h, w)));
                  W[1 + e] = SV<T>{n, f};
               }
            }
         }
         // I_XX .. I_ZZ: n = E wd + w x (E w) (computeDynamicMomentFast, tools/MecanoTools.java:571-598); the off-diagonal bases are symmetric
         W[I0 + 0] = SV<T>{V3<T>{wd.x, T(0), T(0)} + cross(w, V3<T>{w.x, T(0), T(0)}), Z};
         W[I0 + 1] = SV<T>{V3<T>{wd.y, wd.x, T(0)} + cross(w, V3<T>{w.y, w.x, T(0)}), Z};
         W[I0 + 2] = SV<T>{V3<T>{wd.z, T(0), wd.x} + cross(w, V3<T>{w.z, T(0), w.x}), Z};
         W[I0 + 3] = SV<T>{V3<T>{T(0), wd.y, T(0)} + cross(w, V3<T>{T(0), w.y, T(0)}), Z};
         W[I0 + 4] = SV<T>{V3<T>{T(0), wd.z, wd.y} + cross(w, V3<T>{T(0), w.z, w.y}), Z};
         W[I0 + 5] = SV<T>{V3<T>{T(0), T(0), wd.z} + cross(w, V3<T>{T(0), T(0), w.z}), Z};
#pragma unroll
         for (int k = 0; k < NW; k++)
            W[k] = force_to_parent(Xf, W[k]); // to the frame after the joint (InverseDynamicsCalculator.java:936-941)
         // ---- second pass for this body alone (:930-959): the joints from here to the root carry the wrenches
         T *Yb = Y + (long)mi[MI_EXT] * 10 * A.f_es;
         int cur = j, tc = type;
         JX<T> jc = jx;
         XF<T> Xc = Xb;
         for (;;)
         {
            ciptr mc = meta + cur * MI_STRIDE;
            ciptr dc = dof_map + mc[MI_DOF];
            const int nd = dof_count(tc);
            // the parent's joint transform is fetched (workspace / q) before this level's stores and transforms: a lone wave has
            // nothing else to cover that latency with
            const int up = mc[MI_PARENT];
            int tn = JT_FIXED;
            XF<T> Xn = Xc;
            JX<T> jn = jc;
            if (up >= 0)
            {
               ciptr mu = meta + up * MI_STRIDE;
               tn = mu[MI_TYPE];
               Xn = load_xb<T>(CRef<T, LDSC>{CB + up * MC_STRIDE});
               jn = joint_again<T>(tn, cfg_map, mu[MI_CFG], qrow, A.q_es, ws, ws_stride, mu[MI_SLOT_JP]);
            }
            for (int r = 0; r < nd; r++)
            {
               const int cp = dof_comp(tc, r);
               T e[10];
#pragma unroll
               for (int k = 0; k < 10; k++)
                  e[k] = T(0);
#pragma unroll
               for (int k = 0; k < NW; k++)
                  e[k == 0 ? 0 : (FM ? k : k + 3)] = comp(W[k], cp);
               if (A.f_es == 1)
               { // one matrix per configuration: the body's ten columns of this row are 80 (40) contiguous bytes, written as five pairs
                  typedef T pair_t __attribute__((ext_vector_type(2)));
                  pair_t *yr = (pair_t *)(Yb + (long)dc[r] * ycols);
#pragma unroll
                  for (int k = 0; k < 5; k++)
                     yr[k] = pair_t{e[2 * k], e[2 * k + 1]};
               }
               else
               { // MH_LAYOUT_SOA: [nv * 10 n][B], consecutive lanes write consecutive addresses
                  T *yr = Yb + (long)dc[r] * ycols * A.f_es;
#pragma unroll
                  for (int k = 0; k < 10; k++)
                     if (FM || k == 0 || k > 3)
                        yr[(long)k * A.f_es] = e[k];
               }
            }
            if (up < 0)
               break;
#pragma unroll
            for (int k = 0; k < NW; k++)
               W[k] = force_up(tc, jc, Xc, W[k]);
            tc = tn, Xc = Xn, jc = jn;
            cur = up;
         }
      }
   }
}

// ints of synchronisation state per 64 configurations of a bias-split launch (mh_zv_kernels.h): two 128-byte lines, so that no line
// mixes words stored by workgroups behind different L2s
constexpr int ZV_SYNC_STRIDE = 64;

// ---- stamp of everything a topology-specialised code object shares with the library beyond the C-ABI: the argument structs it
// reinterprets, the strides of the per-joint records, the canonical-frame convention.  mh_model_create refuses a code object whose stamp
// differs (a stale libmecano_hip_topo_<key>.so would otherwise read a mis-laid-out struct or fold zeros that are not there).
constexpr unsigned long long spec_abi_stamp()
{
   unsigned long long h = 1469598103934665603ull;
   const unsigned long long parts[] = {sizeof(Args<double>), sizeof(CentArgs<double>), (unsigned long long)MC_STRIDE, (unsigned long long)MI_STRIDE,
                                       (unsigned long long)MH_FRAME_CONVENTION, (unsigned long long)offsetof(Args<double>, joint_wrench),
                                       (unsigned long long)offsetof(Args<double>, q_next), (unsigned long long)ZV_SYNC_STRIDE,
                                       4ull /* revision of the mh_spec_* entry points' signatures: 2 = mh_spec_launch_zv(..., same_l2, stream); 3 = mh_spec_launch_zvb */};
   for (unsigned long long v : parts)
      h = (h ^ v) * 1099511628211ull;
   return h;
}

#undef MH_WS
} // namespace mh
