/*
 * mecano_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A scalar, single-threaded, double-precision restatement of the three Mecano
 * calculators this repository accelerates, written from the reference's
 * algorithm (frames, step order, branch conditions) and NOT copied from it:
 *
 *   RNEA  algorithms/InverseDynamicsCalculator.java:873-959
 *   ABA   algorithms/ForwardDynamicsCalculator.java:1085-1310
 *   CRBA  algorithms/CompositeRigidBodyMassMatrixCalculator.java:588-707,770-798
 *
 * (all citations are relative to /root/reference/src/main/java/us/ihmc/mecano/).
 *
 * PARITY STATUS: the reference is Java 17 + EJML 0.39 + Euclid 0.21.0; no JVM
 * exists in the build container or on the GPU box, and the reference's tests
 * hold no stored golden vectors for this path.  This
 * oracle is therefore pinned only by (i) the reference tests' own invariants
 * and closed-form known answers, restated in tests/ (ABA o RNEA = id,
 * H qdd + bias = RNEA, ballistic free body, sphere wrench, ...), (ii) an
 * independent textbook Featherstone implementation (oracle/featherstone_np.py)
 * and (iii) sympy Lagrangian closed forms (tests/golden/).  Against the Java
 * reference itself:  *** parity unpinned ***.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this file.  The product (mecano_amd/csrc) never does.
 *
 * Like Mecano, the oracle keeps
 *   - body accelerations in the body-fixed (CoM) frames for RNEA
 *     (InverseDynamicsCalculator.java:798,885) and everything of ABA / CRBA in
 *     the frames after the joints (ForwardDynamicsCalculator.java:175-176);
 *   - transforms between neighbouring frames composed through the world frame,
 *     as Euclid's ReferenceFrame.getTransformToDesiredFrame does
 *     (spatial/SpatialAcceleration.java:266-277);
 *   - the articulated-body inertia as three 3x3 blocks (angular A, linear L,
 *     cross C) transformed by "rotate, then translate"
 *     (algorithms/ArticulatedBodyInertia.java:42-55,359-375);
 *   - composite rigid-body inertias as (J, m, c) triples with the |m| >= 1e-7
 *     guard when adding (spatial/interfaces/FixedFrameSpatialInertiaBasics.java:167-176);
 *   - the fast / general Newton-Euler switch at |c|^2 < 1e-11
 *     (spatial/interfaces/SpatialInertiaReadOnly.java:56,104-107).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MO_REVOLUTE 0
#define MO_PRISMATIC 1
#define MO_SIXDOF 2
#define MO_FIXED 3
#define MO_PLANAR 4    /* q = (pitch, x, z), qd = (w_y, v_x, v_z): multiBodySystem/interfaces/PlanarJointReadOnly.java:17-72 */
#define MO_SPHERICAL 5 /* q = quaternion (x, y, z, s), qd = angular velocity: multiBodySystem/interfaces/SphericalJointReadOnly.java:18-104 */

#define MO_MAX_JOINTS 512
#define COM_OFFSET_ZERO_EPSILON 1.0e-11 /* SpatialInertiaReadOnly.java:56 */

typedef struct
{
   double R[9];
   double p[3];
} xf_t;

typedef struct
{
   int n, nq, nv;
   int parent[MO_MAX_JOINTS];
   int type[MO_MAX_JOINTS];
   int ndof[MO_MAX_JOINTS], ncfg[MO_MAX_JOINTS];
   int dof_ofs[MO_MAX_JOINTS], cfg_ofs[MO_MAX_JOINTS]; /* offsets into the concatenated index maps */
   int *dof_idx, *cfg_idx;
   double axis[MO_MAX_JOINTS][3];
   xf_t Xb[MO_MAX_JOINTS];   /* beforeJoint -> parent frame           */
   xf_t Xcom[MO_MAX_JOINTS]; /* body-fixed  -> afterJoint              */
   double J[MO_MAX_JOINTS][9];
   double mass[MO_MAX_JOINTS];
   double com[MO_MAX_JOINTS][3];
} mo_model;

/* ------------------------------------------------------------------ 3-vector / 3x3 helpers */
static void v3_cross(const double a[3], const double b[3], double out[3])
{
   double x = a[1] * b[2] - a[2] * b[1];
   double y = a[2] * b[0] - a[0] * b[2];
   double z = a[0] * b[1] - a[1] * b[0];
   out[0] = x, out[1] = y, out[2] = z;
}
static double v3_dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void m3_mulv(const double M[9], const double v[3], double out[3])
{
   double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
   double y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
   double z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
   out[0] = x, out[1] = y, out[2] = z;
}
static void m3_tmulv(const double M[9], const double v[3], double out[3])
{
   double x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
   double y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
   double z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
   out[0] = x, out[1] = y, out[2] = z;
}
static void m3_mul(const double A[9], const double B[9], double out[9])
{
   double t[9];
   for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
         t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
   memcpy(out, t, sizeof t);
}
static void m3_transpose(const double A[9], double out[9])
{
   double t[9] = {A[0], A[3], A[6], A[1], A[4], A[7], A[2], A[5], A[8]};
   memcpy(out, t, sizeof t);
}
static void m3_tilde(const double v[3], double out[9])
{
   out[0] = 0, out[1] = -v[2], out[2] = v[1];
   out[3] = v[2], out[4] = 0, out[5] = -v[0];
   out[6] = -v[1], out[7] = v[0], out[8] = 0;
}
/* M <- R M R^T   (tools/MecanoTools.java:1053-1071 for symmetric M; RotationMatrix.transform(Matrix3D) in general) */
static void m3_conj(const double R[9], double M[9])
{
   double Rt[9], t[9];
   m3_transpose(R, Rt);
   m3_mul(R, M, t);
   m3_mul(t, Rt, M);
}

/* ------------------------------------------------------------------ rigid transforms */
static void xf_identity(xf_t *X)
{
   memset(X, 0, sizeof *X);
   X->R[0] = X->R[4] = X->R[8] = 1.0;
}
static void xf_mul(const xf_t *a, const xf_t *b, xf_t *out) /* out = a o b */
{
   xf_t t;
   m3_mul(a->R, b->R, t.R);
   m3_mulv(a->R, b->p, t.p);
   for (int k = 0; k < 3; k++)
      t.p[k] += a->p[k];
   *out = t;
}
static void xf_inv(const xf_t *a, xf_t *out)
{
   xf_t t;
   m3_transpose(a->R, t.R);
   m3_mulv(t.R, a->p, t.p);
   for (int k = 0; k < 3; k++)
      t.p[k] = -t.p[k];
   *out = t;
}
/* transform taking coordinates in frame A to frame B, composed through the world like Euclid does */
static void xf_between(const xf_t *W_A, const xf_t *W_B, xf_t *out)
{
   xf_t inv;
   xf_inv(W_B, &inv);
   xf_mul(&inv, W_A, out);
}
/* motion vectors: w' = R w ; v' = R v + p x w'   (spatial/interfaces/FixedFrameSpatialMotionBasics.java:311-321) */
static void xf_motion(const xf_t *X, const double in[6], double out[6])
{
   double w[3], v[3], c[3];
   m3_mulv(X->R, in, w);
   m3_mulv(X->R, in + 3, v);
   v3_cross(X->p, w, c);
   for (int k = 0; k < 3; k++)
      out[k] = w[k], out[3 + k] = v[k] + c[k];
}
/* force vectors: f' = R f ; n' = R n + p x f'   (spatial/interfaces/FixedFrameSpatialForceBasics.java:249-259) */
static void xf_force(const xf_t *X, const double in[6], double out[6])
{
   double n[3], f[3], c[3];
   m3_mulv(X->R, in, n);
   m3_mulv(X->R, in + 3, f);
   v3_cross(X->p, f, c);
   for (int k = 0; k < 3; k++)
      out[k] = n[k] + c[k], out[3 + k] = f[k];
}
/* inverse of xf_motion: v'' = R^T (v - p x w) ; w'' = R^T w  (FixedFrameSpatialMotionBasics.java:343-353) */
static void xf_motion_inv(const xf_t *X, const double in[6], double out[6])
{
   double c[3], t[3];
   v3_cross(X->p, in, c);
   for (int k = 0; k < 3; k++)
      t[k] = in[3 + k] - c[k];
   m3_tmulv(X->R, in, out);
   m3_tmulv(X->R, t, out + 3);
}

/* ------------------------------------------------------------------ joint kinematics */
/* Joint transform afterJoint -> beforeJoint.
 * revolute : tools/MecanoFactories.java:231-260 (axis-angle; within 1e-7 of X / Y / Z the roll / pitch / yaw closed forms, i.e. the exact coordinate axis)
 * prismatic: multiBodySystem/interfaces/PrismaticJointReadOnly.java:18-22
 * sixdof   : multiBodySystem/interfaces/FloatingJointReadOnly.java:34-37 (quaternion x,y,z,s then position) */
static void joint_transform(const mo_model *m, int i, const double *qrow, xf_t *X)
{
   xf_identity(X);
   const int *ci = m->cfg_idx + m->cfg_ofs[i];
   switch (m->type[i])
   {
      case MO_REVOLUTE:
      {
         double q = qrow[ci[0]];
         const double *a = m->axis[i];
         double nrm = sqrt(v3_dot(a, a));
         double ux = a[0] / nrm, uy = a[1] / nrm, uz = a[2] / nrm;
         /* tools/MecanoFactories.java:51, 237-248: an axis that geometricallyEquals X, Y or Z within TRANSFORM_UPDATER_EPSILON = 1e-7 gets the
          * closed-form roll / pitch / yaw matrix -- a rotation about EXACTLY that coordinate axis -- while the joint's unit twist keeps the
          * axis as given (multiBodySystem/OneDoFJoint.java:170).  (Euclid's geometricallyEquals for vectors, un-vendored: norm of the
          * difference <= epsilon.)  An axis exactly on X / Y / Z gives the same matrix either way. */
         for (int k = 0; k < 3; k++)
         {
            double d[3] = {a[0] - (k == 0), a[1] - (k == 1), a[2] - (k == 2)};
            if (sqrt(v3_dot(d, d)) <= 1.0e-7)
            {
               ux = k == 0, uy = k == 1, uz = k == 2;
               break;
            }
         }
         double c = cos(q), s = sin(q), t = 1.0 - c;
         X->R[0] = t * ux * ux + c, X->R[1] = t * ux * uy - s * uz, X->R[2] = t * ux * uz + s * uy;
         X->R[3] = t * ux * uy + s * uz, X->R[4] = t * uy * uy + c, X->R[5] = t * uy * uz - s * ux;
         X->R[6] = t * ux * uz - s * uy, X->R[7] = t * uy * uz + s * ux, X->R[8] = t * uz * uz + c;
         break;
      }
      case MO_PRISMATIC:
      {
         double q = qrow[ci[0]];
         for (int k = 0; k < 3; k++)
            X->p[k] = q * m->axis[i][k];
         break;
      }
      case MO_SIXDOF:
      {
         double x = qrow[ci[0]], y = qrow[ci[1]], z = qrow[ci[2]], s = qrow[ci[3]];
         double nrm = sqrt(x * x + y * y + z * z + s * s); /* Euclid Quaternion.set normalises */
         x /= nrm, y /= nrm, z /= nrm, s /= nrm;
         X->R[0] = 1 - 2 * (y * y + z * z), X->R[1] = 2 * (x * y - z * s), X->R[2] = 2 * (x * z + y * s);
         X->R[3] = 2 * (x * y + z * s), X->R[4] = 1 - 2 * (x * x + z * z), X->R[5] = 2 * (y * z - x * s);
         X->R[6] = 2 * (x * z - y * s), X->R[7] = 2 * (y * z + x * s), X->R[8] = 1 - 2 * (x * x + y * y);
         X->p[0] = qrow[ci[4]], X->p[1] = qrow[ci[5]], X->p[2] = qrow[ci[6]];
         break;
      }
      case MO_PLANAR:
      { /* rotation about y by the pitch, translation (x, 0, z): the pose restricted to the XZ plane (multiBodySystem/PlanarJoint.java:25-26) */
         double pitch = qrow[ci[0]];
         double c = cos(pitch), sn = sin(pitch);
         X->R[0] = c, X->R[1] = 0, X->R[2] = sn;
         X->R[3] = 0, X->R[4] = 1, X->R[5] = 0;
         X->R[6] = -sn, X->R[7] = 0, X->R[8] = c;
         X->p[0] = qrow[ci[1]], X->p[1] = 0, X->p[2] = qrow[ci[2]];
         break;
      }
      case MO_SPHERICAL:
      { /* SphericalJointReadOnly.java:50-53 */
         double x = qrow[ci[0]], y = qrow[ci[1]], z = qrow[ci[2]], s = qrow[ci[3]];
         double nrm = sqrt(x * x + y * y + z * z + s * s);
         x /= nrm, y /= nrm, z /= nrm, s /= nrm;
         X->R[0] = 1 - 2 * (y * y + z * z), X->R[1] = 2 * (x * y - z * s), X->R[2] = 2 * (x * z + y * s);
         X->R[3] = 2 * (x * y + z * s), X->R[4] = 1 - 2 * (x * x + z * z), X->R[5] = 2 * (y * z - x * s);
         X->R[6] = 2 * (x * z - y * s), X->R[7] = 2 * (y * z + x * s), X->R[8] = 1 - 2 * (x * x + y * y);
         break;
      }
      default:
         break;
   }
}
/* Motion subspace S (6 x ndof, column-major by DoF) in afterJoint coordinates:
 * unit twists (multiBodySystem/interfaces/JointReadOnly.java:201-207); SixDoF identity (tools/MecanoTools.java:964-1000) */
static void joint_subspace(const mo_model *m, int i, double S[6][6])
{
   memset(S, 0, 36 * sizeof(double));
   switch (m->type[i])
   {
      case MO_REVOLUTE:
         for (int k = 0; k < 3; k++)
            S[0][k] = m->axis[i][k];
         break;
      case MO_PRISMATIC:
         for (int k = 0; k < 3; k++)
            S[0][3 + k] = m->axis[i][k];
         break;
      case MO_SIXDOF:
         for (int k = 0; k < 6; k++)
            S[k][k] = 1.0;
         break;
      case MO_PLANAR: /* w_y, v_x, v_z (tools/MecanoTools.java:920-952) */
         S[0][1] = 1.0, S[1][3] = 1.0, S[2][5] = 1.0;
         break;
      case MO_SPHERICAL: /* w_x, w_y, w_z (tools/MecanoTools.java:1002-1043) */
         S[0][0] = 1.0, S[1][1] = 1.0, S[2][2] = 1.0;
         break;
      default:
         break;
   }
}
/* S * x for the DoFs of joint i, x read through the dof index map (NULL -> zero) */
static void joint_S_times(const mo_model *m, int i, double S[6][6], const double *xrow, double out[6])
{
   memset(out, 0, 6 * sizeof(double));
   if (!xrow)
      return;
   const int *di = m->dof_idx + m->dof_ofs[i];
   for (int d = 0; d < m->ndof[i]; d++)
      for (int k = 0; k < 6; k++)
         out[k] += S[d][k] * xrow[di[d]];
}

/* ------------------------------------------------------------------ per-configuration kinematics (the frame tree) */
typedef struct
{
   xf_t W_after[MO_MAX_JOINTS]; /* afterJoint_i  -> world */
   xf_t W_body[MO_MAX_JOINTS];  /* bodyFixed_i   -> world */
   double tw_after[MO_MAX_JOINTS][6]; /* twist of afterJoint_i w.r.t. world, in afterJoint_i */
   double tw_body[MO_MAX_JOINTS][6];  /* same body, expressed in bodyFixed_i                  */
   double vJ[MO_MAX_JOINTS][6];       /* joint twist in afterJoint_i                          */
   double S[MO_MAX_JOINTS][6][6];
} mo_kin;

/* rootBody.updateFramesRecursively() + lazy twist propagation (frames/MovingReferenceFrame.java:279-311) */
static void kinematics(const mo_model *m, const double *qrow, const double *qdrow, mo_kin *k)
{
   xf_t W_world;
   double zero6[6] = {0};
   xf_identity(&W_world);
   for (int i = 0; i < m->n; i++)
   {
      int p = m->parent[i];
      const xf_t *W_par = p < 0 ? &W_world : &k->W_after[p];
      const double *tw_par = p < 0 ? zero6 : k->tw_after[p];
      xf_t W_before, XJ, T;
      double tw_before[6], t6[6];

      xf_mul(W_par, &m->Xb[i], &W_before);
      joint_transform(m, i, qrow, &XJ);
      xf_mul(&W_before, &XJ, &k->W_after[i]);
      xf_mul(&k->W_after[i], &m->Xcom[i], &k->W_body[i]);

      joint_subspace(m, i, k->S[i]);
      joint_S_times(m, i, k->S[i], qdrow, k->vJ[i]); /* OneDoFJoint.java:170-172, SixDoFJoint.java:67-69 */

      xf_between(W_par, &W_before, &T);
      xf_motion(&T, tw_par, tw_before);
      xf_between(&W_before, &k->W_after[i], &T);
      xf_motion(&T, tw_before, t6);
      for (int c = 0; c < 6; c++)
         k->tw_after[i][c] = t6[c] + k->vJ[i][c];
      xf_between(&k->W_after[i], &k->W_body[i], &T);
      xf_motion(&T, k->tw_after[i], k->tw_body[i]);
   }
}

/* ------------------------------------------------------------------ Newton-Euler of one body
 * spatial/interfaces/SpatialInertiaReadOnly.java:229-296 with tools/MecanoTools.java:571-598,728-752 (CoM at the origin)
 * or :632-702,785-822 (offset CoM).  acc / tw may be NULL exactly like the Java arguments. */
static _Thread_local int unit_force_general_wrench = 0; /* unit tests only: take the general branch whatever |c| is */
static void dynamic_wrench(const double J[9], double mass, const double c[3], const double *acc, const double *tw, double out[6])
{
   double n[3] = {0, 0, 0}, f[3] = {0, 0, 0};
   if (!unit_force_general_wrench && v3_dot(c, c) < COM_OFFSET_ZERO_EPSILON)
   {
      if (tw)
      {
         double Jw[3];
         m3_mulv(J, tw, Jw);
         v3_cross(tw, Jw, n);
      }
      if (acc)
      {
         double Jwd[3];
         m3_mulv(J, acc, Jwd);
         for (int k = 0; k < 3; k++)
            n[k] += Jwd[k];
      }
      if (acc)
         for (int k = 0; k < 3; k++)
            f[k] = acc[3 + k];
      if (tw)
      {
         double wxv[3];
         v3_cross(tw, tw + 3, wxv);
         for (int k = 0; k < 3; k++)
            f[k] += wxv[k];
      }
      for (int k = 0; k < 3; k++)
         f[k] *= mass;
   }
   else
   {
      /* moment: J wd + w x J w + m (c x a + w (v.c) - v (w.c)) */
      double t[3] = {0, 0, 0};
      if (acc)
         v3_cross(c, acc + 3, t);
      if (tw)
      {
         double wc = v3_dot(tw, c), vc = v3_dot(tw + 3, c), Jw[3], wJw[3];
         for (int k = 0; k < 3; k++)
            t[k] = mass * (t[k] + tw[k] * vc - tw[3 + k] * wc);
         m3_mulv(J, tw, Jw);
         v3_cross(tw, Jw, wJw);
         for (int k = 0; k < 3; k++)
            t[k] += wJw[k];
      }
      /* (without a twist the reference leaves c x a unscaled by the mass: MecanoTools.java:650-692) */
      if (acc)
      {
         double Jwd[3];
         m3_mulv(J, acc, Jwd);
         for (int k = 0; k < 3; k++)
            n[k] = Jwd[k] + t[k];
      }
      else
         memcpy(n, t, sizeof t);
      /* force: m (a - c x wd - w x (c x w - v)) */
      double g[3] = {0, 0, 0};
      if (acc)
      {
         double cxwd[3];
         v3_cross(c, acc, cxwd);
         for (int k = 0; k < 3; k++)
            g[k] = acc[3 + k] - cxwd[k];
      }
      if (tw)
      {
         double cxw[3], u[3], wxu[3];
         v3_cross(c, tw, cxw);
         for (int k = 0; k < 3; k++)
            u[k] = cxw[k] - tw[3 + k];
         v3_cross(tw, u, wxu);
         for (int k = 0; k < 3; k++)
            g[k] -= wxu[k];
      }
      for (int k = 0; k < 3; k++)
         f[k] = mass * g[k];
   }
   for (int k = 0; k < 3; k++)
      out[k] = n[k], out[3 + k] = f[k];
}

/* ================================================================== RNEA
 * InverseDynamicsCalculator.java:873-917 (passOne), :930-959 (passTwo), :961-966 (children) */
/* optional per-body outputs (RigidBodyAccelerationProvider): when set, rnea_one / aba_one also write the
 * spatial acceleration (InverseDynamicsCalculator.java:242-250; ForwardDynamicsCalculator.java:170-180: changed to the body-fixed
 * frame on request) and the twist (frames/MovingReferenceFrame.java:279-311) of every successor body, both relative to the inertial
 * frame and expressed in the body-fixed frame, 6 numbers (angular, linear) per listed joint */
static _Thread_local double *tap_acc = NULL, *tap_twist = NULL;
/* per-joint wrench (InverseDynamicsCalculator.getComputedJointWrench, :578-585): what passTwo leaves in jointWrench, frame after the joint */
static _Thread_local double *tap_wrench = NULL;

/* Root acceleration (InverseDynamicsCalculator.java:343-348 / 413-427, ForwardDynamicsCalculator.java:259-264 / 330-343): setGravity stores
 * (0, -g); setRootAcceleration stores the given spatial acceleration (angular, linear; root-body coordinates) in the same field.  The
 * entry points take the gravity vector; mo_set_root_acceleration (thread-local, like the calculators' field) overrides it until cleared. */
static _Thread_local int root_override_set = 0;
static _Thread_local double root_override[6];
void mo_set_root_acceleration(const double *a6)
{
   root_override_set = a6 != NULL;
   if (a6)
      memcpy(root_override, a6, sizeof root_override);
}
static void root_acceleration(const double g[3], double a_root[6])
{
   if (root_override_set)
      memcpy(a_root, root_override, 6 * sizeof(double));
   else
   {
      a_root[0] = a_root[1] = a_root[2] = 0.0;
      a_root[3] = -g[0], a_root[4] = -g[1], a_root[5] = -g[2];
   }
}

static void rnea_one(const mo_model *m, const double *q, const double *qd, const double *qdd, const double g[3], const double *fext,
                     int coriolis, int accel, double *tau)
{
   static _Thread_local mo_kin K;
   static _Thread_local double acc[MO_MAX_JOINTS][6], wrench[MO_MAX_JOINTS][6];
   xf_t W_world, T;
   double a_root[6]; /* :343-348, :413-427 */
   root_acceleration(g, a_root);
   double zero6[6] = {0};
   xf_identity(&W_world);
   kinematics(m, q, qd, &K);

   for (int i = 0; i < m->n; i++)
   {
      int p = m->parent[i];
      const xf_t *Wp_body = p < 0 ? &W_world : &K.W_body[p];
      const double *a_par = p < 0 ? a_root : acc[p];
      const double *tw_par = p < 0 ? zero6 : K.tw_body[p];
      double a[6];
      memcpy(a, a_par, sizeof a);
      if (coriolis)
      {
         /* predecessor twist = -(joint twist) re-expressed in the predecessor body frame (JointReadOnly.java:270-281) */
         double d[6], c1[3], c2[3], c3[3];
         xf_between(&K.W_after[i], Wp_body, &T);
         xf_motion(&T, K.vJ[i], d);
         for (int k = 0; k < 6; k++)
            d[k] = -d[k];
         /* non-flipped branch of SpatialAccelerationBasics.java:192-200 */
         v3_cross(d + 3, tw_par, c1);     /* v_delta x w_body     */
         v3_cross(d, tw_par + 3, c2);     /* w_delta x v_body     */
         v3_cross(d, tw_par, c3);         /* w_delta x w_body     */
         for (int k = 0; k < 3; k++)
            a[3 + k] += c1[k] + c2[k], a[k] += c3[k];
      }
      xf_between(Wp_body, &K.W_body[i], &T);
      xf_motion(&T, a, acc[i]);
      if (accel)
      {
         double aJ[6], aJb[6];
         joint_S_times(m, i, K.S[i], qdd, aJ); /* :898-899 */
         xf_between(&K.W_after[i], &K.W_body[i], &T);
         xf_motion(&T, aJ, aJb);               /* :907 */
         for (int k = 0; k < 6; k++)
            acc[i][k] += aJb[k];
      }
   }
   for (int i = 0; i < m->n; i++)
   {
      if (tap_acc)
         memcpy(tap_acc + 6 * i, acc[i], 6 * sizeof(double));
      if (tap_twist)
         memcpy(tap_twist + 6 * i, K.tw_body[i], 6 * sizeof(double));
   }
   for (int i = m->n - 1; i >= 0; i--)
   {
      double w[6];
      dynamic_wrench(m->J[i], m->mass[i], m->com[i], acc[i], coriolis ? K.tw_body[i] : NULL, w); /* :935-944 */
      if (fext)
         for (int k = 0; k < 6; k++)
            w[k] -= fext[6 * i + k]; /* :946 */
      xf_between(&K.W_body[i], &K.W_after[i], &T);
      xf_force(&T, w, wrench[i]); /* :947 */
   }
   /* children first (passTwoRecursive :920-928): descending index visits every child before its parent */
   for (int i = m->n - 1; i >= 0; i--)
   {
      const int *di = m->dof_idx + m->dof_ofs[i];
      for (int d = 0; d < m->ndof[i]; d++)
      {
         double s = 0;
         for (int k = 0; k < 6; k++)
            s += K.S[i][d][k] * wrench[i][k]; /* :952-953 */
         tau[di[d]] = s;
      }
      int p = m->parent[i];
      if (p >= 0)
      {
         double w[6];
         xf_between(&K.W_after[i], &K.W_after[p], &T);
         xf_force(&T, wrench[i], w); /* :961-966 */
         for (int k = 0; k < 6; k++)
            wrench[p][k] += w[k];
      }
   }
   if (tap_wrench)
      for (int i = 0; i < m->n; i++)
         memcpy(tap_wrench + 6 * i, wrench[i], 6 * sizeof(double));
}

/* ================================================================== articulated-body inertia as (A, L, C) blocks
 * 6x6 = [[A, C], [C^T, L]]  (ArticulatedBodyInertia.java:42-55,312-319) */
typedef struct
{
   double A[9], L[9], C[9];
} abi_t;

/* ArticulatedBodyInertia.java:176-186 : rigid inertia (J, m, c) -> A = J, L = m 1, C = m [c]x */
static void abi_from_rigid(const double J[9], double mass, const double c[3], abi_t *I)
{
   memcpy(I->A, J, 9 * sizeof(double));
   memset(I->L, 0, sizeof I->L);
   I->L[0] = I->L[4] = I->L[8] = mass;
   m3_tilde(c, I->C);
   for (int k = 0; k < 9; k++)
      I->C[k] *= mass;
}
/* ArticulatedBodyInertia.java:359-375: rotate the three blocks, then translate.
 * Translation (ArticulatedBodyInertiaAlorigthmTools.java:83-101,150-162), in matrix form with P = [p]x :
 *   A' = A + P C^T - C P - P L P ,   C' = C + P L ,   L' = L                                           */
static void abi_apply_transform(const xf_t *X, abi_t *I)
{
   double P[9], Ct[9], t1[9], t2[9], t3[9], PL[9];
   m3_conj(X->R, I->A);
   m3_conj(X->R, I->L);
   m3_conj(X->R, I->C);
   m3_tilde(X->p, P);
   m3_transpose(I->C, Ct);
   m3_mul(P, Ct, t1);
   m3_mul(I->C, P, t2);
   m3_mul(P, I->L, PL);
   m3_mul(PL, P, t3);
   for (int k = 0; k < 9; k++)
   {
      I->A[k] += t1[k] - t2[k] - t3[k];
      I->C[k] += PL[k];
   }
}
static void abi_to_dense(const abi_t *I, double M[36])
{
   for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
      {
         M[6 * r + c] = I->A[3 * r + c];
         M[6 * r + 3 + c] = I->C[3 * r + c];
         M[6 * (3 + r) + c] = I->C[3 * c + r];
         M[6 * (3 + r) + 3 + c] = I->L[3 * r + c];
      }
}
static void abi_sub_dense(abi_t *I, const double M[36]) /* ArticulatedBodyInertia.java:219-241 */
{
   for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
      {
         I->A[3 * r + c] -= M[6 * r + c];
         I->C[3 * r + c] -= M[6 * r + 3 + c];
         I->L[3 * r + c] -= M[6 * (3 + r) + 3 + c];
      }
}
static void abi_mulv(const abi_t *I, const double x[6], double out[6])
{
   double M[36];
   abi_to_dense(I, M);
   for (int r = 0; r < 6; r++)
   {
      double s = 0;
      for (int c = 0; c < 6; c++)
         s += M[6 * r + c] * x[c];
      out[r] = s;
   }
}

/* rigid inertia (J, m, c) algebra used by CRBA and to bring a body inertia to the afterJoint frame
 * SpatialInertiaBasics.java:222-239: rotate J and c, then parallel axis J -= m ([p]x[c]x + [c]x[p]x + [p]x[p]x), c += p
 * (tools/MecanoTools.java:449-547) */
typedef struct
{
   double J[9], m, c[3];
} rigid_t;
static void rigid_apply_transform(const xf_t *X, rigid_t *I)
{
   double P[9], Cx[9], t1[9], t2[9], t3[9], c2[3];
   m3_conj(X->R, I->J);
   m3_mulv(X->R, I->c, c2);
   m3_tilde(X->p, P);
   m3_tilde(c2, Cx);
   m3_mul(P, Cx, t1);
   m3_mul(Cx, P, t2);
   m3_mul(P, P, t3);
   for (int k = 0; k < 9; k++)
      I->J[k] -= I->m * (t1[k] + t2[k] + t3[k]);
   for (int k = 0; k < 3; k++)
      I->c[k] = c2[k] + X->p[k];
}
static void rigid_add(rigid_t *I, const rigid_t *o) /* FixedFrameSpatialInertiaBasics.java:167-176 */
{
   for (int k = 0; k < 9; k++)
      I->J[k] += o->J[k];
   for (int k = 0; k < 3; k++)
      I->c[k] = I->c[k] * I->m + o->m * o->c[k];
   I->m += o->m;
   if (fabs(I->m) >= 1.0e-7)
      for (int k = 0; k < 3; k++)
         I->c[k] *= 1.0 / I->m;
}
/* momentum = I * twist  (SpatialInertiaReadOnly.java:334-357) */
static void rigid_mulv(const rigid_t *I, const double x[6], double out[6])
{
   if (v3_dot(I->c, I->c) < COM_OFFSET_ZERO_EPSILON)
   {
      m3_mulv(I->J, x, out);
      for (int k = 0; k < 3; k++)
         out[3 + k] = I->m * x[3 + k];
   }
   else
   {
      double cxv[3], Jw[3], wxc[3];
      v3_cross(I->c, x + 3, cxv);
      m3_mulv(I->J, x, Jw);
      v3_cross(x, I->c, wxc);
      for (int k = 0; k < 3; k++)
      {
         out[k] = I->m * cxv[k] + Jw[k];
         out[3 + k] = I->m * (wxc[k] + x[3 + k]);
      }
   }
}

/* symmetric positive definite inverse by Cholesky (EJML LinearSolverFactory_DDRM.symmPosDef, ForwardDynamicsCalculator.java:1040,1195-1196;
 * for 2..5 DoFs the reference uses a minor-based unrolled inverse, the same matrix up to rounding) */
static int spd_inverse(int n, const double *D, double *Dinv)
{
   double Lc[36] = {0};
   for (int i = 0; i < n; i++)
      for (int j = 0; j <= i; j++)
      {
         double s = D[i * n + j];
         for (int k = 0; k < j; k++)
            s -= Lc[i * n + k] * Lc[j * n + k];
         if (i == j)
         {
            if (s <= 0)
               return 1;
            Lc[i * n + i] = sqrt(s);
         }
         else
            Lc[i * n + j] = s / Lc[j * n + j];
      }
   for (int c = 0; c < n; c++)
   {
      double y[6], x[6];
      for (int i = 0; i < n; i++)
      {
         double s = (i == c) ? 1.0 : 0.0;
         for (int k = 0; k < i; k++)
            s -= Lc[i * n + k] * y[k];
         y[i] = s / Lc[i * n + i];
      }
      for (int i = n - 1; i >= 0; i--)
      {
         double s = y[i];
         for (int k = i + 1; k < n; k++)
            s -= Lc[k * n + i] * x[k];
         x[i] = s / Lc[i * n + i];
      }
      for (int i = 0; i < n; i++)
         Dinv[i * n + c] = x[i];
   }
   return 0;
}

/* ================================================================== ABA
 * ForwardDynamicsCalculator.java:1085-1127 (passOne), :1136-1254 (passTwo), :1259-1310 (passThree) */
/* locked: NULL or one flag per joint (JointSourceMode.ACCELERATION_SOURCE, ForwardDynamicsCalculator.java:45-57); qdd_in gives the
 * accelerations of the locked joints; tau_out (NULL or nv values) receives tau of every joint: the input for EFFORT_SOURCE joints, the
 * computed effort for ACCELERATION_SOURCE joints (pass four, :1315-1363). */
static int aba_one(const mo_model *m, const double *q, const double *qd, const double *tau, const double g[3], const double *fext, double *qdd,
                   const int *locked, const double *qdd_in, double *tau_out)
{
   static _Thread_local mo_kin K;
   static _Thread_local xf_t Xup[MO_MAX_JOINTS]; /* afterJoint_i -> afterJoint_parent (or root body frame) :1096-1099 */
   static _Thread_local double pb[MO_MAX_JOINTS][6], cb[MO_MAX_JOINTS][6], pA[MO_MAX_JOINTS][6];
   static _Thread_local abi_t IA[MO_MAX_JOINTS];
   static _Thread_local double U[MO_MAX_JOINTS][6][6], Dinv[MO_MAX_JOINTS][36], u[MO_MAX_JOINTS][6], acc[MO_MAX_JOINTS][6];
   xf_t W_world, T;
   xf_identity(&W_world);
   kinematics(m, q, qd, &K);

   /* ---- pass one */
   for (int i = 0; i < m->n; i++)
   {
      int p = m->parent[i];
      double w[6], c1[3], c2[3], c3[3];
      xf_between(&K.W_after[i], p < 0 ? &W_world : &K.W_after[p], &Xup[i]);
      dynamic_wrench(m->J[i], m->mass[i], m->com[i], NULL, K.tw_body[i], w); /* :1110 */
      if (fext)
         for (int k = 0; k < 6; k++)
            w[k] -= fext[6 * i + k];
      xf_between(&K.W_body[i], &K.W_after[i], &T);
      xf_force(&T, w, pb[i]); /* :1112 */
      /* bias acceleration, flipped branch of SpatialAccelerationBasics.java:204-217 with delta = joint twist */
      v3_cross(K.tw_after[i], K.vJ[i] + 3, c1);     /* w_body x v_J */
      v3_cross(K.tw_after[i] + 3, K.vJ[i], c2);     /* v_body x w_J */
      v3_cross(K.tw_after[i], K.vJ[i], c3);         /* w_body x w_J */
      for (int k = 0; k < 3; k++)
         cb[i][k] = c3[k], cb[i][3 + k] = c1[k] + c2[k];
   }
   /* ---- pass two: leaves to root */
   for (int i = 0; i < m->n; i++)
   {
      rigid_t I;
      memcpy(I.J, m->J[i], sizeof I.J);
      I.m = m->mass[i];
      memcpy(I.c, m->com[i], sizeof I.c);
      xf_between(&K.W_body[i], &K.W_after[i], &T);
      rigid_apply_transform(&T, &I); /* spatialInertia.changeFrame(frameAfterJoint) :1149 */
      abi_from_rigid(I.J, I.m, I.c, &IA[i]);
      memcpy(pA[i], pb[i], sizeof pA[i]);
   }
   for (int i = m->n - 1; i >= 0; i--)
   {
      int p = m->parent[i], nd = m->ndof[i];
      const int *di = m->dof_idx + m->dof_ofs[i];
      double D[36];
      if (locked && locked[i])
      { /* ACCELERATION_SOURCE (:1237-1253): Ia = IA ; pa = pA + IA (c + S qdd_given) */
         if (p >= 0)
         {
            double aJ[6], ca[6], Iac[6], pa[6], paP[6];
            joint_S_times(m, i, K.S[i], qdd_in, aJ);
            for (int k = 0; k < 6; k++)
               ca[k] = cb[i][k] + aJ[k];
            abi_mulv(&IA[i], ca, Iac);
            for (int k = 0; k < 6; k++)
               pa[k] = pA[i][k] + Iac[k];
            abi_t Ia = IA[i];
            abi_apply_transform(&Xup[i], &Ia);
            xf_force(&Xup[i], pa, paP);
            for (int k = 0; k < 9; k++)
               IA[p].A[k] += Ia.A[k], IA[p].L[k] += Ia.L[k], IA[p].C[k] += Ia.C[k];
            for (int k = 0; k < 6; k++)
               pA[p][k] += paP[k];
         }
         continue;
      }
      /* U = IA S (:1177) ; D = S^T U (:1179) */
      for (int d = 0; d < nd; d++)
         abi_mulv(&IA[i], K.S[i][d], U[i][d]);
      for (int a = 0; a < nd; a++)
         for (int b = 0; b < nd; b++)
         {
            double s = 0;
            for (int k = 0; k < 6; k++)
               s += K.S[i][a][k] * U[i][b][k];
            D[a * nd + b] = s;
         }
      if (nd == 1)
         Dinv[i][0] = 1.0 / D[0]; /* :1183 */
      else if (nd > 1 && spd_inverse(nd, D, Dinv[i]))
         return 1;
      /* u = tau - S^T pA (:1200-1215) */
      for (int d = 0; d < nd; d++)
      {
         double s = 0;
         for (int k = 0; k < 6; k++)
            s += K.S[i][d][k] * pA[i][k];
         u[i][d] = tau[di[d]] - s;
      }
      if (p >= 0)
      {
         /* Ia = IA - U Dinv U^T (:1220-1226) ; pa = pA + Ia c + U Dinv u (:1229-1234) */
         double UD[6][6] = {{0}}, M[36], pa[6], Iac[6];
         abi_t Ia = IA[i];
         for (int r = 0; r < 6; r++)
            for (int b = 0; b < nd; b++)
            {
               double s = 0;
               for (int a = 0; a < nd; a++)
                  s += U[i][a][r] * Dinv[i][a * nd + b];
               UD[b][r] = s;
            }
         for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++)
            {
               double s = 0;
               for (int b = 0; b < nd; b++)
                  s += UD[b][r] * U[i][b][c];
               M[6 * r + c] = s;
            }
         abi_sub_dense(&Ia, M);
         abi_mulv(&Ia, cb[i], Iac);
         for (int r = 0; r < 6; r++)
         {
            double s = pA[i][r] + Iac[r];
            for (int b = 0; b < nd; b++)
               s += UD[b][r] * u[i][b];
            pa[r] = s;
         }
         /* hand over to the parent (:1156-1166) */
         double paP[6];
         abi_apply_transform(&Xup[i], &Ia);
         xf_force(&Xup[i], pa, paP);
         for (int k = 0; k < 9; k++)
            IA[p].A[k] += Ia.A[k], IA[p].L[k] += Ia.L[k], IA[p].C[k] += Ia.C[k];
         for (int k = 0; k < 6; k++)
            pA[p][k] += paP[k];
      }
   }
   /* ---- pass three: root to leaves */
   double a_root[6]; /* :259-264, :330-343 */
   root_acceleration(g, a_root);
   for (int i = 0; i < m->n; i++)
   {
      int p = m->parent[i], nd = m->ndof[i];
      const int *di = m->dof_idx + m->dof_ofs[i];
      double ap[6], r[6], qddj[6];
      xf_motion_inv(&Xup[i], p < 0 ? a_root : acc[p], ap); /* :1270-1271 */
      for (int k = 0; k < 6; k++)
         ap[k] += cb[i][k]; /* :1273 */
      if (locked && locked[i])
      { /* :1284-1297 */
         for (int a = 0; a < nd; a++)
         {
            qddj[a] = qdd_in[di[a]];
            qdd[di[a]] = qddj[a];
         }
      }
      else
      {
         for (int d = 0; d < nd; d++)
         {
            double s = 0;
            for (int k = 0; k < 6; k++)
               s += U[i][d][k] * ap[k];
            r[d] = u[i][d] - s; /* :1280-1281 */
         }
         for (int a = 0; a < nd; a++)
         {
            double s = 0;
            for (int b = 0; b < nd; b++)
               s += Dinv[i][a * nd + b] * r[b];
            qddj[a] = s; /* :1282 */
            qdd[di[a]] = s;
         }
      }
      memcpy(acc[i], ap, sizeof ap);
      for (int d = 0; d < nd; d++)
         for (int k = 0; k < 6; k++)
            acc[i][k] += K.S[i][d][k] * qddj[d]; /* :1300-1305 */
   }
   for (int i = 0; i < m->n && (tap_acc || tap_twist); i++)
   {
      if (tap_acc)
      { /* :170-180 rigidBodyAcceleration.changeFrame(bodyFixedFrame) */
         xf_between(&K.W_after[i], &K.W_body[i], &T);
         xf_motion(&T, acc[i], tap_acc + 6 * i);
      }
      if (tap_twist)
         memcpy(tap_twist + 6 * i, K.tw_body[i], 6 * sizeof(double));
   }
   if (tau_out)
   {
      /* ---- pass four (:1315-1363): joint wrenches RNEA-style from the accelerations of pass three; tau = S^T wrench for the locked joints */
      static _Thread_local double jw[MO_MAX_JOINTS][6];
      for (int i = m->n - 1; i >= 0; i--)
      {
         double ab[6], w[6], wa[6];
         xf_between(&K.W_after[i], &K.W_body[i], &T);
         xf_motion(&T, acc[i], ab); /* rigidBodyAcceleration.changeFrame(bodyFixedFrame) :1339 */
         xf_between(&K.W_body[i], &K.W_after[i], &T);
         if (v3_dot(m->com[i], m->com[i]) < COM_OFFSET_ZERO_EPSILON)
         { /* velocity terms decouple: add the bias wrench of pass one (:1342-1348) */
            dynamic_wrench(m->J[i], m->mass[i], m->com[i], ab, NULL, w);
            xf_force(&T, w, wa);
            for (int k = 0; k < 6; k++)
               jw[i][k] = wa[k] + pb[i][k];
         }
         else
         { /* :1349-1355 */
            dynamic_wrench(m->J[i], m->mass[i], m->com[i], ab, K.tw_body[i], w);
            if (fext)
               for (int k = 0; k < 6; k++)
                  w[k] -= fext[6 * i + k];
            xf_force(&T, w, jw[i]);
         }
      }
      for (int i = m->n - 1; i >= 0; i--)
      {
         const int *di = m->dof_idx + m->dof_ofs[i];
         for (int d = 0; d < m->ndof[i]; d++)
         {
            if (locked && locked[i])
            {
               double s = 0;
               for (int k = 0; k < 6; k++)
                  s += K.S[i][d][k] * jw[i][k];
               tau_out[di[d]] = s;
            }
            else
               tau_out[di[d]] = tau[di[d]];
         }
         int p = m->parent[i];
         if (p >= 0)
         {
            double w[6];
            xf_force(&Xup[i], jw[i], w);
            for (int k = 0; k < 6; k++)
               jw[p][k] += w[k];
         }
      }
   }
   return 0;
}

/* ================================================================== CRBA
 * CompositeRigidBodyMassMatrixCalculator.java:588-707 + ancestor walk :770-798 */
static void crba_one(const mo_model *m, const double *q, double *H)
{
   static _Thread_local mo_kin K;
   static _Thread_local xf_t Xup[MO_MAX_JOINTS];
   static _Thread_local rigid_t Ic[MO_MAX_JOINTS];
   xf_t W_world, T;
   xf_identity(&W_world);
   kinematics(m, q, NULL, &K);
   memset(H, 0, sizeof(double) * (size_t)m->nv * (size_t)m->nv); /* :298 */
   for (int i = 0; i < m->n; i++)
   {
      int p = m->parent[i];
      if (p >= 0)
         xf_between(&K.W_after[i], &K.W_after[p], &Xup[i]); /* :592-593 */
      else
         xf_identity(&Xup[i]);
      memcpy(Ic[i].J, m->J[i], sizeof Ic[i].J);
      Ic[i].m = m->mass[i];
      memcpy(Ic[i].c, m->com[i], sizeof Ic[i].c);
      xf_between(&K.W_body[i], &K.W_after[i], &T);
      rigid_apply_transform(&T, &Ic[i]); /* :646-650 */
   }
   for (int i = m->n - 1; i >= 0; i--)
   {
      int nd = m->ndof[i];
      const int *di = m->dof_idx + m->dof_ofs[i];
      double F[6][6];
      for (int d = 0; d < nd; d++)
         rigid_mulv(&Ic[i], K.S[i][d], F[d]); /* :663-667 */
      for (int a = 0; a < nd; a++)
         for (int b = 0; b < nd; b++)
         {
            double s = 0;
            for (int k = 0; k < 6; k++)
               s += K.S[i][a][k] * F[b][k];
            H[(size_t)di[a] * m->nv + di[b]] = s; /* :700-707 (setSymmetricEntry :841-845) */
            H[(size_t)di[b] * m->nv + di[a]] = s;
         }
      /* climb the ancestors, re-expressing F on the way (:783-792) */
      int prev = i, anc = m->parent[i];
      while (anc >= 0)
      {
         const int *dj = m->dof_idx + m->dof_ofs[anc];
         for (int b = 0; b < nd; b++)
         {
            double Fn[6];
            xf_force(&Xup[prev], F[b], Fn);
            memcpy(F[b], Fn, sizeof Fn);
            for (int a = 0; a < m->ndof[anc]; a++)
            {
               double s = 0;
               for (int k = 0; k < 6; k++)
                  s += K.S[anc][a][k] * F[b][k];
               H[(size_t)dj[a] * m->nv + di[b]] = s;
               H[(size_t)di[b] * m->nv + dj[a]] = s;
            }
         }
         prev = anc;
         anc = m->parent[anc];
      }
      /* composite inertia to the parent (:651-661) */
      int p = m->parent[i];
      if (p >= 0)
      {
         rigid_t child = Ic[i];
         rigid_apply_transform(&Xup[i], &child);
         rigid_add(&Ic[p], &child);
      }
   }
}

/* ================================================================== Coriolis matrix, centroidal momentum
 * algorithms/FactorizedBodyInertia.java: B = v x* I kept as four 3x3 blocks (angular, linear, top-right, bottom-left) */
typedef struct
{
   double A[9], L[9], TR[9], BL[9];
} fbi_t;
/* FactorizedBodyInertia.setIncludingFrame(SpatialInertiaReadOnly, TwistReadOnly), FactorizedBodyInertia.java:136-158 */
static void fbi_from_rigid(const rigid_t *I, const double tw[6], fbi_t *B)
{
   double Wx[9], Vx[9], Cx[9], t[9];
   m3_tilde(tw, Wx);
   m3_tilde(tw + 3, Vx);
   m3_tilde(I->c, Cx);
   m3_mul(Vx, Cx, B->A); /* w x J - m v x c x */
   m3_mul(Wx, I->J, t);
   for (int k = 0; k < 9; k++)
      B->A[k] = -I->m * B->A[k] + t[k];
   m3_mul(Wx, Cx, B->BL); /* -m w x c x */
   for (int k = 0; k < 9; k++)
      B->BL[k] *= -I->m;
   for (int k = 0; k < 9; k++)
      B->TR[k] = I->m * Vx[k] - B->BL[k]; /* m v x + m w x c x */
   for (int k = 0; k < 9; k++)
      B->L[k] = I->m * Wx[k]; /* m w x */
}
static void fbi_add(fbi_t *B, const fbi_t *o)
{
   for (int k = 0; k < 9; k++)
      B->A[k] += o->A[k], B->L[k] += o->L[k], B->TR[k] += o->TR[k], B->BL[k] += o->BL[k];
}
/* FactorizedBodyInertia.applyTransform(RigidBodyTransform), :314-331: rotate the four blocks, then the four in-place translation updates */
static void fbi_apply_transform(const xf_t *X, fbi_t *B)
{
   double P[9], t[9];
   m3_conj(X->R, B->A);
   m3_conj(X->R, B->L);
   m3_conj(X->R, B->TR);
   m3_conj(X->R, B->BL);
   m3_tilde(X->p, P);
   m3_mul(P, B->L, t);
   for (int k = 0; k < 9; k++)
      B->TR[k] += t[k];
   m3_mul(P, B->BL, t);
   for (int k = 0; k < 9; k++)
      B->A[k] += t[k];
   m3_mul(B->TR, P, t);
   for (int k = 0; k < 9; k++)
      B->A[k] -= t[k];
   m3_mul(B->L, P, t);
   for (int k = 0; k < 9; k++)
      B->BL[k] -= t[k];
}
/* out (+)= B x, :239-263 */
static void fbi_mulv_add(const fbi_t *B, const double x[6], double out[6])
{
   double a[3], b[3];
   m3_mulv(B->A, x, a), m3_mulv(B->TR, x + 3, b);
   for (int k = 0; k < 3; k++)
      out[k] += a[k] + b[k];
   m3_mulv(B->BL, x, a), m3_mulv(B->L, x + 3, b);
   for (int k = 0; k < 3; k++)
      out[3 + k] += a[k] + b[k];
}
/* out = B^T x, :265-276 */
static void fbi_tmulv(const fbi_t *B, const double x[6], double out[6])
{
   double a[3], b[3];
   m3_tmulv(B->A, x, a), m3_tmulv(B->BL, x + 3, b);
   for (int k = 0; k < 3; k++)
      out[k] = a[k] + b[k];
   m3_tmulv(B->TR, x, a), m3_tmulv(B->L, x + 3, b);
   for (int k = 0; k < 3; k++)
      out[3 + k] = a[k] + b[k];
}
static double dot6(const double a[6], const double b[6])
{
   double s = 0;
   for (int k = 0; k < 6; k++)
      s += a[k] * b[k];
   return s;
}

/* Mass matrix, Coriolis matrix and (optionally) the centroidal momentum matrix in one sweep, with the Coriolis calculation enabled:
 * CompositeRigidBodyMassMatrixCalculator.java:588-630 (unit twist derivatives), :642-667 (composite inertia), :669-692 (factorised
 * composite inertia, F1 / F2 / F3), :698-724 (own block), :729-768 (ancestor walk), :801-809 (centroidal momentum matrix = the
 * climbed F2 changed to the centroidal frame).  W_cm = pose of the centroidal momentum frame in the root body ("world") frame.
 * C, Acm may be NULL. */
static void crba_coriolis_one(const mo_model *m, const double *q, const double *qd, const xf_t *W_cm, double *H, double *C, double *Acm)
{
   static _Thread_local mo_kin K;
   static _Thread_local xf_t Xup[MO_MAX_JOINTS];
   static _Thread_local rigid_t Ib[MO_MAX_JOINTS], Ic[MO_MAX_JOINTS];
   static _Thread_local fbi_t Bc[MO_MAX_JOINTS];
   static _Thread_local double Sd[MO_MAX_JOINTS][6][6];
   xf_t T;
   const size_t nv = (size_t)m->nv;
   kinematics(m, q, qd, &K);
   memset(H, 0, sizeof(double) * nv * nv); /* :298-300 */
   if (C)
      memset(C, 0, sizeof(double) * nv * nv);
   if (Acm)
      memset(Acm, 0, sizeof(double) * 6 * nv);
   for (int i = 0; i < m->n; i++)
   {
      int p = m->parent[i];
      if (p >= 0)
         xf_between(&K.W_after[i], &K.W_after[p], &Xup[i]); /* :592-593 */
      else
         xf_identity(&Xup[i]);
      /* derivative of the unit twists of a constant motion subspace: body twist x S (:620-626) */
      const double *tw = K.tw_after[i];
      for (int d = 0; d < m->ndof[i]; d++)
      {
         const double *S = K.S[i][d];
         double c1[3], c2[3];
         v3_cross(tw, S, Sd[i][d]);
         v3_cross(tw + 3, S, c1);
         v3_cross(tw, S + 3, c2);
         for (int k = 0; k < 3; k++)
            Sd[i][d][3 + k] = c1[k] + c2[k];
      }
      memcpy(Ib[i].J, m->J[i], sizeof Ib[i].J);
      Ib[i].m = m->mass[i];
      memcpy(Ib[i].c, m->com[i], sizeof Ib[i].c);
      xf_between(&K.W_body[i], &K.W_after[i], &T);
      rigid_apply_transform(&T, &Ib[i]); /* :646-650 */
      Ic[i] = Ib[i];
      fbi_from_rigid(&Ib[i], tw, &Bc[i]); /* :671-673 */
   }
   for (int i = m->n - 1; i >= 0; i--)
   { /* descending index = children before their parent; Ic[i], Bc[i] already hold the children's contributions */
      int nd = m->ndof[i];
      const int *di = m->dof_idx + m->dof_ofs[i];
      double F1[6][6], F2[6][6], F3[6][6];
      for (int d = 0; d < nd; d++)
      {
         rigid_mulv(&Ic[i], K.S[i][d], F2[d]); /* :663-667 */
         rigid_mulv(&Ic[i], Sd[i][d], F1[d]);  /* :686-688 */
         fbi_mulv_add(&Bc[i], K.S[i][d], F1[d]);
         fbi_tmulv(&Bc[i], K.S[i][d], F3[d]); /* :690-691 */
      }
      for (int a = 0; a < nd; a++)
         for (int b = 0; b < nd; b++)
         {
            double s = dot6(K.S[i][a], F2[b]);
            H[(size_t)di[a] * nv + di[b]] = s; /* :698-707 */
            H[(size_t)di[b] * nv + di[a]] = s;
         }
      if (C)
         for (int a = 0; a < nd; a++)
            for (int b = 0; b < nd; b++)
            { /* :709-724, in the reference's write order */
               C[(size_t)di[a] * nv + di[b]] = dot6(K.S[i][a], F1[b]);
               if (a != b)
                  C[(size_t)di[b] * nv + di[a]] = dot6(Sd[i][a], F2[b]) + dot6(K.S[i][a], F3[b]);
            }
      int prev = i, anc = m->parent[i];
      while (anc >= 0)
      { /* :729-768 */
         const int *dj = m->dof_idx + m->dof_ofs[anc];
         for (int b = 0; b < nd; b++)
         {
            double t[6];
            xf_force(&Xup[prev], F1[b], t), memcpy(F1[b], t, sizeof t);
            xf_force(&Xup[prev], F2[b], t), memcpy(F2[b], t, sizeof t);
            xf_force(&Xup[prev], F3[b], t), memcpy(F3[b], t, sizeof t);
            for (int a = 0; a < m->ndof[anc]; a++)
            {
               double s = dot6(K.S[anc][a], F2[b]);
               H[(size_t)dj[a] * nv + di[b]] = s;
               H[(size_t)di[b] * nv + dj[a]] = s;
               if (C)
               {
                  C[(size_t)dj[a] * nv + di[b]] = dot6(K.S[anc][a], F1[b]);
                  C[(size_t)di[b] * nv + dj[a]] = dot6(Sd[anc][a], F2[b]) + dot6(K.S[anc][a], F3[b]);
               }
            }
         }
         prev = anc;
         anc = m->parent[anc];
      }
      if (Acm)
      { /* :801-809: F2 now lives in the frame after the root-most ancestor joint `prev` */
         xf_between(&K.W_after[prev], W_cm, &T);
         for (int b = 0; b < nd; b++)
         {
            double t[6];
            xf_force(&T, F2[b], t);
            for (int k = 0; k < 6; k++)
               Acm[(size_t)k * nv + di[b]] = t[k];
         }
      }
      int p = m->parent[i];
      if (p >= 0)
      {
         rigid_t child = Ic[i];
         fbi_t childB = Bc[i];
         rigid_apply_transform(&Xup[i], &child); /* :651-661 */
         rigid_add(&Ic[p], &child);
         fbi_apply_transform(&Xup[i], &childB); /* :675-683 */
         fbi_add(&Bc[p], &childB);
      }
   }
}

/* centre of mass of the considered bodies in the root frame (algorithms/CenterOfMassCalculator.java:70-91) */
static void center_of_mass(const mo_model *m, const mo_kin *K, double com[3], double *mass_out)
{
   double M = 0;
   com[0] = com[1] = com[2] = 0;
   for (int i = 0; i < m->n; i++)
   {
      double c[3];
      m3_mulv(K->W_body[i].R, m->com[i], c);
      for (int k = 0; k < 3; k++)
         com[k] += m->mass[i] * (c[k] + K->W_body[i].p[k]);
      M += m->mass[i];
   }
   for (int k = 0; k < 3; k++)
      com[k] *= 1.0 / M;
   if (mass_out)
      *mass_out = M;
}

/* Centroidal convective term b (d/dt h = A qdd + b): CompositeRigidBodyMassMatrixCalculator.java:811-839 -- the Coriolis body
 * accelerations of InverseDynamicsCalculator's first pass with zero root and joint accelerations, each body's dynamic wrench changed
 * to the centroidal frame and summed */
static void convective_one(const mo_model *m, const double *q, const double *qd, const xf_t *W_cm, double b[6])
{
   static _Thread_local mo_kin K;
   static _Thread_local double acc[MO_MAX_JOINTS][6];
   xf_t W_world, T;
   double zero6[6] = {0};
   xf_identity(&W_world);
   kinematics(m, q, qd, &K);
   memset(b, 0, 6 * sizeof(double));
   for (int i = 0; i < m->n; i++)
   {
      int p = m->parent[i];
      const xf_t *Wp_body = p < 0 ? &W_world : &K.W_body[p];
      const double *a_par = p < 0 ? zero6 : acc[p];
      const double *tw_par = p < 0 ? zero6 : K.tw_body[p];
      double a[6], d[6], c1[3], c2[3], c3[3], w[6], wc[6];
      memcpy(a, a_par, sizeof a);
      xf_between(&K.W_after[i], Wp_body, &T);
      xf_motion(&T, K.vJ[i], d);
      for (int k = 0; k < 6; k++)
         d[k] = -d[k];
      v3_cross(d + 3, tw_par, c1);
      v3_cross(d, tw_par + 3, c2);
      v3_cross(d, tw_par, c3);
      for (int k = 0; k < 3; k++)
         a[3 + k] += c1[k] + c2[k], a[k] += c3[k];
      xf_between(Wp_body, &K.W_body[i], &T);
      xf_motion(&T, a, acc[i]); /* :828-831 */
      dynamic_wrench(m->J[i], m->mass[i], m->com[i], acc[i], K.tw_body[i], w); /* :833 */
      xf_between(&K.W_body[i], W_cm, &T);
      xf_force(&T, w, wc); /* :834-835 */
      for (int k = 0; k < 6; k++)
         b[k] += wc[k];
   }
}

/* ================================================================== public C API (ctypes / bench) */
static int joint_ndof(int type) { return type == MO_SIXDOF ? 6 : (type == MO_FIXED ? 0 : (type == MO_PLANAR || type == MO_SPHERICAL ? 3 : 1)); }
static int joint_ncfg(int type) { return type == MO_SIXDOF ? 7 : (type == MO_FIXED ? 0 : (type == MO_PLANAR ? 3 : (type == MO_SPHERICAL ? 4 : 1))); }

/* Joints must be listed parents-first (Mecano's default DFS pre-order is; IT/JointIterator.java:155-161). */
void *mo_model_create(int n, int nq, int nv, const int *parent, const int *type, const double *axis, const double *X_before,
                      const double *X_com, const double *J, const double *mass, const double *com, const int *dof_indices,
                      const int *cfg_indices)
{
   if (n <= 0 || n > MO_MAX_JOINTS)
      return NULL;
   mo_model *m = (mo_model *)calloc(1, sizeof *m);
   m->n = n, m->nq = nq, m->nv = nv;
   int dofs = 0, cfgs = 0;
   for (int i = 0; i < n; i++)
   {
      if (parent[i] >= i || type[i] < 0 || type[i] > MO_SPHERICAL)
      {
         free(m);
         return NULL;
      }
      m->parent[i] = parent[i], m->type[i] = type[i];
      m->ndof[i] = joint_ndof(type[i]), m->ncfg[i] = joint_ncfg(type[i]);
      m->dof_ofs[i] = dofs, m->cfg_ofs[i] = cfgs;
      dofs += m->ndof[i], cfgs += m->ncfg[i];
      memcpy(m->axis[i], axis + 3 * i, 3 * sizeof(double));
      memcpy(m->Xb[i].R, X_before + 12 * i, 9 * sizeof(double));
      memcpy(m->Xb[i].p, X_before + 12 * i + 9, 3 * sizeof(double));
      memcpy(m->Xcom[i].R, X_com + 12 * i, 9 * sizeof(double));
      memcpy(m->Xcom[i].p, X_com + 12 * i + 9, 3 * sizeof(double));
      memcpy(m->J[i], J + 9 * i, 9 * sizeof(double));
      m->mass[i] = mass[i];
      memcpy(m->com[i], com + 3 * i, 3 * sizeof(double));
   }
   m->dof_idx = (int *)malloc(sizeof(int) * (size_t)(dofs > 0 ? dofs : 1));
   m->cfg_idx = (int *)malloc(sizeof(int) * (size_t)(cfgs > 0 ? cfgs : 1));
   memcpy(m->dof_idx, dof_indices, sizeof(int) * (size_t)dofs);
   memcpy(m->cfg_idx, cfg_indices, sizeof(int) * (size_t)cfgs);
   return m;
}
void mo_model_destroy(void *h)
{
   mo_model *m = (mo_model *)h;
   if (!m)
      return;
   free(m->dof_idx);
   free(m->cfg_idx);
   free(m);
}
/* batched wrappers: AoS [B][n] matrices, fext [B][n_joints][6] or NULL */
void mo_rnea(void *h, long B, const double *q, const double *qd, const double *qdd, const double *g, const double *fext, int coriolis,
             int accel, double *tau)
{
   const mo_model *m = (const mo_model *)h;
   for (long b = 0; b < B; b++)
      rnea_one(m, q + b * m->nq, qd + b * m->nv, qdd ? qdd + b * m->nv : NULL, g, fext ? fext + b * 6 * m->n : NULL, coriolis,
               accel && qdd, tau + b * m->nv);
}
int mo_aba(void *h, long B, const double *q, const double *qd, const double *tau, const double *g, const double *fext, double *qdd)
{
   const mo_model *m = (const mo_model *)h;
   int rc = 0;
   for (long b = 0; b < B; b++)
      rc |= aba_one(m, q + b * m->nq, qd + b * m->nv, tau + b * m->nv, g, fext ? fext + b * 6 * m->n : NULL, qdd + b * m->nv, NULL, NULL, NULL);
   return rc;
}
/* ABA with per-joint source modes: locked[n_joints] flags, qdd_in [B][nv] (read for locked joints), tau_out [B][nv] or NULL */
int mo_aba_locked(void *h, long B, const double *q, const double *qd, const double *tau, const double *qdd_in, const double *g, const double *fext,
                  const int *locked, double *qdd, double *tau_out)
{
   const mo_model *m = (const mo_model *)h;
   int rc = 0;
   for (long b = 0; b < B; b++)
      rc |= aba_one(m, q + b * m->nq, qd + b * m->nv, tau + b * m->nv, g, fext ? fext + b * 6 * m->n : NULL, qdd + b * m->nv, locked,
                    qdd_in + b * m->nv, tau_out ? tau_out + b * m->nv : NULL);
   return rc;
}
void mo_crba(void *h, long B, const double *q, double *H)
{
   const mo_model *m = (const mo_model *)h;
   for (long b = 0; b < B; b++)
      crba_one(m, q + b * m->nq, H + (size_t)b * m->nv * m->nv);
}

/* H, C [B][nv][nv] (C may be NULL) */
void mo_crba_coriolis(void *h, long B, const double *q, const double *qd, double *H, double *C)
{
   const mo_model *m = (const mo_model *)h;
   xf_t W;
   xf_identity(&W);
   for (long b = 0; b < B; b++)
      crba_coriolis_one(m, q + b * m->nq, qd ? qd + b * m->nv : NULL, &W, H + (size_t)b * m->nv * m->nv, C ? C + (size_t)b * m->nv * m->nv : NULL, NULL);
}
/* Centroidal momentum matrix A [B][6][nv], convective term b [B][6] (may be NULL), frame origin com_out [B][3] (may be NULL).
 * frame12 = pose (R row-major, p) of the centroidal momentum frame in the root body frame, NULL = the root body frame itself
 * (the calculator's default, CompositeRigidBodyMassMatrixCalculator.java:190-193); at_com != 0 additionally moves the origin to the
 * centre of mass of the considered bodies given in that frame, i.e. a frames/CenterOfMassReferenceFrame whose parent is frame12. */
void mo_centroidal(void *h, long B, const double *q, const double *qd, const double *frame12, int at_com, double *A, double *bout, double *com_out)
{
   const mo_model *m = (const mo_model *)h;
   static _Thread_local mo_kin K;
   double *Htmp = (double *)malloc(sizeof(double) * (size_t)m->nv * (size_t)m->nv);
   for (long b = 0; b < B; b++)
   {
      xf_t W;
      xf_identity(&W);
      if (frame12)
      {
         memcpy(W.R, frame12, 9 * sizeof(double));
         memcpy(W.p, frame12 + 9, 3 * sizeof(double));
      }
      if (at_com)
      {
         double c[3], cf[3], d[3];
         kinematics(m, q + b * m->nq, NULL, &K);
         center_of_mass(m, &K, c, NULL);
         /* CoM in the parent frame: R^T (c - p); the CoM frame is that translation under the parent frame */
         for (int k = 0; k < 3; k++)
            d[k] = c[k] - W.p[k];
         m3_tmulv(W.R, d, cf);
         if (com_out)
            memcpy(com_out + 3 * b, cf, sizeof cf);
         memcpy(W.p, c, sizeof c);
      }
      else if (com_out)
         com_out[3 * b] = com_out[3 * b + 1] = com_out[3 * b + 2] = 0;
      crba_coriolis_one(m, q + b * m->nq, qd ? qd + b * m->nv : NULL, &W, Htmp, NULL, A + (size_t)b * 6 * m->nv);
      if (bout)
         convective_one(m, q + b * m->nq, qd + b * m->nv, &W, bout + 6 * b);
   }
   free(Htmp);
}

/* ------------------------------------------------------------------ state integration
 * tools/MultiBodySystemStateIntegrator.java: explicit constant-acceleration step.
 *   1-DoF   (:433-441, 710-733):  q' = q + dt qd + 0.5 dt^2 qdd ;  qd' = qd + dt qdd
 *   SixDoF  (:503-575): (w, v) = joint twist, (al, a) = joint acceleration, both in the frame after the joint
 *      a_o  = a + w x v                                  linear acceleration at the body origin (SpatialAccelerationReadOnly.java:197-204)
 *      dq   = quaternion of the rotation vector dt w + 0.5 dt^2 al
 *      w'   = w + dt al                                  (not re-expressed, :535)
 *      p'   = p + R(q) (dt v + 0.5 dt^2 a_o)             (:538-546)
 *      v'   = R(dq)^T (v + dt a_o)                       (:548-552)
 *      q'   = q * dq                                     (:554-559)
 *      al'  = al ; a' = R(dq)^T a_o + v' x w'            (:561-562, FixedFrameSpatialAccelerationBasics.java:81-90)
 * The rotation-vector -> quaternion conversion lives in Euclid 0.21.0 (un-vendored): the standard formula is used, with the
 * identity below |rv| = 1e-12 -- that threshold is an assumption (parity unpinned there).  Fixed joints are skipped (:403-404).
 */
static void quat_mul(const double a[4], const double b[4], double o[4])
{ /* Hamilton product, (x, y, z, s) */
   double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
   double y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
   double z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
   double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
   o[0] = x, o[1] = y, o[2] = z, o[3] = w;
}
static void quat_to_R(const double q[4], double R[9])
{
   double x = q[0], y = q[1], z = q[2], s = q[3];
   double nrm = sqrt(x * x + y * y + z * z + s * s);
   x /= nrm, y /= nrm, z /= nrm, s /= nrm;
   R[0] = 1 - 2 * (y * y + z * z), R[1] = 2 * (x * y - z * s), R[2] = 2 * (x * z + y * s);
   R[3] = 2 * (x * y + z * s), R[4] = 1 - 2 * (x * x + z * z), R[5] = 2 * (y * z - x * s);
   R[6] = 2 * (x * z - y * s), R[7] = 2 * (y * z + x * s), R[8] = 1 - 2 * (x * x + y * y);
}
void mo_integrate(void *h, long B, double dt, const double *q, const double *qd, const double *qdd, double *q_out, double *qd_out, double *qdd_out)
{
   const mo_model *m = (const mo_model *)h;
   const double hdd = 0.5 * dt * dt;
   for (long b = 0; b < B; b++)
   {
      const double *qr = q + b * m->nq, *vr = qd + b * m->nv, *ar = qdd + b * m->nv;
      double *qo = q_out + b * m->nq, *vo = qd_out + b * m->nv, *ao = qdd_out ? qdd_out + b * m->nv : NULL;
      for (int i = 0; i < m->n; i++)
      {
         const int *ci = m->cfg_idx + m->cfg_ofs[i], *di = m->dof_idx + m->dof_ofs[i];
         if (m->type[i] == MO_REVOLUTE || m->type[i] == MO_PRISMATIC)
         {
            double q0 = qr[ci[0]], v0 = vr[di[0]], a0 = ar[di[0]];
            qo[ci[0]] = hdd * a0 + dt * v0 + q0; /* :710-713 */
            vo[di[0]] = dt * a0 + v0;            /* :730-733 */
            if (ao)
               ao[di[0]] = a0;
         }
         else if (m->type[i] == MO_SPHERICAL)
         { /* :445-449, 578-625: q' = q * quat(dt w + 0.5 dt^2 al), w' = w + dt al */
            double quat[4] = {qr[ci[0]], qr[ci[1]], qr[ci[2]], qr[ci[3]]}, rv[3], dq[4] = {0, 0, 0, 1}, qn[4];
            for (int k = 0; k < 3; k++)
               rv[k] = dt * vr[di[k]] + hdd * ar[di[k]];
            double th = sqrt(v3_dot(rv, rv));
            if (th >= 1.0e-12)
            {
               double sc = sin(0.5 * th) / th;
               dq[0] = rv[0] * sc, dq[1] = rv[1] * sc, dq[2] = rv[2] * sc, dq[3] = cos(0.5 * th);
            }
            quat_mul(quat, dq, qn);
            for (int k = 0; k < 4; k++)
               qo[ci[k]] = qn[k];
            for (int k = 0; k < 3; k++)
            {
               vo[di[k]] = vr[di[k]] + dt * ar[di[k]];
               if (ao)
                  ao[di[k]] = ar[di[k]];
            }
         }
         else if (m->type[i] == MO_PLANAR)
         { /* the 6-DoF scheme (:503-575) with pose, twist and acceleration confined to the XZ plane (PlanarJoint is a FloatingJointBasics,
            * :415-418): rotation vector (0, th, 0), a_o = a + w x v = (ax + wy vz, az - wy vx), in-plane rotations by the pitch */
            double pitch = qr[ci[0]], px = qr[ci[1]], pz = qr[ci[2]];
            double wy = vr[di[0]], vx = vr[di[1]], vz = vr[di[2]], aly = ar[di[0]], ax = ar[di[1]], az = ar[di[2]];
            double aox = ax + wy * vz, aoz = az - wy * vx;
            double th = dt * wy + hdd * aly;
            double dpx = dt * vx + hdd * aox, dpz = dt * vz + hdd * aoz;
            double c0 = cos(pitch), s0 = sin(pitch), cd = cos(th), sd = sin(th);
            /* R_y(pitch) (x, z) = (c x + s z, -s x + c z) ; R_y(th)^T (x, z) = (c x - s z, s x + c z) */
            double wn = wy + dt * aly;
            double cx = vx + dt * aox, cz = vz + dt * aoz;
            double vnx = cd * cx - sd * cz, vnz = sd * cx + cd * cz;
            qo[ci[0]] = pitch + th;
            qo[ci[1]] = px + c0 * dpx + s0 * dpz;
            qo[ci[2]] = pz - s0 * dpx + c0 * dpz;
            vo[di[0]] = wn, vo[di[1]] = vnx, vo[di[2]] = vnz;
            if (ao)
            { /* a' = R(dq)^T a_o + v' x w' : (v' x w')_x = -vnz wn, (v' x w')_z = vnx wn */
               double anx = cd * aox - sd * aoz, anz = sd * aox + cd * aoz;
               ao[di[0]] = aly, ao[di[1]] = anx - vnz * wn, ao[di[2]] = anz + vnx * wn;
            }
         }
         else if (m->type[i] == MO_SIXDOF)
         {
            double quat[4] = {qr[ci[0]], qr[ci[1]], qr[ci[2]], qr[ci[3]]};
            double p[3] = {qr[ci[4]], qr[ci[5]], qr[ci[6]]};
            double w[3] = {vr[di[0]], vr[di[1]], vr[di[2]]}, v[3] = {vr[di[3]], vr[di[4]], vr[di[5]]};
            double al[3] = {ar[di[0]], ar[di[1]], ar[di[2]]}, a[3] = {ar[di[3]], ar[di[4]], ar[di[5]]};
            double wxv[3], ao3[3], rv[3], dq[4] = {0, 0, 0, 1}, R0[9], Rd[9], dp[3], t[3], vn[3], wn[3], an[3], qn[4], c[3];
            v3_cross(w, v, wxv);
            for (int k = 0; k < 3; k++)
            {
               ao3[k] = a[k] + wxv[k];
               rv[k] = dt * w[k] + hdd * al[k];
               wn[k] = w[k] + dt * al[k];
               dp[k] = dt * v[k] + hdd * ao3[k];
            }
            double th = sqrt(v3_dot(rv, rv));
            if (th >= 1.0e-12)
            {
               double sc = sin(0.5 * th) / th;
               dq[0] = rv[0] * sc, dq[1] = rv[1] * sc, dq[2] = rv[2] * sc, dq[3] = cos(0.5 * th);
            }
            quat_to_R(quat, R0);
            quat_to_R(dq, Rd);
            m3_mulv(R0, dp, t);
            for (int k = 0; k < 3; k++)
               t[k] += p[k], c[k] = v[k] + dt * ao3[k];
            m3_tmulv(Rd, c, vn);
            m3_tmulv(Rd, ao3, an);
            v3_cross(vn, wn, c);
            quat_mul(quat, dq, qn);
            for (int k = 0; k < 4; k++)
               qo[ci[k]] = qn[k];
            for (int k = 0; k < 3; k++)
            {
               qo[ci[4 + k]] = t[k];
               vo[di[k]] = wn[k], vo[di[3 + k]] = vn[k];
               if (ao)
                  ao[di[k]] = al[k], ao[di[3 + k]] = an[k] + c[k];
            }
         }
      }
   }
}

/* RNEA / ABA with the per-body outputs (either may be NULL): body_acc, body_twist [B][n][6] */
void mo_rnea_bodies(void *h, long B, const double *q, const double *qd, const double *qdd, const double *g, const double *fext, int coriolis,
                    int accel, double *tau, double *body_acc, double *body_twist)
{
   const mo_model *m = (const mo_model *)h;
   for (long b = 0; b < B; b++)
   {
      tap_acc = body_acc ? body_acc + b * 6 * m->n : NULL;
      tap_twist = body_twist ? body_twist + b * 6 * m->n : NULL;
      rnea_one(m, q + b * m->nq, qd + b * m->nv, qdd ? qdd + b * m->nv : NULL, g, fext ? fext + b * 6 * m->n : NULL, coriolis, accel && qdd,
               tau + b * m->nv);
   }
   tap_acc = tap_twist = NULL;
}
int mo_aba_bodies(void *h, long B, const double *q, const double *qd, const double *tau, const double *g, const double *fext, double *qdd,
                  double *body_acc, double *body_twist)
{
   const mo_model *m = (const mo_model *)h;
   int rc = 0;
   for (long b = 0; b < B; b++)
   {
      tap_acc = body_acc ? body_acc + b * 6 * m->n : NULL;
      tap_twist = body_twist ? body_twist + b * 6 * m->n : NULL;
      rc |= aba_one(m, q + b * m->nq, qd + b * m->nv, tau + b * m->nv, g, fext ? fext + b * 6 * m->n : NULL, qdd + b * m->nv, NULL, NULL, NULL);
   }
   tap_acc = tap_twist = NULL;
   return rc;
}

/* RNEA plus the 6-D wrench every joint transmits (moment, force), expressed in the frame after the joint:
 * InverseDynamicsCalculator.getComputedJointWrench (:578-585; passTwo :930-959 leaves it there).  ForwardDynamicsCalculator.getJointWrench
 * (:642-650, lazily evaluated by :1330-1363) is the same Newton-Euler sweep run on the accelerations forward dynamics computed, i.e. this
 * function called with qdd = ABA(tau) -- the reference's own test compares the two (ForwardDynamicsCalculatorTest.java:884-901). */
void mo_rnea_wrenches(void *h, long B, const double *q, const double *qd, const double *qdd, const double *g, const double *fext, int coriolis,
                      int accel, double *tau, double *joint_wrench)
{
   const mo_model *m = (const mo_model *)h;
   for (long b = 0; b < B; b++)
   {
      tap_wrench = joint_wrench ? joint_wrench + b * 6 * m->n : NULL;
      rnea_one(m, q + b * m->nq, qd + b * m->nv, qdd ? qdd + b * m->nv : NULL, g, fext ? fext + b * 6 * m->n : NULL, coriolis, accel && qdd,
               tau + b * m->nv);
   }
   tap_wrench = NULL;
}

/* RigidBodyAccelerationProvider.getRelativeAcceleration(base, body) (algorithms/interfaces/RigidBodyAccelerationProvider.java:199-235) on the
 * accelerations RNEA's first pass computes: acceleration of body's body-fixed frame with respect to base's, expressed in body's.
 * base[k] / body[k] index the listed joints (their successor bodies); -1 = the root body (pose identity, no twist, acceleration = the
 * root acceleration -g).  out [B][n_pairs][6].  The base's acceleration is re-expressed in the body's frame with the velocity-dependent
 * terms of SpatialAccelerationBasics.changeFrame(desiredFrame, deltaTwist, bodyTwist) (non-flipped branch, :192-200: the twist of the base
 * frame relative to the body frame and the base's own twist, both in the base frame) when velocities are considered. */
void mo_relative_acceleration(void *h, long B, const double *q, const double *qd, const double *qdd, const double *g, int coriolis, int accel,
                              int n_pairs, const int *base, const int *body, double *out)
{
   const mo_model *m = (const mo_model *)h;
   static _Thread_local mo_kin K;
   static _Thread_local double acc[MO_MAX_JOINTS][6], tw[MO_MAX_JOINTS][6];
   double *tau = (double *)malloc(sizeof(double) * (size_t)(m->nv > 0 ? m->nv : 1));
   double a_root[6];
   const double zero6[6] = {0};
   root_acceleration(g, a_root);
   xf_t W_world, T;
   xf_identity(&W_world);
   for (long b = 0; b < B; b++)
   {
      tap_acc = &acc[0][0], tap_twist = &tw[0][0];
      rnea_one(m, q + b * m->nq, qd + b * m->nv, qdd ? qdd + b * m->nv : NULL, g, NULL, coriolis, accel && qdd, tau);
      tap_acc = tap_twist = NULL;
      kinematics(m, q + b * m->nq, qd + b * m->nv, &K);
      for (int k = 0; k < n_pairs; k++)
      {
         const int b1 = base[k], b2 = body[k];
         const xf_t *W1 = b1 < 0 ? &W_world : &K.W_body[b1], *W2 = b2 < 0 ? &W_world : &K.W_body[b2];
         const double *t1 = b1 < 0 ? zero6 : tw[b1], *t2 = b2 < 0 ? zero6 : tw[b2];
         double a1[6], a1in2[6];
         memcpy(a1, b1 < 0 ? a_root : acc[b1], sizeof a1);
         if (coriolis)
         {
            double t2in1[6], d[6], c1[3], c2[3], c3[3];
            xf_between(W2, W1, &T);
            xf_motion(&T, t2, t2in1);
            for (int c = 0; c < 6; c++)
               d[c] = t1[c] - t2in1[c]; /* baseFrame.getTwistRelativeToOther(bodyFrame): :217 */
            v3_cross(d + 3, t1, c1);    /* v_delta x w_base   */
            v3_cross(d, t1 + 3, c2);    /* w_delta x v_base   */
            v3_cross(d, t1, c3);        /* w_delta x w_base   */
            for (int c = 0; c < 3; c++)
               a1[3 + c] += c1[c] + c2[c], a1[c] += c3[c];
         }
         xf_between(W1, W2, &T);
         xf_motion(&T, a1, a1in2);
         const double *a2 = b2 < 0 ? a_root : acc[b2];
         for (int c = 0; c < 6; c++)
            out[((size_t)b * n_pairs + k) * 6 + c] = a2[c] - a1in2[c]; /* :228 */
      }
   }
   free(tau);
}

/* ================================================================== unit-level entry points
 * The building blocks above, exported one by one so that tests/test_oracle_units.py can restate the reference's own unit tests on
 * them (the only pins this environment allows):
 *   test/.../algorithms/ArticulatedBodyInertiaTest.java:25-130   ABI applyTransform == SpatialInertia applyTransform on rigid inertias
 *   test/.../spatial/SpatialInertiaBasicsTest.java:76-98,129-157,216-340   co-energy, fast == general wrench, frame invariances
 *   test/.../tools/MecanoToolsTest.java:218-460,618-694          parallel-axis translation, dynamic force / moment, co-energy
 * X[12] = R row-major then p, as everywhere in this file. */
static void xf_load(const double X[12], xf_t *T)
{
   memcpy(T->R, X, 9 * sizeof(double));
   memcpy(T->p, X + 9, 3 * sizeof(double));
}
void mo_unit_dynamic_wrench(const double J[9], double mass, const double c[3], const double *acc, const double *tw, int force_general,
                            double out[6])
{
   unit_force_general_wrench = force_general;
   dynamic_wrench(J, mass, c, acc, tw, out);
   unit_force_general_wrench = 0;
}
/* SpatialInertiaBasics.applyTransform / applyInverseTransform (spatial/interfaces/SpatialInertiaBasics.java:222-269) */
void mo_unit_rigid_apply_transform(const double X[12], int inverse, double J[9], double *mass, double c[3])
{
   xf_t T, Ti;
   rigid_t I;
   xf_load(X, &T);
   if (inverse)
   {
      xf_inv(&T, &Ti);
      T = Ti;
   }
   memcpy(I.J, J, sizeof I.J);
   I.m = *mass;
   memcpy(I.c, c, sizeof I.c);
   rigid_apply_transform(&T, &I);
   memcpy(J, I.J, sizeof I.J);
   *mass = I.m;
   memcpy(c, I.c, sizeof I.c);
}
/* ArticulatedBodyInertia.applyTransform / applyInverseTransform (algorithms/ArticulatedBodyInertia.java:359-402) */
void mo_unit_abi_apply_transform(const double X[12], int inverse, double A[9], double L[9], double C[9])
{
   xf_t T, Ti;
   abi_t I;
   xf_load(X, &T);
   if (inverse)
   {
      xf_inv(&T, &Ti);
      T = Ti;
   }
   memcpy(I.A, A, sizeof I.A), memcpy(I.L, L, sizeof I.L), memcpy(I.C, C, sizeof I.C);
   abi_apply_transform(&T, &I);
   memcpy(A, I.A, sizeof I.A), memcpy(L, I.L, sizeof I.L), memcpy(C, I.C, sizeof I.C);
}
/* ArticulatedBodyInertia.setIncludingFrame(SpatialInertia) (:176-186) and both dense 6x6 forms (SpatialInertiaReadOnly.java:394-415) */
void mo_unit_abi_from_rigid(const double J[9], double mass, const double c[3], double A[9], double L[9], double C[9])
{
   abi_t I;
   abi_from_rigid(J, mass, c, &I);
   memcpy(A, I.A, sizeof I.A), memcpy(L, I.L, sizeof I.L), memcpy(C, I.C, sizeof I.C);
}
void mo_unit_abi_to_dense(const double A[9], const double L[9], const double C[9], double M[36])
{
   abi_t I;
   memcpy(I.A, A, sizeof I.A), memcpy(I.L, L, sizeof I.L), memcpy(I.C, C, sizeof I.C);
   abi_to_dense(&I, M);
}
void mo_unit_rigid_mulv(const double J[9], double mass, const double c[3], const double x[6], double out[6])
{
   rigid_t I;
   memcpy(I.J, J, sizeof I.J);
   I.m = mass;
   memcpy(I.c, c, sizeof I.c);
   rigid_mulv(&I, x, out);
}
void mo_unit_motion_transform(const double X[12], int inverse, const double in[6], double out[6])
{
   xf_t T;
   xf_load(X, &T);
   if (inverse)
      xf_motion_inv(&T, in, out);
   else
      xf_motion(&T, in, out);
}
void mo_unit_force_transform(const double X[12], const double in[6], double out[6])
{
   xf_t T;
   xf_load(X, &T);
   xf_force(&T, in, out);
}
/* tools/MecanoTools.java:844-890: T = 1/2 (m v.v + 2 m w.(c x v) + w.J w) */
double mo_unit_kinetic_coenergy(const double J[9], double mass, const double c[3], const double tw[6])
{
   double cxv[3], Jw[3];
   double energy = mass * v3_dot(tw + 3, tw + 3);
   v3_cross(c, tw + 3, cxv);
   energy += 2.0 * mass * v3_dot(tw, cxv);
   m3_mulv(J, tw, Jw);
   energy += v3_dot(tw, Jw);
   return 0.5 * energy;
}
