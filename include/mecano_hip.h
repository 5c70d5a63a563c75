/*
 * mecano_hip.h -- C-ABI of the MI355X batched rigid-body-dynamics engine.
 *
 * This is the drop-in boundary for ONE hot path of ihmcrobotics/mecano (Java):
 *
 *   InverseDynamicsCalculator              (RNEA)  algorithms/InverseDynamicsCalculator.java:481-501,567
 *   ForwardDynamicsCalculator              (ABA)   algorithms/ForwardDynamicsCalculator.java:475-520,556-591
 *   CompositeRigidBodyMassMatrixCalculator (CRBA)  algorithms/CompositeRigidBodyMassMatrixCalculator.java:344-348
 *
 * (paths below are relative to src/main/java/us/ihmc/mecano/ of the reference).
 *
 * The reference evaluates one configuration per compute() call, on one CPU
 * thread, reading q / qd through the MovingReferenceFrame tree.  This library
 * evaluates B configurations per call on one GPU.  A Java (Panama / JNI), C++
 * or Python host flattens a MultiBodySystemReadOnly once into mh_model_desc
 * (recipe: tools/MultiBodySystemFactories.java:401-470,782-868 -- see
 * INTEGRATION.md), then calls mh_rnea / mh_aba / mh_crba with state matrices
 * laid out exactly like the DMatrixRMaj column vectors Mecano uses, stacked
 * along a new leading batch dimension.
 *
 * Conventions (all identical to the reference; B is the only new dimension):
 *   - rigid transform X[12] = { R row-major (9), p (3) } maps coordinates of a
 *     frame into its parent frame: x_parent = R * x_child + p
 *     (ReferenceFrame.getTransformToParent()).
 *   - spatial vectors are ordered angular(3) then linear(3).
 *   - SixDoF configuration = quaternion (x, y, z, s) then position (x, y, z)
 *     (multiBodySystem/interfaces/SixDoFJointReadOnly.java:21-26); it is
 *     normalised on input like Euclid's Quaternion.set does.  SixDoF velocity /
 *     acceleration / effort = (angular, linear) expressed in the frame after
 *     the joint (multiBodySystem/SixDoFJoint.java:64-70).
 *   - gravity is the vector g; the root acceleration is set to -g, expressed in
 *     the root body-fixed frame (InverseDynamicsCalculator.java:343-348).
 *   - external wrenches are per body, expressed in (and about the origin of)
 *     that body's body-fixed (CoM) frame (InverseDynamicsCalculator.java:469-472).
 *   - the mass matrix is dense row-major nv x nv with both triangles filled and
 *     zeros for unrelated branches (CompositeRigidBodyMassMatrixCalculator.java:298,841-845).
 *   - row r of a state matrix belongs to the joint DoF (or configuration entry)
 *     whose entry in dof_indices (cfg_indices) equals r: the
 *     JointMatrixIndexProvider contract
 *     (multiBodySystem/interfaces/JointMatrixIndexProvider.java:71-123).
 *
 * Threading: a model handle is READ-ONLY after mh_model_create and may be shared by any number of host threads and streams.  What
 * compute calls write besides their outputs -- device workspace, scratch matrices, staging buffers, the hand-off flags and the error word
 * of the bias-split forward dynamics -- belongs to a CONTEXT: mh_context_create(model) makes one per host thread / per stream, and a call
 * names it in opts->context.  Calls that leave opts->context NULL use the model's built-in default context and are then subject to the
 * rule of the reference's calculators (one per thread: their scratch fields, InverseDynamicsCalculator.java:706-707): one host thread
 * and one stream at a time per context.  Contexts of one model are independent of each other; so are different models.
 *
 * Calls with device pointers are ASYNCHRONOUS on opts->stream: they return once the work is enqueued.  A failure that shows only while
 * the work runs is therefore reported later: by mh_model_check (synchronises the stream and reports for one context), by
 * mh_stream_synchronize (reports for every context), by the *_host entry points (synchronous: they check before they return) and at the
 * latest by the context's next forward-dynamics call.  One such failure exists: a bias-split forward dynamics launch (small batches of a
 * model with a tree-split code object, mh_aba_f64 / mh_rnea_aba_f64) whose consumer workgroup waited longer than MH_ZV_WAIT_MS
 * (environment, default 2000) for its producer; the accelerations of those 64 configurations are then written as NaN, never as numbers.
 *
 * Last bits: mh_aba_f64 / mh_rnea_aba_f64 choose the formulation of forward dynamics by batch size (bias split while every job gets a CU
 * of its own, one-job tree split up to one group of 64 configurations per CU, beyond that bias and inertia job fused in one workgroup --
 * or two launches where the code object has no fused kernel; MH_ZV / MH_ZVF / MH_ZVB in the environment force each off or on).  The
 * formulations agree to about 1e-13 relative, not bit for bit: the same configuration evaluated inside batches of different sizes -- e.g.
 * in shards of different sizes on different ranks -- may differ in its last bits.  MH_ZV=0 MH_ZVF=0 MH_ZVB=0 pins the one-job form.
 * (Inverse dynamics and the mass matrix have one formulation per code object: their results do not depend on the batch size -- with one
 * exception: beyond one group of 64 configurations per CU mh_rnea_aba_f64 forms its efforts as h + M(q) qdd inside the forward-dynamics
 * launch, within ~1e-15 relative of mh_rnea_f64's, not bit for bit; MH_ZVF_PAIR=0 in the environment keeps the two launches.)
 *
 * No function throws or aborts; every entry point returns an mh_status and
 * mh_last_error() gives a thread-local message.  The library never falls back
 * to a CPU implementation: without a usable HIP device every compute call
 * returns MH_ERR_NO_DEVICE.
 */
#ifndef MECANO_HIP_H
#define MECANO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_ABI_VERSION 5

/* ---- status codes (the Java shim maps them back to Mecano's exception types) ---- */
typedef enum mh_status
{
   MH_OK = 0,
   MH_ERR_INVALID_ARGUMENT = 1,   /* IllegalArgumentException / NullPointerException        */
   MH_ERR_BAD_DIMENSION = 2,      /* MatrixDimensionException (ForwardDynamicsCalculator.java:522-533) */
   MH_ERR_UNSUPPORTED_JOINT = 3,  /* joint kind outside the mh_joint_type enumeration         */
   MH_ERR_LOOP_CLOSURE = 4,       /* kinematic loops: unsupported, as in ForwardDynamicsCalculator.java:207-211 */
   MH_ERR_BAD_TOPOLOGY = 5,       /* parent[] is not a forest / index maps are not a permutation */
   MH_ERR_BAD_AXIS = 6,           /* 1-DoF axis is not a unit vector                          */
   MH_ERR_NO_DEVICE = 7,          /* no HIP device / kernels not loadable                     */
   MH_ERR_HIP = 8,                /* a HIP runtime call failed (message has the HIP error)     */
   MH_ERR_OUT_OF_MEMORY = 9,
   MH_ERR_NOT_RESERVED = 10,      /* batch larger than mh_reserve()d while allocation is forbidden */
   MH_ERR_SINGULAR = 11           /* reserved: the device path does not test joint-space inertias for definiteness -- like the reference's
                                     unguarded 1/D (ForwardDynamicsCalculator.java:1183) a singular block shows as inf / nan in qdd */
} mh_status;

/* ---- joint kinds ---- */
typedef enum mh_joint_type
{
   MH_JOINT_REVOLUTE = 0,  /* multiBodySystem/RevoluteJoint.java   nq=1 nv=1 */
   MH_JOINT_PRISMATIC = 1, /* multiBodySystem/PrismaticJoint.java  nq=1 nv=1 */
   MH_JOINT_SIXDOF = 2,    /* multiBodySystem/SixDoFJoint.java     nq=7 nv=6 */
   MH_JOINT_FIXED = 3,     /* multiBodySystem/FixedJoint.java      nq=0 nv=0 */
   MH_JOINT_PLANAR = 4,    /* multiBodySystem/PlanarJoint.java     nq=3 nv=3   q = (pitch, x, z), qd = (w_y, v_x, v_z) */
   MH_JOINT_SPHERICAL = 5  /* multiBodySystem/SphericalJoint.java  nq=4 nv=3   q = quaternion (x, y, z, s), qd = angular velocity */
} mh_joint_type;

/* ---- memory layout of batched state matrices ---- */
typedef enum mh_layout
{
   MH_LAYOUT_AOS = 0, /* [B][n]: B stacked DMatrixRMaj column vectors (default) */
   MH_LAYOUT_SOA = 1  /* [n][B]: one row per DoF, batch contiguous               */
} mh_layout;

/*
 * Flat description of a MultiBodySystemReadOnly, joints listed in the order of
 * input.getJointMatrixIndexProvider().getIndexedJointsInOrder()
 * (multiBodySystem/interfaces/MultiBodySystemReadOnly.java:57-60,101-104).
 * Joints in getJointsToIgnore() are simply not listed (their subtree inertia,
 * if it must be considered, is lumped by the host into the parent's J/mass/com
 * as InverseDynamicsCalculator.java:832-860 does).
 */
typedef struct mh_model_desc
{
   int32_t n_joints;
   int32_t nq; /* rows of a configuration matrix */
   int32_t nv; /* rows of a velocity / acceleration / effort matrix */
   const int32_t *parent;      /* [n] index of the parent joint (joint.getPredecessor().getParentJoint()), -1 if the predecessor is the root body */
   const int32_t *joint_type;  /* [n] mh_joint_type */
   const double *axis;         /* [3n] joint axis in the joint frame (OneDoFJointReadOnly.getJointAxis()); ignored for SIXDOF/FIXED */
   const double *X_before;     /* [12n] joint.getFrameBeforeJoint().getTransformToParent(); identity when that frame is the parent frame itself (tools/MecanoFactories.java:81-91) */
   const double *X_com;        /* [12n] joint.getSuccessor().getBodyFixedFrame().getTransformToParent() (multiBodySystem/RigidBody.java:170-185) */
   const double *inertia_J;    /* [9n] successor.getInertia().getMomentOfInertia(), row-major, in the body-fixed frame */
   const double *inertia_mass; /* [n]  successor.getInertia().getMass() */
   const double *inertia_com;  /* [3n] successor.getInertia().getCenterOfMassOffset() (normally zero) */
   const int32_t *dof_indices; /* [sum of joint DoFs] getJointDoFIndices(joint), concatenated joint by joint */
   const int32_t *cfg_indices; /* [sum of joint configuration sizes] getJointConfigurationIndices(joint), concatenated */
} mh_model_desc;

/* Per-call switches: mirror of the calculators' setters. */
typedef struct mh_options
{
   int32_t consider_coriolis;      /* InverseDynamicsCalculator.setConsiderCoriolisAndCentrifugalForces (java:291-296); RNEA only; default 1 */
   int32_t consider_accelerations; /* InverseDynamicsCalculator.setConsiderJointAccelerations (java:301-306); RNEA only; default 1 */
   int32_t layout;                 /* mh_layout of every batched matrix of the call */
   int32_t use_root_acceleration;  /* 0 (default): the root body accelerates with (0, -gravity), the `gravity` argument of the call
                                    * (setGravity, InverseDynamicsCalculator.java:318-348, ForwardDynamicsCalculator.java:234-264).
                                    * non-zero: with root_acceleration below; the call's `gravity` argument is ignored and may be NULL */
   void *stream;                   /* hipStream_t to launch on; NULL = the device's null stream */
   double root_acceleration[6];    /* InverseDynamicsCalculator.setRootAcceleration(SpatialAccelerationReadOnly) (java:413-427) and
                                    * ForwardDynamicsCalculator.setRootAcceleration (java:330-343): spatial acceleration of the root body
                                    * (angular x y z, then linear x y z), expressed in the root body's frame, of the root body's frame
                                    * relative to an inertial frame -- what a moving / rotating base contributes; (0, 0, 0, -g) is gravity */
   void *context;                  /* mh_context_t the call's workspace, scratch and flags come from; NULL (default) = the model's own
                                    * default context (then: one host thread and one stream at a time on this model) */
} mh_options;

typedef struct mh_model *mh_model_t;
typedef struct mh_context *mh_context_t;

/* ---- library / device ---- */
int32_t mh_abi_version(void);
/* what a topology-specialised code object (libmecano_hip_topo_<key>.so) must report from its mh_spec_abi() to be accepted by this
 * library build: a hash over the argument structs, record strides and the canonical-frame convention the two share */
uint64_t mh_spec_abi_stamp(void);
/*
 * Build provenance (MH_ABI_VERSION 5).  Every binary carries a hash of what it was compiled from -- FNV-1a 64 over the source files, their
 * headers and the code-generation flags -- as the string "MH_BUILD_ID=<hash>;..." in its file and through these calls:
 *   mh_build_hash()              this library's own sources and flags
 *   mh_spec_sources_hash()       the kernel sources + flags a code object libmecano_hip_topo_<key>.so must have been compiled from to be
 *                                accepted by mh_model_create (it exports the same name; a different hash is refused like a wrong ABI stamp,
 *                                mh_model_kernel_variant says so)
 *   mh_spec_sources_hash_of(dir) the hash of the kernel sources found in `dir` (csrc/), "h" + 16 hex digits + NUL into out[18]: what
 *                                mh_build_code_object checks before it compiles, what mecano_amd/build.py compares instead of file times
 */
const char *mh_build_hash(void);
const char *mh_spec_sources_hash(void);
mh_status mh_spec_sources_hash_of(const char *csrc_dir, char out[18]);
const char *mh_last_error(void);           /* thread-local, never NULL */
mh_status mh_device_count(int32_t *count); /* 0 devices is MH_OK with *count = 0 */
mh_status mh_set_device(int32_t device);   /* device used by subsequent calls of this thread */
void mh_options_default(mh_options *opts); /* coriolis=1, accelerations=1, AoS, null stream */

/* ---- model (replaces the calculators' constructors, InverseDynamicsCalculator.java:226-282) ---- */
mh_status mh_model_create(const mh_model_desc *desc, mh_model_t *model_out);
void mh_model_destroy(mh_model_t model);
int32_t mh_model_nq(mh_model_t model);
int32_t mh_model_nv(mh_model_t model);
int32_t mh_model_n_joints(mh_model_t model);
/* name of the kernel variant compute calls will use for this model: "topo:<key>" when the topology-specialised code object
 * libmecano_hip_topo_<key>.so was found, matches this library build (ABI stamp) and passed the create-time self-check against the
 * run-time-topology kernels; "generic" otherwise, followed by the reason in parentheses when a code object was found but refused */
const char *mh_model_kernel_variant(mh_model_t model);
/*
 * Where this engine consciously departs from the reference (MH_ABI_VERSION 5).  mh_model_create inspects the description and sets:
 *   MH_WARN_NEAR_COORDINATE_AXIS  a revolute axis lies within 1e-7 of +X, +Y or +Z without being that axis.  Mecano then rotates the joint
 *                                 about the EXACT coordinate axis (roll / pitch / yaw closed forms, tools/MecanoFactories.java:51,237-248)
 *                                 while the joint's unit twist keeps the axis as given (multiBodySystem/OneDoFJoint.java:170); the engine
 *                                 uses the given axis for both.  Bound: results differ from Mecano's by <= ~4e-7 relative (instead of 1e-10).
 *                                 Remedy: snap the axis onto the coordinate axis in the description (then both agree to 1e-10).
 *   MH_WARN_TINY_COMPOSITE_MASS   a body's mass plus a child subtree's is under 1e-7: Mecano's SpatialInertia.add skips the
 *                                 renormalisation of the centre of mass (spatial/interfaces/FixedFrameSpatialInertiaBasics.java:174-175);
 *                                 the engine's composite (m, m c, I) stays consistent.  Concerns mh_crba_* / Coriolis / centroidal outputs
 *                                 only (RNEA and ABA never add inertias); bound: <= 1e-6 absolute on H for masses of that size.
 * 0 = the model is in neither class and every output is held to 1e-10 against the reference's arithmetic.  mh_model_warning_text gives
 * the joints concerned ("" when no bit is set; owned by the model); mh_model_create also leaves that text in mh_last_error() while
 * returning MH_OK.
 */
#define MH_WARN_NEAR_COORDINATE_AXIS 1u
#define MH_WARN_TINY_COMPOSITE_MASS 2u
uint32_t mh_model_warnings(mh_model_t model);
const char *mh_model_warning_text(mh_model_t model);

/*
 * Host-only: validates the description and returns the key of its topology (tree shape + joint kinds, in the engine's
 * parents-first order) together with that order's parent / kind arrays.  A topology-specialised code object
 * libmecano_hip_topo_<key>.so placed next to libmecano_hip.so is picked up by mh_model_create (mecano_amd/build.py builds them).
 * key_out holds 16 hex digits + NUL; parents_out / types_out have n_joints entries (either may be NULL).
 */
mh_status mh_topology_key(const mh_model_desc *desc, char key_out[17], int32_t *parents_out, int32_t *types_out);

/*
 * Builds the topology-specialised code object of a model (host-only; runs hipcc -- MH_HIPCC, /opt/rocm/bin/hipcc or the PATH -- on the
 * kernel sources that ship next to the library, csrc/; minutes for a 25-body tree) into out_dir (NULL: next to the library) and returns
 * its path.  Models created afterwards with the same tree shape and joint kinds load it (after the ABI-stamp check and the create-time
 * self-check).  Without it -- and for planar / spherical joints or trees deeper than 16 joints -- a model runs on the run-time-topology
 * kernels (same results; about 2x slower at small batches of a branching tree, 3.5x at device-filling ones).  A host without Python calls
 * this once per robot, e.g. at installation.  With MH_BUILD_FAST=1 in the environment only the tree-split RNEA / ABA / fused kernels for
 * AoS matrices with identity index maps are built -- seconds instead of minutes; every other plan of the model (SoA, per-body outputs,
 * mass matrix, ...) keeps running on the run-time-topology kernels.  MH_AUTO_BUILD=1 (2: the full set) makes mh_model_create do this by
 * itself for a tree it finds no code object for, into MH_SPEC_DIR or next to the library.
 */
mh_status mh_build_code_object(const mh_model_desc *desc, const char *out_dir, char *path_out, size_t path_cap);

/*
 * ---- contexts (see "Threading" at the top) ----
 * A context owns everything a compute call writes besides its outputs; the model handle it was made from stays read-only.  Make one per
 * host thread or stream, pass it in opts->context.  mh_context_reserve is mh_reserve for a context.
 * While contexts of a model exist mh_model_set_joint_source_modes is refused (the contexts hold copies of the joint records' host side).
 * The model is reference-counted by its contexts (they share its device records): mh_model_destroy on a model with live contexts gives
 * up the caller's reference only -- calls through the surviving contexts (with the same handle value) stay valid, no new context can be
 * made -- and the last mh_context_destroy releases the device records.  mh_context_create may run while other threads compute through
 * the model or its contexts.  mh_stream_synchronize reports a pending asynchronous failure of ANY context and clears it -- use
 * mh_model_check where the failing context matters.
 */
mh_status mh_context_create(mh_model_t model, mh_context_t *ctx_out);
void mh_context_destroy(mh_context_t ctx);
mh_status mh_context_reserve(mh_context_t ctx, int64_t max_batch);
/* Synchronises `stream` (NULL = the null stream) and reports failures of the asynchronous calls issued through `ctx` (NULL = the model's
 * default context) that only showed on the device -- see "Calls with device pointers are ASYNCHRONOUS" at the top.  MH_OK: the outputs
 * of every call of this context enqueued on `stream` so far are valid. */
mh_status mh_model_check(mh_model_t model, mh_context_t ctx, void *stream);

/* Pre-allocate device workspace for batches up to max_batch so that compute calls allocate nothing.  After it (and one first call of
 * each entry point, which sets kernel attributes once) the device-pointer entry points only enqueue work on opts->stream -- kernels,
 * memsets, and for a pair call without a fused kernel an event fork / join with a stream of the model's own: they can be captured in a
 * HIP graph and replayed (tests/test_gpu_parity.py::test_entry_points_are_graph_capturable). */
mh_status mh_reserve(mh_model_t model, int64_t max_batch);

/*
 * ---- compute, DEVICE pointers ----
 * q [B][nq], qd/qdd/tau [B][nv] (or transposed with MH_LAYOUT_SOA), gravity[3] is a HOST pointer,
 * f_ext is NULL or a device pointer [B][n_joints][6] (AoS) / [n_joints*6][B] (SoA) holding, for the
 * successor body of each listed joint, the external wrench (moment, force) in its body-fixed frame.
 * Calls are asynchronous on opts->stream.
 */
mh_status mh_rnea_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd,
                      const double gravity[3], const double *f_ext, const mh_options *opts, double *tau_out);
mh_status mh_aba_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *tau,
                     const double gravity[3], const double *f_ext, const mh_options *opts, double *qdd_out);
mh_status mh_crba_f64(mh_model_t model, int64_t B, const double *q, const mh_options *opts, double *H_out);
/*
 * RNEA and ABA of the same B configurations in one call: tau_out = RNEA(q, qd, qdd), qdd_out = ABA(q, qd, tau).  The two are
 * independent.  With a code object the call is ONE launch at every batch size: up to one group of 64 configurations per CU the two run
 * side by side on different workgroups (a 4096-configuration batch is 64 waves, a quarter of what fills an MI355X); beyond that the
 * forward-dynamics kernel also writes tau_out = h + M(q) qdd ("Last bits" above).  Same results as mh_rnea_f64 followed by mh_aba_f64.
 * Because the two run concurrently, tau_out and qdd_out must not overlap q, qd, qdd, tau or each other (MH_ERR_INVALID_ARGUMENT); for
 * in-place use call mh_rnea_f64 and mh_aba_f64 one after the other.
 */
mh_status mh_rnea_aba_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double *tau,
                          const double gravity[3], const double *f_ext, const mh_options *opts, double *tau_out, double *qdd_out);

/*
 * InverseDynamicsCalculator.compute + CompositeRigidBodyMassMatrixCalculator.getMassMatrix for the SAME configurations in one call
 * (what a whole-body controller evaluates per tick: algorithms/InverseDynamicsCalculator.java:496-501,
 * algorithms/CompositeRigidBodyMassMatrixCalculator.java:344-348): tau_out [B][nv], H_out [B][nv][nv].  With a code object the two run
 * side by side in ONE launch; otherwise as two launches, concurrently on small batches.  Same arguments as mh_rnea_f64 / mh_crba_f64.
 */
mh_status mh_rnea_crba_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double gravity[3],
                           const double *f_ext, const mh_options *opts, double *tau_out, double *H_out);

/*
 * ---- joint source modes (ForwardDynamicsCalculator.JointSourceMode, ForwardDynamicsCalculator.java:45-57, 400-444) ----
 * modes[n_joints], one per listed joint: MH_EFFORT_SOURCE (tau is the input, qdd the output; the default) or
 * MH_ACCELERATION_SOURCE (the joint is "locked" onto a given acceleration: qdd is the input, tau the output).  NULL resets every
 * joint to MH_EFFORT_SOURCE (resetJointSourceModes, :441-444).  Synchronises the device; not to be called while compute calls
 * on this model are in flight.  While any joint is an acceleration source, forward dynamics goes through mh_aba_locked_f64 and
 * mh_aba_f64 / mh_rnea_aba_f64 return MH_ERR_INVALID_ARGUMENT (they have no acceleration input).
 */
enum
{
   MH_EFFORT_SOURCE = 0,
   MH_ACCELERATION_SOURCE = 1
};
mh_status mh_model_set_joint_source_modes(mh_model_t model, const int32_t *modes);
/* number of joints currently in MH_ACCELERATION_SOURCE mode */
int32_t mh_model_n_acceleration_sources(mh_model_t model);
/*
 * Forward dynamics with acceleration-source joints (ForwardDynamicsCalculator.compute(tau, qdd), :508-520, passes two/three/four
 * :1237-1253, 1284-1297, 1315-1363).  tau [B][nv] is read at the DoFs of the effort-source joints, qdd_in [B][nv] at the DoFs of
 * the acceleration-source joints (it may be NULL when there are none).  qdd_out receives every joint's acceleration (the given
 * ones copied through); tau_out, when not NULL, every joint's effort (the given ones copied through, the efforts that realise the
 * prescribed accelerations computed).  In-place use (qdd_out == qdd_in, tau_out == tau) is allowed.
 */
mh_status mh_aba_locked_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *tau, const double *qdd_in,
                            const double gravity[3], const double *f_ext, const mh_options *opts, double *qdd_out, double *tau_out);

/*
 * ---- per-body outputs (RigidBodyAccelerationProvider: InverseDynamicsCalculator.getAccelerationProvider, InverseDynamicsCalculator.java:242-250,
 *      660-663; ForwardDynamicsCalculator.getAccelerationProvider, ForwardDynamicsCalculator.java:170-180, 715-718) ----
 * Same as mh_rnea_f64 / mh_aba_f64, plus for the successor body of every listed joint its spatial acceleration and / or twist relative
 * to the inertial frame, expressed in the body-fixed frame: body_acc_out, body_twist_out [B][n_joints][6] (angular, linear), laid
 * out like f_ext; either may be NULL.  As in the reference the acceleration carries the root acceleration -g, and the RNEA switches
 * apply (consider_coriolis = 0: velocity terms dropped and twists reported as zero).  Models with a tree-split code object, identity
 * index maps and AoS matrices run variants of the tree-split kernels that write them as well; every other case the run-time-topology kernels.
 */
mh_status mh_rnea_bodies_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double gravity[3],
                             const double *f_ext, const mh_options *opts, double *tau_out, double *body_acc_out, double *body_twist_out);
mh_status mh_aba_bodies_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *tau, const double gravity[3],
                            const double *f_ext, const mh_options *opts, double *qdd_out, double *body_acc_out, double *body_twist_out);

/*
 * ---- per-joint wrenches (InverseDynamicsCalculator.getComputedJointWrench, InverseDynamicsCalculator.java:578-585, 930-959;
 *      ForwardDynamicsCalculator.getJointWrench, ForwardDynamicsCalculator.java:642-650, 1330-1363) ----
 * Same as mh_rnea_f64 / mh_aba_f64, plus for every listed joint the full 6-D wrench (moment, force) it transmits to its successor body,
 * before projection onto the motion subspace, expressed in the joint's frame after the joint: joint_wrench_out [B][n_joints][6], laid
 * out like f_ext.  tau = S^T wrench.  The forward-dynamics form evaluates, like the reference, a Newton-Euler sweep over the accelerations
 * it has just computed (a second launch on the same stream); it needs every joint to be an effort source.  Run-time-topology kernels.
 */
mh_status mh_rnea_joint_wrenches_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double gravity[3],
                                     const double *f_ext, const mh_options *opts, double *tau_out, double *joint_wrench_out);
mh_status mh_aba_joint_wrenches_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *tau, const double gravity[3],
                                    const double *f_ext, const mh_options *opts, double *qdd_out, double *joint_wrench_out);

/*
 * ---- relative accelerations (RigidBodyAccelerationProvider.getRelativeAcceleration(base, body),
 *      algorithms/interfaces/RigidBodyAccelerationProvider.java:66, 199-235) ----
 * For n_pairs pairs of bodies: the spatial acceleration of body's body-fixed frame with respect to base's, expressed in body's, from the
 * per-body outputs of a previous mh_rnea_bodies_f64 / mh_aba_bodies_f64 call on the same configurations (body_acc, body_twist: DEVICE
 * pointers, [B][n_joints][6]).  base_joints / body_joints are HOST arrays of n_pairs indices into the model's joint list (the successor
 * body of that joint); -1 names the root body, whose acceleration is the root acceleration -g (gravity[3], HOST pointer).
 * opts->consider_coriolis = 0 mirrors a provider that ignores velocities (areVelocitiesConsidered() == false: plain change of frame,
 * body_twist may be NULL).  out [B][n_pairs][6] (MH_LAYOUT_SOA: [n_pairs*6][B]), DEVICE pointer.
 */
mh_status mh_relative_acceleration_f64(mh_model_t model, int64_t B, const double *q, const double *body_acc, const double *body_twist,
                                       const double gravity[3], int32_t n_pairs, const int32_t *base_joints, const int32_t *body_joints,
                                       const mh_options *opts, double *out);

/*
 * ---- joint torque regressor (JointTorqueRegressorCalculator.compute / getJointTorqueRegressorMatrix,
 *      algorithms/JointTorqueRegressorCalculator.java:173-190, 450-453, 749-833; a caller of the inverse dynamics, :118, :181-182, :802-804) ----
 * Y_out [B][nv][10 n_joints], one row-major nv x 10 n matrix per configuration (a DMatrixRMaj each; MH_LAYOUT_SOA: [nv][10 n_joints][B],
 * every store coalesced -- several times faster to produce):  tau = Y pi  for the inverse
 * dynamics without external wrenches, with pi = (mass, com_x, com_y, com_z, Ixx, Ixy, Ixz, Iyy, Iyz, Izz) of every successor body in
 * its body-fixed frame (:877-889; the columns of a body follow SpatialInertiaBasisOption, :514-516).  The ten columns of joint j's
 * successor body start at column 10 j, j in mh_model_desc order -- the reference orders the blocks by the iteration order of a HashMap
 * of rigid bodies (:85, :123, :318-329), which is not specified; getJointTorqueRegressorMatrixBlock(body) (:462-465) is the order-free
 * accessor a shim maps onto this layout.  opts->consider_coriolis / consider_accelerations as in mh_rnea_f64
 * (setConsiderCoriolisAndCentrifugalForces / setConsiderJointAccelerations, :489-502); opts->layout is the layout of q, qd, qdd and Y.
 * first_moment_columns = 0 reproduces the reference: its MCOM_X/Y/Z bases put a centre-of-mass offset on a body of zero mass (:579-581)
 * and every term of the dynamic wrench carries the mass (tools/MecanoTools.java:632-702, 785-822), so those three columns are zero --
 * except with consider_coriolis = 0, where the inverse dynamics passes no twist and computeDynamicMoment leaves c x a unscaled
 * (tools/MecanoTools.java:650-692): the columns then hold the moment e x a, as the reference's do.
 * first_moment_columns = 1 writes d tau / d (m c) there instead (the linear parametrisation used for identification: pi then holds
 * m c in slots 1..3 and the moments of inertia about the origin of the body-fixed frame).  Device pointers, asynchronous on opts->stream;
 * run-time-topology kernel for every model.
 */
mh_status mh_regressor_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double gravity[3],
                           const mh_options *opts, int32_t first_moment_columns, double *Y_out);
mh_status mh_regressor_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *qdd, const double gravity[3],
                           const mh_options *opts, int32_t first_moment_columns, float *Y_out);

/*
 * ---- Coriolis matrix (CompositeRigidBodyMassMatrixCalculator.setEnableCoriolisMatrixCalculation(true) + getMassMatrix / getCoriolisMatrix,
 *      algorithms/CompositeRigidBodyMassMatrixCalculator.java:271-274, 344-365, 604-630, 669-768; algorithms/FactorizedBodyInertia.java) ----
 * H_out and C_out [B][nv][nv] row-major (MH_LAYOUT_SOA: [nv*nv][B]):  tau = H qdd + C qd + G.  C is the reference's matrix entry for entry
 * (the factorisation B = v x* I of the body-level Coriolis terms: C qd = RNEA(q, qd, qdd = 0, g = 0), dH/dt = C + C^T); entries of
 * unrelated joints are zero.  Device pointers, asynchronous on opts->stream.  fp64 models with a specialised code object run its
 * compile-time recursion, everything else the run-time-topology kernel.  For big batches MH_LAYOUT_SOA outputs are 3x faster to produce
 * (coalesced stores) than the AoS matrices.
 */
mh_status mh_crba_coriolis_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const mh_options *opts, double *H_out,
                               double *C_out);
mh_status mh_crba_coriolis_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const mh_options *opts, float *H_out, float *C_out);

/*
 * ---- centroidal momentum (CompositeRigidBodyMassMatrixCalculator.getCentroidalMomentumMatrix / getCentroidalConvectiveTermMatrix,
 *      algorithms/CompositeRigidBodyMassMatrixCalculator.java:316-342, 375-420, 801-839) ----
 * A_out [B][6][nv]: h = A qd is the momentum (angular, linear) of the considered bodies in the centroidal momentum frame; b_out [B][6]
 * (may be NULL; needs qd otherwise): dh/dt = A qdd + b.  frame[12] (HOST pointer, R row-major then p; NULL = identity) is the pose of the
 * centroidal momentum frame in the root body frame -- the reference's setCentroidalMomentumFrame(ReferenceFrame) for a frame fixed in the
 * root body; the default NULL is the calculator's default frame (the root body-fixed frame, :190-193).  frame_mode
 * MH_CENTROIDAL_FRAME_AT_COM re-centres that frame on the centre of mass of the considered bodies, i.e. a
 * frames/CenterOfMassReferenceFrame whose parent is `frame`; com_out [B][3] (may be NULL) then receives the centre of mass in `frame`
 * coordinates (algorithms/CenterOfMassCalculator.java:70-91), zeros in MH_CENTROIDAL_FRAME_FIXED mode.
 */
enum
{
   MH_CENTROIDAL_FRAME_FIXED = 0,
   MH_CENTROIDAL_FRAME_AT_COM = 1
};
mh_status mh_centroidal_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double frame[12], int32_t frame_mode,
                            const mh_options *opts, double *A_out, double *b_out, double *com_out);
mh_status mh_centroidal_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const double frame[12], int32_t frame_mode,
                            const mh_options *opts, float *A_out, float *b_out, float *com_out);

/*
 * ---- state integration (MultiBodySystemStateIntegrator.doubleIntegrateFromAcceleration, tools/MultiBodySystemStateIntegrator.java:365-441,
 *      503-575, 710-733): the step downstream of forward dynamics, so that a simulation loop never leaves the device ----
 * One explicit constant-acceleration step of size dt for every joint of every configuration: 1-DoF q' = q + dt qd + dt^2/2 qdd,
 * qd' = qd + dt qdd; 6-DoF joints integrate the pose with the rotation vector dt w + dt^2/2 dw appended to the quaternion and
 * re-express twist and acceleration in the new frame after the joint, exactly as the reference does.  q_out [B][nq], qd_out [B][nv]
 * and (optional, may be NULL) qdd_out [B][nv] may alias the inputs (in-place step).  Entries no considered joint owns are not written.
 * Device pointers, asynchronous on opts->stream; opts->layout as for the other calls.
 */
mh_status mh_integrate_f64(mh_model_t model, int64_t B, double dt, const double *q, const double *qd, const double *qdd,
                           const mh_options *opts, double *q_out, double *qd_out, double *qdd_out);
/*
 * One simulation step on the device: qdd_out = ABA(q, qd, tau) followed by one integrator step of size dt, q_next / qd_next =
 * integrate(q, qd, qdd_out).  Same results as mh_aba_f64 followed by mh_integrate_f64 (ForwardDynamicsCalculator.compute +
 * MultiBodySystemStateIntegrator.doubleIntegrateFromAcceleration; the loop of MultiBodySystemStateIntegratorTest.java:245-250).  Models
 * with a tree-split code object, identity index maps and AoS matrices run it as ONE launch (the new state is formed from the rows the
 * forward-dynamics kernel already holds in LDS); every other case issues the two launches.  q_next / qd_next may alias q / qd;
 * qdd_out is the forward-dynamics result (not re-expressed by the step).
 */
mh_status mh_aba_integrate_f64(mh_model_t model, int64_t B, double dt, const double *q, const double *qd, const double *tau,
                               const double gravity[3], const double *f_ext, const mh_options *opts, double *qdd_out, double *q_next,
                               double *qd_next);
mh_status mh_integrate_f32(mh_model_t model, int64_t B, double dt, const float *q, const float *qd, const float *qdd,
                           const mh_options *opts, float *q_out, float *qd_out, float *qdd_out);

mh_status mh_rnea_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *qdd,
                      const double gravity[3], const float *f_ext, const mh_options *opts, float *tau_out);
mh_status mh_aba_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *tau,
                     const double gravity[3], const float *f_ext, const mh_options *opts, float *qdd_out);
mh_status mh_crba_f32(mh_model_t model, int64_t B, const float *q, const mh_options *opts, float *H_out);
/* mh_rnea_aba_f64 in fp32 (BASELINE.json configs[4] evaluates both per step): big batches of wide matrices go through ONE depth-first
 * walk that carries the inverse dynamics through the forward dynamics' first pass (AoS callers: through shared transposed scratch
 * copies of q and qd); otherwise mh_rnea_f32 followed by mh_aba_f32.  The results of those two, to fp32 rounding in the fused walk.
 * The outputs must not overlap the inputs or each other (as for mh_rnea_aba_f64). */
mh_status mh_rnea_aba_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *qdd, const float *tau,
                          const double gravity[3], const float *f_ext, const mh_options *opts, float *tau_out, float *qdd_out);
/* fp32 forms of the per-body outputs and of forward dynamics with acceleration-source joints (run-time-topology kernels) */
mh_status mh_rnea_bodies_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *qdd, const double gravity[3],
                             const float *f_ext, const mh_options *opts, float *tau_out, float *body_acc_out, float *body_twist_out);
mh_status mh_aba_bodies_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *tau, const double gravity[3],
                            const float *f_ext, const mh_options *opts, float *qdd_out, float *body_acc_out, float *body_twist_out);
mh_status mh_aba_locked_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *tau, const float *qdd_in,
                            const double gravity[3], const float *f_ext, const mh_options *opts, float *qdd_out, float *tau_out);

/*
 * ---- compute, HOST pointers (what a JNI / Panama shim with heap or off-heap arrays calls) ----
 * Same arguments, all pointers in host memory; the call copies in, launches, copies out and
 * synchronises before returning.
 */
mh_status mh_rnea_f64_host(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd,
                           const double gravity[3], const double *f_ext, const mh_options *opts, double *tau_out);
mh_status mh_aba_f64_host(mh_model_t model, int64_t B, const double *q, const double *qd, const double *tau,
                          const double gravity[3], const double *f_ext, const mh_options *opts, double *qdd_out);
mh_status mh_crba_f64_host(mh_model_t model, int64_t B, const double *q, const mh_options *opts, double *H_out);
mh_status mh_rnea_f32_host(mh_model_t model, int64_t B, const float *q, const float *qd, const float *qdd, const double gravity[3],
                           const float *f_ext, const mh_options *opts, float *tau_out);
mh_status mh_aba_f32_host(mh_model_t model, int64_t B, const float *q, const float *qd, const float *tau, const double gravity[3],
                          const float *f_ext, const mh_options *opts, float *qdd_out);
mh_status mh_crba_f32_host(mh_model_t model, int64_t B, const float *q, const mh_options *opts, float *H_out);
/* tau_out = RNEA(q, qd, qdd) and qdd_out = ABA(q, qd, tau) of the same configurations (mh_rnea_aba_f64 per chunk) */
mh_status mh_rnea_aba_f64_host(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double *tau,
                               const double gravity[3], const double *f_ext, const mh_options *opts, double *tau_out, double *qdd_out);
/*
 * AoS batches above 1024 configurations travel in chunks through three streams (copy-in, kernels, copy-out overlap).  The copies
 * reach PCIe rate only from / to PINNED host memory: let the host keep its state matrices in memory from mh_host_alloc (Panama: wrap the
 * returned address as a MemorySegment; JNI: NewDirectByteBuffer), or pin existing off-heap buffers once with mh_host_register.
 * Pageable pointers work too, at the runtime's staged-copy rate.
 */
mh_status mh_host_alloc(size_t bytes, void **ptr_out); /* hipHostMalloc */
mh_status mh_host_free(void *ptr);
mh_status mh_host_register(void *ptr, size_t bytes);   /* hipHostRegister: pins a range the host already owns */
mh_status mh_host_unregister(void *ptr);

mh_status mh_crba_coriolis_f64_host(mh_model_t model, int64_t B, const double *q, const double *qd, const mh_options *opts, double *H_out,
                                    double *C_out);
mh_status mh_centroidal_f64_host(mh_model_t model, int64_t B, const double *q, const double *qd, const double frame[12], int32_t frame_mode,
                                 const mh_options *opts, double *A_out, double *b_out, double *com_out);

/*
 * ---- device memory for hosts without a HIP binding of their own ----
 * A Java shim keeps its state matrices on the device with these and calls the DEVICE-pointer entry points: a simulation loop
 * (mh_aba_integrate_f64 per step) then never crosses PCIe.  Copies are asynchronous on `stream` (NULL = the null stream); pageable
 * host memory is staged by the runtime, pinned memory (mh_host_alloc) is read / written in place -- keep it alive until
 * mh_stream_synchronize returns.
 */
mh_status mh_device_alloc(size_t bytes, void **ptr_out);
mh_status mh_device_free(void *ptr);
mh_status mh_copy_to_device(void *dst_device, const void *src_host, size_t bytes, void *stream);
mh_status mh_copy_to_host(void *dst_host, const void *src_device, size_t bytes, void *stream);
mh_status mh_stream_synchronize(void *stream);

/*
 * ---- multi-GPU: one process per GPU, the batch sharded by rows, RCCL over xGMI ----
 * The reference is single-threaded Java (InverseDynamicsCalculatorTest.java:124-158 times one calculator on one thread) and has no
 * counterpart.  Every configuration is independent and the model is read-only, so the compute entry points above need no collective:
 * each rank calls them on its own rows.  What a host without torch.distributed needs around that (mecano_amd/distributed.py is the
 * same for the Python host) is
 *   mh_shard_range          which rows of a batch a rank owns (contiguous; sizes differ by at most one);
 *   mh_comm_broadcast_host  the robot description (the arrays of mh_model_desc, packed by the host) from the rank that has it;
 *   mh_comm_all_gather_rows the ranks' output rows side by side on every rank, once, after the steps.
 * Rank 0 calls mh_comm_unique_id and carries the MH_COMM_ID_BYTES bytes to the other processes by its own means (a file, a socket, the
 * launcher's environment); every process then calls mh_comm_create on the device it computes on.  librccl.so.1 is opened at the first
 * of these calls (MH_RCCL_LIBRARY overrides the name): MH_ERR_NO_DEVICE when it cannot be.  A communicator is used by one thread at a time.
 */
#define MH_COMM_ID_BYTES 128
typedef struct mh_comm *mh_comm_t;
mh_status mh_shard_range(int64_t B, int32_t rank, int32_t world, int64_t *lo_out, int64_t *hi_out);
mh_status mh_comm_unique_id(void *id_out); /* MH_COMM_ID_BYTES bytes */
mh_status mh_comm_create(const void *id, int32_t rank, int32_t world, mh_comm_t *comm_out); /* collective: every rank, same id */
mh_status mh_comm_destroy(mh_comm_t comm);
mh_status mh_comm_size(mh_comm_t comm, int32_t *rank_out, int32_t *world_out); /* as the communicator itself counted them */
mh_status mh_comm_broadcast(mh_comm_t comm, void *device_buf, size_t bytes, int32_t root, void *stream); /* in place, asynchronous */
mh_status mh_comm_broadcast_host(mh_comm_t comm, void *host_buf, size_t bytes, int32_t root); /* staged through the device, synchronous */
/* local_rows: [hi - lo][row_bytes] of mh_shard_range(B_total, rank, world), device; all_rows_out: [B_total][row_bytes], device, on every
 * rank.  Asynchronous on `stream`.  Ragged shards travel as they are (one grouped operation, no padding). */
mh_status mh_comm_all_gather_rows(mh_comm_t comm, const void *local_rows, int64_t B_total, size_t row_bytes, void *all_rows_out, void *stream);
/* The operations mh_comm_all_gather_rows issues on `rank` of `world`, without issuing them (host-only: no RCCL, no device).  Step k is a
 * broadcast of `bytes` bytes from rank `root` into [recv_offset, recv_offset + bytes) of every rank's output, in place except on the root
 * (send_local = 1: the bytes come from its local rows); equal shards give ONE step with root = -1, the plain all-gather.  Every rank gets
 * the same list (roots, sizes, offsets, order).  steps may be NULL with cap = 0 to ask for the count only (at most world steps). */
typedef struct mh_gather_step
{
   int32_t root;
   int32_t send_local;
   int64_t recv_offset;
   int64_t bytes;
} mh_gather_step;
mh_status mh_comm_gather_plan(int64_t B_total, size_t row_bytes, int32_t rank, int32_t world, int32_t force_ragged, mh_gather_step *steps,
                              int32_t cap, int32_t *n_steps_out);
mh_status mh_comm_barrier(mh_comm_t comm, void *stream); /* every rank has arrived and `stream` has drained */

/* ---- measurement helper: HIP-event timing of launches on a stream (bench.py, §8d timing protocol) ---- */
typedef struct mh_timer *mh_timer_t;
mh_status mh_timer_create(mh_timer_t *timer_out);
void mh_timer_destroy(mh_timer_t timer);
mh_status mh_timer_start(mh_timer_t timer, void *stream);
mh_status mh_timer_stop(mh_timer_t timer, void *stream);
mh_status mh_timer_elapsed_ms(mh_timer_t timer, float *ms_out); /* synchronises on the stop event */

#ifdef __cplusplus
}
#endif
#endif /* MECANO_HIP_H */
