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
//     stack a lane needs is (deepest path) x (frame), not (bodies) x (record): 146 slots instead of ~2500 on the 128-body tree.  Every
//     frame has a HOME chosen by the host (mh_api.hip: dfs_plan): LDS (slot-major, [slot][64 lanes]: conflict-free) or this wave's block
//     of a global slot-major workspace.  A slot number in the body's record carries its home (DFS_LDS bit).  The host fills LDS from the
//     leaves upwards: a frame goes to LDS when it still fits on top of the deepest LDS path below it, so with a budget of 20 KB per wave
//     (8 waves per CU) the many small subtrees near the leaves -- most of the frames, hence most of the traffic -- never leave the CU,
//     and only the few bodies near the root use the global stack.  Measured on the 128-body tree the all-global stack moved 5 (RNEA) and
//     18 (ABA) times the algorithmic bytes through HBM at 4-6 TB/s: those kernels were bound by their own workspace;
//   * RNEA needs nothing else.  ABA's inward sweep (passes one and two fused into the walk) hands U, 1/D, u' = u - U.c (and cos, sin, c
//     for bodies with children) per body to the outward sweep: 8..16 values per body instead of 30+, in LDS for small models, in the
//     global workspace otherwise.
//
// lane = configuration; state matrices are read per lane with the caller's strides (AoS rows or SoA columns alike: consecutive bodies
// of the walk read consecutive row entries, so every fetched line is consumed before it leaves L2 -- no transposed scratch copies).
// Same arithmetic primitives as the other kernels (mh_device.h, mh_kernels.h); per-body outputs, joint wrenches and
// acceleration-source joints stay on the sweep kernels of mh_kernels.h.
#pragma once
#include <type_traits>
#include "mh_kernels.h"

namespace mh
{
// ---- event program
enum : int
{
   EV_POP = 1,          // else VISIT
   EV_PARENT_REGS = 2,  // VISIT: the previous event was VISIT(parent): its state is in registers
   EV_LEAF = 2,         // POP: the previous event was VISIT(this body): its state is in registers
   EV_LAST_CHILD = 4,   // POP: hand the contribution over in registers (the carry): the parent's last child with children of its own -- or its
                        // first child when all are leaves; only leaf siblings follow before POP(parent), and a leaf's events leave the carry alone
   EV_ACC_FIRST = 8,    // POP (ABA): first contribution to the parent's accumulator: store, do not add
   EV_CARRY_ADD = 16,   // POP of a leaf sibling behind the EV_LAST_CHILD one: ADD the contribution to the carry (round 5: of the 64 child pops
                        // of the 128-body tree of configs[4] that read and wrote 27-33 accumulator slots of their parent's frame, 38 are such leaves)
   EV_ACC_USED = 32,    // POP of a body with children: some child went through the frame's accumulators (else they are never read)
   EV_BODY_SHIFT = 8
};
// stack-frame slots a joint transform takes (revolute: cos, sin; prismatic: q; fixed: nothing)
__host__ __device__ constexpr int jx_slots(int type) { return type == JT_REVOLUTE ? 2 : (type == JT_PRISMATIC ? 1 : (general_x(type) ? 12 : 0)); }
__host__ __device__ constexpr int rnea_frame_slots(int type, int n_children)
{ // [f 6][jx][v, a 12 when later children re-read them]
   return n_children == 0 ? 0 : 6 + jx_slots(type) + (n_children >= 2 ? 12 : 0);
}
__host__ __device__ constexpr int aba_frame_slots(int type, int n_children)
{ // [p 6][w 6: written when several children follow][jx][v 6, articulated-inertia accumulator 21 + bias accumulator 6 when several children contribute]
   return n_children == 0 ? 0 : 12 + jx_slots(type) + (n_children >= 2 ? 33 : 0);
}
// frame of the fused RNEA + ABA walk (aba_dfs_kernel<.., PAIR>): the forward dynamics' frame, then the inverse dynamics' wrench [f 6] and,
// when later children re-read it, its acceleration [a 6] (jx and v are the forward dynamics' own)
__host__ __device__ constexpr int pair_frame_slots(int type, int n_children)
{
   return n_children == 0 ? 0 : aba_frame_slots(type, n_children) + 6 + (n_children >= 2 ? 6 : 0);
}
__host__ __device__ constexpr int aba_hand_slots(int type, int n_children)
{ // inward -> outward hand-over: U (6), 1 / D, u (+ cos, sin) of a 1-DoF joint; the body's a~ of a floating one; 3 U, D^-1 (6), u (3) of a
  // 3-DoF one.  (Until round 5 bodies with children also handed their bias acceleration c over: it is absorbed into the bias wrenches now.)
   (void)n_children;
   return type == JT_REVOLUTE ? 10 : (type == JT_PRISMATIC ? 8 : (type == JT_SIXDOF ? 6 : (type == JT_FIXED ? 0 : 27)));
}

// ---- the per-lane depth stack: slot codes carry the frame's home
enum : int
{
   DFS_LDS = 1 << 20,      // slot code: the frame lives in LDS (else in the wave's block of the global workspace)
   DFS_SLOT = DFS_LDS - 1
};
template <typename T>
using dfs_lds_ptr = T __attribute__((address_space(3))) *;
// MODE: 0 = every frame in LDS, 1 = every frame in the global block (no branch in either), 2 = per frame, as the slot code says
template <typename T, int MODE>
struct DStack
{ // [slot][64 lanes] blocks, this lane's column.  The two homes are pointers of different address spaces on purpose: with two generic
  // pointers the compiler merges the branches of a group into flat_load / flat_store through a selected pointer
   static constexpr int mode = MODE;
   dfs_lds_ptr<T> lds;
   T *glb;
};
// one wave-uniform branch per GROUP of accesses (the code is a scalar): ds_* on one side, global_* on the other, never flat
#define MH_ST_GROUP(code, ...)                               \
   if (SK::mode == 0 || (SK::mode == 2 && ((code)&DFS_LDS))) \
   {                                                         \
      const dfs_lds_ptr<T> sp = S.lds + ((code)&DFS_SLOT) * 64; \
      __VA_ARGS__                                            \
   }                                                         \
   else                                                      \
   {                                                         \
      T *const sp = S.glb + (long)(code)*64;                 \
      __VA_ARGS__                                            \
   }
#define MH_SP(k) sp[(k)*64]
template <typename T, class SK>
MH_DEV void st_store6(const SK &S, int code, const SV<T> &v)
{
   MH_ST_GROUP(code, MH_SP(0) = v.a.x; MH_SP(1) = v.a.y; MH_SP(2) = v.a.z; MH_SP(3) = v.l.x; MH_SP(4) = v.l.y; MH_SP(5) = v.l.z;)
}
template <typename T, class SK>
MH_DEV SV<T> st_load6(const SK &S, int code)
{
   SV<T> v;
   MH_ST_GROUP(code, v.a.x = MH_SP(0); v.a.y = MH_SP(1); v.a.z = MH_SP(2); v.l.x = MH_SP(3); v.l.y = MH_SP(4); v.l.z = MH_SP(5);)
   return v;
}
template <typename T, class SK>
MH_DEV void st_add6(const SK &S, int code, const SV<T> &v)
{
   MH_ST_GROUP(code, MH_SP(0) += v.a.x; MH_SP(1) += v.a.y; MH_SP(2) += v.a.z; MH_SP(3) += v.l.x; MH_SP(4) += v.l.y; MH_SP(5) += v.l.z;)
}
template <typename T, class SK>
MH_DEV void st_store_jx(const SK &S, int code, int type, const JX<T> &jx)
{
   if (type == JT_REVOLUTE)
   {
      MH_ST_GROUP(code, MH_SP(0) = jx.c; MH_SP(1) = jx.s;)
   }
   else if (type == JT_PRISMATIC)
   {
      MH_ST_GROUP(code, MH_SP(0) = jx.d;)
   }
   else if (general_x(type))
   {
      MH_ST_GROUP(code, MH_SP(0) = jx.X.R.xx; MH_SP(1) = jx.X.R.xy; MH_SP(2) = jx.X.R.xz; MH_SP(3) = jx.X.R.yx; MH_SP(4) = jx.X.R.yy; MH_SP(5) = jx.X.R.yz;
                  MH_SP(6) = jx.X.R.zx; MH_SP(7) = jx.X.R.zy; MH_SP(8) = jx.X.R.zz; MH_SP(9) = jx.X.p.x; MH_SP(10) = jx.X.p.y; MH_SP(11) = jx.X.p.z;)
   }
}
template <typename T, class SK>
MH_DEV JX<T> st_load_jx(const SK &S, int code, int type)
{
   JX<T> jx;
   jx.c = T(1), jx.s = T(0), jx.d = T(0);
   if (type == JT_REVOLUTE)
   {
      MH_ST_GROUP(code, jx.c = MH_SP(0); jx.s = MH_SP(1);)
   }
   else if (type == JT_PRISMATIC)
   {
      MH_ST_GROUP(code, jx.d = MH_SP(0);)
   }
   else if (general_x(type))
   {
      MH_ST_GROUP(code, jx.X.R = M3<T>{MH_SP(0), MH_SP(1), MH_SP(2), MH_SP(3), MH_SP(4), MH_SP(5), MH_SP(6), MH_SP(7), MH_SP(8)};
                  jx.X.p = V3<T>{MH_SP(9), MH_SP(10), MH_SP(11)};)
   }
   return jx;
}
template <typename T, class SK>
MH_DEV void st_store_abi(const SK &S, int code, const ABI<T> &I)
{
   MH_ST_GROUP(code, MH_SP(0) = I.A.xx; MH_SP(1) = I.A.xy; MH_SP(2) = I.A.xz; MH_SP(3) = I.A.yy; MH_SP(4) = I.A.yz; MH_SP(5) = I.A.zz;
               MH_SP(6) = I.L.xx; MH_SP(7) = I.L.xy; MH_SP(8) = I.L.xz; MH_SP(9) = I.L.yy; MH_SP(10) = I.L.yz; MH_SP(11) = I.L.zz;
               MH_SP(12) = I.C.xx; MH_SP(13) = I.C.xy; MH_SP(14) = I.C.xz; MH_SP(15) = I.C.yx; MH_SP(16) = I.C.yy; MH_SP(17) = I.C.yz;
               MH_SP(18) = I.C.zx; MH_SP(19) = I.C.zy; MH_SP(20) = I.C.zz;)
}
template <typename T, class SK>
MH_DEV ABI<T> st_load_abi(const SK &S, int code)
{
   ABI<T> I;
   MH_ST_GROUP(code, I.A = S3<T>{MH_SP(0), MH_SP(1), MH_SP(2), MH_SP(3), MH_SP(4), MH_SP(5)};
               I.L = S3<T>{MH_SP(6), MH_SP(7), MH_SP(8), MH_SP(9), MH_SP(10), MH_SP(11)};
               I.C = M3<T>{MH_SP(12), MH_SP(13), MH_SP(14), MH_SP(15), MH_SP(16), MH_SP(17), MH_SP(18), MH_SP(19), MH_SP(20)};)
   return I;
}
// accumulator of a body with several children: (articulated inertia 21, bias wrench 6) += contribution
template <typename T, class SK>
MH_DEV void st_add_abi6(const SK &S, int code, const ABI<T> &I, const SV<T> &p)
{
   MH_ST_GROUP(code, MH_SP(0) += I.A.xx; MH_SP(1) += I.A.xy; MH_SP(2) += I.A.xz; MH_SP(3) += I.A.yy; MH_SP(4) += I.A.yz; MH_SP(5) += I.A.zz;
               MH_SP(6) += I.L.xx; MH_SP(7) += I.L.xy; MH_SP(8) += I.L.xz; MH_SP(9) += I.L.yy; MH_SP(10) += I.L.yz; MH_SP(11) += I.L.zz;
               MH_SP(12) += I.C.xx; MH_SP(13) += I.C.xy; MH_SP(14) += I.C.xz; MH_SP(15) += I.C.yx; MH_SP(16) += I.C.yy; MH_SP(17) += I.C.yz;
               MH_SP(18) += I.C.zx; MH_SP(19) += I.C.zy; MH_SP(20) += I.C.zz; MH_SP(21) += p.a.x; MH_SP(22) += p.a.y; MH_SP(23) += p.a.z;
               MH_SP(24) += p.l.x; MH_SP(25) += p.l.y; MH_SP(26) += p.l.z;)
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
// multi-DoF joints: joint transform / S x from entries already read into registers (same arithmetic as joint_from_q / joint_vec)
template <typename T>
MH_DEV JX<T> joint_of_vals(int type, const T *q)
{
   JX<T> jx;
   jx.c = T(1), jx.s = T(0), jx.d = T(0);
   if (type == JT_SIXDOF)
   {
      jx.X.R = quat_to_R(q[0], q[1], q[2], q[3]);
      jx.X.p = V3<T>{q[4], q[5], q[6]};
   }
   else if (type == JT_SPHERICAL)
   {
      jx.X.R = quat_to_R(q[0], q[1], q[2], q[3]);
      jx.X.p = V3<T>{T(0), T(0), T(0)};
   }
   else if (type == JT_PLANAR)
   { // rotation about y by the pitch, translation (x, 0, z)
      T sp, cp;
      sincos_t(q[0], sp, cp);
      jx.X.R = M3<T>{cp, T(0), sp, T(0), T(1), T(0), -sp, T(0), cp};
      jx.X.p = V3<T>{q[1], T(0), q[2]};
   }
   return jx;
}
template <typename T>
MH_DEV SV<T> joint_vec_vals(int type, const T *x, bool enabled)
{
   SV<T> o{V3<T>{T(0), T(0), T(0)}, V3<T>{T(0), T(0), T(0)}};
   if (!enabled)
      return o;
   if (type == JT_SIXDOF)
      o.a = V3<T>{x[0], x[1], x[2]}, o.l = V3<T>{x[3], x[4], x[5]};
   else if (type == JT_SPHERICAL)
      o.a = V3<T>{x[0], x[1], x[2]};
   else if (type == JT_PLANAR)
      o.a.y = x[0], o.l.x = x[1], o.l.z = x[2];
   return o;
}

// ---- AoS rows through LDS windows (WIN).  With [B][n] matrices a lane's entries are n * sizeof(T) bytes away from its neighbour's: every
// wave-load touches 64 cache lines for 64 useful words, and the 128-body tree makes 1331 such loads per configuration (measured: 1300 us
// against 550 us on SoA matrices).  The walk consumes a matrix row in engine order, so the wave keeps a WINDOW of ROW_WIN consecutive
// entries of its 64 rows in LDS, refilled by coalesced loads (two rows x 128 bytes per wave-instruction) and read back transposed
// (pitch 65: conflict-free both ways).  Needs identity index maps (entry k of engine order is column k); SoA matrices and permuted maps
// read directly.
constexpr int ROW_WIN = 16;
constexpr int ROW_PITCH = 65;
// One matrix's window.  The NEXT window's entries are already on their way into registers (`pre`, 16 per lane) while the current one is
// consumed from LDS: when the walk steps past the window, the registers are committed to LDS and the window after that is requested --
// no exposed memory latency on a sequential walk; a jump (a walk that re-reads, a matrix consumed out of order) refills synchronously.
template <typename T>
struct RowWindow
{
   T *win;         // LDS, [ROW_WIN][ROW_PITCH]
   const T *mat;   // AoS matrix [B][n_cols]
   long n_cols, cfg0, B;
   int w0, pre_w0; // start of the window in LDS / of the one in flight (-huge: none)
   T pre[ROW_WIN];
   MH_DEV void init(T *lds, const T *matrix, long cols, long first_cfg, long batch)
   {
      win = lds, mat = matrix, n_cols = cols, cfg0 = first_cfg, B = batch;
      w0 = pre_w0 = -(1 << 30);
#pragma unroll
      for (int i = 0; i < ROW_WIN; i++)
         pre[i] = T(0);
   }
   MH_DEV void request(int start, int tid)
   { // lane (r, e) = (4 i + tid / 16, tid % 16): four rows x 64 contiguous bytes per wave-instruction
      pre_w0 = start;
#pragma unroll
      for (int i = 0; i < ROW_WIN; i++)
      {
         const long cfg = cfg0 + 4 * i + (tid >> 4);
         const int col = start + (tid & 15);
         pre[i] = (cfg < B && col < n_cols) ? mat[cfg * n_cols + col] : T(0);
      }
   }
   MH_DEV void commit(int tid)
   {
#pragma unroll
      for (int i = 0; i < ROW_WIN; i++)
         win[(tid & 15) * ROW_PITCH + 4 * i + (tid >> 4)] = pre[i];
      w0 = pre_w0;
      __syncthreads(); // one wave per workgroup: orders all lanes' LDS writes before the transposed reads
   }
   MH_DEV T get(int row, int tid)
   {
      if ((unsigned)(row - w0) >= (unsigned)ROW_WIN)
      {
         if ((unsigned)(row - pre_w0) >= (unsigned)ROW_WIN)
            request(row, tid); // not the window in flight: fetch the one that starts here
         commit(tid);
         if (w0 + ROW_WIN < n_cols)
            request(w0 + ROW_WIN, tid);
      }
      return win[(row - w0) * ROW_PITCH + tid];
   }
};

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
// The depth stack: m.rnea_stack slots of LDS ([slot][64]) + this wave's block of the global workspace A.ws ([wave][slot][64]); the slot
// codes in the bodies' records say which (DStack).
// WIN: AoS state rows are read through LDS windows (identity index maps), see window_refill.
// the joint kind of an event as a run-time value (the per-kind dispatch passes std::integral_constant instead)
struct KindRt
{
   int v;
   MH_DEV operator int() const { return v; }
};
template <typename T, bool WIN, int MODE>
__global__ void __launch_bounds__(64) rnea_dfs_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   const ciptr meta = as_const(m.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map), prog = as_const(m.prog);
   const int tid = threadIdx.x;
   const long nlanes = (long)gridDim.x * 64;
   const V3<T> Z{T(0), T(0), T(0)};
   T *const wq = (T *)lds_raw + (long)m.rnea_stack * 64, *const wv = wq + ROW_WIN * ROW_PITCH, *const wx = wv + ROW_WIN * ROW_PITCH;
   const DStack<T, MODE> S{(dfs_lds_ptr<T>)lds_raw + tid, A.ws + (long)blockIdx.x * A.ws_stride + tid};
   {
      // wave-uniform loop over groups of 64 configurations: the lanes of a ragged last group repeat its last configuration (no store)
      for (long cfg0 = (long)blockIdx.x * 64; cfg0 < A.B; cfg0 += nlanes)
      {
         const bool active = cfg0 + tid < A.B;
         const long cfg = active ? cfg0 + tid : A.B - 1;
         const T *qrow = A.q + cfg * A.q_bs;
         const T *qdrow = A.qd + cfg * A.v_bs;
         const T *qddrow = A.in3 + cfg * A.v_bs;
         const T *frow = A.fext ? A.fext + cfg * A.f_bs : nullptr;
         T *trow = A.out + cfg * A.v_bs;
         // entry `row` of this lane's configuration / velocity-sized rows (row = engine-order entry = matrix column with identity maps)
         RowWindow<T> Wq, Wv, Wx;
         if constexpr (WIN)
            Wq.init(wq, A.q, m.nq, cfg0, A.B), Wv.init(wv, A.qd, m.nv, cfg0, A.B), Wx.init(wx, A.in3, m.nv, cfg0, A.B);
         auto getq = [&](int row) -> T {
            if constexpr (WIN)
               return Wq.get(row, tid);
            else
               return qrow[row * A.q_es];
         };
         auto getv = [&](int row, T &vd, T &xd) {
            if constexpr (WIN)
               vd = A.coriolis ? Wv.get(row, tid) : T(0), xd = A.accel ? Wx.get(row, tid) : T(0);
            else
            {
               const long r = row * A.v_es;
               vd = A.coriolis ? qdrow[r] : T(0), xd = A.accel ? qddrow[r] : T(0);
            }
         };
         SV<T> v_reg{Z, Z}, a_reg{Z, Z}, f_reg{Z, Z}, carry{Z, Z};
         JX<T> jx_reg;
         jx_reg.c = T(1), jx_reg.s = T(0), jx_reg.d = T(0);
         In<T> nxt{T(0), T(0), T(0)};
         auto prefetch = [&](int e1) { // the inputs event e1 will consume (a VISIT of a 1-DoF joint: q, qd, qdd; a POP: nothing)
            if (e1 >= m.n_events)
               return;
            const int ev1 = prog[e1];
            if (ev1 & EV_POP)
               return;
            ciptr m1 = meta + (ev1 >> EV_BODY_SHIFT) * MI_STRIDE;
            if (one_dof(m1[MI_TYPE]))
            {
               nxt.q = getq(m1[MI_ROW_Q]);
               getv(m1[MI_ROW_V], nxt.v, nxt.x);
            }
         };
         prefetch(0);
         for (int e = 0; e < m.n_events; e++)
         {
            const int ev = prog[e];
            const int j = ev >> EV_BODY_SHIFT;
            ciptr mi = meta + j * MI_STRIDE;
            const int parent = mi[MI_PARENT], type_rt = mi[MI_TYPE], nch = mi[MI_NCH], fr = mi[MI_DFS_R];
            // one dispatch on the joint kind per event, then straight-line code of that kind (the kind as a compile-time constant folds
            // every test below: the run-time form was a maze of ~150 basic blocks per event)
            auto body = [&](auto kind) {
            const int type = kind; // a compile-time constant after inlining, except for KindRt
            const CRef<T, false> c{CB + j * MC_STRIDE};
            const XF<T> Xb = load_xb<T>(c);
            const In<T> in = nxt;
            JX<T> jxm;
            SV<T> vJm{Z, Z}, aJm{Z, Z};
            if (!(ev & EV_POP) && !one_dof(type) && type != JT_FIXED)
            { // a multi-DoF joint reads its entries now, BEFORE the next event's prefetch may move the windows on
               T mq[7], mv[6], mx[6];
               const int nc = cfg_count(type), nd = dof_count(type);
#pragma unroll
               for (int k = 0; k < 7; k++)
                  mq[k] = k < nc ? getq(WIN ? mi[MI_CFG] + k : cfg_map[mi[MI_CFG] + k]) : T(0);
#pragma unroll
               for (int k = 0; k < 6; k++)
               {
                  mv[k] = T(0), mx[k] = T(0);
                  if (k < nd)
                     getv(WIN ? mi[MI_DOF] + k : dof_map[mi[MI_DOF] + k], mv[k], mx[k]);
               }
               jxm = joint_of_vals<T>(type, mq);
               vJm = joint_vec_vals<T>(type, mv, A.coriolis != 0), aJm = joint_vec_vals<T>(type, mx, A.accel != 0);
            }
            prefetch(e + 1);
            if (!(ev & EV_POP))
            { // ---- VISIT: velocity, acceleration, Newton-Euler wrench of the body (InverseDynamicsCalculator.java:873-917)
               SV<T> vp, ap;
               if (parent < 0)
               {
                  vp = SV<T>{Z, Z};
                  ap = root_acceleration(A); // :343-348
               }
               else if (ev & EV_PARENT_REGS)
                  vp = v_reg, ap = a_reg;
               else
                  vp = st_load6<T>(S, mi[MI_PVA_R]), ap = st_load6<T>(S, mi[MI_PVA_R] + 6);
               JX<T> jx;
               SV<T> vJ{Z, Z}, aJ{Z, Z};
               jx.c = T(1), jx.s = T(0), jx.d = T(0);
               if (type == JT_REVOLUTE)
               {
                  sincos_t(in.q, jx.s, jx.c);
                  vJ.a.z = in.v, aJ.a.z = in.x;
               }
               else if (type == JT_PRISMATIC)
                  jx.d = in.q, vJ.l.z = in.v, aJ.l.z = in.x;
               else if (type != JT_FIXED)
                  jx = jxm, vJ = vJm, aJ = aJm;
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
                  st_store6<T>(S, fr, f);
                  st_store_jx<T>(S, fr + 6, type, jx);
                  if (nch >= 2)
                     st_store6<T>(S, fr + 6 + jx_slots(type), v), st_store6<T>(S, fr + 12 + jx_slots(type), a);
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
                  f = st_load6<T>(S, fr) + carry;
                  jx = st_load_jx<T>(S, fr + 6, type);
               }
               if (active)
                  write_joint_rows<T>(type, dof_map + mi[MI_DOF], trow, A.v_es, f);
               if (parent >= 0)
               {
                  const SV<T> fp = force_up(type, jx, Xb, f);
                  if (ev & EV_LAST_CHILD)
                     carry = fp;
                  else if (ev & EV_CARRY_ADD)
                     carry = carry + fp;
                  else
                     st_add6<T>(S, mi[MI_PFR_R], fp);
               }
            }
            }; // body
            if constexpr (WIN) // (the build with the LDS windows has no registers to spare for six copies of the body: 761 -> 1415 us)
               body(KindRt{type_rt});
            else
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
}

// ============================================================================================ ABA
// The depth stack: m.aba_stack slots of LDS + the wave's block of the global workspace (slot codes, DStack).  HND_LDS: the inward ->
// outward hand-over lives in LDS behind the stack, else at the start of the wave's global block (the global part of the stack follows).
// WIN: the inward sweep reads q and qd of AoS matrices through LDS windows (window_refill); tau (consumed in post-order), the outward
// sweep's re-reads of q and the accelerations written are per-lane accesses.
// PAIR (round 5; mh_rnea_aba_f32 on big batches): the SAME walk also carries the inverse dynamics of (q, qd, A.in3 = qdd) -> A.out = tau,
// while the forward dynamics reads A.in3b = tau and writes A.outb = qdd.  Pass one of the forward dynamics already forms what the inverse
// dynamics' outward sweep needs -- joint transform, velocity v, bias acceleration c = v x vJ, bias wrench p = v x* I v - f_ext
// (ForwardDynamicsCalculator.java:1085-1127 against InverseDynamicsCalculator.java:873-917) -- so the inverse dynamics costs one more
// motion transform (its acceleration), one product I a and, inwards, one force transform per body, and q / qd are read once.  Frames:
// pair_frame_slots; MI_PFR_R / MI_PVA_R hold the parent's wrench / acceleration slots of the inverse dynamics (dfs_plan, algo 2).
// OCC3: a register budget for three waves per SIMD (168 VGPRs; the plain build takes 181-184 and keeps two).  The walk waits on its own
// workspace more than it computes, so a third wave per SIMD pays where the batch has one to offer: forward dynamics at 524 288
// configurations 3.86 -> 3.38 ms, at 1 M 7.35 -> 6.48 (profiles/r05_c5_occ.txt), for 48 bytes of scratch per lane -- which cost 2 % where
// only eight waves per CU exist anyway (131 072), so the host picks the build by batch size (dfs_choose).  The fused pair walk would need
// 176 bytes of scratch for the same budget: 2 x slower, measured (profiles/r05_c5_occ_pair3.txt), not built.
template <typename T, bool HND_LDS, bool WIN, int MODE, bool PAIR = false, bool OCC3 = false>
__global__ void __launch_bounds__(64, OCC3 ? 3 : 1) aba_dfs_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   const DevModel &m = A.m;
   const T *CB = (const T *)m.consts;
   const ciptr meta = as_const(m.meta), dof_map = as_const(m.dof_map), cfg_map = as_const(m.cfg_map), prog = as_const(m.prog);
   const int tid = threadIdx.x;
   const long nlanes = (long)gridDim.x * 64;
   const V3<T> Z{T(0), T(0), T(0)};
   T *const wq = (T *)lds_raw + ((long)m.aba_stack + (HND_LDS ? (long)m.aba_hand : 0)) * 64, *const wv = wq + ROW_WIN * ROW_PITCH;
   T *const glb = A.ws + (long)blockIdx.x * A.ws_stride + tid; // this wave's block of the global workspace, [slot][64 lanes]
   const DStack<T, MODE> S{(dfs_lds_ptr<T>)lds_raw + tid, glb + (HND_LDS ? 0 : (long)m.aba_hand * 64)};
   auto walk = [&](auto hd) {
#define MH_HD(slot) hd[(long)(slot)*64]
      for (long cfg0 = (long)blockIdx.x * 64; cfg0 < A.B; cfg0 += nlanes)
      {
         const bool active = cfg0 + tid < A.B;
         const long cfg = active ? cfg0 + tid : A.B - 1;
         const T *qrow = A.q + cfg * A.q_bs;
         const T *qdrow = A.qd + cfg * A.v_bs;
         const T *taurow = (PAIR ? A.in3b : A.in3) + cfg * A.v_bs;
         const T *frow = A.fext ? A.fext + cfg * A.f_bs : nullptr;
         T *orow = (PAIR ? A.outb : A.out) + cfg * A.v_bs;
         const T *qddrow = A.in3 + cfg * A.v_bs; // PAIR: the inverse dynamics' accelerations ...
         T *trow = A.out + cfg * A.v_bs;         // ... and efforts
         RowWindow<T> Wq, Wv;
         if constexpr (WIN)
            Wq.init(wq, A.q, m.nq, cfg0, A.B), Wv.init(wv, A.qd, m.nv, cfg0, A.B);
         auto getq = [&](int row) -> T {
            if constexpr (WIN)
               return Wq.get(row, tid);
            else
               return qrow[row * A.q_es];
         };
         auto getv = [&](int row) -> T {
            if constexpr (WIN)
               return Wv.get(row, tid);
            else
               return qdrow[row * A.v_es];
         };
         // ---- inward part: passes one and two (ForwardDynamicsCalculator.java:1085-1254) fused into one depth-first walk
         // Round 5: the velocity-product accelerations are ABSORBED INTO THE BIAS WRENCHES.  With w_j = X w_parent + c_j (the acceleration
         // every body would have with all joint accelerations and the root's at zero) and a_j = a~_j + w_j, the recursion in a~ has no bias
         // acceleration at all: a~_j = X a~_parent + S qdd_j, f_j = IA_j a~_j + (pA_j + IA_j w_j).  So pass one adds I_j w_j to the body's bias
         // wrench (one motion transform and one product I w) and passes two and three run with c = 0 (:1224-1235, :1263-1305): no Ia c,
         // no U.c, and -- what pays -- c is no longer handed from the inward to the outward sweep (6 of the 16 hand-over slots of every
         // 1-DoF body with children, 6 of 12 of a floating one) nor kept in the stack frames.  w rides where c used to (slot fr + 6, for
         // later children of a body that has several).  Same qdd; the efforts of the PAIR walk use the untouched c and p.
         SV<T> v_reg{Z, Z}, p_reg{Z, Z}, w_reg{Z, Z}, pcarry{Z, Z};
         SV<T> ar_reg{Z, Z}, fr_reg{Z, Z}, rcarry{Z, Z}; // PAIR: acceleration and wrench of the inverse dynamics, its hand-up
         JX<T> jx_reg;
         jx_reg.c = T(1), jx_reg.s = T(0), jx_reg.d = T(0);
         ABI<T> Icarry = abi_from_rigid(RI<T>{T(0), Z, S3<T>{T(0), T(0), T(0), T(0), T(0), T(0)}});
         In<T> nxt{T(0), T(0), T(0)};
         auto prefetch = [&](int e1) { // the inputs event e1 will consume (1-DoF joints): a VISIT q and qd, a POP the joint's effort
            if (e1 >= m.n_events)
               return;
            const int ev1 = prog[e1];
            ciptr m1 = meta + (ev1 >> EV_BODY_SHIFT) * MI_STRIDE;
            if (!one_dof(m1[MI_TYPE]))
               return;
            if (ev1 & EV_POP)
               nxt.x = taurow[m1[MI_ROW_V] * A.v_es];
            else
            {
               nxt.q = getq(m1[MI_ROW_Q]), nxt.v = getv(m1[MI_ROW_V]);
               if constexpr (PAIR)
                  nxt.x = qddrow[m1[MI_ROW_V] * A.v_es];
            }
         };
         prefetch(0);
         for (int e = 0; e < m.n_events; e++)
         {
            const int ev = prog[e];
            const int j = ev >> EV_BODY_SHIFT;
            ciptr mi = meta + j * MI_STRIDE;
            const int parent = mi[MI_PARENT], type_rt = mi[MI_TYPE], nch = mi[MI_NCH], fr = mi[MI_DFS_A], hf = mi[MI_HAND];
            auto body = [&](auto kind) { // one dispatch on the joint kind per event (rnea_dfs_kernel)
            const int type = kind;
            const int jxs = jx_slots(type);
            const CRef<T, false> c{CB + j * MC_STRIDE};
            const XF<T> Xb = load_xb<T>(c);
            const In<T> in = nxt;
            JX<T> jxm;
            SV<T> vJm{Z, Z}, aJm{Z, Z};
            if (!(ev & EV_POP) && !one_dof(type) && type != JT_FIXED)
            { // a multi-DoF joint reads its entries now, before the next event's prefetch may move the windows on
               T mq[7], mv[6];
               const int nc = cfg_count(type), nd = dof_count(type);
#pragma unroll
               for (int k = 0; k < 7; k++)
                  mq[k] = k < nc ? getq(WIN ? mi[MI_CFG] + k : cfg_map[mi[MI_CFG] + k]) : T(0);
#pragma unroll
               for (int k = 0; k < 6; k++)
                  mv[k] = k < nd ? getv(WIN ? mi[MI_DOF] + k : dof_map[mi[MI_DOF] + k]) : T(0);
               jxm = joint_of_vals<T>(type, mq);
               vJm = joint_vec_vals<T>(type, mv, true);
               if constexpr (PAIR)
               {
                  T mx[6];
#pragma unroll
                  for (int k = 0; k < 6; k++)
                     mx[k] = k < nd ? qddrow[dof_map[mi[MI_DOF] + k] * A.v_es] : T(0);
                  aJm = joint_vec_vals<T>(type, mx, true);
               }
            }
            const int xr = fr + aba_frame_slots(type, nch); // PAIR: the inverse dynamics' part of the frame
            prefetch(e + 1);
            if (!(ev & EV_POP))
            { // ---- VISIT (:1085-1127): velocity, bias wrench p, bias acceleration c
               SV<T> vp;
               if (parent < 0)
                  vp = SV<T>{Z, Z};
               else if (ev & EV_PARENT_REGS)
                  vp = v_reg;
               else
                  vp = st_load6<T>(S, mi[MI_PV_A]);
               JX<T> jx;
               SV<T> vJ{Z, Z};
               jx.c = T(1), jx.s = T(0), jx.d = T(0);
               if (type == JT_REVOLUTE)
               {
                  sincos_t(in.q, jx.s, jx.c);
                  vJ.a.z = in.v;
               }
               else if (type == JT_PRISMATIC)
                  jx.d = in.q, vJ.l.z = in.v;
               else if (type != JT_FIXED)
                  jx = jxm, vJ = vJm;
               const SV<T> v = motion_down(type, jx, Xb, vp) + vJ;
               const RI<T> I = load_inertia<T>(c);
               SV<T> p0 = crf(v, mul(I, v));
               if (frow)
                  p0 = p0 - load_fext<T>(c, frow, A.f_es, mi[MI_EXT]);
               const SV<T> cj = crm(v, vJ);
               SV<T> wp;
               if (parent < 0)
                  wp = SV<T>{Z, Z};
               else if (ev & EV_PARENT_REGS)
                  wp = w_reg;
               else
                  wp = st_load6<T>(S, mi[MI_PFR_A] + 6);
               const SV<T> w = motion_down(type, jx, Xb, wp) + cj;
               const SV<T> p = p0 + mul(I, w);
               if (nch >= 1)
               {
                  st_store6<T>(S, fr, p);
                  st_store_jx<T>(S, fr + 12, type, jx);
                  if (nch >= 2)
                     st_store6<T>(S, fr + 6, w), st_store6<T>(S, fr + 12 + jxs, v);
               }
               if constexpr (PAIR)
               { // InverseDynamicsCalculator.java:873-917 with this walk's jx, v, c and p: a = X a_parent + aJ + c, f = I a + p
                  SV<T> apr, aJ{Z, Z};
                  if (parent < 0)
                     apr = root_acceleration(A); // :343-348
                  else if (ev & EV_PARENT_REGS)
                     apr = ar_reg;
                  else
                     apr = st_load6<T>(S, mi[MI_PVA_R]);
                  if (type == JT_REVOLUTE)
                     aJ.a.z = in.x;
                  else if (type == JT_PRISMATIC)
                     aJ.l.z = in.x;
                  else if (type != JT_FIXED)
                     aJ = aJm;
                  const SV<T> ar = motion_down(type, jx, Xb, apr) + aJ + cj;
                  const SV<T> fr_ = mul(I, ar) + p0;
                  if (nch >= 1)
                  {
                     st_store6<T>(S, xr, fr_);
                     if (nch >= 2)
                        st_store6<T>(S, xr + 6, ar);
                  }
                  ar_reg = ar, fr_reg = fr_;
               }
               v_reg = v, p_reg = p, w_reg = w, jx_reg = jx;
            }
            else
            { // ---- POP (:1136-1254): articulated inertia and bias wrench of the finished subtree, joint-space quantities, hand-up
               ABI<T> IA = abi_from_rigid(load_inertia<T>(c));
               SV<T> pA;
               JX<T> jx;
               if (ev & EV_LEAF)
                  pA = p_reg, jx = jx_reg;
               else
               {
                  pA = st_load6<T>(S, fr) + pcarry;
                  jx = st_load_jx<T>(S, fr + 12, type);
                  add(IA, Icarry);
                  if (ev & EV_ACC_USED)
                  {
                     add(IA, st_load_abi<T>(S, fr + 18 + jxs));
                     pA = pA + st_load6<T>(S, fr + 39 + jxs);
                  }
               }
               if constexpr (PAIR)
               { // the inverse dynamics' half of the POP (:930-966): joint effort, wrench handed to the parent
                  const SV<T> f = (ev & EV_LEAF) ? fr_reg : st_load6<T>(S, xr) + rcarry;
                  if (active)
                     write_joint_rows<T>(type, dof_map + mi[MI_DOF], trow, A.v_es, f);
                  if (parent >= 0)
                  {
                     const SV<T> fp = force_up(type, jx, Xb, f);
                     if (ev & EV_LAST_CHILD)
                        rcarry = fp;
                     else if (ev & EV_CARRY_ADD)
                        rcarry = rcarry + fp;
                     else
                        st_add6<T>(S, mi[MI_PFR_R], fp);
                  }
               }
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
                  const T u = in.x - pz;                   // :1200-1215
                  MH_HD(hf + 0) = ua.x, MH_HD(hf + 1) = ua.y, MH_HD(hf + 2) = ua.z, MH_HD(hf + 3) = ul.x, MH_HD(hf + 4) = ul.y, MH_HD(hf + 5) = ul.z;
                  MH_HD(hf + 6) = dinv;
                  MH_HD(hf + 7) = u; // (pA carries IA w: this is u - U.w, and the outward sweep needs U.(X a~_parent) only)
                  if (type == JT_REVOLUTE)
                     MH_HD(hf + 8) = jx.c, MH_HD(hf + 9) = jx.s;
                  if (parent >= 0)
                  {
                     const T ud = u * dinv;
                     if (type == JT_REVOLUTE)
                     { // Ia S = 0: exact structural zeros -- and the hand-up in the SAME block, so that every product with them folds
                        rank1_down_revolute(Ia, ua, ul, dinv);  // :1220-1226
                        pa = pA + SV<T>{ud * ua, ud * ul};      // :1229-1234 with c = 0
                        revolute_up(jx, Xb, Ia, pa);            // :1156-1166; pa is now expressed in the parent's frame
                        handed_up = true;
                     }
                     else
                     {
                        rank1_down(Ia, ua, ul, dinv);
                        pa = pA + SV<T>{ud * ua, ud * ul};
                     }
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
                  MH_HD(hf + 24) = u3.x, MH_HD(hf + 25) = u3.y, MH_HD(hf + 26) = u3.z;
                  if (parent >= 0)
                  {
                     const SV<T> W0 = Di.xx * U0 + Di.xy * U1 + Di.xz * U2, W1 = Di.xy * U0 + Di.yy * U1 + Di.yz * U2, W2 = Di.xz * U0 + Di.yz * U1 + Di.zz * U2;
                     rank1_pair_down(Ia, W0, U0), rank1_pair_down(Ia, W1, U1), rank1_pair_down(Ia, W2, U2);
                     pa = pA + u3.x * W0 + u3.y * W1 + u3.z * W2;
                  }
               }
               else if (type == JT_SIXDOF)
               { // S = 1_6: x = IA^-1 (tau - pA) is the body acceleration; for the parent Ia = 0, pa = tau
                  const SV<T> tau = joint_vec<T>(type, dof_map, mi[MI_DOF], taurow, A.v_es, true);
                  const SV<T> x = spd6_solve(IA, tau - pA); // = a~ of this body: qdd = x - X a~_parent, and the children start from x
                  MH_HD(hf + 0) = x.a.x, MH_HD(hf + 1) = x.a.y, MH_HD(hf + 2) = x.a.z, MH_HD(hf + 3) = x.l.x, MH_HD(hf + 4) = x.l.y, MH_HD(hf + 5) = x.l.z;
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
                     if (type != JT_SIXDOF) // (a floating joint transmits no inertia: Ia = 0 stays 0)
                        abi_up(type, jx, Xb, Ia); // :1156-1166
                     pp = force_up(type, jx, Xb, pa);
                  }
                  if (ev & EV_LAST_CHILD)
                     Icarry = Ia, pcarry = pp;
                  else if (ev & EV_CARRY_ADD)
                     add(Icarry, Ia), pcarry = pcarry + pp;
                  else
                  {
                     const int acc = mi[MI_PACC_A];
                     if (ev & EV_ACC_FIRST)
                        st_store_abi<T>(S, acc, Ia), st_store6<T>(S, acc + 21, pp);
                     else
                        st_add_abi6<T>(S, acc, Ia, pp);
                  }
               }
            }
            }; // body
            if constexpr (WIN)
               body(KindRt{type_rt});
            else
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
         // ---- outward part: pass three (:1259-1310), joint accelerations root to leaves
         SV<T> a_reg{Z, Z};
         auto prefetch_q = [&](int j1) { // revolute joints take (cos, sin) from the hand-over; prismatic ones re-read q (fetched ahead)
            if (j1 >= m.n)
               return;
            ciptr m1 = meta + j1 * MI_STRIDE;
            if (m1[MI_TYPE] == JT_PRISMATIC)
               nxt.q = qrow[m1[MI_ROW_Q] * A.q_es];
         };
         prefetch_q(0);
         for (int j = 0; j < m.n; j++)
         {
            ciptr mi = meta + j * MI_STRIDE;
            const int parent = mi[MI_PARENT], type_rt = mi[MI_TYPE], nch = mi[MI_NCH], fr = mi[MI_DFS_A], hf = mi[MI_HAND];
            auto body = [&](auto kind) {
            const int type = kind;
            const CRef<T, false> c{CB + j * MC_STRIDE};
            const In<T> in = nxt;
            prefetch_q(j + 1);
            SV<T> ap;
            if (parent < 0)
               ap = root_acceleration(A); // :259-264
            else if (parent == j - 1)
               ap = a_reg;
            else
               ap = st_load6<T>(S, mi[MI_PFR_A]);
            JX<T> jx;
            if (type == JT_REVOLUTE)
               jx.c = MH_HD(hf + 8), jx.s = MH_HD(hf + 9), jx.d = T(0);
            else if (type == JT_PRISMATIC)
               jx.c = T(1), jx.s = T(0), jx.d = in.q;
            else
               jx = joint_from_q<T>(type, cfg_map, mi[MI_CFG], qrow, A.q_es, (T *)nullptr, 0, 0, false);
            const SV<T> apx = motion_down(type, jx, load_xb<T>(c), ap);
            ciptr di = dof_map + mi[MI_DOF];
            SV<T> a = apx;
            if (type == JT_REVOLUTE || type == JT_PRISMATIC)
            {
               const V3<T> ua{MH_HD(hf + 0), MH_HD(hf + 1), MH_HD(hf + 2)}, ul{MH_HD(hf + 3), MH_HD(hf + 4), MH_HD(hf + 5)};
               const T qdd = MH_HD(hf + 6) * (MH_HD(hf + 7) - (dot(ua, apx.a) + dot(ul, apx.l))); // :1280-1282 with u' = u - U.c
               if (active)
                  orow[di[0] * A.v_es] = qdd;
               if (nch >= 1)
               {
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
               if (active)
                  orow[di[0] * A.v_es] = qdd.x, orow[di[1] * A.v_es] = qdd.y, orow[di[2] * A.v_es] = qdd.z;
               if (nch >= 1)
                  a = apx + from_comp3(type, qdd);
            }
            else if (type == JT_SIXDOF)
            {
               const SV<T> xc{V3<T>{MH_HD(hf + 0), MH_HD(hf + 1), MH_HD(hf + 2)}, V3<T>{MH_HD(hf + 3), MH_HD(hf + 4), MH_HD(hf + 5)}};
               const SV<T> qdd = xc - apx;
               if (active)
               {
                  orow[di[0] * A.v_es] = qdd.a.x, orow[di[1] * A.v_es] = qdd.a.y, orow[di[2] * A.v_es] = qdd.a.z;
                  orow[di[3] * A.v_es] = qdd.l.x, orow[di[4] * A.v_es] = qdd.l.y, orow[di[5] * A.v_es] = qdd.l.z;
               }
               if (nch >= 1)
                  a = xc;
            }
            if (nch >= 2)
               st_store6<T>(S, fr, a); // later children re-read it
            a_reg = a;
            }; // body
            if constexpr (WIN)
               body(KindRt{type_rt});
            else
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
#undef MH_HD
   };
   if constexpr (HND_LDS)
      walk((T *)lds_raw + tid + (long)m.aba_stack * 64);
   else
      walk(glb);
}
#undef MH_ST_GROUP
#undef MH_SP
} // namespace mh
