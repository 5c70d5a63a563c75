// mh_api.hip -- host side of the C-ABI declared in include/mecano_hip.h.
//
// Model build (once per MultiBodySystemReadOnly): validation, parents-first ordering, canonical joint frames,
// workspace slot assignment, upload.  Compute calls: argument checks, workspace, kernel launch on the caller's
// stream.  No CPU implementation of the algorithms exists in this library: without a HIP device the compute
// entry points return MH_ERR_NO_DEVICE.
#include "../../include/mecano_hip.h"
#include "mh_dfs_kernels.h"
#include "mh_split_kernels.h"

#include <dlfcn.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cerrno>
extern char **environ;
#include <unistd.h>

#include <algorithm>
#include <functional>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

// Per-joint constants of the generic kernels: staged in LDS (1) or read through scalar loads (0).
#ifndef MH_GENERIC_LDS_CONSTS
#define MH_GENERIC_LDS_CONSTS 0
#endif

namespace
{
thread_local char g_err[512] = "";

mh_status fail(mh_status code, const char *fmt, ...)
{
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(g_err, sizeof g_err, fmt, ap);
   va_end(ap);
   return code;
}
#define HIP_TRY(expr)                                                                                      \
   do                                                                                                      \
   {                                                                                                       \
      hipError_t e_ = (expr);                                                                              \
      if (e_ != hipSuccess)                                                                                \
         return fail(e_ == hipErrorOutOfMemory ? MH_ERR_OUT_OF_MEMORY : MH_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
   } while (0)

// ------------------------------------------------------------------ tiny host 3x3 helpers (double)
struct M3d
{
   double m[9];
};
M3d m3_identity() { return M3d{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
M3d m3_mul(const M3d &a, const M3d &b)
{
   M3d o;
   for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
         o.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
   return o;
}
M3d m3_T(const M3d &a) { return M3d{{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}}; }
void m3_mulv(const M3d &a, const double v[3], double o[3])
{
   double x = a.m[0] * v[0] + a.m[1] * v[1] + a.m[2] * v[2];
   double y = a.m[3] * v[0] + a.m[4] * v[1] + a.m[5] * v[2];
   double z = a.m[6] * v[0] + a.m[7] * v[1] + a.m[8] * v[2];
   o[0] = x, o[1] = y, o[2] = z;
}
// rotation Q with Q * ez = k (k unit): columns (x', y', k) of a right-handed orthonormal basis
M3d frame_with_z(const double k[3])
{
   int least = std::fabs(k[0]) <= std::fabs(k[1]) ? (std::fabs(k[0]) <= std::fabs(k[2]) ? 0 : 2) : (std::fabs(k[1]) <= std::fabs(k[2]) ? 1 : 2);
   double h[3] = {0, 0, 0};
   h[least] = 1.0;
   double d = h[0] * k[0] + h[1] * k[1] + h[2] * k[2];
   double x[3] = {h[0] - d * k[0], h[1] - d * k[1], h[2] - d * k[2]};
   double n = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
   x[0] /= n, x[1] /= n, x[2] /= n;
   double y[3] = {k[1] * x[2] - k[2] * x[1], k[2] * x[0] - k[0] * x[2], k[0] * x[1] - k[1] * x[0]};
   return M3d{{x[0], y[0], k[0], x[1], y[1], k[1], x[2], y[2], k[2]}};
}

// NaN tests on raw 64-bit words that never pass through a `double` value: this translation unit is built with -ffinite-math-only, under
// which x != x folds to false AND double parameters carry nofpclass(nan), so that even a bit test on a double argument may be folded away
inline bool nan_word(unsigned long long u) { return (u & 0x7ff0000000000000ull) == 0x7ff0000000000000ull && (u & 0x000fffffffffffffull) != 0; }
inline bool nan_bits(const double &x)
{
   unsigned long long u;
   std::memcpy(&u, (const void *)&x, sizeof u);
   asm volatile("" : "+r"(u)); // the optimiser must not trace the word back to a floating-point value
   return nan_word(u);
}
// ---- build provenance (mecano_amd/build.py computes the same numbers): FNV-1a 64 over, per file, name 0 contents 0, then the
//      code-generation flags.  The library carries the hash of its own sources (MH_BUILD_HASH) and the hash it expects of a code
//      object's sources (MH_SPEC_SOURCES_HASH); a code object carries the latter and is refused when it differs (try_load_spec).
#define MH_STR2_(...) #__VA_ARGS__
#define MH_STR_(...) MH_STR2_(__VA_ARGS__)
#ifndef MH_BUILD_HASH
#define MH_BUILD_HASH unhashed
#endif
#ifndef MH_SPEC_SOURCES_HASH
#define MH_SPEC_SOURCES_HASH unhashed
#endif
__attribute__((used)) const char kBuildId[] = "MH_BUILD_ID=" MH_STR_(MH_BUILD_HASH) ";spec=" MH_STR_(MH_SPEC_SOURCES_HASH) ";";
const char *const kSpecHashFiles[] = {"mh_spec.hip", "mh_spec_kernels.h", "mh_zv_kernels.h", "mh_kernels.h", "mh_device.h"};
const char kSpecCodegenFlags[] = "--offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -ffinite-math-only -fno-slp-vectorize -mllvm -disable-machine-licm";
uint64_t fnv1a(uint64_t h, const void *data, size_t n)
{
   const unsigned char *p = (const unsigned char *)data;
   for (size_t k = 0; k < n; k++)
      h = (h ^ p[k]) * 0x100000001b3ull;
   return h;
}
bool hash_spec_sources(const std::string &csrc_dir, char out[18])
{
   uint64_t h = 0xcbf29ce484222325ull;
   for (const char *name : kSpecHashFiles)
   {
      FILE *f = fopen((csrc_dir + "/" + name).c_str(), "rb");
      if (!f)
         return false;
      h = fnv1a(h, name, strlen(name) + 1);
      char buf[1 << 16];
      size_t got;
      while ((got = fread(buf, 1, sizeof buf, f)) > 0)
         h = fnv1a(h, buf, got);
      fclose(f);
      h = fnv1a(h, "", 1);
   }
   h = fnv1a(h, kSpecCodegenFlags, sizeof kSpecCodegenFlags - 1);
   snprintf(out, 18, "h%016llx", (unsigned long long)h);
   return true;
}
int joint_ndof(int t) { return mh::dof_count(t); }
int joint_ncfg(int t) { return mh::cfg_count(t); }

struct Workspace
{
   void *ptr = nullptr;
   size_t bytes = 0;
};
// A member of mh_model that a compute call may be WRITING while another thread copies the model into a new context
// (mh_context_create): the copy never reads it -- it starts empty, which is what a context wants anyway.
template <class T>
struct FreshOnCopy : T
{
   FreshOnCopy() = default;
   FreshOnCopy(const FreshOnCopy &) : T() {}
   FreshOnCopy &operator=(const FreshOnCopy &) { return *this; }
   FreshOnCopy &operator=(const T &v)
   {
      T::operator=(v);
      return *this;
   }
};
} // namespace

// entry points of a topology-specialised code object (mh_spec.hip), resolved with dlsym
struct SpecLib
{
   void *handle = nullptr;
   int (*launch)(int algo, int flags, const void *args, int grid, void *stream) = nullptr;
   long (*lds_bytes)(int algo, int flags, int nq, int nv) = nullptr;
   int (*aba_slots)(void) = nullptr;
   int (*supports)(int algo, int flags) = nullptr;
   int (*launch_fused)(int flags, const void *args, int waves, void *stream) = nullptr;
   long (*fused_lds_bytes)(int nq, int nv) = nullptr;
   int (*launch_crba)(int flags, const void *args, int grid, void *stream) = nullptr;
   int (*crba_packed)(int flags) = nullptr;
   int (*split_usable)(void) = nullptr;
   long (*split_lds_bytes)(int algo, int flags, int nq, int nv) = nullptr;
   int (*launch_split)(int algo, int flags, const void *args, int groups, void *stream) = nullptr;
   int (*crba_split_usable)(void) = nullptr;
   int (*launch_crba_split)(const void *args, int groups, int lanes_per_group, void *stream) = nullptr;
   int (*launch_rnea_crba)(const void *args, int rnea_groups, int crba_groups, int lanes_per_group, void *stream) = nullptr;
   long (*rnea_crba_lds_bytes)(int lanes_per_group, int nq, int nv) = nullptr;
   int (*launch_coriolis)(int flags, const void *args, int grid, void *stream) = nullptr;
   int (*launch_centroidal)(int flags, const void *args, int grid, void *stream) = nullptr;
   int (*launch_coriolis_parts)(int flags, const void *args, int grid, int parts, void *stream) = nullptr;
   int (*launch_centroidal_parts)(int flags, const void *args, int grid, int parts, void *stream) = nullptr;
   unsigned long long (*abi)(void) = nullptr;
   const char *(*sources_hash)(void) = nullptr;
   // bias-split forward dynamics (mh_zv_kernels.h)
   int (*zv_usable)(void) = nullptr;
   long (*zv_lds_bytes)(int nq, int nv) = nullptr;
   int (*launch_zv)(int flags, const void *args, void *taup, int *sync_flags, int *error, int epoch, int jobs, int same_l2, unsigned wait_ticks, void *stream) = nullptr;
   int (*zv_self_signal)(void) = nullptr; // 1: identity-map launches expect taup to hold the sentinel wherever no column has been published
   // forward dynamics of device-filling batches as two launches (mh_zv_kernels.h, spec_zvb_*)
   int (*zvb_usable)(void) = nullptr;
   int (*zvb_cs_rows)(void) = nullptr;
   long (*zvb_lds_bytes)(int which, int nq, int nv) = nullptr;
   int (*launch_zvb)(int flags, const void *args, void *taup, void *cs, long cs_stride, int groups, int which, void *stream) = nullptr;
   // ... as one launch, both jobs fused in a workgroup (spec_zvf_kernel)
   int (*zvf_usable)(void) = nullptr;
   int (*zvf_pair_usable)(void) = nullptr; // 1: launch_zvf serves the pair call too (args->in3b = qdd, args->outb = tau)
   int (*launch_zvf)(int flags, const void *args, int groups, void *stream) = nullptr;
   int (*launch_rnea_ahead)(int flags, const void *args, int groups, void *stream) = nullptr;
};
enum : int
{
   SPEC_IO_LDS = 1,
   SPEC_IDENT = 2,
   SPEC_ST_LDS = 4,
   SPEC_BODIES = 16,
   SPEC_OCC3 = 32
};

struct mh_model
{
   int n = 0, nq = 0, nv = 0, n_slots = 0;
   SpecLib spec;
   std::string topo_key;
   int device = 0;
   int cu_count = 256;
   std::vector<int> meta, dof_map, cfg_map;
   std::vector<int> engine_of; // caller joint index -> engine index
   std::vector<double> consts;
   int *d_meta = nullptr, *d_dof = nullptr, *d_cfg = nullptr, *d_prog = nullptr;
   std::vector<int> prog; // event program of the depth-first kernels
   std::vector<int> prog_seq; // the same walk with the siblings in engine order (the kernels that read AoS rows through LDS windows)
   int *d_prog_seq = nullptr;
   int rnea_stack = 0, aba_stack = 0, aba_hand = 0; // per-lane slots: depth stacks, ABA hand-over
   int pair_stack = 0;    // ... of the fused RNEA + ABA walk (aba_dfs_kernel<.., PAIR>)
   int dfs_aba_occ3 = 1;  // MH_DFS_ABA_OCC3=0: never the three-waves-per-SIMD build of the fp32 depth-first forward dynamics
   int use_dfs_pair = 1;  // MH_DFS_PAIR=0: mh_rnea_aba_f32 on big batches issues the two depth-first kernels one after the other, as before round 5
   int use_dfs = 1;       // MH_DFS=0: the sweep kernels of mh_kernels.h serve plain RNEA / ABA calls too (A/B measurements)
   FreshOnCopy<std::map<const void *, size_t>> lds_attr; // dynamic-LDS limit already raised per kernel (the model lives on one device, one host thread at a time)
   double nonleaf_fraction = 1.0; // share of bodies with children: those are the ones that touch the depth stack
   // depth-first kernels: frame homes for a given LDS budget (slots per wave), one copy of the body records per (algorithm, budget) on
   // the device; built on first use (dfs_plan), dropped when the records change (joint source modes)
   struct DfsPlan
   {
      int algo, budget, lds_slots, glb_slots, glb_frames;
      int *d_meta;
   };
   FreshOnCopy<std::deque<DfsPlan>> dfs_plans; // references stay valid across push_back; kept by the model itself (a context uses its model's: dfs_plan)
   struct PlainMutex : std::mutex
   { // a context starts as a copy of its model (context_clone): the copy gets a mutex of its own
      PlainMutex() = default;
      PlainMutex(const PlainMutex &) : std::mutex() {}
      PlainMutex &operator=(const PlainMutex &) { return *this; }
   } dfs_mutex;
   // mh_context_create: a context is a copy of the model's host-side description that SHARES its device records (parent owns them) and
   // owns everything compute calls write -- workspace, scratch matrices, staging buffers, streams, hand-off flags, the error word
   mh_model *parent = nullptr;
   int n_contexts = 0; // live contexts of this model (guarded by g_context_mutex)
   bool destroy_pending = false; // mh_model_destroy was called while contexts were alive: the last mh_context_destroy releases the model
   int use_win = 1;       // MH_DFS_WIN=0: AoS rows are read per lane instead of through LDS windows (A/B measurements)
   int dfs_place = -1;    // MH_DFS_PLACE = 0 | 1 | 2: force all-LDS / stack in LDS + hand-over global / all global
   bool dfs_place_greedy = false; // MH_DFS_GREEDY=1: the frames' homes from the leaves upwards as in rounds 2-4 (A/B measurements; dfs_plan)
   int dfs_budget = -1;   // MH_DFS_BUDGET: cap of the stack's LDS budget in slots per wave (measurements)
   int dfs_aba64 = 0;     // fp64 forward dynamics on the depth-first kernel too: bushy trees (below), or MH_DFS_ABA64=0|1
   int n_nonadjacent = 0; // bodies whose parent is not the body before them in engine order (branch points of the tree)
   int dfs_transpose = -1; // MH_DFS_TRANSPOSE = 0 | 1: depth-first kernels on big AoS batches never / always through transposed scratch copies
   double *d_consts64 = nullptr;
   float *d_consts32 = nullptr;
   Workspace ws;
   // staging buffers of the *_host entry points
   Workspace stage;
   // pipelined host path: copy-in / compute / copy-out streams and per-slot events of a ring of three device chunk slots
   hipStream_t hs_in = nullptr, hs_run = nullptr, hs_out = nullptr;
   hipEvent_t ev_in[3] = {}, ev_run[3] = {}, ev_out[3] = {};
   int host_chunk = 0; // MH_HOST_CHUNK: configurations per chunk of the host-pointer pipeline (0 = choose)
   // mh_rnea_aba_f64 without a fused kernel, small batches: the two launches run side by side, the ABA on this stream with its own workspace
   hipStream_t pair_stream = nullptr;
   hipEvent_t pair_fork = nullptr, pair_join = nullptr;
   Workspace ws_pair;
   int use_pair = 1; // MH_DISABLE_PAIR=1: always one after the other
   // bias-split forward dynamics (mh_zv_kernels.h): tau - h(q, qd) rows, one flag per 64 configurations (a launch stores its epoch there),
   // an error word in mapped host memory that a timed-out wait sets (read at the next call of the model)
   Workspace zv_tau, zv_flags;
   Workspace zv_cols;     // two-stage hand-off with self-signalling limb columns (identity index maps): [groups][nv][64], holds the sentinel
                          // between launches (mh_zv_kernels.h: ZV_SENTINEL) -- written by nothing but those launches
   Workspace zvb_cs;      // two-launch forward dynamics: (cos, sin) of the revolute joints, [2 n_rev][B rounded up to 64]
   int use_rnea_ahead = 1; // MH_RNEA_AHEAD (see rnea_ahead_ok)
   int use_zv_step = 1;    // MH_ZV_STEP=0: simulation steps never ride in the bias-split / fused forward dynamics (the one-job tree-split kernel integrates instead)
   int use_zvb = 1;       // MH_ZVB=0: never; 1: batches of two or more groups of 64 configurations per CU (default); 2: whenever the call qualifies; MH_ZVB_WHICH = 1 | 2: one of the two launches only (timing)
   int zvb_which = 3;
   int use_zvf = 1;       // MH_ZVF=0: never the fused one-launch form; 1: where the two-launch form would be taken (default); 2: whenever the call qualifies
   int use_zvf_pair = 1;  // MH_ZVF_PAIR=0: the pair call of device-filling batches as two launches (A/B measurements)
   int zv_epoch = 0;
   int *zv_error_host = nullptr, *zv_error_dev = nullptr;
   int zv_same_l2 = 0;    // MH_ZV_SAME_L2=1 (experiment, off by default; one-stage hand-off only: the two-stage form of identity index maps is write-through): bias rows and flag of a group whose two jobs prove to sit behind the same L2
                          // stay in that L2 (workgroup-scope stores) -- cache behaviour the memory model does not promise, for no measured gain
   unsigned zv_wait_ticks = 200000000u; // MH_ZV_WAIT_MS: how long an inertia job waits for its bias rows (100 MHz ticks; default 2 s)
   int use_zv = 1;        // MH_ZV=0: never; 1: while every job's workgroup gets a CU of its own (default); 2: whenever the call qualifies
   // run-time tree split (mh_split_kernels.h): plan made at creation, device copies, workspace blocks
   struct SplitRt
   {
      bool usable = false;
      int n_trunk = 0, n_limbs = 0, slots = 0, est = 0, total = 0;
      int n_seg[mh::SPLIT_WAVES] = {};
      int *d_meta[3] = {nullptr, nullptr, nullptr}, *d_trunk = nullptr, *d_seg = nullptr, *d_xl_ofs = nullptr, *d_xl[3] = {nullptr, nullptr, nullptr}; // [0] fp32, [1] fp64, [2] no LDS share
      int lds_slots[3] = {0, 0, 0}; // slots below this number live in LDS (the slot codes of the records say so), per record set
      std::vector<int> meta;     // (body, field, value) patches of the adapted records
      std::vector<int> xl;       // exchange slots of the limbs attached to the trunk bodies (plain slot numbers)
   } split_rt;
   int use_split_rt = -1; // MH_SPLIT_RT = 0 | 1: never / whenever usable (default: small batches)
   int split_rt_lds = -1; // MH_SPLIT_RT_LDS = 0 | 1: the split kernels' workspace never / always with its LDS share (measurements)
   // AoS -> SoA scratch copies of the state matrices for the run-time-topology kernels (big batches of wide matrices); tr_pair: the
   // copies of the forward dynamics call that runs beside the inverse dynamics call on pair_stream (they would share addresses otherwise)
   Workspace tr, tr_pair;
   // scratch of the composite entry points: efforts of the Newton-Euler sweep behind mh_aba_joint_wrenches_f64, pair lists of
   // mh_relative_acceleration_f64
   Workspace aux, pairs;
   FreshOnCopy<std::vector<int>> pairs_host;
   int use_transpose = -1; // MH_GENERIC_TRANSPOSE = 0 | 1 overrides the size heuristic
   std::string variant = "generic";
   uint32_t warnings = 0;    // MH_WARN_* bits set by mh_model_create (mh_model_warnings)
   std::string warning_text; // ... and what they mean for this model
   int use_split = -1;      // MH_SPEC_SPLIT = 0 | 1: never / whenever possible use the tree-split kernels (default: small batches)
   int use_fused = 1;       // MH_DISABLE_FUSED=1: mh_rnea_aba_f64 always issues two launches
   int fused_factor = 4;    // one launch for RNEA + ABA while 2 * ceil(B / 64) workgroups <= cu_count * factor (MH_FUSED_FACTOR)
   int use_spec = 1;        // MH_DISABLE_SPEC=1 in the environment forces the generic kernels (A/B measurements)
   bool spec_minimal = false; // the loaded code object is a minimal (fast) build
   int lds_wave_factor = 1; // ABA hand-over in LDS while waves <= cu_count * factor (MH_ABA_LDS_FACTOR)
   int ident_maps = 0;      // the engine-order index maps are the identity
   int dense_maps = 0;      // nq / nv equal the joints' totals (no unused matrix rows): rows can be staged as dense blocks
   int force_io = -1, force_st = -1; // MH_SPEC_IO / MH_SPEC_ST = 0 | 1 override the heuristics (measurements)
   int n_locked = 0;        // joints in MH_ACCELERATION_SOURCE mode (mh_model_set_joint_source_modes)
   int lds_consts = 0;      // run-time-topology kernels: per-joint constants staged in LDS (large models: they overflow the scalar cache) or read by scalar loads
   int waves_per_cu = 8;    // resident waves per CU the run-time-topology kernels are launched with (MH_WAVES_PER_CU)
   bool waves_per_cu_set = false; // MH_WAVES_PER_CU given: the depth-first kernels take it instead of what their registers allow (dfs_reg_cap)
};

struct mh_context
{
   mh_model *m; // the context's copy of the handle (parent = the model it was created from)
};
namespace
{
std::mutex g_context_mutex;
mh_status ensure_bytes(Workspace &w, size_t bytes)
{
   if (w.bytes >= bytes)
      return MH_OK;
   if (w.ptr)
      HIP_TRY(hipFree(w.ptr));
   w.ptr = nullptr, w.bytes = 0;
   HIP_TRY(hipMalloc(&w.ptr, bytes));
   w.bytes = bytes;
   return MH_OK;
}

struct Launch
{
   int block, grid;
   long lanes;
};
Launch plan_launch(const mh_model *m, int64_t B)
{
   Launch L;
   L.block = 64; // one wave per workgroup: a small batch spreads over as many CUs as it has waves
   long waves = (B + 63) / 64;
   long cap = (long)m->cu_count * m->waves_per_cu; // resident waves: the workspace is sized by the grid, not by B
   L.grid = (int)std::max<long>(1, std::min(waves, cap));
   L.lanes = (long)L.grid * L.block;
   return L;
}
mh_status ensure_workspace(mh_model *m, int64_t B, size_t elem)
{
   Launch L = plan_launch(m, B);
   return ensure_bytes(m->ws, (size_t)m->n_slots * (size_t)L.lanes * elem);
}

// Kernels whose per-body columns are independent (mass matrix, Coriolis matrix, centroidal momentum matrix, joint torque regressor), small
// batches: up to eight waves per group of 64 configurations, each taking every parts-th body (mh_kernels.h) -- as many as keep one
// wave per SIMD (measured: profiles/r02_regressor_rates.txt, profiles/r02_column_parts.txt)
static int regressor_parts(const mh_model *model, const Launch &L)
{
   int parts = (int)std::max<long>(1, std::min<long>(std::min<long>(8, model->n), (long)model->cu_count * 4 / L.grid));
   if (const char *e = getenv("MH_REGRESSOR_PARTS"))
      parts = std::max(1, std::min(64, atoi(e)));
   return parts;
}
static mh_status ensure_parts_workspace(mh_model *m, const Launch &L, int parts, size_t elem)
{
   return ensure_bytes(m->ws, (size_t)m->n_slots * (size_t)L.lanes * (size_t)parts * elem);
}

template <typename T>
mh::DevModel dev_model(const mh_model *m)
{
   mh::DevModel d;
   d.n = m->n, d.nq = m->nq, d.nv = m->nv, d.n_slots = m->n_slots;
   d.meta = m->d_meta, d.dof_map = m->d_dof, d.cfg_map = m->d_cfg;
   d.consts = sizeof(T) == 8 ? (const void *)m->d_consts64 : (const void *)m->d_consts32;
   d.prog = m->d_prog, d.n_events = (int)m->prog.size();
   d.rnea_stack = m->rnea_stack, d.aba_stack = m->aba_stack, d.aba_hand = m->aba_hand;
   return d;
}

// Root acceleration of a call: (0, -g) for the gravity vector, or opts->root_acceleration (angular, linear) when the caller set one
// (InverseDynamicsCalculator.java:343-348 / 413-427, ForwardDynamicsCalculator.java:259-264 / 330-343).  The kernels take minus the linear
// part in (gx, gy, gz) and the angular part in (rax, ray, raz).
template <class ARGS>
void set_root_acceleration(ARGS &A, const mh_options &o, const double *gravity)
{
   using T = decltype(A.gx);
   if (o.use_root_acceleration)
   {
      A.rax = (T)o.root_acceleration[0], A.ray = (T)o.root_acceleration[1], A.raz = (T)o.root_acceleration[2];
      A.gx = (T)-o.root_acceleration[3], A.gy = (T)-o.root_acceleration[4], A.gz = (T)-o.root_acceleration[5];
   }
   else
   {
      A.rax = T(0), A.ray = T(0), A.raz = T(0);
      A.gx = gravity ? (T)gravity[0] : T(0), A.gy = gravity ? (T)gravity[1] : T(0), A.gz = gravity ? (T)gravity[2] : T(0);
   }
}

// Also resolves opts->context: a compute call made with a context runs on the context's copy of the handle (its own workspace, scratch,
// streams, flags); `model` is switched to it here, before the entry point touches anything mutable.
mh_status check_common(mh_model_t &model, int64_t B, const mh_options *opts)
{
   if (!model)
      return fail(MH_ERR_INVALID_ARGUMENT, "model is NULL");
   if (opts && opts->context)
   {
      mh_model *c = ((mh_context *)opts->context)->m;
      if (c->parent != (model->parent ? model->parent : model))
         return fail(MH_ERR_INVALID_ARGUMENT, "opts->context belongs to another model");
      model = c;
   }
   if (B < 0)
      return fail(MH_ERR_BAD_DIMENSION, "negative batch size %lld", (long long)B);
   if (opts && opts->layout != MH_LAYOUT_AOS && opts->layout != MH_LAYOUT_SOA)
      return fail(MH_ERR_INVALID_ARGUMENT, "unknown layout %d", opts->layout);
   int cur = 0;
   if (hipGetDevice(&cur) != hipSuccess)
      return fail(MH_ERR_NO_DEVICE, "no usable HIP device");
   if (cur != model->device)
      return fail(MH_ERR_INVALID_ARGUMENT, "model lives on device %d but the calling thread's device is %d", model->device, cur);
   return MH_OK;
}

// tree-split kernels: 4 waves per 64 configurations; worth it while the batch cannot give every SIMD a wave of its own otherwise
// flags for the tree-split kernels: identity maps; rows staged in LDS when the layout is AoS, the maps are dense and it fits
int split_flags(const mh_model *m, int algo, bool soa)
{
   int flags = m->ident_maps ? SPEC_IDENT : 0;
   if (!soa && m->dense_maps && m->force_io != 0 && m->spec.split_lds_bytes(algo, SPEC_IO_LDS, m->nq, m->nv) <= 160 * 1024)
      flags |= SPEC_IO_LDS;
   return flags;
}
bool split_ok(const mh_model *m, int algo, int64_t B, bool soa)
{
   if (!m->spec.launch_split || !m->spec.split_usable || !m->spec.split_usable() || !m->use_spec || m->use_split == 0)
      return false;
   if (m->spec.split_lds_bytes(algo, split_flags(m, algo, soa), m->nq, m->nv) > 160 * 1024)
      return false;
   if (m->use_split == 1 || algo == 1 || algo == 0)
      return true; // ABA: the split form needs fewer registers; RNEA: two waves per SIMD and a trunk pass that is a fold of parked
                   // wrenches -- both measured faster than the whole-tree kernels at every batch size and in both layouts
   const long groups = (B + 63) / 64;
   const long waves = groups * 4 * (algo == 2 ? 2 : 1);
   return waves <= (long)m->cu_count * 4 * m->fused_factor; // fused: while the batch cannot give every SIMD a wave of its own
}

// Bias-split forward dynamics (mh_zv_kernels.h): AoS matrices, dense index maps, every joint an effort source, no per-body outputs.
// The launch puts `jobs` workgroups of four waves on every 64 configurations, each with a CU's LDS nearly to itself: it pays while they
// all fit the device at once (measured, humanoid: pair 16.6 vs 18.3 us at B = 4096, 25.4 vs 19.3 at 8192; forward dynamics alone 17.0 vs
// 18.9 us at 8192, 31.2 vs 20.0 at 16384 -- profiles/r03_zv_vs_tree_split.txt); beyond that the tree-split kernels serve the call.
bool zv_ok(const mh_model *m, int64_t B, bool soa, int jobs)
{
   if (!m->spec.launch_zv || !m->spec.zv_usable || !m->spec.zv_usable() || !m->use_spec || !m->use_zv || m->use_split == 0)
      return false;
   if (soa || !m->dense_maps || m->force_io == 0 || m->n_locked > 0)
      return false;
   const long lds = m->spec.zv_lds_bytes(m->nq, m->nv);
   return lds > 0 && lds <= 160 * 1024 && (m->use_zv == 2 || (B + 63) / 64 * jobs <= (long)m->cu_count);
}
// A wait of a bias-split launch that ran into its wall-clock limit (the producer workgroup never published): the kernel wrote NaN rows
// for that group and set the model's (context's) error word in mapped host memory.  It is a failure of an ASYNCHRONOUS call, so it is
// reported where the library next synchronises or is asked to: mh_model_check, mh_stream_synchronize, the *_host entry points after
// their own synchronisation, the create-time self-check -- and at the latest by the next bias-split call of the same model / context.
std::mutex g_error_words_mutex;
std::vector<int *> g_error_words; // every live model's / context's mapped error word (mh_stream_synchronize has no handle to ask)
mh_status zv_check_error(mh_model *m)
{
   if (m->zv_error_host && *(volatile int *)m->zv_error_host != 0)
   {
      *(volatile int *)m->zv_error_host = 0;
      return fail(MH_ERR_HIP, "a bias-split forward dynamics launch gave up waiting for its bias rows (the accelerations of those configurations were written as NaN)");
   }
   return MH_OK;
}
mh_status check_all_error_words()
{
   std::lock_guard<std::mutex> lock(g_error_words_mutex);
   bool any = false;
   for (int *w : g_error_words)
      if (*(volatile int *)w != 0)
      {
         *(volatile int *)w = 0;
         any = true;
      }
   if (any)
      return fail(MH_ERR_HIP, "a bias-split forward dynamics launch gave up waiting for its bias rows (the accelerations of those configurations were written as NaN)");
   return MH_OK;
}
// scratch of the bias-split launches for batches up to B: the bias rows, the flags (zeroed on `stream`), the mapped error word
bool zv_self_signalling(const mh_model *m) { return m->ident_maps && m->spec.zv_self_signal && m->spec.zv_self_signal() != 0; }
mh_status zv_prepare(mh_model *m, int64_t B, hipStream_t stream)
{
   const size_t groups = (size_t)((B + 63) / 64);
   mh_status st = ensure_bytes(m->zv_tau, groups * 64 * m->nv * sizeof(double)); // (whole groups: the two-stage hand-off keeps a matrix [nv][64] per group)
   if (st != MH_OK)
      return st;
   if (zv_self_signalling(m) && m->zv_cols.bytes < groups * 64 * m->nv * sizeof(double))
   { // the matrix whose limb columns signal themselves: sentinels wherever nothing has been published (filled ON THE LAUNCH STREAM, like the flags)
      st = ensure_bytes(m->zv_cols, std::max<size_t>(groups, 128) * 64 * m->nv * sizeof(double));
      if (st != MH_OK)
         return st;
      HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)m->zv_cols.ptr, 0x7ff4a5a5, m->zv_cols.bytes / 4, stream));
   }
   if (m->zv_flags.bytes < groups * mh::ZV_SYNC_STRIDE * sizeof(int))
   {
      st = ensure_bytes(m->zv_flags, std::max<size_t>(groups, 1024) * mh::ZV_SYNC_STRIDE * sizeof(int));
      if (st != MH_OK)
         return st;
      // zeroed ON THE LAUNCH STREAM: a plain hipMemset is not ordered against kernels of other streams (seen here as a rare refusal by the
      // create-time self-check: the memset landed after the bias job had stored its flag, and the inertia job ran into its time limit)
      HIP_TRY(hipMemsetAsync(m->zv_flags.ptr, 0, m->zv_flags.bytes, stream));
      m->zv_epoch = 0;
   }
   if (!m->zv_error_host)
   {
      HIP_TRY(hipHostMalloc((void **)&m->zv_error_host, 2 * sizeof(int), hipHostMallocMapped)); // [0] the error word, [1] the poison word's host copy
      m->zv_error_host[0] = m->zv_error_host[1] = 0;
      HIP_TRY(hipHostGetDevicePointer((void **)&m->zv_error_dev, m->zv_error_host, 0));
      std::lock_guard<std::mutex> lock(g_error_words_mutex);
      g_error_words.push_back(m->zv_error_host);
   }
   return MH_OK;
}
// jobs = 2: A.in3b = tau, A.outb = qdd.  jobs = 3: additionally A.in3 = qdd, A.out = tau.  Returns hipErrorNotSupported (as int) in *rc
// when the code object lacks the plan.
mh_status zv_launch(mh_model *m, mh::Args<double> &A, int jobs, hipStream_t stream, int *rc)
{
   mh_status st = zv_prepare(m, A.B, stream);
   if (st != MH_OK)
      return st;
   if (m->zv_epoch == 0x7fffffff)
   { // the flags have seen every positive value: start over
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemsetAsync(m->zv_flags.ptr, 0, m->zv_flags.bytes, stream));
      m->zv_epoch = 0;
   }
   const bool self_signal = zv_self_signalling(m);
   if (self_signal && *(volatile int *)(m->zv_error_host + 1) != 0)
   { // A consumer of this context gave up (mh_zv_kernels.h: zv_take_cols): its producer may have published AFTERWARDS, into a matrix nobody
     // reset -- every launch since has written NaN rows.  Wait for whatever is still running, sentinels everywhere, poison word cleared.
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)m->zv_cols.ptr, 0x7ff4a5a5, m->zv_cols.bytes / 4, stream));
      HIP_TRY(hipMemsetAsync(m->zv_flags.ptr, 0, m->zv_flags.bytes, stream));
      m->zv_epoch = 0;
      *(volatile int *)(m->zv_error_host + 1) = 0;
   }
   const int epoch = ++m->zv_epoch;
   int flags = SPEC_IO_LDS | (m->ident_maps ? SPEC_IDENT : 0);
   *rc = m->spec.launch_zv(flags, &A, self_signal ? m->zv_cols.ptr : m->zv_tau.ptr, (int *)m->zv_flags.ptr, m->zv_error_dev, epoch, jobs, m->zv_same_l2,
                           m->zv_wait_ticks, (void *)stream);
   if (*rc != 0 && *rc != (int)hipErrorNotSupported)
      return fail(MH_ERR_HIP, "bias-split kernel launch failed: %s", hipGetErrorString((hipError_t)*rc));
   return MH_OK;
}

// Forward dynamics of device-filling batches as two launches at two workgroups per CU (mh_zv_kernels.h, spec_zvb_*): the same calls the
// bias split serves (AoS matrices, dense index maps, every joint an effort source), taken from two groups of 64 configurations per CU
// upwards -- below that the one-job kernel gives every group a CU of its own and is faster (humanoid, one MI355X: 24.2 against 20.1 us
// at 16 384, 36.8 against 37.3 at 32 768, 64.9 against 73.1 at 65 536, 214.5 against 248.4 at 262 144: profiles/r04_zvb_vs_tree_split.txt).
bool zvb_ok(const mh_model *m, int64_t B, bool soa)
{
   if (!m->spec.launch_zvb || !m->spec.zvb_usable || !m->spec.zvb_usable() || !m->use_spec || !m->use_zvb || m->use_split == 0)
      return false;
   if (soa || !m->dense_maps || m->force_io == 0 || m->n_locked > 0)
      return false;
   return m->use_zvb == 2 || (B + 63) / 64 >= 2 * (long)m->cu_count;
}
// ... and as ONE launch where the code object has the fused kernel (joints below the root all revolute / fixed, identity index maps)
bool zvf_ok(const mh_model *m, int64_t B, bool soa)
{
   if (!m->spec.launch_zvf || !m->spec.zvf_usable || !m->spec.zvf_usable() || !m->use_spec || !m->use_zvf || m->use_split == 0)
      return false;
   if (soa || !m->dense_maps || !m->ident_maps || m->force_io == 0 || m->n_locked > 0)
      return false;
   // measured (humanoid, one MI355X, profiles/r04_zvf_vs_others.txt): 20.5 us against the one-job kernel's 20.3 at 16 384 (one group per CU:
   // a tie), 27.0 against 36.4 at 24 576, 28.4 against 37.8 at 32 768, 195.9 against 250.5 at 262 144
   return m->use_zvf == 2 || (B + 63) / 64 > (long)m->cu_count;
}
// Inverse dynamics of device-filling batches in a persistent loop that requests the next group's rows behind the trunk pass
// (spec_zvb_bias_kernel<.., BIAS = false>): AoS matrices, dense index maps; from three groups of 64 configurations per CU upwards, where
// every workgroup of the launch takes a second turn.  MH_RNEA_AHEAD=0: never, 2: whenever the call qualifies.
bool rnea_ahead_ok(const mh_model *m, int64_t B, bool soa)
{
   if (!m->spec.launch_rnea_ahead || !m->spec.zvb_usable || !m->spec.zvb_usable() || !m->use_spec || !m->use_rnea_ahead || m->use_split == 0)
      return false;
   if (soa || !m->dense_maps || m->force_io == 0)
      return false;
   return m->use_rnea_ahead == 2 || (B + 63) / 64 > 2 * (long)m->cu_count;
}
mh_status zvb_launch(mh_model *m, mh::Args<double> &A, hipStream_t stream, int *rc)
{
   const size_t padded = (size_t)((A.B + 63) / 64 * 64);
   mh_status st = ensure_bytes(m->zv_tau, (size_t)A.B * m->nv * sizeof(double));
   if (st == MH_OK)
      st = ensure_bytes(m->zvb_cs, std::max<size_t>(1, (size_t)m->spec.zvb_cs_rows()) * padded * sizeof(double));
   if (st != MH_OK)
      return st;
   const int flags = SPEC_IO_LDS | (m->ident_maps ? SPEC_IDENT : 0);
   const long groups = std::min<long>((A.B + 63) / 64, (long)m->cu_count * 2);
   *rc = m->spec.launch_zvb(flags, &A, m->zv_tau.ptr, m->zvb_cs.ptr, (long)padded, (int)groups, m->zvb_which, (void *)stream);
   if (*rc != 0 && *rc != (int)hipErrorNotSupported)
      return fail(MH_ERR_HIP, "two-launch forward dynamics failed to launch: %s", hipGetErrorString((hipError_t)*rc));
   return MH_OK;
}

enum Algo
{
   ALGO_RNEA,
   ALGO_ABA,
   ALGO_CRBA
};

// Depth-first run-time-topology kernels (mh_dfs_kernels.h): homes of the stack frames, where ABA's hand-over lives, grid, launch.
//
// Frame homes.  A frame (non-leaf bodies only) is written when its body is visited and read when the body is popped, plus one
// read-modify-write per child that is not the last: stack traffic is proportional to the number of non-leaf bodies, and most of those
// sit near the leaves.  On an all-global stack the 128-body tree of BASELINE.json's configs[4] moved 5.3x (RNEA) and 17.7x (ABA) its
// algorithmic bytes through HBM at 4.2 / 5.7 TB/s (profiles/r02_config5_dfs_hbm_pmc.json): the kernels were bound by their own
// workspace.  An all-LDS stack needs 37 KB (RNEA) / 100+ KB (ABA) per wave there, i.e. 1-4 waves per CU, and loses more than it saves.
// So LDS is given a BUDGET per wave (what is left of 160 KB at the occupancy the launch wants) and filled from the leaves upwards: a
// frame is placed in LDS if it fits on top of the deepest LDS path below it.  The live frames of a walk are one root-to-leaf path, so
// every path keeps its LDS sum within the budget; the frames that do not fit -- few, near the root -- go to the wave's global block,
// whose offsets count global-homed ancestors only.
const mh_model::DfsPlan *dfs_plan(mh_model *m, int algo, int budget)
{
   if (m->parent)
      m = m->parent; // the plans (device copies of the body records) belong to the model; its contexts share them
   std::lock_guard<std::mutex> lock(m->dfs_mutex);
   for (const mh_model::DfsPlan &p : m->dfs_plans)
      if (p.algo == algo && p.budget == budget)
         return &p;
   const int n = m->n;
   std::vector<int> meta = m->meta, frame(n), below(n, 0), lofs(n, 0), gofs(n, 0);
   std::vector<char> home(n, 0);
   auto MI = [&](int e, int k) -> int & { return meta[(size_t)e * mh::MI_STRIDE + k]; };
   for (int e = 0; e < n; e++) // algo 2: the fused RNEA + ABA walk (the forward dynamics' frame + the inverse dynamics' wrench and acceleration)
      frame[e] = algo == 0 ? mh::rnea_frame_slots(MI(e, mh::MI_TYPE), MI(e, mh::MI_NCH))
                           : (algo == 1 ? mh::aba_frame_slots(MI(e, mh::MI_TYPE), MI(e, mh::MI_NCH)) : mh::pair_frame_slots(MI(e, mh::MI_TYPE), MI(e, mh::MI_NCH)));
   // (The inverse dynamics at twelve waves per CU -- 48 slots per lane -- keeps the old placement: it waits on its frames more than it
   // moves them, and the frames next to the leaves are the ones read back right after they were written: 2.92 ms against 3.00 at 1 M
   // configurations, while at eight waves the knapsack wins 2 %: profiles/r05_c5_frame_placement.txt.)
   bool all_fit = true;
   { // rounds 2-4: from the leaves upwards, whatever the frame is worth
      for (int e = n - 1; e >= 0; e--)
      { // engine order is depth-first: children come after their parent
         int need = below[e];
         if (frame[e] > 0 && below[e] + frame[e] <= budget)
            home[e] = 1, need += frame[e];
         else if (frame[e] > 0)
            all_fit = false;
         const int pe = MI(e, mh::MI_PARENT);
         if (pe >= 0)
            below[pe] = std::max(below[pe], need);
      }
   }
   if (!(m->dfs_place_greedy || (algo == 0 && budget < 64) || all_fit)) // (every frame in LDS already: nothing to choose)
   { // Round 5: by what a frame in LDS SAVES.  A frame is touched 2 (6 + jx) times under a single child, but under k children it is
     // written at the visit, re-read by every later child (v, w / a), read and written by the pop of every child that is not the last
     // (the 27 accumulators of the forward dynamics, the 6 of the inverse dynamics) and read at its own pop: 100 accesses for 47 slots at
     // k = 2, 500 at k = 8, against 16 for 14 under one child.  The budget binds along every root-to-leaf path, so the best set of homes
     // is a knapsack on the tree: best[e][b] = the accesses saved in e's subtree with b slots left for it = max(sum of best[c][b] over
     // the children (e global), worth(e) + sum of best[c][b - frame(e)] (e in LDS)).  128-body tree of configs[4], 80 slots per lane:
     // 5 314 -> 4 680 global slot accesses per configuration in the fused walk (model), 5 144 -> 4 134 for the forward dynamics at 48.
      std::vector<int> worth(n, 0);
      std::vector<int> with_subtree(n, 0); // children that have children of their own: all but the last of them accumulate in the frame
      for (int e = 0; e < n; e++)          // (the leaves are walked behind them and add to the carry: mh_model_create, build_program)
         if (MI(e, mh::MI_PARENT) >= 0 && MI(e, mh::MI_NCH) > 0)
            with_subtree[MI(e, mh::MI_PARENT)]++;
      for (int e = 0; e < n; e++)
      {
         const int k = MI(e, mh::MI_NCH), jx = mh::jx_slots(MI(e, mh::MI_TYPE)), in_frame = std::max(0, with_subtree[e] - 1);
         if (k == 0)
            continue;
         if (algo == 0)
            worth[e] = k == 1 ? 2 * (6 + jx) : (6 + jx + 12) + (k - 1) * 12 + 6 + in_frame * 12 + (6 + jx);
         else
         {
            const int id = algo == 2 ? 6 : 0; // the inverse dynamics' wrench (and acceleration) beside the forward dynamics' slots
            const int acc = 27 + id;
            worth[e] = k == 1 ? 2 * (6 + jx) + 2 * id
                              : (12 + jx + 6 + id) + (k - 1) * (12 + id) + (in_frame > 0 ? acc + (in_frame - 1) * 2 * acc + acc : 0) + (12 + jx);
         }
      }
      const int W = budget + 1;
      std::vector<long> best((size_t)n * W, 0);
      std::vector<char> take((size_t)n * W, 0);
      std::vector<std::vector<int>> kids(n);
      for (int e = 0; e < n; e++)
         if (MI(e, mh::MI_PARENT) >= 0)
            kids[MI(e, mh::MI_PARENT)].push_back(e);
      for (int e = n - 1; e >= 0; e--) // children come after their parent: their rows are complete
         for (int b = 0; b <= budget; b++)
         {
            long out = 0, in = -1;
            for (int c : kids[e])
               out += best[(size_t)c * W + b];
            if (frame[e] > 0 && frame[e] <= b)
            {
               in = worth[e];
               for (int c : kids[e])
                  in += best[(size_t)c * W + b - frame[e]];
            }
            best[(size_t)e * W + b] = std::max(out, in);
            take[(size_t)e * W + b] = in > out;
         }
      std::vector<int> left(n, budget);
      for (int e = 0; e < n; e++)
      {
         const int pe = MI(e, mh::MI_PARENT);
         if (pe >= 0)
            left[e] = left[pe] - (home[pe] ? frame[pe] : 0);
         home[e] = take[(size_t)e * W + left[e]];
      }
   }
   mh_model::DfsPlan plan{algo, budget, 0, 0, 0, nullptr};
   for (int e = 0; e < n; e++)
   {
      const int pe = MI(e, mh::MI_PARENT);
      if (pe >= 0)
         lofs[e] = lofs[pe] + (home[pe] ? frame[pe] : 0), gofs[e] = gofs[pe] + (home[pe] ? 0 : frame[pe]);
      if (home[e])
         plan.lds_slots = std::max(plan.lds_slots, lofs[e] + frame[e]);
      else if (frame[e] > 0)
         plan.glb_slots = std::max(plan.glb_slots, gofs[e] + frame[e]), plan.glb_frames++;
   }
   auto code = [&](int e) { return home[e] ? (lofs[e] | mh::DFS_LDS) : gofs[e]; };
   for (int e = 0; e < n; e++)
   {
      const int pe = MI(e, mh::MI_PARENT);
      const int pj = pe >= 0 ? mh::jx_slots(MI(pe, mh::MI_TYPE)) : 0;
      if (algo == 0)
      {
         MI(e, mh::MI_DFS_R) = code(e);
         if (pe >= 0)
            MI(e, mh::MI_PFR_R) = code(pe), MI(e, mh::MI_PVA_R) = code(pe) + 6 + pj;
      }
      else
      {
         MI(e, mh::MI_DFS_A) = code(e);
         if (pe >= 0)
            MI(e, mh::MI_PFR_A) = code(pe), MI(e, mh::MI_PV_A) = code(pe) + 12 + pj, MI(e, mh::MI_PACC_A) = code(pe) + 18 + pj;
         if (algo == 2 && pe >= 0)
         { // the inverse dynamics' slots of the parent's frame: behind the forward dynamics' part
            const int pa = mh::aba_frame_slots(MI(pe, mh::MI_TYPE), MI(pe, mh::MI_NCH));
            MI(e, mh::MI_PFR_R) = code(pe) + pa, MI(e, mh::MI_PVA_R) = code(pe) + pa + 6;
         }
      }
   }
   plan.glb_slots = std::max(plan.glb_slots, 6);
   if (hipMalloc((void **)&plan.d_meta, meta.size() * sizeof(int)) != hipSuccess)
      return nullptr;
   if (hipMemcpy(plan.d_meta, meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
   {
      (void)hipFree(plan.d_meta);
      return nullptr;
   }
   m->dfs_plans.push_back(plan);
   return &m->dfs_plans.back();
}
void dfs_plans_drop(mh_model *m)
{
   std::lock_guard<std::mutex> lock(m->dfs_mutex);
   for (mh_model::DfsPlan &p : m->dfs_plans)
      (void)hipFree(p.d_meta);
   m->dfs_plans.clear();
}

// Launch policy.  Occupancy first: the grid wants min(resident cap, waves of the batch) waves, spread over the CUs; the LDS a wave may
// use is 160 KB divided by the waves per CU that follow from it (ABA in fp64 holds the whole register file: 4 waves per CU at most).
// Out of that come the row windows (RNEA on AoS matrices), ABA's hand-over if all of it fits (small models at one wave per CU: measured
// 106 vs 116 us on the humanoid at B = 4096), and the rest is the stack's budget.  MH_DFS_PLACE = 0 | 1 | 2 forces an all-LDS stack
// with the hand-over in LDS / an all-LDS stack / an all-global stack (measurements, tests); MH_DFS_BUDGET=<slots> the budget itself.
// ---- run-time tree split (mh_split_kernels.h): trunk / limbs / owners from the tree alone, made once per model.
// Greedy: the limbs start as the trees of the forest; the largest limb is split at its first branching (the chain down to it joins the
// trunk, the branches become limbs) as long as that shortens the estimated critical path  (trunk bodies) + (bodies of the busiest wave).
void split_rt_free(mh_model *m)
{
   mh_model::SplitRt &S = m->split_rt;
   for (int k = 0; k < 3; k++)
      (void)hipFree(S.d_meta[k]), (void)hipFree(S.d_xl[k]), S.d_meta[k] = S.d_xl[k] = nullptr;
   (void)hipFree(S.d_trunk), (void)hipFree(S.d_seg), (void)hipFree(S.d_xl_ofs);
   S.d_trunk = S.d_seg = S.d_xl_ofs = nullptr;
   S.usable = false;
}
mh_status split_rt_upload_meta(mh_model *m)
{ // the adapted body records: the model's with the (body, field, value) patches of the plan applied, then every workspace slot number
  // turned into a slot CODE (home bit) for the precision's LDS share
   mh_model::SplitRt &S = m->split_rt;
   if (!S.usable)
      return MH_OK;
   static const int slot_fields[] = {mh::MI_SLOT_JP, mh::MI_SLOT_F, mh::MI_SLOT_VA, mh::MI_SLOT_C, mh::MI_SLOT_IA, mh::MI_SLOT_LK, mh::MI_HAND};
   for (int k = 0; k < 3; k++)
   {
      const long elem = k == 0 ? 4 : 8;
      const long cap = 160 * 1024 / (64 * elem);
      S.lds_slots[k] = k == 2 ? 0 : (int)(S.slots <= cap ? S.slots : cap - mh::SPLIT_LDS_MARGIN); // everything, or a share with room for a group
      std::vector<int> meta = m->meta;
      for (size_t i = 0; i + 2 < S.meta.size(); i += 3)
         meta[(size_t)S.meta[i] * mh::MI_STRIDE + S.meta[i + 1]] = S.meta[i + 2];
      auto code = [&](int slot) { return slot >= 0 && slot < S.lds_slots[k] ? (slot | mh::DFS_LDS) : slot; };
      for (int e = 0; e < m->n; e++)
         for (int f : slot_fields)
            meta[(size_t)e * mh::MI_STRIDE + f] = code(meta[(size_t)e * mh::MI_STRIDE + f]);
      std::vector<int> xl = S.xl;
      for (int &x : xl)
         x = code(x);
      if (!S.d_meta[k])
         HIP_TRY(hipMalloc((void **)&S.d_meta[k], meta.size() * sizeof(int)));
      HIP_TRY(hipMemcpy(S.d_meta[k], meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice));
      if (!S.d_xl[k])
         HIP_TRY(hipMalloc((void **)&S.d_xl[k], xl.size() * sizeof(int)));
      HIP_TRY(hipMemcpy(S.d_xl[k], xl.data(), xl.size() * sizeof(int), hipMemcpyHostToDevice));
   }
   return MH_OK;
}
void split_rt_plan(mh_model *m)
{
   mh_model::SplitRt &S = m->split_rt;
   const int n = m->n, W = mh::SPLIT_WAVES;
   auto MI = [&](int e, int k) { return m->meta[(size_t)e * mh::MI_STRIDE + k]; };
   std::vector<std::vector<int>> ch(n);
   std::vector<int> sz(n, 1), cnt(n, 1), roots; // sz: cost of the subtree in tenths of a 1-DoF body step; cnt: bodies in it
   for (int e = 0; e < n; e++)
   { // measured on the sweep kernels: a 6-DoF joint (LDL^T solve, general transforms) costs about 2.5 revolute steps, a 3-DoF joint 2
      const int t = MI(e, mh::MI_TYPE);
      sz[e] = t == MH_JOINT_SIXDOF ? 25 : ((t == MH_JOINT_PLANAR || t == MH_JOINT_SPHERICAL) ? 20 : (t == MH_JOINT_FIXED ? 4 : 10));
   }
   const std::vector<int> own = sz;
   for (int e = n - 1; e >= 0; e--)
   {
      const int pe = MI(e, mh::MI_PARENT);
      if (pe >= 0)
         sz[pe] += sz[e], cnt[pe] += cnt[e];
   }
   for (int e = 0; e < n; e++)
   {
      const int pe = MI(e, mh::MI_PARENT);
      (pe >= 0 ? ch[pe] : roots).push_back(e);
   }
   const int trunk_weight = getenv("MH_SPLIT_RT_TRUNK_WEIGHT") ? atoi(getenv("MH_SPLIT_RT_TRUNK_WEIGHT")) : 2; // in half bodies; measured (tools/exp_split_rt_weight.py): 1..3 tie, 4+ splits too little
   std::vector<char> trunk(n, 0);
   std::vector<int> limbs = roots;
   auto estimate = [&](const std::vector<int> &L, int nt, std::vector<int> *owner) {
      std::vector<int> order(L.size());
      for (size_t i = 0; i < L.size(); i++)
         order[i] = (int)i;
      std::sort(order.begin(), order.end(), [&](int a, int b) { return sz[L[a]] != sz[L[b]] ? sz[L[a]] > sz[L[b]] : L[a] < L[b]; });
      int load[mh::SPLIT_WAVES] = {}, cnt[mh::SPLIT_WAVES] = {};
      if (owner)
         owner->assign(L.size(), 0);
      for (int i : order)
      {
         int w = 0;
         for (int k = 1; k < W; k++)
            if (load[k] < load[w])
               w = k;
         load[w] += sz[L[i]], cnt[w]++;
         if (owner)
            (*owner)[i] = w;
      }
      int mx = 0, mc = 0;
      for (int k = 0; k < W; k++)
         mx = std::max(mx, load[k]), mc = std::max(mc, cnt[k]);
      return mc > mh::SPLIT_MAX_SEG ? 1 << 30 : (trunk_weight * nt + 1) / 2 + mx; // a trunk body: light outward steps on every wave + its fold on one while three wait
   };
   int best = estimate(limbs, 0, nullptr), nt = 0, ntc = 0; // trunk cost / trunk bodies
   std::vector<int> best_limbs = limbs;
   std::vector<char> best_trunk = trunk;
   int best_nt = 0, best_ntc = 0;
   for (int iter = 0; iter < n; iter++)
   {
      int big = -1;
      for (size_t i = 0; i < limbs.size(); i++)
         if (big < 0 || sz[limbs[i]] > sz[limbs[big]])
            big = (int)i;
      if (big < 0)
         break;
      int r = limbs[big];
      while (ch[r].size() == 1)
         r = ch[r][0];
      if (ch[r].empty())
         break; // the largest limb is a chain: it cannot be split
      for (int b = limbs[big];; b = ch[b][0])
      {
         trunk[b] = 1, nt += own[b], ntc++;
         if (b == r)
            break;
      }
      limbs.erase(limbs.begin() + big);
      for (int c : ch[r])
         limbs.push_back(c);
      const int est = estimate(limbs, nt, nullptr);
      if (est < best)
         best = est, best_limbs = limbs, best_trunk = trunk, best_nt = nt, best_ntc = ntc;
   }
   int sz_total = 0;
   for (int r0 : roots)
      sz_total += sz[r0];
   S.usable = false;
   if (best_limbs.size() < 2 || best > (3 * sz_total) / 4)
      return; // a chain, or nothing to gain
   limbs = best_limbs, trunk = best_trunk;
   std::vector<int> owner;
   S.est = estimate(limbs, best_nt, &owner);
   // per wave: limbs in ascending order; exchange records behind the sweep kernels' slots
   std::vector<int> seg((size_t)W * mh::SPLIT_MAX_SEG * 2, 0), xslot(n, -1), trunk_list;
   for (int k = 0; k < W; k++)
      S.n_seg[k] = 0;
   std::vector<int> by_start(limbs.size());
   for (size_t i = 0; i < limbs.size(); i++)
      by_start[i] = (int)i;
   std::sort(by_start.begin(), by_start.end(), [&](int a, int b) { return limbs[a] < limbs[b]; });
   int slots = m->n_slots;
   for (int i : by_start)
   {
      const int w = owner[i], r = limbs[i];
      seg[((size_t)w * mh::SPLIT_MAX_SEG + S.n_seg[w]) * 2] = r, seg[((size_t)w * mh::SPLIT_MAX_SEG + S.n_seg[w]) * 2 + 1] = r + cnt[r];
      S.n_seg[w]++;
      if (MI(r, mh::MI_PARENT) >= 0)
         xslot[r] = slots, slots += 27;
   }
   for (int e = 0; e < n; e++)
      if (trunk[e])
         trunk_list.push_back(e);
   // adapted records: (body, field, value) triples applied on top of the model's records
   std::vector<int> nflags(n), nva(n, -1), nia(n, -1);
   for (int e = 0; e < n; e++)
      nflags[e] = MI(e, mh::MI_FLAGS);
   for (int e = 0; e < n; e++)
   {
      if (xslot[e] >= 0)
         nflags[e] &= ~mh::MF_PARENT_ADJ; // a limb root: its parent's state comes from the workspace, its contribution goes to the exchange record
      if (!trunk[e])
         continue;
      bool limb_child = false, nonadj_trunk_child = false;
      int first_acc = -1;
      for (int c : ch[e])
      {
         if (!trunk[c])
            limb_child = true;
         else if (c != e + 1)
            nonadj_trunk_child = true, first_acc = std::max(first_acc, c);
      }
      nflags[e] &= ~(mh::MF_HAS_ACC | mh::MF_STORE_VA);
      if (nonadj_trunk_child)
         nflags[e] |= mh::MF_HAS_ACC;
      if (limb_child || nonadj_trunk_child)
      {
         nflags[e] |= mh::MF_STORE_VA;
         if (!(MI(e, mh::MI_FLAGS) & mh::MF_STORE_VA))
            nva[e] = slots, slots += 12; // the model's records hold no slots for it
      }
      if (nonadj_trunk_child && !(MI(e, mh::MI_FLAGS) & mh::MF_HAS_ACC))
         nia[e] = slots, slots += 40;
      for (int c : ch[e])
         if (trunk[c] && c != e + 1) // first contributor of the trunk-only fold: the highest index
            nflags[c] = (nflags[c] & ~mh::MF_ACC_FIRST) | (c == first_acc ? mh::MF_ACC_FIRST : 0);
   }
   S.meta.clear();
   auto patch = [&](int e, int field, int value) { S.meta.push_back(e), S.meta.push_back(field), S.meta.push_back(value); };
   std::vector<int> xl_ofs(trunk_list.size() + 1, 0), xl;
   for (int e = 0; e < n; e++)
   {
      patch(e, mh::MI_HAND, xslot[e]);
      patch(e, mh::MI_FLAGS, nflags[e]);
      if (nva[e] >= 0)
         patch(e, mh::MI_SLOT_VA, nva[e]);
      if (nia[e] >= 0)
         patch(e, mh::MI_SLOT_IA, nia[e]);
   }
   for (size_t k = 0; k < trunk_list.size(); k++)
   {
      for (size_t i = 0; i < limbs.size(); i++)
         if (MI(limbs[i], mh::MI_PARENT) == trunk_list[k])
            xl.push_back(xslot[limbs[i]]);
      xl_ofs[k + 1] = (int)xl.size();
   }
   if (xl.empty())
      xl.push_back(0);
   if (trunk_list.empty())
      trunk_list.push_back(0);
   S.n_trunk = best_ntc, S.n_limbs = (int)limbs.size(), S.slots = slots, S.est = (S.est + 5) / 10, S.total = (sz_total + 5) / 10;
   auto up = [&](int **dst, const std::vector<int> &v) {
      return hipMalloc((void **)dst, v.size() * sizeof(int)) == hipSuccess && hipMemcpy(*dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
   };
   S.xl = xl;
   S.usable = up(&S.d_trunk, trunk_list) && up(&S.d_seg, seg) && up(&S.d_xl_ofs, xl_ofs);
   if (S.usable && split_rt_upload_meta(m) != MH_OK)
      S.usable = false;
   if (!S.usable)
      split_rt_free(m);
}
template <typename T>
mh_status launch_split_rt(Algo algo, mh_model *model, int64_t B, mh::Args<T> &A, hipStream_t stream)
{
   const mh_model::SplitRt &S = model->split_rt;
   const long groups = (B + 63) / 64;
   // workgroups per CU: the fp64 ABA holds ~300 registers (one wave per SIMD), the others fit two workgroups (measured on the humanoid at
   // B = 32768, two groups per CU: RNEA 47 us with two resident workgroups against 66 looping one; tools/exp_split_rt_wgs.py)
   static const int forced_wgs = getenv("MH_SPLIT_RT_WGS") ? std::max(1, atoi(getenv("MH_SPLIT_RT_WGS"))) : 0;
   const int wgs = forced_wgs ? forced_wgs : ((algo == ALGO_ABA && sizeof(T) == 8) ? 1 : 2);
   const int grid = (int)std::max<long>(1, std::min<long>(groups, (long)model->cu_count * wgs));
   // Which record set: everything in LDS when the block fits (no branches); else a share in LDS once the blocks of the workgroups of an
   // XCD outgrow its L2 (measured on the fp64 humanoid: 44 us all-global vs 47 with a share at B = 4096, 71 vs 50 at 8192); else all global.
   int k = sizeof(T) == 4 ? 0 : 1;
   if (S.lds_slots[k] < S.slots && (size_t)S.slots * 64 * sizeof(T) * ((size_t)grid / 8 + 1) <= (size_t)3 << 20)
      k = 2;
   if (model->split_rt_lds >= 0)
      k = model->split_rt_lds ? (sizeof(T) == 4 ? 0 : 1) : 2;
   if (grid > model->cu_count && (size_t)std::min(S.slots, S.lds_slots[k] + mh::SPLIT_LDS_MARGIN) * 64 * sizeof(T) > 80 * 1024)
      k = 2; // two workgroups per CU: an LDS share above half the CU's would serialise them
   const int mode = S.lds_slots[k] >= S.slots ? 0 : (S.lds_slots[k] == 0 ? 1 : 2);
   mh_status st = ensure_bytes(model->ws, (size_t)S.slots * (size_t)grid * 64 * sizeof(T));
   if (st != MH_OK)
      return st;
   A.ws = (T *)model->ws.ptr;
   mh::SplitDev P{};
   P.meta = S.d_meta[k], P.trunk = S.d_trunk, P.seg = S.d_seg, P.xl_ofs = S.d_xl_ofs, P.xl = S.d_xl[k];
   P.n_trunk = S.n_trunk, P.slots = S.slots;
   for (int w = 0; w < mh::SPLIT_WAVES; w++)
      P.n_seg[w] = S.n_seg[w];
   const size_t lds = mode == 1 ? 0 : (size_t)std::min(S.slots, S.lds_slots[k] + mh::SPLIT_LDS_MARGIN) * 64 * sizeof(T);
   const void *kern = nullptr;
#define MH_SPLIT_KERN(NAME) (mode == 0 ? (const void *)&mh::NAME<T, 0> : (mode == 1 ? (const void *)&mh::NAME<T, 1> : (const void *)&mh::NAME<T, 2>))
   kern = algo == ALGO_RNEA ? MH_SPLIT_KERN(rnea_split_kernel) : (algo == ALGO_ABA ? MH_SPLIT_KERN(aba_split_kernel) : MH_SPLIT_KERN(crba_split_kernel));
   if constexpr (sizeof(T) == 8)
      if (algo == ALGO_ABA)
      { // the machine code the pair call runs (mh::pair_split_kernel): the two calls then agree bit for bit
         P.roles = 2;
         kern = MH_SPLIT_KERN(pair_split_kernel);
      }
#undef MH_SPLIT_KERN
   if (lds > 64 * 1024 && model->lds_attr[kern] < lds)
   {
      HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      model->lds_attr[kern] = lds;
   }
   void *args[] = {(void *)&A, (void *)&P};
   HIP_TRY(hipLaunchKernel(kern, dim3(grid), dim3(256), args, lds, stream));
   return MH_OK;
}

// The pair call of a model without a code object at small batches: both algorithms in ONE launch of the run-time tree split
// (mh::pair_split_kernel): A.in3 = qdd, A.out = tau, A.in3b = tau, A.outb = qdd.  2 * groups workgroups, one per CU.
mh_status launch_split_rt_pair(mh_model *model, int64_t B, mh::Args<double> &A, hipStream_t stream)
{
   using T = double;
   const mh_model::SplitRt &S = model->split_rt;
   const long groups = (B + 63) / 64;
   const int grid = (int)(2 * groups);
   // the record set (hence the kernel instantiation) the single calls of this batch take (launch_split_rt): the pair call and the two
   // single calls then run the same machine code and agree bit for bit
   int k = 1;
   if (S.lds_slots[k] < S.slots && (size_t)S.slots * 64 * sizeof(T) * ((size_t)groups / 8 + 1) <= (size_t)3 << 20)
      k = 2;
   if (model->split_rt_lds >= 0)
      k = model->split_rt_lds ? 1 : 2;
   const int mode = S.lds_slots[k] >= S.slots ? 0 : (S.lds_slots[k] == 0 ? 1 : 2);
   mh_status st = ensure_bytes(model->ws, (size_t)S.slots * (size_t)grid * 64 * sizeof(T));
   if (st != MH_OK)
      return st;
   A.ws = (T *)model->ws.ptr;
   mh::SplitDev P{};
   P.meta = S.d_meta[k], P.trunk = S.d_trunk, P.seg = S.d_seg, P.xl_ofs = S.d_xl_ofs, P.xl = S.d_xl[k];
   P.n_trunk = S.n_trunk, P.slots = S.slots;
   for (int w = 0; w < mh::SPLIT_WAVES; w++)
      P.n_seg[w] = S.n_seg[w];
   const size_t lds = mode == 1 ? 0 : (size_t)std::min(S.slots, S.lds_slots[k] + mh::SPLIT_LDS_MARGIN) * 64 * sizeof(T);
   const void *kern = mode == 0 ? (const void *)&mh::pair_split_kernel<T, 0>
                                : (mode == 1 ? (const void *)&mh::pair_split_kernel<T, 1> : (const void *)&mh::pair_split_kernel<T, 2>);
   if (lds > 64 * 1024 && model->lds_attr[kern] < lds)
   {
      HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      model->lds_attr[kern] = lds;
   }
   void *args[] = {(void *)&A, (void *)&P};
   HIP_TRY(hipLaunchKernel(kern, dim3(grid), dim3(256), args, lds, stream));
   return MH_OK;
}

struct DfsChoice
{
   long per_cu, budget, hand, b_win, slot_bytes;
   bool hand_lds, occ3;
};
DfsChoice dfs_choose(const mh_model *model, Algo algo, size_t elem, int64_t B, bool win, bool pair = false)
{
   DfsChoice c{};
   const long waves = (B + 63) / 64;
   c.b_win = win ? 3L * mh::ROW_WIN * mh::ROW_PITCH * (long)elem : 0;
   c.slot_bytes = 64 * (long)elem;
   const long cus = model->cu_count;
   // resident waves per CU the kernel's registers allow (hipcc -Rpass-analysis=kernel-resource-usage, round 5): fp32 inverse dynamics on
   // SoA / transposed rows 134-136 VGPRs = three waves per SIMD (with the LDS windows of AoS rows 232-234: two); fp32 forward dynamics
   // 181-184 = two, or 168 in the OCC3 build (48 bytes of scratch) = three, taken beyond eight waves per CU, the fused pair walk 216-219 = two; fp64
   // 254-256 = one.  The grid used to be sized for eight everywhere: an inverse dynamics that could keep twelve waves per CU resident ran
   // with eight (1.88 against 1.56 ms at 524 288 configurations of the 128-body tree, profiles/r05_c5_occ.txt), and a fp64 walk planned
   // its LDS for eight waves of which four were resident.
   // Twelve resident waves per CU finish a round 1.32 x later than eight (measured: 13 % more throughput for 50 % more waves), and the
   // waves loop over the groups of 64 configurations: twelve are taken where they save enough ROUNDS to pay for that -- 196 608 (one round
   // of twelve instead of two of eight) and from 393 216 configurations upwards, not at 262 144 (two rounds either way: 2.05 against 1.9 ms)
   long reg_cap = elem == 8 ? 4 : 8;
   const long wpc = (waves + cus - 1) / cus;
   const bool twelve_pays = ((wpc + 11) / 12) * 132 < ((wpc + 7) / 8) * 100;
   if (elem == 4 && algo == ALGO_RNEA && !win && twelve_pays)
      reg_cap = 12;
   if (elem == 4 && algo == ALGO_ABA && !pair && model->dfs_aba_occ3 && twelve_pays)
      reg_cap = 12, c.occ3 = true; // the build with a register budget for three waves per SIMD (mh_dfs_kernels.h: OCC3)
   if (model->waves_per_cu_set)
      reg_cap = std::min<long>(reg_cap, model->waves_per_cu);
   c.per_cu = std::max<long>(1, std::min<long>(reg_cap, (waves + cus - 1) / cus));
   const long full_stack = pair ? model->pair_stack : (algo == ALGO_RNEA ? model->rnea_stack : model->aba_stack);
   c.hand = algo == ALGO_RNEA ? 0 : model->aba_hand;
   if (model->dfs_place >= 0)
   { // forced placements: give the stack what it needs and let the occupancy follow
      c.budget = model->dfs_place == 2 ? 0 : full_stack;
      c.hand_lds = model->dfs_place == 0 && algo == ALGO_ABA && (full_stack + c.hand) * c.slot_bytes + c.b_win <= 160 * 1024;
      if (c.budget * c.slot_bytes + c.b_win > 160 * 1024)
         c.budget = (160 * 1024 - c.b_win) / c.slot_bytes;
   }
   else
   {
      // (a wave's share of the 160 KB, rounded DOWN to 2 KB: LDS is allocated in blocks, and a share that fills 160 KB / per_cu to the byte
      // left room for per_cu - 1 workgroups only -- 98 304 configurations of the 128-body tree, six waves per CU wanted, five resident: 0.99 ms
      // against 0.61 with eight slots less, profiles/r05_c5_rnea_budget.txt)
      const long avail = (160 * 1024 / c.per_cu) / 2048 * 2048 - c.b_win;
      c.hand_lds = algo == ALGO_ABA && (full_stack + c.hand) * c.slot_bytes <= avail;
      c.budget = std::max<long>(0, std::min<long>(full_stack, (avail - (c.hand_lds ? c.hand * c.slot_bytes : 0)) / c.slot_bytes));
      if (model->dfs_budget >= 0)
         c.budget = std::min<long>(model->dfs_budget, c.budget);
   }
   return c;
}
bool dfs_windows(const mh_model *model, Algo algo, size_t elem, bool aos)
{ // AoS matrices with identity index maps and rows that span many cache lines: RNEA reads them through LDS windows (mh_dfs_kernels.h)
   return algo == ALGO_RNEA && aos && model->ident_maps && model->use_win && (long)model->nv * (long)elem >= 512;
}
// pair: the fused RNEA + ABA walk (aba_dfs_kernel<.., PAIR>; algo = ALGO_ABA, A.in3 = qdd, A.out = tau, A.in3b = tau in, A.outb = qdd out)
template <typename T>
mh_status launch_dfs(Algo algo, mh_model *model, int64_t B, mh::Args<T> &A, hipStream_t stream, bool pair = false)
{
   const long waves = (B + 63) / 64;
   const bool win = !pair && dfs_windows(model, algo, sizeof(T), A.q_es == 1 && A.v_es == 1);
   const DfsChoice ch = dfs_choose(model, algo, sizeof(T), B, win, pair);
   const long b_win = ch.b_win, slot_bytes = ch.slot_bytes, hand = ch.hand, budget = ch.budget, cus = model->cu_count;
   long per_cu = ch.per_cu;
   const bool hand_lds = ch.hand_lds;
   const mh_model::DfsPlan *plan = dfs_plan(model, pair ? 2 : (algo == ALGO_RNEA ? 0 : 1), (int)budget);
   if (!plan)
      return fail(MH_ERR_HIP, "depth-first kernels: the body records of the frame plan could not be uploaded");
   const long lds = (plan->lds_slots + (hand_lds ? hand : 0)) * slot_bytes + b_win;
   if (lds > 0)
      per_cu = std::max<long>(1, std::min<long>(per_cu, (160 * 1024) / lds));
   const int grid = (int)std::max<long>(1, std::min(waves, cus * per_cu));
   const long gslots = (hand_lds ? 0 : hand) + plan->glb_slots;
   mh_status st = ensure_bytes(model->ws, (size_t)gslots * (size_t)grid * 64 * sizeof(T));
   if (st != MH_OK)
      return st;
   A.ws = (T *)model->ws.ptr;
   A.ws_stride = gslots * 64; // per-wave block of the global workspace: [grid][slots][64 lanes] -- the same constant slot stride as in LDS
   A.m.meta = plan->d_meta;
   if (algo == ALGO_RNEA)
      A.m.rnea_stack = plan->lds_slots;
   else
      A.m.aba_stack = plan->lds_slots;
   // every frame in LDS / every frame global: builds without the per-group branch
   const int mode = plan->glb_frames == 0 ? 0 : (plan->lds_slots == 0 ? 1 : 2);
   const void *kern = nullptr;
   if (win)
      A.m.prog = model->d_prog_seq; // (the windows follow the matrices in engine order)
   if (algo == ALGO_RNEA)
   {
      if (win)
         kern = mode == 0 ? (const void *)&mh::rnea_dfs_kernel<T, true, 0> : (mode == 1 ? (const void *)&mh::rnea_dfs_kernel<T, true, 1> : (const void *)&mh::rnea_dfs_kernel<T, true, 2>);
      else
         kern = mode == 0 ? (const void *)&mh::rnea_dfs_kernel<T, false, 0> : (mode == 1 ? (const void *)&mh::rnea_dfs_kernel<T, false, 1> : (const void *)&mh::rnea_dfs_kernel<T, false, 2>);
   }
   else if (pair)
   {
      if constexpr (sizeof(T) == 4)
      {
         if (hand_lds)
            kern = mode == 0 ? (const void *)&mh::aba_dfs_kernel<T, true, false, 0, true> : (const void *)&mh::aba_dfs_kernel<T, true, false, 2, true>;
         else
            kern = mode == 0 ? (const void *)&mh::aba_dfs_kernel<T, false, false, 0, true>
                             : (mode == 1 ? (const void *)&mh::aba_dfs_kernel<T, false, false, 1, true> : (const void *)&mh::aba_dfs_kernel<T, false, false, 2, true>);
      }
      else
         return fail(MH_ERR_INVALID_ARGUMENT, "the fused depth-first pair walk is built in fp32 only");
   }
   else if (ch.occ3 && sizeof(T) == 4)
   {
      if constexpr (sizeof(T) == 4)
      {
         if (hand_lds)
            kern = mode == 0 ? (const void *)&mh::aba_dfs_kernel<T, true, false, 0, false, true> : (const void *)&mh::aba_dfs_kernel<T, true, false, 2, false, true>;
         else
            kern = mode == 0 ? (const void *)&mh::aba_dfs_kernel<T, false, false, 0, false, true>
                             : (mode == 1 ? (const void *)&mh::aba_dfs_kernel<T, false, false, 1, false, true> : (const void *)&mh::aba_dfs_kernel<T, false, false, 2, false, true>);
      }
   }
   else if (hand_lds)
      kern = mode == 0 ? (const void *)&mh::aba_dfs_kernel<T, true, false, 0> : (const void *)&mh::aba_dfs_kernel<T, true, false, 2>;
   else
      kern = mode == 0 ? (const void *)&mh::aba_dfs_kernel<T, false, false, 0> : (mode == 1 ? (const void *)&mh::aba_dfs_kernel<T, false, false, 1> : (const void *)&mh::aba_dfs_kernel<T, false, false, 2>);
   if (lds > 64 * 1024 && model->lds_attr[kern] < (size_t)lds)
   {
      HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      model->lds_attr[kern] = (size_t)lds;
   }
   void *args[] = {(void *)&A};
   HIP_TRY(hipLaunchKernel(kern, dim3(grid), dim3(64), args, (size_t)lds, stream));
   return MH_OK;
}

template <typename T>
mh_status launch(Algo algo, mh_model_t model, int64_t B, const T *q, const T *qd, const T *in3, const double gravity[3], const T *fext,
                 const mh_options *opts_in, T *out, const T *locked_in = nullptr, T *locked_out = nullptr, T *body_acc = nullptr,
                 T *body_twist = nullptr, bool bodies = false, double step_dt = 0.0, T *q_next = nullptr, T *qd_next = nullptr,
                 bool *stepped = nullptr, T *joint_wrench = nullptr)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (B == 0)
      return MH_OK; // an empty batch has nothing to read or write: NULL pointers are fine
   if (!q || !out || (algo != ALGO_CRBA && (!qd || !in3 || (!gravity && !opts.use_root_acceleration))))
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer");
   if (algo == ALGO_ABA && model->n_locked > 0 && !locked_in)
      return fail(MH_ERR_INVALID_ARGUMENT, "%d joint(s) are acceleration sources: forward dynamics needs their accelerations, use mh_aba_locked_f64",
                  model->n_locked);
   // the sweep kernels' per-body workspace (plain RNEA / ABA calls run on the depth-first kernels, which size their own)
   // fp64 ABA stays on the sweep kernel: the depth-first walk fuses passes one and two, which in fp64 costs the whole register file plus
   // scratch (512 registers + 320 B against 310 and none) -- measured slower at every batch size on every 25..30-body model (humanoid
   // 112 vs 128 us at B = 4096, 1.04 vs 1.30 ms at 262144; profiles/r02_dfs_kernels_rates.txt).  In fp32 it fits and wins (config 5).
   const bool dfs_aba = sizeof(T) == 4 || model->dfs_place >= 0 || model->dfs_aba64;
   const bool dfs = model->use_dfs && algo != ALGO_CRBA && !bodies && !(algo == ALGO_ABA && (model->n_locked > 0 || !dfs_aba));
   if (!dfs)
   {
      st = ensure_workspace(model, B, sizeof(T));
      if (st != MH_OK)
         return st;
   }
   const Launch L = plan_launch(model, B);
   hipStream_t stream = (hipStream_t)opts.stream;

   mh::Args<T> A{};
   A.m = dev_model<T>(model);
   A.B = B;
   A.q = q, A.qd = qd, A.in3 = in3, A.fext = fext, A.out = out;
   A.in3b = nullptr, A.outb = nullptr;
   A.body_acc = body_acc, A.body_twist = body_twist, A.joint_wrench = joint_wrench;
   A.dt = T(0), A.q_next = nullptr, A.qd_next = nullptr;
   A.ws = (T *)model->ws.ptr;
   A.ws_stride = L.lanes;
   const bool soa = opts.layout == MH_LAYOUT_SOA;
   A.q_bs = soa ? 1 : model->nq, A.q_es = soa ? B : 1;
   A.v_bs = soa ? 1 : model->nv, A.v_es = soa ? B : 1;
   A.f_bs = soa ? 1 : (long)model->n * 6, A.f_es = soa ? B : 1;
   set_root_acceleration(A, opts, gravity);
   A.coriolis = opts.consider_coriolis, A.accel = opts.consider_accelerations;
   const bool ldsc = MH_GENERIC_LDS_CONSTS || model->lds_consts;
   const size_t lds = ldsc ? (size_t)model->n * mh::MC_STRIDE * sizeof(T) : 0;
   if (lds > 160 * 1024)
      return fail(MH_ERR_BAD_DIMENSION, "model constants (%zu B) exceed the 160 KiB LDS of a gfx950 CU", lds);

   if (bodies && algo != ALGO_CRBA)
   { // per-body outputs: run-time-topology kernels (the model's joint source modes must all be effort sources)
      if (model->n_locked > 0 && algo == ALGO_ABA)
         return fail(MH_ERR_INVALID_ARGUMENT, "per-body outputs of forward dynamics are not available while joints are acceleration sources");
      if (sizeof(T) == 8 && !joint_wrench && split_ok(model, algo == ALGO_RNEA ? 0 : 1, B, soa))
      { // the tree-split kernels write them too (identity maps, rows staged in LDS); other plans: the run-time-topology kernels below
         const int sf = split_flags(model, algo == ALGO_RNEA ? 0 : 1, soa);
         if ((sf & SPEC_IDENT) && (sf & SPEC_IO_LDS))
         {
            const long groups = std::min<long>((B + 63) / 64, (long)model->cu_count * 2);
            const int rc = model->spec.launch_split(algo == ALGO_RNEA ? 0 : 1, sf | SPEC_BODIES, &A, (int)groups, (void *)stream);
            if (rc == 0)
               return MH_OK;
            if (rc != (int)hipErrorNotSupported)
               return fail(MH_ERR_HIP, "tree-split kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
         }
      }
      if (model->split_rt.usable && model->n_locked == 0 && (model->use_split_rt == 1 || (B + 63) / 64 <= (long)model->cu_count * 2))
         return launch_split_rt<T>(algo, model, B, A, stream); // small batches: the run-time tree split writes the per-body outputs too
      if (algo == ALGO_RNEA)
         { if (ldsc) hipLaunchKernelGGL((mh::rnea_kernel<T, true, true>), dim3(L.grid), dim3(L.block), lds, stream, A); else hipLaunchKernelGGL((mh::rnea_kernel<T, false, true>), dim3(L.grid), dim3(L.block), lds, stream, A); }
      else
         { if (ldsc) hipLaunchKernelGGL((mh::aba_kernel<T, true, false, true>), dim3(L.grid), dim3(L.block), lds, stream, A); else hipLaunchKernelGGL((mh::aba_kernel<T, false, false, true>), dim3(L.grid), dim3(L.block), lds, stream, A); }
      HIP_TRY(hipGetLastError());
      return MH_OK;
   }
   if (algo == ALGO_ABA && model->n_locked > 0)
   { // acceleration-source joints: run-time flags per joint, generic kernel only
      A.in3b = locked_in, A.outb = locked_out;
      { if (ldsc) hipLaunchKernelGGL((mh::aba_kernel<T, true, true>), dim3(L.grid), dim3(L.block), lds, stream, A); else hipLaunchKernelGGL((mh::aba_kernel<T, false, true>), dim3(L.grid), dim3(L.block), lds, stream, A); }
      HIP_TRY(hipGetLastError());
      return MH_OK;
   }
   if constexpr (sizeof(T) == 8)
   {
      // (a simulation step rides in the inertia job where the index maps are the identity: the two-stage form of the hand-off integrates the
      // rows it holds and writes the new state too -- 17.5 against 20.1 us per step at B = 4096, profiles/r04_step_rates.txt)
      if (algo == ALGO_ABA && (!q_next || (model->ident_maps && model->use_zv_step)) && zv_ok(model, B, soa, 2))
      { // forward dynamics as two jobs side by side: bias efforts | articulated inertias, then the bias fold (mh_zv_kernels.h)
         if (const mh_status se = zv_check_error(model); se != MH_OK)
            return se;
         A.in3b = in3, A.outb = out;
         if (q_next)
            A.dt = (T)step_dt, A.q_next = q_next, A.qd_next = qd_next;
         int rc = 0;
         if (const mh_status sz = zv_launch(model, A, 2, stream, &rc); sz != MH_OK)
            return sz;
         if (rc == 0)
         {
            if (q_next && stepped)
               *stepped = true;
            return MH_OK;
         }
         A.in3b = nullptr, A.outb = nullptr; // not in this code object: the plans below
         A.dt = T(0), A.q_next = nullptr, A.qd_next = nullptr;
      }
      if (algo == ALGO_ABA && (!q_next || model->use_zv_step) && zvf_ok(model, B, soa))
      { // device-filling batches: bias efforts, articulated inertias, fold and outward sweep of a group of 64 configurations by ONE workgroup
        // (a simulation step rides along: 32.3 against 42.4 us per step at 32 768, 226.5 against 286.7 at 262 144 -- profiles/r04_step_rates.txt)
         const long groups = std::min<long>((B + 63) / 64, (long)model->cu_count * 2);
         if (q_next)
            A.dt = (T)step_dt, A.q_next = q_next, A.qd_next = qd_next;
         const int rc = model->spec.launch_zvf(SPEC_IO_LDS | SPEC_IDENT, &A, (int)groups, (void *)stream);
         if (rc == 0)
         {
            if (q_next && stepped)
               *stepped = true;
            return MH_OK;
         }
         if (rc != (int)hipErrorNotSupported)
            return fail(MH_ERR_HIP, "fused forward dynamics failed to launch: %s", hipGetErrorString((hipError_t)rc));
         A.dt = T(0), A.q_next = nullptr, A.qd_next = nullptr;
      }
      if (algo == ALGO_ABA && !q_next && zvb_ok(model, B, soa))
      { // ... or as two launches: bias rows and (cos, sin) pairs by one, articulated inertias + fold + outward sweep by the next
         int rc = 0;
         if (const mh_status sz = zvb_launch(model, A, stream, &rc); sz != MH_OK)
            return sz;
         if (rc == 0)
            return MH_OK;
      }
   }
   if constexpr (sizeof(T) == 8)
   {
      if (algo == ALGO_RNEA && rnea_ahead_ok(model, B, soa))
      {
         const long groups = std::min<long>((B + 63) / 64, (long)model->cu_count * 2);
         const int rc = model->spec.launch_rnea_ahead(SPEC_IO_LDS | (model->ident_maps ? SPEC_IDENT : 0), &A, (int)groups, (void *)stream);
         if (rc == 0)
            return MH_OK;
         if (rc != (int)hipErrorNotSupported)
            return fail(MH_ERR_HIP, "inverse dynamics (rows requested ahead) failed to launch: %s", hipGetErrorString((hipError_t)rc));
      }
   }
   if (algo != ALGO_CRBA && sizeof(T) == 8 && split_ok(model, algo == ALGO_RNEA ? 0 : 1, B, soa))
   {
      int sf = split_flags(model, algo == ALGO_RNEA ? 0 : 1, soa);
      // device-filling RNEA batches without LDS rows (SoA): the build with a 168-register budget keeps three workgroups per CU busy
      // (124 -> 112 us at B = 262144); smaller batches are faster on the plain build
      const bool occ3 = algo == ALGO_RNEA && !(sf & SPEC_IO_LDS) && (sf & SPEC_IDENT) && (B + 63) / 64 > (long)model->cu_count * 2;
      if (occ3)
         sf |= SPEC_OCC3;
      const long groups = std::min<long>((B + 63) / 64, (long)model->cu_count * (occ3 ? 3 : 2));
      if (algo == ALGO_ABA && q_next && (sf & SPEC_IDENT) && (sf & SPEC_IO_LDS))
      { // fused simulation step: the kernel integrates the rows it holds in LDS and writes the new state too
         A.dt = (T)step_dt, A.q_next = q_next, A.qd_next = qd_next;
         if (stepped)
            *stepped = true;
      }
      const int rc = model->spec.launch_split(algo == ALGO_RNEA ? 0 : 1, sf, &A, (int)groups, (void *)stream);
      if (rc == 0)
         return MH_OK;
      if (rc != (int)hipErrorNotSupported)
         return fail(MH_ERR_HIP, "tree-split kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
      // a code object built without this plan (mh_build_code_object in its fast mode): the run-time-topology kernels serve the call
      A.dt = T(0), A.q_next = nullptr, A.qd_next = nullptr;
      if (stepped)
         *stepped = false;
   }
   if (model->spec.launch && algo != ALGO_CRBA && model->use_spec && sizeof(T) == 8)
   {
      // Topology-specialised code object (fp64).  State rows are staged in LDS when the layout is AoS and they fit; ABA's
      // inward -> outward hand-over lives in LDS while the batch is small enough that one wave per CU is all the device
      // would get anyway, otherwise in the global workspace.
      const int a = algo == ALGO_RNEA ? 0 : 1;
      const long waves = (B + 63) / 64;
      const long LDS_MAX = 160 * 1024;
      int flags = model->ident_maps ? SPEC_IDENT : 0;
      bool io = !soa && model->dense_maps && model->spec.supports(a, SPEC_IO_LDS) && model->spec.lds_bytes(a, SPEC_IO_LDS, model->nq, model->nv) <= LDS_MAX;
      if (model->force_io >= 0)
         io = io && model->force_io;
      if (io)
         flags |= SPEC_IO_LDS;
      if (algo == ALGO_ABA)
      {
         bool st = model->spec.lds_bytes(a, flags | SPEC_ST_LDS, model->nq, model->nv) <= LDS_MAX && waves <= (long)model->cu_count * model->lds_wave_factor;
         if (model->force_st >= 0)
            st = model->force_st && model->spec.lds_bytes(a, flags | SPEC_ST_LDS, model->nq, model->nv) <= LDS_MAX;
         if (st && !model->spec.supports(a, flags | SPEC_ST_LDS))
            flags &= ~SPEC_IO_LDS; // small batch: keep the hand-over in LDS, read the state rows directly
         if (st)
            flags |= SPEC_ST_LDS;
      }
      if (model->spec.supports(a, flags))
      {
      const long lds = model->spec.lds_bytes(a, flags, model->nq, model->nv);
      const long per_cu = lds > 0 ? std::max<long>(1, std::min<long>(8, LDS_MAX / lds)) : 8;
      const int grid = (int)std::max<long>(1, std::min(waves, (long)model->cu_count * per_cu));
      if (algo == ALGO_ABA && !(flags & SPEC_ST_LDS))
      {
         mh_status s2 = ensure_bytes(model->ws, (size_t)std::max(model->n_slots, model->spec.aba_slots()) * (size_t)grid * 64 * sizeof(T));
         if (s2 != MH_OK)
            return s2;
         A.ws = (T *)model->ws.ptr;
         A.ws_stride = (long)grid * 64;
      }
      const int rc = model->spec.launch(a, flags, &A, grid, (void *)stream);
      if (rc == 0)
         return MH_OK;
      if (rc != (int)hipErrorNotSupported)
         return fail(MH_ERR_HIP, "specialised kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
      A.ws = (T *)model->ws.ptr, A.ws_stride = L.lanes; // (a fast build without the whole-tree kernels: on to the run-time-topology kernels)
      } // else: this (algorithm, memory plan) is not in the code object -- the run-time-topology kernels below serve the call
   }
   // Small batches of a model whose tree branches: the tree split over the four waves of a workgroup (mh_split_kernels.h)
   if (model->split_rt.usable && algo != ALGO_CRBA && model->n_locked == 0 && !locked_in
       && (model->use_split_rt == 1 || (B + 63) / 64 <= (long)model->cu_count * 2)) // measured: profiles/r02_split_rt_sweep.txt
      return launch_split_rt<T>(algo, model, B, A, stream);
   // Run-time-topology RNEA / ABA on AoS matrices: for big batches of wide matrices go through transposed scratch copies
   // (mh::transpose_kernel).  External wrenches keep their own strides.  The depth-first RNEA reads AoS rows through LDS windows instead
   // (mh_dfs_kernels.h, RowWindow) unless MH_DFS_TRANSPOSE=1; the depth-first ABA has no registers left for windows and takes the copies.
   T *t_out = nullptr;
   if (algo != ALGO_CRBA && !soa)
   {
      bool want = model->use_transpose >= 0 ? model->use_transpose != 0 : (B >= 8192 && model->nq + model->nv >= 64);
      if (dfs && want)
         want = model->dfs_transpose >= 0 ? model->dfs_transpose != 0 : (algo == ALGO_ABA || !(model->ident_maps && model->use_win));
      if (want)
      {
         const size_t nq = model->nq, nv = model->nv;
         mh_status s3 = ensure_bytes(model->tr, (size_t)B * (nq + 3 * nv) * sizeof(T));
         if (s3 != MH_OK)
            return s3;
         T *t_q = (T *)model->tr.ptr, *t_qd = t_q + (size_t)B * nq, *t_in3 = t_qd + (size_t)B * nv;
         t_out = t_in3 + (size_t)B * nv;
         mh::transpose_rows<T>(q, t_q, (long)B, (long)nq, true, stream);
         mh::transpose_rows<T>(qd, t_qd, (long)B, (long)nv, true, stream);
         mh::transpose_rows<T>(in3, t_in3, (long)B, (long)nv, true, stream);
         A.q = t_q, A.qd = t_qd, A.in3 = t_in3, A.out = t_out;
         A.q_bs = 1, A.q_es = B, A.v_bs = 1, A.v_es = B;
      }
   }
   if (dfs)
   {
      st = launch_dfs<T>(algo, model, B, A, stream);
      if (st != MH_OK)
         return st;
   }
   else
   switch (algo)
   {
      case ALGO_RNEA:
         { if (ldsc) hipLaunchKernelGGL((mh::rnea_kernel<T, true>), dim3(L.grid), dim3(L.block), lds, stream, A); else hipLaunchKernelGGL((mh::rnea_kernel<T, false>), dim3(L.grid), dim3(L.block), lds, stream, A); }
         break;
      case ALGO_ABA:
         { if (ldsc) hipLaunchKernelGGL((mh::aba_kernel<T, true>), dim3(L.grid), dim3(L.block), lds, stream, A); else hipLaunchKernelGGL((mh::aba_kernel<T, false>), dim3(L.grid), dim3(L.block), lds, stream, A); }
         break;
      case ALGO_CRBA:
      {
         const size_t hbytes = (size_t)B * model->nv * model->nv * sizeof(T);
         A.v_bs = soa ? 1 : (long)model->nv * model->nv;
         if (model->spec.launch_crba && model->spec.crba_packed && model->use_spec && sizeof(T) == 8)
         {
            const int sflags = (model->ident_maps && model->dense_maps) ? SPEC_IDENT : 0;
            // tree-split form (4 waves per 64 configurations, coalesced write-out): measured faster at every batch size
            bool split = sflags && !soa && model->spec.launch_crba_split && model->spec.crba_split_usable && model->spec.crba_split_usable()
                         && model->use_split != 0;
            if (split)
            {
               // thin workgroups (16 / 32 of the 64 lanes) while that is what it takes to give every CU a workgroup: the write-out is
               // bound by the stores one CU can have in flight
               // (the width that gives every CU exactly one: 16 at 4 096, 24 at 6 000 -- any width up to 64 runs)
               int lpg = (int)std::max<long>(16, std::min<long>(64, (B + model->cu_count - 1) / model->cu_count));
               if (const char *e = getenv("MH_CRBA_LPG"))
                  lpg = std::max(1, std::min(64, atoi(e)));
               const long ng = (B + lpg - 1) / lpg;
               const int rc = model->spec.launch_crba_split(&A, (int)std::min<long>(ng, (long)model->cu_count * 2), lpg, (void *)stream);
               if (rc == 0)
                  return MH_OK;
               if (rc != (int)hipErrorNotSupported)
                  return fail(MH_ERR_HIP, "tree-split CRBA launch failed: %s", hipGetErrorString((hipError_t)rc));
            }
            const bool packed = model->spec.crba_packed(sflags) != 0;
            if (!packed)
               HIP_TRY(hipMemsetAsync(out, 0, hbytes, stream)); // direct-store kernel writes related entries only
            const int grid = packed ? (int)std::min<long>((B + 63) / 64, (long)model->cu_count) : L.grid;
            const int rc = model->spec.launch_crba(sflags, &A, grid, (void *)stream);
            if (rc == 0)
               return MH_OK;
            if (rc != (int)hipErrorNotSupported)
               return fail(MH_ERR_HIP, "specialised CRBA launch failed: %s", hipGetErrorString((hipError_t)rc));
            // not in this code object (fast build): the run-time-topology kernels below
         }
         HIP_TRY(hipMemsetAsync(out, 0, hbytes, stream));
         if (model->split_rt.usable && (model->use_split_rt == 1 || (B + 63) / 64 <= (long)model->cu_count * 2))
            return launch_split_rt<T>(algo, model, B, A, stream); // small batches: the tree split over four waves (mh_split_kernels.h)
         {
            const int parts = regressor_parts(model, L);
            if (const mh_status sp = ensure_parts_workspace(model, L, parts, sizeof(T)); sp != MH_OK)
               return sp;
            A.ws = (T *)model->ws.ptr;
            if (ldsc) hipLaunchKernelGGL((mh::crba_kernel<T, true>), dim3(L.grid, parts), dim3(L.block), lds, stream, A); else hipLaunchKernelGGL((mh::crba_kernel<T, false>), dim3(L.grid, parts), dim3(L.block), lds, stream, A);
         }
         break;
      }
   }
   if (t_out)
   {
      mh::transpose_rows<T>((const T *)t_out, out, (long)B, (long)model->nv, false, stream);
   }
   HIP_TRY(hipGetLastError());
   return MH_OK;
}

// Host-pointer front end (what a JNI / Panama shim with heap or off-heap arrays calls).  The batch is cut into chunks of rows that travel
// through a ring of three device slots on three streams: copy-in of chunk k+1, kernels of chunk k and copy-out of chunk k-1 overlap
// (PCIe is full duplex), kernels stay on ONE stream (they share the model's workspace).  The copies run at PCIe rate when the caller's
// matrices are pinned -- allocated with mh_host_alloc or registered with mh_host_register -- and at the runtime's staged rate otherwise.
// Returns when every output chunk has landed.  kind: ALGO_RNEA / ALGO_ABA / ALGO_CRBA, or PAIR: in3 = qdd, in4 = tau, out = tau_out,
// out2 = qdd_out (mh_rnea_aba_f64 per chunk).
enum
{
   HOST_PAIR = 100
};
mh_status host_pipeline_init(mh_model *m)
{
   if (m->hs_in)
      return MH_OK;
   HIP_TRY(hipStreamCreateWithFlags(&m->hs_in, hipStreamNonBlocking));
   HIP_TRY(hipStreamCreateWithFlags(&m->hs_run, hipStreamNonBlocking));
   HIP_TRY(hipStreamCreateWithFlags(&m->hs_out, hipStreamNonBlocking));
   for (int k = 0; k < 3; k++)
   {
      HIP_TRY(hipEventCreateWithFlags(&m->ev_in[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&m->ev_run[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&m->ev_out[k], hipEventDisableTiming));
   }
   return MH_OK;
}
template <typename T>
mh_status launch_host(int kind, mh_model_t model, int64_t B, const T *q, const T *qd, const T *in3, const T *in4, const double gravity[3],
                      const T *fext, const mh_options *opts_in, T *out, T *out2)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (B == 0)
      return MH_OK;
   const bool crba = kind == ALGO_CRBA, pair = kind == HOST_PAIR;
   if (!q || !out || (!crba && (!qd || !in3 || (!gravity && !opts.use_root_acceleration))) || (pair && (!in4 || !out2)))
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer");
   st = host_pipeline_init(model);
   if (st != MH_OK)
      return st;
   if (opts.stream)
      HIP_TRY(hipStreamSynchronize((hipStream_t)opts.stream)); // work the caller queued before this (synchronous) call
   const size_t nq = model->nq, nv = model->nv, nj = model->n;
   const bool soa = opts.layout == MH_LAYOUT_SOA;
   // rows per chunk: SoA matrices are not cut (a chunk of configurations is not contiguous there); AoS: about an eighth of the batch,
   // 4096 .. 16384 configurations, whole waves
   // (measured, tools/host_path.py: a chunk costs ~60-80 us of copy / event calls, so a 4096-configuration batch is faster whole: 0.39 ms in
   // four chunks against 0.2 ms in one; from 16384 configurations on the overlap wins)
   int64_t chunk = B;
   if (!soa && model->host_chunk > 0 && B > model->host_chunk)
      chunk = std::max<int64_t>(64, ((int64_t)model->host_chunk + 63) / 64 * 64);
   else if (!soa && model->host_chunk == 0 && B >= 16384)
      chunk = std::max<int64_t>(4096, std::min<int64_t>(16384, (B / 8 + 63) / 64 * 64));
   const size_t c_q = (size_t)chunk * nq, c_v = (size_t)chunk * nv, c_f = fext ? (size_t)chunk * nj * 6 : 0;
   const size_t c_out = crba ? (size_t)chunk * nv * nv : c_v;
   const size_t slot = c_q + (crba ? 0 : 2 * c_v) + (pair ? c_v : 0) + c_f + c_out + (pair ? c_v : 0);
   const int64_t n_chunks = (B + chunk - 1) / chunk;
   const int ring = n_chunks > 1 ? 3 : 1;
   st = ensure_bytes(model->stage, slot * ring * sizeof(T));
   if (st != MH_OK)
      return st;
   mh_options o = opts;
   o.stream = (void *)model->hs_run;
   for (int64_t k = 0; k < n_chunks; k++)
   {
      const int s = (int)(k % ring);
      const int64_t r0 = k * chunk, rows = std::min<int64_t>(chunk, B - r0);
      T *d_q = (T *)model->stage.ptr + slot * s, *d_qd = d_q + c_q, *d_in3 = d_qd + (crba ? 0 : c_v), *d_in4 = d_in3 + (crba ? 0 : c_v);
      T *d_f = d_in4 + (pair ? c_v : 0), *d_out = d_f + c_f, *d_out2 = d_out + c_out;
      if (k >= ring)
         HIP_TRY(hipStreamWaitEvent(model->hs_in, model->ev_run[s], 0)); // the kernels of the slot's previous tenant have read their inputs
      const size_t b_q = (size_t)rows * nq * sizeof(T), b_v = (size_t)rows * nv * sizeof(T);
      HIP_TRY(hipMemcpyAsync(d_q, q + (size_t)r0 * nq, b_q, hipMemcpyHostToDevice, model->hs_in));
      if (!crba)
      {
         HIP_TRY(hipMemcpyAsync(d_qd, qd + (size_t)r0 * nv, b_v, hipMemcpyHostToDevice, model->hs_in));
         HIP_TRY(hipMemcpyAsync(d_in3, in3 + (size_t)r0 * nv, b_v, hipMemcpyHostToDevice, model->hs_in));
         if (pair)
            HIP_TRY(hipMemcpyAsync(d_in4, in4 + (size_t)r0 * nv, b_v, hipMemcpyHostToDevice, model->hs_in));
         if (fext)
            HIP_TRY(hipMemcpyAsync(d_f, fext + (size_t)r0 * nj * 6, (size_t)rows * nj * 6 * sizeof(T), hipMemcpyHostToDevice, model->hs_in));
      }
      HIP_TRY(hipEventRecord(model->ev_in[s], model->hs_in));
      HIP_TRY(hipStreamWaitEvent(model->hs_run, model->ev_in[s], 0));
      if (k >= ring)
         HIP_TRY(hipStreamWaitEvent(model->hs_run, model->ev_out[s], 0)); // ... and their outputs have left the slot
      if constexpr (sizeof(T) == 8)
      {
         if (pair)
            st = mh_rnea_aba_f64(model, rows, d_q, d_qd, d_in3, d_in4, gravity, fext ? d_f : nullptr, &o, d_out, d_out2);
         else
            st = launch<T>((Algo)kind, model, rows, d_q, d_qd, d_in3, gravity, fext ? d_f : nullptr, &o, d_out);
      }
      else
         st = launch<T>((Algo)kind, model, rows, d_q, d_qd, d_in3, gravity, fext ? d_f : nullptr, &o, d_out);
      if (st != MH_OK)
      {
         (void)hipDeviceSynchronize();
         return st;
      }
      HIP_TRY(hipEventRecord(model->ev_run[s], model->hs_run));
      HIP_TRY(hipStreamWaitEvent(model->hs_out, model->ev_run[s], 0));
      const size_t b_o = crba ? (size_t)rows * nv * nv * sizeof(T) : b_v;
      HIP_TRY(hipMemcpyAsync(out + (size_t)r0 * (crba ? nv * nv : nv), d_out, b_o, hipMemcpyDeviceToHost, model->hs_out));
      if (pair)
         HIP_TRY(hipMemcpyAsync(out2 + (size_t)r0 * nv, d_out2, b_v, hipMemcpyDeviceToHost, model->hs_out));
      HIP_TRY(hipEventRecord(model->ev_out[s], model->hs_out));
   }
   HIP_TRY(hipStreamSynchronize(model->hs_out));
   return zv_check_error(model); // before the caller reads the outputs
}
} // namespace


namespace
{
// Host-only part of model creation: validation and the engine's joint order (depth-first, parents first).
struct Plan
{
   std::vector<int> order;     // engine index -> caller index
   std::vector<int> engine_of; // caller index -> engine index
   std::vector<int> dofo, cfgo; // per caller joint: offsets into the concatenated index maps
   std::vector<std::vector<int>> children; // by caller index
   std::vector<int> eparent, etype;        // engine order
   std::string key;
};

mh_status plan_model(const mh_model_desc *d, Plan &P)
{
   if (!d)
      return fail(MH_ERR_INVALID_ARGUMENT, "desc is NULL");
   const int n = d->n_joints;
   if (n <= 0)
      return fail(MH_ERR_INVALID_ARGUMENT, "n_joints = %d", n);
   if (!d->parent || !d->joint_type || !d->axis || !d->X_before || !d->X_com || !d->inertia_J || !d->inertia_mass || !d->inertia_com
       || !d->dof_indices || !d->cfg_indices)
      return fail(MH_ERR_INVALID_ARGUMENT, "a model array is NULL");
   if (d->nq < 0 || d->nv < 0)
      return fail(MH_ERR_BAD_DIMENSION, "nq = %d, nv = %d", d->nq, d->nv);
   P.dofo.assign(n + 1, 0), P.cfgo.assign(n + 1, 0);
   for (int i = 0; i < n; i++)
   {
      const int t = d->joint_type[i];
      if (t < MH_JOINT_REVOLUTE || t > MH_JOINT_SPHERICAL)
         return fail(MH_ERR_UNSUPPORTED_JOINT, "joint %d has unsupported kind %d", i, t);
      if (d->parent[i] < -1 || d->parent[i] >= n || d->parent[i] == i)
         return fail(MH_ERR_BAD_TOPOLOGY, "joint %d has parent %d", i, d->parent[i]);
      P.dofo[i + 1] = P.dofo[i] + joint_ndof(t);
      P.cfgo[i + 1] = P.cfgo[i] + joint_ncfg(t);
   }
   {
      std::vector<char> seen_v(d->nv, 0), seen_q(d->nq, 0);
      for (int k = 0; k < P.dofo[n]; k++)
      {
         const int r = d->dof_indices[k];
         if (r < 0 || r >= d->nv || seen_v[r])
            return fail(MH_ERR_BAD_TOPOLOGY, "dof_indices[%d] = %d is out of range or repeated (nv = %d)", k, r, d->nv);
         seen_v[r] = 1;
      }
      for (int k = 0; k < P.cfgo[n]; k++)
      {
         const int r = d->cfg_indices[k];
         if (r < 0 || r >= d->nq || seen_q[r])
            return fail(MH_ERR_BAD_TOPOLOGY, "cfg_indices[%d] = %d is out of range or repeated (nq = %d)", k, r, d->nq);
         seen_q[r] = 1;
      }
   }
   // engine order: depth-first pre-order, children in the caller's order (chains stay contiguous)
   P.children.assign(n, {});
   std::vector<int> roots;
   for (int i = 0; i < n; i++)
      (d->parent[i] < 0 ? roots : P.children[d->parent[i]]).push_back(i);
   P.order.clear();
   P.order.reserve(n);
   {
      std::vector<int> stack(roots.rbegin(), roots.rend());
      while (!stack.empty())
      {
         int i = stack.back();
         stack.pop_back();
         P.order.push_back(i);
         for (auto it = P.children[i].rbegin(); it != P.children[i].rend(); ++it)
            stack.push_back(*it);
      }
   }
   if ((int)P.order.size() != n)
      return fail(MH_ERR_LOOP_CLOSURE, "parent[] contains a cycle: %d of %d joints are reachable from the root", (int)P.order.size(), n);
   P.engine_of.assign(n, 0);
   for (int e = 0; e < n; e++)
      P.engine_of[P.order[e]] = e;
   P.eparent.assign(n, -1), P.etype.assign(n, 0);
   unsigned long long h = 1469598103934665603ull; // FNV-1a over (n, parents, kinds) in engine order
   auto mix = [&](int v) {
      for (int b = 0; b < 4; b++)
      {
         h ^= (unsigned long long)((v >> (8 * b)) & 0xff);
         h *= 1099511628211ull;
      }
   };
   mix(n);
   for (int e = 0; e < n; e++)
   {
      const int i = P.order[e];
      P.eparent[e] = d->parent[i] < 0 ? -1 : P.engine_of[d->parent[i]];
      P.etype[e] = d->joint_type[i];
      mix(P.eparent[e]);
      mix(P.etype[e]);
   }
   char buf[32];
   snprintf(buf, sizeof buf, "%016llx", h);
   P.key = buf;
   return MH_OK;
}

// Looks for libmecano_hip_topo_<key>.so next to this library and checks it was built for exactly this tree.
void try_load_spec(mh_model *m, const Plan &P)
{
   Dl_info info;
   if (!dladdr((const void *)&try_load_spec, &info) || !info.dli_fname)
      return;
   std::string dir(info.dli_fname);
   const size_t slash = dir.find_last_of('/');
   dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
   if (const char *e = getenv("MH_SPEC_DIR")) // experiment builds of the specialised code objects live elsewhere (tools/isa.py)
      dir = e;
   // the full code object, else a minimal one (mh_build_code_object's fast form: tree-split RNEA / ABA / pair for AoS + identity maps)
   std::string path = dir + "/libmecano_hip_topo_" + P.key + ".so";
   void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
   if (!h)
   {
      path = dir + "/libmecano_hip_topo_" + P.key + ".min.so";
      h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
   }
   if (!h)
      return;
   auto f_n = (int (*)(void))dlsym(h, "mh_spec_n");
   auto f_p = (const int *(*)(void))dlsym(h, "mh_spec_parents");
   auto f_t = (const int *(*)(void))dlsym(h, "mh_spec_types");
   SpecLib s;
   s.handle = h;
   s.launch = (decltype(s.launch))dlsym(h, "mh_spec_launch");
   s.lds_bytes = (decltype(s.lds_bytes))dlsym(h, "mh_spec_lds_bytes");
   s.aba_slots = (decltype(s.aba_slots))dlsym(h, "mh_spec_aba_slots");
   s.supports = (decltype(s.supports))dlsym(h, "mh_spec_supports");
   s.launch_fused = (decltype(s.launch_fused))dlsym(h, "mh_spec_launch_fused");
   s.fused_lds_bytes = (decltype(s.fused_lds_bytes))dlsym(h, "mh_spec_fused_lds_bytes");
   s.launch_crba = (decltype(s.launch_crba))dlsym(h, "mh_spec_launch_crba");
   s.crba_packed = (decltype(s.crba_packed))dlsym(h, "mh_spec_crba_packed");
   s.split_usable = (decltype(s.split_usable))dlsym(h, "mh_spec_split_usable");
   s.split_lds_bytes = (decltype(s.split_lds_bytes))dlsym(h, "mh_spec_split_lds_bytes");
   s.launch_split = (decltype(s.launch_split))dlsym(h, "mh_spec_launch_split");
   s.crba_split_usable = (decltype(s.crba_split_usable))dlsym(h, "mh_spec_crba_split_usable");
   s.launch_crba_split = (decltype(s.launch_crba_split))dlsym(h, "mh_spec_launch_crba_split");
   s.launch_rnea_crba = (decltype(s.launch_rnea_crba))dlsym(h, "mh_spec_launch_rnea_crba");
   s.zv_usable = (decltype(s.zv_usable))dlsym(h, "mh_spec_zv_usable");
   s.zv_lds_bytes = (decltype(s.zv_lds_bytes))dlsym(h, "mh_spec_zv_lds_bytes");
   s.launch_zv = (decltype(s.launch_zv))dlsym(h, "mh_spec_launch_zv");
   s.zv_self_signal = (decltype(s.zv_self_signal))dlsym(h, "mh_spec_zv_self_signal");
   s.zvb_usable = (decltype(s.zvb_usable))dlsym(h, "mh_spec_zvb_usable");
   s.zvb_cs_rows = (decltype(s.zvb_cs_rows))dlsym(h, "mh_spec_zvb_cs_rows");
   s.zvb_lds_bytes = (decltype(s.zvb_lds_bytes))dlsym(h, "mh_spec_zvb_lds_bytes");
   s.launch_zvb = (decltype(s.launch_zvb))dlsym(h, "mh_spec_launch_zvb");
   s.zvf_usable = (decltype(s.zvf_usable))dlsym(h, "mh_spec_zvf_usable");
   s.zvf_pair_usable = (decltype(s.zvf_pair_usable))dlsym(h, "mh_spec_zvf_pair_usable");
   s.launch_zvf = (decltype(s.launch_zvf))dlsym(h, "mh_spec_launch_zvf");
   s.launch_rnea_ahead = (decltype(s.launch_rnea_ahead))dlsym(h, "mh_spec_launch_rnea_ahead");
   s.rnea_crba_lds_bytes = (decltype(s.rnea_crba_lds_bytes))dlsym(h, "mh_spec_rnea_crba_lds_bytes");
   s.launch_coriolis = (decltype(s.launch_coriolis))dlsym(h, "mh_spec_launch_coriolis");
   s.launch_centroidal = (decltype(s.launch_centroidal))dlsym(h, "mh_spec_launch_centroidal");
   s.launch_coriolis_parts = (decltype(s.launch_coriolis_parts))dlsym(h, "mh_spec_launch_coriolis_parts");
   s.launch_centroidal_parts = (decltype(s.launch_centroidal_parts))dlsym(h, "mh_spec_launch_centroidal_parts");
   s.abi = (decltype(s.abi))dlsym(h, "mh_spec_abi");
   // the code object reinterprets the library's argument structs and folds parts of the canonical-frame convention at compile time:
   // it must have been built from the same headers (a stale or foreign libmecano_hip_topo_<key>.so is refused, visibly)
   if (!s.abi || s.abi() != mh::spec_abi_stamp())
   {
      char note[256];
      snprintf(note, sizeof note, "generic (code object %s refused: built against another library version, ABI stamp %016llx, this library %016llx)",
               path.c_str(), s.abi ? s.abi() : 0ull, mh::spec_abi_stamp());
      m->variant = note;
      dlclose(h);
      return;
   }
   // ... and from the same KERNEL sources and code-generation flags as the ones this library was built beside: a code object left over
   // from an experiment or an older tree computes something, passes its own self-check against nothing but itself, and is not HEAD's
   s.sources_hash = (decltype(s.sources_hash))dlsym(h, "mh_spec_sources_hash");
   if (!s.sources_hash || strcmp(s.sources_hash(), MH_STR_(MH_SPEC_SOURCES_HASH)) != 0)
   {
      char note[320];
      snprintf(note, sizeof note, "generic (code object %s refused: built from other kernel sources or flags, source hash %s, this library expects %s)",
               path.c_str(), s.sources_hash ? s.sources_hash() : "(none)", MH_STR_(MH_SPEC_SOURCES_HASH));
      m->variant = note;
      dlclose(h);
      return;
   }
   bool ok = f_n && f_p && f_t && s.launch && s.lds_bytes && s.aba_slots && s.supports && f_n() == m->n;
   for (int e = 0; ok && e < m->n; e++)
      ok = f_p()[e] == P.eparent[e] && f_t()[e] == P.etype[e];
   if (!ok)
   {
      m->variant = "generic (code object " + path + " refused: it was built for another tree)";
      dlclose(h);
      return;
   }
   m->spec = s;
   m->variant = "topo:" + P.key;
   auto f_min = (int (*)(void))dlsym(h, "mh_spec_minimal");
   m->spec_minimal = f_min && f_min() != 0;
   if (m->spec_minimal)
      m->variant += " (minimal build: tree-split RNEA / ABA / pair kernels only, every other plan on the run-time-topology kernels)";
}
// Mass matrix + Coriolis matrix (CompositeRigidBodyMassMatrixCalculator with the Coriolis calculation enabled): run-time-topology kernel
template <typename T>
mh_status coriolis_impl(mh_model_t model, int64_t B, const T *q, const T *qd, const mh_options *opts_in, T *H_out, T *C_out)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (B == 0)
      return MH_OK;
   if (!q || !qd || !H_out || !C_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer");
   const Launch L = plan_launch(model, B);
   const int parts = regressor_parts(model, L);
   st = ensure_parts_workspace(model, L, parts, sizeof(T));
   if (st != MH_OK)
      return st;
   hipStream_t stream = (hipStream_t)opts.stream;
   mh::Args<T> A{};
   A.m = dev_model<T>(model);
   A.B = B;
   A.q = q, A.qd = qd, A.out = H_out, A.outb = C_out;
   A.ws = (T *)model->ws.ptr;
   A.ws_stride = L.lanes;
   const bool soa = opts.layout == MH_LAYOUT_SOA;
   A.q_bs = soa ? 1 : model->nq, A.q_es = soa ? B : 1;
   A.v_bs = soa ? 1 : model->nv, A.v_es = soa ? B : 1;
   A.f_bs = soa ? 1 : (long)model->nv * model->nv, A.f_es = soa ? B : 1; // strides of H and C
   const bool ldsc = MH_GENERIC_LDS_CONSTS || model->lds_consts;
   const size_t lds = ldsc ? (size_t)model->n * mh::MC_STRIDE * sizeof(T) : 0;
   const size_t hbytes = (size_t)B * model->nv * model->nv * sizeof(T);
   HIP_TRY(hipMemsetAsync(H_out, 0, hbytes, stream)); // the kernel writes the entries of related joints only (:298-300)
   HIP_TRY(hipMemsetAsync(C_out, 0, hbytes, stream));
   if constexpr (sizeof(T) == 8)
   {
      if (model->spec.launch_coriolis && model->use_spec)
      { // topology-specialised recursion: ancestors' transforms and velocities in registers, no workspace
         const long waves = (B + 63) / 64;
         const int grid = (int)std::max<long>(1, std::min(waves, (long)model->cu_count * 4));
         Launch G = L;
         G.grid = grid; // small batches: several waves per group of configurations, each writing every parts-th body's columns
         const int rc = model->spec.launch_coriolis_parts
                           ? model->spec.launch_coriolis_parts(model->ident_maps ? SPEC_IDENT : 0, &A, grid, regressor_parts(model, G), (void *)stream)
                           : model->spec.launch_coriolis(model->ident_maps ? SPEC_IDENT : 0, &A, grid, (void *)stream);
         if (rc == 0)
            return MH_OK;
         if (rc != (int)hipErrorNotSupported)
            return fail(MH_ERR_HIP, "specialised Coriolis kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
      }
   }
   { if (ldsc) hipLaunchKernelGGL((mh::coriolis_kernel<T, true>), dim3(L.grid, parts), dim3(L.block), lds, stream, A); else hipLaunchKernelGGL((mh::coriolis_kernel<T, false>), dim3(L.grid, parts), dim3(L.block), lds, stream, A); }
   HIP_TRY(hipGetLastError());
   return MH_OK;
}
// Joint torque regressor (JointTorqueRegressorCalculator): run-time-topology kernel
template <typename T>
mh_status regressor_impl(mh_model_t model, int64_t B, const T *q, const T *qd, const T *qdd, const double *gravity, const mh_options *opts_in,
                         int32_t first_moment_columns, T *Y_out)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (B == 0)
      return MH_OK;
   if (!q || !qd || !qdd || !Y_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer");
   const Launch L = plan_launch(model, B);
   const int parts = regressor_parts(model, L);
   st = ensure_parts_workspace(model, L, parts, sizeof(T));
   if (st != MH_OK)
      return st;
   hipStream_t stream = (hipStream_t)opts.stream;
   mh::Args<T> A{};
   A.m = dev_model<T>(model);
   A.B = B;
   A.q = q, A.qd = qd, A.in3 = qdd, A.out = Y_out;
   A.ws = (T *)model->ws.ptr;
   A.ws_stride = L.lanes;
   const bool soa = opts.layout == MH_LAYOUT_SOA;
   A.q_bs = soa ? 1 : model->nq, A.q_es = soa ? B : 1;
   A.v_bs = soa ? 1 : model->nv, A.v_es = soa ? B : 1;
   const long ysize = (long)model->nv * model->n * 10;
   A.f_bs = soa ? 1 : ysize, A.f_es = soa ? B : 1; // strides of Y
   set_root_acceleration(A, opts, gravity);
   A.coriolis = opts.consider_coriolis, A.accel = opts.consider_accelerations;
   const bool ldsc = MH_GENERIC_LDS_CONSTS || model->lds_consts;
   const size_t lds = ldsc ? (size_t)model->n * mh::MC_STRIDE * sizeof(T) : 0;
   // entries of joints that do not support a body are zero, and so are the reference's centre-of-mass columns (mh_kernels.h)
   HIP_TRY(hipMemsetAsync(Y_out, 0, (size_t)B * ysize * sizeof(T), stream));
   // the centre-of-mass columns: d tau / d (m c) on request; else the reference's -- zero, or e x a once the twist is switched off
   const int mode = first_moment_columns ? 1 : (opts.consider_coriolis ? 0 : 2);
#define MH_REG_LAUNCH(MODE) \
   { if (ldsc) hipLaunchKernelGGL((mh::regressor_kernel<T, true, MODE>), dim3(L.grid, parts), dim3(L.block), lds, stream, A); else hipLaunchKernelGGL((mh::regressor_kernel<T, false, MODE>), dim3(L.grid, parts), dim3(L.block), lds, stream, A); }
   if (mode == 0)
      MH_REG_LAUNCH(0)
   else if (mode == 1)
      MH_REG_LAUNCH(1)
   else
      MH_REG_LAUNCH(2)
#undef MH_REG_LAUNCH
   HIP_TRY(hipGetLastError());
   return MH_OK;
}
template <typename T>
mh_status centroidal_impl(mh_model_t model, int64_t B, const T *q, const T *qd, const double *frame, int32_t frame_mode, const mh_options *opts_in,
                          T *A_out, T *b_out, T *com_out)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (frame_mode != MH_CENTROIDAL_FRAME_FIXED && frame_mode != MH_CENTROIDAL_FRAME_AT_COM)
      return fail(MH_ERR_INVALID_ARGUMENT, "unknown centroidal frame mode %d", frame_mode);
   if (B == 0)
      return MH_OK;
   if (!q || !A_out || (b_out && !qd))
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer (the convective term needs qd)");
   const Launch L = plan_launch(model, B);
   const int parts = regressor_parts(model, L);
   st = ensure_parts_workspace(model, L, parts, sizeof(T));
   if (st != MH_OK)
      return st;
   hipStream_t stream = (hipStream_t)opts.stream;
   mh::CentArgs<T> A{};
   A.m = dev_model<T>(model);
   A.B = B;
   A.q = q, A.qd = qd, A.A = A_out, A.b = b_out, A.com = com_out;
   A.ws = (T *)model->ws.ptr;
   A.ws_stride = L.lanes;
   const bool soa = opts.layout == MH_LAYOUT_SOA;
   A.q_bs = soa ? 1 : model->nq, A.q_es = soa ? B : 1;
   A.v_bs = soa ? 1 : model->nv, A.v_es = soa ? B : 1;
   A.a_bs = soa ? 1 : 6L * model->nv, A.a_es = soa ? B : 1;
   A.b_bs = soa ? 1 : 6, A.b_es = soa ? B : 1;
   A.c_bs = soa ? 1 : 3, A.c_es = soa ? B : 1;
   for (int k = 0; k < 9; k++)
      A.fR[k] = frame ? (T)frame[k] : (T)(k % 4 == 0 ? 1 : 0);
   for (int k = 0; k < 3; k++)
      A.fp[k] = frame ? (T)frame[9 + k] : T(0);
   A.at_com = frame_mode == MH_CENTROIDAL_FRAME_AT_COM;
   const bool ldsc = MH_GENERIC_LDS_CONSTS || model->lds_consts;
   const size_t lds = ldsc ? (size_t)model->n * mh::MC_STRIDE * sizeof(T) : 0;
   HIP_TRY(hipMemsetAsync(A_out, 0, (size_t)B * 6 * model->nv * sizeof(T), stream)); // columns no considered joint owns stay zero
   if constexpr (sizeof(T) == 8)
   {
      if (model->spec.launch_centroidal && model->use_spec)
      {
         const long waves = (B + 63) / 64;
         const int grid = (int)std::max<long>(1, std::min(waves, (long)model->cu_count * 4));
         Launch G = L;
         G.grid = grid;
         const int rc = model->spec.launch_centroidal_parts
                           ? model->spec.launch_centroidal_parts(model->ident_maps ? SPEC_IDENT : 0, &A, grid, regressor_parts(model, G), (void *)stream)
                           : model->spec.launch_centroidal(model->ident_maps ? SPEC_IDENT : 0, &A, grid, (void *)stream);
         if (rc == 0)
            return MH_OK;
         if (rc != (int)hipErrorNotSupported)
            return fail(MH_ERR_HIP, "specialised centroidal kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
      }
   }
   { if (ldsc) hipLaunchKernelGGL((mh::centroidal_kernel<T, true>), dim3(L.grid, parts), dim3(L.block), lds, stream, A); else hipLaunchKernelGGL((mh::centroidal_kernel<T, false>), dim3(L.grid, parts), dim3(L.block), lds, stream, A); }
   HIP_TRY(hipGetLastError());
   return MH_OK;
}
template <typename T>
mh_status integrate_impl(mh_model_t model, int64_t B, double dt, const T *q, const T *qd, const T *qdd, const mh_options *opts_in, T *q_out,
                                T *qd_out, T *qdd_out)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (B == 0)
      return MH_OK;
   if (!q || !qd || !qdd || !q_out || !qd_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer");
   if (nan_bits(dt))
      return fail(MH_ERR_INVALID_ARGUMENT, "dt is NaN");
   mh::IntArgs<T> A;
   A.m = dev_model<T>(model);
   A.B = B, A.dt = (T)dt;
   A.q = q, A.qd = qd, A.qdd = qdd, A.q_out = q_out, A.qd_out = qd_out, A.qdd_out = qdd_out;
   const bool soa = opts.layout == MH_LAYOUT_SOA;
   A.q_bs = soa ? 1 : model->nq, A.q_es = soa ? B : 1;
   A.v_bs = soa ? 1 : model->nv, A.v_es = soa ? B : 1;
   const int block = 256;
   if (soa)
   {
      const int grid = (int)std::max<long>(1, std::min<long>((B + block - 1) / block, (long)model->cu_count * 8));
      hipLaunchKernelGGL((mh::integrate_soa_kernel<T>), dim3(grid), dim3(block), 0, (hipStream_t)opts.stream, A);
   }
   else
   {
      // tile: enough workgroups to cover the device at small B, up to 256 configurations each at large B
      const int tile = (int)std::max<long>(16, std::min<long>(256, B / ((long)model->cu_count * 4)));
      const int grid = (int)std::max<long>(1, std::min<long>((B + tile - 1) / tile, (long)model->cu_count * 8));
      hipLaunchKernelGGL((mh::integrate_aos_kernel<T>), dim3(grid), dim3(block), ((size_t)model->n * 3 + 2) * sizeof(int), (hipStream_t)opts.stream, A, tile);
   }
   HIP_TRY(hipGetLastError());
   return MH_OK;
}
} // namespace

static void self_check_spec(mh_model *m);
static mh_status build_code_object(const mh_model_desc *desc, const char *out_dir, char *path_out, size_t path_cap, bool fast);

// =================================================================================================== C-ABI
extern "C" {

int32_t mh_abi_version(void) { return MH_ABI_VERSION; }
uint64_t mh_spec_abi_stamp(void) { return mh::spec_abi_stamp(); }
const char *mh_build_hash(void) { return MH_STR_(MH_BUILD_HASH); }
const char *mh_spec_sources_hash(void) { return MH_STR_(MH_SPEC_SOURCES_HASH); }
mh_status mh_spec_sources_hash_of(const char *csrc_dir, char out[18])
{
   if (!csrc_dir || !out)
      return fail(MH_ERR_INVALID_ARGUMENT, "csrc_dir / out is NULL");
   if (!hash_spec_sources(csrc_dir, out))
      return fail(MH_ERR_INVALID_ARGUMENT, "%s does not hold the kernel sources (mh_spec.hip and its headers)", csrc_dir);
   return MH_OK;
}
const char *mh_last_error(void) { return g_err; }
// the library's other translation units (mh_comm.hip) report through the same thread-local message
mh_status mh_internal_fail(mh_status code, const char *message) { return fail(code, "%s", message); }

mh_status mh_device_count(int32_t *count)
{
   if (!count)
      return fail(MH_ERR_INVALID_ARGUMENT, "count is NULL");
   int c = 0;
   if (hipGetDeviceCount(&c) != hipSuccess)
      c = 0;
   *count = c;
   return MH_OK;
}
mh_status mh_set_device(int32_t device)
{
   HIP_TRY(hipSetDevice(device));
   return MH_OK;
}
void mh_options_default(mh_options *opts)
{
   if (!opts)
      return;
   opts->consider_coriolis = 1;
   opts->consider_accelerations = 1;
   opts->layout = MH_LAYOUT_AOS;
   opts->use_root_acceleration = 0;
   opts->stream = nullptr;
   for (int k = 0; k < 6; k++)
      opts->root_acceleration[k] = 0.0;
   opts->context = nullptr;
}

mh_status mh_model_create(const mh_model_desc *d, mh_model_t *model_out)
{
   if (!d || !model_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "desc / model_out is NULL");
   *model_out = nullptr;
   Plan P;
   mh_status pst = plan_model(d, P);
   if (pst != MH_OK)
      return pst;
   const int n = d->n_joints;
   const std::vector<int> &order = P.order, &engine_of = P.engine_of, &dofo = P.dofo, &cfgo = P.cfgo;
   const std::vector<std::vector<int>> &children = P.children;

   // ---- canonical frames: Q_i maps the canonical after-joint axes of joint i to Mecano's after-joint axes
   std::vector<M3d> Q(n);
   for (int e = 0; e < n; e++)
   {
      const int i = order[e];
      const int t = d->joint_type[i];
      if (t == MH_JOINT_REVOLUTE || t == MH_JOINT_PRISMATIC)
      {
         const double *a = d->axis + 3 * i;
         const double nrm = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
         if (!(std::fabs(nrm - 1.0) <= 1.0e-6))
            return fail(MH_ERR_BAD_AXIS, "joint %d: axis (%g, %g, %g) is not a unit vector", i, a[0], a[1], a[2]);
         const double k[3] = {a[0] / nrm, a[1] / nrm, a[2] / nrm};
         Q[e] = frame_with_z(k);
      }
      else
         Q[e] = m3_identity();
   }

   // The frame after a 1-DoF joint may still turn about and slide along its own axis (both commute with the joint's motion).  That freedom
   // is spent on the joint's FIRST child (engine order: the next joint): origin and x axis are chosen so that the child's origin lies on the
   // x axis, p_b(child) = (a, 0, 0) -- two of the three translation components of that child's pose are structural zeros, which the
   // specialised kernels fold at compile time (Tree<TP>::p_aligned; the run-time-topology kernels just multiply by 0.0).  O[e] = origin of
   // the canonical frame of joint e in Mecano's after-joint frame (on the axis); children first, because a child's own slide moves its origin.
   std::vector<std::array<double, 3>> O(n, std::array<double, 3>{0.0, 0.0, 0.0});
   std::vector<char> aligned(n, 0);
   for (int e = n - 2; e >= 0; e--)
   {
      const int i = order[e], t = d->joint_type[i], ic = order[e + 1];
      if ((t != MH_JOINT_REVOLUTE && t != MH_JOINT_PRISMATIC) || d->parent[ic] != i)
         continue;
      M3d Rbc;
      std::memcpy(Rbc.m, d->X_before + 12 * ic, sizeof Rbc.m);
      double w0[3], w[3];
      m3_mulv(Rbc, O[e + 1].data(), w0);
      for (int k = 0; k < 3; k++)
         w0[k] += d->X_before[12 * ic + 9 + k];
      m3_mulv(m3_T(Q[e]), w0, w);
      const double delta = w[2], rho = std::hypot(w[0], w[1]);
      const double cphi = rho > 1.0e-12 ? w[0] / rho : 1.0, sphi = rho > 1.0e-12 ? w[1] / rho : 0.0;
      const double slide[3] = {0.0, 0.0, delta};
      m3_mulv(Q[e], slide, O[e].data());
      M3d Rz = m3_identity();
      Rz.m[0] = cphi, Rz.m[1] = -sphi, Rz.m[3] = sphi, Rz.m[4] = cphi;
      Q[e] = m3_mul(Q[e], Rz);
      aligned[e + 1] = 1;
   }

   mh_model *m = new mh_model();
   m->n = n, m->nq = d->nq, m->nv = d->nv;
   m->engine_of = engine_of;
   // ---- the two places where this engine consciously departs from the reference: told to the caller, not hidden
   {
      char buf[512];
      // (1) tools/MecanoFactories.java:51, 237-248: a revolute axis that geometricallyEquals X, Y or Z within 1e-7 WITHOUT being that axis
      // gets a joint rotation about the exact coordinate axis while the unit twist keeps the axis as given; the engine uses the given axis for both
      for (int i = 0; i < n; i++)
      {
         if (d->joint_type[i] != MH_JOINT_REVOLUTE)
            continue;
         const double *a = d->axis + 3 * i;
         for (int k = 0; k < 3; k++)
         {
            const double dx = a[0] - (k == 0), dy = a[1] - (k == 1), dz = a[2] - (k == 2);
            const double dist = std::sqrt(dx * dx + dy * dy + dz * dz);
            if (dist <= 1.0e-7 && dist > 0.0)
            {
               if (!(m->warnings & MH_WARN_NEAR_COORDINATE_AXIS))
               {
                  snprintf(buf, sizeof buf,
                           "joint %d: axis (%.17g, %.17g, %.17g) is within 1e-7 of the %c axis but not on it: Mecano rotates such a joint about the exact "
                           "coordinate axis and keeps the given axis in its unit twist (MecanoFactories.java:237-248); this engine uses the given axis for "
                           "both, results differ from Mecano's by up to ~4e-7 relative. ",
                           i, a[0], a[1], a[2], "XYZ"[k]);
                  m->warning_text += buf;
               }
               m->warnings |= MH_WARN_NEAR_COORDINATE_AXIS;
               break;
            }
         }
      }
      // (2) spatial/interfaces/FixedFrameSpatialInertiaBasics.java:167-176: SpatialInertia.add skips the renormalisation of the centre of
      // mass when the summed mass is under 1e-7; the mass matrix of a body whose composite with a child's subtree stays under it differs
      std::vector<double> sub(n, 0.0);
      for (int e = n - 1; e >= 0; e--)
      {
         const int i = order[e];
         sub[i] += d->inertia_mass[i];
         if (d->parent[i] >= 0)
            sub[d->parent[i]] += sub[i];
      }
      for (int i = 0; i < n; i++)
         for (int ch : children[i])
            if (std::fabs(d->inertia_mass[i] + sub[ch]) < 1.0e-7)
            {
               if (!(m->warnings & MH_WARN_TINY_COMPOSITE_MASS))
               {
                  snprintf(buf, sizeof buf,
                           "joint %d: the body's mass plus the subtree of joint %d is %.3g < 1e-7: Mecano's SpatialInertia.add leaves such a composite's "
                           "centre of mass un-normalised (FixedFrameSpatialInertiaBasics.java:174-175); this engine's mass matrix stays consistent "
                           "with its inverse dynamics and differs from Mecano's by less than the masses involved (<= 1e-6). ",
                           i, ch, d->inertia_mass[i] + sub[ch]);
                  m->warning_text += buf;
               }
               m->warnings |= MH_WARN_TINY_COMPOSITE_MASS;
            }
   }
   m->meta.assign((size_t)n * mh::MI_STRIDE, 0);
   m->consts.assign((size_t)n * mh::MC_STRIDE, 0.0);
   // index maps re-concatenated in ENGINE order: the offset of a joint in them is then a function of the topology alone
   std::vector<int> edofo(n + 1, 0), ecfgo(n + 1, 0);
   for (int e = 0; e < n; e++)
   {
      const int i = order[e];
      edofo[e + 1] = edofo[e] + (dofo[i + 1] - dofo[i]);
      ecfgo[e + 1] = ecfgo[e] + (cfgo[i + 1] - cfgo[i]);
      for (int k = dofo[i]; k < dofo[i + 1]; k++)
         m->dof_map.push_back(d->dof_indices[k]);
      for (int k = cfgo[i]; k < cfgo[i + 1]; k++)
         m->cfg_map.push_back(d->cfg_indices[k]);
   }
   m->dense_maps = (edofo[n] == d->nv && ecfgo[n] == d->nq);
   m->ident_maps = m->dense_maps;
   for (int k = 0; m->ident_maps && k < edofo[n]; k++)
      m->ident_maps = m->dof_map[k] == k;
   for (int k = 0; m->ident_maps && k < ecfgo[n]; k++)
      m->ident_maps = m->cfg_map[k] == k;
   if (m->dof_map.empty())
      m->dof_map.push_back(0);
   if (m->cfg_map.empty())
      m->cfg_map.push_back(0);

   int slots = 0;
   for (int e = 0; e < n; e++)
   {
      const int i = order[e];
      const int t = d->joint_type[i];
      const int pe = d->parent[i] < 0 ? -1 : engine_of[d->parent[i]];
      int *mi = &m->meta[(size_t)e * mh::MI_STRIDE];
      double *c = &m->consts[(size_t)e * mh::MC_STRIDE];
      mi[mh::MI_PARENT] = pe;
      mi[mh::MI_TYPE] = t;
      mi[mh::MI_DOF] = edofo[e];
      mi[mh::MI_CFG] = ecfgo[e];
      mi[mh::MI_EXT] = i;
      int flags = 0;
      if (pe >= 0 && pe == e - 1)
         flags |= mh::MF_PARENT_ADJ;
      else if (pe >= 0)
         m->n_nonadjacent++;
      bool nonadj_child = false;
      for (int ch : children[i])
         if (engine_of[ch] != e + 1)
            nonadj_child = true;
      if (nonadj_child)
         flags |= mh::MF_STORE_VA | mh::MF_HAS_ACC;
      if (pe >= 0 && pe != e - 1)
      {
         // first contributor = highest engine index among the non-adjacent children of the parent
         int hi = -1;
         for (int ch : children[d->parent[i]])
            if (engine_of[ch] != pe + 1)
               hi = std::max(hi, engine_of[ch]);
         if (hi == e)
            flags |= mh::MF_ACC_FIRST;
      }
      mi[mh::MI_FLAGS] = flags;
      mi[mh::MI_SLOT_JP] = slots, slots += (t == MH_JOINT_REVOLUTE ? 2 : 0);
      mi[mh::MI_SLOT_F] = slots, slots += 8;
      mi[mh::MI_SLOT_C] = slots, slots += 6;
      mi[mh::MI_SLOT_VA] = slots, slots += (nonadj_child ? 12 : 0);
      mi[mh::MI_SLOT_IA] = slots, slots += (nonadj_child ? 40 : 0); // ABA: 21 | CRBA: 10 | Coriolis: 10 + 30
      mi[mh::MI_SLOT_LK] = slots, slots += (mh::dof_count(t) >= 3 ? 27 : 0); // multi-DoF joints: U, D^-1, u | locked: IA, pA

      // X_before' = Qp^T X_before Q : canonical before-joint frame in the parent's canonical after-joint frame
      const M3d Qp = pe < 0 ? m3_identity() : Q[pe];
      M3d Rb;
      std::memcpy(Rb.m, d->X_before + 12 * i, sizeof Rb.m);
      const M3d Rb2 = m3_mul(m3_mul(m3_T(Qp), Rb), Q[e]);
      double pb2[3], pb0[3];
      m3_mulv(Rb, O[e].data(), pb0); // the canonical origin of this joint, then relative to the parent's canonical origin
      for (int k = 0; k < 3; k++)
         pb0[k] += d->X_before[12 * i + 9 + k] - (pe < 0 ? 0.0 : O[pe][k]);
      m3_mulv(m3_T(Qp), pb0, pb2);
      if (aligned[e])
         pb2[1] = 0.0, pb2[2] = 0.0; // (a, 0, 0) by construction: what is left is rounding
      for (int k = 0; k < 9; k++)
         c[mh::MC_RB + k] = Rb2.m[k];
      for (int k = 0; k < 3; k++)
         c[mh::MC_PB + k] = pb2[k];
      // body-fixed -> canonical after-joint: R' = Q^T Rc, p' = Q^T pc
      M3d Rc;
      std::memcpy(Rc.m, d->X_com + 12 * i, sizeof Rc.m);
      const M3d Rf = m3_mul(m3_T(Q[e]), Rc);
      double pf[3], pf0[3];
      for (int k = 0; k < 3; k++)
         pf0[k] = d->X_com[12 * i + 9 + k] - O[e][k];
      m3_mulv(m3_T(Q[e]), pf0, pf);
      for (int k = 0; k < 9; k++)
         c[mh::MC_RF + k] = Rf.m[k];
      for (int k = 0; k < 3; k++)
         c[mh::MC_PF + k] = pf[k];
      // canonical after-joint -> Mecano's after-joint frame: x = Q x' + O (joint wrench outputs)
      for (int k = 0; k < 9; k++)
         c[mh::MC_QA + k] = Q[e].m[k];
      for (int k = 0; k < 3; k++)
         c[mh::MC_OA + k] = O[e][k];
      // spatial inertia about the canonical after-joint origin.  J is the rotational inertia about the ORIGIN of the
      // body-fixed frame with the CoM at c_b there (spatial/interfaces/SpatialInertiaReadOnly.java:394-415).
      const double mass = d->inertia_mass[i];
      const double *cb = d->inertia_com + 3 * i;
      M3d J;
      std::memcpy(J.m, d->inertia_J + 9 * i, sizeof J.m);
      const M3d Jr = m3_mul(m3_mul(Rf, J), m3_T(Rf)); // about the body-fixed origin, canonical axes
      double cr[3];
      m3_mulv(Rf, cb, cr); // CoM relative to the body-fixed origin, canonical axes
      // shift the origin from the body-fixed origin (at pf) to the after-joint origin: c' = cr + pf
      const double h0[3] = {mass * cr[0], mass * cr[1], mass * cr[2]};
      const double dd = 2.0 * (pf[0] * h0[0] + pf[1] * h0[1] + pf[2] * h0[2]) + mass * (pf[0] * pf[0] + pf[1] * pf[1] + pf[2] * pf[2]);
      double I[9];
      for (int r = 0; r < 3; r++)
         for (int s = 0; s < 3; s++)
            I[3 * r + s] = Jr.m[3 * r + s] + (r == s ? dd : 0.0) - (pf[r] * h0[s] + h0[r] * pf[s] + mass * pf[r] * pf[s]);
      c[mh::MC_M] = mass;
      for (int k = 0; k < 3; k++)
         c[mh::MC_H + k] = h0[k] + mass * pf[k];
      c[mh::MC_I + 0] = I[0], c[mh::MC_I + 1] = 0.5 * (I[1] + I[3]), c[mh::MC_I + 2] = 0.5 * (I[2] + I[6]);
      c[mh::MC_I + 3] = I[4], c[mh::MC_I + 4] = 0.5 * (I[5] + I[7]), c[mh::MC_I + 5] = I[8];
   }
   m->n_slots = std::max(slots, 1);

   // ---- depth-first kernels: children counts, stack-frame / hand-over offsets, event program (mh_dfs_kernels.h)
   {
      std::vector<int> nch(n, 0), ofs_r(n, 0), ofs_a(n, 0), ofs_p(n, 0);
      for (int e = 0; e < n; e++)
         if (P.eparent[e] >= 0)
            nch[P.eparent[e]]++;
      int hand = 0;
      for (int e = 0; e < n; e++)
      {
         const int pe = P.eparent[e], t = P.etype[e];
         ofs_r[e] = pe < 0 ? 0 : ofs_r[pe] + mh::rnea_frame_slots(P.etype[pe], nch[pe]);
         ofs_a[e] = pe < 0 ? 0 : ofs_a[pe] + mh::aba_frame_slots(P.etype[pe], nch[pe]);
         m->rnea_stack = std::max(m->rnea_stack, ofs_r[e] + mh::rnea_frame_slots(t, nch[e]));
         m->aba_stack = std::max(m->aba_stack, ofs_a[e] + mh::aba_frame_slots(t, nch[e]));
         ofs_p[e] = pe < 0 ? 0 : ofs_p[pe] + mh::pair_frame_slots(P.etype[pe], nch[pe]);
         m->pair_stack = std::max(m->pair_stack, ofs_p[e] + mh::pair_frame_slots(t, nch[e]));
         int *mi = &m->meta[(size_t)e * mh::MI_STRIDE];
         mi[mh::MI_NCH] = nch[e], mi[mh::MI_DFS_R] = ofs_r[e], mi[mh::MI_DFS_A] = ofs_a[e], mi[mh::MI_HAND] = hand;
         hand += mh::aba_hand_slots(t, nch[e]);
         if (pe >= 0)
         {
            const int pj = mh::jx_slots(P.etype[pe]);
            mi[mh::MI_PFR_R] = ofs_r[pe], mi[mh::MI_PVA_R] = ofs_r[pe] + 6 + pj;
            mi[mh::MI_PFR_A] = ofs_a[pe], mi[mh::MI_PV_A] = ofs_a[pe] + 12 + pj, mi[mh::MI_PACC_A] = ofs_a[pe] + 18 + pj;
         }
         if (t == MH_JOINT_REVOLUTE || t == MH_JOINT_PRISMATIC)
            mi[mh::MI_ROW_Q] = m->cfg_map[mi[mh::MI_CFG]], mi[mh::MI_ROW_V] = m->dof_map[mi[mh::MI_DOF]];
      }
      m->aba_hand = std::max(hand, 1);
      m->nonleaf_fraction = (double)std::count_if(nch.begin(), nch.end(), [](int c) { return c > 0; }) / (double)n;
      m->rnea_stack = std::max(m->rnea_stack, 1), m->aba_stack = std::max(m->aba_stack, 6);
      // The walk: depth-first, the children of a body in the order [those with children of their own | the leaves].  A child's contribution
      // to its parent (wrench; articulated inertia + bias wrench) is either accumulated in the parent's frame (read-modify-write of 6 / 27 /
      // 33 slots) or handed over in registers, the carry.  The carry survives a LEAF sibling's two events (they never touch it), so the
      // last child with children of its own sets it and every leaf behind it adds to it: only the other children with subtrees go through
      // the frame (128-body tree of configs[4]: 26 of 127 child pops, before the leaves were sorted behind: 64).
      // (The kernels that read AoS rows through LDS windows consume the matrices in engine order and refill synchronously on a jump: they
      // keep a program in engine order -- prog_seq --, with the same carry rule applied to whatever leaves happen to come last.)
      auto build_program = [&](bool leaves_last, std::vector<int> &prog) {
         std::vector<std::vector<int>> kids(n);
         std::vector<int> roots;
         for (int e = 0; e < n; e++)
            (P.eparent[e] >= 0 ? kids[P.eparent[e]] : roots).push_back(e);
         if (leaves_last)
            for (int e = 0; e < n; e++)
               std::stable_partition(kids[e].begin(), kids[e].end(), [&](int c) { return nch[c] > 0; });
         std::vector<size_t> pop_at(n, 0);
         auto visit = [&](int e) {
            int ev = e << mh::EV_BODY_SHIFT;
            if (!prog.empty() && P.eparent[e] >= 0 && !(prog.back() & mh::EV_POP) && (prog.back() >> mh::EV_BODY_SHIFT) == P.eparent[e])
               ev |= mh::EV_PARENT_REGS;
            prog.push_back(ev);
         };
         auto pop = [&](int e) {
            int pv = (e << mh::EV_BODY_SHIFT) | mh::EV_POP;
            if (prog.back() == (e << mh::EV_BODY_SHIFT) + (prog.back() & mh::EV_PARENT_REGS))
               pv |= mh::EV_LEAF; // the previous event is VISIT(e)
            pop_at[e] = prog.size();
            prog.push_back(pv);
         };
         std::vector<std::pair<int, size_t>> path; // (body, next child to walk): an explicit stack -- a chain of 100 000 bodies is a model too
         for (int r : roots)
         {
            visit(r);
            path.emplace_back(r, 0);
            while (!path.empty())
            {
               const int e = path.back().first;
               if (path.back().second < kids[e].size())
               {
                  const int c = kids[e][path.back().second++];
                  visit(c);
                  path.emplace_back(c, 0);
               }
               else
               {
                  pop(e);
                  path.pop_back();
               }
            }
         }
         for (int e = 0; e < n; e++)
         {
            const std::vector<int> &k = kids[e];
            if (k.empty())
               continue;
            int first_carried = 0; // the last child with children of its own (only leaves behind it), or the first child
            for (size_t i = 0; i < k.size(); i++)
               if (nch[k[i]] > 0)
                  first_carried = (int)i;
            for (size_t i = 0; i < k.size(); i++)
            {
               int &ev = prog[pop_at[k[i]]];
               if ((int)i < first_carried)
                  ev |= i == 0 ? mh::EV_ACC_FIRST : 0;
               else
                  ev |= (int)i == first_carried ? mh::EV_LAST_CHILD : mh::EV_CARRY_ADD;
            }
            if (first_carried > 0)
               prog[pop_at[e]] |= mh::EV_ACC_USED;
         }
      };
      build_program(!getenv("MH_DFS_ENGINE_ORDER"), m->prog); // (MH_DFS_ENGINE_ORDER=1: the siblings in engine order everywhere, for measurements)
      build_program(false, m->prog_seq);
   }

   // ---- device side
   int dev = 0, ndev = 0;
   const hipError_t dc = hipGetDeviceCount(&ndev);
   if (dc != hipSuccess || ndev == 0)
   {
      delete m;
      return fail(MH_ERR_NO_DEVICE, "no HIP device (%s, %d device(s)): the model cannot be uploaded (there is no CPU path)", hipGetErrorString(dc), ndev);
   }
   hipError_t e = hipGetDevice(&dev);
   if (e == hipSuccess)
   {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess)
         m->cu_count = prop.multiProcessorCount;
   }
   m->device = dev;
   m->topo_key = P.key;
   std::vector<float> c32(m->consts.begin(), m->consts.end());
   auto up = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
      hipError_t r = hipMalloc(dst, bytes);
      if (r != hipSuccess)
         return r;
      return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
   };
   if (e == hipSuccess)
      e = up((void **)&m->d_meta, m->meta.data(), m->meta.size() * sizeof(int));
   if (e == hipSuccess)
      e = up((void **)&m->d_dof, m->dof_map.data(), m->dof_map.size() * sizeof(int));
   if (e == hipSuccess)
      e = up((void **)&m->d_cfg, m->cfg_map.data(), m->cfg_map.size() * sizeof(int));
   if (e == hipSuccess)
      e = up((void **)&m->d_prog, m->prog.data(), m->prog.size() * sizeof(int));
   if (e == hipSuccess)
      e = up((void **)&m->d_prog_seq, m->prog_seq.data(), m->prog_seq.size() * sizeof(int));
   if (e == hipSuccess)
      e = up((void **)&m->d_consts64, m->consts.data(), m->consts.size() * sizeof(double));
   if (e == hipSuccess)
      e = up((void **)&m->d_consts32, c32.data(), c32.size() * sizeof(float));
   if (e != hipSuccess)
   {
      mh_model_destroy(m);
      return fail(MH_ERR_HIP, "model upload failed: %s", hipGetErrorString(e));
   }
   if (const char *e = getenv("MH_DISABLE_SPEC"))
      m->use_spec = atoi(e) ? 0 : 1;
   if (const char *e = getenv("MH_SPEC_SPLIT"))
      m->use_split = atoi(e);
   if (const char *e = getenv("MH_DISABLE_FUSED"))
      m->use_fused = atoi(e) ? 0 : 1;
   if (const char *e = getenv("MH_ZV"))
      m->use_zv = atoi(e);
   if (const char *e = getenv("MH_ZV_SAME_L2"))
      m->zv_same_l2 = atoi(e) ? 1 : 0;
   if (const char *e = getenv("MH_ZV_WAIT_MS"))
      m->zv_wait_ticks = (unsigned)std::max<long long>(1, std::min<long long>(40000, atoll(e))) * 100000u;
   if (const char *e = getenv("MH_ZVB"))
      m->use_zvb = atoi(e);
   if (const char *e = getenv("MH_ZVF"))
      m->use_zvf = atoi(e);
   if (const char *e = getenv("MH_ZVF_PAIR"))
      m->use_zvf_pair = atoi(e);
   if (const char *e = getenv("MH_RNEA_AHEAD"))
      m->use_rnea_ahead = atoi(e);
   if (const char *e = getenv("MH_ZV_STEP"))
      m->use_zv_step = atoi(e) ? 1 : 0;
   if (const char *e = getenv("MH_ZVB_WHICH"))
      m->zvb_which = std::max(1, std::min(3, atoi(e)));
   if (const char *e = getenv("MH_ABA_LDS_FACTOR"))
      m->lds_wave_factor = atoi(e);
   if (const char *e = getenv("MH_SPEC_IO"))
      m->force_io = atoi(e);
   if (const char *e = getenv("MH_FAKE_CU_COUNT")) // measurements: shrink every grid so that one workgroup loops over the batch
      m->cu_count = std::max(1, atoi(e));
   m->lds_consts = 0; // measured on the 128-body tree (fp32, B = 131072): no difference to scalar loads
   if (const char *e = getenv("MH_GENERIC_LDS"))
      m->lds_consts = atoi(e) != 0;
   if (const char *e = getenv("MH_FUSED_FACTOR"))
      m->fused_factor = std::max(1, atoi(e));
   if (const char *e = getenv("MH_GENERIC_TRANSPOSE"))
      m->use_transpose = atoi(e) != 0;
   if (const char *e = getenv("MH_WAVES_PER_CU"))
      m->waves_per_cu = std::max(1, std::min(32, atoi(e))), m->waves_per_cu_set = true;
   if (const char *e = getenv("MH_SPEC_ST"))
      m->force_st = atoi(e);
   if (const char *e = getenv("MH_DFS"))
      m->use_dfs = atoi(e) != 0;
   if (const char *e = getenv("MH_DFS_PAIR"))
      m->use_dfs_pair = atoi(e) != 0;
   if (const char *e = getenv("MH_DFS_ABA_OCC3"))
      m->dfs_aba_occ3 = atoi(e) != 0;
   if (const char *e = getenv("MH_DFS_TRANSPOSE"))
      m->dfs_transpose = atoi(e) != 0;
   // fp64 forward dynamics at device-filling batches: the sweep kernel accumulates the children of a branching body through the workspace
   // (read-modify-write per extra child), the depth-first one keeps them on its stack -- measured on the reference's 30-joint shapes at
   // B = 262 144 (profiles/r02_generic_fp64_rates.txt): random trees 1253 -> 989 us and 1384 -> 1288 us on the depth-first kernel, chains
   // and the humanoid (4 branches in 24 joints) 3-12 % faster on the sweep.  Bushy = at least three branching bodies in ten.
   m->dfs_aba64 = m->n_nonadjacent * 10 >= 3 * std::max(1, m->n - 1);
   if (const char *e = getenv("MH_DFS_ABA64"))
      m->dfs_aba64 = atoi(e) != 0;
   if (const char *e = getenv("MH_DFS_BUDGET"))
      m->dfs_budget = std::max(0, atoi(e));
   if (const char *e = getenv("MH_DFS_PLACE"))
      m->dfs_place = atoi(e);
   if (const char *e = getenv("MH_DFS_GREEDY"))
      m->dfs_place_greedy = atoi(e) != 0;
   if (const char *e = getenv("MH_DFS_WIN"))
      m->use_win = atoi(e) != 0;
   if (getenv("MH_DISABLE_PAIR"))
      m->use_pair = 0;
   if (const char *e = getenv("MH_HOST_CHUNK"))
      m->host_chunk = std::max(0, atoi(e));
   if (const char *e = getenv("MH_SPLIT_RT"))
      m->use_split_rt = atoi(e);
   if (const char *e = getenv("MH_SPLIT_RT_LDS"))
      m->split_rt_lds = atoi(e) != 0;
   if (m->use_split_rt != 0)
      split_rt_plan(m);
   try_load_spec(m, P);
   // MH_AUTO_BUILD: no usable code object for this tree (none there, or one refused for its ABI stamp / tree) -> build one now (hipcc on the
   // box; 1: the fast form, seconds; 2: the full set, minutes -- also when only a minimal object was found)
   if (const char *ab = getenv("MH_AUTO_BUILD"); ab && atoi(ab) != 0 && m->use_spec
                                                 && ((!m->spec.handle && m->variant.compare(0, 7, "generic") == 0) || (atoi(ab) == 2 && m->spec.handle && m->spec_minimal)))
   {
      char built[1024];
      if (m->spec.handle)
      { // a minimal object is loaded and the full set was asked for
         dlclose(m->spec.handle);
         m->spec = SpecLib{};
      }
      if (build_code_object(d, getenv("MH_SPEC_DIR"), built, sizeof built, atoi(ab) != 2) == MH_OK)
         try_load_spec(m, P);
      else
         m->variant = std::string("generic (MH_AUTO_BUILD: ") + g_err + ")";
   }
   if (!m->use_spec)
      m->variant = "generic";
   if (m->split_rt.usable && m->variant.compare(0, 7, "generic") == 0)
   {
      char buf[160];
      snprintf(buf, sizeof buf, "; small batches: run-time tree split over 4 waves (%d trunk bodies + %d limbs, path %d of %d body steps)", m->split_rt.n_trunk,
               m->split_rt.n_limbs, m->split_rt.est, m->split_rt.total);
      m->variant += buf;
   }
   int selfcheck = 1;
   if (const char *e = getenv("MH_SPEC_SELFCHECK"))
      selfcheck = atoi(e);
   if (m->spec.handle && m->use_spec && selfcheck)
      self_check_spec(m);
   if (m->warnings)
      (void)fail(MH_OK, "warning: %s", m->warning_text.c_str()); // (text for mh_last_error; the status stays MH_OK)
   *model_out = m;
   return MH_OK;
}

// everything compute calls write: owned by a model (its default context) and by every context
static void free_scratch(mh_model *m)
{
   (void)hipFree(m->ws.ptr);
   (void)hipFree(m->ws_pair.ptr);
   (void)hipFree(m->tr_pair.ptr);
   (void)hipFree(m->zv_tau.ptr);
   (void)hipFree(m->zv_cols.ptr);
   (void)hipFree(m->zvb_cs.ptr);
   (void)hipFree(m->zv_flags.ptr);
   if (m->zv_error_host)
   {
      {
         std::lock_guard<std::mutex> lock(g_error_words_mutex);
         g_error_words.erase(std::remove(g_error_words.begin(), g_error_words.end(), m->zv_error_host), g_error_words.end());
      }
      (void)hipHostFree(m->zv_error_host);
   }
   if (m->pair_stream)
   {
      (void)hipStreamDestroy(m->pair_stream);
      (void)hipEventDestroy(m->pair_fork);
      (void)hipEventDestroy(m->pair_join);
   }
   (void)hipFree(m->stage.ptr);
   for (int k = 0; k < 3; k++)
   {
      if (m->ev_in[k])
         (void)hipEventDestroy(m->ev_in[k]);
      if (m->ev_run[k])
         (void)hipEventDestroy(m->ev_run[k]);
      if (m->ev_out[k])
         (void)hipEventDestroy(m->ev_out[k]);
   }
   if (m->hs_in)
      (void)hipStreamDestroy(m->hs_in);
   if (m->hs_run)
      (void)hipStreamDestroy(m->hs_run);
   if (m->hs_out)
      (void)hipStreamDestroy(m->hs_out);
   (void)hipFree(m->tr.ptr);
   (void)hipFree(m->aux.ptr);
   (void)hipFree(m->pairs.ptr);
}
// a fresh set of the above for a copy of a handle
static void reset_scratch(mh_model *m)
{
   m->ws = m->stage = m->ws_pair = m->zv_tau = m->zv_cols = m->zv_flags = m->zvb_cs = m->tr = m->tr_pair = m->aux = m->pairs = Workspace{};
   m->hs_in = m->hs_run = m->hs_out = nullptr;
   for (int k = 0; k < 3; k++)
      m->ev_in[k] = m->ev_run[k] = m->ev_out[k] = nullptr;
   m->pair_stream = nullptr, m->pair_fork = m->pair_join = nullptr;
   m->zv_epoch = 0, m->zv_error_host = m->zv_error_dev = nullptr;
   m->pairs_host.clear();
   m->lds_attr.clear();
}
// the device records, the code object and the host-side description: released once, by whoever holds the last reference
static void release_model(mh_model *m)
{
   dfs_plans_drop(m);
   split_rt_free(m);
   (void)hipFree(m->d_meta);
   (void)hipFree(m->d_dof);
   (void)hipFree(m->d_cfg);
   (void)hipFree(m->d_prog);
   (void)hipFree(m->d_prog_seq);
   (void)hipFree(m->d_consts64);
   (void)hipFree(m->d_consts32);
   free_scratch(m);
   if (m->spec.handle)
      dlclose(m->spec.handle);
   delete m;
}
// The model is reference-counted by its contexts: they share its device records, so a model destroyed while contexts are alive only
// gives up the caller's reference (its handle must not be used again); the last mh_context_destroy releases everything.
void mh_model_destroy(mh_model_t m)
{
   if (!m)
      return;
   if (m->parent)
      return; // (the inner handle of a context cannot reach a caller; should one ever be passed here, its model owns the device records)
   {
      std::lock_guard<std::mutex> lock(g_context_mutex);
      if (m->destroy_pending)
         return; // destroyed twice
      if (m->n_contexts > 0)
      {
         m->destroy_pending = true;
         return;
      }
   }
   release_model(m);
}
// ---- contexts: the model handle is read-only and may be shared by any number of host threads and streams, each calling through a
//      context of its own ("Threading"; the reference keeps this state inside the calculator object, which is why
//      it needs one calculator per thread: InverseDynamicsCalculator.java:706-707)
mh_status mh_context_create(mh_model_t model, mh_context_t *ctx_out)
{
   if (!model || !ctx_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "model / ctx_out is NULL");
   *ctx_out = nullptr;
   mh_model *root = model->parent ? model->parent : model;
   mh_model *c = nullptr;
   {
      std::lock_guard<std::mutex> lock(g_context_mutex);
      if (root->destroy_pending)
         return fail(MH_ERR_INVALID_ARGUMENT, "mh_context_create: the model has been destroyed (it lives on only for its remaining contexts)");
      // The copy reads the immutable description only: what the default context's calls (or another context's first depth-first call, under
      // dfs_mutex) may be inserting into at this moment -- dfs_plans, lds_attr, pairs_host -- is FreshOnCopy and starts empty here; the
      // plain scratch words (Workspace, streams, events) are overwritten by reset_scratch below whatever was read.
      c = new (std::nothrow) mh_model(*root);
      if (!c)
         return fail(MH_ERR_OUT_OF_MEMORY, "out of host memory");
      root->n_contexts++;
   }
   c->parent = root;
   c->n_contexts = 0;
   c->destroy_pending = false;
   reset_scratch(c);
   mh_context *ctx = new (std::nothrow) mh_context{c};
   if (!ctx)
   {
      delete c;
      std::lock_guard<std::mutex> lock(g_context_mutex);
      root->n_contexts--;
      return fail(MH_ERR_OUT_OF_MEMORY, "out of host memory");
   }
   *ctx_out = ctx;
   return MH_OK;
}
void mh_context_destroy(mh_context_t ctx)
{
   if (!ctx)
      return;
   mh_model *c = ctx->m;
   mh_model *root = c->parent;
   free_scratch(c);
   bool last = false;
   {
      std::lock_guard<std::mutex> lock(g_context_mutex);
      last = --root->n_contexts == 0 && root->destroy_pending;
   }
   delete c;
   delete ctx;
   if (last)
      release_model(root); // mh_model_destroy came first: this was the last reference
}
mh_status mh_context_reserve(mh_context_t ctx, int64_t max_batch)
{
   if (!ctx)
      return fail(MH_ERR_INVALID_ARGUMENT, "context is NULL");
   return mh_reserve(ctx->m, max_batch);
}
// Synchronises `stream` and reports what the asynchronous calls issued through this context (NULL: the model's own) left behind.
mh_status mh_model_check(mh_model_t model, mh_context_t ctx, void *stream)
{
   if (!model)
      return fail(MH_ERR_INVALID_ARGUMENT, "model is NULL");
   mh_model *m = ctx ? ctx->m : model;
   if (ctx && m->parent != (model->parent ? model->parent : model))
      return fail(MH_ERR_INVALID_ARGUMENT, "the context belongs to another model");
   HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
   return zv_check_error(m);
}
mh_status mh_topology_key(const mh_model_desc *desc, char key_out[17], int32_t *parents_out, int32_t *types_out)
{
   Plan P;
   mh_status st = plan_model(desc, P);
   if (st != MH_OK)
      return st;
   if (key_out)
      snprintf(key_out, 17, "%s", P.key.c_str());
   for (int e = 0; e < desc->n_joints; e++)
   {
      if (parents_out)
         parents_out[e] = P.eparent[e];
      if (types_out)
         types_out[e] = P.etype[e];
   }
   return MH_OK;
}
int32_t mh_model_nq(mh_model_t m) { return m ? m->nq : -1; }
int32_t mh_model_nv(mh_model_t m) { return m ? m->nv : -1; }
int32_t mh_model_n_joints(mh_model_t m) { return m ? m->n : -1; }
const char *mh_model_kernel_variant(mh_model_t m) { return m ? m->variant.c_str() : ""; }
uint32_t mh_model_warnings(mh_model_t m) { return m ? m->warnings : 0u; }
const char *mh_model_warning_text(mh_model_t m) { return m ? m->warning_text.c_str() : ""; }

// Builds the topology-specialised code object of a model with hipcc (what mecano_amd/build.py does), for hosts without Python.
mh_status mh_build_code_object(const mh_model_desc *desc, const char *out_dir, char *path_out, size_t path_cap)
{
   return build_code_object(desc, out_dir, path_out, path_cap, getenv("MH_BUILD_FAST") && atoi(getenv("MH_BUILD_FAST")) != 0);
}
static mh_status build_code_object(const mh_model_desc *desc, const char *out_dir, char *path_out, size_t path_cap, bool fast)
{
   Plan P;
   mh_status st = plan_model(desc, P);
   if (st != MH_OK)
      return st;
   const int n = desc->n_joints;
   std::vector<int> depth(n, 0);
   int deepest = 0;
   for (int e = 0; e < n; e++)
   {
      if (P.etype[e] > MH_JOINT_FIXED)
         return fail(MH_ERR_UNSUPPORTED_JOINT, "specialised code objects cover revolute, prismatic, 6-DoF and fixed joints; planar / spherical joints run on the run-time-topology kernels");
      depth[e] = 1 + (P.eparent[e] >= 0 ? depth[P.eparent[e]] : 0);
      deepest = std::max(deepest, depth[e]);
   }
   if (deepest > 16)
      return fail(MH_ERR_BAD_TOPOLOGY, "the tree is %d joints deep: a compile-time walk stops paying beyond 16 (512 registers plus hundreds of spills); such models run on the run-time-topology kernels", deepest);
   Dl_info info;
   if (!dladdr((const void *)&mh_build_code_object, &info) || !info.dli_fname)
      return fail(MH_ERR_INVALID_ARGUMENT, "cannot locate libmecano_hip.so");
   std::string dir(info.dli_fname);
   const size_t slash = dir.find_last_of('/');
   dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
   const std::string src = dir + "/csrc/mh_spec.hip";
   if (FILE *f = fopen(src.c_str(), "r"))
      fclose(f);
   else
      return fail(MH_ERR_INVALID_ARGUMENT, "%s not found: the kernel sources must sit next to the library (csrc/)", src.c_str());
   char src_hash[18];
   if (!hash_spec_sources(dir + "/csrc", src_hash))
      return fail(MH_ERR_INVALID_ARGUMENT, "%s/csrc does not hold all the kernel sources", dir.c_str());
   if (strcmp(src_hash, MH_STR_(MH_SPEC_SOURCES_HASH)) != 0)
      return fail(MH_ERR_INVALID_ARGUMENT, "the kernel sources in %s/csrc (hash %s) are not the ones this library was built beside (%s): the code object would be refused at load; rebuild the library",
                  dir.c_str(), src_hash, MH_STR_(MH_SPEC_SOURCES_HASH));
   // The compiler is clang++ itself, not the hipcc wrapper: hipcc assembles a command line of its own and hands it to a shell, so a
   // directory name with shell syntax in it would be interpreted there (seen in tests/test_abi.py).  MH_HIPCC overrides the choice.
   const char *hipcc = getenv("MH_HIPCC");
   std::string cc = hipcc ? hipcc : "";
   if (cc.empty())
   {
      for (const char *candidate : {"/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++"})
         if (FILE *f = fopen(candidate, "r"))
         {
            fclose(f);
            cc = candidate;
            break;
         }
      if (cc.empty())
         cc = "amdclang++";
   }
   const size_t base = cc.find_last_of('/');
   const bool wrapper = cc.compare(base == std::string::npos ? 0 : base + 1, std::string::npos, "hipcc") == 0; // a user's MH_HIPCC=hipcc: its own driver flags
   std::string parents, kinds;
   for (int e = 0; e < n; e++)
   {
      parents += (e ? "," : "") + std::to_string(P.eparent[e]);
      kinds += (e ? "," : "") + std::to_string(P.etype[e]);
   }
   // A fast build (-DMH_SPEC_MINIMAL: tree-split RNEA / ABA / pair kernels for AoS matrices with identity index maps only) gets a name of
   // its own, so that it can neither be taken for the full object by mecano_amd.build (which would never build the full set then) nor
   // replace a full object; the loader prefers the full object and falls back to the minimal one.
   const std::string out = std::string(out_dir ? out_dir : dir.c_str()) + "/libmecano_hip_topo_" + P.key + (fast ? ".min.so" : ".so");
   const std::string tmp = out + ".tmp" + std::to_string((long)getpid());
   // hipcc is started WITHOUT a shell (posix_spawn with an argument vector): a directory name or flag handed in by an application cannot
   // be interpreted as shell syntax.  MH_HIPCC_FLAGS is split at white space into separate arguments.
   // (the flags of mecano_amd/build.py's SPEC_FLAGS: see there for -disable-machine-licm)
   std::vector<std::string> argv_s = {cc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-signed-zeros", "-ffinite-math-only",
                                      "-fno-slp-vectorize", "-mllvm", "-disable-machine-licm"};
   if (!wrapper)
   {
      argv_s.push_back("--driver-mode=g++");
      argv_s.push_back("--hip-link");
   }
   if (fast)
      argv_s.push_back("-DMH_SPEC_MINIMAL");
   argv_s.push_back(std::string("-DMH_SPEC_SOURCES_HASH=") + src_hash);
   {
      // what else shaped this object (the fast form, a caller's extra flags): part of the build id its file carries
      uint64_t xh = fnv1a(0xcbf29ce484222325ull, fast ? "-DMH_SPEC_MINIMAL" : "", fast ? 17 : 0);
      if (const char *extra = getenv("MH_HIPCC_FLAGS"))
         xh = fnv1a(xh, extra, strlen(extra));
      char xs[48];
      snprintf(xs, sizeof xs, "-DMH_BUILD_EXTRA=h%016llx", (unsigned long long)xh);
      argv_s.push_back(xs);
   }
   if (const char *extra = getenv("MH_HIPCC_FLAGS"))
   {
      std::string word;
      for (const char *c = extra;; c++)
      {
         if (*c == 0 || *c == ' ' || *c == '\t' || *c == '\n')
         {
            if (!word.empty())
               argv_s.push_back(word);
            word.clear();
            if (*c == 0)
               break;
         }
         else
            word += *c;
      }
   }
   argv_s.push_back("-DMH_TOPO_N=" + std::to_string(n));
   argv_s.push_back("-DMH_TOPO_PARENTS=" + parents);
   argv_s.push_back("-DMH_TOPO_TYPES=" + kinds);
   argv_s.push_back("-o");
   argv_s.push_back(tmp);
   if (!wrapper)
   {
      argv_s.push_back("-x");
      argv_s.push_back("hip");
   }
   argv_s.push_back(src);
   std::vector<char *> argv;
   for (std::string &a : argv_s)
      argv.push_back(const_cast<char *>(a.c_str()));
   argv.push_back(nullptr);
   pid_t pid = 0;
   const int sp = posix_spawnp(&pid, cc.c_str(), nullptr, nullptr, argv.data(), environ);
   if (sp != 0)
      return fail(MH_ERR_HIP, "cannot start %s: %s", cc.c_str(), strerror(sp));
   int status = 0;
   while (waitpid(pid, &status, 0) < 0)
      if (errno != EINTR)
         return fail(MH_ERR_HIP, "waiting for %s failed: %s", cc.c_str(), strerror(errno));
   if (!WIFEXITED(status) || WEXITSTATUS(status) != 0)
   {
      (void)unlink(tmp.c_str());
      if (WIFEXITED(status))
         return fail(MH_ERR_HIP, "building the code object failed: %s exited with status %d", cc.c_str(), WEXITSTATUS(status));
      return fail(MH_ERR_HIP, "building the code object failed: %s was ended by signal %d", cc.c_str(), WIFSIGNALED(status) ? WTERMSIG(status) : -1);
   }
   if (rename(tmp.c_str(), out.c_str()) != 0) // atomic: a concurrent build of the same tree cannot leave a torn file
   {
      (void)unlink(tmp.c_str());
      return fail(MH_ERR_HIP, "cannot move the code object to %s: %s", out.c_str(), strerror(errno));
   }
   if (path_out && path_cap)
      snprintf(path_out, path_cap, "%s", out.c_str());
   return MH_OK;
}

mh_status mh_reserve(mh_model_t m, int64_t max_batch)
{
   mh_status st = check_common(m, max_batch, nullptr);
   if (st != MH_OK)
      return st;
   st = ensure_workspace(m, max_batch, sizeof(double));
   if (st != MH_OK)
      return st;
   { // mh_regressor_*: one workspace block per wave, several waves per group of configurations on small batches
      const Launch L = plan_launch(m, max_batch);
      st = ensure_bytes(m->ws, (size_t)m->n_slots * (size_t)L.lanes * regressor_parts(m, L) * sizeof(double));
      if (st != MH_OK)
         return st;
   }
   // the whole-tree specialised ABA keeps its hand-over store in the same workspace (more slots than the run-time-topology plan of a
   // chain), and big AoS batches of wide matrices go through transposed scratch copies: reserve both, so that compute calls allocate nothing
   if (m->spec.aba_slots)
   {
      const Launch L = plan_launch(m, max_batch);
      st = ensure_bytes(m->ws, (size_t)std::max(m->n_slots, m->spec.aba_slots()) * (size_t)L.lanes * sizeof(double));
      if (st != MH_OK)
         return st;
   }
   const bool transposes = m->use_transpose >= 0 ? m->use_transpose != 0 : (max_batch >= 8192 && m->nq + m->nv >= 64);
   if (transposes)
      st = ensure_bytes(m->tr, (size_t)max_batch * ((size_t)m->nq + 3 * (size_t)m->nv) * sizeof(double));
   if (st != MH_OK)
      return st;
   // the depth-first kernels: the frame plans a batch of this size gets (both algorithms, both precisions, both layouts) and the
   // global blocks behind them; the second workspace of the side-by-side pair call
   size_t dfs_bytes = 0;
   for (int a = 0; a < 2 && m->use_dfs; a++)
      for (size_t elem : {sizeof(double), sizeof(float)})
         for (int aos = 0; aos < 2; aos++)
         {
            const Algo algo = a == 0 ? ALGO_RNEA : ALGO_ABA;
            const DfsChoice ch = dfs_choose(m, algo, elem, max_batch, dfs_windows(m, algo, elem, aos != 0));
            const mh_model::DfsPlan *plan = dfs_plan(m, a, (int)ch.budget);
            if (!plan)
               return fail(MH_ERR_HIP, "depth-first kernels: the body records of the frame plan could not be uploaded");
            const long lds = (plan->lds_slots + (ch.hand_lds ? ch.hand : 0)) * ch.slot_bytes + ch.b_win;
            const long per_cu = lds > 0 ? std::max<long>(1, std::min<long>(ch.per_cu, (160 * 1024) / lds)) : ch.per_cu;
            const long grid = std::max<long>(1, std::min<long>((max_batch + 63) / 64, (long)m->cu_count * per_cu));
            dfs_bytes = std::max(dfs_bytes, (size_t)((ch.hand_lds ? 0 : ch.hand) + plan->glb_slots) * (size_t)grid * 64 * elem);
         }
   if (m->split_rt.usable) // the run-time tree split: one workspace block per workgroup
      dfs_bytes = std::max(dfs_bytes, (size_t)m->split_rt.slots * (size_t)std::max<long>(1, std::min<long>((max_batch + 63) / 64, 2L * m->cu_count)) * 64 * sizeof(double));
   if (dfs_bytes > 0)
      st = ensure_bytes(m->ws, dfs_bytes);
   if (st == MH_OK && m->use_pair && (max_batch + 63) / 64 <= (long)m->cu_count)
      st = ensure_bytes(m->ws_pair, m->ws.bytes);
   // round 3's plans: the bias-split launches (their scratch for the largest batch they serve: two jobs on every group of 64 within the
   // CUs), the one-launch pair of the run-time tree split (twice the workgroups of a single call), the shared transposed copies of
   // mh_rnea_aba_f32 -- a call right after mh_reserve may be captured into a graph, where an allocation is an error
   if (st == MH_OK && m->spec.launch_zv && m->use_zv)
   {
      st = zv_prepare(m, std::min<int64_t>(max_batch, m->use_zv == 2 ? max_batch : 32L * m->cu_count), nullptr);
      if (st == MH_OK)
         HIP_TRY(hipStreamSynchronize(nullptr)); // the flags are zero before any stream's first launch looks at them
   }
   if (st == MH_OK && zvb_ok(m, max_batch, false) && !zvf_ok(m, max_batch, false))
   { // the two-launch forward dynamics of device-filling batches (models without the fused kernel): bias rows and (cos, sin) pairs
      st = ensure_bytes(m->zv_tau, (size_t)max_batch * m->nv * sizeof(double));
      if (st == MH_OK)
         st = ensure_bytes(m->zvb_cs, std::max<size_t>(1, (size_t)m->spec.zvb_cs_rows()) * (size_t)((max_batch + 63) / 64 * 64) * sizeof(double));
   }
   if (st == MH_OK && m->split_rt.usable)
      st = ensure_bytes(m->ws, (size_t)m->split_rt.slots * (size_t)std::min<long>(2 * ((max_batch + 63) / 64), (long)m->cu_count) * 64 * sizeof(double));
   if (st == MH_OK && transposes && m->use_dfs)
      st = ensure_bytes(m->tr_pair, (size_t)max_batch * ((size_t)m->nq + 5 * (size_t)m->nv) * sizeof(float));
   return st;
}

mh_status mh_rnea_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double gravity[3],
                      const double *f_ext, const mh_options *opts, double *tau_out)
{
   return launch<double>(ALGO_RNEA, model, B, q, qd, qdd, gravity, f_ext, opts, tau_out);
}
mh_status mh_aba_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *tau, const double gravity[3],
                     const double *f_ext, const mh_options *opts, double *qdd_out)
{
   return launch<double>(ALGO_ABA, model, B, q, qd, tau, gravity, f_ext, opts, qdd_out);
}
mh_status mh_crba_f64(mh_model_t model, int64_t B, const double *q, const mh_options *opts, double *H_out)
{
   return launch<double>(ALGO_CRBA, model, B, q, nullptr, nullptr, nullptr, nullptr, opts, H_out);
}
mh_status mh_regressor_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double gravity[3],
                           const mh_options *opts, int32_t first_moment_columns, double *Y_out)
{
   return regressor_impl<double>(model, B, q, qd, qdd, gravity, opts, first_moment_columns, Y_out);
}
mh_status mh_regressor_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *qdd, const double gravity[3],
                           const mh_options *opts, int32_t first_moment_columns, float *Y_out)
{
   return regressor_impl<float>(model, B, q, qd, qdd, gravity, opts, first_moment_columns, Y_out);
}
mh_status mh_crba_coriolis_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const mh_options *opts, double *H_out, double *C_out)
{
   return coriolis_impl<double>(model, B, q, qd, opts, H_out, C_out);
}
mh_status mh_crba_coriolis_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const mh_options *opts, float *H_out, float *C_out)
{
   return coriolis_impl<float>(model, B, q, qd, opts, H_out, C_out);
}
mh_status mh_centroidal_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double frame[12], int32_t frame_mode,
                            const mh_options *opts, double *A_out, double *b_out, double *com_out)
{
   return centroidal_impl<double>(model, B, q, qd, frame, frame_mode, opts, A_out, b_out, com_out);
}
mh_status mh_centroidal_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const double frame[12], int32_t frame_mode,
                            const mh_options *opts, float *A_out, float *b_out, float *com_out)
{
   return centroidal_impl<float>(model, B, q, qd, frame, frame_mode, opts, A_out, b_out, com_out);
}
mh_status mh_integrate_f64(mh_model_t model, int64_t B, double dt, const double *q, const double *qd, const double *qdd, const mh_options *opts,
                           double *q_out, double *qd_out, double *qdd_out)
{
   return integrate_impl<double>(model, B, dt, q, qd, qdd, opts, q_out, qd_out, qdd_out);
}
mh_status mh_aba_integrate_f64(mh_model_t model, int64_t B, double dt, const double *q, const double *qd, const double *tau,
                               const double gravity[3], const double *f_ext, const mh_options *opts, double *qdd_out, double *q_next,
                               double *qd_next)
{
   if (B > 0 && (!q_next || !qd_next))
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer");
   if (nan_bits(dt))
      return fail(MH_ERR_INVALID_ARGUMENT, "dt is NaN");
   bool stepped = false;
   mh_status st = launch<double>(ALGO_ABA, model, B, q, qd, tau, gravity, f_ext, opts, qdd_out, nullptr, nullptr, nullptr, nullptr, false, dt, q_next,
                                 qd_next, &stepped);
   if (st != MH_OK || stepped || B == 0)
      return st;
   return mh_integrate_f64(model, B, dt, q, qd, qdd_out, opts, q_next, qd_next, nullptr);
}
mh_status mh_integrate_f32(mh_model_t model, int64_t B, double dt, const float *q, const float *qd, const float *qdd, const mh_options *opts,
                           float *q_out, float *qd_out, float *qdd_out)
{
   return integrate_impl<float>(model, B, dt, q, qd, qdd, opts, q_out, qd_out, qdd_out);
}
mh_status mh_rnea_bodies_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double gravity[3],
                             const double *f_ext, const mh_options *opts, double *tau_out, double *body_acc_out, double *body_twist_out)
{
   return launch<double>(ALGO_RNEA, model, B, q, qd, qdd, gravity, f_ext, opts, tau_out, nullptr, nullptr, body_acc_out, body_twist_out, true);
}
mh_status mh_aba_bodies_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *tau, const double gravity[3],
                            const double *f_ext, const mh_options *opts, double *qdd_out, double *body_acc_out, double *body_twist_out)
{
   return launch<double>(ALGO_ABA, model, B, q, qd, tau, gravity, f_ext, opts, qdd_out, nullptr, nullptr, body_acc_out, body_twist_out, true);
}
mh_status mh_rnea_joint_wrenches_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double gravity[3],
                                     const double *f_ext, const mh_options *opts, double *tau_out, double *joint_wrench_out)
{
   if (B > 0 && !joint_wrench_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "joint_wrench_out is NULL");
   return launch<double>(ALGO_RNEA, model, B, q, qd, qdd, gravity, f_ext, opts, tau_out, nullptr, nullptr, nullptr, nullptr, true, 0.0, nullptr, nullptr,
                         nullptr, joint_wrench_out);
}
mh_status mh_aba_joint_wrenches_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *tau, const double gravity[3],
                                    const double *f_ext, const mh_options *opts, double *qdd_out, double *joint_wrench_out)
{
   if (B > 0 && !joint_wrench_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "joint_wrench_out is NULL");
   // the scratch below belongs to the CONTEXT of the call: resolve it before anything mutable is touched (launch() does so for itself only)
   mh_options o;
   if (opts)
      o = *opts;
   else
      mh_options_default(&o);
   opts = &o;
   mh_status st = check_common(model, B, opts);
   if (st != MH_OK)
      return st;
   st = launch<double>(ALGO_ABA, model, B, q, qd, tau, gravity, f_ext, opts, qdd_out);
   if (st != MH_OK || B == 0)
      return st;
   // ForwardDynamicsCalculator.getJointWrench (ForwardDynamicsCalculator.java:1330-1363) is a Newton-Euler sweep over the accelerations
   // forward dynamics has just produced: the RNEA kernel with its joint-wrench output, on the same stream; its efforts (= tau up to
   // rounding) go to scratch
   st = ensure_bytes(model->aux, (size_t)B * model->nv * sizeof(double));
   if (st != MH_OK)
      return st;
   return launch<double>(ALGO_RNEA, model, B, q, qd, qdd_out, gravity, f_ext, opts, (double *)model->aux.ptr, nullptr, nullptr, nullptr, nullptr, true,
                         0.0, nullptr, nullptr, nullptr, joint_wrench_out);
}
mh_status mh_relative_acceleration_f64(mh_model_t model, int64_t B, const double *q, const double *body_acc, const double *body_twist,
                                       const double gravity[3], int32_t n_pairs, const int32_t *base_joints, const int32_t *body_joints,
                                       const mh_options *opts_in, double *out)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (n_pairs < 0)
      return fail(MH_ERR_BAD_DIMENSION, "negative number of pairs %d", n_pairs);
   if (B == 0 || n_pairs == 0)
      return MH_OK;
   if (!q || !body_acc || (!gravity && !opts.use_root_acceleration) || !base_joints || !body_joints || !out)
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / pair / output pointer");
   model->pairs_host.resize((size_t)n_pairs * 2);
   for (int k = 0; k < n_pairs; k++)
   {
      const int b1 = base_joints[k], b2 = body_joints[k];
      if (b1 < -1 || b1 >= model->n || b2 < -1 || b2 >= model->n)
         return fail(MH_ERR_INVALID_ARGUMENT, "pair %d names joint %d / %d (the model has %d joints; -1 = the root body)", k, b1, b2, model->n);
      model->pairs_host[2 * k] = b1 < 0 ? -1 : model->engine_of[b1];
      model->pairs_host[2 * k + 1] = b2 < 0 ? -1 : model->engine_of[b2];
   }
   st = ensure_bytes(model->pairs, model->pairs_host.size() * sizeof(int));
   if (st != MH_OK)
      return st;
   hipStream_t stream = (hipStream_t)opts.stream;
   HIP_TRY(hipMemcpyAsync(model->pairs.ptr, model->pairs_host.data(), model->pairs_host.size() * sizeof(int), hipMemcpyHostToDevice, stream));
   mh::RelArgs<double> A{};
   A.m = dev_model<double>(model);
   A.B = B;
   A.q = q, A.body_acc = body_acc, A.body_twist = opts.consider_coriolis ? body_twist : nullptr, A.out = out;
   A.pairs = (const int *)model->pairs.ptr, A.n_pairs = n_pairs;
   const bool soa = opts.layout == MH_LAYOUT_SOA;
   A.q_bs = soa ? 1 : model->nq, A.q_es = soa ? B : 1;
   A.f_bs = soa ? 1 : (long)model->n * 6, A.f_es = soa ? B : 1;
   A.o_bs = soa ? 1 : (long)n_pairs * 6, A.o_es = soa ? B : 1;
   set_root_acceleration(A, opts, gravity);
   if (opts.consider_coriolis && !body_twist)
      return fail(MH_ERR_INVALID_ARGUMENT, "body_twist is NULL but velocities are considered (opts->consider_coriolis)");
   const int block = 64;
   const int grid = (int)std::max<long>(1, std::min<long>((B + block - 1) / block, (long)model->cu_count * 8));
   hipLaunchKernelGGL((mh::relative_acceleration_kernel<double>), dim3(grid), dim3(block), 0, stream, A);
   HIP_TRY(hipGetLastError());
   return MH_OK;
}
mh_status mh_model_set_joint_source_modes(mh_model_t model, const int32_t *modes)
{
   if (!model)
      return fail(MH_ERR_INVALID_ARGUMENT, "model is NULL");
   int cur = 0;
   if (hipGetDevice(&cur) != hipSuccess || cur != model->device)
      return fail(MH_ERR_INVALID_ARGUMENT, "model lives on device %d, which is not the calling thread's device", model->device);
   {
      std::lock_guard<std::mutex> lock(g_context_mutex);
      if (model->parent || model->n_contexts > 0)
         return fail(MH_ERR_INVALID_ARGUMENT, "joint source modes are set on the model itself, before its contexts are created (%d exist)", model->parent ? 1 : model->n_contexts);
   }
   for (int e = 0; modes && e < model->n; e++)
      if (modes[e] != MH_EFFORT_SOURCE && modes[e] != MH_ACCELERATION_SOURCE)
         return fail(MH_ERR_INVALID_ARGUMENT, "joint %d: unknown source mode %d", e, modes[e]);
   int n_locked = 0;
   for (int e = 0; e < model->n; e++)
   {
      int *mi = &model->meta[(size_t)e * mh::MI_STRIDE];
      const bool lk = modes && modes[mi[mh::MI_EXT]] == MH_ACCELERATION_SOURCE;
      mi[mh::MI_FLAGS] = (mi[mh::MI_FLAGS] & ~mh::MF_LOCKED) | (lk ? mh::MF_LOCKED : 0);
      n_locked += lk;
   }
   HIP_TRY(hipDeviceSynchronize());
   HIP_TRY(hipMemcpy(model->d_meta, model->meta.data(), model->meta.size() * sizeof(int), hipMemcpyHostToDevice));
   dfs_plans_drop(model); // copies of the records
   model->n_locked = n_locked;
   return MH_OK;
}
int32_t mh_model_n_acceleration_sources(mh_model_t model) { return model ? model->n_locked : -1; }
mh_status mh_aba_locked_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *tau, const double *qdd_in,
                            const double gravity[3], const double *f_ext, const mh_options *opts, double *qdd_out, double *tau_out)
{
   if (!model)
      return fail(MH_ERR_INVALID_ARGUMENT, "model is NULL");
   if (model->n_locked == 0)
   { // nothing is locked: the ordinary forward dynamics, efforts copied through
      mh_status st = launch<double>(ALGO_ABA, model, B, q, qd, tau, gravity, f_ext, opts, qdd_out);
      if (st == MH_OK && tau_out && tau_out != tau && B > 0)
         HIP_TRY(hipMemcpyAsync(tau_out, tau, (size_t)B * model->nv * sizeof(double), hipMemcpyDeviceToDevice, opts ? (hipStream_t)opts->stream : nullptr));
      return st;
   }
   if (B > 0 && !qdd_in)
      return fail(MH_ERR_INVALID_ARGUMENT, "qdd_in is NULL but %d joint(s) are acceleration sources", model->n_locked);
   return launch<double>(ALGO_ABA, model, B, q, qd, tau, gravity, f_ext, opts, qdd_out, qdd_in, tau_out);
}
// The pair calls run their two algorithms side by side (or phase by phase in one workgroup): an output that overlaps an input of the OTHER
// algorithm would be read half-written.  (mh_rnea_* then mh_aba_* is the call sequence for in-place use.)
static bool pair_outputs_overlap(const void *q, const void *qd, const void *qdd, const void *tau, const void *tau_out, const void *qdd_out, size_t bq, size_t bv)
{
   auto overlap = [](const void *a, size_t na, const void *b, size_t nb) { return (const char *)a < (const char *)b + nb && (const char *)b < (const char *)a + na; };
   const void *ins[4] = {q, qd, qdd, tau};
   const size_t nin[4] = {bq, bv, bv, bv};
   for (int i = 0; i < 4; i++)
      if (overlap(tau_out, bv, ins[i], nin[i]) || overlap(qdd_out, bv, ins[i], nin[i]))
         return true;
   return overlap(tau_out, bv, qdd_out, bv);
}
mh_status mh_rnea_aba_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double *tau,
                          const double gravity[3], const double *f_ext, const mh_options *opts_in, double *tau_out, double *qdd_out)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (B == 0)
      return MH_OK;
   if (!q || !qd || !qdd || !tau || (!gravity && !opts.use_root_acceleration) || !tau_out || !qdd_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer");
   if (pair_outputs_overlap(q, qd, qdd, tau, tau_out, qdd_out, (size_t)B * model->nq * sizeof(double), (size_t)B * model->nv * sizeof(double)))
      return fail(MH_ERR_INVALID_ARGUMENT, "mh_rnea_aba_f64: tau_out / qdd_out must not overlap q, qd, qdd, tau or each other (the two algorithms run concurrently)");
   const long waves = (B + 63) / 64;
   const bool fusable = model->n_locked == 0 && model->spec.launch_fused && model->use_spec && model->use_fused && model->dense_maps && opts.layout == MH_LAYOUT_AOS
                        && opts.consider_coriolis && opts.consider_accelerations && 2 * waves <= (long)model->cu_count * model->fused_factor
                        && model->spec.fused_lds_bytes(model->nq, model->nv) <= 160 * 1024
                        // beyond one group per CU the fused forward-dynamics kernel behind mh_aba_f64 is worth more than one launch for both
                        // (humanoid: 43.1 against 56.6 us at 32 768, 40.0 against 45.5 at 24 576; profiles/r04_pair_call_rates.txt)
                        && !zvf_ok(model, B, false);
   // No fused kernel for this call (no code object, SoA, switches, ...).  While the batch leaves most of the device idle -- the
   // run-time-topology kernels put one wave per 64 configurations on it -- the two launches run SIDE BY SIDE: the ABA on a stream of the
   // model's own, forked from and joined back into the caller's stream with events, on a workspace of its own (humanoid without its code
   // object, B = 4096: 63 + 112 us one after the other, ~115 us together).
   auto two_calls = [&]() -> mh_status {
      hipStream_t s = (hipStream_t)opts.stream;
      // no code object would serve either call and the run-time tree split serves both: one launch, half the grid each
      const bool no_spec = !model->use_spec || !model->spec.launch_split || !model->spec.split_usable || !model->spec.split_usable();
      if (no_spec && model->use_pair && model->split_rt.usable && model->n_locked == 0 && model->use_split_rt != 0 && 2 * waves <= (long)model->cu_count)
      {
         mh::Args<double> P{};
         P.m = dev_model<double>(model);
         P.B = B;
         P.q = q, P.qd = qd, P.in3 = qdd, P.fext = f_ext, P.out = tau_out;
         P.in3b = tau, P.outb = qdd_out;
         const bool soa = opts.layout == MH_LAYOUT_SOA;
         P.q_bs = soa ? 1 : model->nq, P.q_es = soa ? B : 1;
         P.v_bs = soa ? 1 : model->nv, P.v_es = soa ? B : 1;
         P.f_bs = soa ? 1 : (long)model->n * 6, P.f_es = soa ? B : 1;
         set_root_acceleration(P, opts, gravity);
         P.coriolis = opts.consider_coriolis, P.accel = opts.consider_accelerations;
         return launch_split_rt_pair(model, B, P, s);
      }
      if (!model->use_pair || waves > (long)model->cu_count) // measured: pays up to one wave per CU (profiles/r02_generic_pair_side_by_side.txt)
      {
         mh_status r = mh_rnea_f64(model, B, q, qd, qdd, gravity, f_ext, &opts, tau_out);
         return r != MH_OK ? r : mh_aba_f64(model, B, q, qd, tau, gravity, f_ext, &opts, qdd_out);
      }
      if (!model->pair_stream)
      {
         HIP_TRY(hipStreamCreateWithFlags(&model->pair_stream, hipStreamNonBlocking));
         HIP_TRY(hipEventCreateWithFlags(&model->pair_fork, hipEventDisableTiming));
         HIP_TRY(hipEventCreateWithFlags(&model->pair_join, hipEventDisableTiming));
      }
      HIP_TRY(hipEventRecord(model->pair_fork, s));
      HIP_TRY(hipStreamWaitEvent(model->pair_stream, model->pair_fork, 0));
      mh_options ob = opts;
      ob.stream = (void *)model->pair_stream;
      std::swap(model->ws, model->ws_pair); // every launch path sizes and reads model->ws ...
      std::swap(model->tr, model->tr_pair); // ... and model->tr when it goes through transposed copies of the state matrices
      const mh_status rb = mh_aba_f64(model, B, q, qd, tau, gravity, f_ext, &ob, qdd_out);
      std::swap(model->ws, model->ws_pair);
      std::swap(model->tr, model->tr_pair);
      const mh_status ra = mh_rnea_f64(model, B, q, qd, qdd, gravity, f_ext, &opts, tau_out);
      HIP_TRY(hipEventRecord(model->pair_join, model->pair_stream)); // join even after an error: the caller's stream must not run ahead
      HIP_TRY(hipStreamWaitEvent(s, model->pair_join, 0));
      return rb != MH_OK ? rb : ra;
   };
   // Device-filling batches: the fused forward-dynamics kernel computes tau too (its inverse-dynamics phase has h, one more walk without
   // velocities adds M(q) qdd of the caller's accelerations: mh_zv_kernels.h, ZvfDelta) -- one launch instead of the inverse dynamics'
   // own (humanoid, 262 144 configurations: 254 us for the two).  MH_ZVF_PAIR=0: the two launches.
   if (model->n_locked == 0 && model->use_spec && model->use_zvf_pair && opts.layout == MH_LAYOUT_AOS && opts.consider_coriolis && opts.consider_accelerations
       && zvf_ok(model, B, false) && model->spec.zvf_pair_usable && model->spec.zvf_pair_usable())
   {
      mh::Args<double> Z{};
      Z.m = dev_model<double>(model);
      Z.B = B;
      Z.q = q, Z.qd = qd, Z.in3 = tau, Z.fext = f_ext, Z.out = qdd_out;
      Z.in3b = qdd, Z.outb = tau_out;
      Z.q_bs = model->nq, Z.q_es = 1, Z.v_bs = model->nv, Z.v_es = 1, Z.f_bs = (long)model->n * 6, Z.f_es = 1;
      set_root_acceleration(Z, opts, gravity);
      Z.coriolis = 1, Z.accel = 1;
      const long groups = std::min<long>((B + 63) / 64, (long)model->cu_count * 2);
      const int rcz = model->spec.launch_zvf(SPEC_IO_LDS | SPEC_IDENT, &Z, (int)groups, opts.stream);
      if (rcz == 0)
         return MH_OK;
      if (rcz != (int)hipErrorNotSupported)
         return fail(MH_ERR_HIP, "fused forward + inverse dynamics failed to launch: %s", hipGetErrorString((hipError_t)rcz));
   }
   if (!fusable)
      return two_calls();
   mh::Args<double> A{};
   A.m = dev_model<double>(model);
   A.B = B;
   A.q = q, A.qd = qd, A.in3 = qdd, A.fext = f_ext, A.out = tau_out;
   A.in3b = tau, A.outb = qdd_out;
   A.body_acc = nullptr, A.body_twist = nullptr;
   A.dt = 0.0, A.q_next = nullptr, A.qd_next = nullptr;
   A.ws = nullptr, A.ws_stride = 0;
   A.q_bs = model->nq, A.q_es = 1, A.v_bs = model->nv, A.v_es = 1, A.f_bs = (long)model->n * 6, A.f_es = 1;
   set_root_acceleration(A, opts, gravity);
   A.coriolis = 1, A.accel = 1;
   if (zv_ok(model, B, false, 3))
   { // three jobs in one launch: bias efforts | articulated inertias + bias fold | the inverse dynamics output (mh_zv_kernels.h)
      if (const mh_status se = zv_check_error(model); se != MH_OK)
         return se;
      int rc3 = 0;
      if (const mh_status sz = zv_launch(model, A, 3, (hipStream_t)opts.stream, &rc3); sz != MH_OK)
         return sz;
      if (rc3 == 0)
         return MH_OK;
   }
   if (split_ok(model, 2, B, false))
   {
      const int rc2 = model->spec.launch_split(2, split_flags(model, 2, false), &A, (int)waves, opts.stream);
      if (rc2 == 0)
         return MH_OK;
      if (rc2 != (int)hipErrorNotSupported)
         return fail(MH_ERR_HIP, "fused tree-split kernel launch failed: %s", hipGetErrorString((hipError_t)rc2));
      return two_calls(); // not in this code object (fast build)
   }
   if (!model->spec.supports(1, SPEC_ST_LDS | (model->ident_maps ? SPEC_IDENT : 0)))
   // no whole-tree ABA in this code object (trees with a tree-split form) and the tree-split launch was ruled out: two calls
      return two_calls();
   const int rc = model->spec.launch_fused(model->ident_maps ? SPEC_IDENT : 0, &A, (int)waves, opts.stream);
   if (rc == (int)hipErrorNotSupported)
      return two_calls();
   if (rc != 0)
      return fail(MH_ERR_HIP, "fused kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
   return MH_OK;
}
mh_status mh_rnea_crba_f64(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double gravity[3],
                           const double *f_ext, const mh_options *opts_in, double *tau_out, double *H_out)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (B == 0)
      return MH_OK;
   if (!q || !qd || !qdd || (!gravity && !opts.use_root_acceleration) || !tau_out || !H_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer");
   hipStream_t s = (hipStream_t)opts.stream;
   const long groups = (B + 63) / 64;
   // one launch: tree-split RNEA groups and tree-split CRBA groups side by side (code object with identity maps, AoS, no switches, no
   // external wrenches through this path; batches that leave room for both on the device)
   if (model->spec.launch_rnea_crba && model->use_spec && model->use_fused && model->ident_maps && model->dense_maps && opts.layout == MH_LAYOUT_AOS
       && opts.consider_coriolis && opts.consider_accelerations && model->use_split != 0 && model->spec.split_usable && model->spec.split_usable()
       && model->spec.crba_split_usable && model->spec.crba_split_usable() && groups <= (long)model->cu_count * 2)
   {
      mh::Args<double> A{};
      A.m = dev_model<double>(model);
      A.B = B;
      A.q = q, A.qd = qd, A.in3 = qdd, A.fext = f_ext, A.out = tau_out;
      A.in3b = nullptr, A.outb = H_out;
      A.ws = nullptr, A.ws_stride = 0;
      A.q_bs = model->nq, A.q_es = 1, A.v_bs = model->nv, A.v_es = 1, A.f_bs = (long)model->n * 6, A.f_es = 1;
      set_root_acceleration(A, opts, gravity);
      A.coriolis = 1, A.accel = 1;
      // thin CRBA workgroups: its write-out is bound by the stores in flight per CU.  Every workgroup must be resident at once (the RNEA
      // groups would otherwise queue behind the CRBA's): one per CU, two where the thinner image leaves LDS for it (202 registers: two
      // waves per SIMD)
      auto resident = [&](int l) {
         const long lds = model->spec.rnea_crba_lds_bytes ? model->spec.rnea_crba_lds_bytes(l, model->nq, model->nv) : 160 * 1024;
         return (long)model->cu_count * (2 * lds <= 160 * 1024 ? 2 : 1);
      };
      int lpg = 64;
      while (lpg > 16 && (B + lpg / 2 - 1) / (lpg / 2) + groups <= resident(lpg / 2))
         lpg /= 2;
      // ... and where the CUs the RNEA groups leave can take one CRBA workgroup each, the width that does exactly that: a workgroup that
      // shares its CU's SIMDs is the launch's last (humanoid, 4 096: 22 configurations in 187 workgroups beside the 64 RNEA groups 17.4 us,
      // 16 in 256: 18.0, 20 in 205: 19.0, 24: 18.7 -- profiles/r04_rnea_crba_lpg.txt)
      if (groups < (long)model->cu_count)
      {
         const long free_cus = (long)model->cu_count - groups;
         const long fit = (B + free_cus - 1) / free_cus;
         if (fit >= 16 && fit <= 64 && fit > lpg)
            lpg = (int)fit;
      }
      if (const char *e = getenv("MH_RNEA_CRBA_LPG")) // experiments: 16 ... 64
         lpg = std::max(16, std::min(64, atoi(e)));
      const long ng = std::min<long>((B + lpg - 1) / lpg, (long)model->cu_count * 2);
      const int rc = model->spec.launch_rnea_crba(&A, (int)groups, (int)ng, lpg, (void *)s);
      if (rc == 0)
         return MH_OK;
      if (rc != (int)hipErrorNotSupported)
         return fail(MH_ERR_HIP, "fused RNEA + CRBA launch failed: %s", hipGetErrorString((hipError_t)rc));
   }
   // two launches; side by side while the batch leaves most of the device idle (the CRBA on the model's own stream; it needs no workspace
   // of the RNEA's kind when a code object serves it, and gets its own otherwise)
   if (!model->use_pair || groups > (long)model->cu_count)
   {
      st = mh_rnea_f64(model, B, q, qd, qdd, gravity, f_ext, &opts, tau_out);
      return st != MH_OK ? st : mh_crba_f64(model, B, q, &opts, H_out);
   }
   if (!model->pair_stream)
   {
      HIP_TRY(hipStreamCreateWithFlags(&model->pair_stream, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&model->pair_fork, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&model->pair_join, hipEventDisableTiming));
   }
   HIP_TRY(hipEventRecord(model->pair_fork, s));
   HIP_TRY(hipStreamWaitEvent(model->pair_stream, model->pair_fork, 0));
   mh_options ob = opts;
   ob.stream = (void *)model->pair_stream;
   std::swap(model->ws, model->ws_pair);
   const mh_status rb = mh_crba_f64(model, B, q, &ob, H_out);
   std::swap(model->ws, model->ws_pair);
   const mh_status ra = mh_rnea_f64(model, B, q, qd, qdd, gravity, f_ext, &opts, tau_out);
   HIP_TRY(hipEventRecord(model->pair_join, model->pair_stream));
   HIP_TRY(hipStreamWaitEvent(s, model->pair_join, 0));
   return rb != MH_OK ? rb : ra;
}
mh_status mh_rnea_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *qdd, const double gravity[3],
                      const float *f_ext, const mh_options *opts, float *tau_out)
{
   return launch<float>(ALGO_RNEA, model, B, q, qd, qdd, gravity, f_ext, opts, tau_out);
}
mh_status mh_aba_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *tau, const double gravity[3],
                     const float *f_ext, const mh_options *opts, float *qdd_out)
{
   return launch<float>(ALGO_ABA, model, B, q, qd, tau, gravity, f_ext, opts, qdd_out);
}
mh_status mh_crba_f32(mh_model_t model, int64_t B, const float *q, const mh_options *opts, float *H_out)
{
   return launch<float>(ALGO_CRBA, model, B, q, nullptr, nullptr, nullptr, nullptr, opts, H_out);
}
// tau_out = RNEA(q, qd, qdd) and qdd_out = ABA(q, qd, tau) of the same configurations, fp32 (BASELINE.json configs[4]: both per step on
// the 128-body tree).  Where the forward dynamics of a big AoS batch would go through transposed scratch copies anyway (launch<T>), the
// copies of q and qd are made ONCE and serve both algorithms -- the inverse dynamics then runs on SoA strides instead of reading its
// rows through LDS windows (468 against 742 us at B = 131 072 on that tree) -- and both results are transposed back.  Same numbers as
// mh_rnea_f32 followed by mh_aba_f32 (the two layouts of the depth-first kernels agree bit for bit).
mh_status mh_rnea_aba_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *qdd, const float *tau,
                          const double gravity[3], const float *f_ext, const mh_options *opts_in, float *tau_out, float *qdd_out)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   if (model && B > 0 && q && qd && qdd && tau && tau_out && qdd_out
       && pair_outputs_overlap(q, qd, qdd, tau, tau_out, qdd_out, (size_t)B * model->nq * sizeof(float), (size_t)B * model->nv * sizeof(float)))
      return fail(MH_ERR_INVALID_ARGUMENT, "mh_rnea_aba_f32: tau_out / qdd_out must not overlap q, qd, qdd, tau or each other (the two algorithms run concurrently)");
   const bool big = model && B >= 8192 && q && qd && qdd && tau && tau_out && qdd_out && !f_ext && model->use_dfs && model->n_locked == 0
                    && model->use_transpose < 0 && model->dfs_transpose < 0 && model->nq + model->nv >= 64 && opts.consider_coriolis
                    && opts.consider_accelerations;
   const bool shared = big && opts.layout == MH_LAYOUT_AOS;
   // ONE walk for both (round 5; mh_dfs_kernels.h: aba_dfs_kernel<.., PAIR>) unless a run-time tree split or a code object serves the model
   // (small / specialised models hardly get here: 8192 configurations of >= 64 state entries); MH_DFS_PAIR=0 keeps the two launches
   // ... and unless the two single walks, which may keep twelve waves per CU resident where the pair walk keeps eight (dfs_choose), need so
   // many fewer rounds that they win: together they take 1.25 x the pair walk's time at equal occupancy, a round of twelve 1.32 x a round
   // of eight -- which leaves the batches of nine to twelve waves per CU (196 608 configurations: 1.82 against 2.11 ms)
   bool rounds_favour_two = false;
   if (big)
   {
      const long wpc = ((B + 63) / 64 + model->cu_count - 1) / model->cu_count, r8 = (wpc + 7) / 8, r12 = (wpc + 11) / 12;
      const long two = (r12 * 132 < r8 * 100 ? r12 * 132 : r8 * 100) * 125; // (x 1e4)
      rounds_favour_two = !model->waves_per_cu_set && two < r8 * 10000;
   }
   const bool fused = big && model->use_dfs_pair && !rounds_favour_two && !(model->spec.handle && model->use_spec)
                      && !(model->split_rt.usable && (model->use_split_rt == 1 || (B + 63) / 64 <= (long)model->cu_count * 2));
   auto fused_walk = [&](mh_model *mdl, const float *sq, const float *sqd, const float *sqdd, const float *stau, float *o1, float *o2) -> mh_status {
      mh::Args<float> A{};
      A.m = dev_model<float>(mdl);
      A.B = B;
      A.q = sq, A.qd = sqd, A.in3 = sqdd, A.out = o1, A.in3b = stau, A.outb = o2;
      A.fext = nullptr, A.body_acc = nullptr, A.body_twist = nullptr, A.joint_wrench = nullptr;
      A.dt = 0.0f, A.q_next = nullptr, A.qd_next = nullptr;
      A.q_bs = 1, A.q_es = B, A.v_bs = 1, A.v_es = B, A.f_bs = 1, A.f_es = B;
      set_root_acceleration(A, opts, gravity);
      A.coriolis = 1, A.accel = 1;
      const mh_status r = launch_dfs<float>(ALGO_ABA, mdl, B, A, (hipStream_t)opts.stream, true);
      if (r == MH_OK)
         HIP_TRY(hipGetLastError());
      return r;
   };
   if (fused && opts.layout == MH_LAYOUT_SOA)
   { // SoA matrices: the walk reads and writes the caller's buffers
      mh_status st = check_common(model, B, &opts);
      if (st != MH_OK)
         return st;
      if (!gravity && !opts.use_root_acceleration)
         return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer");
      return fused_walk(model, q, qd, qdd, tau, tau_out, qdd_out);
   }
   if (!shared)
   {
      const mh_status r = mh_rnea_f32(model, B, q, qd, qdd, gravity, f_ext, &opts, tau_out);
      return r != MH_OK ? r : mh_aba_f32(model, B, q, qd, tau, gravity, f_ext, &opts, qdd_out);
   }
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (!gravity && !opts.use_root_acceleration)
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer");
   const size_t nq = model->nq, nv = model->nv, Bz = (size_t)B;
   st = ensure_bytes(model->tr_pair, Bz * (nq + 5 * nv) * sizeof(float));
   if (st != MH_OK)
      return st;
   float *t_q = (float *)model->tr_pair.ptr, *t_qd = t_q + Bz * nq, *t_qdd = t_qd + Bz * nv, *t_tau = t_qdd + Bz * nv, *t_o1 = t_tau + Bz * nv,
         *t_o2 = t_o1 + Bz * nv;
   hipStream_t stream = (hipStream_t)opts.stream;
   mh::transpose_rows<float>(q, t_q, (long)B, (long)nq, true, stream);
   mh::transpose_rows<float>(qd, t_qd, (long)B, (long)nv, true, stream);
   mh::transpose_rows<float>(qdd, t_qdd, (long)B, (long)nv, true, stream);
   mh::transpose_rows<float>(tau, t_tau, (long)B, (long)nv, true, stream);
   HIP_TRY(hipGetLastError());
   mh_options so = opts;
   so.layout = MH_LAYOUT_SOA;
   if (fused)
      st = fused_walk(model, t_q, t_qd, t_qdd, t_tau, t_o1, t_o2);
   else
   {
      st = launch<float>(ALGO_RNEA, model, B, t_q, t_qd, t_qdd, gravity, nullptr, &so, t_o1);
      if (st == MH_OK)
         st = launch<float>(ALGO_ABA, model, B, t_q, t_qd, t_tau, gravity, nullptr, &so, t_o2);
   }
   if (st != MH_OK)
      return st;
   mh::transpose_rows<float>(t_o1, tau_out, (long)B, (long)nv, false, stream);
   mh::transpose_rows<float>(t_o2, qdd_out, (long)B, (long)nv, false, stream);
   HIP_TRY(hipGetLastError());
   return MH_OK;
}

mh_status mh_rnea_f64_host(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double gravity[3],
                           const double *f_ext, const mh_options *opts, double *tau_out)
{
   return launch_host<double>(ALGO_RNEA, model, B, q, qd, qdd, nullptr, gravity, f_ext, opts, tau_out, nullptr);
}
mh_status mh_aba_f64_host(mh_model_t model, int64_t B, const double *q, const double *qd, const double *tau, const double gravity[3],
                          const double *f_ext, const mh_options *opts, double *qdd_out)
{
   return launch_host<double>(ALGO_ABA, model, B, q, qd, tau, nullptr, gravity, f_ext, opts, qdd_out, nullptr);
}
mh_status mh_rnea_aba_f64_host(mh_model_t model, int64_t B, const double *q, const double *qd, const double *qdd, const double *tau,
                               const double gravity[3], const double *f_ext, const mh_options *opts, double *tau_out, double *qdd_out)
{
   return launch_host<double>(HOST_PAIR, model, B, q, qd, qdd, tau, gravity, f_ext, opts, tau_out, qdd_out);
}
mh_status mh_rnea_f32_host(mh_model_t model, int64_t B, const float *q, const float *qd, const float *qdd, const double gravity[3],
                           const float *f_ext, const mh_options *opts, float *tau_out)
{
   return launch_host<float>(ALGO_RNEA, model, B, q, qd, qdd, nullptr, gravity, f_ext, opts, tau_out, nullptr);
}
mh_status mh_aba_f32_host(mh_model_t model, int64_t B, const float *q, const float *qd, const float *tau, const double gravity[3],
                          const float *f_ext, const mh_options *opts, float *qdd_out)
{
   return launch_host<float>(ALGO_ABA, model, B, q, qd, tau, nullptr, gravity, f_ext, opts, qdd_out, nullptr);
}
mh_status mh_crba_f32_host(mh_model_t model, int64_t B, const float *q, const mh_options *opts, float *H_out)
{
   return launch_host<float>(ALGO_CRBA, model, B, q, nullptr, nullptr, nullptr, nullptr, nullptr, opts, H_out, nullptr);
}
mh_status mh_rnea_bodies_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *qdd, const double gravity[3],
                             const float *f_ext, const mh_options *opts, float *tau_out, float *body_acc_out, float *body_twist_out)
{
   return launch<float>(ALGO_RNEA, model, B, q, qd, qdd, gravity, f_ext, opts, tau_out, nullptr, nullptr, body_acc_out, body_twist_out, true);
}
mh_status mh_aba_bodies_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *tau, const double gravity[3],
                            const float *f_ext, const mh_options *opts, float *qdd_out, float *body_acc_out, float *body_twist_out)
{
   return launch<float>(ALGO_ABA, model, B, q, qd, tau, gravity, f_ext, opts, qdd_out, nullptr, nullptr, body_acc_out, body_twist_out, true);
}
mh_status mh_aba_locked_f32(mh_model_t model, int64_t B, const float *q, const float *qd, const float *tau, const float *qdd_in,
                            const double gravity[3], const float *f_ext, const mh_options *opts, float *qdd_out, float *tau_out)
{
   if (!model)
      return fail(MH_ERR_INVALID_ARGUMENT, "model is NULL");
   if (model->n_locked == 0)
   {
      mh_status st = launch<float>(ALGO_ABA, model, B, q, qd, tau, gravity, f_ext, opts, qdd_out);
      if (st == MH_OK && tau_out && tau_out != tau && B > 0)
         HIP_TRY(hipMemcpyAsync(tau_out, tau, (size_t)B * model->nv * sizeof(float), hipMemcpyDeviceToDevice, opts ? (hipStream_t)opts->stream : nullptr));
      return st;
   }
   if (B > 0 && !qdd_in)
      return fail(MH_ERR_INVALID_ARGUMENT, "qdd_in is NULL but %d joint(s) are acceleration sources", model->n_locked);
   return launch<float>(ALGO_ABA, model, B, q, qd, tau, gravity, f_ext, opts, qdd_out, qdd_in, tau_out);
}
// ---- device memory for hosts without a HIP binding of their own (a Java shim keeps simulation state resident between steps with these)
mh_status mh_device_alloc(size_t bytes, void **ptr_out)
{
   if (!ptr_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "ptr_out is NULL");
   *ptr_out = nullptr;
   HIP_TRY(hipMalloc(ptr_out, bytes ? bytes : 1));
   return MH_OK;
}
mh_status mh_device_free(void *ptr)
{
   if (ptr)
      HIP_TRY(hipFree(ptr));
   return MH_OK;
}
mh_status mh_copy_to_device(void *dst_device, const void *src_host, size_t bytes, void *stream)
{
   if (bytes && (!dst_device || !src_host))
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL pointer");
   HIP_TRY(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
   return MH_OK;
}
mh_status mh_copy_to_host(void *dst_host, const void *src_device, size_t bytes, void *stream)
{
   if (bytes && (!dst_host || !src_device))
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL pointer");
   HIP_TRY(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
   return MH_OK;
}
mh_status mh_stream_synchronize(void *stream)
{
   HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
   return check_all_error_words(); // an inertia job that gave up waiting (mh_zv_kernels.h): reported here, not at the model's next call
}
mh_status mh_host_alloc(size_t bytes, void **ptr_out)
{
   if (!ptr_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "ptr_out is NULL");
   *ptr_out = nullptr;
   HIP_TRY(hipHostMalloc(ptr_out, bytes ? bytes : 1, hipHostMallocDefault));
   return MH_OK;
}
mh_status mh_host_free(void *ptr)
{
   if (ptr)
      HIP_TRY(hipHostFree(ptr));
   return MH_OK;
}
mh_status mh_host_register(void *ptr, size_t bytes)
{
   if (!ptr || !bytes)
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL pointer / empty range");
   HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
   return MH_OK;
}
mh_status mh_host_unregister(void *ptr)
{
   if (ptr)
      HIP_TRY(hipHostUnregister(ptr));
   return MH_OK;
}
mh_status mh_crba_f64_host(mh_model_t model, int64_t B, const double *q, const mh_options *opts, double *H_out)
{
   return launch_host<double>(ALGO_CRBA, model, B, q, nullptr, nullptr, nullptr, nullptr, nullptr, opts, H_out, nullptr);
}

// ---- HIP-event timer
struct mh_timer
{
   hipEvent_t start, stop;
};
mh_status mh_crba_coriolis_f64_host(mh_model_t model, int64_t B, const double *q, const double *qd, const mh_options *opts_in, double *H_out,
                                    double *C_out)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (B == 0)
      return MH_OK;
   if (!q || !qd || !H_out || !C_out)
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer");
   const size_t s_q = (size_t)B * model->nq, s_v = (size_t)B * model->nv, s_h = (size_t)B * model->nv * model->nv;
   st = ensure_bytes(model->stage, (s_q + s_v + 2 * s_h) * sizeof(double));
   if (st != MH_OK)
      return st;
   hipStream_t stream = (hipStream_t)opts.stream;
   double *d_q = (double *)model->stage.ptr, *d_qd = d_q + s_q, *d_H = d_qd + s_v, *d_C = d_H + s_h;
   HIP_TRY(hipMemcpyAsync(d_q, q, s_q * sizeof(double), hipMemcpyHostToDevice, stream));
   HIP_TRY(hipMemcpyAsync(d_qd, qd, s_v * sizeof(double), hipMemcpyHostToDevice, stream));
   st = coriolis_impl<double>(model, B, d_q, d_qd, &opts, d_H, d_C);
   if (st != MH_OK)
      return st;
   HIP_TRY(hipMemcpyAsync(H_out, d_H, s_h * sizeof(double), hipMemcpyDeviceToHost, stream));
   HIP_TRY(hipMemcpyAsync(C_out, d_C, s_h * sizeof(double), hipMemcpyDeviceToHost, stream));
   HIP_TRY(hipStreamSynchronize(stream));
   return MH_OK;
}
mh_status mh_centroidal_f64_host(mh_model_t model, int64_t B, const double *q, const double *qd, const double frame[12], int32_t frame_mode,
                                 const mh_options *opts_in, double *A_out, double *b_out, double *com_out)
{
   mh_options opts;
   if (opts_in)
      opts = *opts_in;
   else
      mh_options_default(&opts);
   mh_status st = check_common(model, B, &opts);
   if (st != MH_OK)
      return st;
   if (B == 0)
      return MH_OK;
   if (!q || !A_out || (b_out && !qd))
      return fail(MH_ERR_INVALID_ARGUMENT, "NULL state / output pointer (the convective term needs qd)");
   const size_t s_q = (size_t)B * model->nq, s_v = (size_t)B * model->nv, s_a = (size_t)B * 6 * model->nv, s_b = (size_t)B * 6, s_c = (size_t)B * 3;
   st = ensure_bytes(model->stage, (s_q + s_v + s_a + s_b + s_c) * sizeof(double));
   if (st != MH_OK)
      return st;
   hipStream_t stream = (hipStream_t)opts.stream;
   double *d_q = (double *)model->stage.ptr, *d_qd = d_q + s_q, *d_A = d_qd + s_v, *d_b = d_A + s_a, *d_c = d_b + s_b;
   HIP_TRY(hipMemcpyAsync(d_q, q, s_q * sizeof(double), hipMemcpyHostToDevice, stream));
   if (qd)
      HIP_TRY(hipMemcpyAsync(d_qd, qd, s_v * sizeof(double), hipMemcpyHostToDevice, stream));
   st = centroidal_impl<double>(model, B, d_q, qd ? d_qd : nullptr, frame, frame_mode, &opts, d_A, b_out ? d_b : nullptr, com_out ? d_c : nullptr);
   if (st != MH_OK)
      return st;
   HIP_TRY(hipMemcpyAsync(A_out, d_A, s_a * sizeof(double), hipMemcpyDeviceToHost, stream));
   if (b_out)
      HIP_TRY(hipMemcpyAsync(b_out, d_b, s_b * sizeof(double), hipMemcpyDeviceToHost, stream));
   if (com_out)
      HIP_TRY(hipMemcpyAsync(com_out, d_c, s_c * sizeof(double), hipMemcpyDeviceToHost, stream));
   HIP_TRY(hipStreamSynchronize(stream));
   return MH_OK;
}

mh_status mh_timer_create(mh_timer_t *out)
{
   if (!out)
      return fail(MH_ERR_INVALID_ARGUMENT, "timer_out is NULL");
   mh_timer *t = new mh_timer();
   // timing only: without the system-scope fence (cache write-back) a default event performs when it becomes recorded -- measured on the
   // headline's regions of 20 steps: 8 us per region with default events (profiles/r04_region_overhead.txt)
   const unsigned flags = getenv("MH_TIMER_DEFAULT_EVENTS") ? hipEventDefault : hipEventDisableSystemFence;
   if (hipEventCreateWithFlags(&t->start, flags) != hipSuccess || hipEventCreateWithFlags(&t->stop, flags) != hipSuccess)
   {
      delete t;
      return fail(MH_ERR_NO_DEVICE, "cannot create HIP events");
   }
   *out = t;
   return MH_OK;
}
void mh_timer_destroy(mh_timer_t t)
{
   if (!t)
      return;
   (void)hipEventDestroy(t->start);
   (void)hipEventDestroy(t->stop);
   delete t;
}
mh_status mh_timer_start(mh_timer_t t, void *stream)
{
   if (!t)
      return fail(MH_ERR_INVALID_ARGUMENT, "timer is NULL");
   HIP_TRY(hipEventRecord(t->start, (hipStream_t)stream));
   return MH_OK;
}
mh_status mh_timer_stop(mh_timer_t t, void *stream)
{
   if (!t)
      return fail(MH_ERR_INVALID_ARGUMENT, "timer is NULL");
   HIP_TRY(hipEventRecord(t->stop, (hipStream_t)stream));
   return MH_OK;
}
mh_status mh_timer_elapsed_ms(mh_timer_t t, float *ms)
{
   if (!t || !ms)
      return fail(MH_ERR_INVALID_ARGUMENT, "timer / ms_out is NULL");
   HIP_TRY(hipEventSynchronize(t->stop));
   HIP_TRY(hipEventElapsedTime(ms, t->start, t->stop));
   return MH_OK;
}

} // extern "C"

// =================================================================================================== create-time self-check
// A topology-specialised code object is machine-generated straight-line code at the edge of the register file (open issues:
// one whole-tree ABA build of a 25-body tree returned wrong numbers).  So a freshly loaded object is not trusted: 197 seeded
// configurations (three full groups of 64 and a ragged one) go through every call the dispatcher can route to it -- RNEA, ABA, the fused
// pair, the fused simulation step, CRBA, the per-body variants, Coriolis and centroidal quantities, AoS and SoA, with the real CU count
// (small-batch plans) and with a pretended single CU (device-filling plans: global hand-over store, three-wave RNEA build) -- and through
// the run-time-topology kernels.  On disagreement the object is dropped and mh_model_kernel_variant says why; the model then runs on the
// run-time-topology kernels.  MH_SPEC_SELFCHECK=0 skips the check.
namespace
{
struct Lcg
{
   unsigned long long s;
   double u(double lo, double hi)
   {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      return lo + (hi - lo) * (double)(s >> 11) / 9007199254740992.0;
   }
};
enum CheckCase
{
   CK_RNEA,
   CK_ABA,
   CK_CRBA,
   CK_FUSED,
   CK_STEP,
   CK_BODIES_RNEA,
   CK_BODIES_ABA,
   CK_CORIOLIS,
   CK_CENTROIDAL,
   CK_COUNT
};
// fills the check's output buffer with the NaN pattern 0xFF..FF ON THE LAUNCH STREAM: the synchronous hipMemset was observed to be
// unordered against null-stream kernels of this process (results wiped after the kernel had written them, or never wiped), which showed up
// as intermittent refusals / acceptances
__global__ void __launch_bounds__(256) fill_nan_kernel(unsigned long long *p, size_t n)
{
   for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
      p[i] = ~0ull;
}
const char *const kCheckNames[CK_COUNT] = {"RNEA", "ABA", "CRBA", "fused RNEA+ABA", "ABA + integration step", "RNEA with per-body outputs",
                                           "ABA with per-body outputs", "mass + Coriolis matrix", "centroidal momentum"};
} // namespace

static void self_check_spec(mh_model *m)
{
   const int64_t B = 64 * 3 + 5;
   const size_t nq = m->nq, nv = m->nv, n = m->n;
   if (nv == 0)
      return;
   Lcg rng{0x9E3779B97F4A7C15ull ^ (unsigned long long)n};
   std::vector<double> q((size_t)B * nq, 0.0), qd((size_t)B * nv), qdd((size_t)B * nv), tau((size_t)B * nv);
   for (int64_t b = 0; b < B; b++)
      for (int e = 0; e < m->n; e++)
      {
         const int *mi = &m->meta[(size_t)e * mh::MI_STRIDE];
         const int t = mi[mh::MI_TYPE], *ci = &m->cfg_map[mi[mh::MI_CFG]];
         double *row = &q[(size_t)b * nq];
         if (t == MH_JOINT_REVOLUTE)
            row[ci[0]] = rng.u(-3.14159, 3.14159);
         else if (t == MH_JOINT_PRISMATIC)
            row[ci[0]] = rng.u(-1, 1);
         else if (t == MH_JOINT_SIXDOF || t == MH_JOINT_SPHERICAL)
         {
            double qt[4], nrm = 0;
            for (int k = 0; k < 4; k++)
               qt[k] = rng.u(-1, 1), nrm += qt[k] * qt[k];
            nrm = std::sqrt(nrm) + 1e-300;
            for (int k = 0; k < 4; k++)
               row[ci[k]] = qt[k] / nrm;
            for (int k = 4; k < joint_ncfg(t); k++)
               row[ci[k]] = rng.u(-1, 1);
         }
         else if (t == MH_JOINT_PLANAR)
            row[ci[0]] = rng.u(-3.14159, 3.14159), row[ci[1]] = rng.u(-1, 1), row[ci[2]] = rng.u(-1, 1);
      }
   for (size_t k = 0; k < (size_t)B * nv; k++)
      qd[k] = rng.u(-1, 1), qdd[k] = rng.u(-1, 1), tau[k] = rng.u(-1, 1);
   auto transposed = [&](const std::vector<double> &a, size_t cols) {
      std::vector<double> t(a.size());
      for (int64_t b = 0; b < B; b++)
         for (size_t c = 0; c < cols; c++)
            t[c * (size_t)B + (size_t)b] = a[(size_t)b * cols + c];
      return t;
   };
   const size_t out_doubles = (size_t)B * std::max<size_t>({2 * nv * nv, 6 * nv + 9, 2 * nv + 19 * n}) + 64;
   const size_t in_doubles = (size_t)B * (nq + 3 * nv);
   double *d_all = nullptr;
   if (hipMalloc((void **)&d_all, (2 * in_doubles + out_doubles) * sizeof(double)) != hipSuccess)
      return; // cannot check: keep the object (the allocation failure will surface in the first compute call anyway)
   double *d_in[2] = {d_all, d_all + in_doubles}, *d_out = d_all + 2 * in_doubles;
   for (int L = 0; L < 2; L++)
   {
      const std::vector<double> *src[4] = {&q, &qd, &qdd, &tau};
      const size_t cols[4] = {nq, nv, nv, nv};
      size_t ofs = 0;
      for (int k = 0; k < 4; k++)
      {
         const std::vector<double> t = L == 0 ? *src[k] : transposed(*src[k], cols[k]);
         (void)hipMemcpy(d_in[L] + ofs, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice);
         ofs += (size_t)B * cols[k];
      }
   }
   const double gravity[3] = {0.3, -0.2, -9.81};
   auto run = [&](int what, int L, size_t &used) -> mh_status {
      mh_options o;
      mh_options_default(&o);
      o.layout = L == 0 ? MH_LAYOUT_AOS : MH_LAYOUT_SOA;
      // a rotating, accelerating base: every plan starts its outward sweep from the full 6-D root acceleration
      o.use_root_acceleration = 1;
      const double root_acc[6] = {0.11, -0.07, 0.05, -0.3, 0.2, 9.81};
      for (int k = 0; k < 6; k++)
         o.root_acceleration[k] = root_acc[k];
      const double *dq = d_in[L], *dqd = dq + (size_t)B * nq, *dqdd = dqd + (size_t)B * nv, *dtau = dqdd + (size_t)B * nv;
      double *o1 = d_out, *o2 = o1 + (size_t)B * nv, *o3 = o2 + (size_t)B * std::max(nq, 6 * n);
      hipLaunchKernelGGL(fill_nan_kernel, dim3(64), dim3(256), 0, (hipStream_t) nullptr, (unsigned long long *)d_out, out_doubles);
      switch (what)
      {
         case CK_RNEA: used = (size_t)B * nv; return mh_rnea_f64(m, B, dq, dqd, dqdd, gravity, nullptr, &o, o1);
         case CK_ABA: used = (size_t)B * nv; return mh_aba_f64(m, B, dq, dqd, dtau, gravity, nullptr, &o, o1);
         case CK_CRBA: used = (size_t)B * nv * nv; return mh_crba_f64(m, B, dq, &o, o1);
         case CK_FUSED: used = 2 * (size_t)B * nv; return mh_rnea_aba_f64(m, B, dq, dqd, dqdd, dtau, gravity, nullptr, &o, o1, o2);
         case CK_STEP:
            used = (size_t)B * (nv + std::max(nq, 6 * n) + nv);
            return mh_aba_integrate_f64(m, B, 1.0e-3, dq, dqd, dtau, gravity, nullptr, &o, o1, o2, o3);
         case CK_BODIES_RNEA:
            used = (size_t)B * (nv + std::max(nq, 6 * n) + 6 * n);
            return mh_rnea_bodies_f64(m, B, dq, dqd, dqdd, gravity, nullptr, &o, o1, o2, o3);
         case CK_BODIES_ABA:
            used = (size_t)B * (nv + std::max(nq, 6 * n) + 6 * n);
            return mh_aba_bodies_f64(m, B, dq, dqd, dtau, gravity, nullptr, &o, o1, o2, o3);
         case CK_CORIOLIS: used = 2 * (size_t)B * nv * nv; return mh_crba_coriolis_f64(m, B, dq, dqd, &o, d_out, d_out + (size_t)B * nv * nv);
         case CK_CENTROIDAL:
            used = (size_t)B * (6 * nv + 9);
            return mh_centroidal_f64(m, B, dq, dqd, nullptr, MH_CENTROIDAL_FRAME_AT_COM, &o, d_out, d_out + (size_t)B * 6 * nv,
                                     d_out + (size_t)B * (6 * nv + 6));
      }
      return MH_OK;
   };
   const int real_cus = m->cu_count;
   static const char *const kPlanNames[5] = {"small-batch", "device-filling", "tree-split", "device-filling, two launches", "device-filling, one job"};
   const bool verbose = getenv("MH_SPEC_SELFCHECK_VERBOSE") != nullptr;
   std::vector<unsigned long long> ref, got; // raw words: see nan_word
   std::string failure;
   for (int what = 0; what < CK_COUNT && failure.empty(); what++)
      for (int L = 0; L < 2 && failure.empty(); L++)
      {
         if ((what == CK_FUSED || what == CK_STEP) && L == 1)
            continue; // AoS-only entry points
         size_t used = 0;
         m->use_spec = 0;
         mh_status st = run(what, L, used);
         (void)hipDeviceSynchronize();
         m->use_spec = 1;
         if (st != MH_OK)
            continue; // the run-time-topology kernels cannot serve this call either: nothing to compare
         ref.resize(used);
         (void)hipMemcpy(ref.data(), d_out, used * sizeof(double), hipMemcpyDeviceToHost);
         // plan 0: the real CU count; 1: a pretended single CU (device-filling plans); 2: the bias-split forward dynamics switched off
         // (the tree-split kernels it replaced still serve SoA calls, simulation steps and models with acceleration sources)
         // 3: a pretended single CU with the fused forward dynamics switched off (the one-job kernel's device-filling plan, plan 4, is what
         // serves SoA calls and simulation steps at those sizes)
         // 4: ... with the fused one-launch form switched off as well (plan 1 takes the fused kernel where the code object has it, plan 3 then
         // the two launches, plan 4 the one-job kernel)
         // (inverse dynamics: plan 1 takes the loop that requests rows ahead, plan 4 the tree-split kernel's own device-filling loop)
         const int zv_was = m->use_zv, zvb_was = m->use_zvb, zvf_was = m->use_zvf, ahead_was = m->use_rnea_ahead;
         const bool zv_plan = (what == CK_ABA || what == CK_FUSED) && L == 0 && zv_was && zv_ok(m, B, false, what == CK_FUSED ? 3 : 2);
         m->cu_count = 1;
         const bool fd_aos = (what == CK_ABA || what == CK_FUSED) && L == 0;
         const bool zvf_plan = fd_aos && zvf_was && zvf_ok(m, B, false), zvb_plan = fd_aos && zvb_was && zvb_ok(m, B, false);
         const bool ahead_plan = (what == CK_RNEA || what == CK_FUSED) && L == 0 && ahead_was && rnea_ahead_ok(m, B, false);
         m->cu_count = real_cus;
         for (int plan = 0; plan < 5 && failure.empty(); plan++)
         {
            if ((plan == 2 && !zv_plan) || (plan == 3 && !(zvf_plan && zvb_plan)) || (plan == 4 && !(zvf_plan || zvb_plan || ahead_plan)))
               continue;
            const int pretend = plan == 1 || plan >= 3 ? 1 : 0;
            m->cu_count = pretend ? 1 : real_cus;
            m->use_zv = plan == 2 ? 0 : zv_was;
            m->use_zvf = plan >= 3 ? 0 : zvf_was;
            m->use_zvb = plan == 4 ? 0 : zvb_was;
            m->use_rnea_ahead = plan >= 3 ? 0 : ahead_was;
            st = run(what, L, used);
            const hipError_t sync = hipDeviceSynchronize();
            m->cu_count = real_cus;
            m->use_zv = zv_was;
            m->use_zvb = zvb_was;
            m->use_zvf = zvf_was;
            m->use_rnea_ahead = ahead_was;
            if (st == MH_OK)
               st = zv_check_error(m);
            char buf[320];
            if (st != MH_OK || sync != hipSuccess)
            {
               snprintf(buf, sizeof buf, "%s (%s, %s plan) failed: %s", kCheckNames[what], L ? "SoA" : "AoS",
                        kPlanNames[plan], st != MH_OK ? g_err : hipGetErrorString(sync));
               failure = buf;
               break;
            }
            got.resize(used);
            (void)hipMemcpy(got.data(), d_out, used * sizeof(double), hipMemcpyDeviceToHost);
            double scale = 1.0, err = 0.0;
            bool nan_mismatch = false;
            size_t first_bad = 0;
            bool bad_is_unwritten = false;
            for (size_t k = 0; k < used; k++)
            {
               // bit tests, not x != x: this translation unit is built with -ffinite-math-only, under which the compiler folds NaN tests away
               const bool rn = nan_word(ref[k]), gn = nan_word(got[k]);
               if (rn != gn)
               {
                  if (!nan_mismatch)
                     first_bad = k, bad_is_unwritten = gn;
                  nan_mismatch = true;
               }
               else if (!rn)
               {
                  double rv, gv;
                  std::memcpy(&rv, &ref[k], sizeof rv), std::memcpy(&gv, &got[k], sizeof gv);
                  scale = std::max(scale, std::fabs(rv)), err = std::max(err, std::fabs(rv - gv));
               }
            }
            // forward dynamics divides by joint-space inertias: rounding differences between two exact evaluation orders are amplified by
            // their conditioning (1e-8 is what the parity tests grant ill-conditioned random trees); everything else is held to 1e-10
            const double tol = (what == CK_ABA || what == CK_FUSED || what == CK_STEP || what == CK_BODIES_ABA) ? 1.0e-8 : 1.0e-10;
            if (verbose)
               fprintf(stderr, "[mh self-check] %-28s %s %-14s  |diff| %.3e  |ref| %.3e  unwritten-mismatch %d (word %zu)\n", kCheckNames[what], L ? "SoA" : "AoS",
                       kPlanNames[plan], err, scale, (int)nan_mismatch, first_bad);
            if (nan_mismatch || err > tol * scale)
            {
               char where[96] = "";
               if (nan_mismatch)
                  snprintf(where, sizeof where, ", output word %zu %s", first_bad,
                           bad_is_unwritten ? "left unwritten" : "written although the run-time-topology kernels do not write it");
               snprintf(buf, sizeof buf, "%s (%s, %s plan) differs from the run-time-topology kernels by %.3e (|ref| <= %.3e%s)", kCheckNames[what],
                        L ? "SoA" : "AoS", kPlanNames[plan], err, scale, where);
               failure = buf;
            }
         }
      }
   (void)hipFree(d_all);
   if (!failure.empty())
   {
      dlclose(m->spec.handle);
      m->spec = SpecLib{};
      m->variant = "generic (code object topo:" + m->topo_key + " refused by the create-time self-check: " + failure + ")";
   }
}
