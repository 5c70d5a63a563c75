// mh_spec_kernels.h -- topology-specialised RNEA / ABA kernels for gfx950.
//
// Same arithmetic and same lane = configuration mapping as the generic kernels (mh_kernels.h), but the kinematic tree
// (parents and joint kinds, in engine order) is a compile-time constant TP.  The tree walk is then a compile-time
// recursion: every loop over bodies is unrolled, every joint-kind branch is resolved, the per-joint constants sit at
// fixed addresses (wave-uniform scalar loads straight into SGPR operands), and the intermediates of the depth-first
// walk live in registers instead of the per-lane global workspace.
//
// Memory plan of one wave (= one workgroup of 64 lanes = 64 configurations):
//   * IO_LDS: the wave's 64 rows of q, qd and qdd|tau (contiguous in the AoS [B][n] matrices Mecano's layout implies) are
//     copied once, coalesced, into LDS; each lane then reads its own row from LDS (immediate offsets when the index maps
//     are the identity).  Results are written in place over the qdd|tau rows and copied out coalesced.
//   * ABA's hand-over between the inward and the outward sweep (U/D (6), u/D, cos, sin per revolute body) goes through a
//     per-lane store: LDS (slot-major, 64 lanes per slot) for small batches, the global workspace for large ones.
//
// One code object is built per topology (mecano_amd/build.py: hipcc -DMH_TOPO_...); model parameters (poses, inertias,
// index maps) stay run-time data, so every robot with the same tree shape and joint kinds shares it.
#pragma once
#include "mh_kernels.h"

// keeps the instruction scheduler from interleaving the steps of different bodies (which inflates live ranges until
// the 512-VGPR budget spills): nothing may be moved across this point
#define MH_BODY_FENCE() __builtin_amdgcn_sched_barrier(0)

namespace mh
{
// TP must provide: static constexpr int N; static constexpr int parent[N]; static constexpr int type[N];
template <class TP>
struct Tree
{
   static constexpr int N = TP::N;
   static constexpr int ndof(int j) { return TP::type[j] == JT_SIXDOF ? 6 : (TP::type[j] == JT_FIXED ? 0 : 1); }
   static constexpr int ncfg(int j) { return TP::type[j] == JT_SIXDOF ? 7 : (TP::type[j] == JT_FIXED ? 0 : 1); }
   static constexpr int dof_ofs(int j)
   { // offset of joint j in the engine-order index maps
      int s = 0;
      for (int i = 0; i < j; i++)
         s += ndof(i);
      return s;
   }
   static constexpr int cfg_ofs(int j)
   {
      int s = 0;
      for (int i = 0; i < j; i++)
         s += ncfg(i);
      return s;
   }
   static constexpr int total_dofs() { return dof_ofs(N); }
   // joint that owns DoF d of the engine-order velocity vector (run-time look-up in the fused integration step)
   struct DofTable
   {
      short joint[dof_ofs(N) > 0 ? dof_ofs(N) : 1] = {};
      short d0[N] = {}, c0[N] = {}, type[N] = {}; // per joint: first DoF, first configuration entry, kind
   };
   static constexpr DofTable make_dof_table()
   {
      DofTable t;
      for (int j = 0; j < N; j++)
      {
         t.d0[j] = (short)dof_ofs(j), t.c0[j] = (short)cfg_ofs(j), t.type[j] = (short)TP::type[j];
         for (int k = 0; k < ndof(j); k++)
            t.joint[dof_ofs(j) + k] = (short)j;
      }
      return t;
   }
   static constexpr int total_cfgs() { return cfg_ofs(N); }
   // ordinal of joint j among the revolute joints (rows 2 r and 2 r + 1 of the (cos, sin) scratch of the two-launch forward dynamics, mh_zv_kernels.h)
   static constexpr int rev_index(int j)
   {
      int r = 0;
      for (int i = 0; i < j; i++)
         r += TP::type[i] == JT_REVOLUTE ? 1 : 0;
      return r;
   }
   static constexpr int n_revolute() { return rev_index(N); }
   // ABA hand-over slots: revolute 9 (U/D, u/D, cos, sin), prismatic 7, sixdof 6 (IA^-1 u), fixed 0
   static constexpr int aba_slots_of(int j)
   {
      return TP::type[j] == JT_REVOLUTE ? 9 : (TP::type[j] == JT_PRISMATIC ? 7 : (TP::type[j] == JT_SIXDOF ? 6 : 0));
   }
   // bias-split forward dynamics (mh_zv_kernels.h): revolute 9 (U/D, 1/D, cos, sin), prismatic 7, sixdof 21 (LDL^T factor of IA), fixed 0.
   // What the bias fold leaves for the outward sweep (u/D; IA^-1 u) overwrites 1/D / the factor where one wave owns the body (limbs, in
   // registers); trunk bodies are folded by every wave at its own pace through LDS, so there it gets slots of its own (shared = false).
   static constexpr int zv_slots_of(int j, bool shared)
   {
      return TP::type[j] == JT_REVOLUTE ? (shared ? 9 : 10) : (TP::type[j] == JT_PRISMATIC ? (shared ? 7 : 8) : (TP::type[j] == JT_SIXDOF ? (shared ? 21 : 27) : 0));
   }
   static constexpr int zv_result_slot(int j, bool shared)
   { // first slot of u/D (1-DoF) or IA^-1 u (6-DoF)
      return TP::type[j] == JT_REVOLUTE ? (shared ? 6 : 9) : (TP::type[j] == JT_PRISMATIC ? (shared ? 6 : 7) : (shared ? 0 : 21));
   }
   static constexpr int aba_slot(int j)
   {
      int s = 0;
      for (int i = 0; i < j; i++)
         s += aba_slots_of(i);
      return s;
   }
   // The frame after a 1-DoF joint may slide along and turn about its own axis without changing the joint; mh_model_create uses that
   // freedom so that the origin of the joint's FIRST child (engine order: the next joint) lies on its x axis: p_b = (a, 0, 0) exactly.
   static constexpr bool p_aligned(int j)
   {
      return j > 0 && TP::parent[j] == j - 1 && (TP::type[j - 1] == JT_REVOLUTE || TP::type[j - 1] == JT_PRISMATIC);
   }
   static constexpr int n_children(int j)
   {
      int c = 0;
      for (int i = 0; i < N; i++)
         c += TP::parent[i] == j ? 1 : 0;
      return c;
   }
   static constexpr int depth(int j)
   { // number of ancestors (a root has depth 0)
      int d = 0;
      for (int p = TP::parent[j]; p >= 0; p = TP::parent[p])
         d++;
      return d;
   }
   static constexpr int ancestor_at_depth(int j, int d)
   {
      int a = j;
      for (int k = depth(j); k > d; k--)
         a = TP::parent[a];
      return a;
   }
   static constexpr int height(int j)
   { // longest chain of joints below j (a leaf has height 0); children have larger indices than their parent
      int h = 0;
      for (int i = N - 1; i > j; i--)
         if (TP::parent[i] == j)
         {
            const int hc = 1 + height(i);
            h = hc > h ? hc : h;
         }
      return h;
   }
   // k-th child of j (j = -1: k-th root), TALLEST SUBTREE FIRST (ties: lower index first).  The walk descends into the
   // deepest subtree while no partial sum of the siblings is alive yet; the shallower siblings are walked with the
   // accumulated contribution (27 scalars for ABA) held in registers.  This is what keeps the 25-body ABA under 512 VGPRs.
   static constexpr int child(int j, int k)
   {
      int taken[N] = {};
      int pick = -1;
      for (int round = 0; round <= k; round++)
      {
         pick = -1;
         int best = -1;
         for (int i = 0; i < N; i++)
            if (TP::parent[i] == j && !taken[i])
            {
               const int h = height(i);
               if (h > best)
               {
                  best = h;
                  pick = i;
               }
            }
         if (pick >= 0)
            taken[pick] = 1;
      }
      return pick;
   }
};

#ifndef MH_ZVF_MAIL
#define MH_ZVF_MAIL 1 // 0: the fused kernel's inverse dynamics walks every limb on its forward-dynamics owner, as before (A/B measurements)
#endif
#ifndef MH_ZVF_MAIL_COUNT_WALKS
#define MH_ZVF_MAIL_COUNT_WALKS 1 // trunk wrench duties under the mailed owners: by bodies + trunk walks + duties (1) or by the limbs' bodies alone (0)
#endif
// Compile-time partition of the tree for the tree-split kernels: 4 waves of a workgroup walk the SAME 64 configurations.
//   trunk = every body with two or more child subtrees, and all its ancestors  (humanoid: pelvis, spine 1-3)
//   limbs = the chains hanging off the trunk with at least MIN_LIMB bodies       (legs, arms, neck)
// Every wave walks the trunk (redundantly: it would idle otherwise); each limb is walked by one wave only (greedy balance).
template <class TP, int WAVES = 4, int MIN_LIMB = 1>
struct Split
{
   using TR = Tree<TP>;
   static constexpr int N = TP::N;
   struct Plan
   {
      bool trunk[N] = {};
      bool limb_root[N] = {};
      int limb_of[N] = {};    // limb index of a limb body
      int root_of[N] = {};    // body index of limb k's root
      int size_of[N] = {};    // bodies in limb k
      int owner[N] = {};      // wave that walks limb k
      int trunk_slot[N + 1] = {};
      int reg_slot[N] = {};
      int n_limbs = 0, reg_slots = 0;
      // the same placement for the bias-split forward dynamics (mh_zv_kernels.h): a 6-DoF joint keeps its LDL^T factor (21) there
      int zv_trunk_slot[N + 1] = {};
      int zv_reg_slot[N] = {};
      int zv_reg_slots = 0;
      bool usable = false;
      // RNEA / CRBA keep the plain greedy assignment; `owner` is the ABA's, phase-aware when the trunk is staged
      int owner_plain[N] = {};
      // RNEA: the Newton-Euler wrench of every trunk body is formed during the limb phase, by the wave that walks through the body first
      // on its way to a limb (f_limb: that limb), and parked in LDS (8 slots per trunk body: wrench, cos, sin)
      int trunk_rank[N] = {};
      int f_limb[N] = {};
      int f_limb_aba[N] = {}; // the same under the ABA's limb owners (the fused bias + inertia kernel walks its inverse dynamics with those)
      // ... and with ONE one-body limb given to another wave for the inverse dynamics only (mailed < 0: none): see mail_one_limb()
      int owner_f[N] = {};
      int f_limb_f[N] = {};
      int mailed = -1;
      int n_trunk = 0;
      // ---- staged trunk (ABA): see make_stages()
      bool staged = false;
      int root = -1;          // the root trunk body R
      bool late[N] = {};      // limb k hangs off R itself: its owner can still be walking it while a sub-trunk is folded
      int n_sub = 0;
      int sub_top[N] = {};    // trunk children of R
      int sub_owner[N] = {};  // [body] wave that folds the sub-trunk rooted at that body, between the two barriers
      int sub_slot[N] = {};   // [body] exchange slot (an early limb's, consumed by then) its hand-up travels in
      int cut_body[WAVES] = {}; // [wave] body of its late limb after whose children the first barrier sits, -1: explicit barrier
      int cut_limb[WAVES] = {}; // [wave] that late limb, -1: none
   };
   // Longest chain of TRUNK bodies from j downwards (j included).
   static constexpr int trunk_len(const Plan &P, int j)
   {
      int best = 0;
      for (int i = j + 1; i < N; i++)
         if (TP::parent[i] == j && P.trunk[i])
         {
            const int l = trunk_len(P, i);
            best = l > best ? l : best;
         }
      return 1 + best;
   }
   // Staged trunk.  In the plain scheme every wave walks its limbs, ONE barrier, then every wave folds the whole trunk inward
   // (replicated): the serial chain is (longest limb) + (trunk depth) heavy steps -- 6 + 4 for the humanoid.  But only the root R needs
   // every limb; the sub-trunks below it (spine 1-3 under the pelvis) need just the limbs hanging off them (arms, neck).  So:
   //    phase 1  every wave: its early limbs; owners of late limbs (legs, hanging off R): the lower part of the limb
   //    -------- barrier 1 (for a late-limb owner it sits INSIDE the limb's recursion, between two body steps)
   //    phase 2  late-limb owners: the upper part of the limb; one wave without a late limb: the sub-trunk, inward, handing its
   //             articulated inertia up through an exchange slot like a limb does
   //    -------- barrier 2
   //    phase 3  every wave: R alone (children read from the exchange area), then the outward sweep as before
   // Serial chain: max(early limbs, lower parts) + max(upper parts, sub-trunk) + 1 = 4 + 3 + 1 for the humanoid, and the trunk bodies
   // below R are folded once instead of four times.  Early limbs are balanced over the waves by their PHASE-1 load.
   static constexpr void make_stages(Plan &P)
   {
      int nroot = 0;
      for (int j = 0; j < N; j++)
         if (TP::parent[j] < 0)
         {
            nroot++;
            P.root = j;
         }
      for (int w = 0; w < WAVES; w++)
         P.cut_body[w] = -1, P.cut_limb[w] = -1;
      if (!P.usable || nroot != 1 || !P.trunk[P.root])
         return;
      const int R = P.root;
      int sub_depth = 0, n_late = 0;
      for (int j = 0; j < N; j++)
         if (TP::parent[j] == R && P.trunk[j])
         {
            P.sub_top[P.n_sub++] = j;
            const int l = trunk_len(P, j);
            sub_depth = l > sub_depth ? l : sub_depth;
         }
      for (int k = 0; k < P.n_limbs; k++)
      {
         P.late[k] = TP::parent[P.root_of[k]] == R;
         n_late += P.late[k] ? 1 : 0;
      }
      if (P.n_sub < 1 || n_late < 1 || n_late > WAVES - 1)
         return;
      // late limbs: one per wave, largest first
      int load1[WAVES] = {}; // phase-1 load
      bool has_late[WAVES] = {};
      bool done[N] = {};
      int next_wave = 0;
      for (int round = 0; round < n_late; round++)
      {
         int best = -1, bs = -1;
         for (int k = 0; k < P.n_limbs; k++)
            if (P.late[k] && !done[k] && P.size_of[k] > bs)
               bs = P.size_of[k], best = k;
         const int w = next_wave++;
         done[best] = true;
         P.owner[best] = w;
         has_late[w] = true;
         // the cut: `low` bodies of the limb's tallest chain below it, so that the upper part is about as long as the sub-trunk
         const int len = 1 + TR::height(P.root_of[best]);
         if (len >= 2)
         {
            int low = len - sub_depth;
            low = low < 1 ? 1 : (low > len - 1 ? len - 1 : low);
            int j = P.root_of[best];
            for (int step = 0; step < len - low - 1; step++)
               j = TR::child(j, 0);
            P.cut_body[w] = j;
            P.cut_limb[w] = best;
            load1[w] += low;
         }
      }
      // early limbs: decreasing size, each to the wave with the least phase-1 load
      for (;;)
      {
         int best = -1, bs = -1;
         for (int k = 0; k < P.n_limbs; k++)
            if (!P.late[k] && !done[k] && P.size_of[k] > bs)
               bs = P.size_of[k], best = k;
         if (best < 0)
            break;
         int w = 0;
         for (int i = 1; i < WAVES; i++)
            if (load1[i] < load1[w])
               w = i;
         done[best] = true;
         P.owner[best] = w;
         load1[w] += bs;
      }
      // sub-trunks: waves without a late limb, least loaded first
      int extra[WAVES] = {};
      for (int i = 0; i < P.n_sub; i++)
      {
         int w = -1;
         for (int c = 0; c < WAVES; c++)
            if (!has_late[c] && (w < 0 || load1[c] + extra[c] < load1[w] + extra[w]))
               w = c;
         const int s = P.sub_top[i];
         P.sub_owner[s] = w;
         extra[w] += trunk_len(P, s);
         // its hand-up travels in the exchange slot of the first early limb of its own subtree (read by then, by this very wave)
         P.sub_slot[s] = -1;
         for (int k = 0; k < P.n_limbs && P.sub_slot[s] < 0; k++)
            if (!P.late[k])
               for (int a = TP::parent[P.root_of[k]]; a >= 0; a = TP::parent[a])
                  if (a == s)
                  {
                     P.sub_slot[s] = k;
                     break;
                  }
         if (P.sub_slot[s] < 0)
            return; // a sub-trunk without a limb of its own cannot happen (trunk = branching), but stay plain if it does
      }
      P.staged = true;
   }
   // Which limbs really walk the trunk down to their parent (a limb that follows, on the same wave, one with the same parent reuses that
   // walk), and which of them forms the wrench of trunk body j on its way (f[j] = that limb): of the waves that pass j, the one with the
   // least work so far.  Work = the bodies of a wave's limbs + the trunk bodies it walks down to reach them + the wrenches it has already
   // been given (count_walks; without it: the limbs' bodies alone).  (Counting the limbs' bodies only gave all four trunk wrenches of the
   // humanoid to one arm's wave -- four bodies, the "least loaded" -- which also walks four trunk bodies down to its shoulder and was the last
   // wave at the limb barrier of the inverse dynamics by 0.7 us: bias job of the headline 8.23 -> 8.00 us, step 15.7 -> 15.4.)
   static constexpr void assign_wrench_duties(Plan &P, const int (&owner)[N], int (&f)[N], bool count_walks = true)
   {
      bool walks[N] = {};
      int load[WAVES] = {}, last_parent[WAVES] = {};
      bool passes[WAVES][N] = {};
      for (int w = 0; w < WAVES; w++)
         last_parent[w] = -2;
      for (int k = 0; k < P.n_limbs; k++)
      {
         const int w = owner[k], par = TP::parent[P.root_of[k]];
         walks[k] = par != last_parent[w] && par >= 0;
         last_parent[w] = par;
         load[w] += P.size_of[k];
         if (walks[k])
            for (int a = par; a >= 0; a = TP::parent[a])
               if (!passes[w][a])
                  passes[w][a] = true, load[w] += count_walks ? 1 : 0;
      }
      for (int j = 0; j < N; j++)
      {
         f[j] = -1;
         if (!P.trunk[j])
            continue;
         for (int k = 0; k < P.n_limbs; k++)
         {
            if (!walks[k])
               continue;
            bool through = false;
            for (int a = TP::parent[P.root_of[k]]; a >= 0; a = TP::parent[a])
               through = through || a == j;
            if (through && (f[j] < 0 || load[owner[k]] < load[owner[f[j]]]))
               f[j] = k;
         }
         if (f[j] >= 0 && count_walks)
            load[owner[f[j]]]++;
      }
   }
   // Bodies a wave steps through in the limb phase of an inverse dynamics under `owner`: its limbs' bodies + the trunk bodies it walks
   // down to reach them (once per distinct parent chain).
   static constexpr void rnea_loads(const Plan &P, const int (&owner)[N], int (&load)[WAVES])
   {
      bool passes[WAVES][N] = {};
      for (int w = 0; w < WAVES; w++)
         load[w] = 0;
      for (int k = 0; k < P.n_limbs; k++)
      {
         const int w = owner[k];
         load[w] += P.size_of[k];
         for (int a = TP::parent[P.root_of[k]]; a >= 0; a = TP::parent[a])
            if (!passes[w][a])
               passes[w][a] = true, load[w]++;
      }
   }
   // The fused bias + inertia kernel (mh_zv_kernels.h) walks its inverse dynamics with the forward dynamics' limb owners, so that a limb
   // joint's (cos, sin) and tau - h stay in the registers of the wave that needs them next.  On the humanoid that puts the head (one body
   // on the chest) beside a leg: that wave steps through leg + root + the three trunk bodies down to the chest + head = 10 bodies while a
   // leg alone is 7 and an arm 8, and the other three wait 1.1-1.9 us of a 20 us group for it (profiles/r05_zvf_phase_stamps.txt).  A
   // ONE-body revolute limb may therefore be walked by another wave in that phase: its three values travel through three LDS slots
   // (ZvfStore: slot kinds per slot).  Chosen: the move that lowers the largest load most; ties keep the forward dynamics' owner.
   static constexpr void mail_one_limb(Plan &P)
   {
      for (int k = 0; k < P.n_limbs; k++)
         P.owner_f[k] = P.owner[k];
      P.mailed = -1;
      int load[WAVES] = {};
      rnea_loads(P, P.owner, load);
      int best = 0;
      for (int w = 0; w < WAVES; w++)
         best = load[w] > best ? load[w] : best;
      int bk = -1, bw = -1;
      for (int k = 0; k < P.n_limbs; k++)
      {
         if (P.size_of[k] != 1 || TP::type[P.root_of[k]] != JT_REVOLUTE)
            continue;
         for (int w = 0; w < WAVES; w++)
         {
            if (w == P.owner[k])
               continue;
            int trial[N] = {};
            for (int i = 0; i < P.n_limbs; i++)
               trial[i] = i == k ? w : P.owner[i];
            rnea_loads(P, trial, load);
            int worst = 0;
            for (int v = 0; v < WAVES; v++)
               worst = load[v] > worst ? load[v] : worst;
            if (worst < best)
               best = worst, bk = k, bw = w;
         }
      }
      if (bk >= 0)
         P.owner_f[bk] = bw, P.mailed = bk;
   }
   static constexpr Plan make()
   {
      Plan P;
      int nchild[N] = {}, size[N] = {};
      bool branch[N] = {};
      for (int j = 0; j < N; j++)
         if (TP::parent[j] >= 0)
            nchild[TP::parent[j]]++;
      for (int j = N - 1; j >= 0; j--)
      { // children have larger indices than their parent (engine order)
         size[j] += 1;
         branch[j] = branch[j] || nchild[j] >= 2;
         if (TP::parent[j] >= 0)
         {
            size[TP::parent[j]] += size[j];
            branch[TP::parent[j]] = branch[TP::parent[j]] || branch[j];
         }
      }
      for (int j = 0; j < N; j++)
      {
         const int p = TP::parent[j];
         if (branch[j])
            P.trunk[j] = true;
         else if (p < 0)
            P.trunk[j] = false;
         else if (branch[p])
            P.trunk[j] = size[j] < MIN_LIMB; // a chain hanging off the trunk: a limb if long enough
         else
            P.trunk[j] = P.trunk[p];         // further down a chain: same side as the body above
      }
      for (int j = 0; j < N; j++)
      {
         const int p = TP::parent[j];
         P.limb_root[j] = !P.trunk[j] && (p < 0 || P.trunk[p]);
         if (P.limb_root[j])
         {
            P.root_of[P.n_limbs] = j;
            P.size_of[P.n_limbs] = size[j];
            P.limb_of[j] = P.n_limbs++;
         }
         else if (!P.trunk[j])
            P.limb_of[j] = P.limb_of[p];
      }
      // greedy balance: limbs in decreasing size go to the least loaded wave
      int load[WAVES] = {};
      bool done[N] = {};
      for (int round = 0; round < P.n_limbs; round++)
      {
         int best = -1, bs = -1;
         for (int i = 0; i < P.n_limbs; i++)
            if (!done[i] && P.size_of[i] > bs)
            {
               bs = P.size_of[i];
               best = i;
            }
         int w = 0;
         for (int i = 1; i < WAVES; i++)
            if (load[i] < load[w])
               w = i;
         P.owner[best] = w;
         load[w] += bs;
         done[best] = true;
      }
      for (int k = 0; k < P.n_limbs; k++)
         P.owner_plain[k] = P.owner[k];
      for (int j = 0; j < N; j++)
         if (P.trunk[j])
            P.trunk_rank[j] = P.n_trunk++;
      assign_wrench_duties(P, P.owner_plain, P.f_limb);
      P.usable = P.n_limbs >= 2;
      for (int j = 0; j < N; j++)
         if (TP::parent[j] < 0 && !P.trunk[j])
            P.usable = false; // every root must be a trunk body
      make_stages(P);
      if (!P.staged)
         for (int k = 0; k < P.n_limbs; k++)
            P.owner[k] = P.owner_plain[k];
      // ... under the final ABA owners (the fused kernel's inverse dynamics walks with those), by the limbs' bodies alone: there the wave with a
      // leg AND the neck is the last one whatever the others do, and the refined count made the step 2 % slower (27.3 -> 27.8 us at 32 768)
      assign_wrench_duties(P, P.owner, P.f_limb_aba, false);
      mail_one_limb(P);
      assign_wrench_duties(P, P.owner_f, P.f_limb_f, MH_ZVF_MAIL_COUNT_WALKS != 0);
      // ABA hand-over placement: trunk bodies in LDS (all waves write the same values), limb bodies in the owner's registers
      int regs[WAVES] = {};
      for (int j = 0; j < N; j++)
      {
         P.trunk_slot[j + 1] = P.trunk_slot[j] + (P.trunk[j] ? TR::aba_slots_of(j) : 0);
         if (!P.trunk[j])
         {
            const int w = P.owner[P.limb_of[j]];
            P.reg_slot[j] = regs[w];
            regs[w] += TR::aba_slots_of(j);
         }
      }
      for (int w = 0; w < WAVES; w++)
         P.reg_slots = regs[w] > P.reg_slots ? regs[w] : P.reg_slots;
      int zregs[WAVES] = {};
      for (int j = 0; j < N; j++)
      {
         P.zv_trunk_slot[j + 1] = P.zv_trunk_slot[j] + (P.trunk[j] ? TR::zv_slots_of(j, false) : 0);
         if (!P.trunk[j])
         {
            const int w = P.owner[P.limb_of[j]];
            P.zv_reg_slot[j] = zregs[w];
            zregs[w] += TR::zv_slots_of(j, true);
         }
      }
      for (int w = 0; w < WAVES; w++)
         P.zv_reg_slots = zregs[w] > P.zv_reg_slots ? zregs[w] : P.zv_reg_slots;
      return P;
   }
   static constexpr Plan P = make();
   static constexpr bool is_trunk(int j) { return P.trunk[j]; }
   static constexpr int n_limbs() { return P.n_limbs; }
   static constexpr int limb_root(int k) { return P.root_of[k]; }
   static constexpr int limb_index(int root) { return P.limb_of[root]; }
   static constexpr int limb_index_of_body(int j) { return P.limb_of[j]; } // any body of a limb
   static constexpr int owner(int k) { return P.owner[k]; }             // ABA
   static constexpr int owner_plain(int k) { return P.owner_plain[k]; } // RNEA, CRBA
   static constexpr int f_limb(int j) { return P.f_limb[j]; }
   // OWN = 1: the inverse dynamics walks its limbs with the ABA's owners (fused bias + inertia kernel, mh_zv_kernels.h)
   // OWN = 2: the same with one one-body limb walked by another wave (mail_one_limb)
   template <int OWN>
   static constexpr int owner_sel(int k) { return OWN == 2 ? P.owner_f[k] : (OWN ? P.owner[k] : P.owner_plain[k]); }
   template <int OWN>
   static constexpr int f_limb_sel(int j) { return OWN == 2 ? P.f_limb_f[j] : (OWN ? P.f_limb_aba[j] : P.f_limb[j]); }
   static constexpr int mailed_limb() { return P.mailed; }
   static constexpr int rnea_trunk_slot(int j) { return 8 * P.trunk_rank[j]; }
   static constexpr int RNEA_TRUNK_SLOTS = 8 * P.n_trunk;
   static constexpr bool staged() { return P.staged; }
   static constexpr int root() { return P.root; }
   static constexpr bool is_late(int k) { return P.late[k]; }
   static constexpr int n_sub() { return P.n_sub; }
   static constexpr int sub_top(int i) { return P.sub_top[i]; }
   static constexpr int sub_owner(int j) { return P.sub_owner[j]; }
   static constexpr int sub_slot(int j) { return P.sub_slot[j]; }
   static constexpr int cut_limb(int w) { return P.cut_limb[w]; }
   static constexpr bool is_cut(int j)
   {
      for (int w = 0; w < WAVES; w++)
         if (P.staged && P.cut_body[w] == j)
            return true;
      return false;
   }
   static constexpr bool usable() { return P.usable; }
   static constexpr int trunk_slot(int j) { return P.trunk_slot[j]; }
   static constexpr int TRUNK_SLOTS = P.trunk_slot[N];
   static constexpr int reg_slot(int j) { return P.reg_slot[j]; }
   static constexpr int reg_slots() { return P.reg_slots; }
   static constexpr int zv_trunk_slot(int j) { return P.zv_trunk_slot[j]; }
   static constexpr int ZV_TRUNK_SLOTS = P.zv_trunk_slot[N];
   static constexpr int zv_reg_slot(int j) { return P.zv_reg_slot[j]; }
   static constexpr int zv_reg_slots() { return P.zv_reg_slots; }
};

template <typename T>
using lds_ptr = T __attribute__((address_space(3))) *;

// pose of joint J in its parent's frame; for a first child of a 1-DoF joint the y and z components of the position are structural zeros
// (Tree<TP>::p_aligned): literal here, so that every product with them folds away (-fno-signed-zeros -ffinite-math-only) and the two
// scalar loads are never issued
template <class TP, int J, typename T, class CR>
MH_DEV XF<T> load_xb_j(const CR &c)
{
   XF<T> X = load_xb<T>(c);
   if constexpr (Tree<TP>::p_aligned(J))
      X.p.y = T(0), X.p.z = T(0);
   return X;
}

// Per-lane store for values that must survive from ABA's inward sweep to its outward sweep.  Where slot k of body J lives is
// a compile-time decision of the store policy SP:
//   LDS       LDS, slot-major, 64 lanes per slot: "ds_write/read_b64 base offset:imm", no per-slot address register
//   GLOBAL    global workspace ws[slot * stride + lane], wave-uniform base (scalar address arithmetic)
//   REG       a register array of the lane (tree-split kernels: the limb bodies of the wave that owns them)
enum : int
{
   ST_LDS_KIND = 0,
   ST_GLOBAL_KIND = 1,
   ST_REG_KIND = 2
};
template <class TP, int KIND>
struct WholeStore
{ // one wave walks the whole tree: everything in LDS or everything in the global workspace
   static constexpr int REG_SLOTS = 0;
   static constexpr int kind(int) { return KIND; }
   static constexpr int index(int j) { return Tree<TP>::aba_slot(j); }
};
template <class TP>
struct SplitStore
{
   static constexpr int REG_SLOTS = Split<TP>::reg_slots();
   static constexpr int kind(int j) { return Split<TP>::is_trunk(j) ? ST_LDS_KIND : ST_REG_KIND; }
   static constexpr int index(int j) { return Split<TP>::is_trunk(j) ? Split<TP>::trunk_slot(j) : Split<TP>::reg_slot(j); }
};
// Tree-split inverse dynamics with the pairs (cos, sin) of every revolute joint a wave evaluates formed BEFORE its walks, by the
// straight-line fast path of the sincos (rnea_pre_pass below): slots 2 j and 2 j + 1 of the wave's registers.  A context with this store
// policy (Ctx::rnea_pre) makes RneaSub / trunk_va take the pairs from there instead of evaluating sincos_t -- a basic block of its own
// with a 45-instruction dependent chain -- inside every body step.  The pre-pass sits at the head of the wave's OWN `if` chain
// (split_rnea_limbs), so the scheduler interleaves its tail with the first bodies' constant loads.  Measured and not kept
// (profiles/r05_headline_steps.txt): the pairs in LDS and the pre-pass as a phase of its own between the arrival of the rows of q and
// that of the velocities and efforts -- the rows of q are there 0.37 us ahead of the rest, and a pre-pass that overlaps nothing else takes
// 1.2 us (eight or nine chains of ~48 instructions) against ~0.6 us of added walk time here.
#ifndef MH_RNEA_PRE
#define MH_RNEA_PRE 1 // 0: sincos_t inside every body step of the tree-split inverse dynamics, as before round 5 (A/B measurements)
#endif
#ifndef MH_ZVF_PRE
#define MH_ZVF_PRE 1 // the same switch for the fused forward dynamics of device-filling batches (its limb joints' pairs: slots 7, 8 of its own store)
#endif
template <class TP>
struct RneaPreStore
{
   static constexpr int REG_SLOTS = 2 * TP::N;
   static constexpr bool rnea_pre = true;
   static constexpr int kind(int) { return ST_REG_KIND; }
   static constexpr int index(int j) { return 2 * j; }
};
template <class SP, class = void>
struct policy_rnea_pre
{
   static constexpr bool value = false;
};
template <class SP>
struct policy_rnea_pre<SP, std::enable_if_t<SP::rnea_pre>>
{
   static constexpr bool value = true;
};
// a policy may place the slots of one body in different homes (ZvfStore: the mailed limb): slot_kind(j, k) / slot_index(j, k)
template <class SP, class = void>
struct policy_slot_homes
{
   static constexpr int kind(int j, int) { return SP::kind(j); }
   static constexpr int index(int j, int k) { return SP::index(j) + k; }
};
template <class SP>
struct policy_slot_homes<SP, std::enable_if_t<SP::slot_homes>>
{
   static constexpr int kind(int j, int k) { return SP::slot_kind(j, k); }
   static constexpr int index(int j, int k) { return SP::slot_index(j, k); }
};
template <typename T, class SP>
struct LaneStore
{
   lds_ptr<T> lbase; // lds + lane-in-wave
   T *gbase;         // wave-uniform workspace pointer
   long stride, lane;
   mutable T regs[SP::REG_SLOTS > 0 ? SP::REG_SLOTS : 1];
   template <int J, int K>
   MH_DEV void put(T v) const
   {
      constexpr int kind = policy_slot_homes<SP>::kind(J, K), slot = policy_slot_homes<SP>::index(J, K);
      if constexpr (kind == ST_LDS_KIND)
         lbase[slot * 64] = v;
      else if constexpr (kind == ST_GLOBAL_KIND)
         gbase[(long)slot * stride + lane] = v;
      else
         regs[slot] = v;
   }
   template <int J, int K>
   MH_DEV T get() const
   {
      constexpr int kind = policy_slot_homes<SP>::kind(J, K), slot = policy_slot_homes<SP>::index(J, K);
      if constexpr (kind == ST_LDS_KIND)
         return lbase[slot * 64];
      else if constexpr (kind == ST_GLOBAL_KIND)
         return gbase[(long)slot * stride + lane];
      else
         return regs[slot];
   }
};

// Everything one lane needs to walk the tree.  IO_LDS: state rows staged in LDS.  IDENT: the index maps are the identity
// (Mecano's default JointMatrixIndexProvider over joints in depth-first order), so every row index is a compile-time constant.
// OUTMODE 1 (bias job of the bias-split forward dynamics, mh_zv_kernels.h): out(k, v) leaves in3(k) - v, i.e. with in3 = tau and the
// accelerations switched off the inverse-dynamics walk writes tau - h(q, qd) instead of h
// CSMODE (two-launch forward dynamics of device-filling batches, mh_zv_kernels.h): 1 = the walk also leaves (cos, sin) of every revolute
// joint in a slot-major scratch matrix, cs[(2 r + {0, 1}) * cs_stride]; 2 = the walk takes them from there, q and qd are read from the
// caller's matrices (never staged) while in3 / out stay LDS rows, and the bias fold's exchange records are 12 wide instead of 21;
// 3 = bias and inertia job fused in one workgroup (spec_zvf_kernel): the inverse-dynamics walk leaves the pairs and tau - h in the slots
// of the hand-over store (registers of the limb's owner, LDS for the trunk), nothing travels through memory
template <typename T, bool IO_LDS, bool IDENT, class SP, bool BODIES = false, int OUTMODE = 0, int CSMODE = 0>
struct Ctx
{
   using SPolicy = SP;
   static constexpr int csmode = CSMODE;
   static constexpr bool rnea_pre = policy_rnea_pre<SP>::value; // (cos, sin) of the wave's revolute joints already in st (RneaPreStore)
   static constexpr int fold_xw = CSMODE >= 2 ? 12 : 21; // width of a limb's record in the bias fold (ZV_XW while the inertias' records are reused)
   T *cs;          // CSMODE 1, 2: this configuration's column of the (cos, sin) scratch
   long cs_stride;
   lds_ptr<T> park; // CSMODE 3: where the inverse dynamics parks the trunk's wrenches (+ lane); see rnea_park
   // BODIES: the kernel also writes every successor body's spatial acceleration / twist (RigidBodyAccelerationProvider; SURVEY.md
   // section 8f N2) -- a property of the context TYPE, so that the kernels without it stay instruction for instruction what they were
   static constexpr bool bodies = BODIES;
   T *bacc, *btw; // this configuration's rows of the two per-body outputs (either may be NULL), entry stride f_es
   const T *C; // per-joint constants [N][MC_STRIDE], global memory, wave-uniform addresses -> scalar loads
   ciptr dof_map, cfg_map, meta;
   const T *qrow, *qdrow, *in3row, *frow;
   T *orow;
   T *orow2;     // Coriolis kernel: this configuration's C (orow: its H); entry stride f_es
   long q_es, v_es, f_es;
   lds_ptr<T> lq, lqd, lx, lo; // this lane's rows in LDS (IO_LDS); lo = output row (may alias lx)
   V3<T> a0a, a0l;         // angular and linear part of the root acceleration ((0, -g) for a gravity vector)
   int coriolis, accel;
   LaneStore<T, SP> st;
   int nv;                       // CRBA: rows of H
   int wave;                     // tree-split kernels: which wave of the workgroup this is
   unsigned long long own;       // Coriolis / centroidal kernels: bit J = this wave writes the columns of body J (several waves may share a group of configurations)
   lds_ptr<T> xbase;             // tree-split kernels: limb -> trunk exchange area in LDS (+ lane)

   MH_DEV int ci(int k) const { return IDENT ? k : cfg_map[k]; }
   MH_DEV int di(int k) const { return IDENT ? k : dof_map[k]; }
   MH_DEV T q(int k) const
   {
      if constexpr (IO_LDS && CSMODE != 2)
         return lq[ci(k)];
      else
         return qrow[ci(k) * q_es];
   }
   MH_DEV T qd(int k) const
   {
      if constexpr (IO_LDS && CSMODE != 2)
         return lqd[di(k)];
      else
         return qdrow[di(k) * v_es];
   }
   MH_DEV T in3(int k) const
   {
      if constexpr (IO_LDS)
         return lx[di(k)];
      else
         return in3row[di(k) * v_es];
   }
   MH_DEV void out(int k, T v) const
   {
      if constexpr (OUTMODE == 1)
         v = in3(k) - v;
      if constexpr (IO_LDS && OUTMODE != 2) // OUTMODE 2: results go straight to the caller's matrix although the inputs are staged in LDS
         lo[di(k)] = v;
      else
         orow[di(k) * v_es] = v;
   }
};

// raw configuration of a joint: read (LDS / global) at the top of a body together with its constants, turned into the joint
// transform after the fence so that ONE wait covers every request of the body (SMEM returns out of order: any lgkmcnt wait is a
// wait for everything outstanding, so requests are batched up front rather than trickled in between the arithmetic)
template <typename T>
struct JQ
{
   T q[7];
};
template <int TYPE, int CO, class CX, typename T>
MH_DEV JQ<T> spec_joint_read(const CX &cx)
{
   JQ<T> r;
   if constexpr (TYPE == JT_REVOLUTE || TYPE == JT_PRISMATIC)
      r.q[0] = cx.q(CO);
   else if constexpr (TYPE == JT_SIXDOF)
      for (int k = 0; k < 7; k++)
         r.q[k] = cx.q(CO + k);
   return r;
}
template <int TYPE, typename T>
MH_DEV JX<T> spec_joint_from(const JQ<T> &r)
{
   JX<T> jx;
   jx.c = T(1), jx.s = T(0), jx.d = T(0);
   if constexpr (TYPE == JT_REVOLUTE)
      sincos_t(r.q[0], jx.s, jx.c);
   else if constexpr (TYPE == JT_PRISMATIC)
      jx.d = r.q[0];
   else if constexpr (TYPE == JT_SIXDOF)
   {
      jx.X.R = quat_to_R(r.q[0], r.q[1], r.q[2], r.q[3]);
      jx.X.p = V3<T>{r.q[4], r.q[5], r.q[6]};
   }
   return jx;
}
template <int TYPE, int CO, class CX, typename T>
MH_DEV JX<T> spec_joint(const CX &cx)
{
   return spec_joint_from<TYPE, T>(spec_joint_read<TYPE, CO, CX, T>(cx));
}
// slice of qd (WHICH = 0) or of qdd|tau (WHICH = 1) belonging to the joint, as a spatial vector in its canonical frame
template <int TYPE, int DO, int WHICH, class CX, typename T>
MH_DEV SV<T> spec_vec(const CX &cx, bool enabled)
{
   SV<T> o{V3<T>{T(0), T(0), T(0)}, V3<T>{T(0), T(0), T(0)}};
   if (!enabled)
      return o;
   auto rd = [&](int k) { return WHICH == 0 ? cx.qd(k) : cx.in3(k); };
   if constexpr (TYPE == JT_REVOLUTE)
      o.a.z = rd(DO);
   else if constexpr (TYPE == JT_PRISMATIC)
      o.l.z = rd(DO);
   else if constexpr (TYPE == JT_SIXDOF)
   {
      o.a = V3<T>{rd(DO + 0), rd(DO + 1), rd(DO + 2)};
      o.l = V3<T>{rd(DO + 3), rd(DO + 4), rd(DO + 5)};
   }
   return o;
}
template <int TYPE, int DO, class CX, typename T>
MH_DEV void spec_write(const CX &cx, SV<T> w)
{
   if constexpr (TYPE == JT_REVOLUTE)
      cx.out(DO, w.a.z);
   else if constexpr (TYPE == JT_PRISMATIC)
      cx.out(DO, w.l.z);
   else if constexpr (TYPE == JT_SIXDOF)
   {
      cx.out(DO + 0, w.a.x), cx.out(DO + 1, w.a.y), cx.out(DO + 2, w.a.z);
      cx.out(DO + 3, w.l.x), cx.out(DO + 4, w.l.y), cx.out(DO + 5, w.l.z);
   }
}

// Fused bias + inertia kernel (CSMODE 3, mh_zv_kernels.h): tau - h of joint J, left by the inverse-dynamics walk where the bias fold of the
// same workgroup reads it -- the owner's registers for a limb body, the trunk's LDS slots for a trunk body, LDS slots of their own for a
// root body whose other slots are registers of every wave (the store policy says which: ZvfStore).
template <class TP, int J, int K, class CX, typename T>
MH_DEV void zvf_tau_put1(const CX &cx, T v)
{
   using SP = typename CX::SPolicy;
   if constexpr (SP::root_in_regs(J))
      cx.st.lbase[(SP::root_tau_slot() + K) * 64] = v;
   else
      cx.st.template put<J, SP::tau_slot(J) + K>(v);
}
template <class TP, int J, int K, class CX, typename T>
MH_DEV T zvf_tau_get1(const CX &cx)
{ // (what the bias fold reads: a late limb's entries have moved to LDS by then, zvf_park_late_tau)
   using SP = typename CX::SPolicy;
   if constexpr (SP::root_in_regs(J))
      return cx.st.lbase[(SP::root_tau_slot() + K) * 64];
   else if constexpr (SP::late_body(J))
      return cx.st.lbase[(SP::late_tau_slot(J) + K) * 64];
   else
      return cx.st.template get<J, SP::tau_slot(J) + K>();
}
template <class TP, int J, class CX, typename T>
MH_DEV void zvf_put_tau(const CX &cx, const SV<T> &h)
{
   constexpr int TYPE = TP::type[J], DO = Tree<TP>::dof_ofs(J);
   if constexpr (TYPE == JT_REVOLUTE)
      zvf_tau_put1<TP, J, 0, CX, T>(cx, cx.in3(DO) - h.a.z);
   else if constexpr (TYPE == JT_PRISMATIC)
      zvf_tau_put1<TP, J, 0, CX, T>(cx, cx.in3(DO) - h.l.z);
   else if constexpr (TYPE == JT_SIXDOF)
   {
      zvf_tau_put1<TP, J, 0, CX, T>(cx, cx.in3(DO + 0) - h.a.x), zvf_tau_put1<TP, J, 1, CX, T>(cx, cx.in3(DO + 1) - h.a.y);
      zvf_tau_put1<TP, J, 2, CX, T>(cx, cx.in3(DO + 2) - h.a.z), zvf_tau_put1<TP, J, 3, CX, T>(cx, cx.in3(DO + 3) - h.l.x);
      zvf_tau_put1<TP, J, 4, CX, T>(cx, cx.in3(DO + 4) - h.l.y), zvf_tau_put1<TP, J, 5, CX, T>(cx, cx.in3(DO + 5) - h.l.z);
   }
}

// per-body outputs of body J (canonical after-joint frame -> Mecano's body-fixed frame, row MI_EXT of the caller's joint listing)
template <int J, class CX, typename T>
MH_DEV void spec_body_outputs(const CX &cx, const SV<T> &acc, const SV<T> &twist)
{
   if constexpr (CX::bodies)
   {
      const CRef<T, false> c{cx.C + J * MC_STRIDE};
      const int ext = cx.meta[J * MI_STRIDE + MI_EXT];
      if (cx.bacc)
         store_body_motion<T>(c, cx.bacc, cx.f_es, ext, acc);
      if (cx.btw)
         store_body_motion<T>(c, cx.btw, cx.f_es, ext, twist);
   }
}

// ============================================================================================ RNEA
// Returns the wrench the subtree rooted at joint J exerts on its parent, expressed in the parent's frame
// (InverseDynamicsCalculator.java:873-966 as one depth-first recursion).
// limb -> trunk exchange area of the tree-split kernels: XW scalars per limb, slot-major, 64 lanes per slot
template <int LIMB, int XW, int I, class CX, typename T>
MH_DEV void x_put(const CX &cx, T v)
{
   cx.xbase[(LIMB * XW + I) * 64] = v;
}
template <int LIMB, int XW, int I, class CX, typename T>
MH_DEV T x_get(const CX &cx)
{
   return cx.xbase[(LIMB * XW + I) * 64];
}
template <int LIMB, int XW, int I0, class CX, typename T>
MH_DEV void x_put6(const CX &cx, const SV<T> &w)
{
   x_put<LIMB, XW, I0 + 0, CX, T>(cx, w.a.x), x_put<LIMB, XW, I0 + 1, CX, T>(cx, w.a.y), x_put<LIMB, XW, I0 + 2, CX, T>(cx, w.a.z);
   x_put<LIMB, XW, I0 + 3, CX, T>(cx, w.l.x), x_put<LIMB, XW, I0 + 4, CX, T>(cx, w.l.y), x_put<LIMB, XW, I0 + 5, CX, T>(cx, w.l.z);
}
template <int LIMB, int XW, int I0, class CX, typename T>
MH_DEV SV<T> x_get6(const CX &cx)
{
   return SV<T>{V3<T>{x_get<LIMB, XW, I0 + 0, CX, T>(cx), x_get<LIMB, XW, I0 + 1, CX, T>(cx), x_get<LIMB, XW, I0 + 2, CX, T>(cx)},
                V3<T>{x_get<LIMB, XW, I0 + 3, CX, T>(cx), x_get<LIMB, XW, I0 + 4, CX, T>(cx), x_get<LIMB, XW, I0 + 5, CX, T>(cx)}};
}

// MODE 0: the recursion walks the whole subtree.  MODE 1 (tree-split kernels, trunk pass): children that are limb roots are
// not walked; what their limb hands up is read from the exchange area, where the wave that owns the limb left it.
template <class TP, int J, typename T, class CX, int MODE = 0>
struct RneaSub
{
   template <int K>
   static MH_DEV void children(const CX &cx, const SV<T> &v, const SV<T> &a, SV<T> &f)
   {
      if constexpr (K < Tree<TP>::n_children(J))
      {
         constexpr int C = Tree<TP>::child(J, K);
         if constexpr (MODE == 1 && !Split<TP>::is_trunk(C))
            f = f + x_get6<Split<TP>::limb_index(C), 6, 0, CX, T>(cx);
         else
            f = f + RneaSub<TP, C, T, CX, MODE>::run(cx, v, a);
         children<K + 1>(cx, v, a, f);
      }
   }
   static MH_DEV SV<T> run(const CX &cx, const SV<T> &vp, const SV<T> &ap)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J];
      constexpr int DO = Tree<TP>::dof_ofs(J), CO = Tree<TP>::cfg_ofs(J);
      const CRef<T, false> c{cx.C + J * MC_STRIDE};
      // (cos, sin) formed in front of the walk: RneaPreStore (slots 0, 1), or -- fused forward dynamics, the bodies of a limb -- the slots the
      // inertia walk of the same wave reads them from (7, 8; rnea_pre_pass)
      constexpr bool PRE3 = CX::csmode == 3 && MH_ZVF_PRE && TYPE == JT_REVOLUTE && Split<TP>::usable() && !Split<TP>::is_trunk(J);
      constexpr bool PRE = (CX::rnea_pre && TYPE == JT_REVOLUTE) || PRE3;
      constexpr int PS = PRE3 ? 7 : 0;
      JQ<T> jq;
      if constexpr (!PRE)
         jq = spec_joint_read<TYPE, CO, CX, T>(cx);
      const SV<T> vJ = spec_vec<TYPE, DO, 0, CX, T>(cx, cx.coriolis != 0);
      const SV<T> aJ = spec_vec<TYPE, DO, 1, CX, T>(cx, cx.accel != 0);
      const XF<T> Xb = load_xb_j<TP, J, T>(c);
      const RI<T> I = load_inertia<T>(c);
      MH_BODY_FENCE(); // everything the body reads is requested before its arithmetic starts (see JQ)
      JX<T> jx;
      if constexpr (PRE)
         jx.c = cx.st.template get<J, PS>(), jx.s = cx.st.template get<J, PS + 1>(), jx.d = T(0);
      else
         jx = spec_joint_from<TYPE, T>(jq);
      if constexpr (CX::csmode == 1 && TYPE == JT_REVOLUTE)
      {
         constexpr int R = Tree<TP>::rev_index(J);
         cx.cs[(2 * R) * cx.cs_stride] = jx.c, cx.cs[(2 * R + 1) * cx.cs_stride] = jx.s;
      }
      if constexpr (CX::csmode == 3 && TYPE == JT_REVOLUTE && !PRE3)
      { // fused kernel: the pair stays in this wave's registers, in the slots the inertia walk of the same limb reads it from
         cx.st.template put<J, 7>(jx.c);
         cx.st.template put<J, 8>(jx.s);
      }
      const V3<T> Z{T(0), T(0), T(0)};
      SV<T> v = motion_down(TYPE, jx, Xb, vp) + vJ;
      const SV<T> a = motion_down(TYPE, jx, Xb, ap) + aJ + crm(v, vJ);
      if (!cx.coriolis)
         v = SV<T>{Z, Z};
      spec_body_outputs<J, CX, T>(cx, a, v);
      SV<T> f = mul(I, a) + crf(v, mul(I, v));
      if (cx.frow)
         f = f - load_fext<T>(c, cx.frow, cx.f_es, cx.meta[J * MI_STRIDE + MI_EXT]);
      MH_BODY_FENCE();
      children<0>(cx, v, a, f);
      MH_BODY_FENCE();
      if constexpr (CX::csmode == 3)
         zvf_put_tau<TP, J, CX, T>(cx, f);
      else
         spec_write<TYPE, DO, CX, T>(cx, f);
      // the joint pose is read again rather than kept in 24 SGPRs per tree level across the subtree (which overflows the
      // SGPR file and turns every use into a v_readlane); the pointer is laundered so that the reload is not merged away
      const T *c2p = cx.C + J * MC_STRIDE;
      asm volatile("" : "+s"(c2p));
      const SV<T> up = force_up(TYPE, jx, load_xb_j<TP, J, T>(CRef<T, false>{c2p}), f);
      MH_BODY_FENCE();
      return up;
   }
};
template <class TP, typename T, class CX, int MODE = 0, int K = 0>
MH_DEV void rnea_roots(const CX &cx)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      const V3<T> Z{T(0), T(0), T(0)};
      (void)RneaSub<TP, Tree<TP>::child(-1, K), T, CX, MODE>::run(cx, SV<T>{Z, Z}, SV<T>{cx.a0a, cx.a0l});
      rnea_roots<TP, T, CX, MODE, K + 1>(cx);
   }
}
// where the tree-split inverse dynamics parks the trunk's Newton-Euler wrenches: the LDS block of the hand-over store, except in the fused
// bias + inertia kernel (CSMODE 3), whose store holds the forward dynamics' hand-over there
template <class CX, typename T>
MH_DEV lds_ptr<T> rnea_park(const CX &cx)
{
   if constexpr (CX::csmode == 3)
      return cx.park;
   else
      return cx.st.lbase;
}
// velocity and acceleration of trunk body J, walked down from the root (tree-split kernels: every wave needs them for its limbs)
// FK = the limb this walk is for: the walk of limb Split<TP>::f_limb(J) also forms the body's own Newton-Euler wrench
// f = I a + v x* I v - f_ext (InverseDynamicsCalculator.java:935-947) and parks it with (cos, sin) in the LDS trunk area, so that
// the trunk pass after the barrier is a pure fold of 6-vectors (RneaTrunkUp) instead of a second full walk by one wave
template <class TP, int J, typename T, class CX, int FK = -1, int OWN = 0>
MH_DEV void trunk_va(const CX &cx, SV<T> &v, SV<T> &a)
{
   constexpr int TYPE = TP::type[J];
   constexpr int DO = Tree<TP>::dof_ofs(J), CO = Tree<TP>::cfg_ofs(J);
   const V3<T> Z{T(0), T(0), T(0)};
   SV<T> vp{Z, Z}, ap{cx.a0a, cx.a0l};
   if constexpr (TP::parent[J] >= 0)
      trunk_va<TP, TP::parent[J], T, CX, FK, OWN>(cx, vp, ap);
   MH_BODY_FENCE();
   const CRef<T, false> c{cx.C + J * MC_STRIDE};
   constexpr bool PRE = CX::rnea_pre && TYPE == JT_REVOLUTE;
   JQ<T> jq;
   if constexpr (!PRE)
      jq = spec_joint_read<TYPE, CO, CX, T>(cx);
   const SV<T> vJ = spec_vec<TYPE, DO, 0, CX, T>(cx, cx.coriolis != 0);
   const SV<T> aJ = spec_vec<TYPE, DO, 1, CX, T>(cx, cx.accel != 0);
   const XF<T> Xb = load_xb_j<TP, J, T>(c);
   MH_BODY_FENCE();
   JX<T> jx;
   if constexpr (PRE)
      jx.c = cx.st.template get<J, 0>(), jx.s = cx.st.template get<J, 1>(), jx.d = T(0);
   else
      jx = spec_joint_from<TYPE, T>(jq);
   v = motion_down(TYPE, jx, Xb, vp) + vJ;
   a = motion_down(TYPE, jx, Xb, ap) + aJ + crm(v, vJ);
   if (!cx.coriolis)
      v = SV<T>{Z, Z};
   if constexpr (FK >= 0 && Split<TP>::template f_limb_sel<OWN>(J) == FK)
   {
      spec_body_outputs<J, CX, T>(cx, a, v);
      const RI<T> I = load_inertia<T>(c);
      SV<T> f = mul(I, a) + crf(v, mul(I, v));
      if (cx.frow)
         f = f - load_fext<T>(c, cx.frow, cx.f_es, cx.meta[J * MI_STRIDE + MI_EXT]);
      constexpr int S0 = Split<TP>::rnea_trunk_slot(J);
      const lds_ptr<T> t = rnea_park<CX, T>(cx);
      t[(S0 + 0) * 64] = f.a.x, t[(S0 + 1) * 64] = f.a.y, t[(S0 + 2) * 64] = f.a.z;
      t[(S0 + 3) * 64] = f.l.x, t[(S0 + 4) * 64] = f.l.y, t[(S0 + 5) * 64] = f.l.z;
      t[(S0 + 6) * 64] = jx.c, t[(S0 + 7) * 64] = jx.s;
      if constexpr (CX::csmode == 1 && TYPE == JT_REVOLUTE)
      {
         constexpr int R = Tree<TP>::rev_index(J);
         cx.cs[(2 * R) * cx.cs_stride] = jx.c, cx.cs[(2 * R + 1) * cx.cs_stride] = jx.s;
      }
   }
   MH_BODY_FENCE();
}
// The trunk pass of the tree-split RNEA (wave 0, after the barrier): wrench of trunk body J = its own (parked by trunk_va) + what its
// limbs (exchange area) and trunk children hand up; joint effort; hand-up to the parent (InverseDynamicsCalculator.java:949-966).
template <class TP, int J, typename T, class CX>
struct RneaTrunkUp
{
   template <int K>
   static MH_DEV void children(const CX &cx, SV<T> &f)
   {
      if constexpr (K < Tree<TP>::n_children(J))
      {
         constexpr int C = Tree<TP>::child(J, K);
         if constexpr (!Split<TP>::is_trunk(C))
            f = f + x_get6<Split<TP>::limb_index(C), 6, 0, CX, T>(cx);
         else
            f = f + RneaTrunkUp<TP, C, T, CX>::run(cx);
         children<K + 1>(cx, f);
      }
   }
   static MH_DEV SV<T> run(const CX &cx)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J];
      constexpr int DO = Tree<TP>::dof_ofs(J), CO = Tree<TP>::cfg_ofs(J), S0 = Split<TP>::rnea_trunk_slot(J);
      const lds_ptr<T> t = rnea_park<CX, T>(cx);
      SV<T> f{V3<T>{t[(S0 + 0) * 64], t[(S0 + 1) * 64], t[(S0 + 2) * 64]}, V3<T>{t[(S0 + 3) * 64], t[(S0 + 4) * 64], t[(S0 + 5) * 64]}};
      JX<T> jx;
      jx.c = t[(S0 + 6) * 64], jx.s = t[(S0 + 7) * 64], jx.d = T(0);
      if constexpr (TYPE != JT_REVOLUTE && TP::parent[J] >= 0)
         jx = spec_joint_from<TYPE, T>(spec_joint_read<TYPE, CO, CX, T>(cx));
      children<0>(cx, f);
      MH_BODY_FENCE();
      if constexpr (CX::csmode == 3)
      { // fused bias + inertia kernel: tau - h and (cos, sin) go where the forward dynamics of the same workgroup picks them up
         zvf_put_tau<TP, J, CX, T>(cx, f);
         if constexpr (TYPE == JT_REVOLUTE)
         {
            cx.st.template put<J, 7>(jx.c);
            cx.st.template put<J, 8>(jx.s);
         }
      }
      else
         spec_write<TYPE, DO, CX, T>(cx, f);
      SV<T> up = f;
      if constexpr (TP::parent[J] >= 0)
         up = force_up(TYPE, jx, load_xb_j<TP, J, T>(CRef<T, false>{cx.C + J * MC_STRIDE}), f);
      MH_BODY_FENCE();
      return up;
   }
};
template <class TP, typename T, class CX, int K = 0>
MH_DEV void rnea_trunk_roots(const CX &cx)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      (void)RneaTrunkUp<TP, Tree<TP>::child(-1, K), T, CX>::run(cx);
      rnea_trunk_roots<TP, T, CX, K + 1>(cx);
   }
}
// velocity only (ABA)
template <class TP, int J, typename T, class CX>
MH_DEV SV<T> trunk_v(const CX &cx)
{
   constexpr int TYPE = TP::type[J];
   constexpr int DO = Tree<TP>::dof_ofs(J), CO = Tree<TP>::cfg_ofs(J);
   const V3<T> Z{T(0), T(0), T(0)};
   SV<T> vp{Z, Z};
   if constexpr (TP::parent[J] >= 0)
      vp = trunk_v<TP, TP::parent[J], T, CX>(cx);
   MH_BODY_FENCE();
   const CRef<T, false> c{cx.C + J * MC_STRIDE};
   const JQ<T> jq = spec_joint_read<TYPE, CO, CX, T>(cx);
   const SV<T> vJ = spec_vec<TYPE, DO, 0, CX, T>(cx, true);
   const XF<T> Xb = load_xb_j<TP, J, T>(c);
   MH_BODY_FENCE();
   const JX<T> jx = spec_joint_from<TYPE, T>(jq);
   const SV<T> v = motion_down(TYPE, jx, Xb, vp) + vJ;
   MH_BODY_FENCE();
   return v;
}

// ============================================================================================ ABA
template <typename T>
struct AbaUp
{ // what a subtree hands to its parent: articulated inertia and bias wrench, in the parent's frame
   ABI<T> I;
   SV<T> p;
};
template <typename T>
MH_DEV AbaUp<T> aba_up_zero()
{
   AbaUp<T> z;
   z.I.A = S3<T>{T(0), T(0), T(0), T(0), T(0), T(0)};
   z.I.L = z.I.A;
   z.I.C = M3<T>{T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
   z.p = SV<T>{V3<T>{T(0), T(0), T(0)}, V3<T>{T(0), T(0), T(0)}};
   return z;
}

template <int LIMB, class CX, typename T>
MH_DEV void x_put_up(const CX &cx, const AbaUp<T> &u)
{
   constexpr int XW = 27;
   x_put<LIMB, XW, 0, CX, T>(cx, u.I.A.xx), x_put<LIMB, XW, 1, CX, T>(cx, u.I.A.xy), x_put<LIMB, XW, 2, CX, T>(cx, u.I.A.xz);
   x_put<LIMB, XW, 3, CX, T>(cx, u.I.A.yy), x_put<LIMB, XW, 4, CX, T>(cx, u.I.A.yz), x_put<LIMB, XW, 5, CX, T>(cx, u.I.A.zz);
   x_put<LIMB, XW, 6, CX, T>(cx, u.I.L.xx), x_put<LIMB, XW, 7, CX, T>(cx, u.I.L.xy), x_put<LIMB, XW, 8, CX, T>(cx, u.I.L.xz);
   x_put<LIMB, XW, 9, CX, T>(cx, u.I.L.yy), x_put<LIMB, XW, 10, CX, T>(cx, u.I.L.yz), x_put<LIMB, XW, 11, CX, T>(cx, u.I.L.zz);
   x_put<LIMB, XW, 12, CX, T>(cx, u.I.C.xx), x_put<LIMB, XW, 13, CX, T>(cx, u.I.C.xy), x_put<LIMB, XW, 14, CX, T>(cx, u.I.C.xz);
   x_put<LIMB, XW, 15, CX, T>(cx, u.I.C.yx), x_put<LIMB, XW, 16, CX, T>(cx, u.I.C.yy), x_put<LIMB, XW, 17, CX, T>(cx, u.I.C.yz);
   x_put<LIMB, XW, 18, CX, T>(cx, u.I.C.zx), x_put<LIMB, XW, 19, CX, T>(cx, u.I.C.zy), x_put<LIMB, XW, 20, CX, T>(cx, u.I.C.zz);
   x_put6<LIMB, XW, 21, CX, T>(cx, u.p);
}
template <int LIMB, class CX, typename T>
MH_DEV AbaUp<T> x_get_up(const CX &cx)
{
   constexpr int XW = 27;
   AbaUp<T> u;
   u.I.A = S3<T>{x_get<LIMB, XW, 0, CX, T>(cx), x_get<LIMB, XW, 1, CX, T>(cx), x_get<LIMB, XW, 2, CX, T>(cx),
                 x_get<LIMB, XW, 3, CX, T>(cx), x_get<LIMB, XW, 4, CX, T>(cx), x_get<LIMB, XW, 5, CX, T>(cx)};
   u.I.L = S3<T>{x_get<LIMB, XW, 6, CX, T>(cx), x_get<LIMB, XW, 7, CX, T>(cx), x_get<LIMB, XW, 8, CX, T>(cx),
                 x_get<LIMB, XW, 9, CX, T>(cx), x_get<LIMB, XW, 10, CX, T>(cx), x_get<LIMB, XW, 11, CX, T>(cx)};
   u.I.C = M3<T>{x_get<LIMB, XW, 12, CX, T>(cx), x_get<LIMB, XW, 13, CX, T>(cx), x_get<LIMB, XW, 14, CX, T>(cx),
                 x_get<LIMB, XW, 15, CX, T>(cx), x_get<LIMB, XW, 16, CX, T>(cx), x_get<LIMB, XW, 17, CX, T>(cx),
                 x_get<LIMB, XW, 18, CX, T>(cx), x_get<LIMB, XW, 19, CX, T>(cx), x_get<LIMB, XW, 20, CX, T>(cx)};
   u.p = x_get6<LIMB, XW, 21, CX, T>(cx);
   return u;
}

template <class TP, int J, typename T, class CX, int MODE = 0>
struct AbaIn
{ // inward sweep (ForwardDynamicsCalculator.java:1085-1254) as a depth-first recursion.  Only (v, cos, sin, qd) of a body
  // stay live while its subtree is walked; everything that depends on the inertia is formed after the children returned.
  // MODE 0: whole subtree.  1: trunk pass of the tree-split kernels (limb roots come from the exchange area).  2: staged trunk, the
  // root body alone (limbs AND sub-trunks come from the exchange area).  3: a late limb of the staged scheme -- as 0, with the
  // workgroup's first barrier between the steps of two of its bodies (Split<TP>::is_cut).
   template <int K>
   static MH_DEV void children(const CX &cx, const SV<T> &v, AbaUp<T> &acc)
   {
      if constexpr (K < Tree<TP>::n_children(J))
      {
         constexpr int C = Tree<TP>::child(J, K);
         AbaUp<T> c;
         if constexpr ((MODE == 1 || MODE == 2) && !Split<TP>::is_trunk(C))
            c = x_get_up<Split<TP>::limb_index(C), CX, T>(cx);
         else if constexpr (MODE == 2)
            c = x_get_up<Split<TP>::sub_slot(C), CX, T>(cx); // staged trunk: the sub-trunk below R was folded by one wave
         else
            c = AbaIn<TP, C, T, CX, MODE>::run(cx, v);
         if constexpr (K == 0)
            acc = c;
         else
         {
            add(acc.I, c.I);
            acc.p = acc.p + c.p;
         }
         children<K + 1>(cx, v, acc);
      }
   }
   static MH_DEV AbaUp<T> run(const CX &cx, const SV<T> &vp)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J];
      constexpr bool HAS_PARENT = TP::parent[J] >= 0;
      constexpr bool LEAF = Tree<TP>::n_children(J) == 0;
      constexpr int DO = Tree<TP>::dof_ofs(J), CO = Tree<TP>::cfg_ofs(J), S0 = Tree<TP>::aba_slot(J);
      const CRef<T, false> c{cx.C + J * MC_STRIDE};
      const JQ<T> jq = spec_joint_read<TYPE, CO, CX, T>(cx);
      const SV<T> vJ = spec_vec<TYPE, DO, 0, CX, T>(cx, true);
      const XF<T> Xb0 = load_xb_j<TP, J, T>(c);
      MH_BODY_FENCE();
      const JX<T> jx = spec_joint_from<TYPE, T>(jq);
      SV<T> v = motion_down(TYPE, jx, Xb0, vp) + vJ;
      AbaUp<T> up = aba_up_zero<T>();
      MH_BODY_FENCE();
      if constexpr (!LEAF)
         children<0>(cx, v, up);
      if constexpr (MODE == 3 && Split<TP>::is_cut(J))
         __syncthreads(); // barrier 1 of the staged trunk: the early limbs of every wave are in the exchange area
      MH_BODY_FENCE();
      if constexpr (!LEAF)
      {
         // The body velocity is formed again after the subtree returned instead of being kept alive across it.  (Keeping it
         // alive is what the code above asks for, but in the two most register-starved variants of the 25-body kernel hipcc
         // 7.2 handed back a clobbered v after the children -- velocity-dependent terms off by a few percent, chains
         // unaffected; re-forming it from the still-live parent velocity is exact and removes 6 long live ranges.)
         SV<T> vJ2 = vJ;
         asm volatile("" : "+v"(vJ2.a.x), "+v"(vJ2.a.y), "+v"(vJ2.a.z), "+v"(vJ2.l.x), "+v"(vJ2.l.y), "+v"(vJ2.l.z));
         v = motion_down(TYPE, jx, load_xb_j<TP, J, T>(c), vp) + vJ2;
      }
      const RI<T> I = load_inertia<T>(c);
      ABI<T> IA = abi_from_rigid(I);
      SV<T> pA = crf(v, mul(I, v));
      if (cx.frow)
         pA = pA - load_fext<T>(c, cx.frow, cx.f_es, cx.meta[J * MI_STRIDE + MI_EXT]);
      if constexpr (!LEAF)
      {
         add(IA, up.I);
         pA = pA + up.p;
      }
      AbaUp<T> out = aba_up_zero<T>();
      if constexpr (TYPE == JT_REVOLUTE || TYPE == JT_PRISMATIC)
      {
         V3<T> ua, ul;
         T D, pz;
         if constexpr (TYPE == JT_REVOLUTE)
         {
            ua = V3<T>{IA.A.xz, IA.A.yz, IA.A.zz}, ul = V3<T>{IA.C.zx, IA.C.zy, IA.C.zz};
            D = IA.A.zz, pz = pA.a.z;
         }
         else
         {
            ua = V3<T>{IA.C.xz, IA.C.yz, IA.C.zz}, ul = V3<T>{IA.L.xz, IA.L.yz, IA.L.zz};
            D = IA.L.zz, pz = pA.l.z;
         }
         const T dinv = T(1) / D;
         const T ud = (cx.in3(DO) - pz) * dinv;
         const V3<T> sa = dinv * ua, sl = dinv * ul;
         cx.st.template put<J, 0>(sa.x), cx.st.template put<J, 1>(sa.y), cx.st.template put<J, 2>(sa.z);
         cx.st.template put<J, 3>(sl.x), cx.st.template put<J, 4>(sl.y), cx.st.template put<J, 5>(sl.z);
         cx.st.template put<J, 6>(ud);
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
            const SV<T> pa = pA + mul(IA, crm(v, vJ)) + SV<T>{ud * ua, ud * ul};
            const XF<T> Xb = load_xb_j<TP, J, T>(c);
            if constexpr (TYPE == JT_REVOLUTE)
            {
               out.p = pa;
               revolute_up(jx, Xb, IA, out.p);
            }
            else
            {
               abi_up(TYPE, jx, Xb, IA);
               out.p = force_up(TYPE, jx, Xb, pa);
            }
            out.I = IA;
         }
      }
      else if constexpr (TYPE == JT_SIXDOF)
      {
         const SV<T> tau = spec_vec<TYPE, DO, 1, CX, T>(cx, true);
         const SV<T> x = spd6_solve(IA, tau - pA);
         cx.st.template put<J, 0>(x.a.x), cx.st.template put<J, 1>(x.a.y), cx.st.template put<J, 2>(x.a.z);
         cx.st.template put<J, 3>(x.l.x), cx.st.template put<J, 4>(x.l.y), cx.st.template put<J, 5>(x.l.z);
         if constexpr (HAS_PARENT)
            out.p = force_up(TYPE, jx, load_xb_j<TP, J, T>(c), tau); // Ia = 0, pa = tau
      }
      else if constexpr (HAS_PARENT)
      { // fixed joint
         const XF<T> Xb = load_xb_j<TP, J, T>(c);
         abi_up(TYPE, jx, Xb, IA);
         out.I = IA;
         out.p = force_up(TYPE, jx, Xb, pA);
      }
      MH_BODY_FENCE();
      return out;
   }
};
template <class TP, int J, typename T, class CX, int MODE = 0>
struct AbaOut
{ // outward sweep (ForwardDynamicsCalculator.java:1259-1310).  MODE 1: a limb hanging off this trunk body is continued only by the
  // wave that owns it (wave-uniform branch); the trunk itself is walked by every wave, its outputs written by wave 0.
   template <int K>
   static MH_DEV void children(const CX &cx, const SV<T> &v, const SV<T> &a)
   {
      if constexpr (K < Tree<TP>::n_children(J))
      {
         constexpr int C = Tree<TP>::child(J, K);
         if constexpr (MODE == 1 && !Split<TP>::is_trunk(C))
         {
            if (cx.wave == Split<TP>::owner(Split<TP>::limb_index(C)))
               AbaOut<TP, C, T, CX, 0>::run(cx, v, a);
         }
         else
            AbaOut<TP, C, T, CX, MODE>::run(cx, v, a);
         children<K + 1>(cx, v, a);
      }
   }
   static MH_DEV void run(const CX &cx, const SV<T> &vp, const SV<T> &ap)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J];
      constexpr bool LEAF = Tree<TP>::n_children(J) == 0;
      constexpr int DO = Tree<TP>::dof_ofs(J), CO = Tree<TP>::cfg_ofs(J), S0 = Tree<TP>::aba_slot(J);
      const CRef<T, false> c{cx.C + J * MC_STRIDE};
      JX<T> jx;
      JQ<T> jq;
      if constexpr (TYPE == JT_REVOLUTE)
      {
         jx.c = cx.st.template get<J, 7>(), jx.s = cx.st.template get<J, 8>(), jx.d = T(0);
      }
      else
         jq = spec_joint_read<TYPE, CO, CX, T>(cx);
      const SV<T> vJ = spec_vec<TYPE, DO, 0, CX, T>(cx, true);
      const XF<T> Xb = load_xb_j<TP, J, T>(c);
      MH_BODY_FENCE();
      if constexpr (TYPE != JT_REVOLUTE)
         jx = spec_joint_from<TYPE, T>(jq);
      const SV<T> v = motion_down(TYPE, jx, Xb, vp) + vJ;
      SV<T> a = motion_down(TYPE, jx, Xb, ap) + crm(v, vJ);
      if constexpr (TYPE == JT_REVOLUTE || TYPE == JT_PRISMATIC)
      {
         const V3<T> sa{cx.st.template get<J, 0>(), cx.st.template get<J, 1>(), cx.st.template get<J, 2>()}, sl{cx.st.template get<J, 3>(), cx.st.template get<J, 4>(), cx.st.template get<J, 5>()};
         const T qdd = cx.st.template get<J, 6>() - (dot(sa, a.a) + dot(sl, a.l));
         if (MODE == 0 || cx.wave == 0)
            cx.out(DO, qdd);
         if constexpr (TYPE == JT_REVOLUTE)
            a.a.z += qdd;
         else
            a.l.z += qdd;
      }
      else if constexpr (TYPE == JT_SIXDOF)
      {
         const SV<T> x{V3<T>{cx.st.template get<J, 0>(), cx.st.template get<J, 1>(), cx.st.template get<J, 2>()}, V3<T>{cx.st.template get<J, 3>(), cx.st.template get<J, 4>(), cx.st.template get<J, 5>()}};
         if (MODE == 0 || cx.wave == 0)
            spec_write<TYPE, DO, CX, T>(cx, x - a);
         a = x;
      }
      if constexpr (CX::bodies)
         if (MODE == 0 || cx.wave == 0)
            spec_body_outputs<J, CX, T>(cx, a, v);
      MH_BODY_FENCE();
      if constexpr (!LEAF)
         children<0>(cx, v, a);
   }
};
template <class TP, typename T, class CX, int MODE = 0, int K = 0>
MH_DEV void aba_roots_in(const CX &cx)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      const V3<T> Z{T(0), T(0), T(0)};
      (void)AbaIn<TP, Tree<TP>::child(-1, K), T, CX, MODE>::run(cx, SV<T>{Z, Z});
      aba_roots_in<TP, T, CX, MODE, K + 1>(cx);
   }
}
template <class TP, typename T, class CX, int MODE = 0, int K = 0>
MH_DEV void aba_roots_out(const CX &cx)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      const V3<T> Z{T(0), T(0), T(0)};
      AbaOut<TP, Tree<TP>::child(-1, K), T, CX, MODE>::run(cx, SV<T>{Z, Z}, SV<T>{cx.a0a, cx.a0l});
      aba_roots_out<TP, T, CX, MODE, K + 1>(cx);
   }
}

// ============================================================================================ CRBA
// Composite-rigid-body mass matrix as one depth-first recursion (CompositeRigidBodyMassMatrixCalculator.java:588-707, 770-798).
// The joint transforms of the ancestors travel down the recursion in a compile-time-sized path (registers), so that the force
// vectors F = Ic S of a body can climb to the root without recomputing or reloading anything but the constant poses.
template <typename T, int D>
struct CrbaPath
{
   JX<T> jx[D > 0 ? D : 1]; // index = depth of the ancestor
};
// Which entries of H are structurally non-zero (joint of the row is an ancestor or descendant of, or the same as, the joint of
// the column) and where the lower-triangle ones live in the packed per-lane LDS image used by the coalesced write-out.
template <class TP>
struct HMap
{
   using TR = Tree<TP>;
   static constexpr int N = TP::N, NV = TR::total_dofs();
   struct Table
   {
      short slot[NV * NV > 0 ? NV * NV : 1] = {}; // packed slot of entry (r, c), -1 = structural zero
      int n_slots = 0;
   };
   // run-time view of the same table for the coalesced write-out of the tree-split kernel (entry -> slot)
   static MH_DEV int slot_at(int e) { return T.slot[e]; }
   static constexpr Table make()
   {
      Table t;
      int joint_of[NV > 0 ? NV : 1] = {};
      for (int j = 0; j < N; j++)
         for (int k = 0; k < TR::ndof(j); k++)
            joint_of[TR::dof_ofs(j) + k] = j;
      for (int r = 0; r < NV; r++)
         for (int c = 0; c <= r; c++)
         {
            // related: joint_of[c] is joint_of[r] or one of its ancestors (dofs are numbered parents-first)
            bool related = false;
            for (int a = joint_of[r]; a >= 0; a = TP::parent[a])
               if (a == joint_of[c])
                  related = true;
            const short s = related ? (short)t.n_slots++ : (short)-1;
            t.slot[r * NV + c] = s;
            t.slot[c * NV + r] = s;
         }
      return t;
   }
   static constexpr Table T = make();
};

// PACK = false: H[dof R][dof C] and its mirror image go straight to global memory (setSymmetricEntry, :841-845).
// PACK = true : the value goes to the lane's packed LDS image; the kernel writes H out afterwards, zeros included, in address order.
template <class TP, int PACK, int R, int C, class CX, typename T>
MH_DEV void h_put(const CX &cx, T v)
{
   if constexpr (PACK == 2) // lane-major image (tree-split kernel): xbase = image + lane * row pitch
      cx.xbase[HMap<TP>::T.slot[R * HMap<TP>::NV + C]] = v;
   else if constexpr (PACK == 1)
      cx.xbase[HMap<TP>::T.slot[R * HMap<TP>::NV + C] * 64] = v;
   else
   {
      const long r = cx.di(R), c = cx.di(C);
      cx.orow[(r * cx.nv + c) * cx.v_es] = v;
      if constexpr (R != C)
         cx.orow[(c * cx.nv + r) * cx.v_es] = v;
   }
}
// limb -> trunk exchange of the tree-split CRBA: the composite inertia (10 scalars) a limb hands up, behind the packed image of H
template <class TP, int LIMB, class CX, typename T>
MH_DEV void xc_put_ri(const CX &cx, const RI<T> &r)
{
   constexpr int S0 = LIMB * 10; // cx.lx = this lane's records in the exchange area (lane-major, odd pitch: Split-independent of the group's width)
   cx.lx[S0 + 0] = r.m, cx.lx[S0 + 1] = r.h.x, cx.lx[S0 + 2] = r.h.y, cx.lx[S0 + 3] = r.h.z;
   cx.lx[S0 + 4] = r.I.xx, cx.lx[S0 + 5] = r.I.xy, cx.lx[S0 + 6] = r.I.xz, cx.lx[S0 + 7] = r.I.yy;
   cx.lx[S0 + 8] = r.I.yz, cx.lx[S0 + 9] = r.I.zz;
}
template <class TP, int LIMB, class CX, typename T>
MH_DEV RI<T> xc_get_ri(const CX &cx)
{
   constexpr int S0 = LIMB * 10;
   RI<T> r;
   r.m = cx.lx[S0 + 0];
   r.h = V3<T>{cx.lx[S0 + 1], cx.lx[S0 + 2], cx.lx[S0 + 3]};
   r.I = S3<T>{cx.lx[S0 + 4], cx.lx[S0 + 5], cx.lx[S0 + 6], cx.lx[S0 + 7], cx.lx[S0 + 8], cx.lx[S0 + 9]};
   return r;
}
// which revolute joints wave W of a tree-split walk evaluates under the limb owner map OWN (0: inverse dynamics / mass matrix, 1: forward
// dynamics): the bodies of its limbs and the trunk bodies above them
template <class TP, int OWN>
struct RneaPreSet
{
   using S = Split<TP>;
   static constexpr bool evaluates(int W, int J)
   {
      for (int k = 0; k < S::n_limbs(); k++)
         if (S::template owner_sel<OWN>(k) == W)
         {
            for (int a = S::limb_root(k); a >= 0; a = TP::parent[a])
               if (a == J)
                  return true; // the limb's root or a trunk body above it
            for (int a = J; a >= 0; a = TP::parent[a])
               if (a == S::limb_root(k))
                  return true; // a body of the limb
         }
      return false;
   }
};
// joint transform of body J for the mass-matrix walks: a revolute joint's (cos, sin) from the slots of a context with RneaPreStore (formed
// in front of the walk: crba_pre_pass), else evaluated here
template <class TP, int J, class CX, typename T>
MH_DEV JX<T> crba_joint(const CX &cx)
{
   if constexpr (CX::rnea_pre && TP::type[J] == JT_REVOLUTE)
   {
      JX<T> jx;
      jx.c = cx.st.template get<J, 0>(), jx.s = cx.st.template get<J, 1>(), jx.d = T(0);
      return jx;
   }
   else
      return spec_joint<TP::type[J], Tree<TP>::cfg_ofs(J), CX, T>(cx);
}
// MODE 1 (tree-split kernel, trunk pass): a child that is the root of a limb is not walked, its composite inertia comes from the
// exchange area where the limb's owner left it (the limb's own columns of H are complete by then).
template <class TP, int J, typename T, class CX, int D, int PACK, int MODE = 0>
struct CrbaSub
{
   using TR = Tree<TP>;
   template <int K>
   static MH_DEV void children(const CX &cx, const CrbaPath<T, D + 1> &path, RI<T> &acc)
   {
      if constexpr (K < TR::n_children(J))
      {
         RI<T> r;
         if constexpr (MODE == 1 && !Split<TP>::is_trunk(TR::child(J, K)))
            r = xc_get_ri<TP, Split<TP>::limb_index(TR::child(J, K)), CX, T>(cx);
         else
            r = CrbaSub<TP, TR::child(J, K), T, CX, D + 1, PACK, MODE>::run(cx, path);
         if constexpr (K == 0)
            acc = r;
         else
            add(acc, r);
         children<K + 1>(cx, path, acc);
      }
   }
   // rows of ancestor A (at depth DA < D) against column COL of joint J; F arrives expressed in A's frame
   template <int A, int COL>
   static MH_DEV void write_ancestor(const CX &cx, const SV<T> &F)
   {
      constexpr int TA = TP::type[A], DA = TR::dof_ofs(A);
      if constexpr (TA == JT_REVOLUTE)
         h_put<TP, PACK, DA, COL, CX, T>(cx, F.a.z);
      else if constexpr (TA == JT_PRISMATIC)
         h_put<TP, PACK, DA, COL, CX, T>(cx, F.l.z);
      else if constexpr (TA == JT_SIXDOF)
      {
         h_put<TP, PACK, DA + 0, COL, CX, T>(cx, F.a.x), h_put<TP, PACK, DA + 1, COL, CX, T>(cx, F.a.y), h_put<TP, PACK, DA + 2, COL, CX, T>(cx, F.a.z);
         h_put<TP, PACK, DA + 3, COL, CX, T>(cx, F.l.x), h_put<TP, PACK, DA + 4, COL, CX, T>(cx, F.l.y), h_put<TP, PACK, DA + 5, COL, CX, T>(cx, F.l.z);
      }
   }
   // climb from the body at depth DC (frame of F) to its parent, write the parent's rows, continue to the root
   template <int DC, int COL>
   static MH_DEV void climb(const CX &cx, const CrbaPath<T, D + 1> &path, SV<T> F)
   {
      if constexpr (DC > 0)
      {
         constexpr int CUR = TR::ancestor_at_depth(J, DC), PAR = TR::ancestor_at_depth(J, DC - 1);
         const CRef<T, false> c{cx.C + CUR * MC_STRIDE};
         F = force_up(TP::type[CUR], path.jx[DC], load_xb_j<TP, CUR, T>(c), F);
         write_ancestor<PAR, COL>(cx, F);
         climb<DC - 1, COL>(cx, path, F);
      }
   }
   template <int K>
   static MH_DEV void columns(const CX &cx, const CrbaPath<T, D + 1> &path, const RI<T> &Ic)
   {
      constexpr int TYPE = TP::type[J], ND = TR::ndof(J), DO = TR::dof_ofs(J);
      if constexpr (K < ND)
      {
         const SV<T> F = mul(Ic, unit_twist<T>(TYPE, K)); // :663-667
         // diagonal block: rows K..ND-1 of column K (the mirror image is written by h_put)
         if constexpr (TYPE == JT_REVOLUTE)
            h_put<TP, PACK, DO, DO, CX, T>(cx, F.a.z);
         else if constexpr (TYPE == JT_PRISMATIC)
            h_put<TP, PACK, DO, DO, CX, T>(cx, F.l.z);
         else
         {
            if constexpr (K <= 0) h_put<TP, PACK, DO + 0, DO + K, CX, T>(cx, F.a.x);
            if constexpr (K <= 1) h_put<TP, PACK, DO + 1, DO + K, CX, T>(cx, F.a.y);
            if constexpr (K <= 2) h_put<TP, PACK, DO + 2, DO + K, CX, T>(cx, F.a.z);
            if constexpr (K <= 3) h_put<TP, PACK, DO + 3, DO + K, CX, T>(cx, F.l.x);
            if constexpr (K <= 4) h_put<TP, PACK, DO + 4, DO + K, CX, T>(cx, F.l.y);
            if constexpr (K <= 5) h_put<TP, PACK, DO + 5, DO + K, CX, T>(cx, F.l.z);
         }
         climb<D, DO + K>(cx, path, F); // :783-792
         columns<K + 1>(cx, path, Ic);
      }
   }
   static MH_DEV RI<T> run(const CX &cx, const CrbaPath<T, D> &up)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J], CO = TR::cfg_ofs(J);
      constexpr bool LEAF = TR::n_children(J) == 0;
      CrbaPath<T, D + 1> path;
#pragma unroll
      for (int d = 0; d < D; d++)
         path.jx[d] = up.jx[d];
      path.jx[D] = crba_joint<TP, J, CX, T>(cx);
      RI<T> acc;
      MH_BODY_FENCE();
      if constexpr (!LEAF)
         children<0>(cx, path, acc);
      MH_BODY_FENCE();
      const CRef<T, false> c{cx.C + J * MC_STRIDE};
      RI<T> Ic = load_inertia<T>(c);
      if constexpr (!LEAF)
         add(Ic, acc);
      columns<0>(cx, path, Ic);
      if constexpr (TP::parent[J] >= 0)
         rigid_up(TYPE, path.jx[D], load_xb_j<TP, J, T>(c), Ic); // :651-661
      MH_BODY_FENCE();
      return Ic;
   }
};
template <class TP, typename T, class CX, int PACK, int MODE = 0, int K = 0>
MH_DEV void crba_roots(const CX &cx)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      CrbaPath<T, 0> empty;
      (void)CrbaSub<TP, Tree<TP>::child(-1, K), T, CX, 0, PACK, MODE>::run(cx, empty);
      crba_roots<TP, T, CX, PACK, MODE, K + 1>(cx);
   }
}
// joint transforms of the trunk ancestors of a limb root (depth 0 .. D-1), for the owner of the limb
template <class TP, int J, typename T, class CX, int D>
MH_DEV void crba_trunk_path(const CX &cx, CrbaPath<T, D> &path)
{ // J = the ancestor at depth D - 1
   if constexpr (D > 0)
   {
      path.jx[D - 1] = crba_joint<TP, J, CX, T>(cx);
      if constexpr (D > 1)
      {
         CrbaPath<T, D - 1> up;
         crba_trunk_path<TP, TP::parent[J], T, CX, D - 1>(cx, up);
#pragma unroll
         for (int d = 0; d < D - 1; d++)
            path.jx[d] = up.jx[d];
      }
   }
}
// limbs K2 >= K of wave W that hang off the same trunk body as limb K: one set of trunk joint transforms serves them all
template <class TP, int W, int K, int K2, typename T, class CX, int D>
MH_DEV void crba_limbs_same_parent(const CX &cx, const CrbaPath<T, D> &path)
{
   using S = Split<TP>;
   if constexpr (K2 < S::n_limbs())
   {
      if constexpr (S::owner_plain(K2) == W && TP::parent[S::limb_root(K2)] == TP::parent[S::limb_root(K)])
         xc_put_ri<TP, K2, CX, T>(cx, CrbaSub<TP, S::limb_root(K2), T, CX, D, 2, 0>::run(cx, path));
      crba_limbs_same_parent<TP, W, K, K2 + 1, T, CX, D>(cx, path);
   }
}
template <class TP, int W, int K>
constexpr bool crba_first_with_parent()
{
   using S = Split<TP>;
   for (int k = 0; k < K; k++)
      if (S::owner_plain(k) == W && TP::parent[S::limb_root(k)] == TP::parent[S::limb_root(K)])
         return false;
   return true;
}
template <class TP, int W, int K, typename T, class CX>
MH_DEV void crba_limbs_of_wave(const CX &cx)
{
   using S = Split<TP>;
   if constexpr (K < S::n_limbs())
   {
      if constexpr (S::owner_plain(K) == W && crba_first_with_parent<TP, W, K>())
      {
         constexpr int R = S::limb_root(K), D = Tree<TP>::depth(R);
         CrbaPath<T, D> path;
         if constexpr (D > 0)
            crba_trunk_path<TP, TP::parent[R], T, CX, D>(cx, path);
         crba_limbs_same_parent<TP, W, K, K, T, CX, D>(cx, path);
      }
      crba_limbs_of_wave<TP, W, K + 1, T, CX>(cx);
   }
}
// the pre-pass of the tree-split mass matrix (context with RneaPreStore): WHICH = 0..3 the pairs wave WHICH needs for its limbs (their
// bodies and the trunk bodies above them, under the inverse dynamics' owner map); WHICH = -1 the trunk's, for the trunk pass
template <class TP, int WHICH, bool FAST, int J, typename T, class CX>
MH_DEV void crba_pre_bodies(const CX &cx, bool &bad)
{
   if constexpr (J < TP::N)
   {
      constexpr bool MINE = WHICH < 0 ? Split<TP>::is_trunk(J) : RneaPreSet<TP, 0>::evaluates(WHICH, J);
      if constexpr (TP::type[J] == JT_REVOLUTE && MINE)
      {
         const T x = cx.q(Tree<TP>::cfg_ofs(J));
         T s, c;
         if constexpr (FAST)
         {
            sincos_fast(x, s, c);
            bad = bad || !sincos_in_fast_range(x);
         }
         else
            sincos_t(x, s, c);
         cx.st.template put<J, 0>(c);
         cx.st.template put<J, 1>(s);
      }
      crba_pre_bodies<TP, WHICH, FAST, J + 1, T, CX>(cx, bad);
   }
}
template <class TP, int WHICH, typename T, class CX>
MH_DEV void crba_pre_pass(const CX &cx)
{
   if constexpr (CX::rnea_pre)
   {
      bool bad = false;
      crba_pre_bodies<TP, WHICH, true, 0, T, CX>(cx, bad);
      if (__builtin_expect(bad, 0))
         crba_pre_bodies<TP, WHICH, false, 0, T, CX>(cx, bad);
   }
}
template <class TP, int W, typename T, class CX>
MH_DEV void split_crba_limbs(const CX &cx)
{
   if constexpr (W < 4)
   {
      if (cx.wave == W)
      {
         crba_pre_pass<TP, W, T, CX>(cx);
         crba_limbs_of_wave<TP, W, 0, T, CX>(cx);
      }
      else
         split_crba_limbs<TP, W + 1, T, CX>(cx);
   }
}

// ============================================================================================ Coriolis matrix
// Mass matrix and Coriolis matrix in one depth-first recursion (CompositeRigidBodyMassMatrixCalculator.java:604-630, 669-768 with
// FactorizedBodyInertia.java; the arithmetic of coriolis_kernel in mh_kernels.h, see there).  Like CrbaSub the joint transforms of the
// ancestors travel down in a compile-time-sized path -- here together with the ancestors' velocities, which the derivative of every
// ancestor's motion subspace needs -- and a subtree hands (Ic, Bc), 10 + 30 scalars, to its parent.  Entries go straight to global
// memory (both matrices zero-filled by the caller; entry (r, c) at (r * nv + c) * f_es of the configuration's block).
template <typename T, int D>
struct CorPath
{
   JX<T> jx[D > 0 ? D : 1]; // index = depth of the ancestor
   SV<T> v[D > 0 ? D : 1];
};
template <typename T>
struct CorUp
{
   RI<T> I;
   FB<T> B;
};
// FAST: identity index maps and AoS matrices -- the offset of an entry is a compile-time constant, an immediate of the store; otherwise it
// is wave-uniform scalar arithmetic on (nv, f_es), which the kernel launders per configuration (hoisted out of the grid-stride loop the
// ~1 200 offsets of the humanoid's two matrices would live in SGPRs: ~2 000 scalar spills)
template <class TP, int J, typename T, class CX, int D, bool FAST>
struct CorSub
{
   using TR = Tree<TP>;
   template <int R, int C>
   static MH_DEV void put(const CX &cx, T *base, T v)
   {
      if constexpr (FAST)
         base[R * TR::total_dofs() + C] = v;
      else
         base[((long)cx.di(R) * cx.nv + cx.di(C)) * cx.f_es] = v;
   }
   template <int K>
   static MH_DEV void children(const CX &cx, const CorPath<T, D + 1> &path, CorUp<T> &acc)
   {
      if constexpr (K < TR::n_children(J))
      {
         const CorUp<T> r = CorSub<TP, TR::child(J, K), T, CX, D + 1, FAST>::run(cx, path);
         if constexpr (K == 0)
            acc = r;
         else
         {
            add(acc.I, r.I);
            add(acc.B, r.B);
         }
         children<K + 1>(cx, path, acc);
      }
   }
   // rows of ancestor A against column COL of joint J; the three momenta arrive expressed in A's frame, G = F3 - v_A x* F2
   template <int A, int COL, int R = 0>
   static MH_DEV void write_ancestor(const CX &cx, const SV<T> &F1, const SV<T> &F2, const SV<T> &G)
   {
      constexpr int TA = TP::type[A], DA = TR::dof_ofs(A);
      if constexpr (R < TR::ndof(A))
      {
         constexpr int E = dof_comp(TA, R);
         const T h = comp(F2, E);
         put<DA + R, COL>(cx, cx.orow, h);
         put<COL, DA + R>(cx, cx.orow, h);
         put<DA + R, COL>(cx, cx.orow2, comp(F1, E)); // C_ik = S_i . F1   (:756)
         put<COL, DA + R>(cx, cx.orow2, comp(G, E));  // C_ki = Sd_i . F2 + S_i . F3   (:757)
         write_ancestor<A, COL, R + 1>(cx, F1, F2, G);
      }
   }
   template <int DC, int COL>
   static MH_DEV void climb(const CX &cx, const CorPath<T, D + 1> &path, SV<T> F1, SV<T> F2, SV<T> F3)
   {
      if constexpr (DC > 0)
      {
         constexpr int CUR = TR::ancestor_at_depth(J, DC), PAR = TR::ancestor_at_depth(J, DC - 1);
         // the ancestor's pose is loaded afresh at every step (laundered pointer): merged across the columns and bodies that climb through
         // the same ancestor it would stay in 24 SGPRs per tree level for the whole subtree -- 2 000 scalar spills on the humanoid
         const T *cp = cx.C + CUR * MC_STRIDE;
         asm volatile("" : "+s"(cp));
         const XF<T> Xb = load_xb_j<TP, CUR, T>(CRef<T, false>{cp});
         F1 = force_up(TP::type[CUR], path.jx[DC], Xb, F1);
         F2 = force_up(TP::type[CUR], path.jx[DC], Xb, F2);
         F3 = force_up(TP::type[CUR], path.jx[DC], Xb, F3);
         write_ancestor<PAR, COL>(cx, F1, F2, F3 - crf(path.v[DC - 1], F2));
         climb<DC - 1, COL>(cx, path, F1, F2, F3);
      }
   }
   template <int K, int R = 0>
   static MH_DEV void own_block(const CX &cx, const SV<T> &F1, const SV<T> &F2)
   {
      constexpr int TYPE = TP::type[J], DO = TR::dof_ofs(J);
      if constexpr (R < TR::ndof(J))
      {
         constexpr int E = dof_comp(TYPE, R);
         put<DO + R, DO + K>(cx, cx.orow, comp(F2, E));  // :698-707
         put<DO + R, DO + K>(cx, cx.orow2, comp(F1, E)); // :709-724
         own_block<K, R + 1>(cx, F1, F2);
      }
   }
   template <int K>
   static MH_DEV void columns(const CX &cx, const CorPath<T, D + 1> &path, const RI<T> &Ic, const FB<T> &Bc)
   {
      constexpr int TYPE = TP::type[J], DO = TR::dof_ofs(J);
      if constexpr (K < TR::ndof(J))
      {
         const SV<T> S = unit_twist<T>(TYPE, K);
         const SV<T> Sd = crm(path.v[D], S);            // :620-626
         const SV<T> F2 = mul(Ic, S);                   // :663-667
         const SV<T> F1 = mul(Ic, Sd) + mul(Bc, S);     // :686-688
         const SV<T> F3 = tmul(Bc, S);                  // :690-691
         own_block<K>(cx, F1, F2);
         climb<D, DO + K>(cx, path, F1, F2, F3);        // :729-768
         columns<K + 1>(cx, path, Ic, Bc);
      }
   }
   static MH_DEV CorUp<T> run(const CX &cx, const CorPath<T, D> &up)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J], CO = TR::cfg_ofs(J), DO = TR::dof_ofs(J);
      constexpr bool LEAF = TR::n_children(J) == 0;
      const CRef<T, false> c{cx.C + J * MC_STRIDE};
      CorPath<T, D + 1> path;
#pragma unroll
      for (int d = 0; d < D; d++)
         path.jx[d] = up.jx[d], path.v[d] = up.v[d];
      path.jx[D] = spec_joint<TYPE, CO, CX, T>(cx);
      const V3<T> Z{T(0), T(0), T(0)};
      SV<T> vp{Z, Z};
      if constexpr (D > 0)
         vp = up.v[D - 1];
      path.v[D] = motion_down(TYPE, path.jx[D], load_xb_j<TP, J, T>(c), vp) + spec_vec<TYPE, DO, 0, CX, T>(cx, true);
      CorUp<T> acc;
      MH_BODY_FENCE();
      if constexpr (!LEAF)
         children<0>(cx, path, acc);
      MH_BODY_FENCE();
      CorUp<T> out;
      out.I = load_inertia<T>(c);
      out.B = fb_from_rigid(out.I, path.v[D]); // :671-673: the body's own inertia, before the children are added
      if constexpr (!LEAF)
      {
         add(out.I, acc.I);
         add(out.B, acc.B);
      }
      if ((cx.own >> (J & 63)) & 1)
         columns<0>(cx, path, out.I, out.B);
      if constexpr (TP::parent[J] >= 0)
      {
         const T *cp = cx.C + J * MC_STRIDE;
         asm volatile("" : "+s"(cp));
         const XF<T> Xb = load_xb_j<TP, J, T>(CRef<T, false>{cp});
         rigid_up(TYPE, path.jx[D], Xb, out.I); // :651-661
         fb_up(TYPE, path.jx[D], Xb, out.B);    // :675-683
      }
      MH_BODY_FENCE();
      return out;
   }
};
template <class TP, typename T, class CX, bool FAST, int K = 0>
MH_DEV void coriolis_roots(const CX &cx)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      CorPath<T, 0> empty;
      (void)CorSub<TP, Tree<TP>::child(-1, K), T, CX, 0, FAST>::run(cx, empty);
      coriolis_roots<TP, T, CX, FAST, K + 1>(cx);
   }
}

// ============================================================================================ centroidal momentum
// Centroidal momentum matrix and convective term as one depth-first recursion (CompositeRigidBodyMassMatrixCalculator.java:801-839; the
// arithmetic of centroidal_kernel in mh_kernels.h, see there): on the way down the body velocities and Coriolis accelerations (WITH_B),
// on the way up composite inertias and wrenches; column k of A is Ic S_k climbed through the ancestors' transforms (CrbaPath) to the root
// body frame and re-expressed in the centroidal frame.
template <typename T>
struct CentUp
{
   RI<T> I;
   SV<T> f;
};
template <typename T, class BASE>
struct CentCtx : BASE
{
   XF<T> xf;  // centroidal frame -> root body frame
   T *arow;   // this configuration's A: entry (k, col) at (k * nv + col) * a_es
   long a_es;
};
template <class TP, int J, typename T, class CX, int D, bool WITH_B>
struct CentSub
{
   using TR = Tree<TP>;
   template <int K>
   static MH_DEV void children(const CX &cx, const CrbaPath<T, D + 1> &path, const SV<T> &v, const SV<T> &a, CentUp<T> &acc)
   {
      if constexpr (K < TR::n_children(J))
      {
         const CentUp<T> r = CentSub<TP, TR::child(J, K), T, CX, D + 1, WITH_B>::run(cx, path, v, a);
         if constexpr (K == 0)
            acc = r;
         else
         {
            add(acc.I, r.I);
            acc.f = acc.f + r.f;
         }
         children<K + 1>(cx, path, v, a, acc);
      }
   }
   // from the frame after the ancestor at depth DC up to the root body frame
   template <int DC>
   static MH_DEV SV<T> to_root(const CX &cx, const CrbaPath<T, D + 1> &path, SV<T> F)
   {
      constexpr int CUR = TR::ancestor_at_depth(J, DC);
      const T *cp = cx.C + CUR * MC_STRIDE;
      asm volatile("" : "+s"(cp)); // reloaded per step (see CorSub::climb)
      F = force_up(TP::type[CUR], path.jx[DC], load_xb_j<TP, CUR, T>(CRef<T, false>{cp}), F);
      if constexpr (DC > 0)
         return to_root<DC - 1>(cx, path, F);
      else
         return F;
   }
   template <int K>
   static MH_DEV void columns(const CX &cx, const CrbaPath<T, D + 1> &path, const RI<T> &Ic)
   {
      constexpr int TYPE = TP::type[J], DO = TR::dof_ofs(J);
      if constexpr (K < TR::ndof(J))
      {
         const SV<T> F = to_root<D>(cx, path, mul(Ic, unit_twist<T>(TYPE, K))); // :663-667, 783-792, 805
         // root body frame -> centroidal frame: f' = R^T f ; n' = R^T (n - p x f)
         const V3<T> fl = tmul(cx.xf.R, F.l), fa = tmul(cx.xf.R, F.a - cross(cx.xf.p, F.l));
         const long col = cx.di(DO + K), nv = cx.nv, es = cx.a_es;
         cx.arow[(0 * nv + col) * es] = fa.x, cx.arow[(1 * nv + col) * es] = fa.y, cx.arow[(2 * nv + col) * es] = fa.z;
         cx.arow[(3 * nv + col) * es] = fl.x, cx.arow[(4 * nv + col) * es] = fl.y, cx.arow[(5 * nv + col) * es] = fl.z;
         columns<K + 1>(cx, path, Ic);
      }
   }
   static MH_DEV CentUp<T> run(const CX &cx, const CrbaPath<T, D> &up, const SV<T> &vp, const SV<T> &ap)
   {
      MH_BODY_FENCE();
      constexpr int TYPE = TP::type[J], CO = TR::cfg_ofs(J), DO = TR::dof_ofs(J);
      constexpr bool LEAF = TR::n_children(J) == 0;
      const CRef<T, false> c{cx.C + J * MC_STRIDE};
      CrbaPath<T, D + 1> path;
#pragma unroll
      for (int d = 0; d < D; d++)
         path.jx[d] = up.jx[d];
      path.jx[D] = spec_joint<TYPE, CO, CX, T>(cx);
      const V3<T> Z{T(0), T(0), T(0)};
      SV<T> v{Z, Z}, a{Z, Z};
      CentUp<T> out;
      out.I = load_inertia<T>(c);
      out.f = SV<T>{Z, Z};
      if constexpr (WITH_B)
      {
         const XF<T> Xb = load_xb_j<TP, J, T>(c);
         const SV<T> vJ = spec_vec<TYPE, DO, 0, CX, T>(cx, true);
         v = motion_down(TYPE, path.jx[D], Xb, vp) + vJ;
         a = motion_down(TYPE, path.jx[D], Xb, ap) + crm(v, vJ);  // :826-831
         out.f = mul(out.I, a) + crf(v, mul(out.I, v));           // :833
      }
      CentUp<T> acc;
      MH_BODY_FENCE();
      if constexpr (!LEAF)
         children<0>(cx, path, v, a, acc);
      MH_BODY_FENCE();
      if constexpr (!LEAF)
      {
         add(out.I, acc.I);
         out.f = out.f + acc.f;
      }
      if ((cx.own >> (J & 63)) & 1)
         columns<0>(cx, path, out.I);
      {
         const T *cp = cx.C + J * MC_STRIDE;
         asm volatile("" : "+s"(cp));
         const XF<T> Xb = load_xb_j<TP, J, T>(CRef<T, false>{cp});
         rigid_up(TYPE, path.jx[D], Xb, out.I);
         out.f = force_up(TYPE, path.jx[D], Xb, out.f);
      }
      MH_BODY_FENCE();
      return out;
   }
};
template <class TP, typename T, class CX, bool WITH_B, int K = 0>
MH_DEV void centroidal_roots(const CX &cx, CentUp<T> &total)
{
   if constexpr (K < Tree<TP>::n_children(-1))
   {
      const V3<T> Z{T(0), T(0), T(0)};
      CrbaPath<T, 0> empty;
      const CentUp<T> r = CentSub<TP, Tree<TP>::child(-1, K), T, CX, 0, WITH_B>::run(cx, empty, SV<T>{Z, Z}, SV<T>{Z, Z});
      add(total.I, r.I);
      total.f = total.f + r.f;
      centroidal_roots<TP, T, CX, WITH_B, K + 1>(cx, total);
   }
}

// ============================================================================================ kernels
// Coalesced copy of the wave's rows of q, qd and qdd|tau (contiguous blocks of the AoS matrices) into LDS.  ALL loads are
// issued before the first LDS write, so the whole staging costs one memory round trip (about a microsecond) instead of one
// per chunk; 64 * (NQ + 2 NV) elements = NQ + 2 NV loads per lane held in registers for that moment.
template <typename T, int NQ, int NV, int NT = 64>
MH_DEV void wave_stage_in(lds_ptr<T> lq, lds_ptr<T> lqd, lds_ptr<T> lx, const T *q, const T *qd, const T *x, int rows)
{ // NT = threads of the workgroup taking part (64: one wave; 256: the four waves of a tree-split group)
   constexpr int UQ = (64 * NQ + NT - 1) / NT, UV = (64 * NV + NT - 1) / NT;
   T rq[UQ], rd[UV], rx[UV];
   const int nq = rows * NQ, nv = rows * NV, t = threadIdx.x;
#pragma unroll
   for (int u = 0; u < UQ; u++)
      rq[u] = t + NT * u < nq ? q[t + NT * u] : T(0);
#pragma unroll
   for (int u = 0; u < UV; u++)
      rd[u] = t + NT * u < nv ? qd[t + NT * u] : T(0);
#pragma unroll
   for (int u = 0; u < UV; u++)
      rx[u] = t + NT * u < nv ? x[t + NT * u] : T(0);
#pragma unroll
   for (int u = 0; u < UQ; u++)
      if (t + NT * u < 64 * NQ)
         lq[t + NT * u] = rq[u];
#pragma unroll
   for (int u = 0; u < UV; u++)
      if (t + NT * u < 64 * NV)
         lqd[t + NT * u] = rd[u];
#pragma unroll
   for (int u = 0; u < UV; u++)
      if (t + NT * u < 64 * NV)
         lx[t + NT * u] = rx[u];
}
template <typename T, int NT = 64>
MH_DEV void wave_copy_out(T *dst, lds_ptr<T> src, int n)
{
   for (int i = threadIdx.x; i < n; i += NT)
      dst[i] = src[i];
}

template <typename T, class CX>
MH_DEV void fill_ctx(CX &cx, const Args<T> &A, long cfg)
{
   // The model pointers are laundered once per configuration: otherwise loop-invariant code motion lifts all N * 34
   // constant loads out of the grid-stride loop, which overflows the 102 SGPRs and turns every use into a v_readlane.
   const void *pc = A.m.consts;
   const int *pd = A.m.dof_map, *pq = A.m.cfg_map, *pm = A.m.meta;
   asm volatile("" : "+s"(pc), "+s"(pd), "+s"(pq), "+s"(pm));
   cx.C = (const T *)pc;
   cx.dof_map = as_const(pd), cx.cfg_map = as_const(pq), cx.meta = as_const(pm);
   cx.qrow = A.q + cfg * A.q_bs;
   cx.qdrow = A.qd + cfg * A.v_bs;
   cx.in3row = A.in3 + cfg * A.v_bs;
   cx.frow = A.fext ? A.fext + cfg * A.f_bs : nullptr;
   cx.orow = A.out + cfg * A.v_bs;
   cx.bacc = A.body_acc ? A.body_acc + cfg * A.f_bs : nullptr;
   cx.btw = A.body_twist ? A.body_twist + cfg * A.f_bs : nullptr;
   cx.q_es = A.q_es, cx.v_es = A.v_es, cx.f_es = A.f_es;
   cx.a0a = V3<T>{A.rax, A.ray, A.raz}, cx.a0l = V3<T>{-A.gx, -A.gy, -A.gz};
   cx.coriolis = A.coriolis, cx.accel = A.accel;
}

// Touches every 64-byte line of the model constants and index maps with one scalar load each, all in flight together, and waits
// for them.  Round 1 added it so that the per-body s_load batches would hit the scalar cache.  Round 3 measured what it costs with
// real-time stamps around it (tools/exp_zv_probe.py): 6.4 us per launch on the 25-body humanoid (144 lines; the scalar cache takes its
// misses a few at a time) -- and the walk behind it is no faster for it (8.3 vs 8.7 us).  Off by default: the fused launch at B = 4096
// went from 25.9 to 18.1 us per step with this alone.  -DMH_WARM_SCALAR_CACHE=1 brings it back for A/B measurements.
#ifndef MH_WARM_SCALAR_CACHE
#define MH_WARM_SCALAR_CACHE 0
#endif
MH_DEV void warm_scalar_cache(const void *p, int bytes)
{
   if constexpr (!MH_WARM_SCALAR_CACHE)
      return;
   typedef const int __attribute__((address_space(4))) * cip;
   cip a = (cip)(unsigned long long)p;
   int acc = 0;
   for (int ofs = 0; ofs < bytes; ofs += 64)
      acc += a[ofs / 4];
   asm volatile("" ::"s"(acc));
}

// One wave's share of a batch.  ALGO: 0 = RNEA, 1 = ABA.
// (cos, sin) of EVERY revolute joint of the tree into the slots of a context with RneaPreStore (whole-tree inverse dynamics): the
// straight-line fast path for all of them, `sincos_t` for all of them behind one branch when an angle is outside its range
template <class TP, bool FAST, int J, typename T, class CX>
MH_DEV void rnea_pre_all_bodies(const CX &cx, bool &bad)
{
   if constexpr (J < TP::N)
   {
      if constexpr (TP::type[J] == JT_REVOLUTE)
      {
         const T x = cx.q(Tree<TP>::cfg_ofs(J));
         T s, c;
         if constexpr (FAST)
         {
            sincos_fast(x, s, c);
            bad = bad || !sincos_in_fast_range(x);
         }
         else
            sincos_t(x, s, c);
         cx.st.template put<J, 0>(c);
         cx.st.template put<J, 1>(s);
      }
      rnea_pre_all_bodies<TP, FAST, J + 1, T, CX>(cx, bad);
   }
}
template <class TP, typename T, class CX>
MH_DEV void rnea_pre_all(const CX &cx)
{
   if constexpr (CX::rnea_pre)
   {
      bool bad = false;
      rnea_pre_all_bodies<TP, true, 0, T, CX>(cx, bad);
      if (__builtin_expect(bad, 0))
         rnea_pre_all_bodies<TP, false, 0, T, CX>(cx, bad);
   }
}
template <class TP, typename T, int ALGO, bool IO_LDS, bool IDENT, bool ST_LDS>
MH_DEV void spec_wave(const Args<T> &A, long wave, long nwaves, lds_ptr<T> lds)
{
   // (the whole-tree inverse dynamics -- chains, which have no tree split -- forms the pairs of ALL its joints in front of the walk, like the
   // tree-split kernels: RneaPreStore, rnea_pre_all below)
   using CX = Ctx<T, IO_LDS, IDENT, std::conditional_t<ALGO == 0 && MH_RNEA_PRE != 0, RneaPreStore<TP>, WholeStore<TP, ST_LDS ? ST_LDS_KIND : ST_GLOBAL_KIND>>>;
   const int nq = A.m.nq, nv = A.m.nv;
   // LDS map: [64][nq] q | [64][nv] qd | [64][nv] qdd or tau, overwritten by the result | hand-over slots [slot][64]
   const lds_ptr<T> lq = lds, lqd = lq + (IO_LDS ? 64 * nq : 0), lx = lqd + (IO_LDS ? 64 * nv : 0), lst = lx + (IO_LDS ? 64 * nv : 0);
   warm_scalar_cache(A.m.consts, A.m.n * MC_STRIDE * (int)sizeof(T));
   if constexpr (!IDENT)
   {
      warm_scalar_cache(A.m.dof_map, nv * 4);
      warm_scalar_cache(A.m.cfg_map, nq * 4);
   }
#ifdef MH_PROBE // experiment builds: 100 MHz real-time stamps per 64-configuration slice, written behind the B * nv results (tools/exp_c2_floor.py)
#define MH_WSTAMP(k)                                                                                                                       \
   do                                                                                                                                      \
   {                                                                                                                                       \
      const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                                                      \
      if (threadIdx.x == 0)                                                                                                                \
         ((unsigned long long *)(A.out + A.B * nv))[(cfg0 / 64) * 32 + (k)] = t_;                                                          \
   } while (0)
#else
#define MH_WSTAMP(k)
#endif
   for (long cfg0 = wave * 64; cfg0 < A.B; cfg0 += nwaves * 64)
   {
      const int rows = (int)(A.B - cfg0 < 64 ? A.B - cfg0 : 64);
      MH_WSTAMP(0);
      if constexpr (IO_LDS)
      {
         wave_stage_in<T, Tree<TP>::total_cfgs(), Tree<TP>::total_dofs()>(lq, lqd, lx, A.q + cfg0 * nq, A.qd + cfg0 * nv, A.in3 + cfg0 * nv, rows);
         __syncthreads();
      }
      MH_WSTAMP(1);
      if ((int)threadIdx.x < rows)
      {
         CX cx;
         fill_ctx<T>(cx, A, cfg0 + threadIdx.x);
         cx.lq = lq + threadIdx.x * nq, cx.lqd = lqd + threadIdx.x * nv, cx.lx = lx + threadIdx.x * nv;
         cx.lo = cx.lx;
         cx.wave = 0, cx.xbase = lst;
         cx.st.lbase = lst + threadIdx.x;
         cx.st.gbase = A.ws;
         cx.st.stride = A.ws_stride, cx.st.lane = wave * 64 + threadIdx.x;
         asm volatile("" : "+v"(cx.st.lane)); // per configuration: keeps the N * 9 slot addresses from being hoisted out of the loop
         if constexpr (ALGO == 0)
         {
            rnea_pre_all<TP, T, CX>(cx);
            rnea_roots<TP, T, CX>(cx);
         }
         else
         {
            aba_roots_in<TP, T, CX>(cx);
            MH_WSTAMP(4);
            // The hand-over store must really be memory, and the outward sweep must recompute the body velocities from
            // re-read inputs: if the compiler recognises values (or addresses) of the inward sweep it keeps them alive across
            // the turn-around -- 6 N velocities, N * 9 slot addresses -- and spills kilobytes per lane to scratch.
            asm volatile("" ::: "memory");
            asm volatile("" : "+v"(cx.qrow), "+v"(cx.qdrow), "+v"(cx.lq), "+v"(cx.lqd), "+v"(cx.st.lbase), "+v"(cx.st.lane));
            aba_roots_out<TP, T, CX>(cx);
         }
      }
      MH_WSTAMP(2);
      if constexpr (IO_LDS)
      {
         __syncthreads();
         wave_copy_out<T>(A.out + cfg0 * nv, lx, rows * nv);
         __syncthreads();
      }
      MH_WSTAMP(3);
   }
}

// One wave per workgroup.
template <class TP, typename T, int ALGO, bool IO_LDS, bool IDENT, bool ST_LDS>
__global__ void __launch_bounds__(64) spec_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   spec_wave<TP, T, ALGO, IO_LDS, IDENT, ST_LDS>(A, blockIdx.x, gridDim.x, (lds_ptr<T>)lds_raw);
}

// Fused RNEA + ABA for small batches: the first half of the grid computes tau = RNEA(q, qd, qdd), the second half
// qdd = ABA(q, qd, tau_in) on the same configurations.  The two jobs are independent, so a batch that cannot fill the
// device on its own (4096 configurations = 64 waves for 256 CUs) runs both side by side in ONE launch.
// RNEA rows are staged in LDS; ABA reads its rows directly and keeps its hand-over store in LDS.
template <class TP, typename T, bool IDENT>
__global__ void __launch_bounds__(64) spec_fused_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   const long half = gridDim.x / 2;
   if ((long)blockIdx.x < half)
      spec_wave<TP, T, 0, true, IDENT, false>(A, blockIdx.x, half, (lds_ptr<T>)lds_raw);
   else
   {
      Args<T> A2 = A;
      A2.in3 = A.in3b, A2.out = A.outb;
      spec_wave<TP, T, 1, false, IDENT, true>(A2, blockIdx.x - half, half, (lds_ptr<T>)lds_raw);
   }
}

// ============================================================================================ tree-split kernels
// Four waves of a 256-thread workgroup walk the SAME 64 configurations (lane = threadIdx.x & 63): every wave walks the trunk,
// each limb (leg, arm, ...) is walked by the one wave that owns it, and what a limb hands to the trunk crosses waves through a
// small LDS exchange area behind one workgroup barrier.  The serial chain a wave executes shrinks from N bodies to
// (trunk + its longest limb); the batch occupies 4x as many SIMDs.  Built for small batches, where latency is everything.
// The limbs of one owner wave W, in limb order.  PC = trunk body whose velocity (and acceleration) the previous limb of this wave
// hung from (-2: none yet): limbs sharing a parent -- an arm and the neck on the chest -- walk the trunk down to it once.
// ---- the pre-pass of a context with RneaPreStore: (cos, sin) of every revolute joint wave W evaluates -- the bodies of its limbs and the
//      trunk bodies above them -- formed together, in front of the walks.  FAST: the straight-line fast path for all of them (six to
//      nine independent chains that interleave), `bad` = an angle outside its range; the caller repeats with FAST = false behind ONE branch.
template <class TP, int W, int OWN, bool FAST, int J, typename T, class CX>
MH_DEV void rnea_pre_bodies(const CX &cx, bool &bad)
{
   if constexpr (J < TP::N)
   {
      // (the fused forward dynamics: the limbs' bodies only -- its trunk slots lie over the rows of q while the inverse dynamics runs)
      constexpr int PS = CX::rnea_pre ? 0 : 7;
      if constexpr (TP::type[J] == JT_REVOLUTE && RneaPreSet<TP, OWN>::evaluates(W, J) && (CX::rnea_pre || !Split<TP>::is_trunk(J)))
      {
         const T x = cx.q(Tree<TP>::cfg_ofs(J));
         T s, c;
         if constexpr (FAST)
         {
            sincos_fast(x, s, c);
            bad = bad || !sincos_in_fast_range(x);
         }
         else
            sincos_t(x, s, c);
         cx.st.template put<J, PS>(c);
         cx.st.template put<J, PS + 1>(s);
      }
      rnea_pre_bodies<TP, W, OWN, FAST, J + 1, T, CX>(cx, bad);
   }
}
template <class TP, int W, int OWN, typename T, class CX>
MH_DEV void rnea_pre_pass(const CX &cx)
{
   if constexpr (CX::rnea_pre || (CX::csmode == 3 && MH_ZVF_PRE))
   {
      bool bad = false;
      rnea_pre_bodies<TP, W, OWN, true, 0, T, CX>(cx, bad);
      if (__builtin_expect(bad, 0)) // an angle of 2^19 rad or more among them: once more, every pair through the full sincos
         rnea_pre_bodies<TP, W, OWN, false, 0, T, CX>(cx, bad);
   }
}
template <class TP, int W, int K, int PC, typename T, class CX, int OWN = 0>
MH_DEV void split_rnea_limbs_of(const CX &cx, SV<T> &vp, SV<T> &ap)
{
   using S = Split<TP>;
   if constexpr (K < S::n_limbs())
   {
      if constexpr (S::template owner_sel<OWN>(K) == W)
      {
         constexpr int R = S::limb_root(K), P = TP::parent[R];
         if constexpr (P != PC)
         {
            const V3<T> Z{T(0), T(0), T(0)};
            vp = SV<T>{Z, Z}, ap = SV<T>{cx.a0a, cx.a0l};
            if constexpr (P >= 0)
               trunk_va<TP, P, T, CX, K, OWN>(cx, vp, ap);
         }
         x_put6<K, 6, 0, CX, T>(cx, RneaSub<TP, R, T, CX, 0>::run(cx, vp, ap));
         split_rnea_limbs_of<TP, W, K + 1, P, T, CX, OWN>(cx, vp, ap);
      }
      else
         split_rnea_limbs_of<TP, W, K + 1, PC, T, CX, OWN>(cx, vp, ap);
   }
}
template <class TP, int W, int K, int PC, typename T, class CX>
MH_DEV void split_aba_limbs_of(const CX &cx, SV<T> &vp)
{
   using S = Split<TP>;
   if constexpr (K < S::n_limbs())
   {
      if constexpr (S::owner(K) == W)
      {
         constexpr int R = S::limb_root(K), P = TP::parent[R];
         if constexpr (P != PC)
         {
            const V3<T> Z{T(0), T(0), T(0)};
            vp = SV<T>{Z, Z};
            if constexpr (P >= 0)
               vp = trunk_v<TP, P, T, CX>(cx);
         }
         x_put_up<K, CX, T>(cx, AbaIn<TP, R, T, CX, 0>::run(cx, vp));
         split_aba_limbs_of<TP, W, K + 1, P, T, CX>(cx, vp);
      }
      else
         split_aba_limbs_of<TP, W, K + 1, PC, T, CX>(cx, vp);
   }
}
template <class TP, int W, typename T, class CX, int OWN = 0>
MH_DEV void split_rnea_limbs(const CX &cx)
{
   if constexpr (W < 4)
   {
      if (cx.wave == W)
      {
         rnea_pre_pass<TP, W, OWN, T, CX>(cx);
         const V3<T> Z{T(0), T(0), T(0)};
         SV<T> vp{Z, Z}, ap{Z, Z};
         split_rnea_limbs_of<TP, W, 0, -2, T, CX, OWN>(cx, vp, ap);
      }
      else
         split_rnea_limbs<TP, W + 1, T, CX, OWN>(cx);
   }
}
template <class TP, int W, typename T, class CX>
MH_DEV void split_aba_limbs(const CX &cx)
{
   if constexpr (W < 4)
   {
      if (cx.wave == W)
      {
         const V3<T> Z{T(0), T(0), T(0)};
         SV<T> vp{Z, Z};
         split_aba_limbs_of<TP, W, 0, -2, T, CX>(cx, vp);
      }
      else
         split_aba_limbs<TP, W + 1, T, CX>(cx);
   }
}

// ---- staged trunk (Split<TP>::make_stages): the limbs of wave W, early (LATE = 0) or late (LATE = 1), in limb order; the late limb
//      that carries the wave's cut runs in MODE 3 (barrier 1 inside)
template <class TP, int W, int K, int PC, int LATE, typename T, class CX>
MH_DEV void staged_limbs_of(const CX &cx, SV<T> &vp)
{
   using S = Split<TP>;
   if constexpr (K < S::n_limbs())
   {
      if constexpr (S::owner(K) == W && (S::is_late(K) ? 1 : 0) == LATE)
      {
         constexpr int R = S::limb_root(K), P = TP::parent[R];
         if constexpr (P != PC)
         {
            const V3<T> Z{T(0), T(0), T(0)};
            vp = SV<T>{Z, Z};
            if constexpr (P >= 0)
               vp = trunk_v<TP, P, T, CX>(cx);
         }
         x_put_up<K, CX, T>(cx, AbaIn<TP, R, T, CX, (S::cut_limb(W) == K ? 3 : 0)>::run(cx, vp));
         staged_limbs_of<TP, W, K + 1, P, LATE, T, CX>(cx, vp);
      }
      else
         staged_limbs_of<TP, W, K + 1, PC, LATE, T, CX>(cx, vp);
   }
}
// the sub-trunks wave W folds between the two barriers
template <class TP, int W, int I, typename T, class CX>
MH_DEV void staged_subtrunks_of(const CX &cx)
{
   using S = Split<TP>;
   if constexpr (I < S::n_sub())
   {
      constexpr int ST = S::sub_top(I);
      if constexpr (S::sub_owner(ST) == W)
      {
         const SV<T> vr = trunk_v<TP, S::root(), T, CX>(cx);
         x_put_up<S::sub_slot(ST), CX, T>(cx, AbaIn<TP, ST, T, CX, 1>::run(cx, vr));
      }
      staged_subtrunks_of<TP, W, I + 1, T, CX>(cx);
   }
}
// Phases 1 and 2 of wave W.  Every wave passes barrier 1 exactly once: inside its cut limb, or explicitly after its early limbs.
template <class TP, int W, typename T, class CX>
MH_DEV void split_aba_staged(const CX &cx)
{
   using S = Split<TP>;
   if constexpr (W < 4)
   {
      if (cx.wave == W)
      {
         const V3<T> Z{T(0), T(0), T(0)};
         SV<T> vp{Z, Z};
         staged_limbs_of<TP, W, 0, -2, 0, T, CX>(cx, vp);
         if constexpr (S::cut_limb(W) < 0)
            __syncthreads();
         staged_limbs_of<TP, W, 0, -2, 1, T, CX>(cx, vp);
         staged_subtrunks_of<TP, W, 0, T, CX>(cx);
      }
      else
         split_aba_staged<TP, W + 1, T, CX>(cx);
   }
}

// Fused simulation step: lane = configuration of the slice, the joints are dealt round-robin to the four waves; offsets are
// compile-time constants (identity index maps), so a 1-DoF joint is three LDS reads, two FMAs and two LDS writes.
template <class TP, int J, typename T>
MH_DEV void integrate_rows(int wave, lds_ptr<T> q, lds_ptr<T> v, lds_ptr<T> a, T dt, T hdd)
{
   using TR = Tree<TP>;
   if constexpr (J < TP::N)
   {
      if ((J & 3) == wave)
      {
         constexpr int TYPE = TP::type[J], D0 = TR::dof_ofs(J), C0 = TR::cfg_ofs(J);
         if constexpr (TYPE == JT_REVOLUTE || TYPE == JT_PRISMATIC)
         { // MultiBodySystemStateIntegrator.java:433-441, 710-733
            const T q0 = q[C0], v0 = v[D0], a0 = a[D0];
            q[C0] = hdd * a0 + dt * v0 + q0;
            v[D0] = dt * a0 + v0;
         }
         else if constexpr (TYPE == JT_SIXDOF)
         { // :503-575
            T qx = q[C0], qy = q[C0 + 1], qz = q[C0 + 2], qs = q[C0 + 3];
            V3<T> p{q[C0 + 4], q[C0 + 5], q[C0 + 6]}, w{v[D0], v[D0 + 1], v[D0 + 2]}, vl{v[D0 + 3], v[D0 + 4], v[D0 + 5]};
            const V3<T> al{a[D0], a[D0 + 1], a[D0 + 2]}, ac{a[D0 + 3], a[D0 + 4], a[D0 + 5]};
            integrate_sixdof<T>(dt, hdd, qx, qy, qz, qs, p, w, vl, al, ac, nullptr);
            q[C0] = qx, q[C0 + 1] = qy, q[C0 + 2] = qz, q[C0 + 3] = qs, q[C0 + 4] = p.x, q[C0 + 5] = p.y, q[C0 + 6] = p.z;
            v[D0] = w.x, v[D0 + 1] = w.y, v[D0 + 2] = w.z, v[D0 + 3] = vl.x, v[D0 + 4] = vl.y, v[D0 + 5] = vl.z;
         }
      }
      integrate_rows<TP, J + 1, T>(wave, q, v, a, dt, hdd);
   }
}

// One workgroup's share of a batch.  ALGO: 0 = RNEA, 1 = ABA.  IO_LDS: the 64 rows of q, qd, qdd|tau are staged once in LDS
// by all 256 threads (the four waves share them) and the results leave through LDS as one coalesced copy.
template <class TP, typename T, int ALGO, bool IDENT, bool IO_LDS, bool BODIES = false>
MH_DEV void split_group(const Args<T> &A, long group, long ngroups, lds_ptr<T> lds)
{
   using S = Split<TP>;
   using CX = Ctx<T, IO_LDS, IDENT, std::conditional_t<ALGO == 0 && MH_RNEA_PRE != 0, RneaPreStore<TP>, SplitStore<TP>>, BODIES>;
   constexpr int XW = ALGO == 0 ? 6 : 27;
   const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const int lane = threadIdx.x & 63;
   const int nq = A.m.nq, nv = A.m.nv;
   // LDS map: exchange [n_limbs * XW][64] | trunk hand-over slots [TRUNK_SLOTS][64] (ABA) or parked trunk wrenches [RNEA_TRUNK_SLOTS][64] (RNEA) | [64][nq] q | [64][nv] qd | [64][nv] qdd|tau -> result
   // Forward dynamics writes its results to rows of their own (lres): every wave reads the trunk's tau entries in its own inward trunk
   // pass while wave 0, which writes the trunk's accelerations, may already be in its outward sweep -- written in place over tau they
   // could be read back as efforts by a wave that is late (found in round 3 on the bias-split kernel, where the window is wide; here it
   // is a few hundred instructions of the root body's step, never observed, closed all the same).  Inverse dynamics keeps writing in
   // place: an effort is written by the one wave that read the acceleration in that slot.
   const lds_ptr<T> lxc = lds, lst = lxc + S::n_limbs() * XW * 64, lq = lst + (ALGO == 1 ? S::TRUNK_SLOTS : S::RNEA_TRUNK_SLOTS) * 64,
                    lqd = lq + 64 * nq, lx = lqd + 64 * nv, lres = ALGO == 1 && IO_LDS ? lx + 64 * nv : lx;
#ifdef MH_PROBE // experiment builds: s_memtime stamps per wave and phase, written behind the B * nv results (tools/exp_probe.py)
#define MH_STAMP(k)                                                                                                                        \
   do                                                                                                                                      \
   {                                                                                                                                       \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                                          \
      if (lane == 0)                                                                                                                       \
         ((unsigned long long *)(A.out + A.B * nv))[(cfg0 / 64) * 32 + wave * 8 + (k)] = t_;                                               \
   } while (0)
#else
#define MH_STAMP(k)
#endif
   warm_scalar_cache(A.m.consts, A.m.n * MC_STRIDE * (int)sizeof(T));
   for (long cfg0 = group * 64; cfg0 < A.B; cfg0 += ngroups * 64)
   {
      const int rows = (int)(A.B - cfg0 < 64 ? A.B - cfg0 : 64);
      const bool active = lane < rows;
      MH_STAMP(0);
      if constexpr (IO_LDS)
      {
         wave_stage_in<T, Tree<TP>::total_cfgs(), Tree<TP>::total_dofs(), 256>(lq, lqd, lx, A.q + cfg0 * nq, A.qd + cfg0 * nv, A.in3 + cfg0 * nv, rows);
         __syncthreads();
      }
      CX cx;
      fill_ctx<T>(cx, A, active ? cfg0 + lane : cfg0);
      cx.lq = lq + lane * nq, cx.lqd = lqd + lane * nv, cx.lx = lx + lane * nv, cx.lo = lres + lane * nv;
      cx.wave = wave;
      cx.xbase = lxc + lane;
      cx.st.lbase = lst + lane;
      cx.st.gbase = nullptr, cx.st.stride = 0, cx.st.lane = 0;
      MH_STAMP(1);
      if (active)
      { // (lane 0 of every wave is active in every slice, so each wave does reach the barrier the staged ABA carries in here)
         if constexpr (ALGO == 0)
            split_rnea_limbs<TP, 0, T, CX>(cx);
         else if constexpr (S::staged())
            split_aba_staged<TP, 0, T, CX>(cx);
         else
            split_aba_limbs<TP, 0, T, CX>(cx);
      }
      MH_STAMP(2);
      __syncthreads(); // every limb's hand-up is in the exchange area
      MH_STAMP(3);
      if (active)
      {
         if constexpr (ALGO == 0)
         {
            if (wave == 0)
               rnea_trunk_roots<TP, T, CX>(cx);
         }
         else
         {
            aba_roots_in<TP, T, CX, (S::staged() ? 2 : 1)>(cx);
            MH_STAMP(4);
            asm volatile("" ::: "memory");
            asm volatile("" : "+v"(cx.qrow), "+v"(cx.qdrow), "+v"(cx.lq), "+v"(cx.lqd), "+v"(cx.st.lbase));
            aba_roots_out<TP, T, CX, 1>(cx);
         }
      }
      MH_STAMP(5);
      __syncthreads(); // results complete; the exchange area is free for the next batch slice
      MH_STAMP(6);
      if constexpr (ALGO == 1 && IO_LDS && IDENT)
      {
         if (A.q_next)
         { // fused simulation step: q, qd and the fresh qdd rows of the 64 configurations all sit in LDS -- integrate them in place
           // (MultiBodySystemStateIntegrator.java:365-441, 503-575, 710-733) and stream the new state out with the accelerations
            if (lane < rows)
               integrate_rows<TP, 0, T>(wave, lq + lane * nq, lqd + lane * nv, lres + lane * nv, A.dt, T(0.5) * A.dt * A.dt);
            __syncthreads();
            wave_copy_out<T, 256>(A.q_next + cfg0 * nq, lq, rows * nq);
            wave_copy_out<T, 256>(A.qd_next + cfg0 * nv, lqd, rows * nv);
         }
      }
      if constexpr (IO_LDS)
      {
         wave_copy_out<T, 256>(A.out + cfg0 * nv, lres, rows * nv);
         __syncthreads();
      }
      MH_STAMP(7);
   }
}

// Fused RNEA + ABA, tree-split: workgroups [0, G) compute tau = RNEA(q, qd, qdd), workgroups [G, 2G) qdd = ABA(q, qd, tau_in).
template <class TP, typename T, bool IDENT, bool IO_LDS>
__global__ void __launch_bounds__(256) spec_fused_split_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   const long half = gridDim.x / 2;
   if ((long)blockIdx.x < half)
      split_group<TP, T, 0, IDENT, IO_LDS>(A, blockIdx.x, half, (lds_ptr<T>)lds_raw);
   else
   {
      Args<T> A2 = A;
      A2.in3 = A.in3b, A2.out = A.outb;
      split_group<TP, T, 1, IDENT, IO_LDS>(A2, blockIdx.x - half, half, (lds_ptr<T>)lds_raw);
   }
}
template <class TP, typename T, int ALGO, bool IDENT, bool IO_LDS, bool BODIES = false>
__global__ void __launch_bounds__(256) spec_split_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   split_group<TP, T, ALGO, IDENT, IO_LDS, BODIES>(A, blockIdx.x, gridDim.x, (lds_ptr<T>)lds_raw);
}

// The same with a register budget for three waves per SIMD (168 VGPRs): the tree-split RNEA without LDS rows (SoA matrices, read
// directly and coalesced) needs 31 KB of LDS per workgroup, so five workgroups would fit a CU -- the 190 registers of the plain build hold
// it at two.
template <class TP, typename T, int ALGO, bool IDENT, bool IO_LDS>
__global__ void __launch_bounds__(256, 3) spec_split_kernel_occ3(Args<T> A)
{
   extern __shared__ double lds_raw[];
   split_group<TP, T, ALGO, IDENT, IO_LDS, false>(A, blockIdx.x, gridDim.x, (lds_ptr<T>)lds_raw);
}

// CRBA, direct stores: H [B][nv][nv] (or [nv*nv][B]) must be zero-filled by the caller; only entries of related joints are
// written.  Used when the index maps are not the identity.
template <class TP, typename T, bool IDENT>
__global__ void __launch_bounds__(64) spec_crba_kernel(Args<T> A)
{
   using CX = Ctx<T, false, IDENT, WholeStore<TP, ST_GLOBAL_KIND>>;
   const long lane = (long)blockIdx.x * 64 + threadIdx.x, nlanes = (long)gridDim.x * 64;
   warm_scalar_cache(A.m.consts, A.m.n * MC_STRIDE * (int)sizeof(T));
   for (long cfg = lane; cfg < A.B; cfg += nlanes)
   {
      CX cx;
      fill_ctx<T>(cx, A, cfg);
      cx.nv = A.m.nv;
      cx.wave = 0;
      crba_roots<TP, T, CX, false>(cx);
   }
}
// CRBA, packed: identity index maps.  The structurally non-zero lower-triangle entries are collected in a per-lane LDS image
// (slot-major, 64 lanes per slot); afterwards every lane writes its own H -- all nv x nv entries, zeros included, ascending
// addresses, 16 bytes per store -- so no memset pass is needed and every cache line is written exactly once.
template <class TP, typename T>
__global__ void __launch_bounds__(64) spec_crba_packed_kernel(Args<T> A)
{
   extern __shared__ double lds_raw[];
   using CX = Ctx<T, false, true, std::conditional_t<MH_RNEA_PRE != 0, RneaPreStore<TP>, WholeStore<TP, ST_GLOBAL_KIND>>>;
   using HM = HMap<TP>;
   constexpr int NV = HM::NV;
   const lds_ptr<T> img = (lds_ptr<T>)lds_raw + threadIdx.x;
   warm_scalar_cache(A.m.consts, A.m.n * MC_STRIDE * (int)sizeof(T));
   const long wave = blockIdx.x, nwaves = gridDim.x;
   for (long cfg0 = wave * 64; cfg0 < A.B; cfg0 += nwaves * 64)
   {
      const long cfg = cfg0 + threadIdx.x;
      if (cfg < A.B)
      {
         CX cx;
         fill_ctx<T>(cx, A, cfg);
         cx.nv = NV;
         cx.wave = 0;
         cx.xbase = img;
         rnea_pre_all<TP, T, CX>(cx); // the pairs of every revolute joint in one block (RneaPreStore; crba_joint takes them from there)
         crba_roots<TP, T, CX, true>(cx);
         asm volatile("" ::: "memory");
         T *H = A.out + cfg * A.v_bs;
         const long es = A.v_es;
         if (es == 1)
         { // AoS: this lane's matrix is contiguous
#pragma unroll
            for (int e = 0; e + 1 < NV * NV; e += 2)
            {
               const int s0 = HM::T.slot[e], s1 = HM::T.slot[e + 1];
               const T v0 = s0 >= 0 ? img[s0 * 64] : T(0), v1 = s1 >= 0 ? img[s1 * 64] : T(0);
               H[e] = v0;
               H[e + 1] = v1;
            }
            if constexpr ((NV * NV) % 2 == 1)
               H[NV * NV - 1] = HM::T.slot[NV * NV - 1] >= 0 ? img[HM::T.slot[NV * NV - 1] * 64] : T(0);
         }
         else
         { // SoA: entry e of all configurations is contiguous
#pragma unroll
            for (int e = 0; e < NV * NV; e++)
            {
               const int s0 = HM::T.slot[e];
               H[e * es] = s0 >= 0 ? img[s0 * 64] : T(0);
            }
         }
      }
   }
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release fence, which on gfx950 waits for
// EVERY outstanding vector-memory operation of the wave (s_waitcnt vmcnt(0)): stores streamed out just before it would be waited for,
// acknowledgement by acknowledgement.  Safe where no wave of the workgroup reads global memory another wave of it wrote.
MH_DEV void lds_barrier()
{
   asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// Coalesced write-out of the tree-split CRBA: the rows of H that belong to the trunk's dofs (TRUNK) or to the limbs' dofs, for the `rows`
// configurations of a workgroup, from the packed lane-major LDS image (row pitch nsp), zeros included, by threads t of nt.  A run of
// consecutive rows of one kind is contiguous in the AoS result; element g of a run's slice is entry e0 + g % ne of configuration g / ne.
template <class TP>
struct HRows
{
   using TR = Tree<TP>;
   static constexpr int NV = TR::total_dofs();
   static constexpr bool trunk_row(int r)
   {
      for (int j = 0; j < TP::N; j++)
         if (r >= TR::dof_ofs(j) && r < TR::dof_ofs(j) + TR::ndof(j))
            return Split<TP>::is_trunk(j);
      return true;
   }
   static constexpr int n_runs(bool trunk)
   {
      int n = 0;
      for (int r = 0; r < NV; r++)
         if (trunk_row(r) == trunk && (r == 0 || trunk_row(r - 1) != trunk))
            n++;
      return n;
   }
   static constexpr int run_start(bool trunk, int k)
   {
      int n = 0;
      for (int r = 0; r < NV; r++)
         if (trunk_row(r) == trunk && (r == 0 || trunk_row(r - 1) != trunk))
            if (n++ == k)
               return r;
      return NV;
   }
   static constexpr int run_len(bool trunk, int k)
   {
      int r = run_start(trunk, k), n = 0;
      while (r + n < NV && trunk_row(r + n) == trunk)
         n++;
      return n;
   }
};
// (Measured, profiles/r04_crba_writeout_exp.txt: the same loop with constant data and no LDS access at all takes 88 % of the time -- a CU
// issues these stores at 10-13 bytes per clock whatever feeds them, 256 CUs together at the 6 TB/s the memory takes; plain instead of
// nontemporal stores: no faster at 4 096, 6 % slower at 262 144.)
template <typename T, int E0, int NEK, int NE>
MH_DEV void crba_write_run(T *H, lds_ptr<T> img, const short __attribute__((address_space(3))) *tab, int nsp, int rows, int t, int nt)
{
   if constexpr (NEK % 2 == 0 && E0 % 2 == 0 && NE % 2 == 0 && sizeof(T) == 8)
   { // two entries = 16 bytes per lane and store (a pair never straddles two matrices): half the instructions of the loop and twice the
     // bytes each store instruction keeps in flight -- the write-out is bound by the latter
      if ((((unsigned long long)H) & 15) == 0)
      {
         typedef double __attribute__((ext_vector_type(2))) d2;
         constexpr int NP = NEK / 2, UN2 = 4;
         const int pairs = rows * NP;
         for (int gp = t; gp < pairs; gp += UN2 * nt)
         {
            int s0[UN2], s1[UN2], cc[UN2], pp[UN2];
#pragma unroll
            for (int u = 0; u < UN2; u++)
            {
               const int g = gp + u * nt;
               const bool in = g < pairs;
               const int c = in ? g / NP : 0, p2 = in ? g - c * NP : 0;
               s0[u] = in ? (int)tab[E0 + 2 * p2] : -1;
               s1[u] = in ? (int)tab[E0 + 2 * p2 + 1] : -1;
               cc[u] = c, pp[u] = p2;
            }
            d2 v[UN2];
#pragma unroll
            for (int u = 0; u < UN2; u++)
            {
               v[u].x = s0[u] >= 0 ? img[cc[u] * nsp + s0[u]] : 0.0;
               v[u].y = s1[u] >= 0 ? img[cc[u] * nsp + s1[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < UN2; u++)
               if (gp + u * nt < pairs)
                  __builtin_nontemporal_store(v[u], (d2 *)(H + (long)cc[u] * NE + E0) + pp[u]); // streamed once, never read back by this kernel
         }
         return;
      }
   }
   constexpr int UN = 8;
   const int total = rows * NEK;
   for (int g0 = t; g0 < total; g0 += UN * nt)
   { // UN independent look-ups, reads and stores in flight per thread
      int sl[UN], cc[UN], ee[UN];
#pragma unroll
      for (int u = 0; u < UN; u++)
      {
         const int g = g0 + u * nt;
         const bool in = g < total;
         const int c = in ? g / NEK : 0, e = in ? g - c * NEK : 0;
         sl[u] = in ? (int)tab[E0 + e] : -1;
         cc[u] = c, ee[u] = e;
      }
      T v[UN];
#pragma unroll
      for (int u = 0; u < UN; u++)
         v[u] = sl[u] >= 0 ? img[cc[u] * nsp + sl[u]] : T(0);
#pragma unroll
      for (int u = 0; u < UN; u++)
         if (g0 + u * nt < total)
            __builtin_nontemporal_store(v[u], H + (long)cc[u] * NE + E0 + ee[u]);
   }
}
template <class TP, typename T, bool TRUNK, int K = 0>
MH_DEV void crba_write_rows(T *H, lds_ptr<T> img, const short __attribute__((address_space(3))) *tab, int nsp, int rows, int t, int nt)
{
   using HR = HRows<TP>;
   if constexpr (K < HR::n_runs(TRUNK))
   {
      constexpr int NV = HR::NV, R0 = HR::run_start(TRUNK, K), RN = HR::run_len(TRUNK, K);
      crba_write_run<T, R0 * NV, RN * NV, NV * NV>(H, img, tab, nsp, rows, t, nt);
      crba_write_rows<TP, T, TRUNK, K + 1>(H, img, tab, nsp, rows, t, nt);
   }
}

// Tree-split CRBA (identity index maps, AoS): four waves share 64 configurations.  Each limb's owner walks the limb with the joint
// transforms of its trunk ancestors in hand, which completes every column of H that belongs to a limb body and leaves the limb's
// composite inertia in the exchange area; after one barrier wave 0 finishes the trunk bodies' columns; after a second barrier the
// four waves write a quarter of every matrix each, straight from the packed LDS image (zeros included, no memset).
template <class TP, typename T>
MH_DEV void crba_split_group(const Args<T> &A, int lpg, const long block, const long nblocks, double __attribute__((address_space(3))) *lds_raw)
{ // lpg = configurations per workgroup (<= 64).  The write-out of 7.2 KB per configuration is bound by what ONE CU can have in
  // flight, so a small batch is spread over more, thinner workgroups (16 lanes of each wave active) to put every CU's store path to work.
  // block / nblocks: this workgroup's position among the workgroups that do this job (a fused launch gives the rest another job).
   using CX = Ctx<T, false, true, std::conditional_t<MH_RNEA_PRE != 0, RneaPreStore<TP>, WholeStore<TP, ST_GLOBAL_KIND>>>;
   using HM = HMap<TP>;
   constexpr int NV = HM::NV, NE = NV * NV;
   constexpr int NSP = HM::T.n_slots | 1; // odd row pitch of the lane-major image: lanes writing one slot spread over all banks
   const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const int lane = threadIdx.x & 63;
   const lds_ptr<T> img = (lds_ptr<T>)lds_raw;
   // entry -> slot table of the write-out, staged once (the lookups sit in a dependent chain: LDS latency, not global)
   // LDS: image of lpg rows | limb exchange, lpg rows | table -- a thin workgroup asks for thin areas (mh_spec.hip: crba_split_lds),
   // which is what lets two workgroups share a CU
   constexpr int XP = (Split<TP>::n_limbs() * 10) | 1; // pitch of a lane's exchange records (odd, like the image's)
   const int img_words = __builtin_amdgcn_readfirstlane(lpg) * NSP;
   short __attribute__((address_space(3))) *tab = (short __attribute__((address_space(3))) *)(img + img_words + __builtin_amdgcn_readfirstlane(lpg) * XP);
   for (int e = threadIdx.x; e < NE; e += 256)
      tab[e] = (short)HM::slot_at(e);
   warm_scalar_cache(A.m.consts, A.m.n * MC_STRIDE * (int)sizeof(T));
   for (long cfg0 = block * lpg; cfg0 < A.B; cfg0 += nblocks * lpg)
   {
      const long cfg = cfg0 + lane;
      const bool active = lane < lpg && cfg < A.B;
#ifdef MH_PROBE
#define MH_CSTAMP(k)                                                                                                                       \
   do                                                                                                                                      \
   {                                                                                                                                       \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                                          \
      if (lane == 0)                                                                                                                       \
         ((unsigned long long *)(A.out + A.B * NE))[(cfg0 / lpg) * 32 + wave * 8 + (k)] = t_;                                               \
   } while (0)
#else
#define MH_CSTAMP(k)
#endif
      MH_CSTAMP(0);
      CX cx;
      fill_ctx<T>(cx, A, active ? cfg : cfg0);
      cx.nv = NV;
      cx.wave = wave;
      cx.xbase = img + lane * NSP;
      cx.lx = img + img_words + (active ? lane : 0) * XP;
      MH_CSTAMP(1);
      if (active)
         split_crba_limbs<TP, 0, T, CX>(cx);
      MH_CSTAMP(2);
      __syncthreads();
      MH_CSTAMP(3);
      // Every row of H that belongs to a limb dof is complete now (the limbs' owners wrote their columns up to the root; by symmetry the
      // rows).  Thin workgroups (small batches: one slice per workgroup, the write-out at the end of everything): waves 1-3 stream those
      // rows out while wave 0 finishes the trunk bodies' columns, whose entries all lie in rows of trunk dofs; the barrier behind does not
      // wait for the stores (at B = 4096, 16 configurations per workgroup: limbs 6.9 us | trunk 2.6 | write-out 4.4 before,
      // profiles/r04_crba_stamps.txt).  Full groups of 64 keep the one write-out by all four waves: splitting it cost more than the
      // trunk pass it hides (262 144 configurations: 442 -> 525 us).
      // (Each wave streaming its OWN limbs' rows as soon as it is through with them, counted in through an LDS word instead of the barrier,
      // was measured too: 17.3 against 16.9 us -- one wave alone stores at 2.5-3.5 bytes per clock, a quarter of what the CU's four waves
      // reach together: profiles/r04_crba_own_rows_experiment.txt.)
      const int rows = (int)(A.B - cfg0 < lpg ? A.B - cfg0 : lpg);
      const bool early = __builtin_amdgcn_readfirstlane(lpg) < 64;
      if (wave == 0)
      {
         if (active)
         {
            crba_pre_pass<TP, -1, T, CX>(cx); // the trunk's pairs, in one block (the limb phase's were another wave's, or are dead by now)
            crba_roots<TP, T, CX, 2, 1>(cx);
         }
      }
      else if (early)
         crba_write_rows<TP, T, false>(A.out + cfg0 * NE, img, tab, NSP, rows, (int)threadIdx.x - 64, 192);
      MH_CSTAMP(4);
      lds_barrier();
      MH_CSTAMP(5);
      if (early) // what is left: the rows of the trunk's dofs
         crba_write_rows<TP, T, true>(A.out + cfg0 * NE, img, tab, NSP, rows, threadIdx.x, 256);
      else
         crba_write_run<T, 0, NE, NE>(A.out + cfg0 * NE, img, tab, NSP, rows, threadIdx.x, 256);
      MH_CSTAMP(6);
      lds_barrier(); // the image is free for the next slice (the stores took their values from it into registers: nothing waits for their acknowledgements)
      MH_CSTAMP(7);
   }
}

template <class TP, typename T>
__global__ void __launch_bounds__(256) spec_crba_split_kernel(Args<T> A, int lpg)
{
   extern __shared__ double lds_raw[];
   crba_split_group<TP, T>(A, lpg, blockIdx.x, gridDim.x, (double __attribute__((address_space(3))) *)lds_raw);
}
// RNEA and CRBA of the same configurations side by side in ONE launch (what a whole-body controller evaluates per tick:
// InverseDynamicsCalculator + CompositeRigidBodyMassMatrixCalculator on one state): the first rnea_groups workgroups are tree-split RNEA
// groups of 64 configurations (rows staged in LDS, identity maps), the others tree-split CRBA groups of lpg; A.outb = H.
template <class TP, typename T>
__global__ void __launch_bounds__(256) spec_rnea_crba_split_kernel(Args<T> A, int lpg, int rnea_groups)
{
   extern __shared__ double lds_raw[];
   if ((int)blockIdx.x < rnea_groups)
      split_group<TP, T, 0, true, true>(A, blockIdx.x, rnea_groups, (lds_ptr<T>)lds_raw);
   else
   {
      Args<T> A2 = A;
      A2.out = A.outb;
      crba_split_group<TP, T>(A2, lpg, (long)blockIdx.x - rnea_groups, (long)gridDim.x - rnea_groups, (double __attribute__((address_space(3))) *)lds_raw);
   }
}

// Mass + Coriolis matrix, one wave per 64 configurations, direct stores: A.out = H, A.outb = C, both [B][nv][nv] with per-configuration /
// per-entry strides f_bs / f_es and zero-filled by the caller (only entries of related joints are written).
template <class TP, typename T, bool IDENT, bool AOS>
__global__ void __launch_bounds__(64) spec_coriolis_kernel(Args<T> A)
{
   using CX = Ctx<T, false, IDENT, WholeStore<TP, ST_GLOBAL_KIND>>;
   const long lane = (long)blockIdx.x * 64 + threadIdx.x, nlanes = (long)gridDim.x * 64;
   warm_scalar_cache(A.m.consts, A.m.n * MC_STRIDE * (int)sizeof(T));
   // gridDim.y waves share a group of 64 configurations (small batches): each runs the whole recursion for the composite inertias and
   // writes the columns of every gridDim.y-th body
   unsigned long long own = 0;
   for (int j = blockIdx.y; j < TP::N; j += gridDim.y)
      own |= 1ull << (j & 63);
   for (long cfg = lane; cfg < A.B; cfg += nlanes)
   {
      CX cx;
      fill_ctx<T>(cx, A, cfg);
      cx.own = own;
      cx.frow = nullptr;
      cx.orow = A.out + cfg * A.f_bs;
      cx.orow2 = A.outb + cfg * A.f_bs;
      int nv = A.m.nv;
      long es = A.f_es;
      asm volatile("" : "+s"(nv), "+s"(es)); // per configuration: keeps the entry offsets from being hoisted out of the loop
      cx.nv = nv, cx.f_es = es;
      cx.wave = 0;
      coriolis_roots<TP, T, CX, IDENT && AOS>(cx);
   }
}

// Centroidal momentum matrix / convective term / centre of mass, one wave per 64 configurations, direct stores (A zero-filled by the
// caller: columns no considered joint owns stay zero).
template <class TP, typename T, bool IDENT, bool WITH_B>
__global__ void __launch_bounds__(64) spec_centroidal_kernel(CentArgs<T> A)
{
   using BASE = Ctx<T, false, IDENT, WholeStore<TP, ST_GLOBAL_KIND>>;
   using CX = CentCtx<T, BASE>;
   const long lane = (long)blockIdx.x * 64 + threadIdx.x, nlanes = (long)gridDim.x * 64;
   warm_scalar_cache(A.m.consts, A.m.n * MC_STRIDE * (int)sizeof(T));
   const V3<T> Z{T(0), T(0), T(0)};
   unsigned long long own = 0; // (as in spec_coriolis_kernel)
   for (int j = blockIdx.y; j < TP::N; j += gridDim.y)
      own |= 1ull << (j & 63);
   for (long cfg = lane; cfg < A.B; cfg += nlanes)
   {
      CX cx;
      cx.own = own;
      const void *pc = A.m.consts;
      const int *pd = A.m.dof_map, *pq = A.m.cfg_map, *pm = A.m.meta;
      int nv = A.m.nv;
      long es = A.a_es;
      asm volatile("" : "+s"(pc), "+s"(pd), "+s"(pq), "+s"(pm), "+s"(nv), "+s"(es));
      cx.C = (const T *)pc;
      cx.dof_map = as_const(pd), cx.cfg_map = as_const(pq), cx.meta = as_const(pm);
      cx.qrow = A.q + cfg * A.q_bs;
      cx.qdrow = WITH_B ? A.qd + cfg * A.v_bs : nullptr;
      cx.in3row = nullptr, cx.frow = nullptr, cx.orow = nullptr, cx.orow2 = nullptr;
      cx.q_es = A.q_es, cx.v_es = A.v_es, cx.f_es = 0;
      cx.a0a = Z, cx.a0l = Z;
      cx.coriolis = 1, cx.accel = 0;
      cx.nv = nv, cx.wave = 0;
      cx.xf.R = M3<T>{A.fR[0], A.fR[1], A.fR[2], A.fR[3], A.fR[4], A.fR[5], A.fR[6], A.fR[7], A.fR[8]};
      cx.xf.p = V3<T>{A.fp[0], A.fp[1], A.fp[2]};
      cx.arow = A.A + cfg * A.a_bs;
      cx.a_es = es;
      CentUp<T> total;
      total.I = RI<T>{T(0), Z, S3<T>{T(0), T(0), T(0), T(0), T(0), T(0)}};
      total.f = SV<T>{Z, Z};
      centroidal_roots<TP, T, CX, WITH_B>(cx, total);
      // the frame's origin: centre of mass of the considered bodies, in frame coordinates (CenterOfMassCalculator.java:70-91)
      V3<T> shift{T(0), T(0), T(0)};
      if (A.at_com)
      {
         shift = tmul(cx.xf.R, (T(1) / total.I.m) * total.I.h - cx.xf.p);
         T *Am = cx.arow;
         for (int j = blockIdx.y; j < TP::N; j += gridDim.y) // (the columns this wave wrote)
         {
            const int *mj = pm + j * MI_STRIDE;
            const int *dj = pd + mj[MI_DOF];
            for (int k = 0; k < dof_count(mj[MI_TYPE]); k++)
            { // moving the origin by `shift`: n' = n - shift x f
               const long col = dj[k];
               const V3<T> fl{Am[(3L * nv + col) * es], Am[(4L * nv + col) * es], Am[(5L * nv + col) * es]};
               const V3<T> d = cross(shift, fl);
               Am[(0L * nv + col) * es] -= d.x, Am[(1L * nv + col) * es] -= d.y, Am[(2L * nv + col) * es] -= d.z;
            }
         }
      }
      if (blockIdx.y != 0)
         continue; // the centre of mass and the convective term are written once
      if (A.com)
      {
         T *crow = A.com + cfg * A.c_bs;
         crow[0] = shift.x, crow[A.c_es] = shift.y, crow[2 * A.c_es] = shift.z;
      }
      if constexpr (WITH_B)
      {
         const V3<T> fl = tmul(cx.xf.R, total.f.l);
         const V3<T> fa = tmul(cx.xf.R, total.f.a - cross(cx.xf.p, total.f.l)) - cross(shift, fl);
         T *brow = A.b + cfg * A.b_bs;
         brow[0] = fa.x, brow[A.b_es] = fa.y, brow[2 * A.b_es] = fa.z;
         brow[3 * A.b_es] = fl.x, brow[4 * A.b_es] = fl.y, brow[5 * A.b_es] = fl.z;
      }
   }
}

} // namespace mh
