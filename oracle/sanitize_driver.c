/*
 * sanitize_driver.c -- runs every entry point of the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer.
 * TEST INFRASTRUCTURE (see mecano_oracle.c): built and run by tests/test_oracle_sanitizers.py on the CPU
 * (`make -C oracle sanitize`; GPU sanitizers are not available on this pool).  Exit code 0 = no finding.
 */
#include "mecano_oracle.c"

#include <stdio.h>

static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static double urand(double lo, double hi)
{
   lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
   return lo + (hi - lo) * (double)(lcg_state >> 11) / 9007199254740992.0;
}
static void rand_rotation(double R[9])
{
   double q[4], n = 0;
   for (int k = 0; k < 4; k++)
      q[k] = urand(-1, 1), n += q[k] * q[k];
   n = sqrt(n);
   for (int k = 0; k < 4; k++)
      q[k] /= n;
   quat_to_R(q, R);
}

int main(void)
{
   enum { N = 24, B = 5 };
   int parent[N], type[N], dofi[6 * N], cfgi[7 * N], locked[N];
   double axis[3 * N], Xb[12 * N], Xc[12 * N], J[9 * N], mass[N], com[3 * N];
   int nq = 0, nv = 0;
   for (int i = 0; i < N; i++)
   {
      parent[i] = i == 0 ? -1 : (int)urand(0, i - 1e-9); /* random tree, parents first */
      type[i] = i % 6;                                   /* every joint kind */
      double a[3] = {urand(-1, 1), urand(-1, 1), urand(-1, 1)}, na = sqrt(v3_dot(a, a));
      for (int k = 0; k < 3; k++)
         axis[3 * i + k] = a[k] / na;
      rand_rotation(Xb + 12 * i), rand_rotation(Xc + 12 * i);
      for (int k = 0; k < 3; k++)
         Xb[12 * i + 9 + k] = urand(-1, 1), Xc[12 * i + 9 + k] = urand(-1, 1), com[3 * i + k] = i % 2 ? urand(-0.3, 0.3) : 0.0;
      double L[9] = {urand(0.5, 2), 0, 0, urand(-0.5, 0.5), urand(0.5, 2), 0, urand(-0.5, 0.5), urand(-0.5, 0.5), urand(0.5, 2)}, Lt[9];
      m3_transpose(L, Lt);
      m3_mul(L, Lt, J + 9 * i);
      mass[i] = 0.1 + urand(0, 1);
      locked[i] = (i % 4 == 1);
      for (int k = 0; k < joint_ndof(type[i]); k++)
         dofi[nv] = nv, nv++;
      for (int k = 0; k < joint_ncfg(type[i]); k++)
         cfgi[nq] = nq, nq++;
   }
   void *m = mo_model_create(N, nq, nv, parent, type, axis, Xb, Xc, J, mass, com, dofi, cfgi);
   if (!m)
      return 2;
   double *q = calloc((size_t)B * nq, sizeof(double)), *qd = calloc((size_t)B * nv, sizeof(double)), *qdd = calloc((size_t)B * nv, sizeof(double));
   double *tau = calloc((size_t)B * nv, sizeof(double)), *out = calloc((size_t)B * nv, sizeof(double)), *out2 = calloc((size_t)B * nv, sizeof(double));
   double *fext = calloc((size_t)B * N * 6, sizeof(double)), *H = calloc((size_t)B * nv * nv, sizeof(double)), *C = calloc((size_t)B * nv * nv, sizeof(double));
   double *A = calloc((size_t)B * 6 * nv, sizeof(double)), *bacc = calloc((size_t)B * N * 6, sizeof(double)), *btw = calloc((size_t)B * N * 6, sizeof(double));
   double *bw = calloc((size_t)B * N * 6, sizeof(double));
   double *qn = calloc((size_t)B * nq, sizeof(double)), *vn = calloc((size_t)B * nv, sizeof(double)), *an = calloc((size_t)B * nv, sizeof(double));
   double bb[6 * B], cm[3 * B], g[3] = {0.1, -0.2, -9.81}, frame[12];
   rand_rotation(frame);
   frame[9] = 0.3, frame[10] = -0.1, frame[11] = 0.2;
   for (long k = 0; k < (long)B * nq; k++)
      q[k] = urand(-1, 1);
   for (long k = 0; k < (long)B * nv; k++)
      qd[k] = urand(-1, 1), qdd[k] = urand(-1, 1), tau[k] = urand(-1, 1);
   for (long k = 0; k < (long)B * N * 6; k++)
      fext[k] = urand(-1, 1);
   int rc = 0;
   mo_rnea(m, B, q, qd, qdd, g, fext, 1, 1, out);
   mo_rnea(m, B, q, qd, NULL, g, NULL, 0, 0, out);
   rc |= mo_aba(m, B, q, qd, tau, g, fext, out);
   rc |= mo_aba_locked(m, B, q, qd, tau, qdd, g, NULL, locked, out, out2);
   mo_crba(m, B, q, H);
   mo_crba_coriolis(m, B, q, qd, H, C);
   mo_centroidal(m, B, q, qd, frame, 1, A, bb, cm);
   mo_centroidal(m, B, q, NULL, NULL, 0, A, NULL, NULL);
   mo_integrate(m, B, 1.0e-3, q, qd, qdd, qn, vn, an);
   mo_rnea_bodies(m, B, q, qd, qdd, g, fext, 1, 1, out, bacc, btw);
   rc |= mo_aba_bodies(m, B, q, qd, tau, g, fext, out, bacc, btw);
   mo_rnea_wrenches(m, B, q, qd, qdd, g, fext, 1, 1, out, bw);
   {
      int base[3] = {-1, 0, 5}, body[3] = {7, 11, 2};
      double rel[3 * 6 * B];
      mo_relative_acceleration(m, B, q, qd, qdd, g, 1, 1, 3, base, body, rel);
   }
   {
      double Jt[9], cc[3] = {0.1, 0.2, -0.3}, mm = 0.7, X[12], Aa[9], Ll[9], Cc[9], M[36], tw[6] = {1, 2, 3, 4, 5, 6}, w[6];
      memcpy(Jt, J, sizeof Jt);
      memcpy(X, Xb, sizeof X);
      mo_unit_abi_from_rigid(Jt, mm, cc, Aa, Ll, Cc);
      mo_unit_abi_apply_transform(X, 0, Aa, Ll, Cc);
      mo_unit_abi_apply_transform(X, 1, Aa, Ll, Cc);
      mo_unit_abi_to_dense(Aa, Ll, Cc, M);
      mo_unit_rigid_apply_transform(X, 0, Jt, &mm, cc);
      mo_unit_dynamic_wrench(Jt, mm, cc, tw, tw, 1, w);
      mo_unit_dynamic_wrench(Jt, mm, cc, NULL, tw, 0, w);
      mo_unit_rigid_mulv(Jt, mm, cc, tw, w);
      mo_unit_motion_transform(X, 1, tw, w);
      mo_unit_force_transform(X, tw, w);
      if (!(mo_unit_kinetic_coenergy(Jt, mm, cc, tw) == mo_unit_kinetic_coenergy(Jt, mm, cc, tw)))
         rc |= 4;
   }
   double s = 0;
   for (long k = 0; k < (long)B * nv; k++)
      s += out[k] + out2[k];
   printf("sanitize_driver: n = %d, nq = %d, nv = %d, checksum %.6e, rc %d\n", N, nq, nv, s, rc);
   mo_model_destroy(m);
   free(q), free(qd), free(qdd), free(tau), free(out), free(out2), free(fext), free(H), free(C), free(A), free(bacc), free(btw), free(bw);
   free(qn), free(vn), free(an);
   return (s == s) ? 0 : 3; /* rc != 0 only says a random joint-space block was not positive definite: not a sanitizer finding */
}
