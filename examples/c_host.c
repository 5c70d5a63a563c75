/* A C host on the drop-in boundary: builds the flat description of a two-joint arm (revolute about z, prismatic along x), evaluates RNEA
 * and ABA for a few configurations through the host-pointer entry points, and checks the round trip.
 *
 *   gcc -std=c99 -I include examples/c_host.c -o c_host -L mecano_amd -lmecano_hip -Wl,-rpath,$PWD/mecano_amd && ./c_host
 *
 * (needs libmecano_hip.so from `python -m mecano_amd.build` and an MI355X; without a device it reports MH_ERR_NO_DEVICE and exits 0). */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "mecano_hip.h"

int main(void)
{
   int32_t parent[2] = {-1, 0}, type[2] = {MH_JOINT_REVOLUTE, MH_JOINT_PRISMATIC}, dof[2] = {0, 1}, cfg[2] = {0, 1};
   double axis[6] = {0, 0, 1, 1, 0, 0};
   /* frameBeforeJoint.getTransformToParent(): joint 0 at the root, joint 1 half a metre along x of body 0 */
   double X_before[24] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0.5, 0, 0};
   /* bodyFixedFrame.getTransformToParent(): centres of mass 0.25 m / 0.1 m along x */
   double X_com[24] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0.25, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0.1, 0, 0};
   double J[18] = {0.02, 0, 0, 0, 0.05, 0, 0, 0, 0.05, 0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.01}, mass[2] = {2.0, 1.0}, com[6] = {0};
   mh_model_desc d;
   memset(&d, 0, sizeof d);
   d.n_joints = 2, d.nq = 2, d.nv = 2;
   d.parent = parent, d.joint_type = type, d.axis = axis, d.X_before = X_before, d.X_com = X_com;
   d.inertia_J = J, d.inertia_mass = mass, d.inertia_com = com, d.dof_indices = dof, d.cfg_indices = cfg;

   mh_model_t model = NULL;
   mh_status st = mh_model_create(&d, &model);
   if (st == MH_ERR_NO_DEVICE)
   {
      printf("no HIP device: %s\n", mh_last_error());
      return 0;
   }
   if (st != MH_OK)
   {
      printf("mh_model_create failed: %s\n", mh_last_error());
      return 1;
   }
   enum { B = 3 };
   double q[B][2] = {{0.0, 0.0}, {0.7, 0.2}, {-1.3, -0.1}}, qd[B][2] = {{0, 0}, {1.0, -0.5}, {0.3, 0.8}};
   double qdd[B][2] = {{0, 0}, {0.4, 0.1}, {-2.0, 1.5}}, tau[B][2], back[B][2];
   const double g[3] = {0.0, 0.0, -9.81};
   if (mh_rnea_f64_host(model, B, &q[0][0], &qd[0][0], &qdd[0][0], g, NULL, NULL, &tau[0][0]) != MH_OK
       || mh_aba_f64_host(model, B, &q[0][0], &qd[0][0], &tau[0][0], g, NULL, NULL, &back[0][0]) != MH_OK)
   {
      printf("compute failed: %s\n", mh_last_error());
      return 1;
   }
   double err = 0;
   for (int b = 0; b < B; b++)
   {
      printf("config %d: tau = (% .6f, % .6f)   ABA(RNEA(qdd)) - qdd = (% .1e, % .1e)\n", b, tau[b][0], tau[b][1], back[b][0] - qdd[b][0],
             back[b][1] - qdd[b][1]);
      err = fmax(err, fmax(fabs(back[b][0] - qdd[b][0]), fabs(back[b][1] - qdd[b][1])));
   }
   /* a simulation loop that never leaves the device (what a Java host does with mh_device_alloc: no HIP binding of its own needed):
    * state uploaded once, 200 steps of forward dynamics + integration, state downloaded once */
   {
      double *d_q, *d_qd, *d_tau, *d_qdd, zero[B][2] = {{0}}, q1[B][2], qd1[B][2];
      if (mh_device_alloc(sizeof q, (void **)&d_q) || mh_device_alloc(sizeof q, (void **)&d_qd) || mh_device_alloc(sizeof q, (void **)&d_tau)
          || mh_device_alloc(sizeof q, (void **)&d_qdd) || mh_copy_to_device(d_q, q, sizeof q, NULL) || mh_copy_to_device(d_qd, qd, sizeof q, NULL)
          || mh_copy_to_device(d_tau, zero, sizeof q, NULL))
      {
         printf("device memory helpers failed: %s\n", mh_last_error());
         return 1;
      }
      for (int step = 0; step < 200; step++)
         if (mh_aba_integrate_f64(model, B, 1.0e-3, d_q, d_qd, d_tau, g, NULL, NULL, d_qdd, d_q, d_qd) != MH_OK)
         {
            printf("step failed: %s\n", mh_last_error());
            return 1;
         }
      if (mh_copy_to_host(q1, d_q, sizeof q, NULL) || mh_copy_to_host(qd1, d_qd, sizeof q, NULL) || mh_stream_synchronize(NULL))
         return 1;
      /* configuration 0 starts at rest with gravity along the revolute axis and across the slide: it must stay at rest */
      printf("after 0.2 s on the device: q[0] = (% .3e, % .3e), q[1] = (% .6f, % .6f)\n", q1[0][0], q1[0][1], q1[1][0], q1[1][1]);
      if (fabs(q1[0][0]) > 1e-12 || fabs(q1[0][1]) > 1e-12 || !(q1[1][0] == q1[1][0]))
         return 1;
      mh_device_free(d_q), mh_device_free(d_qd), mh_device_free(d_tau), mh_device_free(d_qdd);
   }
   /* gravity acts along the revolute axis and across the slide: at rest no effort is needed */
   printf("kernel variant: %s, round-trip error %.1e\n", mh_model_kernel_variant(model), err);
   mh_model_destroy(model);
   return err < 1e-10 && fabs(tau[0][0]) < 1e-12 && fabs(tau[0][1]) < 1e-12 ? 0 : 1;
}
