"""Inertial-parameter identification with the batched joint torque regressor: what JointTorqueRegressorCalculator is for, B samples at once.

    python examples/identify_parameters.py [samples]

A 7-joint arm whose bodies have offset centres of mass is excited with random states; the torques "measured" by the inverse dynamics are
regressed on Y(q, qd, qdd) (JointTorqueRegressorCalculator.compute for all samples in one launch; firstMomentColumns=True makes tau linear
in (m, m c, J) -- the reference's centre-of-mass columns are zero, see include/mecano_hip.h).  The least-squares parameters are not
unique (only base combinations are identifiable) but they predict the torques of unseen states.  Needs a built library and an MI355X."""
import dataclasses
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), ".."))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
from mecano_amd.multibody import MultiBodySystem


def main(samples=4096, seed=7, verbose=True):
    rng = np.random.default_rng(seed)
    system = MultiBodySystem.toMultiBodySystemInput(rt.nextJointChain(rng, 7, ("revolute",))[0].getPredecessor())
    desc = system.toModelDesc()
    desc = dataclasses.replace(desc, inertia_com=rng.uniform(-0.2, 0.2, 3 * desc.n_joints))  # Mecano's generators leave the offsets at zero
    model = HipModel(desc)
    g = (0.0, 0.0, -9.81)

    def excite(n):
        q, qd, qdd, _ = (torch.tensor(x, device="cuda") for x in rt.nextState(rng, system, n))
        return q, qd, qdd

    q, qd, qdd = excite(samples)
    tau = model.rnea(q, qd, qdd, g)                                   # the "measurements"
    Y = model.regressor(q, qd, qdd, g, first_moment_columns=True)     # [samples, nv, 10 n]
    A, b = Y.reshape(samples * desc.nv, -1), tau.reshape(-1)
    pi_hat = torch.as_tensor(np.linalg.lstsq(A.cpu().numpy(), b.cpu().numpy(), rcond=None)[0], device="cuda")  # host solve: 28 672 x 70
    rank = int(np.linalg.matrix_rank(A.cpu().numpy()))
    fit = float((A @ pi_hat - b).abs().max())
    # unseen states
    q2, qd2, qdd2 = excite(samples)
    predicted = (model.regressor(q2, qd2, qdd2, g, first_moment_columns=True) @ pi_hat)
    err = float((predicted - model.rnea(q2, qd2, qdd2, g)).abs().max())
    scale = float(tau.abs().max())
    if verbose:
        print(f"{samples} samples, {desc.nv} joints, {10 * desc.n_joints} parameters, {rank} identifiable combinations")
        print(f"fit residual {fit:.2e}, prediction error on {samples} new states {err:.2e}  (max |tau| = {scale:.1f})")
    return fit, err, scale, rank


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
