"""Drop-in use of the calculators, batched: 4096 independent 30-DoF humanoids falling under gravity, stepped on the device.

    python examples/simulate_humanoid.py [steps]

Each step is ForwardDynamicsCalculator.compute() followed by MultiBodySystemStateIntegrator.doubleIntegrateFromAcceleration() of the
reference (ForwardDynamicsCalculator.java:475-520, tools/MultiBodySystemStateIntegrator.java:365-441) -- here one kernel launch for the
whole batch (mh_aba_integrate_f64); the state never leaves the GPU.  Needs a built library (python -m mecano_amd.build) and an MI355X."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), ".."))
from mecano_amd import random_tools as rt
from mecano_amd.calculators import (CompositeRigidBodyMassMatrixCalculator, ForwardDynamicsCalculator, InverseDynamicsCalculator,
                                    MultiBodySystemStateIntegrator)

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B, dt = 4096, 1.0e-3
system = rt.nextHumanoid(np.random.default_rng(43))  # SixDoFJoint pelvis + 24 RevoluteJoints, built like a Mecano MultiBodySystem
q, qd, _, _ = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(0), system, B))
tau = torch.zeros_like(qd)  # limp robots

forwardDynamics = ForwardDynamicsCalculator(system)
forwardDynamics.setGravitationalAcceleration(-9.81)
integrator = MultiBodySystemStateIntegrator(dt)

# the same loop written as the reference's tests do (two calls per step) ...
for _ in range(3):
    qdd = forwardDynamics.compute(q, qd, tau)
    q, qd = integrator.doubleIntegrateFromAcceleration(forwardDynamics, q, qd, qdd)
# ... and as one launch per step
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    q, qd, _ = forwardDynamics.model.step(dt, q, qd, tau, (0.0, 0.0, -9.81), inplace=True)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"{steps} steps of {B} humanoids: {el * 1e3:.1f} ms  ({steps * B / el / 1e6:.1f} M robot-steps/s, {el / steps * 1e6:.1f} us per step incl. Python)")

# inverse dynamics of the motion just simulated returns the (zero) efforts that produced it; H and C for a controller
inverseDynamics = InverseDynamicsCalculator(system)
inverseDynamics.setGravitationalAcceleration(-9.81)
qdd = forwardDynamics.compute(q, qd, tau)
print("max |RNEA(ABA(tau)) - tau| =", float(inverseDynamics.compute(q, qd, qdd).abs().max()))
massMatrix = CompositeRigidBodyMassMatrixCalculator(system)
massMatrix.setEnableCoriolisMatrixCalculation(True)
massMatrix.setCentroidalMomentumFrame(None, atCenterOfMass=True)
H = massMatrix.compute(q, qd)
C = massMatrix.getCoriolisMatrix()
h = torch.einsum("bij,bj->bi", massMatrix.getCentroidalMomentumMatrix(), qd)
print("H", tuple(H.shape), "C", tuple(C.shape), "| centroidal momentum of robot 0:", h[0].cpu().numpy().round(3))
