"""mecano_amd -- MI355X-native batched RNEA / ABA / CRBA behind Mecano's calculator API.

Importing this package does not load the HIP library; the first model creation does, and fails loudly
(ImportError) when mecano_amd/libmecano_hip.so has not been built.  There is no CPU fallback.
"""
from .multibody import (FixedJoint, JointMatrixIndexProvider, ModelDesc, MultiBodySystem, PlanarJoint, PrismaticJoint, RevoluteJoint,
                        RigidBody, SixDoFJoint, SphericalJoint)

__all__ = ["FixedJoint", "JointMatrixIndexProvider", "ModelDesc", "MultiBodySystem", "PlanarJoint", "PrismaticJoint", "RevoluteJoint",
           "RigidBody", "SixDoFJoint", "SphericalJoint"]
