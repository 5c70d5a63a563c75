package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import java.util.List;

import org.ejml.data.DMatrix;
import org.ejml.data.DMatrixRMaj;

import us.ihmc.euclid.tuple3D.interfaces.Tuple3DReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.JointBasics;
import us.ihmc.mecano.multiBodySystem.interfaces.JointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.MultiBodySystemReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.RigidBodyReadOnly;
import us.ihmc.mecano.spatial.interfaces.FixedFrameWrenchBasics;
import us.ihmc.mecano.spatial.interfaces.SpatialAccelerationReadOnly;
import us.ihmc.mecano.spatial.interfaces.WrenchReadOnly;

import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;
import static java.lang.foreign.ValueLayout.JAVA_INT;

/**
 * Batched drop-in for InverseDynamicsCalculator (RNEA): same constructor arguments, same setters, compute(...) over B stacked
 * configurations.  Matrices are DMatrixRMaj with one ROW per configuration: q is B x nq, qd / qdd / tau are B x nv, i.e. the column
 * vectors Mecano uses (MultiBodySystemTools.extractJointsState), transposed and stacked -- exactly the [B][n] layout of the C-ABI, so the
 * backing arrays are passed as they are.  NOT compiled in this repository's image (no JVM there).
 * <p>
 * Two ways to call it: {@link #compute(DMatrixRMaj, DMatrixRMaj, DMatrixRMaj)} with host matrices (mh_rnea_f64_host: chunked copies
 * overlapped with the kernels), or {@link #compute(HipDeviceBatch)} on state that already lives on the device, which also fills the
 * per-body accelerations / twists and the joint wrenches.
 */
public class HipInverseDynamicsCalculator implements AutoCloseable
{
   private final MultiBodySystemReadOnly input;
   private final HipMultiBodyModel model;
   private final double[] gravity = new double[3];
   private double[] rootAcceleration; // six components (angular, linear) once setRootAcceleration was called, else null (gravity rules)
   private HipSingleState single;     // the one-configuration face (compute() / compute(DMatrix)), created on first use
   private boolean singleResult;      // the last compute was a one-configuration one: results are column vectors like the reference's
   private boolean considerCoriolisAndCentrifugalForces = true, considerJointAccelerations = true;
   private DMatrixRMaj externalWrenches; // B x 6 n, (moment, force) per successor body in its body-fixed frame; null = none
   private final DMatrixRMaj jointTauMatrix = new DMatrixRMaj(0, 0);
   private final DMatrixRMaj bodyAccelerations = new DMatrixRMaj(0, 0), bodyTwists = new DMatrixRMaj(0, 0), jointWrenches = new DMatrixRMaj(0, 0);
   private HipDeviceBatch lastBatch;

   public HipInverseDynamicsCalculator(MultiBodySystemReadOnly input)
   {
      this(input, true);
   }

   /** InverseDynamicsCalculator(MultiBodySystemReadOnly, boolean considerIgnoredSubtreesInertia) (java:226-236). */
   public HipInverseDynamicsCalculator(MultiBodySystemReadOnly input, boolean considerIgnoredSubtreesInertia)
   {
      this.input = input;
      model = new HipMultiBodyModel(input, considerIgnoredSubtreesInertia);
   }

   /** InverseDynamicsCalculator.setConsiderCoriolisAndCentrifugalForces (java:291-296). */
   public void setConsiderCoriolisAndCentrifugalForces(boolean consider)
   {
      considerCoriolisAndCentrifugalForces = consider;
   }

   /** InverseDynamicsCalculator.setConsiderJointAccelerations (java:301-306). */
   public void setConsiderJointAccelerations(boolean consider)
   {
      considerJointAccelerations = consider;
   }

   /** InverseDynamicsCalculator.setGravitationalAcceleration(double): gravity along z, usually negative (java:388-403). */
   public void setGravitationalAcceleration(double gravity)
   {
      setGravitationalAcceleration(0.0, 0.0, gravity);
   }

   public void setGravitationalAcceleration(Tuple3DReadOnly gravity)
   {
      setGravitationalAcceleration(gravity.getX(), gravity.getY(), gravity.getZ());
   }

   public void setGravitationalAcceleration(double gravityX, double gravityY, double gravityZ)
   {
      gravity[0] = gravityX;
      gravity[1] = gravityY;
      gravity[2] = gravityZ;
      rootAcceleration = null; // the reference stores (0, -g) in the root acceleration: the last setter wins (java:343-348)
   }

   /**
    * InverseDynamicsCalculator.setRootAcceleration(SpatialAccelerationReadOnly) (java:413-427): the root's spatial acceleration, angular
    * and linear part, expressed in the root body's frame -- mh_options.root_acceleration.  It replaces what setGravitationalAcceleration
    * stored (both write the same field in the reference, java:343-348), and the other way round.
    */
   public void setRootAcceleration(SpatialAccelerationReadOnly newRootAcceleration)
   {
      newRootAcceleration.checkReferenceFrameMatch(input.getRootBody().getBodyFixedFrame(), input.getRootBody().getBodyFixedFrame().getRootFrame(),
                                                   input.getRootBody().getBodyFixedFrame()); // java:420: ReferenceFrameMismatchException
      rootAcceleration = new double[] {newRootAcceleration.getAngularPartX(), newRootAcceleration.getAngularPartY(), newRootAcceleration.getAngularPartZ(),
                                       newRootAcceleration.getLinearPartX(), newRootAcceleration.getLinearPartY(), newRootAcceleration.getLinearPartZ()};
   }

   /** Shorthand: the linear part alone (a translating base); gravity g is the root acceleration (0, -g) (java:343-348). */
   public void setRootAcceleration(Tuple3DReadOnly linearAcceleration)
   {
      rootAcceleration = new double[] {0.0, 0.0, 0.0, linearAcceleration.getX(), linearAcceleration.getY(), linearAcceleration.getZ()};
   }

   private HipSingleState single()
   {
      if (single == null)
         single = new HipSingleState(input, model);
      return single;
   }

   /** getExternalWrench(rigidBody) (java:444-461): the live external wrench of that body for the one-configuration calls; modify in place. */
   public FixedFrameWrenchBasics getExternalWrench(RigidBodyReadOnly rigidBody)
   {
      return single().getExternalWrench(rigidBody);
   }

   /** setExternalWrench(rigidBody, externalWrench) (java:463-472): stored in the body-fixed frame (setMatchingFrame). */
   public void setExternalWrench(RigidBodyReadOnly rigidBody, WrenchReadOnly externalWrench)
   {
      single().setExternalWrench(rigidBody, externalWrench);
   }

   /**
    * compute() (java:481-484): the reference's own signature.  Configuration, velocity and desired acceleration are read from the joints
    * (the caller has set them, as it does for the reference), one configuration goes through the HIP path, and the results are the
    * reference's: getJointTauMatrix() is nv x 1, getComputedJointTau(joint) is N x 1, writeComputedJointWrenches writes them back.
    */
   public void compute()
   {
      compute((DMatrix) null);
   }

   /** compute(DMatrix jointAccelerationMatrix) (java:496-501): accelerations from the given nv x 1 matrix instead of the joints. */
   public void compute(DMatrix jointAccelerationMatrix)
   {
      HipSingleState s = single();
      s.readConfigurationAndVelocity();
      s.readAccelerations(jointAccelerationMatrix);
      boolean wrenches = s.packExternalWrenches();
      lastBatch = null;
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment qSeg = arena.allocateFrom(JAVA_DOUBLE, s.q.data), qdSeg = arena.allocateFrom(JAVA_DOUBLE, s.qd.data),
               qddSeg = arena.allocateFrom(JAVA_DOUBLE, s.qdd.data), g = arena.allocateFrom(JAVA_DOUBLE, gravity);
         MemorySegment f = wrenches ? arena.allocateFrom(JAVA_DOUBLE, s.wrenchRow.data) : MemorySegment.NULL;
         MemorySegment options = MecanoHipNative.options(arena, considerCoriolisAndCentrifugalForces, considerJointAccelerations, rootAcceleration);
         MemorySegment tau = arena.allocate(JAVA_DOUBLE, Math.max(1L, model.nv));
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.RNEA_HOST.invokeExact(model.handle, 1L, qSeg, qdSeg, qddSeg, g, f, options, tau));
         jointTauMatrix.reshape(model.nv, 1);
         MemorySegment.copy(tau, JAVA_DOUBLE, 0, jointTauMatrix.data, 0, model.nv);
      }
      singleResult = true;
   }

   /** writeComputedJointWrench(joint) (java:639-653): joint.setJointTau(0, getComputedJointTau(joint)) after a one-configuration compute. */
   public boolean writeComputedJointWrench(JointBasics joint)
   {
      if (!singleResult)
         throw new IllegalStateException("compute() or compute(DMatrix) first: a batch of configurations cannot be written into one joint");
      DMatrixRMaj jointTau = single().rowsOf(joint, jointTauMatrix);
      if (jointTau == null)
         return false;
      joint.setJointTau(0, jointTau);
      return true;
   }

   /** writeComputedJointWrenches(JointBasics[]) (java:613-617). */
   public void writeComputedJointWrenches(JointBasics[] joints)
   {
      for (JointBasics joint : joints)
         writeComputedJointWrench(joint);
   }

   /** writeComputedJointWrenches(List) (java:625-629). */
   public void writeComputedJointWrenches(List<? extends JointBasics> joints)
   {
      for (int i = 0; i < joints.size(); i++)
         writeComputedJointWrench(joints.get(i));
   }

   /**
    * External wrenches of every configuration: B x 6 n, row b = for each listed joint's successor body (moment, force) expressed in its
    * body-fixed frame -- what setExternalWrench(body, wrench) stores after setMatchingFrame (java:444-472).  null = none
    * (setExternalWrenchesToZero, java:430-436).
    */
   public void setExternalWrenches(DMatrixRMaj wrenches)
   {
      if (wrenches != null && wrenches.getNumCols() != 6 * model.numberOfJoints)
         throw new org.ejml.MatrixDimensionException("Expected B x " + 6 * model.numberOfJoints);
      externalWrenches = wrenches;
   }

   public void setExternalWrenchesToZero()
   {
      externalWrenches = null;
      if (single != null)
         single.setExternalWrenchesToZero();
   }

   /** tau = ID(q, qd, qdd) for every row; q: B x nq, qd and qdd: B x nv (InverseDynamicsCalculator.compute(DMatrix), java:496-501). */
   public void compute(DMatrixRMaj q, DMatrixRMaj qd, DMatrixRMaj qdd)
   {
      int B = q.getNumRows();
      if (q.getNumCols() != model.nq || qd.getNumCols() != model.nv || qdd.getNumCols() != model.nv || qd.getNumRows() != B || qdd.getNumRows() != B)
         throw new org.ejml.MatrixDimensionException("Expected q: B x " + model.nq + ", qd and qdd: B x " + model.nv);
      if (externalWrenches != null && externalWrenches.getNumRows() != B)
         throw new org.ejml.MatrixDimensionException("External wrenches: expected " + B + " rows");
      jointTauMatrix.reshape(B, model.nv);
      lastBatch = null;
      singleResult = false;
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment qSeg = arena.allocateFrom(JAVA_DOUBLE, q.data), qdSeg = arena.allocateFrom(JAVA_DOUBLE, qd.data),
               qddSeg = arena.allocateFrom(JAVA_DOUBLE, qdd.data), g = arena.allocateFrom(JAVA_DOUBLE, gravity);
         MemorySegment f = externalWrenches == null ? MemorySegment.NULL : arena.allocateFrom(JAVA_DOUBLE, externalWrenches.data);
         MemorySegment options = MecanoHipNative.options(arena, considerCoriolisAndCentrifugalForces, considerJointAccelerations, rootAcceleration);
         MemorySegment tau = arena.allocate(JAVA_DOUBLE, Math.max(1L, (long) B * model.nv));
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.RNEA_HOST.invokeExact(model.handle, (long) B, qSeg, qdSeg, qddSeg, g, f, options, tau));
         MemorySegment.copy(tau, JAVA_DOUBLE, 0, jointTauMatrix.data, 0, B * model.nv);
      }
   }

   /**
    * The same on device-resident state: reads batch.q / qd / qdd (and batch.fExt when external wrenches were uploaded there), writes
    * batch.tau, the per-body accelerations and twists (getAccelerationProvider, java:242-250, 660-663) and the joint wrenches
    * (getComputedJointWrench, java:578-585).  Nothing crosses PCIe; download what the host needs.
    */
   public void compute(HipDeviceBatch batch, boolean withExternalWrenches)
   {
      lastBatch = batch;
      singleResult = false;
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment g = arena.allocateFrom(JAVA_DOUBLE, gravity);
         MemorySegment f = withExternalWrenches ? batch.fExt : MemorySegment.NULL;
         MemorySegment options = MecanoHipNative.options(arena, considerCoriolisAndCentrifugalForces, considerJointAccelerations, rootAcceleration, batch.context);
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.RNEA_BODIES.invokeExact(model.handle, (long) batch.batchSize, batch.q, batch.qd, batch.qdd, g, f,
                                                                                   options, batch.tau, batch.bodyAcceleration, batch.bodyTwist));
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.RNEA_JOINT_WRENCHES.invokeExact(model.handle, (long) batch.batchSize, batch.q, batch.qd,
                                                                                           batch.qdd, g, f, options, batch.tau, batch.jointWrench));
      }
   }

   /** B x nv; row b, column getJointDoFIndices(joint)[k] is the effort of that DoF for configuration b (InverseDynamicsCalculator.getJointTauMatrix, java:567). */
   public DMatrixRMaj getJointTauMatrix()
   {
      if (lastBatch != null)
         lastBatch.getEffort(jointTauMatrix);
      return jointTauMatrix;
   }

   /** getComputedJointTau(joint) (java:587-602): the joint's columns of the tau matrix, B x N. */
   public DMatrixRMaj getComputedJointTau(JointReadOnly joint)
   {
      if (singleResult)
         return single().rowsOf(joint, jointTauMatrix); // N x 1, like the reference's
      if (model.indexOf(joint) < 0)
         return null;
      int[] columns = input.getJointMatrixIndexProvider().getJointDoFIndices(joint);
      DMatrixRMaj all = getJointTauMatrix(), out = new DMatrixRMaj(all.getNumRows(), columns.length);
      for (int b = 0; b < all.getNumRows(); b++)
         for (int k = 0; k < columns.length; k++)
            out.set(b, k, all.get(b, columns[k]));
      return out;
   }

   /**
    * getComputedJointWrench(joint) (java:578-585): B x 6 (moment, force) in the joint's frame after the joint; null for a joint this
    * calculator does not consider.  Needs compute(HipDeviceBatch, ...).
    */
   public DMatrixRMaj getComputedJointWrench(JointReadOnly joint)
   {
      return rowsOf(joint.getSuccessor(), jointWrenches, lastBatch == null ? null : lastBatch.jointWrench);
   }

   /** getAccelerationProvider().getAccelerationOfBody(body) (java:242-250): B x 6 (angular, linear) in the body-fixed frame. */
   public DMatrixRMaj getAccelerationOfBody(RigidBodyReadOnly body)
   {
      return rowsOf(body, bodyAccelerations, lastBatch == null ? null : lastBatch.bodyAcceleration);
   }

   /** Twist of the body with respect to the inertial frame, in its body-fixed frame (MovingReferenceFrame.getTwistOfFrame), B x 6. */
   public DMatrixRMaj getTwistOfBody(RigidBodyReadOnly body)
   {
      return rowsOf(body, bodyTwists, lastBatch == null ? null : lastBatch.bodyTwist);
   }

   /**
    * getAccelerationProvider().getRelativeAcceleration(base, body) (RigidBodyAccelerationProvider.java:66, 199-235): B x 6, acceleration of
    * body with respect to base expressed in body's body-fixed frame; the root body is a valid base / body.  null when either is not
    * considered.
    */
   public DMatrixRMaj getRelativeAcceleration(RigidBodyReadOnly base, RigidBodyReadOnly body)
   {
      if (lastBatch == null)
         throw new IllegalStateException("compute(HipDeviceBatch, ...) first");
      int baseIndex = base.isRootBody() ? -1 : model.indexOf(base.getParentJoint()), bodyIndex = body.isRootBody() ? -1 : model.indexOf(body.getParentJoint());
      if ((!base.isRootBody() && baseIndex < 0) || (!body.isRootBody() && bodyIndex < 0))
         return null;
      HipDeviceBatch batch = lastBatch;
      DMatrixRMaj out = new DMatrixRMaj(batch.batchSize, 6);
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment g = arena.allocateFrom(JAVA_DOUBLE, gravity), bases = arena.allocateFrom(JAVA_INT, baseIndex), bodies = arena.allocateFrom(JAVA_INT, bodyIndex);
         MemorySegment options = MecanoHipNative.options(arena, considerCoriolisAndCentrifugalForces, considerJointAccelerations, rootAcceleration, batch.context);
         MemorySegment twist = considerCoriolisAndCentrifugalForces ? batch.bodyTwist : MemorySegment.NULL;
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.RELATIVE_ACCELERATION.invokeExact(model.handle, (long) batch.batchSize, batch.q, batch.bodyAcceleration,
                                                                                             twist, g, 1, bases, bodies, options, batch.pairOutput));
         batch.download(batch.pairOutput, batch.batchSize, 6, out);
      }
      return out;
   }

   private DMatrixRMaj rowsOf(RigidBodyReadOnly body, DMatrixRMaj cache, MemorySegment deviceBuffer)
   {
      if (lastBatch == null)
         throw new IllegalStateException("compute(HipDeviceBatch, ...) first");
      int index = body.isRootBody() ? -1 : model.indexOf(body.getParentJoint());
      if (index < 0)
         return null; // like the reference: bodies the calculator does not consider have no entry (java:242-246)
      lastBatch.download(deviceBuffer, lastBatch.batchSize, 6 * model.numberOfJoints, cache);
      DMatrixRMaj out = new DMatrixRMaj(lastBatch.batchSize, 6);
      for (int b = 0; b < lastBatch.batchSize; b++)
         for (int k = 0; k < 6; k++)
            out.set(b, k, cache.get(b, 6 * index + k));
      return out;
   }

   public HipMultiBodyModel getModel()
   {
      return model;
   }

   public MultiBodySystemReadOnly getInput()
   {
      return input;
   }

   @Override
   public void close()
   {
      model.close();
   }
}
