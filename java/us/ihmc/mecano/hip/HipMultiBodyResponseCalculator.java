package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import org.ejml.data.DMatrixRMaj;

import us.ihmc.mecano.multiBodySystem.interfaces.JointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.MultiBodySystemReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.RigidBodyReadOnly;

import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;

/**
 * Batched drop-in for MultiBodyResponseCalculator (java:120-140, 224-250, 288-935): the change in joint and body accelerations (for
 * impulses: velocities) that test wrenches on bodies and test efforts at joints produce, for B configurations at once.  The reference
 * walks the disturbance up and down the articulated-body quantities its forward dynamics left behind (java:1206-1338); that recursion is
 * the forward dynamics' own with velocities, gravity and efforts at zero and the test wrench as the only external wrench, so this class
 * asks mh_aba_bodies_f64 (mh_aba_locked_f64 with acceleration-source joints, whose change stays zero, java:1230-1238) for exactly that.
 * <p>
 * reset(q) sets the B configurations (one ROW each; the reference reads them from the joints' frames).  Test wrenches are B x 6 (moment,
 * force) on the target body expressed in its body-fixed frame -- change the frame of a Wrench before stacking it; efforts are B x dofs.
 * Several apply calls accumulate (MultiBodyResponseCalculatorTest.java:749-812).
 * </p>
 * Source only: this image has no JDK (INTEGRATION.md).
 */
public class HipMultiBodyResponseCalculator implements AutoCloseable
{
   private final HipForwardDynamicsCalculator forwardDynamicsCalculator;
   private final boolean ownsForwardDynamicsCalculator;
   private final HipMultiBodyModel model;
   private HipDeviceBatch batch;
   private DMatrixRMaj testWrenches, testEfforts;
   private boolean upToDate;
   private final DMatrixRMaj motionChangeMatrix = new DMatrixRMaj(0, 0), bodyMotionChange = new DMatrixRMaj(0, 0);

   /** MultiBodyResponseCalculator(MultiBodySystemReadOnly) (java:120-123) */
   public HipMultiBodyResponseCalculator(MultiBodySystemReadOnly input)
   {
      this(new HipForwardDynamicsCalculator(input), true);
   }

   /** MultiBodyResponseCalculator(ForwardDynamicsCalculator) (java:136-140): shares the model and the joint source modes */
   public HipMultiBodyResponseCalculator(HipForwardDynamicsCalculator forwardDynamicsCalculator)
   {
      this(forwardDynamicsCalculator, false);
   }

   private HipMultiBodyResponseCalculator(HipForwardDynamicsCalculator forwardDynamicsCalculator, boolean owns)
   {
      this.forwardDynamicsCalculator = forwardDynamicsCalculator;
      ownsForwardDynamicsCalculator = owns;
      model = forwardDynamicsCalculator.getModel();
   }

   /** java:224-227 */
   public HipForwardDynamicsCalculator getForwardDynamicsCalculator()
   {
      return forwardDynamicsCalculator;
   }

   /** java:232-250 plus the configurations: q is B x nq.  Forgets every disturbance. */
   public void reset(DMatrixRMaj q)
   {
      int B = q.getNumRows();
      if (q.getNumCols() != model.nq)
         throw new org.ejml.MatrixDimensionException("Expected q: B x " + model.nq);
      if (batch == null || batch.batchSize != B)
      {
         if (batch != null)
            batch.close();
         batch = new HipDeviceBatch(model, B);
         batch.setVelocity(new DMatrixRMaj(B, model.nv)); // zero for good: the response does not depend on the velocities
      }
      batch.setConfiguration(q);
      testWrenches = new DMatrixRMaj(B, 6 * model.numberOfJoints);
      testEfforts = new DMatrixRMaj(B, model.nv);
      upToDate = false;
   }

   /** java:608-627 (and applyRigidBodyImpulse, java:640-659: the same linear map); false for a body this calculator does not consider. */
   public boolean applyRigidBodyWrench(RigidBodyReadOnly target, DMatrixRMaj wrenchInBodyFixedFrame)
   {
      int index = target.getParentJoint() == null ? -1 : model.indexOf(target.getParentJoint());
      if (index < 0)
         return false;
      for (int b = 0; b < batch.batchSize; b++)
         for (int c = 0; c < 6; c++)
            testWrenches.add(b, 6 * index + c, wrenchInBodyFixedFrame.get(b, c));
      upToDate = false;
      return true;
   }

   /** java:685-735 (and applyJointImpulse, java:750-815); effort is B x (degrees of freedom of the joint). */
   public boolean applyJointWrench(JointReadOnly target, DMatrixRMaj effort)
   {
      if (model.indexOf(target) < 0)
         return false;
      int[] rows = forwardDynamicsCalculator.getInput().getJointMatrixIndexProvider().getJointDoFIndices(target);
      for (int b = 0; b < batch.batchSize; b++)
         for (int c = 0; c < rows.length; c++)
            testEfforts.add(b, rows[c], effort.get(b, c));
      upToDate = false;
      return true;
   }

   private void propagate()
   {
      if (upToDate)
         return;
      int B = batch.batchSize;
      batch.setEffort(testEfforts);
      batch.setExternalWrenches(testWrenches);
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment zeroGravity = arena.allocateFrom(JAVA_DOUBLE, new double[3]), options = MecanoHipNative.options(arena, true, true);
         if (forwardDynamicsCalculator.hasAccelerationSources())
         { // given accelerations of the acceleration sources: zero (batch.qdd in, change out, in place)
            batch.setAcceleration(new DMatrixRMaj(B, model.nv));
            MecanoHipNative.invoke(() -> (int) MecanoHipNative.ABA_LOCKED.invokeExact(model.handle, (long) B, batch.q, batch.qd, batch.tau, batch.qdd, zeroGravity,
                                                                                     batch.fExt, options, batch.qdd, batch.tau));
         }
         else
            MecanoHipNative.invoke(() -> (int) MecanoHipNative.ABA_BODIES.invokeExact(model.handle, (long) B, batch.q, batch.qd, batch.tau, zeroGravity, batch.fExt,
                                                                                     options, batch.qdd, batch.bodyAcceleration, batch.bodyTwist));
      }
      batch.getAcceleration(motionChangeMatrix);
      upToDate = true;
   }

   /** java:823-829: B x nv, the change in the joint accelerations */
   public DMatrixRMaj propagateWrench()
   {
      propagate();
      return motionChangeMatrix;
   }

   /** java:836-842: B x nv, the change in the joint velocities */
   public DMatrixRMaj propagateImpulse()
   {
      return propagateWrench();
   }

   /**
    * getAccelerationChangeProvider().getAccelerationOfBody(body) / getTwistChangeProvider().getTwistOfBody(body) (java:859-876): B x 6
    * (angular, linear), in the body-fixed frame; null for a body this calculator does not consider.  Effort-source joints only.
    */
   public DMatrixRMaj getMotionChangeOfBody(RigidBodyReadOnly body)
   {
      int index = body.getParentJoint() == null ? -1 : model.indexOf(body.getParentJoint());
      if (index < 0)
         return null;
      if (forwardDynamicsCalculator.hasAccelerationSources())
         throw new UnsupportedOperationException("Per-body changes need every joint to be an effort source.");
      propagate();
      batch.download(batch.bodyAcceleration, batch.batchSize, 6 * model.numberOfJoints, bodyMotionChange);
      DMatrixRMaj out = new DMatrixRMaj(batch.batchSize, 6);
      for (int b = 0; b < batch.batchSize; b++)
         for (int c = 0; c < 6; c++)
            out.set(b, c, bodyMotionChange.get(b, 6 * index + c));
      return out;
   }

   /**
    * computeRigidBodyApparentSpatialInertiaInverse(target, target.getBodyFixedFrame(), ...) (java:288-330): B x 36, row b = the 6 x 6 matrix
    * (row-major) that maps a wrench on the body to the change of its spatial acceleration; null for a body that is not considered.  The
    * disturbances applied so far are kept.
    */
   public DMatrixRMaj computeRigidBodyApparentSpatialInertiaInverse(RigidBodyReadOnly target)
   {
      int index = target.getParentJoint() == null ? -1 : model.indexOf(target.getParentJoint());
      if (index < 0)
         return null;
      int B = batch.batchSize;
      DMatrixRMaj savedWrenches = testWrenches, savedEfforts = testEfforts;
      DMatrixRMaj out = new DMatrixRMaj(B, 36);
      for (int column = 0; column < 6; column++)
      {
         testWrenches = new DMatrixRMaj(B, 6 * model.numberOfJoints);
         testEfforts = new DMatrixRMaj(B, model.nv);
         for (int b = 0; b < B; b++)
            testWrenches.set(b, 6 * index + column, 1.0);
         upToDate = false;
         DMatrixRMaj change = getMotionChangeOfBody(target);
         for (int b = 0; b < B; b++)
            for (int row = 0; row < 6; row++)
               out.set(b, 6 * row + column, change.get(b, row));
      }
      testWrenches = savedWrenches;
      testEfforts = savedEfforts;
      upToDate = false;
      return out;
   }

   @Override
   public void close()
   {
      if (batch != null)
         batch.close();
      if (ownsForwardDynamicsCalculator)
         forwardDynamicsCalculator.close();
   }
}
