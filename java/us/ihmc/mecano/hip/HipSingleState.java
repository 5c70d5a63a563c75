package us.ihmc.mecano.hip;

import java.util.HashMap;
import java.util.List;
import java.util.Map;

import org.ejml.data.DMatrix;
import org.ejml.data.DMatrixRMaj;

import us.ihmc.mecano.multiBodySystem.interfaces.JointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.MultiBodySystemReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.RigidBodyReadOnly;
import us.ihmc.mecano.spatial.Wrench;
import us.ihmc.mecano.spatial.interfaces.FixedFrameWrenchBasics;
import us.ihmc.mecano.spatial.interfaces.WrenchReadOnly;
import us.ihmc.mecano.tools.JointStateType;
import us.ihmc.mecano.tools.MultiBodySystemTools;

/**
 * The ONE-configuration face of the calculators: what InverseDynamicsCalculator.compute() / compute(DMatrix) and
 * ForwardDynamicsCalculator.compute() / compute(DMatrix) / compute(DMatrix, DMatrix) read from the joints, packed into the B = 1 rows the
 * C-ABI takes.  State comes out of the joints with MultiBodySystemTools.extractJointsState (tools/MultiBodySystemTools.java:1433-1491) in
 * the order of the system's JointMatrixIndexProvider -- the order the model was flattened in -- and external wrenches are kept per body,
 * expressed in the body-fixed frame, exactly like the reference's recursion steps keep them (setMatchingFrame,
 * InverseDynamicsCalculator.java:458-472).  NOT compiled in this repository's image (no JVM there).
 */
final class HipSingleState
{
   private final MultiBodySystemReadOnly input;
   private final HipMultiBodyModel model;
   private final List<? extends JointReadOnly> joints;
   private final Map<RigidBodyReadOnly, Wrench> externalWrenches = new HashMap<>();
   final DMatrixRMaj q, qd, qdd, tau; // column vectors, as extractJointsState packs them; their backing arrays ARE the B = 1 rows
   final DMatrixRMaj wrenchRow;       // 1 x 6 n: (moment, force) per listed joint's successor, body-fixed frame

   HipSingleState(MultiBodySystemReadOnly input, HipMultiBodyModel model)
   {
      this.input = input;
      this.model = model;
      joints = input.getJointMatrixIndexProvider().getIndexedJointsInOrder();
      q = new DMatrixRMaj(model.nq, 1);
      qd = new DMatrixRMaj(model.nv, 1);
      qdd = new DMatrixRMaj(model.nv, 1);
      tau = new DMatrixRMaj(model.nv, 1);
      wrenchRow = new DMatrixRMaj(1, 6 * model.numberOfJoints);
      for (JointReadOnly joint : joints)
      {
         RigidBodyReadOnly body = joint.getSuccessor();
         externalWrenches.put(body, new Wrench(body.getBodyFixedFrame(), body.getBodyFixedFrame()));
      }
   }

   /** q and qd of the joints as they are now (the reference reads them through the reference frames the caller updated). */
   void readConfigurationAndVelocity()
   {
      MultiBodySystemTools.extractJointsState(joints, JointStateType.CONFIGURATION, q);
      MultiBodySystemTools.extractJointsState(joints, JointStateType.VELOCITY, qd);
   }

   /** InverseDynamicsCalculator.initializeJointAccelerationMatrix (java:503-524): null = the joints' own accelerations. */
   void readAccelerations(DMatrix given)
   {
      if (given == null)
         MultiBodySystemTools.extractJointsState(joints, JointStateType.ACCELERATION, qdd);
      else
         copyColumn(given, qdd);
   }

   /** ForwardDynamicsCalculator.compute(DMatrix jointTauInput) (java:489-520): null = the joints' own efforts. */
   void readEfforts(DMatrix given)
   {
      if (given == null)
         MultiBodySystemTools.extractJointsState(joints, JointStateType.EFFORT, tau);
      else
         copyColumn(given, tau);
   }

   private void copyColumn(DMatrix from, DMatrixRMaj to)
   {
      if (from.getNumRows() != to.getNumRows() || from.getNumCols() != 1) // ForwardDynamicsCalculator.java:522-533
         throw new org.ejml.MatrixDimensionException("Expected " + to.getNumRows() + " x 1, got " + from.getNumRows() + " x " + from.getNumCols());
      for (int i = 0; i < to.getNumRows(); i++)
         to.set(i, 0, from.get(i, 0));
   }

   /** getExternalWrench(rigidBody) (InverseDynamicsCalculator.java:444-461): the live wrench object of that body; modify it in place. */
   FixedFrameWrenchBasics getExternalWrench(RigidBodyReadOnly body)
   {
      return externalWrenches.get(body);
   }

   /** setExternalWrench(rigidBody, externalWrench) (java:463-472). */
   void setExternalWrench(RigidBodyReadOnly body, WrenchReadOnly wrench)
   {
      getExternalWrench(body).setMatchingFrame(wrench);
   }

   void setExternalWrenchesToZero()
   {
      externalWrenches.values().forEach(Wrench::setToZero);
   }

   /** packs the per-body wrenches into the 1 x 6 n row of the C-ABI; returns false when all of them are zero (pass NULL then) */
   boolean packExternalWrenches()
   {
      boolean any = false;
      for (int j = 0; j < joints.size(); j++)
      {
         Wrench w = externalWrenches.get(joints.get(j).getSuccessor());
         double[] six = {w.getAngularPartX(), w.getAngularPartY(), w.getAngularPartZ(), w.getLinearPartX(), w.getLinearPartY(), w.getLinearPartZ()};
         for (int k = 0; k < 6; k++)
         {
            wrenchRow.set(0, 6 * j + k, six[k]);
            any |= six[k] != 0.0;
         }
      }
      return any;
   }

   /** the rows of a result vector that belong to the joint, N x 1 (getComputedJointTau / getComputedJointAcceleration) */
   DMatrixRMaj rowsOf(JointReadOnly joint, DMatrixRMaj result)
   {
      if (model.indexOf(joint) < 0)
         return null;
      int[] rows = input.getJointMatrixIndexProvider().getJointDoFIndices(joint);
      DMatrixRMaj out = new DMatrixRMaj(rows.length, 1);
      for (int k = 0; k < rows.length; k++)
         out.set(k, 0, result.get(rows[k], 0));
      return out;
   }
}
