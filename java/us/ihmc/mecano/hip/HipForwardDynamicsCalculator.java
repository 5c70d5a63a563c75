package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.util.List;
import java.util.function.Function;

import org.ejml.data.DMatrix;
import org.ejml.data.DMatrixRMaj;

import us.ihmc.euclid.tuple3D.interfaces.Tuple3DReadOnly;
import us.ihmc.mecano.algorithms.ForwardDynamicsCalculator.JointSourceMode;
import us.ihmc.mecano.multiBodySystem.interfaces.JointBasics;
import us.ihmc.mecano.multiBodySystem.interfaces.JointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.MultiBodySystemReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.RigidBodyReadOnly;
import us.ihmc.mecano.spatial.interfaces.FixedFrameWrenchBasics;
import us.ihmc.mecano.spatial.interfaces.SpatialAccelerationReadOnly;
import us.ihmc.mecano.spatial.interfaces.WrenchReadOnly;

import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;

/**
 * Batched drop-in for ForwardDynamicsCalculator (ABA): same constructor arguments, gravity / external-wrench setters, joint source modes
 * (ForwardDynamicsCalculator.java:45-57, 400-444), compute(q, qd, tau[, qdd]) over B stacked configurations (one ROW per configuration, see
 * HipInverseDynamicsCalculator), and a simulation step that stays on the device.  Source only: this image has no JDK (INTEGRATION.md).
 */
public class HipForwardDynamicsCalculator implements AutoCloseable
{
   private final MultiBodySystemReadOnly input;
   private final HipMultiBodyModel model;
   private final double[] gravity = new double[3];
   private double[] rootAcceleration; // six components (angular, linear) once setRootAcceleration was called, else null (gravity rules)
   private HipSingleState single;     // the one-configuration face (compute() / compute(DMatrix) / compute(DMatrix, DMatrix))
   private boolean singleResult;
   private DMatrixRMaj externalWrenches;
   private final int[] sourceModes;
   private boolean anyAccelerationSource;
   private final DMatrixRMaj jointAccelerationMatrix = new DMatrixRMaj(0, 0), jointTauMatrix = new DMatrixRMaj(0, 0);

   public HipForwardDynamicsCalculator(MultiBodySystemReadOnly input)
   {
      this(input, true);
   }

   /** ForwardDynamicsCalculator(MultiBodySystemReadOnly, boolean considerIgnoredSubtreesInertia) (java:154-160). */
   public HipForwardDynamicsCalculator(MultiBodySystemReadOnly input, boolean considerIgnoredSubtreesInertia)
   {
      this.input = input;
      model = new HipMultiBodyModel(input, considerIgnoredSubtreesInertia);
      sourceModes = new int[model.numberOfJoints];
   }

   /** ForwardDynamicsCalculator.setGravitationalAcceleration(double): gravity along z, usually negative (java:304-319). */
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
      rootAcceleration = null; // the last setter wins: both write the root acceleration in the reference (java:259-264, 340)
   }

   /**
    * ForwardDynamicsCalculator.setRootAcceleration(SpatialAccelerationReadOnly) (java:330-343): angular and linear part, root-body
    * coordinates -- mh_options.root_acceleration.
    */
   public void setRootAcceleration(SpatialAccelerationReadOnly newRootAcceleration)
   {
      newRootAcceleration.checkReferenceFrameMatch(input.getRootBody().getBodyFixedFrame(), input.getRootBody().getBodyFixedFrame().getRootFrame(),
                                                   input.getRootBody().getBodyFixedFrame());
      rootAcceleration = new double[] {newRootAcceleration.getAngularPartX(), newRootAcceleration.getAngularPartY(), newRootAcceleration.getAngularPartZ(),
                                       newRootAcceleration.getLinearPartX(), newRootAcceleration.getLinearPartY(), newRootAcceleration.getLinearPartZ()};
   }

   private HipSingleState single()
   {
      if (single == null)
         single = new HipSingleState(input, model);
      return single;
   }

   /** getExternalWrench(rigidBody) (java:353-369): the live external wrench of that body for the one-configuration calls. */
   public FixedFrameWrenchBasics getExternalWrench(RigidBodyReadOnly rigidBody)
   {
      return single().getExternalWrench(rigidBody);
   }

   /** setExternalWrench(rigidBody, externalWrench) (java:371-381). */
   public void setExternalWrench(RigidBodyReadOnly rigidBody, WrenchReadOnly externalWrench)
   {
      single().setExternalWrench(rigidBody, externalWrench);
   }

   /** setExternalWrenchesToZero() (java:348-351). */
   public void setExternalWrenchesToZero()
   {
      externalWrenches = null;
      if (single != null)
         single.setExternalWrenchesToZero();
   }

   /** compute() (java:475-478): configuration, velocity and efforts (accelerations of acceleration-source joints) from the joints. */
   public void compute()
   {
      compute((DMatrix) null, (DMatrix) null);
   }

   /** compute(DMatrix jointTauInput) (java:489-492). */
   public void compute(DMatrix jointTauInput)
   {
      compute(jointTauInput, (DMatrix) null);
   }

   /**
    * compute(DMatrix jointTauInput, DMatrix jointAccelerationInput) (java:508-520), the reference's own signature: one configuration,
    * read from the joints, through the HIP path.  Afterwards getJointAccelerationMatrix() / getJointTauMatrix() are nv x 1 like the
    * reference's, getComputedJointAcceleration(joint) is N x 1, writeComputedJointAccelerations writes them into the joints.
    */
   public void compute(DMatrix jointTauInput, DMatrix jointAccelerationInput)
   {
      HipSingleState s = single();
      s.readConfigurationAndVelocity();
      s.readEfforts(jointTauInput);
      if (anyAccelerationSource)
         s.readAccelerations(jointAccelerationInput);
      boolean wrenches = s.packExternalWrenches();
      DMatrixRMaj keep = externalWrenches;
      externalWrenches = wrenches ? s.wrenchRow : null;
      try
      { // the batched path with B = 1: row vectors share their backing arrays with the column vectors
         DMatrixRMaj q = DMatrixRMaj.wrap(1, model.nq, s.q.data), qd = DMatrixRMaj.wrap(1, model.nv, s.qd.data), tau = DMatrixRMaj.wrap(1, model.nv, s.tau.data);
         compute(q, qd, tau, anyAccelerationSource ? DMatrixRMaj.wrap(1, model.nv, s.qdd.data) : null);
      }
      finally
      {
         externalWrenches = keep;
      }
      jointAccelerationMatrix.reshape(model.nv, 1); // same numbers, the reference's shape
      jointTauMatrix.reshape(model.nv, 1);
      singleResult = true;
   }

   /** getComputedJointAcceleration(joint) (java:600-610): N x 1 after a one-configuration compute, null for a joint that is not considered. */
   public DMatrixRMaj getComputedJointAcceleration(JointReadOnly joint)
   {
      if (!singleResult)
         throw new IllegalStateException("compute(), compute(DMatrix) or compute(DMatrix, DMatrix) first");
      return single().rowsOf(joint, jointAccelerationMatrix);
   }

   /** writeComputedJointAcceleration(joint) (java:699-708). */
   public boolean writeComputedJointAcceleration(JointBasics joint)
   {
      DMatrixRMaj jointAcceleration = getComputedJointAcceleration(joint);
      if (jointAcceleration == null)
         return false;
      joint.setJointAcceleration(0, jointAcceleration);
      return true;
   }

   /** writeComputedJointAccelerations(JointBasics[]) (java:670-674). */
   public void writeComputedJointAccelerations(JointBasics[] joints)
   {
      for (JointBasics joint : joints)
         writeComputedJointAcceleration(joint);
   }

   /** writeComputedJointAccelerations(List) (java:684-688). */
   public void writeComputedJointAccelerations(List<? extends JointBasics> joints)
   {
      for (int i = 0; i < joints.size(); i++)
         writeComputedJointAcceleration(joints.get(i));
   }

   /** B x 6 n, (moment, force) per successor body in its body-fixed frame (setExternalWrench, java:348-381); null = none. */
   public void setExternalWrenches(DMatrixRMaj wrenches)
   {
      if (wrenches != null && wrenches.getNumCols() != 6 * model.numberOfJoints)
         throw new org.ejml.MatrixDimensionException("Expected B x " + 6 * model.numberOfJoints);
      externalWrenches = wrenches;
   }

   /** setJointSourceMode(joint, mode) (java:400-415): ACCELERATION_SOURCE joints take qdd as an input and return tau. */
   public void setJointSourceMode(JointReadOnly joint, JointSourceMode mode)
   {
      int index = model.indexOf(joint);
      if (index < 0)
         throw new IllegalArgumentException("The joint " + joint.getName() + " is not considered by this calculator."); // java:407-409
      sourceModes[index] = mode == JointSourceMode.ACCELERATION_SOURCE ? 1 : 0;
      pushSourceModes();
   }

   /** setJointSourceModes(Function) (java:423-433). */
   public void setJointSourceModes(Function<JointReadOnly, JointSourceMode> modeFunction)
   {
      List<? extends JointReadOnly> joints = input.getJointMatrixIndexProvider().getIndexedJointsInOrder();
      for (int i = 0; i < joints.size(); i++)
         sourceModes[i] = modeFunction.apply(joints.get(i)) == JointSourceMode.ACCELERATION_SOURCE ? 1 : 0;
      pushSourceModes();
   }

   /** resetJointSourceModes() (java:441-444). */
   public void resetJointSourceModes()
   {
      java.util.Arrays.fill(sourceModes, 0);
      pushSourceModes();
   }

   private void pushSourceModes()
   {
      anyAccelerationSource = false;
      for (int mode : sourceModes)
         anyAccelerationSource |= mode == 1;
      model.setJointSourceModes(anyAccelerationSource ? sourceModes : null);
   }

   /** qdd = FD(q, qd, tau) for every row; q: B x nq, qd and tau: B x nv (ForwardDynamicsCalculator.compute(DMatrix), java:475-490). */
   public void compute(DMatrixRMaj q, DMatrixRMaj qd, DMatrixRMaj tau)
   {
      compute(q, qd, tau, null);
   }

   /**
    * compute(DMatrix tau, DMatrix qdd) (java:508-520): tau is read at the DoFs of the effort-source joints, qdd at those of the
    * acceleration-source joints; afterwards getJointAccelerationMatrix() and getJointTauMatrix() hold every joint's acceleration and effort.
    */
   public void compute(DMatrixRMaj q, DMatrixRMaj qd, DMatrixRMaj tau, DMatrixRMaj qddGiven)
   {
      int B = q.getNumRows();
      if (q.getNumCols() != model.nq || qd.getNumCols() != model.nv || tau.getNumCols() != model.nv || qd.getNumRows() != B || tau.getNumRows() != B)
         throw new org.ejml.MatrixDimensionException("Expected q: B x " + model.nq + ", qd and tau: B x " + model.nv); // java:522-533
      if (anyAccelerationSource && (qddGiven == null || qddGiven.getNumRows() != B || qddGiven.getNumCols() != model.nv))
         throw new org.ejml.MatrixDimensionException("Acceleration-source joints need their accelerations: B x " + model.nv);
      jointAccelerationMatrix.reshape(B, model.nv);
      jointTauMatrix.reshape(B, model.nv);
      singleResult = false;
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment g = arena.allocateFrom(JAVA_DOUBLE, gravity);
         MemorySegment options = MecanoHipNative.options(arena, true, true, rootAcceleration);
         if (!anyAccelerationSource)
         { // host-pointer entry point: chunked copies overlapped with the kernels
            MemorySegment qSeg = arena.allocateFrom(JAVA_DOUBLE, q.data), qdSeg = arena.allocateFrom(JAVA_DOUBLE, qd.data),
                  tauSeg = arena.allocateFrom(JAVA_DOUBLE, tau.data);
            MemorySegment f = externalWrenches == null ? MemorySegment.NULL : arena.allocateFrom(JAVA_DOUBLE, externalWrenches.data);
            MemorySegment qdd = arena.allocate(JAVA_DOUBLE, Math.max(1L, (long) B * model.nv));
            MecanoHipNative.invoke(() -> (int) MecanoHipNative.ABA_HOST.invokeExact(model.handle, (long) B, qSeg, qdSeg, tauSeg, g, f, options, qdd));
            MemorySegment.copy(qdd, JAVA_DOUBLE, 0, jointAccelerationMatrix.data, 0, B * model.nv);
            jointTauMatrix.setTo(tau);
            return;
         }
         // acceleration-source joints: mh_aba_locked_f64 takes device pointers
         try (HipDeviceBatch batch = new HipDeviceBatch(model, B))
         {
            batch.setConfiguration(q);
            batch.setVelocity(qd);
            batch.setEffort(tau);
            batch.setAcceleration(qddGiven);
            if (externalWrenches != null)
               batch.upload(externalWrenches, batch.fExt);
            MemorySegment f = externalWrenches == null ? MemorySegment.NULL : batch.fExt;
            // in place: qdd_out == qdd_in, tau_out == tau (allowed by the entry point)
            MecanoHipNative.invoke(() -> (int) MecanoHipNative.ABA_LOCKED.invokeExact(model.handle, (long) B, batch.q, batch.qd, batch.tau, batch.qdd, g, f,
                                                                                     options, batch.qdd, batch.tau));
            batch.getAcceleration(jointAccelerationMatrix);
            batch.getEffort(jointTauMatrix);
         }
      }
   }

   /**
    * On device-resident state: batch.qdd = FD(batch.q, batch.qd, batch.tau), plus the per-body accelerations / twists and the joint
    * wrenches (getJointWrench, java:642-650).  Effort-source joints only.
    */
   public void compute(HipDeviceBatch batch, boolean withExternalWrenches)
   {
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment g = arena.allocateFrom(JAVA_DOUBLE, gravity), options = MecanoHipNative.options(arena, true, true, rootAcceleration, batch.context);
         MemorySegment f = withExternalWrenches ? batch.fExt : MemorySegment.NULL;
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.ABA_BODIES.invokeExact(model.handle, (long) batch.batchSize, batch.q, batch.qd, batch.tau, g, f,
                                                                                  options, batch.qdd, batch.bodyAcceleration, batch.bodyTwist));
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.ABA_JOINT_WRENCHES.invokeExact(model.handle, (long) batch.batchSize, batch.q, batch.qd, batch.tau,
                                                                                          g, f, options, batch.qdd, batch.jointWrench));
      }
   }

   /**
    * One simulation step that never leaves the device: qdd = FD(q, qd, tau), then MultiBodySystemStateIntegrator.doubleIntegrateFromAcceleration
    * with step dt, q and qd updated in place (the loop body of MultiBodySystemStateIntegratorTest.java:245-250).  One kernel launch for models
    * with a tree-split code object.
    */
   public void simulationStep(HipDeviceBatch batch, double dt, boolean withExternalWrenches)
   {
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment g = arena.allocateFrom(JAVA_DOUBLE, gravity), options = MecanoHipNative.options(arena, true, true, rootAcceleration, batch.context);
         MemorySegment f = withExternalWrenches ? batch.fExt : MemorySegment.NULL;
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.ABA_INTEGRATE.invokeExact(model.handle, (long) batch.batchSize, dt, batch.q, batch.qd, batch.tau, g, f,
                                                                                     options, batch.qdd, batch.q, batch.qd));
      }
   }

   /** B x nv (ForwardDynamicsCalculator.getJointAccelerationMatrix, java:556-567). */
   public DMatrixRMaj getJointAccelerationMatrix()
   {
      return jointAccelerationMatrix;
   }

   /** B x nv: the given efforts for effort sources, the computed ones for acceleration sources (getJointTauMatrix, java:580-591). */
   public DMatrixRMaj getJointTauMatrix()
   {
      return jointTauMatrix;
   }

   /** getJointWrench(joint) (java:642-650) after compute(HipDeviceBatch, ...): B x 6 in the frame after the joint; null when not considered. */
   public DMatrixRMaj getJointWrench(HipDeviceBatch batch, JointReadOnly joint)
   {
      int index = model.indexOf(joint);
      if (index < 0)
         return null;
      DMatrixRMaj all = new DMatrixRMaj(0, 0), out = new DMatrixRMaj(batch.batchSize, 6);
      batch.download(batch.jointWrench, batch.batchSize, 6 * model.numberOfJoints, all);
      for (int b = 0; b < batch.batchSize; b++)
         for (int k = 0; k < 6; k++)
            out.set(b, k, all.get(b, 6 * index + k));
      return out;
   }

   public HipMultiBodyModel getModel()
   {
      return model;
   }

   /** for the calculators built on this one (HipMultiBodyResponseCalculator shares the model and the joint source modes) */
   boolean hasAccelerationSources()
   {
      return anyAccelerationSource;
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
