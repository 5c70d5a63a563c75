package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import org.ejml.data.DMatrixRMaj;

import us.ihmc.mecano.multiBodySystem.interfaces.JointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.MultiBodySystemReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.RigidBodyReadOnly;
import us.ihmc.mecano.spatial.interfaces.SpatialInertiaReadOnly;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;

/**
 * Batched drop-in for JointTorqueRegressorCalculator (java:101-133, 173-190, 360-502): tau = Y(q, qd, qdd) pi, ten inertial parameters per
 * body.  compute(q, qd, qdd) takes B stacked states (one ROW per configuration); getJointTorqueRegressorMatrix() is B x (nv * 10 N), row b =
 * the reference's nv x 10 N row-major matrix of configuration b.  One kernel launch evaluates every column of every configuration
 * (mh_regressor_f64); the reference runs one second pass of the inverse dynamics per body and parameter and edits the bodies' inertias while
 * it does so (java:733-745, 795-806) -- this class leaves the multi-body system untouched.
 * <p>
 * Body order: the ten columns of a body start at 10 * (index of its parent joint in input.getJointsToConsider()); the reference orders the
 * blocks by the iteration order of a HashMap (java:85, 123, 318-329).  getJointTorqueRegressorMatrixBlock(body, b, blockToPack) and
 * getParameterVectorSlice(body) are the order-free accessors.
 * </p>
 * Source only: this image has no JDK (INTEGRATION.md).
 */
public class HipJointTorqueRegressorCalculator implements AutoCloseable
{
   public static final int PARAMETERS_PER_BODY = 10;

   private final MultiBodySystemReadOnly input;
   private final HipMultiBodyModel model;
   private final DMatrixRMaj parameterVector;
   private final DMatrixRMaj jointTorqueRegressorMatrix = new DMatrixRMaj(0, 0);
   private final double[] gravity = new double[3];
   private boolean considerCoriolisAndCentrifugalForces = true, considerJointAccelerations = true;
   private boolean firstMomentColumns = false;

   public HipJointTorqueRegressorCalculator(MultiBodySystemReadOnly input)
   {
      this.input = input;
      model = new HipMultiBodyModel(input, false);
      // java:130, 337-348, 877-889: read once, at construction
      parameterVector = new DMatrixRMaj(PARAMETERS_PER_BODY * model.numberOfJoints, 1);
      int i = 0;
      for (JointReadOnly joint : input.getJointsToConsider())
      {
         SpatialInertiaReadOnly inertia = joint.getSuccessor().getInertia();
         int o = PARAMETERS_PER_BODY * i++;
         parameterVector.set(o, 0, inertia.getMass());
         parameterVector.set(o + 1, 0, inertia.getCenterOfMassOffset().getX());
         parameterVector.set(o + 2, 0, inertia.getCenterOfMassOffset().getY());
         parameterVector.set(o + 3, 0, inertia.getCenterOfMassOffset().getZ());
         parameterVector.set(o + 4, 0, inertia.getMomentOfInertia().getM00());
         parameterVector.set(o + 5, 0, inertia.getMomentOfInertia().getM01());
         parameterVector.set(o + 6, 0, inertia.getMomentOfInertia().getM02());
         parameterVector.set(o + 7, 0, inertia.getMomentOfInertia().getM11());
         parameterVector.set(o + 8, 0, inertia.getMomentOfInertia().getM12());
         parameterVector.set(o + 9, 0, inertia.getMomentOfInertia().getM22());
      }
   }

   /** java:360-363 */
   public void setGravitationalAcceleration(double gravity)
   {
      setGravitationalAcceleration(0.0, 0.0, gravity);
   }

   /** java:379-382 */
   public void setGravitationalAcceleration(double gravityX, double gravityY, double gravityZ)
   {
      gravity[0] = gravityX;
      gravity[1] = gravityY;
      gravity[2] = gravityZ;
   }

   /** java:489-492 */
   public void setConsiderJointAccelerations(boolean considerJointAccelerations)
   {
      this.considerJointAccelerations = considerJointAccelerations;
   }

   /** java:499-502 */
   public void setConsiderCoriolisAndCentrifugalForces(boolean considerCoriolisAndCentrifugalForces)
   {
      this.considerCoriolisAndCentrifugalForces = considerCoriolisAndCentrifugalForces;
   }

   /**
    * false (default): columns 1..3 of every body are the reference's (its MCOM bases sit on a body of zero mass: zero, or e x a once the
    * Coriolis terms are switched off -- include/mecano_hip.h).  true: d tau / d (m c), for identification; getParameterVector() keeps the
    * reference's content (the centre-of-mass offset, not the first moment).
    */
   public void setFirstMomentColumns(boolean firstMomentColumns)
   {
      this.firstMomentColumns = firstMomentColumns;
   }

   /** java:173-190 for B configurations.  q: B x nq, qd and qdd: B x nv. */
   public void compute(DMatrixRMaj q, DMatrixRMaj qd, DMatrixRMaj qdd)
   {
      int B = q.getNumRows(), nv = model.nv, columns = PARAMETERS_PER_BODY * model.numberOfJoints;
      if (q.getNumCols() != model.nq || qd.getNumCols() != nv || qd.getNumRows() != B || qdd.getNumCols() != nv || qdd.getNumRows() != B)
         throw new org.ejml.MatrixDimensionException("Expected q: B x " + model.nq + ", qd and qdd: B x " + nv);
      jointTorqueRegressorMatrix.reshape(B, nv * columns);
      MemorySegment dq = deviceCopy(q), dqd = deviceCopy(qd), dqdd = deviceCopy(qdd), dY = deviceAllocate((long) B * nv * columns);
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment g = arena.allocateFrom(JAVA_DOUBLE, gravity);
         MemorySegment options = MecanoHipNative.options(arena, considerCoriolisAndCentrifugalForces, considerJointAccelerations);
         int firstMoments = firstMomentColumns ? 1 : 0;
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.REGRESSOR.invokeExact(model.handle, (long) B, dq, dqd, dqdd, g, options, firstMoments, dY));
         long count = (long) B * nv * columns;
         MemorySegment host = arena.allocate(JAVA_DOUBLE, Math.max(1L, count));
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.COPY_TO_HOST.invokeExact(host, dY, count * Double.BYTES, MemorySegment.NULL));
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.STREAM_SYNCHRONIZE.invokeExact(MemorySegment.NULL));
         MemorySegment.copy(host, JAVA_DOUBLE, 0, jointTorqueRegressorMatrix.data, 0, (int) count);
      }
      finally
      {
         for (MemorySegment buffer : new MemorySegment[] {dq, dqd, dqdd, dY})
            MecanoHipNative.invoke(() -> (int) MecanoHipNative.DEVICE_FREE.invokeExact(buffer));
      }
   }

   private static MemorySegment deviceAllocate(long doubles)
   {
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment out = arena.allocate(ADDRESS);
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.DEVICE_ALLOC.invokeExact(Math.max(1L, doubles) * Double.BYTES, out));
         return out.get(ADDRESS, 0);
      }
   }

   private static MemorySegment deviceCopy(DMatrixRMaj matrix)
   {
      MemorySegment device = deviceAllocate(matrix.getNumElements());
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment host = arena.allocateFrom(JAVA_DOUBLE, matrix.data);
         long bytes = (long) matrix.getNumElements() * Double.BYTES;
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.COPY_TO_DEVICE.invokeExact(device, host, bytes, MemorySegment.NULL));
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.STREAM_SYNCHRONIZE.invokeExact(MemorySegment.NULL));
      }
      return device;
   }

   /** java:450-453; row b = the nv x 10 N matrix of configuration b, row-major. */
   public DMatrixRMaj getJointTorqueRegressorMatrix()
   {
      return jointTorqueRegressorMatrix;
   }

   /** java:462-465 for configuration b: the nv x 10 block of {@code body}. */
   public void getJointTorqueRegressorMatrixBlock(RigidBodyReadOnly body, int b, DMatrixRMaj blockToPack)
   {
      int nv = model.nv, columns = PARAMETERS_PER_BODY * model.numberOfJoints, start = PARAMETERS_PER_BODY * indexOf(body);
      blockToPack.reshape(nv, PARAMETERS_PER_BODY);
      for (int row = 0; row < nv; row++)
         for (int k = 0; k < PARAMETERS_PER_BODY; k++)
            blockToPack.set(row, k, jointTorqueRegressorMatrix.get(b, row * columns + start + k));
   }

   /** java:397-400 (bodies in joints-to-consider order) */
   public DMatrixRMaj getParameterVector()
   {
      return parameterVector;
   }

   /** java:415-421 */
   public DMatrixRMaj getParameterVectorSlice(RigidBodyReadOnly body)
   {
      DMatrixRMaj slice = new DMatrixRMaj(PARAMETERS_PER_BODY, 1);
      System.arraycopy(parameterVector.data, PARAMETERS_PER_BODY * indexOf(body), slice.data, 0, PARAMETERS_PER_BODY);
      return slice;
   }

   private int indexOf(RigidBodyReadOnly body)
   {
      int index = body.getParentJoint() == null ? -1 : model.indexOf(body.getParentJoint());
      if (index < 0)
         throw new IllegalArgumentException("The body is not the successor of a joint this calculator considers: " + body.getName());
      return index;
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
