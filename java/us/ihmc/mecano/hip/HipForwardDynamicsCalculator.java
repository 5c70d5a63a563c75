package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import org.ejml.data.DMatrixRMaj;

import us.ihmc.euclid.tuple3D.interfaces.Tuple3DReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.MultiBodySystemReadOnly;

import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;

/**
 * Batched drop-in for ForwardDynamicsCalculator (ABA): same constructor argument, same gravity setters, compute(q, qd, tau) over B stacked
 * configurations (one ROW per configuration, see HipInverseDynamicsCalculator).  Source only: this image has no JDK (INTEGRATION.md).
 */
public class HipForwardDynamicsCalculator implements AutoCloseable
{
   private final MultiBodySystemReadOnly input;
   private final HipMultiBodyModel model;
   private final double[] gravity = new double[3];
   private final DMatrixRMaj jointAccelerationMatrix = new DMatrixRMaj(0, 0);

   public HipForwardDynamicsCalculator(MultiBodySystemReadOnly input)
   {
      this.input = input;
      model = new HipMultiBodyModel(input);
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
   }

   /** qdd = FD(q, qd, tau) for every row; q: B x nq, qd and tau: B x nv (ForwardDynamicsCalculator.compute(DMatrix), java:508-520). */
   public void compute(DMatrixRMaj q, DMatrixRMaj qd, DMatrixRMaj tau)
   {
      int B = q.getNumRows();
      if (q.getNumCols() != model.nq || qd.getNumCols() != model.nv || tau.getNumCols() != model.nv || qd.getNumRows() != B || tau.getNumRows() != B)
         throw new org.ejml.MatrixDimensionException("Expected q: B x " + model.nq + ", qd and tau: B x " + model.nv);
      jointAccelerationMatrix.reshape(B, model.nv);
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment qSeg = arena.allocateFrom(JAVA_DOUBLE, q.data), qdSeg = arena.allocateFrom(JAVA_DOUBLE, qd.data),
               tauSeg = arena.allocateFrom(JAVA_DOUBLE, tau.data), g = arena.allocateFrom(JAVA_DOUBLE, gravity);
         MemorySegment qdd = arena.allocate(JAVA_DOUBLE, (long) B * model.nv);
         MecanoHipNative.check((int) MecanoHipNative.ABA_HOST.invokeExact(model.handle, (long) B, qSeg, qdSeg, tauSeg, g, MemorySegment.NULL,
                                                                         MemorySegment.NULL, qdd));
         MemorySegment.copy(qdd, JAVA_DOUBLE, 0, jointAccelerationMatrix.data, 0, B * model.nv);
      }
      catch (RuntimeException | Error e)
      {
         throw e;
      }
      catch (Throwable t)
      {
         throw new IllegalStateException(t);
      }
   }

   /** B x nv (ForwardDynamicsCalculator.getJointAccelerationMatrix, java:556-567). */
   public DMatrixRMaj getJointAccelerationMatrix()
   {
      return jointAccelerationMatrix;
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
