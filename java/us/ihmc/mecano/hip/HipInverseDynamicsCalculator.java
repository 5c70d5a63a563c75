package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import org.ejml.data.DMatrixRMaj;

import us.ihmc.euclid.tuple3D.interfaces.Tuple3DReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.MultiBodySystemReadOnly;

import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;

/**
 * Batched drop-in for InverseDynamicsCalculator (RNEA): same constructor argument, same setters, compute(...) over B stacked
 * configurations.  Matrices are DMatrixRMaj with one ROW per configuration: q is B x nq, qd / qdd / tau are B x nv, i.e. the column
 * vectors Mecano uses, transposed and stacked -- exactly the [B][n] layout of the C-ABI, so the backing arrays are passed as they are.
 * (HipForwardDynamicsCalculator and HipCompositeRigidBodyMassMatrixCalculator follow the same pattern with mh_aba_f64_host / mh_crba_f64_host.)
 */
public class HipInverseDynamicsCalculator implements AutoCloseable
{
   private final MultiBodySystemReadOnly input;
   private final HipMultiBodyModel model;
   private final double[] gravity = new double[3];
   private final DMatrixRMaj jointTauMatrix = new DMatrixRMaj(0, 0);

   public HipInverseDynamicsCalculator(MultiBodySystemReadOnly input)
   {
      this.input = input;
      model = new HipMultiBodyModel(input);
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
   }

   /** tau = ID(q, qd, qdd) for every row; q: B x nq, qd and qdd: B x nv. */
   public void compute(DMatrixRMaj q, DMatrixRMaj qd, DMatrixRMaj qdd)
   {
      int B = q.getNumRows();
      if (q.getNumCols() != model.nq || qd.getNumCols() != model.nv || qdd.getNumCols() != model.nv || qd.getNumRows() != B || qdd.getNumRows() != B)
         throw new org.ejml.MatrixDimensionException("Expected q: B x " + model.nq + ", qd and qdd: B x " + model.nv);
      jointTauMatrix.reshape(B, model.nv);
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment qSeg = arena.allocateFrom(JAVA_DOUBLE, q.data), qdSeg = arena.allocateFrom(JAVA_DOUBLE, qd.data),
               qddSeg = arena.allocateFrom(JAVA_DOUBLE, qdd.data), g = arena.allocateFrom(JAVA_DOUBLE, gravity);
         MemorySegment tau = arena.allocate(JAVA_DOUBLE, (long) B * model.nv);
         MecanoHipNative.check((int) MecanoHipNative.RNEA_HOST.invokeExact(model.handle, (long) B, qSeg, qdSeg, qddSeg, g, MemorySegment.NULL,
                                                                          MemorySegment.NULL, tau));
         MemorySegment.copy(tau, JAVA_DOUBLE, 0, jointTauMatrix.data, 0, B * model.nv);
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

   /** B x nv; row b, column getJointDoFIndices(joint)[k] is the effort of that DoF for configuration b (InverseDynamicsCalculator.getJointTauMatrix). */
   public DMatrixRMaj getJointTauMatrix()
   {
      return jointTauMatrix;
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
