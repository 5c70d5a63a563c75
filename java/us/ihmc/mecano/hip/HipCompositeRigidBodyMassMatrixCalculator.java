package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import org.ejml.data.DMatrixRMaj;

import us.ihmc.mecano.multiBodySystem.interfaces.MultiBodySystemReadOnly;

import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;

/**
 * Batched drop-in for CompositeRigidBodyMassMatrixCalculator: mass matrix, Coriolis matrix (setEnableCoriolisMatrixCalculation), centroidal
 * momentum matrix and convective term.  compute(q, qd) takes B stacked configurations (one ROW per configuration); the getters return B
 * stacked row-major matrices: getMassMatrix() is B x (nv * nv), row b = the reference's nv x nv matrix of configuration b, etc.  The
 * centroidal momentum frame is the root body frame, optionally re-centred on the centre of mass (a CenterOfMassReferenceFrame whose parent
 * is the root frame).  Source only: this image has no JDK (INTEGRATION.md).
 */
public class HipCompositeRigidBodyMassMatrixCalculator implements AutoCloseable
{
   private final MultiBodySystemReadOnly input;
   private final HipMultiBodyModel model;
   private boolean enableCoriolisMatrixCalculation = false;
   private boolean centroidalFrameAtCenterOfMass = false;
   private final DMatrixRMaj massMatrix = new DMatrixRMaj(0, 0), coriolisMatrix = new DMatrixRMaj(0, 0);
   private final DMatrixRMaj centroidalMomentumMatrix = new DMatrixRMaj(0, 0), centroidalConvectiveTermMatrix = new DMatrixRMaj(0, 0);

   public HipCompositeRigidBodyMassMatrixCalculator(MultiBodySystemReadOnly input)
   {
      this(input, true);
   }

   /** CompositeRigidBodyMassMatrixCalculator(MultiBodySystemReadOnly, ReferenceFrame, boolean considerIgnoredSubtreesInertia) (java:177-200). */
   public HipCompositeRigidBodyMassMatrixCalculator(MultiBodySystemReadOnly input, boolean considerIgnoredSubtreesInertia)
   {
      this.input = input;
      model = new HipMultiBodyModel(input, considerIgnoredSubtreesInertia);
   }

   /** CompositeRigidBodyMassMatrixCalculator.setEnableCoriolisMatrixCalculation (java:271-274). */
   public void setEnableCoriolisMatrixCalculation(boolean enableCoriolisMatrixCalculation)
   {
      this.enableCoriolisMatrixCalculation = enableCoriolisMatrixCalculation;
   }

   /** true: the centroidal momentum frame is a CenterOfMassReferenceFrame under the root body frame; false: the root body frame (java:190-193). */
   public void setCentroidalMomentumFrameAtCenterOfMass(boolean atCenterOfMass)
   {
      centroidalFrameAtCenterOfMass = atCenterOfMass;
   }

   /** q: B x nq, qd: B x nv. */
   public void compute(DMatrixRMaj q, DMatrixRMaj qd)
   {
      int B = q.getNumRows(), nv = model.nv;
      if (q.getNumCols() != model.nq || qd.getNumCols() != nv || qd.getNumRows() != B)
         throw new org.ejml.MatrixDimensionException("Expected q: B x " + model.nq + ", qd: B x " + nv);
      massMatrix.reshape(B, nv * nv);
      coriolisMatrix.reshape(B, enableCoriolisMatrixCalculation ? nv * nv : 0);
      centroidalMomentumMatrix.reshape(B, 6 * nv);
      centroidalConvectiveTermMatrix.reshape(B, 6);
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment qSeg = arena.allocateFrom(JAVA_DOUBLE, q.data), qdSeg = arena.allocateFrom(JAVA_DOUBLE, qd.data);
         MemorySegment H = arena.allocate(JAVA_DOUBLE, (long) B * nv * nv);
         if (enableCoriolisMatrixCalculation)
         {
            MemorySegment C = arena.allocate(JAVA_DOUBLE, (long) B * nv * nv);
            MecanoHipNative.invoke(() -> (int) MecanoHipNative.CRBA_CORIOLIS_HOST.invokeExact(model.handle, (long) B, qSeg, qdSeg, MemorySegment.NULL, H, C));
            MemorySegment.copy(C, JAVA_DOUBLE, 0, coriolisMatrix.data, 0, B * nv * nv);
         }
         else
            MecanoHipNative.invoke(() -> (int) MecanoHipNative.CRBA_HOST.invokeExact(model.handle, (long) B, qSeg, MemorySegment.NULL, H));
         MemorySegment.copy(H, JAVA_DOUBLE, 0, massMatrix.data, 0, B * nv * nv);
         MemorySegment A = arena.allocate(JAVA_DOUBLE, (long) B * 6 * nv), b = arena.allocate(JAVA_DOUBLE, (long) B * 6);
         int frameMode = centroidalFrameAtCenterOfMass ? 1 : 0;
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.CENTROIDAL_HOST.invokeExact(model.handle, (long) B, qSeg, qdSeg, MemorySegment.NULL, frameMode,
                                                                                       MemorySegment.NULL, A, b, MemorySegment.NULL));
         MemorySegment.copy(A, JAVA_DOUBLE, 0, centroidalMomentumMatrix.data, 0, B * 6 * nv);
         MemorySegment.copy(b, JAVA_DOUBLE, 0, centroidalConvectiveTermMatrix.data, 0, B * 6);
      }
   }

   /** java:344-348 */
   public DMatrixRMaj getMassMatrix()
   {
      return massMatrix;
   }

   /** java:352-365: UnsupportedOperationException while the calculation is disabled. */
   public DMatrixRMaj getCoriolisMatrix()
   {
      if (!enableCoriolisMatrixCalculation)
         throw new UnsupportedOperationException("Coriolis matrix calculation is disabled.");
      return coriolisMatrix;
   }

   /** java:386-398 */
   public DMatrixRMaj getCentroidalMomentumMatrix()
   {
      return centroidalMomentumMatrix;
   }

   /** java:413-420 */
   public DMatrixRMaj getCentroidalConvectiveTermMatrix()
   {
      return centroidalConvectiveTermMatrix;
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
