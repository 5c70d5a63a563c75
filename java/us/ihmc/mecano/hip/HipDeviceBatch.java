package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import org.ejml.data.DMatrixRMaj;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;

/**
 * State of B configurations RESIDENT ON THE DEVICE: q (B x nq), qd, qdd, tau (B x nv), optional external wrenches (B x 6 n) and the
 * per-body / per-joint outputs (B x 6 n).  The calculators' device-side calls read and write these buffers, so a simulation loop
 * (forward dynamics + integration per step, mh_aba_integrate_f64) or a controller that chains inverse dynamics, mass matrix and forward
 * dynamics never crosses PCIe between calls; upload(...) / download(...) move a matrix when the host wants to see it.
 * Row b of every matrix is the column vector Mecano's MultiBodySystemTools.extractJointsState produces for configuration b.
 * <p>
 * A batch owns an mh_context: the calculators' device-side calls on it name that context in their options, so every batch brings its own
 * workspace, scratch and hand-off flags and two threads may drive two batches of one shared (read-only) HipMultiBodyModel at the same
 * time -- the restriction of Mecano's calculators (scratch fields, one calculator per thread: InverseDynamicsCalculator.java:706-707)
 * does not carry over.  {@link #check()} synchronises and surfaces failures of asynchronous calls before their outputs are read.
 */
public final class HipDeviceBatch implements AutoCloseable
{
   final HipMultiBodyModel model;
   final int batchSize;
   final MemorySegment q, qd, qdd, tau, fExt, bodyAcceleration, bodyTwist, jointWrench, pairOutput;
   /** mh_context_t of this batch: pass it as the last argument of MecanoHipNative.options(...) for every call on this batch's buffers */
   final MemorySegment context;
   /** doubles each device buffer holds, in the order of {@link #buffers()}: every copy is checked against it */
   private final long[] capacity;

   public HipDeviceBatch(HipMultiBodyModel model, int batchSize)
   {
      if (batchSize < 0)
         throw new IllegalArgumentException("negative batch size " + batchSize);
      this.model = model;
      this.batchSize = batchSize;
      long B = batchSize, wrenches = B * 6 * model.numberOfJoints;
      capacity = new long[] {B * model.nq, B * model.nv, B * model.nv, B * model.nv, wrenches, wrenches, wrenches, wrenches, B * 6};
      MemorySegment[] made = new MemorySegment[capacity.length];
      MemorySegment madeContext = MemorySegment.NULL;
      try
      {
         for (int i = 0; i < made.length; i++)
            made[i] = allocate(capacity[i]);
         try (Arena arena = Arena.ofConfined())
         {
            MemorySegment out = arena.allocate(ADDRESS);
            MecanoHipNative.invoke(() -> (int) MecanoHipNative.CONTEXT_CREATE.invokeExact(model.handle, out));
            madeContext = out.get(ADDRESS, 0);
         }
         final MemorySegment c = madeContext;
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.CONTEXT_RESERVE.invokeExact(c, (long) batchSize));
      }
      catch (RuntimeException | Error e)
      { // a later allocation failed: the earlier ones must not leak
         for (MemorySegment buffer : made)
            if (buffer != null)
               free(buffer);
         destroyContext(madeContext);
         throw e;
      }
      context = madeContext;
      q = made[0];
      qd = made[1];
      qdd = made[2];
      tau = made[3];
      fExt = made[4];
      bodyAcceleration = made[5];
      bodyTwist = made[6];
      jointWrench = made[7];
      pairOutput = made[8]; // one (base, body) pair of mh_relative_acceleration_f64
   }

   private MemorySegment[] buffers()
   {
      return new MemorySegment[] {q, qd, qdd, tau, fExt, bodyAcceleration, bodyTwist, jointWrench, pairOutput};
   }

   /** doubles the given device buffer of this batch holds; a pointer that is not one of this batch's buffers is refused */
   private long capacityOf(MemorySegment deviceBuffer)
   {
      MemorySegment[] all = buffers();
      for (int i = 0; i < all.length; i++)
         if (all[i].address() == deviceBuffer.address())
            return capacity[i];
      throw new IllegalArgumentException("not a buffer of this batch");
   }

   private static void destroyContext(MemorySegment context)
   {
      if (context.address() == 0)
         return;
      try
      {
         MecanoHipNative.CONTEXT_DESTROY.invokeExact(context);
      }
      catch (Throwable t)
      {
         // best effort on the error path
      }
   }

   /**
    * mh_model_check: waits for the null stream and throws if a call on this batch failed on the device after it had returned (the compute
    * calls with device pointers are asynchronous).  download(...) calls it before it copies, so values never reach the host unchecked.
    */
   public void check()
   {
      MecanoHipNative.invoke(() -> (int) MecanoHipNative.MODEL_CHECK.invokeExact(model.handle, context, MemorySegment.NULL));
   }

   private static void free(MemorySegment buffer)
   {
      try
      {
         int ignored = (int) MecanoHipNative.DEVICE_FREE.invokeExact(buffer);
      }
      catch (Throwable t)
      {
         // freeing is best effort on the error path
      }
   }

   private static MemorySegment allocate(long doubles)
   {
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment out = arena.allocate(ADDRESS);
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.DEVICE_ALLOC.invokeExact(Math.max(1L, doubles) * Double.BYTES, out));
         return out.get(ADDRESS, 0);
      }
   }

   /** host matrix (rows = configurations) -> device buffer */
   public void upload(DMatrixRMaj matrix, MemorySegment deviceBuffer)
   {
      long expected = capacityOf(deviceBuffer);
      if (matrix.getNumElements() != expected) // a larger matrix would be written past the allocation, into neighbouring device buffers
         throw new IllegalArgumentException("matrix has " + matrix.getNumElements() + " elements, the device buffer holds " + expected + " (batch of "
                                            + batchSize + ")");
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment host = arena.allocateFrom(JAVA_DOUBLE, matrix.data);
         long bytes = (long) matrix.getNumElements() * Double.BYTES;
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.COPY_TO_DEVICE.invokeExact(deviceBuffer, host, bytes, MemorySegment.NULL));
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.STREAM_SYNCHRONIZE.invokeExact(MemorySegment.NULL));
      }
   }

   /** device buffer -> host matrix (reshaped to rows x columns) */
   public void download(MemorySegment deviceBuffer, int rows, int columns, DMatrixRMaj matrixToPack)
   {
      if ((long) rows * columns > capacityOf(deviceBuffer))
         throw new IllegalArgumentException(rows + " x " + columns + " exceeds the " + capacityOf(deviceBuffer) + " doubles of the device buffer");
      matrixToPack.reshape(rows, columns);
      check();
      try (Arena arena = Arena.ofConfined())
      {
         long count = (long) rows * columns;
         MemorySegment host = arena.allocate(JAVA_DOUBLE, Math.max(1L, count));
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.COPY_TO_HOST.invokeExact(host, deviceBuffer, count * Double.BYTES, MemorySegment.NULL));
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.STREAM_SYNCHRONIZE.invokeExact(MemorySegment.NULL));
         MemorySegment.copy(host, JAVA_DOUBLE, 0, matrixToPack.data, 0, (int) count);
      }
   }

   public void setConfiguration(DMatrixRMaj q)
   {
      upload(q, this.q);
   }

   public void setVelocity(DMatrixRMaj qd)
   {
      upload(qd, this.qd);
   }

   public void setAcceleration(DMatrixRMaj qdd)
   {
      upload(qdd, this.qdd);
   }

   public void setEffort(DMatrixRMaj tau)
   {
      upload(tau, this.tau);
   }

   /** B x 6 n: (moment, force) on the successor body of every listed joint, in that body's frame. */
   public void setExternalWrenches(DMatrixRMaj wrenches)
   {
      upload(wrenches, fExt);
   }

   public void getConfiguration(DMatrixRMaj qToPack)
   {
      download(q, batchSize, model.nq, qToPack);
   }

   public void getVelocity(DMatrixRMaj qdToPack)
   {
      download(qd, batchSize, model.nv, qdToPack);
   }

   public void getAcceleration(DMatrixRMaj qddToPack)
   {
      download(qdd, batchSize, model.nv, qddToPack);
   }

   public void getEffort(DMatrixRMaj tauToPack)
   {
      download(tau, batchSize, model.nv, tauToPack);
   }

   @Override
   public void close()
   {
      for (MemorySegment buffer : buffers())
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.DEVICE_FREE.invokeExact(buffer));
      destroyContext(context);
   }
}
