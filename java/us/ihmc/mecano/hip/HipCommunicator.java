package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import static java.lang.foreign.ValueLayout.JAVA_BYTE;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * One JVM per GPU, the batch sharded by rows: the three operations a sharded run needs around the calculators, over RCCL (mh_comm_* of
 * include/mecano_hip.h).  Mecano itself has no counterpart -- its calculators are single-threaded objects, one configuration per call
 * (InverseDynamicsCalculatorTest.java:124-158 is the shape of the workload).  Every configuration is independent and the model is
 * read-only, so the calculators' batched calls need no collective: each rank computes its own rows.
 * <pre>
 *   byte[] id = rank == 0 ? HipCommunicator.uniqueId() : receivedFromRankZero;   // 128 bytes, carried by the launcher
 *   MecanoHipNative.invoke(() -> (int) MecanoHipNative.SET_DEVICE.invokeExact(localRank));
 *   try (HipCommunicator comm = new HipCommunicator(id, rank, world))
 *   {
 *      byte[] description = comm.broadcast(rank == 0 ? packedRobotDescription : null, length, 0);
 *      long[] rows = HipCommunicator.shardRange(B, rank, world);                  // [lo, hi)
 *      ... batch of rows[1] - rows[0] configurations, calculators as on one GPU ...
 *      comm.allGatherRows(batch.tauDevice(), B, 8L * nv, allTauDevice);           // optional: every rank sees every row
 *   }
 * </pre>
 */
public final class HipCommunicator implements AutoCloseable
{
   private MemorySegment handle;
   private final int rank, world;

   /** rank 0 makes the id; every other rank receives these 128 bytes from it by the launcher's own means */
   public static byte[] uniqueId()
   {
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment id = arena.allocate(MecanoHipNative.COMM_ID_BYTES);
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.COMM_UNIQUE_ID.invokeExact(id));
         return id.toArray(JAVA_BYTE);
      }
   }

   /** rows [lo, hi) of a batch of B owned by a rank: contiguous, sizes differ by at most one */
   public static long[] shardRange(long B, int rank, int world)
   {
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment lo = arena.allocate(JAVA_LONG), hi = arena.allocate(JAVA_LONG);
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.SHARD_RANGE.invokeExact(B, rank, world, lo, hi));
         return new long[] {lo.get(JAVA_LONG, 0), hi.get(JAVA_LONG, 0)};
      }
   }

   /** collective: every rank calls it with the same id, on the device it computes on (mh_set_device before) */
   public HipCommunicator(byte[] uniqueId, int rank, int world)
   {
      if (uniqueId == null || uniqueId.length != MecanoHipNative.COMM_ID_BYTES)
         throw new IllegalArgumentException("a communicator id has " + MecanoHipNative.COMM_ID_BYTES + " bytes");
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment id = arena.allocateFrom(JAVA_BYTE, uniqueId), out = arena.allocate(java.lang.foreign.ValueLayout.ADDRESS);
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.COMM_CREATE.invokeExact(id, rank, world, out));
         handle = out.get(java.lang.foreign.ValueLayout.ADDRESS, 0);
         MemorySegment r = arena.allocate(JAVA_INT), w = arena.allocate(JAVA_INT);
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.COMM_SIZE.invokeExact(handle, r, w));
         this.rank = r.get(JAVA_INT, 0);
         this.world = w.get(JAVA_INT, 0);
      }
   }

   public int getRank()
   {
      return rank;
   }

   public int getWorldSize()
   {
      return world;
   }

   /** `root` passes the payload (length bytes), the others null; every rank returns the same bytes (the packed robot description) */
   public byte[] broadcast(byte[] payload, int length, int root)
   {
      if (rank == root && (payload == null || payload.length != length))
         throw new IllegalArgumentException("the root passes exactly " + length + " bytes");
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment buffer = arena.allocate(Math.max(length, 1));
         if (rank == root)
            MemorySegment.copy(payload, 0, buffer, JAVA_BYTE, 0, length);
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.COMM_BROADCAST_HOST.invokeExact(handle, buffer, (long) length, root));
         return buffer.asSlice(0, length).toArray(JAVA_BYTE);
      }
   }

   /**
    * localRowsDevice: this rank's rows of shardRange(totalRows, rank, world), [hi - lo][rowBytes] on the device; allRowsDevice:
    * [totalRows][rowBytes] on the device of every rank.  Asynchronous on the null stream: barrier() or a download orders after it.
    */
   public void allGatherRows(MemorySegment localRowsDevice, long totalRows, long rowBytes, MemorySegment allRowsDevice)
   {
      MecanoHipNative.invoke(() -> (int) MecanoHipNative.COMM_ALL_GATHER_ROWS.invokeExact(handle, localRowsDevice, totalRows, rowBytes, allRowsDevice,
                                                                                           MemorySegment.NULL));
   }

   /** every rank has arrived and the null stream of this rank has drained */
   public void barrier()
   {
      MecanoHipNative.invoke(() -> (int) MecanoHipNative.COMM_BARRIER.invokeExact(handle, MemorySegment.NULL));
   }

   @Override
   public void close()
   {
      if (handle != null)
      {
         MemorySegment h = handle;
         handle = null;
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.COMM_DESTROY.invokeExact(h));
      }
   }
}
