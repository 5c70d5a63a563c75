package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemoryLayout;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.StructLayout;
import java.lang.foreign.SymbolLookup;
import java.lang.invoke.MethodHandle;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * Panama (java.lang.foreign, JDK 22+) binding of include/mecano_hip.h, ABI version 5.  NOT compiled in this repository's image (no JVM
 * there); it is the reference-side stub a Mecano maintainer adds.  One downcall handle per C entry point the shim classes use; every
 * entry point returns an mh_status int which {@link #check(int)} maps back to the exception types Mecano itself throws.
 * <p>
 * Pointer arguments named "host" are heap / off-heap memory of the JVM; "device" pointers come from {@link #DEVICE_ALLOC}.
 */
public final class MecanoHipNative
{
   private static final Linker LINKER = Linker.nativeLinker();
   private static final SymbolLookup LIB = SymbolLookup.libraryLookup(System.getProperty("mecano.hip.library", "libmecano_hip.so"), Arena.global());

   private static MethodHandle handle(String name, FunctionDescriptor descriptor)
   {
      return LINKER.downcallHandle(LIB.find(name).orElseThrow(() -> new UnsatisfiedLinkError(name)), descriptor);
   }

   private static FunctionDescriptor status(MemoryLayout... arguments)
   {
      return FunctionDescriptor.of(JAVA_INT, arguments);
   }

   /** the ABI version this binding was written against (include/mecano_hip.h: MH_ABI_VERSION); checked when the class loads */
   static final int ABI = 5;

   /**
    * struct mh_options { int32 consider_coriolis, consider_accelerations, layout, use_root_acceleration; void *stream; double
    * root_acceleration[6]; void *context; }
    */
   static final StructLayout OPTIONS = MemoryLayout.structLayout(JAVA_INT.withName("consider_coriolis"), JAVA_INT.withName("consider_accelerations"),
                                                                JAVA_INT.withName("layout"), JAVA_INT.withName("use_root_acceleration"),
                                                                ADDRESS.withName("stream"), MemoryLayout.sequenceLayout(6, JAVA_DOUBLE).withName("root_acceleration"),
                                                                ADDRESS.withName("context"));

   static final MethodHandle ABI_VERSION = handle("mh_abi_version", FunctionDescriptor.of(JAVA_INT));
   static final MethodHandle LAST_ERROR = handle("mh_last_error", FunctionDescriptor.of(ADDRESS));
   static final MethodHandle MODEL_CREATE = handle("mh_model_create", status(ADDRESS, ADDRESS));
   static final MethodHandle MODEL_DESTROY = handle("mh_model_destroy", FunctionDescriptor.ofVoid(ADDRESS));
   static final MethodHandle MODEL_KERNEL_VARIANT = handle("mh_model_kernel_variant", FunctionDescriptor.of(ADDRESS, ADDRESS));
   /** MH_WARN_* bits: model classes in which the engine consciously departs from Mecano (include/mecano_hip.h). */
   static final MethodHandle MODEL_WARNINGS = handle("mh_model_warnings", FunctionDescriptor.of(JAVA_INT, ADDRESS));
   static final MethodHandle MODEL_WARNING_TEXT = handle("mh_model_warning_text", FunctionDescriptor.of(ADDRESS, ADDRESS));
   static final int WARN_NEAR_COORDINATE_AXIS = 1, WARN_TINY_COMPOSITE_MASS = 2;
   /** (desc, out_dir|NULL, path_out, path_cap): runs hipcc on the kernel sources next to the library; minutes; once per robot. */
   static final MethodHandle BUILD_CODE_OBJECT = handle("mh_build_code_object", status(ADDRESS, ADDRESS, ADDRESS, JAVA_LONG));
   static final MethodHandle RESERVE = handle("mh_reserve", status(ADDRESS, JAVA_LONG));
   /**
    * Contexts: the model handle is read-only and shared; what compute calls write besides their outputs (workspace, scratch, hand-off flags,
    * error word) belongs to a context, one per thread / stream, named in mh_options.context (HipDeviceBatch owns one).
    */
   static final MethodHandle CONTEXT_CREATE = handle("mh_context_create", status(ADDRESS, ADDRESS));
   static final MethodHandle CONTEXT_DESTROY = handle("mh_context_destroy", FunctionDescriptor.ofVoid(ADDRESS));
   static final MethodHandle CONTEXT_RESERVE = handle("mh_context_reserve", status(ADDRESS, JAVA_LONG));
   /** (model, context|NULL, stream|NULL): synchronises the stream and reports failures of asynchronous calls that only showed on the device */
   static final MethodHandle MODEL_CHECK = handle("mh_model_check", status(ADDRESS, ADDRESS, ADDRESS));
   static final MethodHandle SET_JOINT_SOURCE_MODES = handle("mh_model_set_joint_source_modes", status(ADDRESS, ADDRESS));

   /* (model, B, q, qd, qdd|tau, gravity[3] (host), f_ext|NULL, opts|NULL, out) */
   private static final FunctionDescriptor DYNAMICS = status(ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS);
   /* (model, B, q, qd, qdd|tau, gravity, f_ext|NULL, opts|NULL, out, out2, out3): bodies (tau|qdd, body_acc, body_twist) */
   private static final FunctionDescriptor DYNAMICS_3 = status(ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS,
                                                               ADDRESS);
   /* (model, B, q, qd, qdd|tau, gravity, f_ext|NULL, opts|NULL, out, joint_wrench_out) */
   private static final FunctionDescriptor DYNAMICS_2 = status(ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS);
   /* (model, B, q, qd, qdd, tau, gravity, f_ext|NULL, opts|NULL, tau_out, qdd_out) */
   private static final FunctionDescriptor PAIR = status(ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS);

   // ---- host-pointer entry points: copy in (chunked, three streams), launch, copy out, synchronise
   static final MethodHandle RNEA_HOST = handle("mh_rnea_f64_host", DYNAMICS);
   static final MethodHandle ABA_HOST = handle("mh_aba_f64_host", DYNAMICS);
   static final MethodHandle RNEA_ABA_HOST = handle("mh_rnea_aba_f64_host", PAIR);
   static final MethodHandle CRBA_HOST = handle("mh_crba_f64_host", status(ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS));
   /** CompositeRigidBodyMassMatrixCalculator with setEnableCoriolisMatrixCalculation(true): (model, B, q, qd, opts, H_out, C_out). */
   static final MethodHandle CRBA_CORIOLIS_HOST = handle("mh_crba_coriolis_f64_host", status(ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
   /**
    * getCentroidalMomentumMatrix / getCentroidalConvectiveTermMatrix: (model, B, q, qd, frame[12] or NULL, frame_mode, opts, A_out, b_out,
    * com_out); frame = centroidalMomentumFrame.getTransformToDesiredFrame(rootBody.getBodyFixedFrame()) as R row-major + p, frame_mode 1
    * for a CenterOfMassReferenceFrame under it.
    */
   static final MethodHandle CENTROIDAL_HOST = handle("mh_centroidal_f64_host", status(ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS,
                                                                                       ADDRESS, ADDRESS, ADDRESS));

   // ---- device-pointer entry points (asynchronous on opts->stream): what HipDeviceBatch drives
   static final MethodHandle RNEA = handle("mh_rnea_f64", DYNAMICS);
   static final MethodHandle ABA = handle("mh_aba_f64", DYNAMICS);
   static final MethodHandle RNEA_ABA = handle("mh_rnea_aba_f64", PAIR);
   /** the same in fp32 (float matrices on the device) */
   static final MethodHandle RNEA_ABA_F32 = handle("mh_rnea_aba_f32", PAIR);
   /** (model, B, q, qd, qdd, gravity, f_ext|NULL, opts|NULL, tau_out, H_out): inverse dynamics and the mass matrix of the same state, one launch */
   static final MethodHandle RNEA_CRBA = handle("mh_rnea_crba_f64", DYNAMICS_2);
   /** (model, B, q, qd, qdd, gravity, opts|NULL, first_moment_columns, Y_out): JointTorqueRegressorCalculator.compute for B configurations */
   static final MethodHandle REGRESSOR = handle("mh_regressor_f64", status(ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
   /** (model, B, q, qd, tau, qdd_in, gravity, f_ext, opts, qdd_out, tau_out) */
   static final MethodHandle ABA_LOCKED = handle("mh_aba_locked_f64", PAIR);
   static final MethodHandle RNEA_BODIES = handle("mh_rnea_bodies_f64", DYNAMICS_3);
   static final MethodHandle ABA_BODIES = handle("mh_aba_bodies_f64", DYNAMICS_3);
   static final MethodHandle RNEA_JOINT_WRENCHES = handle("mh_rnea_joint_wrenches_f64", DYNAMICS_2);
   static final MethodHandle ABA_JOINT_WRENCHES = handle("mh_aba_joint_wrenches_f64", DYNAMICS_2);
   /** (model, B, q, body_acc, body_twist|NULL, gravity, n_pairs, base_joints (host int[]), body_joints (host int[]), opts, out) */
   static final MethodHandle RELATIVE_ACCELERATION = handle("mh_relative_acceleration_f64", status(ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS,
                                                                                                   JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
   /** (model, B, dt, q, qd, qdd, opts, q_out, qd_out, qdd_out|NULL) */
   static final MethodHandle INTEGRATE = handle("mh_integrate_f64", status(ADDRESS, JAVA_LONG, JAVA_DOUBLE, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS,
                                                                           ADDRESS));
   /** one simulation step: (model, B, dt, q, qd, tau, gravity, f_ext, opts, qdd_out, q_next, qd_next); q_next / qd_next may alias q / qd */
   static final MethodHandle ABA_INTEGRATE = handle("mh_aba_integrate_f64", status(ADDRESS, JAVA_LONG, JAVA_DOUBLE, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS,
                                                                                   ADDRESS, ADDRESS, ADDRESS, ADDRESS));

   // ---- memory: pinned host memory for PCIe-rate copies, device memory for state that stays resident
   static final MethodHandle HOST_ALLOC = handle("mh_host_alloc", status(JAVA_LONG, ADDRESS));
   static final MethodHandle HOST_FREE = handle("mh_host_free", status(ADDRESS));
   static final MethodHandle DEVICE_ALLOC = handle("mh_device_alloc", status(JAVA_LONG, ADDRESS));
   static final MethodHandle DEVICE_FREE = handle("mh_device_free", status(ADDRESS));
   static final MethodHandle COPY_TO_DEVICE = handle("mh_copy_to_device", status(ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
   static final MethodHandle COPY_TO_HOST = handle("mh_copy_to_host", status(ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
   static final MethodHandle STREAM_SYNCHRONIZE = handle("mh_stream_synchronize", status(ADDRESS));

   // ---- multi-GPU: one JVM per GPU, the batch sharded by rows, RCCL over xGMI (HipCommunicator)
   static final int COMM_ID_BYTES = 128;
   static final MethodHandle SET_DEVICE = handle("mh_set_device", status(JAVA_INT));
   static final MethodHandle SHARD_RANGE = handle("mh_shard_range", status(JAVA_LONG, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
   static final MethodHandle COMM_UNIQUE_ID = handle("mh_comm_unique_id", status(ADDRESS));
   static final MethodHandle COMM_CREATE = handle("mh_comm_create", status(ADDRESS, JAVA_INT, JAVA_INT, ADDRESS));
   static final MethodHandle COMM_DESTROY = handle("mh_comm_destroy", status(ADDRESS));
   static final MethodHandle COMM_SIZE = handle("mh_comm_size", status(ADDRESS, ADDRESS, ADDRESS));
   static final MethodHandle COMM_BROADCAST = handle("mh_comm_broadcast", status(ADDRESS, ADDRESS, JAVA_LONG, JAVA_INT, ADDRESS));
   static final MethodHandle COMM_BROADCAST_HOST = handle("mh_comm_broadcast_host", status(ADDRESS, ADDRESS, JAVA_LONG, JAVA_INT));
   static final MethodHandle COMM_ALL_GATHER_ROWS = handle("mh_comm_all_gather_rows", status(ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, ADDRESS));
   static final MethodHandle COMM_BARRIER = handle("mh_comm_barrier", status(ADDRESS, ADDRESS));

   /** An mh_options in `arena`: the calculators' switches, AoS layout (rows = configurations), the null stream, gravity as the root acceleration. */
   static MemorySegment options(Arena arena, boolean considerCoriolis, boolean considerAccelerations)
   {
      return options(arena, considerCoriolis, considerAccelerations, null);
   }

   /**
    * The same with an explicit root acceleration (setRootAcceleration(SpatialAccelerationReadOnly), InverseDynamicsCalculator.java:413-427):
    * six doubles (angular, linear) in root-body coordinates, or null for "the call's gravity argument as (0, -g)".
    */
   static MemorySegment options(Arena arena, boolean considerCoriolis, boolean considerAccelerations, double[] rootAcceleration)
   {
      return options(arena, considerCoriolis, considerAccelerations, rootAcceleration, MemorySegment.NULL);
   }

   /** The same for calls made through a context (HipDeviceBatch.context); MemorySegment.NULL is the model's default context. */
   static MemorySegment options(Arena arena, boolean considerCoriolis, boolean considerAccelerations, double[] rootAcceleration, MemorySegment context)
   {
      MemorySegment options = arena.allocate(OPTIONS);
      options.set(JAVA_INT, 0, considerCoriolis ? 1 : 0);
      options.set(JAVA_INT, 4, considerAccelerations ? 1 : 0);
      options.set(JAVA_INT, 8, 0);
      options.set(JAVA_INT, 12, rootAcceleration == null ? 0 : 1);
      options.set(ADDRESS, 16, MemorySegment.NULL);
      for (int k = 0; k < 6; k++)
         options.set(JAVA_DOUBLE, 24 + 8L * k, rootAcceleration == null ? 0.0 : rootAcceleration[k]);
      options.set(ADDRESS, 72, context);
      return options;
   }

   static
   {
      int version;
      try
      {
         version = (int) ABI_VERSION.invokeExact();
      }
      catch (Throwable t)
      {
         throw new ExceptionInInitializerError(t);
      }
      if (version != ABI)
         throw new UnsatisfiedLinkError("libmecano_hip.so reports ABI version " + version + ", this binding was written for " + ABI);
   }

   /** mh_status -> the exception Mecano's own calculators would have thrown (SURVEY.md section 8b, "Errors"). */
   static void check(int status)
   {
      if (status == 0)
         return;
      String message;
      try
      {
         message = ((MemorySegment) LAST_ERROR.invokeExact()).reinterpret(512).getString(0);
      }
      catch (Throwable t)
      {
         message = "mh_status " + status;
      }
      switch (status)
      {
         case 1: // MH_ERR_INVALID_ARGUMENT
         case 5: // MH_ERR_BAD_TOPOLOGY
         case 6: // MH_ERR_BAD_AXIS
            throw new IllegalArgumentException(message);
         case 2: // MH_ERR_BAD_DIMENSION  (ForwardDynamicsCalculator.java:522-533)
            throw new org.ejml.MatrixDimensionException(message);
         case 3: // MH_ERR_UNSUPPORTED_JOINT
         case 4: // MH_ERR_LOOP_CLOSURE   (ForwardDynamicsCalculator.java:207-211 prints and skips; here it is explicit)
            throw new UnsupportedOperationException(message);
         case 9: // MH_ERR_OUT_OF_MEMORY
            throw new OutOfMemoryError(message);
         default: // MH_ERR_NO_DEVICE, MH_ERR_HIP, ...
            throw new IllegalStateException(message);
      }
   }

   /** Runs a downcall that returns an mh_status and converts checked Throwables (MethodHandle.invoke) into unchecked ones. */
   interface Call
   {
      int run() throws Throwable;
   }

   static void invoke(Call call)
   {
      int status;
      try
      {
         status = call.run();
      }
      catch (RuntimeException | Error e)
      {
         throw e;
      }
      catch (Throwable t)
      {
         throw new IllegalStateException(t);
      }
      check(status);
   }

   private MecanoHipNative()
   {
   }
}
