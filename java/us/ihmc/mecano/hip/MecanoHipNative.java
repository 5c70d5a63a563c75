package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.SymbolLookup;
import java.lang.invoke.MethodHandle;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * Panama (java.lang.foreign, JDK 22+) binding of include/mecano_hip.h.  NOT compiled in this repository's image (no JVM there);
 * it is the reference-side stub a Mecano maintainer adds.  One downcall handle per C entry point; every entry point returns an
 * mh_status int which {@link #check(int)} maps back to the exception types Mecano itself throws.
 */
public final class MecanoHipNative
{
   private static final Linker LINKER = Linker.nativeLinker();
   private static final SymbolLookup LIB = SymbolLookup.libraryLookup(System.getProperty("mecano.hip.library", "libmecano_hip.so"), Arena.global());

   private static MethodHandle handle(String name, FunctionDescriptor descriptor)
   {
      return LINKER.downcallHandle(LIB.find(name).orElseThrow(() -> new UnsatisfiedLinkError(name)), descriptor);
   }

   static final MethodHandle LAST_ERROR = handle("mh_last_error", FunctionDescriptor.of(ADDRESS));
   static final MethodHandle MODEL_CREATE = handle("mh_model_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));
   static final MethodHandle MODEL_DESTROY = handle("mh_model_destroy", FunctionDescriptor.ofVoid(ADDRESS));
   static final MethodHandle RESERVE = handle("mh_reserve", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG));
   /* (model, B, q, qd, qdd|tau, gravity[3], f_ext|NULL, opts|NULL, out) -- host pointers: copies in, launches, copies out, synchronises */
   private static final FunctionDescriptor DYNAMICS = FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS,
                                                                            ADDRESS);
   static final MethodHandle RNEA_HOST = handle("mh_rnea_f64_host", DYNAMICS);
   static final MethodHandle ABA_HOST = handle("mh_aba_f64_host", DYNAMICS);
   static final MethodHandle CRBA_HOST = handle("mh_crba_f64_host", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS));
   /** CompositeRigidBodyMassMatrixCalculator with setEnableCoriolisMatrixCalculation(true): (model, B, q, qd, opts, H_out, C_out). */
   static final MethodHandle CRBA_CORIOLIS_HOST = handle("mh_crba_coriolis_f64_host",
                                                         FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
   /**
    * getCentroidalMomentumMatrix / getCentroidalConvectiveTermMatrix: (model, B, q, qd, frame[12] or NULL, frame_mode, opts, A_out, b_out,
    * com_out); frame = centroidalMomentumFrame.getTransformToDesiredFrame(rootBody.getBodyFixedFrame()) as R row-major + p, frame_mode 1
    * for a CenterOfMassReferenceFrame under it.
    */
   static final MethodHandle CENTROIDAL_HOST = handle("mh_centroidal_f64_host", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS,
                                                                                                       ADDRESS, JAVA_INT, ADDRESS, ADDRESS, ADDRESS,
                                                                                                       ADDRESS));

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

   private MecanoHipNative()
   {
   }
}
