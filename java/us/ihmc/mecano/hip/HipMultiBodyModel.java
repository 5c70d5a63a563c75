package us.ihmc.mecano.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.MemoryLayout;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.StructLayout;
import java.util.List;

import us.ihmc.euclid.transform.RigidBodyTransform;
import us.ihmc.euclid.transform.interfaces.RigidBodyTransformReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.FixedJointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.JointMatrixIndexProvider;
import us.ihmc.mecano.multiBodySystem.interfaces.JointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.MultiBodySystemReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.OneDoFJointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.PlanarJointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.PrismaticJointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.RevoluteJointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.RigidBodyReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.SixDoFJointReadOnly;
import us.ihmc.mecano.multiBodySystem.interfaces.SphericalJointReadOnly;
import us.ihmc.mecano.spatial.SpatialInertia;
import us.ihmc.mecano.spatial.interfaces.SpatialInertiaReadOnly;
import us.ihmc.mecano.tools.MultiBodySystemTools;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;
import static java.lang.foreign.ValueLayout.JAVA_INT;

/**
 * Flattens a MultiBodySystemReadOnly into mh_model_desc -- the model-extraction recipe of MultiBodySystemFactories.java:401-470, 782-868
 * (SURVEY.md appendix B) -- and owns the device-side model handle.  Joints are listed in
 * input.getJointMatrixIndexProvider().getIndexedJointsInOrder(); joints to ignore are not listed, and with
 * considerIgnoredSubtreesInertia the inertia of every ignored subtree is lumped into the body it hangs from, with Mecano's own code and at
 * the configuration the system is in when the model is built -- exactly what InverseDynamicsCalculator.java:226-236, 832-860 does
 * (MultiBodySystemTools.computeSubtreeInertia, changeFrame(bodyFixedFrame), SpatialInertia.add).
 */
public final class HipMultiBodyModel implements AutoCloseable
{
   static final StructLayout DESC = MemoryLayout.structLayout(JAVA_INT.withName("n_joints"), JAVA_INT.withName("nq"), JAVA_INT.withName("nv"),
                                                             MemoryLayout.paddingLayout(4), ADDRESS.withName("parent"), ADDRESS.withName("joint_type"),
                                                             ADDRESS.withName("axis"), ADDRESS.withName("X_before"), ADDRESS.withName("X_com"),
                                                             ADDRESS.withName("inertia_J"), ADDRESS.withName("inertia_mass"),
                                                             ADDRESS.withName("inertia_com"), ADDRESS.withName("dof_indices"),
                                                             ADDRESS.withName("cfg_indices"));

   final MemorySegment handle;
   final int numberOfJoints, nq, nv;
   final List<? extends JointReadOnly> jointList;

   public HipMultiBodyModel(MultiBodySystemReadOnly input)
   {
      this(input, true);
   }

   public HipMultiBodyModel(MultiBodySystemReadOnly input, boolean considerIgnoredSubtreesInertia)
   {
      JointMatrixIndexProvider provider = input.getJointMatrixIndexProvider();
      List<? extends JointReadOnly> joints = provider.getIndexedJointsInOrder();
      jointList = joints;
      int n = joints.size();
      int[] parent = new int[n], type = new int[n];
      double[] axis = new double[3 * n], xBefore = new double[12 * n], xCom = new double[12 * n], J = new double[9 * n], mass = new double[n],
            com = new double[3 * n];
      var dof = new java.util.ArrayList<Integer>();
      var cfg = new java.util.ArrayList<Integer>();
      int maxDof = -1, maxCfg = -1;
      for (int i = 0; i < n; i++)
      {
         JointReadOnly joint = joints.get(i);
         if (joint instanceof RevoluteJointReadOnly)
            type[i] = 0;
         else if (joint instanceof PrismaticJointReadOnly)
            type[i] = 1;
         else if (joint instanceof SixDoFJointReadOnly)
            type[i] = 2;
         else if (joint instanceof FixedJointReadOnly)
            type[i] = 3;
         else if (joint instanceof PlanarJointReadOnly)
            type[i] = 4; // q = (pitch, x, z), qd = (w_y, v_x, v_z): PlanarJointReadOnly.java:17-72
         else if (joint instanceof SphericalJointReadOnly)
            type[i] = 5; // q = quaternion (x, y, z, s), qd = angular velocity: SphericalJointReadOnly.java:18-104
         else
            throw new UnsupportedOperationException("Joint kind not supported by the HIP engine: " + joint.getClass().getSimpleName());
         if (joint.isLoopClosure())
            throw new UnsupportedOperationException("Kinematic loops are not supported: " + joint.getName());

         JointReadOnly parentJoint = joint.getPredecessor().getParentJoint();
         parent[i] = parentJoint == null ? -1 : joints.indexOf(parentJoint);
         if (joint instanceof OneDoFJointReadOnly oneDoF)
            oneDoF.getJointAxis().get(3 * i, axis);

         // identity when the frame before the joint IS the parent frame (MecanoFactories.java:81-91, MultiBodySystemFactories.java:763-769)
         RigidBodyTransformReadOnly before = joint.getFrameBeforeJoint() == parentFrameOf(joint) ? new RigidBodyTransform()
                                                                                                : joint.getFrameBeforeJoint().getTransformToParent();
         pack(before, xBefore, 12 * i);
         RigidBodyReadOnly body = joint.getSuccessor();
         pack(body.getBodyFixedFrame().getTransformToParent(), xCom, 12 * i);
         SpatialInertiaReadOnly inertia = body.getInertia();
         if (considerIgnoredSubtreesInertia)
         { // InverseDynamicsCalculator.java:839-855: ignored children of this body, summed in its body-fixed frame
            SpatialInertia lumped = null;
            for (JointReadOnly childJoint : body.getChildrenJoints())
            {
               if (!input.getJointsToIgnore().contains(childJoint))
                  continue;
               SpatialInertia subtreeInertia = MultiBodySystemTools.computeSubtreeInertia(childJoint);
               subtreeInertia.changeFrame(body.getBodyFixedFrame());
               if (lumped == null)
               {
                  lumped = new SpatialInertia(body.getBodyFixedFrame(), body.getBodyFixedFrame());
                  lumped.setIncludingFrame(body.getInertia());
               }
               lumped.add(subtreeInertia);
            }
            if (lumped != null)
               inertia = lumped;
         }
         for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
               J[9 * i + 3 * r + c] = inertia.getMomentOfInertia().getElement(r, c);
         mass[i] = inertia.getMass();
         inertia.getCenterOfMassOffset().get(3 * i, com);
         for (int index : provider.getJointDoFIndices(joint))
         {
            dof.add(index);
            maxDof = Math.max(maxDof, index);
         }
         for (int index : provider.getJointConfigurationIndices(joint))
         {
            cfg.add(index);
            maxCfg = Math.max(maxCfg, index);
         }
      }
      numberOfJoints = n;
      nv = maxDof + 1;
      nq = maxCfg + 1;

      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment desc = arena.allocate(DESC);
         desc.set(JAVA_INT, 0, n);
         desc.set(JAVA_INT, 4, nq);
         desc.set(JAVA_INT, 8, nv);
         long offset = 16;
         for (MemorySegment array : new MemorySegment[] {arena.allocateFrom(JAVA_INT, parent), arena.allocateFrom(JAVA_INT, type),
               arena.allocateFrom(JAVA_DOUBLE, axis), arena.allocateFrom(JAVA_DOUBLE, xBefore), arena.allocateFrom(JAVA_DOUBLE, xCom),
               arena.allocateFrom(JAVA_DOUBLE, J), arena.allocateFrom(JAVA_DOUBLE, mass), arena.allocateFrom(JAVA_DOUBLE, com),
               arena.allocateFrom(JAVA_INT, dof.stream().mapToInt(Integer::intValue).toArray()),
               arena.allocateFrom(JAVA_INT, cfg.stream().mapToInt(Integer::intValue).toArray())})
         {
            desc.set(ADDRESS, offset, array);
            offset += 8;
         }
         MemorySegment out = arena.allocate(ADDRESS);
         MecanoHipNative.check((int) MecanoHipNative.MODEL_CREATE.invokeExact(desc, out));
         handle = out.get(ADDRESS, 0);
         // the two model classes whose results are NOT held to 1e-10 against Mecano (INTEGRATION.md, "Conscious divergences"): say so once
         if (warnings() != 0)
            System.err.println("[mecano-hip] " + warningText());
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

   private static Object parentFrameOf(JointReadOnly joint)
   {
      RigidBodyReadOnly predecessor = joint.getPredecessor();
      return predecessor.isRootBody() ? predecessor.getBodyFixedFrame() : predecessor.getParentJoint().getFrameAfterJoint();
   }

   /** R row-major (9) then p (3): x_parent = R x_child + p. */
   private static void pack(RigidBodyTransformReadOnly transform, double[] array, int start)
   {
      for (int r = 0; r < 3; r++)
         for (int c = 0; c < 3; c++)
            array[start + 3 * r + c] = transform.getRotation().getElement(r, c);
      array[start + 9] = transform.getTranslationX();
      array[start + 10] = transform.getTranslationY();
      array[start + 11] = transform.getTranslationZ();
   }

   /** MH_WARN_* bits of mh_model_warnings: 1 = a revolute axis within 1e-7 of X / Y / Z but not on it, 2 = a composite mass under 1e-7. */
   public int warnings()
   {
      try
      {
         return (int) MecanoHipNative.MODEL_WARNINGS.invokeExact(handle);
      }
      catch (Throwable t)
      {
         throw new IllegalStateException(t);
      }
   }

   /** Which joints the warning bits concern and the bound on the difference to Mecano; "" when no bit is set. */
   public String warningText()
   {
      try
      {
         return ((MemorySegment) MecanoHipNative.MODEL_WARNING_TEXT.invokeExact(handle)).reinterpret(4096).getString(0);
      }
      catch (Throwable t)
      {
         throw new IllegalStateException(t);
      }
   }

   /** "topo:<key>" when a topology-specialised code object serves this model, "generic" (+ the reason) otherwise. */
   public String kernelVariant()
   {
      try
      {
         return ((MemorySegment) MecanoHipNative.MODEL_KERNEL_VARIANT.invokeExact(handle)).reinterpret(1024).getString(0);
      }
      catch (Throwable t)
      {
         throw new IllegalStateException(t);
      }
   }

   /** JointSourceMode of every listed joint (0 = EFFORT_SOURCE, 1 = ACCELERATION_SOURCE); null resets (ForwardDynamicsCalculator.java:400-444). */
   public void setJointSourceModes(int[] modes)
   {
      try (Arena arena = Arena.ofConfined())
      {
         MemorySegment seg = modes == null ? MemorySegment.NULL : arena.allocateFrom(JAVA_INT, modes);
         MecanoHipNative.invoke(() -> (int) MecanoHipNative.SET_JOINT_SOURCE_MODES.invokeExact(handle, seg));
      }
   }

   /** Position of a joint in the model's joint list (row block of the per-body / per-joint outputs), -1 for a joint that is not listed. */
   public int indexOf(JointReadOnly joint)
   {
      return jointList.indexOf(joint);
   }

   @Override
   public void close()
   {
      try
      {
         MecanoHipNative.MODEL_DESTROY.invokeExact(handle);
      }
      catch (Throwable t)
      {
         throw new IllegalStateException(t);
      }
   }
}
