"""ctypes binding of the C-ABI (include/mecano_hip.h).  Fails loudly when the library is missing: there is no
Python or CPU fallback for the compute path."""
from __future__ import annotations

import ctypes
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MECANO_HIP_LIBRARY") or os.path.join(HERE, "libmecano_hip.so")  # (the override: A/B runs of two builds)

MH_OK = 0
STATUS_NAMES = {0: "MH_OK", 1: "MH_ERR_INVALID_ARGUMENT", 2: "MH_ERR_BAD_DIMENSION", 3: "MH_ERR_UNSUPPORTED_JOINT", 4: "MH_ERR_LOOP_CLOSURE",
                5: "MH_ERR_BAD_TOPOLOGY", 6: "MH_ERR_BAD_AXIS", 7: "MH_ERR_NO_DEVICE", 8: "MH_ERR_HIP", 9: "MH_ERR_OUT_OF_MEMORY",
                10: "MH_ERR_NOT_RESERVED", 11: "MH_ERR_SINGULAR"}
LAYOUT_AOS, LAYOUT_SOA = 0, 1

# every symbol include/mecano_hip.h declares (tests/test_abi.py checks the library exports each one)
ABI_SYMBOLS = [
    "mh_abi_version", "mh_spec_abi_stamp", "mh_build_hash", "mh_spec_sources_hash", "mh_spec_sources_hash_of", "mh_last_error", "mh_device_count", "mh_set_device", "mh_options_default", "mh_model_create", "mh_model_destroy",
    "mh_context_create", "mh_context_destroy", "mh_context_reserve", "mh_model_check",
    "mh_topology_key", "mh_build_code_object", "mh_model_nq", "mh_model_nv", "mh_model_n_joints", "mh_model_kernel_variant", "mh_model_warnings", "mh_model_warning_text", "mh_reserve", "mh_rnea_f64", "mh_aba_f64", "mh_crba_f64", "mh_rnea_aba_f64", "mh_rnea_crba_f64", "mh_regressor_f64", "mh_regressor_f32",
    "mh_model_set_joint_source_modes", "mh_model_n_acceleration_sources", "mh_aba_locked_f64", "mh_rnea_bodies_f64", "mh_aba_bodies_f64", "mh_rnea_joint_wrenches_f64", "mh_aba_joint_wrenches_f64", "mh_relative_acceleration_f64", "mh_crba_coriolis_f64", "mh_crba_coriolis_f32", "mh_centroidal_f64", "mh_centroidal_f32", "mh_integrate_f64", "mh_aba_integrate_f64", "mh_integrate_f32", "mh_rnea_f32", "mh_aba_f32", "mh_crba_f32", "mh_rnea_aba_f32", "mh_rnea_bodies_f32", "mh_aba_bodies_f32", "mh_aba_locked_f32", "mh_rnea_f64_host", "mh_aba_f64_host", "mh_crba_f64_host", "mh_rnea_f32_host", "mh_aba_f32_host", "mh_crba_f32_host", "mh_rnea_aba_f64_host", "mh_host_alloc", "mh_host_free", "mh_host_register", "mh_host_unregister", "mh_device_alloc", "mh_device_free", "mh_copy_to_device", "mh_copy_to_host", "mh_stream_synchronize", "mh_crba_coriolis_f64_host", "mh_centroidal_f64_host", "mh_timer_create",
    "mh_timer_destroy", "mh_timer_start", "mh_timer_stop", "mh_timer_elapsed_ms",
    "mh_shard_range", "mh_comm_unique_id", "mh_comm_create", "mh_comm_destroy", "mh_comm_size", "mh_comm_broadcast", "mh_comm_broadcast_host",
    "mh_comm_all_gather_rows", "mh_comm_gather_plan", "mh_comm_barrier",
]


class MhModelDesc(ctypes.Structure):
    _fields_ = [("n_joints", ctypes.c_int32), ("nq", ctypes.c_int32), ("nv", ctypes.c_int32),
                ("parent", ctypes.c_void_p), ("joint_type", ctypes.c_void_p), ("axis", ctypes.c_void_p), ("X_before", ctypes.c_void_p),
                ("X_com", ctypes.c_void_p), ("inertia_J", ctypes.c_void_p), ("inertia_mass", ctypes.c_void_p),
                ("inertia_com", ctypes.c_void_p), ("dof_indices", ctypes.c_void_p), ("cfg_indices", ctypes.c_void_p)]


class MhOptions(ctypes.Structure):
    _fields_ = [("consider_coriolis", ctypes.c_int32), ("consider_accelerations", ctypes.c_int32), ("layout", ctypes.c_int32),
                ("use_root_acceleration", ctypes.c_int32), ("stream", ctypes.c_void_p), ("root_acceleration", ctypes.c_double * 6),
                ("context", ctypes.c_void_p)]


class MhGatherStep(ctypes.Structure):
    _fields_ = [("root", ctypes.c_int32), ("send_local", ctypes.c_int32), ("recv_offset", ctypes.c_int64), ("bytes", ctypes.c_int64)]


CENTROIDAL_FRAME_FIXED, CENTROIDAL_FRAME_AT_COM = 0, 1


class MecanoHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


_lib = None
_load_lock = threading.Lock()


def load():
    """Returns the loaded library; raises ImportError when it has not been built (python -m mecano_amd.build).  Thread-safe: the order
    "torch first, then the library" below must not be raced (a second thread that sees a half-imported torch in sys.modules would load
    the library against the system HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:
        return _load_locked()


def _load_locked():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m mecano_amd.build` (hipcc, gfx950). "
                          "mecano_amd has no CPU fallback.")
    # PyTorch wheels bundle their own HIP / HSA runtime (torch/lib/libamdhip64.so).  Two HIP runtimes in one process cannot both open
    # the GPU (whichever initialises second reports "no ROCm-capable device"), so when torch is importable it is imported FIRST: the
    # library's libamdhip64.so.7 dependency then binds to the copy torch already loaded and the process holds a single runtime.
    # Hosts without torch (the Java / C++ side of the C-ABI) simply get the system runtime.
    try:
        import torch  # noqa: F401  (a plain import: waits for an import that another thread has in flight)
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    try:  # two HIP runtimes in the process: say so now rather than as a puzzling "no device" later
        with open("/proc/self/maps") as maps:
            runtimes = sorted({line.split()[-1] for line in maps if "libamdhip64" in line})
        if len(runtimes) > 1:
            import warnings
            warnings.warn("two HIP runtimes are loaded (" + ", ".join(runtimes) + "): only the one that initialises first will see the GPU. "
                          "Import torch before anything loads libmecano_hip.so, or run without torch.", RuntimeWarning)
    except OSError:
        pass
    P, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    lib.mh_abi_version.restype = I32
    lib.mh_spec_abi_stamp.restype = ctypes.c_uint64
    lib.mh_build_hash.restype = ctypes.c_char_p
    lib.mh_spec_sources_hash.restype = ctypes.c_char_p
    lib.mh_spec_sources_hash_of.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.mh_last_error.restype = ctypes.c_char_p
    lib.mh_device_count.argtypes = [ctypes.POINTER(I32)]
    lib.mh_set_device.argtypes = [I32]
    lib.mh_options_default.argtypes = [ctypes.POINTER(MhOptions)]
    lib.mh_options_default.restype = None
    lib.mh_model_create.argtypes = [ctypes.POINTER(MhModelDesc), ctypes.POINTER(P)]
    lib.mh_model_destroy.argtypes = [P]
    lib.mh_model_destroy.restype = None
    for f in ("mh_model_nq", "mh_model_nv", "mh_model_n_joints"):
        getattr(lib, f).argtypes = [P]
        getattr(lib, f).restype = I32
    lib.mh_model_kernel_variant.argtypes = [P]
    lib.mh_model_kernel_variant.restype = ctypes.c_char_p
    lib.mh_model_warnings.argtypes = [P]
    lib.mh_model_warnings.restype = ctypes.c_uint32
    lib.mh_model_warning_text.argtypes = [P]
    lib.mh_model_warning_text.restype = ctypes.c_char_p
    lib.mh_topology_key.argtypes = [ctypes.POINTER(MhModelDesc), ctypes.c_char_p, P, P]
    lib.mh_build_code_object.argtypes = [ctypes.POINTER(MhModelDesc), ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    lib.mh_reserve.argtypes = [P, I64]
    lib.mh_context_create.argtypes = [P, ctypes.POINTER(P)]
    lib.mh_context_destroy.argtypes = [P]
    lib.mh_context_destroy.restype = None
    lib.mh_context_reserve.argtypes = [P, I64]
    lib.mh_model_check.argtypes = [P, P, P]
    opt = ctypes.POINTER(MhOptions)
    for f in ("mh_rnea_f64", "mh_aba_f64", "mh_rnea_f32", "mh_aba_f32", "mh_rnea_f64_host", "mh_aba_f64_host", "mh_rnea_f32_host", "mh_aba_f32_host"):
        getattr(lib, f).argtypes = [P, I64, P, P, P, P, P, opt, P]
    for f in ("mh_crba_f64", "mh_crba_f32", "mh_crba_f64_host", "mh_crba_f32_host"):
        getattr(lib, f).argtypes = [P, I64, P, opt, P]
    lib.mh_rnea_aba_f64.argtypes = [P, I64, P, P, P, P, P, P, opt, P, P]
    lib.mh_rnea_aba_f32.argtypes = [P, I64, P, P, P, P, P, P, opt, P, P]
    lib.mh_rnea_crba_f64.argtypes = [P, I64, P, P, P, P, P, opt, P, P]
    lib.mh_regressor_f64.argtypes = [P, I64, P, P, P, P, opt, ctypes.c_int32, P]
    lib.mh_regressor_f32.argtypes = [P, I64, P, P, P, P, opt, ctypes.c_int32, P]
    lib.mh_rnea_aba_f64_host.argtypes = [P, I64, P, P, P, P, P, P, opt, P, P]
    lib.mh_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(P)]
    lib.mh_host_free.argtypes = [P]
    lib.mh_host_register.argtypes = [P, ctypes.c_size_t]
    lib.mh_host_unregister.argtypes = [P]
    lib.mh_device_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(P)]
    lib.mh_device_free.argtypes = [P]
    lib.mh_copy_to_device.argtypes = [P, P, ctypes.c_size_t, P]
    lib.mh_copy_to_host.argtypes = [P, P, ctypes.c_size_t, P]
    lib.mh_stream_synchronize.argtypes = [P]
    lib.mh_model_set_joint_source_modes.argtypes = [P, P]
    lib.mh_model_n_acceleration_sources.argtypes = [P]
    lib.mh_model_n_acceleration_sources.restype = I32
    lib.mh_aba_locked_f64.argtypes = [P, I64, P, P, P, P, P, P, opt, P, P]
    lib.mh_aba_locked_f32.argtypes = [P, I64, P, P, P, P, P, P, opt, P, P]
    for f in ("mh_rnea_bodies_f64", "mh_aba_bodies_f64", "mh_rnea_bodies_f32", "mh_aba_bodies_f32"):
        getattr(lib, f).argtypes = [P, I64, P, P, P, P, P, opt, P, P, P]
    for f in ("mh_rnea_joint_wrenches_f64", "mh_aba_joint_wrenches_f64"):
        getattr(lib, f).argtypes = [P, I64, P, P, P, P, P, opt, P, P]
    lib.mh_relative_acceleration_f64.argtypes = [P, I64, P, P, P, P, I32, P, P, opt, P]
    for f in ("mh_crba_coriolis_f64", "mh_crba_coriolis_f32", "mh_crba_coriolis_f64_host"):
        getattr(lib, f).argtypes = [P, I64, P, P, opt, P, P]
    for f in ("mh_centroidal_f64", "mh_centroidal_f32", "mh_centroidal_f64_host"):
        getattr(lib, f).argtypes = [P, I64, P, P, P, I32, opt, P, P, P]
    for f in ("mh_integrate_f64", "mh_integrate_f32"):
        getattr(lib, f).argtypes = [P, I64, ctypes.c_double, P, P, P, opt, P, P, P]
    lib.mh_aba_integrate_f64.argtypes = [P, I64, ctypes.c_double, P, P, P, P, P, opt, P, P, P]
    lib.mh_timer_create.argtypes = [ctypes.POINTER(P)]
    lib.mh_timer_destroy.argtypes = [P]
    lib.mh_timer_destroy.restype = None
    lib.mh_timer_start.argtypes = [P, P]
    lib.mh_timer_stop.argtypes = [P, P]
    lib.mh_timer_elapsed_ms.argtypes = [P, ctypes.POINTER(ctypes.c_float)]
    lib.mh_shard_range.argtypes = [I64, I32, I32, ctypes.POINTER(I64), ctypes.POINTER(I64)]
    lib.mh_comm_unique_id.argtypes = [P]
    lib.mh_comm_create.argtypes = [P, I32, I32, ctypes.POINTER(P)]
    lib.mh_comm_destroy.argtypes = [P]
    lib.mh_comm_size.argtypes = [P, ctypes.POINTER(I32), ctypes.POINTER(I32)]
    lib.mh_comm_broadcast.argtypes = [P, P, ctypes.c_size_t, I32, P]
    lib.mh_comm_broadcast_host.argtypes = [P, P, ctypes.c_size_t, I32]
    lib.mh_comm_all_gather_rows.argtypes = [P, P, I64, ctypes.c_size_t, P, P]
    lib.mh_comm_barrier.argtypes = [P, P]
    lib.mh_comm_gather_plan.argtypes = [I64, ctypes.c_size_t, I32, I32, I32, ctypes.POINTER(MhGatherStep), I32, ctypes.POINTER(I32)]
    _lib = lib
    return lib


def check(status: int):
    if status != MH_OK:
        raise MecanoHipError(status, load().mh_last_error().decode())
