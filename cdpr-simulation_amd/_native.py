"""ctypes binding of libcdpr_hip.so (include/cdpr.h).

The library is built in-tree (`make -C cdpr-simulation_amd/csrc`, or
`__graft_entry__.build()`); there is no fallback: if it is missing, importing
the engine fails loudly.
"""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("CDPR_LIB", "libcdpr_hip.so"))  # CDPR_LIB: experimental builds (A/B runs)

# every symbol include/cdpr.h declares
EXPORTS = [
    "cdpr_abi_version", "cdpr_config_size", "cdpr_device_count", "cdpr_device_pci_bus_id", "cdpr_bytes_per_state_step", "cdpr_derivative_weights",
    "cdpr_create", "cdpr_destroy", "cdpr_reset", "cdpr_last_error", "cdpr_set_platform_state",
    "cdpr_set_velocity_command", "cdpr_set_position_command", "cdpr_set_velocity_command_device",
    "cdpr_set_position_command_device", "cdpr_bind_velocity_command_device", "cdpr_bind_position_command_device",
    "cdpr_set_velocity_command_masked", "cdpr_set_position_command_masked", "cdpr_set_force_command", "cdpr_set_force_command_device",
    "cdpr_bind_force_command_device", "cdpr_set_force_command_masked", "cdpr_update", "cdpr_update_fused", "cdpr_observable_image_bytes", "cdpr_update_record", "cdpr_update_scheduled", "cdpr_update_scheduled_kind", "cdpr_decode_observables", "cdpr_decode_observables_f64", "cdpr_synchronize", "cdpr_mapping", "cdpr_plan_kernel", "cdpr_kernel_name", "cdpr_step_count",
    "cdpr_get_joint_states", "cdpr_get_platform_state", "cdpr_get_observables", "cdpr_get_pid_debug", "cdpr_get_fk_state", "cdpr_get_td_state", "cdpr_get_limit_state", "cdpr_get_observables_f64", "cdpr_get_raw_state_f64", "cdpr_set_platform_state_f64",
    "cdpr_get_raw_state", "cdpr_rollout_velocity", "cdpr_rollout_velocity_launch", "cdpr_rollout_velocity_fetch",
    "cdpr_rollout_velocity_device", "cdpr_device_malloc", "cdpr_device_free", "cdpr_device_upload", "cdpr_device_download",
    "cdpr_profile_begin", "cdpr_profile_end", "cdpr_solve_ik", "cdpr_solve_fk", "cdpr_solve_td",
]  # fmt: skip

_lib = None


class NativeLibraryMissing(ImportError):
    pass


def lib():
    """Load libcdpr_hip.so once and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or __graft_entry__.build()). There is no CPU fallback for the CDPR step engine."
        )
    L = C.CDLL(LIB_PATH)
    fp, dp, ip = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    cfgp = C.POINTER(_abi.ConfigStruct)
    H = C.c_void_p
    L.cdpr_abi_version.restype = C.c_uint32
    L.cdpr_config_size.restype = C.c_size_t
    L.cdpr_device_count.restype = C.c_int
    L.cdpr_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    L.cdpr_bytes_per_state_step.restype = C.c_size_t
    L.cdpr_bytes_per_state_step.argtypes = [cfgp]
    L.cdpr_derivative_weights.argtypes = [C.c_uint32, C.c_uint32, dp]
    L.cdpr_create.argtypes = [cfgp, C.c_int, C.POINTER(H)]
    L.cdpr_destroy.argtypes = [H]
    L.cdpr_destroy.restype = None
    L.cdpr_reset.argtypes = [H]
    L.cdpr_last_error.argtypes = [H]
    L.cdpr_last_error.restype = C.c_char_p
    L.cdpr_set_platform_state.argtypes = [H, fp, fp]
    for name in ("cdpr_set_velocity_command", "cdpr_set_position_command", "cdpr_set_force_command"):
        getattr(L, name).argtypes = [H, fp, C.c_size_t]
    for name in ("cdpr_set_velocity_command_device", "cdpr_set_position_command_device", "cdpr_bind_velocity_command_device",
                 "cdpr_bind_position_command_device", "cdpr_set_force_command_device", "cdpr_bind_force_command_device"):
        getattr(L, name).argtypes = [H, C.c_void_p, C.c_size_t]
    for name in ("cdpr_set_velocity_command_masked", "cdpr_set_position_command_masked", "cdpr_set_force_command_masked"):
        getattr(L, name).argtypes = [H, fp, C.c_size_t, C.POINTER(C.c_uint8)]
    L.cdpr_update.argtypes = [H, C.c_int]
    L.cdpr_update_fused.argtypes = [H, C.c_int, C.c_int]
    L.cdpr_observable_image_bytes.argtypes = [H, C.POINTER(C.c_size_t)]
    L.cdpr_update_record.argtypes = [H, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.cdpr_update_scheduled.argtypes = [H, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.cdpr_update_scheduled_kind.argtypes = [H, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.cdpr_decode_observables.argtypes = [H, C.c_void_p, fp, fp, fp, fp, fp]
    L.cdpr_decode_observables_f64.argtypes = [H, C.c_void_p, dp, dp, dp, dp, dp]
    L.cdpr_synchronize.argtypes = [H]
    L.cdpr_mapping.argtypes = [H]
    L.cdpr_mapping.restype = C.c_uint32
    L.cdpr_plan_kernel.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_char_p, C.c_size_t]
    L.cdpr_kernel_name.argtypes = [H, C.c_char_p, C.c_size_t]
    L.cdpr_step_count.argtypes = [H]
    L.cdpr_step_count.restype = C.c_uint64
    L.cdpr_get_joint_states.argtypes = [H, fp, fp, fp]
    L.cdpr_get_platform_state.argtypes = [H, fp, fp]
    L.cdpr_get_observables.argtypes = [H, fp, fp, fp, fp, fp]
    L.cdpr_get_raw_state.argtypes = [H, fp, fp]
    L.cdpr_get_pid_debug.argtypes = [H, fp]
    L.cdpr_get_fk_state.argtypes = [H, fp, fp, ip]
    L.cdpr_get_td_state.argtypes = [H, fp, ip]
    L.cdpr_get_limit_state.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.cdpr_get_observables_f64.argtypes = [C.c_void_p, dp, dp, dp, dp, dp]
    L.cdpr_get_raw_state_f64.argtypes = [C.c_void_p, dp, dp]
    L.cdpr_set_platform_state_f64.argtypes = [C.c_void_p, dp, dp]
    L.cdpr_rollout_velocity.argtypes = [H, C.c_int, C.c_int, C.c_void_p, fp, fp]
    L.cdpr_rollout_velocity_launch.argtypes = [H, C.c_int, C.c_int, C.c_void_p, fp]
    L.cdpr_rollout_velocity_fetch.argtypes = [H, fp]
    L.cdpr_rollout_velocity_device.argtypes = [H, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cdpr_device_malloc.argtypes = [H, C.c_size_t, C.POINTER(C.c_void_p)]
    L.cdpr_device_free.argtypes = [H, C.c_void_p]
    L.cdpr_device_upload.argtypes = [H, C.c_void_p, C.c_void_p, C.c_size_t]
    L.cdpr_device_download.argtypes = [H, C.c_void_p, C.c_void_p, C.c_size_t]
    L.cdpr_profile_begin.argtypes = [H]
    L.cdpr_profile_end.argtypes = [H, fp, C.POINTER(C.c_uint64)]
    L.cdpr_solve_ik.argtypes = [H, fp, fp, fp, fp, fp]
    L.cdpr_solve_fk.argtypes = [H, fp, fp, fp, fp, ip]
    L.cdpr_solve_td.argtypes = [H, fp, fp, fp, ip]
    if L.cdpr_abi_version() != _abi.ABI_VERSION:
        raise ImportError("libcdpr_hip.so ABI version mismatch")
    if L.cdpr_config_size() != C.sizeof(_abi.ConfigStruct):
        raise ImportError("cdpr_config_t layout mismatch between include/cdpr.h and _abi.py")
    _lib = L
    return L
