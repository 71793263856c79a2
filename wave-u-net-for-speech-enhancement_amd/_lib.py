"""ctypes binding of csrc/libwunet_hip.so (the C ABI of include/wunet_hip.h).

There is deliberately no fallback: if the HIP library is missing or fails to load the import
raises, so a GPU box can never silently run something else.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (WUNET_LIB_PATH: measurement builds of the same library, e.g. kernel ablations - never set in production)
LIB_PATH = os.environ.get("WUNET_LIB_PATH") or os.path.join(_HERE, "csrc", "libwunet_hip.so")

EXPORTS = [
    "wunet_last_error", "wunet_create", "wunet_destroy", "wunet_workspace_bytes", "wunet_forward",
    "wunet_backward", "wunet_backward_range", "wunet_loss_scratch_bytes", "wunet_loss_forward",
    "wunet_loss_backward", "wunet_layer_info", "wunet_num_conv_layers", "wunet_op_conv1d",
    "wunet_op_conv1d_dgrad", "wunet_op_conv1d_wgrad", "wunet_profile_enable", "wunet_profile_collect",
    "wunet_adam_step", "wunet_set_h3", "wunet_op_conv1d_split", "wunet_op_conv1d_dgrad_split", "wunet_op_conv1d_wgrad_split",
    "wunet_backward_range_async", "wunet_backward_join", "wunet_debug_set_conv_trace", "wunet_crop_windows",
    "wunet_comm_unique_id", "wunet_comm_create", "wunet_comm_allreduce_sum", "wunet_comm_world", "wunet_comm_destroy",
    "wunet_debug_stamps",
]

_vp = ctypes.c_void_p
_i = ctypes.c_int
_sz = ctypes.c_size_t


def declare(lib):
    """Attach argtypes/restypes to a loaded library object (HIP build or the test emulator build)."""
    lib.wunet_last_error.restype = ctypes.c_char_p
    lib.wunet_last_error.argtypes = []
    lib.wunet_create.argtypes = [_i, _i, _i, _i, ctypes.POINTER(_vp)]
    lib.wunet_destroy.argtypes = [_vp]
    lib.wunet_destroy.restype = None
    lib.wunet_workspace_bytes.argtypes = [_vp, _i]
    lib.wunet_workspace_bytes.restype = _sz
    lib.wunet_forward.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]
    lib.wunet_backward.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.wunet_backward_range.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]
    lib.wunet_backward_range_async.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]
    lib.wunet_backward_join.argtypes = [_vp, _vp]
    lib.wunet_loss_scratch_bytes.restype = _sz
    lib.wunet_loss_scratch_bytes.argtypes = []
    lib.wunet_loss_forward.argtypes = [_i, _vp, _vp, _sz, _vp, _vp, _vp]
    lib.wunet_loss_backward.argtypes = [_i, _vp, _vp, _vp, _sz, _vp, _vp]
    lib.wunet_layer_info.argtypes = [_vp, _i, ctypes.POINTER(_sz), ctypes.POINTER(_i), ctypes.POINTER(_i)]
    lib.wunet_num_conv_layers.argtypes = [_vp]
    for name in ("wunet_op_conv1d", "wunet_op_conv1d_split"):
        getattr(lib, name).argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    for name in ("wunet_op_conv1d_dgrad", "wunet_op_conv1d_wgrad", "wunet_op_conv1d_dgrad_split", "wunet_op_conv1d_wgrad_split"):
        getattr(lib, name).argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    lib.wunet_adam_step.argtypes = [_i, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_double, ctypes.c_longlong, ctypes.c_double, _vp, _vp, _vp]
    lib.wunet_set_h3.argtypes = [_vp, _i]
    lib.wunet_profile_enable.argtypes = [_i]
    lib.wunet_profile_collect.argtypes = [ctypes.c_char_p, _sz]
    lib.wunet_profile_collect.restype = ctypes.c_longlong
    if hasattr(lib, "wunet_debug_stamps"):       # (a measurement hook of round 6: saved libraries of earlier rounds - tools/round_vs_round.sh - do not have it)
        lib.wunet_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), _i]
        lib.wunet_debug_stamps.restype = _i
    lib.wunet_crop_windows.argtypes = [_vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp, _vp, _vp]
    lib.wunet_comm_unique_id.argtypes = [_vp]
    lib.wunet_comm_create.argtypes = [_vp, _i, _i, ctypes.POINTER(_vp)]
    lib.wunet_comm_allreduce_sum.argtypes = [_vp, _vp, _sz, _vp]
    lib.wunet_comm_world.argtypes = [_vp]
    lib.wunet_comm_destroy.argtypes = [_vp]
    lib.wunet_comm_destroy.restype = None
    lib.wunet_debug_set_conv_trace.argtypes = [_vp]
    lib.wunet_debug_set_conv_trace.restype = None
    return lib


_HIP = None


def load_hip():
    """Load libwunet_hip.so or raise.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
    (or `make -C <package>/csrc -j8`)."""
    global _HIP
    if _HIP is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the MI355X HIP library has not been built "
                "(run `make -C wave-u-net-for-speech-enhancement_amd/csrc -j8`); there is no CPU fallback")
        _HIP = declare(ctypes.CDLL(LIB_PATH))
    return _HIP
