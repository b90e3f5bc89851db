"""ctypes binding of libspx.so (include/spx.h).  There is NO fallback: if the HIP library is missing
or no GPU is visible, the product path raises -- it never computes on the CPU."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", os.environ.get("SPX_LIB_NAME", "libspx.so"))  # SPX_LIB_NAME: A/B builds (tools/)

c_double_p = ctypes.c_void_p  # device pointers are passed as plain addresses
_i64, _d, _p, _int = ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, ctypes.c_int
_f = ctypes.c_float

# name -> argtypes; must list every symbol declared in include/spx.h (tests/test_abi.py checks this)
SIGNATURES = {
    "spx_abi_version": [],
    "spx_last_error": [],
    "spx_ctx_create": [_int, ctypes.POINTER(_p)],
    "spx_ctx_create_on_stream": [_int, _p, ctypes.POINTER(_p)],
    "spx_ctx_destroy": [_p],
    "spx_sync": [_p],
    "spx_timer_start": [_p],
    "spx_timer_stop": [_p, ctypes.POINTER(ctypes.c_float)],
    "spx_ctx_set_tuning": [_p, _int, _int],
    "spx_ctx_set_value_target": [_p, _p],
    "spx_copy_strided": [_p, _p, _i64, _p, _i64, _i64, _int],
    "spx_synth_fill": [_p, _p, _i64, ctypes.c_uint64, ctypes.c_uint64, _int, _d],
    "spx_check_bounds": [_p, _p, _p, _d, _d, _i64, ctypes.POINTER(_int)],
    "spx_build_mask": [_p, _p, _i64, _p, _i64],
    "spx_prox_l1": [_p, _p, _p, _p, _p, _i64, _d, _d],
    "spx_prox_l0": [_p, _p, _p, _p, _p, _i64, _d, _d],
    "spx_prox_lhalf": [_p, _p, _p, _p, _p, _i64, _d, _d],
    "spx_prox_l1_box": [_p, _p, _p, _p, _p, _i64, _d, _d, _p, _p, _d, _d, _p],
    "spx_prox_l0_box": [_p, _p, _p, _p, _p, _i64, _d, _d, _p, _p, _d, _d, _p],
    "spx_prox_lhalf_box": [_p, _p, _p, _p, _p, _i64, _d, _d, _p, _p, _d, _d, _p],
    "spx_prox_l1_f32": [_p, _p, _p, _p, _p, _i64, _f, _f],
    "spx_prox_l0_f32": [_p, _p, _p, _p, _p, _i64, _f, _f],
    "spx_prox_l1_box_f32": [_p, _p, _p, _p, _p, _i64, _f, _f, _p, _p, _f, _f, _p],
    "spx_prox_l0_box_f32": [_p, _p, _p, _p, _p, _i64, _f, _f, _p, _p, _f, _f, _p],
    "spx_iprox_l1_f32": [_p, _p, _p, _p, _p, _p, _i64, _f, _int],
    "spx_iprox_l0_f32": [_p, _p, _p, _p, _p, _p, _i64, _f, _int],
    "spx_iprox_l1_box_f32": [_p, _p, _p, _p, _p, _p, _i64, _f, _p, _p, _f, _f, _p],
    "spx_iprox_l0_box_f32": [_p, _p, _p, _p, _p, _p, _i64, _f, _p, _p, _f, _f, _p],
    "spx_proxval_l1": [_p, _p, _p, _p, _p, _i64, _d, _d, _d, ctypes.POINTER(_d)],
    "spx_proxval_l0": [_p, _p, _p, _p, _p, _i64, _d, _d, _d, ctypes.POINTER(_d)],
    "spx_proxval_lhalf": [_p, _p, _p, _p, _p, _i64, _d, _d, _d, ctypes.POINTER(_d)],
    "spx_proxval_l1_box": [_p, _p, _p, _p, _p, _i64, _d, _d, _p, _p, _d, _d, _p, _d, ctypes.POINTER(_d)],
    "spx_proxval_l0_box": [_p, _p, _p, _p, _p, _i64, _d, _d, _p, _p, _d, _d, _p, _d, ctypes.POINTER(_d)],
    "spx_proxval_lhalf_box": [_p, _p, _p, _p, _p, _i64, _d, _d, _p, _p, _d, _d, _p, _d, ctypes.POINTER(_d)],
    "spx_iprox_l1": [_p, _p, _p, _p, _p, _p, _i64, _d, _int],
    "spx_iprox_l0": [_p, _p, _p, _p, _p, _p, _i64, _d, _int],
    "spx_iprox_l1_box": [_p, _p, _p, _p, _p, _p, _i64, _d, _p, _p, _d, _d, _p],
    "spx_iprox_l0_box": [_p, _p, _p, _p, _p, _p, _i64, _d, _p, _p, _d, _d, _p],
    "spx_obj_l1": [_p, _p, _p, _p, _i64, _d, ctypes.POINTER(_d)],
    "spx_obj_l0": [_p, _p, _p, _p, _i64, _d, ctypes.POINTER(_d)],
    "spx_obj_lhalf": [_p, _p, _p, _p, _i64, _d, ctypes.POINTER(_d)],
    "spx_obj_l1_box": [_p, _p, _p, _p, _i64, _d, _p, _p, _d, _d, _p, ctypes.POINTER(_d)],
    "spx_obj_l0_box": [_p, _p, _p, _p, _i64, _d, _p, _p, _d, _d, _p, ctypes.POINTER(_d)],
    "spx_obj_lhalf_box": [_p, _p, _p, _p, _i64, _d, _p, _p, _d, _d, _p, ctypes.POINTER(_d)],
    "spx_obj_indball_l0": [_p, _p, _p, _p, _i64, _i64, ctypes.POINTER(_d)],
    "spx_obj_indball_l0_binf": [_p, _p, _p, _p, _i64, _i64, _d, ctypes.POINTER(_d)],
    "spx_obj_group_l2": [_p, _p, _p, _p, _i64, _p, _i64, _i64, _p, ctypes.POINTER(_d)],
    "spx_obj_group_l2_binf": [_p, _p, _p, _p, _i64, _p, _i64, _i64, _p, _d, ctypes.POINTER(_d)],
    "spx_obj_l1_f32": [_p, _p, _p, _p, _i64, _f, ctypes.POINTER(_d)],
    "spx_obj_l0_f32": [_p, _p, _p, _p, _i64, _f, ctypes.POINTER(_d)],
    "spx_obj_lhalf_f32": [_p, _p, _p, _p, _i64, _f, ctypes.POINTER(_d)],
    "spx_obj_l1_box_f32": [_p, _p, _p, _p, _i64, _f, _p, _p, _f, _f, _p, ctypes.POINTER(_d)],
    "spx_obj_l0_box_f32": [_p, _p, _p, _p, _i64, _f, _p, _p, _f, _f, _p, ctypes.POINTER(_d)],
    "spx_obj_lhalf_box_f32": [_p, _p, _p, _p, _i64, _f, _p, _p, _f, _f, _p, ctypes.POINTER(_d)],
    "spx_obj_indball_l0_f32": [_p, _p, _p, _p, _i64, _i64, ctypes.POINTER(_d)],
    "spx_obj_indball_l0_binf_f32": [_p, _p, _p, _p, _i64, _i64, _f, ctypes.POINTER(_d)],
    "spx_obj_group_l2_f32": [_p, _p, _p, _p, _i64, _p, _i64, _i64, _p, ctypes.POINTER(_d)],
    "spx_obj_group_l2_binf_f32": [_p, _p, _p, _p, _i64, _p, _i64, _i64, _p, _f, ctypes.POINTER(_d)],
    "spx_prox_indball_l0": [_p, _p, _p, _p, _p, _i64, _i64],
    "spx_prox_indball_l0_binf": [_p, _p, _p, _p, _p, _i64, _i64, _d],
    "spx_prox_indball_l0_f32": [_p, _p, _p, _p, _p, _i64, _i64],
    "spx_prox_indball_l0_binf_f32": [_p, _p, _p, _p, _p, _i64, _i64, _f],
    "spx_prox_l1_b2": [_p, _p, _p, _p, _p, _i64, _d, _d, _d, _d],
    "spx_obj_l1_b2": [_p, _p, _p, _p, _i64, _d, _d, ctypes.POINTER(_d)],
    "spx_prox_group_l2": [_p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _p, _d],
    "spx_prox_group_l2_f32": [_p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _p, _f],
    "spx_prox_group_l2_binf": [_p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _p, _d, _d],
    "spx_prox_group_l2_gather": [_p, _p, _p, _p, _p, _i64, _p, _p, _i64, _i64, _p, _d],
    "spx_prox_group_l2_binf_gather": [_p, _p, _p, _p, _p, _i64, _p, _p, _i64, _i64, _p, _d, _d],
    "spx_obj_group_l2_gather": [_p, _p, _p, _p, _i64, _p, _p, _i64, _i64, _p, ctypes.POINTER(_d)],
    "spx_obj_group_l2_binf_gather": [_p, _p, _p, _p, _i64, _p, _p, _i64, _i64, _p, _d, ctypes.POINTER(_d)],
}
# host-pointer forms: spx_host_X has the argument list of spx_X (include/spx.h, "host-pointer forms")
SIGNATURES.update({"spx_host_" + k[4:]: list(v) for k, v in list(SIGNATURES.items())
                   if k.startswith(("spx_prox_", "spx_iprox_", "spx_obj_")) and not k.endswith("_f32")})


class SpxError(RuntimeError):
    """A libspx call returned a non-zero spx_status."""

    def __init__(self, status, message):
        super().__init__("libspx status %d: %s" % (status, message))
        self.status = status


_lib = None


def load():
    """Load libspx.so (built by csrc/build.sh / __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libspx.so not found at %s -- build it with shiftedproximaloperators.jl_amd/csrc/build.sh "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            fn.argtypes = args
            fn.restype = ctypes.c_char_p if name == "spx_last_error" else _int
        _lib = L
    return _lib


def check(status):
    if status != 0:
        msg = load().spx_last_error()
        raise SpxError(status, msg.decode() if msg else "")
