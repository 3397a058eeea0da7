"""ctypes binding of libmunit_hip.so (the C ABI declared in include/munit_hip.h).

The product path has NO fallback: if the library is missing or a symbol is absent the import
of the compute ops fails loudly (RuntimeError).  Loading the library needs no GPU, so the
CPU test-suite can check that every declared symbol is exported.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MUNIT_HIP_LIB: developer override used by tools/ab_bench.sh to A/B two builds of the library in one GPU session
LIB_PATH = os.environ.get("MUNIT_HIP_LIB") or os.path.join(_HERE, "libmunit_hip.so")

ACT = {"none": 0, "relu": 1, "lrelu": 2, "tanh": 3}
PAD = {"zero": 0, "reflect": 1}
COMPUTE = {"f32": 0, "bf16": 1, "f32x3": 2}
DTYPE = {"f32": 0, "bf16": 1}     # MUNIT_DTYPE_*: element type of an activation tensor in HBM


class ConvDesc(Structure):
    """munit_conv_desc."""
    _fields_ = [("B", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int),
                ("Cout", c_int), ("KH", c_int), ("KW", c_int),
                ("stride", c_int), ("pad", c_int), ("pad_mode", c_int),
                ("upsample", c_int), ("act", c_int), ("slope", c_float), ("compute", c_int),
                ("in_dtype", c_int), ("out_dtype", c_int)]


class PrepItem(Structure):
    """munit_prep_item."""
    _fields_ = [("w", c_void_p), ("wp", c_void_p), ("Cout", c_int), ("KH", c_int), ("KW", c_int), ("Cin", c_int),
                ("kind", c_int), ("ps", c_int), ("bf16", c_int)]


class ImageDesc(Structure):
    """munit_image_desc."""
    _fields_ = [("src_off", ctypes.c_longlong), ("src_h", c_int), ("src_w", c_int), ("rs_h", c_int), ("rs_w", c_int),
                ("crop_i", c_int), ("crop_j", c_int), ("flip", c_int), ("reserved", c_int)]


_P = c_void_p  # device pointers travel as integers
_DESC = POINTER(ConvDesc)

# name -> (restype, argtypes); the keys are exactly the functions include/munit_hip.h declares
SIGNATURES = {
    "munit_version": (c_int, []),
    "munit_last_error": (c_char_p, []),
    "munit_stream_wait_stream": (c_int, [_P, _P]),
    "munit_comm_unique_id": (c_int, [_P, c_size_t]),
    "munit_comm_init": (c_int, [_P, c_int, c_int, _P]),
    "munit_comm_allreduce": (c_int, [_P, _P, c_size_t, _P]),
    "munit_comm_destroy": (c_int, [_P]),
    "munit_shutdown": (c_int, []),
    "munit_conv2d_out_hw": (c_int, [_DESC, POINTER(c_int), POINTER(c_int)]),
    "munit_conv2d_fwd_workspace_bytes": (c_size_t, [_DESC]),
    "munit_conv2d_fwd": (c_int, [_DESC, _P, _P, _P, _P, _P, c_size_t, _P]),
    "munit_conv2d_dgrad_workspace_bytes": (c_size_t, [_DESC]),
    "munit_conv2d_dgrad": (c_int, [_DESC, _P, _P, _P, _P, _P, c_size_t, _P]),
    "munit_conv2d_wgrad_workspace_bytes": (c_size_t, [_DESC]),
    "munit_conv2d_wgrad": (c_int, [_DESC, _P, _P, _P, _P, c_float, _P, c_size_t, _P]),
    "munit_conv2d_prepared_weight_bytes": (c_size_t, [_DESC, c_int]),
    "munit_conv2d_prep_item": (c_int, [_DESC, c_int, _P, _P, POINTER(PrepItem)]),
    "munit_conv2d_prepare_weights": (c_int, [POINTER(PrepItem), _P]),
    "munit_conv2d_prepare_weights_batch": (c_int, [_P, c_int, _P]),
    "munit_conv2d_fwd_prepared": (c_int, [_DESC, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "munit_conv2d_dgrad_prepared": (c_int, [_DESC, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "munit_conv2d_executed_flops": (c_double, [_DESC, c_int]),
    "munit_conv2d_kernel_name": (c_char_p, [_DESC, c_int]),
    "munit_linear_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "munit_linear_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    "munit_linear_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    "munit_act_bwd": (c_int, [c_int, c_float, _P, _P, _P, c_size_t, _P]),
    "munit_instnorm_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "munit_instnorm_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, c_int,
                                   c_float, _P, c_size_t, _P]),
    "munit_instnorm_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int,
                                   _P, c_size_t, _P]),
    "munit_instnorm_fwd_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, c_int,
                                        c_float, _P, c_size_t, _P]),
    "munit_instnorm_bwd_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int,
                                        _P, c_size_t, _P]),
    "munit_layernorm_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "munit_layernorm_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, c_float, _P, c_size_t, _P]),
    "munit_layernorm_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_int,
                                    c_float, _P, c_size_t, _P]),
    "munit_layernorm_fwd_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, c_float, _P, c_size_t, _P]),
    "munit_layernorm_bwd_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_int,
                                         c_float, _P, c_size_t, _P]),
    "munit_avgpool3s2_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "munit_avgpool3s2_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "munit_gap_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "munit_gap_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "munit_loss_workspace_bytes": (c_size_t, [c_size_t]),
    "munit_l1_mean_fwd": (c_int, [_P, _P, _P, c_size_t, c_int, _P, _P, c_size_t, _P]),
    "munit_l1_mean_bwd": (c_int, [_P, _P, _P, c_size_t, c_int, _P, _P, _P, _P]),
    "munit_l1_mean_fwd_bf16": (c_int, [_P, _P, _P, c_size_t, c_int, _P, _P, c_size_t, _P]),
    "munit_l1_mean_bwd_bf16": (c_int, [_P, _P, _P, c_size_t, c_int, _P, _P, _P, _P]),
    "munit_mse_const_fwd": (c_int, [_P, c_float, c_size_t, _P, _P, c_size_t, _P]),
    "munit_mse_const_bwd": (c_int, [_P, c_float, c_size_t, _P, _P, _P]),
    "munit_weighted_sum": (c_int, [POINTER(c_void_p), POINTER(c_float), c_int, _P, _P]),
    "munit_adam_step": (c_int, [_P, _P, _P, _P, c_size_t, c_double, c_double, c_double, c_double, c_double, c_int,
                                _P]),
    "munit_extraadam_step": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_double, c_double, c_double, c_double,
                                     c_double, c_int, c_int, _P]),
    "munit_scale": (c_int, [_P, _P, c_size_t, c_float, c_int, _P]),
    "munit_image_ksize": (c_int, [c_int, c_int]),
    "munit_image_preprocess_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "munit_image_preprocess": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    "munit_mask_preprocess_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "munit_mask_preprocess": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
}

_lib = None


def load():
    """Load the shared library once; raise RuntimeError (never fall back) when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "munit_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C munit_amd/csrc`). There is no CPU / PyTorch fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError("munit_amd: %s does not export %s" % (LIB_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().munit_last_error()
        raise RuntimeError("munit_hip %s failed (rc=%d): %s" % (what, rc, (msg or b"").decode()))
