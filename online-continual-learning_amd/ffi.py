"""ctypes binding of libocl_hip.so (include/ocl_hip.h).

This is the only place the product path touches native code; there is NO fallback: if the library is
missing, or a call fails, a RuntimeError is raised (the reference's error convention is exceptions).
PyTorch is used for device memory and streams only: every call takes raw `tensor.data_ptr()`s and
torch's current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCL_LIB") or os.path.join(_HERE, "libocl_hip.so")   # (OCL_LIB: a measurement build of the same library)
_lib = None

i64 = C.c_int64
i32 = C.c_int32
f32 = C.c_float
vp = C.c_void_p


class NetDesc(C.Structure):
    _fields_ = [("in_h", i32), ("in_w", i32), ("nf", i32), ("n_classes", i32), ("head", i32),
                ("feat_dim", i32), ("max_batch", i32), ("n_slots", i32)]


# name -> (restype, argtypes); kept in one table so tests can check it against the header
SIGNATURES = {
    "ocl_version": (C.c_int, []),
    "ocl_last_error": (C.c_char_p, []),
    "ocl_init": (C.c_int, [C.c_int]),
    "ocl_upload": (C.c_int, [vp, i64, vp, vp]),
    "ocl_gather_rows": (C.c_int, [vp, vp, i64, i64, vp, vp]),
    "ocl_scatter_rows": (C.c_int, [vp, vp, i64, i64, vp, vp]),
    "ocl_gather_rows_pair": (C.c_int, [vp, i64, vp, vp, i64, vp, vp, vp, i64, vp]),
    "ocl_gather_u8_hwc_to_f32_chw": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ocl_sgd_step": (C.c_int, [vp, vp, i64, f32, f32, f32, vp, vp]),
    "ocl_ce_fwd_bwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "ocl_ce_segmented_fwd_bwd": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "ocl_kd_fwd_bwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp]),
    "ocl_supcon_workspace_bytes": (i64, [C.c_int]),
    "ocl_supcon_fwd_bwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, f32, vp, vp, vp, vp]),
    "ocl_knn_sv": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "ocl_col_reduce": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ocl_aser_score": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ocl_argsort_desc": (C.c_int, [vp, C.c_int, vp, vp]),
    "ocl_ncm_class_means": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp]),
    "ocl_ncm_predict": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]),
    "ocl_mir_scores": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "ocl_cosine_max_workspace_bytes": (i64, [C.c_int]),
    "ocl_cosine_max": (C.c_int, [vp, C.c_int, i64, vp, f32, vp, vp, vp]),
    "ocl_scr_augment": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ocl_scr_augment_uniform": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_double), vp, vp]),
    "ocl_gemm_small": (C.c_int, [vp, i64, i64, vp, i64, i64, vp, i64, C.c_int, C.c_int, C.c_int, vp,
                                 C.c_int, C.c_int, vp]),
    "ocl_net_create": (C.c_int, [C.POINTER(NetDesc), C.POINTER(vp)]),
    "ocl_net_destroy": (None, [vp]),
    "ocl_net_param_count": (i64, [vp]),
    "ocl_net_num_tensors": (i32, [vp]),
    "ocl_net_tensor_info": (C.c_int, [vp, C.c_int, C.c_char_p, C.POINTER(i64), C.POINTER(i32),
                                      C.POINTER(i64)]),
    "ocl_net_num_bn": (i32, [vp]),
    "ocl_net_bn_stat_count": (i64, [vp]),
    "ocl_net_bn_info": (C.c_int, [vp, C.c_int, C.c_char_p, C.POINTER(i64), C.POINTER(i32)]),
    "ocl_net_feature_dim": (i32, [vp]),
    "ocl_net_out_dim": (i32, [vp]),
    "ocl_net_workspace_bytes": (i64, [vp]),
    "ocl_net_bind": (C.c_int, [vp, vp, vp, vp, vp, vp, i64]),
    "ocl_net_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_uint32, vp, vp, vp, C.c_int, vp]),
    "ocl_net_forward_segments": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i32), C.c_int, C.c_int, C.c_uint32, vp, vp, vp, C.c_int, vp]),
    "ocl_net_backward": (C.c_int, [vp, C.c_int, vp, C.c_int, vp]),
    "ocl_net_debug_stop": (C.c_int, [vp, C.c_int]),
    "ocl_net_debug_copy": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, i64, C.POINTER(i64), vp]),
    "ocl_bn_bwd_nhwc": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp]),
    "ocl_set_deterministic": (C.c_int, [C.c_int]),
    "ocl_prof_enable": (C.c_int, [C.c_int]),
    "ocl_prof_reset": (C.c_int, []),
    "ocl_prof_query": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(i64)]),
    "ocl_mfma_calibrate": (C.c_int, [C.c_int, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]),
}

FWD_TRAIN, FWD_SAVE_TAPE, FWD_UPDATE_RUNNING, FWD_FROZEN_BN, FWD_SAME_WEIGHTS, FWD_PACK_ALL = 1, 2, 4, 8, 16, 32


# Every OCL_* variable the package or the library reads (tests/test_cpu_host.py keeps this list equal to the sources).  A variable that is
# not here does nothing -- an A/B script written for a knob that has since been hard-wired would compare the tree with itself -- so
# loading the library warns about it.  (OCL_NONE: the scripts' "no switch" placeholder; the rest of the second line: test / script harness.)
KNOWN_ENV = frozenset("""
OCL_ASER_AUTOGRAD OCL_ASER_PIPELINE OCL_ASER_SPLIT OCL_BNB_EPI OCL_BN_CHAN OCL_BN_FUSED OCL_CBRS_EMULATE OCL_CBRS_VERIFY_EVERY
OCL_CONV_PIPE OCL_CONV_Q4 OCL_CONV_S OCL_CONV_S_NT OCL_CONV_W OCL_CONV_WX OCL_DATA_STREAM OCL_DEBUG_SKIP_BN2FWD OCL_DEBUG_SKIP_SHORTCUT
OCL_DEBUG_SKIP_WGRAD OCL_DETERMINISTIC OCL_DIST_BACKEND OCL_DY_KEEP OCL_GC_FREEZE OCL_GRAPH OCL_GRAPH_VERBOSE OCL_LIB OCL_LOG_PLANS
OCL_PIN OCL_SIDE_EXTRA_MIN OCL_SINGLE_STREAM OCL_WGRAD_ENOUGH OCL_WGRAD_FLUSH OCL_WGRAD_MULTI OCL_WGRAD_MULTI_TARGET OCL_WGRAD_Q
OCL_WGRAD_TARGET""".split())
HARNESS_ENV = frozenset("OCL_NONE OCL_TEST_CASES OCL_TEST_PORT OCL_SHARD_BACKEND OCL_PROBE_STREAM OCL_EAGER_DEVICE".split())


def unknown_env(environ=None):
    """The OCL_* variables of the environment that nothing reads (sorted)."""
    environ = os.environ if environ is None else environ
    return sorted(k for k in environ if k.startswith("OCL_") and k not in KNOWN_ENV and k not in HARNESS_ENV)


def lib():
    """Loads the shared library (once). Raises if it has not been built: no CPU fallback exists."""
    global _lib
    if _lib is None:
        dead = unknown_env()
        if dead:
            import warnings
            warnings.warn("environment variables %s are not read by ocl_amd / libocl_hip.so (removed or misspelt switch?): "
                          "they change nothing" % ", ".join(dead), RuntimeWarning, stacklevel=2)
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "libocl_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ocl_last_error()
        raise RuntimeError("libocl_hip %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


_inited = set()


def init(device_index=None):
    """ocl_init on the current (or given) device: verifies gfx950."""
    if _inited:      # every op calls this: once a device is set up the check is one raw device query
        d = torch._C._cuda_getDevice() if device_index is None else device_index
        if d in _inited:
            return d
    if not torch.cuda.is_available():
        raise RuntimeError("online-continual-learning_amd needs an MI355X (gfx950) visible to PyTorch-ROCm; "
                           "no GPU is visible and there is no CPU fallback.")
    if device_index is None:
        device_index = torch.cuda.current_device()
    if device_index not in _inited:
        check(lib().ocl_init(int(device_index)), "ocl_init")
        _inited.add(device_index)
    return device_index


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device.  The raw accessor is ~0.3 us; going through
    `torch.cuda.current_stream()` builds a Stream object and resolves the device index three times (~9 us, and an op needs it
    for every launch: 0.2 ms of host time per replay step)."""
    if _raw_stream is not None:
        return vp(_raw_stream(torch._C._cuda_getDevice()))
    return vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Raw device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return vp(0)
    if not t.is_contiguous():
        raise RuntimeError("ffi.ptr: tensor must be contiguous")
    return vp(t.data_ptr())
