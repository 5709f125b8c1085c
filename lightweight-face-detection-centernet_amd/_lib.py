"""ctypes binding of libcenterface_hip.so (include/centerface_hip.h).  No CPU fallback: if the
library is missing or a call fails this module raises."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CF_LIB: load another build of the same library (A/B runs of kernel variants: tools/ab_build.sh); default = the in-tree build
LIB_PATH = os.environ.get("CF_LIB") or os.path.join(_HERE, "libcenterface_hip.so")
CSRC = os.path.join(_HERE, "csrc")

CF_OK = 0
CF_F32, CF_BF16, CF_F32_SPLIT = 0, 1, 2
CF_IN_U8_HWC_BGR, CF_IN_F32_NCHW = 0, 1
CF_FLAG_COLLAPSE_HEADS, CF_FLAG_NO_GRAPH, CF_FLAG_NO_FUSE, CF_FLAG_NO_UPHEAD, CF_FLAG_NO_NECK, CF_FLAG_NO_DECODE_STREAM, CF_FLAG_STREAM_HIGH = 1, 2, 4, 8, 16, 32, 64
CF_EOVERFLOW = -6

# every symbol include/centerface_hip.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "cf_version", "cf_strerror", "cf_last_error", "cf_device_count", "cf_create", "cf_destroy",
    "cf_load_weights", "cf_forward", "cf_forward_resized", "cf_forward_images", "cf_upload_images", "cf_forward_uploaded", "cf_get_resized_input", "cf_get_heads", "cf_decode_topk", "cf_decode_topk_post", "cf_affine_from_center_scale", "cf_decode_threshold", "cf_decode_threshold_ex", "cf_decode_threshold_sized", "cf_decode_threshold_enqueue", "cf_set_rescale",
    "cf_detect_topk", "cf_synchronize", "cf_event_record", "cf_event_elapsed_ms",
    "cf_profile_forward", "cf_plan_size", "cf_plan_op", "cf_forward_trace", "cf_graph_stats", "cf_get_streams", "cf_streams_share_queue", "cf_streams_share_queue_ex", "cf_spread_streams", "cf_reroll_streams", "cf_ctdet_loss", "cf_comm_unique_id", "cf_comm_create", "cf_comm_create_all", "cf_comm_create_loopback", "cf_comm_loopback_rank", "cf_comm_destroy", "cf_comm_abort", "cf_comm_query", "cf_comm_synchronize", "cf_comm_last_error", "cf_comm_debug", "cf_comm_set_shard", "cf_comm_stream", "cf_gather_topk", "cf_host_alloc", "cf_host_free", "cf_pinned_alloc", "cf_pinned_free", "cf_host_register", "cf_host_unregister", "cf_device_alloc", "cf_device_free", "cf_memcpy_h2d", "cf_memcpy_d2h",
    "cf_op_last_error", "cf_op_shufflev2", "cf_op_mbconv", "cf_op_expand_dw", "cf_op_ctdet_loss", "cf_op_encode_targets", "cf_op_dwconv", "cf_op_pwconv", "cf_op_stem", "cf_op_idaup", "cf_op_heads",
    "cf_op_ctdet_decode", "cf_op_ctdet_post_process", "cf_op_decode_threshold", "cf_op_decode_threshold_ex", "cf_op_nms", "cf_op_box_match",
)


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32),
                ("dims", C.c_int64 * 4), ("dtype", C.c_int32)]


class OpTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("kind", C.c_char * 16), ("kernel", C.c_char * 160), ("ms", C.c_float),
                ("algo_bytes", C.c_double), ("flops", C.c_double)]


class OpInfo(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("kind", C.c_char * 16), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("fused_away", C.c_int32)]


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    res = subprocess.run(args, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libcenterface_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


_lib = None


def lib():
    """The loaded library; raises (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback path)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.cf_strerror.restype = C.c_char_p
        L.cf_last_error.restype = C.c_char_p
        L.cf_last_error.argtypes = [C.c_void_p]
        L.cf_op_last_error.restype = C.c_char_p
        L.cf_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
        L.cf_destroy.argtypes = [C.c_void_p]
        L.cf_load_weights.argtypes = [C.c_void_p, C.POINTER(TensorDesc), C.c_int]
        L.cf_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        if hasattr(L, "cf_forward_lanes"):              # experiments build only (csrc/cf_experiments.h)
            L.cf_forward_lanes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
            L.cf_forward_lanes_flush.argtypes = [C.c_void_p]
        L.cf_forward_resized.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.cf_get_resized_input.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.cf_get_heads.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.cf_decode_topk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.cf_decode_topk_post.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.cf_affine_from_center_scale.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.cf_op_ctdet_post_process.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.cf_decode_threshold_ex.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cf_decode_threshold_sized.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cf_decode_threshold_enqueue.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        L.cf_op_decode_threshold_ex.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cf_decode_threshold.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cf_detect_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.cf_synchronize.argtypes = [C.c_void_p]
        L.cf_event_record.argtypes = [C.c_void_p, C.c_int]
        L.cf_event_elapsed_ms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.cf_profile_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(OpTime), C.c_int, C.POINTER(C.c_int)]
        L.cf_plan_size.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.cf_plan_op.argtypes = [C.c_void_p, C.c_int, C.POINTER(OpInfo)]
        L.cf_forward_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.cf_ctdet_loss.argtypes = [C.c_void_p] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_void_p]
        L.cf_op_ctdet_loss.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_void_p]
        L.cf_op_encode_targets.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 8
        L.cf_comm_unique_id.argtypes = [C.c_void_p, C.c_int]
        L.cf_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.cf_comm_destroy.argtypes = [C.c_void_p]
        L.cf_comm_create_all.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
        L.cf_comm_create_loopback.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.cf_comm_loopback_rank.argtypes = [C.c_void_p, C.c_int]
        L.cf_comm_abort.argtypes = [C.c_void_p]
        L.cf_comm_query.argtypes = [C.c_void_p]
        L.cf_comm_synchronize.argtypes = [C.c_void_p]
        L.cf_comm_last_error.argtypes = [C.c_void_p]
        L.cf_comm_last_error.restype = C.c_char_p
        L.cf_comm_debug.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cf_comm_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cf_comm_stream.argtypes = [C.c_void_p]
        L.cf_comm_stream.restype = C.c_void_p
        L.cf_gather_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.cf_get_streams.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.cf_graph_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.cf_streams_share_queue.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.cf_streams_share_queue_ex.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.cf_spread_streams.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.cf_reroll_streams.argtypes = [C.c_void_p]
        L.cf_host_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        L.cf_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.cf_pinned_alloc.argtypes = [C.c_uint64, C.POINTER(C.c_void_p)]
        L.cf_pinned_free.argtypes = [C.c_void_p]
        L.cf_host_register.argtypes = [C.c_void_p, C.c_uint64]
        L.cf_host_unregister.argtypes = [C.c_void_p]
        L.cf_forward_images.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
        L.cf_upload_images.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
        L.cf_forward_uploaded.argtypes = [C.c_void_p]
        L.cf_set_rescale.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.cf_device_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        L.cf_device_free.argtypes = [C.c_void_p, C.c_void_p]
        L.cf_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.cf_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        fp, vp, i = C.c_void_p, C.c_void_p, C.c_int
        L.cf_op_dwconv.argtypes = [i, i, fp, fp, fp, fp] + [i] * 9
        L.cf_op_pwconv.argtypes = [i, i, fp, fp, fp, fp, fp] + [i] * 6
        L.cf_op_stem.argtypes = [i, i, vp, i, fp, fp, i, i, i]
        L.cf_op_shufflev2.argtypes = [i, i, fp, fp] + [i] * 8 + [fp] * 10
        L.cf_op_mbconv.argtypes = [i, i, fp, fp, fp, fp, fp] + [i] * 8
        L.cf_op_expand_dw.argtypes = [i, i, fp, fp, fp, fp] + [i] * 7
        L.cf_op_idaup.argtypes = [i, i, fp, fp, fp, fp, fp, fp, C.c_float, fp] + [i] * 5
        L.cf_op_heads.argtypes = [i, i, fp, fp, fp, fp, fp, fp, i, i, i, i]
        L.cf_op_ctdet_decode.argtypes = [i, fp, fp, fp, fp, i, i, i, i, fp, fp, vp]
        L.cf_op_decode_threshold.argtypes = [i, fp, fp, fp, i, i, i, i, i, C.c_float, C.c_float, i, fp, fp, vp]
        L.cf_op_nms.argtypes = [i, fp, fp, i, C.c_float, vp, vp]
        L.cf_op_box_match.argtypes = [i, i, fp, i, vp, fp, i, vp, C.c_float, vp, vp]
        _lib = L
    return _lib


def ptr(a):
    """Host pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class CenterFaceError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libcenterface_hip: %s (code %d)" % (text, code))
        self.code = code


def check(code, ctx=None, op=False):
    if code == CF_OK:
        return
    L = lib()
    detail = (L.cf_op_last_error() if op else L.cf_last_error(ctx)) or b""
    text = L.cf_strerror(code).decode()
    if detail:
        text += ": " + detail.decode(errors="replace")
    if code == -1:
        raise ValueError("libcenterface_hip: " + text)
    raise CenterFaceError(code, text)
