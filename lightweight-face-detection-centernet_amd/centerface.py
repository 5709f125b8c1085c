"""Drop-in host API: the reference's ``centerface.py`` surface on top of libcenterface_hip.so.

``CenterFace(height, width, landmarks=True)`` / ``__call__(img, threshold)`` / ``transform`` /
``decode`` / ``nms`` keep the reference's names, argument meaning, return types and quirks
(centerface.py:11-151); all arithmetic on the hot path runs in hand-written HIP kernels behind the
C ABI (``include/centerface_hip.h``).  This module only marshals numpy arrays; there is no CPU
compute fallback -- if the library is absent every entry point raises.

Additions over the reference (keyword-only, defaults reproduce the reference):
``weights=`` (checkpoint path or state_dict; default: deterministic synthetic weights because the
reference's ``weight/model_epoch_100.pt`` is not distributed), ``dtype=`` ('fp32' parity mode or
'bf16' throughput mode), ``device=``, ``max_batch=``, ``collapse_heads=``; methods ``forward``,
``detect_batch``, ``decode_topk`` (the ``ctdet_decode`` path of centerface_ext.py:52-82).
"""
import bisect
import ctypes as C
import os
import threading
import weakref

import numpy as np

from . import _lib
from . import weights as _weights

_DTYPES = {"fp32": _lib.CF_F32, "float32": _lib.CF_F32, "f32": _lib.CF_F32,
           "bf16": _lib.CF_BF16, "bfloat16": _lib.CF_BF16,
           # tolerance mode: fp32 storage, split-bf16 ("bf16x3") GEMM products -- within 1e-3 of the reference like "fp32"
           "fp32_split": _lib.CF_F32_SPLIT, "bf16x3": _lib.CF_F32_SPLIT}


class Engine(object):
    """One cf_ctx: one GPU, one stream, fixed (H, W), batch up to ``max_batch``."""

    def __init__(self, height, width, max_batch=1, dtype="fp32", device=0, weights=None,
                 collapse_heads=None, fuse=True, graph=True, uphead=True, neck=True, decode_stream=True, high_priority_streams=False):
        L = _lib.lib()
        if dtype not in _DTYPES:
            raise ValueError("dtype must be one of %s" % sorted(_DTYPES))
        self.H, self.W, self.max_batch, self.device = int(height), int(width), int(max_batch), int(device)
        self.h, self.w = self.H // 4, self.W // 4
        self.dtype = dtype
        if collapse_heads is None:
            # the head pair conv3x3+b -> conv1x1+b is linear (model/centernet.py:249-256): folding it (in float64)
            # into one 3x3 24->15 conv is exact algebra and 6x fewer flops -- the default in both modes; pass
            # collapse_heads=False for the two-stage kernel that keeps the reference's operation order
            collapse_heads = True
        flags = ((_lib.CF_FLAG_COLLAPSE_HEADS if collapse_heads else 0) | (0 if fuse else _lib.CF_FLAG_NO_FUSE)
                 | (0 if graph else _lib.CF_FLAG_NO_GRAPH) | (0 if uphead else _lib.CF_FLAG_NO_UPHEAD)
                 | (0 if neck else _lib.CF_FLAG_NO_NECK) | (0 if decode_stream else _lib.CF_FLAG_NO_DECODE_STREAM)
                 | (_lib.CF_FLAG_STREAM_HIGH if high_priority_streams else 0))
        handle = C.c_void_p()
        _lib.check(L.cf_create(self.device, self.max_batch, self.H, self.W, _DTYPES[dtype], flags, C.byref(handle)))
        self._h = handle
        self._L = L
        self.last_B = 0
        if weights is None:
            weights = _weights.synthetic_state_dict(0)
        elif isinstance(weights, str):
            weights = _weights.load_checkpoint(weights)
        self.load_state_dict(weights)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.cf_synchronize(self._h)
            for p in getattr(self, "_pinned", []):
                self._L.cf_host_free(self._h, p)
            self._pinned = []
            self._L.cf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, code):
        _lib.check(code, self._h)

    def load_state_dict(self, sd):
        """Strict load (centerface.py:24)."""
        sd = _weights.validate_state_dict(sd)
        descs = (_lib.TensorDesc * len(sd))()
        keep = []
        for i, (k, v) in enumerate(sd.items()):
            name = k.encode()
            keep.append((name, v))
            descs[i].name = name
            descs[i].data = v.ctypes.data
            descs[i].ndim = v.ndim
            for j, d in enumerate(v.shape):
                descs[i].dims[j] = d
            descs[i].dtype = 1 if v.dtype == np.int64 else 0
        self._chk(self._L.cf_load_weights(self._h, descs, len(sd)))

    # -- forward ----------------------------------------------------------------------------
    def forward_enqueue(self, x, on_device=False, B=None, in_format=None):
        """Enqueue one forward.  ``x``: uint8 [B,H,W,3] BGR or float32 [B,3,H,W] numpy array, or an
        int device pointer with ``on_device=True`` (then ``B`` and ``in_format`` are required)."""
        if on_device:
            p = C.c_void_p(int(x))
        else:
            x = np.ascontiguousarray(x)
            if x.dtype == np.uint8:
                if x.ndim != 4 or x.shape[1:] != (self.H, self.W, 3):
                    raise ValueError("uint8 input must be [B,%d,%d,3], got %s" % (self.H, self.W, x.shape))
                in_format = _lib.CF_IN_U8_HWC_BGR
            else:
                x = np.ascontiguousarray(x, dtype=np.float32)
                if x.ndim != 4 or x.shape[1:] != (3, self.H, self.W):
                    raise ValueError("float input must be [B,3,%d,%d], got %s" % (self.H, self.W, x.shape))
                in_format = _lib.CF_IN_F32_NCHW
            B = x.shape[0]
            self._keep_in = x
            p = _lib.ptr(x)
        self._chk(self._L.cf_forward(self._h, p, int(in_format), 1 if on_device else 0, int(B)))
        self.last_B = int(B)

    def forward_lanes_enqueue(self, prev, x, B, in_format=None):
        """Two-lane schedule (``cf_forward_lanes``; EXPERIMENTS BUILD of the library only -- ``make -C csrc EXP=1`` +
        ``CF_LIB=.../libcenterface_hip_exp.so``; measured slower than EngineRing): enqueue the front of a new batch (device pointer ``x``) on this
        engine and the back half of the batch pending on ``prev`` (another Engine of the same device, or None) underneath its
        mid-size blocks.  Afterwards ``prev`` holds a decodable result; this engine does not until it has been ``prev`` of a
        later call or ``forward_lanes_flush()`` was called."""
        fmt = _lib.CF_IN_U8_HWC_BGR if in_format is None else int(in_format)
        if not hasattr(self._L, "cf_forward_lanes"):
            raise RuntimeError("cf_forward_lanes exists only in an experiments build of the library (make -C csrc EXP=1)")
        self._chk(self._L.cf_forward_lanes(self._h, prev._h if prev is not None else None, C.c_void_p(int(x)), fmt, 1, int(B)))
        self.last_B = 0
        if prev is not None and getattr(prev, "_lane_B", 0):
            prev.last_B, prev._lane_B = prev._lane_B, 0
        self._lane_B = int(B)

    def forward_lanes_flush(self):
        self._chk(self._L.cf_forward_lanes_flush(self._h))
        if getattr(self, "_lane_B", 0):
            self.last_B, self._lane_B = self._lane_B, 0

    def forward_resized_enqueue(self, imgs_u8):
        """cv2.resize + forward (centerface.py:30-41): uint8 [B,h,w,3] BGR images of any (common) size are
        stretch-resized on the device to (H, W) and fed to the network."""
        x = np.ascontiguousarray(imgs_u8, dtype=np.uint8)
        if x.ndim != 4 or x.shape[3] != 3:
            raise ValueError("images must be uint8 [B,h,w,3], got %s" % (x.shape,))
        self._keep_in = x
        self._chk(self._L.cf_forward_resized(self._h, _lib.ptr(x), 0, x.shape[0], x.shape[1], x.shape[2]))
        self.last_B = x.shape[0]

    def forward_images_enqueue(self, images, upload_only=False):
        """The batch as a LIST of uint8 [h,w,3] BGR arrays of one common size (``cf_forward_images``): one asynchronous DMA per
        image, resize on the device when (h, w) is not the network size.  With page-locked arrays (``pinned_empty`` / ``pin``)
        nothing is copied on the host."""
        arrs = [np.asarray(im) for im in images]
        if not arrs:
            raise ValueError("forward_images_enqueue needs at least one image")
        shp = arrs[0].shape
        for a in arrs:
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape != shp or a.shape[2] != 3:
                raise ValueError("images must be uint8 [h,w,3] arrays of one common size, got %s %s (first: %s)" % (a.dtype, a.shape, shp))
        arrs = [a if a.flags["C_CONTIGUOUS"] else np.ascontiguousarray(a) for a in arrs]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        self._keep_in = arrs
        self._chk((self._L.cf_upload_images if upload_only else self._L.cf_forward_images)(self._h, ptrs, len(arrs), int(shp[0]), int(shp[1])))
        if upload_only:
            self._uploaded_B = len(arrs)
        else:
            self.last_B = len(arrs)

    def _upload_addrs(self, addrs, h, w, keep):
        """``upload_images`` for callers that have validated the batch already (CenterFaceBuckets: one raw size per chunk, page-locked
        uint8 arrays): the host addresses go straight into the pointer table."""
        self._keep_in = keep
        self._chk(self._L.cf_upload_images(self._h, (C.c_void_p * len(addrs))(*addrs), len(addrs), int(h), int(w)))
        self._uploaded_B = len(addrs)

    def upload_images(self, images):
        """First half of ``forward_images_enqueue`` (``cf_upload_images``): only the host -> device copies are enqueued."""
        self.forward_images_enqueue(images, upload_only=True)

    def forward_uploaded(self):
        """Second half (``cf_forward_uploaded``): resize + forward on what the last ``upload_images`` brought over."""
        self._chk(self._L.cf_forward_uploaded(self._h))
        self.last_B = self._uploaded_B

    def set_rescale(self, scale_h=0.0, scale_w=0.0):
        """centerface.py:55-62 on the device (``cf_set_rescale``): the threshold decodes of this engine return
        ``x // scale_w``, ``y // scale_h``; (0, 0) = off."""
        self._chk(self._L.cf_set_rescale(self._h, float(scale_h), float(scale_w)))

    def resized_input(self):
        """The resized uint8 [B,H,W,3] batch of the last forward_resized_enqueue (for tests)."""
        out = np.empty((self.last_B, self.H, self.W, 3), np.uint8)
        self._chk(self._L.cf_get_resized_input(self._h, _lib.ptr(out), self.last_B))
        return out

    def synchronize(self):
        self._chk(self._L.cf_synchronize(self._h))

    def heads(self, sigmoid_hm=False):
        """The four head maps of the last forward as NCHW float32 (model/centernet.py:277-280)."""
        B, h, w = self.last_B, self.h, self.w
        out = {"hm": np.empty((B, 1, h, w), np.float32), "wh": np.empty((B, 2, h, w), np.float32),
               "lm": np.empty((B, 10, h, w), np.float32), "reg": np.empty((B, 2, h, w), np.float32)}
        sg = np.empty((B, 1, h, w), np.float32) if sigmoid_hm else None
        self._chk(self._L.cf_get_heads(self._h, _lib.ptr(out["hm"]), _lib.ptr(out["wh"]), _lib.ptr(out["lm"]),
                                       _lib.ptr(out["reg"]), _lib.ptr(sg)))
        if sigmoid_hm:
            out["hm_sigmoid"] = sg
        return out

    def forward(self, x):
        """net(x)[0] (centerface.py:41): dict of hm (raw logits), wh, lm, reg."""
        self.forward_enqueue(x)
        return self.heads()

    # -- decode -----------------------------------------------------------------------------
    def decode_topk(self, K=100, use_reg=True, landmarks=True, post=None):
        """ctdet_decode on the last forward's heads: (dets [B,K,6], lms [B,K,10] | None, inds [B,K]).
        ``post=(c, s)`` (centers [B,2], scales [B] or [B,2]) additionally applies ctdet_post_process's
        coordinate mapping (utils/post_process.py:83-90) inside the decode kernel."""
        B = self.last_B
        dets = np.empty((B, K, 6), np.float32)
        lms = np.empty((B, K, 10), np.float32) if landmarks else None
        inds = np.empty((B, K), np.int64)
        if post is None:
            self._chk(self._L.cf_decode_topk(self._h, int(K), 1 if use_reg else 0, _lib.ptr(dets), _lib.ptr(lms),
                                             _lib.ptr(inds), 0))
        else:
            c = np.ascontiguousarray(np.asarray(post[0], np.float32).reshape(B, 2))
            s = np.asarray(post[1], np.float32)
            s = np.ascontiguousarray(np.repeat(s.reshape(B, 1), 2, axis=1) if s.size == B else s.reshape(B, 2))
            self._chk(self._L.cf_decode_topk_post(self._h, int(K), 1 if use_reg else 0, _lib.ptr(c), _lib.ptr(s),
                                                  self.w, self.h, _lib.ptr(dets), _lib.ptr(lms), _lib.ptr(inds), 0))
        return dets, lms, inds

    def decode_topk_device(self, K, dets_ptr, lms_ptr=None, inds_ptr=None, use_reg=True):
        """Same, writing into caller-owned DEVICE buffers (asynchronous)."""
        self._chk(self._L.cf_decode_topk(self._h, int(K), 1 if use_reg else 0, C.c_void_p(int(dets_ptr)),
                                         C.c_void_p(int(lms_ptr)) if lms_ptr else None,
                                         C.c_void_p(int(inds_ptr)) if inds_ptr else None, 1))

    def decode_threshold_enqueue(self, score_thresh=0.3, nms_thresh=0.3, max_out=1024, mode="d1", size=None):
        """Enqueue the threshold decode + NMS kernels of the last forward right behind it (no host wait); a following
        ``decode_threshold`` with the same arguments only waits and copies the results out (``cf_decode_threshold_enqueue``)."""
        ih, iw = (self.H, self.W) if size is None else (int(size[0]), int(size[1]))
        self._chk(self._L.cf_decode_threshold_enqueue(self._h, {"d1": 0, "d2": 1}[mode], float(score_thresh), float(nms_thresh), ih, iw, int(max_out)))

    def decode_threshold(self, score_thresh=0.3, nms_thresh=0.3, max_out=1024, mode="d1", size=None):
        """CenterFace.decode + nms on the last forward: list of (boxes [n,5], lms [n,10]) per image.
        mode "d2" = eval_widerface.decode (eval_widerface.py:92-110): threshold honoured, offsets used.
        ``size`` = (h, w) the boxes are clamped to (default: the engine's input size)."""
        ih, iw = (self.H, self.W) if size is None else (int(size[0]), int(size[1]))
        B = self.last_B
        while True:
            dets = np.empty((B, max_out, 5), np.float32)
            lms = np.empty((B, max_out, 10), np.float32)
            counts = np.empty((B,), np.int32)
            self._chk(self._L.cf_decode_threshold_sized(self._h, {"d1": 0, "d2": 1}[mode], float(score_thresh), float(nms_thresh),
                                                        ih, iw, int(max_out), _lib.ptr(dets), _lib.ptr(lms), _lib.ptr(counts)))
            if int(counts.max(initial=0)) <= max_out:
                break
            max_out = int(counts.max())          # more survivors than rows: the reference keeps them all, so do we
        return [(dets[b, :counts[b]].copy(), lms[b, :counts[b]].copy()) for b in range(B)]

    # -- launch plan / layer trace (parity tests) ---------------------------------------------
    def plan(self):
        """The context's launch plan: list of dicts name/kind/C/H/W/fused_away."""
        n = C.c_int()
        self._chk(self._L.cf_plan_size(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            info = _lib.OpInfo()
            self._chk(self._L.cf_plan_op(self._h, i, C.byref(info)))
            out.append(dict(index=i, name=info.name.decode(), kind=info.kind.decode(), C=info.C, H=info.H, W=info.W,
                            fused_away=bool(info.fused_away)))
        return out

    def trace(self, x, op_index):
        """Run the forward up to plan entry ``op_index`` and return that entry's output as NCHW float32."""
        x = np.ascontiguousarray(x)
        in_format = _lib.CF_IN_U8_HWC_BGR if x.dtype == np.uint8 else _lib.CF_IN_F32_NCHW
        if x.dtype != np.uint8:
            x = np.ascontiguousarray(x, dtype=np.float32)
        op = self.plan()[op_index]
        out = np.empty((x.shape[0], op["C"], op["H"], op["W"]), np.float32)
        self._chk(self._L.cf_forward_trace(self._h, _lib.ptr(x), in_format, 0, x.shape[0], int(op_index), _lib.ptr(out)))
        self.last_B = x.shape[0] if op_index == len(self.plan()) - 1 else 0
        return out

    # -- timing -----------------------------------------------------------------------------
    def event_record(self, slot):
        self._chk(self._L.cf_event_record(self._h, int(slot)))

    def event_elapsed_ms(self, a, b):
        ms = C.c_float()
        self._chk(self._L.cf_event_elapsed_ms(self._h, int(a), int(b), C.byref(ms)))
        return ms.value

    def profile_forward(self, x, on_device=False, B=None, in_format=None, K=0):
        """Per-kernel times of one forward: list of dicts name/kind/ms/algo_bytes/flops."""
        if on_device:
            p = C.c_void_p(int(x))
        else:
            x = np.ascontiguousarray(x)
            in_format = _lib.CF_IN_U8_HWC_BGR if x.dtype == np.uint8 else _lib.CF_IN_F32_NCHW
            B = x.shape[0]
            p = _lib.ptr(x)
        rec = (_lib.OpTime * 64)()
        n = C.c_int()
        self._chk(self._L.cf_profile_forward(self._h, p, int(in_format), 1 if on_device else 0, int(B), int(K),
                                             rec, 64, C.byref(n)))
        self.last_B = int(B)
        return [dict(name=r.name.decode(), kind=r.kind.decode(), kernel=r.kernel.decode(), ms=r.ms, algo_bytes=r.algo_bytes, flops=r.flops)
                for r in rec[:n.value]]

    def streams(self):
        """(main, decode) hipStream_t handles as ints, e.g. for torch.cuda.ExternalStream."""
        a, b = C.c_void_p(), C.c_void_p()
        self._chk(self._L.cf_get_streams(self._h, C.byref(a), C.byref(b)))
        return a.value or 0, b.value or 0

    def shares_queue_with(self, other):
        """True when this context's main stream waits behind ``other``'s (same hardware queue).  Both must be idle."""
        sh = C.c_int(0)
        self._chk(self._L.cf_streams_share_queue(other._h, self._h, C.byref(sh)))
        return bool(sh.value)

    def queue_shared(self, which, other, which_other):
        """The same probe for any pair of streams (``cf_streams_share_queue_ex``): which = 0 main, 1 decode, 2 the device's copy stream."""
        sh = C.c_int(0)
        self._chk(self._L.cf_streams_share_queue_ex(other._h, int(which_other), self._h, int(which), C.byref(sh)))
        return bool(sh.value)

    def reroll_streams(self):
        """Replace the main and decode streams by newly created ones (cf_reroll_streams).  The context must be idle."""
        self._chk(self._L.cf_reroll_streams(self._h))

    def graph_stats(self):
        """(captured forward graphs, keys that fell back to eager launches)."""
        a, b = C.c_int(), C.c_int()
        self._chk(self._L.cf_graph_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pinned_array(self, shape, dtype=np.uint8):
        """A numpy array in page-locked host memory (cf_host_alloc): forward_enqueue / forward_resized_enqueue from it
        are asynchronous DMA copies.  Freed with the engine (close)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._chk(self._L.cf_host_alloc(self._h, max(n, 1), C.byref(p)))
        if not hasattr(self, "_pinned"):
            self._pinned = []
        self._pinned.append(p)
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def pinned_raw(self, nbytes):
        """(pointer, capacity) of a fresh page-locked host buffer owned by the engine (freed at close or by pinned_free)."""
        p = C.c_void_p()
        self._chk(self._L.cf_host_alloc(self._h, max(int(nbytes), 1), C.byref(p)))
        if not hasattr(self, "_pinned"):
            self._pinned = []
        self._pinned.append(p)
        return p, int(nbytes)

    def pinned_free(self, p):
        """Release one buffer of pinned_array / pinned_raw now (no copy from it may be in flight)."""
        self._pinned = [q for q in getattr(self, "_pinned", []) if q.value != p.value]
        self._L.cf_host_free(self._h, p)

    def device_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self._L.cf_device_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def device_free(self, p):
        self._chk(self._L.cf_device_free(self._h, C.c_void_p(int(p))))

    def memcpy_h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self._L.cf_memcpy_h2d(self._h, C.c_void_p(int(dptr)), _lib.ptr(arr), arr.nbytes))

    def memcpy_d2h(self, arr, dptr):
        """Blocking copy of ``arr.nbytes`` device bytes at ``dptr`` into the C-contiguous numpy array ``arr``."""
        if not arr.flags["C_CONTIGUOUS"]:
            raise ValueError("memcpy_d2h needs a C-contiguous destination")
        self._chk(self._L.cf_memcpy_d2h(self._h, _lib.ptr(arr), C.c_void_p(int(dptr)), arr.nbytes))


class EngineRing(object):
    """``depth`` independent contexts of one geometry used round-robin: batch i runs on context i % depth.

    Inside one context a forward is a chain on one stream.  Its back half -- the project GEMMs on the 40x40 / 20x20 maps
    (HBM-bound, small grids), the up3+heads kernel (memory-latency-bound) and the decode -- leaves the VALU idle, while
    its front half (stem, layer1.x / 2.x: VALU-issue-bound on Swish) leaves HBM idle.  With two batches in flight on two
    contexts (each has its own main / copy / decode streams and buffers) the GPU overlaps the back half of batch i with
    the front half of batch i+1: 44k -> 48k img/s at 64 x 640x640 bf16 (tools/dual_stream_probe.py; depth 3 adds
    nothing).  The reference has no counterpart (centerface.py:39-48 is one synchronous call per image); a C host does
    the same with two cf_ctx handles.  Contexts are independent: results of batch i are on context
    i % depth until batch i + depth is submitted.

        ring = EngineRing(640, 640, max_batch=64, dtype="bf16")
        t0 = ring.submit(batch0, K=100); t1 = ring.submit(batch1, K=100)
        dets, lms, inds = ring.collect(t0)          # waits for batch0 only
    """

    def __init__(self, height, width, depth=2, placement=None, **engine_kwargs):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        # three or more contexts (small batches: BASELINE configs[4] shards of four 1280x1280 images): six streams on HIP's four hardware
        # queues serialise more than they overlap -- the decodes stay on the main streams (CF_FLAG_NO_DECODE_STREAM): 9.5 -> 12.2 k img/s at depth 3
        engine_kwargs.setdefault("decode_stream", depth < 3)
        # Where the ring's streams land decides whether its contexts overlap at all.  HIP folds a process's streams onto four hardware queues
        # PER PRIORITY CLASS and gives a new stream the least-used queue of its class; two main streams on one queue (or on two queues of one
        # dispatch pipe) run their forwards strictly one after the other (48.2 k img/s instead of 54.4 k at 64 x 640x640, and which of the two a
        # ring got depended on every stream the process had created before).
        #   placement="probe" (default): default-priority streams, every pair tested with spin kernels (cf_streams_share_queue_ex, + 16 = the
        #     dispatch-pipe form), cf_spread_streams re-places them all when two clash: 51.0-54.5 k in EVERY process history measured
        #     (tools/queue_order_probe.py, tools/ring_sequence_probe.py), ~15-30 ms at creation.
        #   placement="priority" (round 6): the ring's streams are created in the HIGHEST priority class (CF_FLAG_STREAM_HIGH) and nothing is
        #     probed.  A class of its own is proof against everything the process did in the DEFAULT class -- 54.3-54.5 k with an identical queue
        #     map after 0 / 1 / 2 / 3 / 5 dummy streams or live plain contexts, where default-class streams as created give 48.2-54.4 k -- but not
        #     against this library's own earlier use of the highest class: the device's copy stream lives there, five streams share four queues,
        #     and which two share depends on what was created AND DESTROYED before (a ring created after two plain contexts were closed: 47.5 k;
        #     after two single-context rings: 45.7-47.8 k; with the copy stream in the lowest class instead: 51.7 k always, 4 % under the best).
        #     For a server that creates its ring first and keeps it: the same rate as "probe" with no probe kernel ever launched.
        #   placement="none": streams as the runtime creates them.
        if placement is None:
            placement = os.environ.get("CF_RING_PLACEMENT") or "probe"
        if os.environ.get("CF_RING_PLACE", "1") == "0":        # (round-5 switch: streams as created)
            placement = "none"
        if placement not in ("priority", "probe", "none"):
            raise ValueError("placement must be 'priority', 'probe' or 'none'")
        if placement == "priority":
            engine_kwargs.setdefault("high_priority_streams", True)
        self.placement = placement
        self.engines = [Engine(height, width, **engine_kwargs) for _ in range(int(depth))]
        self.queue_rerolls = 0
        if placement == "probe":
            which = (0, 1) if engine_kwargs["decode_stream"] else (0,)
            streams = [(e, w) for e in self.engines for w in which]
            # (+ 16: the dispatch-pipe probe; two DECODE streams on one pipe are harmless and sometimes all the process's queues allow)
            clash = any(a[0].queue_shared(a[1] + 16, b[0], b[1]) for k, a in enumerate(streams) for b in streams[:k] if not (a[1] == 1 and b[1] == 1))
            if clash:
                hs = (C.c_void_p * len(self.engines))(*[e._h for e in self.engines])
                nd = C.c_int(0)
                self.engines[0]._chk(self.engines[0]._L.cf_spread_streams(hs, len(self.engines), 0, C.byref(nd)))
                self.queue_rerolls = 1
        self.depth = int(depth)
        self._n = 0
        self._out = [None] * self.depth               # per slot: (K, dets_ptr, lms_ptr, inds_ptr, B)

    def next_engine(self):
        """The context the next batch goes to (advances the ring): for callers that drive Engine directly."""
        e = self.engines[self._n % self.depth]
        self._n += 1
        return e

    def submit(self, x, K=100, on_device=False, B=None, in_format=None, use_reg=True):
        """Enqueue forward + top-K decode of one batch on the next context; returns a ticket for ``collect``."""
        slot = self._n % self.depth
        e = self.next_engine()
        e.forward_enqueue(x, on_device=on_device, B=B, in_format=in_format)
        o = self._out[slot]
        if o is None or o[0] < K:
            if o is not None:
                e.synchronize()
                for q in o[4]:
                    e.pinned_free(q)
            nb = e.max_batch
            # the decode's "device" outputs are page-locked HOST buffers (device-visible): the select kernel writes the B x K rows over
            # PCIe itself, and collect() is a wait + three numpy copies instead of three blocking device -> pageable transfers
            raw = [e.pinned_raw(nb * K * 6 * 4)[0], e.pinned_raw(nb * K * 10 * 4)[0], e.pinned_raw(nb * K * 8)[0]]
            o = (int(K), raw[0].value, raw[1].value, raw[2].value, raw)
        self._out[slot] = o
        e.decode_topk_device(K, o[1], o[2], o[3], use_reg=use_reg)
        return (slot, int(K), e.last_B)

    def collect(self, ticket):
        """(dets [B,K,6], lms [B,K,10], inds [B,K]) of a submitted batch (host arrays; waits for that context only)."""
        slot, K, B = ticket
        e, o = self.engines[slot], self._out[slot]
        # the decode wrote compact [B][K] rows (the K of its call) at the start of the slot's buffers
        e.synchronize()

        def host(addr, shape, dtype):
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_char * n).from_address(addr), dtype=dtype).reshape(shape).copy()
        return host(o[1], (B, K, 6), np.float32), host(o[2], (B, K, 10), np.float32), host(o[3], (B, K), np.int64)

    def load_state_dict(self, sd):
        for e in self.engines:
            e.load_state_dict(sd)

    def synchronize(self):
        for e in self.engines:
            e.synchronize()

    def close(self):
        for slot, e in enumerate(self.engines):
            self._out[slot] = None                # (the page-locked output buffers belong to the engine: freed with it)
            e.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CenterFace(object):
    """Same construction and call surface as the reference class (centerface.py:11-66)."""
    mean = np.array([0.408, 0.447, 0.470], dtype=np.float32).reshape(1, 1, 3)   # centerface.py:12-13
    std = np.array([0.289, 0.274, 0.278], dtype=np.float32).reshape(1, 1, 3)    # centerface.py:14-15

    def __init__(self, height, width, landmarks=True, *, weights=None, dtype="fp32", device=0,
                 max_batch=1, collapse_heads=None, nms_thresh=0.3, max_dets=1024):
        self.landmarks = landmarks
        self.img_h_new, self.img_w_new, self.scale_h, self.scale_w = self.transform(height, width)
        self.nms_thresh = nms_thresh
        self.max_dets = max_dets
        self.device = device
        # (threshold decodes run on the main stream: these contexts never need a decode stream of their own)
        self._engine_kw = dict(max_batch=max_batch, dtype=dtype, device=device, weights=weights, collapse_heads=collapse_heads, decode_stream=False)
        self.engine = Engine(self.img_h_new, self.img_w_new, **self._engine_kw)
        self._engine2 = None                      # second context of detect_stream, created on first use

    def close(self):
        self.engine.close()
        if self._engine2 is not None:
            self._engine2.close()
            self._engine2 = None

    # centerface.py:68-71
    def transform(self, h, w):
        img_h_new, img_w_new = int(np.ceil(h / 32) * 32), int(np.ceil(w / 32) * 32)
        scale_h, scale_w = img_h_new / h, img_w_new / w
        return img_h_new, img_w_new, scale_h, scale_w

    @staticmethod
    def _floordiv(a, scale):
        """``a // scale`` for a float32 array and a Python float exactly as numpy evaluates it (float32 operands, result the
        mathematically exact floor of the quotient), computed as floor of the float64 quotient: two float32 values whose
        ratio is not an integer differ from it by far more than a double's rounding, so the results are identical
        (tests/test_abi.py::test_fast_floor_division_equals_numpy_floor_divide, adversarial near-multiples included) --
        and numpy's own float32 floor_divide (an fmod per element) is 12x slower, which made this rescale half of the
        host time of a VGA-bucket batch."""
        return np.floor(a.astype(np.float64) / np.float64(np.float32(scale))).astype(np.float32)

    def _postprocess(self, dets, lms, rescaled=False):
        # centerface.py:55-62: floor-division rescale (``rescaled``: already done by the decode kernel, Engine.set_rescale),
        # empty -> [0,5] / [0,10]
        if len(dets) > 0 and rescaled:
            pass
        elif len(dets) > 0:
            dets[:, 0:4:2], dets[:, 1:4:2] = self._floordiv(dets[:, 0:4:2], self.scale_w), self._floordiv(dets[:, 1:4:2], self.scale_h)
            if self.landmarks:
                lms[:, 0:10:2], lms[:, 1:10:2] = self._floordiv(lms[:, 0:10:2], self.scale_w), self._floordiv(lms[:, 1:10:2], self.scale_h)
        else:
            dets = np.empty(shape=[0, 5], dtype=np.float32)
            if self.landmarks:
                lms = np.empty(shape=[0, 10], dtype=np.float32)
        return (dets, lms) if self.landmarks else dets

    def _postprocess_many(self, results, rescaled=False):
        """_postprocess for a list of (dets, lms) that share one (scale_h, scale_w): ONE floor division over the
        concatenated rows (elementwise, so identical to per-image calls), then split again.  ``rescaled``: the decode kernel
        has divided already (Engine.set_rescale) -- only the reference's empty-result shapes are left to do."""
        if rescaled:
            return [self._postprocess(d, l, True) for d, l in results]
        n = [len(d) for d, _ in results]
        if sum(n) == 0:
            return [self._postprocess(d, l) for d, l in results]
        dets = np.concatenate([d for d, _ in results if len(d)])
        lms = np.concatenate([l for d, l in results if len(d)])
        dets, lms = self._postprocess(dets, lms) if self.landmarks else (self._postprocess(dets, lms), None)
        out, o = [], 0
        for k in n:
            if k == 0:
                out.append(self._postprocess(np.empty((0, 5), np.float32), np.empty((0, 10), np.float32)))
            else:
                out.append((dets[o:o + k], lms[o:o + k]) if self.landmarks else dets[o:o + k])
            o += k
        return out

    def __call__(self, img, threshold=0.2):
        """img: BGR uint8 HWC (what cv2.imread returns).  Returns (dets [N,5], lms [N,10]) or dets."""
        return self.detect_batch([img], threshold)[0]

    def detect_batch(self, imgs, threshold=0.2):
        """Batched ``__call__`` (the shape of eval_widerface.get_detections, :76-90).  The reference's
        decode ignores ``threshold`` and uses 0.3 (centerface.py:77); so does this."""
        del threshold
        imgs = [np.asarray(im, dtype=np.uint8) for im in imgs]
        direct = all(is_pinned(im) for im in imgs)                           # page-locked images: one DMA each, no host copy
        batch = imgs if direct else np.stack(imgs)                           # one common (h, w) per instance
        identity = batch[0].shape[:2] == (self.img_h_new, self.img_w_new)
        out = []
        self.engine.set_rescale(self.scale_h, self.scale_w)                  # centerface.py:55-62 inside the decode kernel
        try:
            for i in range(0, len(batch), self.engine.max_batch):
                chunk = batch[i:i + self.engine.max_batch]
                if direct:
                    self.engine.forward_images_enqueue(chunk)
                elif identity:
                    self.engine.forward_enqueue(chunk)
                else:
                    self.engine.forward_resized_enqueue(chunk)               # cv2.resize stand-in, on the device
                out.extend(self._postprocess_many(self.engine.decode_threshold(0.3, self.nms_thresh, self.max_dets), rescaled=True))
        finally:
            self.engine.set_rescale(0.0, 0.0)
        return out

    def detect_stream(self, imgs, threshold=0.2):
        """``__call__`` over an iterable of same-sized images, results yielded in order (the loop of demo.py:30-38 /
        eval_widerface.py:76-90).  The reference runs one synchronous call per image; here two contexts alternate, the
        forward of chunk i+1 (``max_batch`` images) is enqueued before the host waits for the decode of chunk i, so the GPU
        works on one chunk while the host decodes and rescales the other.  Same results as ``__call__`` (tests)."""
        del threshold
        if self._engine2 is None:
            self._engine2 = Engine(self.img_h_new, self.img_w_new, **self._engine_kw)
            if self._engine2.shares_queue_with(self.engine):     # keep the two main streams on different hardware queues (see EngineRing)
                hs = (C.c_void_p * 2)(self.engine._h, self._engine2._h)
                self.engine._chk(self.engine._L.cf_spread_streams(hs, 2, 0, None))
        engs = (self.engine, self._engine2)
        nb = self.engine.max_batch

        def enqueue(e, chunk):
            chunk = [np.asarray(im, dtype=np.uint8) for im in chunk]
            e.set_rescale(self.scale_h, self.scale_w)
            try:
                if all(is_pinned(im) for im in chunk):
                    return e.forward_images_enqueue(chunk)
                batch = np.stack(chunk)
                if batch.shape[1:3] == (self.img_h_new, self.img_w_new):
                    e.forward_enqueue(batch)
                else:
                    e.forward_resized_enqueue(batch)
            except BaseException:
                e.set_rescale(0.0, 0.0)                  # the rescale is per-context state: a failed enqueue must not leave it behind
                raise

        def finish(e):
            try:
                return self._postprocess_many(e.decode_threshold(0.3, self.nms_thresh, self.max_dets), rescaled=True)
            finally:
                e.set_rescale(0.0, 0.0)
        it, k, pending = iter(imgs), 0, None
        while True:
            chunk = []
            for im in it:
                chunk.append(im)
                if len(chunk) == nb:
                    break
            if chunk:
                enqueue(engs[k & 1], chunk)
            if pending is not None:
                for r in finish(pending):
                    yield r
            if not chunk:
                return
            pending = engs[k & 1]
            k += 1

    def forward(self, x):
        return self.engine.forward(x)

    def decode_topk(self, K=100):
        return self.engine.decode_topk(K)

    # centerface.py:73-109 on explicit arrays
    def decode(self, heatmap, scale, offset, landmark, size, threshold=0.1):
        del offset, threshold       # read but unused / ignored by the reference (:77,:86-88)
        L = _lib.lib()
        hm = _lib.f32(np.asarray(heatmap).reshape((1, 1) + np.squeeze(heatmap).shape))
        h, w = hm.shape[2:]
        wh = _lib.f32(scale)
        lm = _lib.f32(landmark) if landmark is not None else np.zeros((1, 10, h, w), np.float32)
        cap = max(1, min(h * w, 4096))
        while True:
            dets = np.empty((1, cap, 5), np.float32)
            lms = np.empty((1, cap, 10), np.float32)
            cnt = np.zeros((1,), np.int32)
            _lib.check(L.cf_op_decode_threshold(self.device, _lib.ptr(hm), _lib.ptr(wh), _lib.ptr(lm), 1, h, w,
                                                int(size[0]), int(size[1]), 0.3, float(self.nms_thresh), cap,
                                                _lib.ptr(dets), _lib.ptr(lms), _lib.ptr(cnt)), op=True)
            if int(cnt[0]) <= cap:
                break
            cap = int(cnt[0])
        n = int(cnt[0])
        if n == 0:
            return ([], []) if self.landmarks else []
        return (dets[0, :n].copy(), lms[0, :n].copy()) if self.landmarks else dets[0, :n].copy()

    # centerface.py:111-151
    def nms(self, boxes, scores, nms_thresh):
        L = _lib.lib()
        boxes, scores = _lib.f32(boxes), _lib.f32(scores)
        n = boxes.shape[0]
        keep = np.empty((max(n, 1),), np.int32)
        nk = C.c_int32()
        _lib.check(L.cf_op_nms(self.device, _lib.ptr(boxes), _lib.ptr(scores), n, float(nms_thresh),
                               _lib.ptr(keep), C.byref(nk)), op=True)
        return [int(k) for k in keep[:nk.value]]


# ---- page-locked caller memory ------------------------------------------------------------------------------------------
# Round 6.  Round 5 page-locked numpy HEAP memory in place (hipHostRegister on `im.copy()`, and on an aligned window of an np.empty
# array for pinned_empty).  That is what killed the driver's GPU run of round 5: the C library grows, trims and reuses its heap
# underneath a live registration, the registration's pages stop being the ones the GPU was given, and a later DMA from the array
# faults ("Memory access fault by GPU ... on address <heap address>": SIGABRT from the HSA runtime's event thread, no Python
# exception possible).  tools/diag/pin_churn_probe.py reproduces it in under a minute of allocation churn: 6 faults in 22 runs from heap
# memory, 1 in 6 even from whole pages inside a heap array, none from mmap regions or hipHostMalloc memory.  Hence:
#   * pinned_empty / pinned_copy hand out hipHostMalloc memory (cf_pinned_alloc) -- the library never registers numpy's heap;
#   * pin() only accepts whole pages of a mapping of its own (np.memmap / mmap / shared memory: a frame pool) and refuses the heap.
_pin_lock = threading.Lock()
_pin_bases, _pin_sizes = [], {}          # sorted base addresses / base -> bytes, of every page-locked range handed out or registered
_pin_ids = {}                            # id(array object handed out by pin / pinned_empty) -> its address (the per-image fast path)
_PAGE = 4096


def _track(addr, nbytes):
    with _pin_lock:
        bisect.insort(_pin_bases, addr)
        _pin_sizes[addr] = nbytes


def _untrack(addr):
    with _pin_lock:
        if _pin_sizes.pop(addr, None) is None:
            return False
        _pin_bases.remove(addr)
        return True


def _unregister(addr):
    if not _untrack(addr):
        return
    try:
        _lib.lib().cf_host_unregister(C.c_void_p(addr))
    except Exception:                                              # noqa: BLE001  (interpreter shutdown)
        pass


def _owner(arr):
    """The ndarray at the end of ``arr``'s .base chain: the object whose lifetime bounds the memory ``arr`` views."""
    o = arr
    while isinstance(o.base, np.ndarray):
        o = o.base
    return o


def _heap_range():
    """[lo, hi) of the C library's main heap (the ``[heap]`` line of /proc/self/maps), or None."""
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                if ln.rstrip().endswith("[heap]"):
                    lo, hi = ln.split()[0].split("-")
                    return int(lo, 16), int(hi, 16)
    except OSError:
        pass
    return None


def pin(arr):
    """Page-lock the memory of a C-contiguous numpy array IN PLACE (``cf_host_register``) and return it: batches taken from it
    reach the GPU by asynchronous DMA with no staging copy (``Engine.forward_images_enqueue``, ``CenterFaceBuckets``).  Only for
    memory that is a mapping of its own -- ``np.memmap``, ``mmap.mmap``, a shared-memory segment: a decoder's frame pool -- in whole
    pages: the address a multiple of 4096, the size too.  Arrays from ``np.empty`` / ``.copy()`` / ``cv2.imread`` live on the C
    library's heap and are REFUSED (ValueError): registering heap memory faults the GPU once the heap has been trimmed and regrown
    (the abort of GPUTEST_r05; see the comment above) -- copy such images into ``pinned_empty`` / ``pinned_copy`` buffers instead.
    ``unpin`` releases the registration; so does the garbage collection of the array that owns the memory (a finalizer on the
    owner unregisters the range before the mapping can go away)."""
    if not isinstance(arr, np.ndarray) or not arr.flags["C_CONTIGUOUS"] or arr.nbytes == 0:
        raise ValueError("pin needs a non-empty C-contiguous numpy array")
    addr, nbytes = arr.ctypes.data, arr.nbytes
    if addr % _PAGE or nbytes % _PAGE:
        raise ValueError("pin needs whole pages of a mapping of its own (address and size multiples of %d; got 0x%x, %d bytes): "
                         "numpy heap arrays cannot be page-locked in place -- use pinned_empty / pinned_copy" % (_PAGE, addr, nbytes))
    heap = _heap_range()
    if heap is not None and addr < heap[1] and addr + nbytes > heap[0]:
        raise ValueError("pin: the array lives on the C library's heap; page-locking heap memory in place faults the GPU when the heap "
                         "is trimmed and regrown -- use pinned_empty / pinned_copy")
    with _pin_lock:
        if addr in _pin_sizes:
            if _pin_sizes[addr] >= nbytes:
                return arr
            raise ValueError("a shorter range at the same address is already pinned")
        k = bisect.bisect_right(_pin_bases, addr) - 1
        lo_clash = k >= 0 and _pin_bases[k] + _pin_sizes[_pin_bases[k]] > addr
        hi_clash = k + 1 < len(_pin_bases) and _pin_bases[k + 1] < addr + nbytes
        if lo_clash or hi_clash:
            raise ValueError("the array overlaps a range that is already pinned (pin the enclosing array once)")
    owner = _owner(arr)
    _lib.check(_lib.lib().cf_host_register(C.c_void_p(addr), nbytes), op=True)
    _track(addr, nbytes)
    weakref.finalize(owner, _unregister, addr)                     # (a no-op after unpin: _unregister forgets the address first)
    _remember(arr)
    return arr


def _remember(arr):
    key = id(arr)
    _pin_ids[key] = arr.ctypes.data
    weakref.finalize(arr, _pin_ids.pop, key, None)


def _direct_addr(im):
    """Address of a page-locked C-contiguous uint8 array, else None (one dict probe for the objects pin / pinned_empty returned)."""
    if type(im) is not np.ndarray or im.dtype != np.uint8:
        return None
    addr = _pin_ids.get(id(im))
    if addr is not None and im.flags.c_contiguous:
        return addr
    return im.ctypes.data if is_pinned(im) else None


def unpin(arr):
    """Release a ``pin`` registration (``pinned_empty`` memory is released with its last view instead)."""
    _pin_ids.pop(id(arr), None)
    _unregister(arr.ctypes.data)


class _PinnedBlock(object):
    """One hipHostMalloc allocation (``cf_pinned_alloc``) exposed through the array interface; freed when the last numpy view of it
    is gone (numpy keeps this object as the .base of every view)."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _lib.check(_lib.lib().cf_pinned_alloc(int(nbytes), C.byref(p)), op=True)
        self.addr, self.nbytes = p.value, int(nbytes)
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.addr, False), "version": 3}
        _track(self.addr, self.nbytes)

    def __del__(self):
        addr, self.addr = getattr(self, "addr", None), None
        if addr:
            _untrack(addr)
            try:
                _lib.lib().cf_pinned_free(C.c_void_p(addr))
            except Exception:                                      # noqa: BLE001  (interpreter shutdown)
                pass


def pinned_empty(shape, dtype=np.uint8):
    """``np.empty`` in page-locked host memory (hipHostMalloc through ``cf_pinned_alloc``; released when the array and its views are
    gone): images written here reach the GPU by asynchronous DMA with no staging copy."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    block = _PinnedBlock(max(n, 1))
    out = np.asarray(block)[:n].view(dtype).reshape(shape)
    _remember(out)
    return out


def pinned_copy(arr):
    """A copy of ``arr`` in page-locked memory (``pinned_empty`` + one host copy): for images that arrive in pageable memory
    (``cv2.imread``, a decoder) and are used more than once."""
    arr = np.asarray(arr)
    out = pinned_empty(arr.shape, arr.dtype)
    np.copyto(out, arr)
    return out


def is_pinned(arr):
    """True when the array's bytes lie inside a range registered by ``pin`` or handed out by ``pinned_empty``."""
    if not isinstance(arr, np.ndarray) or not arr.flags["C_CONTIGUOUS"]:
        return False
    addr = arr.ctypes.data
    with _pin_lock:
        k = bisect.bisect_right(_pin_bases, addr) - 1
        return k >= 0 and addr + arr.nbytes <= _pin_bases[k] + _pin_sizes[_pin_bases[k]]


_copy_pool, _copy_pool_lock = None, threading.Lock()


def _stage_copy_begin(stage, images, background=False):
    """Host images -> rows of a page-locked staging array; returns a function that waits for the copy.  The copy is the largest
    host cost of a mixed-size batch (118 MB for 128 VGA images: 5 of 9 ms on one core).  numpy releases the GIL while it copies,
    so a few worker threads (one task per worker: a submit costs ~25 us of Python) move it faster -- and, with ``background``,
    entirely off the calling thread, which collects the results of an earlier chunk meanwhile.  Small jobs are copied at once."""
    global _copy_pool
    total = sum(int(np.asarray(im).nbytes) for im in images)
    workers = min(int(os.environ.get("CF_STAGE_THREADS", "4")), os.cpu_count() or 1, len(images))
    if total < (4 << 20) or workers < 2:
        for k, im in enumerate(images):
            np.copyto(stage[k], np.asarray(im, dtype=np.uint8))
        return lambda: None
    with _copy_pool_lock:
        if _copy_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            _copy_pool = ThreadPoolExecutor(max_workers=max(2, min(int(os.environ.get("CF_STAGE_THREADS", "4")), os.cpu_count() or 1)),
                                            thread_name_prefix="cf-stage")

    def copy_slice(k0):
        for k in range(k0, len(images), workers):
            np.copyto(stage[k], np.asarray(images[k], dtype=np.uint8))
    futs = [_copy_pool.submit(copy_slice, k0) for k0 in range(0 if background else 1, workers)]
    if not background:
        copy_slice(0)                                              # the calling thread takes a share

    def wait():
        for f in futs:
            f.result()                                             # re-raises a worker's exception (shape mismatch, ...)
    return wait


class CenterFaceBuckets(object):
    """Variable-size input (BASELINE configs[3]: WIDER-style images of different shapes in one batch).

    The reference builds one ``CenterFace(h, w)`` per image shape (demo.py:76 even per image).  Here images are
    bucketed by their NETWORK shape ``transform(h, w)[:2]`` (multiples of 32, centerface.py:68-71): every bucket
    owns one ``Engine`` (one cf_ctx on the GPU, created on first use, all sharing the same weights, graphs
    reused), so all raw sizes that round up to the same network shape share a context.  Inside a bucket the
    images of one raw size run as full ``forward_resized`` batches (the device resize takes one source size per
    launch); the floor-division rescale uses each image's own ``scale_h`` / ``scale_w``.  Results come back in
    the order of ``imgs``; every image gets exactly what ``CenterFace(h, w)(img)`` returns for it."""

    def __init__(self, landmarks=True, *, weights=None, dtype="fp32", device=0, max_batch=32, max_buckets=8,
                 nms_thresh=0.3, max_dets=1024, collapse_heads=None):
        self.landmarks, self.dtype, self.device, self.max_batch, self.max_buckets = landmarks, dtype, device, max_batch, max_buckets
        self.nms_thresh, self.max_dets, self.collapse_heads = nms_thresh, max_dets, collapse_heads
        self._weights = _weights.synthetic_state_dict(0) if weights is None else (
            _weights.load_checkpoint(weights) if isinstance(weights, str) else weights)
        self._buckets = {}          # network (H, W) -> Engine, in least-recently-used order
        self.created = 0            # contexts created so far (tests: raw sizes sharing a network shape share one)

    def _engine(self, H, W):
        key = (int(H), int(W))
        eng = self._buckets.pop(key, None)
        if eng is None:
            if len(self._buckets) >= self.max_buckets:             # evict the least recently used context
                old = next(iter(self._buckets))
                self._buckets.pop(old).close()
            eng = Engine(H, W, max_batch=self.max_batch, dtype=self.dtype, device=self.device, weights=self._weights, decode_stream=False,
                         collapse_heads=self.collapse_heads)
            self.created += 1
            self._placement_dirty = True
        self._buckets[key] = eng
        return eng

    def _staging(self, eng, n, h, w):
        """Page-locked staging view [n, h, w, 3] into the engine's ONE grow-only pinned buffer (sized to the largest
        n * h * w * 3 seen).  A WIDER-style run has hundreds of distinct raw sizes and tail lengths: one pinned array per
        (n, h, w) would pile up GBs of page-locked memory, each allocation a device-synchronising hipHostMalloc.  Reuse
        is safe: an engine stages one chunk per round and the round ends with its decode_threshold (a synchronise)."""
        need = int(n) * int(h) * int(w) * 3
        st = eng.__dict__.get("_stage")
        if st is None or st[1] < need:
            if st is not None:
                eng.synchronize()
                eng.pinned_free(st[0])
            cap = max(need, int(1.25 * st[1]) if st is not None else need)
            eng.__dict__["_stage"] = st = eng.pinned_raw(cap)
        buf = (C.c_char * need).from_address(st[0].value)
        return np.frombuffer(buf, dtype=np.uint8, count=need).reshape(n, h, w, 3)

    def detect(self, imgs, threshold=0.2):
        del threshold                                              # ignored by the reference's decode (centerface.py:77)
        groups = {}                                                # network shape -> raw shape -> indices
        net = self.__dict__.setdefault("_net_shape", {})           # raw (h, w) -> network (H, W): transform() once per size
        addrs = []                                                 # per image: address if page-locked uint8 (pin / pinned_empty), else None
        for i, im in enumerate(imgs):
            hw = im.shape[:2] if type(im) is np.ndarray else np.asarray(im).shape[:2]
            HW = net.get(hw)
            if HW is None:
                HW = net[hw] = CenterFace.transform(None, hw[0], hw[1])[:2]
            groups.setdefault(HW, {}).setdefault(hw, []).append(i)
            addrs.append(_direct_addr(im))
        out = [None] * len(imgs)
        post = CenterFace.__new__(CenterFace)                      # only for _postprocess (no engine of its own)
        post.landmarks = self.landmarks
        keys = list(groups)
        for g0 in range(0, len(keys), self.max_buckets):           # at most max_buckets contexts alive at a time (LRU eviction)
            self._run_buckets({k: groups[k] for k in keys[g0:g0 + self.max_buckets]}, imgs, out, post, addrs)
        return out

    def _run_buckets(self, groups, imgs, out, post, addrs):
        # one work list per bucket (engine): chunks of one raw size, at most max_batch images
        work = []
        for (H, W), raws in groups.items():
            eng = self._engine(H, W)
            work.append([eng, [((h, w), idx[j:j + eng.max_batch]) for (h, w), idx in raws.items()
                               for j in range(0, len(idx), eng.max_batch)]])
        # New contexts since the last call: put the main streams on dispatch pipes such that contexts used one after the other never share
        # one (cf_spread_streams, window 2).  A process's queues offer three or four pipes; five contexts as the runtime places them had
        # three on one pipe (contexts 0, 3, 4), so the forwards of chunks 3 and 4 took turns kernel by kernel: 31.0 -> 33.9-35.4 k img/s on
        # the configs[3] mix with (0, 3) (1, 4) (2) instead.  Costs ~15 ms, once per new context.
        if self.__dict__.pop("_placement_dirty", False) and len(self._buckets) >= 2:
            engs = list(self._buckets.values())
            hs = (C.c_void_p * len(engs))(*[e._h for e in engs])
            engs[0]._chk(engs[0]._L.cf_spread_streams(hs, len(engs), 2, None))
        # Software pipeline over the chunks, taken round-robin over the buckets, on TWO host threads: this thread copies chunk i into
        # its context's page-locked buffer (with the staging threads) and enqueues it (asynchronous DMA + resize + forward on that
        # context's own streams); a collector thread waits for the oldest chunk in flight (decode_threshold: the wait happens inside
        # the C call, without the GIL), copies its results out and rescales them.  Round 3 did both on one thread and was host-bound
        # (6.3 ms per 128 images of which the device needs 4.2: profiles/r04_depth_sweep.txt).  A context has one staging buffer and
        # one decode state, so its previous chunk is collected before its next one is staged (the future of that chunk).
        order = []
        while any(chunks for _, chunks in work):
            for eng, chunks in work:
                if chunks:
                    order.append((eng,) + chunks.pop(0))

        def collect(item):
            eng, (h, w), idx = item
            pp = CenterFace.__new__(CenterFace)                    # only the reference's empty-result shapes are left to the host
            pp.landmarks = post.landmarks
            try:
                res = pp._postprocess_many(eng.decode_threshold(0.3, self.nms_thresh, self.max_dets), rescaled=True)
            finally:
                eng.set_rescale(0.0, 0.0)
            for i, r in zip(idx, res):
                out[i] = r

        collector = self.__dict__.get("_collector")
        if collector is None:
            from concurrent.futures import ThreadPoolExecutor
            collector = self.__dict__["_collector"] = ThreadPoolExecutor(max_workers=1, thread_name_prefix="cf-collect")
        last, futs = {}, []
        main_exc = None
        # Page-locked chunks are uploaded AHEAD: the copies of the next chunks (one per context at most) are enqueued before the
        # forward of this one, so the device's copy queue holds nothing but copies back to back and never stands behind a forward.
        direct = [all(addrs[i] is not None for i in idx) for _, _, idx in order]
        uploaded, up_next = {}, 0              # id(engine) -> index into ``order`` of the chunk whose copies are enqueued

        def upload_ahead(k):
            # the copies of chunks k, k+1, ... in order, while they are page-locked and their context has no upload waiting
            nonlocal up_next
            up_next = max(up_next, k)
            while up_next < len(order) and up_next <= k + len(work):
                e2, hw2, idx2 = order[up_next]
                if not direct[up_next] or id(e2) in uploaded:
                    return
                # calls on ONE context are serialised by the caller (include/centerface_hip.h): the collector thread may still be inside
                # this context's decode of its previous chunk -- no upload ahead into it until that chunk has been collected
                busy = last.get(id(e2))
                if busy is not None and not busy.done():
                    return
                e2._upload_addrs([addrs[i] for i in idx2], hw2[0], hw2[1], [imgs[i] for i in idx2])
                uploaded[id(e2)] = up_next
                up_next += 1
        try:
            for k, item in enumerate(order):
                eng, (h, w), idx = item
                prev = last.get(id(eng))
                if prev is not None:
                    prev.result()                                  # this context's earlier chunk has been collected
                eng.set_rescale(eng.H / h, eng.W / w)              # this chunk's scale_h, scale_w (centerface.py:68-71)
                if direct[k]:
                    upload_ahead(k)                                # (a no-op when an earlier step has uploaded this chunk already)
                    if uploaded.pop(id(eng), None) != k:
                        raise RuntimeError("CenterFaceBuckets: upload order lost track of chunk %d" % k)
                    eng.forward_uploaded()                         # page-locked images (pin / pinned_empty): DMA straight from them
                    upload_ahead(k + 1)                            # (after the forward: a context holds ONE upload at a time)
                else:
                    chunk = [imgs[i] for i in idx]
                    stage = self._staging(eng, len(idx), h, w)
                    _stage_copy_begin(stage, chunk)()
                    if (h, w) == (eng.H, eng.W):
                        eng.forward_enqueue(stage)
                    else:
                        eng.forward_resized_enqueue(stage)
                eng.decode_threshold_enqueue(0.3, self.nms_thresh, self.max_dets)       # decode + NMS run as soon as the forward is done
                f = collector.submit(collect, item)                # FIFO on one worker: collected in enqueue order
                last[id(eng)] = f
                futs.append(f)
        except BaseException as exc:                               # noqa: BLE001  (re-raised below, after the drain)
            main_exc = exc
        # drain even when staging / enqueue raised: no chunk stays in flight.  The exception of this thread wins; a collector
        # failure is raised only when nothing else is propagating (and is not re-raised a second time when it already surfaced
        # through prev.result() above).
        err = None
        for f in futs:
            try:
                f.result()
            except Exception as exc:                               # noqa: BLE001
                err = err or exc
        if main_exc is not None or err is not None:
            for eng, _ in work:                                    # the rescale is per-context state: never left behind by a failed chunk
                try:
                    eng.set_rescale(0.0, 0.0)
                except Exception:                                  # noqa: BLE001
                    pass
        if main_exc is not None:
            if err is not None and err is not main_exc:
                raise main_exc from err
            raise main_exc
        if err is not None:
            raise err

    __call__ = detect

    def close(self):
        col = self.__dict__.pop("_collector", None)
        if col is not None:
            col.shutdown(wait=True)
        for eng in self.__dict__.get("_buckets", {}).values():
            eng.close()
        self._buckets = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()                                           # the collector thread does not outlive the object
        except Exception:                                          # noqa: BLE001
            pass
