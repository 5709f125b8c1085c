/*
 * centerface_hip.h -- C ABI of libcenterface_hip.so: the MI355X (gfx950) CenterFace inference hot path.
 *
 * The reference has no FFI: its boundary is the Python class in centerface.py.  Every entry point
 * below names the reference code it replaces; INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add to centerface.py to route through this library.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types cross the boundary (device pointers travel as void*).
 *   - every function returns 0 (CF_OK) or a negative CF_E* code; cf_last_error(ctx) gives the text.
 *     The library never aborts and never falls back to a CPU path.
 *   - one cf_ctx = one GPU + its own HIP streams (forward, decode, input copy), buffers and graphs; calls on a ctx are
 *     serialised by the caller; different ctxs may be driven from different threads / processes (one process per GPU
 *     for multi-GPU; two ctxs on one GPU used alternately keep two batches in flight).
 *   - host outputs are written into CALLER-ALLOCATED buffers; nothing allocated here crosses back.
 *   - activations live in HBM as NHWC (channels contiguous), fp32 or bf16 storage, fp32 accumulate.
 */
#ifndef CENTERFACE_HIP_H
#define CENTERFACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CF_VERSION 100           /* 0.1.0 */

/* error codes */
#define CF_OK             0
#define CF_EINVAL        -1      /* bad argument (shape not multiple of 32, K > h*w, ...) */
#define CF_ENOMEM        -2
#define CF_EHIP          -3      /* a HIP runtime call failed; see cf_last_error */
#define CF_ESTATE        -4      /* call order violated (forward before load_weights, ...) */
#define CF_ESCHEMA       -5      /* weight set does not match the 94-tensor checkpoint schema */
#define CF_EOVERFLOW     -6      /* candidate capacity exceeded in threshold decode */

/* storage dtype of activations and packed weights (accumulation is always fp32) */
#define CF_F32   0               /* parity mode: fp32 storage, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) */
#define CF_BF16  1               /* throughput mode: bf16 storage, bf16 MFMA, fp32 accumulate */
#define CF_F32_SPLIT 2           /* tolerance mode: fp32 storage, every GEMM product as a split-bf16 ("bf16x3") product on the bf16
                                  * matrix pipe (hi.hi + lo.hi + hi.lo, fp32 accumulate): within north_star's 1e-3 of the reference
                                  * like CF_F32 (products exact to ~2^-16 relative), several times its speed */

/* network input formats accepted by cf_forward */
#define CF_IN_U8_HWC_BGR  0      /* uint8 [B,H,W,3] BGR, as cv2 gives it; /255, -mean, /std fused
                                    into the stem (replaces centerface.py:32-37) */
#define CF_IN_F32_NCHW    1      /* float [B,3,H,W], already normalised: the tensor the reference
                                    hands to net() at centerface.py:41 / eval_widerface.py:83 */

/* cf_create flags */
#define CF_FLAG_COLLAPSE_HEADS  1u   /* fold each head's conv3x3+b -> conv1x1+b (a linear pair,
                                        model/centernet.py:249-256) into one 3x3 conv 24->15 */
#define CF_FLAG_NO_GRAPH        2u   /* launch kernels eagerly instead of replaying a hipGraph */
#define CF_FLAG_NO_FUSE         4u   /* run every MBConv block as three kernels (expand, dw, project)
                                        instead of the fused kernel that keeps the 6x tensor in LDS */
#define CF_FLAG_NO_UPHEAD       8u   /* keep the last IDAUp stage and the heads as two kernels (bf16 +
                                        collapsed heads fuse them, the neck output stays in LDS) */
#define CF_FLAG_NO_NECK        16u   /* keep conv_last and the first two IDAUp stages as three kernels (bf16 fuses them
                                        into one: the 1/32 and 1/16 neck maps stay in LDS) */
#define CF_FLAG_NO_DECODE_STREAM 32u  /* device-output decodes (cf_decode_topk, cf_gather_topk) run on the context's main stream instead of a
                                      * decode stream of its own.  For hosts that keep THREE OR MORE contexts in flight (small batches):
                                      * HIP has four hardware queues per process, and three main + three decode streams share them
                                      * (configs[4] shard, B = 4: 3 contexts 9.5 k img/s with decode streams, 12.2 k without; 2 contexts 11.3 k) */
#define CF_FLAG_STREAM_HIGH 64u       /* the context's main and decode streams are created in the HIGHEST stream-priority class.  The HIP runtime
                                      * keeps one pool of (four) hardware queues per priority class and gives a new stream the least-used queue of its
                                      * class, so streams in a class of their own are placed independently of everything the process created in the
                                      * default class: 54.3-54.5 k img/s for two contexts from five different process histories, against 48.2-54.4 k
                                      * for default-class streams as created (round 6, tools/queue_order_probe.py).  The class is shared with the
                                      * device's copy stream and with this library's earlier contexts: after contexts have been destroyed in it the
                                      * map is no longer the same (45.7-47.8 k, tools/ring_sequence_probe.py) -- for a host that creates its contexts
                                      * once; cf_streams_share_queue_ex / cf_spread_streams verify and repair a placement in any history. */

typedef struct cf_ctx cf_ctx;

/* One checkpoint tensor, exactly as torch.save(model.state_dict()) holds it (train.py:165):
 * conv weights OIHW float32; num_batches_tracked may be passed (int64) and is ignored. */
typedef struct cf_tensor_desc {
    const char* name;            /* state_dict key, e.g. "layer1.0.conv.1.1.weight" */
    const void* data;            /* host pointer */
    int32_t     ndim;
    int64_t     dims[4];
    int32_t     dtype;           /* 0 = float32, 1 = int64 */
} cf_tensor_desc;

int         cf_version(void);
const char* cf_strerror(int code);
const char* cf_last_error(const cf_ctx* ctx);          /* ctx may be NULL: last create error */
int         cf_device_count(int* n);

/* ---- lifetime: replaces CenterFace.__init__ (centerface.py:16-27) -------------------------- */
/* H, W must be multiples of 32 (CenterFace.transform guarantees it, centerface.py:69). */
int cf_create(int device, int max_batch, int H, int W, int dtype, uint32_t flags, cf_ctx** out);
int cf_destroy(cf_ctx* ctx);
/* strict load of the 94-tensor state_dict (centerface.py:23-24): validates names and shapes,
 * folds BatchNorm (eval mode, centerface.py:25), repacks to kernel layouts, uploads. */
int cf_load_weights(cf_ctx* ctx, const cf_tensor_desc* tensors, int n);

/* ---- training-side pieces that share the detector's tensors (forward evaluation only) ----------------- */
/* CtdetLoss.forward (model/losses.py:347-374: focal loss :142-167 on clamp(sigmoid(hm), 1e-5, 1-1e-5),
 * RegL1Loss :239-250 on wh / reg / lm gathered at ind) evaluated on the head maps of the LAST cf_forward of
 * ctx.  Targets are host arrays as dataset/dataset.py:223-226 returns them: gt_hm [B,1,h,w] f32,
 * reg_mask / lm_mask [B,M] u8, ind / lm_ind [B,M] i64, wh_t / reg_t [B,M,2], lm_t [B,M,10].
 * weights = {hm_w, wh_w, off_w, lm_w} (reference defaults 1, 0.1, 1, 1).  out[5] = loss, hm_loss, wh_loss,
 * off_loss, lm_loss.  Blocking. */
int cf_ctdet_loss(cf_ctx* ctx, const float* gt_hm, const uint8_t* reg_mask, const int64_t* ind, const float* wh_t,
                  const float* reg_t, const uint8_t* lm_mask, const int64_t* lm_ind, const float* lm_t,
                  int max_objs, const float* weights4, float* out5);

/* ---- forward: replaces net(img)[0] (centerface.py:41, eval_widerface.py:83-84) ------------- */
/* `in` is a host pointer (in_on_device = 0; copied H2D on a copy stream into one of two staging buffers, so the
 * copy overlaps the previous forward -- keep the buffer unchanged until a blocking call on this context has
 * returned: cf_synchronize, cf_get_heads or a decode with host outputs) or a device pointer on
 * ctx's GPU (in_on_device = 1; 4-byte aligned -- CF_EINVAL otherwise).  Asynchronous: returns after
 * enqueueing. */
int cf_forward(cf_ctx* ctx, const void* in, int in_format, int in_on_device, int B);
/* cv2.resize + forward in one enqueue (centerface.py:30-41): imgs uint8 [B,h,w,3] BGR of ANY size are
 * stretch-resized on the device to the ctx's (H, W) with OpenCV's fixed-point INTER_LINEAR arithmetic for uint8
 * (11-bit coefficients, int32 passes; csrc/cf_util.hip restates it) and fed to the network.  Pinned to the
 * published algorithm, not to a particular cv2 binary (cv2 is not installable where this was built). */
int cf_forward_resized(cf_ctx* ctx, const void* imgs_u8, int in_on_device, int B, int h, int w);
/* The same for a batch held as B SEPARATE host images, imgs[b] -> uint8 [h,w,3] BGR -- what the loop of eval_widerface.py:76-90
 * has after B cv2.imread calls.  Every image is one asynchronous DMA into the context's buffer ((h, w) == (H, W): straight into
 * the network input, no resize launch).  From page-locked memory (cf_host_alloc, cf_host_register) there is no host-side
 * staging copy at all; pageable pointers work too, but then the call returns only once the runtime has staged them.  The images
 * must stay unchanged until a blocking call on ctx has returned (as for cf_forward). */
int cf_forward_images(cf_ctx* ctx, const void* const* imgs, int B, int h, int w);
/* The two halves of cf_forward_images: cf_upload_images only enqueues the host -> device copies (large batches: on the device's
 * copy streams, shared by every context of the process, in call order), cf_forward_uploaded enqueues resize + forward on what the
 * last cf_upload_images of ctx brought over (CF_ESTATE without one; any other upload or forward on ctx in between replaces it).
 * A host that keeps several contexts in flight issues the uploads of the next batches BEFORE the forwards of the earlier ones, so
 * that the copy queue never stands behind a forward (CenterFaceBuckets: 5 contexts, 128 VGA images, profiles/r05_vga_pipeline.md). */
int cf_upload_images(cf_ctx* ctx, const void* const* imgs, int B, int h, int w);
int cf_forward_uploaded(cf_ctx* ctx);
/* the resized uint8 [B,H,W,3] batch of the last cf_forward_resized (tests) */
int cf_get_resized_input(cf_ctx* ctx, void* out_u8, int B);
/* copies the four head maps of the last forward to host as NCHW float32: hm [B,1,h,w] raw logits
 * (what net() returns), wh [B,2,h,w], lm [B,10,h,w], reg [B,2,h,w]; h=H/4, w=W/4 (model/centernet.py
 * :277-280).  Any pointer may be NULL.  hm_sigmoid (optional) receives clamp(sigmoid(hm),1e-4,1-1e-4)
 * (centerface.py:43).  Synchronises. */
int cf_get_heads(cf_ctx* ctx, float* hm, float* wh, float* lm, float* reg, float* hm_sigmoid);

/* ---- decode D3: replaces ctdet_decode / _nms / _topk / _transpose_and_gather_feat
 *      (centerface_ext.py:11-82) on the last forward's heads --------------------------------- */
/* dets [B,K,6] = x1,y1,x2,y2,score,cls in heat-map units; lms [B,K,10] raw landmark rows at the
 * same cells (may be NULL); inds [B,K] flat cell index y*w+x (may be NULL).  Equal scores are
 * ordered lower-index-first (torch.topk leaves it unspecified).  use_reg = 0 gives the +0.5
 * branch (centerface_ext.py:65-67).  out_on_device selects host or device destination buffers.
 * 1 <= K <= h*w, any map size (K > 1024 sorts through global memory: slower, same results). */
int cf_decode_topk(cf_ctx* ctx, int K, int use_reg, float* dets, float* lms, int64_t* inds,
                   int out_on_device);

/* Same, followed by ctdet_post_process (utils/post_process.py:83-100): both box corners are mapped
 * from heat-map coordinates back to source-image coordinates with the inverse affine of
 * get_affine_transform(center, scale, rot=0, (out_w, out_h), inv=1) (utils/image.py:19-66), fused into
 * the decode kernel's epilogue.  centers [B,2] (c[i]), scales [B,2] (s[i]; [0] is used, as the
 * reference does), out_w/out_h = heat-map size.  cv2.getAffineTransform is replaced by a float64
 * 3-point solve. */
int cf_decode_topk_post(cf_ctx* ctx, int K, int use_reg, const float* centers, const float* scales,
                        int out_w, int out_h, float* dets, float* lms, int64_t* inds, int out_on_device);
/* the 2x3 float64 matrix used above (row-major), for callers that want transform_preds themselves */
int cf_affine_from_center_scale(float cx, float cy, float scale_w, int out_w, int out_h, double* trans6);

/* ---- decode D1: replaces CenterFace.decode + nms (centerface.py:73-151) -------------------- */
/* For each image: cells with hm > score_thresh in row-major order, boxes/landmarks with the
 * reference's arithmetic (offsets ignored, x2 = min(x1c + w, W)), greedy IoU >= nms_thresh
 * suppression in descending score order.  dets [B,max_out,5], lms [B,max_out,10] (may be NULL),
 * counts [B] = number of boxes that survive NMS (in the reference's keep order); only the first max_out
 * rows are written, so counts[b] > max_out tells the caller that image b was truncated (call again with a
 * larger max_out).  Any number of cells above the threshold is accepted, as in the reference: the candidate
 * workspace grows to the largest count seen (CF_ENOMEM if that does not fit).  Host buffers.
 * The reference ignores its `threshold` argument and uses 0.3 (centerface.py:77); pass 0.3f. */
int cf_decode_threshold(cf_ctx* ctx, float score_thresh, float nms_thresh, int max_out,
                        float* dets, float* lms, int32_t* counts);

/* mode 0 = the above (D1); mode 1 = D2, eval_widerface.decode (eval_widerface.py:92-110): the threshold
 * argument is honoured and the offsets are added -- reg channel 1 to x and channel 0 to y plus the
 * 0.5, exactly as that file does (:102-104) -- no landmarks; same greedy NMS (:112-152). */
int cf_decode_threshold_ex(cf_ctx* ctx, int mode, float score_thresh, float nms_thresh, int max_out,
                           float* dets, float* lms, int32_t* counts);

/* Same with an explicit clamp size: the reference's get_detections clamps boxes to a hard-coded (640, 640)
 * whatever the input size (eval_widerface.py:88); cf_decode_threshold[_ex] clamp to the context's (H, W). */
int cf_decode_threshold_sized(cf_ctx* ctx, int mode, float score_thresh, float nms_thresh, int img_h, int img_w,
                              int max_out, float* dets, float* lms, int32_t* counts);
/* centerface.py:55-62 on the device: from now on the threshold decodes of ctx write floor(x / scale_w) and floor(y / scale_h) for
 * the box corners and the landmark points (numpy's float32 `//`: the exact floor of the quotient), so the host has nothing left to
 * do per box.  scale_h = scale_w = 0 switches it off (the default: network coordinates).  CF_EINVAL for negative / mixed values. */
int cf_set_rescale(cf_ctx* ctx, float scale_h, float scale_w);
/* Optional asynchronous first half: enqueue the decode kernels right behind the last forward, no host wait.  A later
 * cf_decode_threshold_sized (or _ex / plain, which call it) with the SAME parameters then only waits and copies the results out;
 * with other parameters, or after another forward, it launches its own decode as usual.  For hosts that keep several contexts
 * in flight and collect them later (CenterFaceBuckets): the decode runs when the forward finishes, not when the host gets there. */
int cf_decode_threshold_enqueue(cf_ctx* ctx, int mode, float score_thresh, float nms_thresh, int img_h, int img_w, int max_out);

/* ---- fused convenience: forward + D3 decode in one enqueue (eval_widerface.py:76-90 shape) -- */
int cf_detect_topk(cf_ctx* ctx, const void* in, int in_format, int in_on_device, int B, int K,
                   float* dets, float* lms, int64_t* inds, int out_on_device);

/* ---- multi-GPU: one process (or thread) per GPU, batch sharded by rank, final boxes gathered over RCCL / xGMI -- */
/* The only exchange of the data path (images are independent end to end): an all-gather of the fixed-size
 * detection records.  The reference has no counterpart (train.py:11,17 import torch.distributed without using
 * it).  librccl is loaded on first use.  Rendezvous is the caller's: rank 0 calls cf_comm_unique_id and ships the
 * 128 bytes to the other ranks by any means (file, socket, MPI, torch.distributed store), then every rank calls
 * cf_comm_create (collective: returns when all `world` ranks have joined). */
#define CF_COMM_ID_BYTES 128
typedef struct cf_comm cf_comm;
int cf_comm_unique_id(void* id, int bytes);
/* ONE communicator per rank (= per GPU), whatever the number of contexts on that GPU: it owns the rank's single gather
 * stream, and every all-gather of every context of the rank is enqueued there in call order -- all ranks then see the same
 * collective order as long as they call cf_gather_topk in the same order (concurrent communicators on free-running streams
 * have no such guarantee and may deadlock).  `ctx` only names the device. */
int cf_comm_create(cf_ctx* ctx, int rank, int world, const void* id, cf_comm** out);
/* Single process / single thread that owns one context per GPU: the n ncclCommInitRank calls inside one
 * ncclGroupStart/End (un-grouped, the first call would block forever waiting for the others).  out[i] = rank i = ctxs[i]. */
int cf_comm_create_all(cf_ctx** ctxs, int n, cf_comm** out);
/* Dry run of a `world`-rank gather on ONE GPU, without RCCL: the process plays the ranks in turn.  cf_comm_loopback_rank(comm, r)
 * names the rank whose cf_gather_topk comes next; that call decodes the context's last forward (rank r's shard), writes the slot
 * header and copies the slot into rank r's place of the landing area (the stand-in for the all-gather).  The call of the LAST rank
 * of a step (every rank exactly once, any order) runs the header check + unpack and fills `records` [world * B, K, 16]; the calls
 * before it write nothing.  Everything else -- slot sizes, shard agreement, header protocol, step numbers, the rank-major unpack
 * arithmetic, the mismatch latch -- is the code of the real gather: a deployment checks its shard math for 2 / 4 / 8 ranks on one
 * GPU (tests/test_multigpu.py asserts the 8-rank result equals the unsharded order bit for bit). */
int cf_comm_create_loopback(cf_ctx* ctx, int world, cf_comm** out);
int cf_comm_loopback_rank(cf_comm* comm, int rank);
int cf_comm_destroy(cf_comm* comm);
/* Give up on a communicator whose collective does not complete (ncclCommAbort: in-flight RCCL kernels exit), then release
 * it.  cf_comm_query: 0 = everything enqueued on the gather stream (shard agreement, gathers) has completed, 1 = still
 * running, CF_EINVAL = the agreement or a slot header found unequal (B, K) or unequal step numbers on the ranks (latched;
 * text from cf_comm_last_error).  Never blocks -- a host polls it against its own deadline and aborts instead of hanging.
 * cf_comm_synchronize waits by polling cf_comm_query, so the latch ends the wait too. */
int cf_comm_abort(cf_comm* comm);
int cf_comm_query(cf_comm* comm);
int cf_comm_synchronize(cf_comm* comm);
const char* cf_comm_last_error(cf_comm* comm);
/* Declare the largest shard every rank gathers: B images x K records (= the fixed slot size of every later gather).  Collective by
 * contract: every rank calls it with the same values at the same point of its call sequence (the first cf_gather_topk of a
 * communicator calls it implicitly; call it again on every rank to change the geometry).  It enqueues the ONLY extra
 * collective of the gather path -- a 2-int all-gather of each rank's (B, K) plus a device-side compare that latches a
 * mismatch -- and returns without waiting; poll cf_comm_query (0 = agreed, CF_EINVAL = unequal shards) against a deadline.
 * The first gather of the geometry reads that verdict (waiting for it if the host has not) BEFORE it enqueues a record
 * gather, so an all-gather with per-rank-unequal counts is never launched. */
int cf_comm_set_shard(cf_comm* comm, int B, int K);
/* Test hooks.  what = 0: park the gather stream behind a spin kernel of `value` ms (a collective that does not complete in
 * time); what = 1: the header of the next gather's slot carries B + value (the mismatch path on a single rank). */
int cf_comm_debug(cf_comm* comm, int what, int value);
void* cf_comm_stream(cf_comm* comm);                    /* the gather stream (hipStream_t) */
/* D3 decode of the last forward (as cf_decode_topk) followed by the all-gather: records [world * B, K, 16] =
 * x1,y1,x2,y2,score,cls,lm0..lm9 per detection, rank-major = exactly the batch order of the unsharded run.  Every rank
 * must pass the same B and K.  ONE collective per call: each rank sends a fixed-size slot = a 64-byte header {magic, B, K,
 * step number} + its B x K x 16 records; behind the all-gather a device kernel checks every rank's header against THIS
 * rank's (B, K) and step count and strips the headers into `records`.  The slot agreed by cf_comm_set_shard is a CAPACITY:
 * any (B, K) with B x K no larger than the agreed product -- the ragged last batch of a run, a smaller K -- is gathered in
 * the same slot without another agreement, as long as it is the same on every rank.  A rank whose shard does not fit
 * still sends a full-size slot (carrying its real B, K) and returns CF_EINVAL; a rank whose (B, K) differs from the
 * others' is found by every rank's header check: the counts and the collective sequence stay identical on all ranks,
 * every rank latches the mismatch, nobody hangs.  A latched mismatch
 * is reported by the blocking form itself, and by cf_comm_query / cf_comm_synchronize / the next cf_gather_topk for the
 * asynchronous form.  The decode runs on the context's decode stream, the all-gather on the communicator's stream behind
 * it, both underneath the next forward.  out_on_device = 1: `records` is a device buffer, the call never waits on the host
 * once the shard is agreed (cf_comm_query / cf_comm_synchronize before reading it); 0: host buffer, blocking. */
int cf_gather_topk(cf_ctx* ctx, cf_comm* comm, int K, int use_reg, float* records, int out_on_device);

/* ---- stream / timing plumbing -------------------------------------------------------------- */
int cf_synchronize(cf_ctx* ctx);
/* HIP events on the ctx stream (the stream the kernels are launched on). slot in [0, 64). */
int cf_event_record(cf_ctx* ctx, int slot);
int cf_event_elapsed_ms(cf_ctx* ctx, int slot_begin, int slot_end, float* ms);
/* Per-kernel timing of one forward (+ optional top-K decode when K > 0): runs the launches eagerly
 * with an event pair around each, returns up to `cap` records.  Replaces the datetime prints at
 * centerface.py:38,47,49. */
typedef struct cf_op_time {
    char    name[48];            /* e.g. "layer1.0.dw" */
    char    kind[16];            /* stem | pw | dw | head | decode */
    char    kernel[160];         /* demangled kernel symbol, as rocprofv3 --kernel-trace prints it */
    float   ms;
    double  algo_bytes;          /* algorithmic HBM bytes of this launch: unpadded in + out */
    double  flops;               /* 2 * MACs */
} cf_op_time;
int cf_profile_forward(cf_ctx* ctx, const void* in, int in_format, int in_on_device, int B, int K,
                       cf_op_time* out, int cap, int* n_out);
/* The launch plan of a context (model/centernet.py:263-280 as kernels) and a layer-by-layer trace for parity
 * tests: cf_forward_trace runs the forward eagerly up to and including plan entry `op_index` and copies that
 * entry's output to the host as NCHW float32 [B, C, H, W] (the head entry: C = 16 record channels hm_sigmoid,
 * wh0-1, lm0-9, reg0-1, hm_raw).  Entries with fused_away != 0 are not launched (their work happens inside the
 * next entry) and cannot be traced.  Blocking. */
typedef struct cf_op_info {
    char    name[48];            /* e.g. "layer1.0.mbconv" */
    char    kind[16];            /* stem0 | mbconv | expdw | pw | dw | head | stem */
    int32_t C, H, W;             /* output tensor */
    int32_t fused_away;
} cf_op_info;
int cf_plan_size(cf_ctx* ctx, int* n);
int cf_plan_op(cf_ctx* ctx, int i, cf_op_info* out);
int cf_forward_trace(cf_ctx* ctx, const void* in, int in_format, int in_on_device, int B, int op_index, float* out_nchw);
/* The context's HIP streams as opaque hipStream_t values: `main_stream` carries the forward (and the
 * host-output decodes), `decode_stream` the device-output top-K decode.  For callers that chain their own
 * device work (e.g. an RCCL all-gather of the decoded boxes on another stream) with stream/event waits
 * instead of cf_synchronize(). */
int cf_get_streams(cf_ctx* ctx, void** main_stream, void** decode_stream);
/* Two contexts with a batch in flight on each overlap their forwards on the GPU -- unless their MAIN streams were folded
 * onto the same hardware queue (HIP maps a process's streams onto 4 queues; which one a new stream gets depends on every
 * stream the process created before): then the two forwards run strictly one after the other (42 k instead of 46 k img/s
 * at 64 x 640x640).  cf_streams_share_queue tells (a ~0.3 ms spin on a's main stream, an empty kernel on b's, both contexts
 * idle); cf_reroll_streams replaces the context's main and decode streams by new ones (new streams are created BEFORE the
 * old ones are destroyed; the context must be idle; captured graphs stay valid).  Re-rolling helps in a fresh process; where it keeps
 * landing on the same queue, cf_spread_streams (below) is the tool.  What EngineRing calls at construction when its streams clash. */
int cf_streams_share_queue(cf_ctx* a, cf_ctx* b, int* shared);
/* The same probe for any pair of the contexts' streams: which = 0 main, 1 decode, 2 the device's copy stream (a == b allowed).
 * + 16 on either selector: the DISPATCH-PIPE probe -- the first stream runs a kernel whose grid cannot be resident at once (its pipe stays busy
 * launching workgroups), so streams on different queues of one pipe are caught too: such a pair costs a ring of two contexts its overlap
 * just like a shared queue (49-51 k instead of 53-54 k img/s; tools/queue_order_probe.py). */
int cf_streams_share_queue_ex(cf_ctx* a, int which_a, cf_ctx* b, int which_b, int* shared);
/* Put the main streams (and the decode streams of contexts created without CF_FLAG_NO_DECODE_STREAM) of n idle contexts of one device on
 * pairwise different hardware queues: candidates are created and probed one after the other, those that land on a used queue are kept as
 * ballast until the end, so the runtime's fewest-streams-first placement moves on (a create-then-destroy re-roll can come back to the same
 * queue forever in a process with unevenly loaded queues).  Main streams first; at most four queues exist.  window = 0: all pairwise different;
 * window = w > 0: a main stream only differs from those of the w contexts before it in ctxs (more contexts than pipes, used round-robin in
 * that order).  Decode streams may end up sharing with each other, never with a main stream.  *n_distinct (may be NULL) =
 * streams placed on a queue of their own.  Captured graphs stay valid.  Call it only when streams clash as created: the placement the runtime
 * gives the first contexts of a process measured 4 % faster than a fresh one (54.1 against 51.9 k img/s, no queue shared in either). */
int cf_spread_streams(cf_ctx** ctxs, int n, int window, int* n_distinct);
int cf_reroll_streams(cf_ctx* ctx);
/* hipGraph replay state: number of captured forward graphs held by the context, and how many
 * (input, format, batch) keys could not be captured and run as eager launches instead. */
int cf_graph_stats(cf_ctx* ctx, int* n_graphs, int* n_uncapturable);
/* page-locked host memory: a batch staged here is copied by DMA, asynchronously, underneath the previous forward
 * (cf_forward from pageable memory stages through the driver and blocks the caller for the copy) */
int cf_host_alloc(cf_ctx* ctx, uint64_t bytes, void** hptr);
int cf_host_free(cf_ctx* ctx, void* hptr);
/* The same without a context (any device may DMA from it; hipHostMalloc, portable): for a host's frame pool. */
int cf_pinned_alloc(uint64_t bytes, void** hptr);
int cf_pinned_free(void* hptr);
/* Page-lock / release memory the caller owns (no context needed; any device may then DMA from it).  For buffers that are reused:
 * registering costs a page-table walk (~0.1 ms per MB).  Errors: cf_op_last_error().
 * CONTRACT (round 6): `hptr` is page-aligned (4096), `bytes` a multiple of the page size, and the range is a mapping OF ITS OWN
 * (mmap, shm, a device driver's buffer) -- NOT memory from malloc / new / numpy.  Registration works on pages and on the kernel's
 * view of the mapping; the C library grows, trims and reuses its heap (brk) underneath live registrations, and the GPU then faults on
 * a page it was told is locked ("Memory access fault by GPU ... Reason: Unknown", SIGABRT from the HSA runtime's event thread).
 * Measured on MI355X / ROCm 7 (tools/diag/pin_churn_probe.py, 45 s of allocation churn per run): heap memory 6 faults in 22 runs,
 * whole pages inside a heap array 1 in 6, mmap regions 0 in 6, cf_pinned_alloc memory 0 in 6.  An unaligned pointer or size is
 * CF_EINVAL; whose mapping it is cannot be checked here (cfa.pin refuses the [heap] segment).  Prefer cf_pinned_alloc. */
int cf_host_register(void* hptr, uint64_t bytes);
int cf_host_unregister(void* hptr);
/* device memory helpers so a host language without a GPU allocator can keep inputs resident */
int cf_device_alloc(cf_ctx* ctx, uint64_t bytes, void** dptr);
int cf_device_free(cf_ctx* ctx, void* dptr);
int cf_memcpy_h2d(cf_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int cf_memcpy_d2h(cf_ctx* ctx, void* dst, const void* src, uint64_t bytes);

/* ---- per-op entry points (tests; host pointers; NCHW float32 like the torch ops they replace) */
const char* cf_op_last_error(void);      /* text of the last failing cf_op_* call on this thread */
/* ConvReLU depthwise / ShuffleV2 dw: pad -> conv2d(groups=C, bias=False) -> act.
 * x [B,C,H,W], w [C,1,k,k], y [B,C,Ho,Wo]; pad_lo/pad_hi as ZeroPad2d (model/centernet.py:63,68-70;
 * model/blocks.py:28); act: 0 none, 1 swish.  bias may be NULL (folded BN shift, blocks.py:29).
 * C a multiple of 8; one image (H x W x C in the storage type) smaller than 4 GiB (CF_EINVAL otherwise). */
int cf_op_dwconv(int device, int dtype, const float* x, const float* w, const float* bias, float* y,
                 int B, int C, int H, int W, int k, int stride, int pad_lo, int pad_hi, int act);
/* 1x1 conv [+bias] [+act] [+residual]: x [B,Cin,H,W], w [Cout,Cin], y [B,Cout,H,W]
 * (model/centernet.py:109-110,117-118,134-137,179-184; blocks.py:22-24,31-33). act 0/1 swish/2 relu */
int cf_op_pwconv(int device, int dtype, const float* x, const float* w, const float* bias,
                 const float* residual, float* y, int B, int Cin, int Cout, int H, int W, int act);
/* MBConvBlock.forward, se=False (model/centernet.py:89-140) as ONE fused kernel: x [B,Cin,H,W],
 * w_exp [hid,Cin], w_dw [hid,1,k,k], w_proj [Cout,hid]; residual when Cin==Cout and stride==1.
 * Returns CF_EINVAL for shapes the fused kernel does not cover (t == 1, Cout > 96). */
int cf_op_mbconv(int device, int dtype, const float* x, const float* w_exp, const float* w_dw,
                 const float* w_proj, float* y, int B, int Cin, int hid, int Cout, int H, int W,
                 int k, int stride);
/* The first two thirds of MBConvBlock.forward (model/centernet.py:109-114): expand 1x1 + Swish, depthwise
 * k x k (stride, `_get_padding`) + Swish, as ONE kernel (bf16 storage only; the path the engine uses for
 * the blocks whose Cout is too wide to fuse the project conv as well).  y [B,hid,Ho,Wo].
 * Returns CF_EINVAL for shapes / dtypes the kernel does not cover. */
int cf_op_expand_dw(int device, int dtype, const float* x, const float* w_exp, const float* w_dw, float* y,
                    int B, int Cin, int hid, int H, int W, int k, int stride);
/* CtdetLoss.forward on explicit NCHW head maps (hm as logits), same targets as cf_ctdet_loss. */
int cf_op_ctdet_loss(int device, const float* hm_raw, const float* wh, const float* reg, const float* lm,
                     int B, int h, int w, const float* gt_hm, const uint8_t* reg_mask, const int64_t* ind,
                     const float* wh_t, const float* reg_t, const uint8_t* lm_mask, const int64_t* lm_ind,
                     const float* lm_t, int max_objs, const float* weights4, float* out5);
/* Target maps of dataset/dataset.py:160-217 for boxes [B,M,4] / landmarks [B,M,10] given in OUTPUT-MAP
 * coordinates (after the affine of :172-179), counts [B]: Gaussian heat map (utils/image.py:95-141, radius in
 * float64), wh, reg, ind, reg_mask, landmarks, lm_ind, lm_mask -- all [B,M,...] except hm [B,1,h,w]. */
int cf_op_encode_targets(int device, const float* boxes, const float* lms, const int32_t* counts, int B, int h, int w,
                         int max_objs, float* hm, float* wh, float* reg, int64_t* ind, uint8_t* reg_mask,
                         float* landmarks, int64_t* lm_ind, uint8_t* lm_mask);
/* ShuffleV2Block.forward in eval mode (model/blocks.py:4-62) as ONE entry point: x [B, 2*inp, H, W] (stride 1) or
 * [B, inp, H, W] (stride 2) -> y [B, oup, Ho, Wo].  Weights as the reference's state_dict holds them: m_w0
 * [mid, inp] (branch_main.0), m_wdw [mid,1,k,k] (.3), m_w5 [oup-inp, mid] (.5); p_wdw [inp,1,k,k] (branch_proj.0),
 * p_w2 [inp, inp] (.2) for stride 2 (NULL otherwise); every *_bn* is 4 rows [C]: weight, bias, running_mean,
 * running_var (eps 1e-5), folded here.  The channel shuffle (:56-62) and the concat (:50,54) are channel
 * addressing inside the kernels.  inp, mid, oup-inp multiples of 8. */
int cf_op_shufflev2(int device, int dtype, const float* x, float* y, int B, int inp, int oup, int mid, int H, int W,
                    int ksize, int stride, const float* m_w0, const float* m_bn1, const float* m_wdw, const float* m_bn4,
                    const float* m_w5, const float* m_bn6, const float* p_wdw, const float* p_bn1, const float* p_w2,
                    const float* p_bn3);
/* stem: ConvReLU(3,32,3,stride 2) on a normalised float NCHW tensor or a uint8 HWC BGR image
 * (model/centernet.py:224 ; centerface.py:32-37). y [B,32,H/2,W/2] */
int cf_op_stem(int device, int dtype, const void* x, int in_format, const float* w, float* y,
               int B, int H, int W);
/* IDAUp.forward (model/centernet.py:200-204) with raw (unfolded) BN parameters, 5 floats rows:
 * bn_up[4][C] / bn_cv[4][C] = weight, bias, running_mean, running_var. lo [B,C,h,w], skip [B,Cs,2h,2w] */
int cf_op_idaup(int device, int dtype, const float* lo, const float* skip, const float* w_up,
                const float* bn_up, const float* w_cv, const float* bn_cv, float eps, float* y,
                int B, int C, int Cs, int h, int w);
/* the four heads (model/centernet.py:247-261,277-279) on x [B,24,h,w]; weights concatenated in
 * head order hm,wh,lm,reg: w0 [4][24,24,3,3], b0 [4][24], w1 [15,24], b1 [15].
 * out [B,15,h,w] = hm(raw),wh(2),lm(10),reg(2).  collapse != 0 uses the folded 3x3 24->15 conv. */
int cf_op_heads(int device, int dtype, const float* x, const float* w0, const float* b0,
                const float* w1, const float* b1, float* out, int B, int h, int w, int collapse);
/* ctdet_decode (centerface_ext.py:52-82) on explicit maps: heat [B,1,h,w] (already sigmoid'ed),
 * wh [B,2,h,w], reg [B,2,h,w] or NULL, lm [B,10,h,w] or NULL. */
int cf_op_ctdet_decode(int device, const float* heat, const float* wh, const float* reg,
                       const float* lm, int B, int h, int w, int K,
                       float* dets, float* lms, int64_t* inds);
/* CenterFace.decode + nms (centerface.py:73-151) on explicit maps for ONE image size (H,W). */
int cf_op_decode_threshold(int device, const float* hm, const float* wh, const float* lm,
                           int B, int h, int w, int img_h, int img_w, float score_thresh,
                           float nms_thresh, int max_out, float* dets, float* lms, int32_t* counts);
int cf_op_decode_threshold_ex(int device, int mode, const float* hm, const float* wh, const float* reg,
                              const float* lm, int B, int h, int w, int img_h, int img_w, float score_thresh,
                              float nms_thresh, int max_out, float* dets, float* lms, int32_t* counts);
/* ctdet_post_process's coordinate part on explicit detections dets [B,K,dim] (in place, host array) */
int cf_op_ctdet_post_process(int device, float* dets, const float* centers, const float* scales, int B, int K,
                             int dim, int out_w, int out_h);
/* CenterFace.nms alone (centerface.py:111-151): keep[] receives kept indices in keep order. */
int cf_op_nms(int device, const float* boxes, const float* scores, int n, float nms_thresh,
              int32_t* keep, int32_t* n_keep);
/* bbox_overlap (eval_widerface.py:48-74: the "+1" IoU of every detection against every annotation, float32 arithmetic as numpy
 * computes it for float32 inputs) and the two counts evaluate (:195-206) takes from it, for n_img images in one call: rows of
 * box_stride / query_stride floats (x1,y1,x2,y2 first), concatenated; box_off / query_off [n_img + 1] = first row of each image.
 * overlaps (optional): the dense [N_i][K_i] float64 matrices back to back.  counts (optional) [n_img][2]: detections whose best
 * overlap exceeds thresh (evaluate's "detected_num"), annotations whose best overlap does ("true_positives"). */
int cf_op_box_match(int device, int n_img, const float* boxes, int box_stride, const int32_t* box_off, const float* query,
                    int query_stride, const int32_t* query_off, float thresh, double* overlaps, int32_t* counts);

#ifdef __cplusplus
}
#endif
#endif /* CENTERFACE_HIP_H */
