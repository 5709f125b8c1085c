// C-ABI runtime of libcenterface_hip.so: context, execution plan, BN folding + weight packing,
// forward, decode and profiling entry points.  See include/centerface_hip.h for the contract and
// the reference lines each entry point replaces.
//
// The network graph (model/centernet.py:205-280) is held here as a static launch plan built at
// cf_create(): 1 stem + 12 MBConv blocks (expand pw -> dw -> project pw [+residual]) + conv_last +
// 3 IDAUp (one fused pw launch each) + 1 fused head launch = 40 kernel launches per forward,
// versus ~140 ATen ops in the reference.  Activations are NHWC in a handful of
// reused HBM buffers (ping/pong + expanded + depthwise) so the working set of a batch stays small
// and L2 / Infinity-Cache resident between producer and consumer where it fits.
#include "cf_exp.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <mutex>
#include <vector>

#include "centerface_hip.h"
#include CF_EXP_INC(cf_runtime_0)
#include "cf_common.h"
#include "cf_kernels.h"

using namespace cf;

namespace {

thread_local std::string g_create_error;

const int kSettings[7][5] = {  // t, c, n, s, k -- model/centernet.py:211-219
    {1, 16, 1, 1, 3}, {6, 24, 2, 2, 3}, {6, 32, 2, 2, 5}, {6, 64, 2, 2, 3},
    {6, 96, 2, 1, 5}, {6, 160, 2, 2, 5}, {6, 320, 1, 1, 3}};

enum OpKind { OP_STEM = 0, OP_PW, OP_DW, OP_HEAD, OP_MB, OP_STEM0, OP_EXPDW };
const char* kKindName[] = {"stem", "pw", "dw", "head", "mbconv", "stem0", "expdw"};

struct Op {
    OpKind kind;
    std::string name;
    int in = -1, out = -1, res = -1, low = -1;       // buffer ids
    int Hin = 0, Win = 0, Cin = 0, Hout = 0, Wout = 0, Cout = 0;
    int k = 1, s = 1, pad_lo = 0, act = 0;
    // weight source
    std::string wkey, bnkey, upkey, upbnkey;
    float bn_eps = 0.f;
    // device weights
    void* wp = nullptr; float* bias = nullptr; float* upw = nullptr; float* upb = nullptr;
    float* b1 = nullptr; float* w1d = nullptr;
    // fused MBConv (OP_MB): Cin -> hid -> Cout
    int hid = 0; bool residual = false; std::string wkey_dw, wkey_proj;
    MbGeom geo{}; void* wexp = nullptr; float* wdw = nullptr; void* wproj = nullptr;
    double macs = 0;                                  // per image
    // IDAUp stage 3 fused into the head kernel (cf_uphead.hip): the OP_PW op keeps its weights but is not
    // launched (fused_away); the OP_HEAD op launches the fused kernel with its partner's operands
    bool fused_away = false; int partner = -1;
    int neck_cl = -1, neck_u1 = -1;                   // on the up2 op: conv_last and up1 are computed by its (fused) launch
    bool neck_part = false;                           // conv_last / up1 when fused into up2's launch
    // expand+dw -> project pairs (bf16): the depthwise tensor between them is kept in pixel-block order
    // [m / 32][hid / 8][m % 32][8] (MbParams::yblock / PwParams::xblock): the project GEMM's activation loads coalesce
    // The same order for the block outputs of layer3.1 ... layer6.0 (layout_pass): in / out / res say which operands of
    // this launch are in block order.
    bool in_blk = false, out_blk = false, res_blk = false;
};

struct Buf { std::string name; size_t elems = 0; bool f32 = false; void* p = nullptr; };

}  // namespace

struct cf_ctx {
    int device = 0, max_batch = 0, H = 0, W = 0, dtype = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    // device-output top-K decode runs on a second stream so that it overlaps the NEXT forward (it
    // occupies 64 workgroups of a 256-CU chip); the head kernel of that forward waits for it
    hipStream_t stream2 = nullptr; hipEvent_t ev_fwd = nullptr, ev_dec = nullptr; bool dec_pending = false;
    std::vector<Buf> bufs;
    std::vector<Op> ops;
    std::vector<void*> owned;                 // device allocations to free
    int buf_in = -1, buf_heads = -1, buf_resized = -1;
    // host inputs: H2D copies run on their own stream into two alternating staging buffers, so the copy of
    // batch i+1 overlaps the forward of batch i (PCIe-inclusive rate ~ max(copy, compute), not their sum)
    hipStream_t stream_in = nullptr, stream_in2 = nullptr; hipEvent_t ev_copy2 = nullptr; int buf_in2 = -1; int in_slot = 0; int in_slot_used = -1;
    struct Upload { bool pending; int B, h, w, slot; bool small, dual; } up = {false, 0, 0, 0, 0, false, false};      // cf_upload_images -> cf_forward_uploaded
    hipEvent_t ev_src_copy = nullptr, ev_src_free = nullptr; bool src_busy = false;     // src_stage: filled on stream_in, read by the resize on stream
    hipEvent_t ev_copy[2] = {nullptr, nullptr}, ev_slot_free[2] = {nullptr, nullptr}; bool slot_busy[2] = {false, false};
    bool weights_loaded = false;
    int last_B = 0;
    std::string err;
    hipEvent_t events[64] = {};
    // decode workspaces (lazy)
    unsigned long long* keys = nullptr; int* key_count = nullptr; unsigned long long* big = nullptr; size_t big_stride = 0;
    float* d_rec = nullptr; float* d_slot = nullptr;                    // d_rec = d_slot + 16: the gather slot of a rank is [header | records]
    hipEvent_t ev_main_dec = nullptr; bool main_dec_pending = false;
    // two-lane schedule (cf_forward_lanes): segment events and the back half still to be launched
    hipEvent_t ev_seg1 = nullptr, ev_seg2 = nullptr; bool seg2_recorded = false;
    bool lane_pending = false; const void* lane_in = nullptr; int lane_fmt = 0, lane_B = 0; int lane_cut1 = -1, lane_cut2 = -1;
    hipEvent_t ev_gather = nullptr; bool gather_pending = false;       // the last all-gather still reads d_rec (communicator's stream)
    float* hm_plane = nullptr; double* d_trans = nullptr; uint8_t* src_stage = nullptr; size_t src_stage_bytes = 0;
    float* d_dets = nullptr; float* d_lms = nullptr; long long* d_inds = nullptr; int decK = 0;
    float* t_cand = nullptr; int* t_count = nullptr; int* t_order = nullptr; unsigned long long* t_mask = nullptr; int t_B = 0;
    float* t_dets = nullptr; float* t_lms = nullptr; int* t_counts = nullptr; int* t_overflow = nullptr;
    int t_cap = 0, t_maxout = 0;
    // results of the threshold decode in page-locked host memory (written by the kernels over PCIe): [overflow | counts | dets | lms]
    uint8_t* h_thr = nullptr; bool t_host = false; hipEvent_t ev_thr = nullptr;
    // cf_decode_threshold_enqueue: the decode kernels of the last forward are already in the stream with these parameters
    bool thr_pending = false; int thr_mode = 0, thr_h = 0, thr_w = 0, thr_maxout = 0, thr_B = 0; float thr_score = 0.f, thr_nms = 0.f, thr_rs_h = 0.f, thr_rs_w = 0.f;
    int prio = 0;                      // stream priority class of the context's main / decode streams: -1 lowest, 0 normal, +1 highest
    float rs_h = 0.f, rs_w = 0.f;      // cf_set_rescale: the threshold decode floor-divides x by rs_w and y by rs_h (0 = off)
    // hipGraph replay of the backbone + neck launches, one executable graph per (input pointer,
    // input format, batch): the second forward with a key captures it, later ones replay it
    struct FwdGraph { const void* in; int fmt, B; hipGraphExec_t exec; bool broken; unsigned long long used; };
    std::vector<FwdGraph> graphs;
    unsigned long long graph_clock = 0;

    int fail(int code, const char* fmt, ...) {
        char b[512];
        va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap);
        err = b;
        return code;
    }
};

#define HIPCHK(ctx, call)                                                                        \
    do { hipError_t e_ = (call);                                                                 \
         if (e_ != hipSuccess) return (ctx)->fail(CF_EHIP, "%s failed: %s (%s:%d)", #call,       \
                                                  hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

namespace {

int add_buf(cf_ctx* c, const char* name, bool f32 = false) {
    c->bufs.push_back(Buf{name, 0, f32, nullptr});
    return (int)c->bufs.size() - 1;
}
void need(cf_ctx* c, int id, size_t elems) { if (c->bufs[id].elems < elems) c->bufs[id].elems = elems; }

// Pixel-block order for block outputs: the input of an expand+depthwise kernel is read as MFMA operand fragments (lane =
// halo pixel, 16 bytes per k-step) by EVERY hidden-chunk workgroup of the tile; from NHWC rows each such load touches 64
// cache lines.  If the tensor's producer can write block order (a project GEMM or a bf16 fused MBConv kernel) and all of its
// readers can read it (expand+dw input, GEMM input, GEMM residual), the tensor is kept in block order instead.
void layout_pass(cf_ctx* c) {
    static const bool off = cf_ab_int("CF_IN_XBLOCK", 1) == 0;      // A/B
    if (off || c->dtype == CF_F32) return;                          // bf16, and (round 5) the split mode: cf_mbconv5.hip reads block order too
    auto& ops = c->ops;
    for (size_t i = 0; i < ops.size(); ++i) {
        if (ops[i].kind != OP_EXPDW || !ops[i].out_blk || ops[i].in_blk || (ops[i].Cin % 8)) continue;
        const int buf = ops[i].in;
        int j = (int)i - 1;
        while (j >= 0 && ops[j].out != buf) --j;
        if (j < 0) continue;
        Op& pr = ops[j];
        const bool can_write = !pr.fused_away && pr.low < 0 &&
                               ((pr.kind == OP_PW && pr.bnkey.empty()) || (pr.kind == OP_MB && (pr.geo.kind == 1 || pr.geo.kind == 5 || pr.geo.kind == 6)) ||
                                (pr.kind == OP_MB && c->dtype != CF_BF16 && pr.geo.kind == 0));      // cf_mbconv.hip's fp32-tile kernel (layer3.1 in the split mode)
        if (!can_write) continue;
        bool ok = true;
        size_t end = j + 1;
        for (; end < ops.size() && ops[end].out != buf; ++end) {
            const Op& r = ops[end];
            if (r.low == buf) ok = false;
            if (r.in == buf && !(r.kind == OP_EXPDW || r.kind == OP_PW) ) ok = false;
            if (r.kind == OP_MB && r.in == buf) ok = false;
            if (r.kind == OP_HEAD && r.partner >= 0 && (ops[r.partner].in == buf || ops[r.partner].low == buf)) ok = false;
            if (r.fused_away && !r.neck_part && (r.in == buf || r.low == buf)) ok = false;          // read by the fused up3+heads kernel
            if (r.neck_part && r.low == buf) ok = false;                                             // (the fused neck kernel reads either order)
        }
        if (!ok) continue;
        pr.out_blk = true;
        for (size_t k = j + 1; k < end; ++k) {
            if (ops[k].in == buf) ops[k].in_blk = true;
            if (ops[k].res == buf) ops[k].res_blk = true;
        }
    }
}

void build_plan(cf_ctx* c) {
    const int H = c->H, W = c->W;
    int A = add_buf(c, "ping"), B = add_buf(c, "pong"), E = add_buf(c, "expanded"), D = add_buf(c, "depthwise");
    int S1 = add_buf(c, "skip_l1"), S2 = add_buf(c, "skip_l2"), S4 = add_buf(c, "skip_l4");
    int N0 = add_buf(c, "neck0"), N1 = add_buf(c, "neck1"), N2 = add_buf(c, "neck2"), N3 = add_buf(c, "neck3");
    c->buf_heads = add_buf(c, "heads", true);

    auto push = [&](Op op) {
        need(c, op.out, (size_t)op.Hout * op.Wout * (op.kind == OP_HEAD ? 16 : op.Cout));
        c->ops.push_back(op);
    };
    const bool fuse = !(c->flags & CF_FLAG_NO_FUSE);
    if (fuse) {
        // first_conv + layer0.0 (dw 3x3 + project 32->16) fused: cf_stem0.hip
        Op st; st.kind = OP_STEM0; st.name = "first_conv+layer0.0"; st.in = -1; st.out = A;
        st.Hin = H; st.Win = W; st.Cin = 3; st.Hout = H / 2; st.Wout = W / 2; st.Cout = 16; st.k = 3; st.s = 2; st.act = 1;
        st.wkey = "first_conv.0.1.weight"; st.wkey_dw = "layer0.0.conv.0.1.weight"; st.wkey_proj = "layer0.0.conv.1.weight";
        st.macs = (double)st.Hout * st.Wout * (32 * 27 + 32 * 9 + 32 * 16);
        push(st);
    } else {
    // stem: first_conv (model/centernet.py:224)
    Op st; st.kind = OP_STEM; st.name = "first_conv"; st.in = -1; st.out = A;
    st.Hin = H; st.Win = W; st.Cin = 3; st.Hout = H / 2; st.Wout = W / 2; st.Cout = 32; st.k = 3; st.s = 2; st.act = 1;
    st.wkey = "first_conv.0.1.weight"; st.macs = (double)st.Hout * st.Wout * 32 * 27;
    push(st);
    }

    int cur = A, curH = H / 2, curW = W / 2, cin = fuse ? 16 : 32;
    for (int li = fuse ? 1 : 0; li < 7; ++li) {
        const int t = kSettings[li][0], cout = kSettings[li][1], n = kSettings[li][2], k = kSettings[li][4];
        for (int i = 0; i < n; ++i) {
            const int s = (i == 0) ? kSettings[li][3] : 1;
            const int hid = cin * t;
            char pre[32]; snprintf(pre, sizeof pre, "layer%d.%d", li, i);
            const int p = std::max(k - s, 0);      // _get_padding (:68-70)
            const int Ho = (curH + p - k) / s + 1, Wo = (curW + p - k) / s + 1;
            const bool residual = (cin == cout && s == 1);      // :101
            int dst = (cur == A) ? B : A;
            if (li == 1 && i == n - 1) dst = S1;                 // x1 (:266)
            if (li == 2 && i == n - 1) dst = S2;                 // x2 (:267)
            if (li == 4 && i == n - 1) dst = S4;                 // x4 (:269)
            MbGeom geo = mb_geometry(c->dtype, cin, hid, cout, k, s);
            // Cout = 96 (layer4.0 / 4.1): three project accumulator blocks leave the fully fused kernel 2 waves per SIMD at
            // 246 VGPRs and it runs at 0.47 of its own instruction-issue bound (profiles/r02b_valu_bound.md); expand + depthwise
            // in one kernel (no accumulators: 6 waves per SIMD) + the LDS-weight GEMM for the project conv is faster although
            // the depthwise output makes a round trip through HBM: 0.111 -> 0.095 ms and 0.166 -> 0.141 ms.  CF_SPLIT_WIDE=0: A/B.
            static const bool fuse_wide = cf_ab_int("CF_SPLIT_WIDE", 1) == 0;
            if (!fuse_wide && c->dtype == CF_BF16 && cout > 64 && geo.kind == 1) geo.ok = false;
            // fp32-storage modes, round 5: the same split for Cout = 96 (layer4.x) wherever cf_mbconv5.hip has the shape -- the fused
            // kernel's three accumulator blocks hold it to 8x16 tiles (1.9x halo recompute) and two waves per SIMD
            if (!fuse_wide && c->dtype != CF_BF16 && cout > 64 && expdw_geometry(c->dtype, cin, hid, k, s).ok) geo.ok = false;
            if (t != 1 && geo.ok && !(c->flags & CF_FLAG_NO_FUSE)) {
                // fused expand -> dw -> project (cf_mbconv.hip): one launch, expanded tensor stays in LDS
                Op m; m.kind = OP_MB; m.name = std::string(pre) + ".mbconv"; m.in = cur; m.out = dst;
                m.Hin = curH; m.Win = curW; m.Cin = cin; m.hid = hid; m.Hout = Ho; m.Wout = Wo; m.Cout = cout;
                m.k = k; m.s = s; m.pad_lo = p / 2; m.residual = residual; m.geo = geo;
                m.wkey = std::string(pre) + ".conv.0.1.weight"; m.wkey_dw = std::string(pre) + ".conv.1.1.weight";
                m.wkey_proj = std::string(pre) + ".conv.2.weight";
                m.macs = (double)curH * curW * cin * hid + (double)Ho * Wo * hid * (k * k + cout);
                push(m);
                cur = dst; curH = Ho; curW = Wo; cin = cout;
                continue;
            }
            int src = cur, j = 0;
            MbGeom xg = (t != 1 && !(c->flags & CF_FLAG_NO_FUSE)) ? expdw_geometry(c->dtype, cin, hid, k, s) : MbGeom{};
            if (xg.ok) {
                // expand + depthwise in one launch (cf_mbconv2.hip), project stays a GEMM: blocks too wide to fuse fully
                Op m; m.kind = OP_EXPDW; m.name = std::string(pre) + ".expand+dw"; m.in = cur; m.out = D;
                m.Hin = curH; m.Win = curW; m.Cin = cin; m.hid = hid; m.Hout = Ho; m.Wout = Wo; m.Cout = hid;
                m.k = k; m.s = s; m.pad_lo = p / 2; m.geo = xg;
                m.wkey = std::string(pre) + ".conv.0.1.weight"; m.wkey_dw = std::string(pre) + ".conv.1.1.weight";
                m.macs = (double)curH * curW * cin * hid + (double)Ho * Wo * hid * k * k;
                static const bool blk_off = cf_ab_int("CF_PW_XBLOCK", 1) == 0;      // A/B
                m.out_blk = !blk_off && hid <= 960;
                push(m);
                j = 1;
            } else {
            if (t != 1) {                         // expand pw + Swish (:109-110)
                Op e; e.kind = OP_PW; e.name = std::string(pre) + ".expand"; e.in = cur; e.out = E;
                e.Hin = e.Hout = curH; e.Win = e.Wout = curW; e.Cin = cin; e.Cout = hid; e.act = 1;
                e.wkey = std::string(pre) + ".conv.0.1.weight"; e.macs = (double)curH * curW * cin * hid;
                push(e); src = E; j = 1;
            }
            Op d; d.kind = OP_DW; d.name = std::string(pre) + ".dw"; d.in = src; d.out = D;
            d.Hin = curH; d.Win = curW; d.Cin = d.Cout = hid; d.Hout = Ho; d.Wout = Wo; d.k = k; d.s = s;
            d.pad_lo = p / 2; d.act = 1;
            d.wkey = std::string(pre) + ".conv." + std::to_string(j) + ".1.weight";
            d.macs = (double)Ho * Wo * hid * k * k;
            push(d);
            }
            Op pr; pr.kind = OP_PW; pr.name = std::string(pre) + ".project"; pr.in = D; pr.out = dst;
            pr.in_blk = !c->ops.empty() && c->ops.back().kind == OP_EXPDW && c->ops.back().out_blk;
            pr.res = residual ? cur : -1;
            pr.Hin = pr.Hout = Ho; pr.Win = pr.Wout = Wo; pr.Cin = hid; pr.Cout = cout; pr.act = 0;
            pr.wkey = std::string(pre) + ".conv." + std::to_string(j + 1) + ".weight";
            pr.macs = (double)Ho * Wo * hid * cout;
            push(pr);
            cur = dst; curH = Ho; curW = Wo; cin = cout;
        }
    }
    // conv_last = conv_1x1_bn(320, 24) (:236, :179-184)
    Op cl; cl.kind = OP_PW; cl.name = "conv_last"; cl.in = cur; cl.out = N0;
    cl.Hin = cl.Hout = curH; cl.Win = cl.Wout = curW; cl.Cin = cin; cl.Cout = 24; cl.act = 1;
    cl.wkey = "conv_last.0.weight"; cl.bnkey = "conv_last.1"; cl.bn_eps = 1e-5f;
    cl.macs = (double)curH * curW * cin * 24;
    push(cl);
    // IDAUp x3 (:237-239, :186-204): one fused launch each
    struct { const char* nm; int skip, skipC, out; } ups[3] = {{"up1", S4, 96, N1}, {"up2", S2, 32, N2}, {"up3", S1, 24, N3}};
    int low = N0;
    for (auto& u : ups) {
        curH *= 2; curW *= 2;
        Op o; o.kind = OP_PW; o.name = u.nm; o.in = u.skip; o.out = u.out; o.low = low;
        o.Hin = o.Hout = curH; o.Win = o.Wout = curW; o.Cin = u.skipC; o.Cout = 24; o.act = 2;
        o.wkey = std::string(u.nm) + ".conv.0.weight"; o.bnkey = std::string(u.nm) + ".conv.1";
        o.upkey = std::string(u.nm) + ".up.weight"; o.upbnkey = std::string(u.nm) + ".bn_up"; o.bn_eps = 1e-3f;
        o.macs = (double)curH * curW * (u.skipC * 24 + 24);
        push(o); low = u.out;
    }
    // heads (:240-261, :277-279)
    Op hd; hd.kind = OP_HEAD; hd.name = "heads"; hd.in = N3; hd.out = c->buf_heads;
    hd.Hin = hd.Hout = curH; hd.Win = hd.Wout = curW; hd.Cin = 24; hd.Cout = 15;
    hd.macs = (c->flags & CF_FLAG_COLLAPSE_HEADS) ? (double)curH * curW * 216 * 15
                                                  : (double)curH * curW * (4 * 216 * 24 + 15 * 24);
    push(hd);
    // (bf16 and the split-mode tolerance path; the exact-fp32 test mode keeps the two launches)
    if (fuse && !(c->flags & CF_FLAG_NO_UPHEAD) && c->dtype != CF_F32 && (c->flags & CF_FLAG_COLLAPSE_HEADS)) {
        const int ih = (int)c->ops.size() - 1, iu = ih - 1;               // heads, up3
        c->ops[iu].fused_away = true;
        c->ops[ih].partner = iu;
        c->ops[ih].name = "up3+heads";
        c->ops[ih].macs += c->ops[iu].macs;
    }
    static const bool neck_off = cf_ab_int("CF_NECK", 1) == 0;      // A/B in an experiments build; the product switch is CF_FLAG_NO_NECK
    if (fuse && !neck_off && !(c->flags & CF_FLAG_NO_NECK) && (c->dtype == CF_BF16 || c->dtype == CF_F32_SPLIT) && cin == 320) {
        int icl = -1, iu1 = -1, iu2 = -1;
        for (size_t i = 0; i < c->ops.size(); ++i) {
            if (c->ops[i].name == "conv_last") icl = (int)i;
            if (c->ops[i].name == "up1") iu1 = (int)i;
            if (c->ops[i].name == "up2") iu2 = (int)i;
        }
        if (icl >= 0 && iu1 == icl + 1 && iu2 == iu1 + 1 && c->ops[iu1].Cin == 96 && c->ops[iu2].Cin == 32) {
            c->ops[icl].fused_away = c->ops[iu1].fused_away = true;
            c->ops[icl].neck_part = c->ops[iu1].neck_part = true;
            c->ops[iu2].neck_cl = icl; c->ops[iu2].neck_u1 = iu1;
            c->ops[iu2].name = "conv_last+up1+up2";
            c->ops[iu2].macs += c->ops[icl].macs + c->ops[iu1].macs;
        }
    }
    layout_pass(c);
}

struct WeightSet {
    std::map<std::string, const cf_tensor_desc*> m;
    const float* f(const std::string& k) const { return (const float*)m.at(k)->data; }
};

bool shape_is(const cf_tensor_desc* t, std::initializer_list<int64_t> dims) {
    if (t->ndim != (int)dims.size()) return false;
    int i = 0;
    for (auto d : dims) if (t->dims[i++] != d) return false;
    return true;
}

// fold eval-mode BatchNorm y = (x - mean) * gamma / sqrt(var + eps) + beta into scale/shift
void bn_fold(const WeightSet& ws, const std::string& pre, int C, float eps, std::vector<double>& scale, std::vector<double>& shift) {
    const float* g = ws.f(pre + ".weight"); const float* b = ws.f(pre + ".bias");
    const float* mu = ws.f(pre + ".running_mean"); const float* var = ws.f(pre + ".running_var");
    scale.resize(C); shift.resize(C);
    for (int c = 0; c < C; ++c) {
        scale[c] = (double)g[c] / std::sqrt((double)var[c] + (double)eps);
        shift[c] = (double)b[c] - (double)mu[c] * scale[c];
    }
}

template <typename T>
int upload(cf_ctx* c, const std::vector<T>& host, T** dptr) {
    void* d = nullptr;
    HIPCHK(c, hipMalloc(&d, host.size() * sizeof(T) + 64));
    c->owned.push_back(d);
    // on the context's own (non-blocking) stream, never the legacy stream: another thread may be capturing a forward
    // graph of ITS context, and a legacy-stream copy would implicitly synchronise with it ("would make the legacy stream
    // depend on a capturing stream")
    HIPCHK(c, hipMemcpyAsync(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *dptr = (T*)d;
    return CF_OK;
}
int upload_bytes(cf_ctx* c, const std::vector<char>& host, void** dptr) {
    void* d = nullptr;
    HIPCHK(c, hipMalloc(&d, host.size() + 64));
    c->owned.push_back(d);
    HIPCHK(c, hipMemcpyAsync(d, host.data(), host.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *dptr = d;
    return CF_OK;
}

int expect(cf_ctx* c, const WeightSet& ws, const std::string& key, std::initializer_list<int64_t> dims) {
    auto it = ws.m.find(key);
    if (it == ws.m.end()) return c->fail(CF_ESCHEMA, "missing key in state_dict: %s", key.c_str());
    if (!shape_is(it->second, dims)) return c->fail(CF_ESCHEMA, "size mismatch for %s", key.c_str());
    if (it->second->dtype != 0) return c->fail(CF_ESCHEMA, "%s must be float32", key.c_str());
    return CF_OK;
}
int expect_bn(cf_ctx* c, const WeightSet& ws, const std::string& pre, int C) {
    for (const char* leaf : {".weight", ".bias", ".running_mean", ".running_var"}) {
        int r = expect(c, ws, pre + leaf, {C}); if (r) return r;
    }
    if (!ws.m.count(pre + ".num_batches_tracked")) return c->fail(CF_ESCHEMA, "missing key in state_dict: %s.num_batches_tracked", pre.c_str());
    return CF_OK;
}

}  // namespace

// spins for `ticks` of the 100 MHz wall clock (cf_streams_share_queue)
__global__ void cf_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
}

// The same with a grid that cannot be resident at once (64 KB of LDS per workgroup = two per CU): the dispatcher of the stream's hardware
// PIPE stays busy launching workgroups for the whole run of the kernel, so a kernel on another queue of the SAME pipe has to wait --
// which is what two forwards on such a pair do to each other, launch after launch (each kernel holds far more workgroups than fit)
__global__ void cf_fat_spin_kernel(long long ticks) {
    extern __shared__ char fat_lds[];
    if (ticks < 0) fat_lds[threadIdx.x] = 0;             // (keeps the allocation)
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
}

// Shard agreement of the gather (cf_comm_set_shard): publish this rank's (B, K), all-gather the pairs ONCE per geometry, ...
__global__ void cf_comm_publish_kernel(int* mine, int B, int K) { mine[0] = B; mine[1] = K; }
// ... and compare every rank's against ours; a mismatch is latched in host-visible memory
__global__ void cf_comm_compare_kernel(const int* all, int world, int B, int K, int* flag) {
    if (threadIdx.x != 0 || flag[0]) return;
    for (int r = 0; r < world; ++r)
        if (all[2 * r] != B || all[2 * r + 1] != K) {
            flag[1] = r; flag[2] = all[2 * r]; flag[3] = all[2 * r + 1]; flag[4] = B; flag[5] = K;
            __threadfence_system();
            flag[0] = 1;
            __threadfence_system();
            return;
        }
}
// Every gather step sends ONE fixed-size slot per rank: a 16-float header {magic, B, K, step} in front of the B x K x 16 records.
constexpr unsigned kSlotMagic = 0x43464731u;      // "CFG1"
constexpr int kSlotHeader = 16;                   // floats (64 bytes: the records stay line-aligned)
__global__ void cf_comm_header_kernel(float* slot, int B, int K, unsigned step) {
    unsigned* h = reinterpret_cast<unsigned*>(slot);
    h[0] = kSlotMagic; h[1] = (unsigned)B; h[2] = (unsigned)K; h[3] = step;
}
// Behind the all-gather of the slots: validate every rank's header against the agreed shard (B, K) and this step's number --
// a mismatch is latched in host-visible memory, as above -- and copy the records of all ranks, rank-major, into `dst`
// (nullptr: validate only; the host-destination path strips the headers with a strided copy instead).
__global__ void cf_comm_unpack_kernel(const float* slots, int world, size_t slot_floats, int B, int K, unsigned step, float* dst, int* flag) {
    const size_t n = (size_t)B * K * 16;
    if (blockIdx.x == 0 && threadIdx.x == 0 && !flag[0])
        for (int r = 0; r < world; ++r) {
            const unsigned* h = reinterpret_cast<const unsigned*>(slots + (size_t)r * slot_floats);
            if (h[0] != kSlotMagic || h[1] != (unsigned)B || h[2] != (unsigned)K || h[3] != step) {
                flag[1] = r; flag[2] = (int)h[1]; flag[3] = (int)h[2]; flag[4] = B; flag[5] = K; flag[6] = (int)h[3]; flag[7] = (int)step;
                __threadfence_system();
                flag[0] = h[0] != kSlotMagic ? 3 : (h[3] != step && h[1] == (unsigned)B && h[2] == (unsigned)K) ? 2 : 1;
                __threadfence_system();
                break;
            }
        }
    if (!dst) return;
    const size_t n4 = n / 4, total = n4 * world;                       // 16 floats per record: n is a multiple of 4
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / n4, j = i - r * n4;
        reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(slots + r * slot_floats + kSlotHeader)[j];
    }
}

// One host-to-device copy stream per DEVICE, shared by every context on it.  Two contexts with a copy stream each
// (cfa.EngineRing: two batches in flight) ran their 78 MB transfers concurrently and the pair moved ~27 GB/s instead of
// the 46 GB/s one transfer at a time reaches (39.4k -> 22.6k img/s with the batch starting in pinned host memory every
// step); in one queue the transfers run back to back, in submission order, which is the order the ring needs.
// (What is left depends on how the HIP runtime folds the process's streams onto its hardware queues -- four by default,
// GPU_MAX_HW_QUEUES: two contexts own five streams, and a copy stream that shares a queue with a compute stream waits
// behind its kernels: 28-39k img/s host-fed with two contexts against 39.5k with one.  Host-fed serving is PCIe-bound
// at ~39k img/s either way; use one context for it.)
namespace {
struct CopyStream { hipStream_t s = nullptr, s2 = nullptr; int refs = 0; };
std::mutex g_copy_mu;
CopyStream g_copy[64];
// highest priority: HIP folds a process's streams onto a few hardware queues, and a copy stream that shares its queue with some context's
// main stream stands behind that context's whole forward; streams of another priority get hardware queues of their own
static hipError_t make_copy_stream(hipStream_t* out) {
    int lo = 0, hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    static const bool prio = cf_ab_int("CF_COPY_PRIO", 1) != 0;       // (round 6, with the ring in the highest class too: lowest-class copy streams 50.1 k against 52.4 k)
    if (e == hipSuccess) e = hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio ? hi : lo);
    if (e != hipSuccess) *out = nullptr;
    return e;
}
hipError_t make_ctx_stream(cf_ctx* c, hipStream_t* out) {
    if (c->prio == 0) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    int least = 0, greatest = 0;                       // (numerically the greatest priority is the smaller number)
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, c->prio > 0 ? greatest : least);
}
hipError_t acquire_copy_stream(int device, hipStream_t* out) {
    std::lock_guard<std::mutex> lk(g_copy_mu);
    CopyStream& cs = g_copy[device & 63];
    if (!cs.s) { hipError_t e = make_copy_stream(&cs.s); if (e != hipSuccess) return e; }
    ++cs.refs;
    *out = cs.s;
    return hipSuccess;
}
// The device's SECOND copy stream (cf_upload_images deals large batches of separately allocated images to both) is created at its first
// use, not with the first context: one more stream at context-creation time changes which hardware queues the contexts' own streams
// get, and the two-context ring lost its overlap to that (54.4 k -> 47.2 k img/s at 64 x 640x640; profiles/r05_vga_pipeline.md section 4).
hipError_t second_copy_stream(int device, hipStream_t* out) {
    std::lock_guard<std::mutex> lk(g_copy_mu);
    CopyStream& cs = g_copy[device & 63];
    if (!cs.s2) { hipError_t e = make_copy_stream(&cs.s2); if (e != hipSuccess) return e; }
    *out = cs.s2;
    return hipSuccess;
}
void release_copy_stream(int device) {
    std::lock_guard<std::mutex> lk(g_copy_mu);
    CopyStream& cs = g_copy[device & 63];
    if (--cs.refs == 0 && cs.s) {
        hipStreamDestroy(cs.s);
        if (cs.s2) hipStreamDestroy(cs.s2);
        cs.s = cs.s2 = nullptr;
    }
}
}  // namespace

extern "C" {

int cf_version(void) { return CF_VERSION; }

const char* cf_strerror(int code) {
    switch (code) {
        case CF_OK: return "ok";
        case CF_EINVAL: return "invalid argument";
        case CF_ENOMEM: return "out of memory";
        case CF_EHIP: return "HIP runtime error";
        case CF_ESTATE: return "invalid call order";
        case CF_ESCHEMA: return "state_dict does not match the CenterFace checkpoint schema";
        case CF_EOVERFLOW: return "candidate capacity exceeded";
        default: return "unknown error";
    }
}

const char* cf_last_error(const cf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int cf_device_count(int* n) {
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) { g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); *n = 0; return CF_EHIP; }
    *n = k;
    return CF_OK;
}

int cf_create(int device, int max_batch, int H, int W, int dtype, uint32_t flags, cf_ctx** out) {
    if (!out) return CF_EINVAL;
    *out = nullptr;
    if (max_batch < 1 || H < 32 || W < 32 || (H % 32) || (W % 32) || (dtype != CF_F32 && dtype != CF_BF16 && dtype != CF_F32_SPLIT)) {
        g_create_error = "cf_create: H and W must be positive multiples of 32, max_batch >= 1, dtype CF_F32|CF_BF16|CF_F32_SPLIT";
        return CF_EINVAL;
    }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return CF_EHIP; }
    cf_ctx* c = new cf_ctx();
    c->device = device; c->max_batch = max_batch; c->H = H; c->W = W; c->dtype = dtype; c->flags = flags;
    auto bail = [&](int code, const char* what, hipError_t he) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(he);
        cf_destroy(c);
        return code;
    };
    c->prio = (flags & CF_FLAG_STREAM_HIGH) ? 1 : 0;
    if ((e = make_ctx_stream(c, &c->stream)) != hipSuccess) return bail(CF_EHIP, "hipStreamCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_fwd, hipEventDisableTiming)) != hipSuccess) return bail(CF_EHIP, "hipEventCreate", e);
    if ((e = acquire_copy_stream(c->device, &c->stream_in)) != hipSuccess) return bail(CF_EHIP, "hipStreamCreate", e);
    for (int i = 0; i < 2; ++i) {
        if ((e = hipEventCreateWithFlags(&c->ev_copy[i], hipEventDisableTiming)) != hipSuccess) return bail(CF_EHIP, "hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&c->ev_slot_free[i], hipEventDisableTiming)) != hipSuccess) return bail(CF_EHIP, "hipEventCreate", e);
    }
    if ((e = hipEventCreateWithFlags(&c->ev_dec, hipEventDisableTiming)) != hipSuccess) return bail(CF_EHIP, "hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_main_dec, hipEventDisableTiming)) != hipSuccess) return bail(CF_EHIP, "hipEventCreate", e);
    for (auto& ev : c->events) if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(CF_EHIP, "hipEventCreate", e);
    build_plan(c);
    // input staging: the larger of u8 HWC and f32 NCHW
    c->buf_in = add_buf(c, "input", true);
    need(c, c->buf_in, (size_t)3 * H * W);
    c->buf_in2 = add_buf(c, "input2", false);
    need(c, c->buf_in2, ((size_t)3 * H * W + elem_size(dtype) - 1) / elem_size(dtype));     // 3*H*W BYTES per image (u8 only)
    // cf_forward_resized writes here (not into a host-input staging slot: those belong to the copy stream's protocol)
    c->buf_resized = add_buf(c, "input_resized", false);
    need(c, c->buf_resized, ((size_t)3 * H * W + elem_size(dtype) - 1) / elem_size(dtype));
    for (auto& b : c->bufs) {
        // + slack: kernels may over-read one 16-byte chunk; a tensor in pixel-block order is padded to whole 32-pixel blocks
        // (at most 31 pixels x 960 channels x 2 bytes)
        size_t bytes = b.elems * (size_t)max_batch * (b.f32 ? 4 : elem_size(dtype)) + 256 + (b.f32 ? 0 : (size_t)32 * 960 * elem_size(dtype));
        if ((e = hipMalloc(&b.p, bytes)) != hipSuccess) return bail(e == hipErrorOutOfMemory ? CF_ENOMEM : CF_EHIP, "hipMalloc(activations)", e);
        // zero-initialised incl. the slack: kernels may over-read (never write) one 16-byte chunk
        if ((e = hipMemsetAsync(b.p, 0, bytes, c->stream)) != hipSuccess) return bail(CF_EHIP, "hipMemset(activations)", e);
    }
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail(CF_EHIP, "hipMemset(activations)", e);
    if ((e = hipMalloc((void**)&c->hm_plane, (size_t)max_batch * (H / 4) * (W / 4) * sizeof(float))) != hipSuccess)
        return bail(CF_EHIP, "hipMalloc(hm_plane)", e);
    *out = c;
    return CF_OK;
}

int cf_destroy(cf_ctx* c) {
    if (!c) return CF_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->stream2) { hipStreamSynchronize(c->stream2); hipStreamDestroy(c->stream2); }
    if (c->ev_fwd) hipEventDestroy(c->ev_fwd);
    if (c->stream_in) { hipStreamSynchronize(c->stream_in); if (c->stream_in2) hipStreamSynchronize(c->stream_in2); release_copy_stream(c->device); }
    for (int i = 0; i < 2; ++i) { if (c->ev_copy[i]) hipEventDestroy(c->ev_copy[i]); if (c->ev_slot_free[i]) hipEventDestroy(c->ev_slot_free[i]); }
    if (c->ev_copy2) hipEventDestroy(c->ev_copy2);
    if (c->ev_src_copy) hipEventDestroy(c->ev_src_copy);
    if (c->ev_src_free) hipEventDestroy(c->ev_src_free);
    if (c->ev_dec) hipEventDestroy(c->ev_dec);
    if (c->ev_gather) { hipEventSynchronize(c->ev_gather); hipEventDestroy(c->ev_gather); }
    if (c->ev_main_dec) hipEventDestroy(c->ev_main_dec);
    if (c->ev_seg1) hipEventDestroy(c->ev_seg1);
    if (c->ev_seg2) hipEventDestroy(c->ev_seg2);
    for (auto& b : c->bufs) if (b.p) hipFree(b.p);
    for (void* p : c->owned) hipFree(p);
    for (void* p : {(void*)c->src_stage, (void*)c->d_trans, (void*)c->hm_plane, (void*)c->keys, (void*)c->key_count, (void*)c->big, (void*)c->d_slot, (void*)c->d_dets, (void*)c->d_lms, (void*)c->d_inds, (void*)c->t_cand, (void*)c->t_count,
                    (void*)c->t_order, (void*)c->t_mask, (void*)(c->t_host ? nullptr : c->t_dets), (void*)(c->t_host ? nullptr : c->t_lms), (void*)c->t_counts, (void*)c->t_overflow})
        if (p) hipFree(p);
    if (c->h_thr) hipHostFree(c->h_thr);
    if (c->ev_thr) hipEventDestroy(c->ev_thr);
    for (auto& ev : c->events) if (ev) hipEventDestroy(ev);
    for (auto& g : c->graphs) if (g.exec) hipGraphExecDestroy(g.exec);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return CF_OK;
}

int cf_load_weights(cf_ctx* c, const cf_tensor_desc* tensors, int n) {
    if (!c || !tensors) return CF_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    WeightSet ws;
    for (int i = 0; i < n; ++i) {
        if (!tensors[i].name || !tensors[i].data) return c->fail(CF_EINVAL, "tensor %d has a null name or data pointer", i);
        ws.m[tensors[i].name] = &tensors[i];
    }
    // strict schema check (centerface.py:24): no unexpected keys
    size_t expected_keys = 0;
    const int dt = c->dtype;
    for (auto& op : c->ops) {
        int r = CF_OK;
        if (op.kind == OP_STEM) { r = expect(c, ws, op.wkey, {32, 3, 3, 3}); expected_keys += 1; }
        else if (op.kind == OP_DW) { r = expect(c, ws, op.wkey, {op.Cin, 1, op.k, op.k}); expected_keys += 1; }
        else if (op.kind == OP_STEM0) {
            r = expect(c, ws, op.wkey, {32, 3, 3, 3});
            if (!r) r = expect(c, ws, op.wkey_dw, {32, 1, 3, 3});
            if (!r) r = expect(c, ws, op.wkey_proj, {16, 32, 1, 1});
            expected_keys += 3;
        }
        else if (op.kind == OP_MB) {
            r = expect(c, ws, op.wkey, {op.hid, op.Cin, 1, 1});
            if (!r) r = expect(c, ws, op.wkey_dw, {op.hid, 1, op.k, op.k});
            if (!r) r = expect(c, ws, op.wkey_proj, {op.Cout, op.hid, 1, 1});
            expected_keys += 3;
        }
        else if (op.kind == OP_EXPDW) {
            r = expect(c, ws, op.wkey, {op.hid, op.Cin, 1, 1});
            if (!r) r = expect(c, ws, op.wkey_dw, {op.hid, 1, op.k, op.k});
            expected_keys += 2;
        }
        else if (op.kind == OP_PW) {
            r = expect(c, ws, op.wkey, {op.Cout, op.Cin, 1, 1}); expected_keys += 1;
            if (!r && !op.bnkey.empty()) { r = expect_bn(c, ws, op.bnkey, op.Cout); expected_keys += 5; }
            if (!r && !op.upkey.empty()) {
                r = expect(c, ws, op.upkey, {op.Cout, 1, 2, 2});
                if (!r) r = expect_bn(c, ws, op.upbnkey, op.Cout);
                expected_keys += 6;
            }
        } else {
            const char* hn[4] = {"hm", "wh", "lm", "reg"}; const int hc[4] = {1, 2, 10, 2};
            for (int h = 0; h < 4 && !r; ++h) {
                std::string p = hn[h];
                r = expect(c, ws, p + ".0.weight", {24, 24, 3, 3});
                if (!r) r = expect(c, ws, p + ".0.bias", {24});
                if (!r) r = expect(c, ws, p + ".1.weight", {hc[h], 24, 1, 1});
                if (!r) r = expect(c, ws, p + ".1.bias", {hc[h]});
                expected_keys += 4;
            }
        }
        if (r) return r;
    }
    if (ws.m.size() != expected_keys) {
        return c->fail(CF_ESCHEMA, "state_dict has %zu tensors, the CenterFace schema has %zu (unexpected keys present)",
                       ws.m.size(), expected_keys);
    }

    // A reload on a live context (e.g. one checkpoint per epoch for the validation-loss path): the captured
    // hipGraphs have the OLD weight pointers baked into their kernel parameters and the old buffers would stay
    // allocated until cf_destroy -- drain the streams, drop every graph, free the previous weight set.
    if (c->weights_loaded) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
        HIPCHK(c, hipStreamSynchronize(c->stream_in));
        c->dec_pending = false;
        for (auto& g : c->graphs) if (g.exec) hipGraphExecDestroy(g.exec);
        c->graphs.clear();
        for (void* p : c->owned) hipFree(p);
        c->owned.clear();
        for (auto& op : c->ops) { op.wp = nullptr; op.bias = op.upw = op.upb = op.b1 = op.w1d = nullptr; op.wexp = nullptr; op.wdw = nullptr; op.wproj = nullptr; }
        c->weights_loaded = false;
    }

    for (auto& op : c->ops) {
        if (op.kind == OP_STEM) {
            std::vector<char> w(stem_packed_bytes(dt));
            stem_pack_weights(dt, ws.f(op.wkey), w.data());
            int r = upload_bytes(c, w, &op.wp); if (r) return r;
        } else if (op.kind == OP_STEM0) {
            static const bool px_off = cf_ab_int("CF_STEM0_KIND", 1) == 0;
            const bool px = dt == CF_BF16 && !px_off;                 // second-generation kernel (cf_stem0.hip)
            // bit 1 = XCD-aware tile order: it cuts the stem's input fetch from 252 to 79 MB per launch (the excess is halo
            // lines fetched by up to three XCD L2s, served by the Infinity Cache).  The kernel is VALU-bound: alone on the
            // chip it runs 2-3 % slower with it (profiles/r02_ablation.md), with two batches in flight (EngineRing, the
            // benchmarked schedule) there is no difference (44.8k img/s either way) and the fabric traffic is a third: on by
            // default, CF_XCD_ORDER=0 switches it off
            static const bool swz_on = cf_env_int("CF_XCD_ORDER", 1) >= 1;      // product switch
            static const bool mx_off = cf_env_int("CF_DW_MATRIX", 1) == 0 || cf_ab_int("CF_STEM_MX", 1) == 0;      // product switch: depthwise on the matrix cores
            const bool mx = px && !mx_off;
            op.geo.kind = px ? ((swz_on ? 3 : 1) | (mx ? 4 : 0)) : 0;
            std::vector<char> w(px ? stem0px_wstem_bytes() : stem_packed_bytes(dt)), wp(stem0_proj_bytes(dt));
            std::vector<float> wd(mx ? stem0mx_wdw_dwords() : px ? stem0px_wdw_dwords() : 9 * 32), lut(768);
            if (mx) {
                stem0mx_pack(ws.f(op.wkey), ws.f(op.wkey_dw), ws.f(op.wkey_proj), w.data(), reinterpret_cast<uint32_t*>(wd.data()), wp.data());
            } else if (px) {
                stem0px_pack(ws.f(op.wkey), ws.f(op.wkey_dw), ws.f(op.wkey_proj), w.data(), reinterpret_cast<uint32_t*>(wd.data()), wp.data());
            } else if (dt == CF_F32_SPLIT) {
                // Swish factors folded into the weights (stem0_kernel<sp32_t>: swish2_sel<true>): -log2(e) into the stem conv, -ln 2 into the project conv
                std::vector<float> w_s(ws.f(op.wkey), ws.f(op.wkey) + 32 * 27), p_s(ws.f(op.wkey_proj), ws.f(op.wkey_proj) + 16 * 32);
                for (float& v : w_s) v *= kCfNegLog2e;
                for (float& v : p_s) v *= kCfNegLn2;
                stem_pack_weights(dt, w_s.data(), w.data());
                dw_pack_weights(ws.f(op.wkey_dw), 32, 3, wd.data());
                stem0_pack_proj(dt, p_s.data(), wp.data());
            } else {
                stem_pack_weights(dt, ws.f(op.wkey), w.data());
                dw_pack_weights(ws.f(op.wkey_dw), 32, 3, wd.data());
                stem0_pack_proj(dt, ws.f(op.wkey_proj), wp.data());
            }
            stem0_lut(lut.data());
            int r = upload_bytes(c, w, &op.wp); if (r) return r;
            r = upload(c, wd, &op.wdw); if (r) return r;
            r = upload_bytes(c, wp, &op.wproj); if (r) return r;
            r = upload(c, lut, &op.upw); if (r) return r;
        } else if (op.kind == OP_EXPDW) {
            std::vector<char> we(op.geo.wexp_bytes);
            std::vector<float> wd(op.geo.wdw_floats);
            mb_pack_weights(dt, op.geo, op.Cin, op.hid, op.hid, op.k, ws.f(op.wkey), ws.f(op.wkey_dw), nullptr, we.data(), wd.data(), nullptr);
            int r = upload_bytes(c, we, &op.wexp); if (r) return r;
            r = upload(c, wd, &op.wdw); if (r) return r;
        } else if (op.kind == OP_MB) {
            std::vector<char> we(op.geo.wexp_bytes), wp(op.geo.wproj_bytes);
            std::vector<float> wd(op.geo.wdw_floats);
            mb_pack_weights(dt, op.geo, op.Cin, op.hid, op.Cout, op.k, ws.f(op.wkey), ws.f(op.wkey_dw), ws.f(op.wkey_proj),
                            we.data(), wd.data(), wp.data());
            int r = upload_bytes(c, we, &op.wexp); if (r) return r;
            r = upload(c, wd, &op.wdw); if (r) return r;
            r = upload_bytes(c, wp, &op.wproj); if (r) return r;
        } else if (op.kind == OP_DW) {
            std::vector<float> w((size_t)op.k * op.k * op.Cin);
            dw_pack_weights(ws.f(op.wkey), op.Cin, op.k, w.data());
            float* d; int r = upload(c, w, &d); if (r) return r;
            op.wp = d;
        } else if (op.kind == OP_PW) {
            const int K = op.Cin, N = op.Cout;
            std::vector<float> w(ws.f(op.wkey), ws.f(op.wkey) + (size_t)K * N);
            if (!op.bnkey.empty()) {
                std::vector<double> sc, sh;
                bn_fold(ws, op.bnkey, N, op.bn_eps, sc, sh);
                for (int nn = 0; nn < N; ++nn)
                    for (int kk = 0; kk < K; ++kk) w[(size_t)nn * K + kk] = (float)((double)w[(size_t)nn * K + kk] * sc[nn]);
                std::vector<float> b(N);
                for (int nn = 0; nn < N; ++nn) b[nn] = (float)sh[nn];
                int r = upload(c, b, &op.bias); if (r) return r;
            }
            if (!op.upkey.empty()) {
                std::vector<double> sc, sh;
                bn_fold(ws, op.upbnkey, N, op.bn_eps, sc, sh);
                const float* wu = ws.f(op.upkey);                 // [C][1][2][2]
                std::vector<float> uw(4 * N), ub(N);
                for (int cc = 0; cc < N; ++cc) {
                    for (int tap = 0; tap < 4; ++tap) uw[(size_t)tap * N + cc] = (float)((double)wu[cc * 4 + tap] * sc[cc]);
                    ub[cc] = (float)sh[cc];
                }
                int r = upload(c, uw, &op.upw); if (r) return r;
                r = upload(c, ub, &op.upb); if (r) return r;
            }
            std::vector<char> packed(pw_packed_bytes(dt, K, N));
            pw_pack_weights(dt, w.data(), K, N, packed.data());
            int r = upload_bytes(c, packed, &op.wp); if (r) return r;
        } else {
            const char* hn[4] = {"hm", "wh", "lm", "reg"}; const int hc[4] = {1, 2, 10, 2};
            std::vector<float> w0(4 * 24 * 216), b0(96), w1(15 * 24), b1(15);
            int o = 0;
            for (int h = 0; h < 4; ++h) {
                std::string p = hn[h];
                memcpy(&w0[(size_t)h * 24 * 216], ws.f(p + ".0.weight"), sizeof(float) * 24 * 216);
                memcpy(&b0[h * 24], ws.f(p + ".0.bias"), sizeof(float) * 24);
                memcpy(&w1[(size_t)o * 24], ws.f(p + ".1.weight"), sizeof(float) * hc[h] * 24);
                memcpy(&b1[o], ws.f(p + ".1.bias"), sizeof(float) * hc[h]);
                o += hc[h];
            }
            const int col = (c->flags & CF_FLAG_COLLAPSE_HEADS) ? (op.partner >= 0 ? 2 : 1) : 0;   // 2: row order of the fused up3+heads kernel
            std::vector<char> packed(head_packed_bytes(dt, col));
            std::vector<float> b0h(96), w1d(96 * 16), b1h(16);
            head_pack_weights(dt, col, w0.data(), b0.data(), w1.data(), b1.data(), packed.data(), b0h.data(), w1d.data(), b1h.data());
            int r = upload_bytes(c, packed, &op.wp); if (r) return r;
            r = upload(c, b0h, &op.bias); if (r) return r;
            r = upload(c, w1d, &op.w1d); if (r) return r;
            r = upload(c, b1h, &op.b1); if (r) return r;
        }
    }
    c->weights_loaded = true;
    return CF_OK;
}

}  // extern "C"

namespace {

// img0 > 0: a SUB-BATCH of an expand+depthwise / project-GEMM pair (launch_plan_range): the block input, residual and output
// are addressed from image img0, the depthwise tensor between the two launches always from the start of its buffer (every
// sub-batch reuses the same few MB, which the Infinity Cache can keep between the write and the read)
hipError_t launch_plan_at(cf_ctx* c, size_t i, const void* net_in, int in_format, int B, int* consumed);
// The decode stream (device-output top-K decode and the gather records run on it, underneath the next forward) exists from its first use:
// contexts that only ever decode on their main stream (threshold decode: CenterFace.__call__, CenterFaceBuckets) never own one, and every
// stream a process creates takes a share of HIP's four hardware queues (five two-stream contexts: three main streams on one queue).
int ensure_decode_stream(cf_ctx* c) {
    if (c->stream2) return CF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, make_ctx_stream(c, &c->stream2));
    return CF_OK;
}

hipError_t launch_op(cf_ctx* c, const Op& op, const void* net_in, int in_format, int B, int img0 = 0) {
    auto bp = [&](int id) -> void* { return id < 0 ? nullptr : c->bufs[id].p; };
    // byte offset of image img0 in a [B][H][W][C] tensor, NHWC or pixel-block order (whole 32-pixel blocks: checked by the caller)
    auto ioff = [&](int H, int W, int C) -> size_t { return (size_t)img0 * H * W * C * elem_size(c->dtype); };
    auto at = [&](int id, int H, int W, int C) -> void* { return id < 0 ? nullptr : (char*)c->bufs[id].p + ioff(H, W, C); };
    if (op.fused_away) return hipSuccess;
    if (op.kind == OP_HEAD && op.partner >= 0) {
        const Op& u = c->ops[op.partner];
        UpHeadParams p{}; p.skip = bp(u.in); p.low = bp(u.low); p.wcv = u.wp; p.bias = u.bias; p.upw = u.upw; p.upb = u.upb;
        p.w0p = op.wp; p.b0 = op.bias; p.heads = (float*)bp(op.out); p.hm_plane = c->hm_plane; p.B = B; p.h = op.Hout; p.w = op.Wout;
        return launch_uphead(c->stream, c->dtype, p);
    }
    if (op.neck_cl >= 0) {
        const Op& cl = c->ops[op.neck_cl]; const Op& u1 = c->ops[op.neck_u1];
        NeckParams p{}; p.x = bp(cl.in); p.skip1 = bp(u1.in); p.skip2 = bp(op.in);
        p.x_blk = cl.in_blk ? 1 : 0; p.skip1_blk = u1.in_blk ? 1 : 0; p.skip2_blk = op.in_blk ? 1 : 0;
        p.w0 = cl.wp; p.b0 = cl.bias;
        p.w1 = u1.wp; p.b1 = u1.bias; p.upw1 = u1.upw; p.upb1 = u1.upb;
        p.w2 = op.wp; p.b2 = op.bias; p.upw2 = op.upw; p.upb2 = op.upb;
        p.y = bp(op.out); p.B = B; p.h = cl.Hout; p.w = cl.Wout;
        return launch_neck(c->stream, c->dtype, p);
    }
    switch (op.kind) {
        case OP_STEM: {
            StemParams p{}; p.x = net_in; p.in_format = in_format; p.w = op.wp; p.y = bp(op.out);
            p.B = B; p.H = op.Hin; p.W = op.Win;
            return launch_stem(c->stream, c->dtype, p);
        }
        case OP_DW: {
            DwParams p{}; p.x = bp(op.in); p.w = (const float*)op.wp; p.bias = nullptr; p.y = bp(op.out);
            p.B = B; p.C = op.Cin; p.H = op.Hin; p.W = op.Win; p.Ho = op.Hout; p.Wo = op.Wout;
            p.k = op.k; p.s = op.s; p.pad_lo = op.pad_lo; p.act = op.act;
            return launch_dw(c->stream, c->dtype, p);
        }
        case OP_PW: {
            PwParams p{}; p.x = bp(op.in); p.wp = op.wp; p.bias = op.bias; p.res = bp(op.res); p.y = bp(op.out);
            if (img0) { p.res = at(op.res, op.Hout, op.Wout, op.Cout); p.y = at(op.out, op.Hout, op.Wout, op.Cout); }      // x: the depthwise tensor, from its start
            p.M = (long long)B * op.Hout * op.Wout; p.K = op.Cin; p.N = op.Cout; p.act = op.act;
            p.low = bp(op.low); p.upw = op.upw; p.upb = op.upb; p.Ho = op.Hout; p.Wo = op.Wout;
            p.xblock = op.in_blk ? 1 : 0; p.yblock = op.out_blk ? 1 : 0; p.resblock = op.res_blk ? 1 : 0;
            return launch_pw(c->stream, c->dtype, p);
        }
        case OP_STEM0: {
            Stem0Params p{}; p.x = net_in; p.in_format = in_format; p.lut = op.upw; p.wstem = op.wp; p.wdw = op.wdw;
            p.wproj = op.wproj; p.y = bp(op.out); p.B = B; p.H = op.Hin; p.W = op.Win; p.kind = op.geo.kind;
            return launch_stem0(c->stream, c->dtype, p);
        }
        case OP_EXPDW:
        case OP_MB: {
            MbParams p{}; p.x = bp(op.in); p.y = bp(op.out); p.wexp = op.wexp; p.wdw = op.wdw; p.wproj = op.wproj;
            if (img0) p.x = at(op.in, op.Hin, op.Win, op.Cin);                                                              // y: the depthwise tensor, from its start
            p.B = B; p.Hin = op.Hin; p.Win = op.Win; p.Hout = op.Hout; p.Wout = op.Wout; p.Cin = op.Cin; p.hid = op.hid; p.Cout = op.Cout;
            p.k = op.k; p.s = op.s; p.pad_lo = op.pad_lo; p.residual = op.residual ? 1 : 0;
            p.HC = op.geo.HC; p.nq = op.geo.nq; p.NBE = op.geo.NBE; p.JX = op.geo.JX; p.HALF = op.geo.HALF; p.rowb = op.geo.rowb;
            p.lds_bytes = op.geo.lds_bytes; p.kind = op.geo.kind; p.yblock = op.out_blk ? 1 : 0; p.xblock = op.in_blk ? 1 : 0;
            return launch_mbconv(c->stream, c->dtype, p);
        }
        case OP_HEAD: {
            HeadParams p{}; p.x = bp(op.in); p.w0p = op.wp; p.b0 = op.bias; p.w1d = op.w1d; p.b1 = op.b1;
            p.heads = (float*)bp(op.out); p.hm_plane = c->hm_plane; p.B = B; p.h = op.Hout; p.w = op.Wout;
            p.collapsed = (c->flags & CF_FLAG_COLLAPSE_HEADS) ? 1 : 0;
            return launch_heads(c->stream, c->dtype, p);
        }
    }
    return hipErrorInvalidValue;
}

// algorithmic HBM bytes of one launch: unpadded input(s) read once + output written once
double op_bytes(const cf_ctx* c, const Op& op, int in_format, int B) {
    const double es = (double)elem_size(c->dtype);
    double in_b, out_b;
    if (op.kind == OP_STEM || op.kind == OP_STEM0) in_b = (double)op.Hin * op.Win * 3 * (in_format == CF_IN_U8_HWC_BGR ? 1 : 4);
    else in_b = (double)op.Hin * op.Win * op.Cin * es;
    if (op.kind == OP_HEAD) out_b = (double)op.Hout * op.Wout * 16 * 4;
    else out_b = (double)op.Hout * op.Wout * op.Cout * es;
    if (op.kind == OP_HEAD && op.partner >= 0) in_b += (double)(op.Hout / 2) * (op.Wout / 2) * 24 * es;   // + the low IDAUp input
    if (op.neck_cl >= 0) {                                   // fused neck: layer6 + layer4 skip + layer2 skip in, up2 out (no low input from HBM)
        const Op& cl = c->ops[op.neck_cl]; const Op& u1 = c->ops[op.neck_u1];
        in_b += (double)cl.Hin * cl.Win * cl.Cin * es + (double)u1.Hin * u1.Win * u1.Cin * es;
        return (in_b + out_b) * B;
    }
    if (op.res >= 0) in_b += (double)op.Hout * op.Wout * op.Cout * es;
    if (op.low >= 0) in_b += (double)(op.Hout / 2) * (op.Wout / 2) * op.Cout * es;
    return (in_b + out_b) * B;
}

int stage_input(cf_ctx* c, const void* in, int in_format, int in_on_device, int B, const void** net_in) {
    if (!c->weights_loaded) return c->fail(CF_ESTATE, "cf_forward before cf_load_weights");
    if (!in || B < 1 || B > c->max_batch) return c->fail(CF_EINVAL, "cf_forward: B=%d outside [1, %d] or null input", B, c->max_batch);
    if (in_format != CF_IN_U8_HWC_BGR && in_format != CF_IN_F32_NCHW) return c->fail(CF_EINVAL, "unknown input format %d", in_format);
    HIPCHK(c, hipSetDevice(c->device));
    *net_in = in;
    if (in_on_device && (reinterpret_cast<uintptr_t>(in) & 3))
        return c->fail(CF_EINVAL, "cf_forward: device input must be 4-byte aligned (the stem reads it as dwords)");
    if (!in_on_device) {
        size_t bytes = (size_t)B * 3 * c->H * c->W * (in_format == CF_IN_U8_HWC_BGR ? 1 : 4);
        // only the u8 format fits the second staging buffer; f32 NCHW input (tests) keeps the single buffer
        const bool two = in_format == CF_IN_U8_HWC_BGR;
        if (bytes < ((size_t)8 << 20)) {
            // small batches (single-image calls: 1.2 MB): the copy goes on the MAIN stream into slot 0 -- there is nothing to
            // overlap it with, and the copy stream's event round trip costs 0.3 ms of latency (CenterFace(640,640)(img):
            // 0.81 -> 0.5 ms).  Stream order covers earlier forwards; a later copy-stream transfer into slot 0 waits for
            // ev_slot_free[0] like after any other forward.
            void* dst = c->bufs[c->buf_in].p;
            if (c->slot_busy[0]) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_slot_free[0], 0));
            HIPCHK(c, hipMemcpyAsync(dst, in, bytes, hipMemcpyHostToDevice, c->stream));
            c->in_slot_used = 0;
            *net_in = dst;
            return CF_OK;
        }
        const int slot = two ? (c->in_slot ^= 1) : 0;
        void* dst = c->bufs[slot == 0 ? c->buf_in : c->buf_in2].p;
        if (c->slot_busy[slot]) HIPCHK(c, hipStreamWaitEvent(c->stream_in, c->ev_slot_free[slot], 0));   // its last reader (a stem) is done
        HIPCHK(c, hipMemcpyAsync(dst, in, bytes, hipMemcpyHostToDevice, c->stream_in));
        HIPCHK(c, hipEventRecord(c->ev_copy[slot], c->stream_in));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copy[slot], 0));
        c->in_slot_used = slot;
        *net_in = dst;
    }
    return CF_OK;
}

// get_affine_transform(center, scale, rot=0, output_size=(out_w, out_h), inv=1) (utils/image.py:27-60):
// the three point pairs are built in float32 exactly as the numpy code does, then the 2x3 map
// dst -> src is solved in float64 (Cramer) where the reference calls cv2.getAffineTransform.
void inverse_affine(float cx, float cy, float sw, int out_w, int out_h, double t[6]) {
    const float dw = (float)out_w, dh = (float)out_h;
    // dst triangle (heat-map side)
    const float d0x = dw * 0.5f, d0y = dh * 0.5f;
    const float d1x = d0x + 0.0f, d1y = d0y + dw * -0.5f;
    const float d2x = d1x - (d0y - d1y), d2y = d1y + (d0x - d1x);
    // src triangle (image side)
    const float s0x = cx, s0y = cy;
    const float s1x = (float)((double)cx + 0.0), s1y = (float)((double)cy + (double)(sw * -0.5f));
    const float s2x = s1x - (s0y - s1y), s2y = s1y + (s0x - s1x);
    // solve [dx dy 1] * [a b c]^T = sx (and sy) for the three points
    const double x0 = d0x, y0 = d0y, x1 = d1x, y1 = d1y, x2 = d2x, y2 = d2y;
    const double det = x0 * (y1 - y2) - y0 * (x1 - x2) + (x1 * y2 - x2 * y1);
    auto solve = [&](double u0, double u1, double u2, double* o) {
        o[0] = (u0 * (y1 - y2) - y0 * (u1 - u2) + (u1 * y2 - u2 * y1)) / det;
        o[1] = (x0 * (u1 - u2) - u0 * (x1 - x2) + (x1 * u2 - x2 * u1)) / det;
        o[2] = (x0 * (y1 * u2 - y2 * u1) - y0 * (x1 * u2 - x2 * u1) + u0 * (x1 * y2 - x2 * y1)) / det;
    };
    solve(s0x, s1x, s2x, t);
    solve(s0y, s1y, s2y, t + 3);
}

int ensure_topk_ws(cf_ctx* c, int K) {
    const size_t HW = (size_t)(c->H / 4) * (c->W / 4);
    if (!c->keys) HIPCHK(c, hipMalloc((void**)&c->keys, HW * c->max_batch * sizeof(unsigned long long)));
    if (!c->key_count) {
        HIPCHK(c, hipMalloc((void**)&c->key_count, (size_t)c->max_batch * kTopkCountStride * sizeof(int)));
        HIPCHK(c, hipMemsetAsync(c->key_count, 0, (size_t)c->max_batch * kTopkCountStride * sizeof(int), c->stream));   // the select kernel leaves it zero
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (c->decK < K) {
        HIPCHK(c, hipStreamSynchronize(c->stream));           // a decode in flight may still write the old buffers
        if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
        if (c->gather_pending) { HIPCHK(c, hipEventSynchronize(c->ev_gather)); c->gather_pending = false; }   // ... and a gather may still read d_rec
        for (void* p : {(void*)c->d_dets, (void*)c->d_lms, (void*)c->d_inds, (void*)c->d_slot, (void*)c->big}) if (p) hipFree(p);
        c->d_dets = nullptr; c->d_lms = nullptr; c->d_inds = nullptr; c->d_slot = nullptr; c->d_rec = nullptr; c->big = nullptr; c->big_stride = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_dets, (size_t)c->max_batch * K * 6 * sizeof(float)));
        HIPCHK(c, hipMalloc((void**)&c->d_lms, (size_t)c->max_batch * K * 10 * sizeof(float)));
        HIPCHK(c, hipMalloc((void**)&c->d_inds, (size_t)c->max_batch * K * sizeof(long long)));
        // the gather records live behind a 16-float header: [header | B x K x 16] is the slot one rank sends per gather step
        HIPCHK(c, hipMalloc((void**)&c->d_slot, ((size_t)c->max_batch * K * 16 + kSlotHeader) * sizeof(float)));
        c->d_rec = c->d_slot + kSlotHeader;
        if (K > 1024) {
            c->big_stride = topk_big_stride(K);
            HIPCHK(c, hipMalloc((void**)&c->big, (size_t)c->max_batch * c->big_stride * sizeof(unsigned long long)));
        }
        c->decK = K;
    }
    return CF_OK;
}

// One scratch list serves both streams: a decode on the main stream waits for an overlapped decode still running on
// the decode stream and vice versa (events, no host sync).
int enqueue_topk(cf_ctx* c, int B, int K, int use_reg, float* dets, float* lms, long long* inds, const double* trans = nullptr,
                 hipStream_t on = nullptr, float* rec16 = nullptr) {
    TopkParams p{};
    p.trans = trans;
    p.heads = (const float*)c->bufs[c->buf_heads].p; p.hm_plane = c->hm_plane; p.scratch = c->keys; p.count = c->key_count;
    p.big = c->big; p.big_stride = c->big_stride;
    p.B = B; p.h = c->H / 4; p.w = c->W / 4; p.K = K; p.use_reg = use_reg;
    p.dets = dets; p.lms = lms; p.inds = inds; p.rec16 = rec16;
    const bool on_main = (on == nullptr || on == c->stream);
    if (on_main) {
        if (c->dec_pending) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_dec, 0));
    } else if (c->main_dec_pending) {
        HIPCHK(c, hipStreamWaitEvent(on, c->ev_main_dec, 0));
        c->main_dec_pending = false;
    }
    HIPCHK(c, launch_peak_topk(on_main ? c->stream : on, p));
    if (on_main) { HIPCHK(c, hipEventRecord(c->ev_main_dec, c->stream)); c->main_dec_pending = true; }
    return CF_OK;
}

// Everything in front of the head kernel as one executable graph (the head kernel stays an eager
// launch: it may have to wait for an overlapped decode of the previous batch, a dependency on work
// outside the capture).  Returns nullptr when this key has not been seen twice yet or cannot be captured.
hipGraphExec_t forward_graph(cf_ctx* c, const void* net_in, int in_format, int B) {
    constexpr size_t kMaxGraphs = 16;
    cf_ctx::FwdGraph* g = nullptr;
    for (auto& e : c->graphs) if (e.in == net_in && e.fmt == in_format && e.B == B) g = &e;
    if (!g) {
        if (c->graphs.size() >= kMaxGraphs) {                         // evict the least recently used
            size_t lru = 0;
            for (size_t i = 1; i < c->graphs.size(); ++i) if (c->graphs[i].used < c->graphs[lru].used) lru = i;
            if (c->graphs[lru].exec) hipGraphExecDestroy(c->graphs[lru].exec);
            c->graphs.erase(c->graphs.begin() + lru);
        }
        c->graphs.push_back({net_in, in_format, B, nullptr, false, ++c->graph_clock});
        return nullptr;                                               // first sighting: eager (also does the one-time kernel attribute setup)
    }
    g->used = ++c->graph_clock;
    if (g->exec || g->broken) return g->exec;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); g->broken = true; return nullptr; }
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < c->ops.size();) {
        if (c->ops[i].kind == OP_HEAD) break;
        int used = 1;
        if ((e = launch_plan_at(c, i, net_in, in_format, B, &used)) != hipSuccess) break;
        i += used;
    }
    hipError_t e2 = hipStreamEndCapture(c->stream, &graph);
    if (e == hipSuccess && e2 == hipSuccess && graph && hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) == hipSuccess) {
        hipGraphDestroy(graph);
        return g->exec;
    }
    if (graph) hipGraphDestroy(graph);
    (void)hipGetLastError();
    g->exec = nullptr; g->broken = true;
    return nullptr;
}

// One plan entry -- or, with sub-batching on (experiments build: CF_SUBBATCH=n), an expand+depthwise launch TOGETHER with the project
// GEMM behind it, n images at a time.  VERDICT r04 next-5: the depthwise tensor of layer4.0-6.0 is the one 6x tensor that reaches
// HBM (1.04 GB per batch of 64 in bf16); n = 16 keeps it at <= 30 MB per round trip.  Returns the number of plan entries consumed.
hipError_t launch_plan_at(cf_ctx* c, size_t i, const void* net_in, int in_format, int B, int* consumed) {
    static const int sub = cf_ab_int("CF_SUBBATCH", 0);
    const Op& op = c->ops[i];
    *consumed = 1;
    if (sub > 0 && op.kind == OP_EXPDW && i + 1 < c->ops.size() && c->ops[i + 1].kind == OP_PW && c->ops[i + 1].in == op.out && B > sub && B % sub == 0 &&
        ((long long)sub * op.Hin * op.Win) % 32 == 0 && ((long long)sub * op.Hout * op.Wout) % 32 == 0) {
        const Op& pr = c->ops[i + 1];
        for (int img0 = 0; img0 < B; img0 += sub) {
            hipError_t e = launch_op(c, op, net_in, in_format, sub, img0);
            if (e == hipSuccess) e = launch_op(c, pr, net_in, in_format, sub, img0);
            if (e != hipSuccess) return e;
        }
        *consumed = 2;
        return hipSuccess;
    }
    return launch_op(c, op, net_in, in_format, B);
}

int launch_all_ops(cf_ctx* c, const void* net_in, int in_format, int B) {
    c->thr_pending = false;                               // an enqueued threshold decode belongs to the forward before this one
    c->up.pending = false;                                // ... and so does an upload nobody asked to run
    hipGraphExec_t exec = (c->flags & CF_FLAG_NO_GRAPH) ? nullptr : forward_graph(c, net_in, in_format, B);
    if (exec) HIPCHK(c, hipGraphLaunch(exec, c->stream));
    for (size_t i = 0; i < c->ops.size();) {
        const Op& op = c->ops[i];
        if (op.kind == OP_HEAD && c->dec_pending) {       // the overlapped decode still reads heads / hm_plane
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_dec, 0));
            c->dec_pending = false;
        }
        int used = 1;
        if (!(exec && op.kind != OP_HEAD)) HIPCHK(c, launch_plan_at(c, i, net_in, in_format, B, &used));
        i += used;
    }
    HIPCHK(c, hipEventRecord(c->ev_fwd, c->stream));
    if (c->in_slot_used >= 0) {                     // this forward read a host-input staging slot: mark when it is free again
        HIPCHK(c, hipEventRecord(c->ev_slot_free[c->in_slot_used], c->stream));
        c->slot_busy[c->in_slot_used] = true;
        c->in_slot_used = -1;
    }
    return CF_OK;
}

}  // namespace

extern "C" {

#include CF_EXP_INC(cf_runtime_1)
#if !CF_EXP_ON
#define CF_FLUSH_LANE(c) do { } while (0)
#endif

// Host images of a size other than the network's land in src_stage first (the resize kernel reads them from there).  The copies go
// on the device's ONE copy stream, like the network-sized batches of stage_input: with several contexts in flight (CenterFaceBuckets:
// one per network shape) copies issued on each context's own stream share the PCIe link, ALL finish late and no forward can start
// under them; queued on one stream the first chunk is complete after 1/n of the time and its forward runs under the other copies.
// Small batches (single-image calls) stay on the main stream: nothing to overlap, and the event round trip costs 0.3 ms of latency.
static hipStream_t src_stage_stream(cf_ctx* c, size_t bytes) { return bytes < ((size_t)8 << 20) ? c->stream : c->stream_in; }
static int src_stage_begin(cf_ctx* c, size_t bytes) {
    if (!c->ev_src_copy) HIPCHK(c, hipEventCreateWithFlags(&c->ev_src_copy, hipEventDisableTiming));
    if (!c->ev_src_free) HIPCHK(c, hipEventCreateWithFlags(&c->ev_src_free, hipEventDisableTiming));
    if (c->src_stage_bytes < bytes) {
        if (c->src_stage) HIPCHK(c, hipFree(c->src_stage));        // (a device-wide synchronisation: no reader is left)
        c->src_stage = nullptr; c->src_stage_bytes = 0; c->src_busy = false;
        HIPCHK(c, hipMalloc((void**)&c->src_stage, bytes));
        c->src_stage_bytes = bytes;
    }
    hipStream_t cs = src_stage_stream(c, bytes);
    if (c->src_busy && cs != c->stream) HIPCHK(c, hipStreamWaitEvent(cs, c->ev_src_free, 0));      // the resize of the batch before has read it
    return CF_OK;
}
static int src_stage_resize(cf_ctx* c, size_t bytes, uint8_t* dst, int B, int h, int w) {
    if (src_stage_stream(c, bytes) != c->stream) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_src_copy, 0));      // recorded behind the copies
    HIPCHK(c, launch_resize_u8(c->stream, c->src_stage, dst, B, h, w, c->H, c->W));
    HIPCHK(c, hipEventRecord(c->ev_src_free, c->stream));
    c->src_busy = true;
    return CF_OK;
}

int cf_forward(cf_ctx* c, const void* in, int in_format, int in_on_device, int B) {
    if (!c) return CF_EINVAL;
    CF_FLUSH_LANE(c);
    const void* net_in = nullptr;
    int r = stage_input(c, in, in_format, in_on_device, B, &net_in);
    if (r) return r;
    r = launch_all_ops(c, net_in, in_format, B);
    if (r) return r;
    c->last_B = B;
    return CF_OK;
}

#include CF_EXP_INC(cf_runtime_2)   // measured 2-8 % slower than the free-running pair of contexts: not in the release library

int cf_forward_resized(cf_ctx* c, const void* imgs, int in_on_device, int B, int h, int w) {
    if (!c || !imgs || h < 1 || w < 1) return CF_EINVAL;
    if (!c->weights_loaded) return c->fail(CF_ESTATE, "cf_forward_resized before cf_load_weights");
    if (B < 1 || B > c->max_batch) return c->fail(CF_EINVAL, "cf_forward_resized: B=%d outside [1, %d]", B, c->max_batch);
    HIPCHK(c, hipSetDevice(c->device));
    CF_FLUSH_LANE(c);
    uint8_t* dst = (uint8_t*)c->bufs[c->buf_resized].p;
    if (!in_on_device) {
        const size_t bytes = (size_t)B * h * w * 3;
        int r = src_stage_begin(c, bytes); if (r) return r;
        HIPCHK(c, hipMemcpyAsync(c->src_stage, imgs, bytes, hipMemcpyHostToDevice, src_stage_stream(c, bytes)));
        if (src_stage_stream(c, bytes) != c->stream) HIPCHK(c, hipEventRecord(c->ev_src_copy, c->stream_in));
        r = src_stage_resize(c, bytes, dst, B, h, w); if (r) return r;
    } else {
        HIPCHK(c, launch_resize_u8(c->stream, (const uint8_t*)imgs, dst, B, h, w, c->H, c->W));
    }
    int r = launch_all_ops(c, dst, CF_IN_U8_HWC_BGR, B);
    if (r) return r;
    c->last_B = B;
    return CF_OK;
}

// The batch as B separate host images (one pointer each) instead of one [B,h,w,3] block: what a caller holds after B cv2.imread calls
// (eval_widerface.py:76-90).  Each image is its own asynchronous DMA into the context's device buffer -- from page-locked memory
// (cf_host_alloc / cf_host_register) without any host-side staging copy, which was the largest host cost of a mixed-size batch
// (2.7 of 5.8 ms per 128 VGA images).  Pageable pointers are accepted too (the runtime stages them: correct, slower, and the call
// then returns only when the copies have left the caller's memory).
int cf_upload_images(cf_ctx* c, const void* const* imgs, int B, int h, int w) {
    if (!c || !imgs || h < 1 || w < 1) return CF_EINVAL;
    if (!c->weights_loaded) return c->fail(CF_ESTATE, "cf_upload_images before cf_load_weights");
    if (B < 1 || B > c->max_batch) return c->fail(CF_EINVAL, "cf_upload_images: B=%d outside [1, %d]", B, c->max_batch);
    for (int b = 0; b < B; ++b) if (!imgs[b]) return c->fail(CF_EINVAL, "cf_upload_images: image %d is a null pointer", b);
    HIPCHK(c, hipSetDevice(c->device));
    CF_FLUSH_LANE(c);
    const size_t one = (size_t)h * w * 3;
    // page-locked images (device-visible, 16-byte aligned): ONE kernel reads them all over PCIe instead of one DMA command each
    std::vector<const void*> dev;
    static const bool use_kernel = cf_ab_int("CF_UPLOAD_KERNEL", 0) != 0;      // measured: the kernel reaches 53 GB/s but doubles the time of forwards running beside it
    if (use_kernel && one % 16 == 0) {
        dev.resize(B);
        for (int b = 0; b < B; ++b) {
            void* dp = nullptr;
            if ((reinterpret_cast<uintptr_t>(imgs[b]) & 15) || hipHostGetDevicePointer(&dp, const_cast<void*>(imgs[b]), 0) != hipSuccess || !dp ||
                (reinterpret_cast<uintptr_t>(dp) & 15)) {
                (void)hipGetLastError();
                dev.clear();
                break;
            }
            dev[b] = dp;
        }
    }
    // One DMA command per run of images that are adjacent in host memory (a caller's frame pool often is).  A command costs the engine
    // 8-10 us of idle link on top of ~22 us per VGA image, so large batches alternate between the device's two copy streams (two
    // engines): one's set-up hides under the other's transfer.  Both streams wait for `after` (the buffer's last reader), each records
    // its own event behind its copies, and the forward waits for both -- no stream waits for the other (every such hop costs ~0.1 ms).
    bool dual = false;
    auto copy_in = [&](uint8_t* dst, hipStream_t cs, hipEvent_t after) -> hipError_t {
        hipError_t e = hipSuccess;
        if (after && (e = hipStreamWaitEvent(cs, after, 0)) != hipSuccess) return e;
        if (!dev.empty()) return launch_upload_images(cs, dev.data(), dst, B, (long long)one);
        static const bool dual_ok = cf_ab_int("CF_COPY_DUAL", 1) != 0;
        dual = dual_ok && cs == c->stream_in && B > 1;
        if (dual) {
            if (!c->stream_in2 && (e = second_copy_stream(c->device, &c->stream_in2)) != hipSuccess) return e;
            if (!c->ev_copy2 && (e = hipEventCreateWithFlags(&c->ev_copy2, hipEventDisableTiming)) != hipSuccess) return e;
            if (after && (e = hipStreamWaitEvent(c->stream_in2, after, 0)) != hipSuccess) return e;
        }
        int k = 0;
        for (int b = 0; b < B;) {
            int n = 1;
            while (b + n < B && (const uint8_t*)imgs[b + n] == (const uint8_t*)imgs[b] + (size_t)n * one) ++n;
            e = hipMemcpyAsync(dst + b * one, imgs[b], one * n, hipMemcpyHostToDevice, (dual && (k++ & 1)) ? c->stream_in2 : cs);
            if (e != hipSuccess) return e;
            b += n;
        }
        if (dual && k < 2) dual = false;                    // (everything was one run: the second stream got nothing)
        if (dual && (e = hipEventRecord(c->ev_copy2, c->stream_in2)) != hipSuccess) return e;
        return hipSuccess;
    };
    if (h == c->H && w == c->W) {                           // network-sized: straight into an input slot, on the copy stream
        static const bool own_stream = cf_ab_int("CF_UPLOAD_STREAM", 1) == 0;      // A/B: uploads on the context's main stream
        const bool small = own_stream || one * B < ((size_t)8 << 20);             // as in stage_input: slot 0 on the main stream, no event round trip
        const int slot = small ? 0 : (c->in_slot ^= 1);
        hipStream_t cs = small ? c->stream : c->stream_in;
        uint8_t* dst = (uint8_t*)c->bufs[slot == 0 ? c->buf_in : c->buf_in2].p;
        HIPCHK(c, copy_in(dst, cs, c->slot_busy[slot] ? c->ev_slot_free[slot] : nullptr));
        if (!small) HIPCHK(c, hipEventRecord(c->ev_copy[slot], cs));
        c->up = {true, B, h, w, slot, small, dual};
        return CF_OK;
    }
    const size_t bytes = one * B;
    if (!c->ev_src_copy) HIPCHK(c, hipEventCreateWithFlags(&c->ev_src_copy, hipEventDisableTiming));
    if (!c->ev_src_free) HIPCHK(c, hipEventCreateWithFlags(&c->ev_src_free, hipEventDisableTiming));
    if (c->src_stage_bytes < bytes) {
        if (c->src_stage) HIPCHK(c, hipFree(c->src_stage));        // (a device-wide synchronisation: no reader is left)
        c->src_stage = nullptr; c->src_stage_bytes = 0; c->src_busy = false;
        HIPCHK(c, hipMalloc((void**)&c->src_stage, bytes));
        c->src_stage_bytes = bytes;
    }
    hipStream_t cs = src_stage_stream(c, bytes);
    HIPCHK(c, copy_in(c->src_stage, cs, c->src_busy && cs != c->stream ? c->ev_src_free : nullptr));
    if (cs != c->stream) HIPCHK(c, hipEventRecord(c->ev_src_copy, cs));
    c->up = {true, B, h, w, -1, false, dual};
    return CF_OK;
}

int cf_forward_uploaded(cf_ctx* c) {
    if (!c) return CF_EINVAL;
    if (!c->up.pending) return c->fail(CF_ESTATE, "cf_forward_uploaded without a cf_upload_images before it");
    HIPCHK(c, hipSetDevice(c->device));
    const cf_ctx::Upload u = c->up;
    c->up.pending = false;
    const uint8_t* net_in;
    if (u.dual) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copy2, 0));
    if (u.slot >= 0) {
        if (!u.small) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copy[u.slot], 0));
        c->in_slot_used = u.slot;
        net_in = (const uint8_t*)c->bufs[u.slot == 0 ? c->buf_in : c->buf_in2].p;
    } else {
        uint8_t* dst = (uint8_t*)c->bufs[c->buf_resized].p;
        int r = src_stage_resize(c, (size_t)u.B * u.h * u.w * 3, dst, u.B, u.h, u.w); if (r) return r;
        net_in = dst;
    }
    int r = launch_all_ops(c, net_in, CF_IN_U8_HWC_BGR, u.B);
    if (r) return r;
    c->last_B = u.B;
    return CF_OK;
}

int cf_forward_images(cf_ctx* c, const void* const* imgs, int B, int h, int w) {
    int r = cf_upload_images(c, imgs, B, h, w);
    return r ? r : cf_forward_uploaded(c);
}


// centerface.py:55-62 on the device: the threshold decode writes floor(x / scale_w), floor(y / scale_h) for the four box corners and
// the five landmark points (numpy's float32 `//` by a python float: the exact floor of the quotient -- evaluated here as the floor of
// the float64 quotient, which is identical, see CenterFace._floordiv).  0, 0 switches it off (boxes in network coordinates).
int cf_set_rescale(cf_ctx* c, float scale_h, float scale_w) {
    if (!c) return CF_EINVAL;
    if (!(scale_h >= 0.f) || !(scale_w >= 0.f) || (scale_h == 0.f) != (scale_w == 0.f))
        return c->fail(CF_EINVAL, "cf_set_rescale: scales must both be positive, or both 0 (off); got %g, %g", (double)scale_h, (double)scale_w);
    c->rs_h = scale_h; c->rs_w = scale_w;
    return CF_OK;
}

int cf_get_resized_input(cf_ctx* c, void* out_u8, int B) {
    if (!c || !out_u8 || B < 1 || B > c->max_batch) return CF_EINVAL;
    HIPCHK(c, hipMemcpyAsync(out_u8, c->bufs[c->buf_resized].p, (size_t)B * c->H * c->W * 3, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CF_OK;
}

int cf_synchronize(cf_ctx* c) {
    if (!c) return CF_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
    if (c->gather_pending) { HIPCHK(c, hipEventSynchronize(c->ev_gather)); c->gather_pending = false; }   // the communicator's stream
    return CF_OK;
}

int cf_get_heads(cf_ctx* c, float* hm, float* wh, float* lm, float* reg, float* hm_sigmoid) {
    if (!c) return CF_EINVAL;
    if (c->last_B < 1) return c->fail(CF_ESTATE, "cf_get_heads before cf_forward");
    const int B = c->last_B, h = c->H / 4, w = c->W / 4;
    const size_t HW = (size_t)h * w;
    std::vector<float> host((size_t)B * HW * 16);
    HIPCHK(c, hipMemcpyAsync(host.data(), c->bufs[c->buf_heads].p, host.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int b = 0; b < B; ++b)
        for (size_t i = 0; i < HW; ++i) {
            const float* r = &host[((size_t)b * HW + i) * 16];
            if (hm_sigmoid) hm_sigmoid[(size_t)b * HW + i] = r[0];
            if (hm) hm[(size_t)b * HW + i] = r[15];
            if (wh) for (int ch = 0; ch < 2; ++ch) wh[((size_t)b * 2 + ch) * HW + i] = r[1 + ch];
            if (lm) for (int ch = 0; ch < 10; ++ch) lm[((size_t)b * 10 + ch) * HW + i] = r[3 + ch];
            if (reg) for (int ch = 0; ch < 2; ++ch) reg[((size_t)b * 2 + ch) * HW + i] = r[13 + ch];
        }
    return CF_OK;
}

int cf_decode_topk(cf_ctx* c, int K, int use_reg, float* dets, float* lms, int64_t* inds, int out_on_device) {
    if (!c || !dets) return CF_EINVAL;
    if (c->last_B < 1) return c->fail(CF_ESTATE, "cf_decode_topk before cf_forward");
    const int B = c->last_B, HW = (c->H / 4) * (c->W / 4);
    if (K < 1 || K > HW) return c->fail(CF_EINVAL, "K=%d must be in [1, %d]", K, HW);
    HIPCHK(c, hipSetDevice(c->device));
    int r = ensure_topk_ws(c, K); if (r) return r;
    if (out_on_device) {
        static const bool overlap = cf_env_int("CF_DECODE_OVERLAP", 1) != 0;      // product switch
        if (!overlap || (c->flags & CF_FLAG_NO_DECODE_STREAM)) return enqueue_topk(c, B, K, use_reg, dets, lms, (long long*)inds);
        r = ensure_decode_stream(c); if (r) return r;
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fwd, 0));
        r = enqueue_topk(c, B, K, use_reg, dets, lms, (long long*)inds, nullptr, c->stream2);
        if (r) return r;
        HIPCHK(c, hipEventRecord(c->ev_dec, c->stream2));
        c->dec_pending = true;
        return CF_OK;
    }
    r = enqueue_topk(c, B, K, use_reg, c->d_dets, lms ? c->d_lms : nullptr, inds ? c->d_inds : nullptr);
    if (r) return r;
    HIPCHK(c, hipMemcpyAsync(dets, c->d_dets, (size_t)B * K * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (lms) HIPCHK(c, hipMemcpyAsync(lms, c->d_lms, (size_t)B * K * 10 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (inds) HIPCHK(c, hipMemcpyAsync(inds, c->d_inds, (size_t)B * K * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CF_OK;
}

int cf_decode_topk_post(cf_ctx* c, int K, int use_reg, const float* centers, const float* scales, int out_w, int out_h,
                        float* dets, float* lms, int64_t* inds, int out_on_device) {
    if (!c || !dets || !centers || !scales) return CF_EINVAL;
    if (c->last_B < 1) return c->fail(CF_ESTATE, "cf_decode_topk_post before cf_forward");
    const int B = c->last_B, HW = (c->H / 4) * (c->W / 4);
    if (K < 1 || K > HW) return c->fail(CF_EINVAL, "K=%d must be in [1, %d]", K, HW);
    HIPCHK(c, hipSetDevice(c->device));
    int r = ensure_topk_ws(c, K); if (r) return r;
    if (!c->d_trans) HIPCHK(c, hipMalloc((void**)&c->d_trans, (size_t)c->max_batch * 6 * sizeof(double)));
    std::vector<double> t((size_t)B * 6);
    for (int b = 0; b < B; ++b) inverse_affine(centers[2 * b], centers[2 * b + 1], scales[2 * b], out_w, out_h, &t[(size_t)b * 6]);
    HIPCHK(c, hipMemcpyAsync(c->d_trans, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // t is a stack-lifetime host buffer
    if (out_on_device) return enqueue_topk(c, B, K, use_reg, dets, lms, (long long*)inds, c->d_trans);
    r = enqueue_topk(c, B, K, use_reg, c->d_dets, lms ? c->d_lms : nullptr, inds ? c->d_inds : nullptr, c->d_trans);
    if (r) return r;
    HIPCHK(c, hipMemcpyAsync(dets, c->d_dets, (size_t)B * K * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (lms) HIPCHK(c, hipMemcpyAsync(lms, c->d_lms, (size_t)B * K * 10 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (inds) HIPCHK(c, hipMemcpyAsync(inds, c->d_inds, (size_t)B * K * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CF_OK;
}

int cf_affine_from_center_scale(float cx, float cy, float scale_w, int out_w, int out_h, double* trans6) {
    if (!trans6 || out_w < 1 || out_h < 1) return CF_EINVAL;
    inverse_affine(cx, cy, scale_w, out_w, out_h, trans6);
    return CF_OK;
}

int cf_detect_topk(cf_ctx* c, const void* in, int in_format, int in_on_device, int B, int K,
                   float* dets, float* lms, int64_t* inds, int out_on_device) {
    int r = cf_forward(c, in, in_format, in_on_device, B);
    if (r) return r;
    return cf_decode_topk(c, K, 1, dets, lms, inds, out_on_device);
}

static void free_thresh_ws(cf_ctx* c) {
    for (void* p : {(void*)c->t_cand, (void*)c->t_count, (void*)c->t_order, (void*)c->t_mask, (void*)c->t_counts, (void*)c->t_overflow})
        if (p) hipFree(p);
    c->t_cand = nullptr; c->t_count = nullptr; c->t_order = nullptr; c->t_mask = nullptr; c->t_counts = nullptr; c->t_overflow = nullptr;
    c->t_cap = 0; c->t_B = 0;
}
// The suppression matrix is dense: cap x cap / 64 words per image.  The workspace is sized for the batch of the CURRENT call
// (not max_batch) and for the candidate count seen; a decode that needed more than kThreshKeepBytes of it (one noisy batch,
// e.g. untrained weights: 1.3 GB per image at 1280x1280 when every cell passes) releases it again afterwards instead of
// pinning GBs of HBM for the life of the context.  Practical limit: cap^2 / 8 bytes x B must fit the free HBM.
static constexpr size_t kThreshKeepBytes = (size_t)512 << 20;
static size_t thresh_mask_bytes(int B, int cap) { return (size_t)B * cap * ((cap + 63) / 64) * sizeof(unsigned long long); }
static constexpr size_t kThreshHostBytes = (size_t)16 << 20;          // larger result tables (max_out grown after an overflow) stay on the device
static size_t thr_host_head(cf_ctx* c) { return (256 + (size_t)c->max_batch * sizeof(int) + 255) / 256 * 256; }     // [overflow: 256 B][counts]
static int ensure_thresh_ws(cf_ctx* c, int max_out, int cap, int B) {
    if (c->t_cap < cap || c->t_B < B) {
        cap = std::max(cap, c->t_cap); B = std::max(B, c->t_B);
        free_thresh_ws(c);
        const size_t mb = B, words = (cap + 63) / 64;
        hipError_t e = hipMalloc((void**)&c->t_cand, mb * cap * 16 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&c->t_count, mb * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&c->t_order, mb * cap * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&c->t_mask, mb * cap * words * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc((void**)&c->t_counts, mb * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&c->t_overflow, sizeof(int));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            free_thresh_ws(c);
            return c->fail(e == hipErrorOutOfMemory ? CF_ENOMEM : CF_EHIP, "threshold-decode workspace for %d candidates per image x %d images (%.1f MB of suppression bits): %s",
                           cap, B, thresh_mask_bytes(B, cap) / 1e6, hipGetErrorString(e));
        }
        c->t_cap = cap; c->t_B = B;
    }
    if (c->t_maxout < max_out) {
        if (!c->t_host && c->t_dets) hipFree(c->t_dets);
        if (!c->t_host && c->t_lms) hipFree(c->t_lms);
        if (c->h_thr) { HIPCHK(c, hipStreamSynchronize(c->stream)); hipHostFree(c->h_thr); }
        c->t_dets = nullptr; c->t_lms = nullptr; c->h_thr = nullptr; c->t_host = false; c->t_maxout = 0;
        const size_t nd = (size_t)c->max_batch * max_out * 5 * sizeof(float), nl = 2 * nd, head = thr_host_head(c);
        if (nd + nl <= kThreshHostBytes) {
            // the usual case: the sweep kernel writes the few hundred result rows straight into page-locked host memory.  Copy
            // commands cost more than the data here (two synchronising round trips and three staged copies per collected batch,
            // and a hipStreamSynchronize that came back milliseconds late with five contexts in flight: profiles/r05_vga_pipeline.md)
            HIPCHK(c, hipHostMalloc((void**)&c->h_thr, head + nd + nl, hipHostMallocDefault));
            memset(c->h_thr, 0, head);
            c->t_dets = (float*)(c->h_thr + head); c->t_lms = (float*)(c->h_thr + head + nd);
            c->t_host = true;
        } else {
            HIPCHK(c, hipMalloc((void**)&c->t_dets, nd));
            HIPCHK(c, hipMalloc((void**)&c->t_lms, nl));
        }
        c->t_maxout = max_out;
    }
    if (!c->ev_thr) HIPCHK(c, hipEventCreateWithFlags(&c->ev_thr, hipEventDisableTiming));
    return CF_OK;
}

int cf_decode_threshold(cf_ctx* c, float score_thresh, float nms_thresh, int max_out,
                        float* dets, float* lms, int32_t* counts) {
    return cf_decode_threshold_ex(c, 0, score_thresh, nms_thresh, max_out, dets, lms, counts);
}

int cf_decode_threshold_ex(cf_ctx* c, int mode, float score_thresh, float nms_thresh, int max_out,
                           float* dets, float* lms, int32_t* counts) {
    if (!c) return CF_EINVAL;
    return cf_decode_threshold_sized(c, mode, score_thresh, nms_thresh, c->H, c->W, max_out, dets, lms, counts);
}

// the decode kernels of the last forward (threshold compaction, rank, suppression matrix, sweep) on the context's stream; no host wait
static int thresh_launch(cf_ctx* c, int mode, float score_thresh, float nms_thresh, int img_h, int img_w, int max_out, int cap) {
    const int B = c->last_B;
    int r = ensure_thresh_ws(c, max_out, cap, B); if (r) return r;
    ThreshParams p{};
    p.heads = (const float*)c->bufs[c->buf_heads].p; p.hm_plane = c->hm_plane; p.B = B; p.h = c->H / 4; p.w = c->W / 4;
    p.img_h = img_h; p.img_w = img_w; p.score_thresh = score_thresh; p.nms_thresh = nms_thresh; p.cap = c->t_cap; p.mode = mode;
    p.cand = c->t_cand; p.cand_count = c->t_count; p.order = c->t_order; p.mask = c->t_mask;
    p.max_out = max_out; p.dets = c->t_dets; p.lms = c->t_lms; p.counts = c->t_counts; p.overflow = c->t_overflow;
    p.rs_h = c->rs_h; p.rs_w = c->rs_w;
    if (c->t_host) { p.host_overflow = (int*)c->h_thr; p.host_counts = (int*)(c->h_thr + 256); }
    HIPCHK(c, hipMemsetAsync(c->t_overflow, 0, sizeof(int), c->stream));
    HIPCHK(c, launch_decode_threshold(c->stream, p));
    HIPCHK(c, hipEventRecord(c->ev_thr, c->stream));
    return CF_OK;
}

// Asynchronous first half of cf_decode_threshold_sized: the decode kernels are enqueued right behind the forward (so they run as
// soon as it finishes, not when the host gets around to collecting this context); a later cf_decode_threshold_sized with the same
// parameters only waits, checks the overflow flag and copies the results out.  Any other forward / decode on the context in between
// simply makes that call launch its own decode.
int cf_decode_threshold_enqueue(cf_ctx* c, int mode, float score_thresh, float nms_thresh, int img_h, int img_w, int max_out) {
    if (!c || max_out < 1 || (mode != 0 && mode != 1) || img_h < 1 || img_w < 1) return CF_EINVAL;
    if (c->last_B < 1) return c->fail(CF_ESTATE, "cf_decode_threshold_enqueue before cf_forward");
    HIPCHK(c, hipSetDevice(c->device));
    const int HW = (c->H / 4) * (c->W / 4);
    const int cap = c->t_cap > 0 ? c->t_cap : (HW < 4096 ? (HW + 63) / 64 * 64 : 4096);
    int r = thresh_launch(c, mode, score_thresh, nms_thresh, img_h, img_w, max_out, cap); if (r) return r;
    c->thr_pending = true; c->thr_mode = mode; c->thr_h = img_h; c->thr_w = img_w; c->thr_maxout = max_out; c->thr_B = c->last_B;
    c->thr_score = score_thresh; c->thr_nms = nms_thresh; c->thr_rs_h = c->rs_h; c->thr_rs_w = c->rs_w;
    return CF_OK;
}

int cf_decode_threshold_sized(cf_ctx* c, int mode, float score_thresh, float nms_thresh, int img_h, int img_w, int max_out,
                              float* dets, float* lms, int32_t* counts) {
    if (!c || !dets || !counts || max_out < 1 || (mode != 0 && mode != 1) || img_h < 1 || img_w < 1) return CF_EINVAL;
    if (c->last_B < 1) return c->fail(CF_ESTATE, "cf_decode_threshold before cf_forward");
    HIPCHK(c, hipSetDevice(c->device));
    const int B = c->last_B, HW = (c->H / 4) * (c->W / 4);
    // candidate capacity starts at 4096 per image and grows to the largest count seen (the reference's decode takes
    // any number of cells above the threshold, centerface.py:78-79): on overflow the collect kernel reports the
    // count, the workspace is reallocated and the decode reruns
    int cap = c->t_cap > 0 ? c->t_cap : (HW < 4096 ? (HW + 63) / 64 * 64 : 4096);
    bool launched = c->thr_pending && c->thr_mode == mode && c->thr_h == img_h && c->thr_w == img_w && c->thr_maxout == max_out && c->thr_B == B &&
                    c->thr_score == score_thresh && c->thr_nms == nms_thresh && c->thr_rs_h == c->rs_h && c->thr_rs_w == c->rs_w;      // cf_decode_threshold_enqueue did the launch
    c->thr_pending = false;
    // the pre-enqueued launch is only valid while the workspace it wrote still exists with the geometry it was launched on
    if (launched && (!c->t_overflow || !c->t_counts || c->t_B < B || c->t_cap < 1 || c->t_maxout < max_out)) launched = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!launched) { int r = thresh_launch(c, mode, score_thresh, nms_thresh, img_h, img_w, max_out, cap); if (r) return r; }
        launched = false;
        int overflow = 0;
        if (c->t_host) {                           // results are in page-locked host memory once the decode's event has fired
            HIPCHK(c, hipEventSynchronize(c->ev_thr));
            overflow = *(volatile int*)c->h_thr;
            memcpy(counts, c->h_thr + 256, (size_t)B * sizeof(int));
        } else {
            // `overflow` lives on this stack frame: never return while a copy into it may still be in flight
            const hipError_t e1 = hipMemcpyAsync(&overflow, c->t_overflow, sizeof(int), hipMemcpyDeviceToHost, c->stream);
            const hipError_t e2 = e1 == hipSuccess ? hipMemcpyAsync(counts, c->t_counts, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, c->stream) : e1;
            const hipError_t e3 = hipStreamSynchronize(c->stream);
            HIPCHK(c, e1); HIPCHK(c, e2); HIPCHK(c, e3);
        }
        if (overflow > c->t_cap && attempt == 0) { cap = (std::min(overflow, HW) + 63) / 64 * 64; continue; }
        if (overflow > c->t_cap) return c->fail(CF_EOVERFLOW, "more than %d cells above the score threshold in one image", c->t_cap);
        break;
    }
    // only the rows that exist: [B][rows][5 | 10] out of [B][max_out][.] (the rows past an image's count are never read by the caller)
    int rows = 0;
    for (int b = 0; b < B; ++b) rows = std::max(rows, std::min((int)counts[b], max_out));
    if (rows > 0 && c->t_host) {
        for (int b = 0; b < B; ++b) {
            const size_t n = (size_t)std::min((int)counts[b], max_out);
            if (!n) continue;
            memcpy(dets + (size_t)b * max_out * 5, c->t_dets + (size_t)b * max_out * 5, n * 5 * sizeof(float));
            if (lms) memcpy(lms + (size_t)b * max_out * 10, c->t_lms + (size_t)b * max_out * 10, n * 10 * sizeof(float));
        }
    } else if (rows > 0) {
        HIPCHK(c, hipMemcpy2DAsync(dets, (size_t)max_out * 5 * sizeof(float), c->t_dets, (size_t)max_out * 5 * sizeof(float),
                                   (size_t)rows * 5 * sizeof(float), B, hipMemcpyDeviceToHost, c->stream));
        if (lms) HIPCHK(c, hipMemcpy2DAsync(lms, (size_t)max_out * 10 * sizeof(float), c->t_lms, (size_t)max_out * 10 * sizeof(float),
                                            (size_t)rows * 10 * sizeof(float), B, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (thresh_mask_bytes(c->t_B, c->t_cap) > kThreshKeepBytes) free_thresh_ws(c);      // an oversized decode does not keep its workspace
    return CF_OK;
}

int cf_event_record(cf_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= 64) return CF_EINVAL;
    HIPCHK(c, hipEventRecord(c->events[slot], c->stream));
    return CF_OK;
}
int cf_event_elapsed_ms(cf_ctx* c, int a, int b, float* ms) {
    if (!c || !ms || a < 0 || a >= 64 || b < 0 || b >= 64) return CF_EINVAL;
    HIPCHK(c, hipEventSynchronize(c->events[b]));
    HIPCHK(c, hipEventElapsedTime(ms, c->events[a], c->events[b]));
    return CF_OK;
}

int cf_profile_forward(cf_ctx* c, const void* in, int in_format, int in_on_device, int B, int K,
                       cf_op_time* out, int cap, int* n_out) {
    if (!c || !out || !n_out) return CF_EINVAL;
    const void* net_in = nullptr;
    int r = stage_input(c, in, in_format, in_on_device, B, &net_in);
    if (r) return r;
    c->thr_pending = false;
    int nlaunch = 0;
    for (auto& op : c->ops) nlaunch += op.fused_away ? 0 : 1;
    const int nops = nlaunch + (K > 0 ? 1 : 0);
    if (cap < nops) return c->fail(CF_EINVAL, "cf_profile_forward: need room for %d records", nops);
    if (K > 0) { r = ensure_topk_ws(c, K); if (r) return r; }
    std::vector<hipEvent_t> ev(nops + 1);
    std::vector<std::string> tags;
    for (auto& e : ev) HIPCHK(c, hipEventCreate(&e));
    HIPCHK(c, hipEventRecord(ev[0], c->stream));
    int i = 0;
    for (auto& op : c->ops) {
        if (op.fused_away) continue;
        HIPCHK(c, launch_op(c, op, net_in, in_format, B));
        tags.push_back(last_kernel_tag());
        HIPCHK(c, hipEventRecord(ev[++i], c->stream));
    }
    c->last_B = B;
    if (K > 0) {
        r = enqueue_topk(c, B, K, 1, c->d_dets, c->d_lms, c->d_inds); if (r) return r;
        tags.push_back("cf::peak_collect_kernel(cf::TopkParams) + cf::topk_select_kernel");
        HIPCHK(c, hipEventRecord(ev[++i], c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    i = 0;
    for (auto& op : c->ops) {
        if (op.fused_away) continue;
        cf_op_time& t = out[i];
        memset(&t, 0, sizeof t);
        snprintf(t.name, sizeof t.name, "%s", op.name.c_str());
        snprintf(t.kind, sizeof t.kind, "%s", kKindName[op.kind]);
        snprintf(t.kernel, sizeof t.kernel, "%s", tags[i].c_str());
        HIPCHK(c, hipEventElapsedTime(&t.ms, ev[i], ev[i + 1]));
        t.algo_bytes = op_bytes(c, op, in_format, B);
        t.flops = 2.0 * op.macs * B;
        ++i;
    }
    if (K > 0) {
        cf_op_time& t = out[i];
        memset(&t, 0, sizeof t);
        snprintf(t.name, sizeof t.name, "peak_topk");
        snprintf(t.kind, sizeof t.kind, "decode");
        snprintf(t.kernel, sizeof t.kernel, "%s", tags[i].c_str());
        HIPCHK(c, hipEventElapsedTime(&t.ms, ev[i], ev[i + 1]));
        const double HW = (double)(c->H / 4) * (c->W / 4);
        t.algo_bytes = B * (HW * 4 + (double)K * (14 * 4 + 16 * 4 + 8));
        ++i;
    }
    for (auto& e : ev) hipEventDestroy(e);
    *n_out = i;
    return CF_OK;
}

int cf_plan_size(cf_ctx* c, int* n) {
    if (!c || !n) return CF_EINVAL;
    *n = (int)c->ops.size();
    return CF_OK;
}

int cf_plan_op(cf_ctx* c, int i, cf_op_info* out) {
    if (!c || !out || i < 0 || i >= (int)c->ops.size()) return CF_EINVAL;
    const Op& op = c->ops[i];
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", op.name.c_str());
    snprintf(out->kind, sizeof out->kind, "%s", kKindName[op.kind]);
    out->C = op.kind == OP_HEAD ? 16 : op.Cout; out->H = op.Hout; out->W = op.Wout;
    out->fused_away = op.fused_away ? 1 : 0;
    return CF_OK;
}

int cf_forward_trace(cf_ctx* c, const void* in, int in_format, int in_on_device, int B, int op_index, float* out_nchw) {
    if (!c || !out_nchw) return CF_EINVAL;
    if (op_index < 0 || op_index >= (int)c->ops.size()) return c->fail(CF_EINVAL, "cf_forward_trace: op_index %d outside the plan", op_index);
    if (c->ops[op_index].fused_away) return c->fail(CF_EINVAL, "cf_forward_trace: plan entry %d is fused into the next one", op_index);
    const void* net_in = nullptr;
    int r = stage_input(c, in, in_format, in_on_device, B, &net_in);
    if (r) return r;
    if (c->dec_pending) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_dec, 0)); c->dec_pending = false; }
    for (int i = 0; i <= op_index; ++i) HIPCHK(c, launch_op(c, c->ops[i], net_in, in_format, B));
    if (c->in_slot_used >= 0) {
        HIPCHK(c, hipEventRecord(c->ev_slot_free[c->in_slot_used], c->stream));
        c->slot_busy[c->in_slot_used] = true; c->in_slot_used = -1;
    }
    const Op& op = c->ops[op_index];
    const bool head = op.kind == OP_HEAD;
    const int C = head ? 16 : op.Cout;
    const size_t n = (size_t)B * C * op.Hout * op.Wout;
    float* tmp = nullptr;
    HIPCHK(c, hipMalloc((void**)&tmp, n * sizeof(float)));
    hipError_t e = op.out_blk
        ? launch_blocked_to_nchw(c->stream, c->dtype, c->bufs[op.out].p, tmp, B, C, op.Hout, op.Wout)
        : launch_nhwc_to_nchw(c->stream, head ? CF_F32 : c->dtype, c->bufs[op.out].p, tmp, B, C, op.Hout, op.Wout);
    if (e == hipSuccess) e = hipMemcpyAsync(out_nchw, tmp, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return c->fail(CF_EHIP, "cf_forward_trace: %s", hipGetErrorString(e));
    c->last_B = op_index + 1 == (int)c->ops.size() ? B : 0;      // heads are only valid after the whole plan ran
    c->thr_pending = false;
    return CF_OK;
}

int cf_ctdet_loss(cf_ctx* c, const float* gt_hm, const uint8_t* reg_mask, const int64_t* ind, const float* wh_t,
                  const float* reg_t, const uint8_t* lm_mask, const int64_t* lm_ind, const float* lm_t,
                  int max_objs, const float* weights4, float* out5) {
    if (!c) return CF_EINVAL;
    if (!gt_hm || !reg_mask || !ind || !wh_t || !reg_t || !lm_mask || !lm_ind || !lm_t || !weights4 || !out5 || max_objs < 1)
        return c->fail(CF_EINVAL, "cf_ctdet_loss: null argument or max_objs < 1");
    if (c->last_B < 1) return c->fail(CF_ESTATE, "cf_ctdet_loss before cf_forward");
    HIPCHK(c, hipSetDevice(c->device));
    const int B = c->last_B, h = c->H / 4, w = c->W / 4, nblocks = 256;
    const size_t hw = (size_t)h * w, bm = (size_t)B * max_objs;
    // one staging allocation per call: validation-loss evaluation is not on the per-image hot path
    const size_t off_gt = 0, off_wh = off_gt + B * hw * 4, off_reg = off_wh + bm * 8, off_lm = off_reg + bm * 8,
                 off_ind = off_lm + bm * 40, off_lmind = off_ind + bm * 8, off_ws = off_lmind + bm * 8,
                 off_out = off_ws + (3 * nblocks + 6) * 8, off_rm = off_out + 64, off_lmm = off_rm + ((bm + 15) / 16) * 16,
                 total = off_lmm + ((bm + 15) / 16) * 16;
    char* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, total));
    auto up = [&](size_t off, const void* src, size_t bytes) { return hipMemcpyAsync(d + off, src, bytes, hipMemcpyHostToDevice, c->stream); };
    hipError_t e = up(off_gt, gt_hm, B * hw * 4);
    if (e == hipSuccess) e = up(off_wh, wh_t, bm * 8);
    if (e == hipSuccess) e = up(off_reg, reg_t, bm * 8);
    if (e == hipSuccess) e = up(off_lm, lm_t, bm * 40);
    if (e == hipSuccess) e = up(off_ind, ind, bm * 8);
    if (e == hipSuccess) e = up(off_lmind, lm_ind, bm * 8);
    if (e == hipSuccess) e = up(off_rm, reg_mask, bm);
    if (e == hipSuccess) e = up(off_lmm, lm_mask, bm);
    LossParams p{};
    p.heads = (const float*)c->bufs[c->buf_heads].p;
    p.gt_hm = (const float*)(d + off_gt); p.wh_t = (const float*)(d + off_wh); p.reg_t = (const float*)(d + off_reg);
    p.lm_t = (const float*)(d + off_lm); p.ind = (const long long*)(d + off_ind); p.lm_ind = (const long long*)(d + off_lmind);
    p.reg_mask = (const unsigned char*)(d + off_rm); p.lm_mask = (const unsigned char*)(d + off_lmm);
    p.B = B; p.h = h; p.w = w; p.M = max_objs;
    p.hm_w = weights4[0]; p.wh_w = weights4[1]; p.off_w = weights4[2]; p.lm_w = weights4[3];
    if (e == hipSuccess) e = launch_ctdet_loss(c->stream, p, (double*)(d + off_ws), nblocks, (float*)(d + off_out));
    if (e == hipSuccess) e = hipMemcpyAsync(out5, d + off_out, 5 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return c->fail(CF_EHIP, "cf_ctdet_loss: %s", hipGetErrorString(e));
    return CF_OK;
}

int cf_get_streams(cf_ctx* c, void** main_stream, void** decode_stream) {
    if (!c) return CF_EINVAL;
    if (main_stream) *main_stream = (void*)c->stream;
    if (decode_stream) { int r = ensure_decode_stream(c); if (r) return r; *decode_stream = (void*)c->stream2; }
    return CF_OK;
}

static hipStream_t pick_stream(cf_ctx* c, int which) {
    if (which == 1 && ensure_decode_stream(c) != CF_OK) return nullptr;
    return which == 0 ? c->stream : which == 1 ? c->stream2 : which == 2 ? c->stream_in : nullptr;
}

// Do two streams sit on one hardware queue?  ~0.3 ms spin on the first, then an empty kernel on the second: on one queue the second waits
// for the first.  One timing sample can be fooled by anything else using the GPU: three probes, majority decides.  Both streams idle.
static int streams_share(cf_ctx* a, hipStream_t sa, hipStream_t sb, int* shared, bool fat = false) {
    if (sa == sb) { *shared = 1; return CF_OK; }
    if (fat) {
        static thread_local bool configured_dev[64] = {};
        if (!configured_dev[a->device & 63]) {
            HIPCHK(a, hipFuncSetAttribute(reinterpret_cast<const void*>(cf_fat_spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            configured_dev[a->device & 63] = true;
        }
    }
    HIPCHK(a, hipStreamSynchronize(sa));
    HIPCHK(a, hipStreamSynchronize(sb));
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    hipError_t err = hipSuccess;
    for (int i = 0; i < 3 && err == hipSuccess; ++i) err = hipEventCreate(&ev[i]);
    int votes = 0;
    for (int rep = 0; rep < 3 && err == hipSuccess; ++rep) {
        auto step = [&](hipError_t e) { if (err == hipSuccess) err = e; };
        step(hipEventRecord(ev[0], sa));
        if (fat) hipLaunchKernelGGL(cf_fat_spin_kernel, dim3(2560), dim3(64), 64 * 1024, sa, (long long)6000);       // 5 rounds of 60 us
        else hipLaunchKernelGGL(cf_spin_kernel, dim3(1), dim3(64), 0, sa, (long long)30000);
        step(hipGetLastError());
        step(hipEventRecord(ev[1], sa));
        hipLaunchKernelGGL(cf_spin_kernel, dim3(1), dim3(64), 0, sb, (long long)0);
        step(hipGetLastError());
        step(hipEventRecord(ev[2], sb));
        step(hipStreamSynchronize(sa));
        step(hipStreamSynchronize(sb));
        float ta = 0.0f, tb = 0.0f;
        step(hipEventElapsedTime(&ta, ev[0], ev[1]));
        step(hipEventElapsedTime(&tb, ev[0], ev[2]));
        if (err == hipSuccess && tb > 0.5f * ta) ++votes;
        if (rep == 1 && (votes == 0 || votes == 2)) break;        // decided after two agreeing probes
    }
    for (hipEvent_t e : ev) if (e) hipEventDestroy(e);          // on every path
    if (err != hipSuccess) return a->fail(CF_EHIP, "hardware-queue probe: %s", hipGetErrorString(err));
    *shared = votes >= 2 ? 1 : 0;
    return CF_OK;
}

// which_a / which_b: 0 = the context's main stream, 1 = its decode stream, 2 = the device's copy stream
int cf_streams_share_queue_ex(cf_ctx* a, int which_a, cf_ctx* b, int which_b, int* shared) {
    if (!a || !b || !shared) return CF_EINVAL;
    const bool fat = ((which_a | which_b) & 16) != 0;                      // + 16: the dispatch-pipe probe (a grid that cannot be resident at once)
    hipStream_t sa = pick_stream(a, which_a & 15), sb = pick_stream(b, which_b & 15);
    if (!sa || !sb) return a->fail(CF_EINVAL, "cf_streams_share_queue_ex: stream selector outside 0..2");
    if (a->device != b->device) { *shared = 0; return CF_OK; }
    HIPCHK(a, hipSetDevice(a->device));
    return streams_share(a, sa, sb, shared, fat);
}

// Put the main streams -- and the decode streams, for contexts that use one -- of n contexts of ONE device on pairwise different hardware
// queues.  The runtime binds a new stream to the queue with the fewest streams on it (of at most four per priority), so what a context
// gets depends on everything the process created before, and a create-then-destroy re-roll can land on the same crowded queue forever
// (bench.py's second and third ring: both main streams on one queue through 12 re-rolls, the ring worth nothing).  Here candidates are
// created one after the other and probed against the streams already chosen: a candidate on a new queue is adopted, one on a used queue
// is kept alive as BALLAST until the end -- it weighs its queue down, so the next candidate goes elsewhere.  Main streams are placed
// first.  window > 0: a main stream only has to differ from those of the `window` contexts before it in ctxs (a pool of more contexts than
// there are pipes, used in that order round-robin: neighbours in time must not share).  *n_distinct = streams placed.  All contexts idle.
int cf_spread_streams(cf_ctx** ctxs, int n, int window, int* n_distinct) {
    if (!ctxs || n < 1 || n > 16 || window < 0) return CF_EINVAL;
    for (int i = 0; i < n; ++i) if (!ctxs[i] || ctxs[i]->device != ctxs[0]->device) return CF_EINVAL;
    cf_ctx* c0 = ctxs[0];
    HIPCHK(c0, hipSetDevice(c0->device));
    struct Want { cf_ctx* c; bool decode; };
    std::vector<Want> want;
    for (int i = 0; i < n; ++i) want.push_back({ctxs[i], false});
    for (int i = 0; i < n; ++i) if (!(ctxs[i]->flags & CF_FLAG_NO_DECODE_STREAM)) want.push_back({ctxs[i], true});
    for (int i = 0; i < n; ++i) { int r = cf_synchronize(ctxs[i]); if (r) return r; }
    std::vector<hipStream_t> chosen, ballast;
    int rc = CF_OK;
    // Strict rounds: a candidate must clash with nothing placed so far.  The process's hardware queues may sit on fewer than four dispatch
    // pipes, though: when a decode stream finds no pipe of its own it settles for one it shares with another DECODE stream only (two decodes
    // of 0.15 ms per step on one pipe cost nothing measurable; a decode on the other context's main pipe costs the ring 8 %).
    for (int relaxed = 0; relaxed < 2 && rc == CF_OK && chosen.size() < want.size(); ++relaxed) {
        for (int tries = 0; tries < 24 && chosen.size() < want.size(); ++tries) {
            hipStream_t s = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
            hipLaunchKernelGGL(cf_spin_kernel, dim3(1), dim3(64), 0, s, (long long)0);       // first use: the stream has its queue now
            (void)hipStreamSynchronize(s);
            const bool placing_decode = want[chosen.size()].decode;
            bool clash = false;
            for (size_t k = 0; k < chosen.size() && !clash; ++k) {
                if (relaxed && placing_decode && want[k].decode) continue;
                if (window > 0 && !placing_decode && (int)(chosen.size() - k) > window) continue;       // only the `window` contexts before this one
                int sh = 0;
                rc = streams_share(c0, chosen[k], s, &sh, true);             // the dispatch-pipe probe: a shared queue fails it too
                if (rc) break;
                clash = sh != 0;
            }
            if (rc) { hipStreamDestroy(s); break; }
            if (clash && relaxed && !placing_decode) { ballast.push_back(s); continue; }
            (clash ? ballast : chosen).push_back(s);
        }
    }
    for (hipStream_t s : ballast) hipStreamDestroy(s);
    if (rc) { for (hipStream_t s : chosen) hipStreamDestroy(s); return rc; }
    for (size_t k = 0; k < chosen.size(); ++k) {                  // adopt: the contexts are idle (synchronised above)
        cf_ctx* c = want[k].c;
        hipStream_t& slot = want[k].decode ? c->stream2 : c->stream;
        if (slot) hipStreamDestroy(slot);
        slot = chosen[k];
        c->dec_pending = false; c->main_dec_pending = false;
        for (int i = 0; i < 2; ++i) c->slot_busy[i] = false;
    }
    if (n_distinct) *n_distinct = (int)chosen.size();
    return CF_OK;
}

int cf_streams_share_queue(cf_ctx* a, cf_ctx* b, int* shared) {
    if (!a || !b || !shared || a == b) return CF_EINVAL;
    return cf_streams_share_queue_ex(a, 0, b, 0, shared);
}

int cf_reroll_streams(cf_ctx* c) {
    if (!c) return CF_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
    HIPCHK(c, hipStreamSynchronize(c->stream_in));
    hipStream_t s1 = nullptr, s2 = nullptr;
    HIPCHK(c, make_ctx_stream(c, &s1));     // new ones first: the old ones still hold their queues
    if (c->stream2) HIPCHK(c, make_ctx_stream(c, &s2));
    hipStreamDestroy(c->stream);
    if (c->stream2) hipStreamDestroy(c->stream2);
    c->stream = s1; c->stream2 = s2;
    c->dec_pending = false; c->main_dec_pending = false;               // everything was drained above
    for (int i = 0; i < 2; ++i) c->slot_busy[i] = false;
    return CF_OK;
}

int cf_graph_stats(cf_ctx* c, int* n_graphs, int* n_uncapturable) {
    if (!c) return CF_EINVAL;
    int ng = 0, nb = 0;
    for (auto& g : c->graphs) { ng += g.exec != nullptr; nb += g.broken; }
    if (n_graphs) *n_graphs = ng;
    if (n_uncapturable) *n_uncapturable = nb;
    return CF_OK;
}

int cf_host_alloc(cf_ctx* c, uint64_t bytes, void** hptr) {
    if (!c || !hptr) return CF_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return CF_OK;
}
int cf_host_free(cf_ctx* c, void* hptr) {
    if (!c) return CF_EINVAL;
    HIPCHK(c, hipHostFree(hptr));
    return CF_OK;
}
int cf_device_alloc(cf_ctx* c, uint64_t bytes, void** dptr) {
    if (!c || !dptr) return CF_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(dptr, bytes));
    return CF_OK;
}
int cf_device_free(cf_ctx* c, void* dptr) {
    if (!c) return CF_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipFree(dptr));
    return CF_OK;
}
int cf_memcpy_h2d(cf_ctx* c, void* dst, const void* src, uint64_t bytes) {
    if (!c) return CF_EINVAL;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CF_OK;
}
int cf_memcpy_d2h(cf_ctx* c, void* dst, const void* src, uint64_t bytes) {
    if (!c) return CF_EINVAL;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CF_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- multi-GPU: gather of the final boxes over RCCL
// One process per GPU, images sharded across ranks, nothing exchanged on the data path except the fixed-size
// detection records [B, K, 16] after the decode (the reference has no counterpart: its
// torch.distributed imports at train.py:11,17 are unused).  RCCL is loaded with dlopen on first use, so a
// process that never creates a communicator carries no dependency on librccl.
struct cf_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;                       // THE gather stream of this rank: every all-gather of every context goes here, in call order
    hipEvent_t ev_done = nullptr;                       // recorded after the last enqueued all-gather (cf_comm_query / _synchronize)
    // ONE collective per gather step: every rank sends a fixed-size slot [16-float header {magic, B, K, step} | B x K x 16
    // records].  The slot geometry (B, K) is agreed ONCE (cf_comm_set_shard, or the first gather): a 2-int all-gather + a
    // device-side compare whose verdict the host reads before the first record gather of that geometry is enqueued -- so an
    // all-gather with per-rank-unequal counts is never launched.  After that, the header of every slot is validated on the
    // device behind the all-gather (a rank whose shard changed still sends a full-size slot carrying its real (B, K): every
    // rank latches the mismatch, nobody hangs).
    int slot_B = 0, slot_K = 0; bool slot_verified = false; size_t slot_floats = 0;
    unsigned step = 0;                                  // gathers enqueued so far: carried in the header, equal on all ranks by construction
    float* recv = nullptr; size_t recv_elems = 0;       // [world][slot] landing area of the all-gather
    float* d_spare = nullptr; size_t spare_elems = 0;   // a slot-size send buffer for the rank-local mismatch path
    int* d_chk = nullptr;                               // [world][2] gathered (B, K) + [2] this rank's pair
    int* h_flag = nullptr; int* d_flag = nullptr;       // pinned + mapped: {verdict, peer rank, peer B, peer K, my B, my K, peer step, my step} (sticky)
    hipEvent_t ev_chk = nullptr;                        // behind the compare kernel of the agreement
    int debug_skew = 0;                                 // cf_comm_debug(1, v): the next gather's header carries B + v (tests of the mismatch path)
    // cf_comm_create_loopback: no RCCL communicator -- this process plays every rank of a `world`-rank job in turn on ONE GPU
    // (cf_comm_loopback_rank), the all-gather is the device copy of each rank's slot into its place of the landing area, and the
    // header check + unpack run when the last rank has deposited: slot sizes, header protocol and the rank-major unpack arithmetic
    // of an N-rank gather are exercised without N GPUs
    bool loopback = false; unsigned deposited = 0;      // bit r: rank r's slot of the current step is in `recv`
    std::string err;
};

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

// librccl is loaded once, by whichever thread gets here first (std::call_once: a second thread blocks until the table is
// complete instead of seeing a half-filled one); any missing symbol is an error before the first call through the table
RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) { const char* de = dlerror(); api.err = std::string("dlopen(librccl): ") + (de ? de : "not found"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(api.handle, n); if (!p && api.err.empty()) api.err = std::string("librccl lacks ") + n; return p; };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    return &api;
}

// stream + event + check buffer of a new communicator
int comm_resources(cf_ctx* c, cf_comm* m) {
    HIPCHK(c, hipSetDevice(m->device));
    HIPCHK(c, hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&m->ev_chk, hipEventDisableTiming));
    HIPCHK(c, hipMalloc((void**)&m->d_chk, (size_t)(m->world + 1) * 2 * sizeof(int)));
    HIPCHK(c, hipHostMalloc((void**)&m->h_flag, 8 * sizeof(int), hipHostMallocMapped));
    memset(m->h_flag, 0, 8 * sizeof(int));
    HIPCHK(c, hipHostGetDevicePointer((void**)&m->d_flag, m->h_flag, 0));
    return CF_OK;
}

}  // namespace

extern "C" {

int cf_comm_unique_id(void* id, int bytes) {
    if (!id || bytes < (int)sizeof(ncclUniqueId)) { g_create_error = "cf_comm_unique_id: need a 128-byte buffer"; return CF_EINVAL; }
    RcclApi* r = rccl();
    if (!r->err.empty()) { g_create_error = r->err; return CF_EHIP; }
    ncclUniqueId u;
    ncclResult_t e = r->GetUniqueId(&u);
    if (e != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + r->GetErrorString(e); return CF_EHIP; }
    memcpy(id, &u, sizeof u);
    return CF_OK;
}

int cf_comm_create(cf_ctx* c, int rank, int world, const void* id, cf_comm** out) {
    if (!c || !id || !out || world < 1 || rank < 0 || rank >= world) return CF_EINVAL;
    *out = nullptr;
    RcclApi* r = rccl();
    if (!r->err.empty()) return c->fail(CF_EHIP, "%s", r->err.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    cf_comm* m = new cf_comm();
    m->rank = rank; m->world = world; m->device = c->device;
    ncclResult_t e = r->CommInitRank(&m->comm, world, u, rank);
    if (e != ncclSuccess) { delete m; return c->fail(CF_EHIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, r->GetErrorString(e)); }
    int rr = comm_resources(c, m);
    if (rr) { cf_comm_destroy(m); return rr; }
    *out = m;
    return CF_OK;
}

// One process, one thread, n GPUs: rank i = ctxs[i]'s device.  ncclCommInitRank blocks until every rank has joined, so a
// host that owns all the contexts must issue the n calls inside one ncclGroupStart / ncclGroupEnd.
int cf_comm_create_all(cf_ctx** ctxs, int n, cf_comm** out) {
    if (!ctxs || !out || n < 1) return CF_EINVAL;
    for (int i = 0; i < n; ++i) { if (!ctxs[i]) return CF_EINVAL; out[i] = nullptr; }
    cf_ctx* c0 = ctxs[0];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (ctxs[i]->device == ctxs[j]->device) return c0->fail(CF_EINVAL, "cf_comm_create_all: contexts %d and %d share device %d (one rank per GPU)", j, i, ctxs[i]->device);
    RcclApi* r = rccl();
    if (!r->err.empty()) return c0->fail(CF_EHIP, "%s", r->err.c_str());
    ncclUniqueId u;
    ncclResult_t e = r->GetUniqueId(&u);
    if (e != ncclSuccess) return c0->fail(CF_EHIP, "ncclGetUniqueId: %s", r->GetErrorString(e));
    std::vector<cf_comm*> ms(n, nullptr);
    for (int i = 0; i < n; ++i) { ms[i] = new cf_comm(); ms[i]->rank = i; ms[i]->world = n; ms[i]->device = ctxs[i]->device; }
    e = r->GroupStart();
    for (int i = 0; i < n && e == ncclSuccess; ++i) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) { e = ncclUnhandledCudaError; break; }
        e = r->CommInitRank(&ms[i]->comm, n, u, i);
    }
    ncclResult_t e2 = r->GroupEnd();
    if (e == ncclSuccess) e = e2;
    int rr = CF_OK;
    if (e != ncclSuccess) rr = c0->fail(CF_EHIP, "ncclCommInitRank (grouped, %d ranks): %s", n, r->GetErrorString(e));
    for (int i = 0; i < n && !rr; ++i) rr = comm_resources(ctxs[i], ms[i]);
    if (rr) { for (cf_comm* m : ms) cf_comm_destroy(m); return rr; }
    for (int i = 0; i < n; ++i) out[i] = ms[i];
    return CF_OK;
}

// Dry run of a `world`-rank gather on one GPU: no RCCL, this process plays the ranks one after the other.
int cf_comm_create_loopback(cf_ctx* c, int world, cf_comm** out) {
    if (!c || !out || world < 1 || world > 32) return CF_EINVAL;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    cf_comm* m = new cf_comm();
    m->rank = 0; m->world = world; m->device = c->device; m->loopback = true;
    int rr = comm_resources(c, m);
    if (rr) { cf_comm_destroy(m); return rr; }
    *out = m;
    return CF_OK;
}
// the rank whose cf_gather_topk comes next (any order; every rank exactly once per step)
int cf_comm_loopback_rank(cf_comm* m, int rank) {
    if (!m || !m->loopback || rank < 0 || rank >= m->world) return CF_EINVAL;
    m->rank = rank;
    return CF_OK;
}

int cf_comm_destroy(cf_comm* m) {
    if (!m) return CF_OK;
    hipSetDevice(m->device);
    if (m->stream) hipStreamSynchronize(m->stream);
    if (m->recv) hipFree(m->recv);
    if (m->d_spare) hipFree(m->d_spare);
    if (m->d_chk) hipFree(m->d_chk);
    if (m->h_flag) hipHostFree(m->h_flag);
    if (m->comm) rccl()->CommDestroy(m->comm);
    if (m->ev_done) hipEventDestroy(m->ev_done);
    if (m->ev_chk) hipEventDestroy(m->ev_chk);
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
    return CF_OK;
}

// Give up on a communicator whose collective does not complete (a peer died, a first-run deadlock): ncclCommAbort makes the
// in-flight RCCL kernels exit, the gather stream drains, every resource is released.  The handle is dead afterwards.
int cf_comm_abort(cf_comm* m) {
    if (!m) return CF_OK;
    hipSetDevice(m->device);
    if (m->comm) { rccl()->CommAbort(m->comm); m->comm = nullptr; }
    return cf_comm_destroy(m);
}

// the latched verdict of the device-side checks (agreement compare, slot headers): CF_EINVAL + text once any of them failed
static int comm_mismatch(cf_comm* m) {
    volatile int* f = m->h_flag;
    if (!f || !f[0]) return CF_OK;
    char buf[320];
    if (f[0] == 2)
        snprintf(buf, sizeof buf, "cf_gather_topk: rank %d is at gather step %d, rank %d sent step %d -- every rank must call cf_gather_topk the same number of times in the same order",
                 m->rank, f[7], f[1], f[6]);
    else if (f[0] == 3)
        snprintf(buf, sizeof buf, "cf_gather_topk: the slot received from rank %d carries no gather header (another payload size or library version on that rank?)", f[1]);
    else
        snprintf(buf, sizeof buf, "cf_gather_topk: rank %d has (B=%d, K=%d), rank %d has (B=%d, K=%d) -- the gather needs equal shards (pad the last one)",
                 m->rank, f[4], f[5], f[1], f[2], f[3]);
    m->err = buf;
    return CF_EINVAL;
}

// 0: everything enqueued on the gather stream (agreement, gathers) has completed, 1: still running, < 0: error (CF_EINVAL: the
// agreement or a slot header found unequal shards / steps on the ranks).  Never blocks.
int cf_comm_query(cf_comm* m) {
    if (!m) return CF_EINVAL;
    hipSetDevice(m->device);
    int mm = comm_mismatch(m); if (mm) return mm;
    hipError_t e = hipStreamQuery(m->stream);
    if (e == hipSuccess) return comm_mismatch(m);
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 1; }
    m->err = std::string("hipStreamQuery: ") + hipGetErrorString(e);
    return CF_EHIP;
}

// Waits for the gather stream by POLLING (a latched mismatch ends the wait: CF_EINVAL, and cf_comm_abort is the way out).
int cf_comm_synchronize(cf_comm* m) {
    if (!m) return CF_EINVAL;
    for (unsigned spins = 0;; ++spins) {
        int q = cf_comm_query(m);
        if (q <= 0) return q;
        if (spins > 200) usleep(20);
    }
}

const char* cf_comm_last_error(cf_comm* m) { return m ? m->err.c_str() : "null communicator"; }

// Test hooks.  what = 0: park the gather stream behind a spin kernel of `value` milliseconds (a collective that does not
// complete in time, for the deadline tests); what = 1: the header of the next gather's slot carries B + value (the mismatch path at world 1).
int cf_comm_debug(cf_comm* m, int what, int value) {
    if (!m) return CF_EINVAL;
    hipSetDevice(m->device);
    if (what == 0) {
        if (value < 0 || value > 60000) return CF_EINVAL;
        hipLaunchKernelGGL(cf_spin_kernel, dim3(1), dim3(64), 0, m->stream, (long long)value * 100000LL);
        return hipGetLastError() == hipSuccess ? CF_OK : CF_EHIP;
    }
    if (what == 1) { m->debug_skew = value; return CF_OK; }
    return CF_EINVAL;
}

void* cf_comm_stream(cf_comm* m) { return m ? (void*)m->stream : nullptr; }

// Declare the shard every rank will gather: B images x K records.  Collective by contract (every rank calls it with the same
// values at the same point of its call sequence; the first cf_gather_topk of a communicator calls it implicitly).  Enqueues,
// on the gather stream, the only extra collective of the gather path -- a 2-int all-gather of each rank's (B, K) and a
// device-side compare that latches a mismatch -- and returns without waiting: a host with a deadline polls cf_comm_query
// (0 = agreed, CF_EINVAL = unequal shards) and aborts instead of blocking.  The first gather of the geometry reads the verdict
// (waiting for it if the host did not) BEFORE it enqueues a record gather: unequal counts never reach ncclAllGather.
int cf_comm_set_shard(cf_comm* m, int B, int K) {
    if (!m || B < 1 || K < 1) return CF_EINVAL;
    if (!m->comm && !m->loopback) { m->err = "communicator was aborted"; return CF_ESTATE; }
    if (hipSetDevice(m->device) != hipSuccess) { m->err = "hipSetDevice failed"; return CF_EHIP; }
    { int mm = comm_mismatch(m); if (mm) return mm; }
    const size_t slot = (size_t)B * K * 16 + kSlotHeader;
    auto hipfail = [&](hipError_t e, const char* what) { m->err = std::string(what) + ": " + hipGetErrorString(e); return CF_EHIP; };
    if (m->recv_elems < slot * m->world || m->spare_elems < slot) {
        hipError_t e = hipSuccess;
        if (m->recv || m->d_spare) {                                     // gathers of the previous geometry still use the old buffers
            e = hipStreamSynchronize(m->stream);                         // (never on the first agreement: nothing to wait for, no host wait)
            if (e != hipSuccess) return hipfail(e, "hipStreamSynchronize");
        }
        if (m->recv) (void)hipFree(m->recv);
        if (m->d_spare) (void)hipFree(m->d_spare);
        m->recv = nullptr; m->recv_elems = 0; m->d_spare = nullptr; m->spare_elems = 0;
        if ((e = hipMalloc((void**)&m->recv, slot * m->world * sizeof(float))) != hipSuccess) return hipfail(e, "hipMalloc (gather landing area)");
        m->recv_elems = slot * m->world;
        if ((e = hipMalloc((void**)&m->d_spare, slot * sizeof(float))) != hipSuccess) return hipfail(e, "hipMalloc (spare slot)");
        if ((e = hipMemsetAsync(m->d_spare, 0, slot * sizeof(float), m->stream)) != hipSuccess) return hipfail(e, "hipMemsetAsync");
        m->spare_elems = slot;
    }
    m->slot_B = B; m->slot_K = K; m->slot_floats = slot; m->slot_verified = false;
    int* mine = m->d_chk + 2 * m->world;
    hipLaunchKernelGGL(cf_comm_publish_kernel, dim3(1), dim3(1), 0, m->stream, mine, B, K);
    hipError_t le = hipGetLastError(); if (le != hipSuccess) return hipfail(le, "cf_comm_publish_kernel");
    if (m->loopback) {                                   // one caller speaks for every rank: its pair in all `world` places
        for (int r = 0; r < m->world; ++r)
            if ((le = hipMemcpyAsync(m->d_chk + 2 * r, mine, 2 * sizeof(int), hipMemcpyDeviceToDevice, m->stream)) != hipSuccess) return hipfail(le, "hipMemcpyAsync (loopback agreement)");
        m->deposited = 0;
    } else {
        ncclResult_t e = rccl()->AllGather(mine, m->d_chk, 2, ncclInt32, m->comm, m->stream);
        if (e != ncclSuccess) { m->err = std::string("ncclAllGather (shard agreement): ") + rccl()->GetErrorString(e); return CF_EHIP; }
    }
    hipLaunchKernelGGL(cf_comm_compare_kernel, dim3(1), dim3(64), 0, m->stream, (const int*)m->d_chk, m->world, B, K, m->d_flag);
    if ((le = hipGetLastError()) != hipSuccess) return hipfail(le, "cf_comm_compare_kernel");
    if ((le = hipEventRecord(m->ev_chk, m->stream)) != hipSuccess) return hipfail(le, "hipEventRecord");
    return CF_OK;
}

int cf_gather_topk(cf_ctx* c, cf_comm* m, int K, int use_reg, float* records, int out_on_device) {
    if (!c || !m || !records) return CF_EINVAL;
    if (c->last_B < 1) return c->fail(CF_ESTATE, "cf_gather_topk before cf_forward");
    if (!m->comm && !m->loopback) return c->fail(CF_ESTATE, "communicator was aborted");
    if (m->device != c->device) return c->fail(CF_EINVAL, "communicator was created for device %d, context runs on %d", m->device, c->device);
    const int B = c->last_B, HW = (c->H / 4) * (c->W / 4);
    if (K < 1 || K > HW) return c->fail(CF_EINVAL, "K=%d must be in [1, %d]", K, HW);
    HIPCHK(c, hipSetDevice(c->device));
    int r = CF_OK;
    if (!c->ev_gather) HIPCHK(c, hipEventCreateWithFlags(&c->ev_gather, hipEventDisableTiming));
    { int mm = comm_mismatch(m); if (mm) return c->fail(CF_EINVAL, "%s", m->err.c_str()); }
    // ncclAllGather takes ONE count for all ranks.  The count of every gather is the slot agreed by cf_comm_set_shard (implicitly
    // here on the first gather): its verdict is read -- waited for by polling if the host has not done so through
    // cf_comm_query -- before the first record gather of the geometry is enqueued.
    if (!m->slot_B) { r = cf_comm_set_shard(m, B, K); if (r) return c->fail(r, "%s", m->err.c_str()); }
    // the all-gather reads a whole agreed slot from this context's record buffer, whatever this step's K is: size it for both
    {
        const size_t slot_rows = ((size_t)m->slot_B * m->slot_K + c->max_batch - 1) / c->max_batch;
        r = ensure_topk_ws(c, (int)std::max<size_t>((size_t)K, slot_rows)); if (r) return r;
    }
    if (!m->slot_verified) {
        for (unsigned spins = 0;; ++spins) {
            hipError_t q = hipEventQuery(m->ev_chk);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return c->fail(CF_EHIP, "hipEventQuery (shard agreement): %s", hipGetErrorString(q));
            (void)hipGetLastError();
            if (spins > 200) usleep(20);
        }
        int mm = comm_mismatch(m); if (mm) return c->fail(CF_EINVAL, "%s", m->err.c_str());
        m->slot_verified = true;
    }
    // The agreed slot is a CAPACITY (round 6): a step whose B x K records fit into it -- the ragged last batch of a run, a
    // smaller K -- travels in the same fixed-size slot with its real (B, K) in the header, and the header check behind the all-gather
    // requires every rank's header to equal THIS rank's (B, K, step): a geometry that is identical on every rank needs no second
    // agreement and no second collective, one that differs between ranks latches the mismatch on every rank.  A shard that does NOT
    // fit is a rank-LOCAL fact: this rank still enqueues a full-size slot (from the spare buffer, the header carrying its real
    // (B, K)), so the collective sequence and every count stay identical on all ranks, every other rank's header check latches,
    // and the call returns CF_EINVAL here (a larger geometry needs cf_comm_set_shard on every rank).
    const size_t n = (size_t)B * K * 16;
    const bool local_ok = n <= (size_t)m->slot_B * m->slot_K * 16;
    // (loopback: the ranks of one step deposit one after the other and share its number; the step ends with the last of them)
    const bool last_rank = !m->loopback || (m->deposited | (1u << m->rank)) == (m->world >= 32 ? ~0u : (1u << m->world) - 1u);
    if (m->loopback && (m->deposited >> m->rank & 1u)) return c->fail(CF_ESTATE, "cf_gather_topk (loopback): rank %d has deposited its slot of this step already", m->rank);
    const unsigned step = last_rank ? m->step++ : m->step;
    float* src = c->d_slot;
    if (local_ok) {
        // decode stream of the context: [wait forward] [wait the previous gather: it reads d_slot] decode -> records -> event
        // gather stream of the communicator (one per rank, shared by all its contexts): [wait that event] header, all-gather,
        // header check + unpack (-> D2H).  With ONE communicator and ONE stream per rank every rank enqueues its collectives in
        // the same order as long as it calls cf_gather_topk in the same order -- no cross-communicator ordering to get wrong.
        hipStream_t ds = c->stream;                          // CF_FLAG_NO_DECODE_STREAM: the decode stays on the main stream
        if (!(c->flags & CF_FLAG_NO_DECODE_STREAM)) {
            r = ensure_decode_stream(c); if (r) return r;
            ds = c->stream2;
            HIPCHK(c, hipStreamWaitEvent(ds, c->ev_fwd, 0));
        }
        if (c->gather_pending) HIPCHK(c, hipStreamWaitEvent(ds, c->ev_gather, 0));
        r = enqueue_topk(c, B, K, use_reg, nullptr, nullptr, nullptr, nullptr, ds, c->d_rec);
        if (r) return r;
        HIPCHK(c, hipEventRecord(c->ev_dec, ds));
        c->dec_pending = true;
        HIPCHK(c, hipStreamWaitEvent(m->stream, c->ev_dec, 0));
    } else {
        src = m->d_spare;
    }
    hipLaunchKernelGGL(cf_comm_header_kernel, dim3(1), dim3(1), 0, m->stream, src, B + m->debug_skew, K, step);
    m->debug_skew = 0;
    HIPCHK(c, hipGetLastError());
    if (m->loopback) {
        HIPCHK(c, hipMemcpyAsync(m->recv + (size_t)m->rank * m->slot_floats, src, m->slot_floats * sizeof(float), hipMemcpyDeviceToDevice, m->stream));
        m->deposited |= 1u << m->rank;
        if (!last_rank) {                                // nothing to unpack yet; the next rank's decode must not overwrite d_slot under the copy
            if (local_ok) { HIPCHK(c, hipEventRecord(c->ev_gather, m->stream)); c->gather_pending = true; }
            if (!local_ok) return c->fail(CF_EINVAL, "cf_gather_topk (loopback): rank %d's shard (B=%d, K=%d) does not fit the agreed slot (B=%d, K=%d) -- the gather needs equal shards", m->rank, B, K, m->slot_B, m->slot_K);
            return CF_OK;
        }
        m->deposited = 0;
    } else {
        ncclResult_t e = rccl()->AllGather(src, m->recv, m->slot_floats, ncclFloat, m->comm, m->stream);
        if (e != ncclSuccess) return c->fail(CF_EHIP, "ncclAllGather: %s", rccl()->GetErrorString(e));
    }
    float* dev_dst = (local_ok && out_on_device) ? records : nullptr;
    const int ublocks = dev_dst ? (int)std::min<size_t>(1024, (n / 4 * m->world + 255) / 256) : 1;
    hipLaunchKernelGGL(cf_comm_unpack_kernel, dim3(ublocks), dim3(256), 0, m->stream, (const float*)m->recv, m->world, m->slot_floats, B, K, step, dev_dst, m->d_flag);
    HIPCHK(c, hipGetLastError());
    if (local_ok) {
        HIPCHK(c, hipEventRecord(c->ev_gather, m->stream));
        c->gather_pending = true;
    }
    if (!local_ok) {
        volatile int* f = m->h_flag;                     // the device check will latch the same verdict; make it visible to the host now
        if (!f[0]) { f[1] = m->rank; f[2] = B; f[3] = K; f[4] = m->slot_B; f[5] = m->slot_K; f[0] = 1; }
        return c->fail(CF_EINVAL, "cf_gather_topk: this rank's shard (B=%d, K=%d) does not fit the slot the communicator agreed on (B=%d, K=%d) -- the gather needs equal shards no larger than the agreed one (cf_comm_set_shard on every rank for a larger geometry)",
                       B, K, m->slot_B, m->slot_K);
    }
    if (!out_on_device) {
        // host destination: strip the headers with a strided copy (row = one rank's records), then the verdict of this step's check
        HIPCHK(c, hipMemcpy2DAsync(records, n * sizeof(float), m->recv + kSlotHeader, m->slot_floats * sizeof(float), n * sizeof(float), (size_t)m->world,
                                   hipMemcpyDeviceToHost, m->stream));
        HIPCHK(c, hipStreamSynchronize(m->stream));
        c->gather_pending = false;
        int mm = comm_mismatch(m); if (mm) return c->fail(CF_EINVAL, "%s", m->err.c_str());
    }
    return CF_OK;
}

}  // extern "C"
