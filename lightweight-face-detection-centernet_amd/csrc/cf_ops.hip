// Per-op C-ABI entry points (cf_op_*): each runs ONE production kernel on host NCHW float32
// arrays, exactly as the torch op it replaces would be called, so the parity tests can check every
// kernel in isolation against the oracle / the reference's golden vectors.  Layout conversion
// NCHW<->NHWC happens on the device around the kernel; there is no CPU compute path here.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "centerface_hip.h"
#include "cf_common.h"
#include "cf_kernels.h"

using namespace cf;

namespace {

thread_local std::string g_op_error;

struct Scope {
    hipStream_t s = nullptr;
    std::vector<void*> ptrs;
    hipError_t err = hipSuccess;
    explicit Scope(int device) {
        err = hipSetDevice(device);
        if (err == hipSuccess) err = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);   // never the legacy stream (see upload() in cf_runtime.hip)
    }
    ~Scope() {
        if (s) { hipStreamSynchronize(s); hipStreamDestroy(s); }
        for (void* p : ptrs) hipFree(p);
    }
    void* alloc(size_t bytes) {
        if (err != hipSuccess) return nullptr;
        void* p = nullptr;
        err = hipMalloc(&p, bytes + 256);
        if (err == hipSuccess) { ptrs.push_back(p); hipMemsetAsync(p, 0, bytes + 256, s); }
        return p;
    }
    void* up(const void* host, size_t bytes) {
        void* p = alloc(bytes);
        if (p && err == hipSuccess) err = hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, s);
        return p;
    }
    template <typename T> T* upv(const std::vector<T>& v) { return (T*)up(v.data(), v.size() * sizeof(T)); }
    void chk(hipError_t e) { if (err == hipSuccess) err = e; }
    // host f32 NCHW -> device T NHWC
    void* to_nhwc(int dtype, const float* host, int B, int C, int H, int W) {
        size_t n = (size_t)B * C * H * W;
        float* src = (float*)up(host, n * 4);
        void* dst = alloc(n * elem_size(dtype));
        if (err == hipSuccess) chk(launch_nchw_to_nhwc(s, dtype, src, dst, B, C, H, W));
        return dst;
    }
    // device T NHWC -> host f32 NCHW
    void to_host_nchw(int dtype, const void* dev, float* host, int B, int C, int H, int W) {
        size_t n = (size_t)B * C * H * W;
        float* tmp = (float*)alloc(n * 4);
        if (err == hipSuccess) chk(launch_nhwc_to_nchw(s, dtype, dev, tmp, B, C, H, W));
        if (err == hipSuccess) chk(hipMemcpyAsync(host, tmp, n * 4, hipMemcpyDeviceToHost, s));
        if (err == hipSuccess) chk(hipStreamSynchronize(s));
    }
    int result(const char* what) {
        if (err == hipSuccess) err = hipStreamSynchronize(s);
        if (err == hipSuccess) return CF_OK;
        g_op_error = std::string(what) + ": " + hipGetErrorString(err);
        return CF_EHIP;
    }
};

bool bad_dtype(int d) { return d != CF_F32 && d != CF_BF16 && d != CF_F32_SPLIT; }

void fold(const float* bn /*[4][C]: weight,bias,mean,var*/, int C, float eps, std::vector<double>& sc, std::vector<double>& sh) {
    sc.resize(C); sh.resize(C);
    for (int c = 0; c < C; ++c) {
        sc[c] = (double)bn[c] / std::sqrt((double)bn[3 * C + c] + (double)eps);
        sh[c] = (double)bn[C + c] - (double)bn[2 * C + c] * sc[c];
    }
}

// explicit maps (NCHW) -> the 16-float head record layout the decode kernels read
std::vector<float> make_records(const float* hm, const float* wh, const float* reg, const float* lm, int B, int h, int w) {
    const size_t HW = (size_t)h * w;
    std::vector<float> rec((size_t)B * HW * 16, 0.0f);
    for (int b = 0; b < B; ++b)
        for (size_t i = 0; i < HW; ++i) {
            float* r = &rec[((size_t)b * HW + i) * 16];
            r[0] = hm[(size_t)b * HW + i];
            if (wh) { r[1] = wh[((size_t)b * 2 + 0) * HW + i]; r[2] = wh[((size_t)b * 2 + 1) * HW + i]; }
            if (lm) for (int c = 0; c < 10; ++c) r[3 + c] = lm[((size_t)b * 10 + c) * HW + i];
            if (reg) { r[13] = reg[((size_t)b * 2 + 0) * HW + i]; r[14] = reg[((size_t)b * 2 + 1) * HW + i]; }
        }
    return rec;
}

}  // namespace

extern "C" {

const char* cf_op_last_error(void) { return g_op_error.c_str(); }

// Page-locked host memory without a context (hipHostMalloc, portable: every device may DMA from it): what cfa.pinned_empty hands out.
int cf_pinned_alloc(uint64_t bytes, void** hptr) {
    if (!hptr || bytes == 0) { g_op_error = "cf_pinned_alloc: null pointer or zero size"; return CF_EINVAL; }
    *hptr = nullptr;
    const hipError_t e = hipHostMalloc(hptr, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *hptr = nullptr;
        g_op_error = std::string("hipHostMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? CF_ENOMEM : CF_EHIP;
    }
    return CF_OK;
}
int cf_pinned_free(void* hptr) {
    if (!hptr) return CF_OK;
    const hipError_t e = hipHostFree(hptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        g_op_error = std::string("hipHostFree: ") + hipGetErrorString(e);
        return CF_EHIP;
    }
    return CF_OK;
}

// Page-lock memory the caller already owns (a mapped frame pool, a shared-memory segment) so that cf_forward / cf_forward_resized /
// cf_forward_images read it by asynchronous DMA, without the staging copy into cf_host_alloc memory.  Costs a page-table walk per call
// (~0.1 ms per MB): for buffers that are reused, not for one-shot images.  Whole pages of a mapping of its own only (the header's
// contract): round 5 registered numpy heap arrays, and the GPU faulted on them once the C library had trimmed and regrown its heap
// underneath (the SIGABRT of GPUTEST_r05).
int cf_host_register(void* hptr, uint64_t bytes) {
    if (!hptr || bytes == 0) { g_op_error = "cf_host_register: null pointer or zero size"; return CF_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(hptr) & 4095u) || (bytes & 4095u)) {
        g_op_error = "cf_host_register: the range must be whole pages (pointer and size multiples of 4096) of a mapping of its own -- not malloc / numpy heap memory; use cf_pinned_alloc";
        return CF_EINVAL;
    }
    const hipError_t e = hipHostRegister(hptr, bytes, hipHostRegisterPortable | hipHostRegisterMapped);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        g_op_error = std::string("hipHostRegister: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? CF_ENOMEM : CF_EHIP;
    }
    return CF_OK;
}
int cf_host_unregister(void* hptr) {
    if (!hptr) { g_op_error = "cf_host_unregister: null pointer"; return CF_EINVAL; }
    const hipError_t e = hipHostUnregister(hptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        g_op_error = std::string("hipHostUnregister: ") + hipGetErrorString(e);
        return CF_EHIP;
    }
    return CF_OK;
}

int cf_op_dwconv(int device, int dtype, const float* x, const float* w, const float* bias, float* y,
                 int B, int C, int H, int W, int k, int stride, int pad_lo, int pad_hi, int act) {
    if (bad_dtype(dtype) || !x || !w || !y || (C % 8) || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return CF_EINVAL;
    const int Ho = (H + pad_lo + pad_hi - k) / stride + 1, Wo = (W + pad_lo + pad_hi - k) / stride + 1;
    if (Ho < 1 || Wo < 1) return CF_EINVAL;
    Scope sc(device);
    std::vector<float> wp((size_t)k * k * C);
    dw_pack_weights(w, C, k, wp.data());
    DwParams p{};
    p.x = sc.to_nhwc(dtype, x, B, C, H, W);
    p.w = sc.upv(wp);
    p.bias = bias ? (const float*)sc.up(bias, (size_t)C * 4) : nullptr;
    p.y = sc.alloc((size_t)B * Ho * Wo * C * elem_size(dtype));
    p.B = B; p.C = C; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.k = k; p.s = stride; p.pad_lo = pad_lo; p.act = act;
    if (sc.err == hipSuccess) sc.chk(launch_dw(sc.s, dtype, p));
    sc.to_host_nchw(dtype, p.y, y, B, C, Ho, Wo);
    return sc.result("cf_op_dwconv");
}

int cf_op_pwconv(int device, int dtype, const float* x, const float* w, const float* bias,
                 const float* residual, float* y, int B, int Cin, int Cout, int H, int W, int act) {
    if (bad_dtype(dtype) || !x || !w || !y || (Cin % 8) || (Cout % 8) || act < 0 || act > 2) return CF_EINVAL;
    Scope sc(device);
    std::vector<char> packed(pw_packed_bytes(dtype, Cin, Cout));
    pw_pack_weights(dtype, w, Cin, Cout, packed.data());
    PwParams p{};
    p.x = sc.to_nhwc(dtype, x, B, Cin, H, W);
    p.wp = sc.up(packed.data(), packed.size());
    p.bias = bias ? (const float*)sc.up(bias, (size_t)Cout * 4) : nullptr;
    p.res = residual ? sc.to_nhwc(dtype, residual, B, Cout, H, W) : nullptr;
    p.y = sc.alloc((size_t)B * H * W * Cout * elem_size(dtype));
    p.M = (long long)B * H * W; p.K = Cin; p.N = Cout; p.act = act; p.Ho = H; p.Wo = W;      // the map size picks the kernel (cf_pw.hip)
    if (sc.err == hipSuccess) sc.chk(launch_pw(sc.s, dtype, p));
    sc.to_host_nchw(dtype, p.y, y, B, Cout, H, W);
    return sc.result("cf_op_pwconv");
}

int cf_op_mbconv(int device, int dtype, const float* x, const float* w_exp, const float* w_dw,
                 const float* w_proj, float* y, int B, int Cin, int hid, int Cout, int H, int W, int k, int stride) {
    if (bad_dtype(dtype) || !x || !w_exp || !w_dw || !w_proj || !y || B < 1) return CF_EINVAL;
    MbGeom g = mb_geometry(dtype, Cin, hid, Cout, k, stride);
    if (!g.ok || hid == Cin) { g_op_error = "shape not covered by the fused MBConv kernel"; return CF_EINVAL; }
    const int pd = k - stride > 0 ? k - stride : 0;
    const int Ho = (H + pd - k) / stride + 1, Wo = (W + pd - k) / stride + 1;
    Scope sc(device);
    std::vector<char> we(g.wexp_bytes), wp(g.wproj_bytes);
    std::vector<float> wd(g.wdw_floats);
    mb_pack_weights(dtype, g, Cin, hid, Cout, k, w_exp, w_dw, w_proj, we.data(), wd.data(), wp.data());
    MbParams p{};
    p.x = sc.to_nhwc(dtype, x, B, Cin, H, W);
    p.y = sc.alloc((size_t)B * Ho * Wo * Cout * elem_size(dtype));
    p.wexp = sc.up(we.data(), we.size()); p.wdw = sc.upv(wd); p.wproj = sc.up(wp.data(), wp.size());
    p.B = B; p.Hin = H; p.Win = W; p.Hout = Ho; p.Wout = Wo; p.Cin = Cin; p.hid = hid; p.Cout = Cout;
    p.k = k; p.s = stride; p.pad_lo = pd / 2; p.residual = (Cin == Cout && stride == 1) ? 1 : 0;
    p.HC = g.HC; p.nq = g.nq; p.NBE = g.NBE; p.JX = g.JX; p.HALF = g.HALF; p.rowb = g.rowb; p.lds_bytes = g.lds_bytes; p.kind = g.kind;
#ifdef CF_X5_TIMING      // A/B build only: per-wave phase cycle sums of mbconv_f32_kernel (kind 7), printed to stderr
    const size_t tn = (size_t)1 << 22;
    unsigned long long* tdev = (unsigned long long*)sc.alloc(tn * 8);
    (void)hipMemsetAsync(tdev, 0, tn * 8, sc.s); p.dbg = tdev;
    for (int rep = 0; rep < 2 && sc.err == hipSuccess; ++rep) sc.chk(launch_mbconv(sc.s, dtype, p));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, sc.s);
#endif
    if (sc.err == hipSuccess) sc.chk(launch_mbconv(sc.s, dtype, p));
#ifdef CF_X5_TIMING
    (void)hipEventRecord(e1, sc.s); (void)hipStreamSynchronize(sc.s);
    { float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> th(tn);
      (void)hipMemcpy(th.data(), tdev, tn * 8, hipMemcpyDeviceToHost);
      double sum[4] = {0, 0, 0, 0}; size_t nw = 0;
      for (size_t i = 0; i + 3 < tn; i += 4) if (th[i] | th[i + 1] | th[i + 2] | th[i + 3]) { for (int k = 0; k < 4; ++k) sum[k] += (double)th[i + k]; ++nw; }
      if (nw) fprintf(stderr, "mbconv timing (kind %d): %.1f us; %zu waves, %d chunks each; mean cycles per wave and chunk: wait-top %.0f  expand %.0f  barrier %.0f  dw+project %.0f  (sum %.0f)\n",
                      g.kind, ms * 1e3, nw, g.nq, sum[0] / nw / g.nq, sum[1] / nw / g.nq, sum[2] / nw / g.nq, sum[3] / nw / g.nq, (sum[0] + sum[1] + sum[2] + sum[3]) / nw / g.nq); }
#endif
    sc.to_host_nchw(dtype, p.y, y, B, Cout, Ho, Wo);
    return sc.result("cf_op_mbconv");
}

int cf_op_expand_dw(int device, int dtype, const float* x, const float* w_exp, const float* w_dw, float* y,
                    int B, int Cin, int hid, int H, int W, int k, int stride) {
    if (bad_dtype(dtype) || !x || !w_exp || !w_dw || !y || B < 1) return CF_EINVAL;
    MbGeom g = expdw_geometry(dtype, Cin, hid, k, stride);
    if (!g.ok) { g_op_error = "shape / dtype not covered by the expand+depthwise kernel"; return CF_EINVAL; }
    const int pd = k - stride > 0 ? k - stride : 0;
    const int Ho = (H + pd - k) / stride + 1, Wo = (W + pd - k) / stride + 1;
    Scope sc(device);
    std::vector<char> we(g.wexp_bytes);
    std::vector<float> wd(g.wdw_floats);
    mb_pack_weights(dtype, g, Cin, hid, hid, k, w_exp, w_dw, nullptr, we.data(), wd.data(), nullptr);
    MbParams p{};
    p.x = sc.to_nhwc(dtype, x, B, Cin, H, W);
    p.y = sc.alloc((size_t)B * Ho * Wo * hid * elem_size(dtype));
    p.wexp = sc.up(we.data(), we.size()); p.wdw = sc.upv(wd);
    p.B = B; p.Hin = H; p.Win = W; p.Hout = Ho; p.Wout = Wo; p.Cin = Cin; p.hid = hid; p.Cout = hid;
    p.k = k; p.s = stride; p.pad_lo = pd / 2;
    p.HC = g.HC; p.nq = g.nq; p.NBE = g.NBE; p.JX = g.JX; p.HALF = g.HALF; p.rowb = g.rowb; p.lds_bytes = g.lds_bytes; p.kind = g.kind;
#ifdef CF_X5_TIMING      // A/B build only (tools/ab_build.sh): per-wave phase cycle sums of expdw_f32_kernel, printed to stderr
    const size_t tn = (size_t)1 << 20;
    unsigned long long* tdev = (unsigned long long*)sc.alloc(tn * 8);
    if (g.kind == 8) { (void)hipMemsetAsync(tdev, 0, tn * 8, sc.s); p.wproj = tdev; }
    for (int rep = 0; rep < 3 && sc.err == hipSuccess; ++rep) sc.chk(launch_mbconv(sc.s, dtype, p));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, sc.s);
#endif
    if (sc.err == hipSuccess) sc.chk(launch_mbconv(sc.s, dtype, p));
#ifdef CF_X5_TIMING
    (void)hipEventRecord(e1, sc.s); (void)hipStreamSynchronize(sc.s);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (g.kind == 8) {
        std::vector<unsigned long long> th(tn);
        (void)hipMemcpy(th.data(), tdev, tn * 8, hipMemcpyDeviceToHost);
        double sum[4] = {0, 0, 0, 0}; size_t nw = 0;
        for (size_t i = 0; i + 3 < tn; i += 4) if (th[i] | th[i + 1] | th[i + 2] | th[i + 3]) { for (int k = 0; k < 4; ++k) sum[k] += (double)th[i + k]; ++nw; }
        const int chunks = g.HALF < g.nq ? g.HALF : g.nq;
        fprintf(stderr, "x5 timing: %.1f us; %zu waves, %d chunks each; mean cycles per wave and chunk: wait-top %.0f  phase1 %.0f  barrier %.0f  phase2 %.0f  (sum %.0f)\n",
                ms * 1e3, nw, chunks, sum[0] / nw / chunks, sum[1] / nw / chunks, sum[2] / nw / chunks, sum[3] / nw / chunks, (sum[0] + sum[1] + sum[2] + sum[3]) / nw / chunks);
    }
#endif
    sc.to_host_nchw(dtype, p.y, y, B, hid, Ho, Wo);
    return sc.result("cf_op_expand_dw");
}

int cf_op_ctdet_loss(int device, const float* hm_raw, const float* wh, const float* reg, const float* lm,
                     int B, int h, int w, const float* gt_hm, const uint8_t* reg_mask, const int64_t* ind,
                     const float* wh_t, const float* reg_t, const uint8_t* lm_mask, const int64_t* lm_ind,
                     const float* lm_t, int max_objs, const float* weights4, float* out5) {
    if (!hm_raw || !wh || !reg || !lm || !gt_hm || !reg_mask || !ind || !wh_t || !reg_t || !lm_mask || !lm_ind || !lm_t ||
        !weights4 || !out5 || B < 1 || h < 1 || w < 1 || max_objs < 1) return CF_EINVAL;
    Scope sc(device);
    const size_t hw = (size_t)h * w, bm = (size_t)B * max_objs;
    LossParams p{};
    p.hm_raw = (const float*)sc.up(hm_raw, B * hw * 4); p.wh = (const float*)sc.up(wh, B * 2 * hw * 4);
    p.reg = (const float*)sc.up(reg, B * 2 * hw * 4); p.lm = (const float*)sc.up(lm, B * 10 * hw * 4);
    p.gt_hm = (const float*)sc.up(gt_hm, B * hw * 4);
    p.reg_mask = (const unsigned char*)sc.up(reg_mask, bm); p.ind = (const long long*)sc.up(ind, bm * 8);
    p.wh_t = (const float*)sc.up(wh_t, bm * 8); p.reg_t = (const float*)sc.up(reg_t, bm * 8);
    p.lm_mask = (const unsigned char*)sc.up(lm_mask, bm); p.lm_ind = (const long long*)sc.up(lm_ind, bm * 8);
    p.lm_t = (const float*)sc.up(lm_t, bm * 40);
    p.B = B; p.h = h; p.w = w; p.M = max_objs;
    p.hm_w = weights4[0]; p.wh_w = weights4[1]; p.off_w = weights4[2]; p.lm_w = weights4[3];
    const int nblocks = 256;
    double* ws = (double*)sc.alloc((3 * nblocks + 6) * sizeof(double));
    float* out = (float*)sc.alloc(5 * sizeof(float));
    if (sc.err == hipSuccess) sc.chk(launch_ctdet_loss(sc.s, p, ws, nblocks, out));
    if (sc.err == hipSuccess) sc.chk(hipMemcpyAsync(out5, out, 5 * sizeof(float), hipMemcpyDeviceToHost, sc.s));
    return sc.result("cf_op_ctdet_loss");
}

int cf_op_encode_targets(int device, const float* boxes, const float* lms, const int32_t* counts, int B, int h, int w,
                         int max_objs, float* hm, float* wh, float* reg, int64_t* ind, uint8_t* reg_mask,
                         float* landmarks, int64_t* lm_ind, uint8_t* lm_mask) {
    if (!boxes || !lms || !counts || !hm || !wh || !reg || !ind || !reg_mask || !landmarks || !lm_ind || !lm_mask ||
        B < 1 || h < 1 || w < 1 || max_objs < 1) return CF_EINVAL;
    for (int b = 0; b < B; ++b) if (counts[b] < 0 || counts[b] > max_objs) return CF_EINVAL;
    Scope sc(device);
    const size_t hw = (size_t)h * w, bm = (size_t)B * max_objs;
    EncodeParams p{};
    p.boxes = (const float*)sc.up(boxes, bm * 16); p.lms = (const float*)sc.up(lms, bm * 40); p.counts = (const int*)sc.up(counts, (size_t)B * 4);
    p.hm = (float*)sc.alloc(B * hw * 4); p.wh = (float*)sc.alloc(bm * 8); p.reg = (float*)sc.alloc(bm * 8);
    p.ind = (long long*)sc.alloc(bm * 8); p.reg_mask = (unsigned char*)sc.alloc(bm);
    p.landmarks = (float*)sc.alloc(bm * 40); p.lm_ind = (long long*)sc.alloc(bm * 8); p.lm_mask = (unsigned char*)sc.alloc(bm);
    p.B = B; p.h = h; p.w = w; p.M = max_objs;
    if (sc.err == hipSuccess) sc.chk(launch_encode_targets(sc.s, p));
    auto down = [&](void* dst, const void* src, size_t bytes) { if (sc.err == hipSuccess) sc.chk(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, sc.s)); };
    down(hm, p.hm, B * hw * 4); down(wh, p.wh, bm * 8); down(reg, p.reg, bm * 8); down(ind, p.ind, bm * 8);
    down(reg_mask, p.reg_mask, bm); down(landmarks, p.landmarks, bm * 40); down(lm_ind, p.lm_ind, bm * 8); down(lm_mask, p.lm_mask, bm);
    return sc.result("cf_op_encode_targets");
}

int cf_op_stem(int device, int dtype, const void* x, int in_format, const float* w, float* y, int B, int H, int W) {
    if (bad_dtype(dtype) || !x || !w || !y || (H % 2) || (W % 2)) return CF_EINVAL;
    if (in_format != CF_IN_U8_HWC_BGR && in_format != CF_IN_F32_NCHW) return CF_EINVAL;
    Scope sc(device);
    std::vector<char> wp(stem_packed_bytes(dtype));
    stem_pack_weights(dtype, w, wp.data());
    StemParams p{};
    p.x = sc.up(x, (size_t)B * 3 * H * W * (in_format == CF_IN_U8_HWC_BGR ? 1 : 4));
    p.in_format = in_format; p.w = sc.up(wp.data(), wp.size());
    p.y = sc.alloc((size_t)B * (H / 2) * (W / 2) * 32 * elem_size(dtype));
    p.B = B; p.H = H; p.W = W;
    if (sc.err == hipSuccess) sc.chk(launch_stem(sc.s, dtype, p));
    sc.to_host_nchw(dtype, p.y, y, B, 32, H / 2, W / 2);
    return sc.result("cf_op_stem");
}

int cf_op_idaup(int device, int dtype, const float* lo, const float* skip, const float* w_up,
                const float* bn_up, const float* w_cv, const float* bn_cv, float eps, float* y,
                int B, int C, int Cs, int h, int w) {
    if (bad_dtype(dtype) || !lo || !skip || !w_up || !bn_up || !w_cv || !bn_cv || !y || (C % 8) || (Cs % 8)) return CF_EINVAL;
    Scope sc(device);
    std::vector<double> s1, h1, s2, h2;
    fold(bn_cv, C, eps, s1, h1);
    fold(bn_up, C, eps, s2, h2);
    std::vector<float> wf((size_t)C * Cs), bias(C), uw(4 * C), ub(C);
    for (int n = 0; n < C; ++n) {
        for (int k = 0; k < Cs; ++k) wf[(size_t)n * Cs + k] = (float)((double)w_cv[(size_t)n * Cs + k] * s1[n]);
        bias[n] = (float)h1[n];
        for (int t = 0; t < 4; ++t) uw[(size_t)t * C + n] = (float)((double)w_up[n * 4 + t] * s2[n]);
        ub[n] = (float)h2[n];
    }
    std::vector<char> packed(pw_packed_bytes(dtype, Cs, C));
    pw_pack_weights(dtype, wf.data(), Cs, C, packed.data());
    const int Ho = 2 * h, Wo = 2 * w;
    PwParams p{};
    p.x = sc.to_nhwc(dtype, skip, B, Cs, Ho, Wo);
    p.low = sc.to_nhwc(dtype, lo, B, C, h, w);
    p.wp = sc.up(packed.data(), packed.size());
    p.bias = sc.upv(bias); p.upw = sc.upv(uw); p.upb = sc.upv(ub);
    p.y = sc.alloc((size_t)B * Ho * Wo * C * elem_size(dtype));
    p.M = (long long)B * Ho * Wo; p.K = Cs; p.N = C; p.act = 2; p.Ho = Ho; p.Wo = Wo;
    if (sc.err == hipSuccess) sc.chk(launch_pw(sc.s, dtype, p));
    sc.to_host_nchw(dtype, p.y, y, B, C, Ho, Wo);
    return sc.result("cf_op_idaup");
}

int cf_op_heads(int device, int dtype, const float* x, const float* w0, const float* b0,
                const float* w1, const float* b1, float* out, int B, int h, int w, int collapse) {
    if (bad_dtype(dtype) || !x || !w0 || !b0 || !w1 || !b1 || !out) return CF_EINVAL;
    Scope sc(device);
    std::vector<char> packed(head_packed_bytes(dtype, collapse ? 1 : 0));
    std::vector<float> b0h(96), w1d(96 * 16), b1h(16);
    head_pack_weights(dtype, collapse ? 1 : 0, w0, b0, w1, b1, packed.data(), b0h.data(), w1d.data(), b1h.data());
    HeadParams p{};
    p.x = sc.to_nhwc(dtype, x, B, 24, h, w);
    p.w0p = sc.up(packed.data(), packed.size());
    p.b0 = sc.upv(b0h); p.w1d = sc.upv(w1d); p.b1 = sc.upv(b1h);
    const size_t HW = (size_t)h * w;
    p.heads = (float*)sc.alloc((size_t)B * HW * 16 * 4);
    p.B = B; p.h = h; p.w = w; p.collapsed = collapse ? 1 : 0;
    if (sc.err == hipSuccess) sc.chk(launch_heads(sc.s, dtype, p));
    std::vector<float> rec((size_t)B * HW * 16);
    if (sc.err == hipSuccess) sc.chk(hipMemcpyAsync(rec.data(), p.heads, rec.size() * 4, hipMemcpyDeviceToHost, sc.s));
    int r = sc.result("cf_op_heads");
    if (r) return r;
    for (int b = 0; b < B; ++b)
        for (size_t i = 0; i < HW; ++i) {
            const float* rr = &rec[((size_t)b * HW + i) * 16];
            out[((size_t)b * 15 + 0) * HW + i] = rr[15];                      // raw hm logit
            for (int c = 1; c < 15; ++c) out[((size_t)b * 15 + c) * HW + i] = rr[c];
        }
    return CF_OK;
}

// ShuffleV2Block.forward, eval mode (model/blocks.py:47-62).  BatchNorm is folded here (float64) into the conv
// weights + an fp32 bias; the channel shuffle is ADDRESSING, not data movement on the host: the first 1x1 conv of
// branch_main reads the odd input channels through a zero-interleaved weight matrix over all 2*inp channels (exact:
// the even channels meet zero weights), the pass-through half is copied by a strided-channel kernel straight into
// channels [0, inp) of the block output, and both final 1x1 convs write their slice of the output rows
// (PwParams::ldy / yoff) -- the torch.cat of the reference never materialises.
int cf_op_shufflev2(int device, int dtype, const float* x, float* y, int B, int inp, int oup, int mid, int H, int W,
                    int ksize, int stride, const float* m_w0, const float* m_bn1, const float* m_wdw, const float* m_bn4,
                    const float* m_w5, const float* m_bn6, const float* p_wdw, const float* p_bn1, const float* p_w2,
                    const float* p_bn3) {
    const int outputs = oup - inp;
    if (bad_dtype(dtype) || !x || !y || B < 1 || (ksize != 3 && ksize != 5) || (stride != 1 && stride != 2) || outputs < 8 ||
        (inp % 8) || (mid % 8) || (outputs % 8) || !m_w0 || !m_bn1 || !m_wdw || !m_bn4 || !m_w5 || !m_bn6) return CF_EINVAL;
    if (stride == 2 && (!p_wdw || !p_bn1 || !p_w2 || !p_bn3)) return CF_EINVAL;
    const int Cin = stride == 1 ? 2 * inp : inp, pad = ksize / 2;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const float eps = 1e-5f;                                             // nn.BatchNorm2d default (blocks.py:23,29,32)
    Scope sc(device);
    void* xd = sc.to_nhwc(dtype, x, B, Cin, H, W);
    void* out = sc.alloc((size_t)B * Ho * Wo * oup * elem_size(dtype));
    const long long Min = (long long)B * H * W, Mout = (long long)B * Ho * Wo;

    auto pw = [&](const void* src, long long M, int K, int N, const float* w /*[N][Kw]*/, int Kw, bool interleave, const float* bn,
                  void* dst, int ldy, int yoff) {
        std::vector<double> s1, h1;
        fold(bn, N, eps, s1, h1);
        std::vector<float> wf((size_t)N * K, 0.0f), bias(N);
        for (int n = 0; n < N; ++n) {
            for (int k = 0; k < Kw; ++k) wf[(size_t)n * K + (interleave ? 2 * k + 1 : k)] = (float)((double)w[(size_t)n * Kw + k] * s1[n]);
            bias[n] = (float)h1[n];
        }
        std::vector<char> packed(pw_packed_bytes(dtype, K, N));
        pw_pack_weights(dtype, wf.data(), K, N, packed.data());
        PwParams p{};
        p.x = src; p.wp = sc.up(packed.data(), packed.size()); p.bias = sc.upv(bias); p.y = dst;
        p.M = M; p.K = K; p.N = N; p.act = 2; p.ldy = ldy; p.yoff = yoff;
        if (sc.err == hipSuccess) sc.chk(launch_pw(sc.s, dtype, p));
    };
    auto dw = [&](const void* src, int C, const float* w /*[C][1][k][k]*/, const float* bn) -> void* {
        std::vector<double> s1, h1;
        fold(bn, C, eps, s1, h1);
        std::vector<float> wf((size_t)C * ksize * ksize), wp((size_t)C * ksize * ksize), bias(C);
        for (int c = 0; c < C; ++c) {
            for (int t = 0; t < ksize * ksize; ++t) wf[(size_t)c * ksize * ksize + t] = (float)((double)w[(size_t)c * ksize * ksize + t] * s1[c]);
            bias[c] = (float)h1[c];
        }
        dw_pack_weights(wf.data(), C, ksize, wp.data());
        DwParams p{};
        p.x = src; p.w = sc.upv(wp); p.bias = sc.upv(bias);
        p.y = sc.alloc((size_t)Mout * C * elem_size(dtype));
        p.B = B; p.C = C; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.k = ksize; p.s = stride; p.pad_lo = pad; p.act = 0;
        if (sc.err == hipSuccess) sc.chk(launch_dw(sc.s, dtype, p));
        return p.y;
    };

    // branch_main (:20-34): pw + BN + ReLU -> dw + BN -> pw + BN + ReLU, into channels [inp, oup)
    void* t1 = sc.alloc((size_t)Min * mid * elem_size(dtype));
    pw(xd, Min, Cin, mid, m_w0, inp, stride == 1, m_bn1, t1, 0, 0);
    void* t2 = dw(t1, mid, m_wdw, m_bn4);
    pw(t2, Mout, mid, outputs, m_w5, mid, false, m_bn6, out, oup, inp);
    if (stride == 1) {
        // x_proj = the even channels (:56-62), passed through into channels [0, inp)
        if (sc.err == hipSuccess) sc.chk(launch_shuffle_copy(sc.s, dtype, xd, out, Min, inp, 0, oup, 0));
    } else {
        // branch_proj (:36-45): dw + BN -> pw + BN + ReLU, into channels [0, inp)
        void* p1 = dw(xd, inp, p_wdw, p_bn1);
        pw(p1, Mout, inp, inp, p_w2, inp, false, p_bn3, out, oup, 0);
    }
    sc.to_host_nchw(dtype, out, y, B, oup, Ho, Wo);
    return sc.result("cf_op_shufflev2");
}

int cf_op_ctdet_decode(int device, const float* heat, const float* wh, const float* reg, const float* lm,
                       int B, int h, int w, int K, float* dets, float* lms, int64_t* inds) {
    if (!heat || !wh || !dets || B < 1 || K < 1 || (long long)K > (long long)h * w) return CF_EINVAL;
    Scope sc(device);
    std::vector<float> rec = make_records(heat, wh, reg, lm, B, h, w);
    TopkParams p{};
    p.heads = sc.upv(rec);
    p.scratch = (unsigned long long*)sc.alloc((size_t)B * h * w * 8);
    p.count = (int*)sc.alloc((size_t)B * kTopkCountStride * 4);               // zero-initialised by Scope::alloc
    if (K > 1024) { p.big_stride = topk_big_stride(K); p.big = (unsigned long long*)sc.alloc((size_t)B * p.big_stride * 8); }
    p.B = B; p.h = h; p.w = w; p.K = K; p.use_reg = reg ? 1 : 0;
    p.dets = (float*)sc.alloc((size_t)B * K * 6 * 4);
    p.lms = (lms && lm) ? (float*)sc.alloc((size_t)B * K * 10 * 4) : nullptr;
    p.inds = inds ? (long long*)sc.alloc((size_t)B * K * 8) : nullptr;
    if (sc.err == hipSuccess) sc.chk(launch_peak_topk(sc.s, p));
    if (sc.err == hipSuccess) sc.chk(hipMemcpyAsync(dets, p.dets, (size_t)B * K * 6 * 4, hipMemcpyDeviceToHost, sc.s));
    if (p.lms && sc.err == hipSuccess) sc.chk(hipMemcpyAsync(lms, p.lms, (size_t)B * K * 10 * 4, hipMemcpyDeviceToHost, sc.s));
    if (p.inds && sc.err == hipSuccess) sc.chk(hipMemcpyAsync(inds, p.inds, (size_t)B * K * 8, hipMemcpyDeviceToHost, sc.s));
    return sc.result("cf_op_ctdet_decode");
}

static int run_threshold(Scope& sc, int mode, const float* d_heads, int B, int h, int w, int img_h, int img_w,
                         float score_thresh, float nms_thresh, int cap, int max_out,
                         float* dets, float* lms, int32_t* counts, int* overflow_out) {
    const size_t words = (cap + 63) / 64;
    ThreshParams p{};
    p.heads = d_heads; p.B = B; p.h = h; p.w = w; p.img_h = img_h; p.img_w = img_w;
    p.score_thresh = score_thresh; p.nms_thresh = nms_thresh; p.cap = cap; p.mode = mode;
    p.cand = (float*)sc.alloc((size_t)B * cap * 16 * 4);
    p.cand_count = (int*)sc.alloc((size_t)B * 4);
    p.order = (int*)sc.alloc((size_t)B * cap * 4);
    p.mask = (unsigned long long*)sc.alloc((size_t)B * cap * words * 8);
    p.max_out = max_out;
    p.dets = (float*)sc.alloc((size_t)B * max_out * 5 * 4);
    p.lms = lms ? (float*)sc.alloc((size_t)B * max_out * 10 * 4) : nullptr;
    p.counts = (int*)sc.alloc((size_t)B * 4);
    p.overflow = (int*)sc.alloc(4);
    if (sc.err == hipSuccess) sc.chk(launch_decode_threshold(sc.s, p));
    if (sc.err == hipSuccess) sc.chk(hipMemcpyAsync(dets, p.dets, (size_t)B * max_out * 5 * 4, hipMemcpyDeviceToHost, sc.s));
    if (lms && sc.err == hipSuccess) sc.chk(hipMemcpyAsync(lms, p.lms, (size_t)B * max_out * 10 * 4, hipMemcpyDeviceToHost, sc.s));
    if (sc.err == hipSuccess) sc.chk(hipMemcpyAsync(counts, p.counts, (size_t)B * 4, hipMemcpyDeviceToHost, sc.s));
    if (sc.err == hipSuccess) sc.chk(hipMemcpyAsync(overflow_out, p.overflow, 4, hipMemcpyDeviceToHost, sc.s));
    return sc.result("decode_threshold");
}

int cf_op_decode_threshold(int device, const float* hm, const float* wh, const float* lm,
                           int B, int h, int w, int img_h, int img_w, float score_thresh,
                           float nms_thresh, int max_out, float* dets, float* lms, int32_t* counts) {
    return cf_op_decode_threshold_ex(device, 0, hm, wh, nullptr, lm, B, h, w, img_h, img_w, score_thresh, nms_thresh,
                                     max_out, dets, lms, counts);
}

int cf_op_decode_threshold_ex(int device, int mode, const float* hm, const float* wh, const float* reg, const float* lm,
                              int B, int h, int w, int img_h, int img_w, float score_thresh,
                              float nms_thresh, int max_out, float* dets, float* lms, int32_t* counts) {
    if (!hm || !wh || !dets || !counts || B < 1 || max_out < 1 || (mode != 0 && mode != 1) || (mode == 1 && !reg)) return CF_EINVAL;
    Scope sc(device);
    std::vector<float> rec = make_records(hm, wh, reg, lm, B, h, w);
    const float* d_heads = sc.upv(rec);
    const int HW = h * w;
    int cap = HW < 4096 ? (HW + 63) / 64 * 64 : 4096;
    for (int attempt = 0; attempt < 2; ++attempt) {            // grow to the largest candidate count and rerun (see cf_decode_threshold_ex)
        int overflow = 0;
        int r = run_threshold(sc, mode, d_heads, B, h, w, img_h, img_w, score_thresh, nms_thresh, cap, max_out, dets, lms, counts, &overflow);
        if (r) return r;
        if (overflow <= cap) return CF_OK;
        cap = ((overflow < HW ? overflow : HW) + 63) / 64 * 64;
    }
    g_op_error = "candidate capacity exceeded";
    return CF_EOVERFLOW;
}

int cf_op_ctdet_post_process(int device, float* dets, const float* centers, const float* scales, int B, int K, int dim,
                             int out_w, int out_h) {
    if (!dets || !centers || !scales || B < 1 || K < 1 || dim < 4) return CF_EINVAL;
    Scope sc(device);
    std::vector<double> t((size_t)B * 6);
    for (int b = 0; b < B; ++b) cf_affine_from_center_scale(centers[2 * b], centers[2 * b + 1], scales[2 * b], out_w, out_h, &t[(size_t)b * 6]);
    float* d = (float*)sc.up(dets, (size_t)B * K * dim * 4);
    double* dt = sc.upv(t);
    if (sc.err == hipSuccess) sc.chk(launch_affine_boxes(sc.s, d, dt, B, K, dim));
    if (sc.err == hipSuccess) sc.chk(hipMemcpyAsync(dets, d, (size_t)B * K * dim * 4, hipMemcpyDeviceToHost, sc.s));
    return sc.result("cf_op_ctdet_post_process");
}

int cf_op_nms(int device, const float* boxes, const float* scores, int n, float nms_thresh, int32_t* keep, int32_t* n_keep) {
    // CenterFace.nms alone: feed the candidates through the D1 kernels' rank/mask/sweep stages by
    // presenting them as pre-collected candidates.
    if (!boxes || !scores || !keep || !n_keep || n < 0) return CF_EINVAL;
    if (n == 0) { *n_keep = 0; return CF_OK; }
    Scope sc(device);
    const int cap = (n + 63) / 64 * 64;
    const size_t words = cap / 64;
    std::vector<float> cand((size_t)cap * 16, 0.0f);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < 4; ++j) cand[(size_t)i * 16 + j] = boxes[i * 4 + j];
        cand[(size_t)i * 16 + 4] = scores[i];
        cand[(size_t)i * 16 + 5] = (float)i;           // carried through as "landmark 0" = original index
    }
    ThreshParams p{};
    p.B = 1; p.cap = cap; p.nms_thresh = nms_thresh;
    p.cand = sc.upv(cand);
    p.cand_count = (int*)sc.up(&n, 4);
    p.order = (int*)sc.alloc((size_t)cap * 4);
    p.mask = (unsigned long long*)sc.alloc((size_t)cap * words * 8);
    p.max_out = n;
    p.dets = (float*)sc.alloc((size_t)n * 5 * 4);
    p.lms = (float*)sc.alloc((size_t)n * 10 * 4);
    p.counts = (int*)sc.alloc(4);
    p.overflow = (int*)sc.alloc(4);
    if (sc.err == hipSuccess) sc.chk(launch_nms_stages(sc.s, p));
    std::vector<float> l((size_t)n * 10);
    int cnt = 0;
    if (sc.err == hipSuccess) sc.chk(hipMemcpyAsync(l.data(), p.lms, l.size() * 4, hipMemcpyDeviceToHost, sc.s));
    if (sc.err == hipSuccess) sc.chk(hipMemcpyAsync(&cnt, p.counts, 4, hipMemcpyDeviceToHost, sc.s));
    int r = sc.result("cf_op_nms");
    if (r) return r;
    for (int i = 0; i < cnt; ++i) keep[i] = (int32_t)l[(size_t)i * 10];
    *n_keep = cnt;
    return CF_OK;
}

// bbox_overlap (eval_widerface.py:48-74) and the match counts of evaluate (:195-206) for n_img images at once: boxes / query are
// the concatenated rows, box_off / query_off [n_img + 1] the first row of each image.  overlaps (optional): the dense [N_i][K_i]
// float64 matrices back to back; counts (optional) [n_img][2].
int cf_op_box_match(int device, int n_img, const float* boxes, int box_stride, const int32_t* box_off, const float* query, int query_stride,
                    const int32_t* query_off, float thresh, double* overlaps, int32_t* counts) {
    if (n_img < 0 || !box_off || !query_off || box_stride < 4 || query_stride < 4) return CF_EINVAL;
    if (n_img == 0) return CF_OK;
    const int NB = box_off[n_img], NQ = query_off[n_img];
    if (box_off[0] != 0 || query_off[0] != 0 || (NB > 0 && !boxes) || (NQ > 0 && !query)) return CF_EINVAL;
    std::vector<long long> ooff(n_img + 1, 0);
    for (int i = 0; i < n_img; ++i) {
        const int N = box_off[i + 1] - box_off[i], K = query_off[i + 1] - query_off[i];
        if (N < 0 || K < 0) return CF_EINVAL;
        ooff[i + 1] = ooff[i] + (long long)N * K;
    }
    Scope sc(device);
    OverlapParams p{};
    p.n_img = n_img; p.box_stride = box_stride; p.query_stride = query_stride; p.thresh = thresh;
    float dummy[4] = {0, 0, 0, 0};
    p.boxes = (const float*)sc.up(NB ? (const void*)boxes : (const void*)dummy, NB ? (size_t)NB * box_stride * 4 : 16);
    p.query = (const float*)sc.up(NQ ? (const void*)query : (const void*)dummy, NQ ? (size_t)NQ * query_stride * 4 : 16);
    p.box_off = (const int*)sc.up(box_off, (size_t)(n_img + 1) * 4);
    p.query_off = (const int*)sc.up(query_off, (size_t)(n_img + 1) * 4);
    if (overlaps && ooff[n_img] > 0) {
        p.overlaps = (double*)sc.alloc((size_t)ooff[n_img] * 8);
        p.overlaps_off = (const long long*)sc.up(ooff.data(), ooff.size() * 8);
    }
    if (counts) p.counts = (int*)sc.alloc((size_t)n_img * 2 * 4);
    if (sc.err == hipSuccess) sc.chk(launch_box_match(sc.s, p));
    if (sc.err == hipSuccess && p.overlaps) sc.chk(hipMemcpyAsync(overlaps, p.overlaps, (size_t)ooff[n_img] * 8, hipMemcpyDeviceToHost, sc.s));
    if (sc.err == hipSuccess && counts) sc.chk(hipMemcpyAsync(counts, p.counts, (size_t)n_img * 2 * 4, hipMemcpyDeviceToHost, sc.s));
    return sc.result("cf_op_box_match");
}

}  // extern "C"
