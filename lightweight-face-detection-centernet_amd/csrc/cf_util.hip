// Small shared pieces: kernel-symbol tags for the profiler and the NCHW<->NHWC converters used at
// the test boundary (cf_get_heads / cf_op_*).
#include "cf_common.h"
#include "cf_kernels.h"
#include <cstdarg>
#include <cstdio>

namespace cf {

static thread_local char g_kernel_tag[160] = "";
const char* last_kernel_tag() { return g_kernel_tag; }
void set_kernel_tag(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_kernel_tag, sizeof g_kernel_tag, fmt, ap); va_end(ap);
}

// ---------------------------------------------------------------- bilinear resize (centerface.py:30)
// cv2.resize(img, (W, H)) with the default INTER_LINEAR on uint8 is FIXED-POINT in OpenCV (third-party code, not
// under /root/reference and not installable here; algorithm restated from OpenCV 4.x modules/imgproc/src/resize.cpp,
// the generic path all SIMD paths are bit-exact with):
//   * per destination column: fx = (float)((dx + 0.5) * scale_x - 0.5) with scale_x = 1.0 / ((double)W / w),
//     sx = floor(fx), fx -= sx; sx < 0 -> (sx, fx) = (0, 0); sx >= w - 1 -> (w - 1, 0);
//     coefficients as shorts with 11 fractional bits: a0 = cvRound((1.f - fx) * 2048), a1 = cvRound(fx * 2048);
//   * rows likewise (fy, sy, b0, b1), except that out-of-range rows are CLAMPED (sy + k -> [0, h - 1]) and the
//     coefficients kept;
//   * horizontal pass in int32: r = S[sx] * a0 + S[sx + 1] * a1;
//   * vertical pass (VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>):
//     dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
// cvRound = round-half-to-even (rintf).  Parity with an actual cv2 build is UNPINNED (no cv2 anywhere we can run);
// the oracle restates the same published algorithm and the known answers in the tests (identity, exact 2x
// patterns) are derived by hand from it.
__device__ __forceinline__ void cv_linear_coeffs(int d, int src, int dst, bool clamp_coeff, int& s0, int& s1, int& c0, int& c1) {
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (clamp_coeff) {                                   // columns: coefficient reset at the borders
        if (si < 0) { f = 0.0f; si = 0; }
        if (si >= src - 1) { f = 0.0f; si = src - 1; }
        s0 = si; s1 = min(si + 1, src - 1);
    } else {                                             // rows: indices clamped, coefficients kept
        s0 = min(max(si, 0), src - 1); s1 = min(max(si + 1, 0), src - 1);
    }
    c0 = (int)rintf((1.0f - f) * 2048.0f);
    c1 = (int)rintf(f * 2048.0f);
}
__global__ void resize_u8_kernel(const uint8_t* src, uint8_t* dst, int B, int h, int w, int H, int W) {
    const long long n = (long long)B * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int X = (int)(i % W); const long long t = i / W;
    const int Y = (int)(t % H); const int b = (int)(t / H);
    int x0, x1, a0, a1, y0, y1, b0, b1;
    cv_linear_coeffs(X, w, W, true, x0, x1, a0, a1);
    cv_linear_coeffs(Y, h, H, false, y0, y1, b0, b1);
    const uint8_t* p00 = src + (((size_t)b * h + y0) * w + x0) * 3;
    const uint8_t* p01 = src + (((size_t)b * h + y0) * w + x1) * 3;
    const uint8_t* p10 = src + (((size_t)b * h + y1) * w + x0) * 3;
    const uint8_t* p11 = src + (((size_t)b * h + y1) * w + x1) * 3;
    uint8_t* d = dst + (size_t)i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int r0 = (int)p00[c] * a0 + (int)p01[c] * a1;
        const int r1 = (int)p10[c] * a0 + (int)p11[c] * a1;
        const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        d[c] = (uint8_t)min(max(v, 0), 255);
    }
}
// ---- host images -> HBM by a kernel (zero-copy reads of page-locked host memory over PCIe) -------------------------------------
// One launch moves up to 64 separately allocated images: the DMA engines need one command per image (8-10 us of idle link between
// two ~22 us copies of a VGA image) and ~0.3 ms from the end of a copy to the start of a kernel that waits for it on another stream
// (profiles/r05_vga_pipeline.md); a kernel has neither.  blockIdx.y = image, 16-byte loads, UNR in flight per lane: 48 workgroups x
// 256 lanes x 4 x 16 B = 786 KB in flight covers the link's bandwidth-delay product many times.
struct UploadPtrs { const uint8_t* src[64]; };
template <int UNR>
__global__ void __launch_bounds__(256) upload_images_kernel(UploadPtrs tab, uint8_t* dst, long long bytes) {
    const uint8_t* src = tab.src[blockIdx.y];
    uint8_t* out = dst + (size_t)blockIdx.y * bytes;
    const long long nvec = bytes >> 4;                                   // both ends 16-byte aligned (checked by the launcher)
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* s4 = reinterpret_cast<const u32x4*>(src);
    u32x4* d4 = reinterpret_cast<u32x4*>(out);
    for (; i + (UNR - 1) * stride < nvec; i += UNR * stride) {
        u32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(s4 + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNR; ++u) d4[i + u * stride] = v[u];
    }
    for (; i < nvec; i += stride) d4[i] = __builtin_nontemporal_load(s4 + i);
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) out[(nvec << 4) + threadIdx.x] = src[(nvec << 4) + threadIdx.x];
}
// imgs: DEVICE-visible addresses of B page-locked host images of `bytes` each; dst: [B][bytes] on the device (16-byte aligned)
hipError_t launch_upload_images(hipStream_t s, const void* const* imgs, uint8_t* dst, int B, long long bytes) {
    if (B <= 0 || bytes <= 0) return hipSuccess;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int nb = B - b0 < 64 ? B - b0 : 64;
        UploadPtrs tab{};
        for (int b = 0; b < nb; ++b) tab.src[b] = (const uint8_t*)imgs[b0 + b];
        int gx = (int)((bytes / 16 + 256 * 4 - 1) / (256 * 4));
        const int want = (96 + nb - 1) / nb;                             // ~96 workgroups per launch: the link, not the CUs, is the limit
        if (gx > want) gx = want;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL(upload_images_kernel<4>, dim3(gx, nb), dim3(256), 0, s, tab, dst + (size_t)b0 * bytes, bytes);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_resize_u8(hipStream_t s, const uint8_t* src, uint8_t* dst, int B, int h, int w, int H, int W) {
    const long long n = (long long)B * H * W;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(resize_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, B, h, w, H, W);
    return hipGetLastError();
}

// ---------------------------------------------------------------- layout converters (test boundary)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* src, T* dst, int B, int C, int H, int W) {
    const long long n = (long long)B * C * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C); long long t = i / C;
        int x = (int)(t % W); t /= W;
        int y = (int)(t % H); int b = (int)(t / H);
        float v = src[(((size_t)b * C + c) * H + y) * W + x];
        if constexpr (sizeof(T) == 4) dst[i] = v;
        else dst[i] = (T)(pack_bf16x2(v, 0.0f) & 0xffffu);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* src, float* dst, int B, int C, int H, int W) {
    const long long n = (long long)B * C * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % W); long long t = i / W;
        int y = (int)(t % H); t /= H;
        int c = (int)(t % C); int b = (int)(t / C);
        T v = src[(((size_t)b * H + y) * W + x) * C + c];
        if constexpr (sizeof(T) == 4) dst[i] = v;
        else dst[i] = bf16_to_f32(v);
    }
}
// pixel-block order [m / 32][C / P][m % 32][P] (P = channels per 16 bytes: 8 bf16 / 4 fp32), m = (b H + y) W + x  ->  NCHW float32
template <typename T>
__global__ void blocked_to_nchw_kernel(const T* src, float* dst, int B, int C, int H, int W) {
    constexpr int P = 16 / (int)sizeof(T);
    const long long n = (long long)B * C * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % W); long long t = i / W;
        int y = (int)(t % H); t /= H;
        int c = (int)(t % C); int b = (int)(t / C);
        const size_t m = ((size_t)b * H + y) * W + x;
        const T v = src[(((m >> 5) * (size_t)(C / P) + (size_t)(c / P)) * 32 + (m & 31)) * P + (c % P)];
        if constexpr (sizeof(T) == 4) dst[i] = v;
        else dst[i] = bf16_to_f32(v);
    }
}
static unsigned conv_grid(long long n) { long long g = (n + 255) / 256; return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

hipError_t launch_nchw_to_nhwc(hipStream_t s, int dtype, const float* src, void* dst, int B, int C, int H, int W) {
    long long n = (long long)B * C * H * W;
    if (n == 0) return hipSuccess;
    if (dtype != 1) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(conv_grid(n)), dim3(256), 0, s, src, (float*)dst, B, C, H, W);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(conv_grid(n)), dim3(256), 0, s, src, (bf16_t*)dst, B, C, H, W);
    return hipGetLastError();
}
hipError_t launch_blocked_to_nchw(hipStream_t s, int dtype, const void* src, float* dst, int B, int C, int H, int W) {
    long long n = (long long)B * C * H * W;
    if (n == 0) return hipSuccess;
    if (dtype != 1) hipLaunchKernelGGL(blocked_to_nchw_kernel<float>, dim3(conv_grid(n)), dim3(256), 0, s, (const float*)src, dst, B, C, H, W);
    else hipLaunchKernelGGL(blocked_to_nchw_kernel<bf16_t>, dim3(conv_grid(n)), dim3(256), 0, s, (const bf16_t*)src, dst, B, C, H, W);
    return hipGetLastError();
}
hipError_t launch_nhwc_to_nchw(hipStream_t s, int dtype, const void* src, float* dst, int B, int C, int H, int W) {
    long long n = (long long)B * C * H * W;
    if (n == 0) return hipSuccess;
    if (dtype != 1) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(conv_grid(n)), dim3(256), 0, s, (const float*)src, dst, B, C, H, W);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(conv_grid(n)), dim3(256), 0, s, (const bf16_t*)src, dst, B, C, H, W);
    return hipGetLastError();
}

}  // namespace cf
