// Stem: ConvReLU(3, 32, 3, stride=2) = ZeroPad2d(0,1,0,1) -> Conv2d(3,32,3,s2,bias=False) -> Swish
// (model/centernet.py:224 via :58-70), with the reference's host-side preprocessing fused in front
// when the input is a uint8 BGR image: x/255, (x-mean)/std (centerface.py:12-15,32-33).
//
// One lane = one output pixel, all 32 output channels in registers (the output, 64 B/pixel in bf16,
// dominates the traffic: stores are four 16-byte vectors per lane, a wave writes 4 KiB contiguous).
// The 27x32 weight table is wave-uniform and lives in LDS (broadcast reads).  Also contains the
// NCHW<->NHWC converters used at the test boundary.
#include "cf_common.h"
#include "cf_kernels.h"
#include "centerface_hip.h"
#include <cstdarg>
#include <cstdio>

namespace cf {

static thread_local char g_kernel_tag[160] = "";
const char* last_kernel_tag() { return g_kernel_tag; }
void set_kernel_tag(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_kernel_tag, sizeof g_kernel_tag, fmt, ap); va_end(ap);
}

void stem_pack_weights(const float* w, float* out_host) {
    for (int co = 0; co < 32; ++co)
        for (int ci = 0; ci < 3; ++ci)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx)
                    out_host[((ky * 3 + kx) * 3 + ci) * 32 + co] = w[((co * 3 + ci) * 3 + ky) * 3 + kx];
}

template <typename T, int FMT>
__global__ __launch_bounds__(256) void stem_kernel(StemParams p) {
    __shared__ __attribute__((aligned(16))) float wl[27 * 32];
    for (int i = threadIdx.x; i < 27 * 32; i += 256) wl[i] = p.w[i];
    __syncthreads();

    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const long long total = (long long)p.B * Ho * Wo;
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= total) return;
    const int b = (int)(m / ((long long)Ho * Wo));
    const int rem = (int)(m - (long long)b * Ho * Wo);
    const int yo = rem / Wo, xo = rem - yo * Wo;

    // centerface.py:12-15 (BGR order)
    const float mean[3] = {0.408f, 0.447f, 0.470f};
    const float stdv[3] = {0.289f, 0.274f, 0.278f};

    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.0f;

#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * yo + ky;                       // pad_lo = 0, pad_hi = 1
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * xo + kx;
            const bool ok = iy < p.H && ix < p.W;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                float v = 0.0f;
                if (ok) {
                    if constexpr (FMT == CF_IN_U8_HWC_BGR) {
                        const uint8_t* px = (const uint8_t*)p.x + (((size_t)b * p.H + iy) * p.W + ix) * 3;
                        v = ((float)px[ci] / 255.0f - mean[ci]) / stdv[ci];   // IEEE divisions, as numpy
                    } else {
                        v = ((const float*)p.x)[(((size_t)b * 3 + ci) * p.H + iy) * p.W + ix];
                    }
                }
                const float* wr = &wl[((ky * 3 + kx) * 3 + ci) * 32];
#pragma unroll
                for (int c = 0; c < 32; ++c) acc[c] = fmaf(v, wr[c], acc[c]);
            }
        }
    }
    constexpr int P = Elem<T>::PER16;
    T* out = (T*)p.y + (size_t)m * 32;
#pragma unroll
    for (int g = 0; g < 32 / P; ++g) {
        float o[P];
#pragma unroll
        for (int e = 0; e < P; ++e) o[e] = swish_f(acc[g * P + e]);
        st16(out + g * P, pack16<T>(o));
    }
}

hipError_t launch_stem(hipStream_t s, int dtype, const StemParams& p) {
    if (p.B <= 0) return hipSuccess;
    const long long total = (long long)p.B * (p.H / 2) * (p.W / 2);
    dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    set_kernel_tag("void cf::stem_kernel<%s, %d>(cf::StemParams)", dtype == 0 ? "float" : "unsigned short", p.in_format);
    if (dtype == 0) {
        if (p.in_format == CF_IN_U8_HWC_BGR) hipLaunchKernelGGL((stem_kernel<float, CF_IN_U8_HWC_BGR>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_kernel<float, CF_IN_F32_NCHW>), grid, blk, 0, s, p);
    } else {
        if (p.in_format == CF_IN_U8_HWC_BGR) hipLaunchKernelGGL((stem_kernel<bf16_t, CF_IN_U8_HWC_BGR>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_kernel<bf16_t, CF_IN_F32_NCHW>), grid, blk, 0, s, p);
    }
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
static unsigned conv_grid(long long n) { long long g = (n + 255) / 256; return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

hipError_t launch_nchw_to_nhwc(hipStream_t s, int dtype, const float* src, void* dst, int B, int C, int H, int W) {
    long long n = (long long)B * C * H * W;
    if (n == 0) return hipSuccess;
    if (dtype == 0) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(conv_grid(n)), dim3(256), 0, s, src, (float*)dst, B, C, H, W);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(conv_grid(n)), dim3(256), 0, s, src, (bf16_t*)dst, B, C, H, W);
    return hipGetLastError();
}
hipError_t launch_nhwc_to_nchw(hipStream_t s, int dtype, const void* src, float* dst, int B, int C, int H, int W) {
    long long n = (long long)B * C * H * W;
    if (n == 0) return hipSuccess;
    if (dtype == 0) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(conv_grid(n)), dim3(256), 0, s, (const float*)src, dst, B, C, H, W);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(conv_grid(n)), dim3(256), 0, s, (const bf16_t*)src, dst, B, C, H, W);
    return hipGetLastError();
}

}  // namespace cf
