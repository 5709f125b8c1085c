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
// dst(Y, X) samples src at ((Y + 0.5) * h / H - 0.5, (X + 0.5) * w / W - 0.5), clamped to the image,
// float32 arithmetic, round-to-nearest-even.  cv2's INTER_LINEAR uses 11-bit fixed-point weights for
// uint8; cv2 is not installable here, so bit parity with it is UNPINNED (documented in DESIGN.md).
__global__ void resize_u8_kernel(const uint8_t* src, uint8_t* dst, int B, int h, int w, int H, int W) {
    const long long n = (long long)B * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int X = (int)(i % W); const long long t = i / W;
    const int Y = (int)(t % H); const int b = (int)(t / H);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    float fy = ((float)Y + 0.5f) * sy - 0.5f, fx = ((float)X + 0.5f) * sx - 0.5f;
    fy = fminf(fmaxf(fy, 0.0f), (float)(h - 1)); fx = fminf(fmaxf(fx, 0.0f), (float)(w - 1));
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const uint8_t* p00 = src + (((size_t)b * h + y0) * w + x0) * 3;
    const uint8_t* p01 = src + (((size_t)b * h + y0) * w + x1) * 3;
    const uint8_t* p10 = src + (((size_t)b * h + y1) * w + x0) * 3;
    const uint8_t* p11 = src + (((size_t)b * h + y1) * w + x1) * 3;
    uint8_t* d = dst + (size_t)i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = (float)p00[c] * (1.0f - wx) + (float)p01[c] * wx;
        const float bot = (float)p10[c] * (1.0f - wx) + (float)p11[c] * wx;
        const float v = top * (1.0f - wy) + bot * wy;
        d[c] = (uint8_t)fminf(fmaxf(rintf(v), 0.0f), 255.0f);
    }
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
