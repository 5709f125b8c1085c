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
