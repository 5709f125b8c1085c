// Does a small MFMA overlap with VALU / transcendental issue on one SIMD?  cycles per instruction per SIMD for
//   M: 8 independent v_mfma_f32_4x4x4_16b_f16 per iteration          T: 8 independent v_exp_f32 per iteration
//   MT: both interleaved 1:1 (sum of the two = no overlap, max = full overlap)
// and the same with v_mfma_f32_16x16x32_f16 / 32x32x16_f16.  2048 workgroups x 256 threads (8 waves per SIMD) and 256 x 256 (1 wave).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ub_mfma_coissue_probe.hip -o tools/bin/ub_mfma_coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(4))) _Float16 h4;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// KIND 0: 4x4x4 f16, 1: 16x16x32 f16, 2: 32x32x16 f16, 3: 16x16x16 f16 (legacy K), 4: 32x32x8 f16 (legacy K)
template <int KIND, bool DO_M, bool DO_T, int NT>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    const int lane = threadIdx.x;
    h4 a4, b4; h8 a8, b8;
    for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)(0.01f * (lane + i)); b4[i] = (_Float16)(0.02f * (lane - i)); }
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * (lane + i)); b8[i] = (_Float16)(0.02f * (lane - i)); }
    f32x4 c4[8]; f32x16 c16[4];
    for (int g = 0; g < 8; ++g) c4[g] = f32x4{0, 0, 0, 0};
    for (int g = 0; g < 4; ++g) for (int r = 0; r < 16; ++r) c16[g][r] = 0.f;
    float t[8];
    for (int i = 0; i < 8; ++i) t[i] = seed + 0.001f * lane + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if constexpr (DO_M) {
                if constexpr (KIND == 0) c4[g] = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, c4[g], 0, 0, 0);
                else if constexpr (KIND == 1) c4[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c4[g], 0, 0, 0);
                else if constexpr (KIND == 2) { if (g < 4) c16[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c16[g], 0, 0, 0); }
                else if constexpr (KIND == 3) c4[g] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c4[g], 0, 0, 0);
                else { if (g < 4) c16[g] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c16[g], 0, 0, 0); }
            }
            if constexpr (DO_T) {
#pragma unroll
                for (int n = 0; n < NT; ++n) asm volatile("v_exp_f32 %0, %0" : "+v"(t[(g * NT + n) & 7]));
            }
        }
    }
    float s = 0;
    for (int g = 0; g < 8; ++g) for (int r = 0; r < 4; ++r) s += c4[g][r];
    for (int g = 0; g < 4; ++g) for (int r = 0; r < 16; ++r) s += c16[g][r];
    for (int i = 0; i < 8; ++i) s += t[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, bool DO_M, bool DO_T, int NT>
static void run(const char* name, float* d, int blocks, int nm) {
    const int iters = 20000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<KIND, DO_M, DO_T, NT>), dim3(blocks), dim3(256), 0, 0, d, 100, 0.5f);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<KIND, DO_M, DO_T, NT>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double groups = (double)blocks * 4 / 1024 * iters;      // per SIMD: iterations of {nm MFMA, 8 NT exp}
    printf("%-40s blocks %4d  %8.3f ms  %7.1f cycles per iteration per SIMD  (%d MFMA + %d exp)\n", name, blocks, ms,
           ms * 1e-3 * 2.4e9 / groups, DO_M ? nm : 0, DO_T ? 8 * NT : 0);
}

int main() {
    float* d; (void)hipMalloc(&d, 2048 * 256 * 4);
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0, true, false, 1>("4x4x4 f16 only", d, blocks, 8);
        run<0, false, true, 1>("v_exp only (8)", d, blocks, 8);
        run<0, true, true, 1>("4x4x4 + 1 exp each", d, blocks, 8);
        run<0, true, true, 2>("4x4x4 + 2 exp each", d, blocks, 8);
        run<0, true, true, 4>("4x4x4 + 4 exp each", d, blocks, 8);
        run<3, true, false, 1>("16x16x16 f16 only", d, blocks, 8);
        run<3, true, true, 2>("16x16x16 + 2 exp each", d, blocks, 8);
        run<1, true, false, 1>("16x16x32 f16 only", d, blocks, 8);
        run<1, true, true, 2>("16x16x32 + 2 exp each", d, blocks, 8);
        run<2, true, false, 1>("32x32x16 f16 only (4)", d, blocks, 4);
        run<2, true, true, 2>("32x32x16 (4) + 16 exp", d, blocks, 4);
    }
    return 0;
}
