#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
template <int MODE>
__global__ void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  unsigned ub = __float_as_uint(seed) | 0x3f803f80u;
  bf2 bx = __builtin_bit_cast(bf2, ub), by = __builtin_bit_cast(bf2, ub ^ 0x00010001u);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {       // 8 scalar fma
      a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 1.0001f, 0.5f); a2 = fmaf(a2, 1.0001f, 0.5f); a3 = fmaf(a3, 1.0001f, 0.5f);
      a4 = fmaf(a4, 1.0001f, 0.5f); a5 = fmaf(a5, 1.0001f, 0.5f); a6 = fmaf(a6, 1.0001f, 0.5f); a7 = fmaf(a7, 1.0001f, 0.5f);
    } else if (MODE == 1) { // 4 packed fma (8 flops-pairs)
      f2 m = {1.0001f, 1.0001f}, c = {0.5f, 0.5f};
      p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c);
      p2 = __builtin_elementwise_fma(p2, m, c); p3 = __builtin_elementwise_fma(p3, m, c);
    } else if (MODE == 2) { // 8 dot2 bf16
      a0 = __builtin_amdgcn_fdot2_f32_bf16(bx, by, a0, false); a1 = __builtin_amdgcn_fdot2_f32_bf16(bx, by, a1, false);
      a2 = __builtin_amdgcn_fdot2_f32_bf16(bx, by, a2, false); a3 = __builtin_amdgcn_fdot2_f32_bf16(bx, by, a3, false);
      a4 = __builtin_amdgcn_fdot2_f32_bf16(bx, by, a4, false); a5 = __builtin_amdgcn_fdot2_f32_bf16(bx, by, a5, false);
      a6 = __builtin_amdgcn_fdot2_f32_bf16(bx, by, a6, false); a7 = __builtin_amdgcn_fdot2_f32_bf16(bx, by, a7, false);
    } else if (MODE == 3) { // 8 exp2
      a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
      a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
    } else if (MODE == 4) { // 8 shifts (unpack-like)
      unsigned u0=__float_as_uint(a0),u1=__float_as_uint(a1),u2=__float_as_uint(a2),u3=__float_as_uint(a3),u4=__float_as_uint(a4),u5=__float_as_uint(a5),u6=__float_as_uint(a6),u7=__float_as_uint(a7);
      a0=__uint_as_float((u0<<1)^i); a1=__uint_as_float((u1<<1)^i); a2=__uint_as_float((u2<<1)^i); a3=__uint_as_float((u3<<1)^i);
      a4=__uint_as_float((u4<<1)^i); a5=__uint_as_float((u5<<1)^i); a6=__uint_as_float((u6<<1)^i); a7=__uint_as_float((u7<<1)^i);
    }
  }
  out[blockIdx.x*blockDim.x+threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7+p0.x+p0.y+p1.x+p1.y+p2.x+p2.y+p3.x+p3.y;
}
template <int MODE> void run(const char* name, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 256 * 8, threads = 256;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.0f);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: waves = blocks*4 / 1024 SIMDs = 8 waves/SIMD ; instrs per wave = iters * 8 (or 4)
  double instr_per_simd = (double)blocks * 4 / 1024 * iters * (MODE == 1 ? 4 : 8);
  printf("%-12s %.3f ms  -> %.2f cycles/instr/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() { float* d; hipMalloc(&d, 256*8*256*4); run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_dot2_bf16", d); run<3>("v_exp_f32", d); run<4>("shift+xor(2)", d); return 0; }
