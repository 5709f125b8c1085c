#include <hip/hip_runtime.h>
#include <cstdio>
#define OP8(INS) asm volatile(INS " %0, %0" : "+v"(a0)); asm volatile(INS " %0, %0" : "+v"(a1)); asm volatile(INS " %0, %0" : "+v"(a2)); asm volatile(INS " %0, %0" : "+v"(a3)); \
                 asm volatile(INS " %0, %0" : "+v"(a4)); asm volatile(INS " %0, %0" : "+v"(a5)); asm volatile(INS " %0, %0" : "+v"(a6)); asm volatile(INS " %0, %0" : "+v"(a7));
template <int MODE>
__global__ void k(float* out, int iters) {
  float a0=threadIdx.x*0.001f+0.5f,a1=a0+.1f,a2=a0+.2f,a3=a0+.3f,a4=a0+.4f,a5=a0+.5f,a6=a0+.6f,a7=a0+.7f;
#pragma unroll 4
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { OP8("v_exp_f32") }
    else if (MODE == 1) { OP8("v_rcp_f32") }
    else if (MODE == 2) { OP8("v_exp_f16") }
    else if (MODE == 3) { OP8("v_rcp_f16") }
    else if (MODE == 4) { OP8("v_cvt_f16_f32") }
    else if (MODE == 5) { OP8("v_mov_b32") }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7;
}
template <int MODE> void run(const char* name, float* d) {
  const int blocks = 2048, iters = 20000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100);
  (void)hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double instr_per_simd = (double)blocks * 4 / 1024 * iters * 8;
  printf("%-20s %.3f ms  -> %.2f cycles/instr/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() { float *d; (void)hipMalloc(&d, 2048*256*4);
  run<0>("v_exp_f32", d); run<1>("v_rcp_f32", d); run<2>("v_exp_f16", d); run<3>("v_rcp_f16", d); run<4>("v_cvt_f16_f32", d); run<5>("v_mov_b32", d); return 0; }
