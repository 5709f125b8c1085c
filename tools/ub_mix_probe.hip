#include <hip/hip_runtime.h>
#include <cstdio>
#define MIXLO(acc, e, w, C) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(e), C(w))
#define MIXHI(acc, e, w, C) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(e), C(w))
template <int MODE>
__global__ void k(float* out, int iters, unsigned seed, float ws) {
  float a0=threadIdx.x*0.001f,a1=a0+1,a2=a0+2,a3=a0+3,a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
  unsigned e0 = 0x3c003c00u + (threadIdx.x & 7), e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3;
  float wv = ws + threadIdx.x * 1e-9f;
#pragma unroll 4
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { MIXLO(a0,e0,wv,"v"); MIXHI(a1,e0,wv,"v"); MIXLO(a2,e1,wv,"v"); MIXHI(a3,e1,wv,"v"); MIXLO(a4,e2,wv,"v"); MIXHI(a5,e2,wv,"v"); MIXLO(a6,e3,wv,"v"); MIXHI(a7,e3,wv,"v"); }
    else if (MODE == 1) { MIXLO(a0,e0,ws,"s"); MIXHI(a1,e0,ws,"s"); MIXLO(a2,e1,ws,"s"); MIXHI(a3,e1,ws,"s"); MIXLO(a4,e2,ws,"s"); MIXHI(a5,e2,ws,"s"); MIXLO(a6,e3,ws,"s"); MIXHI(a7,e3,ws,"s"); }
    else if (MODE == 3) {
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(e0), "v"(wv)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(e0), "v"(wv));
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a2) : "v"(e1), "v"(wv)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a3) : "v"(e1), "v"(wv));
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a4) : "v"(e2), "v"(wv)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a5) : "v"(e2), "v"(wv));
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a6) : "v"(e3), "v"(wv)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a7) : "v"(e3), "v"(wv));
    } else if (MODE == 4) {
      asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a0) : "v"(e0), "s"(ws)); asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a1) : "v"(e0), "s"(ws));
      asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a2) : "v"(e1), "s"(ws)); asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a3) : "v"(e1), "s"(ws));
      asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a4) : "v"(e2), "s"(ws)); asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a5) : "v"(e2), "s"(ws));
      asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a6) : "v"(e3), "s"(ws)); asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a7) : "v"(e3), "s"(ws));
    } else if (MODE == 5) {
      asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(a0) : "v"(e0), "s"(ws)); asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(a1) : "v"(e0), "s"(ws));
      asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(a2) : "v"(e1), "s"(ws)); asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(a3) : "v"(e1), "s"(ws));
      asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(a4) : "v"(e2), "s"(ws)); asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(a5) : "v"(e2), "s"(ws));
      asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(a6) : "v"(e3), "s"(ws)); asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(a7) : "v"(e3), "s"(ws));
    } else { // plain v_fma_f32 with sgpr for reference
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(e0), "s"(ws)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(e0), "s"(ws));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(e1), "s"(ws)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(e1), "s"(ws));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a4) : "v"(e2), "s"(ws)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a5) : "v"(e2), "s"(ws));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a6) : "v"(e3), "s"(ws)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a7) : "v"(e3), "s"(ws));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7;
}
template <int MODE> void run(const char* name, float* d) {
  const int blocks = 2048, iters = 20000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 12345u, 0.999f);
  (void)hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u, 0.999f); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double instr_per_simd = (double)blocks * 4 / 1024 * iters * 8;
  printf("%-28s %.3f ms  -> %.2f cycles/instr/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() { float *d; (void)hipMalloc(&d, 2048*256*4);
  run<0>("fma_mix f16 x vgpr", d); run<1>("fma_mix f16 x sgpr", d); run<2>("v_fma_f32 (vop3) x sgpr", d); run<3>("v_fmac_f32 (vop2) vgpr", d); run<4>("v_fmac_f32 (vop2) sgpr", d); run<5>("v_dot2c_f32_bf16 sgpr", d); return 0; }
