// Issue cost of the PACKED fp16 VALU instructions on gfx950 -- the question behind "a packed-fp16 polynomial Swish
// (v_pk_fma_f16: 128 elements per instruction) instead of two quarter-rate transcendentals per element".
// Same method as tools/ub_mix_probe.hip: 2048 workgroups x 256 threads, 8 independent accumulators, inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP3(name, a, b, c) asm volatile(name " %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c))
#define OP2(name, a, b) asm volatile(name " %0, %0, %1" : "+v"(a) : "v"(b))
#define OP1(name, a) asm volatile(name " %0, %0" : "+v"(a))
template <int MODE>
__global__ void k(unsigned* out, int iters) {
  unsigned a0 = 0x3c003800u + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned b = 0x38003a00u + (threadIdx.x & 3), c = 0x34003400u;
#pragma unroll 4
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { OP3("v_pk_fma_f16", a0, b, c); OP3("v_pk_fma_f16", a1, b, c); OP3("v_pk_fma_f16", a2, b, c); OP3("v_pk_fma_f16", a3, b, c);
                     OP3("v_pk_fma_f16", a4, b, c); OP3("v_pk_fma_f16", a5, b, c); OP3("v_pk_fma_f16", a6, b, c); OP3("v_pk_fma_f16", a7, b, c); }
    else if (MODE == 1) { OP2("v_pk_mul_f16", a0, b); OP2("v_pk_mul_f16", a1, b); OP2("v_pk_mul_f16", a2, b); OP2("v_pk_mul_f16", a3, b);
                          OP2("v_pk_mul_f16", a4, b); OP2("v_pk_mul_f16", a5, b); OP2("v_pk_mul_f16", a6, b); OP2("v_pk_mul_f16", a7, b); }
    else if (MODE == 2) { OP2("v_pk_add_f16", a0, b); OP2("v_pk_add_f16", a1, b); OP2("v_pk_add_f16", a2, b); OP2("v_pk_add_f16", a3, b);
                          OP2("v_pk_add_f16", a4, b); OP2("v_pk_add_f16", a5, b); OP2("v_pk_add_f16", a6, b); OP2("v_pk_add_f16", a7, b); }
    else if (MODE == 3) { OP2("v_pk_max_f16", a0, b); OP2("v_pk_max_f16", a1, b); OP2("v_pk_max_f16", a2, b); OP2("v_pk_max_f16", a3, b);
                          OP2("v_pk_max_f16", a4, b); OP2("v_pk_max_f16", a5, b); OP2("v_pk_max_f16", a6, b); OP2("v_pk_max_f16", a7, b); }
    else if (MODE == 4) { OP1("v_exp_f16_e32", a0); OP1("v_exp_f16_e32", a1); OP1("v_exp_f16_e32", a2); OP1("v_exp_f16_e32", a3);
                          OP1("v_exp_f16_e32", a4); OP1("v_exp_f16_e32", a5); OP1("v_exp_f16_e32", a6); OP1("v_exp_f16_e32", a7); }
    else if (MODE == 5) { OP1("v_rcp_f16_e32", a0); OP1("v_rcp_f16_e32", a1); OP1("v_rcp_f16_e32", a2); OP1("v_rcp_f16_e32", a3);
                          OP1("v_rcp_f16_e32", a4); OP1("v_rcp_f16_e32", a5); OP1("v_rcp_f16_e32", a6); OP1("v_rcp_f16_e32", a7); }
    else if (MODE == 6) { OP3("v_fma_f16", a0, b, c); OP3("v_fma_f16", a1, b, c); OP3("v_fma_f16", a2, b, c); OP3("v_fma_f16", a3, b, c);
                          OP3("v_fma_f16", a4, b, c); OP3("v_fma_f16", a5, b, c); OP3("v_fma_f16", a6, b, c); OP3("v_fma_f16", a7, b, c); }
    else { OP1("v_exp_f32_e32", a0); OP1("v_exp_f32_e32", a1); OP1("v_exp_f32_e32", a2); OP1("v_exp_f32_e32", a3);
           OP1("v_exp_f32_e32", a4); OP1("v_exp_f32_e32", a5); OP1("v_exp_f32_e32", a6); OP1("v_exp_f32_e32", a7); }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int MODE> void run(const char* name, unsigned* d) {
  const int blocks = 2048, iters = 20000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100);
  (void)hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double instr_per_simd = (double)blocks * 4 / 1024 * iters * 8;
  printf("| `%s` | %.2f |\n", name, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() { unsigned* d; (void)hipMalloc(&d, 2048 * 256 * 4);
  printf("| instruction | cycles / wave-instr / SIMD @2.4 GHz |\n|---|---:|\n");
  run<0>("v_pk_fma_f16", d); run<1>("v_pk_mul_f16", d); run<2>("v_pk_add_f16", d); run<3>("v_pk_max_f16", d); run<6>("v_fma_f16 (VOP3, one element per lane)", d);
  run<4>("v_exp_f16", d); run<5>("v_rcp_f16", d); run<7>("v_exp_f32", d); return 0; }
