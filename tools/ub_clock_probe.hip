// VALU issue rates in SHADER cycles (not wall time at a nominal clock), and whether transcendentals overlap plain VALU.
//
// tools/ub_valu_probe.hip reports v_fma_f32 at 2.88 "cycles" per wave64 instruction per SIMD, the
// hardware guide says 2.0 (= the 157 TFLOP/s vector peak).  That probe divides WALL time by a nominal 2.4 GHz.  This one
// also reads s_memtime (shader-clock ticks) and s_memrealtime (100 MHz constant clock) around the loop in every wave, so the
// same run yields (a) cycles per instruction in shader cycles, (b) the shader clock the chip actually sustained while the
// loop ran (DVFS: a dense VALU body clocks below 2.4 GHz), (c) wall time.  Mixed loops (8 v_exp_f32 + n x 8 v_fma_f32,
// independent accumulators, interleaved) answer whether the quarter-rate transcendental pipe runs BESIDE the plain VALU
// or in its issue slots.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ub_clock_probe tools/ub_clock_probe.hip && tools/bin/ub_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define FMA(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c))
#define FMAS(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "s"(ms), "v"(c))
#define FMAC(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(m), "v"(c))
#define FMACS(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "s"(ms), "v"(c))
#define MUL(a) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m))
#define EXP(a) asm volatile("v_exp_f32 %0, %0" : "+v"(a))
#define RCP(a) asm volatile("v_rcp_f32 %0, %0" : "+v"(a))
#define PKFMA(p) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(pm), "v"(pc))
#define PKMUL(p) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(pm))
#define PKFMAS(p) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(pc), "s"(sp))       // taps as an SGPR pair
#define PKFMAS1(p) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p) : "v"(pc), "s"(sp))   // ... one SGPR broadcast to both halves
#define DOT2CV(a) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a) : "v"(hv), "v"(hw))        // VOP2, both operands VGPR
#define DOT2CS(a) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a) : "s"(hs), "v"(hw))        // VOP2, taps in an SGPR (the bf16 kernels' form)
#define DOT2V(a) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a) : "v"(hv), "v"(hw))      // VOP3P
#define FMACD(a, b) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(m))   // neighbour lane's value as src0
#define X8(OP) OP(a0); OP(a1); OP(a2); OP(a3); OP(a4); OP(a5); OP(a6); OP(a7);
#define Y8(OP) OP(b0); OP(b1); OP(b2); OP(b3); OP(b4); OP(b5); OP(b6); OP(b7);
#define Z8(OP) OP(c0); OP(c1); OP(c2); OP(c3); OP(c4); OP(c5); OP(c6); OP(c7);
#define W8(OP) OP(d0); OP(d1); OP(d2); OP(d3); OP(d4); OP(d5); OP(d6); OP(d7);
// 8 transcendentals on a0..a7 interleaved one-for-one (two / three-for-one) with plain ops on other registers
#define I1(T, P) T(a0); P(b0); T(a1); P(b1); T(a2); P(b2); T(a3); P(b3); T(a4); P(b4); T(a5); P(b5); T(a6); P(b6); T(a7); P(b7);
#define I2(T, P) T(a0); P(b0); P(c0); T(a1); P(b1); P(c1); T(a2); P(b2); P(c2); T(a3); P(b3); P(c3); T(a4); P(b4); P(c4); T(a5); P(b5); P(c5); T(a6); P(b6); P(c6); T(a7); P(b7); P(c7);
#define I3(T, P) T(a0); P(b0); P(c0); P(d0); T(a1); P(b1); P(c1); P(d1); T(a2); P(b2); P(c2); P(d2); T(a3); P(b3); P(c3); P(d3); T(a4); P(b4); P(c4); P(d4); T(a5); P(b5); P(c5); P(d5); T(a6); P(b6); P(c6); P(d6); T(a7); P(b7); P(c7); P(d7);

typedef __attribute__((ext_vector_type(2))) float f2;

struct Stamp { unsigned long long t0, t1, r0, r1; };

template <int MODE>
__global__ void k(float* out, Stamp* st, int iters, float ms) {
    float a0 = threadIdx.x * 0.001f + 0.5f, a1 = a0 + .1f, a2 = a0 + .2f, a3 = a0 + .3f, a4 = a0 + .4f, a5 = a0 + .5f, a6 = a0 + .6f, a7 = a0 + .7f;
    float b0 = a0 + 1, b1 = a1 + 1, b2 = a2 + 1, b3 = a3 + 1, b4 = a4 + 1, b5 = a5 + 1, b6 = a6 + 1, b7 = a7 + 1;
    float c0 = a0 + 2, c1 = a1 + 2, c2 = a2 + 2, c3 = a3 + 2, c4 = a4 + 2, c5 = a5 + 2, c6 = a6 + 2, c7 = a7 + 2;
    float d0 = a0 + 3, d1 = a1 + 3, d2 = a2 + 3, d3 = a3 + 3, d4 = a4 + 3, d5 = a5 + 3, d6 = a6 + 3, d7 = a7 + 3;
    float m = 0.9999f + threadIdx.x * 1e-9f, c = 1e-4f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {b0, b1}, p5 = {b2, b3}, p6 = {b4, b5}, p7 = {b6, b7};
    f2 pm = {m, m}, pc = {c, c};
    typedef __attribute__((ext_vector_type(2))) float sf2;
    unsigned long long sp;
    { float lo = ms, hi = ms * 0.5f; unsigned long long t = ((unsigned long long)__float_as_uint(hi) << 32) | __float_as_uint(lo);
      sp = __builtin_amdgcn_readfirstlane((unsigned)t) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(t >> 32)) << 32); }
    unsigned hv = 0x3c003800u + (threadIdx.x & 3), hw = 0x38003c00u + (threadIdx.x & 1);
    const unsigned hs = __builtin_amdgcn_readfirstlane(0x3c003a00u + (unsigned)(ms * 4.0f));
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    // the body is repeated eight times by hand: one loop branch (s_add / s_cmp / s_cbranch) per 64+ instructions
#define R8(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
    for (int i = 0; i < iters; ++i) { R8(
        if (MODE == 0) { X8(FMA) }
        else if (MODE == 1) { X8(FMAC) }
        else if (MODE == 2) { X8(FMAS) }
        else if (MODE == 3) { X8(FMACS) }
        else if (MODE == 4) { X8(EXP) }
        else if (MODE == 5) { X8(RCP) }
        else if (MODE == 6) { PKFMA(p0); PKFMA(p1); PKFMA(p2); PKFMA(p3); PKFMA(p4); PKFMA(p5); PKFMA(p6); PKFMA(p7); }
        else if (MODE == 7) { I1(EXP, FMA) }
        else if (MODE == 8) { I2(EXP, FMA) }
        else if (MODE == 9) { I3(EXP, FMA) }
        else if (MODE == 10) { X8(EXP) Y8(FMA) Z8(FMA) }          // same multiset as 8, blocked instead of interleaved
        else if (MODE == 11) { X8(MUL) }
        else if (MODE == 12) { PKMUL(p0); PKMUL(p1); PKMUL(p2); PKMUL(p3); PKMUL(p4); PKMUL(p5); PKMUL(p6); PKMUL(p7); }
        else if (MODE == 15) { PKFMAS(p0); PKFMAS(p1); PKFMAS(p2); PKFMAS(p3); PKFMAS(p4); PKFMAS(p5); PKFMAS(p6); PKFMAS(p7); }
        else if (MODE == 16) { PKFMAS1(p0); PKFMAS1(p1); PKFMAS1(p2); PKFMAS1(p3); PKFMAS1(p4); PKFMAS1(p5); PKFMAS1(p6); PKFMAS1(p7); }
        else if (MODE == 17) { FMACD(a0, b0); FMACD(a1, b1); FMACD(a2, b2); FMACD(a3, b3); FMACD(a4, b4); FMACD(a5, b5); FMACD(a6, b6); FMACD(a7, b7); }
        else if (MODE == 18) { X8(DOT2CV) }
        else if (MODE == 19) { X8(DOT2CS) }
        else if (MODE == 20) { X8(DOT2V) }
        else if (MODE == 13) { X8(EXP) X8(RCP) Y8(FMA) Z8(FMA) W8(FMA) }     // a Swish-like group, blocked: 16 transcendentals, 24 plain
        else if (MODE == 14) { I1(EXP, FMA) I2(RCP, FMA) }                      // ... the same multiset with every transcendental between plain ops
    ) }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    if ((threadIdx.x & 63) == 0) {
        Stamp s; s.t0 = t0; s.t1 = t1; s.r0 = r0; s.r1 = r1;
        st[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 +
                                                 d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

static float* g_out; static Stamp* g_st;

template <int MODE>
void run(const char* name, int instr_per_iter, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 20000 / waves_per_simd / (instr_per_iter / 8);
    instr_per_iter *= 8;                                          // the hand-unrolled body
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, g_out, g_st, 200, 0.9999f);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, g_out, g_st, iters, 0.9999f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<Stamp> h(blocks * 4);
    (void)hipMemcpy(h.data(), g_st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    // per wave: shader ticks and 100 MHz ticks spent in the loop; the chip-wide figure is the median wave
    std::vector<double> cyc, clk;
    for (const Stamp& s : h) { const double dt = double(s.t1 - s.t0), dr = double(s.r1 - s.r0); cyc.push_back(dt); clk.push_back(dr > 0 ? dt / dr * 100e6 : 0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double wave_cycles = cyc[cyc.size() / 2], ghz = clk[clk.size() / 2] * 1e-9;
    // a wave's loop spans wave_cycles; meanwhile its SIMD issued waves_per_simd x iters x instr_per_iter wave-instructions
    const double per_instr = wave_cycles / ((double)waves_per_simd * iters * instr_per_iter);
    const double wall_nominal = ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * instr_per_iter);
    const double wall_true = ms * 1e-3 * ghz * 1e9 / ((double)waves_per_simd * iters * instr_per_iter);
    printf("%-34s w/SIMD %d  %8.3f ms  shader clock %.3f GHz  cycles per wave-instruction per SIMD: %5.2f (wall time x measured clock)  %5.2f (median wave's own span)  %5.2f (wall time x nominal 2.4 GHz)\n",
           name, waves_per_simd, ms, ghz, wall_true, per_instr, wall_nominal);
}

int main() {
    (void)hipMalloc(&g_out, 2048 * 256 * 4); (void)hipMalloc(&g_st, 2048 * 4 * sizeof(Stamp));
    for (int w : {8, 4, 2, 1}) {
        if (w == 8) { run<0>("v_fma_f32 vop3 vgpr", 8, 8); run<1>("v_fmac_f32 vop2 vgpr", 8, 8); run<2>("v_fma_f32 vop3 sgpr", 8, 8); run<3>("v_fmac_f32 vop2 sgpr", 8, 8);
                      run<11>("v_mul_f32", 8, 8); run<6>("v_pk_fma_f32", 8, 8); run<15>("v_pk_fma_f32 sgpr pair", 8, 8); run<16>("v_pk_fma_f32 one sgpr, op_sel", 8, 8); run<17>("v_fmac_f32_dpp row_shl:1", 8, 8);
                      run<18>("v_dot2c_f32_f16 vop2 vgpr", 8, 8); run<19>("v_dot2c_f32_f16 vop2 sgpr", 8, 8); run<20>("v_dot2_f32_f16 vop3p vgpr", 8, 8); run<12>("v_pk_mul_f32", 8, 8); run<4>("v_exp_f32", 8, 8); run<5>("v_rcp_f32", 8, 8);
                      run<7>("8 exp + 8 fma interleaved", 16, 8); run<8>("8 exp + 16 fma interleaved", 24, 8); run<9>("8 exp + 24 fma interleaved", 32, 8);
                      run<10>("8 exp + 16 fma blocked", 24, 8); run<13>("8 exp 8 rcp 24 fma blocked", 40, 8); run<14>("8 exp 8 rcp 24 fma interleaved", 40, 8); }
        if (w == 4) { run<0>("v_fma_f32 vop3 vgpr", 8, 4); run<4>("v_exp_f32", 8, 4); run<8>("8 exp + 16 fma interleaved", 24, 4); run<10>("8 exp + 16 fma blocked", 24, 4); }
        if (w == 2) { run<0>("v_fma_f32 vop3 vgpr", 8, 2); run<4>("v_exp_f32", 8, 2); run<8>("8 exp + 16 fma interleaved", 24, 2); }
        if (w == 1) { run<0>("v_fma_f32 vop3 vgpr", 8, 1); run<4>("v_exp_f32", 8, 1); run<8>("8 exp + 16 fma interleaved", 24, 1); }
    }
    return 0;
}
