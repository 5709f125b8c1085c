// Probe for "depthwise taps on the matrix cores" (VERDICT r02 next-1b): v_mfma_f32_4x4x4_16B_f16 with per-channel
// Toeplitz A operands against the v_dot2c path of cf_mbconv2.hip.
//   part A: lane layout of the 4x4x4 MFMA (A/B/D) and the CBSZ/ABID broadcast, checked against a host model
//   part B: cycles per (64 pixels x 32 channels) depthwise + Swish + project step, dot2c path vs MFMA path
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ub_dwmfma_probe.hip -o tools/bin/ub_dwmfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(4))) _Float16 h4;
typedef __attribute__((ext_vector_type(2))) _Float16 hf2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ------------------------------------------------------------------ part A
template <int CBSZ, int ABID>
__global__ void layout_kernel(const _Float16* a, const _Float16* b, float* d) {
    const int l = threadIdx.x;
    h4 av, bv;
    for (int k = 0; k < 4; ++k) { av[k] = a[l * 4 + k]; bv[k] = b[l * 4 + k]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, c, CBSZ, ABID, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

template <int CBSZ, int ABID> static int check_layout() {
    std::vector<_Float16> a(256), b(256);
    // assumed: lane l = 4 blk + i holds A_blk[i][k = 0..3]; lane l = 4 blk + j holds B_blk[k = 0..3][j]
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
            a[l * 4 + k] = (_Float16)(float)((l * 7 + k * 3) % 11 - 5);
            b[l * 4 + k] = (_Float16)(float)((l * 5 + k * 13) % 9 - 4);
        }
    _Float16 *da, *db; float* dd;
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 1024);
    (void)hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((layout_kernel<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<float> d(256);
    (void)hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 16; ++blk) {
        const int ablk = CBSZ ? ((blk >> CBSZ) << CBSZ) + ABID : blk;      // A broadcast inside groups of 2^CBSZ blocks
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float ref = 0;
                for (int k = 0; k < 4; ++k) ref += (float)a[(ablk * 4 + i) * 4 + k] * (float)b[(blk * 4 + j) * 4 + k];
                const float got = d[(blk * 4 + j) * 4 + i];               // assumed: lane 4 blk + j, register i
                if (got != ref) ++bad;
            }
    }
    printf("layout cbsz=%d abid=%d: %s (%d mismatches of 256)\n", CBSZ, ABID, bad ? "MISMATCH" : "ok", bad);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dd);
    return bad;
}

// ------------------------------------------------------------------ part B
__device__ __forceinline__ f32x2 swish2_prescaled(f32x2 u) {
    f32x2 e; e.x = __builtin_amdgcn_exp2f(u.x); e.y = __builtin_amdgcn_exp2f(u.y);
    const f32x2 den = e + 1.0f;
    f32x2 r; r.x = __builtin_amdgcn_rcpf(den.x); r.y = __builtin_amdgcn_rcpf(den.y);
    return u * r;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    uint32_t r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r;
}
__device__ __forceinline__ void dot2c(float& acc, uint32_t w, uint32_t e) {
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(hf2, w), __builtin_bit_cast(hf2, e), acc, false);
}


// one 4x4x4 MFMA, then its share of the step's 64 transcendentals / 48 other VALU, a DS read every other MFMA
template <int n, int NM> __device__ __forceinline__ void sched_interleave() {
    if constexpr (n < NM) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr ((n & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        constexpr int nt = (64 * (n + 1)) / NM - (64 * n) / NM, nv = (48 * (n + 1)) / NM - (48 * n) / NM;
        if constexpr (nt > 0) __builtin_amdgcn_sched_group_barrier(0x400, nt, 0);
        if constexpr (nv > 0) __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);
        sched_interleave<n + 1, NM>();
    }
}

// MODE 0: dot2c path (cf_mbconv2.hip dw_chunk): per 8-channel chunk KS rows x NT tap pairs x 2 ds_read_b128 + 8 KS NT dot2c
//         (SGPR weights), 8 Swish, pack; 4 chunks -> 2 x (permlane swap x4 + 2 MFMA 32x32x16)
// MODE 1: MFMA path: 8 channel-quad groups x KS x KSTEPS {ds_read_b64 + mfma 4x4x4 (cbsz 2)}, 32 Swish, 16 pack, 4 x NM MFMA 16x16x32
// MODE 2: Swish + pack + project only (the floor both share)
template <int MODE, int KS, int KSTEPS, int NT>
__global__ __launch_bounds__(256) void step_kernel(const uint32_t* __restrict__ wtab, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[40 * 1024];
    for (int i = threadIdx.x; i < 10 * 1024; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = 0x2c002c00u + ((i * 37) & 0x3ff);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 pacc[4][2];
    f32x16 qacc[2];
    for (int i = 0; i < 4; ++i) for (int m = 0; m < 2; ++m) pacc[i][m] = f32x4{0, 0, 0, 0};
    for (int k = 0; k < 2; ++k) for (int r = 0; r < 16; ++r) qacc[k][r] = 0.f;
    const u32x4 wp = *reinterpret_cast<const u32x4*>(wtab + 1024 + lane * 4);
    unsigned off = 0;
    if constexpr (MODE == 3) {
        // MODE 3 = MODE 1 software-pipelined across steps: the 4x4x4 MFMAs of step `it` are interleaved (sched_group_barrier)
        // with the Swish / pack / project of step it-1, so the matrix pipe runs under the transcendental issue of the SAME wave
        u32x2 A[2][KS][KSTEPS];
        for (int a = 0; a < 2; ++a) for (int ky = 0; ky < KS; ++ky) for (int ks = 0; ks < KSTEPS; ++ks)
            A[a][ky][ks] = *reinterpret_cast<const u32x2*>(wtab + ((a * KS + ky) * KSTEPS + ks) * 128 + lane * 2);
        const unsigned lb128 = (lane & 3) * 1088 + ((lane >> 2) & 3) * 272 + (lane >> 4) * 64;
        constexpr int NSTEP = KS * KSTEPS;
        f32x4 ping[8], pong[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) pong[g] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
        auto body = [&](f32x4* acc, f32x4* prev) {
#pragma unroll
            for (int g = 0; g < 8; ++g) acc[g] = f32x4{0, 0, 0, 0};
            u32x4 bq[2][4];
#pragma unroll
            for (int g2 = 0; g2 < 4; ++g2) bq[0][g2] = *reinterpret_cast<const u32x4*>(lds + off + lb128 + g2 * 16);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + 1 < NSTEP) {
#pragma unroll
                    for (int g2 = 0; g2 < 4; ++g2) bq[(st + 1) & 1][g2] = *reinterpret_cast<const u32x4*>(lds + off + lb128 + (st + 1) * 1088 + g2 * 16);
                }
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const h4 av = __builtin_bit_cast(h4, A[g >> 2][st / KSTEPS][st % KSTEPS]);
                    u32x2 b2; b2.x = (g & 1) ? bq[st & 1][g >> 1].z : bq[st & 1][g >> 1].x; b2.y = (g & 1) ? bq[st & 1][g >> 1].w : bq[st & 1][g >> 1].y;
                    const h4 bv = __builtin_bit_cast(h4, b2);
                    switch (g & 3) {
                        case 0: acc[g] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc[g], 2, 0, 0); break;
                        case 1: acc[g] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc[g], 2, 1, 0); break;
                        case 2: acc[g] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc[g], 2, 2, 0); break;
                        default: acc[g] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc[g], 2, 3, 0); break;
                    }
                }
            }
            // Swish / pack / project of the PREVIOUS step's results
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    f32x2 u; u.x = prev[g][i]; u.y = prev[g][i + 1];
                    const f32x2 y = swish2_prescaled(u);
                    prev[g][i] = y.x; prev[g][i + 1] = y.y;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4 d;
#pragma unroll
                for (int p = 0; p < 4; ++p) d[p] = pack_bf16x2(prev[2 * p][i], prev[2 * p + 1][i]);
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    pacc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, wp), __builtin_bit_cast(bf8, d), pacc[i][m], 0, 0, 0);
            }
            // interleave: per 4x4x4 MFMA about 64 / (8 NSTEP) transcendentals and as many other VALU; a DS read every other MFMA
            sched_interleave<0, 8 * NSTEP>();
            off = (off + 16) & 63;
        };
        for (int it = 0; it < iters; it += 2) { body(ping, pong); body(pong, ping); }
    } else
    if constexpr (MODE == 1) {
        // A operands: [2 group-quads][KS][KSTEPS] x 2 VGPRs, resident
        u32x2 A[2][KS][KSTEPS];
        for (int a = 0; a < 2; ++a) for (int ky = 0; ky < KS; ++ky) for (int ks = 0; ks < KSTEPS; ++ks)
            A[a][ky][ks] = *reinterpret_cast<const u32x2*>(wtab + ((a * KS + ky) * KSTEPS + ks) * 128 + lane * 2);
        const unsigned lbase = (lane & 31) * 8 + (lane >> 5) * 264;       // conflict-free b64 reads per 32-lane half
        const unsigned lb128 = (lane & 3) * 1088 * 16 / 16 + ((lane >> 2) & 3) * 272 + (lane >> 4) * 64;   // row j, quad pg, channel group kg: conflict-free b128
        for (int it = 0; it < iters; ++it) {
            f32x4 acc[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) acc[g] = f32x4{0, 0, 0, 0};
            // B operands: a lane's 8 channels are 64 contiguous bytes ([row][quad][channel] cells of 8 bytes) -> 4 ds_read_b128 per
            // (ky, kstep) group, double-buffered one group ahead
            constexpr int NSTEP = KS * KSTEPS;
            u32x4 bq[2][4];
#pragma unroll
            for (int g2 = 0; g2 < 4; ++g2) bq[0][g2] = *reinterpret_cast<const u32x4*>(lds + off + lb128 + g2 * 16);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + 1 < NSTEP) {
#pragma unroll
                    for (int g2 = 0; g2 < 4; ++g2) bq[(st + 1) & 1][g2] = *reinterpret_cast<const u32x4*>(lds + off + lb128 + (st + 1) * 1088 + g2 * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const h4 av = __builtin_bit_cast(h4, A[g >> 2][st / KSTEPS][st % KSTEPS]);
                    u32x2 b2; b2.x = (g & 1) ? bq[st & 1][g >> 1].z : bq[st & 1][g >> 1].x; b2.y = (g & 1) ? bq[st & 1][g >> 1].w : bq[st & 1][g >> 1].y;
                    const h4 bv = __builtin_bit_cast(h4, b2);
                    switch (g & 3) {
                        case 0: acc[g] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc[g], 2, 0, 0); break;
                        case 1: acc[g] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc[g], 2, 1, 0); break;
                        case 2: acc[g] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc[g], 2, 2, 0); break;
                        default: acc[g] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc[g], 2, 3, 0); break;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // Swish on the 32 results in place (register pairs of one MFMA result: no moves), then pack channel pairs
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    f32x2 u; u.x = acc[g][i]; u.y = acc[g][i + 1];
                    const f32x2 y = swish2_prescaled(u);
                    acc[g][i] = y.x; acc[g][i + 1] = y.y;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4 d;
#pragma unroll
                for (int p = 0; p < 4; ++p) d[p] = pack_bf16x2(acc[2 * p][i], acc[2 * p + 1][i]);
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    pacc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, wp), __builtin_bit_cast(bf8, d), pacc[i][m], 0, 0, 0);
            }
            off = (off + 16) & 63;
        }
    } else {
        const unsigned lbase = lane * 16;
        for (int it = 0; it < iters; ++it) {
            u32x4 dch[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) a8[i] = (float)it;
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const unsigned o = off + lbase + (((c & 1) * KS + ky) * NT + t) * 1040;
                            const u32x4 e0 = *reinterpret_cast<const u32x4*>(lds + o);
                            const u32x4 e1 = *reinterpret_cast<const u32x4*>(lds + o + 1024 + 16);
                            typedef __attribute__((ext_vector_type(8))) uint32_t u32x8;
                            const u32x8 w = ((const __attribute__((address_space(4))) u32x8*)wtab)[((c * KS + ky) * NT + t) + (it & 7) * 64];
                            dot2c(a8[0], w[0], e0.x); dot2c(a8[1], w[1], e0.y); dot2c(a8[2], w[2], e0.z); dot2c(a8[3], w[3], e0.w);
                            dot2c(a8[4], w[4], e1.x); dot2c(a8[5], w[5], e1.y); dot2c(a8[6], w[6], e1.z); dot2c(a8[7], w[7], e1.w);
                        }
                } else {
                    const u32x4 e0 = *reinterpret_cast<const u32x4*>(lds + off + lbase + c * 2064);
                    a8[0] += __uint_as_float(e0.x); a8[3] += __uint_as_float(e0.y);
                }
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    f32x2 u; u.x = a8[i]; u.y = a8[i + 1];
                    const f32x2 y = swish2_prescaled(u);
                    a8[i] = y.x; a8[i + 1] = y.y;
                }
                dch[c].x = pack_bf16x2(a8[0], a8[1]); dch[c].y = pack_bf16x2(a8[2], a8[3]);
                dch[c].z = pack_bf16x2(a8[4], a8[5]); dch[c].w = pack_bf16x2(a8[6], a8[7]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 dA = dch[j], dB = dch[2 + j];
                u32x4 x0, x1;
                auto s0 = __builtin_amdgcn_permlane32_swap(dA.x, dB.x, false, false); x0.x = s0[0]; x1.x = s0[1];
                auto s1 = __builtin_amdgcn_permlane32_swap(dA.y, dB.y, false, false); x0.y = s1[0]; x1.y = s1[1];
                auto s2 = __builtin_amdgcn_permlane32_swap(dA.z, dB.z, false, false); x0.z = s2[0]; x1.z = s2[1];
                auto s3 = __builtin_amdgcn_permlane32_swap(dA.w, dB.w, false, false); x0.w = s3[0]; x1.w = s3[1];
                qacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, wp), __builtin_bit_cast(bf8, x0), qacc[0], 0, 0, 0);
                qacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, wp), __builtin_bit_cast(bf8, x1), qacc[1], 0, 0, 0);
            }
            off = (off + 16) & 63;
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int m = 0; m < 2; ++m) for (int r = 0; r < 4; ++r) s += pacc[i][m][r];
    for (int k = 0; k < 2; ++k) for (int r = 0; r < 16; ++r) s += qacc[k][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int KS, int KSTEPS, int NT>
static double run_step(const char* name, const uint32_t* wtab, float* out, int blocks_per_cu) {
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((step_kernel<MODE, KS, KSTEPS, NT>), dim3(blocks), dim3(256), 0, 0, wtab, out, 20);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((step_kernel<MODE, KS, KSTEPS, NT>), dim3(blocks), dim3(256), 0, 0, wtab, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    // wave-steps per SIMD = blocks * 4 waves / 1024 SIMDs * iters
    const double steps_per_simd = (double)blocks * 4 / 1024 * iters;
    const double cyc = ms * 1e-3 * 2.4e9 / steps_per_simd;
    printf("%-44s waves/SIMD %d  %.3f ms  %.0f cycles per (64 px x 32 ch) step per SIMD\n", name, blocks_per_cu, ms, cyc);
    return cyc;
}

int main() {
    int bad = 0;
    bad += check_layout<0, 0>();
    bad += check_layout<2, 0>(); bad += check_layout<2, 1>(); bad += check_layout<2, 2>(); bad += check_layout<2, 3>();
    bad += check_layout<1, 1>();
    uint32_t* wtab; float* out;
    (void)hipMalloc(&wtab, 64 * 1024); (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    std::vector<uint32_t> h(16 * 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x2e002e00u + (uint32_t)((i * 2654435761u) & 0x1ff);
    (void)hipMemcpy(wtab, h.data(), 64 * 1024, hipMemcpyHostToDevice);
    for (int w = 1; w <= 4; w *= 2) {
        run_step<2, 3, 2, 2>("floor: Swish + pack + project only", wtab, out, w);
        run_step<0, 3, 2, 2>("3x3 s1 dot2c (6 per channel)", wtab, out, w);
        run_step<1, 3, 2, 2>("3x3 s1 mfma 4x4x4 (48 per step)", wtab, out, w);
        run_step<3, 3, 2, 2>("3x3 s1 mfma 4x4x4, pipelined+interleaved", wtab, out, w);
        run_step<0, 5, 2, 3>("5x5 s1 dot2c (15 per channel)", wtab, out, w);
        run_step<1, 5, 2, 3>("5x5 s1 mfma 4x4x4 (80 per step)", wtab, out, w);
        run_step<3, 5, 2, 3>("5x5 s1 mfma 4x4x4, pipelined+interleaved", wtab, out, w);
        run_step<1, 3, 3, 2>("3x3 s2 mfma 4x4x4 (72 per step)", wtab, out, w);
        run_step<1, 5, 3, 3>("5x5 s2 mfma 4x4x4 (120 per step)", wtab, out, w);
        run_step<3, 5, 3, 3>("5x5 s2 mfma 4x4x4, pipelined+interleaved", wtab, out, w);
    }
    return bad != 0;
}
