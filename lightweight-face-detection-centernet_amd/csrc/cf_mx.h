// Matrix-core depthwise building blocks shared by cf_mbconv3.hip and cf_stem0.hip (see the header comment of cf_mbconv3.hip):
// quad-cell tile E[halo quad][channel][4 x fp16], Toeplitz A operands on v_mfma_f32_4x4x4_16b_f16, the lane -> output-quad
// table that keeps ds_read_b128 conflict-free, compiler-visible bf16 packing for MFMA operands.
#pragma once
#include "cf_common.h"

namespace cf {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 mfma_f16x4;
static constexpr float kNegLog2e3 = -1.44269504088896341f, kNegLn23 = -0.69314718055994531f;

// two fp32 -> packed bf16x2 (RNE) through the compiler's own v_cvt_pk_bf16_f32 selection, NOT the inline-asm pack_bf16x2 of
// cf_common.h: here the packed dwords are MFMA operands a few instructions later, and hipcc pads no VALU-write -> MFMA-read
// wait states for a register written inside an asm statement (seen as stale project operands on ~half the waves)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_v;
__device__ __forceinline__ uint32_t packb(float lo, float hi) {
    f32x2 v; v.x = lo; v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_v));
}
__device__ __forceinline__ u32x4 pack16b(const float* f) {
    u32x4 c; c.x = packb(f[0], f[1]); c.y = packb(f[2], f[3]); c.z = packb(f[4], f[5]); c.w = packb(f[6], f[7]); return c;
}

__device__ __forceinline__ f32x2 swish2_pre(f32x2 u) {      // u = -log2(e) x  ->  u / (1 + 2^u) = -log2(e) swish(x)
    f32x2 e; e.x = __builtin_amdgcn_exp2f(u.x); e.y = __builtin_amdgcn_exp2f(u.y);
    const f32x2 den = e + 1.0f;
    f32x2 r; r.x = __builtin_amdgcn_rcpf(den.x); r.y = __builtin_amdgcn_rcpf(den.y);
    return u * r;
}

static inline uint16_t host_f32_to_f16_3(float f) {       // round-to-nearest-even, saturating (= cf_mbconv2.hip)
    uint32_t u; __builtin_memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int32_t exp = (int32_t)((u >> 23) & 0xff) - 127 + 15;
    uint32_t man = u & 0x7fffffu;
    if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
    if (exp >= 31) return (uint16_t)(sign | 0x7bffu);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - exp;
        uint32_t half = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1))) ++half;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)exp << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half;
    if (half >= 0x7c00u) half = 0x7bffu;
    return (uint16_t)(sign | half);
}

// lane & 15 -> output quad of a set (entry: bit 15 = no quad, bits 6.. = oy, bits 0-5 = x-quad)
template <int TOH, int TOW, int IWQ, int S = 1>
struct SetMap {
    static constexpr int OWQ = TOW / 4, NOQ = TOH * OWQ, NSET = (NOQ + 15) / 16, NSLOT = NSET * 16;
    static_assert(OWQ <= 64 && TOH <= 256, "entry packing");
    uint16_t v[NSLOT];
    constexpr SetMap() : v() {
        int bucket[16][NOQ] = {}; int cnt[16] = {}, used[16] = {};
        for (int r = 0; r < NOQ; ++r) { const int b = (S * (r / OWQ) * IWQ + (r % OWQ)) & 15; bucket[b][cnt[b]++] = r; }     // class of the quad's first INPUT cell (stride 2: rows keep even x-quads first, so cell = x-quad / 2)
        int slot[NSLOT] = {};
        for (int s = 0; s < NSET; ++s)
            for (int q = 0; q < 16; ++q) {
                const int b = ((q >> 2) + 4 * (q & 3)) & 15;                     // slot (pg, j) wants class pg + 4 j
                slot[s * 16 + q] = used[b] < cnt[b] ? bucket[b][used[b]++] : -1;
            }
        for (int b = 0; b < 16; ++b)                                            // uneven classes: leftovers fill the holes
            while (used[b] < cnt[b]) {
                int s = 0; while (slot[s] >= 0) ++s;
                slot[s] = bucket[b][used[b]++];
            }
        for (int s = 0; s < NSLOT; ++s) {
            const int r = slot[s] >= 0 ? slot[s] : 0;
            v[s] = (uint16_t)((slot[s] < 0 ? 0x8000 : 0) | ((r / OWQ) << 6) | (r % OWQ));
        }
    }
};

#define CF_MX_MFMA(ACC, AV, BV, ABID) ACC = __builtin_amdgcn_mfma_f32_4x4x4f16(AV, BV, ACC, 2, ABID, 0)

// depthwise of one set (16 output quads x 32 channels): acc[g][i] = output pixel i of this lane's quad, channel kg*8 + g
template <int KS, int IWQ, int CP>
__device__ __forceinline__ void mx_depthwise(const char* bb, const u32x2 (*A)[KS][2], f32x4* acc) {
    constexpr int NSTEP = KS * 2;
#pragma unroll
    for (int g = 0; g < 8; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    u32x4 bq[2][4];
#pragma unroll
    for (int g2 = 0; g2 < 4; ++g2) bq[0][g2] = ld16(bb + g2 * 16);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        if (st + 1 < NSTEP) {
            const int ky = (st + 1) >> 1, ks = (st + 1) & 1;
#pragma unroll
            for (int g2 = 0; g2 < 4; ++g2) bq[(st + 1) & 1][g2] = ld16(bb + (ky * IWQ + ks) * CP + g2 * 16);
        }
        __builtin_amdgcn_sched_barrier(0);        // keep the next group's reads in flight under this group's MFMAs
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const mfma_f16x4 av = __builtin_bit_cast(mfma_f16x4, A[g >> 2][st >> 1][st & 1]);
            u32x2 b2;
            b2.x = (g & 1) ? bq[st & 1][g >> 1].z : bq[st & 1][g >> 1].x;
            b2.y = (g & 1) ? bq[st & 1][g >> 1].w : bq[st & 1][g >> 1].y;
            const mfma_f16x4 bv = __builtin_bit_cast(mfma_f16x4, b2);
            switch (g & 3) {
                case 0: CF_MX_MFMA(acc[g], av, bv, 0); break;
                case 1: CF_MX_MFMA(acc[g], av, bv, 1); break;
                case 2: CF_MX_MFMA(acc[g], av, bv, 2); break;
                default: CF_MX_MFMA(acc[g], av, bv, 3); break;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the same with the Toeplitz operands read from an LDS copy of the round's table ([2][KS][2][64 lanes] x 8 B) one step ahead
// instead of living in 24 / 40 VGPRs for the whole kernel
template <int KS, int IWQ, int CP>
__device__ __forceinline__ void mx_depthwise_lds(const char* bb, const char* at /* table + lane * 8 */, f32x4* acc) {
    constexpr int NSTEP = KS * 2;
#pragma unroll
    for (int g = 0; g < 8; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    u32x4 bq[2][4];
    u32x2 aq[2][2];
#pragma unroll
    for (int g2 = 0; g2 < 4; ++g2) bq[0][g2] = ld16(bb + g2 * 16);
    aq[0][0] = *reinterpret_cast<const u32x2*>(at);
    aq[0][1] = *reinterpret_cast<const u32x2*>(at + NSTEP * 512);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        if (st + 1 < NSTEP) {
            const int ky = (st + 1) >> 1, ks = (st + 1) & 1;
#pragma unroll
            for (int g2 = 0; g2 < 4; ++g2) bq[(st + 1) & 1][g2] = ld16(bb + (ky * IWQ + ks) * CP + g2 * 16);
            aq[(st + 1) & 1][0] = *reinterpret_cast<const u32x2*>(at + (st + 1) * 512);
            aq[(st + 1) & 1][1] = *reinterpret_cast<const u32x2*>(at + (NSTEP + st + 1) * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const mfma_f16x4 av = __builtin_bit_cast(mfma_f16x4, aq[st & 1][g >> 2]);
            u32x2 b2;
            b2.x = (g & 1) ? bq[st & 1][g >> 1].z : bq[st & 1][g >> 1].x;
            b2.y = (g & 1) ? bq[st & 1][g >> 1].w : bq[st & 1][g >> 1].y;
            const mfma_f16x4 bv = __builtin_bit_cast(mfma_f16x4, b2);
            switch (g & 3) {
                case 0: CF_MX_MFMA(acc[g], av, bv, 0); break;
                case 1: CF_MX_MFMA(acc[g], av, bv, 1); break;
                case 2: CF_MX_MFMA(acc[g], av, bv, 2); break;
                default: CF_MX_MFMA(acc[g], av, bv, 3); break;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// single-buffered form (16 VGPRs less): the reads of a step are issued and consumed in place; for kernels that run four waves per
// SIMD, where the other waves cover the LDS latency
template <int KS, int IWQ, int CP>
__device__ __forceinline__ void mx_depthwise_lds1(const char* bb, const char* at, f32x4* acc) {
    constexpr int NSTEP = KS * 2;
#pragma unroll
    for (int g = 0; g < 8; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        const int ky = st >> 1, ks = st & 1;
        u32x4 bq[4];
#pragma unroll
        for (int g2 = 0; g2 < 4; ++g2) bq[g2] = ld16(bb + (ky * IWQ + ks) * CP + g2 * 16);
        u32x2 aq[2];
        aq[0] = *reinterpret_cast<const u32x2*>(at + st * 512);
        aq[1] = *reinterpret_cast<const u32x2*>(at + (NSTEP + st) * 512);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const mfma_f16x4 av = __builtin_bit_cast(mfma_f16x4, aq[g >> 2]);
            u32x2 b2;
            b2.x = (g & 1) ? bq[g >> 1].z : bq[g >> 1].x;
            b2.y = (g & 1) ? bq[g >> 1].w : bq[g >> 1].y;
            const mfma_f16x4 bv = __builtin_bit_cast(mfma_f16x4, b2);
            switch (g & 3) {
                case 0: CF_MX_MFMA(acc[g], av, bv, 0); break;
                case 1: CF_MX_MFMA(acc[g], av, bv, 1); break;
                case 2: CF_MX_MFMA(acc[g], av, bv, 2); break;
                default: CF_MX_MFMA(acc[g], av, bv, 3); break;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Half a set's channels (g = 4 half .. 4 half + 3: 32 contiguous bytes of a cell row) for the stride-2 blocks, where a wave is
// (set, channel half); KSTEPS k-steps per kernel row (stride 2: inputs x .. x + 10 of an output quad = three quads); the Toeplitz
// operand table in LDS is [2 channel quads][KS][KSTEPS][64 lanes] x 8 B, `at` points at this half's quad + lane * 8
template <int KS, int KSTEPS, int IWQ, int CP>
__device__ __forceinline__ void mx_depthwise_half(const char* bb, const char* at, f32x4* acc) {
    constexpr int NSTEP = KS * KSTEPS;
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    u32x4 bq[2][2];
    u32x2 aq[2];
    bq[0][0] = ld16(bb); bq[0][1] = ld16(bb + 16);
    aq[0] = *reinterpret_cast<const u32x2*>(at);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        if (st + 1 < NSTEP) {
            const int ky = (st + 1) / KSTEPS, ks = (st + 1) % KSTEPS;
            constexpr int HQ = (IWQ + 1) / 2;                      // cell rows: even x-quads, then odd ones
            bq[(st + 1) & 1][0] = ld16(bb + (ky * IWQ + (ks & 1) * HQ + (ks >> 1)) * CP);
            bq[(st + 1) & 1][1] = ld16(bb + (ky * IWQ + (ks & 1) * HQ + (ks >> 1)) * CP + 16);
            aq[(st + 1) & 1] = *reinterpret_cast<const u32x2*>(at + (st + 1) * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
        const mfma_f16x4 av = __builtin_bit_cast(mfma_f16x4, aq[st & 1]);
        u32x2 t0, t1, t2, t3;
        t0.x = bq[st & 1][0].x; t0.y = bq[st & 1][0].y; t1.x = bq[st & 1][0].z; t1.y = bq[st & 1][0].w;
        t2.x = bq[st & 1][1].x; t2.y = bq[st & 1][1].y; t3.x = bq[st & 1][1].z; t3.y = bq[st & 1][1].w;
        CF_MX_MFMA(acc[0], av, __builtin_bit_cast(mfma_f16x4, t0), 0);
        CF_MX_MFMA(acc[1], av, __builtin_bit_cast(mfma_f16x4, t1), 1);
        CF_MX_MFMA(acc[2], av, __builtin_bit_cast(mfma_f16x4, t2), 2);
        CF_MX_MFMA(acc[3], av, __builtin_bit_cast(mfma_f16x4, t3), 3);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the same with this half's Toeplitz operands resident in registers (KS x KSTEPS pairs)
template <int KS, int KSTEPS, int IWQ, int CP>
__device__ __forceinline__ void mx_depthwise_half_reg(const char* bb, const u32x2 (*A)[KSTEPS], f32x4* acc) {
    constexpr int NSTEP = KS * KSTEPS;
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    u32x4 bq[2][2];
    bq[0][0] = ld16(bb); bq[0][1] = ld16(bb + 16);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        if (st + 1 < NSTEP) {
            const int ky = (st + 1) / KSTEPS, ks = (st + 1) % KSTEPS;
            constexpr int HQ = (IWQ + 1) / 2;
            bq[(st + 1) & 1][0] = ld16(bb + (ky * IWQ + (ks & 1) * HQ + (ks >> 1)) * CP);
            bq[(st + 1) & 1][1] = ld16(bb + (ky * IWQ + (ks & 1) * HQ + (ks >> 1)) * CP + 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        const mfma_f16x4 av = __builtin_bit_cast(mfma_f16x4, A[st / KSTEPS][st % KSTEPS]);
        u32x2 t0, t1, t2, t3;
        t0.x = bq[st & 1][0].x; t0.y = bq[st & 1][0].y; t1.x = bq[st & 1][0].z; t1.y = bq[st & 1][0].w;
        t2.x = bq[st & 1][1].x; t2.y = bq[st & 1][1].y; t3.x = bq[st & 1][1].z; t3.y = bq[st & 1][1].w;
        CF_MX_MFMA(acc[0], av, __builtin_bit_cast(mfma_f16x4, t0), 0);
        CF_MX_MFMA(acc[1], av, __builtin_bit_cast(mfma_f16x4, t1), 1);
        CF_MX_MFMA(acc[2], av, __builtin_bit_cast(mfma_f16x4, t2), 2);
        CF_MX_MFMA(acc[3], av, __builtin_bit_cast(mfma_f16x4, t3), 3);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Toeplitz A operands of nq rounds of 32 channels: [round][channel quad q][ky][k-step][lane (kg, pg, i)] = 4 fp16 (cf_mbconv3.hip)
// tap of output offset i against input k of k-step ks: w[ky][4 ks + k - stride * i]
void mx_pack_taps(int nq, int k, const float* wd /*[channels][k*k]*/, uint32_t* out, int stride = 1, int ksteps = 2);

}  // namespace cf
