// Fused MBConv block, second generation (bf16 storage only): the same expand 1x1 (+Swish) ->
// depthwise k x k (+Swish) -> project 1x1 (+residual) fusion as cf_mbconv.hip
// (MBConvBlock.forward, model/centernet.py:89-140), restructured around what the VALU
// microbenchmarks (profiles/r01_valu_microbench.md) say the depthwise taps cost:
//
//   * the expanded tile E lives in LDS as fp16 PIXEL PAIRS: one dword = (E[y][x even][c], E[y][x+1][c]).
//     The expand GEMM is run as D = X . We^T (lane = hidden channel, registers = 16 pixels, of which
//     rows 2t / 2t+1 are x-neighbours), so a pair is two registers of one lane: Swish, one
//     v_cvt_pkrtz_f16_f32, one ds_write_b32 -- no cross-lane traffic.
//   * a depthwise row of k taps is then ceil((k+1)/2) `v_dot2c_f32_f16` per channel instead of k
//     (unpack + fma): 6 instead of 2x9 VALU instructions for 3x3, 15 instead of 2x25 for 5x5, with fp32
//     accumulation.  The second operand is the tap PAIR (w[2t], w[2t+1]) (or (0,w0),(w1,w2).. for odd
//     x) as fp16 -- wave-uniform, so it comes from SGPRs (s_load from a pre-paired table): no LDS
//     reads and no VGPRs for depthwise weights at all.
//   * wave-uniform tap pairs need every lane of a wave to work on the same hidden channels and the
//     same x parity: a wave owns 64 output pixels (two 32-pixel blocks, same parity), computes the
//     depthwise for channel chunk A (8 channels) and chunk B on all 64 lanes, and one
//     v_permlane32_swap per register turns (A, B) into the two MFMA B-operand fragments of the project
//     GEMM (lanes 0-31 = k-slots 0-7, lanes 32-63 = k-slots 8-15) for pixel block 0 and block 1;
//     both MFMAs share one project-weight fragment.
//   * Swish arguments arrive pre-scaled by -log2(e) (folded into the expand weights; the depthwise is
//     linear, so its output is pre-scaled too) and the leftover factor is folded into the project
//     weights (x -ln 2): x * sigmoid(x) costs exp2 + rcp + one packed add + one packed multiply.
// fp16 has a narrower range than bf16: v_cvt_pkrtz saturates to +-65504 instead of overflowing, and
// E is post-Swish (>= -0.28).  Everything outside the tile E stays bf16 (HBM tensors, project operand).
#include "cf_exp.h"
#include "cf_common.h"
#include "cf_kernels.h"
#include <cstdlib>
#include <type_traits>

namespace cf {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8;
typedef __attribute__((ext_vector_type(2))) __fp16 f16x2;
#define CF_AS4 __attribute__((address_space(4)))

static inline int slot_channel2(int nb, int i) {
    int h = (i >> 2) & 1;
    int r = (i & 3) + 4 * (i >> 3);
    return nb * 32 + h * 16 + r;
}

static inline uint16_t host_f32_to_f16(float f) {       // round-to-nearest-even, saturating
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

// acc += w.lo * e.lo + w.hi * e.hi (fp16 inputs, fp32 accumulate); w is wave-uniform (SGPR)
// (the builtin, not inline asm: DOT results have read-after-write wait states the compiler must see)
typedef __attribute__((ext_vector_type(2))) _Float16 hf2;
__device__ __forceinline__ void dot2c(float& acc, uint32_t w, uint32_t e) {
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(hf2, w), __builtin_bit_cast(hf2, e), acc, false);
}

// Swish with the input pre-scaled by -log2(e) (folded into the weights that produce it):
// u = -log2(e) x  ->  u / (1 + 2^u) = -log2(e) swish(x); the constant factor of the result is folded into
// the NEXT linear layer's weights (x -ln 2).  One packed multiply less per pair than swish2().
__device__ __forceinline__ f32x2 swish2_prescaled(f32x2 u) {
    f32x2 e; e.x = __builtin_amdgcn_exp2f(u.x); e.y = __builtin_amdgcn_exp2f(u.y);
    const f32x2 den = e + 1.0f;
    f32x2 r; r.x = __builtin_amdgcn_rcpf(den.x); r.y = __builtin_amdgcn_rcpf(den.y);
    return u * r;
}
static constexpr float kNegLog2e = -1.44269504088896341f, kNegLn2 = -0.69314718055994531f;

#define CF_DOT8_ACC(A, W, E0, E1)                                                                        \
    dot2c(A[0], W[0], E0.x); dot2c(A[1], W[1], E0.y); dot2c(A[2], W[2], E0.z); dot2c(A[3], W[3], E0.w);  \
    dot2c(A[4], W[4], E1.x); dot2c(A[5], W[5], E1.y); dot2c(A[6], W[6], E1.z); dot2c(A[7], W[7], E1.w)
// first tap through the builtin: acc = w . e + 0 (the compiler zeroes the accumulators with v_mov and uses v_dot2c).
// Measured per kernel under rocprofv3 (B = 64, 640x640): the asm block dot8_first() is 1-3 % faster on the fused 3x3
// kernels and the stem, neutral on the fused 5x5 ones and 3-7 % SLOWER on the expand+depthwise kernels (six waves per
// SIMD: the block is a scheduling barrier in their row pipeline), so only the 3x3 fused path and the stem use it.
#define CF_DOT8_Z(A, W, E0, E1)                                                                          \
    do { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) A[i_] = 0.0f; CF_DOT8_ACC(A, W, E0, E1); } while (0)
#define CF_DOT8(FIRST, A, W, E0, E1)                                                                     \
    do { if constexpr (FIRST) dot8_first(A, W, E0, E1); else { CF_DOT8_ACC(A, W, E0, E1); } } while (0)

#ifndef CF_WDIRECT_COND
#define CF_WDIRECT_COND (KS == 5 && S == 2 && NW == 3)
#endif
// ---------------------------------------------------------------- geometry shared by host and device
template <int KS, int S, int HC, int TOH, int TOW, int JX, int NW>
struct Px {
    static constexpr int IH = (TOH - 1) * S + KS, IW0 = (TOW - 1) * S + KS, IWP = (IW0 + 1) & ~1;
    static constexpr int IPX = IH * IWP, NIB = (IPX + 31) / 32, NPAIR = IPX / 2;
    static constexpr int NT = (KS + 1) / 2, NPARW = S == 1 ? 2 : 1;
    // a wave owns 64 output pixels of one x parity (stride 1) -- the last wave of a parity class may be
    // partly filled (8x40 tile: 160 pixels per parity = 2.5 waves)
    static constexpr int NPIX = TOH * TOW, PPX = S == 1 ? NPIX / 2 : NPIX, WPP = (PPX + 63) / 64;
    static constexpr int NPP = NPARW * WPP, KG = NW / NPP;
    static constexpr int NBE = (HC + 31) / 32, HALF = HC / 16, JS = HALF / KG;
    // a trailing half block of 16 channels (hid = 144 = 3 x 48) runs on two 16x16x32 MFMAs (16 pixels x 16
    // channels each, K = 32 >= Cin) instead of a half-empty 32x32 block: no Swish on dead lanes
    static constexpr bool PART = (HC % 32 == 16);
    static constexpr int NBF = HC / 32;
    static constexpr int PITCH = HC * 4 + 16;
    static constexpr int WXB = (NBF * JX + (PART ? 1 : 0)) * 1024;
    static_assert(!PART || JX <= 2, "half block needs Cin <= 32");
    static constexpr int EBYTES = NIB * 16 * PITCH;          // whole pixel blocks: phase 1 stores are unconditional
    static constexpr int RED = (KG - 1) * NPP * 64 * 64;
    // three-wave workgroups (layer2.0) read the expand weights straight from global memory (L1/L2 hits): without the
    // staging buffers the tile fits four times into a CU, 12 waves = 3 on every SIMD.  With 46 KB three workgroups = 9
    // waves were resident, one SIMD carried three waves against two on the others, and every workgroup ran at the pace of
    // its wave on the crowded SIMD (barrier per phase).
    static constexpr bool WDIRECT = CF_WDIRECT_COND;
    static constexpr int WLDS = WDIRECT ? 0 : 2 * WXB;
    static constexpr int LDS = (EBYTES + WLDS) > RED ? (EBYTES + WLDS) : RED;
    static_assert(NPP * KG == NW && TOW % 2 == 0, "tile / wave geometry");
    static_assert(HC % 16 == 0 && JS * KG == HALF, "hidden chunk / k-group geometry");
};

// ---------------------------------------------------------------- device
#ifndef CF_XRELOAD_COND
#define CF_XRELOAD_COND (KS == 5 && S == 2 && NBO == 1 && NW == 3)
#endif
template <int KS, int S, int NBO, bool RESID, int NW, int JX, int HC, int TOH, int TOW>
__global__ __launch_bounds__(NW * 64) void mbconv_px_kernel(MbParams p) {
    typedef Px<KS, S, HC, TOH, TOW, JX, NW> G;
    constexpr int IWP = G::IWP, IPX = G::IPX, NIB = G::NIB, NPAIR = G::NPAIR, NT = G::NT, NPARW = G::NPARW;
    constexpr int NPP = G::NPP, KG = G::KG, NBF = G::NBF, HALF = G::HALF, JS = G::JS, PITCH = G::PITCH, WXB = G::WXB;
    constexpr bool PART = G::PART;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* E = smem;
    char* Wst = smem + G::EBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    unsigned tbx = blockIdx.x, tby = blockIdx.y, tbz = blockIdx.z;
    if (p.nw) xcd_tile_order(tbx, tby, tbz);        // neighbouring tiles (shared halo rows / cache lines) on one XCD's L2
    const int ox0 = tbx * TOW, oy0 = tby * TOH, b = tbz;
    const int nq = p.nq;
    const int pp = wave % NPP, jg = wave / NPP;

    // wave-uniform x parity (stride 2: x0 = 2 ox is always even) and the wave's 64 pixels of that class
    const int par = S == 1 ? pp / G::WPP : 0;
    static constexpr LaneMap<S, TOH, TOW, IWP> kLanes{};
    const int lm0 = (S == 1 ? pp - par * G::WPP : pp) * 64;
    // pixel `blk * 32 + lane` of this wave -> (oy, ox) through the bank-conflict-free lane map (cf_common.h);
    // false when the lane has no pixel (partly filled last wave of a class)
    auto tile_pixel = [&](int blk, int lane31, int& oy, int& ox) -> bool {
        const uint32_t e = kLanes.v[lm0 + blk * 32 + lane31];
        oy = (e >> 6) & 0x1ff;
        ox = S == 1 ? 2 * (int)(e & 63) + par : (int)(e & 63);
        return (e & 0x8000u) == 0;
    };
    // phase 2: this lane's depthwise pixel = pixel pl of block h of the wave
    int dy, dx; tile_pixel(h, pl, dy, dx);
    const unsigned e_pix = (unsigned)(((dy * S) * IWP + (dx * S - par)) / 2) * (unsigned)PITCH;

    f32x16 acc[2][NBO];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < NBO; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][i][r] = 0.0f;

    const char* xbase = (const char*)p.x + (size_t)b * p.Hin * p.Win * p.Cin * 2;
    const unsigned rowbytes = (unsigned)p.Cin * 2;

    auto stage_weights = [&](int q) {
        char* dst = Wst + (q & 1) * WXB;
        const char* srcx = (const char*)p.wexp + (size_t)q * WXB;
        for (int c = wave; c < WXB / 1024; c += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcx + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
    };

    // X fragments of this wave's halo pixel blocks (MFMA A operand: lane = pixel, 8 contiguous Cin per
    // half), loaded once; clamped address + zero select = ZeroPad2d without predicated loads
    constexpr int MAXI = (NIB + NW - 1) / NW;
    // XRELOAD: the fragments are fetched again (from L2) at the top of every hidden-chunk round instead of living in
    // registers across the rounds -- 64 VGPRs less on the 5x5 stride-2 block (layer2.0: 203 -> occupancy 3)
    constexpr bool XRELOAD = CF_XRELOAD_COND;
    u32x4 xf[MAXI][JX];
    u32x4 xh[PART ? MAXI : 1][2];
    auto load_x = [&]() {
#pragma unroll
    for (int t = 0; t < MAXI; ++t) {
        const int ib = wave + NW * t;
        const int ip = ib * 32 + pl;
        const int ipc = ip < IPX ? ip : IPX - 1;
        const int iy = ipc / IWP, ix = ipc - iy * IWP;
        const int gy = oy0 * S - p.pad_lo + iy, gx = ox0 * S - p.pad_lo + ix;
        const bool valid = ip < IPX && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
        const int cy = min(max(gy, 0), p.Hin - 1), cx = min(max(gx, 0), p.Win - 1);
        const unsigned off = ((unsigned)cy * (unsigned)p.Win + (unsigned)cx) * rowbytes + (unsigned)(h * JX * 16);
#pragma unroll
        for (int j = 0; j < JX; ++j) {
            const u32x4 v = ld16(xbase + off + j * 16);
            xf[t][j].x = valid ? v.x : 0u; xf[t][j].y = valid ? v.y : 0u;
            xf[t][j].z = valid ? v.z : 0u; xf[t][j].w = valid ? v.w : 0u;
        }
    }

    // half block: A operands of the 16x16x32 MFMAs, lane (m = pixel of the 16-pixel sub-block, kg = Cin chunk)
    if constexpr (PART) {
#pragma unroll
        for (int t = 0; t < MAXI; ++t)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int ib = wave + NW * t;
                const int ip = ib * 32 + sub * 16 + (lane & 15), kg = lane >> 4;
                const int ipc = ip < IPX ? ip : IPX - 1;
                const int iy = ipc / IWP, ix = ipc - iy * IWP;
                const int gy = oy0 * S - p.pad_lo + iy, gx = ox0 * S - p.pad_lo + ix;
                const bool valid = ip < IPX && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win && kg * 8 < p.Cin;
                const int cy = min(max(gy, 0), p.Hin - 1), cx = min(max(gx, 0), p.Win - 1);
                const u32x4 v = ld16(xbase + ((unsigned)cy * (unsigned)p.Win + (unsigned)cx) * rowbytes + (unsigned)(min(kg * 8, p.Cin - 8) * 2));
                xh[t][sub].x = valid ? v.x : 0u; xh[t][sub].y = valid ? v.y : 0u;
                xh[t][sub].z = valid ? v.z : 0u; xh[t][sub].w = valid ? v.w : 0u;
            }
    }
    };
    if constexpr (!XRELOAD) load_x();

    // expand one halo pixel block: D[pixel][channel] = X . We^T, Swish, pixel pairs -> E
    // The LAST halo block holds only VP = (IPX - 32 (NIB - 1)) / 2 real pixel pairs (2 of 16 on an 18x18 halo); the Swish and
    // the store of a register pair whose pixels lie past the halo are skipped there (LASTB, compile-time per call site):
    // 5-7 % of the expand-phase VALU work of the 3x3 kernels.
    constexpr int VP = (IPX - 32 * (NIB - 1)) / 2;
    auto expand_block = [&](auto lastb, int ib, const u32x4* xfr, const u32x4* xhr, const char* wx) {
        constexpr bool LASTB = decltype(lastb)::value && VP < 16;
        if constexpr (PART) {
            const u32x4 wv = ld16(wx + (NBF * JX * 64 + lane) * 16);
            char* ecol = E + (NBF * 32 + (lane & 15)) * 4 + (unsigned)(ib * 16 + 2 * (lane >> 4)) * (unsigned)PITCH;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x4 a4 = {0.0f, 0.0f, 0.0f, 0.0f};
                a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, xhr[sub]),
                                                             __builtin_bit_cast(mfma_bf16x8, wv), a4, 0, 0, 0);
                // lane (channel n, row group g): rows 4g .. 4g+3 of the sub-block = pixel pairs 2g, 2g+1
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (LASTB && sub * 8 + t >= VP) continue;     // pairs sub*8 + 2g + t, g = lane >> 4: none of them is real
                    f32x2 x2; x2.x = a4[2 * t]; x2.y = a4[2 * t + 1];
                    const f32x2 y2 = swish2_prescaled(x2);
                    *reinterpret_cast<uint32_t*>(ecol + (sub * 8 + t) * PITCH) =
                        __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y2.x, y2.y));
                }
            }
        }
#pragma unroll
        for (int nbl = 0; nbl < NBF; ++nbl) {
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.0f;
            const char* wb = wx + (nbl * JX * 64 + lane) * 16;
#pragma unroll
            for (int j = 0; j < JX; ++j) {
                const u32x4 wv = ld16(wb + j * 1024);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, xfr[j]),
                                                            __builtin_bit_cast(mfma_bf16x8, wv), a, 0, 0, 0);
            }
            const int ch = nbl * 32 + pl;
            // registers 2t, 2t+1 = MFMA rows m, m+1 (m even): pixel pair (ib*32 + m) / 2 = ib*16 + m2(t)
            char* ecol = E + ch * 4 + (unsigned)(ib * 16 + 2 * h) * (unsigned)PITCH;
            uint32_t d[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (LASTB && (t & 1) + 4 * (t >> 1) >= VP) continue;      // pair (t & 1) + 4 (t >> 1) + 2 h: past the halo on both halves
                f32x2 x2; x2.x = a[2 * t]; x2.y = a[2 * t + 1];
                const f32x2 y2 = swish2_prescaled(x2);            // E' = -log2(e) swish(expand)
                d[t] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y2.x, y2.y));
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (LASTB && (t & 1) + 4 * (t >> 1) >= VP) continue;
                *reinterpret_cast<uint32_t*>(ecol + ((t & 1) + 4 * (t >> 1)) * PITCH) = d[t];
            }
        }
    };

    const CF_AS4 u32x8* wtab = (const CF_AS4 u32x8*)p.wdw;

    // depthwise + Swish for hidden chunk c (8 channels) of chunk group q on this lane's pixel -> bf16x8
    auto dw_chunk = [&](int q, int c) -> u32x4 {
        float a8[8];
        // p.nw is 0 or 1, never negative: the select on a kernel argument only keeps the table address a run-time value.  With a
        // compile-time offset the compiler clusters the tap loads of all chunks of a round in front of the depthwise loop
        // (layer1.0: 33 SGPR spills, 132 -> 177 VGPRs, three -> two waves per SIMD, 0.256 -> 0.318 ms; an opaque inline-asm
        // barrier on the offset does not prevent it).
        const CF_AS4 u32x8* wq = wtab + (p.nw < 0 ? (size_t)0 : (size_t)(((q * NPARW + par) * (HC / 8) + c) * KS) * NT);
        const char* eb = E + e_pix + c * 32;
        if constexpr (KS == 3) {
        // 3x3: all six tap-pair vectors of the chunk (48 SGPRs) in ONE batch.  Tap pairs (s_load) and tile
        // dwords (ds_read) share a wait counter and scalar loads return out of order, so a tile read can only
        // be waited for with "everything outstanding"; with the taps loaded up front only the tile reads are
        // pipelined row by row (layer3.1: 0.073 -> 0.065 ms).  For 5x5 the per-row pipeline below measured faster.
        u32x8 wa[KS * NT];
#pragma unroll
        for (int i = 0; i < KS * NT; ++i) wa[i] = wq[i];
        u32x4 en[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) { en[t][0] = ld16(eb + t * PITCH); en[t][1] = ld16(eb + t * PITCH + 16); }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            u32x4 ec[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) { ec[t][0] = en[t][0]; ec[t][1] = en[t][1]; }
            if (ky + 1 < KS) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const char* et = eb + ((ky + 1) * (IWP / 2) + t) * PITCH;
                    en[t][0] = ld16(et); en[t][1] = ld16(et + 16);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const u32x8 wv = wa[ky * NT + t];
                if (ky == 0 && t == 0) { CF_DOT8(true, a8, wv, ec[t][0], ec[t][1]); }
                else { CF_DOT8(false, a8, wv, ec[t][0], ec[t][1]); }
            }
        }
        } else {
        // software pipeline by one kernel row: row ky+1's tap pairs (SGPRs) and tile dwords (VGPRs) are
        // requested before row ky's 8*NT dot products issue
        u32x8 wn[NT]; u32x4 en[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) { wn[t] = wq[t]; en[t][0] = ld16(eb + t * PITCH); en[t][1] = ld16(eb + t * PITCH + 16); }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            u32x8 wc[NT]; u32x4 ec[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) { wc[t] = wn[t]; ec[t][0] = en[t][0]; ec[t][1] = en[t][1]; }
            if (ky + 1 < KS) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const char* et = eb + ((ky + 1) * (IWP / 2) + t) * PITCH;
                    wn[t] = wq[(ky + 1) * NT + t]; en[t][0] = ld16(et); en[t][1] = ld16(et + 16);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (ky == 0 && t == 0) { CF_DOT8_Z(a8, wc[t], ec[t][0], ec[t][1]); }
                else { CF_DOT8(false, a8, wc[t], ec[t][0], ec[t][1]); }
            }
        }
        }
        // a8 = sum w E' = -log2(e) * depthwise output: already the pre-scaled Swish argument
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            f32x2 u; u.x = a8[i]; u.y = a8[i + 1];
            const f32x2 y = swish2_prescaled(u);
            a8[i] = y.x; a8[i + 1] = y.y;
        }
        return pack16<bf16_t>(a8);
    };

    if constexpr (!G::WDIRECT) stage_weights(0);
    for (int q = 0; q < nq; ++q) {
        const char* wx = G::WDIRECT ? (const char*)p.wexp + (size_t)q * WXB : Wst + (q & 1) * WXB;
        if constexpr (XRELOAD) load_x();
        cf_sync_lds_dma();    // previous chunk's depthwise done with E; this stage's expand weights (LDS-DMA) landed for every wave

#pragma unroll
        for (int t = 0; t < MAXI; ++t) {
            const int ib = wave + NW * t;
            if (t == (NIB - 1) / NW && ib == NIB - 1) expand_block(std::true_type{}, ib, xf[t], xh[PART ? t : 0], wx);   // only this round can hold it
            else if (ib < NIB) expand_block(std::false_type{}, ib, xf[t], xh[PART ? t : 0], wx);
        }
        __syncthreads();
        if (!G::WDIRECT && q + 1 < nq) stage_weights(q + 1);

#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            if (KG > 1 && (j / JS) != jg) continue;                // wave-uniform: k-steps of this k-group
            u32x4 wpc[NBO];
#pragma unroll
            for (int i = 0; i < NBO; ++i)
                wpc[i] = ld16((const char*)p.wproj + ((((size_t)i * nq + q) * HALF + j) * 64 + lane) * 16);
            const u32x4 dA = dw_chunk(q, j), dB = dw_chunk(q, HALF + j);
            // (A, B) -> project fragments: block 0 = {A.lo, B.lo}, block 1 = {A.hi, B.hi}
            u32x4 x0, x1;
            {
                auto s0 = __builtin_amdgcn_permlane32_swap(dA.x, dB.x, false, false); x0.x = s0[0]; x1.x = s0[1];
                auto s1 = __builtin_amdgcn_permlane32_swap(dA.y, dB.y, false, false); x0.y = s1[0]; x1.y = s1[1];
                auto s2 = __builtin_amdgcn_permlane32_swap(dA.z, dB.z, false, false); x0.z = s2[0]; x1.z = s2[1];
                auto s3 = __builtin_amdgcn_permlane32_swap(dA.w, dB.w, false, false); x0.w = s3[0]; x1.w = s3[1];
            }
#pragma unroll
            for (int i = 0; i < NBO; ++i) {
                acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, wpc[i]),
                                                                    __builtin_bit_cast(mfma_bf16x8, x0), acc[0][i], 0, 0, 0);
                acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, wpc[i]),
                                                                    __builtin_bit_cast(mfma_bf16x8, x1), acc[1][i], 0, 0, 0);
            }
        }
    }

    // ---- combine the k-groups through LDS, one accumulator block at a time
    if constexpr (KG > 1) {
        float* red = reinterpret_cast<float*>(smem) + (size_t)(pp * 64 + lane) * 16;
        constexpr int GSTRIDE = NPP * 64 * 16;                         // floats per k-group slab
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < NBO; ++i) {
                __syncthreads();
                if (jg > 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float t[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = acc[k][i][g * 4 + e];
                        st16(red + (jg - 1) * GSTRIDE + g * 4, pack16<float>(t));
                    }
                }
                __syncthreads();
                if (jg == 0) {
#pragma unroll
                    for (int gk = 0; gk < KG - 1; ++gk)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float t[4];
                            unpack16<float>(ld16(red + gk * GSTRIDE + g * 4), t);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[k][i][g * 4 + e] += t[e];
                        }
                }
            }
        if (jg > 0) return;
    }

    // ---- epilogue: accumulator k holds pixel block 2 pp + k (lane = pixel pl, half h = 16 channels)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int oy, ox;
        const bool livepx = tile_pixel(k, pl, oy, ox);
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (!livepx || gy >= p.Hout || gx >= p.Wout) continue;
        const size_t opix = ((size_t)b * p.Hout + gy) * p.Wout + gx;
#pragma unroll
        for (int i = 0; i < NBO; ++i) {
            const int cb = i * 32 + h * 16;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ch = cb + g * 8;
                if (ch >= p.Cout) break;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[k][i][g * 8 + e];
                if constexpr (RESID) {
                    float r[8];
                    unpack16<bf16_t>(ld16((const char*)p.x + (opix * p.Cin + ch) * 2), r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = r[e] + v[e];
                }
                st16((char*)p.y + (p.yblock ? blk_off(opix, p.Cout / 8, ch / 8) : (opix * p.Cout + ch) * 2), pack16<bf16_t>(v));
            }
        }
    }
}


#ifndef CF_ILP_TU        // expand+depthwise kernel, tables and host side: main translation unit only
// ================================================================== expand + depthwise only
// The blocks whose output is too wide for the accumulator budget (layer5.0 .. layer6.0: Cout 160 / 320 on
// 20x20 maps) keep their project 1x1 as a GEMM launch (cf_pw.hip), but expand + Swish + depthwise + Swish
// run here with the same fp16 pixel-pair tile / SGPR tap pairs, writing only the depthwise output:
// the 6x-expanded tensor (the largest of the block) never reaches HBM.  With no project accumulators a
// workgroup handles ONE hidden chunk of one tile, so the grid is (tiles, hid / HC, batch): enough
// workgroups to fill the chip even on 20x20 maps.  Pixel -> lane mapping allows partly filled waves
// (a 10x20 tile has 100 pixels per x parity: two waves of 64 lanes, 50 live in the second).
template <int KS, int S, int HC, int TOH, int TOW, int JX>
struct Xd {
    static constexpr int IH = (TOH - 1) * S + KS, IW0 = (TOW - 1) * S + KS, IWP = (IW0 + 1) & ~1;
    static constexpr int IPX = IH * IWP, NIB = (IPX + 31) / 32;
    static constexpr int NT = (KS + 1) / 2, NPARW = S == 1 ? 2 : 1;
    static constexpr int NPIX = TOH * TOW, PPX = S == 1 ? NPIX / 2 : NPIX;     // pixels per parity class
    static constexpr int WPP = (PPX + 63) / 64, NW = NPARW * WPP;
    static constexpr int NBE = HC / 32, PITCH = HC * 4 + 16, WXB = NBE * JX * 1024;
    static constexpr int EBYTES = NIB * 16 * PITCH, LDS = EBYTES + WXB;
    static_assert(HC % 32 == 0 && TOW % 2 == 0, "hidden chunk / tile geometry");
};

template <int KS, int S, int JX, int HC, int TOH, int TOW>
__global__ __launch_bounds__((Xd<KS, S, HC, TOH, TOW, JX>::NW) * 64) void expdw_px_kernel(MbParams p) {
    typedef Xd<KS, S, HC, TOH, TOW, JX> G;
    constexpr int IWP = G::IWP, IPX = G::IPX, NIB = G::NIB, NT = G::NT, NPARW = G::NPARW, NW = G::NW;
    constexpr int NBE = G::NBE, PITCH = G::PITCH, WXB = G::WXB, PPX = G::PPX, WPP = G::WPP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* E = smem;
    char* Wst = smem + G::EBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const int tiles_x = (p.Wout + TOW - 1) / TOW;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int ox0 = txi * TOW, oy0 = tyi * TOH, b = blockIdx.z;
    const int grp = blockIdx.y;                                      // hidden chunk of this workgroup

    // expand weights of this chunk -> LDS (DMA); lands under the X loads, fenced by the barrier below
    {
        const char* srcx = (const char*)p.wexp + (size_t)grp * WXB;
        for (int c = wave; c < WXB / 1024; c += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcx + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(Wst + c * 1024), 16, 0, 0);
    }
    const char* xbase = (const char*)p.x + (size_t)b * p.Hin * p.Win * p.Cin * 2;
    const unsigned rowbytes = (unsigned)p.Cin * 2;
    // raw loads from a clamped address; the zero-padding select is applied when the fragment is CONSUMED
    // (a select at issue time would make the prefetch wait for its own data)
    auto load_x = [&](int ib, u32x4* xf) -> bool {
        const int ip = ib * 32 + pl;
        const int ipc = ip < IPX ? ip : IPX - 1;
        const int iy = ipc / IWP, ix = ipc - iy * IWP;
        const int gy = oy0 * S - p.pad_lo + iy, gx = ox0 * S - p.pad_lo + ix;
        const int cy = min(max(gy, 0), p.Hin - 1), cx = min(max(gx, 0), p.Win - 1);
        if (p.xblock) {        // pixel-block order: 32 neighbouring halo pixels read one 512-byte run per chunk
            const char* xb = (const char*)p.x + blk_off(((size_t)b * p.Hin + cy) * p.Win + cx, p.Cin / 8, h * JX);
#pragma unroll
            for (int j = 0; j < JX; ++j) xf[j] = ld16(xb + j * 512);
        } else {
            const unsigned off = ((unsigned)cy * (unsigned)p.Win + (unsigned)cx) * rowbytes + (unsigned)(h * JX * 16);
#pragma unroll
            for (int j = 0; j < JX; ++j) xf[j] = ld16(xbase + off + j * 16);
        }
        return ip < IPX && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
    };
    auto mask_x = [&](u32x4* xf, bool valid) {
#pragma unroll
        for (int j = 0; j < JX; ++j) {
            xf[j].x = valid ? xf[j].x : 0u; xf[j].y = valid ? xf[j].y : 0u; xf[j].z = valid ? xf[j].z : 0u; xf[j].w = valid ? xf[j].w : 0u;
        }
    };
    u32x4 xa[JX];
    bool va = false;
    if (wave < NIB) va = load_x(wave, xa);
    cf_sync_lds_dma();        // the expand weights (LDS-DMA) landed for every wave
    mask_x(xa, va);

    // ---- phase 1: expand + Swish -> pixel-pair tile
    for (int ib = wave; ib < NIB; ib += NW) {
        u32x4 xn[JX];
        const bool more = ib + NW < NIB;
        bool vn = false;
        if (more) vn = load_x(ib + NW, xn);                        // next block's X under this block's math
        // (skipping the activation of the padding pairs of the LAST halo block, as the fused kernel does, was measured here:
        // the uniform branch breaks this loop's load/compute overlap and the kernels run 4-10 % slower)
#pragma unroll
        for (int nbl = 0; nbl < NBE; ++nbl) {
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.0f;
            const char* wb = Wst + (nbl * JX * 64 + lane) * 16;
#pragma unroll
            for (int j = 0; j < JX; ++j) {
                const u32x4 wv = ld16(wb + j * 1024);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, xa[j]),
                                                            __builtin_bit_cast(mfma_bf16x8, wv), a, 0, 0, 0);
            }
            char* ecol = E + (nbl * 32 + pl) * 4 + (unsigned)(ib * 16 + 2 * h) * (unsigned)PITCH;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                f32x2 x2; x2.x = a[2 * t]; x2.y = a[2 * t + 1];
                const f32x2 y2 = swish2_prescaled(x2);
                *reinterpret_cast<uint32_t*>(ecol + ((t & 1) + 4 * (t >> 1)) * PITCH) =
                    __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y2.x, y2.y));
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < JX; ++j) xa[j] = xn[j];
            mask_x(xa, vn);
        }
    }
    __syncthreads();

    // ---- phase 2: depthwise + Swish, one output pixel per lane, all HC channels of this chunk
    const int par = S == 1 ? wave / WPP : 0;
    // lane -> pixel of the wave's parity class through the bank-conflict-free lane map (cf_common.h)
    static constexpr LaneMap<S, TOH, TOW, IWP> kLanes{};
    const uint32_t lme = kLanes.v[(S == 1 ? wave - par * WPP : wave) * 64 + lane];
    const bool live = (lme & 0x8000u) == 0;
    const int oy = (lme >> 6) & 0x1ff;
    const int ox = S == 1 ? 2 * (int)(lme & 63) + par : (int)(lme & 63);
    const char* eb0 = E + (unsigned)(((oy * S) * IWP + (ox * S - par)) / 2) * (unsigned)PITCH;
    const int gy = oy0 + oy, gx = ox0 + ox;
    const bool store = live && gy < p.Hout && gx < p.Wout;
    // NHWC rows, or pixel-block order for the project GEMM that follows (PwParams::xblock): chunk stride 16 / 512 bytes
    const size_t opix = ((size_t)b * p.Hout + gy) * p.Wout + gx;
    const size_t ostep = p.yblock ? 512 : 16;
    char* out = p.yblock ? (char*)p.y + ((opix >> 5) * (size_t)(p.hid / 8) * 32 + (opix & 31)) * 16 + (size_t)grp * (HC / 8) * 512
                         : (char*)p.y + (opix * p.hid + (size_t)grp * HC) * 2;
    const CF_AS4 u32x8* wtab = (const CF_AS4 u32x8*)p.wdw;
#pragma unroll
    for (int c = 0; c < HC / 8; ++c) {
        float a8[8];
        const CF_AS4 u32x8* wq = wtab + (size_t)(((grp * NPARW + par) * (HC / 8) + c) * KS) * NT;
        const char* eb = eb0 + c * 32;
        u32x8 wn[NT]; u32x4 en[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) { wn[t] = wq[t]; en[t][0] = ld16(eb + t * PITCH); en[t][1] = ld16(eb + t * PITCH + 16); }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            u32x8 wc[NT]; u32x4 ec[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) { wc[t] = wn[t]; ec[t][0] = en[t][0]; ec[t][1] = en[t][1]; }
            if (ky + 1 < KS) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const char* et = eb + ((ky + 1) * (IWP / 2) + t) * PITCH;
                    wn[t] = wq[(ky + 1) * NT + t]; en[t][0] = ld16(et); en[t][1] = ld16(et + 16);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (ky == 0 && t == 0) { CF_DOT8_Z(a8, wc[t], ec[t][0], ec[t][1]); }
                else { CF_DOT8(false, a8, wc[t], ec[t][0], ec[t][1]); }
            }
        }
        // a8 = -log2(e) * depthwise output -> swish, with the leftover -log2(e) taken out again
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            f32x2 u; u.x = a8[i]; u.y = a8[i + 1];
            const f32x2 y = swish2_prescaled(u) * kNegLn2;
            a8[i] = y.x; a8[i + 1] = y.y;
        }
        if (store) st16(out + c * ostep, pack16<bf16_t>(a8));
    }
}

struct XdEntry {
    int k, s, jx, hc, toh, tow, var, lds_bytes, nt, nparw, nw;
    hipError_t (*fn)(hipStream_t, const MbParams&);
};
template <int KS, int S, int JX, int HC, int TOH, int TOW>
static hipError_t xd_launch_t(hipStream_t s, const MbParams& p) {
    typedef Xd<KS, S, HC, TOH, TOW, JX> G;
    auto kfn = expdw_px_kernel<KS, S, JX, HC, TOH, TOW>;
    static thread_local bool configured_dev[32] = {};               // function attributes are per device
    int dev = 0; (void)hipGetDevice(&dev);
    bool& configured = configured_dev[dev & 31];
    if (G::LDS > 64 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    dim3 grid(((p.Wout + TOW - 1) / TOW) * ((p.Hout + TOH - 1) / TOH), p.hid / HC, p.B), blk(G::NW * 64);
    set_kernel_tag("void cf::expdw_px_kernel<%d, %d, %d, %d, %d, %d>(cf::MbParams)", KS, S, JX, HC, TOH, TOW);
    hipLaunchKernelGGL(kfn, grid, blk, G::LDS, s, p);
    return hipGetLastError();
}
#define XD(V, KS, S, JX, HC, TOH, TOW)                                                                             \
    {KS, S, JX, HC, TOH, TOW, V, Xd<KS, S, HC, TOH, TOW, JX>::LDS, Xd<KS, S, HC, TOH, TOW, JX>::NT, Xd<KS, S, HC, TOH, TOW, JX>::NPARW, \
     Xd<KS, S, HC, TOH, TOW, JX>::NW, &xd_launch_t<KS, S, JX, HC, TOH, TOW>}
static const XdEntry kXdTable[] = {
    //  var KS S JX HC  tile
    XD(0, 5, 2, 6, 32, 10, 20),     // 5.0   96 -> 576, 40x40 -> 20x20
    XD(0, 5, 1, 10, 32, 10, 20),    // 5.1  160 -> 960, 20x20
    XD(0, 3, 1, 10, 32, 10, 20),    // 6.0  160 -> 960, 20x20
    XD(0, 5, 1, 4, 32, 10, 40),     // 4.0   64 -> 384, 40x40: full-width tiles (10x20 0.071, 20x20 0.067, 10x40 0.066, 20x40 0.083 ms)
    XD(0, 5, 1, 6, 32, 10, 40),     // 4.1   96 -> 576, 40x40   (0.106 / 0.098 / 0.096 / 0.129)
#include CF_EXP_INC(cf_mbconv2_0)   // never-default variants: A/B runs of an experiments build only
};
#undef XD
static const XdEntry* xd_find(int k, int s, int jx) {
    static const int want = cf_ab_int("CF_XD_VARIANT", 0);
    const XdEntry* base = nullptr;
    for (const XdEntry& e : kXdTable)
        if (e.k == k && e.s == s && e.jx == jx) {
            if (e.var == want) return &e;
            if (e.var == 0) base = &e;
        }
    return base;
}

// geometry of the expand+depthwise kernel for a block (bf16 storage): MbGeom with kind = 2
MbGeom expdw_geometry(int dtype, int Cin, int hid, int k, int s) {
    if (dtype != 1) return expdw_f32_geometry(dtype, Cin, hid, k, s);   // fp32 storage: cf_mbconv5.hip (kind 8)
    MbGeom g = expdw_mx_geometry(dtype, Cin, hid, k, s);          // stride 1: depthwise on the matrix cores (cf_mbconv3.hip)
    if (g.ok) return g;
    static const bool off = cf_ab_int("CF_XD_KIND", 1) == 0;
    if (off || dtype != 1 || (Cin % 8) || hid == Cin) return g;
    const int jx = (Cin * 2 / 16 + 1) / 2;
    const XdEntry* e = xd_find(k, s, jx);
    if (!e || hid % e->hc) return g;
    g.ok = true; g.kind = 2; g.S = s;
    g.JX = jx; g.NBO = 0; g.HC = e->hc; g.nq = hid / e->hc; g.NBE = e->hc / 32; g.HALF = 0;
    g.rowb = e->hc * 4 + 16;
    g.lds_bytes = (size_t)e->lds_bytes;
    g.wexp_bytes = (size_t)g.nq * g.NBE * g.JX * 64 * 16;
    g.wdw_floats = (size_t)g.nq * e->nparw * (g.HC / 8) * k * e->nt * 8;
    g.wproj_bytes = 0;
    return g;
}
hipError_t expdw_launch(hipStream_t s, const MbParams& p) {
    const XdEntry* e = xd_find(p.k, p.s, p.JX);
    if (!e || e->hc != p.HC) return hipErrorInvalidValue;
    return e->fn(s, p);
}

// ---------------------------------------------------------------- host side
struct Mb2Entry {
    int k, s, jx, hc, nbo, res, toh, tow, nw, var;
    int lds_bytes, nt, nparw, nbe, half, wxb;
    hipError_t (*fn)(hipStream_t, const MbParams&);
};

#endif  // !CF_ILP_TU
template <int KS, int S, int NBO, bool RESID, int NW, int JX, int HC, int TOH, int TOW>
hipError_t mb2_launch_t(hipStream_t s, const MbParams& p) {
    auto kfn = mbconv_px_kernel<KS, S, NBO, RESID, NW, JX, HC, TOH, TOW>;
    constexpr int LDS = Px<KS, S, HC, TOH, TOW, JX, NW>::LDS;
    static thread_local bool configured_dev[32] = {};               // function attributes are per device
    int dev = 0; (void)hipGetDevice(&dev);
    bool& configured = configured_dev[dev & 31];
    if (LDS > 64 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    dim3 grid((p.Wout + TOW - 1) / TOW, (p.Hout + TOH - 1) / TOH, p.B), blk(NW * 64);
    // XCD-aware tile order: measured neutral-to-slower on these VALU-bound kernels (layer1.0 0.234 -> 0.237-0.244 ms; their
    // PMC traffic is already 1.0-1.13x algorithmic), so it stays off here (CF_MB2_XCD=1 switches it on in an experiments build);
    // the stem and the up3+heads kernel take it with CF_XCD_ORDER=1 (it removes their cross-XCD halo re-fetches, not time)
    static const bool xcd_on = cf_ab_int("CF_MB2_XCD", 0) == 1;
    MbParams q = p; q.nw = xcd_on ? 1 : 0;
    set_kernel_tag("void cf::mbconv_px_kernel<%d, %d, %d, %s, %d, %d, %d, %d, %d>(cf::MbParams)", KS, S, NBO, RESID ? "true" : "false", NW, JX, HC, TOH, TOW);
    hipLaunchKernelGGL(kfn, grid, blk, LDS, s, q);
    return hipGetLastError();
}

// Two instances are compiled in their own translation unit (cf_mbconv2_ilp.hip = this file with CF_ILP_TU defined) under
// `-mllvm -amdgpu-sched-strategy=max-ilp`: the ILP-first machine scheduler gains 1-2 % on them (same box, two runs each: layer1.0 0.2460 -> 0.2435 ms,
// layer3.0 0.0563 -> 0.0552; 0.243 -> 0.236 with the whole file under it) but costs the other kernels of this file their occupancy (layer3.1 +6 us,
// layer5.0's expand+dw +33 us: 74 -> 256 VGPRs), and the option is per translation unit.  bench.py: 52.29 -> 52.56 k img/s together with layer2.1.
#define CF_MB2_ILP_INSTANCES(X) \
    X(3, 2, 1, false, 4, 1, 32, 8, 16)  /* layer1.0 */ \
    X(3, 2, 2, false, 4, 2, 32, 8, 16)  /* layer3.0 */
#ifdef CF_ILP_TU
#define CF_X(KS, S, NBO, RES, NW, JX, HC, TOH, TOW) template hipError_t mb2_launch_t<KS, S, NBO, RES, NW, JX, HC, TOH, TOW>(hipStream_t, const MbParams&);
CF_MB2_ILP_INSTANCES(CF_X)
#undef CF_X
#else
#define CF_X(KS, S, NBO, RES, NW, JX, HC, TOH, TOW) extern template hipError_t mb2_launch_t<KS, S, NBO, RES, NW, JX, HC, TOH, TOW>(hipStream_t, const MbParams&);
CF_MB2_ILP_INSTANCES(CF_X)
#undef CF_X
#endif

#ifndef CF_ILP_TU
#define MB2(V, KS, S, JX, HC, NBO, RES, TOH, TOW, NW)                                                               \
    {KS, S, JX, HC, NBO, RES, TOH, TOW, NW, V, Px<KS, S, HC, TOH, TOW, JX, NW>::LDS, Px<KS, S, HC, TOH, TOW, JX, NW>::NT, \
     Px<KS, S, HC, TOH, TOW, JX, NW>::NPARW, Px<KS, S, HC, TOH, TOW, JX, NW>::NBE, Px<KS, S, HC, TOH, TOW, JX, NW>::HALF,  \
     Px<KS, S, HC, TOH, TOW, JX, NW>::WXB, &mb2_launch_t<KS, S, NBO, (RES != 0), NW, JX, HC, TOH, TOW>}
static const Mb2Entry kMb2Table[] = {
    //  var KS S JX HC NBO res  tile  waves
    MB2(0, 3, 2, 1, 32, 1, 0, 8, 16, 4),    // 1.0  16 ->  96 -> 24
    MB2(0, 3, 1, 2, 48, 1, 1, 16, 16, 4),   // 1.1  24 -> 144 -> 24 (+res)
    MB2(0, 5, 2, 2, 48, 1, 0, 8, 8, 3),     // 2.0  24 -> 144 -> 32   (small tile: 48 KB of LDS, 3 workgroups per CU)
    MB2(0, 5, 1, 2, 64, 1, 1, 8, 16, 4),    // 2.1  32 -> 192 -> 32 (+res)
    MB2(0, 3, 2, 2, 32, 2, 0, 8, 16, 4),    // 3.0  32 -> 192 -> 64
    MB2(0, 3, 1, 4, 64, 2, 1, 8, 16, 4),    // 3.1  64 -> 384 -> 64 (+res)
    MB2(0, 5, 1, 4, 64, 3, 0, 8, 16, 4),    // 4.0  64 -> 384 -> 96
    MB2(0, 5, 1, 6, 64, 3, 1, 8, 16, 4),    // 4.1  96 -> 576 -> 96 (+res)
#include CF_EXP_INC(cf_mbconv2_1)   // never-default variants: A/B runs of an experiments build only
};
#undef MB2

static const Mb2Entry* mb2_find(int k, int s, int jx, int nbo, int res) {
    static const int want = cf_ab_int("CF_MB2_VARIANT", 0);
    const Mb2Entry* base = nullptr;
    for (const Mb2Entry& e : kMb2Table)
        if (e.k == k && e.s == s && e.jx == jx && e.nbo == nbo && e.res == res) {
            if (e.var == want) return &e;
            if (e.var == 0) base = &e;
        }
    return base;
}

bool mb2_geometry(MbGeom& g, int Cin, int hid, int Cout, int k, int s) {
    static const bool off = cf_ab_int("CF_MB_KIND", 1) == 0;
    if (off) return false;
    const int jx = (Cin * 2 / 16 + 1) / 2, nbo = (Cout + 31) / 32;
    const Mb2Entry* e = mb2_find(k, s, jx, nbo, (Cin == Cout && s == 1) ? 1 : 0);
    if (!e || hid % e->hc) return false;
    g.ok = true; g.kind = 1; g.S = s;
    g.JX = jx; g.NBO = nbo; g.HC = e->hc; g.nq = hid / e->hc; g.NBE = e->nbe; g.HALF = e->half;
    g.rowb = e->hc * 4 + 16;
    g.lds_bytes = (size_t)e->lds_bytes;
    g.wexp_bytes = (size_t)g.nq * e->wxb;
    g.wdw_floats = (size_t)g.nq * e->nparw * (g.HC / 8) * k * e->nt * 8;    // dwords (fp16 tap pairs)
    g.wproj_bytes = (size_t)g.NBO * g.nq * g.HALF * 64 * 16;
    return true;
}

// we [hid][Cin], wd [hid][k*k], wp [Cout][hid]
void mb2_pack_weights(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
                      void* wexp_host, float* wdw_host, void* wproj_host) {
    const int NCx = Cin * 2 / 16, NT = (k + 1) / 2, NPARW = g.S == 1 ? 2 : 1;
    __builtin_memset(wexp_host, 0, g.wexp_bytes);
    if (wproj_host) __builtin_memset(wproj_host, 0, g.wproj_bytes);
    uint32_t* wt = reinterpret_cast<uint32_t*>(wdw_host);
    for (int q = 0; q < g.nq; ++q) {
        // expand, MFMA B operand: lane (n = channel, half h) holds Cin chunk h*JX + j of hidden channel q*HC + nbl*32 + n;
        // a trailing 16-channel half block (kind 1 only) is one 16x16x32 fragment: lane (n = channel, kg = Cin chunk)
        const bool part = g.kind == 1 && (g.HC % 32 == 16);
        const int nbf = part ? g.HC / 32 : g.NBE;
        const size_t wxb = ((size_t)nbf * g.JX + (part ? 1 : 0)) * 1024;
        for (int nbl = 0; nbl < nbf; ++nbl)
            for (int j = 0; j < g.JX; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = lane & 31, hh = lane >> 5;
                    const int cl = nbl * 32 + n, c = hh * g.JX + j;
                    if (cl >= g.HC || c >= NCx) continue;
                    uint16_t* dst = (uint16_t*)((char*)wexp_host + (size_t)q * wxb + (((size_t)nbl * g.JX + j) * 64 + lane) * 16);
                    for (int e = 0; e < 8; ++e) dst[e] = host_f32_to_bf16(kNegLog2e * we[(size_t)(q * g.HC + cl) * Cin + (size_t)c * 8 + e]);
                }
        if (part)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 15, kg = lane >> 4;
                if (kg >= NCx) continue;
                uint16_t* dst = (uint16_t*)((char*)wexp_host + (size_t)q * wxb + ((size_t)nbf * g.JX * 64 + lane) * 16);
                for (int e = 0; e < 8; ++e) dst[e] = host_f32_to_bf16(kNegLog2e * we[(size_t)(q * g.HC + nbf * 32 + n) * Cin + (size_t)kg * 8 + e]);
            }
        // depthwise tap pairs: [q][parity][chunk][ky][t][8 channels]; even x0: (w[2t], w[2t+1]), odd x0: (w[2t-1], w[2t])
        for (int par = 0; par < NPARW; ++par)
            for (int c = 0; c < g.HC / 8; ++c)
                for (int ky = 0; ky < k; ++ky)
                    for (int t = 0; t < NT; ++t)
                        for (int i = 0; i < 8; ++i) {
                            const float* wrow = wd + (size_t)(q * g.HC + c * 8 + i) * k * k + (size_t)ky * k;
                            const int k0 = 2 * t - par, k1 = k0 + 1;
                            const uint16_t lo = (k0 >= 0 && k0 < k) ? host_f32_to_f16(wrow[k0]) : 0;
                            const uint16_t hi = (k1 >= 0 && k1 < k) ? host_f32_to_f16(wrow[k1]) : 0;
                            wt[((((size_t)(q * NPARW + par) * (g.HC / 8) + c) * k + ky) * NT + t) * 8 + i] = (uint32_t)lo | ((uint32_t)hi << 16);
                        }
        // project, MFMA A operand (same layout as cf_mbconv.hip)
        for (int nbo = 0; wproj_host && nbo < g.NBO; ++nbo)
            for (int j = 0; j < g.HALF; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, hh = lane >> 5;
                    const int co = slot_channel2(nbo, i);
                    if (co >= Cout) continue;
                    const int hc = q * g.HC + (hh * g.HALF + j) * 8;
                    uint16_t* dst = (uint16_t*)((char*)wproj_host + ((((size_t)nbo * g.nq + q) * g.HALF + j) * 64 + lane) * 16);
                    for (int e = 0; e < 8; ++e) dst[e] = host_f32_to_bf16(kNegLn2 * wp[(size_t)co * hid + hc + e]);
                }
    }
}

hipError_t mb2_launch(hipStream_t s, const MbParams& p) {
    const Mb2Entry* e = mb2_find(p.k, p.s, p.JX, (p.Cout + 31) / 32, p.residual ? 1 : 0);
    if (!e || e->hc != p.HC) return hipErrorInvalidValue;
    return e->fn(s, p);
}

#endif  // !CF_ILP_TU
}  // namespace cf
