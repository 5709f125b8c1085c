// Fused MBConv block for the fp32 PARITY MODE, second generation (MbGeom::kind = 7): expand 1x1 (+Swish) -> depthwise k x k
// (+Swish) -> project 1x1 (+residual), MBConvBlock.forward (model/centernet.py:89-140), everything in fp32 on the exact
// v_mfma_f32_32x32x2_f32 -- the only mode that meets north_star's 1e-3 against the reference.
//
// cf_mbconv.hip's fp32 instance maps a lane to (pixel, half of the hidden chunk): its depthwise reads the tile AND the taps from
// LDS (two ds_read_b128 per tap per four channels) and spends most of its time on the LDS pipe.  Here, as in the bf16 kernels of
// cf_mbconv2.hip:
//   * a wave owns 64 output pixels and walks the hidden chunk four channels at a time, so the taps are WAVE-UNIFORM and come
//     from SGPRs (s_load_dwordx4 from a [chunk][group][tap][4] table): one ds_read_b128 + four v_fmac_f32 (SGPR operand) per
//     tap, no LDS or VGPRs for depthwise weights;
//   * two channel groups A and B are computed on all 64 lanes and one v_permlane32_swap per register turns them into the two
//     B-operand fragments of the project MFMA (lanes 0-31 = k-slot 0 = a channel of A, lanes 32-63 = k-slot 1 = a channel of
//     B) for pixel block 0 and pixel block 1 of the wave;
//   * stride 2: a tile row keeps its even input columns first, then the odd ones, so the 16 lanes of a ds_read_b128 group walk
//     consecutive rows of the tile (one 16-byte bank slot apart) instead of every second one (two-way conflict).
// Exact mode: no weight folding and the same swish2() as the other fp32 kernels (its arithmetic is what the goldens pin);
// split mode (SP): Swish factors folded into the expand / project weights like cf_mbconv.hip (swish2_sel<true>).
#include "cf_exp.h"
#include "cf_common.h"
#include "cf_kernels.h"
#include <cstdlib>
#include <type_traits>

namespace cf {

#define CF_AS4 __attribute__((address_space(4)))

template <int KS, int S, int HC, int TOH, int TOW, int JX, int NW>
struct F4 {
    static constexpr int IH = (TOH - 1) * S + KS, IW0 = (TOW - 1) * S + KS;
    static constexpr int HW = (IW0 + 1) / 2, IW = S == 2 ? 2 * HW : IW0;          // stride 2: [even columns | odd columns]
    static constexpr int IPX = IH * IW, NIB = (IPX + 31) / 32;
    static constexpr int NPIX = TOH * TOW, NPW = NPIX / 64, KG = NW / NPW;
    static constexpr int NBE = (HC + 31) / 32, NPAIR = HC / 8, JS = NPAIR / KG;
    static constexpr bool PART = (HC % 32 == 16);
    static constexpr int ROWB = HC * 4 + 16;
    static constexpr int WXB = NBE * JX * 1024;
    static constexpr int EBYTES = (IPX * ROWB + 15) / 16 * 16;
    static constexpr int RED = (KG - 1) * NPW * 64 * 64;                          // one n-block of partial sums per extra k-group
    static constexpr int LDS = (EBYTES + 2 * WXB) > RED ? (EBYTES + 2 * WXB) : RED;
    static_assert(NPIX % 64 == 0 && NPW * KG == NW && TOW % 16 == 0, "tile / wave geometry");
    static_assert(HC % 8 == 0 && JS * KG == NPAIR, "hidden chunk / k-group geometry");
};

// SP: GEMM products as split-bf16 (CfMma<sp32_t>, dtype 2) instead of the exact fp32 MFMA; everything else identical
template <int KS, int S, int NBO, bool RESID, int NW, int JX, int HC, int TOH, int TOW, bool XRELOAD, bool SP = false>
__global__ __launch_bounds__(NW * 64) void mbconv_f32_kernel(MbParams p) {
    typedef CfMma<typename std::conditional<SP, sp32_t, float>::type> MMA;
    typedef F4<KS, S, HC, TOH, TOW, JX, NW> G;
    constexpr int IW = G::IW, HW = G::HW, IPX = G::IPX, NIB = G::NIB, NPW = G::NPW, KG = G::KG, NBE = G::NBE, JS = G::JS;
    constexpr int ROWB = G::ROWB, WXB = G::WXB;
    constexpr bool PART = G::PART;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* E = smem;
    char* Wst = smem + G::EBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const int ox0 = blockIdx.x * TOW, oy0 = blockIdx.y * TOH, b = blockIdx.z;
    const int nq = p.nq;

    // phase 2: this wave's 64 output pixels and its share of the chunk's channel pairs
    const int pw = wave % NPW, kg = wave / NPW;
    const int o = pw * 64 + (lane & 32) + lds_group_pixel(lane & 31), oy = o / TOW, ox = o % TOW;      // conflict-free ds_read_b128 groups (cf_common.h)
    const unsigned e_pix = (unsigned)((oy * S) * IW + ox) * (unsigned)ROWB;       // stride 2: column 2 ox = position ox of the even half

    f32x16 acc[2][NBO];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int i = 0; i < NBO; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[blk][i][r] = 0.0f;

    const char* xbase = (const char*)p.x + (size_t)b * p.Hin * p.Win * p.Cin * 4;
    const unsigned rowbytes = (unsigned)p.Cin * 4;

    auto stage_weights = [&](int q) {
        char* dst = Wst + (q & 1) * WXB;
        const char* srcx = (const char*)p.wexp + (size_t)q * WXB;
        for (int c = wave; c < WXB / 1024; c += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcx + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
    };

    // this wave's tile pixel blocks -> X fragments (zero outside the image = ZeroPad2d); resident, or re-read per chunk when
    // Cin is wide (JX >= 8: 2 x 12 x 4 registers would not leave room for the accumulators)
    constexpr int MAXI = (NIB + NW - 1) / NW;
    u32x4 xf[XRELOAD ? 1 : MAXI][JX];
    auto load_block = [&](int ib, u32x4* dst) {
        const int ip = ib * 32 + pl;
        const int ipc = ip < IPX ? ip : IPX - 1;
        const int iy = ipc / IW, xp = ipc - iy * IW;
        const int ix = S == 2 ? (xp < HW ? 2 * xp : 2 * (xp - HW) + 1) : xp;
        const int gy = oy0 * S - p.pad_lo + iy, gx = ox0 * S - p.pad_lo + ix;
        const bool valid = ip < IPX && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
        const int cy = min(max(gy, 0), p.Hin - 1), cx = min(max(gx, 0), p.Win - 1);
        const unsigned off = ((unsigned)cy * (unsigned)p.Win + (unsigned)cx) * rowbytes + (unsigned)(h * JX * 16);
#pragma unroll
        for (int j = 0; j < JX; ++j) {
            const u32x4 v = ld16(xbase + off + j * 16);
            dst[j].x = valid ? v.x : 0u; dst[j].y = valid ? v.y : 0u;
            dst[j].z = valid ? v.z : 0u; dst[j].w = valid ? v.w : 0u;
        }
    };
    if constexpr (!XRELOAD) {
#pragma unroll
        for (int t = 0; t < MAXI; ++t) load_block(wave + NW * t, xf[t]);
        // split mode (round 5): the resident fragments are split into bf16 (hi, lo) ONCE, here, instead of once per hidden chunk inside
        // the MFMA chain (a chunk pair -> [8 x hi], [8 x lo]; an odd last chunk -> [4 x hi | 4 x lo]): same products, same order
        if constexpr (SP) {
#pragma unroll
            for (int t = 0; t < MAXI; ++t) {
#pragma unroll
                for (int j = 0; j + 1 < JX; j += 2) { const SplitPair s2 = split8(xf[t][j], xf[t][j + 1]); xf[t][j] = s2.hi; xf[t][j + 1] = s2.lo; }
                if constexpr (JX & 1) { u32x2 xh, xl; split4(xf[t][JX - 1], xh, xl); xf[t][JX - 1].x = xh.x; xf[t][JX - 1].y = xh.y; xf[t][JX - 1].z = xl.x; xf[t][JX - 1].w = xl.y; }
            }
        }
    }

    auto expand_block = [&](int ib, const u32x4* xfr, const char* wx) {
        const int ip = ib * 32 + pl;
        const bool ipok = ip < IPX;
        char* erow = E + (unsigned)(ipok ? ip : 0) * (unsigned)ROWB;
#pragma unroll
        for (int nbl = 0; nbl < NBE; ++nbl) {
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.0f;
            const char* wb = wx + (nbl * JX * 64 + lane) * 16;
            if constexpr (SP && !XRELOAD) {                    // fragments already split (above)
#pragma unroll
                for (int j = 0; j + 1 < JX; j += 2) {
                    const u32x4 whi = ld16(wb + j * 1024), wlo = ld16(wb + (j + 1) * 1024);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, wlo), __builtin_bit_cast(cf_bf16x8, xfr[j]), a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, whi), __builtin_bit_cast(cf_bf16x8, xfr[j + 1]), a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, whi), __builtin_bit_cast(cf_bf16x8, xfr[j]), a, 0, 0, 0);
                }
                if constexpr (JX & 1) {
                    u32x2 xh, xl; xh.x = xfr[JX - 1].x; xh.y = xfr[JX - 1].y; xl.x = xfr[JX - 1].z; xl.y = xfr[JX - 1].w;
                    mma_split_parts(a, ld16(wb + (JX - 1) * 1024), xh, xl);
                }
            } else
            mma_chain<typename std::conditional<SP, sp32_t, float>::type, JX>(a, [&](int j) { return ld16(wb + j * 1024); }, [&](int j) { return xfr[j]; });
            const bool half_block = PART && nbl == NBE - 1;       // 8 channels on each lane half (mb_pack_weights)
            const int ch0 = half_block ? nbl * 32 + h * 8 : nbl * 32 + h * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (half_block && g >= 2) break;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    f32x2 x2; x2.x = a[g * 4 + e]; x2.y = a[g * 4 + e + 1];
                    const f32x2 y2 = swish2_sel<SP>(x2);
                    v[e] = y2.x; v[e + 1] = y2.y;
                }
                if (ipok) st16(erow + (ch0 + g * 4) * 4, pack16<float>(v));
            }
        }
    };

    const CF_AS4 f32x4* wtab = (const CF_AS4 f32x4*)p.wdw;                       // [chunk][group of 4 channels][tap]
#ifdef CF_X5_TIMING      // phase stamps (s_memtime), summed over the chunks of a wave: tools/x5_timing.py
    unsigned long long tph[4] = {0, 0, 0, 0}, tq = 0;
#define M4_STAMP(k) { unsigned long long t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tph[k] += t_ - tq; tq = t_; }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tq) :: "memory");
#else
#define M4_STAMP(k)
#endif
    stage_weights(0);
    for (int q = 0; q < nq; ++q) {
        const char* wx = Wst + (q & 1) * WXB;
        cf_sync_lds_dma();            // previous chunk's depthwise is done with E; this chunk's expand weights landed
        M4_STAMP(0)

        // ---- phase 1: expand + Swish -> E
#pragma unroll
        for (int t = 0; t < MAXI; ++t) {
            const int ib = wave + NW * t;
            if (ib >= NIB) break;
            if constexpr (XRELOAD) { load_block(ib, xf[0]); expand_block(ib, xf[0], wx); }
            else expand_block(ib, xf[t], wx);
        }
        M4_STAMP(1)
        __syncthreads();
        M4_STAMP(2)
        if (q + 1 < nq) stage_weights(q + 1);

        // ---- phase 2 + 3: depthwise + Swish on channel groups A, B; swap halves; project MFMAs of both pixel blocks
        // one step = the pair of channel groups 2 jp, 2 jp + 1 -> the operand chunks x0 (pixel block 0) and x1 (pixel block 1)
        auto dw_step = [&](int jp, u32x4& x0, u32x4& x1) {
            float d[2][4];
#pragma unroll
            for (int ab = 0; ab < 2; ++ab) {
                const int g = 2 * jp + ab;
                const CF_AS4 f32x4* wq = wtab + ((size_t)q * (HC / 4) + g) * (KS * KS);
                const char* eb = E + e_pix + g * 16;
                float a4[4];
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const int xo = S == 2 ? (kx & 1) * HW + (kx >> 1) : kx;
                        float ev[4];
                        unpack16<float>(ld16(eb + (ky * IW + xo) * ROWB), ev);
                        const f32x4 w = wq[ky * KS + kx];
#pragma unroll
                        for (int c = 0; c < 4; ++c) a4[c] = (ky == 0 && kx == 0) ? w[c] * ev[c] : __builtin_fmaf(w[c], ev[c], a4[c]);
                    }
#pragma unroll
                for (int c = 0; c < 4; c += 2) {
                    f32x2 x2; x2.x = a4[c]; x2.y = a4[c + 1];
                    const f32x2 y2 = swish2_sel<SP>(x2);
                    d[ab][c] = y2.x; d[ab][c + 1] = y2.y;
                }
            }
            auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[0][0]), __float_as_uint(d[1][0]), false, false); x0.x = s0[0]; x1.x = s0[1];
            auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[0][1]), __float_as_uint(d[1][1]), false, false); x0.y = s1[0]; x1.y = s1[1];
            auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[0][2]), __float_as_uint(d[1][2]), false, false); x0.z = s2[0]; x1.z = s2[1];
            auto s3 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[0][3]), __float_as_uint(d[1][3]), false, false); x0.w = s3[0]; x1.w = s3[1];
        };
        auto wproj_at = [&](int i, int jp) { return ld16((const char*)p.wproj + ((((size_t)i * nq + q) * G::NPAIR + jp) * 64 + lane) * 16); };
        if constexpr (SP) {
            // split mode: the JS steps of this k-group in pairs (one k = 16 MFMA set per pair and pixel block), an odd last step alone
#pragma unroll
            for (int js = 0; js + 1 < JS; js += 2) {
                const int jp = kg * JS + js;
                u32x4 w0[NBO], w1[NBO];
#pragma unroll
                for (int i = 0; i < NBO; ++i) { w0[i] = wproj_at(i, jp); w1[i] = wproj_at(i, jp + 1); }
                u32x4 xa0, xa1, xb0, xb1;
                dw_step(jp, xa0, xa1);
                dw_step(jp + 1, xb0, xb1);
#pragma unroll
                for (int i = 0; i < NBO; ++i) { MMA::run2(acc[0][i], w0[i], w1[i], xa0, xb0); MMA::run2(acc[1][i], w0[i], w1[i], xa1, xb1); }
            }
            if constexpr (JS & 1) {
                const int jp = kg * JS + JS - 1;
                u32x4 w0[NBO];
#pragma unroll
                for (int i = 0; i < NBO; ++i) w0[i] = wproj_at(i, jp);
                u32x4 x0, x1;
                dw_step(jp, x0, x1);
#pragma unroll
                for (int i = 0; i < NBO; ++i) { MMA::run(acc[0][i], w0[i], x0); MMA::run(acc[1][i], w0[i], x1); }
            }
        } else {
#pragma unroll
        for (int js = 0; js < JS; ++js) {
            const int jp = kg * JS + js;                                        // pair of groups 2 jp, 2 jp + 1 (wave-uniform)
            u32x4 wpc[NBO];
#pragma unroll
            for (int i = 0; i < NBO; ++i) wpc[i] = wproj_at(i, jp);
            u32x4 x0, x1;
            dw_step(jp, x0, x1);
#pragma unroll
            for (int i = 0; i < NBO; ++i) { MMA::run(acc[0][i], wpc[i], x0); MMA::run(acc[1][i], wpc[i], x1); }
        }
        }
        M4_STAMP(3)
    }

#ifdef CF_X5_TIMING
    if (p.dbg && lane == 0) {
        unsigned long long* o = (unsigned long long*)p.dbg + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + wave) * 4;
        o[0] = tph[0]; o[1] = tph[1]; o[2] = tph[2]; o[3] = tph[3];
    }
#endif
    // ---- combine the k-groups through LDS (one output n-block of one pixel block at a time), in k-group order
    if constexpr (KG > 1) {
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < NBO; ++i) {
                __syncthreads();                              // everyone is done with E / the previous buffer
                if (kg > 0) {
                    float* dst = red + ((size_t)((kg - 1) * NPW + pw) * 64 + lane) * 16;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float t[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = acc[blk][i][g * 4 + e];
                        st16(dst + g * 4, pack16<float>(t));
                    }
                }
                __syncthreads();
                if (kg == 0) {
#pragma unroll
                    for (int k2 = 1; k2 < KG; ++k2) {
                        const float* src = red + ((size_t)((k2 - 1) * NPW + pw) * 64 + lane) * 16;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float t[4];
                            unpack16<float>(ld16(src + g * 4), t);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[blk][i][g * 4 + e] += t[e];
                        }
                    }
                }
            }
        if (kg > 0) return;
    }

    // ---- epilogue: lane (pl, h) holds 16 contiguous output channels of pixel pl of each of the wave's two pixel blocks
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int o2 = pw * 64 + blk * 32 + lds_group_pixel(pl), oy2 = o2 / TOW, ox2 = o2 % TOW;       // the pixel lane pl of block blk computed
        const int gy = oy0 + oy2, gx = ox0 + ox2;
        if (gy >= p.Hout || gx >= p.Wout) continue;
        const size_t opix = ((size_t)b * p.Hout + gy) * p.Wout + gx;
#pragma unroll
        for (int i = 0; i < NBO; ++i) {
            const int cb = i * 32 + h * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = cb + g * 4;
                if (ch >= p.Cout) break;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[blk][i][g * 4 + e];
                if constexpr (RESID) {
                    float r[4];
                    unpack16<float>(ld16((const char*)p.x + (opix * p.Cin + ch) * 4), r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = r[e] + v[e];
                }
                st16((char*)p.y + (opix * p.Cout + ch) * 4, pack16<float>(v));
            }
        }
    }
}

// ---------------------------------------------------------------- host side
struct F4Entry {
    int k, s, jx, hc, nbo, res, var, lds_bytes, kg;  // var 0: default (both modes), 1: CF_F4_VARIANT=1, 2: default in split mode only
    hipError_t (*fn)(hipStream_t, const MbParams&);
    hipError_t (*fn_sp)(hipStream_t, const MbParams&);      // split-bf16 products (dtype 2)
};
template <int KS, int S, int NBO, bool RESID, int NW, int JX, int HC, int TOH, int TOW, bool SP, bool XR>
static hipError_t f4_launch_t(hipStream_t s, const MbParams& p) {
    typedef F4<KS, S, HC, TOH, TOW, JX, NW> G;
    auto kfn = mbconv_f32_kernel<KS, S, NBO, RESID, NW, JX, HC, TOH, TOW, XR, SP>;
    static thread_local bool configured_dev[32] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    bool& configured = configured_dev[dev & 31];
    if (G::LDS > 64 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    dim3 grid((p.Wout + TOW - 1) / TOW, (p.Hout + TOH - 1) / TOH, p.B), blk(NW * 64);
    set_kernel_tag("void cf::mbconv_f32_kernel<%d, %d, %d, %s, %d, %d, %d, %d, %d, %s, %s>(cf::MbParams)", KS, S, NBO, RESID ? "true" : "false", NW, JX, HC, TOH, TOW,
                   XR ? "true" : "false", SP ? "true" : "false");
    hipLaunchKernelGGL(kfn, grid, blk, G::LDS, s, p);
    return hipGetLastError();
}
#define F4X(V, KS, S, JX, HC, NBO, RES, TOH, TOW, NW, XR) \
    {KS, S, JX, HC, NBO, RES, V, F4<KS, S, HC, TOH, TOW, JX, NW>::LDS, F4<KS, S, HC, TOH, TOW, JX, NW>::KG, &f4_launch_t<KS, S, NBO, (RES != 0), NW, JX, HC, TOH, TOW, false, (XR != 0)>, \
     &f4_launch_t<KS, S, NBO, (RES != 0), NW, JX, HC, TOH, TOW, true, (XR != 0)>}
#define F4E(V, KS, S, JX, HC, NBO, RES, TOH, TOW, NW) F4X(V, KS, S, JX, HC, NBO, RES, TOH, TOW, NW, (JX >= 8))
static const F4Entry kF4Table[] = {
    // Measured against cf_mbconv.hip's fp32 instances (B = 64, 640x640, ms, HIP events of tools/profile_ops.py; this kernel / that one):
    //   1.0  0.706-0.786 / 0.565    1.1  0.492 / 0.497    2.0  0.471-0.485 / 0.439    2.1  0.259 / 0.325
    //   3.0  0.152-0.169 / 0.150    3.1  0.221 / 0.192    4.0  0.335-0.356 / 0.277    4.1  0.631-0.652 / 0.505
    // (four or eight waves on an 8x16 tile; 2- and 1-wave workgroups and 4x16 tiles: 20-60 % slower still).  Halving the LDS reads
    // and removing the tap reads pays only where the depthwise dominates (5x5 at Cout = 32); elsewhere the fp32 blocks are bound
    // by the SUM of their MFMA, VALU and LDS time (a 16-pass fp32 MFMA keeps the SIMD's issue port most of its duration), and the
    // extra accumulators (two pixel blocks per wave: 32 NBO registers more) cost occupancy.  Only layer2.1 runs here.
    //  var KS S JX HC NBO res tile   waves
    F4E(0, 5, 1, 4, 32, 1, 1, 8, 16, 4),     // 2.1  32 -> 192 -> 32
    // split mode (dtype 2: the MFMA share of the block is 2.7x smaller, the LDS share decides): layer1.1 0.371 -> 0.337 ms here
    F4E(2, 3, 1, 3, 48, 1, 1, 8, 16, 4),     // 1.1  24 -> 144 -> 24
    // split-mode sweep (tools/split_sweep.sh, profiles/r04_split_sweep.txt): 4x16 stride-2 tiles with four k-groups for 1.0
    // (0.438 on cf_mbconv.hip -> 0.414), eight waves / HC 64 for 2.1 (0.2175 -> 0.2056)
    F4X(2, 3, 2, 2, 32, 1, 0, 4, 16, 4, 0),  // 1.0  16 ->  96 -> 24
    // second sweep, after the conflict-free lane map (profiles/r04_split_sweep2.txt): 2.0 here on an 8x16 tile with HC 16 (the [even |
    // odd] column layout has no stride-2 bank conflicts: 0.297 on cf_mbconv.hip -> 0.261), 2.1 with HC 32 on eight waves (0.191 -> 0.179)
    F4X(2, 5, 2, 3, 16, 1, 0, 8, 16, 4, 0),  // 2.0  24 -> 144 -> 32
    F4X(2, 5, 1, 4, 32, 1, 1, 8, 16, 8, 0),  // 2.1  32 -> 192 -> 32
    // CF_F4_VARIANT=1: every block shape on this kernel (A/B runs, parity tests)
    F4E(1, 3, 2, 2, 32, 1, 0, 8, 16, 4),     // 1.0  16 ->  96 -> 24
    F4E(1, 3, 1, 3, 48, 1, 1, 8, 16, 4),     // 1.1  24 -> 144 -> 24
    F4E(1, 5, 2, 3, 48, 1, 0, 8, 16, 4),     // 2.0  24 -> 144 -> 32
    F4E(1, 5, 1, 4, 32, 1, 1, 8, 16, 4),     // 2.1  32 -> 192 -> 32
    F4E(1, 3, 2, 4, 32, 2, 0, 8, 16, 4),     // 3.0  32 -> 192 -> 64
    F4E(1, 3, 1, 8, 32, 2, 1, 8, 16, 4),     // 3.1  64 -> 384 -> 64
    F4E(1, 5, 1, 8, 32, 3, 0, 8, 16, 4),     // 4.0  64 -> 384 -> 96
    F4E(1, 5, 1, 12, 32, 3, 1, 8, 16, 4),    // 4.1  96 -> 576 -> 96
#include CF_EXP_INC(cf_mbconv4_0)   // A/B sweep of the split mode (CF_F4_VARIANT=3..6): small stride-2 tiles, X fragments resident for wide Cin
};
#undef F4E
#undef F4X

static const F4Entry* f4_find(int dtype, int k, int s, int jx, int nbo, int res) {
    // product switch: 1 = every fp32 block on this file's kernel.  Other values select sweep rows and exist in the experiments
    // build only (var 2 = the split mode's own defaults: never selectable for the exact mode, whose goldens pin the arithmetic)
    static const int want = [] {
        const int v = cf_env_int("CF_F4_VARIANT", 0);
#include CF_EXP_INC(cf_mbconv4_1)
#if !CF_EXP_ON
        return v == 1 ? 1 : 0;
#endif
    }();
    const F4Entry *def0 = nullptr, *def2 = nullptr;
    for (const F4Entry& e : kF4Table)
        if (e.k == k && e.s == s && e.jx == jx && e.nbo == nbo && e.res == res) {
            if (want && e.var == want) return &e;
            if (e.var == 0 && !def0) def0 = &e;
            if (e.var == 2 && !def2) def2 = &e;
        }
    return (dtype == 2 && def2) ? def2 : def0;      // split mode prefers its own row explicitly; nullptr: the block stays on cf_mbconv.hip
}

bool mb4_geometry(int dtype, MbGeom& g, int Cin, int hid, int Cout, int k, int s) {
    // fp32 and split mode share the table (same storage, same tile geometry); some rows are the default in split mode only
    static const int on = cf_ab_int("CF_F4", 1);      // A/B: 0 = cf_mbconv.hip's fp32 instance
    if (!on || (Cin % 8) || (Cout % 8) || Cout > 96 || Cin > 96 || hid == Cin) return false;
    const int jx = (Cin * 4 / 16 + 1) / 2, nbo = (Cout + 31) / 32;
    const F4Entry* e = f4_find(dtype, k, s, jx, nbo, (Cin == Cout && s == 1) ? 1 : 0);
    if (!e || hid % e->hc) return false;
    g = MbGeom{};
    g.ok = true; g.kind = 7; g.S = s;
    g.JX = jx; g.NBO = nbo; g.HC = e->hc; g.nq = hid / e->hc;
    g.NBE = (g.HC + 31) / 32; g.HALF = g.HC / 8; g.rowb = g.HC * 4 + 16; g.KG = e->kg;
    g.lds_bytes = (size_t)e->lds_bytes;
    g.wexp_bytes = (size_t)g.nq * g.NBE * g.JX * 64 * 16;
    g.wdw_floats = (size_t)g.nq * k * k * g.HC;
    g.wproj_bytes = (size_t)g.NBO * g.nq * g.HALF * 64 * 16;
    return true;
}

// depthwise taps [chunk][group of 4 channels][tap][4] and project fragments [n-block][chunk][pair][lane] x 16 B: lane (row slot
// i, half h) holds w[co(i)][chunk base + 8 pair + 4 h + e], e = 0..3 -- k-slot h of the e-th MFMA = channel e of group A / B.
// The expand fragments are cf_mbconv.hip's (mb_pack_weights packs them before calling this).
void mb4_repack(int dtype, const MbGeom& g, int hid, int Cout, int k, const float* wd, const float* wp, float* wdw_host, void* wproj_host) {
    for (int q = 0; q < g.nq; ++q)
        for (int grp = 0; grp < g.HC / 4; ++grp)
            for (int t = 0; t < k * k; ++t)
                for (int c = 0; c < 4; ++c)
                    wdw_host[(((size_t)q * (g.HC / 4) + grp) * k * k + t) * 4 + c] = wd[(size_t)(q * g.HC + grp * 4 + c) * k * k + t];
    __builtin_memset(wproj_host, 0, g.wproj_bytes);
    for (int nbo = 0; nbo < g.NBO; ++nbo)
        for (int q = 0; q < g.nq; ++q)
            for (int jp = 0; jp < g.HALF; ++jp)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    const int hh = (i >> 2) & 1, rr = (i & 3) + 4 * (i >> 3);
                    const int co = nbo * 32 + hh * 16 + rr;                     // = slot_channel(nbo, i) of cf_mbconv.hip
                    if (co >= Cout) continue;
                    char* dst = (char*)wproj_host + ((((size_t)nbo * g.nq + q) * g.HALF + jp) * 64 + lane) * 16;
                    pack_chunk(dtype, wp + (size_t)co * hid + q * g.HC + 8 * jp + 4 * h, dst);
                }
    if (dtype == 2) split_pairs_inplace(wproj_host, (size_t)g.NBO * g.nq * g.KG, g.HALF / g.KG);      // the JS steps of a k-group in pairs
}

hipError_t mb4_launch(hipStream_t s, int dtype, const MbParams& p) {
    const F4Entry* e = f4_find(dtype, p.k, p.s, p.JX, (p.Cout + 31) / 32, p.residual ? 1 : 0);
    if (!e || e->hc != p.HC) return hipErrorInvalidValue;
    return dtype == 2 ? e->fn_sp(s, p) : e->fn(s, p);
}

}  // namespace cf
