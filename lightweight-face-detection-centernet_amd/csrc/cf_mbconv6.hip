// Fused MBConv block for the split-bf16 TOLERANCE MODE, third generation (MbGeom::kind = 9): expand 1x1 (+Swish) -> depthwise
// k x k (+Swish) -> project 1x1 (+residual), MBConvBlock.forward (model/centernet.py:89-140), for the narrow early blocks
// (Cout <= 32: layer1.0 ... 2.1) where the depthwise dominates.
//
// cf_mbconv4.hip (round 4) maps a lane to ONE output pixel: per tap and 4-channel group one ds_read_b128 and four v_fmac_f32 with
// an SGPR tap -- an instruction form that issues at half rate on this chip (profiles/r05_valu_clock_probe.md) -- and the LDS
// array was busy 40-50 % of those launches.  Here the depthwise is cf_mbconv5.hip's:
//   * the expanded tile lives in LDS as x-quad cells E[row][quad][4-channel group][4 pixels] x 16 B (odd quad pitch in 16-byte
//     slots); a lane owns a strip of FOUR x-adjacent output pixels of one channel group and reads 4 + KS - 1 cells per kernel
//     row (8 reads for 20 taps at 5x5) instead of one per tap: LDS reads / 2-2.5;
//   * taps through LDS (DMA'd per chunk, broadcast ds_read_b128), two packed FMAs per tap and pixel -- two multiply-adds per
//     4.2 cycles where the SGPR-tap v_fmac_f32 did one per 4.1;
//   * a lane's (strip, channel group) item is free, so the wave's 64 lanes are dealt items strip-major: every lane works whatever
//     the tile's strip count.
// The depthwise output of a chunk goes to a second LDS tile D[pixel][HC] (fp32, odd pitch), from which the project MFMAs read
// their B fragments (lane = pixel, 16 bytes per k-step); the project accumulators stay in registers across the hidden chunks
// (one 32-channel block: Cout <= 32).  X fragments are loaded and SPLIT into bf16 (hi, lo) once per workgroup and stay in
// registers for all chunks.  Two barriers per chunk: [expand -> E] B1 [depthwise E -> D] B2 [project D -> acc].
#include "cf_exp.h"
#include "cf_common.h"
#include "cf_kernels.h"
#include <cstdlib>
#include <type_traits>

namespace cf {

template <int KS, int S, int HC, int TOH, int TOW, int JX, int NW>
struct M6 {
    static_assert(TOW % 4 == 0 && (S == 1 || S == 2) && (KS == 3 || KS == 5), "strip geometry");
    static constexpr int IH = (TOH - 1) * S + KS, IW0 = (TOW - 1) * S + KS;
    static constexpr int HWQ = ((IW0 + 1) / 2 + 3) / 4;
    static constexpr int IWQ = S == 2 ? 2 * HWQ : (IW0 + 3) / 4, IW = 4 * IWQ;
    static constexpr int IPX = IH * IW, NIB = (IPX + 31) / 32, MAXI = (NIB + NW - 1) / NW;
    static constexpr int NG = HC / 4, QSTRIDE = NG * 64 + 16;
    static constexpr int NPIX = TOH * TOW, NPB = NPIX / 32, KG = NW / NPB;
    static constexpr int SPR = TOW / 4, NSTRIP = NPIX / 4, NITEM = NSTRIP * NG, UNITS = (NITEM + 63) / 64;
    static constexpr int NBE = (HC + 31) / 32;
    static constexpr bool PART = (HC % 32 == 16);
    static constexpr int HALF = HC / 8, JS = HALF / KG;           // project k-steps (4 channels each) per lane half / per k-group
    static constexpr int WXB = NBE * JX * 1024;
    // row pitch: S row pitches == SPR slots (mod 16), so that sixteen consecutive strips -- across tile rows too -- hit sixteen
    // different 16-byte slots for every cell of the window
    static constexpr int RS0 = (IWQ * (QSTRIDE / 16)) % 16, RSW = S == 1 ? SPR % 16 : (SPR / 2) % 8;
    static constexpr int RPAD = S == 1 ? (RSW - RS0 + 16) % 16 : (RSW - RS0 % 8 + 8) % 8;
    static constexpr int ROWP = IWQ * QSTRIDE + RPAD * 16;
    static constexpr int EBYTES = IH * ROWP;
    static constexpr int DROW = HC * 4 + 16, DBYTES = NPIX * DROW;
    static constexpr int TAPB = (NG * KS * KS * 16 + 1023) / 1024 * 1024;
    static constexpr int RED = (KG - 1) * NPB * 64 * 64;          // k-group partial sums (reuses E)
    static constexpr int LDS0 = EBYTES + DBYTES + WXB + 2 * TAPB;
    static constexpr int LDS = LDS0 > RED ? LDS0 : RED;
    static constexpr int NE = S == 1 ? 4 + KS - 1 : 4 + (KS - 1) / 2;
    static constexpr int NO = S == 1 ? 0 : 4 + (KS - 3) / 2;
    static_assert(NPIX % 32 == 0 && NPB * KG == NW && HALF % KG == 0 && HC % 8 == 0 && JX % 2 == 0, "tile / wave geometry");
};

template <int KS, int S, int HC, int TOH, int TOW, int JX, int NW, bool RESID, int MW>
__global__ __launch_bounds__(NW * 64, MW) void mbconv6_kernel(MbParams p) {
    typedef M6<KS, S, HC, TOH, TOW, JX, NW> G;
    constexpr int IW = G::IW, IWQ = G::IWQ, HWQ = G::HWQ, IPX = G::IPX, NIB = G::NIB, MAXI = G::MAXI, NG = G::NG, NBE = G::NBE;
    constexpr int QSTRIDE = G::QSTRIDE, ROWP = G::ROWP, WXB = G::WXB, TAPB = G::TAPB, SPR = G::SPR, NSTRIP = G::NSTRIP, NITEM = G::NITEM, UNITS = G::UNITS;
    constexpr int NPB = G::NPB, KG = G::KG, HALF = G::HALF, JS = G::JS, DROW = G::DROW;
    constexpr bool PART = G::PART;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* E = smem;
    char* D = smem + G::EBYTES;
    char* Wst = D + G::DBYTES;
    char* Tap = Wst + WXB;

    const int tid = threadIdx.x;
    int lane = tid & 63;                                            // (not const: re-"defined" per chunk, see the chunk loop)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pl = lane & 31, h = lane >> 5;
    const int ox0 = blockIdx.x * TOW, oy0 = blockIdx.y * TOH, b = blockIdx.z;
    const int nq = p.nq;
    const char* xbase = (const char*)p.x + (size_t)b * p.Hin * p.Win * p.Cin * 4;
    const unsigned rowbytes = (unsigned)p.Cin * 4;

    // expand fragments of chunk q -> Wst, its depthwise taps -> Tap[q & 1] (inline-asm LDS DMA: see cf_mbconv5.hip)
    auto stage_weights = [&](int q) {
        const char* srcx = (const char*)p.wexp + (size_t)q * WXB;
        const char* srct = (const char*)p.wdw + (size_t)q * (NG * KS * KS * 16);
        const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
        const unsigned wst = lds0 + G::EBYTES + G::DBYTES, tdst = wst + WXB + (q & 1) * TAPB;
        for (int c = wave; c < (WXB + TAPB) / 1024; c += NW) {
            const bool isw = c < WXB / 1024;
            const char* src = (isw ? srcx + c * 1024 : srct + (c - WXB / 1024) * 1024) + lane * 16;
            const unsigned dst = __builtin_amdgcn_readfirstlane(isw ? wst + c * 1024 : tdst + (c - WXB / 1024) * 1024);
            unsigned m0save;                                       // m0 is the compiler's: saved and restored around the DMA
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0" : "=&s"(m0save) : "s"(dst), "v"(src));
        }
    };

    // ---- X fragments of this wave's halo pixel blocks: loaded once, split once into bf16 (hi, lo) chunk pairs, resident
    u32x4 xs[MAXI][JX];
    bool xsv[MAXI];
#pragma unroll
    for (int t = 0; t < MAXI; ++t) {
        const int ib = wave + NW * t;
        xsv[t] = false;
        if (ib < NIB) {
            const int ip = ib * 32 + pl;
            const int ipc = ip < IPX ? ip : IPX - 1;
            const int iy = ipc / IW, xp = ipc - iy * IW;
            const int ix = S == 2 ? (xp < 4 * HWQ ? 2 * xp : 2 * (xp - 4 * HWQ) + 1) : xp;
            const int gy = oy0 * S - p.pad_lo + iy, gx = ox0 * S - p.pad_lo + ix;
            xsv[t] = ip < IPX && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
            const int cy = min(max(gy, 0), p.Hin - 1), cx = min(max(gx, 0), p.Win - 1);
            // half h reads the row's chunks h JH .. h JH + JX - 1 (JH = real chunks per half; a padded slot meets zero weights)
            const unsigned off = ((unsigned)cy * (unsigned)p.Win + (unsigned)cx) * rowbytes + (unsigned)(h * ((p.Cin / 4 + 1) / 2) * 16);
            u32x4 raw[JX];
#pragma unroll
            for (int j = 0; j < JX; ++j) raw[j] = ld16(xbase + off + j * 16);
#pragma unroll
            for (int j = 0; j < JX; j += 2) { const SplitPair sp2 = split8(raw[j], raw[j + 1]); xs[t][j] = sp2.hi; xs[t][j + 1] = sp2.lo; }
        }
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int pb = wave % NPB, kg = wave / NPB;                     // phase 3: this wave's pixel block and k-group

    stage_weights(0);
    cf_sync_lds_dma();                // chunk 0's expand weights and taps have landed (later chunks: published by barrier B2)
    for (int q = 0; q < nq; ++q) {
        const char* tapq = Tap + (q & 1) * TAPB;
        // loop-invariant per-lane addresses are recomputed per chunk instead of being held (and spilled): cf_mbconv5.hip
        asm volatile("" : "+v"(lane), "+v"(pl), "+v"(h));

        // ---- phase 1: expand + Swish -> E (x-quad cells).  E was last read by the depthwise of chunk q - 1, which every wave
        // left through barrier B2 of that chunk.
#pragma unroll
        for (int t = 0; t < MAXI; ++t) {
            const int ib = wave + NW * t;
            if (ib >= NIB) break;
            const int ip = ib * 32 + pl;
            const bool ipok = ip < IPX;
            const int ipc = ipok ? ip : 0;
            const int iy = ipc / IW, xp = ipc - iy * IW;
            char* ecell = E + (unsigned)iy * (unsigned)ROWP + (unsigned)(xp >> 2) * (unsigned)QSTRIDE + (unsigned)(xp & 3) * 16u;
#pragma unroll
            for (int nbl = 0; nbl < NBE; ++nbl) {
                f32x16 a;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = 0.0f;
                const char* wb = Wst + (nbl * JX * 64 + lane) * 16;
#pragma unroll
                for (int j = 0; j < JX; j += 2) {
                    const u32x4 whi = ld16(wb + j * 1024), wlo = ld16(wb + (j + 1) * 1024);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, wlo), __builtin_bit_cast(cf_bf16x8, xs[t][j]), a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, whi), __builtin_bit_cast(cf_bf16x8, xs[t][j + 1]), a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, whi), __builtin_bit_cast(cf_bf16x8, xs[t][j]), a, 0, 0, 0);
                }
                const bool half_block = PART && nbl == NBE - 1;       // 8 channels on each lane half (mb_pack_weights)
                const int ch0 = half_block ? nbl * 32 + h * 8 : nbl * 32 + h * 16;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (half_block && g >= 2) break;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        f32x2 x2; x2.x = a[g * 4 + e]; x2.y = a[g * 4 + e + 1];
                        const f32x2 y2 = swish2_sel<true>(x2);
                        v[e] = xsv[t] ? y2.x : 0.0f; v[e + 1] = xsv[t] ? y2.y : 0.0f;      // outside the image: ZeroPad2d of the expanded tensor
                    }
                    if (ipok) st16(ecell + (ch0 / 4 + g) * 64, pack16<float>(v));
                }
            }
        }
        __syncthreads();                                              // B1: E complete; D free (the project of chunk q - 1 is done)
        if (q + 1 < nq) stage_weights(q + 1);

        // ---- phase 2: depthwise + Swish, one (strip of four pixels, channel group) item per lane -> D
        for (int u = wave; u < UNITS; u += NW) {
            const int v0 = (lane & 32) + lds_group_pixel(lane & 31);             // hardware read groups = sixteen consecutive items
            const int it = u * 64 + v0;
            const bool iok = it < NITEM;
            const int itc = iok ? it : NITEM - 1;
            const int g = itc / NSTRIP, st = itc - g * NSTRIP;                   // strip-major: consecutive lanes = consecutive strips
            const int oy = st / SPR, sx = st - oy * SPR;
            const char* wq = tapq + g * (KS * KS * 16);
            const char* eb = E + (unsigned)(oy * S) * (unsigned)ROWP + (unsigned)sx * (unsigned)QSTRIDE + (unsigned)g * 64u;
            f32x2 sacc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { sacc[i][0].x = sacc[i][0].y = 0.0f; sacc[i][1].x = sacc[i][1].y = 0.0f; }
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const char* er = eb + (unsigned)ky * (unsigned)ROWP;
                u32x4 ce[G::NE], co[G::NO > 0 ? G::NO : 1], wr[KS];
#pragma unroll
                for (int j = 0; j < G::NE; ++j) ce[j] = ld16(er + (j >> 2) * QSTRIDE + (j & 3) * 16);
                if constexpr (S == 2) {
#pragma unroll
                    for (int j = 0; j < G::NO; ++j) co[j] = ld16(er + (HWQ + (j >> 2)) * QSTRIDE + (j & 3) * 16);
                }
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) wr[kx] = ld16(wq + (ky * KS + kx) * 16);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const u32x4 w = wr[kx];
                    f32x2 w01, w23; w01.x = __uint_as_float(w.x); w01.y = __uint_as_float(w.y); w23.x = __uint_as_float(w.z); w23.y = __uint_as_float(w.w);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4& c = S == 1 ? ce[i + kx] : ((kx & 1) ? co[i + (kx >> 1)] : ce[i + (kx >> 1)]);
                        f32x2 e01, e23; e01.x = __uint_as_float(c.x); e01.y = __uint_as_float(c.y); e23.x = __uint_as_float(c.z); e23.y = __uint_as_float(c.w);
                        sacc[i][0] = fma2(e01, w01, sacc[i][0]);
                        sacc[i][1] = fma2(e23, w23, sacc[i][1]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(sacc[i][0]), "+v"(sacc[i][1]));      // (pins the FMAs in front of the next row's reads)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (iok) {
                char* drow = D + (unsigned)(oy * TOW + sx * 4) * (unsigned)DROW + (unsigned)g * 16u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 y0 = swish2_sel<true>(sacc[i][0]), y1 = swish2_sel<true>(sacc[i][1]);
                    float vv[4] = {y0.x, y0.y, y1.x, y1.y};
                    st16(drow + i * DROW, pack16<float>(vv));
                }
            }
        }
        cf_sync_lds_dma();                                            // B2: D complete; E free; the next chunk's weights / taps are published

        // ---- phase 3: project MFMAs of this wave's pixel block over its k-group's share of the chunk
        {
            const int o = pb * 32 + lds_group_pixel(pl);                          // conflict-free ds_read_b128 groups: sixteen consecutive pixels
            const char* dr = D + (unsigned)o * (unsigned)DROW + (unsigned)((h * HALF + kg * JS) * 16);
            const char* wp = (const char*)p.wproj + (((size_t)q * KG + kg) * JS * 64 + lane) * 16;
#pragma unroll
            for (int js = 0; js + 1 < JS; js += 2)
                CfMma<sp32_t>::run2(acc, ld16(wp + js * 1024), ld16(wp + (js + 1) * 1024), ld16(dr + js * 16), ld16(dr + (js + 1) * 16));
            if constexpr (JS & 1) CfMma<sp32_t>::run(acc, ld16(wp + (JS - 1) * 1024), ld16(dr + (JS - 1) * 16));
        }
    }

    // ---- combine the k-groups through LDS, in k-group order
    if constexpr (KG > 1) {
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();                                              // everyone is done with E / D
        if (kg > 0) {
            float* dst = red + ((size_t)((kg - 1) * NPB + pb) * 64 + lane) * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = acc[g * 4 + e];
                st16(dst + g * 4, pack16<float>(t));
            }
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int k2 = 1; k2 < KG; ++k2) {
            const float* src = red + ((size_t)((k2 - 1) * NPB + pb) * 64 + lane) * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float t[4];
                unpack16<float>(ld16(src + g * 4), t);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[g * 4 + e] += t[e];
            }
        }
    }

    // ---- epilogue: lane (pl, h) holds 16 contiguous output channels of its pixel
    {
        const int o = pb * 32 + lds_group_pixel(pl), oy = o / TOW, ox = o % TOW;
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (gy >= p.Hout || gx >= p.Wout) return;
        const size_t opix = ((size_t)b * p.Hout + gy) * p.Wout + gx;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = h * 16 + g * 4;
            if (ch >= p.Cout) break;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[g * 4 + e];
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

// ---------------------------------------------------------------- host side
struct M6Entry {
    int var, k, s, jx, hc, res, kg, half;
    int lds_bytes;
    hipError_t (*fn)(hipStream_t, const MbParams&);
};
template <int KS, int S, int HC, int TOH, int TOW, int JX, int NW, bool RESID, int MW>
static hipError_t m6_launch_t(hipStream_t s, const MbParams& p) {
    typedef M6<KS, S, HC, TOH, TOW, JX, NW> G;
    auto kfn = mbconv6_kernel<KS, S, HC, TOH, TOW, JX, NW, RESID, MW>;
    static thread_local bool configured_dev[32] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    bool& configured = configured_dev[dev & 31];
    if (G::LDS > 64 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    dim3 grid((p.Wout + TOW - 1) / TOW, (p.Hout + TOH - 1) / TOH, p.B), blk(NW * 64);
    set_kernel_tag("void cf::mbconv6_kernel<%d, %d, %d, %d, %d, %d, %d, %s, %d>(cf::MbParams)", KS, S, HC, TOH, TOW, JX, NW, RESID ? "true" : "false", MW);
    hipLaunchKernelGGL(kfn, grid, blk, G::LDS, s, p);
    return hipGetLastError();
}
#define M6E(V, KS, S, JX, HC, RES, TOH, TOW, NW, MW) \
    {V, KS, S, JX, HC, RES, M6<KS, S, HC, TOH, TOW, JX, NW>::KG, M6<KS, S, HC, TOH, TOW, JX, NW>::HALF, M6<KS, S, HC, TOH, TOW, JX, NW>::LDS, \
     &m6_launch_t<KS, S, HC, TOH, TOW, JX, NW, (RES != 0), MW>}
static const M6Entry kM6Table[] = {
    // Measured against cf_mbconv4.hip's one-pixel-per-lane kernel (B = 64, 640x640, ms, same box, gpurun_out/r05k/m6_variants.txt ->
    // profiles/r05_split_restructure.md): this kernel / that one, best variant of each block
    //   1.0 (3x3 s2)  0.584 / 0.387      1.1 (3x3 s1)  0.404 / 0.320      2.0 (5x5 s2)  0.322 / 0.277      2.1 (5x5 s1)  0.176 / 0.184
    // The register window halves the LDS reads and the packed taps halve the tap issue time, but the depthwise output now makes a
    // round trip through a second LDS tile and a second barrier per chunk, and 8x16 tiles on whole x-quads expand 25 % more halo
    // columns (20 instead of 18): it pays only where the depthwise dominates the block -- 5x5 stride 1.  Only layer2.1 runs here.
    // var KS S JX HC res tile   waves regs-for-waves/SIMD
    M6E(0, 5, 1, 4, 32, 1, 8, 16, 8, 4),     // 2.1  32 -> 192 -> 32: eight waves (two k-groups), two workgroups per CU
#include CF_EXP_INC(cf_mbconv6_0)   // the sweep (CF_M6_VARIANT=1..4)
};
#undef M6E

static const M6Entry* m6_find(int k, int s, int jx, int res) {
    static const int want = cf_ab_int("CF_M6_VARIANT", 0);
    const M6Entry* base = nullptr;
    for (const M6Entry& e : kM6Table)
        if (e.k == k && e.s == s && e.jx == jx && e.res == res) {
            if (e.var == want) return &e;
            if (e.var == 0) base = &e;
        }
    return base;
}

// JX is padded to an even chunk count per lane half (chunk pairs): Cin = 24 has three 16-byte chunks per half, the fourth is zero
// weights against a clamped re-read (mb6_pack zeroes the fragment)
static int m6_jx(int Cin) { const int j = (Cin * 4 / 16 + 1) / 2; return (j + 1) & ~1; }

bool mb6_geometry(int dtype, MbGeom& g, int Cin, int hid, int Cout, int k, int s) {
    static const int on = cf_ab_int("CF_M6", 1);
    if (!on || dtype != 2 || (Cin % 8) || (Cout % 8) || Cout > 32 || Cin > 32 || hid == Cin) return false;
    const int jx = m6_jx(Cin);
    const M6Entry* e = m6_find(k, s, jx, (Cin == Cout && s == 1) ? 1 : 0);
    if (!e || hid % e->hc) return false;
    g = MbGeom{};
    g.ok = true; g.kind = 9; g.S = s;
    g.JX = jx; g.NBO = 1; g.HC = e->hc; g.nq = hid / e->hc;
    g.NBE = (g.HC + 31) / 32; g.HALF = e->half; g.rowb = 0; g.KG = e->kg;
    g.lds_bytes = (size_t)e->lds_bytes;
    g.wexp_bytes = (size_t)g.nq * g.NBE * g.JX * 64 * 16;
    g.wdw_floats = (size_t)g.nq * k * k * g.HC + 256;
    g.wproj_bytes = (size_t)g.nq * g.HALF * 64 * 16;
    return true;
}

// expand fragments [chunk][n-block][JX][lane] x 16 B (split pairs; -log2 e folded), taps [chunk][group][tap][4], project fragments
// [chunk][k-group][JS][lane] x 16 B (-ln 2 folded; split pairs within a k-group): lane (row slot i -> output channel, half h) holds
// w[co][chunk base + (h HALF + kg JS + js) 4 + e]
void mb6_pack(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
              void* wexp_host, float* wdw_host, void* wproj_host) {
    const int NCx = Cin * 4 / 16;                                  // real 16-byte chunks of an input row
    const int JH = (NCx + 1) / 2;                                  // real chunks per lane half
    __builtin_memset(wexp_host, 0, g.wexp_bytes);
    for (int q = 0; q < g.nq; ++q)
        for (int nbl = 0; nbl < g.NBE; ++nbl)
            for (int j = 0; j < g.JX; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    int cl = (nbl * 32 + ((i >> 2) & 1) * 16 + (i & 3) + 4 * (i >> 3));          // slot_channel(nbl, i)
                    if (nbl == g.NBE - 1 && g.HC % 32 == 16) {
                        const int hh = (i >> 2) & 1, rr = (i & 3) + 4 * (i >> 3);
                        cl = rr < 8 ? nbl * 32 + hh * 8 + rr : g.HC;
                    }
                    const int c = h * JH + j;                          // 16-byte chunk of the Cin row read by (half h, slot j)
                    if (cl >= g.HC || j >= JH || c >= NCx) continue;
                    float v[4];
                    for (int e = 0; e < 4; ++e) v[e] = kCfNegLog2e * we[(size_t)(q * g.HC + cl) * Cin + c * 4 + e];
                    __builtin_memcpy((char*)wexp_host + ((((size_t)q * g.NBE + nbl) * g.JX + j) * 64 + lane) * 16, v, 16);
                }
    split_pairs_inplace(wexp_host, (size_t)g.nq * g.NBE, g.JX);
    for (int q = 0; q < g.nq; ++q)
        for (int grp = 0; grp < g.HC / 4; ++grp)
            for (int t = 0; t < k * k; ++t)
                for (int c = 0; c < 4; ++c)
                    wdw_host[(((size_t)q * (g.HC / 4) + grp) * k * k + t) * 4 + c] = wd[(size_t)(q * g.HC + grp * 4 + c) * k * k + t];
    __builtin_memset(wproj_host, 0, g.wproj_bytes);
    const int JS = g.HALF / g.KG;
    for (int q = 0; q < g.nq; ++q)
        for (int kg = 0; kg < g.KG; ++kg)
            for (int js = 0; js < JS; ++js)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    const int co = ((i >> 2) & 1) * 16 + (i & 3) + 4 * (i >> 3);                 // slot_channel(0, i)
                    if (co >= Cout) continue;
                    float v[4];
                    for (int e = 0; e < 4; ++e) v[e] = kCfNegLn2 * wp[(size_t)co * hid + q * g.HC + (h * g.HALF + kg * JS + js) * 4 + e];
                    __builtin_memcpy((char*)wproj_host + ((((size_t)q * g.KG + kg) * JS + js) * 64 + lane) * 16, v, 16);
                }
    split_pairs_inplace(wproj_host, (size_t)g.nq * g.KG, JS);
}

hipError_t mb6_launch(hipStream_t s, const MbParams& p) {
    const M6Entry* e = m6_find(p.k, p.s, p.JX, p.residual ? 1 : 0);
    if (!e || e->hc != p.HC) return hipErrorInvalidValue;
    return e->fn(s, p);
}

}  // namespace cf
