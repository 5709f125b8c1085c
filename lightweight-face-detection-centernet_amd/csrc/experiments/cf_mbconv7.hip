// Fused MBConv block for the split-bf16 tolerance mode, register-window depthwise feeding the project MFMAs DIRECTLY
// (MbGeom::kind = 10; cf_mbconv6.hip without its second LDS tile): MBConvBlock.forward (model/centernet.py:89-140), Cout <= 32.
//
// cf_mbconv6.hip sends the depthwise output of a chunk through a second LDS tile so that the project MFMAs can read lane = pixel
// fragments: a round trip and a barrier per chunk, and it lost to cf_mbconv4.hip on every 3x3 block.  The strip layout does not
// need that: in the depthwise phase lanes 0-31 of a wave own the 32 strips of a strip block for channel group A, lanes 32-63 the
// SAME strips for group B.  For pixel i of the strips (i = 0..3) the lane's four channels are then exactly the B operand of
// v_mfma_f32_32x32x8_bf16_1k: column n = lane & 31 = strip, k-slots 0-3 = group A (lanes 0-31), 4-7 = group B (lanes 32-63).
// One depthwise step (a pair of channel groups) therefore ends in 4 x 3 split-bf16 MFMAs on four accumulator blocks (pixels
// i = 0..3 of the 32 strips = 128 pixels per wave), no cross-lane traffic, no LDS.  The waves of a workgroup split the group
// pairs of a chunk (k-groups) and add their partial sums once, at the end, through LDS in k-group order.
//
// MEASURED (profiles/r05_split_restructure.md section 3): correct on every EXACT test, 0.173 ms on layer2.1 (cf_mbconv6.hip 0.175,
// cf_mbconv4.hip 0.184) and 0.48-0.78 ms on the 3x3 blocks (cf_mbconv4.hip 0.31-0.39): with 128-pixel tiles and three hidden chunks
// the per-workgroup fixed costs (fragment load + split, the k-group reduction with eight barriers, the epilogue) outweigh the
// chunk loop, and the k = 8 MFMA form runs the matrix pipe at half efficiency.  Experiments build only (CF_M7=1); the release
// library contains none of it.
#include "../cf_common.h"
#include "../cf_kernels.h"
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace cf {

template <int KS, int S, int HC, int TOH, int TOW, int JX, int NW>
struct M7 {
    static_assert(TOW % 4 == 0 && (S == 1 || S == 2) && (KS == 3 || KS == 5), "strip geometry");
    static constexpr int IH = (TOH - 1) * S + KS, IW0 = (TOW - 1) * S + KS;
    static constexpr int HWQ = ((IW0 + 1) / 2 + 3) / 4;
    static constexpr int IWQ = S == 2 ? 2 * HWQ : (IW0 + 3) / 4, IW = 4 * IWQ;
    static constexpr int IPX = IH * IW, NIB = (IPX + 31) / 32, MAXI = (NIB + NW - 1) / NW;
    static constexpr int NG = HC / 4, QSTRIDE = NG * 64 + 16;
    static constexpr int NPIX = TOH * TOW, SPR = TOW / 4, NSTRIP = NPIX / 4, NSB = NSTRIP / 32;     // strip blocks of 32 strips = 128 pixels
    static constexpr int KG = NW / NSB, NGP = HC / 8, NST = NGP / KG;                                   // k-groups, group pairs per chunk, steps per wave and chunk
    static constexpr int NBE = (HC + 31) / 32;
    static constexpr bool PART = (HC % 32 == 16);
    static constexpr int WXB = NBE * JX * 1024;
    // row pitch: S row pitches == SPR slots (mod 16), so that sixteen consecutive strips -- across tile rows too -- hit sixteen
    // different 16-byte slots for every cell of the window
    static constexpr int RS0 = (IWQ * (QSTRIDE / 16)) % 16, RSW = S == 1 ? SPR % 16 : (SPR / 2) % 8;
    static constexpr int RPAD = S == 1 ? (RSW - RS0 + 16) % 16 : (RSW - RS0 % 8 + 8) % 8;
    static constexpr int ROWP = IWQ * QSTRIDE + RPAD * 16;
    static constexpr int EBYTES = IH * ROWP;
    static constexpr int TAPB = (NG * KS * KS * 16 + 1023) / 1024 * 1024;
    static constexpr int RED = KG > 1 ? NW * 4096 : 0;            // one accumulator block per wave at a time
    static constexpr int LDS0 = EBYTES + WXB + 2 * TAPB;
    static constexpr int LDS = LDS0 > RED ? LDS0 : RED;
    static constexpr int NE = S == 1 ? 4 + KS - 1 : 4 + (KS - 1) / 2;
    static constexpr int NO = S == 1 ? 0 : 4 + (KS - 3) / 2;
    static_assert(NSTRIP % 32 == 0 && NSB * KG == NW && NGP % KG == 0 && HC % 8 == 0 && JX % 2 == 0, "tile / wave geometry");
};

template <int KS, int S, int HC, int TOH, int TOW, int JX, int NW, bool RESID, int MW>
__global__ __launch_bounds__(NW * 64, MW) void mbconv7_kernel(MbParams p) {
    typedef M7<KS, S, HC, TOH, TOW, JX, NW> G;
    constexpr int IW = G::IW, IWQ = G::IWQ, HWQ = G::HWQ, IPX = G::IPX, NIB = G::NIB, MAXI = G::MAXI, NG = G::NG, NBE = G::NBE;
    constexpr int QSTRIDE = G::QSTRIDE, ROWP = G::ROWP, WXB = G::WXB, TAPB = G::TAPB, SPR = G::SPR;
    constexpr int NSB = G::NSB, KG = G::KG, NGP = G::NGP, NST = G::NST;
    constexpr bool PART = G::PART;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* E = smem;
    char* Wst = smem + G::EBYTES;
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
        const unsigned wst = lds0 + G::EBYTES, tdst = wst + WXB + (q & 1) * TAPB;
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

    f32x16 acc[4];                                                  // pixel i = 0..3 of this wave's 32 strips x 32 output channels
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const int sb = wave % NSB, kg = wave / NSB;                     // this wave's strip block and k-group

#ifdef CF_X5_TIMING
    unsigned long long tph[4] = {0, 0, 0, 0}, tq = 0;
#define M7_STAMP(k) { unsigned long long t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tph[k] += t_ - tq; tq = t_; }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tq) :: "memory");
#else
#define M7_STAMP(k)
#endif
    stage_weights(0);
    cf_sync_lds_dma();                // chunk 0's expand weights and taps have landed (later chunks: published by barrier B2)
    for (int q = 0; q < nq; ++q) {
        M7_STAMP(0)
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
        M7_STAMP(1)
        __syncthreads();                                              // B1: E complete
        M7_STAMP(2)
        if (q + 1 < nq) stage_weights(q + 1);

        // ---- phase 2 + 3: per step one PAIR of channel groups: depthwise + Swish on (strip, group A | B) per lane, then the project MFMAs
#pragma unroll
        for (int stp = 0; stp < NST; ++stp) {
            const int gp = kg + stp * KG;                                        // wave-uniform
            const u32x4 wfrag = ld16((const char*)p.wproj + (((size_t)q * NGP + gp) * 64 + lane) * 16);
            const int st = sb * 32 + lds_group_pixel(lane & 31);                 // hardware read groups = sixteen consecutive strips
            const int g = 2 * gp + (lane >> 5);
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
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 y0 = swish2_sel<true>(sacc[i][0]), y1 = swish2_sel<true>(sacc[i][1]);
                float vv[4] = {y0.x, y0.y, y1.x, y1.y};
                CfMma<sp32_t>::run(acc[i], wfrag, pack16<float>(vv));           // k = 8: group A's four channels (lanes 0-31) | group B's (32-63)
            }
        }
        M7_STAMP(3)
        if (q + 1 < nq) cf_sync_lds_dma();                            // every wave is done with E; the next chunk's weights / taps are published
    }
#ifdef CF_X5_TIMING
    if (p.dbg && lane == 0) {
        unsigned long long* o = (unsigned long long*)p.dbg + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + wave) * 4;
        o[0] = tph[0]; o[1] = tph[1]; o[2] = tph[2]; o[3] = tph[3];
    }
#endif

    // ---- combine the k-groups through LDS: block i is finished by k-group i % KG; the others hand it over in k-group order
    if constexpr (KG > 1) {
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int owner = i % KG;
            __syncthreads();                                          // everyone is done with E / the previous block's buffer
            if (kg != owner) {
                float* dst = red + ((size_t)wave * 64 + lane) * 16;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = acc[i][g * 4 + e];
                    st16(dst + g * 4, pack16<float>(t));
                }
            }
            __syncthreads();
            if (kg == owner) {
#pragma unroll
                for (int k2 = 0; k2 < KG; ++k2) {
                    if (k2 == owner) continue;
                    const float* src = red + ((size_t)(k2 * NSB + sb) * 64 + lane) * 16;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float t[4];
                        unpack16<float>(ld16(src + g * 4), t);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][g * 4 + e] += t[e];
                    }
                }
            }
        }
    }

    // ---- epilogue: for block i, lane (n, h) holds 16 contiguous output channels of pixel i of strip n
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (KG > 1 && kg != i % KG) continue;
        const int st = sb * 32 + lds_group_pixel(pl), oy = st / SPR, sx = st - oy * SPR;
        const int gy = oy0 + oy, gx = ox0 + sx * 4 + i;
        if (gy >= p.Hout || gx >= p.Wout) continue;
        const size_t opix = ((size_t)b * p.Hout + gy) * p.Wout + gx;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = h * 16 + g * 4;
            if (ch >= p.Cout) break;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][g * 4 + e];
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
struct M7Entry {
    int var, k, s, jx, hc, res;
    int lds_bytes;
    hipError_t (*fn)(hipStream_t, const MbParams&);
};
template <int KS, int S, int HC, int TOH, int TOW, int JX, int NW, bool RESID, int MW>
static hipError_t m7_launch_t(hipStream_t s, const MbParams& p) {
    typedef M7<KS, S, HC, TOH, TOW, JX, NW> G;
    auto kfn = mbconv7_kernel<KS, S, HC, TOH, TOW, JX, NW, RESID, MW>;
    static thread_local bool configured_dev[32] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    bool& configured = configured_dev[dev & 31];
    if (G::LDS > 64 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    dim3 grid((p.Wout + TOW - 1) / TOW, (p.Hout + TOH - 1) / TOH, p.B), blk(NW * 64);
    set_kernel_tag("void cf::mbconv7_kernel<%d, %d, %d, %d, %d, %d, %d, %s, %d>(cf::MbParams)", KS, S, HC, TOH, TOW, JX, NW, RESID ? "true" : "false", MW);
    hipLaunchKernelGGL(kfn, grid, blk, G::LDS, s, p);
    return hipGetLastError();
}
#define M7E(V, KS, S, JX, HC, RES, TOH, TOW, NW, MW) \
    {V, KS, S, JX, HC, RES, M7<KS, S, HC, TOH, TOW, JX, NW>::LDS, &m7_launch_t<KS, S, HC, TOH, TOW, JX, NW, (RES != 0), MW>}
static const M7Entry kM7Table[] = {
    // var KS S JX HC res tile   waves regs-for-waves/SIMD
    M7E(0, 5, 1, 4, 32, 1, 8, 16, 4, 2),     // 2.1  32 -> 192 -> 32: four waves = four k-groups, one group pair per wave and chunk
    M7E(0, 3, 1, 4, 48, 1, 8, 16, 6, 3),     // 1.1  24 -> 144 -> 24: six waves (JX: Cin = 24 -> 3 chunks per half, padded to 4)
    M7E(0, 3, 2, 2, 32, 0, 8, 16, 4, 2),     // 1.0  16 ->  96 -> 24
    M7E(0, 5, 2, 4, 16, 0, 8, 16, 2, 2),     // 2.0  24 -> 144 -> 32
    // CF_M7_VARIANT=1..3
    M7E(1, 5, 1, 4, 32, 1, 8, 16, 2, 2),     // two waves, two steps each
    M7E(1, 3, 1, 4, 48, 1, 8, 16, 3, 2),     // three waves, two steps each
    M7E(1, 3, 2, 2, 32, 0, 8, 16, 2, 2),
    M7E(2, 5, 1, 4, 32, 1, 8, 32, 8, 2),     // 8x32 tiles: two strip blocks x four k-groups
    M7E(2, 3, 1, 4, 48, 1, 8, 32, 6, 2),     // two strip blocks x three k-groups
    M7E(2, 3, 2, 2, 32, 0, 4, 32, 4, 2),     // stride 2: 4x32 tiles
    M7E(3, 5, 1, 4, 32, 1, 16, 16, 8, 2),    // 16x16 tiles
    M7E(3, 3, 1, 4, 48, 1, 16, 16, 6, 2),
    M7E(3, 3, 2, 2, 16, 0, 8, 16, 2, 2),     // stride 2, chunks of 16
};
#undef M7E

static const M7Entry* m7_find(int k, int s, int jx, int res) {
    static const int want = cf_ab_int("CF_M7_VARIANT", 0);
    const M7Entry* base = nullptr;
    for (const M7Entry& e : kM7Table)
        if (e.k == k && e.s == s && e.jx == jx && e.res == res) {
            if (e.var == want) return &e;
            if (e.var == 0) base = &e;
        }
    return base;
}

// JX is padded to an even chunk count per lane half (chunk pairs): Cin = 24 has three 16-byte chunks per half, the fourth is zero
// weights against a clamped re-read (mb7_pack zeroes the fragment)
static int m7_jx(int Cin) { const int j = (Cin * 4 / 16 + 1) / 2; return (j + 1) & ~1; }

bool mb7_geometry(int dtype, MbGeom& g, int Cin, int hid, int Cout, int k, int s) {
    static const int on = cf_ab_int("CF_M7", 0);                  // experiments build: 1 = every Cout <= 32 block of the split mode on this kernel
    if (!on || dtype != 2 || (Cin % 8) || (Cout % 8) || Cout > 32 || Cin > 32 || hid == Cin) return false;
    const int jx = m7_jx(Cin);
    const M7Entry* e = m7_find(k, s, jx, (Cin == Cout && s == 1) ? 1 : 0);
    if (!e || hid % e->hc) return false;
    g = MbGeom{};
    g.ok = true; g.kind = 10; g.S = s;
    g.JX = jx; g.NBO = 1; g.HC = e->hc; g.nq = hid / e->hc;
    g.NBE = (g.HC + 31) / 32; g.HALF = g.HC / 8; g.rowb = 0; g.KG = 1;
    g.lds_bytes = (size_t)e->lds_bytes;
    g.wexp_bytes = (size_t)g.nq * g.NBE * g.JX * 64 * 16;
    g.wdw_floats = (size_t)g.nq * k * k * g.HC + 256;
    g.wproj_bytes = (size_t)g.nq * g.HALF * 64 * 16;              // [chunk][group pair][lane] x 16 B
    return true;
}

// expand fragments and taps as mb6_pack; project fragments [chunk][group pair gp][lane] x 16 B ([4 x hi | 4 x lo]; -ln 2 folded): lane (row
// slot i -> output channel, half h) holds w[co][chunk base + (2 gp + h) 4 + e] -- k-slots 0-3 = group 2 gp, 4-7 = group 2 gp + 1
void mb7_pack(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
              void* wexp_host, float* wdw_host, void* wproj_host) {
    MbGeom g6 = g; g6.KG = 1;
    std::vector<char> scratch(g.wproj_bytes);
    mb6_pack(g6, Cin, hid, Cout, k, we, wd, wp, wexp_host, wdw_host, scratch.data());       // (its project fragments are not used)
    __builtin_memset(wproj_host, 0, g.wproj_bytes);
    const int NGP = g.HC / 8;
    for (int q = 0; q < g.nq; ++q)
        for (int gp = 0; gp < NGP; ++gp)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                const int co = ((i >> 2) & 1) * 16 + (i & 3) + 4 * (i >> 3);                 // slot_channel(0, i)
                if (co >= Cout) continue;
                float v[4];
                for (int e = 0; e < 4; ++e) v[e] = kCfNegLn2 * wp[(size_t)co * hid + q * g.HC + (2 * gp + h) * 4 + e];
                __builtin_memcpy((char*)wproj_host + (((size_t)q * NGP + gp) * 64 + lane) * 16, v, 16);
            }
    split_pairs_inplace(wproj_host, (size_t)g.nq * NGP, 1);
}

hipError_t mb7_launch(hipStream_t s, const MbParams& p) {
    const M7Entry* e = m7_find(p.k, p.s, p.JX, p.residual ? 1 : 0);
    if (!e || e->hc != p.HC) return hipErrorInvalidValue;
    return e->fn(s, p);
}

}  // namespace cf
