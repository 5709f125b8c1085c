// Third-generation fused MBConv kernels (bf16 storage, stride 1): the depthwise taps run on the MATRIX cores.
//
// MBConvBlock.forward (model/centernet.py:89-140): expand 1x1 (+Swish) -> depthwise k x k (+Swish) -> project 1x1.
// cf_mbconv2.hip evaluates the depthwise with v_dot2c_f32_f16 (6 / 15 VALU instructions per output element for 3x3 / 5x5):
// 129 M of them per 640x640 image, ~20 % of the VALU issue budget of a forward whose every large kernel is VALU-issue-bound
// (profiles/r02h_valu_bound.md).  The matrix pipe is idle in those kernels.  Here a depthwise row is a banded (Toeplitz) 4x4
// matrix per channel and runs on v_mfma_f32_4x4x4_16b_f16 -- sixteen independent 4x4x4 products per instruction:
//
//   D_blk[i][j] = sum_k A_blk[i][k] * B_blk[k][j]      blk = (channel group kg, pixel-quad slot pg), 16 blocks
//     B: lane (blk, j)  = 4 x-consecutive fp16 of ONE channel of the expanded tile (one 8-byte LDS cell), quad (pg, j)
//     A: lane (blk, i)  = taps of output x-offset i against the 4 inputs of the quad: w[ky][4 ks + k - i] (0 outside)
//     D: lane (blk, j) register i = output pixel i of output quad (pg, j) for channel (kg, g)
//   one output quad needs KS rows x 2 k-steps (inputs x .. x+7) -> 6 (3x3) / 10 (5x5) MFMAs per channel-of-4-pixels, each
//   ~4 issue cycles beside other VALU work (tools/ub_mfma_coissue_probe.hip) instead of 4 x 6 / 4 x 15 v_dot2c.
//   CBSZ = 2 broadcasts the A operand of block `abid` inside each group of four blocks, so ONE register pair holds the
//   Toeplitz operands of four channels (block pg of group kg keeps channel 4 q + pg): A stays resident in 24 / 40 VGPRs.
//
// The operand layouts chain without any cross-lane traffic:
//   expand MFMA 32x32x16 (lane = hidden channel, registers = 4 x-quads of the halo pixel block) -> Swish -> cvt_pkrtz ->
//   ds_write_b64 into cells E[halo quad][channel][4 x fp16]  (a quad's 32 channels = 256 contiguous bytes + 16 pad)
//   -> a lane's eight channels (kg*8 .. +7) of a quad are 64 contiguous bytes: four ds_read_b128 feed eight MFMAs
//   -> D registers -> Swish -> bf16: for output pixel i the lane holds 8 consecutive channels = one 16-byte store (or one
//   MFMA 16x16x32 B fragment: n = lane & 15 = output quad, k-group = lane >> 4 = kg).
// The lane -> output-quad table deals quads to the 16 slots of a set so that the 16-lane hardware groups of ds_read_b128
// (MI355X_MICROARCH.md, LDS) hit sixteen distinct 16-byte slots: cell pitch 272 B = 17 slots, so a quad's slot class is
// its linear halo index mod 16 and slot (pg, j) takes class pg + 4 j.
//
// Numerics are those of cf_mbconv2.hip (same storage points: fp16 round-toward-zero tile, fp16 taps, fp32 accumulation,
// pre-scaled Swish); only the fp32 summation order inside a row differs.  oracle/bf16_emulation.py is the checker.
#include "cf_exp.h"
#include "cf_common.h"
#include "cf_kernels.h"
#include "cf_mx.h"
#include <cstdlib>

namespace cf {

// ---------------------------------------------------------------- geometry shared by host and device (stride 1)
template <int KS, int JX, int TOH, int TOW, int NW, bool ALDS = true>
struct Mx {
    static constexpr int HC = 32;                                   // hidden channels per round = 4 groups x 8
    static constexpr int IH = TOH + KS - 1, IWQ = TOW / 4 + 1, IWP = IWQ * 4;   // halo tile in x-quads (covers TOW + KS - 1)
    static constexpr int NQD = IH * IWQ, NIB = (NQD + 7) / 8, IPX = NQD * 4;
    static constexpr int CP = HC * 8 + 16;                          // bytes per quad cell row: 32 channels x 8 B + 16 (17 slots)
    static constexpr int EBYTES = NIB * 8 * CP;                     // whole pixel blocks: phase 1 stores are unconditional
    static constexpr int NOQ = TOH * (TOW / 4), NSET = (NOQ + 15) / 16;
    static constexpr int WXB = JX * 1024;                           // expand weight fragments of one round
    static constexpr int NSTEP = KS * 2;                            // (ky, k-step) pairs per output quad
    static constexpr int ATB = 2 * NSTEP * 512;                     // Toeplitz operand table of one round
    static constexpr int LDS = EBYTES + WXB + (ALDS ? ATB : 0);
    static_assert(TOW % 4 == 0 && KS <= 5, "x-quads; two k-steps cover 4 + KS - 1 <= 8 inputs");
};

// ================================================================== expand + depthwise (project stays a GEMM launch)
// grid = (tiles, hid / 32, batch); one round of 32 hidden channels per workgroup; only the depthwise output reaches HBM
template <int KS, int JX, int TOH, int TOW, int NW, bool ALDS>
__global__ __launch_bounds__(NW * 64) void expdw_mx_kernel(MbParams p) {
    typedef Mx<KS, JX, TOH, TOW, NW, ALDS> G;
    constexpr int IWQ = G::IWQ, IWP = G::IWP, IPX = G::IPX, NIB = G::NIB, CP = G::CP, NSET = G::NSET, WXB = G::WXB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* E = smem;
    char* Wst = smem + G::EBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const int tiles_x = (p.Wout + TOW - 1) / TOW;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int ox0 = txi * TOW, oy0 = tyi * TOH, b = blockIdx.z;
    const int grp = blockIdx.y;

    {
        const char* srcx = (const char*)p.wexp + (size_t)grp * WXB;
        for (int c = wave; c < WXB / 1024; c += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcx + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(Wst + c * 1024), 16, 0, 0);
    }
    asm volatile("" ::: "memory");        // the loads below stay behind the DMAs in program order (cf_sync_lds_dma_keep counts on it)
    // Toeplitz A operands of this round ([2 channel quads][KS][2 k-steps][lane] x 8 B): an LDS copy next to the expand
    // weights (ALDS: 81-110 VGPRs, + 6 / 10 KB of LDS) or resident register pairs (132-148 VGPRs)
    char* Ats = Wst + WXB;
    u32x2 A[ALDS ? 1 : 2][ALDS ? 1 : KS][2];
    if constexpr (ALDS) {
        const char* srca = (const char*)p.wdw + (size_t)grp * G::ATB;
        for (int c = wave; c < G::ATB / 1024; c += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srca + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(Ats + c * 1024), 16, 0, 0);
    } else {
        const u32x2* at = reinterpret_cast<const u32x2*>(p.wdw) + (size_t)grp * (2 * KS * 2) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) A[q][ky][ks] = at[((q * KS + ky) * 2 + ks) * 64];
    }
    asm volatile("" ::: "memory");
    const char* xbase = (const char*)p.x + (size_t)b * p.Hin * p.Win * p.Cin * 2;
    const unsigned rowbytes = (unsigned)p.Cin * 2;
    auto load_x = [&](int ib, u32x4* xf) -> bool {
        const int ip = ib * 32 + pl;
        const int ipc = ip < IPX ? ip : IPX - 1;
        const int iy = ipc / IWP, ix = ipc - iy * IWP;
        const int gy = oy0 - p.pad_lo + iy, gx = ox0 - p.pad_lo + ix;
        const int cy = min(max(gy, 0), p.Hin - 1), cx = min(max(gx, 0), p.Win - 1);
        if (p.xblock) {
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
#ifdef CF_ABLATION
    const int abl = p.nw;            // timing experiments only (-DCF_ABLATION, CF_MX_ABL): 1 no depthwise MFMAs, 2 no output Swish, 4 no expand Swish, 8 one X load, 16 no stores, 32 stores as 1 KB runs
#else
    constexpr int abl = 0;           // (the run-time tests cost the production kernel 4-7 %: compiled out)
#endif
    u32x4 xa[JX];
    bool va = false;
    if (wave < NIB) va = load_x(wave, xa);
    static_assert(NIB >= NW, "every wave issues the X loads the wait below keeps in flight");
    cf_sync_lds_dma_keep<(ALDS ? 0 : 2 * KS * 2) + JX>();      // expand weights / operand table (DMA, issued first) landed for every wave
    mask_x(xa, va);

    // ---- phase 1: expand + Swish -> quad cells.  D rows (r & 3) + 8 (r >> 2) + 4 h: register quad t = halo quad ib*8 + 2t + h
    for (int ib = wave; ib < NIB; ib += NW) {
        u32x4 xn[JX];
        const bool more = ib + NW < NIB;
        bool vn = false;
        if (more) { if (abl & 8) { for (int j = 0; j < JX; ++j) xn[j] = xa[j]; vn = va; } else vn = load_x(ib + NW, xn); }
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.0f;
        const char* wb = Wst + lane * 16;
#pragma unroll
        for (int j = 0; j < JX; ++j) {
            const u32x4 wv = ld16(wb + j * 1024);
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, xa[j]),
                                                        __builtin_bit_cast(mfma_bf16x8, wv), a, 0, 0, 0);
        }
        char* ecell = E + (unsigned)(ib * 8 + h) * (unsigned)CP + pl * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x2 u0, u1; u0.x = a[4 * t]; u0.y = a[4 * t + 1]; u1.x = a[4 * t + 2]; u1.y = a[4 * t + 3];
            f32x2 y0 = u0, y1 = u1;
            if (!(abl & 4)) { y0 = swish2_pre(u0); y1 = swish2_pre(u1); }
            u32x2 d;
            d.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y0.x, y0.y));
            d.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y1.x, y1.y));
            *reinterpret_cast<u32x2*>(ecell + 2 * t * CP) = d;
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < JX; ++j) xa[j] = xn[j];
            mask_x(xa, vn);
        }
    }
    __syncthreads();

    // ---- phase 2: depthwise on the matrix cores + Swish, 16 output quads x 32 channels per wave step
    static constexpr SetMap<TOH, TOW, IWQ> kSets{};
    const int kg = lane >> 4;
    for (int set = wave; set < NSET; set += NW) {
        const uint32_t e = kSets.v[set * 16 + (lane & 15)];
        const int oy = (e >> 6) & 0x1ff, oxq = e & 63;
        const bool live = (e & 0x8000u) == 0;
        f32x4 acc[8];
        const char* bb = E + (unsigned)(oy * IWQ + oxq) * (unsigned)CP + kg * 64;
        if (abl & 1) {
#pragma unroll
            for (int g = 0; g < 8; ++g) acc[g] = f32x4{(float)oy, (float)oxq, (float)g, 1.0f};
        } else {
        if constexpr (ALDS) mx_depthwise_lds<KS, IWQ, CP>(bb, Ats + lane * 8, acc);
        else mx_depthwise<KS, IWQ, CP>(bb, reinterpret_cast<const u32x2 (*)[KS][2]>(A), acc);
        }
        // a = -log2(e) * depthwise output -> Swish, leftover factor out again (the project GEMM has plain weights)
        if (!(abl & 2))
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                f32x2 u; u.x = acc[g][i]; u.y = acc[g][i + 1];
                const f32x2 y = swish2_pre(u) * kNegLn23;
                acc[g][i] = y.x; acc[g][i + 1] = y.y;
            }
        const int gy = oy0 + oy, gx0 = ox0 + 4 * oxq;
        if (!live || gy >= p.Hout || ((abl & 16) && acc[0][0] != 123.0f)) continue;
        const size_t opix0 = ((size_t)b * p.Hout + gy) * p.Wout + gx0;
        const int chunk = grp * 4 + kg;                               // 8-channel chunk of the depthwise tensor
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (gx0 + i >= p.Wout) break;
            u32x4 o;
            o.x = packb(acc[0][i], acc[1][i]); o.y = packb(acc[2][i], acc[3][i]);
            o.z = packb(acc[4][i], acc[5][i]); o.w = packb(acc[6][i], acc[7][i]);
            const size_t opix = opix0 + i;
            if (abl & 32) {         // timing experiment (results invalid): the same bytes, 1 KB of consecutive addresses per store instruction
                const size_t region = (((size_t)b * gridDim.x + blockIdx.x) * gridDim.y + grp) * (size_t)(TOH * TOW * 64);
                st16((char*)p.y + region + ((size_t)((set * 4 + i) * 1024 + lane * 16)) % (size_t)(TOH * TOW * 64), o);
                continue;
            }
            st16((char*)p.y + (p.yblock ? blk_off(opix, p.hid / 8, chunk) : (opix * p.hid + (size_t)chunk * 8) * 2), o);
        }
    }
}

#include CF_EXP_INC(cf_mbconv3_0)   // rounds-persistent variant: measured slower (profiles/r03_mfma_depthwise.md), experiments build only

// ================================================================== fully fused block: expand -> depthwise -> project (+residual)
// One workgroup per output tile; the hidden channels go through the tile in rounds of 32 (+ an optional last round of 16:
// hid = 144): expand the halo for the round -> quad cells in LDS -> matrix-core depthwise -> Swish -> the lane's eight (four)
// channels of output pixel i are the B fragment of v_mfma_f32_16x16x32_bf16 (16x16x16 for the 16-channel round) whose n index
// is the lane's quad slot: project accumulators D[out channel][quad slot] per pixel i, in registers across the rounds.
// MFMA row m <-> output channel 32 (mb >> 1) + 8 (m >> 2) + 4 (mb & 1) + (m & 3): lane group kq = lane >> 4 ends up with the
// eight consecutive channels 8 kq .. 8 kq + 7 (+32 for the second pair of M blocks) of its pixels = 16-byte stores.
template <int KS, int JX, int NMB, int TOH, int TOW, int NW, bool TAIL16, bool ALDS = false>
struct Fx {
    static constexpr int IH = TOH + KS - 1, IWQ = TOW / 4 + 1, IWP = IWQ * 4;
    static constexpr int NQD = IH * IWQ, NIB = (NQD + 7) / 8, IPX = NQD * 4, MAXI = (NIB + NW - 1) / NW;
    static constexpr int CP8 = 32 * 8 + 16, CP4 = 16 * 8 + 16;
    static constexpr int EBYTES = NIB * 8 * CP8;
    static constexpr int NOQ = TOH * (TOW / 4), NSET = (NOQ + 15) / 16, SPW = NSET / NW;
    static constexpr int WXB = JX * 1024;
    static constexpr int ATB = 2 * KS * 2 * 512;                    // Toeplitz operand table of one round (LDS copy when ALDS)
    static constexpr int LDS = EBYTES + 2 * WXB + (ALDS ? ATB : 0);
    static_assert(NSET % NW == 0, "every wave owns SPW whole sets (project accumulators live in registers)");
    static_assert(TOW % 4 == 0 && KS <= 5 && (NMB == 2 || NMB == 4 || NMB == 6), "geometry");
    static_assert(!TAIL16 || JX <= 2, "16-channel round: one 16x16x32 expand MFMA, Cin <= 32");
};

template <int KS, int JX, int NMB, bool RESID, int TOH, int TOW, int NW, bool TAIL16, bool XRELOAD, bool ALDS, bool SB = false>
__global__ __launch_bounds__(NW * 64) void mbconv_mx_kernel(MbParams p) {
    typedef Fx<KS, JX, NMB, TOH, TOW, NW, TAIL16, ALDS> G;
    constexpr int IWQ = G::IWQ, IWP = G::IWP, IPX = G::IPX, NIB = G::NIB, MAXI = G::MAXI, CP8 = G::CP8, CP4 = G::CP4;
    constexpr int SPW = G::SPW, WXB = G::WXB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* E = smem;
    char* Wst = smem + G::EBYTES;
    char* Ats = Wst + 2 * G::WXB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5, kg = lane >> 4;
    const int ox0 = blockIdx.x * TOW, oy0 = blockIdx.y * TOH, b = blockIdx.z;
    const int nq = p.nq;                                            // full rounds of 32 hidden channels
#ifdef CF_ABLATION
    const int abl = p.nw;      // timing experiments only (-DCF_ABLATION, CF_FX_ABL): 1 no depthwise MFMAs, 2 no output Swish, 4 no expand Swish, 8 no project MFMAs, 16 no stores
#else
    constexpr int abl = 0;     // (the run-time tests cost the production kernel 3-4 %: compiled out)
#endif

    const char* xbase = (const char*)p.x + (size_t)b * p.Hin * p.Win * p.Cin * 2;
    const unsigned rowbytes = (unsigned)p.Cin * 2;

    auto stage_weights = [&](int q) {
        char* dst = Wst + (q & 1) * WXB;
        const char* srcx = (const char*)p.wexp + (size_t)q * WXB;
        for (int c = wave; c < WXB / 1024; c += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcx + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
    };

    // X fragments of this wave's halo pixel blocks (MFMA A operand: lane = pixel, 8 contiguous Cin per half): resident, or
    // (XRELOAD) fetched again from L2 at the top of every round; clamped address + zero select = ZeroPad2d
    u32x4 xf[MAXI][JX];
    auto load_x = [&]() {
#pragma unroll
    for (int t = 0; t < MAXI; ++t) {
        const int ib = wave + NW * t;
        const int ip = ib * 32 + pl;
        const int ipc = ip < IPX ? ip : IPX - 1;
        const int iy = ipc / IWP, ix = ipc - iy * IWP;
        const int gy = oy0 - p.pad_lo + iy, gx = ox0 - p.pad_lo + ix;
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
    };
    if constexpr (!XRELOAD) load_x();

    // this wave's output quads: one per (set, lane & 15)
    static constexpr SetMap<TOH, TOW, IWQ> kSets{};
    unsigned qcell[SPW];                     // linear halo-quad index of the output quad's top-left input quad
#pragma unroll
    for (int sw = 0; sw < SPW; ++sw) {
        const uint32_t e = kSets.v[(wave * SPW + sw) * 16 + (lane & 15)];
        qcell[sw] = ((e >> 6) & 0x1ff) * IWQ + (e & 63);
    }

    f32x4 pacc[SPW][4][NMB];
#pragma unroll
    for (int sw = 0; sw < SPW; ++sw)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) pacc[sw][i][mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    stage_weights(0);
    asm volatile("" ::: "memory");
    for (int q = 0; q < nq; ++q) {
        const char* wx = Wst + (q & 1) * WXB;
        // Toeplitz operands + project fragments of this round (and the X fragments when reloaded): requested before the barrier,
        // used after it -- they stay in flight across it, only the older expand-weight DMA is drained
        u32x2 A[ALDS ? 1 : 2][ALDS ? 1 : KS][2];
        if constexpr (!ALDS) {
            const u32x2* at = reinterpret_cast<const u32x2*>(p.wdw) + (size_t)q * (2 * KS * 2) * 64 + lane;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) A[a][ky][ks] = at[((a * KS + ky) * 2 + ks) * 64];
        }
        u32x4 wpc[NMB];
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) wpc[mb] = ld16((const char*)p.wproj + (((size_t)q * NMB + mb) * 64 + lane) * 16);
        if constexpr (XRELOAD) load_x();
        cf_sync_lds_dma_keep<(ALDS ? 0 : 2 * KS * 2) + NMB + (XRELOAD ? MAXI * JX : 0)>();   // previous round's depthwise done with E / the table; expand weights landed
        // this round's Toeplitz table -> LDS under the expand phase (drained before the next barrier).  Staging it one round
        // ahead in a second buffer was measured: no gain (0.192 vs 0.189 ms on layer1.1), 6-10 KB more LDS
        if constexpr (ALDS) {
            const char* srca = (const char*)p.wdw + (size_t)q * G::ATB;
            for (int c = wave; c < G::ATB / 1024; c += NW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srca + c * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(Ats + c * 1024), 16, 0, 0);
        }
        const char* Atq = Ats;

        // ---- phase 1: expand + Swish -> quad cells
#pragma unroll
        for (int t = 0; t < MAXI; ++t) {
            const int ib = wave + NW * t;
            if (ib >= NIB) break;
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.0f;
#pragma unroll
            for (int j = 0; j < JX; ++j) {
                const u32x4 wv = ld16(wx + (j * 64 + lane) * 16);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, xf[t][j]),
                                                            __builtin_bit_cast(mfma_bf16x8, wv), a, 0, 0, 0);
            }
            char* ecell = E + (unsigned)(ib * 8 + h) * (unsigned)CP8 + pl * 8;
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                f32x2 u0, u1; u0.x = a[4 * tq]; u0.y = a[4 * tq + 1]; u1.x = a[4 * tq + 2]; u1.y = a[4 * tq + 3];
                f32x2 y0 = u0, y1 = u1;
                if (!(abl & 4)) { y0 = swish2_pre(u0); y1 = swish2_pre(u1); }
                u32x2 d;
                d.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y0.x, y0.y));
                d.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y1.x, y1.y));
                *reinterpret_cast<u32x2*>(ecell + 2 * tq * CP8) = d;
            }
        }
        cf_sync_lds_dma();          // E complete; the operand table (DMA) landed
        if (q + 1 < nq) stage_weights(q + 1);
        asm volatile("" ::: "memory");        // later loads stay behind the DMA in program order (cf_sync_lds_dma_keep counts on it)

        // ---- phase 2: depthwise (matrix cores) + Swish + project
#pragma unroll
        for (int sw = 0; sw < SPW; ++sw) {
            f32x4 acc[8];
            if (abl & 1) {
#pragma unroll
                for (int g = 0; g < 8; ++g) acc[g] = f32x4{(float)q, (float)g, 1.0f, 2.0f};
            } else {
            if constexpr (ALDS && SB) mx_depthwise_lds1<KS, IWQ, CP8>(E + qcell[sw] * (unsigned)CP8 + kg * 64, Atq + lane * 8, acc);
            else if constexpr (ALDS) mx_depthwise_lds<KS, IWQ, CP8>(E + qcell[sw] * (unsigned)CP8 + kg * 64, Atq + lane * 8, acc);
            else mx_depthwise<KS, IWQ, CP8>(E + qcell[sw] * (unsigned)CP8 + kg * 64, reinterpret_cast<const u32x2 (*)[KS][2]>(A), acc);
            }
            if (!(abl & 2))
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    f32x2 u; u.x = acc[g][i]; u.y = acc[g][i + 1];
                    const f32x2 y = swish2_pre(u);
                    acc[g][i] = y.x; acc[g][i + 1] = y.y;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4 d;
                d.x = packb(acc[0][i], acc[1][i]); d.y = packb(acc[2][i], acc[3][i]);
                d.z = packb(acc[4][i], acc[5][i]); d.w = packb(acc[6][i], acc[7][i]);
                if (abl & 8) { pacc[sw][i][0][0] += __uint_as_float(d.x ^ d.y ^ d.z ^ d.w); continue; }
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb)
                    pacc[sw][i][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, wpc[mb]),
                                                                             __builtin_bit_cast(mfma_bf16x8, d), pacc[sw][i][mb], 0, 0, 0);
            }
        }
    }

    if constexpr (TAIL16) {
        // ---- last round: 16 hidden channels (4 groups x 4).  Expand on v_mfma_f32_16x16x32_bf16: D[16 pixels][16 channels],
        // lane (n = channel, qd = lane >> 4) holds quad qd of the 16-pixel group -> one 8-byte cell; cell row 144 B
        typedef __attribute__((ext_vector_type(4))) __bf16 mfma_bf16x4;
        const u32x4 wv = ld16((const char*)p.wexp + (size_t)nq * WXB + lane * 16);
        u32x2 A4[KS][2];
        {
            const u32x2* at = reinterpret_cast<const u32x2*>(p.wdw) + (size_t)nq * (2 * KS * 2) * 64 + lane;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) A4[ky][ks] = at[(ky * 2 + ks) * 64];
        }
        u32x2 wp4[NMB];
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
            wp4[mb] = *reinterpret_cast<const u32x2*>((const char*)p.wproj + (size_t)nq * NMB * 1024 + ((size_t)mb * 64 + lane) * 8);
        __syncthreads();
        for (int sb = wave; sb < NIB * 2; sb += NW) {               // 16-pixel sub-blocks = 4 quads each
            const int ip = sb * 16 + (lane & 15), kc = lane >> 4;   // A operand: lane (pixel, Cin chunk)
            const int ipc = ip < IPX ? ip : IPX - 1;
            const int iy = ipc / IWP, ix = ipc - iy * IWP;
            const int gy = oy0 - p.pad_lo + iy, gx = ox0 - p.pad_lo + ix;
            const bool valid = ip < IPX && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win && kc * 8 < p.Cin;
            const int cy = min(max(gy, 0), p.Hin - 1), cx = min(max(gx, 0), p.Win - 1);
            u32x4 xv = ld16(xbase + ((unsigned)cy * (unsigned)p.Win + (unsigned)cx) * rowbytes + (unsigned)(min(kc * 8, p.Cin - 8) * 2));
            xv.x = valid ? xv.x : 0u; xv.y = valid ? xv.y : 0u; xv.z = valid ? xv.z : 0u; xv.w = valid ? xv.w : 0u;
            f32x4 a4 = {0.0f, 0.0f, 0.0f, 0.0f};
            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, xv), __builtin_bit_cast(mfma_bf16x8, wv), a4, 0, 0, 0);
            f32x2 u0, u1; u0.x = a4[0]; u0.y = a4[1]; u1.x = a4[2]; u1.y = a4[3];
            const f32x2 y0 = swish2_pre(u0), y1 = swish2_pre(u1);
            u32x2 d;
            d.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y0.x, y0.y));
            d.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y1.x, y1.y));
            *reinterpret_cast<u32x2*>(E + (unsigned)(sb * 4 + kc) * (unsigned)CP4 + (lane & 15) * 8) = d;
        }
        __syncthreads();
#pragma unroll
        for (int sw = 0; sw < SPW; ++sw) {
            const char* bb = E + qcell[sw] * (unsigned)CP4 + kg * 32;
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int st = 0; st < KS * 2; ++st) {
                const char* bs = bb + ((st >> 1) * IWQ + (st & 1)) * CP4;
                const u32x4 b0 = ld16(bs), b1 = ld16(bs + 16);
                const mfma_f16x4 av = __builtin_bit_cast(mfma_f16x4, A4[st >> 1][st & 1]);
                u32x2 t0, t1, t2, t3; t0.x = b0.x; t0.y = b0.y; t1.x = b0.z; t1.y = b0.w; t2.x = b1.x; t2.y = b1.y; t3.x = b1.z; t3.y = b1.w;
                CF_MX_MFMA(acc[0], av, __builtin_bit_cast(mfma_f16x4, t0), 0);
                CF_MX_MFMA(acc[1], av, __builtin_bit_cast(mfma_f16x4, t1), 1);
                CF_MX_MFMA(acc[2], av, __builtin_bit_cast(mfma_f16x4, t2), 2);
                CF_MX_MFMA(acc[3], av, __builtin_bit_cast(mfma_f16x4, t3), 3);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    f32x2 u; u.x = acc[g][i]; u.y = acc[g][i + 1];
                    const f32x2 y = swish2_pre(u);
                    acc[g][i] = y.x; acc[g][i + 1] = y.y;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x2 d; d.x = packb(acc[0][i], acc[1][i]); d.y = packb(acc[2][i], acc[3][i]);
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb)
                    pacc[sw][i][mb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(mfma_bf16x4, wp4[mb]),
                                                                                __builtin_bit_cast(mfma_bf16x4, d), pacc[sw][i][mb], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane (kq = lane >> 4, quad slot): channels 8 kq .. + 7 (+ 32) of the four pixels of its quad
#pragma unroll
    for (int sw = 0; sw < SPW; ++sw) {
        const uint32_t e = kSets.v[(wave * SPW + sw) * 16 + (lane & 15)];
        const int oy = (e >> 6) & 0x1ff, oxq = e & 63;
        const int gy = oy0 + oy, gx0 = ox0 + 4 * oxq;
        if ((e & 0x8000u) || gy >= p.Hout || ((abl & 16) && pacc[sw][0][0][0] != 123.0f)) continue;
        const size_t opix0 = ((size_t)b * p.Hout + gy) * p.Wout + gx0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (gx0 + i >= p.Wout) break;
            const size_t opix = opix0 + i;
#pragma unroll
            for (int mp = 0; mp < NMB / 2; ++mp) {
                const int ch = mp * 32 + kg * 8;
                if (ch >= p.Cout) break;
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = pacc[sw][i][2 * mp][r]; v[4 + r] = pacc[sw][i][2 * mp + 1][r]; }
                if constexpr (RESID) {
                    float rr[8];
                    unpack16<bf16_t>(ld16((const char*)p.x + (opix * p.Cin + ch) * 2), rr);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = rr[r] + v[r];
                }
                st16((char*)p.y + (p.yblock ? blk_off(opix, p.Cout / 8, ch / 8) : (opix * p.Cout + ch) * 2), pack16b(v));
            }
        }
    }
}

#include CF_EXP_INC(cf_mbconv3_1)   // role-specialised waves: measured slower, experiments build only

// ================================================================== fully fused block, stride 2 (layer1.0, layer2.0)
// The input halo of a stride-2 tile is four times the output tile, so the LDS budget allows 32 output quads = two sets per
// workgroup.  Four waves = (set, channel half): every wave runs the depthwise of ITS four channels per group (4 x KS x 3
// MFMAs per round: an output quad reads inputs x .. x+10 = three k-steps), Swish, and projects them with
// v_mfma_f32_16x16x16_bf16 (K = 16 = its 4 channels x 4 groups) into its own partial accumulators; the two halves of a set
// are added through LDS once at the end.  Everything else as mbconv_mx_kernel.  The operand table always lives in LDS.
template <int KS, int JX, int NMB, int TOH, int TOW, bool TAIL16, bool ALDS = true>
struct Fs {
    static constexpr int NW = 4, KSTEPS = 3, NSTEP = KS * KSTEPS;
    static constexpr int IH = (TOH - 1) * 2 + KS, IW0 = (TOW - 1) * 2 + KS, IWQ = (IW0 + 3) / 4, IWP = IWQ * 4;
    static constexpr int NQD = IH * IWQ, NIB = (NQD + 7) / 8, IPX = NQD * 4, MAXI = (NIB + NW - 1) / NW;
    static constexpr int CP8 = 32 * 8 + 16, CP4 = 16 * 8 + 16;
    static constexpr int EBYTES = NIB * 8 * CP8;
    static constexpr int NOQ = TOH * (TOW / 4);
    static constexpr int WXB = JX * 1024, ATB = 2 * NSTEP * 512;
    static constexpr int RED = 2 * 64 * 4 * NMB * 16;                  // partial accumulators of the two half-1 waves
    static constexpr int LDS = (EBYTES + 2 * WXB + (ALDS ? ATB : 0)) > RED ? (EBYTES + 2 * WXB + (ALDS ? ATB : 0)) : RED;
    static_assert(NOQ == 32 && TOW % 4 == 0 && KS <= 5, "two sets of 16 output quads per tile");
    static_assert(!TAIL16 || JX <= 2, "16-channel round: one 16x16x32 expand MFMA, Cin <= 32");
};

template <int KS, int JX, int NMB, int TOH, int TOW, bool TAIL16, bool XRELOAD, bool ALDS>
__global__ __launch_bounds__(256) void mbconv_mx2_kernel(MbParams p) {
    typedef Fs<KS, JX, NMB, TOH, TOW, TAIL16, ALDS> G;
    constexpr int IWQ = G::IWQ, IWP = G::IWP, IPX = G::IPX, NIB = G::NIB, MAXI = G::MAXI, CP8 = G::CP8, CP4 = G::CP4;
    constexpr int WXB = G::WXB, NW = 4, KSTEPS = 3, NSTEP = G::NSTEP, HQ = (IWQ + 1) / 2;
    typedef __attribute__((ext_vector_type(4))) __bf16 mfma_bf16x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* E = smem;
    char* Wst = smem + G::EBYTES;
    char* Ats = Wst + 2 * WXB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = wave & 1, half = wave >> 1;
    const int pl = lane & 31, h = lane >> 5, kg = lane >> 4;
    const int ox0 = blockIdx.x * TOW, oy0 = blockIdx.y * TOH, b = blockIdx.z;
    const int nq = p.nq;

    const char* xbase = (const char*)p.x + (size_t)b * p.Hin * p.Win * p.Cin * 2;
    const unsigned rowbytes = (unsigned)p.Cin * 2;

    auto stage_weights = [&](int q) {
        char* dst = Wst + (q & 1) * WXB;
        const char* srcx = (const char*)p.wexp + (size_t)q * WXB;
        for (int c = wave; c < WXB / 1024; c += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcx + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
    };
    u32x4 xf[MAXI][JX];
    auto load_x = [&]() {
#pragma unroll
    for (int t = 0; t < MAXI; ++t) {
        const int ib = wave + NW * t;
        const int ip = ib * 32 + pl;
        const int ipc = ip < IPX ? ip : IPX - 1;
        const int cc = ipc >> 2, iy = cc / IWQ, pos = cc - iy * IWQ;                 // cell position -> x-quad: even quads first, then the odd ones
        const int ix = 4 * (pos < HQ ? 2 * pos : 2 * (pos - HQ) + 1) + (ipc & 3);
        const int gy = oy0 * 2 - p.pad_lo + iy, gx = ox0 * 2 - p.pad_lo + ix;
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
    };
    if constexpr (!XRELOAD) load_x();

    static constexpr SetMap<TOH, TOW, IWQ, 2> kSets{};
    const uint32_t se = kSets.v[set * 16 + (lane & 15)];
    const int soy = (se >> 6) & 0x1ff, soxq = se & 63;
    // a row of cells holds its even x-quads first, then the odd ones: the three input quads 2 oxq + ks of an output quad sit at
    // oxq, HQ + oxq, oxq + 1, and the 16 lanes of a ds_read_b128 group (cells 2 oy IWQ + oxq) can cover 16 distinct bank slots --
    // with the plain row order every first cell is even and each read is a two-way conflict (0.63 conflict cycles per active cycle)
    const unsigned qcell = (unsigned)(2 * soy * IWQ + soxq);

    f32x4 pacc[4][NMB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) pacc[i][mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    stage_weights(0);
    asm volatile("" ::: "memory");
    for (int q = 0; q < nq; ++q) {
        const char* wx = Wst + (q & 1) * WXB;
        // this half's Toeplitz operands + project fragments (+ X fragments when reloaded): requested before the barrier, in flight across it
        u32x2 Ah[ALDS ? 1 : KS][KSTEPS];
        if constexpr (!ALDS) {
            const u32x2* at = reinterpret_cast<const u32x2*>(p.wdw) + ((size_t)q * 2 + half) * NSTEP * 64 + lane;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) Ah[ky][ks] = at[(ky * KSTEPS + ks) * 64];
        }
        u32x2 wp4[NMB];      // project fragments of this wave's 16 channels (4 per group): A operand of 16x16x16, 8 bytes per lane
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
            wp4[mb] = *reinterpret_cast<const u32x2*>((const char*)p.wproj + ((((size_t)q * 2 + half) * NMB + mb) * 64 + lane) * 8);
        if constexpr (XRELOAD) load_x();
        cf_sync_lds_dma_keep<(ALDS ? 0 : NSTEP) + NMB + (XRELOAD ? MAXI * JX : 0)>();   // previous round's depthwise done with E / the table; expand weights landed
        if constexpr (ALDS) {
            const char* srca = (const char*)p.wdw + (size_t)q * G::ATB;
            for (int c = wave; c < G::ATB / 1024; c += NW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srca + c * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(Ats + c * 1024), 16, 0, 0);
        }

        // ---- phase 1: expand + Swish -> quad cells
#pragma unroll
        for (int t = 0; t < MAXI; ++t) {
            const int ib = wave + NW * t;
            if (ib >= NIB) break;
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.0f;
#pragma unroll
            for (int j = 0; j < JX; ++j) {
                const u32x4 wv = ld16(wx + (j * 64 + lane) * 16);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, xf[t][j]),
                                                            __builtin_bit_cast(mfma_bf16x8, wv), a, 0, 0, 0);
            }
            char* ecell = E + (unsigned)(ib * 8 + h) * (unsigned)CP8 + pl * 8;
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                f32x2 u0, u1; u0.x = a[4 * tq]; u0.y = a[4 * tq + 1]; u1.x = a[4 * tq + 2]; u1.y = a[4 * tq + 3];
                const f32x2 y0 = swish2_pre(u0), y1 = swish2_pre(u1);
                u32x2 d;
                d.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y0.x, y0.y));
                d.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y1.x, y1.y));
                *reinterpret_cast<u32x2*>(ecell + 2 * tq * CP8) = d;
            }
        }
        cf_sync_lds_dma();          // E complete; the operand table landed
        if (q + 1 < nq) stage_weights(q + 1);
        asm volatile("" ::: "memory");        // later loads stay behind the DMA in program order (cf_sync_lds_dma_keep counts on it)

        // ---- phase 2: this wave's (set, channel half): depthwise + Swish + project
        {
            f32x4 acc[4];
            if constexpr (ALDS) mx_depthwise_half<KS, KSTEPS, IWQ, CP8>(E + qcell * (unsigned)CP8 + kg * 64 + half * 32, Ats + half * (NSTEP * 512) + lane * 8, acc);
            else mx_depthwise_half_reg<KS, KSTEPS, IWQ, CP8>(E + qcell * (unsigned)CP8 + kg * 64 + half * 32, reinterpret_cast<const u32x2 (*)[KSTEPS]>(Ah), acc);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    f32x2 u; u.x = acc[g][i]; u.y = acc[g][i + 1];
                    const f32x2 y = swish2_pre(u);
                    acc[g][i] = y.x; acc[g][i + 1] = y.y;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x2 d; d.x = packb(acc[0][i], acc[1][i]); d.y = packb(acc[2][i], acc[3][i]);
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb)
                    pacc[i][mb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(mfma_bf16x4, wp4[mb]),
                                                                            __builtin_bit_cast(mfma_bf16x4, d), pacc[i][mb], 0, 0, 0);
            }
        }
    }

    if constexpr (TAIL16) {
        // ---- last round: 16 hidden channels (4 groups x 4), done by the half-0 wave of each set (1 / 9 of the work at hid = 144)
        const u32x4 wv = ld16((const char*)p.wexp + (size_t)nq * WXB + lane * 16);
        u32x2 A4[KS][KSTEPS];
        {
            const u32x2* at = reinterpret_cast<const u32x2*>(p.wdw) + (size_t)nq * (2 * NSTEP) * 64 + lane;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) A4[ky][ks] = at[(ky * KSTEPS + ks) * 64];
        }
        u32x2 wp4[NMB];
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
            wp4[mb] = *reinterpret_cast<const u32x2*>((const char*)p.wproj + (size_t)nq * 2 * NMB * 512 + ((size_t)mb * 64 + lane) * 8);
        __syncthreads();
        for (int sb = wave; sb < NIB * 2; sb += NW) {               // 16-pixel sub-blocks = 4 quads each
            const int ip = sb * 16 + (lane & 15), kc = lane >> 4;
            const int ipc = ip < IPX ? ip : IPX - 1;
            const int cc = ipc >> 2, iy = cc / IWQ, pos = cc - iy * IWQ;
            const int ix = 4 * (pos < HQ ? 2 * pos : 2 * (pos - HQ) + 1) + (ipc & 3);
            const int gy = oy0 * 2 - p.pad_lo + iy, gx = ox0 * 2 - p.pad_lo + ix;
            const bool valid = ip < IPX && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win && kc * 8 < p.Cin;
            const int cy = min(max(gy, 0), p.Hin - 1), cx = min(max(gx, 0), p.Win - 1);
            u32x4 xv = ld16(xbase + ((unsigned)cy * (unsigned)p.Win + (unsigned)cx) * rowbytes + (unsigned)(min(kc * 8, p.Cin - 8) * 2));
            xv.x = valid ? xv.x : 0u; xv.y = valid ? xv.y : 0u; xv.z = valid ? xv.z : 0u; xv.w = valid ? xv.w : 0u;
            f32x4 a4 = {0.0f, 0.0f, 0.0f, 0.0f};
            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, xv), __builtin_bit_cast(mfma_bf16x8, wv), a4, 0, 0, 0);
            f32x2 u0, u1; u0.x = a4[0]; u0.y = a4[1]; u1.x = a4[2]; u1.y = a4[3];
            const f32x2 y0 = swish2_pre(u0), y1 = swish2_pre(u1);
            u32x2 d;
            d.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y0.x, y0.y));
            d.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y1.x, y1.y));
            *reinterpret_cast<u32x2*>(E + (unsigned)(sb * 4 + kc) * (unsigned)CP4 + (lane & 15) * 8) = d;
        }
        __syncthreads();
        if (half == 0) {
            const char* bb = E + qcell * (unsigned)CP4 + kg * 32;
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                const char* bs = bb + ((st / KSTEPS) * IWQ + ((st % KSTEPS) & 1) * HQ + ((st % KSTEPS) >> 1)) * CP4;
                const u32x4 b0 = ld16(bs), b1 = ld16(bs + 16);
                const mfma_f16x4 av = __builtin_bit_cast(mfma_f16x4, A4[st / KSTEPS][st % KSTEPS]);
                u32x2 t0, t1, t2, t3; t0.x = b0.x; t0.y = b0.y; t1.x = b0.z; t1.y = b0.w; t2.x = b1.x; t2.y = b1.y; t3.x = b1.z; t3.y = b1.w;
                CF_MX_MFMA(acc[0], av, __builtin_bit_cast(mfma_f16x4, t0), 0);
                CF_MX_MFMA(acc[1], av, __builtin_bit_cast(mfma_f16x4, t1), 1);
                CF_MX_MFMA(acc[2], av, __builtin_bit_cast(mfma_f16x4, t2), 2);
                CF_MX_MFMA(acc[3], av, __builtin_bit_cast(mfma_f16x4, t3), 3);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    f32x2 u; u.x = acc[g][i]; u.y = acc[g][i + 1];
                    const f32x2 y = swish2_pre(u);
                    acc[g][i] = y.x; acc[g][i + 1] = y.y;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x2 d; d.x = packb(acc[0][i], acc[1][i]); d.y = packb(acc[2][i], acc[3][i]);
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb)
                    pacc[i][mb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(mfma_bf16x4, wp4[mb]),
                                                                            __builtin_bit_cast(mfma_bf16x4, d), pacc[i][mb], 0, 0, 0);
            }
        }
    }

    // ---- add the two channel halves of every set through LDS, then the epilogue by the half-0 waves
    __syncthreads();
    {
        f32x4* red = reinterpret_cast<f32x4*>(smem) + ((size_t)set * 64 + lane) * (4 * NMB);
        if (half == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb) red[i * NMB + mb] = pacc[i][mb];
        }
        __syncthreads();
        if (half == 1) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) pacc[i][mb] += red[i * NMB + mb];
    }
    const int gy = oy0 + soy, gx0 = ox0 + 4 * soxq;
    if ((se & 0x8000u) || gy >= p.Hout) return;
    const size_t opix0 = ((size_t)b * p.Hout + gy) * p.Wout + gx0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (gx0 + i >= p.Wout) break;
        const size_t opix = opix0 + i;
#pragma unroll
        for (int mp = 0; mp < NMB / 2; ++mp) {
            const int ch = mp * 32 + kg * 8;
            if (ch >= p.Cout) break;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = pacc[i][2 * mp][r]; v[4 + r] = pacc[i][2 * mp + 1][r]; }
            st16((char*)p.y + (p.yblock ? blk_off(opix, p.Cout / 8, ch / 8) : (opix * p.Cout + ch) * 2), pack16b(v));
        }
    }
}
#ifndef CF_ILP_TU        // expand+depthwise host side: main translation unit only

// ---------------------------------------------------------------- host side
struct MxEntry {
    int k, jx, toh, tow, nw, var, lds_bytes;
    hipError_t (*fn)(hipStream_t, const MbParams&);
};
template <int KS, int JX, int TOH, int TOW, int NW, bool ALDS>
static hipError_t xmx_launch_t(hipStream_t s, const MbParams& p) {
    typedef Mx<KS, JX, TOH, TOW, NW, ALDS> G;
    auto kfn = expdw_mx_kernel<KS, JX, TOH, TOW, NW, ALDS>;
    static thread_local bool configured_dev[32] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    bool& configured = configured_dev[dev & 31];
    if (G::LDS > 64 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    dim3 grid(((p.Wout + TOW - 1) / TOW) * ((p.Hout + TOH - 1) / TOH), p.hid / 32, p.B), blk(NW * 64);
    set_kernel_tag("void cf::expdw_mx_kernel<%d, %d, %d, %d, %d, %s>(cf::MbParams)", KS, JX, TOH, TOW, NW, ALDS ? "true" : "false");
    static const int abl = cf_ab_int("CF_MX_ABL", 0);      // timing experiments only: results invalid
    MbParams q = p; q.nw = abl;
    hipLaunchKernelGGL(kfn, grid, blk, G::LDS, s, q);
    return hipGetLastError();
}
#include CF_EXP_INC(cf_mbconv3_2)
#include CF_EXP_INC(cf_mbconv3_3)
#define XMT(V, KS, JX, TOH, TOW, NW, T) {KS, JX, TOH, TOW, NW, V, Mx<KS, JX, TOH, TOW, NW, true>::LDS, &xmxt_launch_t<KS, JX, TOH, TOW, NW, T>}
#define XMR(V, KS, JX, TOH, TOW, NW, R) {KS, JX, TOH, TOW, NW, V, Mx<KS, JX, TOH, TOW, NW, true>::EBYTES + 2 * (Mx<KS, JX, TOH, TOW, NW, true>::WXB + Mx<KS, JX, TOH, TOW, NW, true>::ATB), &xmxr_launch_t<KS, JX, TOH, TOW, NW, R>}
#define XMX(V, KS, JX, TOH, TOW, NW, AL) {KS, JX, TOH, TOW, NW, V, Mx<KS, JX, TOH, TOW, NW, (AL != 0)>::LDS, &xmx_launch_t<KS, JX, TOH, TOW, NW, (AL != 0)>}
static const MxEntry kXmxTable[] = {
    //  var KS JX  tile   waves  A in LDS          (B = 64, 640x640, HIP events; expdw_px_kernel of the layer in brackets)
    XMX(0, 5, 4, 10, 40, 8, 1),      // 4.0   64 -> 384, 40x40: 0.055 ms [0.066]
    XMX(0, 5, 6, 10, 40, 8, 1),      // 4.1   96 -> 576, 40x40: 0.083 ms [0.101]
    XMX(0, 5, 10, 20, 20, 4, 0),     // 5.1  160 -> 960, 20x20: 0.047 ms [0.057]
    XMX(0, 3, 10, 10, 20, 4, 1),     // 6.0  160 -> 960, 20x20: 0.042 ms [0.044]
#include CF_EXP_INC(cf_mbconv3_4)   // never-default variants: A/B runs of an experiments build only
};
#undef XMT
#undef XMX
#undef XMR
static const MxEntry* xmx_find(int k, int jx) {
    static const int want = cf_ab_int("CF_MX_VARIANT", 0);
    const MxEntry* base = nullptr;
    for (const MxEntry& e : kXmxTable)
        if (e.k == k && e.jx == jx) {
            if (e.var == want) return &e;
            if (e.var == 0) base = &e;
        }
    return base;
}

// geometry of the matrix-core expand+depthwise kernel (MbGeom::kind = 4); stride 1 only
MbGeom expdw_mx_geometry(int dtype, int Cin, int hid, int k, int s) {
    MbGeom g{};
    static const bool off = cf_env_int("CF_DW_MATRIX", 1) == 0 || cf_ab_int("CF_MX", 1) == 0;      // product switch: the v_dot2c family instead
    if (off || dtype != 1 || s != 1 || (Cin % 8) || (hid % 32) || hid == Cin) return g;
    const int jx = (Cin * 2 / 16 + 1) / 2;
    const MxEntry* e = xmx_find(k, jx);
    if (!e) return g;
    g.ok = true; g.kind = 4; g.S = 1;
    g.JX = jx; g.NBO = 0; g.HC = 32; g.nq = hid / 32; g.NBE = 1; g.HALF = 0;
    g.rowb = 32 * 8 + 16;
    g.lds_bytes = (size_t)e->lds_bytes;
    g.wexp_bytes = (size_t)g.nq * g.JX * 64 * 16;
    g.wdw_floats = (size_t)g.nq * 2 * k * 2 * 64 * 2;              // dwords: [round][2][k][2][64 lanes] x 2
    g.wproj_bytes = 0;
    return g;
}

// Toeplitz A operands: [round][channel quad q][ky][k-step][lane (kg, pg, i)] = 4 fp16: tap w[ky][4 ks + k - i] of channel
// round*32 + kg*8 + q*4 + pg (MFMA abid = pg selects it for the whole group kg)
void mx_pack_taps(int nq, int k, const float* wd /*[hid][k*k]*/, uint32_t* out, int stride, int ksteps) {
    for (int r = 0; r < nq; ++r)
        for (int q = 0; q < 2; ++q)
            for (int ky = 0; ky < k; ++ky)
                for (int ks = 0; ks < ksteps; ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int kg = lane >> 4, pg = (lane >> 2) & 3, i = lane & 3;
                        const float* wrow = wd + (size_t)(r * 32 + kg * 8 + q * 4 + pg) * k * k + (size_t)ky * k;
                        uint16_t v[4];
                        for (int kk = 0; kk < 4; ++kk) {
                            const int kx = 4 * ks + kk - stride * i;
                            v[kk] = (kx >= 0 && kx < k) ? host_f32_to_f16_3(wrow[kx]) : 0;
                        }
                        uint32_t* dst = out + ((((size_t)(r * 2 + q) * k + ky) * ksteps + ks) * 64 + lane) * 2;
                        dst[0] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
                        dst[1] = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
                    }
}

// we [hid][Cin], wd [hid][k*k]; expand fragments as cf_mbconv2.hip (MFMA B operand: lane (n = channel, half) holds Cin chunk)
void mx_pack_weights(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
                     void* wexp_host, float* wdw_host, void* wproj_host) {
    (void)hid; (void)Cout; (void)wp; (void)wproj_host;
    const int NCx = Cin * 2 / 16;
    __builtin_memset(wexp_host, 0, g.wexp_bytes);
    for (int q = 0; q < g.nq; ++q)
        for (int j = 0; j < g.JX; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 31, hh = lane >> 5, c = hh * g.JX + j;
                if (c >= NCx) continue;
                uint16_t* dst = (uint16_t*)((char*)wexp_host + (((size_t)q * g.JX + j) * 64 + lane) * 16);
                for (int e = 0; e < 8; ++e) dst[e] = host_f32_to_bf16(kNegLog2e3 * we[(size_t)(q * 32 + n) * Cin + (size_t)c * 8 + e]);
            }
    mx_pack_taps(g.nq, k, wd, reinterpret_cast<uint32_t*>(wdw_host));
}

hipError_t mx_launch(hipStream_t s, const MbParams& p) {
    const MxEntry* e = xmx_find(p.k, p.JX);
    if (!e || p.s != 1) return hipErrorInvalidValue;
    return e->fn(s, p);
}


// ---------------------------------------------------------------- fused block, host side (MbGeom::kind = 5)
#endif  // !CF_ILP_TU
struct FxEntry {
    int k, jx, nmb, res, tail, toh, tow, nw, var, lds_bytes;
    hipError_t (*fn)(hipStream_t, const MbParams&);
};
template <int KS, int JX, int NMB, bool RESID, int TOH, int TOW, int NW, bool TAIL16, bool XRELOAD, bool ALDS, bool SB = false>
hipError_t fx_launch_t(hipStream_t s, const MbParams& p) {
    typedef Fx<KS, JX, NMB, TOH, TOW, NW, TAIL16, ALDS> G;
    auto kfn = mbconv_mx_kernel<KS, JX, NMB, RESID, TOH, TOW, NW, TAIL16, XRELOAD, ALDS, SB>;
    static thread_local bool configured_dev[32] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    bool& configured = configured_dev[dev & 31];
    if (G::LDS > 64 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    dim3 grid((p.Wout + TOW - 1) / TOW, (p.Hout + TOH - 1) / TOH, p.B), blk(NW * 64);
    static const int abl = cf_ab_int("CF_FX_ABL", 0);      // timing experiments only: results invalid
    MbParams q = p; q.nw = abl;
    set_kernel_tag(SB ? "void cf::mbconv_mx_kernel<%d, %d, %d, %s, %d, %d, %d, %s, %s, %s, true>(cf::MbParams)"
                      : "void cf::mbconv_mx_kernel<%d, %d, %d, %s, %d, %d, %d, %s, %s, %s, false>(cf::MbParams)", KS, JX, NMB, RESID ? "true" : "false",
                   TOH, TOW, NW, TAIL16 ? "true" : "false", XRELOAD ? "true" : "false", ALDS ? "true" : "false");
    hipLaunchKernelGGL(kfn, grid, blk, G::LDS, s, q);
    return hipGetLastError();
}
// layer2.1 is compiled in its own translation unit (cf_mbconv3_ilp.hip = this file with CF_ILP_TU defined) under
// `-mllvm -amdgpu-sched-strategy=max-ilp`: 0.0806 -> 0.0792 ms (same box, two runs each); layer1.1 is unchanged by it (0.1799 / 0.1805) and
// the same option costs mbconv_mx2_kernel a wave per SIMD (0.140 -> 0.159), and the option is per translation unit
#define CF_FX_ILP_INSTANCES(X) \
    X(5, 2, 2, true, 16, 16, 4, false, false, true)  /* layer2.1 */
#ifdef CF_ILP_TU
#define CF_X(KS, JX, NMB, RES, TOH, TOW, NW, TAIL, XR, AL) template hipError_t fx_launch_t<KS, JX, NMB, RES, TOH, TOW, NW, TAIL, XR, AL, false>(hipStream_t, const MbParams&);
CF_FX_ILP_INSTANCES(CF_X)
#undef CF_X
#else
#define CF_X(KS, JX, NMB, RES, TOH, TOW, NW, TAIL, XR, AL) extern template hipError_t fx_launch_t<KS, JX, NMB, RES, TOH, TOW, NW, TAIL, XR, AL, false>(hipStream_t, const MbParams&);
CF_FX_ILP_INSTANCES(CF_X)
#undef CF_X
#endif

#ifndef CF_ILP_TU
#include CF_EXP_INC(cf_mbconv3_5)
#define FXE(V, KS, JX, NMB, RES, TAIL, TOH, TOW, NW, XR, AL) \
    {KS, JX, NMB, RES, TAIL, TOH, TOW, NW, V, Fx<KS, JX, NMB, TOH, TOW, NW, (TAIL != 0), (AL != 0)>::LDS, \
     &fx_launch_t<KS, JX, NMB, (RES != 0), TOH, TOW, NW, (TAIL != 0), (XR != 0), (AL != 0)>}
static const FxEntry kFxTable[] = {
    //  var KS JX NMB res tail  tile   waves  X reload  A in LDS        (B = 64, 640x640, HIP events; cf_mbconv2.hip kernel of the layer in brackets)
    FXE(0, 5, 2, 2, 1, 0, 16, 16, 4, 0, 1),     // 2.1  32 -> 192 -> 32 (+res), 80x80: 0.076 ms [0.120]
    FXE(0, 3, 2, 2, 1, 1, 16, 16, 4, 0, 1),     // 1.1  24 -> 144 -> 24 (+res), 160x160, four rounds of 32 + one of 16: 0.175 ms [0.205]
    // variants for A/B runs (CF_FX_VARIANT=n); 3.1 (64 -> 384 -> 64, 40x40) stays on cf_mbconv2.hip: 0.097-0.106 ms here against 0.063,
    // and the Cout = 96 blocks (4.0 / 4.1) stay split (fused here: 0.115 / 0.210 ms against 0.082 / 0.120 for the two launches)
#include CF_EXP_INC(cf_mbconv3_6)   // never-default variants: A/B runs of an experiments build only
};
#undef FXE
static const FxEntry* fx_find(int k, int jx, int nmb, int res, int tail) {
    static const int want = cf_ab_int("CF_FX_VARIANT", 0);
    const FxEntry* base = nullptr;
    for (const FxEntry& e : kFxTable)
        if (e.k == k && e.jx == jx && e.nmb == nmb && e.res == res && e.tail == tail) {
            if (e.var == want) return &e;
            if (e.var == 0) base = &e;
        }
    return base;
}

bool mx_fused_geometry(MbGeom& g, int Cin, int hid, int Cout, int k, int s) {
    static const bool off = cf_env_int("CF_DW_MATRIX", 1) == 0 || cf_ab_int("CF_FX", 1) == 0;
    if (off || s != 1 || (Cin % 8) || (Cout % 8) || Cout > 96 || (hid % 16) || hid == Cin) return false;
    const int jx = (Cin * 2 / 16 + 1) / 2, nmb = 2 * ((Cout + 31) / 32), tail = (hid % 32) ? 1 : 0;
    const FxEntry* e = fx_find(k, jx, nmb, (Cin == Cout) ? 1 : 0, tail);
    if (!e) return false;
    g = MbGeom{};
    g.ok = true; g.kind = 5; g.S = 1;
    g.JX = jx; g.NBO = nmb; g.HC = 32; g.nq = hid / 32; g.NBE = 1; g.HALF = tail;
    g.rowb = 32 * 8 + 16;
    g.lds_bytes = (size_t)e->lds_bytes;
    g.wexp_bytes = (size_t)g.nq * g.JX * 1024 + (tail ? 1024 : 0);
    g.wdw_floats = ((size_t)g.nq * 2 + (tail ? 1 : 0)) * k * 2 * 64 * 2;
    g.wproj_bytes = (size_t)g.nq * nmb * 1024 + (tail ? (size_t)nmb * 512 : 0);
    return true;
}

static inline int fx_out_channel(int mb, int m) { return 32 * (mb >> 1) + 8 * (m >> 2) + 4 * (mb & 1) + (m & 3); }

void mx_fused_pack_weights(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
                           void* wexp_host, float* wdw_host, void* wproj_host) {
    const int NCx = Cin * 2 / 16, nq = g.nq, tail = g.HALF, nmb = g.NBO;
    __builtin_memset(wexp_host, 0, g.wexp_bytes);
    __builtin_memset(wproj_host, 0, g.wproj_bytes);
    for (int q = 0; q < nq; ++q)
        for (int j = 0; j < g.JX; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 31, hh = lane >> 5, c = hh * g.JX + j;
                if (c >= NCx) continue;
                uint16_t* dst = (uint16_t*)((char*)wexp_host + (((size_t)q * g.JX + j) * 64 + lane) * 16);
                for (int e = 0; e < 8; ++e) dst[e] = host_f32_to_bf16(kNegLog2e3 * we[(size_t)(q * 32 + n) * Cin + (size_t)c * 8 + e]);
            }
    uint32_t* wt = reinterpret_cast<uint32_t*>(wdw_host);
    mx_pack_taps(nq, k, wd, wt);
    for (int q = 0; q < nq; ++q)
        for (int mb = 0; mb < nmb; ++mb)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = fx_out_channel(mb, lane & 15), kc = lane >> 4;
                if (co >= Cout) continue;
                uint16_t* dst = (uint16_t*)((char*)wproj_host + (((size_t)q * nmb + mb) * 64 + lane) * 16);
                for (int e = 0; e < 8; ++e) dst[e] = host_f32_to_bf16(kNegLn23 * wp[(size_t)co * hid + q * 32 + kc * 8 + e]);
            }
    if (tail) {
        const int c0 = nq * 32;
        for (int lane = 0; lane < 64; ++lane) {                     // expand B fragment of 16x16x32: lane (n = channel, Cin chunk)
            const int n = lane & 15, kc = lane >> 4;
            if (kc >= NCx) continue;
            uint16_t* dst = (uint16_t*)((char*)wexp_host + (size_t)nq * g.JX * 1024 + (size_t)lane * 16);
            for (int e = 0; e < 8; ++e) dst[e] = host_f32_to_bf16(kNegLog2e3 * we[(size_t)(c0 + n) * Cin + (size_t)kc * 8 + e]);
        }
        uint32_t* tt = wt + (size_t)nq * 2 * k * 2 * 64 * 2;        // Toeplitz operands: channel c0 + kg*4 + pg
        for (int ky = 0; ky < k; ++ky)
            for (int ks = 0; ks < 2; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int kgq = lane >> 4, pg = (lane >> 2) & 3, i = lane & 3;
                    const float* wrow = wd + (size_t)(c0 + kgq * 4 + pg) * k * k + (size_t)ky * k;
                    uint16_t v[4];
                    for (int kk = 0; kk < 4; ++kk) {
                        const int kx = 4 * ks + kk - i;
                        v[kk] = (kx >= 0 && kx < k) ? host_f32_to_f16_3(wrow[kx]) : 0;
                    }
                    uint32_t* dst = tt + (((size_t)ky * 2 + ks) * 64 + lane) * 2;
                    dst[0] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
                    dst[1] = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
                }
        for (int mb = 0; mb < nmb; ++mb)                             // project A fragment of 16x16x16: lane (m, 4 channels)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = fx_out_channel(mb, lane & 15), kc = lane >> 4;
                if (co >= Cout) continue;
                uint16_t* dst = (uint16_t*)((char*)wproj_host + (size_t)nq * nmb * 1024 + ((size_t)mb * 64 + lane) * 8);
                for (int e = 0; e < 4; ++e) dst[e] = host_f32_to_bf16(kNegLn23 * wp[(size_t)co * hid + c0 + kc * 4 + e]);
            }
    }
}

hipError_t mx_fused_launch(hipStream_t s, const MbParams& p) {
    const FxEntry* e = fx_find(p.k, p.JX, 2 * ((p.Cout + 31) / 32), p.residual ? 1 : 0, (p.hid % 32) ? 1 : 0);
    if (!e || p.s != 1) return hipErrorInvalidValue;
    return e->fn(s, p);
}


// ---------------------------------------------------------------- fused block, stride 2, host side (MbGeom::kind = 6)
struct FsEntry {
    int k, jx, nmb, tail, toh, tow, xr, var, lds_bytes;
    hipError_t (*fn)(hipStream_t, const MbParams&);
};
template <int KS, int JX, int NMB, int TOH, int TOW, bool TAIL16, bool XRELOAD, bool ALDS>
static hipError_t fs_launch_t(hipStream_t s, const MbParams& p) {
    typedef Fs<KS, JX, NMB, TOH, TOW, TAIL16, ALDS> G;
    auto kfn = mbconv_mx2_kernel<KS, JX, NMB, TOH, TOW, TAIL16, XRELOAD, ALDS>;
    static thread_local bool configured_dev[32] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    bool& configured = configured_dev[dev & 31];
    if (G::LDS > 64 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    dim3 grid((p.Wout + TOW - 1) / TOW, (p.Hout + TOH - 1) / TOH, p.B), blk(256);
    set_kernel_tag("void cf::mbconv_mx2_kernel<%d, %d, %d, %d, %d, %s, %s, %s>(cf::MbParams)", KS, JX, NMB, TOH, TOW,
                   TAIL16 ? "true" : "false", XRELOAD ? "true" : "false", ALDS ? "true" : "false");
    hipLaunchKernelGGL(kfn, grid, blk, G::LDS, s, p);
    return hipGetLastError();
}
#define FSE(V, KS, JX, NMB, TAIL, TOH, TOW, XR, AL) \
    {KS, JX, NMB, TAIL, TOH, TOW, XR, V, Fs<KS, JX, NMB, TOH, TOW, (TAIL != 0), (AL != 0)>::LDS, &fs_launch_t<KS, JX, NMB, TOH, TOW, (TAIL != 0), (XR != 0), (AL != 0)>}
static const FsEntry kFsTable[] = {
    //  var KS JX NMB tail  tile  X reload  A in LDS       (B = 64, 640x640, HIP events; cf_mbconv2.hip kernel of the layer in brackets)
    FSE(0, 5, 2, 2, 1, 8, 16, 0, 0),      // 2.0  24 -> 144 -> 32, 160x160 -> 80x80, four rounds of 32 + one of 16: 0.150 ms [0.177]
    // layer1.0 (16 -> 96 -> 24, 3x3, 320x320 -> 160x160) stays on cf_mbconv2.hip: its depthwise is 4 % of the block's work and
    // 12 800 small workgroups pay the operand-table fetch three times each: 0.264 ms here (0.327 with the table in LDS) vs 0.246
#include CF_EXP_INC(cf_mbconv3_7)   // never-default variants: A/B runs of an experiments build only
};
#undef FSE
static const FsEntry* fs_find(int k, int jx, int nmb, int tail) {
    static const int want = cf_ab_int("CF_FS_VARIANT", 0);
    const FsEntry* base = nullptr;
    for (const FsEntry& e : kFsTable)
        if (e.k == k && e.jx == jx && e.nmb == nmb && e.tail == tail) {
            if (e.var == want) return &e;
            if (e.var == 0) base = &e;
        }
    return base;
}

bool mx_fused2_geometry(MbGeom& g, int Cin, int hid, int Cout, int k, int s) {
    static const bool off = cf_env_int("CF_DW_MATRIX", 1) == 0 || cf_ab_int("CF_FS", 1) == 0;
    if (off || s != 2 || (Cin % 8) || (Cout % 8) || Cout > 64 || (hid % 16) || hid == Cin) return false;
    const int jx = (Cin * 2 / 16 + 1) / 2, nmb = 2 * ((Cout + 31) / 32), tail = (hid % 32) ? 1 : 0;
    const FsEntry* e = fs_find(k, jx, nmb, tail);
    if (!e) return false;
    g = MbGeom{};
    g.ok = true; g.kind = 6; g.S = 2;
    g.JX = jx; g.NBO = nmb; g.HC = 32; g.nq = hid / 32; g.NBE = 1; g.HALF = tail;
    g.rowb = 32 * 8 + 16;
    g.lds_bytes = (size_t)e->lds_bytes;
    g.wexp_bytes = (size_t)g.nq * g.JX * 1024 + (tail ? 1024 : 0);
    g.wdw_floats = ((size_t)g.nq * 2 + (tail ? 1 : 0)) * k * 3 * 64 * 2;
    g.wproj_bytes = (size_t)g.nq * 2 * nmb * 512 + (tail ? (size_t)nmb * 512 : 0);
    return true;
}

void mx_fused2_pack_weights(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
                            void* wexp_host, float* wdw_host, void* wproj_host) {
    const int NCx = Cin * 2 / 16, nq = g.nq, tail = g.HALF, nmb = g.NBO;
    __builtin_memset(wexp_host, 0, g.wexp_bytes);
    __builtin_memset(wproj_host, 0, g.wproj_bytes);
    for (int q = 0; q < nq; ++q)
        for (int j = 0; j < g.JX; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 31, hh = lane >> 5, c = hh * g.JX + j;
                if (c >= NCx) continue;
                uint16_t* dst = (uint16_t*)((char*)wexp_host + (((size_t)q * g.JX + j) * 64 + lane) * 16);
                for (int e = 0; e < 8; ++e) dst[e] = host_f32_to_bf16(kNegLog2e3 * we[(size_t)(q * 32 + n) * Cin + (size_t)c * 8 + e]);
            }
    uint32_t* wt = reinterpret_cast<uint32_t*>(wdw_host);
    mx_pack_taps(nq, k, wd, wt, 2, 3);
    // project, per (round, channel half): A fragment of 16x16x16: lane (m = output channel slot, kc = group) holds the 4
    // hidden channels q*32 + kc*8 + half*4 + e
    for (int q = 0; q < nq; ++q)
        for (int hf = 0; hf < 2; ++hf)
            for (int mb = 0; mb < nmb; ++mb)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = fx_out_channel(mb, lane & 15), kc = lane >> 4;
                    if (co >= Cout) continue;
                    uint16_t* dst = (uint16_t*)((char*)wproj_host + ((((size_t)q * 2 + hf) * nmb + mb) * 64 + lane) * 8);
                    for (int e = 0; e < 4; ++e) dst[e] = host_f32_to_bf16(kNegLn23 * wp[(size_t)co * hid + q * 32 + kc * 8 + hf * 4 + e]);
                }
    if (tail) {
        const int c0 = nq * 32;
        for (int lane = 0; lane < 64; ++lane) {
            const int n = lane & 15, kc = lane >> 4;
            if (kc >= NCx) continue;
            uint16_t* dst = (uint16_t*)((char*)wexp_host + (size_t)nq * g.JX * 1024 + (size_t)lane * 16);
            for (int e = 0; e < 8; ++e) dst[e] = host_f32_to_bf16(kNegLog2e3 * we[(size_t)(c0 + n) * Cin + (size_t)kc * 8 + e]);
        }
        uint32_t* tt = wt + (size_t)nq * 2 * k * 3 * 64 * 2;
        for (int ky = 0; ky < k; ++ky)
            for (int ks = 0; ks < 3; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int kgq = lane >> 4, pg = (lane >> 2) & 3, i = lane & 3;
                    const float* wrow = wd + (size_t)(c0 + kgq * 4 + pg) * k * k + (size_t)ky * k;
                    uint16_t v[4];
                    for (int kk = 0; kk < 4; ++kk) {
                        const int kx = 4 * ks + kk - 2 * i;
                        v[kk] = (kx >= 0 && kx < k) ? host_f32_to_f16_3(wrow[kx]) : 0;
                    }
                    uint32_t* dst = tt + (((size_t)ky * 3 + ks) * 64 + lane) * 2;
                    dst[0] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
                    dst[1] = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
                }
        for (int mb = 0; mb < nmb; ++mb)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = fx_out_channel(mb, lane & 15), kc = lane >> 4;
                if (co >= Cout) continue;
                uint16_t* dst = (uint16_t*)((char*)wproj_host + (size_t)nq * 2 * nmb * 512 + ((size_t)mb * 64 + lane) * 8);
                for (int e = 0; e < 4; ++e) dst[e] = host_f32_to_bf16(kNegLn23 * wp[(size_t)co * hid + c0 + kc * 4 + e]);
            }
    }
}

hipError_t mx_fused2_launch(hipStream_t s, const MbParams& p) {
    const FsEntry* e = fs_find(p.k, p.JX, 2 * ((p.Cout + 31) / 32), (p.hid % 32) ? 1 : 0);
    if (!e || p.s != 2) return hipErrorInvalidValue;
    return e->fn(s, p);
}

#endif  // !CF_ILP_TU
}  // namespace cf
