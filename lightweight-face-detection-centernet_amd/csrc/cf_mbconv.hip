// Fused MBConv block for gfx950: expand 1x1 (+Swish) -> depthwise k x k (+Swish) -> project 1x1
// (+residual) in ONE kernel; the 6x-expanded tensor never touches HBM.
//
// Replaces MBConvBlock.forward (model/centernet.py:89-140; SE is off :232, no BN :117-120) for the
// blocks whose output fits the accumulator budget (Cout <= 96: layer1.0 .. layer4.1).  Layer by
// layer these blocks move 53.6/33/20 MB per image (bf16) through HBM for layer1.0/1.1/2.0; fused
// they move block input + block output only (4.5/2.5/1.6 MB) -- SURVEY.md section 7 step 8.
//
// Structure (one workgroup = 4 waves = one 8x16 output tile, all output channels):
//   for each chunk of HC hidden channels:
//     phase 1  expand: the (8-1)s+k x (16-1)s+k input tile (halo included, zero outside the image =
//              the reference's ZeroPad2d, since swish(0*W) = 0) is read straight from HBM into MFMA
//              operand registers, D^T = We . X^T on the matrix core, Swish, and the result goes to
//              LDS as E[pixel][HC] (row stride padded to an odd multiple of 16 B: conflict-free
//              ds_read_b128 / ds_write_b128 for consecutive pixels).
//     phase 2  depthwise: lane (pixel, h) computes its pixel's k*k taps for one 16-byte channel
//              chunk at a time from LDS (weights from LDS, broadcast), Swish, packs the result --
//              which is EXACTLY the MFMA B-operand fragment of the project GEMM (lane = pixel,
//              8 contiguous k) -- so the depthwise output never leaves registers:
//     phase 3  project: acc[out n-block] += Wp . D^T, accumulating over all hidden chunks.
//   epilogue: + residual, 16 contiguous output channels per lane -> 16-byte stores.
// The same two free permutations as cf_pw.hip are used (output channel <-> MFMA row, k <-> slot).
#include "cf_exp.h"
#include "cf_common.h"
#include "cf_kernels.h"
#include <type_traits>
#include <cstdlib>
#include <vector>

namespace cf {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

static inline int slot_channel(int nb, int i) {
    int h = (i >> 2) & 1;
    int r = (i & 3) + 4 * (i >> 3);
    return nb * 32 + h * 16 + r;
}



template <typename T> using MbMma = CfMma<T>;

// ---------------------------------------------------------------- host: packing (geometry: mb_geometry below the kernel table)
// we [hid][Cin], wd [hid][k*k], wp [Cout][hid]
void mb_pack_weights(int dtype, const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we,
                     const float* wd, const float* wp, void* wexp_host, float* wdw_host, void* wproj_host) {
    if (g.kind == 1 || g.kind == 2) { mb2_pack_weights(g, Cin, hid, Cout, k, we, wd, wp, wexp_host, wdw_host, wproj_host); return; }
    if (g.kind == 4) { mx_pack_weights(g, Cin, hid, Cout, k, we, wd, wp, wexp_host, wdw_host, wproj_host); return; }
    if (g.kind == 5) { mx_fused_pack_weights(g, Cin, hid, Cout, k, we, wd, wp, wexp_host, wdw_host, wproj_host); return; }
    if (g.kind == 6) { mx_fused2_pack_weights(g, Cin, hid, Cout, k, we, wd, wp, wexp_host, wdw_host, wproj_host); return; }
#include CF_EXP_INC(cf_mbconv_m7_pack)
    if (g.kind == 9) { mb6_pack(g, Cin, hid, Cout, k, we, wd, wp, wexp_host, wdw_host, wproj_host); return; }
    if (g.kind == 8) {                       // cf_mbconv5.hip: this file's expand fragments, taps as [chunk][group of 4 channels][tap][4]
        MbGeom g0 = g; g0.kind = 0; g0.NBO = 0; g0.HALF = 0; g0.wproj_bytes = 0;
        std::vector<float> generic(g.wdw_floats);
        mb_pack_weights(dtype, g0, Cin, hid, Cout, k, we, wd, nullptr, wexp_host, generic.data(), nullptr);
        for (int q = 0; q < g.nq; ++q)
            for (int grp = 0; grp < g.HC / 4; ++grp)
                for (int t = 0; t < k * k; ++t)
                    for (int c = 0; c < 4; ++c)
                        wdw_host[(((size_t)q * (g.HC / 4) + grp) * k * k + t) * 4 + c] = wd[(size_t)(q * g.HC + grp * 4 + c) * k * k + t];
        return;
    }
    if (g.kind == 7) {                       // cf_mbconv4.hip: this file's expand fragments, its own tap table and project fragments
        MbGeom g0 = g; g0.kind = 0;
        mb_pack_weights(dtype, g0, Cin, hid, Cout, k, we, wd, wp, wexp_host, wdw_host, wproj_host);      // (its project fragments are overwritten)
        std::vector<float> wp7;
        if (dtype == 2) { wp7.assign(wp, wp + (size_t)Cout * hid); for (float& v : wp7) v *= kCfNegLn2; wp = wp7.data(); }   // as the recursive call did for its own copy
        mb4_repack(dtype, g, hid, Cout, k, wd, wp, wdw_host, wproj_host);
        return;
    }
    // split mode: -log2(e) folded into the expand weights, the leftover -ln 2 into the project weights (swish2_sel<true>, cf_common.h);
    // the depthwise taps stay as they are (their input and their output both carry the -log2(e) factor)
    std::vector<float> we_s, wp_s;
    if (dtype == 2) {
        we_s.assign(we, we + (size_t)hid * Cin);
        for (float& v : we_s) v *= kCfNegLog2e;
        we = we_s.data();
        if (wp) { wp_s.assign(wp, wp + (size_t)Cout * hid); for (float& v : wp_s) v *= kCfNegLn2; wp = wp_s.data(); }
    }
    const int P = per16(dtype);
    const int NCx = Cin * (int)elem_size(dtype) / 16;
    __builtin_memset(wexp_host, 0, g.wexp_bytes);
    if (wproj_host && g.wproj_bytes) __builtin_memset(wproj_host, 0, g.wproj_bytes);
    auto put = [&](char* dst, const float* src, int n) { (void)n; pack_chunk(dtype, src, dst); };      // n = P: one 16-byte chunk
    for (int q = 0; q < g.nq; ++q) {
        for (int nbl = 0; nbl < g.NBE; ++nbl)
            for (int j = 0; j < g.JX; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    int cl = slot_channel(nbl, i);                 // channel within the hidden chunk
                    if (nbl == g.NBE - 1 && g.HC % 32 == 16) {
                        // half-filled last block: 8 channels on EACH half of the MFMA rows (registers
                        // 0..7 of both lane halves) instead of 16 on one half -- the Swish epilogue
                        // then works on 8 live registers in every lane, none on dead ones
                        const int hh = (i >> 2) & 1, rr = (i & 3) + 4 * (i >> 3);
                        cl = rr < 8 ? nbl * 32 + hh * 8 + rr : g.HC;
                    }
                    const int c = h * g.JX + j;                     // 16-byte chunk of the Cin row
                    if (cl >= g.HC || c >= NCx) continue;
                    char* dst = (char*)wexp_host + ((((size_t)q * g.NBE + nbl) * g.JX + j) * 64 + lane) * 16;
                    put(dst, we + (size_t)(q * g.HC + cl) * Cin + (size_t)c * P, P);
                }
        for (int t = 0; t < k * k; ++t)
            for (int cl = 0; cl < g.HC; ++cl)
                wdw_host[((size_t)q * k * k + t) * g.HC + cl] = wd[(size_t)(q * g.HC + cl) * k * k + t];
        for (int nbo = 0; nbo < g.NBO; ++nbo)
            for (int j = 0; j < g.HALF; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    const int co = slot_channel(nbo, i);
                    if (co >= Cout) continue;
                    const int hc = q * g.HC + (h * g.HALF + j) * P;  // first hidden channel of the chunk
                    char* dst = (char*)wproj_host + ((((size_t)nbo * g.nq + q) * g.HALF + j) * 64 + lane) * 16;
                    put(dst, wp + (size_t)co * hid + hc, P);
                }
    }
    if (dtype == 2) {                                               // split mode: chunk pairs per MFMA chain (cf_common.h)
        split_pairs_inplace(wexp_host, (size_t)g.nq * g.NBE, g.JX);                             // expand: the JX chunks of a lane half
        // project: one chunk per MFMA set.  Pairing the k-steps here (two depthwise chunks + two weight fragments per n-block live
        // at once) cost layer4.1 its second wave per SIMD (0.331 -> 0.463 ms) and gained nothing on 3.x: measured, not kept
        if (wproj_host) split_pairs_inplace(wproj_host, (size_t)g.NBO * g.nq * g.HALF, 1);
    }
}

// ---------------------------------------------------------------- device
// Template geometry: TOH x TOW output tile (NPB = TOH*TOW/32 pixel blocks), NW waves = NPB x KG
// (KG = 1 or 2 k-groups: with KG = 2, waves w and w + NPB share pixel block w % NPB and split the
// k-steps of every hidden chunk; their partial project sums are combined once, through LDS, in the
// epilogue -- twice the waves per LDS byte).  JX = 16-byte chunks of a Cin row per lane half, HC =
// hidden chunk, EF = keep the expanded tile in LDS as fp32 (no bf16 unpack in the depthwise inner
// loop: half the VALU work per tap, at 2x the LDS bytes per channel -> smaller HC).  Everything is
// compile-time so the expand / depthwise / project loops are straight-line code.
template <typename T, int KS, int S, int NBO, bool RESID, int NW, int JX, int HC, int TOH, int TOW, bool EF>
__global__ __launch_bounds__(NW * 64) void mbconv_kernel(MbParams p) {
    typedef typename std::conditional<EF, float, T>::type ET;     // element type of E in LDS
    constexpr int P = Elem<T>::PER16;                             // channels per project k-chunk
    constexpr bool PRE = std::is_same<T, sp32_t>::value;          // split mode: Swish factors folded into the expand / project weights (mb_pack_weights)
    constexpr int EP = 16 / (int)sizeof(ET);                      // E elements per 16 bytes
    constexpr int IH = (TOH - 1) * S + KS, IW = (TOW - 1) * S + KS, IPX = IH * IW;
    constexpr int NIB = (IPX + 31) / 32;
    constexpr int NPB = TOH * TOW / 32, KG = NW / NPB;
    static_assert(NPB * 32 == TOH * TOW && KG * NPB == NW && (KG == 1 || KG == 2), "tile / wave geometry");
    constexpr int NBE = (HC + 31) / 32, HALF = HC * (int)sizeof(T) / 16 / 2;
    constexpr int ROWB = HC * (int)sizeof(ET) + 16;
    constexpr int WXB = NBE * JX * 1024;                         // expand fragments per chunk (bytes)
    constexpr int WDB = KS * KS * HC * 4;                        // depthwise taps (fp32)
    constexpr int WSTAGE = WXB + WDB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* E = smem;
    char* Wst = smem + ((IPX * ROWB + 15) / 16 * 16);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int ox0 = blockIdx.x * TOW, oy0 = blockIdx.y * TOH, b = blockIdx.z;
    const int nq = p.nq;

    // this lane's output pixel (phase 2/3 and epilogue) and its share of the k-steps
    const int pbk = wave % NPB, jg = wave / NPB;
    const int o = pbk * 32 + (sizeof(T) == 4 ? lds_group_pixel(pl) : pl);       // fp32 tile: conflict-free ds_read_b128 groups (cf_common.h)
    const int oy = o / TOW, ox = o % TOW;
    const unsigned e_pix = (unsigned)((oy * S) * IW + ox * S) * (unsigned)ROWB;
    constexpr int JSPLIT = KG == 2 ? (HALF + 1) / 2 : HALF;

    f32x16 acc[NBO];
#pragma unroll
    for (int i = 0; i < NBO; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const char* xbase = (const char*)p.x + (size_t)b * p.Hin * p.Win * p.Cin * sizeof(T);
    const unsigned rowbytes = (unsigned)p.Cin * sizeof(T);

    // weights of hidden chunk q -> stage (q & 1) by DMA (global_load_lds): nothing waits here, the
    // copy lands under the following phase and is fenced by the next __syncthreads()
    auto stage_weights = [&](int q) {
        char* dst = Wst + (q & 1) * WSTAGE;
        const char* srcx = (const char*)p.wexp + (size_t)q * WXB;
        for (int c = wave; c < WXB / 1024; c += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcx + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
        const char* srcd = (const char*)(p.wdw + (size_t)q * KS * KS * HC);
        for (int c = wave; c < (WDB + 1023) / 1024; c += NW)
            if (c * 1024 + lane * 16 < WDB)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcd + c * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(dst + WXB + c * 1024), 16, 0, 0);
    };

    // This wave's input-tile pixel blocks (ib = wave + NW * t) -> X fragments, loaded ONCE: the input
    // tile is the same for every hidden chunk.  Branch-free: always load from a clamped (valid)
    // address, then zero what lies outside the image (= ZeroPad2d) -- predicated loads would each
    // end in a full vmcnt(0) wait.  A row whose 16-byte chunk count is odd is over-read by one chunk
    // on the h = 1 half: the matching weight fragment is zero and activation buffers are
    // zero-initialised with slack, so it is inert.
    constexpr int MAXI = (NIB + NW - 1) / NW;
    u32x4 xf[MAXI][JX];
#pragma unroll
    for (int t = 0; t < MAXI; ++t) {
        const int ib = wave + NW * t;
        const int ip = ib * 32 + pl;
        const int ipc = ip < IPX ? ip : IPX - 1;
        const int iy = ipc / IW, ix = ipc - iy * IW;
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
        // split mode (round 5): split into bf16 (hi, lo) ONCE, here, not once per hidden chunk inside the MFMA chain (a chunk pair ->
        // [8 x hi], [8 x lo]; an odd last chunk -> [4 x hi | 4 x lo]): the same products in the same order
        if constexpr (std::is_same<T, sp32_t>::value) {
#pragma unroll
            for (int j = 0; j + 1 < JX; j += 2) { const SplitPair s2 = split8(xf[t][j], xf[t][j + 1]); xf[t][j] = s2.hi; xf[t][j + 1] = s2.lo; }
            if constexpr (JX & 1) { u32x2 xh, xl; split4(xf[t][JX - 1], xh, xl); xf[t][JX - 1].x = xh.x; xf[t][JX - 1].y = xh.y; xf[t][JX - 1].z = xl.x; xf[t][JX - 1].w = xl.y; }
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
            if constexpr (std::is_same<T, sp32_t>::value) {     // fragments already split (above)
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
            mma_chain<T, JX>(a, [&](int j) { return ld16(wb + j * 1024); }, [&](int j) { return xfr[j]; });
            constexpr bool PARTIAL = (HC % 32 == 16);
            if (PARTIAL && nbl == NBE - 1) {
                // half-filled last block: channels nbl*32 + h*8 + [0,8) live in registers 0..7
                const int ch0 = nbl * 32 + h * 8;
#pragma unroll
                for (int g = 0; g < 8 / EP; ++g) {
                    float v[EP];
#pragma unroll
                    for (int e = 0; e < EP; e += 2) {
                        f32x2 x2; x2.x = a[g * EP + e]; x2.y = a[g * EP + e + 1];
                        const f32x2 y2 = swish2_sel<PRE>(x2);
                        v[e] = y2.x; v[e + 1] = y2.y;
                    }
                    if (ipok) st16(erow + (ch0 + g * EP) * (int)sizeof(ET), pack16<ET>(v));
                }
            } else {
                const int ch0 = nbl * 32 + h * 16;
#pragma unroll
                for (int g = 0; g < 16 / EP; ++g) {
                    float v[EP];
#pragma unroll
                    for (int e = 0; e < EP; e += 2) {
                        f32x2 x2; x2.x = a[g * EP + e]; x2.y = a[g * EP + e + 1];
                        const f32x2 y2 = swish2_sel<PRE>(x2);
                        v[e] = y2.x; v[e + 1] = y2.y;
                    }
                    if (ipok) st16(erow + (ch0 + g * EP) * (int)sizeof(ET), pack16<ET>(v));
                }
            }
        }
    };

    stage_weights(0);
    for (int q = 0; q < nq; ++q) {
        const char* wx = Wst + (q & 1) * WSTAGE;
        const char* wdq = wx + WXB;
        cf_sync_lds_dma();    // previous chunk's phase 2 done with E; this stage's weights (LDS-DMA) landed for every wave

        // ---- phase 1: expand + Swish -> E (X fragments are register-resident)
#pragma unroll
        for (int t = 0; t < MAXI; ++t) {
            const int ib = wave + NW * t;
            if (ib < NIB) expand_block(ib, xf[t], wx);
        }
        __syncthreads();
        if (q + 1 < nq) stage_weights(q + 1);       // streams in under this chunk's depthwise

        // ---- phase 2 + 3: depthwise + Swish in registers, straight into the project MFMA
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            // KG == 2: wave group jg owns k-steps [0, JSPLIT) or [JSPLIT, HALF) -- wave-uniform
            if (KG == 2 && ((j < JSPLIT) != (jg == 0))) continue;
            u32x4 wpc[NBO];
#pragma unroll
            for (int i = 0; i < NBO; ++i)
                wpc[i] = ld16((const char*)p.wproj + ((((size_t)i * nq + q) * HALF + j) * 64 + lane) * 16);
            const int c = h * HALF + j;                               // project k-chunk: P hidden channels
            f32x2 d2[P / 2];
            const char* eb = E + e_pix + c * (P * (int)sizeof(ET));
            const char* wdb = wdq + c * (P * 4);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    float ev[P], wv[P];
                    const char* et = eb + (ky * IW + kx) * ROWB;
#pragma unroll
                    for (int g = 0; g < P / EP; ++g) unpack16<ET>(ld16(et + g * 16), ev + g * EP);
                    const char* wt = wdb + (ky * KS + kx) * HC * 4;
#pragma unroll
                    for (int g = 0; g < P / 4; ++g) unpack16<float>(ld16(wt + g * 16), wv + g * 4);
#pragma unroll
                    for (int e = 0; e < P / 2; ++e) {
                        f32x2 e2, w2; e2.x = ev[2 * e]; e2.y = ev[2 * e + 1]; w2.x = wv[2 * e]; w2.y = wv[2 * e + 1];
                        d2[e] = (ky == 0 && kx == 0) ? e2 * w2 : fma2(e2, w2, d2[e]);   // v_pk_mul / v_pk_fma
                    }
                }
            float d[P];
#pragma unroll
            for (int e = 0; e < P / 2; ++e) { const f32x2 y2 = swish2_sel<PRE>(d2[e]); d[2 * e] = y2.x; d[2 * e + 1] = y2.y; }
            const u32x4 xc = pack16<T>(d);
#pragma unroll
            for (int i = 0; i < NBO; ++i) MbMma<T>::run(acc[i], wpc[i], xc);
        }
    }

    // ---- combine the two k-groups: the upper group hands its partial sums over through LDS, one
    // output n-block at a time (the buffer stays smaller than the E tile it reuses)
    if constexpr (KG == 2) {
        float* red = reinterpret_cast<float*>(smem) + (size_t)(pbk * 64 + lane) * 16;
#pragma unroll
        for (int i = 0; i < NBO; ++i) {
            __syncthreads();                              // everyone is done with E / the previous n-block
            if (jg == 1) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = acc[i][g * 4 + e];
                    st16(red + g * 4, pack16<float>(t));
                }
            }
            __syncthreads();
            if (jg == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float t[4];
                    unpack16<float>(ld16(red + g * 4), t);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][g * 4 + e] += t[e];
                }
            }
        }
        if (jg == 1) return;
    }

    // ---- epilogue: (+ residual) -> y
    const int gy = oy0 + oy, gx = ox0 + ox;
    if (gy >= p.Hout || gx >= p.Wout) return;
    const size_t opix = ((size_t)b * p.Hout + gy) * p.Wout + gx;
#pragma unroll
    for (int i = 0; i < NBO; ++i) {
        const int cb = i * 32 + h * 16;
#pragma unroll
        for (int g = 0; g < 16 / P; ++g) {
            const int ch = cb + g * P;
            if (ch >= p.Cout) break;
            float v[P];
#pragma unroll
            for (int e = 0; e < P; ++e) v[e] = acc[i][g * P + e];
            if constexpr (RESID) {                                    // Cin == Cout, stride 1: same pixel of x
                float r[P];
                unpack16<T>(ld16((const char*)p.x + (opix * p.Cin + ch) * sizeof(T)), r);
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = r[e] + v[e];
            }
            // yblock: pixel-block order [m / 32][Cout / P][m % 32][P] (the next block's expand+depthwise kernel reads it as operand fragments)
            st16((char*)p.y + (p.yblock ? blk_off(opix, p.Cout / P, ch / P) : (opix * p.Cout + ch) * sizeof(T)), pack16<T>(v));
        }
    }
}

// One table row = one kernel instantiation = one block shape of the CenterFace backbone
// (model/centernet.py:211-219, layer1.0 .. layer4.1) in one storage type.
struct MbEntry {
    int dtype, k, s, jx, hc, nbo, res;
    int toh, tow, ef, nw, var, lds_bytes;     // var > 0: experimental variant, selected with CF_MB_VARIANT=var
    hipError_t (*fn)(hipStream_t, const MbParams&);
};

template <typename T, int KS, int S, int NBO, bool RESID, int NW, int JX, int HC, int TOH, int TOW, bool EF>
static hipError_t mb_launch(hipStream_t s, const MbParams& p) {
    auto kfn = mbconv_kernel<T, KS, S, NBO, RESID, NW, JX, HC, TOH, TOW, EF>;
    // function attributes are per device: remember what was set for each
    static thread_local size_t configured_dev[32] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    size_t& configured = configured_dev[dev & 31];
    if (p.lds_bytes > 64 * 1024 && configured < p.lds_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return e;
        configured = p.lds_bytes;
    }
    dim3 grid((p.Wout + TOW - 1) / TOW, (p.Hout + TOH - 1) / TOH, p.B), blk(NW * 64);
    set_kernel_tag("void cf::mbconv_kernel<%s, %d, %d, %d, %s, %d, %d, %d, %d, %d, %s>(cf::MbParams)", type_tag<T>(), KS, S, NBO,
                   RESID ? "true" : "false", NW, JX, HC, TOH, TOW, EF ? "true" : "false");
    hipLaunchKernelGGL(kfn, grid, blk, p.lds_bytes, s, p);
    return hipGetLastError();
}

template <typename T, int KS, int S, int JX, int HC, int TOH, int TOW, bool EF, int NBO, int NW>
constexpr int mb_lds_bytes() {
    constexpr int ES = EF ? 4 : (int)sizeof(T);
    constexpr int IPX = ((TOH - 1) * S + KS) * ((TOW - 1) * S + KS);
    constexpr int NBE = (HC + 31) / 32;
    constexpr int main_bytes = (IPX * (HC * ES + 16) + 15) / 16 * 16 + 2 * (NBE * JX * 1024 + KS * KS * HC * 4);
    // KG == 2: the epilogue reuses the space for the k-group reduction, 64 lanes x 16 floats per pixel block
    constexpr int NPB = TOH * TOW / 32;
    constexpr int red_bytes = NW > NPB ? NPB * 64 * 16 * 4 : 0;
    return main_bytes > red_bytes ? main_bytes : red_bytes;
}

#define MB_ENTRY(T, DT, KS, S, JX, HC, NBO, RES, TOH, TOW, EF, NW) MB_VARIANT(0, T, DT, KS, S, JX, HC, NBO, RES, TOH, TOW, EF, NW)
#define MB_VARIANT(V, T, DT, KS, S, JX, HC, NBO, RES, TOH, TOW, EF, NW) \
    {DT, KS, S, JX, HC, NBO, RES, TOH, TOW, EF, NW, V, mb_lds_bytes<T, KS, S, JX, HC, TOH, TOW, (EF != 0), NBO, NW>(), \
     &mb_launch<T, KS, S, NBO, (RES != 0), NW, JX, HC, TOH, TOW, (EF != 0)>}
static const MbEntry kMbTable[] = {
    // bf16 storage: KS S JX HC NBO res | tile  E-fp32 waves        layer   (per-layer best of the
    // measured variants: E as bf16 vs fp32, tile 8x16 / 4x16 / 16x16 / 8x20 / 8x32 / 8x40, 4-16 waves, HC --
    // profiles/r01_mbconv_variants.md)
    MB_ENTRY(bf16_t, 1, 3, 2, 1, 32, 1, 0, 8, 16, 0, 4),   // 1.0  16 ->  96 -> 24
    MB_ENTRY(bf16_t, 1, 3, 1, 2, 48, 1, 1, 8, 16, 0, 4),   // 1.1  24 -> 144 -> 24 (+res)
    MB_ENTRY(bf16_t, 1, 5, 2, 2, 48, 1, 0, 8, 16, 0, 8),   // 2.0  24 -> 144 -> 32
    MB_ENTRY(bf16_t, 1, 5, 1, 2, 48, 1, 1, 8, 16, 0, 8),   // 2.1  32 -> 192 -> 32 (+res)
    MB_ENTRY(bf16_t, 1, 3, 2, 2, 32, 2, 0, 8, 16, 0, 4),   // 3.0  32 -> 192 -> 64
    MB_ENTRY(bf16_t, 1, 3, 1, 4, 32, 2, 1, 8, 16, 1, 4),   // 3.1  64 -> 384 -> 64 (+res)
    MB_ENTRY(bf16_t, 1, 5, 1, 4, 32, 3, 0, 8, 20, 0, 10),  // 4.0  64 -> 384 -> 96      8x20 tiles: a 40x40 map
    MB_ENTRY(bf16_t, 1, 5, 1, 6, 32, 3, 1, 8, 20, 0, 10),  // 4.1  96 -> 576 -> 96 (+res)   has no edge waste
#include CF_EXP_INC(cf_mbconv_0)   // never-default variants (CF_MB_VARIANT=n): A/B runs of an experiments build only
    // fp32 storage (parity mode).  Round-3 A/B over hidden chunk / tile / k-groups (B = 64, 640x640, ms; previous entry in brackets):
    // what matters is the LDS footprint of the fp32 tile (1.0 at 8x16 / HC 32 = 87 KB = ONE four-wave workgroup per CU)
    MB_ENTRY(float, 0, 3, 2, 2, 32, 1, 0, 4, 16, 1, 4),    // 1.0  4x16 tile, two k-groups, 49 KB: 0.564 [8x16: 0.786; 8x16 HC 16: 0.635]
    MB_ENTRY(float, 0, 3, 1, 3, 48, 1, 1, 8, 16, 1, 8),    // 1.1  two k-groups: 0.497 [0.505; HC 16: 0.590; 16x16: 0.589]
    MB_ENTRY(float, 0, 5, 2, 3, 16, 1, 0, 8, 16, 1, 8),    // 2.0  HC 16, 62 KB: 0.428 [HC 48, 138 KB: 0.441; 4x16: 0.454 / 0.636]
    MB_ENTRY(float, 0, 5, 1, 4, 48, 1, 1, 8, 16, 1, 8),    // 2.1  HC 48: 0.319 [HC 32: 0.365; HC 16: 0.363]
    MB_ENTRY(float, 0, 3, 2, 4, 32, 2, 0, 8, 16, 1, 8),    // 3.0  two k-groups: 0.150 [0.170; HC 16: 0.203]
    MB_ENTRY(float, 0, 3, 1, 8, 32, 2, 1, 8, 16, 1, 4),    // 3.1  0.192 [two k-groups: 0.191; HC 64: 0.220; HC 48: 0.232]
    MB_ENTRY(float, 0, 5, 1, 8, 32, 3, 0, 8, 16, 1, 8),    // 4.0  0.277 [HC 16: 0.363; HC 48: 0.344; one k-group: 0.378]
    MB_ENTRY(float, 0, 5, 1, 12, 32, 3, 1, 8, 16, 1, 4),   // 4.1  one k-group: 0.512 [0.534; HC 16: 0.724; HC 48: 0.596]
    // fp32 storage + split-bf16 GEMM products (dtype 2): the fp32 geometries (same LDS tile, same epilogues)
    MB_ENTRY(sp32_t, 2, 3, 2, 2, 32, 1, 0, 4, 16, 1, 4),   // 1.0
    MB_ENTRY(sp32_t, 2, 3, 1, 3, 48, 1, 1, 8, 16, 1, 8),   // 1.1
    MB_ENTRY(sp32_t, 2, 5, 2, 3, 16, 1, 0, 8, 16, 1, 8),   // 2.0
    MB_ENTRY(sp32_t, 2, 5, 1, 4, 48, 1, 1, 8, 16, 1, 8),   // 2.1
    MB_ENTRY(sp32_t, 2, 3, 2, 4, 32, 2, 0, 4, 16, 1, 4),   // 3.0  4x16 tile: 0.111 -> 0.095 (tools/split_sweep.sh)
    MB_ENTRY(sp32_t, 2, 3, 1, 8, 32, 2, 1, 8, 16, 1, 8),   // 3.1  two k-groups: 0.1275 -> 0.123
    MB_ENTRY(sp32_t, 2, 5, 1, 8, 32, 3, 0, 8, 16, 1, 8),   // 4.0
    MB_ENTRY(sp32_t, 2, 5, 1, 12, 32, 3, 1, 8, 16, 1, 4),  // 4.1
#include CF_EXP_INC(cf_mbconv_1)   // A/B sweep of the split mode (CF_MB_VARIANT=1..3)
};
#undef MB_ENTRY

static const MbEntry* mb_find(int dtype, int k, int s, int jx, int nbo, int res) {
    static const int want = cf_ab_int("CF_MB_VARIANT", 0);
    const MbEntry* base = nullptr;
    for (const MbEntry& e : kMbTable)
        if (e.dtype == dtype && e.k == k && e.s == s && e.jx == jx && e.nbo == nbo && e.res == res) {
            if (e.var == want) return &e;
            if (e.var == 0) base = &e;
        }
    return base;
}

MbGeom mb_geometry(int dtype, int Cin, int hid, int Cout, int k, int s) {
    MbGeom g{};
    const int sz = (int)elem_size(dtype);
    if ((Cin % 8) || (Cout % 8) || Cout > 96 || Cin > 96 || hid == Cin) return g;
    if (dtype == 1 && mx_fused_geometry(g, Cin, hid, Cout, k, s)) return g;      // stride 1: depthwise on the matrix cores
    if (dtype == 1 && mx_fused2_geometry(g, Cin, hid, Cout, k, s)) return g;     // stride 2
    if (dtype == 1 && mb2_geometry(g, Cin, hid, Cout, k, s)) return g;
#include CF_EXP_INC(cf_mbconv_m7_geometry)
    if (dtype == 2 && mb6_geometry(dtype, g, Cin, hid, Cout, k, s)) return g;     // round 5: register-window depthwise (cf_mbconv6.hip)
    if (dtype != 1 && mb4_geometry(dtype, g, Cin, hid, Cout, k, s)) return g;
    g.JX = (Cin * sz / 16 + 1) / 2;
    g.NBO = (Cout + 31) / 32;
    const MbEntry* e = mb_find(dtype, k, s, g.JX, g.NBO, (Cin == Cout && s == 1) ? 1 : 0);
    if (!e || hid % e->hc) return g;
    g.ok = true;
    g.HC = e->hc; g.nq = hid / g.HC;
    g.NBE = (g.HC + 31) / 32;
    g.HALF = g.HC * sz / 16 / 2;
    g.rowb = g.HC * (e->ef ? 4 : sz) + 16;
    g.lds_bytes = (size_t)e->lds_bytes;
    g.KG = e->nw / (e->toh * e->tow / 32);
    g.wexp_bytes = (size_t)g.nq * g.NBE * g.JX * 64 * 16;
    g.wdw_floats = (size_t)g.nq * k * k * g.HC;
    g.wproj_bytes = (size_t)g.NBO * g.nq * g.HALF * 64 * 16;
    return g;
}

hipError_t launch_mbconv(hipStream_t s, int dtype, const MbParams& p) {
    if (p.B <= 0) return hipSuccess;
    if (p.kind == 1) return dtype == 1 ? mb2_launch(s, p) : hipErrorInvalidValue;
    if (p.kind == 2) return dtype == 1 ? expdw_launch(s, p) : hipErrorInvalidValue;
    if (p.kind == 4) return dtype == 1 ? mx_launch(s, p) : hipErrorInvalidValue;
    if (p.kind == 5) return dtype == 1 ? mx_fused_launch(s, p) : hipErrorInvalidValue;
    if (p.kind == 6) return dtype == 1 ? mx_fused2_launch(s, p) : hipErrorInvalidValue;
    if (p.kind == 7) return dtype != 1 ? mb4_launch(s, dtype, p) : hipErrorInvalidValue;
    if (p.kind == 8) return expdw_f32_launch(s, dtype, p);
    if (p.kind == 9) return dtype == 2 ? mb6_launch(s, p) : hipErrorInvalidValue;
#include CF_EXP_INC(cf_mbconv_m7_launch)
    const MbEntry* e = mb_find(dtype, p.k, p.s, p.JX, (p.Cout + 31) / 32, p.residual ? 1 : 0);
    if (!e || e->hc != p.HC) return hipErrorInvalidValue;
    return e->fn(s, p);
}

}  // namespace cf
