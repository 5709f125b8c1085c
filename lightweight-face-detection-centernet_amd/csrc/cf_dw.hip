// Depthwise k x k convolution, NHWC, fused asymmetric zero padding + bias + Swish, for gfx950.
//
// Replaces ConvReLU(hidden, hidden, k, stride, groups=hidden) = ZeroPad2d -> Conv2d(groups=C,
// bias=False) -> Swish (model/centernet.py:58-70,111-113) and the ShuffleV2 depthwise conv + BN
// (model/blocks.py:26-29,37-39, symmetric pad k//2, no activation).
//
// The op is pure streaming (1.8 - 12.5 flop/B): the design goal is HBM bandwidth.
//  * the halo tile of one channel chunk is copied HBM -> LDS by the DMA path (global_load_lds_dwordx4), out-of-image chunks from a
//    zero constant (= ZeroPad2d), the chunk's tap weights (fp32) next to it;
//  * stride 1: the compute phase works on strips of four output pixels per 16-byte channel group (dw_strip_kernel below: what it reads
//    from LDS and how many VALU operations it spends per output element); stride 2: one output vector per work item (dw_lds_kernel);
//  * ZeroPad2d, bias, Swish and the bf16 pack are fused: the op reads its unpadded input once and writes its output once.
//  * workgroups take their tile from an XCD-contiguous work list (dw_xcd_remap: the tile below and the other channel chunks of the same
//    pixels under ONE L2; PMC fetch 1.22x -> 0.98x the input bytes on layer1.0, profiles/r06_dw_xcd.md).
// Measured per layer of the 640x640 network at B = 64 (profiles/r06_dw_xcd.md, r06_dw_strip.md): layer1.0 (3x3 stride 2, the largest depthwise of the
// network) 4.77-5.02 TB/s = 60-63 % of the 8 TB/s spec by box, layer0.0 (3x3 stride 1) 4.41-4.80, the other 3x3 layers 3.1-4.6 TB/s, 5x5 2.1-3.0 TB/s.
// The product path fuses this op into the MBConv kernels (the depthwise tensor never reaches HBM); it runs standalone in the
// unfused path (CF_FLAG_NO_FUSE), in cf_op_dwconv and in the ShuffleV2 block.
#include "cf_exp.h"
#include "cf_common.h"
#include "cf_kernels.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace cf {

void dw_pack_weights(const float* w, int C, int k, float* out_host) {
    for (int c = 0; c < C; ++c)
        for (int t = 0; t < k * k; ++t) out_host[(size_t)t * C + c] = w[(size_t)c * k * k + t];
}

#include CF_EXP_INC(cf_dw_0)   // round 1's register-marching kernel (CF_DW_MARCH=1)

// ================================================================== LDS-staged variant
// The input tile (halo included) of one channel chunk is copied HBM -> LDS by the DMA path
// (global_load_lds_dwordx4: no VGPR round trip, a wave keeps 1 KiB per instruction in flight with
// almost no registers), out-of-image chunks are sourced from a 16-byte zero constant (= ZeroPad2d),
// then the workgroup computes from LDS.  Consecutive lanes own consecutive 16-byte channel groups of a
// pixel, so LDS reads are conflict-free and global stores are whole pixels' worth of contiguous bytes.
__device__ __attribute__((aligned(16))) const uint32_t g_dw_zero16[4] = {0u, 0u, 0u, 0u};

struct DwLdsGeom { int Cc, nchunk, cpp, rc, nch, magic_rc, magic_cpp, nt, xcd; size_t lds_bytes;
                   int srow, spx, spart; };      // one pass of the workgroup through the chunk list (blockDim chunks) = srow tile rows + spx pixels + spart chunks
static void dw_set_step(DwLdsGeom& g, int threads) {
    g.srow = threads / g.rc; const int r1 = threads - g.srow * g.rc;
    g.spx = r1 / g.cpp; g.spart = r1 - g.spx * g.cpp;
}
// Workgroups are dealt to the eight XCDs round-robin by their linear id, and each XCD has an L2 of its own: with the plain order
// the tile below (its first halo rows = this tile's last ones) and the other channel chunks of the same pixels (the other half of
// the same 128-byte lines) run on OTHER XCDs and fetch those lines again (round 6, PMC: 1.22x the input bytes on layer1.0).  The
// remap gives XCD k the k-th contiguous eighth of the work list, ordered channel chunk -> tile column -> tile row -> image.
__device__ __forceinline__ unsigned dw_xcd_remap(unsigned l, unsigned n) {
    const unsigned q = n >> 3, r = n & 7u, k = l & 7u;
    return k * q + (k < r ? k : r) + (l >> 3);
}

// The halo tile of one channel chunk, HBM -> LDS by DMA: chunk q = (tile row, pixel, 16-byte part) lands at tile + 16 q; a chunk outside the image comes from the
// zero constant (= ZeroPad2d).  A thread owns chunks tid, tid + blockDim, ...: (row, pixel, part) are taken apart ONCE (two magic divisions) and then advanced by
// the workgroup's step (DwLdsGeom::srow / spx / spart) with adds and two wrap tests, the byte offset inside the image stays in 32 bits.  (Round 6's PMC counts:
// with the divisions, the 64-bit products and a two-way branch per chunk the staging of a stride-2 tile cost as many VALU instructions as its arithmetic,
// several of them quarter-rate integer multiplies.)  The image must be smaller than 4 GiB (launch_dw checks).
template <typename T, int IW>
__device__ __forceinline__ void dw_stage_tile(const DwParams& p, const DwLdsGeom& g, const char* xb, int iy0, int ix0, char* tile, int tid, int nthr) {
    const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
    const int ngroups = (g.nch + 63) >> 6;
    int q = tid;
    int row = (int)__umulhi((unsigned)q, (unsigned)g.magic_rc);
    const int rem = q - row * g.rc;
    int px = (g.cpp == 1 ? rem : (int)__umulhi((unsigned)rem, (unsigned)g.magic_cpp));
    int part = rem - px * g.cpp;
    const unsigned pxb = (unsigned)p.C * (unsigned)sizeof(T), rowb = (unsigned)p.W * pxb;
    unsigned rowoff = (unsigned)(iy0 + row) * rowb;                 // (wraps for rows above the image: never used there)
    const unsigned srowb = (unsigned)g.srow * rowb;
    const char* zsrc = reinterpret_cast<const char*>(g_dw_zero16);
    (void)lane;
    for (int grp = wave; grp < ngroups; grp += nwave) {
        const int gy = iy0 + row, gx = ix0 + px;
        const bool ok = (q < g.nch) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);      // (bitwise: no short-circuit branches)
        const unsigned off = rowoff + __umul24((unsigned)gx, pxb) + (unsigned)part * 16u;
        const char* src = ok ? xb + off : zsrc;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tile + grp * 1024), 16, 0, 0);
        q += nthr; part += g.spart; px += g.spx; row += g.srow; rowoff += srowb;
        if (part >= g.cpp) { part -= g.cpp; px += 1; }
        if (px >= IW) { px -= IW; row += 1; rowoff += rowb; }
    }
}
static int magic_div(int d) { return (int)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }   // exact for n < 2^16, d >= 2 (d = 1 does not fit 32 bits: the kernels test for it)

// ---- one output vector per work item (round 1): the form the STRIDE-2 layers keep (dw_by_stride below)
template <typename T, int KS, int S, int TH, int TW, int ACT, bool BIAS>
__global__ __launch_bounds__(256) void dw_lds_kernel(DwParams p, DwLdsGeom g) {
    constexpr int P = Elem<T>::PER16;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tile = smem;                                            // [IH][IW][Cc] T, linear 16-byte chunks
    float* wl = reinterpret_cast<float*>(smem + (((size_t)g.nch * 16 + 1023) / 1024) * 1024);   // [k*k][Cc]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int x0, y0, b, c0;
    if (g.xcd) {
        unsigned l = dw_xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
        const unsigned ck = l % (unsigned)g.nchunk; l /= (unsigned)g.nchunk;
        const unsigned tx = l % gridDim.x; l /= gridDim.x;
        const unsigned ty = l % gridDim.y;
        x0 = tx * TW; y0 = ty * TH; b = l / gridDim.y; c0 = ck * g.Cc;
    } else {
        x0 = blockIdx.x * TW; y0 = blockIdx.y * TH;
        b = blockIdx.z / g.nchunk; c0 = (blockIdx.z - b * g.nchunk) * g.Cc;
    }
    const int iy0 = y0 * S - p.pad_lo, ix0 = x0 * S - p.pad_lo;
    const char* xb = (const char*)p.x + ((size_t)b * p.H * p.W * p.C + c0) * sizeof(T);

    // ---- DMA the tile: chunk q -> (row, pixel, part); LDS address = q * 16 (lane-linear per wave)
    dw_stage_tile<T, IW>(p, g, xb, iy0, ix0, tile, tid, 256);
    for (int i = tid; i < KS * KS * (g.Cc / 4); i += 256) {       // tap weights of this channel chunk
        const int t = i / (g.Cc / 4), c4 = i - t * (g.Cc / 4);
        st16(wl + t * g.Cc + c4 * 4, ld16(p.w + (size_t)t * p.C + c0 + c4 * 4));
    }
    cf_sync_lds_dma();                                            // every wave drains its DMAs (vmcnt), then the barrier

    // ---- compute: output vector v -> (pixel, channel group)
    const int nvec = TH * TW * g.cpp;
    for (int v = tid; v < nvec; v += 256) {
        const int opx = (g.cpp == 1 ? v : (int)__umulhi((unsigned)v, (unsigned)g.magic_cpp));
        const int cg = v - opx * g.cpp;
        const int oy = opx / TW, ox = opx % TW;
        const int gy = y0 + oy, gx = x0 + ox;
        if (gy >= p.Ho || gx >= p.Wo) continue;
        const char* tb = tile + ((size_t)((oy * S) * IW + ox * S) * g.cpp + cg) * 16;
        const float* wb = wl + cg * P;
        float d[P];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                float ev[P], wv[P];
                unpack16<T>(ld16(tb + (size_t)(ky * IW + kx) * g.cpp * 16), ev);
                const float* wt = wb + (ky * KS + kx) * g.Cc;
                unpack16<float>(ld16(wt), wv);
                if constexpr (P == 8) unpack16<float>(ld16(wt + 4), wv + 4);
                if (ky == 0 && kx == 0) {
#pragma unroll
                    for (int e = 0; e < P; ++e) d[e] = ev[e] * wv[e];
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) d[e] = fmaf(ev[e], wv[e], d[e]);
                }
            }
        const int ch = c0 + cg * P;
#pragma unroll
        for (int e = 0; e < P; ++e) d[e] = d[e] + (BIAS ? p.bias[ch + e] : 0.0f);
        act_arr<ACT, P>(d);
        u32x4* dstp = reinterpret_cast<u32x4*>((char*)p.y + ((((size_t)b * p.Ho + gy) * p.Wo + gx) * p.C + ch) * sizeof(T));
        *dstp = pack16<T>(d);
    }
}


template <typename T, int KS, int S, int TH, int TW>
static hipError_t dw_lds_dispatch(hipStream_t s, const DwParams& p, int cap_default = 48) {
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int P = 16 / (int)sizeof(T);
    // channel chunk: largest divisor of C (multiple of one 16-byte group) whose tile fits the budget (48 KiB unless the caller's table says otherwise: tile +
    // tap weights stay under the 64 KiB default dynamic-LDS limit; a smaller budget = more workgroups per CU, shorter DMA runs per pixel)
    int Cc = 0;
    static const int cap_env = cf_ab_int("CF_DW_CAP2", 0);           // A/B: LDS budget of the tile in KB
    const int cap_kb = cap_env > 0 ? cap_env : cap_default;
    for (int c = p.C; c >= P; c -= P)
        if (p.C % c == 0 && (size_t)IH * IW * c * sizeof(T) <= (size_t)cap_kb * 1024) { Cc = c; break; }
    if (!Cc) return hipErrorInvalidValue;
    DwLdsGeom g;
    g.Cc = Cc; g.nchunk = p.C / Cc; g.cpp = Cc / P; g.rc = IW * g.cpp; g.nch = IH * g.rc;
    if (g.nch >= 65536) return hipErrorInvalidValue;
    g.magic_rc = magic_div(g.rc); g.magic_cpp = magic_div(g.cpp);
    g.nt = 0;          // (non-temporal DMA loads cost 8-35 %, non-temporal stores nothing: profiles/r01_dw_variants.md, r06_dw_xcd.md)
    dw_set_step(g, 256);
    { static const int xcd_env = cf_ab_int("CF_DW_XCD", 1); g.xcd = xcd_env; }
    g.lds_bytes = (((size_t)g.nch * 16 + 1023) / 1024) * 1024 + (size_t)KS * KS * Cc * 4;
    dim3 grid((p.Wo + TW - 1) / TW, (p.Ho + TH - 1) / TH, p.B * g.nchunk), blk(256);
    const bool bias = p.bias != nullptr;
    set_kernel_tag("void cf::dw_lds_kernel<%s, %d, %d, %d, %d, %d, %s>(cf::DwParams, cf::DwLdsGeom)", type_tag<T>(), KS, S, TH, TW,
                   p.act, bias ? "true" : "false");
#define CF_DW_LAUNCH(ACT, BIAS) \
    hipLaunchKernelGGL((dw_lds_kernel<T, KS, S, TH, TW, ACT, BIAS>), grid, blk, g.lds_bytes, s, p, g); return hipGetLastError();
    if (p.act == 1 && !bias) { CF_DW_LAUNCH(1, false) }
    if (p.act == 0 && bias) { CF_DW_LAUNCH(0, true) }
    if (p.act == 0 && !bias) { CF_DW_LAUNCH(0, false) }
    CF_DW_LAUNCH(1, true)
#undef CF_DW_LAUNCH
}


// ================================================================== LDS-staged, strip form (round 6)
// Same staging as dw_lds_kernel (the halo tile of one channel chunk by LDS DMA, tap weights as fp32 next to it); the compute phase is
// restructured around what bound the first form -- LDS traffic and unpack instructions, not HBM: there every output vector (8 channels
// of one pixel) read k*k tile chunks AND k*k x 32 bytes of weights from LDS (54 B per output element at 3x3, 150 B at 5x5, against
// ~128 B per cycle and CU) and unpacked every tile chunk once per tap.  Here a work item is a STRIP of SX = 4 output pixels of one row
// for one 16-byte channel group: per kernel row it reads the (SX - 1) S + k tile chunks behind the strip once, unpacks them once, and
// reads that row's k weight vectors once for all four outputs -- 18 B (3x3) / 45 B (5x5) of LDS traffic per output element, 13.5 / 35
// VALU operations instead of 18 / 50, as v_pk_fma_f32 on channel pairs.  The taps of an output accumulate in the same order as before
// (ky outer, kx inner, first tap a plain multiply): results are bit-identical to dw_lds_kernel.  Items are laid out channel group
// fastest, so consecutive lanes read consecutive 16-byte chunks (conflict-free) and store whole pixels' worth of contiguous bytes.
template <typename T, int KS, int S, int TH, int TW>
__global__ __launch_bounds__(256) void dw_strip_kernel(DwParams p, DwLdsGeom g, int ntx, int nty, int ntiles) {
    constexpr int P = Elem<T>::PER16, H2 = P / 2, SX = 4;
    constexpr int IW = (TW - 1) * S + KS;
    constexpr int WIN = (SX - 1) * S + KS, NSTRIP = TW / SX;
    static_assert(TW % SX == 0, "strips of four output pixels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two buffers of [tile | tap weights]: the DMA of tile t + 1 runs underneath the arithmetic of tile t (persistent workgroups)
    const size_t tile_bytes = (((size_t)g.nch * 16 + 1023) / 1024) * 1024;
    const size_t buf_bytes = tile_bytes + (((size_t)KS * KS * g.Cc * 4 + 1023) / 1024) * 1024;       // (the weights arrive in whole 1 KiB DMA groups too)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
    const int ngroups = (g.nch + 63) >> 6;

    // tile t -> (image, channel chunk, tile row, tile column); x fastest, so the tiles in flight at any moment are neighbours (shared halo lines in L2)
    auto decode = [&](int t, int& b, int& c0, int& y0, int& x0) {
        if (g.xcd && (int)gridDim.x == ntiles) {                      // one workgroup per tile: XCD-contiguous work list (dw_xcd_remap)
            t = (int)dw_xcd_remap((unsigned)t, (unsigned)ntiles);
            const int ck = t % g.nchunk; t /= g.nchunk;
            const int tx = t % ntx; t /= ntx;
            const int ty = t % nty; b = t / nty;
            c0 = ck * g.Cc; y0 = ty * TH; x0 = tx * TW;
            return;
        }
        const int tx = t % ntx; t /= ntx;
        const int ty = t % nty; t /= nty;
        const int ck = t % g.nchunk; b = t / g.nchunk;
        c0 = ck * g.Cc; y0 = ty * TH; x0 = tx * TW;
    };
    auto stage = [&](int t, char* buf) {
        int b, c0, y0, x0; decode(t, b, c0, y0, x0);
        const int iy0 = y0 * S - p.pad_lo, ix0 = x0 * S - p.pad_lo;
        const char* xb = (const char*)p.x + ((size_t)b * p.H * p.W * p.C + c0) * sizeof(T);
        dw_stage_tile<T, IW>(p, g, xb, iy0, ix0, buf, tid, nthr);
        // tap weights of the tile's channel chunk: whole 1 KiB groups by DMA as well ([k*k][Cc] fp32 = k*k*Cc/4 chunks of 16 bytes)
        char* wdst = buf + tile_bytes;
        const int nwch = KS * KS * (g.Cc / 4), nwg = (nwch + 63) >> 6;
        for (int grp = wave; grp < nwg; grp += nwave) {
            const int i = grp * 64 + lane;
            const int ic = i < nwch ? i : nwch - 1;
            const int t2 = ic / (g.Cc / 4), c4 = ic - t2 * (g.Cc / 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.w + (size_t)t2 * p.C + c0 + c4 * 4),
                                             (__attribute__((address_space(3))) void*)(wdst + grp * 1024), 16, 0, 0);
        }
    };
    auto unpack2 = [](const u32x4& c, f32x2* f) {                 // one 16-byte chunk -> H2 channel pairs
        if constexpr (P == 8) {
            f[0].x = bf16lo(c.x); f[0].y = bf16hi(c.x); f[1].x = bf16lo(c.y); f[1].y = bf16hi(c.y);
            f[2].x = bf16lo(c.z); f[2].y = bf16hi(c.z); f[3].x = bf16lo(c.w); f[3].y = bf16hi(c.w);
        } else {
            f[0].x = __uint_as_float(c.x); f[0].y = __uint_as_float(c.y); f[1].x = __uint_as_float(c.z); f[1].y = __uint_as_float(c.w);
        }
    };

    int t = blockIdx.x;
    if (t < ntiles) stage(t, smem);
    for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
        char* tile = smem + (size_t)(it & 1) * buf_bytes;
        const float* wl = reinterpret_cast<const float*>(tile + tile_bytes);
        cf_sync_lds_dma();          // tile t has landed for every wave, and every wave is done with tile t - 1 (the other buffer)
        if (t + (int)gridDim.x < ntiles) stage(t + gridDim.x, smem + (size_t)((it + 1) & 1) * buf_bytes);
        int b, c0, y0, x0; decode(t, b, c0, y0, x0);
        const int nitem = TH * NSTRIP * g.cpp;
        for (int v = tid; v < nitem; v += nthr) {
            const int sidx = (g.cpp == 1 ? v : (int)__umulhi((unsigned)v, (unsigned)g.magic_cpp));
            const int cg = v - sidx * g.cpp;
            const int oy = sidx / NSTRIP, xs = sidx - oy * NSTRIP;
            const int gy = y0 + oy, gx0 = x0 + xs * SX;
            if (gy >= p.Ho || gx0 >= p.Wo) continue;
            const char* tb = tile + ((size_t)((oy * S) * IW + xs * SX * S) * g.cpp + cg) * 16;
            const float* wb = wl + cg * P;
            f32x2 acc[SX][H2];
            // One kernel row at a time: fully unrolled, the compiler hoists every row's window and weights (460 VGPRs at 5x5, one wave per SIMD).  The FIRST
            // row is peeled off the rolled loop: its first tap is a plain multiply (the order dw_lds_kernel fixes), and with `ky == 0` a run-time test inside
            // the loop the compiler issued BOTH forms for every kx = 0 tap of every row and selected between them -- 16 v_pk_mul_f32 + 32 v_cndmask_b32 per
            // row next to the row's 48 (3x3) / 80 (5x5) v_pk_fma_f32 (ISA of round 6's first version; same results, a third fewer tap instructions)
            auto row = [&](auto first, int ky) {
                f32x2 xw[WIN][H2];
#pragma unroll
                for (int c = 0; c < WIN; ++c) unpack2(ld16(tb + (size_t)(ky * IW + c) * g.cpp * 16), xw[c]);
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const float* wt = wb + (ky * KS + kx) * g.Cc;
                    f32x2 w2[H2];
                    {
                        const u32x4 w0 = ld16(wt);
                        w2[0].x = __uint_as_float(w0.x); w2[0].y = __uint_as_float(w0.y); w2[1].x = __uint_as_float(w0.z); w2[1].y = __uint_as_float(w0.w);
                        if constexpr (P == 8) {
                            const u32x4 w1 = ld16(wt + 4);
                            w2[2].x = __uint_as_float(w1.x); w2[2].y = __uint_as_float(w1.y); w2[3].x = __uint_as_float(w1.z); w2[3].y = __uint_as_float(w1.w);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < SX; ++j)
#pragma unroll
                        for (int h = 0; h < H2; ++h) {
                            if (decltype(first)::value && kx == 0) acc[j][h] = xw[j * S + kx][h] * w2[h];
                            else acc[j][h] = fma2(xw[j * S + kx][h], w2[h], acc[j][h]);
                        }
                }
            };
            row(std::true_type{}, 0);
#pragma unroll 1
            for (int ky = 1; ky < KS; ++ky) row(std::false_type{}, ky);
            const int ch = c0 + cg * P;
            float bv[P];
#pragma unroll
            for (int e = 0; e < P; ++e) bv[e] = p.bias ? p.bias[ch + e] : 0.0f;
            char* yrow = (char*)p.y + ((((size_t)b * p.Ho + gy) * p.Wo + gx0) * p.C + ch) * sizeof(T);
#pragma unroll
            for (int j = 0; j < SX; ++j) {
                if (gx0 + j >= p.Wo) break;
                float d[P];
#pragma unroll
                for (int h = 0; h < H2; ++h) { d[2 * h] = acc[j][h].x + bv[2 * h]; d[2 * h + 1] = acc[j][h].y + bv[2 * h + 1]; }
                if (p.act == 1) act_arr<1, P>(d);
                *reinterpret_cast<u32x4*>(yrow + (size_t)j * p.C * sizeof(T)) = pack16<T>(d);
            }
        }
    }
}

#include CF_EXP_INC(cf_dw_5)   // the strip form in two row bands (dw_band_kernel): measured, +2-7 % on some layers, -7 % on others -- experiments build only
struct DwTileCfg { int th, tw; };
template <typename T, int KS, int S, int TH, int TW>
static hipError_t dw_strip_launch(hipStream_t s, const DwParams& p, const DwLdsGeom& g, int threads) {
    auto kfn = dw_strip_kernel<T, KS, S, TH, TW>;
    static thread_local bool big[64] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    if (!big[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return e;
        big[dev & 63] = true;
    }
    const int ntx = (p.Wo + TW - 1) / TW, nty = (p.Ho + TH - 1) / TH;
    const long long nt = (long long)p.B * g.nchunk * nty * ntx;
    if (nt > 0x7fffffffLL) return hipErrorInvalidValue;
#include CF_EXP_INC(cf_dw_6)   // CF_DW_BAND=1: dw_band_kernel
    // CF_DW_WGS = n > 0: n PERSISTENT workgroups per CU, each walking tiles with two LDS buffers (the DMA of the next tile under the arithmetic of
    // this one); 0: one workgroup per tile, one buffer -- the overlap comes from the other workgroups of the CU
    static const int per_cu = cf_ab_int("CF_DW_WGS", 0);
    int ncu = 256; { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ncu = pr.multiProcessorCount; }
    const bool persist = per_cu > 0 && nt > (long long)ncu * per_cu;
    const int nwg = persist ? ncu * per_cu : (int)nt;
    set_kernel_tag("void cf::dw_strip_kernel<%s, %d, %d, %d, %d>(cf::DwParams, cf::DwLdsGeom, int, int, int)", type_tag<T>(), KS, S, TH, TW);
    DwLdsGeom gs = g; dw_set_step(gs, threads);
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(threads), (persist ? 2 : 1) * g.lds_bytes, s, p, gs, ntx, nty, (int)nt);
    return hipGetLastError();
}

// geometry of one candidate tile: the largest channel chunk (a divisor of C, whole 16-byte groups) whose tile + tap weights fit `cap`
template <typename T>
static bool dw_strip_geom(const DwParams& p, int TH, int TW, size_t cap, DwLdsGeom& g, int& threads, double& score) {
    constexpr int P = 16 / (int)sizeof(T);
    const int IH = (TH - 1) * p.s + p.k, IW = (TW - 1) * p.s + p.k;
    int Cc = 0;
    for (int c = p.C; c >= P; c -= P) {
        if (p.C % c) continue;
        const size_t bytes = (((size_t)IH * IW * (c / P) * 16 + 1023) / 1024) * 1024 + (((size_t)p.k * p.k * c * 4 + 1023) / 1024) * 1024;
        if (bytes <= cap && (size_t)IH * IW * (c / P) < 65536) { Cc = c; break; }
    }
    if (!Cc) return false;
    g.Cc = Cc; g.nchunk = p.C / Cc; g.cpp = Cc / P; g.rc = IW * g.cpp; g.nch = IH * g.rc;
    g.magic_rc = magic_div(g.rc); g.magic_cpp = magic_div(g.cpp); g.nt = 0;
    { static const int xcd_env = cf_ab_int("CF_DW_XCD", 1); g.xcd = xcd_env; }
    g.lds_bytes = (((size_t)g.nch * 16 + 1023) / 1024) * 1024 + (((size_t)p.k * p.k * Cc * 4 + 1023) / 1024) * 1024;
    const int items = TH * (TW / 4) * g.cpp;
    threads = items >= 256 ? 256 : (items + 63) / 64 * 64;
    const int passes = (items + threads - 1) / threads;
    const double eff_thr = (double)items / ((double)passes * threads);
    const int nty = (p.Ho + TH - 1) / TH, ntx = (p.Wo + TW - 1) / TW;
    const double eff_sp = (double)p.Ho * p.Wo / ((double)nty * TH * ntx * TW);
    const double halo = (double)(TH * p.s) * (TW * p.s) / ((double)IH * IW);
    const double run = g.cpp >= 4 ? 1.0 : 0.6 + 0.1 * g.cpp;                  // short DMA runs (one or two chunks per pixel) waste line fetches
    const double fill = threads >= 256 ? 1.0 : 0.85;
    score = eff_thr * eff_sp * (0.5 + 0.5 * halo) * run * fill;
    return true;
}

#define CF_DW_TILES(X) X(4, 16) X(4, 32) X(8, 16) X(8, 32) X(16, 16) X(16, 32) X(8, 40) X(10, 20) X(20, 20)
// Tile and LDS budget per shape class: the best of an exhaustive sweep (nine tiles x five LDS caps per layer of the 640x640 network at B = 64,
// tools/dw_sweep2.sh, profiles/r06_dw_strip.md); shapes the table does not fit fall back to the score of dw_strip_geom.
static void dw_strip_table(int k, int s, int Wo, int& th, int& tw, int& cap_kb) {
    (void)s;                                                         // (stride 1 only: dw_by_stride)
    if (k == 3) { if (Wo >= 256) { th = 16; tw = 32; cap_kb = 60; } else if (Wo >= 96) { th = 8; tw = 40; cap_kb = 48; } else { th = 10; tw = 20; cap_kb = 48; } }
    else { if (Wo >= 64) { th = 16; tw = 16; cap_kb = 32; } else if (Wo >= 32) { th = 20; tw = 20; cap_kb = 48; } else { th = 10; tw = 20; cap_kb = 32; } }
}
template <typename T, int KS, int S>
static hipError_t dw_strip_pick(hipStream_t s, const DwParams& p) {
    static const int force = cf_ab_int("CF_DW_TILE", -1);           // A/B: index into the tile list
    static const int cap_env = cf_ab_int("CF_DW_CAP", 0);           //      LDS budget of the tile buffer in KB
    int tth = 0, ttw = 0, tcap = 48;
    dw_strip_table(KS, S, p.Wo, tth, ttw, tcap);
    int best = -1, bthreads = 0, idx = 0; double bscore = -1.0; DwLdsGeom bg{};
    // first choice: the table's tile; otherwise (or when the table's tile cannot hold one 16-byte channel group) the best score
    for (int pass = 0; pass < 2 && best < 0; ++pass) {
        const size_t cap = (size_t)(cap_env > 0 ? cap_env : (pass == 0 ? tcap : 48)) * 1024;
        idx = 0;
#define CF_DW_TRY(TH_, TW_) { DwLdsGeom g{}; int th = 0; double sc = 0; \
            const bool want = force >= 0 ? idx == force : (pass == 0 ? (TH_ == tth && TW_ == ttw) : true); \
            if (want && dw_strip_geom<T>(p, TH_, TW_, cap, g, th, sc) && (force >= 0 || pass == 0 || sc > bscore)) { best = idx; bscore = sc; bg = g; bthreads = th; } ++idx; }
        CF_DW_TILES(CF_DW_TRY)
#undef CF_DW_TRY
        if (force >= 0) break;
    }
    if (best < 0) return hipErrorInvalidValue;
    idx = 0;
#define CF_DW_GO(TH_, TW_) if (idx++ == best) return dw_strip_launch<T, KS, S, TH_, TW_>(s, p, bg, bthreads);
    CF_DW_TILES(CF_DW_GO)
#undef CF_DW_GO
    return hipErrorInvalidValue;
}
// Stride 1: the strip form (3.3-4.2 TB/s at 3x3 against 2.5-3.8, 1.9-2.3 TB/s at 5x5 against 1.3-1.7).  Stride 2: round 1's one-vector form -- a strip of four
// stride-2 outputs sits on 9 / 11 input columns per row, the reuse is small and the work items are four times fewer: measured 4.18 / 2.73 / 3.73 / 1.84 TB/s
// (layer1.0 / 2.0 / 3.0 / 5.0, best of 45 tile x LDS-budget combinations each) against 4.63 / 2.70 / 3.97 / 1.98 (profiles/r06_dw_strip.md).
template <typename T>
static hipError_t dw_by_stride(hipStream_t s, const DwParams& p) {
    if (p.k == 3 && p.s == 1) return dw_strip_pick<T, 3, 1>(s, p);
    if (p.k == 5 && p.s == 1) return dw_strip_pick<T, 5, 1>(s, p);
#include CF_EXP_INC(cf_dw_4)   // CF_DW2_TILE: tile sweep of the stride-2 form
    if (p.k == 3 && p.s == 2) {
        if (p.Ho <= 20 && p.Wo <= 20 && p.Ho > 8) return dw_lds_dispatch<T, 3, 2, 10, 20>(s, p);      // small late maps: one tile = the whole map
        // (tile x LDS budget sweep with the XCD-contiguous order, gpurun_out/r06q_sweep_s2.txt: layer3.0 3.77 -> 4.33 TB/s at 24 KB)
        return p.Wo >= 64 ? dw_lds_dispatch<T, 3, 2, 4, 32>(s, p, 48) : dw_lds_dispatch<T, 3, 2, 4, 16>(s, p, 24);
    }
    // (same sweep: layer2.0 2.55 -> 2.93 TB/s on 8x16 tiles at 24 KB, layer5.0 1.94 -> 2.16 at 32 KB)
    if (p.k == 5 && p.s == 2) return p.Wo >= 64 ? dw_lds_dispatch<T, 5, 2, 8, 16>(s, p, 24) : dw_lds_dispatch<T, 5, 2, 4, 32>(s, p, 32);
    return hipErrorInvalidValue;
}

#include CF_EXP_INC(cf_dw_2)   // ... and their per-shape dispatch

hipError_t launch_dw(hipStream_t s, int dtype, const DwParams& p) {
    if (p.B <= 0) return hipSuccess;
    if (p.C % 8) return hipErrorInvalidValue;
    if ((unsigned long long)p.H * p.W * p.C * (dtype != 1 ? 4ull : 2ull) >= (1ull << 32)) return hipErrorInvalidValue;       // 32-bit byte offsets inside an image (dw_stage_tile)
#include CF_EXP_INC(cf_dw_3)   // CF_DW_MARCH=1 / CF_DW_STRIP=0: the older kernels, for A/B runs
    return dtype != 1 ? dw_by_stride<float>(s, p) : dw_by_stride<bf16_t>(s, p);      // dtype 2: fp32 storage, no GEMM here
}

}  // namespace cf
