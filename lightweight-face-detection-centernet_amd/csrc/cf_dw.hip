// Depthwise k x k convolution, NHWC, fused asymmetric zero padding + bias + Swish, for gfx950.
//
// Replaces ConvReLU(hidden, hidden, k, stride, groups=hidden) = ZeroPad2d -> Conv2d(groups=C,
// bias=False) -> Swish (model/centernet.py:58-70,111-113) and the ShuffleV2 depthwise conv + BN
// (model/blocks.py:26-29,37-39, symmetric pad k//2, no activation).
//
// The op is pure streaming (1.8 - 12.5 flop/B): the design goal is HBM bandwidth.
//  * NHWC makes an image row one contiguous run of W*C elements; a thread owns VEC consecutive
//    channels of one output column, so a wave's loads are back-to-back 16-byte pieces of that run
//    (fully coalesced, whole 128-B lines).
//  * A thread marches DOWN a strip of TH output rows: every input row of the strip is loaded once
//    per thread (k horizontally shifted vectors; the shifts overlap the neighbouring lanes' loads
//    and are served by the CU's L1/TA, not by HBM) and is folded into the <= ceil(k/s) output rows
//    it contributes to while they are live in registers -- vertical reuse never leaves the VGPRs.
//  * The k*k*VEC weights of the thread's channels sit in registers for the whole strip (fp32).
//  * ZeroPad2d, Swish and the bf16 pack are fused: the op reads its unpadded input once and
//    writes its output once.
#include "cf_common.h"
#include "cf_kernels.h"
#include <cstdlib>
#include <cstring>

namespace cf {

void dw_pack_weights(const float* w, int C, int k, float* out_host) {
    for (int c = 0; c < C; ++c)
        for (int t = 0; t < k * k; ++t) out_host[(size_t)t * C + c] = w[(size_t)c * k * k + t];
}

template <typename T, int VEC> struct VecIO;
template <> struct VecIO<bf16_t, 8> {
    static __device__ __forceinline__ void load(const void* p, float* f) { unpack16<bf16_t>(ld16(p), f); }
    static __device__ __forceinline__ void store(void* p, const float* f) { st16(p, pack16<bf16_t>(f)); }
};
template <> struct VecIO<bf16_t, 4> {
    static __device__ __forceinline__ void load(const void* p, float* f) {
        u32x2 c = *reinterpret_cast<const u32x2*>(p);
        f[0] = bf16lo(c.x); f[1] = bf16hi(c.x); f[2] = bf16lo(c.y); f[3] = bf16hi(c.y);
    }
    static __device__ __forceinline__ void store(void* p, const float* f) {
        u32x2 c; c.x = pack_bf16x2(f[0], f[1]); c.y = pack_bf16x2(f[2], f[3]);
        *reinterpret_cast<u32x2*>(p) = c;
    }
};
template <> struct VecIO<float, 4> {
    static __device__ __forceinline__ void load(const void* p, float* f) { unpack16<float>(ld16(p), f); }
    static __device__ __forceinline__ void store(void* p, const float* f) { st16(p, pack16<float>(f)); }
};

template <typename T, int KS, int S, int VEC, int TH, int ACT, bool BIAS>
__global__ __launch_bounds__(256) void dw_kernel(DwParams p) {
    const int CG = p.C / VEC;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= p.Wo * CG) return;
    const int xo = f / CG;
    const int c0 = (f - xo * CG) * VEC;
    const int y0 = blockIdx.y * TH;
    const int b = blockIdx.z;

    float wreg[KS * KS][VEC];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t)
#pragma unroll
        for (int v4 = 0; v4 < VEC / 4; ++v4)           // 16-byte weight loads (c0 is a multiple of 4)
            unpack16<float>(ld16(p.w + (size_t)t * p.C + c0 + 4 * v4), &wreg[t][4 * v4]);
    float breg[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) breg[e] = BIAS ? p.bias[c0 + e] : 0.0f;

    const int ix0 = xo * S - p.pad_lo;
    bool xok[KS];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) xok[kx] = (unsigned)(ix0 + kx) < (unsigned)p.W;

    const T* xin = (const T*)p.x + (size_t)b * p.H * p.W * p.C + c0;
    T* yout = (T*)p.y + ((size_t)b * p.Ho * p.Wo + xo) * p.C + c0;

    float acc[TH][VEC];
#pragma unroll
    for (int t = 0; t < TH; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[t][e] = 0.0f;

    constexpr int ROWS = (TH - 1) * S + KS;       // input rows touched by the strip
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int iy = y0 * S - p.pad_lo + r;
        const bool yok = (unsigned)iy < (unsigned)p.H;
        float v[KS][VEC];
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            if (yok && xok[kx]) {
                VecIO<T, VEC>::load(xin + ((size_t)iy * p.W + (ix0 + kx)) * p.C, v[kx]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[kx][e] = 0.0f;
            }
        }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            // output row t of the strip uses input row r when t*S + ky == r
            if ((r - ky) >= 0 && (r - ky) % S == 0 && (r - ky) / S < TH) {
                const int t = (r - ky) / S;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        acc[t][e] = fmaf(v[kx][e], wreg[ky * KS + kx][e], acc[t][e]);
            }
        }
        // output row t is complete after its last input row r = t*S + KS-1
        if (r >= KS - 1 && (r - (KS - 1)) % S == 0) {
            const int t = (r - (KS - 1)) / S;
            const int yo = y0 + t;
            if (yo < p.Ho) {
                float o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = acc[t][e] + breg[e];
                act_arr<ACT, VEC>(o);
                VecIO<T, VEC>::store(yout + (size_t)yo * p.Wo * p.C, o);
            }
        }
    }
}

template <typename T, int KS, int S, int VEC, int TH>
static hipError_t dw_dispatch(hipStream_t s, const DwParams& p) {
    dim3 blk(256);
    dim3 grid((unsigned)((p.Wo * (p.C / VEC) + 255) / 256), (unsigned)((p.Ho + TH - 1) / TH), (unsigned)p.B);
    const bool bias = p.bias != nullptr;
    set_kernel_tag("void cf::dw_kernel<%s, %d, %d, %d, %d, %d, %s>(cf::DwParams)", type_tag<T>(), KS, S, VEC, TH, p.act, bias ? "true" : "false");
    if (p.act == 1 && !bias) hipLaunchKernelGGL((dw_kernel<T, KS, S, VEC, TH, 1, false>), grid, blk, 0, s, p);
    else if (p.act == 0 && bias) hipLaunchKernelGGL((dw_kernel<T, KS, S, VEC, TH, 0, true>), grid, blk, 0, s, p);
    else if (p.act == 0 && !bias) hipLaunchKernelGGL((dw_kernel<T, KS, S, VEC, TH, 0, false>), grid, blk, 0, s, p);
    else hipLaunchKernelGGL((dw_kernel<T, KS, S, VEC, TH, 1, true>), grid, blk, 0, s, p);
    return hipGetLastError();
}

// ================================================================== LDS-staged variant
// The input tile (halo included) of one channel chunk is copied HBM -> LDS by the DMA path
// (global_load_lds_dwordx4: no VGPR round trip, a wave keeps 1 KiB per instruction in flight with
// almost no registers), out-of-image chunks are sourced from a 16-byte zero constant (= ZeroPad2d),
// then every thread computes output vectors from LDS: k*k ds_read_b128 of the tile + the tap weights
// (fp32, staged once per workgroup).  Consecutive lanes own consecutive 16-byte channel groups of a
// pixel, so LDS reads are conflict-free and global stores are whole pixels' worth of contiguous bytes.
__device__ __attribute__((aligned(16))) const uint32_t g_dw_zero16[4] = {0u, 0u, 0u, 0u};

struct DwLdsGeom { int Cc, nchunk, cpp, rc, nch, magic_rc, magic_cpp, nt; size_t lds_bytes; };

template <typename T, int KS, int S, int TH, int TW, int ACT, bool BIAS>
__global__ __launch_bounds__(256) void dw_lds_kernel(DwParams p, DwLdsGeom g) {
    constexpr int P = Elem<T>::PER16;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tile = smem;                                            // [IH][IW][Cc] T, linear 16-byte chunks
    float* wl = reinterpret_cast<float*>(smem + (((size_t)g.nch * 16 + 1023) / 1024) * 1024);   // [k*k][Cc]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int b = blockIdx.z / g.nchunk, c0 = (blockIdx.z - b * g.nchunk) * g.Cc;
    const int iy0 = y0 * S - p.pad_lo, ix0 = x0 * S - p.pad_lo;
    const char* xb = (const char*)p.x + ((size_t)b * p.H * p.W * p.C + c0) * sizeof(T);

    // ---- DMA the tile: chunk q -> (row, pixel, part); LDS address = q * 16 (lane-linear per wave)
    const int ngroups = (g.nch + 63) >> 6;
    for (int grp = wave; grp < ngroups; grp += 4) {
        const int q = grp * 64 + lane;
        const int row = __umulhi((unsigned)q, (unsigned)g.magic_rc);
        const int rem = q - row * g.rc;
        const int px = __umulhi((unsigned)rem, (unsigned)g.magic_cpp);
        const int part = rem - px * g.cpp;
        const int gy = iy0 + row, gx = ix0 + px;
        const bool ok = q < g.nch && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const char* src = ok ? xb + ((size_t)gy * p.W + gx) * p.C * sizeof(T) + part * 16
                             : reinterpret_cast<const char*>(g_dw_zero16);
        if (g.nt & 1)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(tile + grp * 1024), 16, 0, 2);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(tile + grp * 1024), 16, 0, 0);
    }
    for (int i = tid; i < KS * KS * (g.Cc / 4); i += 256) {       // tap weights of this channel chunk
        const int t = i / (g.Cc / 4), c4 = i - t * (g.Cc / 4);
        st16(wl + t * g.Cc + c4 * 4, ld16(p.w + (size_t)t * p.C + c0 + c4 * 4));
    }
    cf_sync_lds_dma();                                            // every wave drains its DMAs (vmcnt), then the barrier

    // ---- compute: output vector v -> (pixel, channel group)
    const int nvec = TH * TW * g.cpp;
    for (int v = tid; v < nvec; v += 256) {
        const int opx = __umulhi((unsigned)v, (unsigned)g.magic_cpp);
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
        if (g.nt & 2) __builtin_nontemporal_store(pack16<T>(d), dstp);
        else *dstp = pack16<T>(d);
    }
}

static int magic_div(int d) { return (int)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }   // exact for n < 2^16

template <typename T, int KS, int S, int TH, int TW>
static hipError_t dw_lds_dispatch(hipStream_t s, const DwParams& p) {
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int P = 16 / (int)sizeof(T);
    // channel chunk: largest divisor of C (multiple of one 16-byte group) whose tile fits 48 KiB (tile + tap weights stay under the 64 KiB default dynamic-LDS limit)
    int Cc = 0;
    for (int c = p.C; c >= P; c -= P)
        if (p.C % c == 0 && (size_t)IH * IW * c * sizeof(T) <= 48 * 1024) { Cc = c; break; }
    if (!Cc) return hipErrorInvalidValue;
    DwLdsGeom g;
    g.Cc = Cc; g.nchunk = p.C / Cc; g.cpp = Cc / P; g.rc = IW * g.cpp; g.nch = IH * g.rc;
    if (g.nch >= 65536) return hipErrorInvalidValue;
    g.magic_rc = magic_div(g.rc); g.magic_cpp = magic_div(g.cpp);
    { static const int nt_env = cf_ab_int("CF_DW_NT", 0); g.nt = nt_env; }
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

template <typename T>
static hipError_t dw_lds_by_shape(hipStream_t s, const DwParams& p) {
    static const int tv = cf_ab_int("CF_DW_TILE", 0);     // A/B of tile shapes
    if (tv == 1) {
        if (p.k == 3 && p.s == 1) return dw_lds_dispatch<T, 3, 1, 16, 16>(s, p);
        if (p.k == 3 && p.s == 2) return dw_lds_dispatch<T, 3, 2, 8, 16>(s, p);
        if (p.k == 5 && p.s == 1) return dw_lds_dispatch<T, 5, 1, 16, 16>(s, p);
        if (p.k == 5 && p.s == 2) return dw_lds_dispatch<T, 5, 2, 8, 16>(s, p);
    }
    if (tv == 2) {
        if (p.k == 3 && p.s == 1) return dw_lds_dispatch<T, 3, 1, 8, 32>(s, p);
        if (p.k == 3 && p.s == 2) return dw_lds_dispatch<T, 3, 2, 4, 32>(s, p);
        if (p.k == 5 && p.s == 1) return dw_lds_dispatch<T, 5, 1, 8, 32>(s, p);
        if (p.k == 5 && p.s == 2) return dw_lds_dispatch<T, 5, 2, 4, 32>(s, p);
    }
    // small late maps (20x20 at 640x640 input): one tile = the whole map (no spatial padding waste)
    if (p.Ho <= 20 && p.Wo <= 20 && p.Ho > 8) {
        if (p.k == 3 && p.s == 1) return dw_lds_dispatch<T, 3, 1, 20, 20>(s, p);
        if (p.k == 5 && p.s == 1) return dw_lds_dispatch<T, 5, 1, 20, 20>(s, p);
        if (p.k == 3 && p.s == 2) return dw_lds_dispatch<T, 3, 2, 10, 20>(s, p);
        // 5x5 stride 2: the 43-wide halo tile forces a 24-channel chunk; 4x32 tiles measured faster
    }
    // measured on MI355X, B=64 (profiles/r01_dw_variants.md): wide tiles for the big early maps
    // (less halo per byte), smaller ones where the channel chunk would otherwise drop below a pixel
    if (p.k == 3 && p.s == 1) return p.C <= 32 ? dw_lds_dispatch<T, 3, 1, 16, 16>(s, p) : dw_lds_dispatch<T, 3, 1, 8, 16>(s, p);
    if (p.k == 3 && p.s == 2) return p.Wo >= 64 ? dw_lds_dispatch<T, 3, 2, 4, 32>(s, p) : dw_lds_dispatch<T, 3, 2, 4, 16>(s, p);
    if (p.k == 5 && p.s == 1) return p.Wo >= 32 ? dw_lds_dispatch<T, 5, 1, 16, 16>(s, p) : dw_lds_dispatch<T, 5, 1, 8, 32>(s, p);
    if (p.k == 5 && p.s == 2) return p.Wo >= 64 ? dw_lds_dispatch<T, 5, 2, 4, 16>(s, p) : dw_lds_dispatch<T, 5, 2, 4, 32>(s, p);
    return hipErrorInvalidValue;
}

template <typename T, int VEC3, int VEC5>
static hipError_t dw_by_shape(hipStream_t s, const DwParams& p) {
    if (p.k == 3 && p.s == 1) return dw_dispatch<T, 3, 1, VEC3, 8>(s, p);
    if (p.k == 3 && p.s == 2) return dw_dispatch<T, 3, 2, VEC3, 4>(s, p);
    if (p.k == 5 && p.s == 1) return dw_dispatch<T, 5, 1, VEC5, 8>(s, p);
    if (p.k == 5 && p.s == 2) return dw_dispatch<T, 5, 2, VEC5, 4>(s, p);
    return hipErrorInvalidValue;
}

hipError_t launch_dw(hipStream_t s, int dtype, const DwParams& p) {
    if (p.B <= 0) return hipSuccess;
    if (p.C % 8) return hipErrorInvalidValue;
    // CF_DW_MARCH=1 (experiments build) selects the register-marching kernel (A/B against the LDS-staged one)
    static const bool march = cf_ab_int("CF_DW_MARCH", 0) == 1;
    if (march) {
        if (dtype != 1) return dw_by_shape<float, 4, 4>(s, p);
        return dw_by_shape<bf16_t, 8, 4>(s, p);
    }
    return dtype != 1 ? dw_lds_by_shape<float>(s, p) : dw_lds_by_shape<bf16_t>(s, p);      // dtype 2: fp32 storage, no GEMM here
}

}  // namespace cf
