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
#include "cf_common.h"
#include "cf_kernels.h"

namespace cf {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

static inline int slot_channel(int nb, int i) {
    int h = (i >> 2) & 1;
    int r = (i & 3) + 4 * (i >> 3);
    return nb * 32 + h * 16 + r;
}

constexpr int MB_TOH = 8, MB_TOW = 16;
bool mb_supported(int dtype, int k, int s, int jx, int hc, int nbo, int res);

template <typename T> struct MbMma;
template <> struct MbMma<bf16_t> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, w),
                                                      __builtin_bit_cast(mfma_bf16x8, x), acc, 0, 0, 0);
    }
};
template <> struct MbMma<float> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
    }
};

// ---------------------------------------------------------------- host: geometry + packing
MbGeom mb_geometry(int dtype, int Cin, int hid, int Cout, int k, int s) {
    MbGeom g{};
    const int sz = (int)elem_size(dtype);
    g.ok = (Cin % 8 == 0) && (hid % 48 == 0) && (Cout % 8 == 0) && Cout <= 96 && Cin <= 96 && (k == 3 || k == 5) && (s == 1 || s == 2);
    if (!g.ok) return g;
    // hidden chunk: whole 32-channel MFMA blocks where hid allows; smaller chunks for the stride-2
    // tiles (4.6x more input pixels per output pixel in LDS) and for fp32 storage
    if (dtype == 0) g.HC = (hid % 32 == 0) ? 32 : 48;
    else if (s == 2) g.HC = (hid % 32 == 0) ? 32 : 48;
    else g.HC = (hid % 96 == 0) ? 96 : 48;
    g.nq = hid / g.HC;
    g.NBE = (g.HC + 31) / 32;
    g.JX = (Cin * sz / 16 + 1) / 2;
    g.HALF = g.HC * sz / 16 / 2;
    g.NBO = (Cout + 31) / 32;
    const int IH = (MB_TOH - 1) * s + k, IW = (MB_TOW - 1) * s + k;
    g.rowb = g.HC * sz + 16;
    g.lds_bytes = (size_t)((IH * IW * g.rowb + 15) / 16 * 16) + 2 * ((size_t)g.NBE * g.JX * 1024 + (size_t)k * k * g.HC * 4);
    g.wexp_bytes = (size_t)g.nq * g.NBE * g.JX * 64 * 16;
    g.wdw_floats = (size_t)g.nq * k * k * g.HC;
    g.wproj_bytes = (size_t)g.NBO * g.nq * g.HALF * 64 * 16;
    g.ok = mb_supported(dtype, k, s, g.JX, g.HC, g.NBO, (Cin == Cout && s == 1) ? 1 : 0);
    return g;
}

// we [hid][Cin], wd [hid][k*k], wp [Cout][hid]
void mb_pack_weights(int dtype, const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we,
                     const float* wd, const float* wp, void* wexp_host, float* wdw_host, void* wproj_host) {
    const int P = per16(dtype);
    const int NCx = Cin * (int)elem_size(dtype) / 16;
    __builtin_memset(wexp_host, 0, g.wexp_bytes);
    __builtin_memset(wproj_host, 0, g.wproj_bytes);
    auto put = [&](char* dst, const float* src, int n) {
        if (dtype == 0) for (int e = 0; e < n; ++e) ((float*)dst)[e] = src[e];
        else for (int e = 0; e < n; ++e) ((uint16_t*)dst)[e] = host_f32_to_bf16(src[e]);
    };
    for (int q = 0; q < g.nq; ++q) {
        for (int nbl = 0; nbl < g.NBE; ++nbl)
            for (int j = 0; j < g.JX; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    const int cl = slot_channel(nbl, i);           // channel within the hidden chunk
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
}

// ---------------------------------------------------------------- device
// NW waves per workgroup (4 or 8).  With 8 waves the tile still has 4 pixel blocks: waves w and w+4
// share pixel block w&3 and split the k-steps of every hidden chunk between them; their partial
// project sums are combined once, through LDS, in the epilogue.  Twice the waves per LDS byte.
// The block geometry (JX = 16-byte chunks of a Cin row per lane half, HC = hidden chunk) is a
// template parameter: the expand / depthwise / project loops are straight-line code that the
// compiler can interleave (MFMA next to VALU next to LDS) -- with run-time trip counts every k-step
// became its own basic block and nothing overlapped.
template <typename T, int KS, int S, int NBO, bool RESID, int NW, int JX, int HC>
__global__ __launch_bounds__(NW * 64) void mbconv_kernel(MbParams p) {
    constexpr int P = Elem<T>::PER16;
    constexpr int IH = (MB_TOH - 1) * S + KS, IW = (MB_TOW - 1) * S + KS, IPX = IH * IW;
    constexpr int NIB = (IPX + 31) / 32;
    constexpr int MAXJX = JX;
    constexpr int NT = NW * 64;
    constexpr int NBE = (HC + 31) / 32, HALF = HC * (int)sizeof(T) / 16 / 2;
    constexpr int ROWB = HC * (int)sizeof(T) + 16;
    constexpr int WXB = NBE * JX * 1024;                         // expand fragments per chunk (bytes)
    constexpr int WSTAGE = WXB + KS * KS * HC * 4;               // + depthwise taps (fp32)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* E = smem;
    char* Wst = smem + ((IPX * ROWB + 15) / 16 * 16);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int ox0 = blockIdx.x * MB_TOW, oy0 = blockIdx.y * MB_TOH, b = blockIdx.z;
    const int nq = p.nq;

    // this lane's output pixel (phase 2/3 and epilogue) and its share of the k-steps
    const int pbk = wave & 3, jg = wave >> 2;
    const int o = pbk * 32 + pl;
    const int oy = o / MB_TOW, ox = o % MB_TOW;
    const unsigned e_pix = (unsigned)((oy * S) * IW + ox * S) * (unsigned)ROWB;
    constexpr int JSPLIT = NW == 8 ? (HALF + 1) / 2 : HALF;

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
        constexpr int WDB = KS * KS * HC * 4;                         // multiple of 16, not of 1024
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
    }
    auto expand_block = [&](int ib, const u32x4* xf, const char* wx) {
        const int ip = ib * 32 + pl;
        const bool ipok = ip < IPX;
        char* erow = E + (unsigned)(ipok ? ip : 0) * (unsigned)ROWB;
#pragma unroll
        for (int nbl = 0; nbl < NBE; ++nbl) {
            {
                f32x16 a;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = 0.0f;
                const char* wb = wx + (nbl * JX * 64 + lane) * 16;
#pragma unroll
                for (int j = 0; j < JX; ++j) MbMma<T>::run(a, ld16(wb + j * 1024), xf[j]);
                const int ch0 = nbl * 32 + h * 16;
#pragma unroll
                for (int g = 0; g < 16 / P; ++g) {
                    float v[P];
#pragma unroll
                    for (int e = 0; e < P; ++e) v[e] = swish_f(a[g * P + e]);
                    const u32x4 pk = pack16<T>(v);
                    const int ch = ch0 + g * P;
                    if (ch < HC) { if (ipok) st16(erow + ch * (int)sizeof(T), pk); }
                }
            }
        }
    };

    stage_weights(0);
    for (int q = 0; q < nq; ++q) {
        const char* wx = Wst + (q & 1) * WSTAGE;
        const char* wdq = wx + WXB;
        __syncthreads();      // previous chunk's phase 2 done with E; this stage's weights landed

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
            // NW == 8: wave group jg owns k-steps [0, JSPLIT) or [JSPLIT, HALF) -- wave-uniform
            if (NW == 8 && ((j < JSPLIT) != (jg == 0))) continue;
            u32x4 wpc[NBO];
#pragma unroll
            for (int i = 0; i < NBO; ++i)
                wpc[i] = ld16((const char*)p.wproj + ((((size_t)i * nq + q) * HALF + j) * 64 + lane) * 16);
            const int c = h * HALF + j;                               // 16-byte chunk within HC
            float d[P];
            const char* eb = E + e_pix + c * 16;
            const char* wdb = wdq + c * (P * 4);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    float ev[P];
                    unpack16<T>(ld16(eb + (ky * IW + kx) * ROWB), ev);
                    const char* wt = wdb + (ky * KS + kx) * HC * 4;
                    float wv[P];
                    unpack16<float>(ld16(wt), wv);
                    if constexpr (P == 8) unpack16<float>(ld16(wt + 16), wv + 4);
                    if (ky == 0 && kx == 0) {
#pragma unroll
                        for (int e = 0; e < P; ++e) d[e] = ev[e] * wv[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < P; ++e) d[e] = fmaf(ev[e], wv[e], d[e]);
                    }
                }
#pragma unroll
            for (int e = 0; e < P; ++e) d[e] = swish_f(d[e]);
            const u32x4 xc = pack16<T>(d);
#pragma unroll
            for (int i = 0; i < NBO; ++i) MbMma<T>::run(acc[i], wpc[i], xc);
        }
    }

    // ---- combine the two k-halves (NW == 8): waves 4..7 hand their partial sums over through LDS
    if constexpr (NW == 8) {
        __syncthreads();                                  // everyone is done with E
        float* red = reinterpret_cast<float*>(smem) + (size_t)(pbk * 64 + lane) * (NBO * 16);
        if (jg == 1) {
#pragma unroll
            for (int i = 0; i < NBO; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = acc[i][g * 4 + e];
                    st16(red + i * 16 + g * 4, pack16<float>(t));
                }
        }
        __syncthreads();
        if (jg == 1) return;
#pragma unroll
        for (int i = 0; i < NBO; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float t[4];
                unpack16<float>(ld16(red + i * 16 + g * 4), t);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][g * 4 + e] += t[e];
            }
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
            st16((char*)p.y + (opix * p.Cout + ch) * sizeof(T), pack16<T>(v));
        }
    }
}

template <typename T, int KS, int S, int NBO, bool RESID, int NW, int JX, int HC>
static hipError_t mb_launch_nw(hipStream_t s, const MbParams& p) {
    auto kfn = mbconv_kernel<T, KS, S, NBO, RESID, NW, JX, HC>;
    static thread_local size_t configured = 0;
    if (p.lds_bytes > 64 * 1024 && configured < p.lds_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return e;
        configured = p.lds_bytes;
    }
    dim3 grid((p.Wout + MB_TOW - 1) / MB_TOW, (p.Hout + MB_TOH - 1) / MB_TOH, p.B), blk(NW * 64);
    set_kernel_tag("void cf::mbconv_kernel<%s, %d, %d, %d, %s, %d, %d, %d>(cf::MbParams)", type_tag<T>(), KS, S, NBO,
                   RESID ? "true" : "false", NW, JX, HC);
    hipLaunchKernelGGL(kfn, grid, blk, p.lds_bytes, s, p);
    return hipGetLastError();
}

template <typename T, int KS, int S, int NBO, bool RESID, int JX, int HC>
static hipError_t mb_launch(hipStream_t s, const MbParams& p) {
    // 8 waves need >= 2 k-steps per hidden chunk and room in the E region for the 4 partial-sum slabs
    constexpr int IPX = ((MB_TOH - 1) * S + KS) * ((MB_TOW - 1) * S + KS);
    constexpr int HALF = HC * (int)sizeof(T) / 16 / 2;
    constexpr bool can8 = HALF >= 2 && (size_t)4 * 64 * NBO * 16 * 4 <= (size_t)IPX * (HC * sizeof(T) + 16);
    const bool want8 = p.nw == 8 || (p.nw == 0 && KS == 5);
    if constexpr (can8) { if (want8) return mb_launch_nw<T, KS, S, NBO, RESID, 8, JX, HC>(s, p); }
    return mb_launch_nw<T, KS, S, NBO, RESID, 4, JX, HC>(s, p);
}

// The block shapes of the CenterFace backbone (model/centernet.py:211-219), layer1.0 .. layer4.1:
//   X(dtype-independent: KS, S, Cin, hid-chunk rule, NBO, residual)
struct MbEntry { int dtype, k, s, jx, hc, nbo, res; hipError_t (*fn)(hipStream_t, const MbParams&); };
#define MB_ENTRY(T, DT, KS, S, JX, HC, NBO, RES) {DT, KS, S, JX, HC, NBO, RES, &mb_launch<T, KS, S, NBO, (RES != 0), JX, HC>}
static const MbEntry kMbTable[] = {
    // bf16 storage                           layer
    MB_ENTRY(bf16_t, 1, 3, 2, 1, 32, 1, 0),   // 1.0  16 ->  96 -> 24
    MB_ENTRY(bf16_t, 1, 3, 1, 2, 48, 1, 1),   // 1.1  24 -> 144 -> 24 (+res)
    MB_ENTRY(bf16_t, 1, 5, 2, 2, 48, 1, 0),   // 2.0  24 -> 144 -> 32
    MB_ENTRY(bf16_t, 1, 5, 1, 2, 96, 1, 1),   // 2.1  32 -> 192 -> 32 (+res)
    MB_ENTRY(bf16_t, 1, 3, 2, 2, 32, 2, 0),   // 3.0  32 -> 192 -> 64
    MB_ENTRY(bf16_t, 1, 3, 1, 4, 96, 2, 1),   // 3.1  64 -> 384 -> 64 (+res)
    MB_ENTRY(bf16_t, 1, 5, 1, 4, 96, 3, 0),   // 4.0  64 -> 384 -> 96
    MB_ENTRY(bf16_t, 1, 5, 1, 6, 96, 3, 1),   // 4.1  96 -> 576 -> 96 (+res)
    // fp32 storage (parity mode)
    MB_ENTRY(float, 0, 3, 2, 2, 32, 1, 0),
    MB_ENTRY(float, 0, 3, 1, 3, 48, 1, 1),
    MB_ENTRY(float, 0, 5, 2, 3, 48, 1, 0),
    MB_ENTRY(float, 0, 5, 1, 4, 32, 1, 1),
    MB_ENTRY(float, 0, 3, 2, 4, 32, 2, 0),
    MB_ENTRY(float, 0, 3, 1, 8, 32, 2, 1),
    MB_ENTRY(float, 0, 5, 1, 8, 32, 3, 0),
    MB_ENTRY(float, 0, 5, 1, 12, 32, 3, 1),
};
#undef MB_ENTRY

static const MbEntry* mb_find(int dtype, int k, int s, int jx, int hc, int nbo, int res) {
    for (const MbEntry& e : kMbTable)
        if (e.dtype == dtype && e.k == k && e.s == s && e.jx == jx && e.hc == hc && e.nbo == nbo && e.res == res) return &e;
    return nullptr;
}

bool mb_supported(int dtype, int k, int s, int jx, int hc, int nbo, int res) { return mb_find(dtype, k, s, jx, hc, nbo, res) != nullptr; }

hipError_t launch_mbconv(hipStream_t s, int dtype, const MbParams& p) {
    if (p.B <= 0) return hipSuccess;
    const MbEntry* e = mb_find(dtype, p.k, p.s, p.JX, p.HC, (p.Cout + 31) / 32, p.residual ? 1 : 0);
    if (!e) return hipErrorInvalidValue;
    return e->fn(s, p);
}

}  // namespace cf
