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
    if (dtype == 0) g.HC = 48;
    else if (s == 2) g.HC = 48;
    else g.HC = (hid % 96 == 0) ? 96 : 48;
    g.nq = hid / g.HC;
    g.NBE = (g.HC + 31) / 32;
    g.JX = (Cin * sz / 16 + 1) / 2;
    g.HALF = g.HC * sz / 16 / 2;
    g.NBO = (Cout + 31) / 32;
    const int IH = (MB_TOH - 1) * s + k, IW = (MB_TOW - 1) * s + k;
    g.rowb = g.HC * sz + 16;
    g.lds_bytes = (size_t)((IH * IW * g.rowb + 15) / 16 * 16) + (size_t)k * k * g.HC * 4;
    g.wexp_bytes = (size_t)g.nq * g.NBE * g.JX * 64 * 16;
    g.wdw_floats = (size_t)g.nq * k * k * g.HC;
    g.wproj_bytes = (size_t)g.NBO * g.nq * g.HALF * 64 * 16;
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
template <typename T, int KS, int S, int NBO, bool RESID>
__global__ __launch_bounds__(256) void mbconv_kernel(MbParams p) {
    constexpr int P = Elem<T>::PER16;
    constexpr int IH = (MB_TOH - 1) * S + KS, IW = (MB_TOW - 1) * S + KS, IPX = IH * IW;
    constexpr int NIB = (IPX + 31) / 32;
    constexpr int MAXJX = sizeof(T) == 4 ? 12 : 6;               // Cin <= 96
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ROWB = p.rowb;
    char* E = smem;
    float* Wd = reinterpret_cast<float*>(smem + ((IPX * ROWB + 15) / 16 * 16));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int ox0 = blockIdx.x * MB_TOW, oy0 = blockIdx.y * MB_TOH, b = blockIdx.z;
    const int NCx = p.Cin * (int)sizeof(T) / 16;
    const int JX = p.JX, jxmax = h ? NCx - JX : JX;
    const int HC = p.HC, NBE = p.NBE, HALF = p.HALF, nq = p.nq;

    // this lane's output pixel (phase 2/3 and epilogue)
    const int o = wave * 32 + pl;
    const int oy = o / MB_TOW, ox = o % MB_TOW;
    const int ipo = (oy * S) * IW + ox * S;

    f32x16 acc[NBO];
#pragma unroll
    for (int i = 0; i < NBO; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const char* xbase = (const char*)p.x + (size_t)b * p.Hin * p.Win * p.Cin * sizeof(T);

    for (int q = 0; q < nq; ++q) {
        // stage this chunk's depthwise weights [tap][HC] (fp32)
        for (int i = tid; i < KS * KS * HC; i += 256) Wd[i] = p.wdw[(size_t)q * KS * KS * HC + i];

        // ---- phase 1: expand + Swish -> E
        for (int ib = wave; ib < NIB; ib += 4) {
            const int ip = ib * 32 + pl;
            const int ipc = ip < IPX ? ip : IPX - 1;
            const int iy = ipc / IW, ix = ipc - iy * IW;
            const int gy = oy0 * S - p.pad_lo + iy, gx = ox0 * S - p.pad_lo + ix;
            const bool valid = ip < IPX && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
            const char* xrow = xbase + ((size_t)(valid ? gy : 0) * p.Win + (valid ? gx : 0)) * p.Cin * sizeof(T) + (size_t)h * JX * 16;
            u32x4 xf[MAXJX];
#pragma unroll
            for (int j = 0; j < MAXJX; ++j) xf[j] = (valid && j < jxmax) ? ld16(xrow + j * 16) : zero16();
            for (int nbl = 0; nbl < NBE; ++nbl) {
                f32x16 a;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = 0.0f;
                const char* wb = (const char*)p.wexp + ((((size_t)q * NBE + nbl) * JX) * 64 + lane) * 16;
#pragma unroll
                for (int j = 0; j < MAXJX; ++j)
                    if (j < JX) MbMma<T>::run(a, ld16(wb + (size_t)j * 1024), xf[j]);
                if (ip < IPX) {
                    const int ch0 = nbl * 32 + h * 16;
#pragma unroll
                    for (int g = 0; g < 16 / P; ++g) {
                        const int ch = ch0 + g * P;
                        if (ch < HC) {
                            float v[P];
#pragma unroll
                            for (int e = 0; e < P; ++e) v[e] = swish_f(a[g * P + e]);
                            st16(E + (size_t)ip * ROWB + (size_t)ch * sizeof(T), pack16<T>(v));
                        }
                    }
                }
            }
        }
        __syncthreads();

        // ---- phase 2 + 3: depthwise + Swish in registers, straight into the project MFMA
        for (int j = 0; j < HALF; ++j) {
            const int c = h * HALF + j;                               // 16-byte chunk within HC
            float d[P];
#pragma unroll
            for (int e = 0; e < P; ++e) d[e] = 0.0f;
            const char* eb = E + (size_t)ipo * ROWB + (size_t)c * 16;
            const float* wdb = Wd + c * P;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    float ev[P];
                    unpack16<T>(ld16(eb + (size_t)(ky * IW + kx) * ROWB), ev);
                    const float* wt = wdb + (ky * KS + kx) * HC;
#pragma unroll
                    for (int e = 0; e < P; ++e) d[e] = fmaf(ev[e], wt[e], d[e]);
                }
#pragma unroll
            for (int e = 0; e < P; ++e) d[e] = swish_f(d[e]);
            const u32x4 xc = pack16<T>(d);
#pragma unroll
            for (int i = 0; i < NBO; ++i) {
                const char* wb = (const char*)p.wproj + ((((size_t)i * nq + q) * HALF + j) * 64 + lane) * 16;
                MbMma<T>::run(acc[i], ld16(wb), xc);
            }
        }
        __syncthreads();
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

template <typename T, int KS, int S, int NBO, bool RESID>
static hipError_t mb_launch(hipStream_t s, const MbParams& p) {
    auto kfn = mbconv_kernel<T, KS, S, NBO, RESID>;
    static thread_local size_t configured = 0;
    if (p.lds_bytes > 64 * 1024 && configured < p.lds_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return e;
        configured = p.lds_bytes;
    }
    dim3 grid((p.Wout + MB_TOW - 1) / MB_TOW, (p.Hout + MB_TOH - 1) / MB_TOH, p.B), blk(256);
    set_kernel_tag("void cf::mbconv_kernel<%s, %d, %d, %d, %s>(cf::MbParams)", type_tag<T>(), KS, S, NBO, RESID ? "true" : "false");
    hipLaunchKernelGGL(kfn, grid, blk, p.lds_bytes, s, p);
    return hipGetLastError();
}

template <typename T, int KS, int S>
static hipError_t mb_by_out(hipStream_t s, const MbParams& p) {
    const int nbo = (p.Cout + 31) / 32;
    if (p.residual) {
        if (S != 1) return hipErrorInvalidValue;
        switch (nbo) {
            case 1: return mb_launch<T, KS, 1, 1, true>(s, p);
            case 2: return mb_launch<T, KS, 1, 2, true>(s, p);
            case 3: return mb_launch<T, KS, 1, 3, true>(s, p);
        }
    } else {
        switch (nbo) {
            case 1: return mb_launch<T, KS, S, 1, false>(s, p);
            case 2: return mb_launch<T, KS, S, 2, false>(s, p);
            case 3: return mb_launch<T, KS, S, 3, false>(s, p);
        }
    }
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t mb_by_shape(hipStream_t s, const MbParams& p) {
    if (p.k == 3 && p.s == 1) return mb_by_out<T, 3, 1>(s, p);
    if (p.k == 3 && p.s == 2) return mb_by_out<T, 3, 2>(s, p);
    if (p.k == 5 && p.s == 1) return mb_by_out<T, 5, 1>(s, p);
    if (p.k == 5 && p.s == 2) return mb_by_out<T, 5, 2>(s, p);
    return hipErrorInvalidValue;
}

hipError_t launch_mbconv(hipStream_t s, int dtype, const MbParams& p) {
    if (p.B <= 0) return hipSuccess;
    return dtype == 0 ? mb_by_shape<float>(s, p) : mb_by_shape<bf16_t>(s, p);
}

}  // namespace cf
