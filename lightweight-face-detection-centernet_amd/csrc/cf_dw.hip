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
                for (int e = 0; e < VEC; ++e) o[e] = act_f<ACT>(acc[t][e] + breg[e]);
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
    if (dtype == 0) return dw_by_shape<float, 4, 4>(s, p);
    return dw_by_shape<bf16_t, 8, 4>(s, p);
}

}  // namespace cf
