// Stem: ConvReLU(3, 32, 3, stride=2) = ZeroPad2d(0,1,0,1) -> Conv2d(3,32,3,s2,bias=False) -> Swish
// (model/centernet.py:224 via :58-70), with the reference's host-side preprocessing fused in front
// when the input is a uint8 BGR image: x/255, (x-mean)/std (centerface.py:12-15,32-33).
//
// The 3x3x3 window is a K=27 contraction against 32 output channels: small, but dense, so it goes
// on the matrix core like the 1x1 convs (cf_pw.hip): a wave owns 32 output pixels, lane (pixel, h)
// gathers its half of the 27 taps, D^T = W . X^T, and each lane ends up with 16 contiguous output
// channels of its pixel -> Swish -> two/four 16-byte stores.  bf16 mode: two v_mfma_f32_32x32x16_bf16
// (K padded to 32); fp32 mode: fourteen exact v_mfma_f32_32x32x2_f32.
// The uint8 -> normalised float map is a 3x256 table built once per workgroup in LDS with the
// reference's exact IEEE arithmetic ((u/255 - mean)/std), so the hot loop has no divisions.
// The output (64 B/pixel in bf16) is 5x the input bytes: the kernel is store-bandwidth bound.
#include "cf_common.h"
#include "cf_kernels.h"
#include "centerface_hip.h"

namespace cf {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

static inline int slot_channel0(int i) {
    int h = (i >> 2) & 1;
    int r = (i & 3) + 4 * (i >> 3);
    return h * 16 + r;
}
// tap index of lane-half h, slot s (16 slots per lane); taps >= 27 are zero padding
static inline int tap_of(int dtype, int s, int h) { return dtype != 1 ? 2 * s + h : (s >> 3) * 16 + h * 8 + (s & 7); }

size_t stem_packed_bytes(int dtype) { return (size_t)(dtype != 1 ? 4 : 2) * 64 * 16; }

// w [32][3][3][3] (co, ci, ky, kx)  ->  [chunk][lane][16 B]; tap t = ky*9 + kx*3 + ci
void stem_pack_weights(int dtype, const float* w, void* out_host) {
    const int P = per16(dtype), NCH = dtype != 1 ? 4 : 2;
    __builtin_memset(out_host, 0, stem_packed_bytes(dtype));
    for (int c = 0; c < NCH; ++c)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31, h = lane >> 5, co = slot_channel0(i);
            char* dst = (char*)out_host + ((size_t)c * 64 + lane) * 16;
            float vv[8];
            for (int e = 0; e < P; ++e) {
                const int t = tap_of(dtype, c * P + e, h);
                float v = 0.0f;
                if (t < 27) {
                    const int ky = t / 9, kx = (t % 9) / 3, ci = t % 3;
                    v = w[((co * 3 + ci) * 3 + ky) * 3 + kx];
                }
                vv[e] = v;
            }
            pack_chunk(dtype, vv, dst);
        }
    if (dtype == 2) split_pairs_inplace(out_host, 1, NCH);
}

template <typename T, int FMT>
__global__ __launch_bounds__(256) void stem_kernel(StemParams p) {
    __shared__ float lut[FMT == CF_IN_U8_HWC_BGR ? 768 : 1];
    if constexpr (FMT == CF_IN_U8_HWC_BGR) {
        // centerface.py:12-15 (BGR order), :32-33: exact float32 arithmetic, IEEE divisions
        const float mean[3] = {0.408f, 0.447f, 0.470f};
        const float stdv[3] = {0.289f, 0.274f, 0.278f};
        for (int i = threadIdx.x; i < 768; i += 256) {
            const int ci = i >> 8;
            lut[i] = ((float)(i & 255) / 255.0f - mean[ci]) / stdv[ci];
        }
        __syncthreads();
    }
    constexpr bool F32 = sizeof(T) == 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const long long total = (long long)p.B * Ho * Wo;
    const long long pb = (long long)blockIdx.x * 4 + wave;
    if (pb * 32 >= total) return;
    const long long m = pb * 32 + pl;
    const bool mvalid = m < total;
    const long long mr = mvalid ? m : total - 1;
    const int b = (int)(mr / ((long long)Ho * Wo));
    const int rem = (int)(mr - (long long)b * Ho * Wo);
    const int yo = rem / Wo, xo = rem - yo * Wo;

    float v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int t = F32 ? 2 * s + h : (s >> 3) * 16 + h * 8 + (s & 7);
        const int ky = t / 9, r = t - 9 * ky, kx = r / 3, ci = r - 3 * kx;
        const int iy = 2 * yo + ky, ix = 2 * xo + kx;             // pad_lo = 0, pad_hi = 1
        const bool ok = t < 27 && iy < p.H && ix < p.W;
        float val = 0.0f;
        if (ok) {
            if constexpr (FMT == CF_IN_U8_HWC_BGR) {
                const uint8_t u = ((const uint8_t*)p.x)[(((size_t)b * p.H + iy) * p.W + ix) * 3 + ci];
                val = lut[ci * 256 + u];
            } else {
                val = ((const float*)p.x)[(((size_t)b * 3 + ci) * p.H + iy) * p.W + ix];
            }
        }
        v[s] = val;
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const char* wbase = (const char*)p.w + (size_t)lane * 16;
    if constexpr (F32) {
        mma_chain<T, 4>(acc, [&](int c) { return ld16(wbase + (size_t)c * 1024); }, [&](int c) { return pack16<float>(&v[4 * c]); });
    } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const u32x4 wc = ld16(wbase + (size_t)c * 1024);
            const u32x4 xc = pack16<bf16_t>(&v[8 * c]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, wc),
                                                          __builtin_bit_cast(mfma_bf16x8, xc), acc, 0, 0, 0);
        }
    }
    if (!mvalid) return;
    constexpr int P = Elem<T>::PER16;
    T* out = (T*)p.y + (size_t)m * 32 + h * 16;
#pragma unroll
    for (int g = 0; g < 16 / P; ++g) {
        float o[P];
#pragma unroll
        for (int e = 0; e < P; ++e) o[e] = acc[g * P + e];
        act_arr<1, P>(o);
        st16(out + g * P, pack16<T>(o));
    }
}

hipError_t launch_stem(hipStream_t s, int dtype, const StemParams& p) {
    if (p.B <= 0) return hipSuccess;
    const long long total = (long long)p.B * (p.H / 2) * (p.W / 2);
    dim3 grid((unsigned)((total + 127) / 128)), blk(256);
    set_kernel_tag("void cf::stem_kernel<%s, %d>(cf::StemParams)", dtype == 0 ? "float" : dtype == 2 ? "sp32_t" : "unsigned short", p.in_format);
    if (dtype == 0) {
        if (p.in_format == CF_IN_U8_HWC_BGR) hipLaunchKernelGGL((stem_kernel<float, CF_IN_U8_HWC_BGR>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_kernel<float, CF_IN_F32_NCHW>), grid, blk, 0, s, p);
    } else if (dtype == 2) {
        if (p.in_format == CF_IN_U8_HWC_BGR) hipLaunchKernelGGL((stem_kernel<sp32_t, CF_IN_U8_HWC_BGR>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_kernel<sp32_t, CF_IN_F32_NCHW>), grid, blk, 0, s, p);
    } else {
        if (p.in_format == CF_IN_U8_HWC_BGR) hipLaunchKernelGGL((stem_kernel<bf16_t, CF_IN_U8_HWC_BGR>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((stem_kernel<bf16_t, CF_IN_F32_NCHW>), grid, blk, 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace cf
