// The four CenterFace heads as ONE kernel: 3x3 conv (padding 1, bias) on the matrix core as an
// implicit GEMM, then the per-head 1x1 conv (bias), then sigmoid+clamp on the heat-map channel.
//
// Replaces, per head in {hm, wh, lm, reg}: nn.Conv2d(24,24,3,padding=1,bias=True) ->
// nn.Conv2d(24,c,1,bias=True) (model/centernet.py:247-261, applied :277-279) and
// torch.clamp(out['hm'].sigmoid_(), 1e-4, 1-1e-4) (centerface.py:43).  The reference reads the
// 24x160x160 neck tensor four times and writes four 24-channel intermediates; here it is read once
// and the intermediates never leave registers.
//
// Implicit GEMM mapping (same free-permutation idea as cf_pw.hip): for an output pixel, the 3 input
// pixels (x-1,x,x+1) of one kernel row dy are 72 CONTIGUOUS NHWC elements; lane half h owns the
// contiguous half of those 72 (bf16: 16-byte chunks 0-4 | 5-8; fp32: 0-8 | 9-17), a chunk never
// straddles a pixel (24 ch = 3 or 6 chunks), so border zero-padding is a per-chunk predicate.
// Two-stage mode keeps the reference's operation order (3x3 24->4x24, +b, 1x1 block-diagonal, +b);
// collapsed mode (CF_FLAG_COLLAPSE_HEADS) folds the linear pair into one 3x3 24->15 conv.
#include "cf_common.h"
#include "cf_kernels.h"

namespace cf {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

static inline int slot_channel(int nb, int i) {
    int h = (i >> 2) & 1;
    int r = (i & 3) + 4 * (i >> 3);
    return nb * 32 + h * 16 + r;
}
// chunks per kernel row (3 pixels x 24 ch) and MFMA steps per kernel row
static inline int cpd(int dtype) { return dtype != 1 ? 18 : 9; }
static inline int spd(int dtype) { return dtype != 1 ? 9 : 5; }

size_t head_packed_bytes(int dtype, int collapsed) {
    int NB = collapsed ? 1 : 3;
    return (size_t)NB * 3 * spd(dtype) * 64 * 16;
}

// output slot order of the 16-float head record: hm, wh0, wh1, lm0..9, reg0, reg1, (hm_raw)
void head_pack_weights(int dtype, int collapsed, const float* w0, const float* b0, const float* w1,
                       const float* b1, void* w0p_host, float* b0_host, float* w1d_host, float* b1_host) {
    const int P = per16(dtype), CPD = cpd(dtype), SPD = spd(dtype);
    const int NB = collapsed ? 1 : 3;
    const int NQ = collapsed ? 15 : 96;
    static const int head_of_out[15] = {0, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3};
    // effective 3x3 weight table W[q][ci][ky][kx]
    double* W = new double[(size_t)NQ * 24 * 9];
    double bq[96];
    if (!collapsed) {
        for (int q = 0; q < 96; ++q) {
            for (int t = 0; t < 24 * 9; ++t) W[(size_t)q * 216 + t] = w0[(size_t)q * 216 + t];
            bq[q] = b0[q];
        }
    } else {
        for (int o = 0; o < 15; ++o) {
            int hd = head_of_out[o];
            for (int t = 0; t < 216; ++t) {
                double s = 0;
                for (int c = 0; c < 24; ++c) s += (double)w1[o * 24 + c] * (double)w0[((size_t)hd * 24 + c) * 216 + t];
                W[(size_t)o * 216 + t] = s;
            }
            double s = b1[o];
            for (int c = 0; c < 24; ++c) s += (double)w1[o * 24 + c] * (double)b0[hd * 24 + c];
            bq[o] = s;
        }
    }
    __builtin_memset(w0p_host, 0, head_packed_bytes(dtype, collapsed));
    for (int nb = 0; nb < NB; ++nb)
        for (int dy = 0; dy < 3; ++dy)
            for (int j = 0; j < SPD; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    int i = lane & 31, h = lane >> 5;
                    int q = slot_channel(nb, i);
                    // collapsed == 2 (cf_uphead.hip): MFMA row i = record slot i, so that BOTH lane halves of the
                    // result hold eight slots of the pixel (rows 0-3, 8-11 | 4-7, 12-15) and store 32 bytes each;
                    // slot 15 (hm_raw) repeats slot 0's weights -- the same dot product, bit for bit
                    if (collapsed == 2) q = i < 15 ? i : i == 15 ? 0 : NQ;
                    int c = h * SPD + j;
                    if (q >= NQ || c >= CPD) continue;
                    char* dst = (char*)w0p_host + ((((size_t)nb * 3 + dy) * SPD + j) * 64 + lane) * 16;
                    float vv[8];
                    for (int e = 0; e < P; ++e) {
                        int idx = c * P + e;          // element within the 72-wide kernel row
                        int dx = idx / 24, ci = idx % 24;
                        vv[e] = (float)W[(size_t)q * 216 + (ci * 3 + dy) * 3 + dx];
                    }
                    pack_chunk(dtype, vv, dst);
                }
    if (dtype == 2) split_pairs_inplace(w0p_host, (size_t)NB * 3, SPD);        // the SPD chunk steps of a kernel row in pairs
    for (int q = 0; q < (collapsed ? 16 : 96); ++q) b0_host[q] = q < NQ ? (float)bq[q] : 0.0f;
    if (collapsed == 2) b0_host[15] = (float)bq[0];
    for (int i = 0; i < 96 * 16; ++i) w1d_host[i] = 0.0f;
    for (int o = 0; o < 16; ++o) b1_host[o] = (o < 15 && !collapsed) ? b1[o] : 0.0f;
    if (!collapsed)
        for (int o = 0; o < 15; ++o)
            for (int c = 0; c < 24; ++c) w1d_host[(head_of_out[o] * 24 + c) * 16 + o] = w1[o * 24 + c];
    delete[] W;
}

template <typename T> using HMma = CfMma<T>;

template <typename T, bool COLLAPSED>
__global__ __launch_bounds__(256) void head_kernel(HeadParams p) {
    constexpr int P = Elem<T>::PER16;
    constexpr int CPP = 24 / P;            // chunks per pixel
    constexpr int CPD = 3 * CPP;           // chunks per kernel row
    constexpr int SPD = (CPD + 1) / 2;     // MFMA steps per kernel row
    constexpr int NB = COLLAPSED ? 1 : 3;

    __shared__ __attribute__((aligned(16))) float w1s[COLLAPSED ? 16 : 96 * 16];
    if constexpr (!COLLAPSED) {
        for (int i = threadIdx.x; i < 96 * 16; i += 256) w1s[i] = p.w1d[i];
        __syncthreads();
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const long long M = (long long)p.B * p.h * p.w;
    const long long pb = (long long)blockIdx.x * 4 + wave;
    if (pb * 32 >= M) return;
    const long long m = pb * 32 + pl;
    const bool mvalid = m < M;
    const long long mr = mvalid ? m : M - 1;
    const int b = (int)(mr / ((long long)p.h * p.w));
    const int rem = (int)(mr - (long long)b * p.h * p.w);
    const int y = rem / p.w, x = rem - y * p.w;

    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const char* wbase = (const char*)p.w0p + (size_t)lane * 16;
    // Branch-free operand loads: every chunk is read from a clamped (valid) address and zeroed afterwards when it lies outside
    // the map (= the conv's zero padding) -- a predicated load ends in a full vmcnt(0) wait each, and the 27 chunks of a pixel
    // would be 27 serial memory round trips.  All chunks of kernel row dy + 1 are in flight while row dy's MFMAs run.
    const char* img = (const char*)p.x + (size_t)b * p.h * p.w * 24 * sizeof(T);
    auto load_row = [&](int dy, u32x4* xc) {
        const int iy = y + dy - 1;
        const int cy = min(max(iy, 0), p.h - 1);
#pragma unroll
        for (int j = 0; j < SPD; ++j) {
            const int c = min(h * SPD + j, CPD - 1);
            const int cx = min(max(x - 1 + c / CPP, 0), p.w - 1);
            xc[j] = ld16(img + ((size_t)cy * p.w + cx) * 24 * sizeof(T) + (size_t)(c % CPP) * 16);
        }
    };
    u32x4 xn[SPD];
    load_row(0, xn);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = y + dy - 1;
        const bool yok = (unsigned)iy < (unsigned)p.h;
        u32x4 xc[SPD];
#pragma unroll
        for (int j = 0; j < SPD; ++j) {
            const int c = h * SPD + j;
            const int ix = x - 1 + c / CPP;
            const bool ok = yok && c < CPD && (unsigned)ix < (unsigned)p.w;
            xc[j].x = ok ? xn[j].x : 0u; xc[j].y = ok ? xn[j].y : 0u; xc[j].z = ok ? xn[j].z : 0u; xc[j].w = ok ? xn[j].w : 0u;
        }
        if (dy + 1 < 3) load_row(dy + 1, xn);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            mma_chain<T, SPD>(acc[i], [&](int j) { return ld16(wbase + (((size_t)i * 3 + dy) * SPD + j) * 1024); }, [&](int j) { return xc[j]; });
    }

    // ---- second stage.  Lane (pixel, h) holds first-stage channels nb*32 + h*16 + r.
    float out[16];
    if constexpr (COLLAPSED) {
        // slots 0..15 of n-block 0: lane half h holds slots h*16 + r; only h == 0 is meaningful
#pragma unroll
        for (int r = 0; r < 16; ++r) out[r] = acc[0][r] + p.b0[r];
    } else {
#pragma unroll
        for (int o = 0; o < 16; ++o) out[o] = 0.0f;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = i * 32 + h * 16 + r;
                const float t = acc[i][r] + p.b0[q];
                const f32x4* wr = reinterpret_cast<const f32x4*>(&w1s[q * 16]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 wv = wr[g];
                    out[g * 4 + 0] = fmaf(t, wv.x, out[g * 4 + 0]);
                    out[g * 4 + 1] = fmaf(t, wv.y, out[g * 4 + 1]);
                    out[g * 4 + 2] = fmaf(t, wv.z, out[g * 4 + 2]);
                    out[g * 4 + 3] = fmaf(t, wv.w, out[g * 4 + 3]);
                }
            }
        // combine the two halves of the pixel (lanes l and l^32) and add the 1x1 bias
#pragma unroll
        for (int o = 0; o < 16; ++o) out[o] = out[o] + __shfl_xor(out[o], 32) + p.b1[o];
    }
    if (!mvalid || h != 0) return;
    const float raw = out[0];
    // centerface.py:43: clamp(sigmoid(hm), 1e-4, 1 - 1e-4); precise exp + IEEE divide
    float sg = 1.0f / (1.0f + expf(-raw));
    sg = fminf(fmaxf(sg, 1e-4f), 1.0f - 1e-4f);
    out[0] = sg;
    out[15] = raw;
    if (p.hm_plane) p.hm_plane[m] = sg;
    float* dst = p.heads + (size_t)m * 16;
#pragma unroll
    for (int g = 0; g < 4; ++g) st16(dst + g * 4, pack16<float>(&out[g * 4]));
}

hipError_t launch_heads(hipStream_t s, int dtype, const HeadParams& p) {
    const long long M = (long long)p.B * p.h * p.w;
    if (M <= 0) return hipSuccess;
    dim3 grid((unsigned)((M + 127) / 128)), blk(256);
    set_kernel_tag("void cf::head_kernel<%s, %s>(cf::HeadParams)", dtype == 0 ? "float" : dtype == 2 ? "sp32_t" : "unsigned short", p.collapsed ? "true" : "false");
    if (dtype == 0) {
        if (p.collapsed) hipLaunchKernelGGL((head_kernel<float, true>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((head_kernel<float, false>), grid, blk, 0, s, p);
    } else if (dtype == 2) {
        if (p.collapsed) hipLaunchKernelGGL((head_kernel<sp32_t, true>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((head_kernel<sp32_t, false>), grid, blk, 0, s, p);
    } else {
        if (p.collapsed) hipLaunchKernelGGL((head_kernel<bf16_t, true>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((head_kernel<bf16_t, false>), grid, blk, 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace cf
