// conv_last + the first two IDAUp stages as ONE kernel (bf16 storage): three tiny launches (9 + 12 + 12 us at B = 64, each the
// latency of its own K chain at every batch size) become one whose loads are all in flight from the start.
//
// Replaces, fused: conv_1x1_bn(320, 24) + Swish (model/centernet.py:179-184, :236) and IDAUp.forward for up1 and up2
// (:186-204: relu(bn(conv1x1(skip))) + relu(bn_up(deconv2x2(low)))).  The up-branch is a depthwise 2x2 stride-2 deconvolution, so
// every low-resolution cell feeds exactly its own 2x2 block: no halo.  A workgroup owns 4x8 cells of the 1/32 map (conv_last: one
// full MFMA pixel block), hence 8x16 cells of the 1/16 map (up1) and 16x32 of the 1/8 map (up2); the two intermediate maps exist only in LDS, rounded to
// bf16 exactly where the three-kernel path stores them.  Same arithmetic and operation order as cf_pw.hip's pw_kernel (k-steps
// in order, lane half h owns the second half of K; bias, activation, up-branch add), so the result is bit-identical to the
// three launches (tests/test_gpu_parity.py).
#include "cf_common.h"
#include "cf_kernels.h"

namespace cf {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

__device__ __forceinline__ f32x16 nk_mma(f32x16 acc, const u32x4& w, const u32x4& x) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, w), __builtin_bit_cast(mfma_bf16x8, x), acc, 0, 0, 0);
}

constexpr int NK_CH = 4, NK_CW = 8;                 // conv_last cells per workgroup = one 32-pixel MFMA block
constexpr int NK_N = 24, NK_PIT = 48;               // neck channels, bytes per cell in LDS

// address of 16-byte chunk `chunk` of pixel m: rows [m][K] or pixel-block order [m / 32][K / 8][m % 32][8]
__device__ __forceinline__ const char* nk_chunk(const void* base, int blocked, size_t m, int NC, int chunk) {
    return blocked ? (const char*)base + blk_off(m, NC, chunk) : (const char*)base + (m * (size_t)NC + (size_t)chunk) * 16;
}

// bias (+ Swish | ReLU) (+ IDAUp up-branch from a low-resolution cell held in LDS) on this lane's 16 accumulators -> bf16 pieces;
// `bias`, `upw`, `upb` are indexed by channel (LDS tables, or this lane's preloaded bias with ch0 = its first channel)
template <int ACT, bool UP>
__device__ __forceinline__ void nk_epilogue(const f32x16& acc, int h, const float* bias, int ch0, const char* lowcell, const float* upw, const float* upb,
                                            int tap, u32x4* out /*[2]*/) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int ch = h * 16 + g * 8;
        if (ch >= NK_N) break;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[g * 8 + e] + bias[ch - ch0 + e];
        act_arr<ACT, 8>(v);
        if constexpr (UP) {
            float r[8];
            unpack16<bf16_t>(ld16(lowcell + ch * 2), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += relu_f(r[e] * upw[tap * NK_N + ch + e] + upb[ch + e]);
        }
        out[g] = pack16<bf16_t>(v);
    }
}

__global__ __launch_bounds__(256) void neck_kernel(NeckParams p) {
    __shared__ __attribute__((aligned(16))) char Cs[NK_CH * NK_CW * NK_PIT];            // conv_last tile (4 x 8 cells), bf16
    __shared__ __attribute__((aligned(16))) char U1[4 * NK_CH * NK_CW * NK_PIT];        // up1 tile (8 x 16 cells), bf16
    __shared__ float Tb[2 * (NK_N + 4 * NK_N + NK_N)];                                  // b1, upw1, upb1, b2, upw2, upb2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const int cx0 = blockIdx.x * NK_CW, cy0 = blockIdx.y * NK_CH, b = blockIdx.z;
    const int h1 = 2 * p.h, w1 = 2 * p.w, h2 = 4 * p.h, w2 = 4 * p.w;
    constexpr int NC0 = 40, NCh0 = 20;                     // K = 320
    constexpr int NC1 = 12, NCh1 = 6;                      // K = 96
    constexpr int NC2 = 4, NCh2 = 2;                       // K = 32
    constexpr int TN = 6 * NK_N;                           // floats per stage in Tb

    // ---- every global load of the workgroup is issued up front: the three K chains and their epilogue tables cost one memory
    //      round trip instead of eight (the chain of dependent loads was the kernel time at every batch size)
    for (int k = tid; k < 2 * TN; k += 256) {              // epilogue tables of up1 / up2 -> LDS (used after the first barrier)
        const int st = k >= TN, i = k - st * TN;
        const float* src = i < NK_N ? (st ? p.b2 : p.b1) + i : (i < 5 * NK_N ? (st ? p.upw2 : p.upw1) + (i - NK_N) : (st ? p.upb2 : p.upb1) + (i - 5 * NK_N));
        Tb[k] = *src;
    }
    // phase 2: wave w owns pixel block w of the 8x16 up1 tile (two tile rows)
    const int o1 = wave * 32 + pl, uy = o1 >> 4, ux = o1 & 15;
    const int gy1 = 2 * cy0 + uy, gx1 = 2 * cx0 + ux;
    const bool ok1 = gy1 < h1 && gx1 < w1;
    u32x4 x1[NCh1], wq1[NCh1], wq2[NCh2];
    {
        const size_t m1 = ((size_t)b * h1 + min(gy1, h1 - 1)) * w1 + min(gx1, w1 - 1);
#pragma unroll
        for (int j = 0; j < NCh1; ++j) {
            x1[j] = ld16(nk_chunk(p.skip1, p.skip1_blk, m1, NC1, h * NCh1 + j));
            wq1[j] = ld16((const char*)p.w1 + ((size_t)j * 64 + lane) * 16);
        }
#pragma unroll
        for (int j = 0; j < NCh2; ++j) wq2[j] = ld16((const char*)p.w2 + ((size_t)j * 64 + lane) * 16);
    }
    // phase 3: wave w owns pixel blocks 4 w .. 4 w + 3 of the 16x32 up2 tile (block = one tile row of 32 cells); wave 0 issues
    // these loads after its conv_last chain (registers), the other waves -- idle until the first barrier -- now
    u32x4 x2[4][NCh2];
    bool ok2[4]; size_t m2[4];
    auto load_x2 = [&]() {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int gy = 4 * cy0 + wave * 4 + t, gx = 4 * cx0 + pl;
            ok2[t] = gy < h2 && gx < w2;
            m2[t] = ((size_t)b * h2 + min(gy, h2 - 1)) * w2 + min(gx, w2 - 1);
#pragma unroll
            for (int j = 0; j < NCh2; ++j) x2[t][j] = ld16(nk_chunk(p.skip2, p.skip2_blk, m2[t], NC2, h * NCh2 + j));
        }
    };
    if (wave != 0) load_x2();

    // ---- phase 1 (wave 0): conv_last on the 4x8 cells, bias + Swish -> LDS
    if (wave == 0) {
        const int cy = pl >> 3, cx = pl & 7;
        const bool ok0 = cy0 + cy < p.h && cx0 + cx < p.w;
        const size_t m0 = ((size_t)b * p.h + min(cy0 + cy, p.h - 1)) * p.w + min(cx0 + cx, p.w - 1);
        u32x4 x0[NCh0];
#pragma unroll
        for (int j = 0; j < NCh0; ++j) x0[j] = ld16(nk_chunk(p.x, p.x_blk, m0, NC0, h * NCh0 + j));
        float bb[16];                                      // this lane's conv_last shifts: channels 16 h .. (h = 1: 16..23 only)
#pragma unroll
        for (int i = 0; i < 16; ++i) bb[i] = (h * 16 + i < NK_N) ? p.b0[h * 16 + i] : 0.0f;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        u32x4 wq[2][5];                                    // weight fragments (L2-resident): batch k + 1 in flight under the MFMAs of batch k
#pragma unroll
        for (int u = 0; u < 5; ++u) wq[0][u] = ld16((const char*)p.w0 + ((size_t)u * 64 + lane) * 16);
#pragma unroll
        for (int k = 0; k < NCh0 / 5; ++k) {
            if (k + 1 < NCh0 / 5) {
#pragma unroll
                for (int u = 0; u < 5; ++u) wq[(k + 1) & 1][u] = ld16((const char*)p.w0 + ((size_t)((k + 1) * 5 + u) * 64 + lane) * 16);
            }
#pragma unroll
            for (int u = 0; u < 5; ++u) acc = nk_mma(acc, wq[k & 1][u], x0[k * 5 + u]);
        }
        load_x2();
        u32x4 o[2];
        nk_epilogue<1, false>(acc, h, bb, h * 16, nullptr, nullptr, nullptr, 0, o);
        if (!ok0) { o[0] = zero16(); o[1] = zero16(); }
        st16(Cs + pl * NK_PIT + h * 32, o[0]);
        if (h == 0) st16(Cs + pl * NK_PIT + 16, o[1]);
    }
    __syncthreads();

    // ---- phase 2 (all waves): up1 on the 8x16 cells -> LDS
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int j = 0; j < NCh1; ++j) acc = nk_mma(acc, wq1[j], x1[j]);
        const int tap = ((uy & 1) << 1) | (ux & 1);         // tile origins are even: local parity = map parity
        u32x4 o[2];
        nk_epilogue<2, true>(acc, h, Tb, 0, Cs + ((uy >> 1) * NK_CW + (ux >> 1)) * NK_PIT, Tb + NK_N, Tb + 5 * NK_N, tap, o);
        if (!ok1) { o[0] = zero16(); o[1] = zero16(); }
        st16(U1 + o1 * NK_PIT + h * 32, o[0]);
        if (h == 0) st16(U1 + o1 * NK_PIT + 16, o[1]);
    }
    __syncthreads();

    // ---- phase 3 (all waves): up2 on the 16x32 cells -> HBM
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int j = 0; j < NCh2; ++j) acc = nk_mma(acc, wq2[j], x2[t][j]);
        const int oy = wave * 4 + t, ox = pl;
        const int tap = ((oy & 1) << 1) | (ox & 1);
        u32x4 o[2];
        nk_epilogue<2, true>(acc, h, Tb + TN, 0, U1 + ((oy >> 1) * (2 * NK_CW) + (ox >> 1)) * NK_PIT, Tb + TN + NK_N, Tb + TN + 5 * NK_N, tap, o);
        if (ok2[t]) {
            char* dst = (char*)p.y + m2[t] * (size_t)(NK_N * 2);
            st16(dst + h * 32, o[0]);
            if (h == 0) st16(dst + 16, o[1]);
        }
    }
}

hipError_t launch_neck(hipStream_t s, const NeckParams& p) {
    if (p.B <= 0) return hipSuccess;
    dim3 grid((p.w + NK_CW - 1) / NK_CW, (p.h + NK_CH - 1) / NK_CH, p.B), blk(256);
    set_kernel_tag("cf::neck_kernel(cf::NeckParams)");
    hipLaunchKernelGGL(neck_kernel, grid, blk, 0, s, p);
    return hipGetLastError();
}

}  // namespace cf
