// conv_last + the first two IDAUp stages as ONE kernel (bf16 storage; round 5: the split-bf16 tolerance mode too): three tiny launches (9 + 12 + 12 us at B = 64, each the
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

constexpr int NK_CH = 4, NK_CW = 8;                 // conv_last cells per workgroup = one 32-pixel MFMA block
constexpr int NK_N = 24;                            // neck channels

// address of 16-byte chunk `chunk` of pixel m: rows [m][K] or pixel-block order [m / 32][chunks][m % 32][16 B]
__device__ __forceinline__ const char* nk_chunk(const void* base, int blocked, size_t m, int NC, int chunk) {
    return blocked ? (const char*)base + blk_off(m, NC, chunk) : (const char*)base + (m * (size_t)NC + (size_t)chunk) * 16;
}

// bias (+ Swish | ReLU) (+ IDAUp up-branch from a low-resolution cell held in LDS) on this lane's 16 accumulators -> 16-byte pieces
// of the storage type (bf16: two pieces of eight channels, fp32 storage: four of four); `bias`, `upw`, `upb` are indexed by channel
// (LDS tables, or this lane's preloaded bias with ch0 = its first channel)
template <typename T, int ACT, bool UP>
__device__ __forceinline__ void nk_epilogue(const f32x16& acc, int h, const float* bias, int ch0, const char* lowcell, const float* upw, const float* upb,
                                            int tap, u32x4* out /*[16 / P]*/) {
    constexpr int P = Elem<T>::PER16, ES = 16 / P;
#pragma unroll
    for (int g = 0; g < 16 / P; ++g) {
        const int ch = h * 16 + g * P;
        if (ch >= NK_N) break;
        float v[P];
#pragma unroll
        for (int e = 0; e < P; ++e) v[e] = acc[g * P + e] + bias[ch - ch0 + e];
        act_arr<ACT, P>(v);
        if constexpr (UP) {
            float r[P];
            unpack16<T>(ld16(lowcell + ch * ES), r);
#pragma unroll
            for (int e = 0; e < P; ++e) v[e] += relu_f(r[e] * upw[tap * NK_N + ch + e] + upb[ch + e]);
        }
        out[g] = pack16<T>(v);
    }
}
// the pieces of a cell: into an LDS tile / a row of the output tensor (`valid` false: zeros, tile only)
template <typename T>
__device__ __forceinline__ void nk_store(char* cell, int h, const u32x4* o, bool valid) {
    constexpr int P = Elem<T>::PER16, ES = 16 / P;
#pragma unroll
    for (int g = 0; g < 16 / P; ++g) {
        const int ch = h * 16 + g * P;
        if (ch >= NK_N) break;
        st16(cell + ch * ES, valid ? o[g] : zero16());
    }
}

// T = bf16_t (round 3) or sp32_t (round 5: the tolerance mode; fp32 tiles, every product through CfMma<sp32_t>'s pair steps in
// pw_kernel<sp32_t>'s order)
template <typename T>
__global__ __launch_bounds__(256) void neck_kernel(NeckParams p) {
    constexpr int P = Elem<T>::PER16, ES = 16 / P, PIT = NK_N * ES;
    constexpr bool WIDE = ES == 4;                         // fp32 storage: twice the fragments per K chain
    __shared__ __attribute__((aligned(16))) char Cs[NK_CH * NK_CW * PIT];            // conv_last tile (4 x 8 cells)
    __shared__ __attribute__((aligned(16))) char U1[4 * NK_CH * NK_CW * PIT];        // up1 tile (8 x 16 cells)
    __shared__ float Tb[2 * (NK_N + 4 * NK_N + NK_N)];                               // b1, upw1, upb1, b2, upw2, upb2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const int cx0 = blockIdx.x * NK_CW, cy0 = blockIdx.y * NK_CH, b = blockIdx.z;
    const int h1 = 2 * p.h, w1 = 2 * p.w, h2 = 4 * p.h, w2 = 4 * p.w;
    constexpr int NC0 = 320 / P, NCh0 = NC0 / 2;           // 16-byte chunks of a pixel's K values, per lane half
    constexpr int NC1 = 96 / P, NCh1 = NC1 / 2;
    constexpr int NC2 = 32 / P, NCh2 = NC2 / 2;
    constexpr int KB = WIDE ? 8 : 5;                       // conv_last weight fragments per batch (even in the split mode: pair steps)
    constexpr int TN = 6 * NK_N;                           // floats per stage in Tb
    static_assert(NCh0 % KB == 0 && (!WIDE || (NCh1 % 2 == 0 && NCh2 % 2 == 0)), "K chains in whole batches / pairs");

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
    auto load_p2 = [&]() {
        const size_t m1 = ((size_t)b * h1 + min(gy1, h1 - 1)) * w1 + min(gx1, w1 - 1);
#pragma unroll
        for (int j = 0; j < NCh1; ++j) {
            x1[j] = ld16(nk_chunk(p.skip1, p.skip1_blk, m1, NC1, h * NCh1 + j));
            wq1[j] = ld16((const char*)p.w1 + ((size_t)j * 64 + lane) * 16);
        }
#pragma unroll
        for (int j = 0; j < NCh2; ++j) wq2[j] = ld16((const char*)p.w2 + ((size_t)j * 64 + lane) * 16);
    };
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
    if (!WIDE || wave != 0) load_p2();                     // (fp32 storage: wave 0's conv_last fragments alone are 160 registers)
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
        u32x4 wq[2][KB];                                   // weight fragments (L2-resident): batch k + 1 in flight under the MFMAs of batch k
#pragma unroll
        for (int u = 0; u < KB; ++u) wq[0][u] = ld16((const char*)p.w0 + ((size_t)u * 64 + lane) * 16);
#pragma unroll
        for (int k = 0; k < NCh0 / KB; ++k) {
            if (k + 1 < NCh0 / KB) {
#pragma unroll
                for (int u = 0; u < KB; ++u) wq[(k + 1) & 1][u] = ld16((const char*)p.w0 + ((size_t)((k + 1) * KB + u) * 64 + lane) * 16);
            }
            mma_chain<T, KB>(acc, [&](int u) -> const u32x4& { return wq[k & 1][u]; }, [&](int u) -> const u32x4& { return x0[k * KB + u]; });
        }
        if (WIDE) load_p2();
        load_x2();
        u32x4 o[16 / P];
        nk_epilogue<T, 1, false>(acc, h, bb, h * 16, nullptr, nullptr, nullptr, 0, o);
        nk_store<T>(Cs + pl * PIT, h, o, ok0);
    }
    __syncthreads();

    // ---- phase 2 (all waves): up1 on the 8x16 cells -> LDS
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        mma_chain<T, NCh1>(acc, [&](int j) -> const u32x4& { return wq1[j]; }, [&](int j) -> const u32x4& { return x1[j]; });
        const int tap = ((uy & 1) << 1) | (ux & 1);         // tile origins are even: local parity = map parity
        u32x4 o[16 / P];
        nk_epilogue<T, 2, true>(acc, h, Tb, 0, Cs + ((uy >> 1) * NK_CW + (ux >> 1)) * PIT, Tb + NK_N, Tb + 5 * NK_N, tap, o);
        nk_store<T>(U1 + o1 * PIT, h, o, ok1);
    }
    __syncthreads();

    // ---- phase 3 (all waves): up2 on the 16x32 cells -> HBM
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        mma_chain<T, NCh2>(acc, [&](int j) -> const u32x4& { return wq2[j]; }, [&](int j) -> const u32x4& { return x2[t][j]; });
        const int oy = wave * 4 + t, ox = pl;
        const int tap = ((oy & 1) << 1) | (ox & 1);
        u32x4 o[16 / P];
        nk_epilogue<T, 2, true>(acc, h, Tb + TN, 0, U1 + ((oy >> 1) * (2 * NK_CW) + (ox >> 1)) * PIT, Tb + TN + NK_N, Tb + TN + 5 * NK_N, tap, o);
        if (ok2[t]) nk_store<T>((char*)p.y + m2[t] * (size_t)PIT, h, o, true);
    }
}

hipError_t launch_neck(hipStream_t s, int dtype, const NeckParams& p) {
    if (p.B <= 0) return hipSuccess;
    dim3 grid((p.w + NK_CW - 1) / NK_CW, (p.h + NK_CH - 1) / NK_CH, p.B), blk(256);
    if (dtype == 1) {
        set_kernel_tag("void cf::neck_kernel<%s>(cf::NeckParams)", type_tag<bf16_t>());
        hipLaunchKernelGGL(neck_kernel<bf16_t>, grid, blk, 0, s, p);
    } else if (dtype == 2) {
        set_kernel_tag("void cf::neck_kernel<%s>(cf::NeckParams)", type_tag<sp32_t>());
        hipLaunchKernelGGL(neck_kernel<sp32_t>, grid, blk, 0, s, p);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace cf
