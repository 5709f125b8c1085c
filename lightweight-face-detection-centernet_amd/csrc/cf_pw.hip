// Pointwise (1x1) convolution on MFMA for gfx950: y[M][N] = epi(x[M][K] . w[N][K]^T).
//
// Replaces every nn.Conv2d(kernel_size=1) instance on the hot path: MBConv expand (+Swish,
// model/centernet.py:109-110), MBConv project (+residual, :117-118,:134-137), conv_1x1_bn (:179-184),
// IDAUp.conv + the whole IDAUp up-branch as an epilogue (:186-204), ShuffleV2 pw convs
// (model/blocks.py:22-24,31-33).
//
// Design (MI355X-first, not a GEMM library call):
//  * The op is HBM-bound (AI ~ 29 flop/B at bf16 vs machine balance ~312), so the kernel is a
//    streaming kernel that happens to use the matrix core.  No LDS: the activation operand is read
//    exactly once, straight into MFMA operand registers; weights (<= 307 KB) are pre-packed on the
//    host in fragment order so that every wave-load is one contiguous 1 KiB line burst from L2.
//  * Computes D^T = W . X^T with v_mfma_f32_32x32x16_bf16 (bf16 storage) or the exact-fp32
//    v_mfma_f32_32x32x2_f32 (fp32 parity mode): the pixel index is the MFMA column, so each lane ends
//    up owning one pixel, and -- because the output-channel <-> MFMA-row assignment is free (weights
//    are repacked) -- 16 CONTIGUOUS output channels of it: epilogue and stores are 16-byte vectors.
//  * The K <-> MFMA k-slot assignment is free as well (sum over k): lane half h owns the contiguous
//    half [h*K/2, (h+1)*K/2) of its pixel's row, so successive 16-byte loads of a lane walk one
//    cache line instead of striding across rows.
#include "cf_common.h"
#include "cf_kernels.h"
#include <cstdlib>
#include <type_traits>

namespace cf {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

// channel held by MFMA row-slot i of n-block nb (see header comment): lane (pixel, h) accumulator
// register r  <->  row i = (r&3) + 8*(r>>2) + 4*h  <->  channel nb*32 + h*16 + r
static inline int slot_channel(int nb, int i) {
    int h = (i >> 2) & 1;
    int r = (i & 3) + 4 * (i >> 3);
    return nb * 32 + h * 16 + r;
}

static inline int pw_chunks(int dtype, int K) { return (int)((size_t)K * elem_size(dtype) / 16); }

size_t pw_packed_bytes(int dtype, int K, int N) {
    int NC = pw_chunks(dtype, K), NCh = (NC + 1) / 2;
    int NB = (N + 31) / 32;
    int NBpad = (NB + 7) / 8 * 8;                  // kernels may touch up to NBW-1 blocks past NB
    return (size_t)NBpad * NCh * 64 * 16;
}

void pw_pack_weights(int dtype, const float* w, int K, int N, void* out_host) {
    const int P = per16(dtype);
    int NC = pw_chunks(dtype, K), NCh = (NC + 1) / 2;
    int NB = (N + 31) / 32;
    size_t total = pw_packed_bytes(dtype, K, N);
    __builtin_memset(out_host, 0, total);
    for (int nb = 0; nb < NB; ++nb)
        for (int j = 0; j < NCh; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                int i = lane & 31, h = lane >> 5;
                int ch = slot_channel(nb, i);
                int c = h * NCh + j;
                if (ch >= N || c >= NC) continue;
                char* dst = (char*)out_host + (((size_t)nb * NCh + j) * 64 + lane) * 16;
                const float* src = w + (size_t)ch * K + (size_t)c * P;
                pack_chunk(dtype, src, dst);
            }
    if (dtype == 2) split_pairs_inplace(out_host, (size_t)((NB + 7) / 8 * 8), NCh);      // per n-block: the K chain of a lane half in pairs
}

template <typename T> using Mma = CfMma<T>;      // bf16 / exact fp32 / split-bf16 MFMA over one 16-byte chunk (cf_common.h)

// RES: 0 none, 1 residual add, 2 IDAUp up-branch add
template <typename T, int NBW, int ACT, int RES, bool BIAS>
__global__ __launch_bounds__(256) void pw_kernel(PwParams p) {
    constexpr int P = Elem<T>::PER16;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const long long pb = (long long)blockIdx.x * 4 + wave;
    if (pb * 32 >= p.M) return;                               // wave-uniform
    const long long m = pb * 32 + pl;
    const bool mvalid = m < p.M;
    const long long mr = mvalid ? m : p.M - 1;
    const int NC = (int)((size_t)p.K * sizeof(T) / 16), NCh = (NC + 1) >> 1;
    const int NB = (p.N + 31) >> 5;
    const int nb0 = blockIdx.y * NBW;

    // x rows [m][K], or pixel-block order [m / 32][K / 8][m % 32][8] (xblock): consecutive 16-byte chunks of a pixel are then
    // 512 bytes apart and the 32 lanes of a wave half read one contiguous 512-byte run
    const size_t xs = p.xblock ? 512 : 16;
    const char* xrow = p.xblock ? (const char*)p.x + (((size_t)pb * NC + (size_t)h * NCh) * 32 + pl) * 16
                                : (const char*)p.x + (size_t)mr * p.K * sizeof(T) + (size_t)h * NCh * 16;
    const char* wbase = (const char*)p.wp + ((size_t)nb0 * NCh * 64 + lane) * 16;
    const int jmax = (h == 0) ? NCh : NC - NCh;               // valid chunks of this half

    f32x16 acc[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // four k-steps per trip, every load of the trip in flight before its first MFMA; a chunk past this half's
    // share is read from a clamped address and meets zero weights (pw_pack_weights), so no load is predicated
    const int nbv = NB - nb0 < NBW ? NB - nb0 : NBW;
    for (int j0 = 0; j0 < NCh; j0 += 4) {
        u32x4 xc[4], wc[4][NBW];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u < NCh ? j0 + u : NCh - 1;
            xc[u] = ld16(xrow + (size_t)(j < jmax ? j : 0) * xs);
#pragma unroll
            for (int i = 0; i < NBW; ++i)
                wc[u][i] = ld16(wbase + ((size_t)(i < nbv ? i : 0) * NCh + j) * 1024);
        }
        if constexpr (std::is_same<T, sp32_t>::value) {       // split mode: chunk pairs = one k = 16 step, an odd last chunk alone
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                if (j0 + u + 1 < NCh) {                       // uniform
#pragma unroll
                    for (int i = 0; i < NBW; ++i)
                        if (i < nbv) Mma<T>::run2(acc[i], wc[u][i], wc[u + 1][i], xc[u], xc[u + 1]);
                } else if (j0 + u < NCh) {
#pragma unroll
                    for (int i = 0; i < NBW; ++i)
                        if (i < nbv) Mma<T>::run(acc[i], wc[u][i], xc[u]);
                }
            }
        } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + u < NCh) {                               // uniform
#pragma unroll
                for (int i = 0; i < NBW; ++i)
                    if (i < nbv) Mma<T>::run(acc[i], wc[u][i], xc[u]);
            }
        }
        }
    }

    if (!mvalid) return;
    // ---- epilogue: this lane owns pixel m, channels (nb0+i)*32 + h*16 + [0,16)
    size_t low_row = 0; int tap = 0;
    if constexpr (RES == 2) {
        long long hw = (long long)p.Ho * p.Wo;
        long long b = m / hw; int rem = (int)(m - b * hw);
        int yy = rem / p.Wo, xx = rem - yy * p.Wo;
        low_row = ((size_t)b * (p.Ho >> 1) + (yy >> 1)) * (p.Wo >> 1) + (xx >> 1);
        tap = ((yy & 1) << 1) | (xx & 1);
    }
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        if (nb0 + i >= NB) break;
        const int cb = (nb0 + i) * 32 + h * 16;
#pragma unroll
        for (int g = 0; g < 16 / P; ++g) {
            const int ch = cb + g * P;
            if (ch >= p.N) break;
            float v[P];
#pragma unroll
            for (int e = 0; e < P; ++e) {
                float t = acc[i][g * P + e];
                if constexpr (BIAS) t += p.bias[ch + e];
                v[e] = t;
            }
            act_arr<ACT, P>(v);
            if constexpr (RES == 1) {
                float r[P];
                unpack16<T>(ld16((const char*)p.res + (p.resblock ? blk_off((size_t)m, p.N / P, ch / P) : ((size_t)m * p.N + ch) * sizeof(T))), r);
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] += r[e];
            } else if constexpr (RES == 2) {
                float r[P];
                unpack16<T>(ld16((const char*)p.low + (low_row * p.N + ch) * sizeof(T)), r);
#pragma unroll
                for (int e = 0; e < P; ++e)
                    v[e] += relu_f(r[e] * p.upw[tap * p.N + ch + e] + p.upb[ch + e]);
            }
            const size_t ld = p.ldy ? (size_t)p.ldy : (size_t)p.N;
            st16((char*)p.y + (p.yblock ? blk_off((size_t)m, p.N / P, ch / P) : ((size_t)m * ld + p.yoff + ch) * sizeof(T)), pack16<T>(v));
        }
    }
}

// ------------------------------------------------------------------ weight-in-LDS variant
// For the GEMM-heavy late layers (K, N up to 960 on 20x20 / 40x40 maps) the plain kernel is bound
// by the L1 path that feeds one 1 KiB weight fragment per MFMA (64 B/clk/CU).  Here the four waves
// of a workgroup share every weight fragment: KT k-steps x NBW n-blocks of fragments are DMA'd
// HBM/L2 -> LDS (global_load_lds_dwordx4, lane-linear = fragment order, conflict-free) once per
// workgroup and read back at 256 B/clk/CU.  Activations still stream straight into registers.
template <typename T, int NBW, int ACT, int RES, int NST>
__global__ __launch_bounds__(256) void pw_wlds_kernel(PwParams p) {
    constexpr int P = Elem<T>::PER16;
    constexpr int KT = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [NBW][KT][1 KiB]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const long long pb = (long long)blockIdx.x * 4 + wave;
    const long long m = pb * 32 + pl;
    const bool mvalid = m < p.M;
    const long long mr = mvalid ? m : p.M - 1;
    const int NC = (int)((size_t)p.K * sizeof(T) / 16), NCh = (NC + 1) >> 1;
    const int NB = (p.N + 31) >> 5;
    const int nb0 = blockIdx.y * NBW;
    const size_t xs = p.xblock ? 512 : 16;                        // pixel-block order: see pw_kernel
    const long long pbc = pb * 32 < p.M ? pb : (p.M - 1) / 32;    // waves past the last block read it again (results unused)
    const char* xrow = p.xblock ? (const char*)p.x + (((size_t)pbc * NC + (size_t)h * NCh) * 32 + pl) * 16
                                : (const char*)p.x + (size_t)mr * p.K * sizeof(T) + (size_t)h * NCh * 16;
    const int jmax = (h == 0) ? NCh : NC - NCh;

    f32x16 acc[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // NST-stage ring: the weight DMA and the activation loads of K tile t + NST - 1 are issued while tile t
    // runs.  The activations of these late layers come from HBM (~2 us); the K loop of one workgroup is a
    // chain of 15 such tiles, so the kernel time was that chain's latency at every batch size.  Waits are
    // PARTIAL (vmcnt(n): loads return in order, so "all but the n youngest" = tile t has landed): a
    // __syncthreads() would wait for the prefetches as well.
    constexpr int STAGE = NBW * KT * 1024;
    constexpr int VM_PER_TILE = NBW * KT / 4 + KT;               // vm instructions one wave issues per tile
    const int NT = (NCh + KT - 1) / KT;
    auto prefetch = [&](int t, char* stage, u32x4* xn) {
        const int j0 = t * KT;
        // always NBW*KT/4 DMA instructions per wave (clamped source for the tail / missing n-blocks) so that the
        // outstanding-instruction count per tile is a compile-time constant
#pragma unroll
        for (int q = 0; q < NBW * KT / 4; ++q) {                 // straight-line: the compiler must be able to count vm ops
            const int c = wave + 4 * q;
            const int i = c / KT, jj = c - i * KT;
            const int ii = nb0 + i < NB ? nb0 + i : NB - 1, jc = j0 + jj < NCh ? j0 + jj : NCh - 1;
            const char* src = (const char*)p.wp + ((((size_t)ii * NCh + jc) * 64) + lane) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(stage + (i * KT + jj) * 1024), 16, 0, 0);
        }
        // no select on the loaded value here (it would force the wait at issue time): a chunk past this
        // half's share is read from a clamped address and meets ZERO weights (pw_pack_weights leaves the
        // fragment of a missing chunk zero); whole k-steps past NCh are skipped at consumption
#pragma unroll
        for (int jj = 0; jj < KT; ++jj) xn[jj] = ld16(xrow + (size_t)(j0 + jj < jmax ? j0 + jj : 0) * xs);
    };
    const int nbv = NB - nb0 < NBW ? NB - nb0 : NBW;
    u32x4 xr[NST][KT];
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
        if (s0 < NT) prefetch(s0, smem + s0 * STAGE, xr[s0]);
    for (int t0 = 0; t0 < NT; t0 += NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int t = t0 + u;
            if (t >= NT) break;
            // tiles t+1 .. t+NST-2 may still be in flight: wait for everything older than them
            const int younger = min(NT - 1 - t, NST - 2);
            if (NST >= 4 && younger >= 2) __builtin_amdgcn_s_waitcnt(((2 * VM_PER_TILE) & 0xF) | (((2 * VM_PER_TILE) >> 4) << 14) | 0x0F70);
            else if (NST >= 3 && younger >= 1) __builtin_amdgcn_s_waitcnt((VM_PER_TILE & 0xF) | ((VM_PER_TILE >> 4) << 14) | 0x0F70);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();     // every wave's share of tile t is in LDS; all are done with tile t-1's stage
            if (t + NST - 1 < NT) prefetch(t + NST - 1, smem + ((u + NST - 1) % NST) * STAGE, xr[(u + NST - 1) % NST]);
            const int kt = NCh - t * KT < KT ? NCh - t * KT : KT;
            const char* st = smem + u * STAGE;
            if constexpr (std::is_same<T, sp32_t>::value) {      // split mode: chunk pairs (KT is even: pairs never straddle a tile)
#pragma unroll
                for (int jj = 0; jj < KT; jj += 2) {
                    if (jj + 1 < kt) {
#pragma unroll
                        for (int i = 0; i < NBW; ++i)
                            if (i < nbv) Mma<T>::run2(acc[i], ld16(st + (i * KT + jj) * 1024 + lane * 16), ld16(st + (i * KT + jj + 1) * 1024 + lane * 16),
                                                      xr[u][jj], xr[u][jj + 1]);
                    } else if (jj < kt) {
#pragma unroll
                        for (int i = 0; i < NBW; ++i)
                            if (i < nbv) Mma<T>::run(acc[i], ld16(st + (i * KT + jj) * 1024 + lane * 16), xr[u][jj]);
                    }
                }
            } else if (nbv == NBW) {                              // wave-uniform: all n-blocks of this workgroup exist
#pragma unroll
                for (int jj = 0; jj < KT; ++jj) {
                    if (jj < kt) {
#pragma unroll
                        for (int i = 0; i < NBW; ++i) Mma<T>::run(acc[i], ld16(st + (i * KT + jj) * 1024 + lane * 16), xr[u][jj]);
                    }
                }
            } else {                                       // last n-group of a layer whose N is not a multiple of 32 NBW
#pragma unroll
                for (int jj = 0; jj < KT; ++jj) {
                    if (jj < kt) {
#pragma unroll
                        for (int i = 0; i < NBW; ++i)
                            if (i < nbv) Mma<T>::run(acc[i], ld16(st + (i * KT + jj) * 1024 + lane * 16), xr[u][jj]);
                    }
                }
            }
        }
    }
    if (!mvalid) return;
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        if (nb0 + i >= NB) break;
        const int cb = (nb0 + i) * 32 + h * 16;
#pragma unroll
        for (int g = 0; g < 16 / P; ++g) {
            const int ch = cb + g * P;
            if (ch >= p.N) break;
            float v[P];
#pragma unroll
            for (int e = 0; e < P; ++e) v[e] = acc[i][g * P + e];
            act_arr<ACT, P>(v);
            if constexpr (RES == 1) {
                float r[P];
                unpack16<T>(ld16((const char*)p.res + (p.resblock ? blk_off((size_t)m, p.N / P, ch / P) : ((size_t)m * p.N + ch) * sizeof(T))), r);
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] += r[e];
            }
            st16((char*)p.y + (p.yblock ? blk_off((size_t)m, p.N / P, ch / P) : ((size_t)m * p.N + ch) * sizeof(T)), pack16<T>(v));
        }
    }
}

// ------------------------------------------------------------------ K split over the four waves of a workgroup
// Small batches of the late layers (K up to 960 on 40x40 / 20x20 maps; configs[4]'s four images per GPU): M / 128 workgroups
// leave most CUs idle and the kernel time is the latency of one wave's K chain (15 tiles of loads, ~2 us each).  Here a
// workgroup owns 32 pixels instead of 128 and its four waves each walk a QUARTER of K (weights straight from L2: the fragment
// stream of a wave is contiguous), so there are four times as many workgroups and the chain is a quarter as long.  The four
// partial accumulators meet in LDS and are added in wave order 0, 1, 2, 3 (deterministic), each wave finishing a quarter of the
// accumulator registers (4 channels x 32 pixels x NBW n-blocks).
template <int NBW, int ACT, int RES>
__global__ __launch_bounds__(256) void pw_ksplit_kernel(PwParams p) {
    typedef bf16_t T;                                                // bf16 storage only (the fp32 parity mode keeps pw_wlds_kernel)
    constexpr int DEPTH = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [wave][NBW][4 register quads][64 lanes] x 16 B
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const long long pb = blockIdx.x;
    const long long m = pb * 32 + pl;
    const bool mvalid = m < p.M;
    const long long mr = mvalid ? m : p.M - 1;
    const int NC = (int)((size_t)p.K * sizeof(T) / 16), NCh = (NC + 1) >> 1;
    const int NB = (p.N + 31) >> 5;
    const int nb0 = blockIdx.y * NBW;
    const size_t xs = p.xblock ? 512 : 16;
    const char* xrow = p.xblock ? (const char*)p.x + (((size_t)pb * NC + (size_t)h * NCh) * 32 + pl) * 16
                                : (const char*)p.x + (size_t)mr * p.K * sizeof(T) + (size_t)h * NCh * 16;
    const int jmax = (h == 0) ? NCh : NC - NCh;
    const int per = (NCh + 3) >> 2;                                   // k-steps per wave
    const int j0 = wave * per, j1 = min(j0 + per, NCh);
    // n-blocks past NB (last group of a layer whose NB is not a multiple of NBW) read the layer's last block: finite values, never stored
    size_t woff[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) woff[i] = ((size_t)min(nb0 + i, NB - 1) * NCh * 64 + lane) * 16;

    f32x16 acc[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    u32x4 xr[DEPTH], wr[DEPTH][NBW];
    auto fetch = [&](int j, int slot) {
        // a chunk past this half's share is read from a clamped address and meets zero weights (pw_pack_weights)
        xr[slot] = ld16(xrow + (size_t)(j < jmax ? j : 0) * xs);
#pragma unroll
        for (int i = 0; i < NBW; ++i) wr[slot][i] = ld16((const char*)p.wp + woff[i] + (size_t)j * 1024);
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
        if (j0 + d < j1) fetch(j0 + d, d);
    for (int jb = j0; jb < j1; jb += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int j = jb + u;
            if (j >= j1) break;
            if (j + DEPTH - 1 < j1) fetch(j + DEPTH - 1, (u + DEPTH - 1) % DEPTH);
#pragma unroll
            for (int i = 0; i < NBW; ++i) Mma<T>::run(acc[i], wr[u][i], xr[u]);
        }
    }
    // ---- partial sums -> LDS, then wave w finishes register quad w of every n-block
    {
        f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
        for (int i = 0; i < NBW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v; v[0] = acc[i][4 * q]; v[1] = acc[i][4 * q + 1]; v[2] = acc[i][4 * q + 2]; v[3] = acc[i][4 * q + 3];
                red[((wave * NBW + i) * 4 + q) * 64 + lane] = v;
            }
    }
    __syncthreads();
    if (!mvalid) return;
    const f32x4* red = reinterpret_cast<const f32x4*>(smem);
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        if (nb0 + i >= NB) break;
        const int ch = (nb0 + i) * 32 + h * 16 + wave * 4;            // channels of register quad `wave`
        if (ch >= p.N) continue;
        f32x4 v = red[((0 * NBW + i) * 4 + wave) * 64 + lane];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) v += red[((w2 * NBW + i) * 4 + wave) * 64 + lane];
        float o[4] = {v[0], v[1], v[2], v[3]};
        act_arr<ACT, 4>(o);
        // 8-byte element group inside the 16-byte group of eight channels (row-major or pixel-block order)
        const size_t off = (p.yblock ? blk_off((size_t)m, p.N / 8, ch / 8) : ((size_t)m * p.N + (ch & ~7)) * sizeof(T)) + (ch & 4) * sizeof(T);
        if constexpr (RES == 1) {
            const size_t roff = (p.resblock ? blk_off((size_t)m, p.N / 8, ch / 8) : ((size_t)m * p.N + (ch & ~7)) * sizeof(T)) + (ch & 4) * sizeof(T);
            const u32x2 rv = *reinterpret_cast<const u32x2*>((const char*)p.res + roff);
            o[0] += bf16_to_f32((uint16_t)(rv.x & 0xffffu)); o[1] += bf16_to_f32((uint16_t)(rv.x >> 16));
            o[2] += bf16_to_f32((uint16_t)(rv.y & 0xffffu)); o[3] += bf16_to_f32((uint16_t)(rv.y >> 16));
        }
        u32x2 pk; pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]);
        *reinterpret_cast<u32x2*>((char*)p.y + off) = pk;
    }
}

template <int NBW>
static hipError_t dispatch_ksplit(hipStream_t s, const PwParams& p) {
    const int NB = (p.N + 31) / 32;
    dim3 grid((unsigned)((p.M + 31) / 32), (unsigned)((NB + NBW - 1) / NBW)), blk(256);
    const size_t lds = (size_t)4 * NBW * 4 * 64 * 16;
    const int res = p.res ? 1 : 0;
#define CF_PWK_LAUNCH(ACT, RES) \
    set_kernel_tag("void cf::pw_ksplit_kernel<%d, %d, %d>(cf::PwParams)", NBW, ACT, RES); \
    hipLaunchKernelGGL((pw_ksplit_kernel<NBW, ACT, RES>), grid, blk, lds, s, p); return hipGetLastError();
    if (p.act == 1 && res == 0) { CF_PWK_LAUNCH(1, 0) }
    if (p.act == 0 && res == 0) { CF_PWK_LAUNCH(0, 0) }
    if (p.act == 0 && res == 1) { CF_PWK_LAUNCH(0, 1) }
#undef CF_PWK_LAUNCH
    return hipErrorInvalidValue;
}

template <typename T, int NBW, int NST>
static hipError_t dispatch_wlds(hipStream_t s, const PwParams& p, dim3 grid) {
    dim3 blk(256);
    const size_t lds = (size_t)NST * NBW * 4 * 1024;            // NST stages of [NBW][KT = 4][1 KiB]
    const int res = p.res ? 1 : 0;
#define CF_PWL_LAUNCH(ACT, RES) \
    set_kernel_tag("void cf::pw_wlds_kernel<%s, %d, %d, %d, %d>(cf::PwParams)", type_tag<T>(), NBW, ACT, RES, NST); \
    hipLaunchKernelGGL((pw_wlds_kernel<T, NBW, ACT, RES, NST>), grid, blk, lds, s, p); return hipGetLastError();
    if (p.act == 1 && res == 0) { CF_PWL_LAUNCH(1, 0) }
    if (p.act == 0 && res == 0) { CF_PWL_LAUNCH(0, 0) }
    if (p.act == 0 && res == 1) { CF_PWL_LAUNCH(0, 1) }
#undef CF_PWL_LAUNCH
    return hipErrorInvalidValue;
}

template <typename T, int NBW>
static hipError_t dispatch_epi(hipStream_t s, const PwParams& p, dim3 grid) {
    dim3 blk(256);
    const bool bias = p.bias != nullptr;
    const int res = p.low ? 2 : (p.res ? 1 : 0);
#define CF_PW_LAUNCH(ACT, RES, BIAS) \
    set_kernel_tag("void cf::pw_kernel<%s, %d, %d, %d, %s>(cf::PwParams)", type_tag<T>(), NBW, ACT, RES, BIAS ? "true" : "false"); \
    hipLaunchKernelGGL((pw_kernel<T, NBW, ACT, RES, BIAS>), grid, blk, 0, s, p); return hipGetLastError();
    if (p.act == 1 && res == 0 && !bias) { CF_PW_LAUNCH(1, 0, false) }   // expand + Swish
    if (p.act == 0 && res == 0 && !bias) { CF_PW_LAUNCH(0, 0, false) }   // project
    if (p.act == 0 && res == 1 && !bias) { CF_PW_LAUNCH(0, 1, false) }   // project + residual
    if (p.act == 1 && res == 0 && bias)  { CF_PW_LAUNCH(1, 0, true) }    // conv_1x1_bn
    if (p.act == 2 && res == 2 && bias)  { CF_PW_LAUNCH(2, 2, true) }    // IDAUp
    if (p.act == 2 && res == 0 && bias)  { CF_PW_LAUNCH(2, 0, true) }    // ShuffleV2 pw+BN+ReLU
    if (p.act == 0 && res == 0 && bias)  { CF_PW_LAUNCH(0, 0, true) }    // plain conv + bias
#undef CF_PW_LAUNCH
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t dispatch_nbw(hipStream_t s, const PwParams& p) {
    const int NB = (p.N + 31) / 32;
    const long long gx = (p.M + 127) / 128;
    // GEMM-heavy layers (no bias / IDAUp epilogue): share weight fragments through LDS
    static const int wl_env = cf_ab_int("CF_PW_WLDS", -1);     // A/B: 0 off, N = force NBW
    if (wl_env != 0 && !p.bias && !p.low && !p.ldy && p.K >= 64 && NB >= 3 && (p.act == 1 || p.act == 0)) {
        (void)wl_env;                          // NBW = 4 measured best of {4,5,6,8} on every late layer
        if constexpr (sizeof(T) == 2) {
            // K split over the waves on the >= 32x32 late maps of 1280-class inputs (configs[4]: four images per GPU leave M / 128 =
            // 50 workgroups).  Chosen by the LAYER's shape, never by the batch: the two kernels sum K in different orders, and an
            // image's result must not depend on the batch it travels in.  B = 4, 1280x1280: 20.6 -> 10.8, 24.0 -> 14.6 us (layer5.1 /
            // 6.0 project, K = 960); slower than pw_wlds_kernel at B = 64, 640x640 (20x20 maps: 23 -> 29, 30 -> 49 us; layer4.1's K = 576
            // on its 40x40 map: 32 -> 35 us, N = 96), hence K >= 512 AND N >= 128: layer5.0 / 5.1 / 6.0 of 1280-class inputs (13.7 -> 8.8 us for 5.0)
            static const int ks_env = cf_ab_int("CF_PW_KSPLIT", -1);   // A/B: 0 off, 1 force
            if (ks_env != 0 && p.K >= 256 && (p.act == 0 || !p.res) /* the k-split epilogues: Swish, none, none + residual */ && (ks_env > 0 || (p.K >= 512 && p.N >= 128 && (long long)p.Ho * p.Wo >= 1024 && (long long)p.Ho * p.Wo < 4096)))
                return (NB % 3 == 0 || NB == 5) ? dispatch_ksplit<3>(s, p) : dispatch_ksplit<2>(s, p);
        }
        static const int nst_env = cf_ab_int("CF_PW_NST", 0);       // A/B: ring depth 2..4
        static const int nbw_env = cf_ab_int("CF_PW_NBW", 0);
        // N = 320: five n-blocks per wave (activations read twice); N = 160: 3 + 2 (two workgroup rows: 200 -> 400 workgroups on
        // the 20x20 maps at B = 64, 16.7 -> 16.0 and 25.2 -> 23.1 us); N = 96: three
        // fp32-width outputs (split / exact mode) of the expand GEMMs: three n-blocks per wave (N = 576: six balanced workgroup rows instead of 4 + 4 + 4 + 4 + 2;
        // B = 64 split mode: layer5.0 / 5.1 / 6.0 expand 0.132 / 0.071 / 0.070 -> 0.118 / 0.064 / 0.064 ms; the project GEMMs keep 3 / 5: 6.0 project 0.088 -> 0.108 with three)
        const bool wide_f32_expand = sizeof(T) == 4 && p.act == 1 && NB % 3 == 0 && NB >= 12;
        const int nbw = nbw_env ? nbw_env : wide_f32_expand ? 3 : (NB == 3 || NB == 5 ? 3 : (NB % 5 == 0 ? 5 : 4));
        dim3 grid((unsigned)gx, (unsigned)((NB + nbw - 1) / nbw));
        // ring depth: 3-4 stages (<= 64 KB of LDS) when the grid is at most half a workgroup per CU (small batches: the
        // K chain's latency is the kernel time), 2 stages (more workgroups per CU) otherwise -- measured
        const int nst = nst_env ? nst_env : ((long long)grid.x * grid.y <= 128 ? (nbw == 5 ? 3 : 4) : 2);
        if (nbw == 5) return nst == 2 ? dispatch_wlds<T, 5, 2>(s, p, grid) : dispatch_wlds<T, 5, 3>(s, p, grid);
        if (nbw == 3) return nst == 2 ? dispatch_wlds<T, 3, 2>(s, p, grid) : dispatch_wlds<T, 3, 4>(s, p, grid);
        switch (nst) {
            case 2: return dispatch_wlds<T, 4, 2>(s, p, grid);
            case 3: return dispatch_wlds<T, 4, 3>(s, p, grid);
            default: return dispatch_wlds<T, 4, 4>(s, p, grid);
        }
    }
    // n-blocks per wave: as many as fit 64..128 accumulator VGPRs, fewer when the grid would be
    // too small to fill 256 CUs x 8 waves.
    int nbw = NB >= 4 ? 4 : NB;
    if (NB > 4 && (NB % 3 == 0) && (NB % 4 != 0)) nbw = 3;
    if (NB == 5) nbw = 5;
    while (nbw > 1 && gx * ((NB + nbw - 1) / nbw) < 1024) nbw = (nbw + 1) / 2;
    dim3 grid((unsigned)gx, (unsigned)((NB + nbw - 1) / nbw));
    switch (nbw) {
        case 1: return dispatch_epi<T, 1>(s, p, grid);
        case 2: return dispatch_epi<T, 2>(s, p, grid);
        case 3: return dispatch_epi<T, 3>(s, p, grid);
        case 4: return dispatch_epi<T, 4>(s, p, grid);
        default: return dispatch_epi<T, 5>(s, p, grid);
    }
}

template <typename T>
__global__ void shuffle_copy_kernel(const T* x, T* y, long long M, int C, int phase, int ldy, int yoff) {
    const long long n = M * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C; const int c = (int)(i - m * C);
        y[m * ldy + yoff + c] = x[m * (2 * C) + 2 * c + phase];
    }
}
hipError_t launch_shuffle_copy(hipStream_t s, int dtype, const void* x, void* y, long long M, int C, int phase, int ldy, int yoff) {
    if (M <= 0 || C <= 0) return hipSuccess;
    long long g = (M * C + 255) / 256; if (g > 16384) g = 16384;
    if (dtype != 1) hipLaunchKernelGGL(shuffle_copy_kernel<float>, dim3((unsigned)g), dim3(256), 0, s, (const float*)x, (float*)y, M, C, phase, ldy, yoff);
    else hipLaunchKernelGGL(shuffle_copy_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, M, C, phase, ldy, yoff);
    return hipGetLastError();
}

hipError_t launch_pw(hipStream_t s, int dtype, const PwParams& p) {
    if (p.M <= 0) return hipSuccess;
    if (p.K % 8 || p.N % 8) return hipErrorInvalidValue;
    if (p.ldy && ((p.ldy * (int)elem_size(dtype)) % 16 || (p.yoff * (int)elem_size(dtype)) % 16 || p.yoff + p.N > p.ldy)) return hipErrorInvalidValue;
    return dtype == 0 ? dispatch_nbw<float>(s, p) : dtype == 2 ? dispatch_nbw<sp32_t>(s, p) : dispatch_nbw<bf16_t>(s, p);
}

}  // namespace cf
