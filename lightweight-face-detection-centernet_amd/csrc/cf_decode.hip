// Peak decode on the GPU.
//
// D3 (peak_collect_kernel + topk_select_kernel) replaces ctdet_decode and its helpers (centerface_ext.py:11-82):
//   _nms   (:44-50)  3x3 max-pool equality mask            -> fused peak test, no pooled tensor
//   _topk  (:11-27)  torch.topk over H*W, index -> (y, x)   -> exact radix select + bitonic sort
//   _transpose_and_gather_feat (:28-42) full NCHW->NHWC permute + gather -> K 64-byte row reads
//   box assembly (:60-82)
// Scores are made totally ordered by the composite 64-bit key (orderable(score) << 32) | ~index, so "equal
// scores: lower index first" is part of the key and the selected set is exact and deterministic (torch leaves
// ties unspecified).
//
// Two kernels.  After `heat * (hmax == heat)` every cell that is not a 3x3 peak scores exactly +0.0, and those are
// ~8/9 of a map: they need no key at all -- among themselves they are ordered by index.  So
//   1. peak_collect_kernel, grid (cells / 4096, B): the peak test (all 9 loads of 4 cells per thread in flight) and a
//      wave-aggregated append of the composite keys of the cells whose kept score is NOT +0.0 to a per-image list.
//      Many workgroups per image: a 4-image 1280x1280 batch fills 100 CUs instead of 4.
//   2. topk_select_kernel, grid (B): MSB-first radix select of the K-th largest key over the LIST (typically 11 %
//      of the map; staged in LDS when it fits), compaction, bitonic sort (wave shuffles / LDS; global memory for
//      K > 1024), then -- only when the list holds fewer than K positive scores -- the lowest-index zero cells in
//      index order, then the gather and box assembly.
// HBM traffic: the heat plane once (4 B/cell) + 8 B per listed cell + K records, against 8 B/cell written and
// re-read up to four times by the one-kernel version this replaces.  No limit on K (<= H*W) or on the map size.
//
// D1 (threshold decode + greedy NMS) replaces CenterFace.decode / CenterFace.nms
// (centerface.py:73-151) with the reference's arithmetic (float64 intermediates rounded to fp32,
// float32 IoU with +1 areas, `>=` threshold), row-major candidate order and score-descending
// suppression order (ties: higher candidate index first, = argsort()[::-1] of a stable sort).
#include "cf_common.h"
#include "cf_kernels.h"

namespace cf {

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
constexpr uint32_t kZeroOk = 0x80000000u;              // orderable(+0.0f)

// heat * keep for cell (y, x) of one image plane: keep = 1.0 where the 3x3 maximum equals the cell
// (max_pool2d pads with -inf = clamped in-range duplicates), "+ 0.0f" canonicalises -0 to +0
__device__ __forceinline__ float kept_score(const float* hm, int hs, int h, int w, int i) {
    const int y = i / w, x = i - y * w;
    const float v = hm[(size_t)i * hs];
    float mx = v;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = min(max(y + dy, 0), h - 1), xx = min(max(x + dx, 0), w - 1);
            mx = fmaxf(mx, hm[((size_t)yy * w + xx) * hs]);
        }
    return (mx == v) ? v : (v * 0.0f + 0.0f);
}

constexpr int kCollectCells = 1024;                    // cells per workgroup of peak_collect_kernel: 4 per thread

// One group of four cells per thread: the chain loads -> ballot -> reserve -> stores is walked once, and a
// 64 x 640x640 batch is 1600 workgroups (6 waves per SIMD in flight).
__global__ __launch_bounds__(256) void peak_collect_kernel(TopkParams p) {
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int HW = p.h * p.w;
    const float* hm = p.hm_plane ? p.hm_plane + (size_t)b * HW : p.heads + (size_t)b * HW * 16;
    const int hs = p.hm_plane ? 1 : 16;
    u64* keys = p.scratch + (size_t)b * HW;
    const int base = blockIdx.x * kCollectCells;
    float v[4], mx[4];
    int idx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = base + u * 256 + tid;
        idx[u] = i;
        const int ic = i < HW ? i : HW - 1;
        const int y = ic / p.w, x = ic - y * p.w;
        v[u] = hm[(size_t)ic * hs];
        mx[u] = v[u];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = min(max(y + dy, 0), p.h - 1), xx = min(max(x + dx, 0), p.w - 1);
                mx[u] = fmaxf(mx[u], hm[((size_t)yy * p.w + xx) * hs]);
            }
    }
    uint32_t ok[4]; bool hit[4]; unsigned long long bal[4]; uint32_t tot = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float kept = (mx[u] == v[u]) ? v[u] : (v[u] * 0.0f + 0.0f);
        ok[u] = orderable(kept);
        hit[u] = idx[u] < HW && ok[u] != kZeroOk;
        bal[u] = __ballot(hit[u]);
        tot += (uint32_t)__popcll(bal[u]);
    }
    // one returning atomic per WORKGROUP, on a counter that owns its cache line (kTopkCountStride): returning atomics
    // on one line serialise in the L2 (~8 ns each) -- per-wave atomics on 64 adjacent counters made this kernel 36-52 us
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t wbase;
    const int wave = tid >> 6;
    if (lane == 0) wtot[wave] = tot;
    __syncthreads();
    if (tid == 0) {
        const uint32_t all = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        wbase = all ? (uint32_t)atomicAdd(&p.count[(size_t)b * kTopkCountStride], (int)all) : 0u;
    }
    __syncthreads();
    if (tot == 0) return;                               // wave-uniform
    uint32_t off = wbase;
    for (int w2 = 0; w2 < wave; ++w2) off += wtot[w2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (hit[u]) keys[off + (uint32_t)__popcll(bal[u] & ((1ull << lane) - 1ull))] = ((u64)ok[u] << 32) | (u64)(0xffffffffu - (uint32_t)idx[u]);
        off += (uint32_t)__popcll(bal[u]);
    }
}

// suffix-inclusive scan over NBINS LDS counters by wave 0; finds digit d with
// sum(hist[d+1..]) < kth <= sum(hist[d..]); returns d and the count above it via LDS result slots.
template <int NBINS>
__device__ __forceinline__ void find_digit(const uint32_t* hist, uint32_t kth, uint32_t* res /*[3]: digit, count above, count in bin*/) {
    // called by all threads; wave 0 does the work
    if (threadIdx.x < 64) {
        constexpr int PER = NBINS / 64;
        const int lane = threadIdx.x;
        // lane L owns bins [ (63-L)*PER, (63-L+1)*PER ) so that lane order == descending digit order
        const int base = (63 - lane) * PER;
        uint32_t local = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) local += hist[base + i];
        // inclusive prefix over lanes (descending digits)
        uint32_t incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        const uint32_t excl = incl - local;          // count in strictly higher lanes' bins
        if (excl < kth && kth <= incl) {             // exactly one lane
            uint32_t above = excl;
            for (int i = PER - 1; i >= 0; --i) {
                uint32_t c = hist[base + i];
                if (above + c >= kth) { res[0] = (uint32_t)(base + i); res[1] = above; res[2] = c; break; }
                above += c;
            }
        }
    }
}

constexpr int kStageCap = 4096;                        // list entries staged in LDS (32 KB)

// BIG = false: K <= NT, survivors live in registers / LDS.  BIG = true: any K, survivors, sort and the final
// order live in p.big (global; one workgroup per image, so __syncthreads orders its accesses).  NT = 256 threads
// for K <= 256 (the usual top-100: four waves make every barrier cheap), 1024 otherwise.
template <int NT, bool BIG>
__global__ __launch_bounds__(NT) void topk_select_kernel(TopkParams p) {
    __shared__ uint32_t hist3[3][2048];                 // one histogram per score pass, zeroed once up front
    __shared__ uint32_t res[3];
    __shared__ uint32_t nsel, npos_s, zfound;
    __shared__ uint32_t wave_cnt[NT / 64];
    __shared__ u64 sel[BIG ? 64 : NT];
    __shared__ u64 staged[kStageCap];

    const int b = blockIdx.x;
    const int HW = p.h * p.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CF_TOPK_TIMING
    unsigned long long tm[12]; int tn = 0;
#define CF_TT() tm[tn++] = __builtin_amdgcn_s_memtime()
#else
#define CF_TT()
#endif
    CF_TT();
    const int K = p.K;
    const float* hm = p.hm_plane ? p.hm_plane + (size_t)b * HW : p.heads + (size_t)b * HW * 16;
    const int hs = p.hm_plane ? 1 : 16;
    const int L = p.count[(size_t)b * kTopkCountStride];
    const u64* list = p.scratch + (size_t)b * HW;
    for (int i = tid; i < 3 * 2048; i += NT) (&hist3[0][0])[i] = 0;
    if (tid == 0) npos_s = 0;
    if (L <= kStageCap) {                               // workgroup-uniform
        for (int i = tid; i < L; i += NT) staged[i] = list[i];
        list = staged;
    }
    __syncthreads();
    CF_TT();
    // ---- pass 0: histogram of the top 11 score bits over the list; positives = bins >= 1024 (the list never holds +0)
    {
        uint32_t mypos = 0;
        for (int i = tid; i < L; i += NT) {
            const uint32_t ok = (uint32_t)(list[i] >> 32);
            atomicAdd(&hist3[0][ok >> 21], 1u);
            mypos += ok > kZeroOk ? 1u : 0u;
        }
        for (int off = 32; off > 0; off >>= 1) mypos += __shfl_xor(mypos, off);
        if (lane == 0 && mypos) atomicAdd(&npos_s, mypos);
    }
    __syncthreads();
    CF_TT();
    const uint32_t Ppos = npos_s;
    const uint32_t Z = (uint32_t)(HW - L);              // cells whose kept score is exactly +0: ordered by index
    // how many list entries the result holds (M), how many zero cells (nz): positives, then zeros, then negatives
    uint32_t M, nz;
    if (Ppos >= (uint32_t)K) { M = K; nz = 0; }
    else { nz = min((uint32_t)K - Ppos, Z); M = (uint32_t)K - nz; }
    const uint32_t lead = min(Ppos, M);                 // list entries that precede the zero cells in the output

    // ---- radix select of the M-th largest composite key of the list, MSB first: 11+11+10 score bits, then the
    //      index bits (only when scores tie at the threshold)
    u64 thresh = 0;                                     // M == L: everything
    if (M > 0 && M < (uint32_t)L) {
        u64 prefix = 0, mask = 0;
        uint32_t kth = M;
        int ibits = 1;
        while ((1ll << ibits) < (long long)HW) ++ibits;
        bool done = false;
        for (int pass = 0; pass < 3 && !done; ++pass) {
            const int sh = pass == 0 ? 53 : (pass == 1 ? 42 : 32), wd = pass == 2 ? 10 : 11;
            uint32_t* hist = hist3[pass];
            if (pass > 0) {                                  // pass 0's histogram is already there
                for (int i = tid; i < L; i += NT) {
                    const u64 k = list[i];
                    if ((k & mask) == prefix) atomicAdd(&hist[(uint32_t)(k >> sh) & ((1u << wd) - 1u)], 1u);
                }
                __syncthreads();
            }
            find_digit<2048>(hist, kth, res);
            __syncthreads();
            prefix |= (u64)res[0] << sh;
            mask |= (u64)((1u << wd) - 1u) << sh;
            kth -= res[1];
            // every key that shares the fixed bits is needed: nothing left to break
            if (kth == res[2]) done = true;
        }
        if (!done) {
            // ~index occupies the low 32 bits; its bits above `ibits` are all ones for every key
            const u64 hi_ones = (ibits < 32) ? ((0xffffffffull >> ibits) << ibits) : 0ull;
            prefix |= hi_ones; mask |= hi_ones;
            uint32_t* hist = hist3[0];
            for (int rem = ibits; rem > 0 && !done;) {
                const int wd = rem < 11 ? rem : 11, sh = rem - wd;
                __syncthreads();                             // everyone has read res / the previous histogram
                for (int i = tid; i < 2048; i += NT) hist[i] = 0;
                __syncthreads();
                for (int i = tid; i < L; i += NT) {
                    const u64 k = list[i];
                    if ((k & mask) == prefix) atomicAdd(&hist[(uint32_t)(k >> sh) & ((1u << wd) - 1u)], 1u);
                }
                __syncthreads();
                find_digit<2048>(hist, kth, res);
                __syncthreads();
                prefix |= (u64)res[0] << sh;
                mask |= (u64)((1u << wd) - 1u) << sh;
                kth -= res[1];
                if (kth == res[2]) done = true;
                rem = sh;
            }
        }
        thresh = prefix;                                // keys >= thresh (masked bits only) are exactly the M survivors
    }

    CF_TT();
    // ---- compaction of the M survivors (order irrelevant: sorted next)
    int sortn = 64;
    while (sortn < (int)M) sortn <<= 1;
    u64* gbuf = BIG ? p.big + (size_t)b * p.big_stride : nullptr;      // [sortn] sort buffer, then [K] final order
    if (tid == 0) nsel = 0;
    if constexpr (BIG) for (int i = tid; i < sortn; i += NT) gbuf[i] = 0ull;
    __syncthreads();
    if (M > 0) {
        for (int i = tid; i < L; i += NT) {
            const u64 k = list[i];
            if (k >= thresh) {
                const uint32_t pos = atomicAdd(&nsel, 1u);
                if constexpr (BIG) gbuf[pos] = k; else if (pos < (uint32_t)NT) sel[pos] = k;
            }
        }
    }
    __syncthreads();

    CF_TT();
    u64 mine = 0ull;
    if constexpr (!BIG) {
        mine = (tid < (int)M) ? sel[tid] : 0ull;
        __syncthreads();
        // bitonic sort, descending, one key per thread over the next power of two >= M (padding keys 0 sink to the end)
        for (int k2 = 2; k2 <= sortn; k2 <<= 1) {
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                u64 other;
                if (j >= 64) {
                    sel[tid] = mine;
                    __syncthreads();
                    other = sel[tid ^ j];
                    __syncthreads();
                } else {
                    uint32_t lo = __shfl_xor((uint32_t)mine, j), hi = __shfl_xor((uint32_t)(mine >> 32), j);
                    other = ((u64)hi << 32) | lo;
                }
                const bool up = (tid & k2) == 0;            // descending block
                const bool lower = (tid & j) == 0;
                const bool take_max = (up == lower);
                mine = take_max ? (mine > other ? mine : other) : (mine < other ? mine : other);
            }
        }
        __syncthreads();
        // final order: list entries before the zero cells, the zero cells, the rest of the list entries
        if (tid < (int)M) sel[(uint32_t)tid < lead ? tid : tid + nz] = mine;
    } else {
        for (int k2 = 2; k2 <= sortn; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < sortn / 2; t += NT) {
                    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;       // the pair (lo, lo ^ j)
                    const u64 a = gbuf[lo], c = gbuf[hi];
                    const bool desc = (lo & k2) == 0;
                    if (desc ? (a < c) : (a > c)) { gbuf[lo] = c; gbuf[hi] = a; }
                }
                __syncthreads();
            }
        u64* fin = gbuf + sortn;
        for (int t = tid; t < (int)M; t += NT) fin[(uint32_t)t < lead ? t : t + nz] = gbuf[t];
    }

    CF_TT();
    // ---- the zero cells (kept score exactly +0.0), lowest index first: ordered scan in rounds of NT cells
    if (nz > 0) {
        if (tid == 0) zfound = 0;
        __syncthreads();
        for (int i0 = 0; i0 < HW; i0 += NT) {
            const uint32_t have = zfound;                   // uniform (read after the barrier below / above)
            if (have >= nz) break;
            const int i = i0 + tid;
            const bool hit = i < HW && orderable(kept_score(hm, hs, p.h, p.w, i)) == kZeroOk;
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t off = have, total = 0;
            for (int w2 = 0; w2 < NT / 64; ++w2) { if (w2 < wave) off += wave_cnt[w2]; total += wave_cnt[w2]; }
            if (hit) {
                const uint32_t pos = off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (pos < nz) {
                    const u64 k = ((u64)kZeroOk << 32) | (u64)(0xffffffffu - (uint32_t)i);
                    if constexpr (BIG) gbuf[sortn + lead + pos] = k; else sel[lead + pos] = k;
                }
            }
            __syncthreads();
            if (tid == 0) zfound = have + total;
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid == 0) p.count[(size_t)b * kTopkCountStride] = 0;     // leave the list empty for the next launch

    CF_TT();
    // ---- gather + box assembly (centerface_ext.py:60-82)
    for (int t = tid; t < K; t += NT) {
        const u64 key = BIG ? gbuf[sortn + t] : sel[t];
        const uint32_t idx = 0xffffffffu - (uint32_t)key;
        const float score = from_orderable((uint32_t)(key >> 32));
        // the 64-byte head record as four 16-byte loads issued together (read field by field, each landmark's load -> store pair waited for
        // the one before: ten dependent memory round trips, 14 of the 50 us of a K = 1000 decode)
        const float4* r4 = reinterpret_cast<const float4*>(p.heads + ((size_t)b * HW + idx) * 16);
        const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2], q3 = r4[3];
        const float rec[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        const uint32_t yi = idx / (uint32_t)p.w;
        float xs = (float)(int)(idx - yi * (uint32_t)p.w);
        float ys = (float)(int)yi;
        if (p.use_reg) { xs = xs + rec[13]; ys = ys + rec[14]; }
        else { xs = xs + 0.5f; ys = ys + 0.5f; }
        const float hw0 = rec[1] / 2.0f, hw1 = rec[2] / 2.0f;
        float bx1 = xs - hw0, by1 = ys - hw1, bx2 = xs + hw0, by2 = ys + hw1;
        if (p.trans) {
            // ctdet_post_process (utils/post_process.py:83-90): transform_preds on both corners with the
            // inverse affine of get_affine_transform (utils/image.py:19-66); float64 like np.dot(t, pt)
            const double* tr = p.trans + (size_t)b * 6;
            const double ax1 = tr[0] * (double)bx1 + tr[1] * (double)by1 + tr[2], ay1 = tr[3] * (double)bx1 + tr[4] * (double)by1 + tr[5];
            const double ax2 = tr[0] * (double)bx2 + tr[1] * (double)by2 + tr[2], ay2 = tr[3] * (double)bx2 + tr[4] * (double)by2 + tr[5];
            bx1 = (float)ax1; by1 = (float)ay1; bx2 = (float)ax2; by2 = (float)ay2;
        }
        const size_t o = (size_t)b * K + t;
        if (p.dets) {                                   // rows of 24 / 40 bytes: 8-byte aligned
            float2* d = reinterpret_cast<float2*>(p.dets + o * 6);
            d[0] = make_float2(bx1, by1); d[1] = make_float2(bx2, by2); d[2] = make_float2(score, 0.0f);
        }
        if (p.lms) {
            float2* l = reinterpret_cast<float2*>(p.lms + o * 10);
#pragma unroll
            for (int j = 0; j < 5; ++j) l[j] = make_float2(rec[3 + 2 * j], rec[4 + 2 * j]);
        }
        if (p.inds) p.inds[o] = (long long)idx;
        if (p.rec16) {                                     // the gather record: box, score, class, landmarks
            float4* r = reinterpret_cast<float4*>(p.rec16 + o * 16);       // 64-byte records
            r[0] = make_float4(bx1, by1, bx2, by2); r[1] = make_float4(score, 0.0f, rec[3], rec[4]);
            r[2] = make_float4(rec[5], rec[6], rec[7], rec[8]); r[3] = make_float4(rec[9], rec[10], rec[11], rec[12]);
        }
    }
#ifdef CF_TOPK_TIMING
    CF_TT();
    if (tid == 0 && b == 0) printf("topk L=%d M=%u nz=%u | init+stage %llu hist0 %llu select %llu compact %llu sort %llu zeros %llu gather %llu (memtime ticks)\n", L, M, nz,
                                   tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], tm[5] - tm[4], tm[6] - tm[5], tm[7] - tm[6]);
#endif
#undef CF_TT
}

size_t topk_big_stride(int K) {                           // u64 per image of TopkParams::big for K > 1024
    size_t n = 64;
    while (n < (size_t)K) n <<= 1;
    return n + (size_t)K;
}

hipError_t launch_peak_topk(hipStream_t s, const TopkParams& p) {
    if (p.B <= 0) return hipSuccess;
    const long long HW = (long long)p.h * p.w;
    if (p.K < 1 || p.K > HW || HW >= (1ll << 31) || !p.scratch || !p.count) return hipErrorInvalidValue;
    if (p.K > 1024 && (!p.big || p.big_stride < topk_big_stride(p.K))) return hipErrorInvalidValue;
    set_kernel_tag("cf::peak_collect_kernel(cf::TopkParams)");
    hipLaunchKernelGGL(peak_collect_kernel, dim3((unsigned)((HW + kCollectCells - 1) / kCollectCells), p.B), dim3(256), 0, s, p);
    if (p.K <= 256) {
        set_kernel_tag("void cf::topk_select_kernel<256, false>(cf::TopkParams)");
        hipLaunchKernelGGL((topk_select_kernel<256, false>), dim3(p.B), dim3(256), 0, s, p);
    } else if (p.K <= 1024) {
        set_kernel_tag("void cf::topk_select_kernel<1024, false>(cf::TopkParams)");
        hipLaunchKernelGGL((topk_select_kernel<1024, false>), dim3(p.B), dim3(1024), 0, s, p);
    } else {
        set_kernel_tag("void cf::topk_select_kernel<1024, true>(cf::TopkParams)");
        hipLaunchKernelGGL((topk_select_kernel<1024, true>), dim3(p.B), dim3(1024), 0, s, p);
    }
    return hipGetLastError();
}

// ================================================================== D1: threshold decode + NMS
// Stage 1: one workgroup per image collects the cells above the threshold IN ROW-MAJOR ORDER (the reference's
// np.where order, centerface.py:78; it decides ties in the NMS order) with their boxes and landmarks.  Each of the 16
// waves owns a contiguous segment of the map: pass 1 counts its hits (ballot + popcount per 64 cells, no barrier), one
// barrier publishes the 16 counts, pass 2 re-scans the segment and writes every hit at segment base + running offset.
// (The previous version walked the map in 1024-cell rounds with three workgroup barriers each: 74 us for one 160x160 map.)
__device__ __forceinline__ void thresh_emit(const ThreshParams& p, const float* heads, float* cand, int i, uint32_t pos) {
    const float4* r4 = reinterpret_cast<const float4*>(heads + (size_t)i * 16);      // the 64-byte record as four loads in flight together
    const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2], q3 = r4[3];
    const float rec[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    const float s = rec[0];
    const int cy = i / p.w, cx = i - cy * p.w;
    // centerface.py:84-91 -- float32 sizes, float64 centre arithmetic, cast at the end
    const float s0 = rec[1] * 4.0f, s1 = rec[2] * 4.0f;
    // D2 (eval_widerface.py:102-104) adds the offsets -- channel 1 to x, channel 0 to y, as the
    // reference does -- in float64 (int64 + float32 promotes to float64 in numpy)
    const double ox = p.mode == 1 ? (double)rec[14] : 0.0, oy = p.mode == 1 ? (double)rec[13] : 0.0;
    double x1 = fmax(0.0, ((double)cx + ox + 0.5) * 4.0 - (double)(s0 / 2.0f));
    double y1 = fmax(0.0, ((double)cy + oy + 0.5) * 4.0 - (double)(s1 / 2.0f));
    x1 = fmin(x1, (double)p.img_w); y1 = fmin(y1, (double)p.img_h);
    const double x2 = fmin(x1 + (double)s0, (double)p.img_w);
    const double y2 = fmin(y1 + (double)s1, (double)p.img_h);
    float c[16];
    c[0] = (float)x1; c[1] = (float)y1; c[2] = (float)x2; c[3] = (float)y2; c[4] = s;
#pragma unroll
    for (int j = 0; j < 5; ++j) {                       // centerface.py:94-99
        c[5 + 2 * j] = (float)(((double)rec[3 + 2 * j] + (double)cx + 0.5) * 4.0);
        c[6 + 2 * j] = (float)(((double)rec[4 + 2 * j] + (double)cy + 0.5) * 4.0);
    }
    c[15] = 0.0f;
    float4* o = reinterpret_cast<float4*>(cand + (size_t)pos * 16);       // 64-byte candidate records
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = make_float4(c[4 * j], c[4 * j + 1], c[4 * j + 2], c[4 * j + 3]);
}

__global__ __launch_bounds__(1024) void thresh_collect_kernel(ThreshParams p) {
    __shared__ uint32_t wave_cnt[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = p.h * p.w;
    const float* heads = p.heads + (size_t)b * HW * 16;
    float* cand = p.cand + (size_t)b * p.cap * 16;
    int* idx = p.order + (size_t)b * p.cap;                // scratch until the rank kernel overwrites it with the sort order
    // the threshold scan reads the dense heat plane when the head kernel wrote one (256 contiguous bytes per wave load
    // instead of one float out of each of 64 records)
    const float* hm = p.hm_plane ? p.hm_plane + (size_t)b * HW : heads;
    const int hs = p.hm_plane ? 1 : 16;
    const int seg = ((HW + 15) / 16 + 63) / 64 * 64;       // cells per wave, a multiple of 64
    const int lo = wave * seg, hi = min(lo + seg, HW);
    uint32_t mine = 0;
    for (int i0 = lo; i0 < hi; i0 += 256) {                // four independent 64-cell groups per trip (loads in flight)
        bool hit[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * 64 + lane; hit[u] = i < hi && hm[(size_t)i * hs] > p.score_thresh; }   // hm > 0.3, centerface.py:77
#pragma unroll
        for (int u = 0; u < 4; ++u) mine += (uint32_t)__popcll(__ballot(hit[u]));
    }
    if (lane == 0) wave_cnt[wave] = mine;
    __syncthreads();
    uint32_t off = 0, total = 0;
    for (int w2 = 0; w2 < 16; ++w2) { if (w2 < wave) off += wave_cnt[w2]; total += wave_cnt[w2]; }
    for (int i0 = lo; i0 < hi; i0 += 256) {
        bool hit[4]; unsigned long long bal[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * 64 + lane; hit[u] = i < hi && hm[(size_t)i * hs] > p.score_thresh; }
#pragma unroll
        for (int u = 0; u < 4; ++u) bal[u] = __ballot(hit[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (hit[u]) {
                const uint32_t pos = off + (uint32_t)__popcll(bal[u] & ((1ull << lane) - 1ull));
                if (pos < (uint32_t)p.cap) idx[pos] = i0 + u * 64 + lane;          // only the cell index: 4 bytes, no divergent box code
            }
            off += (uint32_t)__popcll(bal[u]);
        }
    }
    __syncthreads();
    // pass 3: one candidate per thread, densely (the box / landmark arithmetic inside the scan ran once per 64-cell group
    // with a hit -- ~25 times per wave for a few hundred candidates -- with one or two lanes active: 40 of the 48 us)
    const int ncand = min((int)total, p.cap);
    for (int t = tid; t < ncand; t += 1024) thresh_emit(p, heads, cand, idx[t], (uint32_t)t);
    if (tid == 0) {
        uint32_t n = total;
        if (n > (uint32_t)p.cap) { atomicMax(p.overflow, (int)n); n = p.cap; }     // the host grows the workspace to the largest count and reruns
        p.cand_count[b] = (int)n;
    }
}

// Stage 2: rank candidates by (score desc, index desc) -- rank = number of candidates that precede.  The scores are
// staged in LDS in chunks (a thread's n comparisons were n dependent 64-byte-stride global loads: 41 us for n = 500).
__global__ __launch_bounds__(256) void thresh_rank_kernel(ThreshParams p) {
    constexpr int CH = 4096;
    __shared__ float sc[CH];
    const int b = blockIdx.y;
    const int n = p.cand_count[b];
    if ((int)(blockIdx.x * 256) >= n) return;                      // workgroup-uniform
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float* cand = p.cand + (size_t)b * p.cap * 16;
    const float si = i < n ? cand[(size_t)i * 16 + 4] : 0.0f;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += CH) {
        const int m = min(CH, n - j0);
        __syncthreads();
        for (int j = threadIdx.x; j < m; j += 256) sc[j] = cand[(size_t)(j0 + j) * 16 + 4];
        __syncthreads();
        if (i < n)
            for (int j = 0; j < m; ++j) {
                const float sj = sc[j];
                rank += (sj > si) || (sj == si && (j0 + j) > i);
            }
    }
    if (i < n) p.order[(size_t)b * p.cap + rank] = i;
}

// Stage 3: suppression bit matrix in sorted order: bit (r, c) set when sorted candidate r
// suppresses sorted candidate c > r  (centerface.py:134-149).  One wave per (row, 64-column word)
// work item, upper triangle only, grid-strided so the launch does not depend on the count.
__global__ __launch_bounds__(256) void thresh_mask_kernel(ThreshParams p) {
    const int b = blockIdx.y;
    const int n = p.cand_count[b];
    const int nw = (n + 63) >> 6;
    const int words = (p.cap + 63) >> 6;
    const int lane = threadIdx.x & 63;
    const float* cand = p.cand + (size_t)b * p.cap * 16;
    const int* order = p.order + (size_t)b * p.cap;
    const long long items = (long long)n * nw;
    for (long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); t < items; t += (long long)gridDim.x * 4) {
        const int r = (int)(t / nw), cw = (int)(t - (long long)r * nw);
        if (cw * 64 + 63 <= r) { if (lane == 0) p.mask[((size_t)b * p.cap + r) * words + cw] = 0ull; continue; }
        const int c = cw * 64 + lane;
        bool sup = false;
        if (c < n && c > r) {
            const float* a = cand + (size_t)order[r] * 16;
            const float* q = cand + (size_t)order[c] * 16;
            const float ia = (a[2] - a[0] + 1.0f) * (a[3] - a[1] + 1.0f);
            const float qa = (q[2] - q[0] + 1.0f) * (q[3] - q[1] + 1.0f);
            const float xx1 = fmaxf(a[0], q[0]), yy1 = fmaxf(a[1], q[1]);
            const float xx2 = fminf(a[2], q[2]), yy2 = fminf(a[3], q[3]);
            const float w = fmaxf(0.0f, xx2 - xx1 + 1.0f), h = fmaxf(0.0f, yy2 - yy1 + 1.0f);
            const float inter = w * h;
            const float ovr = inter / (ia + qa - inter);
            sup = ovr >= p.nms_thresh;
        }
        const unsigned long long bal = __ballot(sup);
        if (lane == 0) p.mask[((size_t)b * p.cap + r) * words + cw] = bal;
    }
}

// Stage 4: greedy sweep by one wave per image in blocks of 64 sorted candidates; emits kept rows in keep order.
// Inside a block the 64 x 64 diagonal part of the suppression matrix sits in registers (lane i = row i) and the
// sequential dependence is resolved with scalar bit tests + v_readlane -- no memory access in the serial loop; the
// kept rows of the block are then written by their own lanes in parallel and OR-ed into the `removed` bitmap of the
// later blocks with coalesced row loads.  (One candidate per iteration with a dependent global load each: 200 us for
// 350 boxes, longer than the network forward of a single image.)
__global__ __launch_bounds__(64) void thresh_sweep_kernel(ThreshParams p) {
    extern __shared__ unsigned long long sweep_lds[];     // removed[words] | keptbits[words] | keptbase[words] (int)
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = p.cand_count[b];
    const int words = (p.cap + 63) >> 6;
    const int nw = (n + 63) >> 6;
    unsigned long long* removed = sweep_lds;
    unsigned long long* keptbits = sweep_lds + words;
    int* keptbase = reinterpret_cast<int*>(sweep_lds + 2 * words);
    for (int w = lane; w < nw; w += 64) removed[w] = 0ull;
    const float* cand = p.cand + (size_t)b * p.cap * 16;
    const int* order = p.order + (size_t)b * p.cap;
    const unsigned long long* gmask = p.mask + (size_t)b * p.cap * words;
    // (staging the n x nw words that matter in LDS first was tried: 44 -> 55 us for 500 candidates -- the sweep is bound by
    //  its serial scalar loop, not by the row loads)
    __syncthreads();
    const unsigned long long* mask = gmask;
    const int mstride = words;
    int kept = 0;
    for (int blk = 0; blk < nw; ++blk) {
        const int r = blk * 64 + lane;
        const unsigned long long diag = r < n ? mask[(size_t)r * mstride + blk] : 0ull;
        const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
        const unsigned long long remv = removed[blk];      // the same value in every lane
        // scalar state (SGPRs): removed bits and kept bits of this block as 32-bit halves; candidate i can only suppress
        // candidates j > i, so while i < 32 both halves of its row matter, afterwards only the high one
        uint32_t rlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)remv);
        uint32_t rhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(remv >> 32));
        uint32_t klo = 0u, khi = 0u;
        const int nvalid = min(64, n - blk * 64);
        // branch-free and fully unrolled (constant lane numbers, constant shifts): ~10 scalar instructions per candidate
        // instead of two taken branches; rows past n are all-zero and are masked out of the kept bits afterwards
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const uint32_t alive = ((rlo >> i) & 1u) - 1u;                    // all ones when candidate i is still alive
            klo |= (1u << i) & alive;
            rlo |= (uint32_t)__builtin_amdgcn_readlane((int)dlo, i) & alive;
            rhi |= (uint32_t)__builtin_amdgcn_readlane((int)dhi, i) & alive;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const uint32_t alive = ((rhi >> i) & 1u) - 1u;
            khi |= (1u << i) & alive;
            rhi |= (uint32_t)__builtin_amdgcn_readlane((int)dhi, 32 + i) & alive;
        }
        if (nvalid < 64) {
            const unsigned long long vm = (1ull << nvalid) - 1ull;
            klo &= (uint32_t)vm; khi &= (uint32_t)(vm >> 32);
        }
        const unsigned long long kb = ((unsigned long long)khi << 32) | klo;
        if (lane == 0) { keptbits[blk] = kb; keptbase[blk] = kept; }
        kept += __popcll(kb);
        // the kept rows of this block suppress candidates of the later blocks
        const int rest = nw - (blk + 1);
        if (rest > 0 && rest <= 16) {
            // lane = row: every kept lane reads the remaining words of ITS row (contiguous), then one wave-wide OR per word
            const bool kl = (kb >> lane) & 1ull;
            unsigned long long v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (kl && j < rest) ? mask[(size_t)r * mstride + blk + 1 + j] : 0ull;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < rest) {                              // uniform
                    uint32_t vl = (uint32_t)v[j], vh = (uint32_t)(v[j] >> 32);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { vl |= __shfl_xor(vl, o); vh |= __shfl_xor(vh, o); }
                    if (lane == 0) removed[blk + 1 + j] |= ((unsigned long long)vh << 32) | vl;
                }
            }
        } else {
            for (int w0 = blk + 1; w0 < nw; w0 += 64) {     // many words: lane = word, one coalesced row load per kept row
                const int w = w0 + lane;
                unsigned long long acc = 0ull;
                unsigned long long k2 = kb;
                while (k2) {                                // uniform
                    const int i = __builtin_ctzll(k2);
                    k2 &= k2 - 1;
                    if (w < nw) acc |= mask[(size_t)(blk * 64 + i) * mstride + w];
                }
                if (w < nw) removed[w] |= acc;
            }
        }
        __syncthreads();
    }
    // emit the kept rows in keep order: every candidate's lane knows its position from the block's kept bits
    for (int blk = 0; blk < nw; ++blk) {
        const int r = blk * 64 + lane;
        const unsigned long long kb = keptbits[blk];
        if (!((kb >> lane) & 1ull)) continue;
        const int pos = keptbase[blk] + __popcll(kb & ((1ull << lane) - 1ull));
        if (pos >= p.max_out) continue;
        const float4* c4 = reinterpret_cast<const float4*>(cand + (size_t)order[r] * 16);      // the 64-byte candidate record: four loads in flight
        const float4 q0 = c4[0], q1 = c4[1], q2 = c4[2], q3 = c4[3];
        const float c[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        float* d = p.dets + ((size_t)b * p.max_out + pos) * 5;
        float* l = p.lms ? p.lms + ((size_t)b * p.max_out + pos) * 10 : nullptr;
        if (p.rs_w > 0.f) {                        // centerface.py:55-62: x // scale_w, y // scale_h (exact floor of the quotient)
            const double sw = (double)p.rs_w, sh = (double)p.rs_h;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = (float)floor((double)c[j] / ((j & 1) ? sh : sw));
            d[4] = c[4];
            if (l) {
#pragma unroll
                for (int j = 0; j < 10; ++j) l[j] = (float)floor((double)c[5 + j] / ((j & 1) ? sh : sw));
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) d[j] = c[j];
        if (l) {
#pragma unroll
            for (int j = 0; j < 10; ++j) l[j] = c[5 + j];
        }
    }
    if (lane == 0) p.counts[b] = kept;            // may exceed max_out: rows past max_out are not written, the caller sees the truncation
    if (lane == 0 && p.host_counts) p.host_counts[b] = kept;
    if (lane == 0 && b == 0 && p.host_overflow) p.host_overflow[0] = *p.overflow;      // final: thresh_collect_kernel finished launches ago
}

__global__ void affine_boxes_kernel(float* dets, const double* trans, int B, int K, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const double* t = trans + (size_t)(i / K) * 6;
    float* d = dets + (size_t)i * stride;
    const double x1 = d[0], y1 = d[1], x2 = d[2], y2 = d[3];
    d[0] = (float)(t[0] * x1 + t[1] * y1 + t[2]); d[1] = (float)(t[3] * x1 + t[4] * y1 + t[5]);
    d[2] = (float)(t[0] * x2 + t[1] * y2 + t[2]); d[3] = (float)(t[3] * x2 + t[4] * y2 + t[5]);
}
hipError_t launch_affine_boxes(hipStream_t s, float* dets, const double* trans, int B, int K, int stride) {
    if (B * K <= 0) return hipSuccess;
    hipLaunchKernelGGL(affine_boxes_kernel, dim3((B * K + 255) / 256), dim3(256), 0, s, dets, trans, B, K, stride);
    return hipGetLastError();
}

// ---------------------------------------------------------------- box match (eval_widerface.py:48-74, 172-211)
// bbox_overlap's "+1" IoU between the detections of an image and its annotations, in the arithmetic numpy gives float32
// inputs (every operation rounded to float32, no contraction; the quotient is a float32 division under NumPy >= 2, where
// the python-float union is a weak scalar), and the two counts `evaluate` derives from it: rows (detections) whose best
// overlap exceeds the threshold, columns (annotations) whose best overlap does.  One workgroup per image.
__device__ __forceinline__ float overlap_pair(const float* b, const float* q, float qarea) {
    const float iw = (fminf(b[2], q[2]) - fmaxf(b[0], q[0])) + 1.0f;
    if (!(iw > 0.0f)) return 0.0f;
    const float ih = (fminf(b[3], q[3]) - fmaxf(b[1], q[1])) + 1.0f;
    if (!(ih > 0.0f)) return 0.0f;
    const float ua = ((((b[2] - b[0]) + 1.0f) * ((b[3] - b[1]) + 1.0f)) + qarea) - iw * ih;
    return (iw * ih) / ua;
}
__global__ __launch_bounds__(256) void box_match_kernel(OverlapParams p) {
    const int img = blockIdx.x;
    const int n0 = p.box_off[img], n1 = p.box_off[img + 1], k0 = p.query_off[img], k1 = p.query_off[img + 1];
    const int N = n1 - n0, K = k1 - k0;
    const float* B = p.boxes + (size_t)n0 * p.box_stride;
    const float* Q = p.query + (size_t)k0 * p.query_stride;
    int rows = 0, cols = 0;
    for (int n = threadIdx.x; n < N; n += 256) {             // a detection against every annotation
        const float* b = B + (size_t)n * p.box_stride;
        float best = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float* q = Q + (size_t)k * p.query_stride;
            const float qa = ((q[2] - q[0]) + 1.0f) * ((q[3] - q[1]) + 1.0f);
            const float ov = overlap_pair(b, q, qa);
            if (p.overlaps) p.overlaps[p.overlaps_off[img] + (size_t)n * K + k] = (double)ov;
            best = fmaxf(best, ov);
        }
        rows += (K > 0 && best > p.thresh) ? 1 : 0;
    }
    if (p.counts) {
        for (int k = threadIdx.x; k < K; k += 256) {         // an annotation against every detection
            const float* q = Q + (size_t)k * p.query_stride;
            const float qa = ((q[2] - q[0]) + 1.0f) * ((q[3] - q[1]) + 1.0f);
            float best = 0.0f;
            for (int n = 0; n < N; ++n) best = fmaxf(best, overlap_pair(B + (size_t)n * p.box_stride, q, qa));
            cols += (N > 0 && best > p.thresh) ? 1 : 0;
        }
        if (rows) atomicAdd(&p.counts[2 * img], rows);
        if (cols) atomicAdd(&p.counts[2 * img + 1], cols);
    }
}
hipError_t launch_box_match(hipStream_t s, const OverlapParams& p) {
    if (p.n_img <= 0) return hipSuccess;
    hipLaunchKernelGGL(box_match_kernel, dim3(p.n_img), dim3(256), 0, s, p);
    return hipGetLastError();
}

// dynamic LDS of the sweep: removed / kept bitmaps and kept bases
static size_t sweep_lds_bytes(int words) { return ((size_t)2 * words + (words + 1) / 2) * sizeof(unsigned long long); }
static hipError_t sweep_configure() {                      // > 64 KB of dynamic LDS needs the function attribute, once per device
    static thread_local bool configured_dev[32] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    if (configured_dev[dev & 31]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(thresh_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) configured_dev[dev & 31] = true;
    return e;
}

hipError_t launch_nms_stages(hipStream_t s, const ThreshParams& p) {
    if (p.B <= 0) return hipSuccess;
    const int words = (p.cap + 63) >> 6;
    { hipError_t e = sweep_configure(); if (e != hipSuccess) return e; }
    if (sweep_lds_bytes(words) > 160 * 1024) return hipErrorInvalidValue;   // > ~520 k candidates per image
    hipLaunchKernelGGL(thresh_rank_kernel, dim3((p.cap + 255) / 256, p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(thresh_mask_kernel, dim3(128, p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(thresh_sweep_kernel, dim3(p.B), dim3(64), sweep_lds_bytes(words), s, p);
    return hipGetLastError();
}

hipError_t launch_decode_threshold(hipStream_t s, const ThreshParams& p) {
    if (p.B <= 0) return hipSuccess;
    const int words = (p.cap + 63) >> 6;
    { hipError_t e = sweep_configure(); if (e != hipSuccess) return e; }
    if (sweep_lds_bytes(words) > 160 * 1024) return hipErrorInvalidValue;   // > ~520 k candidates per image
    hipLaunchKernelGGL(thresh_collect_kernel, dim3(p.B), dim3(1024), 0, s, p);
    hipLaunchKernelGGL(thresh_rank_kernel, dim3((p.cap + 255) / 256, p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(thresh_mask_kernel, dim3(128, p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(thresh_sweep_kernel, dim3(p.B), dim3(64), sweep_lds_bytes(words), s, p);
    return hipGetLastError();
}

}  // namespace cf
