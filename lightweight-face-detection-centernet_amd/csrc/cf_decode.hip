// Peak decode on the GPU.
//
// D3 (peak_topk_kernel) replaces ctdet_decode and its helpers (centerface_ext.py:11-82):
//   _nms   (:44-50)  3x3 max-pool equality mask            -> fused peak test, no pooled tensor
//   _topk  (:11-27)  torch.topk over H*W, index -> (y, x)   -> exact radix select + bitonic sort
//   _transpose_and_gather_feat (:28-42) full NCHW->NHWC permute + gather -> K 64-byte row reads
//   box assembly (:60-82)
// One workgroup (1024 threads, 16 waves) per image.  Scores are made totally ordered by the
// composite 64-bit key (orderable(score) << 32) | ~index, so "equal scores: lower index first" is
// part of the key and the selected set is exact and deterministic (torch leaves ties unspecified).
// The K-th largest key is found by MSB-first radix select (LDS histograms, suffix scans with wave
// shuffles); the K survivors are sorted descending by a bitonic network that exchanges through
// __shfl_xor inside a wave and through LDS across waves.
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

__global__ __launch_bounds__(1024) void peak_topk_kernel(TopkParams p) {
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t res[3];
    __shared__ uint32_t nsel;
    __shared__ u64 sel[1024];

    const int b = blockIdx.x;
    const int HW = p.h * p.w;
    const int tid = threadIdx.x;
    // dense heat-map plane when the head kernel provided one (coalesced), else channel 0 of the records
    const float* hm = p.hm_plane ? p.hm_plane + (size_t)b * HW : p.heads + (size_t)b * HW * 16;
    const int hs = p.hm_plane ? 1 : 16;
    u64* keys = p.scratch + (size_t)b * HW;

    // ---- pass 0: peak test (_nms), composite keys, and the histogram of the top 11 score bits
    for (int i = tid; i < 2048; i += 1024) hist[i] = 0;
    __syncthreads();
    // four cells per thread and iteration: all 4 x 9 loads are in flight before the first compare
    for (int i0 = tid; i0 < HW; i0 += 4096) {
        float v[4], mx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 1024;
            const int ic = i < HW ? i : HW - 1;
            const int y = ic / p.w, x = ic - y * p.w;
            v[u] = hm[(size_t)ic * hs];
            mx[u] = v[u];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = min(max(y + dy, 0), p.h - 1), xx = min(max(x + dx, 0), p.w - 1);   // clamped = in-range duplicate
                    mx[u] = fmaxf(mx[u], hm[((size_t)yy * p.w + xx) * hs]);
                }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 1024;
            if (i >= HW) break;
            // heat * keep  (keep = 1.0 where hmax == heat else 0.0); "+ 0.0f" canonicalises -0 to +0
            const float kept = (mx[u] == v[u]) ? v[u] : (v[u] * 0.0f + 0.0f);
            const uint32_t ok = orderable(kept);
            keys[i] = ((u64)ok << 32) | (u64)(0xffffffffu - (uint32_t)i);
            atomicAdd(&hist[ok >> 21], 1u);
        }
    }
    __syncthreads();

    // ---- radix select of the K-th largest composite key, MSB first: 11+11+10 score bits, then
    //      17 index bits (9+8) which only matter when scores tie at the threshold
    u64 prefix = 0, mask = 0;
    uint32_t kth = (uint32_t)p.K;
    const int shifts[5] = {53, 42, 32, 8, 0};
    const int widths[5] = {11, 11, 10, 9, 8};
    // the low 32 bits hold ~index: its top 15 bits are all ones for index < 2^17, include them in
    // the prefix up front so that the two index passes cover bits [16:8] and [7:0]
    for (int pass = 0; pass < 5; ++pass) {
        if (pass == 3) { prefix |= 0xfffe0000ull; mask |= 0xfffe0000ull; }
        const int sh = shifts[pass], wd = widths[pass];
        if (pass > 0) {                                  // pass 0's histogram was built with the keys
            for (int i = tid; i < 2048; i += 1024) hist[i] = 0;
            __syncthreads();
            for (int i0 = tid; i0 < HW; i0 += 4096) {
                u64 k[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) k[u] = keys[min(i0 + u * 1024, HW - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u * 1024 < HW && (k[u] & mask) == prefix) atomicAdd(&hist[(uint32_t)(k[u] >> sh) & ((1u << wd) - 1u)], 1u);
            }
            __syncthreads();
        }
        find_digit<2048>(hist, kth, res);
        __syncthreads();
        prefix |= (u64)res[0] << sh;
        mask |= (u64)((1u << wd) - 1u) << sh;
        kth -= res[1];
        const uint32_t in_bin = res[2];
        __syncthreads();
        // all 32 score bits fixed and every key with that score is needed: no tie to break by index
        if (pass == 2 && kth == in_bin) break;
    }
    const u64 thresh = prefix;          // exact K-th largest composite key (keys are distinct)

    // ---- compaction of the K survivors (order irrelevant: sorted next)
    if (tid == 0) nsel = 0;
    __syncthreads();
    for (int i0 = tid; i0 < HW; i0 += 4096) {
        u64 k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = keys[min(i0 + u * 1024, HW - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * 1024 < HW && k[u] >= thresh) { uint32_t pos = atomicAdd(&nsel, 1u); if (pos < 1024) sel[pos] = k[u]; }
    }
    __syncthreads();
    u64 mine = (tid < p.K) ? sel[tid] : 0ull;
    __syncthreads();

    // ---- bitonic sort, descending, over the next power of two >= K elements (one per thread; the padding
    //      keys are 0 and sink to the end)
    int sortn = 64;
    while (sortn < p.K) sortn <<= 1;
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

    // ---- gather + box assembly (centerface_ext.py:60-82)
    if (tid < p.K) {
        const uint32_t idx = 0xffffffffu - (uint32_t)mine;
        const float score = from_orderable((uint32_t)(mine >> 32));
        const float* rec = p.heads + ((size_t)b * HW + idx) * 16;
        float xs = (float)(int)(idx % (uint32_t)p.w);
        float ys = (float)(int)(idx / (uint32_t)p.w);
        if (p.use_reg) { xs = xs + rec[13]; ys = ys + rec[14]; }
        else { xs = xs + 0.5f; ys = ys + 0.5f; }
        const float hw0 = rec[1] / 2.0f, hw1 = rec[2] / 2.0f;
        float* d = p.dets + ((size_t)b * p.K + tid) * 6;
        float bx1 = xs - hw0, by1 = ys - hw1, bx2 = xs + hw0, by2 = ys + hw1;
        if (p.trans) {
            // ctdet_post_process (utils/post_process.py:83-90): transform_preds on both corners with the
            // inverse affine of get_affine_transform (utils/image.py:19-66); float64 like np.dot(t, pt)
            const double* t = p.trans + (size_t)b * 6;
            const double ax1 = t[0] * (double)bx1 + t[1] * (double)by1 + t[2], ay1 = t[3] * (double)bx1 + t[4] * (double)by1 + t[5];
            const double ax2 = t[0] * (double)bx2 + t[1] * (double)by2 + t[2], ay2 = t[3] * (double)bx2 + t[4] * (double)by2 + t[5];
            bx1 = (float)ax1; by1 = (float)ay1; bx2 = (float)ax2; by2 = (float)ay2;
        }
        d[0] = bx1; d[1] = by1; d[2] = bx2; d[3] = by2; d[4] = score; d[5] = 0.0f;
        if (p.lms) {
            float* l = p.lms + ((size_t)b * p.K + tid) * 10;
#pragma unroll
            for (int j = 0; j < 10; ++j) l[j] = rec[3 + j];
        }
        if (p.inds) p.inds[(size_t)b * p.K + tid] = (long long)idx;
    }
}

hipError_t launch_peak_topk(hipStream_t s, const TopkParams& p) {
    if (p.B <= 0) return hipSuccess;
    if (p.K < 1 || p.K > 1024 || p.K > p.h * p.w || p.h * p.w > (1 << 17)) return hipErrorInvalidValue;
    set_kernel_tag("cf::peak_topk_kernel(cf::TopkParams)");
    hipLaunchKernelGGL(peak_topk_kernel, dim3(p.B), dim3(1024), 0, s, p);
    return hipGetLastError();
}

// ================================================================== D1: threshold decode + NMS
// Stage 1: one workgroup per image scans the heat map in row-major order in rounds of 1024 cells
// and appends the cells above the threshold, in order, with their boxes and landmarks.
__global__ __launch_bounds__(1024) void thresh_collect_kernel(ThreshParams p) {
    __shared__ uint32_t wave_cnt[16];
    __shared__ uint32_t base_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = p.h * p.w;
    const float* heads = p.heads + (size_t)b * HW * 16;
    float* cand = p.cand + (size_t)b * p.cap * 16;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < HW; i0 += 1024) {
        const int i = i0 + tid;
        float s = 0.0f;
        bool hit = false;
        if (i < HW) { s = heads[(size_t)i * 16]; hit = s > p.score_thresh; }     // hm > 0.3, centerface.py:77
        const unsigned long long bal = __ballot(hit);
        const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        uint32_t off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        uint32_t total = 0;
        for (int w = 0; w < 16; ++w) total += wave_cnt[w];
        if (hit) {
            const uint32_t pos = off + before;
            if (pos < (uint32_t)p.cap) {
                const float* rec = heads + (size_t)i * 16;
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
                float* c = cand + (size_t)pos * 16;
                c[0] = (float)x1; c[1] = (float)y1; c[2] = (float)x2; c[3] = (float)y2; c[4] = s;
#pragma unroll
                for (int j = 0; j < 5; ++j) {                       // centerface.py:94-99
                    c[5 + 2 * j] = (float)(((double)rec[3 + 2 * j] + (double)cx + 0.5) * 4.0);
                    c[6 + 2 * j] = (float)(((double)rec[4 + 2 * j] + (double)cy + 0.5) * 4.0);
                }
                c[15] = 0.0f;
            }
        }
        __syncthreads();
        if (tid == 0) base_s += total;
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t n = base_s;
        if (n > (uint32_t)p.cap) { atomicMax(p.overflow, (int)n); n = p.cap; }     // the host grows the workspace to the largest count and reruns
        p.cand_count[b] = (int)n;
    }
}

// Stage 2: rank candidates by (score desc, index desc) -- rank = number of candidates that precede
__global__ __launch_bounds__(256) void thresh_rank_kernel(ThreshParams p) {
    const int b = blockIdx.y;
    const int n = p.cand_count[b];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* cand = p.cand + (size_t)b * p.cap * 16;
    const float si = cand[(size_t)i * 16 + 4];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        const float sj = cand[(size_t)j * 16 + 4];
        rank += (sj > si) || (sj == si && j > i);
    }
    p.order[(size_t)b * p.cap + rank] = i;
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

// Stage 4: sequential greedy sweep by one wave per image; emits kept rows in keep order.
__global__ __launch_bounds__(64) void thresh_sweep_kernel(ThreshParams p) {
    extern __shared__ unsigned long long removed[];      // words
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = p.cand_count[b];
    const int words = (p.cap + 63) >> 6;
    const int nw = (n + 63) >> 6;
    for (int w = lane; w < nw; w += 64) removed[w] = 0ull;
    __syncthreads();
    const float* cand = p.cand + (size_t)b * p.cap * 16;
    const int* order = p.order + (size_t)b * p.cap;
    int kept = 0;
    for (int r = 0; r < n; ++r) {
        const bool dead = (removed[r >> 6] >> (r & 63)) & 1ull;      // uniform
        if (dead) continue;
        const unsigned long long* row = p.mask + ((size_t)b * p.cap + r) * words;
        for (int w = (r >> 6) + lane; w < nw; w += 64) removed[w] |= row[w];
        if (kept < p.max_out) {
            const float* c = cand + (size_t)order[r] * 16;
            if (lane < 5) p.dets[((size_t)b * p.max_out + kept) * 5 + lane] = c[lane];
            if (p.lms && lane >= 5 && lane < 15) p.lms[((size_t)b * p.max_out + kept) * 10 + (lane - 5)] = c[lane];
        }
        ++kept;
        __syncthreads();
    }
    if (lane == 0) p.counts[b] = kept;            // may exceed max_out: rows past max_out are not written, the caller sees the truncation
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

hipError_t launch_nms_stages(hipStream_t s, const ThreshParams& p) {
    if (p.B <= 0) return hipSuccess;
    const int words = (p.cap + 63) >> 6;
    hipLaunchKernelGGL(thresh_rank_kernel, dim3((p.cap + 255) / 256, p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(thresh_mask_kernel, dim3(128, p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(thresh_sweep_kernel, dim3(p.B), dim3(64), words * sizeof(unsigned long long), s, p);
    return hipGetLastError();
}

hipError_t launch_decode_threshold(hipStream_t s, const ThreshParams& p) {
    if (p.B <= 0) return hipSuccess;
    const int words = (p.cap + 63) >> 6;
    hipLaunchKernelGGL(thresh_collect_kernel, dim3(p.B), dim3(1024), 0, s, p);
    hipLaunchKernelGGL(thresh_rank_kernel, dim3((p.cap + 255) / 256, p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(thresh_mask_kernel, dim3(128, p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(thresh_sweep_kernel, dim3(p.B), dim3(64), words * sizeof(unsigned long long), s, p);
    return hipGetLastError();
}

}  // namespace cf
