// Training-side pieces that share the detector's tensors: the CenterNet
// detection loss evaluated on the head maps of a forward, and the target encoder that produces what it is
// compared with.  Forward evaluation only (validation loss / fine-tuning monitors), fp32 terms summed in
// double in a FIXED order, so results are reproducible run to run.
//
//   focal loss     model/losses.py:142-167 (_neg_loss) on clamp(sigmoid(hm), 1e-5, 1 - 1e-5) (:343-345)
//   RegL1Loss      model/losses.py:239-250 with _tranpose_and_gather_feat (:86-90): gather the head map at
//                  `ind`, masked L1 sum / (mask.sum() * C + 1e-4)
//   CtdetLoss      model/losses.py:347-374: hm_w * focal + wh_w * L1(wh) + off_w * L1(reg) + lm_w * L1(lm)
//   target maps    dataset/dataset.py:160-217 per-object loop with utils/image.py:95-141
//                  (gaussian_radius in float64, draw_umich_gaussian max-blend, ind / reg / masks / landmarks)
// Compiled with -ffp-contract=off: gaussian_radius must evaluate b*b - 4ac exactly as numpy does.
#include "cf_common.h"
#include "cf_kernels.h"

namespace cf {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum of three doubles in a fixed order: lanes (xor tree) -> waves 0..N-1 sequentially
template <int NT>
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c) {
    __shared__ double red[3][NT / 64];
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = c; }
    __syncthreads();
    a = b = c = 0.0;
    for (int w = 0; w < NT / 64; ++w) { a += red[0][w]; b += red[1][w]; c += red[2][w]; }
    __syncthreads();
}

// partial[blockIdx] = {num_pos, sum pos_loss, sum neg_loss} over this block's contiguous slice of cells
__global__ __launch_bounds__(256) void focal_partial_kernel(const float* hm_raw, int stride, const float* gt, long long n,
                                                            double* partial) {
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double npos = 0.0, pos = 0.0, neg = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const float x = hm_raw[i * stride];
        float p = 1.0f / (1.0f + expf(-x));
        p = fminf(fmaxf(p, 1e-5f), 1.0f - 1e-5f);
        const float g = gt[i];
        if (g == 1.0f) {
            const float q = 1.0f - p;
            npos += 1.0; pos += (double)(logf(p) * (q * q));
        } else if (g < 1.0f) {
            const float q = 1.0f - g, q2 = q * q;
            neg += (double)(logf(1.0f - p) * (p * p) * (q2 * q2));
        }
    }
    block_sum3<256>(npos, pos, neg);
    if (threadIdx.x == 0) { partial[blockIdx.x * 3 + 0] = npos; partial[blockIdx.x * 3 + 1] = pos; partial[blockIdx.x * 3 + 2] = neg; }
}

// One block per regression head q in {wh, reg, lm}: sums[q] = {sum |pred - target| over masked objects, mask count * C}
__global__ __launch_bounds__(256) void regl1_kernel(LossParams p, double* sums) {
    const int q = blockIdx.x;
    const int C = q == 2 ? 10 : 2;
    const int slot0 = q == 0 ? 1 : q == 1 ? 13 : 3;                       // head record slots: wh 1-2, lm 3-12, reg 13-14
    const unsigned char* mask = q == 2 ? p.lm_mask : p.reg_mask;
    const long long* ind = q == 2 ? p.lm_ind : p.ind;
    const float* tgt = q == 0 ? p.wh_t : q == 1 ? p.reg_t : p.lm_t;
    const float* map = q == 0 ? p.wh : q == 1 ? p.reg : p.lm;            // explicit NCHW maps, or nullptr -> p.heads
    const long long hw = (long long)p.h * p.w;
    double s = 0.0, cnt = 0.0, unused = 0.0;
    const int total = p.B * p.M * C;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int c = i % C, bm = i / C, b = bm / p.M;
        if (!mask[bm]) continue;
        long long cell = ind[bm];
        cell = cell < 0 ? 0 : (cell >= hw ? hw - 1 : cell);
        const float pred = map ? map[((long long)b * C + c) * hw + cell] : p.heads[((long long)b * hw + cell) * 16 + slot0 + c];
        s += (double)fabsf(pred - tgt[(long long)bm * C + c]);
        cnt += 1.0;
    }
    block_sum3<256>(s, cnt, unused);
    if (threadIdx.x == 0) { sums[q * 2 + 0] = s; sums[q * 2 + 1] = cnt; }
}

// out[5] = loss, hm_loss, wh_loss, off_loss, lm_loss (float32, as the reference's loss_stats)
__global__ void loss_final_kernel(const double* partial, int nblocks, const double* sums, LossParams p, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double npos = 0.0, pos = 0.0, neg = 0.0;
    for (int i = 0; i < nblocks; ++i) { npos += partial[i * 3]; pos += partial[i * 3 + 1]; neg += partial[i * 3 + 2]; }
    const float fpos = (float)pos, fneg = (float)neg, fn = (float)npos;
    const float focal = npos == 0.0 ? -fneg : -(fpos + fneg) / fn;
    float l[3];
    for (int q = 0; q < 3; ++q) l[q] = (float)sums[q * 2] / ((float)sums[q * 2 + 1] + 1e-4f);
    const float hm_loss = p.hm_w * focal, wh_loss = p.wh_w * l[0], off_loss = p.off_w * l[1], lm_loss = p.lm_w * l[2];
    out[0] = hm_loss + wh_loss + off_loss + lm_loss;
    out[1] = hm_loss; out[2] = wh_loss; out[3] = off_loss; out[4] = lm_loss;
}

hipError_t launch_ctdet_loss(hipStream_t s, const LossParams& p, double* ws /*[3 * nblocks + 6]*/, int nblocks, float* out_dev) {
    const long long n = (long long)p.B * p.h * p.w;
    const float* raw = p.hm_raw ? p.hm_raw : p.heads + 15;
    const int stride = p.hm_raw ? 1 : 16;
    hipLaunchKernelGGL(focal_partial_kernel, dim3(nblocks), dim3(256), 0, s, raw, stride, p.gt_hm, n, ws);
    hipLaunchKernelGGL(regl1_kernel, dim3(3), dim3(256), 0, s, p, ws + 3 * nblocks);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, s, ws, nblocks, ws + 3 * nblocks, p, out_dev);
    return hipGetLastError();
}

// ------------------------------------------------------------------ target encoder
__device__ double gaussian_radius_dev(int height, int width) {
    const double mo = 0.7;
    const double b1 = (double)(height + width);
    const double c1 = (double)(width * height) * (1 - mo) / (1 + mo);
    const double sq1 = sqrt(b1 * b1 - 4 * c1);
    const double r1 = (b1 + sq1) / 2;
    const double b2 = (double)(2 * (height + width));
    const double c2 = (1 - mo) * width * height;
    const double sq2 = sqrt(b2 * b2 - 16 * c2);
    const double r2 = (b2 + sq2) / 2;
    const double a3 = 4 * mo;
    const double b3 = -2 * mo * (height + width);
    const double c3 = (mo - 1) * width * height;
    const double sq3 = sqrt(b3 * b3 - 4 * a3 * c3);
    const double r3 = (b3 + sq3) / 2;
    return fmin(r1, fmin(r2, r3));
}

// grid (M, B), one block per object slot.  All outputs are zero-initialised by the caller.
__global__ __launch_bounds__(256) void encode_targets_kernel(EncodeParams p) {
    const int k = blockIdx.x, b = blockIdx.y;
    if (k >= p.counts[b]) return;
    const float* bb = p.boxes + ((size_t)b * p.M + k) * 4;
    const float* lm = p.lms + ((size_t)b * p.M + k) * 10;
    const float fw = (float)(p.w - 1), fh = (float)(p.h - 1);
    const float x1 = fminf(fmaxf(bb[0], 0.0f), fw), x2 = fminf(fmaxf(bb[2], 0.0f), fw);
    const float y1 = fminf(fmaxf(bb[1], 0.0f), fh), y2 = fminf(fmaxf(bb[3], 0.0f), fh);
    const float bh = y2 - y1, bw = x2 - x1;
    if (!(bh > 0.0f && bw > 0.0f)) return;
    double rad = gaussian_radius_dev((int)ceilf(bh), (int)ceilf(bw));
    const int radius = rad > 0.0 ? (int)rad : 0;
    const float cx = (x1 + x2) / 2, cy = (y1 + y2) / 2;
    const int ix = (int)cx, iy = (int)cy;
    const size_t o = (size_t)b * p.M + k;
    if (threadIdx.x == 0) {
        p.wh[o * 2] = bw; p.wh[o * 2 + 1] = bh;
        p.ind[o] = (long long)iy * p.w + ix;
        p.reg[o * 2] = (float)((double)cx - ix); p.reg[o * 2 + 1] = (float)((double)cy - iy);
        p.reg_mask[o] = 1;
        if (lm[0] > 0 && lm[1] < p.h && lm[2] < p.w && lm[3] < p.h && lm[6] > 0 && lm[7] > 0 && lm[8] < p.w && lm[9] > 0) {
            p.lm_ind[o] = (long long)iy * p.w + ix;
            if (bh * bw > 10.0f) p.lm_mask[o] = 1;
            for (int j = 0; j < 10; ++j) p.landmarks[o * 10 + j] = (float)((double)lm[j] - ((j & 1) ? iy : ix));
        }
    }
    // draw_umich_gaussian: max-blend exp(-(x^2+y^2) / (2 sigma^2)), sigma = (2r+1)/6, float64 -> float32
    const int diameter = 2 * radius + 1;
    const double sigma = diameter / 6.0;
    const int left = min(ix, radius), right = min(p.w - ix, radius + 1);
    const int top = min(iy, radius), bottom = min(p.h - iy, radius + 1);
    const int ww = left + right, hh = top + bottom;
    if (ww <= 0 || hh <= 0) return;
    int* hmi = reinterpret_cast<int*>(p.hm + (size_t)b * p.h * p.w);
    for (int t = threadIdx.x; t < ww * hh; t += 256) {
        const int dy = t / ww - top, dx = t % ww - left;
        double g = exp(-(double)(dx * dx + dy * dy) / (2 * sigma * sigma));
        if (g < 2.220446049250313e-16) g = 0.0;                       // eps * max (max = 1 at the centre)
        const float gf = (float)g;
        atomicMax(&hmi[(iy + dy) * p.w + (ix + dx)], __float_as_int(gf));   // values >= 0: int order = float order
    }
}

hipError_t launch_encode_targets(hipStream_t s, const EncodeParams& p) {
    if (p.B <= 0 || p.M <= 0) return hipSuccess;
    hipLaunchKernelGGL(encode_targets_kernel, dim3(p.M, p.B), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace cf
