// Host-callable launchers for the gfx950 kernels + weight packers.  dtype: 0 = fp32, 1 = bf16, 2 = fp32 storage with split-bf16
// GEMM products (sp32_t, cf_common.h): storage, layouts and sizes of 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct sp32_t;

namespace cf {

// Demangled symbol of the kernel the last launch_* call on this thread selected, in the form
// rocprofv3 prints it -- lets cf_profile_forward attribute times to kernel symbols.
const char* last_kernel_tag();
void set_kernel_tag(const char* fmt, ...);
template <typename T> inline const char* type_tag() { return sizeof(T) == 4 ? "float" : "unsigned short"; }
template <> inline const char* type_tag<sp32_t>() { return "sp32_t"; }

inline size_t elem_size(int dtype) { return dtype == 1 ? 2 : 4; }
inline int per16(int dtype) { return dtype == 1 ? 8 : 4; }

// ------------------------------------------------------------------ pointwise (1x1) conv, MFMA
// y[m][n] = act( sum_k x[m][k] * w[n][k] + bias[n] ) (+ residual[m][n]) (+ IDAUp up-branch)
struct PwParams {
    const void* x;        // [M][K]  T
    const void* wp;       // packed weights, see pw_pack_weights
    const float* bias;    // [N] fp32 or nullptr
    const void* res;      // [M][N] T residual (added after act) or nullptr
    void* y;              // [M][N] T
    long long M;
    int K, N;
    int act;              // 0 none, 1 swish, 2 relu
    // IDAUp fusion (model/centernet.py:200-204): y += relu(low[b][y/2][x/2][n] * upw[tap][n] + upb[n])
    const void* low;      // [B][Ho/2][Wo/2][N] T or nullptr
    const float* upw;     // [4][N]  deconv tap * BN scale
    const float* upb;     // [N]     BN shift
    int Ho, Wo;           // spatial dims of y per image (IDAUp fusion; the map size also picks pw_ksplit_kernel, cf_pw.hip)
    // channel-addressed output (ShuffleV2 concat, model/blocks.py:47-54): y rows have ldy elements and this conv
    // writes channels [yoff, yoff + N) of them; ldy = 0 means a dense [M][N] output.  Plain pw_kernel only.
    int ldy, yoff;
    // x in PIXEL-BLOCK order (bf16 only): [m / 32][K / 8][m % 32][8 channels], what expdw_px_kernel writes with
    // MbParams::yblock -- the 32 lanes of a wave half then read one contiguous 512-byte run per k-step instead of 32 rows
    int xblock;
    int yblock;           // y written in pixel-block order [m / 32][N / 8][m % 32][8] (bf16; dense output only)
    int resblock;         // res read in pixel-block order
};
// out[m][c] = x[m][2 c + phase], c < C: the pass-through half of channel_shuffle (model/blocks.py:56-62) written
// straight into its slice of the block output (rows of ldy elements, channel offset yoff)
hipError_t launch_shuffle_copy(hipStream_t s, int dtype, const void* x, void* y, long long M, int C, int phase, int ldy, int yoff);
size_t pw_packed_bytes(int dtype, int K, int N);
void pw_pack_weights(int dtype, const float* w /*[N][K]*/, int K, int N, void* out_host);
hipError_t launch_pw(hipStream_t s, int dtype, const PwParams& p);

// ------------------------------------------------------------------ depthwise k x k conv
struct DwParams {
    const void* x;        // [B][H][W][C] T
    const float* w;       // [k][k][C] fp32 (tap-major, channel contiguous)
    const float* bias;    // [C] fp32 or nullptr
    void* y;              // [B][Ho][Wo][C] T
    int B, C, H, W, Ho, Wo;
    int k, s, pad_lo;     // pad_hi is implied by Ho/Wo
    int act;              // 0 none, 1 swish
};
void dw_pack_weights(const float* w /*[C][1][k][k]*/, int C, int k, float* out_host /*[k][k][C]*/);
hipError_t launch_dw(hipStream_t s, int dtype, const DwParams& p);

// ------------------------------------------------------------------ fused MBConv block
struct MbGeom {
    bool ok;              // false: this block shape is not supported by the fused kernel
    int HC, nq;           // hidden-channel chunk and number of chunks (hid = HC * nq)
    int NBE, JX, HALF, NBO, rowb;
    size_t lds_bytes, wexp_bytes, wdw_floats, wproj_bytes;
    int kind, S;          // kind 0: cf_mbconv.hip, 1: cf_mbconv2.hip (fp16 pixel-pair tile, bf16 storage only)
    int KG;               // k-groups of the project loop (fp32 / split kernels: the wave groups that split a hidden chunk's k-steps);
                          // the split mode packs the project fragments in pairs per k-group (split_pairs_inplace)
};
MbGeom mb_geometry(int dtype, int Cin, int hid, int Cout, int k, int s);
void mb_pack_weights(int dtype, const MbGeom& g, int Cin, int hid, int Cout, int k,
                     const float* we /*[hid][Cin]*/, const float* wd /*[hid][k*k]*/, const float* wp /*[Cout][hid]*/,
                     void* wexp_host, float* wdw_host, void* wproj_host);
struct MbParams {
    const void* x;        // [B][Hin][Win][Cin] T
    void* y;              // [B][Hout][Wout][Cout] T
    const void* wexp; const float* wdw; const void* wproj;
    int B, Hin, Win, Hout, Wout, Cin, hid, Cout;
    int k, s, pad_lo, residual;
    int HC, nq, NBE, JX, HALF, rowb;
    size_t lds_bytes;
    int nw;               // mbconv_px_kernel: 1 = XCD-aware tile order (set by the launcher)
    int kind;             // MbGeom::kind
    int yblock;           // y (expdw: the depthwise tensor; mbconv_px: the block output) in pixel-block order
                          // [m / 32][C / 8][m % 32][8], m = linear pixel index over the batch (PwParams::xblock)
    int xblock;           // expdw_px_kernel: x in pixel-block order
    void* dbg;            // -DCF_X5_TIMING builds only: per-wave phase cycle sums (tools/x5_timing.py); nullptr otherwise
};
hipError_t launch_mbconv(hipStream_t s, int dtype, const MbParams& p);
// cf_mbconv2.hip
bool mb2_geometry(MbGeom& g, int Cin, int hid, int Cout, int k, int s);
void mb2_pack_weights(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
                      void* wexp_host, float* wdw_host, void* wproj_host);
hipError_t mb2_launch(hipStream_t s, const MbParams& p);
// expand + depthwise only (MbGeom::kind == 2): y = depthwise output [B][Hout][Wout][hid]; packs with mb_pack_weights(wproj = nullptr)
MbGeom expdw_geometry(int dtype, int Cin, int hid, int k, int s);
hipError_t expdw_launch(hipStream_t s, const MbParams& p);

// cf_mbconv4.hip: the fp32 parity mode's fused block, second generation (MbGeom::kind = 7): wave = 64 pixels, SGPR taps, permlane32
// swap into the project MFMA; expand fragments as cf_mbconv.hip, taps and project fragments repacked by mb4_repack
bool mb4_geometry(int dtype, MbGeom& g, int Cin, int hid, int Cout, int k, int s);
void mb4_repack(int dtype, const MbGeom& g, int hid, int Cout, int k, const float* wd, const float* wp, float* wdw_host, void* wproj_host);
hipError_t mb4_launch(hipStream_t s, int dtype, const MbParams& p);

// cf_mbconv5.hip: expand + depthwise for the fp32-storage modes (MbGeom::kind = 8, split-bf16 tolerance mode): register-window
// depthwise on an x-quad-cell tile, packed taps; y = depthwise output in NHWC or pixel-block order; MbGeom::HALF = hidden chunks
// per workgroup.  Expand fragments as cf_mbconv.hip, taps [chunk][group of 4 channels][tap][4] (mb_pack_weights)
MbGeom expdw_f32_geometry(int dtype, int Cin, int hid, int k, int s);
hipError_t expdw_f32_launch(hipStream_t s, int dtype, const MbParams& p);

// cf_mbconv6.hip: the split mode's fused block for Cout <= 32 (MbGeom::kind = 9): cf_mbconv5.hip's register-window depthwise + project
// MFMAs from an LDS tile of the depthwise output
bool mb6_geometry(int dtype, MbGeom& g, int Cin, int hid, int Cout, int k, int s);
void mb6_pack(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
              void* wexp_host, float* wdw_host, void* wproj_host);
hipError_t mb6_launch(hipStream_t s, const MbParams& p);

// experiments/cf_mbconv7.hip (experiments build only): as cf_mbconv6.hip, the depthwise feeding the project MFMAs directly (MbGeom::kind = 10; experiments switch CF_M7)
bool mb7_geometry(int dtype, MbGeom& g, int Cin, int hid, int Cout, int k, int s);
void mb7_pack(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
              void* wexp_host, float* wdw_host, void* wproj_host);
hipError_t mb7_launch(hipStream_t s, const MbParams& p);

// cf_mbconv3.hip: depthwise on the matrix cores (v_mfma_f32_4x4x4_16b_f16, Toeplitz operands), stride 1, bf16 storage.
// MbGeom::kind 4 = expand + depthwise (project stays a GEMM launch)
MbGeom expdw_mx_geometry(int dtype, int Cin, int hid, int k, int s);
void mx_pack_weights(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
                     void* wexp_host, float* wdw_host, void* wproj_host);
hipError_t mx_launch(hipStream_t s, const MbParams& p);
// MbGeom::kind 5 = fully fused block (expand -> matrix-core depthwise -> project (+residual)), MbGeom::HALF = 1: last round of 16
bool mx_fused_geometry(MbGeom& g, int Cin, int hid, int Cout, int k, int s);
void mx_fused_pack_weights(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
                           void* wexp_host, float* wdw_host, void* wproj_host);
hipError_t mx_fused_launch(hipStream_t s, const MbParams& p);
// MbGeom::kind 6 = the same for the stride-2 blocks (two sets per tile, waves = (set, channel half))
bool mx_fused2_geometry(MbGeom& g, int Cin, int hid, int Cout, int k, int s);
void mx_fused2_pack_weights(const MbGeom& g, int Cin, int hid, int Cout, int k, const float* we, const float* wd, const float* wp,
                            void* wexp_host, float* wdw_host, void* wproj_host);
hipError_t mx_fused2_launch(hipStream_t s, const MbParams& p);

// ------------------------------------------------------------------ stem 3x3 s2 3->32 + Swish
struct StemParams {
    const void* x;        // u8 [B][H][W][3] (BGR) or f32 [B][3][H][W]
    int in_format;        // CF_IN_U8_HWC_BGR / CF_IN_F32_NCHW
    const void* w;        // packed MFMA fragments, see stem_pack_weights
    void* y;              // [B][H/2][W/2][32] T
    int B, H, W;
};
size_t stem_packed_bytes(int dtype);
void stem_pack_weights(int dtype, const float* w /*[32][3][3][3]*/, void* out_host);
hipError_t launch_stem(hipStream_t s, int dtype, const StemParams& p);

// ------------------------------------------------------------------ fused stem + layer0 (dw 3x3 + project 32->16)
struct Stem0Params {
    const void* x;        // u8 [B][H][W][3] (BGR) or f32 [B][3][H][W]
    int in_format;
    const float* lut;     // [3][256] normalisation table (u8 input only)
    const void* wstem;    // stem_pack_weights
    const float* wdw;     // [9][32] fp32
    const void* wproj;    // stem0_pack_proj
    void* y;              // [B][H/2][W/2][16] T
    int B, H, W;
    int kind;             // 0: stem0_kernel; 1: stem0_px_kernel (bf16 storage; weights from stem0px_pack)
};
size_t stem0px_wstem_bytes();
size_t stem0px_wdw_dwords();
void stem0px_pack(const float* ws /*[32][3][3][3]*/, const float* wd /*[32][9]*/, const float* wp /*[16][32]*/,
                  void* wstem_out, uint32_t* wdw_out, void* wproj_out);
size_t stem0mx_wdw_dwords();
void stem0mx_pack(const float* ws, const float* wd, const float* wp, void* wstem_out, uint32_t* wdw_out, void* wproj_out);     // Stem0Params::kind bit 2
void stem0_lut(float* lut /*[3][256]*/);
size_t stem0_proj_bytes(int dtype);
void stem0_pack_proj(int dtype, const float* wp /*[16][32]*/, void* out_host);
hipError_t launch_stem0(hipStream_t s, int dtype, const Stem0Params& p);

// ------------------------------------------------------------------ heads: 3x3 conv (MFMA) + 1x1
struct HeadParams {
    const void* x;        // [B][h][w][24] T
    const void* w0p;      // packed 3x3 weights (96 or 16 output slots), see head_pack_weights
    const float* b0;      // [96] (two-stage) or [16] (collapsed)
    const float* w1d;     // [96][16] dense second-stage table (two-stage only)
    const float* b1;      // [16]
    float* heads;         // [B][h][w][16] fp32: hm_sigmoid, wh0, wh1, lm0..9, reg0, reg1, hm_raw
    float* hm_plane;      // optional dense [B][h][w] copy of hm_sigmoid for the peak test (coalesced reads)
    int B, h, w;
    int collapsed;
};
size_t head_packed_bytes(int dtype, int collapsed);
// w0 [4][24][24][3][3], b0 [4][24], w1 [15][24] (rows: hm, wh0, wh1, lm0..9, reg0, reg1), b1 [15]
void head_pack_weights(int dtype, int collapsed, const float* w0, const float* b0, const float* w1,
                       const float* b1, void* w0p_host, float* b0_host /*[96]|[16]*/,
                       float* w1d_host /*[96][16]*/, float* b1_host /*[16]*/);
hipError_t launch_heads(hipStream_t s, int dtype, const HeadParams& p);

// ------------------------------------------------------------------ detection loss + target encoder (cf_loss.hip)
struct LossParams {
    const float* heads;   // [B][h][w][16] head records of a forward (used when the explicit maps are null)
    const float* hm_raw;  // explicit NCHW maps (op-level entry): hm logits [B,1,h,w], wh [B,2,h,w], reg [B,2,h,w], lm [B,10,h,w]
    const float* wh; const float* reg; const float* lm;
    const float* gt_hm;   // [B][h][w]
    const unsigned char* reg_mask; const long long* ind; const float* wh_t; const float* reg_t;   // [B][M](,2)
    const unsigned char* lm_mask; const long long* lm_ind; const float* lm_t;                     // [B][M](,10)
    int B, h, w, M;
    float hm_w, wh_w, off_w, lm_w;
};
hipError_t launch_ctdet_loss(hipStream_t s, const LossParams& p, double* ws /*[3 * nblocks + 6]*/, int nblocks, float* out_dev /*[5]*/);
struct EncodeParams {
    const float* boxes;   // [B][M][4] x1,y1,x2,y2 in output-map coordinates
    const float* lms;     // [B][M][10], lms[.][0] < 0: no landmarks
    const int* counts;    // [B] objects per image (<= M)
    float* hm; float* wh; float* reg; long long* ind; unsigned char* reg_mask;
    float* landmarks; long long* lm_ind; unsigned char* lm_mask;
    int B, h, w, M;
};
hipError_t launch_encode_targets(hipStream_t s, const EncodeParams& p);

// ------------------------------------------------------------------ IDAUp stage 3 + heads fused (bf16, collapsed heads)
struct UpHeadParams {
    const void* skip;     // [B][h][w][24] T: the IDAUp skip input (layer1 output)
    const void* low;      // [B][h/2][w/2][24] T: previous IDAUp stage
    const void* wcv;      // pw_pack_weights(24 -> 24, BN folded)
    const float* bias;    // [24] BN shift of the 1x1 conv
    const float* upw;     // [4][24] deconv tap * BN scale
    const float* upb;     // [24]
    const void* w0p;      // head_pack_weights(collapsed)
    const float* b0;      // [16]
    float* heads;         // [B][h][w][16] fp32
    float* hm_plane;      // [B][h][w] or nullptr
    int B, h, w;
    int xcd;              // 1 = XCD-aware tile order (set by the launcher)
};
hipError_t launch_uphead(hipStream_t s, int dtype, const UpHeadParams& p);

// conv_last + up1 + up2 as one kernel (cf_neck.hip, bf16): the 1/32 and 1/16 neck maps exist only in LDS
struct NeckParams {
    const void* x;        // layer6 output [B][h][w][320] bf16 (rows, or pixel-block order with x_blk)
    const void* skip1;    // layer4 output [B][2h][2w][96]
    const void* skip2;    // layer2 output [B][4h][4w][32]
    int x_blk, skip1_blk, skip2_blk;
    const void* w0; const float* b0;                                      // conv_last: pw_pack_weights(320 -> 24, BN folded), shift
    const void* w1; const float* b1; const float* upw1; const float* upb1; // up1: conv (96 -> 24), [4][24] tap * scale, [24] shift
    const void* w2; const float* b2; const float* upw2; const float* upb2; // up2: conv (32 -> 24)
    void* y;              // up2 output [B][4h][4w][24] bf16 rows
    int B, h, w;          // h, w: the 1/32 map
};
hipError_t launch_neck(hipStream_t s, int dtype, const NeckParams& p);

// ------------------------------------------------------------------ decode
// D3: 3x3 peak test + top-K (radix select + bitonic sort) + gather: a multi-workgroup collect kernel + one select
// workgroup per image (cf_decode.hip).
struct TopkParams {
    const float* heads;   // [B][h*w][16]
    const float* hm_plane; // optional dense [B][h*w] heat map (else channel 0 of the records is used)
    unsigned long long* scratch;   // [B][h*w] composite keys of the cells whose kept score is not +0 (the peak list)
    int* count;           // [B * kTopkCountStride] list lengths, one per 128-byte line: zero before the launch, zero again after it
    unsigned long long* big;       // K > 1024 only: [B][big_stride] sort buffer + final order (topk_big_stride(K))
    size_t big_stride;
    int B, h, w, K, use_reg;
    float* dets;          // [B][K][6] or nullptr
    float* lms;           // [B][K][10] or nullptr
    long long* inds;      // [B][K] or nullptr
    float* rec16;         // [B][K][16] or nullptr: x1,y1,x2,y2,score,cls,lm0..9 -- the record the multi-GPU gather ships
    const double* trans;  // optional [B][6]: row-major 2x3 affine (heat-map -> source image) applied to both box corners
};
constexpr int kTopkCountStride = 32;   // ints between two images' list counters (each on its own cache line)
size_t topk_big_stride(int K);
hipError_t launch_peak_topk(hipStream_t s, const TopkParams& p);

// D1: threshold compaction (row-major) + box/landmark arithmetic + greedy NMS
struct ThreshParams {
    const float* heads;   // [B][h*w][16]
    const float* hm_plane; // optional dense [B][h*w] copy of channel 0 (coalesced threshold scan), else nullptr
    int B, h, w, img_h, img_w;
    float score_thresh, nms_thresh;
    int cap;              // candidate capacity per image
    int mode;             // 0 = D1 CenterFace.decode (centerface.py:73-109), 1 = D2 eval_widerface.decode (:92-110)
    // workspace (device)
    float* cand;          // [B][cap][16]: x1,y1,x2,y2,score, lm[10], pad
    int* cand_count;      // [B]
    int* order;           // [B][cap] candidate indices sorted by score desc
    unsigned long long* mask;   // [B][cap][cap/64]
    // outputs (device)
    int max_out;
    float* dets;          // [B][max_out][5]
    float* lms;           // [B][max_out][10] or nullptr
    int* counts;          // [B]
    int* overflow;        // [1] largest candidate count seen when some image exceeded cap (else untouched)
    float rs_h, rs_w;     // > 0: emit floor(y / rs_h), floor(x / rs_w) (centerface.py:55-62); 0 = network coordinates
    int* host_counts;     // optional page-locked HOST mirrors, written by the sweep kernel over PCIe ([B] counts, [1] the final overflow word):
    int* host_overflow;   // with dets / lms in page-locked memory too the host needs no copy command at all, only an event wait
};
hipError_t launch_decode_threshold(hipStream_t s, const ThreshParams& p);
// apply per-image 2x3 affines to the (x1,y1),(x2,y2) corners of dets [B][K][stride] in place (utils/post_process.py:83-90)
hipError_t launch_affine_boxes(hipStream_t s, float* dets, const double* trans, int B, int K, int stride);
// rank + suppression matrix + greedy sweep only (candidates already collected)
hipError_t launch_nms_stages(hipStream_t s, const ThreshParams& p);

// bbox_overlap + the two match counts of evaluate (eval_widerface.py:48-74, 172-211); images concatenated, one workgroup each
struct OverlapParams {
    const float* boxes;   // detections, rows of box_stride floats (x1,y1,x2,y2,...)
    const float* query;   // annotations, rows of query_stride floats
    const int* box_off;   // [n_img + 1] first row of each image in boxes
    const int* query_off; // [n_img + 1]
    int n_img, box_stride, query_stride;
    float thresh;
    double* overlaps;     // optional: per image a dense [N][K] block at overlaps_off[img]
    const long long* overlaps_off;
    int* counts;          // optional [n_img][2] (zeroed): detections with a best overlap > thresh, annotations with one
};
hipError_t launch_box_match(hipStream_t s, const OverlapParams& p);

// bilinear stretch-resize of uint8 HWC images (cv2.resize(img, (W, H)) at centerface.py:30; half-pixel centres)
hipError_t launch_resize_u8(hipStream_t s, const uint8_t* src, uint8_t* dst, int B, int h, int w, int H, int W);
// B page-locked host images (device-visible addresses, 16-byte aligned, `bytes` each) -> dst [B][bytes], read over PCIe by a kernel
hipError_t launch_upload_images(hipStream_t s, const void* const* imgs, uint8_t* dst, int B, long long bytes);

// layout converters used by cf_get_heads and the per-op test entry points
hipError_t launch_nchw_to_nhwc(hipStream_t s, int dtype, const float* src /*f32 NCHW*/, void* dst /*T NHWC*/,
                               int B, int C, int H, int W);
hipError_t launch_blocked_to_nchw(hipStream_t s, int dtype, const void* src /*pixel-block order*/, float* dst /*f32 NCHW*/, int B, int C, int H, int W);
hipError_t launch_nhwc_to_nchw(hipStream_t s, int dtype, const void* src /*T NHWC*/, float* dst /*f32 NCHW*/,
                               int B, int C, int H, int W);

}  // namespace cf
