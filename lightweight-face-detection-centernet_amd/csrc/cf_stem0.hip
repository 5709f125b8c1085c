// Fused network entry for gfx950: uint8 image -> normalise -> stem 3x3 s2 3->32 + Swish ->
// layer0 depthwise 3x3 + Swish -> layer0 project 32->16, in ONE kernel.
//
// Replaces centerface.py:32-37 (x/255, (x-mean)/std, HWC->CHW), first_conv (model/centernet.py:224)
// and layer0.0 = MBConvBlock(32, 16, expand_ratio=1, k=3, s=1) (model/centernet.py:213, :105-122).
// Layer by layer these three ops move 1.2 + 6.6 | 6.6 + 6.6 | 6.6 + 3.3 MB per image (bf16); fused,
// only the uint8 image (1.2 MB) is read and the 16-channel map (3.3 MB) is written: the 32-channel
// stem output lives only in LDS.
//
// One workgroup = one 8x16 tile of the H/2 x W/2 map:
//   stage   the (2*10+1) x (2*18+1) image patch behind the tile is read with coalesced byte loads,
//           mapped through the 3x256 normalisation table (computed on the host with the reference's
//           exact float32 arithmetic) and stored in LDS in the storage type; outside the image: 0
//           (the stem's own ZeroPad2d(0,1,0,1)).
//   phase 1 stem: lane (pixel, h) gathers its half of the 27 taps from LDS, D^T = Ws . X^T on MFMA
//           (as cf_stem.hip), Swish -> E[pixel][32] in LDS; pixels outside the map are written as 0
//           (the depthwise conv's ZeroPad2d(1,1,1,1) acts on the stem OUTPUT).
//   phase 2 depthwise 3x3 + Swish per lane from LDS, feeding the project MFMA directly (cf_mbconv.hip).
#include <type_traits>
#include "cf_common.h"
#include "cf_kernels.h"
#include "cf_mx.h"
#include <vector>
#include "centerface_hip.h"

namespace cf {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

static inline int slot_channel(int nb, int i) {
    int h = (i >> 2) & 1;
    int r = (i & 3) + 4 * (i >> 3);
    return nb * 32 + h * 16 + r;
}

constexpr int S0_TOH = 16, S0_TOW = 16;                 // 8 waves: one 32-pixel block (2 rows) each
constexpr int S0_NT = (S0_TOH * S0_TOW / 32) * 64;      // 512 threads
constexpr int S0_IH = S0_TOH + 2, S0_IW = S0_TOW + 2, S0_IPX = S0_IH * S0_IW;     // 18 x 18 = 324
constexpr int S0_PH = 2 * S0_IH + 1, S0_PW = 2 * S0_IW + 1;                         // 37 x 37 image patch
constexpr int S0_PROW = S0_PW * 3 + 1;                                              // 112 elements per patch row

void stem0_lut(float* lut /*[3][256]*/) {
    // centerface.py:12-15 (BGR), :32-33 -- float32 arithmetic, IEEE division, exactly as numpy
    const float mean[3] = {0.408f, 0.447f, 0.470f};
    const float stdv[3] = {0.289f, 0.274f, 0.278f};
    for (int c = 0; c < 3; ++c)
        for (int u = 0; u < 256; ++u) {
            volatile float a = (float)u / 255.0f;
            volatile float b = a - mean[c];
            lut[c * 256 + u] = b / stdv[c];
        }
}

size_t stem0_proj_bytes(int dtype) { return (size_t)(32 * elem_size(dtype) / 16 / 2) * 64 * 16; }

// project weights wp [16][32] -> [j][lane][16 B]; hidden chunk = all 32 stem channels
void stem0_pack_proj(int dtype, const float* wp, void* out_host) {
    const int P = per16(dtype), HALF = 32 * (int)elem_size(dtype) / 16 / 2;
    __builtin_memset(out_host, 0, stem0_proj_bytes(dtype));
    for (int j = 0; j < HALF; ++j)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31, h = lane >> 5, co = slot_channel(0, i);
            if (co >= 16) continue;
            char* dst = (char*)out_host + ((size_t)j * 64 + lane) * 16;
            pack_chunk(dtype, wp + co * 32 + (h * HALF + j) * P, dst);
        }
    if (dtype == 2) split_pairs_inplace(out_host, 1, HALF);
}

template <typename T> using S0Mma = CfMma<T>;

#ifdef CF_X5_TIMING      // phase stamps of stem0_kernel, summed over all waves (tools/x5_timing.py reads them through cf_debug_stem_stamps)
__device__ unsigned long long g_s0_stamps[8];
__device__ unsigned long long g_s0_buf[204800 * 8];      // one record per wave of a B = 64, 640x640 launch (no atomics: they perturb the loads being timed)
#define S0_STAMP(k) { unsigned long long t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); s0t[k] += t_ - s0q; s0q = t_; }
#else
#define S0_STAMP(k)
#endif
template <typename T, int FMT>
__global__ __launch_bounds__(S0_NT) void stem0_kernel(Stem0Params p) {
#ifdef CF_X5_TIMING
    unsigned long long s0t[6] = {0, 0, 0, 0, 0, 0}, s0q;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(s0q) :: "memory");
#endif
    constexpr int P = Elem<T>::PER16;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr bool PRE = std::is_same<T, sp32_t>::value;          // split mode: -log2(e) in the stem weights, -ln 2 in the project weights (cf_runtime.hip)
    constexpr int ROWB = 32 * sizeof(T) + 16;
    constexpr int HALF = 32 * sizeof(T) / 16 / 2;                 // k-steps of the project GEMM per lane half
    constexpr int NIB = (S0_IPX + 31) / 32;                       // 11
    constexpr int NW = S0_NT / 64;
    // split mode, uint8 input (round 5): the patch is staged with DWORD loads (below), its rows start 2 elements early and are 116 wide
    constexpr bool DWST = PRE && FMT == CF_IN_U8_HWC_BGR;
    constexpr int PROW = DWST ? 120 : S0_PROW, XOFF = DWST ? 2 : 0;
    __shared__ __attribute__((aligned(16))) char E[S0_IPX * ROWB];
    __shared__ __attribute__((aligned(16))) T Xs[S0_PH * PROW];
    __shared__ __attribute__((aligned(16))) float Wd[9 * 32];
    __shared__ float lut[FMT == CF_IN_U8_HWC_BGR ? 768 : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const int ox0 = blockIdx.x * S0_TOW, oy0 = blockIdx.y * S0_TOH, b = blockIdx.z;


    // ---- stage the normalised image patch: patch row r, element e = col*3 + ci
    const int iy0 = 2 * (oy0 - 1), ix0 = 2 * (ox0 - 1);
    // Thread t owns one element column e = col*3 + ci of the patch (fixed for the whole loop) and walks
    // down the rows with stride RSTEP: every per-element quantity except the row is loop-invariant, so
    // the loop body is a load, a compare/select, a table read and an LDS store -- no index division.
    constexpr int ECOLS = S0_PW * 3;                              // 111 elements per patch row
    constexpr int RSTEP = S0_NT / ECOLS;                          // 4 rows per pass (68 lanes idle)
    constexpr int NIT = (S0_PH + RSTEP - 1) / RSTEP;              // 10 passes over the 37 rows
    if constexpr (DWST) {
        // Round 5 (phase stamps, tools/x5_timing.py: the byte-wise staging below was 46 % of this kernel's wave time in the split mode --
        // 4107 single-byte loads per tile, a table gather and a 4-byte LDS store each).  As in stem0_px_kernel: the patch row starts
        // 6 (ox0 - 1) bytes into the image row = 2 bytes past a dword boundary, so it is read as 29 ALIGNED dwords starting 2 bytes early
        // (a dword lies entirely inside or outside the image row); thread = one dword column walking down the rows; normalisation
        // (u / 255 - mean) / std as one fma per byte (1 ulp from the reference's two divisions: the exact mode keeps its table, this mode
        // is held to 1e-3); outside the image: exact 0 (ZeroPad2d).  Element e of a patch row sits at position e + 2.
        constexpr int ND = (S0_PW * 3 + 2 + 3) / 4, RG = S0_NT / ND, NITD = (S0_PH + RG - 1) / RG;   // 29 dwords cover a patch row (+2 bytes in front)
        const int d = tid % ND, rg = tid / ND;
        const bool tact = rg < RG;
        const int boff = ix0 * 3 - 2 + 4 * d;
        const bool din = tact && boff >= 0 && boff + 4 <= p.W * 3;
        const int cboff = min(max(boff, 0), p.W * 3 - 4);
        float sc[4], sh[4]; bool bok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * d - 2 + i;
            bok[i] = din && e >= 0 && e < S0_PW * 3;
            const int ci = (e + 3) % 3;
            sc[i] = ci == 0 ? 1.0f / (255.0f * 0.289f) : ci == 1 ? 1.0f / (255.0f * 0.274f) : 1.0f / (255.0f * 0.278f);
            sh[i] = ci == 0 ? -0.408f / 0.289f : ci == 1 ? -0.447f / 0.274f : -0.470f / 0.278f;
        }
        uint32_t v[NITD];
#pragma unroll
        for (int it = 0; it < NITD; ++it) {
            const int cy = min(max(iy0 + rg + it * RG, 0), p.H - 1);
            v[it] = *reinterpret_cast<const uint32_t*>((const uint8_t*)p.x + ((size_t)b * p.H + cy) * p.W * 3 + cboff);
        }
        for (int i = tid; i < 9 * 32; i += S0_NT) Wd[i] = p.wdw[i];
#pragma unroll
        for (int it = 0; it < NITD; ++it) {
            const int r = rg + it * RG, iy = iy0 + r;
            const bool rowok = (unsigned)iy < (unsigned)p.H;
            float f[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float u = (float)((v[it] >> (8 * i)) & 0xffu);             // v_cvt_f32_ubyteN
                f[i] = (rowok && bok[i]) ? fmaf(u, sc[i], sh[i]) : 0.0f;
            }
            if (tact && r < S0_PH) st16(reinterpret_cast<char*>(Xs) + (r * PROW + 4 * d) * 4, pack16<float>(f));
        }
    } else {
        const int e = tid % ECOLS, r0 = tid / ECOLS;
        const bool tact = r0 < RSTEP;
        const int col = e / 3, ci = e - col * 3;
        const int ix = ix0 + col;
        const bool xok = tact && (unsigned)ix < (unsigned)p.W;
        const int cx = min(max(ix, 0), p.W - 1);
        // all loads of the patch are issued before any is consumed (one memory latency per tile);
        // branch-free: loads come from a clamped (valid) address and are discarded when outside
        float v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = r0 + it * RSTEP;
            const int iy = iy0 + r;
            const bool ok = xok && r < S0_PH && (unsigned)iy < (unsigned)p.H;
            const int cy = min(max(iy, 0), p.H - 1);
            if constexpr (FMT == CF_IN_U8_HWC_BGR) {
                const uint32_t u = ((const uint8_t*)p.x)[(((size_t)b * p.H + cy) * p.W + cx) * 3 + ci];
                v[it] = __uint_as_float(ok ? (u | (uint32_t)(ci << 8)) : 0xffffffffu);   // table index | "outside"
            } else {
                const float f = ((const float*)p.x)[(((size_t)b * 3 + ci) * p.H + cy) * p.W + cx];
                v[it] = ok ? f : 0.0f;
            }
        }
        // table + tap weights -> LDS while the patch loads are in flight
        if constexpr (FMT == CF_IN_U8_HWC_BGR) {
            for (int i = tid; i < 768; i += S0_NT) lut[i] = p.lut[i];
        }
        for (int i = tid; i < 9 * 32; i += S0_NT) Wd[i] = p.wdw[i];
        if constexpr (FMT == CF_IN_U8_HWC_BGR) __syncthreads();      // table is in LDS
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = r0 + it * RSTEP;
            float val = v[it];
            if constexpr (FMT == CF_IN_U8_HWC_BGR) {
                const uint32_t idx = __float_as_uint(v[it]);
                val = idx == 0xffffffffu ? 0.0f : lut[idx];
            }
            if (tact && r < S0_PH) {
                if constexpr (F32) reinterpret_cast<float*>(Xs)[r * PROW + e] = val;
                else Xs[r * PROW + e] = (T)(pack_bf16x2(val, 0.0f) & 0xffffu);
            }
        }
    }
    S0_STAMP(0)
    __syncthreads();
    S0_STAMP(1)

    // ---- phase 1: stem conv on MFMA + Swish -> E
    u32x4 ws[F32 ? 4 : 2];
#pragma unroll
    for (int c = 0; c < (F32 ? 4 : 2); ++c) ws[c] = ld16((const char*)p.wstem + ((size_t)c * 64 + lane) * 16);
    for (int ib = wave; ib < NIB; ib += NW) {
        const int ip = ib * 32 + pl;
        const int ipc = ip < S0_IPX ? ip : S0_IPX - 1;
        const int ty = ipc / S0_IW, tx = ipc - ty * S0_IW;           // tile pixel on the H/2 grid
        const int y = oy0 - 1 + ty, x = ox0 - 1 + tx;
        const bool inmap = (unsigned)y < (unsigned)Ho && (unsigned)x < (unsigned)Wo;
        typedef typename std::conditional<F32, float, T>::type XT;
        const XT* xp = reinterpret_cast<const XT*>(Xs) + (2 * ty) * PROW + (2 * tx) * 3 + XOFF;
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.0f;
        if constexpr (F32) {
            float v[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int t = 2 * s + h;
                const int ky = t / 9, rr = t - 9 * ky;
                v[s] = t < 27 ? xp[ky * PROW + rr] : 0.0f;
            }
            mma_chain<T, 4>(a, [&](int c) { return ws[c]; }, [&](int c) { return pack16<float>(&v[4 * c]); });
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                uint32_t w4[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const int t0 = c * 16 + h * 8 + 2 * e2, t1 = t0 + 1;
                    const int ky0 = t0 / 9, r0 = t0 - 9 * ky0, ky1 = t1 / 9, r1 = t1 - 9 * ky1;
                    const uint32_t lo = t0 < 27 ? (uint32_t)xp[ky0 * PROW + r0] : 0u;
                    const uint32_t hi = t1 < 27 ? (uint32_t)xp[ky1 * PROW + r1] : 0u;
                    w4[e2] = lo | (hi << 16);
                }
                u32x4 xc; xc.x = w4[0]; xc.y = w4[1]; xc.z = w4[2]; xc.w = w4[3];
                S0Mma<T>::run(a, ws[c], xc);
            }
        }
        if (ip < S0_IPX) {
            char* erow = E + ip * ROWB + h * 16 * (int)sizeof(T);
#pragma unroll
            for (int g = 0; g < 16 / P; ++g) {
                float v[P];
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = inmap ? a[g * P + e] : 0.0f;     // swish(0) = 0
                if constexpr (PRE) {
#pragma unroll
                    for (int e = 0; e < P; e += 2) { f32x2 x2; x2.x = v[e]; x2.y = v[e + 1]; const f32x2 y2 = swish2_sel<true>(x2); v[e] = y2.x; v[e + 1] = y2.y; }
                } else act_arr<1, P>(v);
                st16(erow + g * 16, pack16<T>(v));
            }
        }
    }
    S0_STAMP(2)
    __syncthreads();
    S0_STAMP(3)

    // ---- phase 2 + 3: depthwise 3x3 + Swish -> project 32->16
    const int o = wave * 32 + (F32 ? lds_group_pixel(pl) : pl);                 // fp32 tile: conflict-free ds_read_b128 groups (cf_common.h)
    const int oy = o / S0_TOW, ox = o % S0_TOW;
    const char* eb0 = E + (oy * S0_IW + ox) * ROWB;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    auto dw_chunk = [&](int j) -> u32x4 {
        const int c = h * HALF + j;
        f32x2 d2[P / 2];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float ev[P], wv[P];
                unpack16<T>(ld16(eb0 + (ky * S0_IW + kx) * ROWB + c * 16), ev);
                const char* wt = (const char*)Wd + ((ky * 3 + kx) * 32 + c * P) * 4;
                unpack16<float>(ld16(wt), wv);
                if constexpr (P == 8) unpack16<float>(ld16(wt + 16), wv + 4);
#pragma unroll
                for (int e = 0; e < P / 2; ++e) {
                    f32x2 e2, w2; e2.x = ev[2 * e]; e2.y = ev[2 * e + 1]; w2.x = wv[2 * e]; w2.y = wv[2 * e + 1];
                    d2[e] = (ky == 0 && kx == 0) ? e2 * w2 : fma2(e2, w2, d2[e]);
                }
            }
        float d[P];
#pragma unroll
        for (int e = 0; e < P / 2; ++e) { const f32x2 y2 = swish2_sel<PRE>(d2[e]); d[2 * e] = y2.x; d[2 * e + 1] = y2.y; }
        return pack16<T>(d);
    };
    mma_chain<T, HALF>(acc, [&](int j) { return ld16((const char*)p.wproj + ((size_t)j * 64 + lane) * 16); }, dw_chunk);
    S0_STAMP(4)
#ifdef CF_X5_TIMING
    if (lane == 0) {
        const size_t wv = (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + wave;
        if (wv < 204800) { for (int k = 0; k < 5; ++k) g_s0_buf[wv * 8 + k] = s0t[k]; g_s0_buf[wv * 8 + 5] = 1ull; }
    }
#endif
    const int gy = oy0 + oy, gx = ox0 + ox;
    if (h != 0 || gy >= Ho || gx >= Wo) return;                  // channels 0..15 live in the h == 0 lanes
    T* out = (T*)p.y + (((size_t)b * Ho + gy) * Wo + gx) * 16;
#pragma unroll
    for (int g = 0; g < 16 / P; ++g) {
        float v[P];
#pragma unroll
        for (int e = 0; e < P; ++e) v[e] = acc[g * P + e];
        st16(out + g * P, pack16<T>(v));
    }
}


// ================================================================== second generation (bf16 storage)
// Same fusion, with the depthwise restructured as in cf_mbconv2.hip: the stem GEMM runs as
// D = X . Ws^T (lane = stem channel, registers = 16 pixels), so x-neighbour pairs sit in one lane and go
// to LDS as fp16 pixel pairs; the 3x3 depthwise is then 6 v_dot2c_f32_f16 per channel against
// wave-uniform tap pairs from SGPRs; a wave owns 64 same-parity pixels and a v_permlane32_swap builds the
// two project-MFMA fragments.  Swish arguments are pre-scaled by -log2(e) through the stem weights and
// the leftover -ln 2 sits in the project weights.  4 waves per 16x16 tile.
typedef __attribute__((ext_vector_type(8))) uint32_t s0_u32x8;
typedef __attribute__((ext_vector_type(2))) _Float16 s0_hf2;
#define S0_AS4 __attribute__((address_space(4)))
constexpr int S0P_NT = 256, S0P_NW = 4;
constexpr int S0P_ND = (S0_PW * 3 + 2 + 3) / 4;                     // 29 dwords cover a patch row (+2 bytes in front)
constexpr int S0P_PROW = 120;                                      // >= 4 * S0P_ND elements, row pitch a multiple of 8 bytes
constexpr int S0P_PITCH = 32 * 4 + 16;                             // one pixel pair, 32 channels (fp16 x 2) + pad
constexpr int S0P_NIB = (S0_IPX + 31) / 32;                        // 11 halo pixel blocks
static constexpr float kS0NegLog2e = -1.44269504088896341f, kS0NegLn2 = -0.69314718055994531f;

static inline uint16_t s0_f32_to_f16(float f) {                    // RNE, saturating
    uint32_t u; __builtin_memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int32_t exp = (int32_t)((u >> 23) & 0xff) - 127 + 15;
    uint32_t man = u & 0x7fffffu;
    if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
    if (exp >= 31) return (uint16_t)(sign | 0x7bffu);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - exp;
        uint32_t half = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1))) ++half;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)exp << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half;
    if (half >= 0x7c00u) half = 0x7bffu;
    return (uint16_t)(sign | half);
}

size_t stem0px_wstem_bytes() { return 2 * 64 * 16; }
size_t stem0px_wdw_dwords() { return 2 * 4 * 3 * 2 * 8; }
// ws [32][3][3][3] (co, ci, ky, kx), wd [32][9], wp [16][32]
void stem0px_pack(const float* ws, const float* wd, const float* wp, void* wstem_out, uint32_t* wdw_out, void* wproj_out) {
    // stem, MFMA B operand: lane (n = channel, half h), chunk c, element e.  The k-slot <-> tap assignment is free
    // (sum over k): group g = 2c + h < 3 holds taps 0..7 of kernel row g (8 CONTIGUOUS patch elements = four aligned
    // dwords in LDS), group 3 holds tap 8 of rows 0, 1, 2 in its even slots (odd slots and slots 6, 7: zero weight,
    // the operand there is whatever follows in the patch).  tap t = ky*9 + kx*3 + ci.
    __builtin_memset(wstem_out, 0, stem0px_wstem_bytes());
    for (int c = 0; c < 2; ++c)
        for (int lane = 0; lane < 64; ++lane) {
            const int co = lane & 31, h = lane >> 5;
            uint16_t* dst = (uint16_t*)((char*)wstem_out + ((size_t)c * 64 + lane) * 16);
            for (int e = 0; e < 8; ++e) {
                const int g = 2 * c + h;
                int t;
                if (g < 3) t = g * 9 + e;
                else if (e < 6 && (e & 1) == 0) t = (e >> 1) * 9 + 8;
                else continue;
                const int ky = t / 9, kx = (t % 9) / 3, ci = t % 3;
                dst[e] = host_f32_to_bf16(kS0NegLog2e * ws[((co * 3 + ci) * 3 + ky) * 3 + kx]);
            }
        }
    // depthwise tap pairs [parity][chunk][ky][t][8]: even x0 -> (w[2t], w[2t+1]); odd x0 -> (w[2t-1], w[2t])
    for (int par = 0; par < 2; ++par)
        for (int c = 0; c < 4; ++c)
            for (int ky = 0; ky < 3; ++ky)
                for (int t = 0; t < 2; ++t)
                    for (int i = 0; i < 8; ++i) {
                        const float* wrow = wd + (c * 8 + i) * 9 + ky * 3;
                        const int k0 = 2 * t - par, k1 = k0 + 1;
                        const uint16_t lo = (k0 >= 0 && k0 < 3) ? s0_f32_to_f16(wrow[k0]) : 0;
                        const uint16_t hi = (k1 >= 0 && k1 < 3) ? s0_f32_to_f16(wrow[k1]) : 0;
                        wdw_out[((((par * 4 + c) * 3 + ky) * 2 + t) * 8) + i] = (uint32_t)lo | ((uint32_t)hi << 16);
                    }
    // project (A operand, as stem0_pack_proj) x -ln 2
    __builtin_memset(wproj_out, 0, stem0_proj_bytes(1));
    for (int j = 0; j < 2; ++j)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31, h = lane >> 5, co = slot_channel(0, i);
            if (co >= 16) continue;
            uint16_t* dst = (uint16_t*)((char*)wproj_out + ((size_t)j * 64 + lane) * 16);
            for (int e = 0; e < 8; ++e) dst[e] = host_f32_to_bf16(kS0NegLn2 * wp[co * 32 + (h * 2 + j) * 8 + e]);
        }
}

__device__ __forceinline__ f32x2 s0_swish2_prescaled(f32x2 u) {
    f32x2 e; e.x = __builtin_amdgcn_exp2f(u.x); e.y = __builtin_amdgcn_exp2f(u.y);
    const f32x2 den = e + 1.0f;
    f32x2 r; r.x = __builtin_amdgcn_rcpf(den.x); r.y = __builtin_amdgcn_rcpf(den.y);
    return u * r;
}
__device__ __forceinline__ void s0_dot2c(float& acc, uint32_t w, uint32_t e) {
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(s0_hf2, w), __builtin_bit_cast(s0_hf2, e), acc, false);
}

template <int FMT>
__global__ __launch_bounds__(S0P_NT) void stem0_px_kernel(Stem0Params p) {
    typedef bf16_t T;
    // The normalised patch Xs is dead once every wave has gathered its MFMA operands, so it shares the
    // LDS bytes of the tile E that phase 1 then writes: 25 KB per workgroup instead of 34 (6 instead of 4
    // workgroups per CU).
    __shared__ __attribute__((aligned(16))) char E[S0P_NIB * 16 * S0P_PITCH];
    static_assert(S0_PH * S0P_PROW * 2 <= S0P_NIB * 16 * S0P_PITCH, "patch must fit under the tile");
    T* Xs = reinterpret_cast<T*>(E);                                     // patch row: 2 pad elements, then e = col*3 + ci

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    unsigned tbx = blockIdx.x, tby = blockIdx.y, tbz = blockIdx.z;
    if (p.kind & 2) xcd_tile_order(tbx, tby, tbz);
    const int ox0 = tbx * S0_TOW, oy0 = tby * S0_TOH, b = tbz;

    // ---- stage the normalised image patch
    const int iy0 = 2 * (oy0 - 1), ix0 = 2 * (ox0 - 1);
    if constexpr (FMT == CF_IN_U8_HWC_BGR) {
        // uint8 input: DWORD loads.  The patch row starts 6 (ox0 - 1) bytes into the image row, i.e. 2 bytes
        // past a dword boundary (ox0 is a multiple of 16, W of 32), so the row is read as 29 aligned
        // dwords starting 2 bytes early, and a dword lies entirely inside or entirely outside the image row.
        // Thread = one dword column (fixed channel phase, fixed column validity) walking down the rows;
        // normalisation is (u/255 - mean)/std as one fma per byte (1 ulp from the reference's two
        // divisions, far below the bf16 rounding that follows); outside the image: exact 0 (ZeroPad2d).
        constexpr int ND = S0P_ND, RG = S0P_NT / ND, NITD = (S0_PH + RG - 1) / RG;
        const int d = tid % ND, rg = tid / ND;
        const bool tact = rg < RG;
        const int boff = ix0 * 3 - 2 + 4 * d;
        const bool din = tact && boff >= 0 && boff + 4 <= p.W * 3;
        const int cboff = min(max(boff, 0), p.W * 3 - 4);
        float sc[4], sh[4]; bool bok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * d - 2 + i;                         // element of the patch row, e = col*3 + ci
            bok[i] = din && e >= 0 && e < S0_PW * 3;
            const int ci = (e + 3) % 3;
            // centerface.py:12-15 (BGR): mean 0.408 0.447 0.470, std 0.289 0.274 0.278
            sc[i] = ci == 0 ? 1.0f / (255.0f * 0.289f) : ci == 1 ? 1.0f / (255.0f * 0.274f) : 1.0f / (255.0f * 0.278f);
            sh[i] = ci == 0 ? -0.408f / 0.289f : ci == 1 ? -0.447f / 0.274f : -0.470f / 0.278f;
        }
        uint32_t v[NITD];
#pragma unroll
        for (int it = 0; it < NITD; ++it) {
            const int cy = min(max(iy0 + rg + it * RG, 0), p.H - 1);
            v[it] = *reinterpret_cast<const uint32_t*>((const uint8_t*)p.x + ((size_t)b * p.H + cy) * p.W * 3 + cboff);
        }
        // Interior tiles (81 % at 640x640: the whole patch inside the image) need no zero-padding selects.  Elements of the
        // dword columns that lie outside the PATCH then hold neighbouring pixels instead of 0: nothing reads them except
        // k-slots whose stem weight is zero (stem0px_pack), and they are finite.
        const bool interior = iy0 >= 0 && iy0 + S0_PH <= p.H && ix0 * 3 - 2 >= 0 && ix0 * 3 - 2 + 4 * ND <= p.W * 3;
        auto convert = [&](auto inside) {
#pragma unroll
            for (int it = 0; it < NITD; ++it) {
                const int r = rg + it * RG, iy = iy0 + r;
                const bool rowok = (unsigned)iy < (unsigned)p.H;
                float f[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float u = (float)((v[it] >> (8 * i)) & 0xffu);             // v_cvt_f32_ubyteN
                    const float nv = fmaf(u, sc[i], sh[i]);
                    f[i] = (decltype(inside)::value || (rowok && bok[i])) ? nv : 0.0f;
                }
                if (tact && r < S0_PH) {
                    u32x2 o; o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]);
                    *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(Xs) + r * (S0P_PROW * 2) + d * 8) = o;
                }
            }
        };
        if (interior) convert(std::true_type{}); else convert(std::false_type{});
    } else {
        constexpr int ECOLS = S0_PW * 3, RSTEP = S0P_NT / ECOLS, NIT = (S0_PH + RSTEP - 1) / RSTEP;
        const int e = tid % ECOLS, r0 = tid / ECOLS;
        const bool tact = r0 < RSTEP;
        const int col = e / 3, ci = e - col * 3;
        const int ix = ix0 + col;
        const bool xok = tact && (unsigned)ix < (unsigned)p.W;
        const int cx = min(max(ix, 0), p.W - 1);
        float v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = r0 + it * RSTEP;
            const int iy = iy0 + r;
            const bool ok = xok && r < S0_PH && (unsigned)iy < (unsigned)p.H;
            const int cy = min(max(iy, 0), p.H - 1);
            const float f = ((const float*)p.x)[(((size_t)b * 3 + ci) * p.H + cy) * p.W + cx];
            v[it] = ok ? f : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = r0 + it * RSTEP;
            if (tact && r < S0_PH) Xs[r * S0P_PROW + 2 + e] = (T)(pack_bf16x2(v[it], 0.0f) & 0xffffu);
        }
    }
    __syncthreads();

    // ---- phase 1: stem conv D[pixel][channel] = X . Ws^T, Swish (pre-scaled), pixel pairs -> E
    u32x4 ws[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) ws[c] = ld16((const char*)p.wstem + ((size_t)c * 64 + lane) * 16);
    constexpr int MAXB = (S0P_NIB + S0P_NW - 1) / S0P_NW;           // 3 halo pixel blocks per wave
    u32x4 xg[MAXB][2];
    // tiles whose whole 18 x 18 halo lies inside the H/2 x W/2 map need no zero-padding selects either (the lanes past the
    // halo in the last block then carry a copy of its last pixel: their results are never activated or stored)
    const bool halo_inside = oy0 >= 1 && oy0 + S0_TOH + 1 <= Ho && ox0 >= 1 && ox0 + S0_TOW + 1 <= Wo;
    auto gather = [&](auto inside) {
#pragma unroll
    for (int t = 0; t < MAXB; ++t) {
        const int ib = wave + S0P_NW * t;
        const int ip = ib * 32 + pl;
        const int ipc = ip < S0_IPX ? ip : S0_IPX - 1;
        const int ty = ipc / S0_IW, tx = ipc - ty * S0_IW;
        const int y = oy0 - 1 + ty, x = ox0 - 1 + tx;
        // a halo pixel outside the map is the depthwise conv's zero padding: zero operand row -> swish(0) = 0
        const bool inmap = decltype(inside)::value || (ip < S0_IPX && (unsigned)y < (unsigned)Ho && (unsigned)x < (unsigned)Wo);
        // eight aligned dword reads per lane, already in operand order (see stem0px_pack): no packing ops
        const char* xp = reinterpret_cast<const char*>(Xs) + ((2 * ty) * S0P_PROW + (2 * tx) * 3 + 2) * 2;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint32_t w4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = c == 0 ? h * (S0P_PROW * 2) + 4 * i
                                       : (h == 0 ? 2 * (S0P_PROW * 2) + 4 * i : 16 + (i < 2 ? i : 2) * (S0P_PROW * 2));
                const uint32_t v = *reinterpret_cast<const uint32_t*>(xp + off);
                w4[i] = inmap ? v : 0u;
            }
            xg[t][c].x = w4[0]; xg[t][c].y = w4[1]; xg[t][c].z = w4[2]; xg[t][c].w = w4[3];
        }
    }
    };
    if (halo_inside) gather(std::true_type{}); else gather(std::false_type{});
    __syncthreads();                                              // every wave has its operands: Xs may be overwritten
#pragma unroll
    for (int t = 0; t < MAXB; ++t) {
        const int ib = wave + S0P_NW * t;
        if (ib >= S0P_NIB) break;
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, xg[t][c]),
                                                        __builtin_bit_cast(mfma_bf16x8, ws[c]), a, 0, 0, 0);
        char* ecol = E + pl * 4 + (unsigned)(ib * 16 + 2 * h) * (unsigned)S0P_PITCH;
        // the last halo block (18 x 18 = 10 blocks + 4 pixels) holds VP = 2 real pixel pairs: the register pairs past them
        // are neither activated nor stored (wave-uniform branch; only the round that can hold the last block tests it)
        constexpr int VP = (S0_IPX - 32 * (S0P_NIB - 1)) / 2;
        auto activate = [&](auto lastb) {
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                if (decltype(lastb)::value && (tt & 1) + 4 * (tt >> 1) >= VP) continue;
                f32x2 x2; x2.x = a[2 * tt]; x2.y = a[2 * tt + 1];
                const f32x2 y2 = s0_swish2_prescaled(x2);
                *reinterpret_cast<uint32_t*>(ecol + ((tt & 1) + 4 * (tt >> 1)) * S0P_PITCH) =
                    __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y2.x, y2.y));
            }
        };
        if (VP < 16 && t == (S0P_NIB - 1) / S0P_NW && ib == S0P_NIB - 1) activate(std::true_type{});
        else activate(std::false_type{});
    }
    __syncthreads();

    // ---- phase 2 + 3: depthwise 3x3 + Swish on 64 same-parity pixels per wave -> project 32 -> 16
    const int par = wave >> 1;                                      // waves 0,1: even x; 2,3: odd x
    // row-major lane -> pixel order.  (The bank-conflict-free LaneMap of cf_common.h was measured here too: LDS conflicts
    // 47 % -> 19 % of the LDS cycles, kernel 190 -> 194 us -- this kernel is not LDS-bound and its 32-byte output rows
    // coalesce better when neighbouring lanes own neighbouring pixels.)
    auto tile_pixel = [](int u, int& oy, int& ox) {
        const int pr = u >> 7, r = u & 127;
        oy = r >> 3; ox = 2 * (r & 7) + pr;
    };
    int dy, dx; tile_pixel((wave * 2 + h) * 32 + pl, dy, dx);
    const char* eb0 = E + (unsigned)((dy * S0_IW + (dx - par)) / 2) * (unsigned)S0P_PITCH;
    const S0_AS4 s0_u32x8* wtab = (const S0_AS4 s0_u32x8*)p.wdw;

    auto dw_chunk = [&](int c) -> u32x4 {
        float a8[8];
        const S0_AS4 s0_u32x8* wq = wtab + ((par * 4 + c) * 3) * 2;
        const char* eb = eb0 + c * 32;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const s0_u32x8 wv = wq[ky * 2 + t];
                const char* et = eb + (ky * (S0_IW / 2) + t) * S0P_PITCH;
                const u32x4 e0 = ld16(et), e1 = ld16(et + 16);
                if (ky == 0 && t == 0) { dot8_first(a8, wv, e0, e1); continue; }      // a8 = w . e + 0: no zero-init moves
                s0_dot2c(a8[0], wv[0], e0.x); s0_dot2c(a8[1], wv[1], e0.y);
                s0_dot2c(a8[2], wv[2], e0.z); s0_dot2c(a8[3], wv[3], e0.w);
                s0_dot2c(a8[4], wv[4], e1.x); s0_dot2c(a8[5], wv[5], e1.y);
                s0_dot2c(a8[6], wv[6], e1.z); s0_dot2c(a8[7], wv[7], e1.w);
            }
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            f32x2 u; u.x = a8[i]; u.y = a8[i + 1];
            const f32x2 yv = s0_swish2_prescaled(u);
            a8[i] = yv.x; a8[i + 1] = yv.y;
        }
        return pack16<T>(a8);
    };

    f32x16 acc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const u32x4 wpc = ld16((const char*)p.wproj + ((size_t)j * 64 + lane) * 16);
        const u32x4 dA = dw_chunk(j), dB = dw_chunk(2 + j);
        u32x4 x0, x1;
        {
            auto s0 = __builtin_amdgcn_permlane32_swap(dA.x, dB.x, false, false); x0.x = s0[0]; x1.x = s0[1];
            auto s1 = __builtin_amdgcn_permlane32_swap(dA.y, dB.y, false, false); x0.y = s1[0]; x1.y = s1[1];
            auto s2 = __builtin_amdgcn_permlane32_swap(dA.z, dB.z, false, false); x0.z = s2[0]; x1.z = s2[1];
            auto s3 = __builtin_amdgcn_permlane32_swap(dA.w, dB.w, false, false); x0.w = s3[0]; x1.w = s3[1];
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, wpc), __builtin_bit_cast(mfma_bf16x8, x0), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, wpc), __builtin_bit_cast(mfma_bf16x8, x1), acc[1], 0, 0, 0);
    }
    if (h != 0) return;                                            // channels 0..15 live in the h == 0 lanes
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int oy, ox; tile_pixel((wave * 2 + k) * 32 + pl, oy, ox);
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (gy >= Ho || gx >= Wo) continue;
        T* out = (T*)p.y + (((size_t)b * Ho + gy) * Wo + gx) * 16;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[k][g * 8 + e];
            st16(out + g * 8, pack16<T>(v));
        }
    }
}

// ---- third generation: the 3x3 depthwise on the matrix cores (cf_mx.h, cf_mbconv3.hip) ----
constexpr int S0M_IWQ = S0_TOW / 4 + 1, S0M_IWP = 4 * S0M_IWQ;      // halo rows of 5 x-quads (20 columns, 18 used)
constexpr int S0M_IPX = S0_IH * S0M_IWP, S0M_NIB = (S0_IH * S0M_IWQ + 7) / 8, S0M_CP = 32 * 8 + 16;
size_t stem0mx_wdw_dwords() { return 2 * 3 * 2 * 64 * 2; }
// stem weights as stem0px_pack; Toeplitz depthwise operands (mx_pack_taps, one round of 32 channels); project 32 -> 16 as
// the A fragment of v_mfma_f32_16x16x32_bf16 (lane: row m = output channel, k-group lane >> 4 = 8 hidden channels), x -ln 2
void stem0mx_pack(const float* ws, const float* wd, const float* wp, void* wstem_out, uint32_t* wdw_out, void* wproj_out) {
    std::vector<uint32_t> scratch(stem0px_wdw_dwords());
    std::vector<char> pscratch(stem0_proj_bytes(1));
    stem0px_pack(ws, wd, wp, wstem_out, scratch.data(), pscratch.data());
    mx_pack_taps(1, 3, wd, wdw_out);
    __builtin_memset(wproj_out, 0, stem0_proj_bytes(1));
    for (int lane = 0; lane < 64; ++lane) {
        const int co = lane & 15, kc = lane >> 4;
        uint16_t* dst = (uint16_t*)((char*)wproj_out + (size_t)lane * 16);
        for (int e = 0; e < 8; ++e) dst[e] = host_f32_to_bf16(kS0NegLn2 * wp[co * 32 + kc * 8 + e]);
    }
}

template <int FMT>
__global__ __launch_bounds__(S0P_NT) void stem0_mx_kernel(Stem0Params p) {
    typedef bf16_t T;
    // The normalised patch Xs is dead once every wave has gathered its MFMA operands, so it shares the
    // LDS bytes of the tile E that phase 1 then writes: 25 KB per workgroup instead of 34 (6 instead of 4
    // workgroups per CU).
    __shared__ __attribute__((aligned(16))) char E[S0M_NIB * 8 * S0M_CP];
    static_assert(S0_PH * S0P_PROW * 2 <= S0M_NIB * 8 * S0M_CP, "patch must fit under the tile");
    T* Xs = reinterpret_cast<T*>(E);                                     // patch row: 2 pad elements, then e = col*3 + ci

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    unsigned tbx = blockIdx.x, tby = blockIdx.y, tbz = blockIdx.z;
    if (p.kind & 2) xcd_tile_order(tbx, tby, tbz);
    const int ox0 = tbx * S0_TOW, oy0 = tby * S0_TOH, b = tbz;

    // ---- stage the normalised image patch
    const int iy0 = 2 * (oy0 - 1), ix0 = 2 * (ox0 - 1);
    if constexpr (FMT == CF_IN_U8_HWC_BGR) {
        // uint8 input: DWORD loads.  The patch row starts 6 (ox0 - 1) bytes into the image row, i.e. 2 bytes
        // past a dword boundary (ox0 is a multiple of 16, W of 32), so the row is read as 29 aligned
        // dwords starting 2 bytes early, and a dword lies entirely inside or entirely outside the image row.
        // Thread = one dword column (fixed channel phase, fixed column validity) walking down the rows;
        // normalisation is (u/255 - mean)/std as one fma per byte (1 ulp from the reference's two
        // divisions, far below the bf16 rounding that follows); outside the image: exact 0 (ZeroPad2d).
        constexpr int ND = S0P_ND, RG = S0P_NT / ND, NITD = (S0_PH + RG - 1) / RG;
        const int d = tid % ND, rg = tid / ND;
        const bool tact = rg < RG;
        const int boff = ix0 * 3 - 2 + 4 * d;
        const bool din = tact && boff >= 0 && boff + 4 <= p.W * 3;
        const int cboff = min(max(boff, 0), p.W * 3 - 4);
        float sc[4], sh[4]; bool bok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * d - 2 + i;                         // element of the patch row, e = col*3 + ci
            bok[i] = din && e >= 0 && e < S0_PW * 3;
            const int ci = (e + 3) % 3;
            // centerface.py:12-15 (BGR): mean 0.408 0.447 0.470, std 0.289 0.274 0.278
            sc[i] = ci == 0 ? 1.0f / (255.0f * 0.289f) : ci == 1 ? 1.0f / (255.0f * 0.274f) : 1.0f / (255.0f * 0.278f);
            sh[i] = ci == 0 ? -0.408f / 0.289f : ci == 1 ? -0.447f / 0.274f : -0.470f / 0.278f;
        }
        uint32_t v[NITD];
#pragma unroll
        for (int it = 0; it < NITD; ++it) {
            const int cy = min(max(iy0 + rg + it * RG, 0), p.H - 1);
            v[it] = *reinterpret_cast<const uint32_t*>((const uint8_t*)p.x + ((size_t)b * p.H + cy) * p.W * 3 + cboff);
        }
        // Interior tiles (81 % at 640x640: the whole patch inside the image) need no zero-padding selects.  Elements of the
        // dword columns that lie outside the PATCH then hold neighbouring pixels instead of 0: nothing reads them except
        // k-slots whose stem weight is zero (stem0px_pack), and they are finite.
        const bool interior = iy0 >= 0 && iy0 + S0_PH <= p.H && ix0 * 3 - 2 >= 0 && ix0 * 3 - 2 + 4 * ND <= p.W * 3;
        auto convert = [&](auto inside) {
#pragma unroll
            for (int it = 0; it < NITD; ++it) {
                const int r = rg + it * RG, iy = iy0 + r;
                const bool rowok = (unsigned)iy < (unsigned)p.H;
                float f[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float u = (float)((v[it] >> (8 * i)) & 0xffu);             // v_cvt_f32_ubyteN
                    const float nv = fmaf(u, sc[i], sh[i]);
                    f[i] = (decltype(inside)::value || (rowok && bok[i])) ? nv : 0.0f;
                }
                if (tact && r < S0_PH) {
                    u32x2 o; o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]);
                    *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(Xs) + r * (S0P_PROW * 2) + d * 8) = o;
                }
            }
        };
        if (interior) convert(std::true_type{}); else convert(std::false_type{});
    } else {
        constexpr int ECOLS = S0_PW * 3, RSTEP = S0P_NT / ECOLS, NIT = (S0_PH + RSTEP - 1) / RSTEP;
        const int e = tid % ECOLS, r0 = tid / ECOLS;
        const bool tact = r0 < RSTEP;
        const int col = e / 3, ci = e - col * 3;
        const int ix = ix0 + col;
        const bool xok = tact && (unsigned)ix < (unsigned)p.W;
        const int cx = min(max(ix, 0), p.W - 1);
        float v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = r0 + it * RSTEP;
            const int iy = iy0 + r;
            const bool ok = xok && r < S0_PH && (unsigned)iy < (unsigned)p.H;
            const int cy = min(max(iy, 0), p.H - 1);
            const float f = ((const float*)p.x)[(((size_t)b * 3 + ci) * p.H + cy) * p.W + cx];
            v[it] = ok ? f : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = r0 + it * RSTEP;
            if (tact && r < S0_PH) Xs[r * S0P_PROW + 2 + e] = (T)(pack_bf16x2(v[it], 0.0f) & 0xffffu);
        }
    }
    __syncthreads();

    // ---- phase 1: stem conv D[pixel][channel] = X . Ws^T, Swish (pre-scaled), pixel pairs -> E
    u32x4 ws[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) ws[c] = ld16((const char*)p.wstem + ((size_t)c * 64 + lane) * 16);
    // Toeplitz operands of the 32 stem channels: [2 channel quads][3 rows][2 k-steps] register pairs (mx_pack_taps)
    u32x2 A[2][3][2];
    {
        const u32x2* at = reinterpret_cast<const u32x2*>(p.wdw) + lane;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) A[q][ky][ks] = at[((q * 3 + ky) * 2 + ks) * 64];
    }
    constexpr int MAXB = (S0M_NIB + S0P_NW - 1) / S0P_NW;           // 3 halo pixel blocks per wave (18 rows x 5 quads = 90 quads)
    u32x4 xg[MAXB][2];
    // tiles whose whole 18 x 18 halo lies inside the H/2 x W/2 map need no zero-padding selects either (the lanes past the
    // halo in the last block then carry a copy of its last pixel: their results are never activated or stored)
    const bool halo_inside = oy0 >= 1 && oy0 + S0_TOH + 1 <= Ho && ox0 >= 1 && ox0 + S0_TOW + 1 <= Wo;
    auto gather = [&](auto inside) {
#pragma unroll
    for (int t = 0; t < MAXB; ++t) {
        const int ib = wave + S0P_NW * t;
        const int ip = ib * 32 + pl;
        // halo rows of 5 x-quads = 20 columns: columns 18, 19 lie past the 18-wide halo (no output reads them) and repeat column 17
        const int ipc = ip < S0M_IPX ? ip : S0M_IPX - 1;
        const int ty = ipc / S0M_IWP, txq = ipc - ty * S0M_IWP, tx = txq < S0_IW ? txq : S0_IW - 1;
        const int y = oy0 - 1 + ty, x = ox0 - 1 + tx;
        // a halo pixel outside the map is the depthwise conv's zero padding: zero operand row -> swish(0) = 0
        const bool inmap = decltype(inside)::value || (ip < S0M_IPX && (unsigned)y < (unsigned)Ho && (unsigned)x < (unsigned)Wo);
        // eight aligned dword reads per lane, already in operand order (see stem0px_pack): no packing ops
        const char* xp = reinterpret_cast<const char*>(Xs) + ((2 * ty) * S0P_PROW + (2 * tx) * 3 + 2) * 2;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint32_t w4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = c == 0 ? h * (S0P_PROW * 2) + 4 * i
                                       : (h == 0 ? 2 * (S0P_PROW * 2) + 4 * i : 16 + (i < 2 ? i : 2) * (S0P_PROW * 2));
                const uint32_t v = *reinterpret_cast<const uint32_t*>(xp + off);
                w4[i] = inmap ? v : 0u;
            }
            xg[t][c].x = w4[0]; xg[t][c].y = w4[1]; xg[t][c].z = w4[2]; xg[t][c].w = w4[3];
        }
    }
    };
    if (halo_inside) gather(std::true_type{}); else gather(std::false_type{});
    __syncthreads();                                              // every wave has its operands: Xs may be overwritten
#pragma unroll
    for (int t = 0; t < MAXB; ++t) {
        const int ib = wave + S0P_NW * t;
        if (ib >= S0M_NIB) break;
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, xg[t][c]),
                                                        __builtin_bit_cast(mfma_bf16x8, ws[c]), a, 0, 0, 0);
        // D rows (r & 3) + 8 (r >> 2) + 4 h: register quad tq = halo quad ib * 8 + 2 tq + h -> one 8-byte cell of lane channel pl
        char* ecell = E + (unsigned)(ib * 8 + h) * (unsigned)S0M_CP + pl * 8;
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            f32x2 u0, u1; u0.x = a[4 * tq]; u0.y = a[4 * tq + 1]; u1.x = a[4 * tq + 2]; u1.y = a[4 * tq + 3];
            const f32x2 y0 = swish2_pre(u0), y1 = swish2_pre(u1);
            u32x2 d;
            d.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y0.x, y0.y));
            d.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y1.x, y1.y));
            *reinterpret_cast<u32x2*>(ecell + 2 * tq * S0M_CP) = d;
        }
    }
    __syncthreads();

    // ---- phase 2 + 3: depthwise 3x3 on the matrix cores (cf_mx.h) + Swish -> project 32 -> 16.  One set of 16 output quads
    // per wave; the lane's 8 channels of output pixel i are the B fragment of v_mfma_f32_16x16x32_bf16 (n = quad slot).
    static constexpr SetMap<S0_TOH, S0_TOW, S0M_IWQ> kSets{};
    const uint32_t se = kSets.v[wave * 16 + (lane & 15)];
    const int soy = (se >> 6) & 0x1ff, soxq = se & 63, kg = lane >> 4;
    f32x4 acc[8];
    mx_depthwise<3, S0M_IWQ, S0M_CP>(E + (unsigned)(soy * S0M_IWQ + soxq) * (unsigned)S0M_CP + kg * 64,
                                      reinterpret_cast<const u32x2 (*)[3][2]>(A), acc);
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            f32x2 u; u.x = acc[g][i]; u.y = acc[g][i + 1];
            const f32x2 yv = swish2_pre(u);
            acc[g][i] = yv.x; acc[g][i + 1] = yv.y;
        }
    const u32x4 wpc = ld16((const char*)p.wproj + (size_t)lane * 16);
    const int gy = oy0 + soy, gx0 = ox0 + 4 * soxq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32x4 d;
        d.x = packb(acc[0][i], acc[1][i]); d.y = packb(acc[2][i], acc[3][i]);
        d.z = packb(acc[4][i], acc[5][i]); d.w = packb(acc[6][i], acc[7][i]);
        f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, wpc), __builtin_bit_cast(mfma_bf16x8, d), o, 0, 0, 0);
        if (gy < Ho && gx0 + i < Wo) {                             // rows 4 kg .. 4 kg + 3 = output channels of pixel (quad, i)
            u32x2 ov; ov.x = packb(o[0], o[1]); ov.y = packb(o[2], o[3]);
            *reinterpret_cast<u32x2*>((T*)p.y + (((size_t)b * Ho + gy) * Wo + gx0 + i) * 16 + kg * 4) = ov;
        }
    }
}

hipError_t launch_stem0(hipStream_t s, int dtype, const Stem0Params& p) {
    if (p.B <= 0) return hipSuccess;
    const int Ho = p.H / 2, Wo = p.W / 2;
    if (p.kind & 4) {                      // matrix-core depthwise (bf16 storage)
        if (dtype != 1) return hipErrorInvalidValue;
        dim3 grid((Wo + S0_TOW - 1) / S0_TOW, (Ho + S0_TOH - 1) / S0_TOH, p.B), blk(S0P_NT);
        set_kernel_tag("void cf::stem0_mx_kernel<%d>(cf::Stem0Params)", p.in_format);
        if (p.in_format == CF_IN_U8_HWC_BGR) hipLaunchKernelGGL((stem0_mx_kernel<CF_IN_U8_HWC_BGR>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((stem0_mx_kernel<CF_IN_F32_NCHW>), grid, blk, 0, s, p);
        return hipGetLastError();
    }
    if (p.kind & 1) {
        if (dtype != 1) return hipErrorInvalidValue;
        dim3 grid((Wo + S0_TOW - 1) / S0_TOW, (Ho + S0_TOH - 1) / S0_TOH, p.B), blk(S0P_NT);
        set_kernel_tag("void cf::stem0_px_kernel<%d>(cf::Stem0Params)", p.in_format);
        if (p.in_format == CF_IN_U8_HWC_BGR) hipLaunchKernelGGL((stem0_px_kernel<CF_IN_U8_HWC_BGR>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((stem0_px_kernel<CF_IN_F32_NCHW>), grid, blk, 0, s, p);
        return hipGetLastError();
    }
    dim3 grid((Wo + S0_TOW - 1) / S0_TOW, (Ho + S0_TOH - 1) / S0_TOH, p.B), blk(S0_NT);
    set_kernel_tag("void cf::stem0_kernel<%s, %d>(cf::Stem0Params)", dtype == 0 ? "float" : dtype == 2 ? "sp32_t" : "unsigned short", p.in_format);
    if (dtype == 0) {
        if (p.in_format == CF_IN_U8_HWC_BGR) hipLaunchKernelGGL((stem0_kernel<float, CF_IN_U8_HWC_BGR>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((stem0_kernel<float, CF_IN_F32_NCHW>), grid, blk, 0, s, p);
    } else if (dtype == 2) {
        if (p.in_format == CF_IN_U8_HWC_BGR) hipLaunchKernelGGL((stem0_kernel<sp32_t, CF_IN_U8_HWC_BGR>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((stem0_kernel<sp32_t, CF_IN_F32_NCHW>), grid, blk, 0, s, p);
    } else {
        if (p.in_format == CF_IN_U8_HWC_BGR) hipLaunchKernelGGL((stem0_kernel<bf16_t, CF_IN_U8_HWC_BGR>), grid, blk, 0, s, p);
        else hipLaunchKernelGGL((stem0_kernel<bf16_t, CF_IN_F32_NCHW>), grid, blk, 0, s, p);
    }
    return hipGetLastError();
}

#ifdef CF_X5_TIMING
}  // namespace cf
extern "C" int cf_debug_stem_stamps(unsigned long long* out6, int reset) {
    static unsigned long long host[204800 * 8];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(cf::g_s0_buf), sizeof host) != hipSuccess) return -1;
    if (out6) { for (int k = 0; k < 6; ++k) out6[k] = 0; for (size_t w = 0; w < 204800; ++w) for (int k = 0; k < 6; ++k) out6[k] += host[w * 8 + k]; }
    if (reset) { static unsigned long long z[204800 * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(cf::g_s0_buf), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
namespace cf {
#endif
}  // namespace cf
