// Last IDAUp stage + the four heads as ONE kernel (bf16 storage, collapsed heads): the 24-channel
// stride-4 neck output lives only in LDS.
//
// Replaces, fused: IDAUp.forward for up3 (model/centernet.py:200-204: relu(bn(conv1x1(skip))) +
// relu(bn_up(deconv2x2(low)))), called at :274) and, per head, Conv2d(24,24,3,padding=1) -> Conv2d(24,c,1)
// (:247-261, :277-279) + the sigmoid/clamp on hm (centerface.py:43).  Layer by layer the neck output is
// written once (79 MB per batch of 64) and read back ~1.5-2x through the 3x3 window; here a workgroup
// computes it for an 8x32 tile plus a one-pixel halo straight into LDS (exactly the bf16 values the
// unfused kernel would store, pixels outside the map = the 3x3 conv's zero padding) and the head conv
// reads its 3 x 72 contiguous elements per output pixel from there.  Same arithmetic, same operation
// order as cf_pw.hip (IDAUp epilogue) followed by cf_head.hip (collapsed), so results are bit-identical
// to the two-kernel path.
#include "cf_exp.h"
#include "cf_common.h"
#include "cf_kernels.h"
#include <cstdlib>

namespace cf {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

constexpr int UH_TW = 32, UH_IW = UH_TW + 2;      // tile width; the tile height TH and the wave count NW are template parameters

// T = bf16_t (benchmarked mode), sp32_t (tolerance mode: fp32 tile, split-bf16 products) or float (exact fp32 MFMA).
// TH x 32 output tile (TH + 2 halo rows: 10 x 34 = 340 pixels at TH = 8, 18 x 34 = 612 at TH = 16), NW waves; NT: the head
// records (105 MB per batch of 64, read back only at the K decoded cells) are stored non-temporally.
template <typename T, int UH_TH, int NW>
struct Uh {
    static constexpr int P = Elem<T>::PER16;
    static constexpr int PIT = 24 * (int)sizeof(T);                 // bytes per tile pixel: 24 channels
    static constexpr int NC = PIT / 16, NCH = (NC + 1) / 2;        // 1x1 conv: chunks per pixel row / per lane half (k-steps)
    static constexpr int CPD = 3 * NC, SPD = (CPD + 1) / 2;        // head conv: chunks per kernel row (3 pixels) / steps per lane half
    static constexpr int G = 16 / P;                                // channel groups of P per lane (16 output channels)
    static constexpr int IH = UH_TH + 2, IPX = IH * UH_IW, NIB = (IPX + 31) / 32, MAXB = (NIB + NW - 1) / NW;
    static constexpr int T3B = NIB * 32 * PIT, WHB = 3 * SPD * 1024, RSB = NW * 2048;
    static constexpr int LDS = T3B + WHB + RSB;
};

template <typename T, bool COALESCE, int UH_TH = 8, int NW = 4, bool NT = false>
__global__ __launch_bounds__(NW * 64) void uphead_kernel(UpHeadParams p) {
    typedef Uh<T, UH_TH, NW> U;
    constexpr int P = U::P, PIT = U::PIT, NC = U::NC, NCH = U::NCH, CPD = U::CPD, SPD = U::SPD, G = U::G;
    constexpr int UH_IPX = U::IPX, UH_NIB = U::NIB, MAXB = U::MAXB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* T3 = smem;                       // neck tile incl. halo, storage type T
    char* Wh = smem + U::T3B;              // head weight fragments
    char* Rs = Wh + U::WHB;                // one row of 32 records per wave, staged for full-line stores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    unsigned tbx = blockIdx.x, tby = blockIdx.y, tbz = blockIdx.z;
    if (p.xcd) xcd_tile_order(tbx, tby, tbz);       // the 3x3 halo rows and the half-resolution `low` rows are shared by neighbours
    const int ox0 = tbx * UH_TW, oy0 = tby * UH_TH, b = tbz;

    // head weights -> LDS by DMA; lands under phase A, fenced by the barrier
    for (int c = wave; c < U::WHB / 1024; c += NW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)p.w0p + c * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(Wh + c * 1024), 16, 0, 0);

    // ---- phase A: up3 on the halo tile.  1x1 conv 24 -> 24 as in cf_pw.hip (lane half h owns chunks h NCH .. of the pixel row; a
    // chunk past the row meets a zeroed operand), epilogue bias + ReLU, + ReLU(low * tap weight + shift), storage type -> LDS
    u32x4 wc[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) wc[j] = ld16((const char*)p.wcv + (size_t)(j * 64 + lane) * 16);
    // All global loads of this wave's halo blocks are issued before the first result is consumed; padding selects are applied
    // afterwards, on the registers.
    u32x4 xs[MAXB][NCH], lw[MAXB][G];
    bool valid[MAXB]; int tapv[MAXB];
#pragma unroll
    for (int t = 0; t < MAXB; ++t) {
        const int ib = wave + NW * t;
        const int ip = ib * 32 + pl;
        const int ipc = ip < UH_IPX ? ip : UH_IPX - 1;
        const int ty = ipc / UH_IW, tx = ipc - ty * UH_IW;
        const int gy = oy0 - 1 + ty, gx = ox0 - 1 + tx;
        valid[t] = ib < UH_NIB && ip < UH_IPX && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w;
        const int cy = min(max(gy, 0), p.h - 1), cx = min(max(gx, 0), p.w - 1);
        const char* xrow = (const char*)p.skip + (((size_t)b * p.h + cy) * p.w + cx) * PIT;
#pragma unroll
        for (int j = 0; j < NCH; ++j) xs[t][j] = ld16(xrow + min(h * NCH + j, NC - 1) * 16);      // a chunk past the row: zeroed below
        const size_t low_row = ((size_t)b * (p.h >> 1) + (cy >> 1)) * (p.w >> 1) + (cx >> 1);
        tapv[t] = ((cy & 1) << 1) | (cx & 1);
#pragma unroll
        for (int g = 0; g < G; ++g) lw[t][g] = ld16((const char*)p.low + (low_row * 24 + min(h * 16 + g * P, 24 - P)) * sizeof(T));
    }
#pragma unroll
    for (int t = 0; t < MAXB; ++t) {
        const int ib = wave + NW * t;
        if (ib >= UH_NIB) break;
        const int ip = ib * 32 + pl;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        mma_chain<T, NCH>(acc, [&](int j) { return wc[j]; }, [&](int j) { return (h * NCH + j) < NC ? xs[t][j] : zero16(); });
        const int tap = tapv[t];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int ch = h * 16 + g * P;
            if (ch >= 24) break;
            float v[P], r[P];
#pragma unroll
            for (int e = 0; e < P; ++e) v[e] = relu_f(acc[g * P + e] + p.bias[ch + e]);
            unpack16<T>(lw[t][g], r);
#pragma unroll
            for (int e = 0; e < P; ++e) v[e] += relu_f(r[e] * p.upw[tap * 24 + ch + e] + p.upb[ch + e]);
            u32x4 o = pack16<T>(v);
            if (!valid[t]) o = zero16();
            st16(T3 + ip * PIT + ch * (int)sizeof(T), o);
        }
    }
    cf_sync_lds_dma();            // the tile is complete and the head weights (LDS-DMA) have landed for every wave

    // ---- phase B: collapsed 3x3 head conv from the LDS tile (cf_head.hip: kernel row dy = 72 contiguous elements = CPD chunks;
    // lane half h owns chunks h SPD ..)
    for (int ob = wave; ob < UH_TH * UH_TW / 32; ob += NW) {
        const int o = ob * 32 + pl;
        const int oy = o / UH_TW, ox = o - oy * UH_TW;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const char* row = T3 + ((oy + dy) * UH_IW + ox) * PIT;
            mma_chain<T, SPD>(acc, [&](int j) { return ld16(Wh + ((dy * SPD + j) * 64 + lane) * 16); },
                              [&](int j) { const int c = h * SPD + j; u32x4 xc = ld16(row + (c < CPD ? c : CPD - 1) * 16);
                                           if (c >= CPD) xc = zero16(); return xc; });
        }
        const int gy = oy0 + oy, gx = ox0 + ox;
        if constexpr (!COALESCE) { if (gy >= p.h || gx >= p.w) continue; }
        // head_pack_weights(collapsed = 2): MFMA row = record slot, so lane half h holds slots 4h..4h+3 (acc 0-3) and
        // 8+4h..8+4h+3 (acc 4-7) of its pixel: every lane stores two 16-byte pieces and a 64-byte record is written by
        // two store instructions of the wave instead of four half-empty ones.  Slot 15 = the raw hm logit (slot 0's
        // weights again: identical arithmetic), slot 0 = its clamped sigmoid.
        float out[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { out[r] = acc[r] + p.b0[h * 4 + r]; out[4 + r] = acc[4 + r] + p.b0[8 + h * 4 + r]; }
        const size_t m = ((size_t)b * p.h + gy) * p.w + gx;
        const bool inmap = gy < p.h && gx < p.w;
        if (h == 0) {
            // centerface.py:43: clamp(sigmoid(hm), 1e-4, 1 - 1e-4); precise exp + IEEE divide
            float sg = 1.0f / (1.0f + expf(-out[0]));
            sg = fminf(fmaxf(sg, 1e-4f), 1.0f - 1e-4f);
            out[0] = sg;
            if (p.hm_plane && inmap) p.hm_plane[m] = sg;
        }
        if constexpr (COALESCE) {
            // a pixel block is one tile row of 32 cells = 2 KB of consecutive records: staged through LDS so that every store
            // instruction of the wave writes 1 KB of consecutive bytes instead of 64 16-byte pieces 64 bytes apart
            char* rs = Rs + wave * 2048;
            st16(rs + pl * 64 + h * 16, pack16<float>(&out[0]));
            st16(rs + pl * 64 + 32 + h * 16, pack16<float>(&out[4]));
            __builtin_amdgcn_wave_barrier();
            if (gy < p.h) {                                                     // wave-uniform (oy is)
                char* row0 = (char*)(p.heads + (((size_t)b * p.h + gy) * p.w + ox0) * 16);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int q = lane + 64 * k;                                // 16-byte piece q of the row: record q / 4
                    const u32x4 v = ld16(rs + q * 16);
                    if (ox0 + (q >> 2) < p.w) {
                        if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(row0 + q * 16));
                        else st16(row0 + q * 16, v);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (!COALESCE) {
            float* dst = p.heads + m * 16 + h * 4;
            st16(dst, pack16<float>(&out[0]));
            st16(dst + 8, pack16<float>(&out[4]));
        }
    }
}

template <typename T, bool CO, int TH, int NW, bool NT>
static hipError_t uphead_launch_t(hipStream_t s, const UpHeadParams& q) {
    typedef Uh<T, TH, NW> U;
    auto kfn = uphead_kernel<T, CO, TH, NW, NT>;
    static thread_local bool configured_dev[32] = {};               // function attributes are per device
    int dev = 0; (void)hipGetDevice(&dev);
    bool& configured = configured_dev[dev & 31];
    if (U::LDS > 64 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, U::LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    dim3 grid((q.w + UH_TW - 1) / UH_TW, (q.h + TH - 1) / TH, q.B), blk(NW * 64);
    set_kernel_tag("void cf::uphead_kernel<%s, %s, %d, %d, %s>(cf::UpHeadParams)", type_tag<T>(), CO ? "true" : "false", TH, NW, NT ? "true" : "false");
    hipLaunchKernelGGL(kfn, grid, blk, U::LDS, s, q);
    return hipGetLastError();
}

hipError_t launch_uphead(hipStream_t s, int dtype, const UpHeadParams& p) {
    if (p.B <= 0) return hipSuccess;
    static const bool xcd_on = cf_ab_int("CF_UH_XCD", 0) >= 1;      // A/B only: 0.090 -> 0.094 ms with it
    UpHeadParams q = p; q.xcd = xcd_on ? 1 : 0;
#include CF_EXP_INC(cf_uphead_0)
    // bf16: 16 x 32 tiles on eight waves, non-temporal record stores: halo rows 2 / 8 -> 2 / 16 of the skip / low fetch (B = 64, 640x640,
    // HIP events, same box: 8x32 / 4 waves 79.0 us, + non-temporal 78.1, 16x32 / 8 waves 73.2, + non-temporal 72.3; 16x32 / 4 waves 84.9,
    // 32x32 / 8 waves 86.4, 8x32 / 8 waves 91.4); every variant is bit-identical to the two-kernel path (test_fused_up3_heads_...)
    if (dtype == 1) return uphead_launch_t<bf16_t, true, 16, 8, true>(s, q);
    // fp32 tile (twice the LDS per pixel): 8 x 32 tiles on eight waves, 77.8 KB = two workgroups per CU.  B = 64, 640x640: 0.132 ms
    // (four waves 0.154, 16x32 / 8 waves 0.159, 4x32 / 4 waves 0.187) against 0.103 + 0.149 for the two launches
    if (dtype == 2) return uphead_launch_t<sp32_t, true, 8, 8, true>(s, q);
    return uphead_launch_t<float, true, 8, 4, true>(s, q);
}

}  // namespace cf
