// Expand 1x1 (+Swish) -> depthwise k x k (+Swish) for the fp32-STORAGE modes (the split-bf16 tolerance mode, MbGeom::kind = 8):
// the first two convolutions of MBConvBlock.forward (model/centernet.py:89-140) in one launch, the 6x tensor only in LDS, the
// depthwise output written once to HBM for the project GEMM (pw_wlds_kernel) -- the wide blocks layer4.0 ... 6.0, whose project
// accumulators (pixels x Cout fp32) do not fit next to a useful tile.  Round 4 ran those blocks either fully fused on 8x16 tiles
// (4.x: 1.9x halo recompute, X fragments re-read per chunk with the load latency exposed, two waves per SIMD) or as three
// launches with TWO fp32 6x tensors through HBM (5.x, 6.0): 1.13 ms of the 3.12 ms forward.
//
// What is new against cf_mbconv4.hip's depthwise (one pixel per lane, one ds_read_b128 + four v_fmac_f32 with an SGPR operand
// per tap and 4-channel group):
//   * REGISTER WINDOW: a lane owns a strip of FOUR x-adjacent output pixels of one 4-channel group.  Per kernel row it reads
//     4 + KS - 1 cells (stride 2: 4 + (KS-1)/2 even and 4 + (KS-3)/2 odd columns) instead of 4 KS: 8 reads for 20 taps (5x5),
//     6 for 12 (3x3) -- the LDS traffic of the depthwise, which was half of these kernels' time, falls 2-2.5x;
//   * the tile is stored as X-QUAD CELLS  E[row][quad][4-channel group][4 pixels] x 16 B  (quad pitch = 64 NG + 16 bytes: an odd
//     number of 16-byte slots), so the sixteen lanes of a ds_read_b128 hardware group -- sixteen consecutive strips -- hit
//     sixteen different slots for every cell of the window (a pixel-major tile gives strips four pixels apart only four);
//   * PACKED taps: an SGPR source operand halves the fp32 FMA rate on this chip (profiles/r05_valu_clock_probe.md: 4.15 against
//     2.25 shader cycles), v_pk_fma_f32 does two multiply-adds in 4.3: the taps of a 4-channel group are two register pairs
//     and a tap costs 8 packed FMAs for 4 pixels x 4 channels;
//   * no project accumulators: 10x40 / 20x20 tiles (halo recompute 1.5x / 1.2-1.4x instead of 1.9x), X fragments resident in
//     registers or double-buffered (the next block's loads are in flight under this block's expand), a workgroup walks a range
//     of hidden chunks with the expand weights DMA'd one chunk ahead.
// The depthwise output leaves in PIXEL-BLOCK order [m / 32][hid / 4][m % 32][4 channels] (PwParams::xblock: the project GEMM's
// B operand is then one 512-byte run per wave half and k-step) or NHWC.  Split mode (SP): Swish factors as cf_mbconv.hip
// (-log2 e folded into the expand weights, the leftover -ln 2 multiplied in here, before the store).
#include "cf_exp.h"
#include "cf_common.h"
#include "cf_kernels.h"
#include <cstdlib>
#include <type_traits>

namespace cf {

#define CF_AS4 __attribute__((address_space(4)))

template <int KS, int S, int HC, int TOH, int TOW, int JX, int NW>
struct X5 {
    static_assert(TOW % 4 == 0 && (S == 1 || S == 2) && (KS == 3 || KS == 5), "strip geometry");
    static constexpr int IH = (TOH - 1) * S + KS, IW0 = (TOW - 1) * S + KS;
    // stride 1: IWQ quads of four consecutive columns.  Stride 2: [even columns | odd columns], HWQ quads each.
    static constexpr int HWQ = ((IW0 + 1) / 2 + 3) / 4;
    static constexpr int IWQ = S == 2 ? 2 * HWQ : (IW0 + 3) / 4, IW = 4 * IWQ;
    static constexpr int IPX = IH * IW, NIB = (IPX + 31) / 32, MAXI = (NIB + NW - 1) / NW;
    static constexpr int NG = HC / 4, QSTRIDE = NG * 64 + 16;
    static constexpr int SPR = TOW / 4, NSTRIP = TOH * SPR, NSG = (NSTRIP + 63) / 64, UNITS = NSG * NG;
    static constexpr int NBE = (HC + 31) / 32;
    static constexpr bool PART = (HC % 32 == 16);
    static constexpr int WXB = NBE * JX * 1024;
    static constexpr int EBYTES = IH * IWQ * QSTRIDE;
    static constexpr int TAPB = (NG * KS * KS * 16 + 1023) / 1024 * 1024;     // the chunk's taps [group][tap][4] x fp32, in whole 1 KiB DMA pieces
    // E | expand weights of ONE chunk (the next chunk's are DMA'd behind the phase-1 barrier, under the depthwise) | taps x 2
    static constexpr int LDS = EBYTES + WXB + 2 * TAPB;
    // cells of a strip's window per kernel row
    static constexpr int NE = S == 1 ? 4 + KS - 1 : 4 + (KS - 1) / 2;      // consecutive (stride 1) / even columns
    static constexpr int NO = S == 1 ? 0 : 4 + (KS - 3) / 2;               // odd columns
    static_assert(HC % 8 == 0, "hidden chunk geometry");
};

// RES: the X fragments of the wave's blocks stay in registers for all chunks, ALREADY SPLIT into bf16 (hi, lo) pairs in the split mode
// (re-read and re-split per chunk otherwise: 144 of the ~250 VALU instructions of a block at Cin = 96); costs the second workgroup
// per CU (256-register budget instead of 128)
// MW: waves per SIMD the register budget is set for (4: 128 registers, two 8-wave workgroups per CU; 2: 256 registers)
template <int KS, int S, int HC, int TOH, int TOW, int JX, int NW, bool PIPE, bool SP, bool RES = false, int MW = 4>
__global__ __launch_bounds__(NW * 64, MW) void expdw_f32_kernel(MbParams p) {
    typedef typename std::conditional<SP, sp32_t, float>::type MT;
    typedef X5<KS, S, HC, TOH, TOW, JX, NW> G;
    constexpr int IW = G::IW, IWQ = G::IWQ, HWQ = G::HWQ, IPX = G::IPX, NIB = G::NIB, MAXI = G::MAXI, NG = G::NG, NBE = G::NBE;
    constexpr int QSTRIDE = G::QSTRIDE, WXB = G::WXB, TAPB = G::TAPB, SPR = G::SPR, NSTRIP = G::NSTRIP, NSG = G::NSG, UNITS = G::UNITS;
    constexpr bool PART = G::PART;
    static_assert(RES || NW <= NIB, "re-read mode: every wave owns a first block");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* E = smem;
    char* Wst = smem + G::EBYTES;
    char* Tap = Wst + WXB;

    const int tid = threadIdx.x;
    int lane = tid & 63;                                            // (not const: re-"defined" per chunk, see the chunk loop)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pl = lane & 31, h = lane >> 5;
    const int ntx = (p.Wout + TOW - 1) / TOW;
    const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx;
    const int ox0 = tx * TOW, oy0 = ty * TOH, b = blockIdx.z;
    const int q0 = blockIdx.y * p.HALF, q1 = min(q0 + p.HALF, p.nq);          // MbParams::HALF: hidden chunks per workgroup (this kernel)

    const char* xbase = (const char*)p.x + (size_t)b * p.Hin * p.Win * p.Cin * 4;
    const unsigned rowbytes = (unsigned)p.Cin * 4;

    // expand fragments of chunk q -> Wst, its depthwise taps -> Tap[(q - q0) & 1]: LDS DMA, 1 KiB per wave instruction.  The tap
    // table lives in LDS (broadcast ds_read_b128, VGPR operands of the packed FMAs) rather than in SGPRs: scalar loads share the
    // lgkmcnt counter with the LDS reads and return out of order, so a row's FMAs would wait for EVERYTHING in flight.
    // The DMA is issued through inline asm: hipcc orders every later LDS read behind a __builtin_amdgcn_global_load_lds with a
    // full s_waitcnt vmcnt(0) (it cannot tell the DMA's LDS range from the tile's), which parked each wave at the head of the
    // depthwise phase until the next chunk's weights had arrived (phase stamps: tools/x5_timing.py).  The ranges are disjoint by
    // construction; cf_sync_lds_dma() publishes the DMA'd range at the top of the next chunk.
    auto stage_weights = [&](int q) {
        const char* srcx = (const char*)p.wexp + (size_t)q * WXB;
        const char* srct = (const char*)p.wdw + (size_t)q * (NG * KS * KS * 16);
        const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
        const unsigned wst = lds0 + G::EBYTES, tdst = wst + WXB + ((q - q0) & 1) * TAPB;
        for (int c = wave; c < (WXB + TAPB) / 1024; c += NW) {
            const bool isw = c < WXB / 1024;
            const char* src = (isw ? srcx + c * 1024 : srct + (c - WXB / 1024) * 1024) + lane * 16;
            const unsigned dst = __builtin_amdgcn_readfirstlane(isw ? wst + c * 1024 : tdst + (c - WXB / 1024) * 1024);
            unsigned m0save;                                       // m0 is the compiler's: saved and restored around the DMA
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0" : "=&s"(m0save) : "s"(dst), "v"(src));
        }
    };

    // tile position ip (row-major over IH x IW tile columns) -> its image coordinates.  The loads are issued from a CLAMPED address and
    // not touched here (a select on the loaded value would wait for it at once): a position outside the image -- ZeroPad2d -- or past
    // the tile zeroes the block's OUTPUT instead (expand has no bias: the expanded value of a zero pixel is Swish(0) = 0).
    // The K chunks of a lane half are walked in NPASS passes of JP (wide Cin: the fragment buffer is JP x 4 registers).
    constexpr int NPASS = JX > 12 ? 2 : 1, JP = JX / NPASS;
    static_assert(JP * NPASS == JX && (NPASS == 1 || JP % 2 == 0), "K passes");
    const char* xbase_blk = nullptr; unsigned xoff = 0; bool xvalid = false;      // address of the block whose fragments are being loaded
    auto block_addr = [&](int ib) {
        const int ip = ib * 32 + pl;
        const int ipc = ip < IPX ? ip : IPX - 1;
        const int iy = ipc / IW, xp = ipc - iy * IW;
        const int ix = S == 2 ? (xp < 4 * HWQ ? 2 * xp : 2 * (xp - 4 * HWQ) + 1) : xp;
        const int gy = oy0 * S - p.pad_lo + iy, gx = ox0 * S - p.pad_lo + ix;
        xvalid = ip < IPX && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
        const int cy = min(max(gy, 0), p.Hin - 1), cx = min(max(gx, 0), p.Win - 1);
        if (p.xblock) {
            // x in pixel-block order [m / 32][Cin / 4][m % 32][4]: consecutive tile columns = consecutive 16-byte chunks
            xbase_blk = (const char*)p.x + blk_off(((size_t)b * p.Hin + cy) * p.Win + cx, p.Cin / 4, h * JX);
        } else {
            xoff = ((unsigned)cy * (unsigned)p.Win + (unsigned)cx) * rowbytes + (unsigned)(h * JX * 16);
        }
    };
    auto load_pass = [&](int pass, u32x4* dst) {
        if (p.xblock) {
#pragma unroll
            for (int j = 0; j < JP; ++j) dst[j] = ld16(xbase_blk + (pass * JP + j) * 512);
        } else {
#pragma unroll
            for (int j = 0; j < JP; ++j) dst[j] = ld16(xbase + xoff + (pass * JP + j) * 16);
        }
    };
    // ... Swish + the stores into the x-quad cells of E
    auto expand_store = [&](int ib, const f32x16* a, bool valid) {
        const int ip = ib * 32 + pl;
        const bool ipok = ip < IPX;
        const int ipc = ipok ? ip : 0;
        const int iy = ipc / IW, xp = ipc - iy * IW;
        char* ecell = E + (unsigned)(iy * IWQ + (xp >> 2)) * (unsigned)QSTRIDE + (unsigned)(xp & 3) * 16u;
#pragma unroll
        for (int nbl = 0; nbl < NBE; ++nbl) {
            const bool half_block = PART && nbl == NBE - 1;       // 8 channels on each lane half (mb_pack_weights)
            const int ch0 = half_block ? nbl * 32 + h * 8 : nbl * 32 + h * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (half_block && g >= 2) break;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    f32x2 x2; x2.x = a[nbl][g * 4 + e]; x2.y = a[nbl][g * 4 + e + 1];
                    const f32x2 y2 = swish2_sel<SP>(x2);
                    v[e] = valid ? y2.x : 0.0f; v[e + 1] = valid ? y2.y : 0.0f;
                }
                if (ipok) st16(ecell + (ch0 / 4 + g) * 64, pack16<float>(v));
            }
        }
    };

    // X fragments: ONE buffer of JP chunks, refilled with the next pass's / next block's loads as soon as the MFMA chain has consumed
    // it -- the loads are in flight under the Swish / store half of the block
    u32x4 xf[JP];
    u32x4 xs[RES ? MAXI : 1][RES ? JX : 1];
    bool xsv[RES ? MAXI : 1];
    if constexpr (RES) {
        static_assert(JX % 2 == 0 && JP % 2 == 0, "resident fragments are kept as chunk pairs");
#pragma unroll
        for (int t = 0; t < MAXI; ++t) {
            const int ib = wave + NW * t;
            xsv[t] = false;
            if (ib < NIB) {
                block_addr(ib);
                xsv[t] = xvalid;
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    load_pass(pass, xf);
#pragma unroll
                    for (int j = 0; j < JP; j += 2) {
                        if constexpr (SP) { const SplitPair sp2 = split8(xf[j], xf[j + 1]); xs[t][pass * JP + j] = sp2.hi; xs[t][pass * JP + j + 1] = sp2.lo; }
                        else { xs[t][pass * JP + j] = xf[j]; xs[t][pass * JP + j + 1] = xf[j + 1]; }
                    }
                }
            }
        }
    } else {
        block_addr(wave);
        load_pass(0, xf);
    }

    const int NC = p.hid / 4;                                                    // 16-byte chunks of a depthwise-output pixel
#ifdef CF_X5_TIMING      // phase stamps (s_memtime), summed over the chunks of a wave: tools/x5_timing.py
    unsigned long long tph[5] = {0, 0, 0, 0, 0}, tq = 0;
#define X5_STAMP(k) { unsigned long long t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tph[k] += t_ - tq; tq = t_; }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tq) :: "memory");
#else
#define X5_STAMP(k)
#endif
    stage_weights(q0);
    for (int q = q0; q < q1; ++q) {
        const char* wx = Wst;
        const char* tapq = Tap + ((q - q0) & 1) * TAPB;
        cf_sync_lds_dma();            // previous chunk's depthwise is done with E; this chunk's expand weights have landed
        X5_STAMP(0)
        // Everything below that depends only on the lane (tile coordinates, LDS and HBM addresses of the wave's blocks and strips) is
        // loop-invariant, and hipcc hoists all of it out of the chunk loop -- 30-60 live registers, spilled to scratch at the
        // 128-register budget, with a vmcnt(0) wait at every reload.  It costs a few dozen integer instructions per chunk to recompute:
        // the lane id is made opaque once per chunk.
        asm volatile("" : "+v"(lane), "+v"(pl), "+v"(h));

        // ---- phase 1: expand + Swish -> E (x-quad cells)
#pragma unroll
        for (int t = 0; t < MAXI; ++t) {
            const int ib = wave + NW * t;
            if (ib >= NIB) break;
            f32x16 a[NBE];
#pragma unroll
            for (int nbl = 0; nbl < NBE; ++nbl)
#pragma unroll
                for (int r = 0; r < 16; ++r) a[nbl][r] = 0.0f;
            if constexpr (RES) {
#pragma unroll
                for (int nbl = 0; nbl < NBE; ++nbl) {
                    const char* wb = wx + (nbl * JX * 64 + lane) * 16;
                    if constexpr (SP) {
#pragma unroll
                        for (int j = 0; j < JX; j += 2) {
                            const u32x4 whi = ld16(wb + j * 1024), wlo = ld16(wb + (j + 1) * 1024);
                            a[nbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, wlo), __builtin_bit_cast(cf_bf16x8, xs[t][j]), a[nbl], 0, 0, 0);
                            a[nbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, whi), __builtin_bit_cast(cf_bf16x8, xs[t][j + 1]), a[nbl], 0, 0, 0);
                            a[nbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, whi), __builtin_bit_cast(cf_bf16x8, xs[t][j]), a[nbl], 0, 0, 0);
                        }
                    } else {
                        mma_chain<MT, JX>(a[nbl], [&](int j) { return ld16(wb + j * 1024); }, [&](int j) { return xs[t][j]; });
                    }
                }
                expand_store(ib, a, xsv[t]);
                continue;
            }
            const bool vcur = xvalid;
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
                for (int nbl = 0; nbl < NBE; ++nbl) {
                    const char* wb = wx + ((nbl * JX + pass * JP) * 64 + lane) * 16;
                    mma_chain<MT, JP>(a[nbl], [&](int j) { return ld16(wb + j * 1024); }, [&](int j) { return xf[j]; });
                }
                if (pass + 1 < NPASS) load_pass(pass + 1, xf);
                else if (ib + NW < NIB) { block_addr(ib + NW); load_pass(0, xf); }   // (the next chunk's first block: behind phase 2)
            }
            expand_store(ib, a, vcur);
        }
        X5_STAMP(1)
        __syncthreads();
        X5_STAMP(2)
        if (q + 1 < q1) stage_weights(q + 1);

        // ---- phase 2: depthwise + Swish, a strip of four output pixels x four channels per lane, -> HBM
        for (int u = wave; u < UNITS; u += NW) {
            const int sg = u % NSG, g = u / NSG;                                 // wave-uniform: strip group, channel group
            const int v = (lane & 32) + lds_group_pixel(lane & 31);              // hardware read groups = sixteen consecutive strips
            const int st = sg * 64 + v;
            const bool sok = st < NSTRIP;
            const int stc = sok ? st : NSTRIP - 1;
            const int oy = stc / SPR, sx = stc - oy * SPR;
            const char* wq = tapq + g * (KS * KS * 16);                          // wave-uniform address: a broadcast read
            const char* eb = E + (unsigned)((oy * S) * IWQ + sx) * (unsigned)QSTRIDE + (unsigned)g * 64u;
            f32x2 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[i][0].x = acc[i][0].y = 0.0f; acc[i][1].x = acc[i][1].y = 0.0f; }
            // One kernel row at a time: the row's window cells and taps are requested, then consumed.  PIPE: the reads of row ky + 1
            // are in flight under the FMAs of row ky (twice the window registers); without it the other waves of the SIMD (four at the
            // 128-register budget) cover the LDS latency.  __builtin_amdgcn_sched_barrier pins the order -- left alone, the machine
            // scheduler hoists the reads of ALL rows to the top (160 window registers) -- and LDS reads return in order, so the wait
            // in front of a row's FMAs is a partial lgkmcnt.
            constexpr int NBUF = PIPE ? 2 : 1;
            u32x4 ce[NBUF][G::NE], co[NBUF][G::NO > 0 ? G::NO : 1];
            u32x4 wr[NBUF][KS];
            auto fetch_row = [&](int ky, int buf) {
                const char* er = eb + (unsigned)(ky * IWQ) * (unsigned)QSTRIDE;
#pragma unroll
                for (int j = 0; j < G::NE; ++j) ce[buf][j] = ld16(er + (j >> 2) * QSTRIDE + (j & 3) * 16);
                if constexpr (S == 2) {
#pragma unroll
                    for (int j = 0; j < G::NO; ++j) co[buf][j] = ld16(er + (HWQ + (j >> 2)) * QSTRIDE + (j & 3) * 16);
                }
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) wr[buf][kx] = ld16(wq + (ky * KS + kx) * 16);
            };
            if constexpr (PIPE) fetch_row(0, 0);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int cb = PIPE ? (ky & 1) : 0;
                if constexpr (PIPE) { if (ky + 1 < KS) fetch_row(ky + 1, cb ^ 1); }
                else fetch_row(ky, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const u32x4 w = wr[cb][kx];
                    f32x2 w01, w23; w01.x = __uint_as_float(w.x); w01.y = __uint_as_float(w.y); w23.x = __uint_as_float(w.z); w23.y = __uint_as_float(w.w);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4& c = S == 1 ? ce[cb][i + kx] : ((kx & 1) ? co[cb][i + (kx >> 1)] : ce[cb][i + (kx >> 1)]);
                        f32x2 e01, e23; e01.x = __uint_as_float(c.x); e01.y = __uint_as_float(c.y); e23.x = __uint_as_float(c.z); e23.y = __uint_as_float(c.w);
                        acc[i][0] = fma2(e01, w01, acc[i][0]);
                        acc[i][1] = fma2(e23, w23, acc[i][1]);
                    }
                }
                // the sums are only used by the (lane-conditional) stores below: without this pin LLVM SINKS all the FMAs of all rows
                // under that condition, behind the reads of every row
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]));
                __builtin_amdgcn_sched_barrier(0);
            }
            const int gy = oy0 + oy, gx0 = ox0 + sx * 4;
            if (!sok || gy >= p.Hout) continue;
            // byte offset of pixel m0's chunk (32-bit: the depthwise tensor of a batch stays far below 4 GB); computed HERE, per chunk, from
            // a value the compiler cannot hoist -- hoisted out of the chunk loop the eight 64-bit addresses were spilled to scratch
            unsigned m0 = ((unsigned)b * (unsigned)p.Hout + (unsigned)gy) * (unsigned)p.Wout + (unsigned)gx0;
            asm volatile("" : "+v"(m0));
            const unsigned cg = (unsigned)(q * NG + g);                          // 16-byte chunk of the output pixel
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 y0 = swish2_sel<SP>(acc[i][0]), y1 = swish2_sel<SP>(acc[i][1]);
                if constexpr (SP) { y0 = y0 * kCfNegLn2; y1 = y1 * kCfNegLn2; }  // -log2(e) swish -> swish: the project GEMM takes plain weights
                if (gx0 + i >= p.Wout) break;
                float vv[4] = {y0.x, y0.y, y1.x, y1.y};
                const unsigned m = m0 + i;
                const unsigned off = p.yblock ? ((m >> 5) * (unsigned)NC + cg) * 512u + (m & 31u) * 16u : (m * (unsigned)p.hid + cg * 4u) * 4u;
                st16((char*)p.y + off, pack16<float>(vv));
            }
        }
        X5_STAMP(3)
        if constexpr (!RES) { if (q + 1 < q1) { block_addr(wave); load_pass(0, xf); } }   // in flight across the barrier at the top of the next chunk
    }
#ifdef CF_X5_TIMING
    if (p.wproj && lane == 0) {
        unsigned long long* o = (unsigned long long*)p.wproj + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + wave) * 4;
        o[0] = tph[0]; o[1] = tph[1]; o[2] = tph[2]; o[3] = tph[3];
    }
#endif
}

// ---------------------------------------------------------------- host side
struct X5Entry {
    int var, k, s, jx, hc, toh, tow, qpw;     // var: CF_X5_VARIANT (experiments build); qpw: hidden chunks per workgroup
    int lds_bytes;
    hipError_t (*fn)(hipStream_t, const MbParams&);
    hipError_t (*fn_sp)(hipStream_t, const MbParams&);
};
template <int KS, int S, int HC, int TOH, int TOW, int JX, int NW, bool PIPE, bool SP, bool RES, int MW>
static hipError_t x5_launch_t(hipStream_t s, const MbParams& p) {
    typedef X5<KS, S, HC, TOH, TOW, JX, NW> G;
    auto kfn = expdw_f32_kernel<KS, S, HC, TOH, TOW, JX, NW, PIPE, SP, RES, MW>;
    static thread_local bool configured_dev[32] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    bool& configured = configured_dev[dev & 31];
    if (G::LDS > 64 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int ntx = (p.Wout + TOW - 1) / TOW, nty = (p.Hout + TOH - 1) / TOH;
    dim3 grid(ntx * nty, (p.nq + p.HALF - 1) / p.HALF, p.B), blk(NW * 64);
    set_kernel_tag("void cf::expdw_f32_kernel<%d, %d, %d, %d, %d, %d, %d, %s, %s, %s, %d>(cf::MbParams)", KS, S, HC, TOH, TOW, JX, NW, PIPE ? "true" : "false", SP ? "true" : "false",
                   RES ? "true" : "false", MW);
    hipLaunchKernelGGL(kfn, grid, blk, G::LDS, s, p);
    return hipGetLastError();
}
#define X5M(V, KS, S, JX, HC, TOH, TOW, NW, PIPE, RES, MW, QPW) \
    {V, KS, S, JX, HC, TOH, TOW, QPW, X5<KS, S, HC, TOH, TOW, JX, NW>::LDS, &x5_launch_t<KS, S, HC, TOH, TOW, JX, NW, (PIPE != 0), false, (RES != 0), MW>, \
     &x5_launch_t<KS, S, HC, TOH, TOW, JX, NW, (PIPE != 0), true, (RES != 0), MW>}
#define X5E(V, KS, S, JX, HC, TOH, TOW, NW, PIPE, RES, QPW) X5M(V, KS, S, JX, HC, TOH, TOW, NW, PIPE, RES, ((RES) ? 2 : 4), QPW)
static const X5Entry kX5Table[] = {
    // 10x20 tiles (E = 38-44 KB, two 8-wave workgroups per CU); round-5 sweep: profiles/r05_expdw_f32.md
    // Sweep of the round (gpurun_out/r05j/x5_variants*.txt -> profiles/r05_split_restructure.md; B = 64, 640x640, ms, one box;
    // rows: the default of every block, CF_X5_VARIANT = 1 ... 6):
    //            4.0     4.1     5.0     5.1     6.0
    //   two workgroups per CU, 10x20 tiles, X re-read and re-split per chunk              0.116   0.198   0.205   0.135   0.119
    //   1: the same tiles, X fragments resident and pre-split, one workgroup per CU       0.134   0.208   0.219   0.141   0.104
    //   2: 10x40 / 20x20 tiles, resident, every chunk in one workgroup                    0.110   0.176   0.135   0.348   0.252
    //   3: as 1 with every chunk in one workgroup                                         0.127   0.199   0.204   0.179   0.135
    //   4: hidden chunks of 64, re-read X                                                 0.125   0.211   0.161   0.195   0.167
    //   5 / 6: big tiles, re-read X, 18 / 6 chunks per workgroup                          0.126 / 0.122   0.213 / 0.205   0.162 / 0.150   0.306 / 0.143   0.259 / 0.125
    // Resident, pre-split X fragments win wherever they fit without spills (4.x on 10x40 tiles, 6.0), layer5.0 wants hidden chunks of
    // 32 on one workgroup per CU (its stride-2 tile is 82 KB), layer5.1 (Cin = 160: 80 fragment registers per block) fits them with
    // twelve waves per workgroup (one halo block per wave).
    // var KS S JX  HC  tile    waves row-pipe resident-X chunks/workgroup
    X5E(0, 5, 1, 8, 32, 10, 40, 8, 1, 1, 12),     // 4.0   64 -> 384, 40x40
    X5E(0, 5, 1, 12, 32, 10, 40, 8, 0, 1, 18),    // 4.1   96 -> 576, 40x40
    X5E(0, 5, 2, 12, 32, 5, 20, 8, 0, 1, 18),     // 5.0   96 -> 576, 40x40 -> 20x20
    X5M(0, 5, 1, 20, 32, 10, 20, 12, 0, 1, 3, 10), // 5.1  160 -> 960, 20x20: twelve waves = one halo block each, its 80 fragment registers resident (0.134 -> 0.119)
    X5E(0, 3, 1, 20, 32, 10, 20, 8, 1, 1, 10),    // 6.0  160 -> 960, 20x20
#include CF_EXP_INC(cf_mbconv5_0)
};
#undef X5E
#undef X5M

static const X5Entry* x5_find(int k, int s, int jx) {
    static const int want = cf_ab_int("CF_X5_VARIANT", 0);
    const X5Entry* base = nullptr;
    for (const X5Entry& e : kX5Table)
        if (e.k == k && e.s == s && e.jx == jx) {
            if (e.var == want) return &e;
            if (e.var == 0) base = &e;
        }
    return base;
}

// geometry of the fp32-storage expand+depthwise kernel for a block: MbGeom with kind = 8 (HALF = hidden chunks per workgroup)
MbGeom expdw_f32_geometry(int dtype, int Cin, int hid, int k, int s) {
    MbGeom g{};
    static const bool on = cf_ab_int("CF_X5", 1) != 0;      // A/B: 0 = round 4's fused 4.x / three-launch 5.x, 6.0
    if (!on || dtype != 2 || (Cin % 8) || hid == Cin || (hid % 4)) return g;
    const int jx = (Cin * 4 / 16 + 1) / 2;
    const X5Entry* e = x5_find(k, s, jx);
    if (!e || hid % e->hc) return g;
    g.ok = true; g.kind = 8; g.S = s;
    g.JX = jx; g.NBO = 0; g.HC = e->hc; g.nq = hid / e->hc; g.NBE = (e->hc + 31) / 32; g.HALF = e->qpw;
    g.rowb = 0; g.KG = 1;
    g.lds_bytes = (size_t)e->lds_bytes;
    g.wexp_bytes = (size_t)g.nq * g.NBE * g.JX * 64 * 16;
    g.wdw_floats = (size_t)g.nq * k * k * g.HC + 256;     // + 1 KiB: the tap DMA of the last chunk reads whole 1 KiB pieces
    g.wproj_bytes = 0;
    return g;
}

hipError_t expdw_f32_launch(hipStream_t s, int dtype, const MbParams& p) {
    const X5Entry* e = x5_find(p.k, p.s, p.JX);
    if (!e || e->hc != p.HC || dtype == 1) return hipErrorInvalidValue;
    return dtype == 2 ? e->fn_sp(s, p) : e->fn(s, p);
}

}  // namespace cf
