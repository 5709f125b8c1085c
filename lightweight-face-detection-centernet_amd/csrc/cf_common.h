// Shared device helpers for the gfx950 CenterFace kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef uint16_t bf16_t;   // raw bfloat16 bits in HBM

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA bf16 operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define CF_WAVE 64

__device__ __forceinline__ float bf16_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ float bf16lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }

// two fp32 -> packed bf16x2, round-to-nearest-even (v_cvt_pk_bf16_f32 has no builtin on gfx950)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Swish (model/centernet.py:34-40): x * sigmoid(x) = x / (1 + exp(-x)); v_exp_f32 + v_rcp_f32
__device__ __forceinline__ float swish_f(float x) {
    return x * fast_rcp(1.0f + __expf(-x));
}
__device__ __forceinline__ float relu_f(float x) { return fmaxf(x, 0.0f); }

// Packed fp32 math on register pairs (v_pk_mul/add/fma_f32).  Measured (profiles/r01_valu_microbench.md):
// v_pk_fma_f32 issues in 5.65 cycles vs 2.88 for v_fma_f32, i.e. the same per-element rate -- what it
// buys is fewer instructions to fetch/issue (5-12 % on the fused kernels).  The cost that matters in
// Swish is the two transcendentals (v_exp_f32, v_rcp_f32: 9.45 cycles each).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 swish2(f32x2 x) {
    const f32x2 t = x * -1.44269504088896341f;                    // v_pk_mul_f32: -x * log2(e)
    f32x2 e; e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
    const f32x2 den = e + 1.0f;                                   // v_pk_add_f32
    f32x2 r; r.x = __builtin_amdgcn_rcpf(den.x); r.y = __builtin_amdgcn_rcpf(den.y);
    return x * r;                                                 // v_pk_mul_f32
}
// Swish with the -log2(e) factor already in the operand (folded into the producing conv's weights): u = -log2(e) x ->
// u / (1 + 2^u) = -log2(e) swish(x): one packed multiply fewer per pair; the consumer's weights carry the leftover -ln 2.
// The bf16 kernels have always done this (cf_mx.h: swish2_pre); the split-mode fp32-tile kernels do since round 4 (PRE = true),
// the exact-fp32 mode keeps swish2 (its arithmetic is what the goldens pin to the last bit).
static constexpr float kCfNegLog2e = -1.44269504088896341f, kCfNegLn2 = -0.69314718055994531f;
template <bool PRE> __device__ __forceinline__ f32x2 swish2_sel(f32x2 x) {
    if constexpr (PRE) {
        f32x2 e; e.x = __builtin_amdgcn_exp2f(x.x); e.y = __builtin_amdgcn_exp2f(x.y);
        const f32x2 den = e + 1.0f;
        f32x2 r; r.x = __builtin_amdgcn_rcpf(den.x); r.y = __builtin_amdgcn_rcpf(den.y);
        return x * r;
    } else {
        return swish2(x);
    }
}
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// activation over a small register array, two lanes' worth per packed instruction (N even)
template <int ACT, int N> __device__ __forceinline__ void act_arr(float* v) {
    if constexpr (ACT == 1) {
#pragma unroll
        for (int e = 0; e < N; e += 2) {
            f32x2 x2; x2.x = v[e]; x2.y = v[e + 1];
            const f32x2 y2 = swish2(x2);
            v[e] = y2.x; v[e + 1] = y2.y;
        }
    } else if constexpr (ACT == 2) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = relu_f(v[e]);
    }
}
template <int ACT> __device__ __forceinline__ float act_f(float x) {
    if constexpr (ACT == 1) return swish_f(x);
    else if constexpr (ACT == 2) return relu_f(x);
    else return x;
}

// element <-> storage traits ------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int PER16 = 4;   // elements per 16-byte chunk
};
template <> struct Elem<bf16_t> {
    static constexpr int PER16 = 8;
};

// Third storage/arithmetic type: fp32 in HBM and LDS like `float`, but every GEMM product runs as a SPLIT-bf16 ("bf16x3")
// product on the bf16 matrix pipe -- x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits kept), and
// w.x ~= w_hi.x_hi + w_lo.x_hi + w_hi.x_lo (the dropped lo.lo term and the rounding of lo are ~2^-17 relative), accumulated
// in fp32 by the MFMA.  The native fp32 MFMA (v_mfma_f32_32x32x2_f32: 64 cycles for 4 Kflop) runs at 1/16 of the bf16 pipe;
// three v_mfma_f32_32x32x8_bf16_1k (32 cycles for 16 Kflop each) do the same product 2.7x faster, three 32x32x16 5.3x.
// Storage, layouts and every non-GEMM operation (Swish, depthwise taps, residual adds, epilogues) are those of `float`.
struct sp32_t { float v; };
template <> struct Elem<sp32_t> {
    static constexpr int PER16 = 4;
};

// unpack one 16-byte chunk to fp32 values
template <typename T> __device__ __forceinline__ void unpack16(const u32x4& c, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4& c, float* f) {
    f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y);
    f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4& c, float* f) {
    f[0] = bf16lo(c.x); f[1] = bf16hi(c.x); f[2] = bf16lo(c.y); f[3] = bf16hi(c.y);
    f[4] = bf16lo(c.z); f[5] = bf16hi(c.z); f[6] = bf16lo(c.w); f[7] = bf16hi(c.w);
}
template <> __device__ __forceinline__ void unpack16<sp32_t>(const u32x4& c, float* f) { unpack16<float>(c, f); }
template <typename T> __device__ __forceinline__ u32x4 pack16(const float* f);
template <> __device__ __forceinline__ u32x4 pack16<float>(const float* f) {
    u32x4 c; c.x = __float_as_uint(f[0]); c.y = __float_as_uint(f[1]);
    c.z = __float_as_uint(f[2]); c.w = __float_as_uint(f[3]); return c;
}
template <> __device__ __forceinline__ u32x4 pack16<sp32_t>(const float* f) { return pack16<float>(f); }
template <> __device__ __forceinline__ u32x4 pack16<bf16_t>(const float* f) {
    u32x4 c; c.x = pack_bf16x2(f[0], f[1]); c.y = pack_bf16x2(f[2], f[3]);
    c.z = pack_bf16x2(f[4], f[5]); c.w = pack_bf16x2(f[6], f[7]); return c;
}

// ---- the three MFMA flavours behind one call: acc[32x32] += W-fragment . X-fragment over one 16-byte operand chunk per lane
// (lane half h supplies k-slot group h: 8 bf16, or 4 fp32 values)
typedef __attribute__((ext_vector_type(8))) __bf16 cf_bf16x8;
typedef __attribute__((ext_vector_type(4))) short cf_s16x4;
// two fp32 -> packed bf16x2 (RNE) through the COMPILER's v_cvt_pk_bf16_f32 selection, not the inline-asm pack_bf16x2 above: the
// packed dwords below are MFMA operands a few instructions later, and hipcc pads no VALU-write -> MFMA-read wait states for a
// register written inside an asm statement (stale operands on part of the waves: profiles/r03_mfma_depthwise.md section 6)
typedef __attribute__((ext_vector_type(2))) __bf16 cf_bf16x2;
__device__ __forceinline__ uint32_t cvt_bf16x2(float lo, float hi) {
    f32x2 v; v.x = lo; v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, cf_bf16x2));
}
// four fp32 values -> (hi, lo) bf16 quads
__device__ __forceinline__ void split4(const u32x4& x, u32x2& hi, u32x2& lo) {
    const float x0 = __uint_as_float(x.x), x1 = __uint_as_float(x.y), x2 = __uint_as_float(x.z), x3 = __uint_as_float(x.w);
    hi.x = cvt_bf16x2(x0, x1); hi.y = cvt_bf16x2(x2, x3);
    lo.x = cvt_bf16x2(x0 - bf16lo(hi.x), x1 - bf16hi(hi.x));
    lo.y = cvt_bf16x2(x2 - bf16lo(hi.y), x3 - bf16hi(hi.y));
}
__device__ __forceinline__ void mma_split_parts(f32x16& acc, const u32x4& w, const u32x2& xh, const u32x2& xl) {
    u32x2 wh, wl; wh.x = w.x; wh.y = w.y; wl.x = w.z; wl.y = w.w;               // host packing: [4 x hi | 4 x lo] (pack_split4)
    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(cf_s16x4, wl), __builtin_bit_cast(cf_s16x4, xh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(cf_s16x4, wh), __builtin_bit_cast(cf_s16x4, xl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(cf_s16x4, wh), __builtin_bit_cast(cf_s16x4, xh), acc, 0, 0, 0);
}
// run2: two consecutive operand chunks (k-steps j, j + 1 of the lane) at once.  bf16 / exact fp32: two run() calls.  Split mode:
// the eight fp32 values of the pair are ONE k = 16 step of v_mfma_f32_32x32x16_bf16 (twice the rate of the k = 8 form): the
// host stores the pair's weights as [8 x hi] in fragment slot j and [8 x lo] in slot j + 1 (split_pairs_inplace); an unpaired
// last chunk keeps the [4 x hi | 4 x lo] form of run().
template <typename T> struct CfMma;
template <> struct CfMma<bf16_t> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, w), __builtin_bit_cast(cf_bf16x8, x), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void run2(f32x16& acc, const u32x4& w0, const u32x4& w1, const u32x4& x0, const u32x4& x1) {
        run(acc, w0, x0); run(acc, w1, x1);
    }
};
template <> struct CfMma<float> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void run2(f32x16& acc, const u32x4& w0, const u32x4& w1, const u32x4& x0, const u32x4& x1) {
        run(acc, w0, x0); run(acc, w1, x1);
    }
};
// the split halves of a chunk pair, computed once and reused against several weight blocks
struct SplitPair { u32x4 hi, lo; };
__device__ __forceinline__ SplitPair split8(const u32x4& x0, const u32x4& x1) {
    u32x2 h0, l0, h1, l1;
    split4(x0, h0, l0); split4(x1, h1, l1);
    SplitPair s;
    s.hi.x = h0.x; s.hi.y = h0.y; s.hi.z = h1.x; s.hi.w = h1.y;
    s.lo.x = l0.x; s.lo.y = l0.y; s.lo.z = l1.x; s.lo.w = l1.y;
    return s;
}
template <> struct CfMma<sp32_t> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& w, const u32x4& x) {
        u32x2 xh, xl; split4(x, xh, xl);
        mma_split_parts(acc, w, xh, xl);
    }
    static __device__ __forceinline__ void run2(f32x16& acc, const u32x4& whi, const u32x4& wlo, const u32x4& x0, const u32x4& x1) {
        const SplitPair s = split8(x0, x1);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, wlo), __builtin_bit_cast(cf_bf16x8, s.hi), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, whi), __builtin_bit_cast(cf_bf16x8, s.lo), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, whi), __builtin_bit_cast(cf_bf16x8, s.hi), acc, 0, 0, 0);
    }
};
// acc += sum over the J operand chunks of a lane: pairs through run2, an odd last chunk through run (straight-line: J, w(j), x(j)
// are compile-time indexable)
template <typename T, int J, typename WF, typename XF>
__device__ __forceinline__ void mma_chain(f32x16& acc, WF w, XF x) {
#pragma unroll
    for (int j = 0; j + 1 < J; j += 2) CfMma<T>::run2(acc, w(j), w(j + 1), x(j), x(j + 1));
    if constexpr (J & 1) CfMma<T>::run(acc, w(J - 1), x(J - 1));
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ u32x4 zero16() { u32x4 z; z.x = z.y = z.z = z.w = 0u; return z; }

// byte offset of the 16-byte chunk `chunk` (8 bf16 channels) of pixel m in pixel-block order [m / 32][nchunks][m % 32][8]
__device__ __forceinline__ size_t blk_off(size_t m, int nchunks, int chunk) {
    return ((m >> 5) * (size_t)nchunks + (size_t)chunk) * 512 + (m & 31) * 16;
}
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8_t;
// First tap of the eight accumulators: A[i] = w . e + 0 as the three-address VOP3P v_dot2_f32_f16 with an inline-constant
// addend.  (The builtin selects the two-address v_dot2c_f32_f16 and has to zero the accumulator with a v_mov first:
// eight extra VALU instructions per chunk.)  Inline asm hides the DOT result hazard from the compiler (a VALU
// instruction of a DIFFERENT opcode -- the accumulating v_dot2c -- may read a DOT result no earlier than three wait
// states after it issued): the block ends with s_nop 2, and within the block every result is independent.
__device__ __forceinline__ void dot8_first(float* A, const u32x8_t& W, const u32x4& E0, const u32x4& E1) {
    asm("v_dot2_f32_f16 %0, %8, %16, 0\n\t"
        "v_dot2_f32_f16 %1, %9, %17, 0\n\t"
        "v_dot2_f32_f16 %2, %10, %18, 0\n\t"
        "v_dot2_f32_f16 %3, %11, %19, 0\n\t"
        "v_dot2_f32_f16 %4, %12, %20, 0\n\t"
        "v_dot2_f32_f16 %5, %13, %21, 0\n\t"
        "v_dot2_f32_f16 %6, %14, %22, 0\n\t"
        "v_dot2_f32_f16 %7, %15, %23, 0\n\t"
        "s_nop 2"
        : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]), "=&v"(A[3]), "=&v"(A[4]), "=&v"(A[5]), "=&v"(A[6]), "=&v"(A[7])
        : "s"(W[0]), "s"(W[1]), "s"(W[2]), "s"(W[3]), "s"(W[4]), "s"(W[5]), "s"(W[6]), "s"(W[7]),
          "v"(E0.x), "v"(E0.y), "v"(E0.z), "v"(E0.w), "v"(E1.x), "v"(E1.y), "v"(E1.z), "v"(E1.w));
}

// Publish LDS-DMA data (__builtin_amdgcn_global_load_lds) to the other waves of the workgroup.  The compiler makes the
// ISSUING wave wait (vmcnt) before its own reads of that LDS range, but those come after the barrier: the s_barrier itself
// gets no vmcnt wait, so another wave can pass it and read the range before the DMA has landed (seen as rare garbage expand
// weights with two contexts sharing the chip).  Every wave drains its own DMAs, THEN the barrier.
__device__ __forceinline__ void cf_sync_lds_dma() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ... keeping the N most recently issued vector-memory operations in flight across the barrier (register prefetches issued
// AFTER the DMAs: vmcnt retires in order, so vmcnt(N) still covers every DMA).  Every wave must really have issued those N.
template <int N> __device__ __forceinline__ void cf_sync_lds_dma_keep() {
    static_assert(N >= 0 && N <= 63, "vmcnt field");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
    __syncthreads();
}

// Lane -> output pixel map of the depthwise phase (pixel-pair tile kernels: cf_mbconv2.hip, cf_stem0.hip).
//
// A lane reads its pixel's pixel-pair rows from the LDS tile with ds_read_b128 at  e * PITCH + const, PITCH / 16 odd,
// so the 16-byte slot it hits in the 256-byte bank row is  (e * odd + const) mod 16.  The LDS serves a wave64
// ds_read_b128 in four groups of sixteen lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS) -- one cycle per group when the sixteen slots differ, one more per extra address on a slot.
// With the row-major pixel order (lane = oy * TOW/2 + ox/2) the tile's row pitch in pairs (9, 10, 18, 34 ...) is not 8
// mod 16 and every group lands two or three lanes on a slot: SQ_LDS_BANK_CONFLICT was 39-63 % of SQ_LDS_IDX_ACTIVE on
// every kernel of this family, and the 5x5 blocks were bound by the LDS, not by the VALU.  Which pixel a lane owns is
// free (the wave's 64 pixels only have to share the x parity), so the map is built here, at compile time: the
// pixels of a parity class are dealt to the lane groups so that the sixteen of a group have distinct  e mod 16.
// entry: bit 15 = no pixel (lane idles on a valid address), bits 6.. = oy, bits 0-5 = ox / 2 (stride 1) or ox (stride 2).
template <int S, int TOH, int TOW, int IWP>
struct LaneMap {
    static constexpr int NPIX = TOH * TOW, PPX = S == 1 ? NPIX / 2 : NPIX, WPP = (PPX + 63) / 64, NSLOT = WPP * 64;
    static constexpr int ROWW = S == 1 ? TOW / 2 : TOW;                     // class pixels per tile row
    static_assert(ROWW <= 64 && TOH <= 256, "entry packing: 6 bits of x, 9 bits of y below the idle flag");
    uint16_t v[NSLOT];
    static constexpr int pair_index(int r) {                                // e of class pixel r (row-major)
        const int oy = r / ROWW, oxh = r - oy * ROWW;
        return S == 1 ? oy * (IWP / 2) + oxh : oy * IWP + oxh;
    }
    constexpr LaneMap() : v() {
        constexpr int NG = NSLOT / 16;
        int bucket[16][PPX] = {}; int cnt[16] = {}, used[16] = {};
        for (int r = 0; r < PPX; ++r) { const int b = pair_index(r) & 15; bucket[b][cnt[b]++] = r; }
        int slot[NSLOT] = {};
        for (int g = 0; g < NG; ++g)
            for (int b = 0; b < 16; ++b) slot[g * 16 + b] = used[b] < cnt[b] ? bucket[b][used[b]++] : -1;
        for (int b = 0; b < 16; ++b)                                        // uneven classes: leftovers fill the holes
            while (used[b] < cnt[b]) {
                int s = 0; while (slot[s] >= 0) ++s;
                slot[s] = bucket[b][used[b]++];
            }
        for (int g = 0; g < NG; ++g)
            for (int j = 0; j < 16; ++j) {
                const int hw = g & 3;                                       // hardware lane group of the wave
                const int base = (hw & 1) ? (j < 8 ? 4 + j : j < 12 ? 16 + (j - 8) : 28 + (j - 12))
                                          : (j < 4 ? j : j < 8 ? 12 + (j - 4) : 20 + (j - 8));
                const int lane = base + (hw >> 1) * 32;
                const int r = slot[g * 16 + j];
                const int rc = r >= 0 ? r : cnt[j] > 0 ? bucket[j][0] : 0;   // idle lane: an address on its own slot
                const int oy = rc / ROWW, oxh = rc - oy * ROWW;
                v[(g >> 2) * 64 + lane] = (uint16_t)((r < 0 ? 0x8000 : 0) | (oy << 6) | oxh);
            }
    }
};

// Lane (0..31 inside a wave half) -> index of the pixel it owns inside a 32-pixel block (fp32-tile kernels: cf_mbconv.hip,
// cf_mbconv4.hip, stem0_kernel).  A lane reads its pixel's 16-byte channel group with ds_read_b128 at pixel * ROWB, ROWB / 16 odd,
// and the hardware serves that instruction in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32): with the row-major
// order (lane = pixel) a group mixes eight pixels of one 16-wide tile row with eight of the next, whose slots collide unless the
// row pitch is a multiple of 16 pixels (SQ_LDS_BANK_CONFLICT was 29-44 % of SQ_LDS_IDX_ACTIVE on these kernels:
// profiles/r04a_pmc_lds_split.txt).  Which pixel a lane owns is free, so each hardware group gets sixteen CONSECUTIVE pixels
// (one tile row when the tile is 16 wide): odd pitch x consecutive pixels = sixteen distinct slots.
__device__ __forceinline__ int lds_group_pixel(int pl) {
    return pl < 4 ? pl : pl < 12 ? pl + 12 : pl < 16 ? pl - 8 : pl < 20 ? pl + 8 : pl < 28 ? pl - 12 : pl;
}

// XCD-aware tile order.  The hardware deals consecutive workgroup ids round-robin onto the 8 XCDs (observed, for
// speed only -- not a contract; MI355X_MICROARCH.md), so spatially adjacent tiles land on different L2s and every
// halo row / shared cache line is fetched once per XCD that touches it.  This remaps the linear workgroup id so that
// each XCD walks a CONTIGUOUS range of tiles (x fastest, then y, then batch): neighbours share an L2.
__device__ __forceinline__ void xcd_tile_order(unsigned& bx, unsigned& by, unsigned& bz) {
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    unsigned l = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if ((n & 7u) == 0u) l = (l & 7u) * (n >> 3) + (l >> 3);
    bx = l % gx; l /= gx; by = l % gy; bz = l / gy;
}

// Run-time switches (host).  cf_env_int: the few PRODUCT switches, each exercised by the driver-run suite
// (tests/test_switches.py): CF_DW_MATRIX, CF_F4_VARIANT, CF_XCD_ORDER, CF_DECODE_OVERLAP.  cf_ab_int: the A/B switches of the
// kernel-variant tables and fusion choices -- they exist only in an experiments build (make EXP=1: -DCF_EXPERIMENTS ->
// libcenterface_hip_exp.so, loaded with CF_LIB=...); the release library returns the default, reads no such variable and
// does not contain the never-default variants.
static inline int cf_env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#ifdef CF_EXPERIMENTS
static inline int cf_ab_int(const char* name, int dflt) { return cf_env_int(name, dflt); }
#else
static inline int cf_ab_int(const char*, int dflt) { return dflt; }
#endif

// host-side fp32 -> bf16 (RNE), identical rounding to the device instruction for finite values
static inline uint16_t host_f32_to_bf16(float f) {
    uint32_t u; __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);   // inf / nan: truncate
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float host_bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16; float f; __builtin_memcpy(&f, &u, 4); return f;
}
// host: four fp32 weights -> one 16-byte split fragment [hi0..3 | lo0..3] (CfMma<sp32_t>)
static inline void pack_split4(const float* src, void* dst16) {
    uint16_t* d = (uint16_t*)dst16;
    for (int e = 0; e < 4; ++e) {
        const uint16_t hi = host_f32_to_bf16(src[e]);
        d[e] = hi;
        d[4 + e] = host_f32_to_bf16(src[e] - host_bf16_to_f32(hi));
    }
}
// host: fragments packed as fp32 ([group][J slots][64 lanes] x 16 B = four fp32 each) -> the split mode's operand format, in
// place: slots (2i, 2i + 1) of a group become [8 x hi] and [8 x lo] of the pair's eight weights (CfMma<sp32_t>::run2), an odd
// last slot becomes [4 x hi | 4 x lo] (run).  The kernel walks the same J slots in the same pairs (mma_chain).
static inline void split_pairs_inplace(void* frags, size_t ngroups, int J) {
    char* base = (char*)frags;
    for (size_t g = 0; g < ngroups; ++g) {
        char* grp = base + g * (size_t)J * 1024;
        for (int j = 0; j + 1 < J; j += 2)
            for (int lane = 0; lane < 64; ++lane) {
                char* c0 = grp + ((size_t)j * 64 + lane) * 16, * c1 = c0 + 1024;
                float v[8];
                __builtin_memcpy(v, c0, 16); __builtin_memcpy(v + 4, c1, 16);
                uint16_t hi[8], lo[8];
                for (int e = 0; e < 8; ++e) { hi[e] = host_f32_to_bf16(v[e]); lo[e] = host_f32_to_bf16(v[e] - host_bf16_to_f32(hi[e])); }
                __builtin_memcpy(c0, hi, 16); __builtin_memcpy(c1, lo, 16);
            }
        if (J & 1)
            for (int lane = 0; lane < 64; ++lane) {
                char* c = grp + ((size_t)(J - 1) * 64 + lane) * 16;
                float v[4]; __builtin_memcpy(v, c, 16);
                pack_split4(v, c);
            }
    }
}
// host: P elements of storage/operand type `dtype` into a 16-byte fragment chunk: 0 fp32 and 2 split (raw fp32 here -- every
// packer of the split mode finishes with split_pairs_inplace over its fragment groups), 1 bf16
static inline void pack_chunk(int dtype, const float* src, void* dst16) {
    if (dtype != 1) __builtin_memcpy(dst16, src, 16);
    else for (int e = 0; e < 8; ++e) ((uint16_t*)dst16)[e] = host_f32_to_bf16(src[e]);
}
