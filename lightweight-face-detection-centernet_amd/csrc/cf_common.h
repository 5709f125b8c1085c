// Shared device helpers for the gfx950 CenterFace kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
template <typename T> __device__ __forceinline__ u32x4 pack16(const float* f);
template <> __device__ __forceinline__ u32x4 pack16<float>(const float* f) {
    u32x4 c; c.x = __float_as_uint(f[0]); c.y = __float_as_uint(f[1]);
    c.z = __float_as_uint(f[2]); c.w = __float_as_uint(f[3]); return c;
}
template <> __device__ __forceinline__ u32x4 pack16<bf16_t>(const float* f) {
    u32x4 c; c.x = pack_bf16x2(f[0], f[1]); c.y = pack_bf16x2(f[2], f[3]);
    c.z = pack_bf16x2(f[4], f[5]); c.w = pack_bf16x2(f[6], f[7]); return c;
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ u32x4 zero16() { u32x4 z; z.x = z.y = z.z = z.w = 0u; return z; }

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
