// common.h -- shared device/host helpers for the LSeg gfx950 engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace lseg {

// ----- element types ------------------------------------------------------------------
// Activations/weights travel as raw 16-bit words; the tag types pick the MFMA flavour.
struct BF16 { static constexpr int code = 2; };
struct F16  { static constexpr int code = 1; };

typedef __attribute__((ext_vector_type(8))) __bf16   bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float    f32x4_t;
typedef __attribute__((ext_vector_type(16))) float   f32x16_t;
typedef __attribute__((ext_vector_type(4))) int      i32x4_t;   // a 16-byte MFMA operand fragment

enum DType { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2, DT_I64 = 3 };

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {      // round-to-nearest-even
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float f16_to_f32(uint16_t v) {
    _Float16 h = __builtin_bit_cast(_Float16, v);
    return (float)h;
}
__device__ __forceinline__ uint16_t f32_to_f16(float f) {       // v_cvt_f16_f32, RNE
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float round_f16(float f) { return (float)((_Float16)f); }
// (1-ly)*((1-lx)*a + lx*b) + ly*((1-lx)*c + lx*d) with the fused multiply-adds spelled out: every kernel that evaluates output_conv's
// bilinear (materialised logits, masks / metrics read through it on the fly, the one-pass x4 upsample) gets the same bits whatever
// the compiler would have contracted in its context
// source taps of output index i under align_corners=True (r = (n_in - 1) / (n_out - 1)): i0 = floor(r i), i1 = min(i0 + 1, n - 1),
// l = r i - i0 -- with the product ROUNDED before the subtraction in every kernel (no contraction into an fma here)
__device__ __forceinline__ void src_tap(float r, int i, int n, int& i0, int& i1, float& l) {
#pragma clang fp contract(off)
    const float s = r * (float)i;
    i0 = (int)s;
    i1 = i0 + (i0 < n - 1);
    l = s - (float)i0;
}
__device__ __forceinline__ float bilerp(float a, float b, float c, float d, float lx, float ly) {
    const float wx = 1.f - lx, wy = 1.f - ly;
    const float h0 = __builtin_fmaf(lx, b, wx * a), h1 = __builtin_fmaf(lx, d, wx * c);
    return __builtin_fmaf(ly, h1, wy * h0);
}

template <typename T> __device__ __forceinline__ float to_f32(uint16_t v);
template <> __device__ __forceinline__ float to_f32<BF16>(uint16_t v) { return bf16_to_f32(v); }
template <> __device__ __forceinline__ float to_f32<F16>(uint16_t v) { return f16_to_f32(v); }
template <typename T> __device__ __forceinline__ uint16_t from_f32(float f);
template <> __device__ __forceinline__ uint16_t from_f32<BF16>(float f) { return f32_to_bf16(f); }
template <> __device__ __forceinline__ uint16_t from_f32<F16>(float f) { return f32_to_f16(f); }

// Two fp32 -> one 32-bit word of two 16-bit floats (lo in bits 0..15), round-to-nearest-even.
// bf16: gfx950's v_cvt_pk_bf16_f32 (no builtin; cdna_hip_programming.md T12), 1 instruction
// instead of the ~10 of the bit-twiddling form.
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<BF16>(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
template <> __device__ __forceinline__ uint32_t pack2<F16>(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    h2 v; v[0] = (_Float16)lo; v[1] = (_Float16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t pack2_dt(float lo, float hi, int dtype) {
    return dtype == DT_F16 ? pack2<F16>(lo, hi) : pack2<BF16>(lo, hi);
}

__device__ __forceinline__ float load_as_f32(const void* p, size_t i, int dtype) {
    if (dtype == DT_F32) return ((const float*)p)[i];
    if (dtype == DT_F16) return f16_to_f32(((const uint16_t*)p)[i]);
    return bf16_to_f32(((const uint16_t*)p)[i]);
}
__device__ __forceinline__ void store_from_f32(void* p, size_t i, int dtype, float v) {
    if (dtype == DT_F32) ((float*)p)[i] = v;
    else if (dtype == DT_F16) ((uint16_t*)p)[i] = f32_to_f16(v);
    else ((uint16_t*)p)[i] = f32_to_bf16(v);
}

// ----- MFMA wrappers ---------------------------------------------------------------------
// v_mfma_f32_16x16x32_{bf16,f16}:  D[16x16] += A[16x32] * B[32x16]
//   A operand: lane l holds row (l&15), k = (l>>4)*8 .. +7      (8 contiguous k)
//   B operand: lane l holds col (l&15), k = (l>>4)*8 .. +7
//   D: lane l, reg r  ->  row (l>>4)*4 + r, col (l&15)
template <typename T> __device__ __forceinline__ f32x4_t mfma16(i32x4_t a, i32x4_t b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t mfma16<BF16>(i32x4_t a, i32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma16<F16>(i32x4_t a, i32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// v_mfma_f32_32x32x16_{bf16,f16}:  D[32x32] += A[32x16] * B[16x32]
//   A: lane l holds row (l&31), k = (l>>5)*8 .. +7 ; B: col (l&31), same k
//   D: lane l, reg r -> row (r&3) + 8*(r>>2) + 4*(l>>5), col (l&31)
template <typename T> __device__ __forceinline__ f32x16_t mfma32(i32x4_t a, i32x4_t b, f32x16_t c);
template <> __device__ __forceinline__ f32x16_t mfma32<BF16>(i32x4_t a, i32x4_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16_t mfma32<F16>(i32x4_t a, i32x4_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// ----- LDS tile convention ---------------------------------------------------------------
// Every MFMA operand tile in LDS is [rows][64 x 16-bit] = 128-byte rows, i.e. 8 chunks of
// 16 B per row.  It is filled by direct-to-LDS loads (global_load_lds_dwordx4: the LDS
// destination is wave-uniform base + lane*16, so one wave instruction lands 8 rows x 128 B
// linearly) and the XOR swizzle therefore lives on the SOURCE address: LDS slot s of row r
// holds logical chunk  s ^ swz(r),  swz(r) = (r>>1)&7.  Fragment reads apply the same XOR.
// With that swizzle both the 16-row (16x16x32) and the 32-row (32x32x16) ds_read_b128
// fragment patterns are bank-conflict free (see DESIGN.md "LDS layout").
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// One wave loads an 8-row x 128-byte slab: lane l -> row l>>3, LDS slot l&7.
// `src_row` must point at the start (logical chunk 0) of the row this lane serves.
__device__ __forceinline__ void glds_slab_row(const uint16_t* src_row, int row_in_tile, int lane,
                                              char* lds_slab /* wave-uniform */) {
    const int chunk = (lane & 7) ^ swz(row_in_tile);
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(src_row + chunk * 8),
        (__attribute__((address_space(3))) void*)lds_slab, 16, 0, 0);
}
// A wave-uniform pointer forced into SGPRs (v_readfirstlane is free when the value already is scalar): keeps address arithmetic on it out
// of the compiler's per-lane strength reduction.
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
// Same, with a wave-uniform base pointer and a 32-bit per-lane byte offset (lets the compiler use
// the SGPR-base + VGPR-offset addressing form: no 64-bit VALU adds in the K-loop).
__device__ __forceinline__ void glds_slab_off(const char* base_uniform, uint32_t lane_byte_off,
                                              char* lds_slab /* wave-uniform */) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(base_uniform + lane_byte_off),
        (__attribute__((address_space(3))) void*)lds_slab, 16, 0, 0);
}
// Byte offset of logical 16-byte chunk `c` of row `r` inside a swizzled tile.
__device__ __forceinline__ int tile_off(int r, int c) { return r * 128 + ((c ^ swz(r)) << 4); }

// ----- host-side error plumbing ------------------------------------------------------------
#define LSEG_HIP_TRY(expr)                                                         \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) return lseg::set_error_hip(_e, #expr, __FILE__, __LINE__); \
    } while (0)

int set_error(int code, const char* fmt, ...);
int set_error_hip(hipError_t e, const char* what, const char* file, int line);
const char* last_error();

}  // namespace lseg
