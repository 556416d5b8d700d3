// Shared host/device helpers of liblade_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lade_hip.h"

namespace lade {

// ---- host: error reporting (thread local, read through lade_last_error_string) --------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// one experiment switch of LADE_DEBUG=name[=value],name2[=value2],... (kernel ablation bits and forced launch shapes: tools/ only) as an int;
// 0 when the variable or the name is absent, 1 for a bare name.  Callers cache the answer (the variable is read once per process).
int debug_int(const char* name);

#define LADE_REQUIRE(cond, code, ...)      \
    do {                                   \
        if (!(cond)) {                     \
            lade::set_error(__VA_ARGS__);  \
            return (code);                 \
        }                                  \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// seal of a step record (include/lade_hip.h: lade_record_seal): one definition for the kernel that writes it and the host that checks it
__host__ __device__ static inline uint32_t seal_words(const uint32_t* rec, uint32_t step_no) {
    uint32_t x = step_no * 0x9E3779B1u + 0x7F4A7C15u;
    for (uint32_t w = 0; w + 1 < LADE_REC_WORDS; ++w) x ^= (rec[w] + w) * (2u * w + 0x85EBCA6Bu);
    return x;
}

// ---- device: 16-bit float storage <-> fp32 ---------------------------------------------
struct BF16 {};  // tag types: storage is uint16_t
struct F16 {};
struct F32 {};  // 4-byte storage

template <typename T>
__device__ __forceinline__ float to_f32(uint16_t v);
template <>
__device__ __forceinline__ float to_f32<BF16>(uint16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
template <>
__device__ __forceinline__ float to_f32<F16>(uint16_t v) {
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return (float)h;
}

template <typename T>
__device__ __forceinline__ uint16_t from_f32(float f);
template <>
__device__ __forceinline__ uint16_t from_f32<BF16>(float f) {  // round to nearest even, NaN kept quiet
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
template <>
__device__ __forceinline__ uint16_t from_f32<F16>(float f) {
    _Float16 h = (_Float16)f;  // v_cvt_f16_f32, RTE
    uint16_t v;
    __builtin_memcpy(&v, &h, 2);
    return v;
}

// two fp32 -> packed pair of 16-bit floats (lo in bits 0..15)
template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <>
__device__ __forceinline__ uint32_t pack2<BF16>(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
template <>
__device__ __forceinline__ uint32_t pack2<F16>(float lo, float hi) {
    return (uint32_t)from_f32<F16>(lo) | ((uint32_t)from_f32<F16>(hi) << 16);
}

// torch.argmax's order, one definition for lade_argmax_rows, lade_argmax_pairs and the lm_head GEMM's argmax epilogue: a NaN is the maximum
// (the first NaN of a row wins), among equal values the lowest index wins; bi == 0x7fffffff = nothing seen yet
__device__ __forceinline__ bool argmax_better(float v, int i, float best, int bi) {
    const bool vn = v != v, bn = best != best;
    if (bi == 0x7fffffff) return true;
    if (vn) return !bn || i < bi;
    return !bn && (v > best || (v == best && i < bi));
}

// storage-type helpers: 16-bit types round after every elementwise op exactly like torch does
// (fp32 math, one rounding per op); fp32 uses separately rounded mul / add (no contraction).
template <typename T> struct Elem;
template <> struct Elem<BF16> { typedef uint16_t S; __device__ static float ld(S v) { return to_f32<BF16>(v); } __device__ static S st(float f) { return from_f32<BF16>(f); } };
template <> struct Elem<F16> { typedef uint16_t S; __device__ static float ld(S v) { return to_f32<F16>(v); } __device__ static S st(float f) { return from_f32<F16>(f); } };
template <> struct Elem<F32> { typedef float S; __device__ static float ld(S v) { return v; } __device__ static S st(float f) { return f; } };

// one rounding to the storage type.  The empty asm keeps the compiler from narrowing the fp32
// expression to native f16 ops and contracting mul+add into v_fma_f16, which would skip a rounding
// the reference's separate torch ops perform.
template <typename T>
__device__ __forceinline__ float rnd(float f) {
    float r = Elem<T>::ld(Elem<T>::st(f));
    asm volatile("" : "+v"(r));
    return r;
}

}  // namespace lade
