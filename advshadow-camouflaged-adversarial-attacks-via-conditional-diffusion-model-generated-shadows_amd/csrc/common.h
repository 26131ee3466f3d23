// Shared helpers for the gfx950 kernels behind the C-ABI in include/advshadow.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/advshadow.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one MFMA A/B fragment
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8; // 8 fp16 = one MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- error plumbing -------------------------------------------------------
void advs_set_error(const char* fmt, ...);
#define ADVS_FAIL(code, ...) do { advs_set_error(__VA_ARGS__); return (code); } while (0)
#define ADVS_REQUIRE(cond, ...) do { if (!(cond)) ADVS_FAIL(ADVS_ERR_ARG, __VA_ARGS__); } while (0)
#define ADVS_CHECK_LAUNCH(name) do { hipError_t e__ = hipGetLastError(); \
    if (e__ != hipSuccess) ADVS_FAIL(ADVS_ERR_HIP, "%s: %s", name, hipGetErrorString(e__)); } while (0)
#define ADVS_HIP(call) do { hipError_t e__ = (call); \
    if (e__ != hipSuccess) ADVS_FAIL(ADVS_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); } while (0)

// ---- element types ----------------------------------------------------------
// Activations and GEMM weights are f32 (parity mode, exact-f32 MFMA), bf16 or fp16 (throughput modes,
// 16-bit MFMA with f32 accumulation; fp16 is the dtype BASELINE.json's config 4 names).  The 16-bit
// storage types are raw u16.
struct BF16 { unsigned short v; };
struct F16 { unsigned short v; };

__device__ __forceinline__ float bf16_to_f32(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned short, b);
}

__device__ __forceinline__ float f16_to_f32(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ unsigned short f32_to_f16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }   // RNE

template <typename T> struct Elt;
template <> struct Elt<float> {
    static constexpr int VEC = 4;               // elements per 16 bytes
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elt<BF16> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(const BF16* p) { return bf16_to_f32(p->v); }
    __device__ static __forceinline__ void st(BF16* p, float v) { p->v = f32_to_bf16(v); }
};

template <> struct Elt<F16> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(const F16* p) { return f16_to_f32(p->v); }
    __device__ static __forceinline__ void st(F16* p, float v) { p->v = f32_to_f16(v); }
};

// 16-byte vector <-> floats
template <typename T> __device__ __forceinline__ void unpack16(const u32x4& raw, float* out);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4& raw, float* out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = __uint_as_float(raw[i]);
}
template <> __device__ __forceinline__ void unpack16<BF16>(const u32x4& raw, float* out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[2 * i] = __uint_as_float(raw[i] << 16);
        out[2 * i + 1] = __uint_as_float(raw[i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void unpack16<F16>(const u32x4& raw, float* out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[2 * i] = f16_to_f32((unsigned short)(raw[i] & 0xffffu));
        out[2 * i + 1] = f16_to_f32((unsigned short)(raw[i] >> 16));
    }
}
template <typename T> __device__ __forceinline__ u32x4 pack16(const float* in);
template <> __device__ __forceinline__ u32x4 pack16<float>(const float* in) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __float_as_uint(in[i]);
    return r;
}

// two f32 -> one packed pair of 16-bit T
template <typename T> __device__ __forceinline__ unsigned pack2(float lo, float hi);
// (as ONE two-source conversion, v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32: RNE, NaN stays NaN -- the same rounding as two single conversions,
// a third of the instructions: two conversions plus a shift-or is what the scalar form compiles to)
typedef float f32x2_cv __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_cv __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_cv __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ unsigned pack2<BF16>(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_cv{lo, hi}, bf16x2_cv));
}
template <> __device__ __forceinline__ unsigned pack2<F16>(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_cv{lo, hi}, f16x2_cv));
}
template <> __device__ __forceinline__ u32x4 pack16<BF16>(const float* in) {
    return u32x4{pack2<BF16>(in[0], in[1]), pack2<BF16>(in[2], in[3]), pack2<BF16>(in[4], in[5]), pack2<BF16>(in[6], in[7])};
}
template <> __device__ __forceinline__ u32x4 pack16<F16>(const float* in) {
    return u32x4{pack2<F16>(in[0], in[1]), pack2<F16>(in[2], in[3]), pack2<F16>(in[4], in[5]), pack2<F16>(in[6], in[7])};
}
// K = 16 MFMA on one 16-byte fragment per operand
template <typename T> __device__ __forceinline__ f32x16 mma16(const u32x4& a, const u32x4& b, const f32x16& c);
template <> __device__ __forceinline__ f32x16 mma16<BF16>(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mma16<F16>(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// SiLU for values that are rounded to bf16 right after: v_exp_f32 + v_rcp_f32 (each ~1 ulp) instead of
// expf + IEEE division (~25 VALU instructions), which otherwise makes the HBM-bound norm pass VALU-bound.
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
// The same SiLU from the PRE-SCALED argument vp = v * log2(e): v * sigmoid(v) = vp / (log2e * (1 + 2^-vp)) = vp * rcp(fma(2^-vp, log2e, log2e)).
// One VALU slot less per element than silu_fast (the multiply by -log2e is folded into the caller's scale / shift, the add into an FMA): used
// where GroupNorm's affine feeds the activation (gn_apply_kernel<T, true> and the GroupNorm-on-load transform of conv_halo2.hip, which must
// agree bit for bit).
#define ADVS_LOG2E 1.4426950408889634f
__device__ __forceinline__ float silu_fast_prescaled(float vp) {
    return vp * __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_exp2f(-vp), ADVS_LOG2E, ADVS_LOG2E));
}
template <typename T> __device__ __forceinline__ float apply_act_t(float v, int act);

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ADVS_ACT_RELU: return v > 0.f ? v : 0.f;
        case ADVS_ACT_SILU: return v / (1.0f + expf(-v));
        case ADVS_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case ADVS_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
        case ADVS_ACT_LRELU01: return v > 0.f ? v : 0.1f * v;
        case ADVS_ACT_LRELU001: return v > 0.f ? v : 0.01f * v;
        case ADVS_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

template <> __device__ __forceinline__ float apply_act_t<float>(float v, int act) { return apply_act(v, act); }
template <> __device__ __forceinline__ float apply_act_t<BF16>(float v, int act) {
    return act == ADVS_ACT_SILU ? silu_fast(v) : apply_act(v, act);
}
template <> __device__ __forceinline__ float apply_act_t<F16>(float v, int act) {
    return act == ADVS_ACT_SILU ? silu_fast(v) : apply_act(v, act);
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t dtype_size(int dt) { return dt == ADVS_F32 ? 4 : 2; }
static inline bool dtype_ok(int dt) { return dt == ADVS_F32 || dt == ADVS_BF16 || dt == ADVS_F16; }
// Instantiate `...` with T = float / BF16 / F16 according to the runtime dtype code.
#define ADVS_SWITCH_T(dt, ...)                                              \
    do {                                                                    \
        if ((dt) == ADVS_BF16) { using T = BF16; __VA_ARGS__; }             \
        else if ((dt) == ADVS_F16) { using T = F16; __VA_ARGS__; }          \
        else { using T = float; __VA_ARGS__; }                              \
    } while (0)

// 128 zero bytes every lane may read instead of an out-of-image / out-of-range row.
const void* advs_zero_page();
