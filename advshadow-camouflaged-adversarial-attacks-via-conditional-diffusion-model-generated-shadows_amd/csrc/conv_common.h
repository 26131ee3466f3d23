// Shared pieces of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_halo.hip).
#pragma once
#include "common.h"

#define SLAB 128                 // bytes of K per row per step

// Division by a launch-invariant divisor (Granlund-Montgomery): q = (umulhi(n, m) + n) >> s, exact
// for n < 2^31.  Replaces ~35-instruction integer divides in the tile prologue / epilogue.
struct FastDiv {
    unsigned d, m, s;
    __host__ void init(unsigned dd) {
        d = dd; s = 0;
        while ((1ull << s) < dd) ++s;
        m = (unsigned)(((1ull << 32) * ((1ull << s) - dd)) / dd + 1);
    }
    __device__ __forceinline__ unsigned div(unsigned n) const { return (__umulhi(n, m) + n) >> s; }
};

struct ConvKP {
    const char* x1; const char* x2; const char* w;
    const char* e1; const char* e2;   // extra 1x1 operand (two concat sources) appended to K, or null
    int E1, E2; unsigned e1_bytes, e2_bytes;
    const float* bias; const float* temb; const char* res; char* y;
    const float* norm;           // [B][C1 + C2][2] (scale, shift): the conv reads SiLU(scale * x + shift) (advs_conv_args.norm), or null
    const char* mask;            // ReLU-backward mask (advs_conv_args.relu_mask), read only by the MASK instantiations
    float* stats;                // [ceil(M/WM)][Cout][2] per-channel (sum, sum of squares) of y, or null
    unsigned x1_bytes, x2_bytes, w_bytes;
    int B, H, W, C1, C2, Cout;
    int LD1, LD2;                // pixel strides of x1 / x2 in elements (<= C1 / C2, see advs_conv_args.ld1)
    int R, stride, pad, ups;
    int Ho, Wo, M, K;            // K in elements
    int act, temb_stride;
    int nMt, nNt;
    int fast_epi;                // 16-bit dtype, no residual / activation / mask: conv_epilogue_fast (bias + temb start in the accumulators)
    FastDiv dHoWo, dWo;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define OOB_OFFSET 0xF0000000u   // > every num_records we accept: the bounds check returns zeros

__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_base, 16, voff, soff, 0, 0);
}

template <typename T> struct Mma;
template <> struct Mma<BF16> {
    static constexpr int ESZ = 2;
    // one 16-byte fragment per operand = K of 16 (two lane halves x 8)
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) { c = mma16<BF16>(a, b, c); }
};
template <> struct Mma<F16> {
    static constexpr int ESZ = 2;
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) { c = mma16<F16>(a, b, c); }
};
template <> struct Mma<float> {
    static constexpr int ESZ = 4;
    // 16 bytes = 4 floats per lane half; float j of both halves forms one K=2 step
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
    }
};


// ---- epilogue shared by both kernels.  C/D layout of a 32x32 MFMA tile: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Per 32-row strip the wave parks its 32 x WN accumulators in
// a private LDS patch (f32, row-major) and re-reads it one 16-byte output vector per lane, so bias /
// temb / residual are read and y is written along the channel axis in whole 128-byte lines.
//   row_to_m(lr): output pixel index of the wave's local row lr (0 .. TM*32), or -1 when out of range
//   temb_b: batch index when every row of the wave shares it (temb folded into the bias), -1 = per row
//   rb: row-block index for the epilogue statistics, -1 = this wave writes none
//   wave_stats: null, or WN (sum, sum of squares) pairs in LDS that take the wave's statistics INSTEAD of p.stats -- the caller then
//               adds the waves of its workgroup up (conv_halo2.hip: one entry per workgroup, a quarter / an eighth of the bytes)
template <typename T, int WN, int TM, int TN, bool MASK = false, typename RowMap>
__device__ __forceinline__ void conv_epilogue(const ConvKP& p, f32x16 (&acc)[TM][TN], float* patch, int lane,
                                              int ncol0, RowMap row_to_m, int temb_b, int rb, float2* wave_stats = nullptr) {
    constexpr int VEC = 16 / Mma<T>::ESZ;
    constexpr int LPR = WN / VEC;                 // lanes per patch row
    constexpr int RPI = 64 / LPR;                 // rows per wave-instruction
    const int l31 = lane & 31, lh = lane >> 5;
    T* y = (T*)p.y;
    const T* res = (const T*)p.res;
    const int prow = lane / LPR, pcv = lane - prow * LPR;
    const int n = ncol0 + pcv * VEC;
    const bool n_ok = n < p.Cout;                 // Cout is a multiple of VEC
    float bias[VEC], ssum[VEC], ssq[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { bias[e] = (p.bias && n_ok) ? p.bias[n + e] : 0.f; ssum[e] = 0.f; ssq[e] = 0.f; }
    const bool temb_rowwise = p.temb && temb_b < 0;
    const bool res_after = (p.act & ADVS_GN_RESIDUAL_AFTER_ACT) != 0;       // same flag bit as advs_groupnorm's act
    const int act = p.act & ~ADVS_GN_RESIDUAL_AFTER_ACT;
    if (p.temb && temb_b >= 0 && n_ok) {
        const float* tp = p.temb + (size_t)temb_b * p.temb_stride + n;
#pragma unroll
        for (int e = 0; e < VEC; ++e) bias[e] += tp[e];
    }
    constexpr int NIT = 32 / RPI;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        // this strip's residual vectors are requested first so their latency overlaps the LDS transpose
        u32x4 rraw[NIT], mraw[MASK ? NIT : 1];
        if (MASK) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int m = row_to_m(i * 32 + it * RPI + prow);
                mraw[it] = (m >= 0 && n_ok) ? *(const u32x4*)((const T*)p.mask + (size_t)m * p.Cout + n) : u32x4{0, 0, 0, 0};
            }
        }
        if (res) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int m = row_to_m(i * 32 + it * RPI + prow);
                rraw[it] = (m >= 0 && n_ok) ? *(const u32x4*)(res + (size_t)m * p.Cout + n) : u32x4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * WN + j * 32 + l31] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int row = it * RPI + prow;
            const int m = row_to_m(i * 32 + row);
            float v[VEC];
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                const f32x4 t = *(const f32x4*)(patch + row * WN + pcv * VEC + e);
                v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
            }
            if (m >= 0 && n_ok) {
                const size_t o = (size_t)m * p.Cout + n;
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] += bias[e];
                if (temb_rowwise) {
                    const float* tp = p.temb + (size_t)p.dHoWo.div((unsigned)m) * p.temb_stride + n;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += tp[e];
                }
                float rv[VEC];
                if (res) unpack16<T>(rraw[it], rv);
                if (res && !res_after) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += rv[e];
                }
                if (act != ADVS_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] = apply_act(v[e], act);
                }
                if (res && res_after) {                  // y = act(conv) + x (FusedMBConv with expand 1: Conv-BN-SiLU, then add)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += rv[e];
                }
                if (MASK) {                              // ReLU backward: gradient only where the forward activation was positive
                    float mv[VEC];
                    unpack16<T>(mraw[it], mv);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
                }
                const u32x4 packed = pack16<T>(v);
                *(u32x4*)(y + o) = packed;
                if (p.stats) {                    // statistics of the values as stored (after rounding)
                    float sv[VEC];
                    unpack16<T>(packed, sv);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { ssum[e] += sv[e]; ssq[e] = fmaf(sv[e], sv[e], ssq[e]); }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (p.stats) {
        // fold the RPI row-lanes that share a channel vector (fixed butterfly order), lanes 0..LPR-1 store
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { ssum[e] += __shfl_xor(ssum[e], o); ssq[e] += __shfl_xor(ssq[e], o); }
        if (wave_stats) {                            // channels past Cout carry zeros
            if (prow == 0) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) wave_stats[pcv * VEC + e] = make_float2(ssum[e], ssq[e]);
            }
        } else if (prow == 0 && n_ok && rb >= 0) {
            float* sp = p.stats + ((size_t)rb * p.Cout + n) * 2;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { sp[2 * e] = ssum[e]; sp[2 * e + 1] = ssq[e]; }
        }
    }
}


// ---- fast epilogue of the halo kernels for the common case (16-bit storage, no residual, no activation, no mask; every row of the
// tile inside the image, one image per workgroup).  The generic epilogue above is VALU-bound: ~70 vector instructions per
// 8-row store, two waves per SIMD, 6.5 us per workgroup at the level-0 shapes = 27 % of those launches (profiles/round2_ablation.txt).
// Here everything that can be done in the ACCUMULATOR layout (lane = channel, registers = 16 pixel rows) is done there:
//   * bias and the time-embedding slice are the accumulators' initial value (conv_acc_init), so they cost nothing afterwards;
//   * rounding packs two pixel rows of one channel per register, and the GroupNorm statistics of the ROUNDED values come from two
//     v_dot2c per packed pair (sum against a pair of ones, sum of squares against itself), summed over the lane's rows;
//   * the transposition goes through a 16-bit patch (half the LDS bytes), and the read side is ds_read_b128 -> global_store_dwordx4
//     with no arithmetic at all.
template <typename T> __device__ __forceinline__ float dot2_acc(unsigned a, unsigned b, float c);
template <> __device__ __forceinline__ float dot2_acc<BF16>(unsigned a, unsigned b, float c) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a), __builtin_bit_cast(bf2, b), c, false);
}
template <> __device__ __forceinline__ float dot2_acc<F16>(unsigned a, unsigned b, float c) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
}
template <> __device__ __forceinline__ float dot2_acc<float>(unsigned, unsigned, float c) { return c; }   // never used (fast_epi is 16-bit only)
template <typename T> __device__ __forceinline__ unsigned ones_pair() { return 0x3c003c00u; }          // (1.0, 1.0) in fp16
template <> __device__ __forceinline__ unsigned ones_pair<BF16>() { return 0x3f803f80u; }

// Accumulator start value: zero, or -- for the fast epilogue -- bias (+ the image's time-embedding slice) of the lane's channel.
// (Tried and dropped: also starting from the residual, gathered in the accumulator layout with 64 two-byte loads per lane in front
// of the prologue -- 853 us against 825 us for the generic epilogue at the level-0 shape; the residual layers keep the generic path.)
template <int TM, int TN>
__device__ __forceinline__ void conv_acc_init(const ConvKP& p, f32x16 (&acc)[TM][TN], int lane, int ncol0, int temb_b) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = ncol0 + j * 32 + (lane & 31);
        float v = 0.f;
        if (p.fast_epi && c < p.Cout) {
            if (p.bias) v = p.bias[c];
            if (p.temb) v += p.temb[(size_t)temb_b * p.temb_stride + c];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = v;
    }
}

template <typename T, int WN, int TM, int TN, typename RowMap>
__device__ __forceinline__ void conv_epilogue_fast(const ConvKP& p, f32x16 (&acc)[TM][TN], char* patch, int lane, int ncol0,
                                                   RowMap row_to_m, int rb, float2* wave_stats = nullptr) {
    static_assert(WN == TN * 32, "one wave owns TN channel tiles");
    constexpr int ROWB = WN * 2;                  // bytes of one patch row (16-bit elements)
    constexpr int LPR = ROWB / 16, RPI = 64 / LPR;
    const int l31 = lane & 31, lh = lane >> 5;
    const int prow = lane / LPR, pcv = lane - prow * LPR;
    const int n = ncol0 + pcv * 8;
    const bool n_ok = n < p.Cout;
    const unsigned ones = ones_pair<T>();
    char* y = p.y;
    float ssum[TN], ssq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = 2 * q;                                 // registers r, r + 1 = pixel rows row, row + 1
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const unsigned pk = pack2<T>(acc[i][j][r], acc[i][j][r + 1]);
                ssum[j] = dot2_acc<T>(pk, ones, ssum[j]);
                ssq[j] = dot2_acc<T>(pk, pk, ssq[j]);
                unsigned short* dst = (unsigned short*)(patch + row * ROWB + (j * 32 + l31) * 2);
                dst[0] = (unsigned short)pk;
                dst[ROWB / 2] = (unsigned short)(pk >> 16);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int row = it * RPI + prow;
            const int m = row_to_m(i * 32 + row);
            const u32x4 v = *(const u32x4*)(patch + row * ROWB + pcv * 16);
            // y is read next by a GroupNorm pass over a tensor far larger than the L2: streaming it past the caches leaves them to the
            // halo and weight re-reads (conv -1.5 %, round 2; the generic epilogue keeps plain stores: the victims' small maps want the L2)
            if (m >= 0 && n_ok) __builtin_nontemporal_store(v, (u32x4*)(y + ((size_t)m * p.Cout + n) * 2));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (p.stats && rb >= 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {                               // the two lane halves hold the other 16 rows of the same channel
            const float s = ssum[j] + __shfl_xor(ssum[j], 32), q = ssq[j] + __shfl_xor(ssq[j], 32);
            const int c = ncol0 + j * 32 + l31;
            if (wave_stats) {
                if (lh == 0) wave_stats[j * 32 + l31] = make_float2(s, q);
            } else if (lh == 0 && c < p.Cout) {
                float2* sp = (float2*)(p.stats + ((size_t)rb * p.Cout + c) * 2);
                *sp = make_float2(s, q);
            }
        }
    }
}
