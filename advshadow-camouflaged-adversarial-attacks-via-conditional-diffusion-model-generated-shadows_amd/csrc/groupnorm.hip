// GroupNorm (+ activation, + residual) on NHWC activations.  HBM-bound: the statistics pass
// reads x once; the apply pass reads x (and the residual) once and writes y once.
// Algorithmic traffic: 3 * B*HW*C*sizeof(T) bytes (4x with a residual).
//
// Determinism: statistics are reduced in a fixed order (per-thread f32 sums -> LDS tree ->
// per-chunk partials -> f64 combine), no atomics, so a sample's result does not depend on
// the batch it is in or on the GPU it runs on.
#include "common.h"

#define GN_MAX_CHUNKS 256
#define GN_THREADS 256

extern "C" size_t advs_groupnorm_scratch_bytes(int b, int groups) {
    return (size_t)b * GN_MAX_CHUNKS * groups * 2 * sizeof(float);
}

// thread layout shared by both passes: tid -> (pixel lane pl, channel vector cv)
template <typename T>
__global__ void __launch_bounds__(GN_THREADS)
gn_partial_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1, float* __restrict__ partials,
                  int HW, int C, int G, int nchunk) {
    constexpr int VEC = Elt<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [PIXB][C][2]
    const int vpp = C / VEC, PIXB = GN_THREADS / vpp;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x, pl = tid / vpp, cv = tid - pl * vpp;
    const int per = (HW + nchunk - 1) / nchunk;
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    float s[VEC], q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (pl < PIXB) {
        // channel vector cv lives in source 1 (C1 channels) or source 2 (C - C1 channels)
        const int v1 = C1 / VEC;
        const bool in1 = cv < v1;
        const int vs = in1 ? v1 : vpp - v1;                       // vectors per pixel in that source
        const u32x4* base = in1 ? (const u32x4*)(x + (size_t)b * HW * C1) + cv
                                : (const u32x4*)(x2 + (size_t)b * HW * (C - C1)) + (cv - v1);
        int p = p0 + pl;
        for (; p + 3 * PIXB < p1; p += 4 * PIXB) {                 // 4 loads in flight
            u32x4 r0 = base[(size_t)p * vs], r1 = base[(size_t)(p + PIXB) * vs];
            u32x4 r2 = base[(size_t)(p + 2 * PIXB) * vs], r3 = base[(size_t)(p + 3 * PIXB) * vs];
            float f[VEC];
            unpack16<T>(r0, f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { s[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
            unpack16<T>(r1, f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { s[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
            unpack16<T>(r2, f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { s[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
            unpack16<T>(r3, f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { s[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
        }
        for (; p < p1; p += PIXB) {
            u32x4 r0 = base[(size_t)p * vs];
            float f[VEC];
            unpack16<T>(r0, f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { s[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            sm[((size_t)pl * C + cv * VEC + j) * 2] = s[j];
            sm[((size_t)pl * C + cv * VEC + j) * 2 + 1] = q[j];
        }
    }
    __syncthreads();
    // fold the pixel lanes: channel c -> sm[c][0..1] of lane 0
    for (int c = tid; c < C; c += GN_THREADS) {
        float a = sm[c * 2], d = sm[c * 2 + 1];
        for (int l = 1; l < PIXB; ++l) { a += sm[((size_t)l * C + c) * 2]; d += sm[((size_t)l * C + c) * 2 + 1]; }
        sm[c * 2] = a; sm[c * 2 + 1] = d;
    }
    __syncthreads();
    // fold channels of a group: one wave per group, butterfly in a fixed order
    const int cpg = C / G, wave = tid >> 6, lane = tid & 63;
    for (int g = wave; g < G; g += GN_THREADS / 64) {
        float a = 0.f, d = 0.f;
        for (int c = lane; c < cpg; c += 64) { a += sm[(g * cpg + c) * 2]; d += sm[(g * cpg + c) * 2 + 1]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); d += __shfl_xor(d, o); }
        if (lane == 0) {
            float* out = partials + (((size_t)b * nchunk + chunk) * G + g) * 2;
            out[0] = a; out[1] = d;
        }
    }
}

// Per-group mean / rstd of sample b from the chunk partials: LPG lanes per group walk the chunks (fixed interleave, loads issued
// together), then a butterfly: G <= 32 groups x 8 lanes, or G <= 64 groups x 4 lanes = 256 threads.  Shared by the apply pass and
// the (scale, shift) table of the fused form, so both see the same bits.
__device__ __forceinline__ void gn_combine_partials(const float* __restrict__ partials, int b, int tid, int HW, int cpg, int G,
                                                    int nchunk, float* s_mean, float* s_rstd) {
    const int lsh = G <= 32 ? 3 : 2, LPG = 1 << lsh;
    const int g = tid >> lsh, sub = tid & (LPG - 1);
    double a = 0.0, d = 0.0;
    if (g < G) {
        const float2* p = (const float2*)partials + ((size_t)b * nchunk * G + g);
        int k = sub;
        for (; k + 3 * LPG < nchunk; k += 4 * LPG) {
            const float2 e0 = p[(size_t)k * G], e1 = p[(size_t)(k + LPG) * G], e2 = p[(size_t)(k + 2 * LPG) * G],
                         e3 = p[(size_t)(k + 3 * LPG) * G];
            a += (double)e0.x; d += (double)e0.y; a += (double)e1.x; d += (double)e1.y;
            a += (double)e2.x; d += (double)e2.y; a += (double)e3.x; d += (double)e3.y;
        }
        for (; k < nchunk; k += LPG) { const float2 e = p[(size_t)k * G]; a += (double)e.x; d += (double)e.y; }
    }
    a += __shfl_xor(a, 1); d += __shfl_xor(d, 1);
    a += __shfl_xor(a, 2); d += __shfl_xor(d, 2);
    if (lsh == 3) { a += __shfl_xor(a, 4); d += __shfl_xor(d, 4); }
    if (g < G && sub == 0) {
        double n = (double)HW * cpg, mean = a / n, var = d / n - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[g] = (float)mean;
        s_rstd[g] = (float)(1.0 / sqrt(var + 1e-5));
    }
}
// scale / shift of channel c: y = fma(x, scale, shift).  One definition for the apply pass and the fused form's table.
__device__ __forceinline__ void gn_channel_affine(float mean, float rstd, float gamma, float beta, float& scale, float& shift) {
    scale = rstd * gamma;
    shift = fmaf(-mean, scale, beta);
}

// FAST: the dominant launch of the eps-predictor -- GroupNorm + SiLU, no residual, no per-channel add, 16-bit
// storage -- with the activation fixed at compile time (no per-element dispatch, no dead adds).
template <typename T, bool FAST>
__global__ void __launch_bounds__(GN_THREADS)
gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1, const float* __restrict__ gamma,
                const float* __restrict__ beta, const T* __restrict__ res, T* __restrict__ y,
                const float* __restrict__ partials, int HW, int C, int G, int nchunk, int nblk, int act,
                const float* __restrict__ cadd, int cadd_stride, const float* __restrict__ meanrstd) {
    constexpr int VEC = Elt<T>::VEC;
    __shared__ float s_mean[64], s_rstd[64];
    const int vpp = C / VEC, PIXB = GN_THREADS / vpp;
    const int b = blockIdx.y, tid = threadIdx.x, cpg = C / G;
    if (meanrstd) {
        for (int g = tid; g < G; g += GN_THREADS) {
            s_mean[g] = meanrstd[((size_t)b * G + g) * 2];
            s_rstd[g] = meanrstd[((size_t)b * G + g) * 2 + 1];
        }
    } else {
        gn_combine_partials(partials, b, tid, HW, cpg, G, nchunk, s_mean, s_rstd);
    }
    __syncthreads();
    const int pl = tid / vpp, cv = tid - pl * vpp;
    if (pl >= PIXB) return;
    float ca[VEC], cb[VEC], cc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        int c = cv * VEC + j, g = c / cpg;
        gn_channel_affine(s_mean[g], s_rstd[g], gamma[c], beta[c], ca[j], cb[j]);
        if constexpr (FAST) { ca[j] *= ADVS_LOG2E; cb[j] *= ADVS_LOG2E; }      // silu_fast_prescaled's argument (same two products as gn_affine_kernel)
        cc[j] = cadd ? cadd[(size_t)b * cadd_stride + c] : 0.f;
    }
    const int per = (HW + nblk - 1) / nblk;
    const int p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
    const size_t sample = (size_t)b * HW * vpp;
    const int v1 = C1 / VEC;
    const bool in1 = cv < v1;
    const int vs = in1 ? v1 : vpp - v1;
    const u32x4* xb = in1 ? (const u32x4*)x + (size_t)b * HW * v1 + cv
                          : (const u32x4*)x2 + (size_t)b * HW * (vpp - v1) + (cv - v1);
    const u32x4* rb = res ? (const u32x4*)res + sample + cv : nullptr;
    u32x4* yb = (u32x4*)y + sample + cv;
    const bool res_after = (act & ADVS_GN_RESIDUAL_AFTER_ACT) != 0;     // act(norm(x)) + residual (module.py:45-46)
    act &= ~ADVS_GN_RESIDUAL_AFTER_ACT;
    auto one = [&](const u32x4& raw, const u32x4& rraw) -> u32x4 {
        float f[VEC], r[VEC];
        unpack16<T>(raw, f);
        if constexpr (FAST) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) f[j] = silu_fast_prescaled(fmaf(f[j], ca[j], cb[j]));
            return pack16<T>(f);
        }
        if (rb) unpack16<T>(rraw, r);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float v = fmaf(f[j], ca[j], cb[j]);
            if (rb && !res_after) v += r[j];
            v = apply_act_t<T>(v, act) + cc[j];
            if (rb && res_after) v += r[j];
            f[j] = v;
        }
        return pack16<T>(f);
    };
    int p = p0 + pl;
    for (; p + 3 * PIXB < p1; p += 4 * PIXB) {                     // four 16-byte loads in flight per lane
        const size_t i0 = (size_t)p * vs, i1 = (size_t)(p + PIXB) * vs, i2 = (size_t)(p + 2 * PIXB) * vs,
                     i3 = (size_t)(p + 3 * PIXB) * vs;
        // x is dead after this pass: stream it past the caches (nontemporal), they are better spent on y
        const u32x4 a0 = __builtin_nontemporal_load(xb + i0), a1 = __builtin_nontemporal_load(xb + i1),
                    a2 = __builtin_nontemporal_load(xb + i2), a3 = __builtin_nontemporal_load(xb + i3);
        u32x4 r0 = a0, r1 = a0, r2 = a0, r3 = a0;
        if (rb) {
            r0 = rb[(size_t)p * vpp]; r1 = rb[(size_t)(p + PIXB) * vpp];
            r2 = rb[(size_t)(p + 2 * PIXB) * vpp]; r3 = rb[(size_t)(p + 3 * PIXB) * vpp];
        }
        yb[(size_t)p * vpp] = one(a0, r0);
        yb[(size_t)(p + PIXB) * vpp] = one(a1, r1);
        yb[(size_t)(p + 2 * PIXB) * vpp] = one(a2, r2);
        yb[(size_t)(p + 3 * PIXB) * vpp] = one(a3, r3);
    }
    for (; p < p1; p += PIXB) {
        const u32x4 a0 = xb[(size_t)p * vs];
        const u32x4 r0 = rb ? rb[(size_t)p * vpp] : a0;
        yb[(size_t)p * vpp] = one(a0, r0);
    }
}

// Statistics already emitted by the producing conv epilogues (advs_conv_args.stats): per source a
// [row blocks][C][2] array of per-channel (sum, sum of squares).  gn_stats_fold_kernel turns a slice
// of one sample's row blocks into per-group partial sums in the SAME layout gn_partial_kernel writes
// (partials[b][chunk][g][2]), so gn_apply_kernel combines them exactly as in the unfused path.
// Reads are channel-contiguous (a wave covers 512 consecutive bytes); order of summation is fixed.
#define GNF_MAXC 1024
__global__ void __launch_bounds__(256)
gn_stats_fold_kernel(const float* __restrict__ st1, int rbpi1, int C1, const float* __restrict__ st2, int rbpi2, int C2,
                     float* __restrict__ partials, int G, int nsplit) {
    __shared__ double tmp[4 * 256 * 2];           // [row lane][channel in chunk][2]
    __shared__ double chs[GNF_MAXC * 2];          // per concatenated channel (sum, sumsq)
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int C = C1 + C2;
    for (int src = 0; src < 2; ++src) {
        const float* st = src ? st2 : st1;
        const int Cs = src ? C2 : C1, rbpi = src ? rbpi2 : rbpi1, cbase = src ? C1 : 0;
        if (Cs == 0) continue;
        const int per = (rbpi + nsplit - 1) / nsplit;
        const int r0 = chunk * per, r1 = min(rbpi, r0 + per);
        const int rows_par = Cs < 256 ? (256 / Cs < 4 ? 256 / Cs : 4) : 1;      // tmp holds 4 row lanes
        const int rl = Cs < 256 ? tid / Cs : 0, cl = Cs < 256 ? tid - rl * Cs : tid;
        for (int c0 = 0; c0 < Cs; c0 += 256) {
            const int c = c0 + cl;
            double s = 0.0, q = 0.0;
            if (c < Cs && rl < rows_par) {
                // eight independent loads in flight per lane, summed in row order (the loop was latency-bound: one
                // L2 round trip per row block, 16-32 of them in series = 8.6 us per launch, 51 launches per forward)
                const float2* sp = (const float2*)st + ((size_t)b * rbpi * Cs + c);
                int r = r0 + rl;
                for (; r + 7 * rows_par < r1; r += 8 * rows_par) {
                    float2 e[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) e[u] = sp[(size_t)(r + u * rows_par) * Cs];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { s += (double)e[u].x; q += (double)e[u].y; }
                }
                for (; r < r1; r += rows_par) {
                    const float2 e = sp[(size_t)r * Cs];
                    s += (double)e.x; q += (double)e.y;
                }
            }
            if (rl < rows_par) { tmp[(rl * 256 + cl) * 2] = s; tmp[(rl * 256 + cl) * 2 + 1] = q; }
            __syncthreads();
            if (rl == 0 && c < Cs) {
                for (int l = 1; l < rows_par; ++l) { s += tmp[(l * 256 + cl) * 2]; q += tmp[(l * 256 + cl) * 2 + 1]; }
                chs[(cbase + c) * 2] = s; chs[(cbase + c) * 2 + 1] = q;
            }
            __syncthreads();
        }
    }
    const int cpg = C / G;
    for (int g = tid; g < G; g += 256) {
        double s = 0.0, q = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s += chs[c * 2]; q += chs[c * 2 + 1]; }
        float* out = partials + (((size_t)b * nsplit + chunk) * G + g) * 2;
        out[0] = (float)s; out[1] = (float)q;
    }
}

// (scale, shift) per (sample, channel) for advs_conv_args.norm: the coefficients gn_apply_kernel would use
__global__ void __launch_bounds__(GN_THREADS)
gn_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ partials,
                 float* __restrict__ table, int HW, int C, int G, int nchunk) {
    __shared__ float s_mean[64], s_rstd[64];
    const int b = blockIdx.x, tid = threadIdx.x, cpg = C / G;
    gn_combine_partials(partials, b, tid, HW, cpg, G, nchunk, s_mean, s_rstd);
    __syncthreads();
    for (int c = tid; c < C; c += GN_THREADS) {
        float sc, sh;
        gn_channel_affine(s_mean[c / cpg], s_rstd[c / cpg], gamma[c], beta[c], sc, sh);
        // stored times log2(e): the consumer evaluates SiLU as silu_fast_prescaled(fma(x, scale', shift')), like gn_apply_kernel<T, true>
        *(float2*)(table + ((size_t)b * C + c) * 2) = make_float2(sc * ADVS_LOG2E, sh * ADVS_LOG2E);
    }
}

static int gn_fold_chunks(int rbpi1, int rbpi2) {
    const int rbmax = rbpi1 > rbpi2 ? rbpi1 : rbpi2;
    // a FIXED number of chunks (not a function of the batch): the rounding points of the per-chunk f32 partials,
    // and with them the statistics of an image, do not depend on which batch the image is evaluated in.
    // 32 x batch workgroups: the fold is latency-bound, a full-size batch fills the chip four times over.
    int nchunk = 32;
    if (nchunk > rbmax / 8) nchunk = rbmax / 8;
    if (nchunk > GN_MAX_CHUNKS) nchunk = GN_MAX_CHUNKS;
    if (nchunk < 1) nchunk = 1;
    return nchunk;
}

extern "C" int advs_groupnorm_affine_stats(const float* stats1, int rbpi1, const float* stats2, int rbpi2, const float* gamma,
                                           const float* beta, void* scratch, float* table, int b, int hw, int c1, int c2,
                                           int groups, void* stream) {
    ADVS_REQUIRE(stats1 && rbpi1 > 0 && gamma && beta && scratch && table, "groupnorm_affine_stats: null pointer");
    ADVS_REQUIRE(c1 > 0 && c2 >= 0 && (c2 == 0) == (stats2 == nullptr), "groupnorm_affine_stats: c2/stats2 mismatch");
    const int c = c1 + c2;
    ADVS_REQUIRE(b > 0 && hw > 0 && groups > 0 && groups <= 64 && c % groups == 0 && c <= GNF_MAXC, "groupnorm_affine_stats: bad shape");
    const int nchunk = gn_fold_chunks(rbpi1, rbpi2);
    hipStream_t st = (hipStream_t)stream;
    gn_stats_fold_kernel<<<dim3(nchunk, b), 256, 0, st>>>(stats1, rbpi1, c1, stats2, rbpi2, c2, (float*)scratch, groups, nchunk);
    ADVS_CHECK_LAUNCH("gn_stats_fold");
    gn_affine_kernel<<<b, GN_THREADS, 0, st>>>(gamma, beta, (const float*)scratch, table, hw, c, groups, nchunk);
    ADVS_CHECK_LAUNCH("gn_affine");
    return ADVS_OK;
}

template <typename T>
static int gn_launch(const void* x, const void* x2, int c1, const float* gamma, const float* beta, const void* res,
                     void* y, float* partials, int b, int hw, int c, int groups, int act, const float* cadd,
                     int cadd_stride, hipStream_t st, const float* st1 = nullptr, int rbpi1 = 0,
                     const float* st2 = nullptr, int rbpi2 = 0) {
    constexpr int VEC = Elt<T>::VEC;
    const int vpp = c / VEC, PIXB = GN_THREADS / vpp;
    const float* meanrstd = nullptr;
    int nchunk = 1;
    if (st1) {
        nchunk = gn_fold_chunks(rbpi1, rbpi2);
        gn_stats_fold_kernel<<<dim3(nchunk, b), 256, 0, st>>>(st1, rbpi1, c1, st2, rbpi2, c - c1, partials, groups, nchunk);
        ADVS_CHECK_LAUNCH("gn_stats_fold");
    } else {
    // a fixed chunk count (batch-independent rounding points, see above), >= 4 pixels per lane per chunk
    nchunk = 64;
    int maxc = hw / (PIXB * 4);
    if (nchunk > maxc) nchunk = maxc;
    if (nchunk > GN_MAX_CHUNKS) nchunk = GN_MAX_CHUNKS;
    if (nchunk < 1) nchunk = 1;
    size_t lds = (size_t)PIXB * c * 2 * sizeof(float);
    gn_partial_kernel<T><<<dim3(nchunk, b), GN_THREADS, lds, st>>>((const T*)x, (const T*)x2, c1, partials, hw, c, groups, nchunk);
    ADVS_CHECK_LAUNCH("gn_partial");
    }
    int nblk = 4096 / b;
    int maxb = hw / (PIXB * 2);
    if (nblk > maxb) nblk = maxb;
    if (nblk < 1) nblk = 1;
    if (sizeof(T) == 2 && act == ADVS_ACT_SILU && !res && !cadd)
        gn_apply_kernel<T, true><<<dim3(nblk, b), GN_THREADS, 0, st>>>((const T*)x, (const T*)x2, c1, gamma, beta, nullptr,
                                                                       (T*)y, partials, hw, c, groups, nchunk, nblk, act,
                                                                       nullptr, 0, meanrstd);
    else
        gn_apply_kernel<T, false><<<dim3(nblk, b), GN_THREADS, 0, st>>>((const T*)x, (const T*)x2, c1, gamma, beta, (const T*)res,
                                                                        (T*)y, partials, hw, c, groups, nchunk, nblk, act,
                                                                        cadd, cadd_stride, meanrstd);
    ADVS_CHECK_LAUNCH("gn_apply");
    return ADVS_OK;
}

extern "C" int advs_groupnorm(const void* x, const void* x2, const float* gamma, const float* beta,
                              const void* residual_in, const float* chan_add, int chan_add_stride, void* y,
                              void* partials, int b, int hw, int c1, int c2, int groups, int act, int dtype,
                              void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_groupnorm: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && gamma && beta && y && partials, "groupnorm: null pointer");
    ADVS_REQUIRE(c1 > 0 && c2 >= 0 && (c2 == 0) == (x2 == nullptr), "groupnorm: x2/c2 mismatch");
    const int c = c1 + c2;
    ADVS_REQUIRE(b > 0 && hw > 0 && c > 0 && groups > 0 && groups <= 64 && c % groups == 0,
                 "groupnorm: bad shape b=%d hw=%d c=%d groups=%d", b, hw, c, groups);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c1 % vec == 0 && c2 % vec == 0 && c / vec <= GN_THREADS, "groupnorm: c=%d+%d unsupported for dtype %d", c1, c2, dtype);
    ADVS_REQUIRE((size_t)(GN_THREADS / (c / vec)) * c * 8 <= 65536, "groupnorm: LDS budget exceeded for c=%d", c);
    ADVS_SWITCH_T(dtype, return gn_launch<T>(x, x2, c1, gamma, beta, residual_in, y, (float*)partials, b, hw, c, groups, act, chan_add, chan_add_stride, (hipStream_t)stream));
    return ADVS_ERR_ARG;                    // not reached
}

// Same as advs_groupnorm, but the per-channel statistics were already emitted by the conv
// epilogues that produced x (and x2): no pass over the tensors for statistics.
extern "C" int advs_groupnorm_stats(const void* x, const void* x2, const float* stats1, int row_blocks_per_image1,
                                    const float* stats2, int row_blocks_per_image2, const float* gamma,
                                    const float* beta, const void* residual_in, const float* chan_add,
                                    int chan_add_stride, void* y, void* scratch, int b, int hw, int c1, int c2,
                                    int groups, int act, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_groupnorm_stats: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && gamma && beta && y && scratch && stats1 && row_blocks_per_image1 > 0, "groupnorm_stats: null pointer");
    ADVS_REQUIRE(c1 > 0 && c2 >= 0 && (c2 == 0) == (x2 == nullptr) && (c2 == 0) == (stats2 == nullptr),
                 "groupnorm_stats: x2/c2/stats2 mismatch");
    const int c = c1 + c2;
    ADVS_REQUIRE(b > 0 && hw > 0 && groups > 0 && groups <= 64 && c % groups == 0, "groupnorm_stats: bad shape");
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c1 % vec == 0 && c2 % vec == 0 && c / vec <= GN_THREADS && c <= GNF_MAXC, "groupnorm_stats: c=%d+%d unsupported", c1, c2);
    ADVS_SWITCH_T(dtype, return gn_launch<T>(x, x2, c1, gamma, beta, residual_in, y, (float*)scratch, b, hw, c, groups, act, chan_add, chan_add_stride, (hipStream_t)stream, stats1, row_blocks_per_image1, stats2, row_blocks_per_image2));
    return ADVS_ERR_ARG;                    // not reached
}
