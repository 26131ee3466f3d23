// Backward-to-the-image pieces of the ViT victim (`loss.backward(); image.grad` of
// tools/train_shadow.py:204-212 when the classifier is the HF ViT of ASR_fast.py:47-58, BASELINE config 4's victim).
// Data gradients only (the attack never updates weights): every Linear's gradient is an advs_conv2d on the transposed
// weight; what is here are the non-GEMM pieces -- LayerNorm, exact GELU, softmax attention, the patch embedding -- in the
// order a reverse sweep of HF's ViTLayer needs them.  Token tensors are [B][n_pad][C] T with rows >= n_valid padding:
// padding rows carry zero gradient in and out of every kernel.
#include "common.h"

// ---------------------------------------------------------------------------------------------- LayerNorm backward
// y = (x - mean) * rstd * gamma + beta per row.  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma,
// plus an optional second gradient stream `add` (the residual branch of the pre-norm block).  One wave per row, f32 math.
template <typename T>
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma, const T* __restrict__ add,
                     T* __restrict__ dx, long long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long stride = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long row = row0; row < rows; row += stride) {
        const T* xr = x + (size_t)row * C;
        const T* gr = dy + (size_t)row * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += Elt<T>::ld(xr + c);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = Elt<T>::ld(xr + c) - mean; q = fmaf(d, d, q); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
        float m1 = 0.f, m2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float g = Elt<T>::ld(gr + c) * gamma[c];
            const float xh = (Elt<T>::ld(xr + c) - mean) * rstd;
            m1 += g; m2 = fmaf(g, xh, m2);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
        m1 /= (float)C; m2 /= (float)C;
        for (int c = lane; c < C; c += 64) {
            const float g = Elt<T>::ld(gr + c) * gamma[c];
            const float xh = (Elt<T>::ld(xr + c) - mean) * rstd;
            float v = rstd * (g - m1 - xh * m2);
            if (add) v += Elt<T>::ld(add + (size_t)row * C + c);
            Elt<T>::st(dx + (size_t)row * C + c, v);
        }
    }
}
// 16-bit storage, C a multiple of 8 and <= 2048: the row lives in registers (one pass over x and dy, 16-byte accesses).  LPR lanes
// (a power of two >= C / 8 vectors, at most 64) share a row, NV vectors each; 64 / LPR rows per wave.  Same arithmetic as above
// (mean, then the centred second moment, then the two gradient means), so the result is the same up to summation order.
template <typename T, int NV>
__global__ void __launch_bounds__(256)
layernorm_bwd16_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma, const T* __restrict__ add,
                       T* __restrict__ dx, long long rows, int C, float eps, int LPR) {
    const int lane = threadIdx.x & 63, sub = lane & (LPR - 1), rpw = 64 / LPR;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int nvec = C / 8;
    const float invC = 1.0f / (float)C;
    for (long long row = wave * rpw + lane / LPR; row < (rows + rpw - 1) / rpw * rpw; row += nwaves * rpw) {
        const bool rok = row < rows;
        float xv[NV][8], gv[NV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = sub + i * LPR;
            const bool ok = rok && v < nvec;
            u32x4 xr{0, 0, 0, 0}, gr{0, 0, 0, 0};
            if (ok) {
                xr = *((const u32x4*)(x + (size_t)row * C) + v);
                gr = *((const u32x4*)(dy + (size_t)row * C) + v);
            }
            unpack16<T>(xr, xv[i]);
            unpack16<T>(gr, gv[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s += xv[i][e];
                gv[i][e] *= ok ? gamma[v * 8 + e] : 0.f;
            }
        }
        for (int o = LPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool ok = sub + i * LPR < nvec;
#pragma unroll
            for (int e = 0; e < 8; ++e) { xv[i][e] = ok ? xv[i][e] - mean : 0.f; q = fmaf(xv[i][e], xv[i][e], q); }
        }
        for (int o = LPR >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q * invC + eps);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { xv[i][e] *= rstd; m1 += gv[i][e]; m2 = fmaf(gv[i][e], xv[i][e], m2); }
        for (int o = LPR >> 1; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
        m1 *= invC; m2 *= invC;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = sub + i * LPR;
            if (!(rok && v < nvec)) continue;
            float out[8], av[8];
            if (add) unpack16<T>(*((const u32x4*)(add + (size_t)row * C) + v), av);
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = rstd * (gv[i][e] - m1 - xv[i][e] * m2) + (add ? av[e] : 0.f);
            *((u32x4*)(dx + (size_t)row * C) + v) = pack16<T>(out);
        }
    }
}
template <typename T>
static bool layernorm_bwd16_launch(const void* dy, const void* x, const float* gamma, const void* add, void* dx, long long rows, int c,
                                   float eps, hipStream_t st) {
    if constexpr (sizeof(T) != 2) return false;
    else {
        if (c % 8 || c > 2048 || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)add) & 15)) return false;
        const int nvec = c / 8;
        int lpr = 1;
        while (lpr < nvec && lpr < 64) lpr <<= 1;
        const int nv = (nvec + lpr - 1) / lpr;                 // 1, 2 or (3,) 4
        const long long nw = (rows + (64 / lpr) - 1) / (64 / lpr);
        const long long blocks = (nw + 3) / 4;
        const int grid = (int)(blocks < 16384 ? blocks : 16384);
        if (nv == 1) layernorm_bwd16_kernel<T, 1><<<grid, 256, 0, st>>>((const T*)dy, (const T*)x, gamma, (const T*)add, (T*)dx, rows, c, eps, lpr);
        else if (nv == 2) layernorm_bwd16_kernel<T, 2><<<grid, 256, 0, st>>>((const T*)dy, (const T*)x, gamma, (const T*)add, (T*)dx, rows, c, eps, lpr);
        else layernorm_bwd16_kernel<T, 4><<<grid, 256, 0, st>>>((const T*)dy, (const T*)x, gamma, (const T*)add, (T*)dx, rows, c, eps, lpr);
        return true;
    }
}
extern "C" int advs_layernorm_bwd(const void* dy, const void* x, const float* gamma, const void* add, void* dx, long long rows,
                                  int c, float eps, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_layernorm_bwd: unknown dtype code %d", dtype);
    ADVS_REQUIRE(dy && x && gamma && dx && rows > 0 && c > 0, "layernorm_bwd: bad args");
    if (dtype != ADVS_F32) {
        const bool done = dtype == ADVS_BF16 ? layernorm_bwd16_launch<BF16>(dy, x, gamma, add, dx, rows, c, eps, (hipStream_t)stream)
                                             : layernorm_bwd16_launch<F16>(dy, x, gamma, add, dx, rows, c, eps, (hipStream_t)stream);
        if (done) { ADVS_CHECK_LAUNCH("layernorm_bwd"); return ADVS_OK; }
    }
    const long long blocks = (rows + 3) / 4;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);
    ADVS_SWITCH_T(dtype, layernorm_bwd_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)dy, (const T*)x, gamma, (const T*)add,
                                                                                       (T*)dx, rows, c, eps));
    ADVS_CHECK_LAUNCH("layernorm_bwd");
    return ADVS_OK;
}

// ---------------------------------------------------------------------------------------------- exact GELU, both ways
// forward: y = 0.5 x (1 + erf(x / sqrt 2)) (the activation of HF's ViTIntermediate, kept apart from the GEMM in the gradient plan so
// that the pre-activation survives); backward: dx = dy * (Phi(x) + x phi(x)).
template <typename T>
__global__ void gelu_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, size_t n, int backward) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = Elt<T>::ld(x + i);
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
        float r;
        if (backward) r = Elt<T>::ld(dy + i) * (cdf + v * 0.39894228040143267794f * expf(-0.5f * v * v));
        else r = v * cdf;
        Elt<T>::st(out + i, r);
    }
}
// 16-bit storage: eight elements per lane, and erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far inside a 16-bit ulp) on
// v_exp_f32 / v_rcp_f32 -- libm's erff + expf make the pass VALU-bound (49 us per ViT-B layer instead of the 19 its bytes take).
template <typename T, bool BWD>
__global__ void __launch_bounds__(256)
gelu16_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ dy, u32x4* __restrict__ out, size_t nvec) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        float v[8], g[8];
        unpack16<T>(x[i], v);
        if (BWD) unpack16<T>(dy[i], g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float z = fabsf(v[e]) * 0.70710678118654752440f;
            const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
            const float ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);             // exp(-z^2) = exp(-x^2 / 2)
            const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
            const float erfz = fmaf(-poly, ex, 1.0f);                                           // erf(|x| / sqrt 2)
            const float cdf = 0.5f * (1.0f + copysignf(erfz, v[e]));
            v[e] = BWD ? g[e] * fmaf(v[e] * 0.39894228040143267794f, ex, cdf) : v[e] * cdf;
        }
        out[i] = pack16<T>(v);
    }
}
template <typename T>
static void gelu_launch(const void* x, const void* dy, void* out, size_t n, int backward, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        if (n % 8 == 0 && ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)out) % 16 == 0) {
            const size_t nvec = n / 8;
            const int grid = (int)((nvec + 255) / 256 < 16384 ? (nvec + 255) / 256 : 16384);
            if (backward) gelu16_kernel<T, true><<<grid, 256, 0, st>>>((const u32x4*)x, (const u32x4*)dy, (u32x4*)out, nvec);
            else gelu16_kernel<T, false><<<grid, 256, 0, st>>>((const u32x4*)x, nullptr, (u32x4*)out, nvec);
            return;
        }
    }
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    gelu_kernel<T><<<grid, 256, 0, st>>>((const T*)x, (const T*)dy, (T*)out, n, backward);
}
extern "C" int advs_gelu(const void* x, void* y, long long n, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype) && x && y && n > 0, "advs_gelu: bad args");
    ADVS_SWITCH_T(dtype, gelu_launch<T>(x, nullptr, y, (size_t)n, 0, (hipStream_t)stream));
    ADVS_CHECK_LAUNCH("gelu");
    return ADVS_OK;
}
extern "C" int advs_gelu_bwd(const void* x, const void* dy, void* dx, long long n, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype) && x && dy && dx && n > 0, "advs_gelu_bwd: bad args");
    ADVS_SWITCH_T(dtype, gelu_launch<T>(x, dy, dx, (size_t)n, 1, (hipStream_t)stream));
    ADVS_CHECK_LAUNCH("gelu_bwd");
    return ADVS_OK;
}

// ---------------------------------------------------------------------------------------------- attention backward
// softmax(q k^T / sqrt d) v per (image, head), tokens padded to n rows of which the first n_valid are real keys (as
// advs_attention_masked).  Given dO: D_i = dO_i . O_i;  P recomputed from q, k;  dP_ij = dO_i . v_j;  dS = P o (dP - D);
//   dq_i = s sum_j dS_ij k_j,   dk_j = s sum_i dS_ij q_i,   dv_j = sum_i P_ij dO_i      (s = 1 / sqrt d).
// Two passes, f32 VALU arithmetic (the sequences are a few hundred tokens: 197 for ViT-B/16; the layer's GEMMs dominate):
//   pass 1, a thread per query row: online row max / sum over the keys, then a second sweep that recomputes the scores, forms
//           P and dS, accumulates dq and writes P / dS TRANSPOSED to scratch [B][heads][key j][query i] (lanes = queries: coalesced);
//   pass 2, a thread per key row: dk, dv from its scratch row.
// K / V (pass 1) and Q / dO (pass 2) of the head sit in LDS as f32 and are read as broadcasts.  D = head width rounded up
// (compile time, so q / dO / the accumulators stay in registers).
template <typename T, int D>
__global__ void __launch_bounds__(256)
attn_bwd_q_kernel(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ dO, T* __restrict__ dqkv,
                  float* __restrict__ sP, float* __restrict__ sdS, int n, int n_valid, int heads, int d, int ld, int q_off, int k_off,
                  int v_off, int head_stride, float scale, const float* __restrict__ bias, int bias_mod) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // K [n_valid][D], V [n_valid][D], zero beyond d
    float* sK = lds;
    float* sV = lds + (size_t)n_valid * D;
    const int b = blockIdx.z, hd = blockIdx.y, tid = threadIdx.x;
    const T* base = qkv + (size_t)b * n * ld;
    for (int i = tid; i < n_valid * D; i += blockDim.x) {
        const int j = i / D, c = i - j * D;
        sK[i] = c < d ? Elt<T>::ld(base + (size_t)j * ld + k_off + hd * head_stride + c) : 0.f;
        sV[i] = c < d ? Elt<T>::ld(base + (size_t)j * ld + v_off + hd * head_stride + c) : 0.f;
    }
    __syncthreads();
    const int C = heads * d;
    for (int i = blockIdx.x * blockDim.x + tid; i < n; i += gridDim.x * blockDim.x) {
        float* pc = sP + ((size_t)b * heads + hd) * n * n + i;         // element (j, i) at pc[j * n]
        float* dc = sdS + ((size_t)b * heads + hd) * n * n + i;
        T* dq = dqkv + ((size_t)b * n + i) * ld + q_off + hd * head_stride;
        if (i >= n_valid) {                                            // padding query: no gradient flows through it
            for (int j = 0; j < n; ++j) { pc[(size_t)j * n] = 0.f; dc[(size_t)j * n] = 0.f; }
            for (int c = 0; c < d; ++c) Elt<T>::st(dq + c, 0.f);
            continue;
        }
        float q[D], g[D], acc[D];
        const T* qp = base + (size_t)i * ld + q_off + hd * head_stride;
        const T* gp = dO + ((size_t)b * n + i) * C + hd * d;
        const T* op = o + ((size_t)b * n + i) * C + hd * d;
        float Dn = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            q[c] = c < d ? Elt<T>::ld(qp + c) * scale : 0.f;           // the scale rides on q
            g[c] = c < d ? Elt<T>::ld(gp + c) : 0.f;
            if (c < d) Dn = fmaf(g[c], Elt<T>::ld(op + c), Dn);
            acc[c] = 0.f;
        }
        // additive score bias (Swin), given in units of log2(e): row i of block b % bias_mod
        const float* brow = bias ? bias + (((size_t)(b % bias_mod) * heads + hd) * n + i) * (size_t)n : nullptr;
        float mx = -INFINITY, l = 0.f;
        for (int j = 0; j < n_valid; ++j) {
            float s = brow ? brow[j] * 0.6931471805599453f : 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) s = fmaf(q[c], sK[j * D + c], s);
            const float m2 = fmaxf(mx, s);
            l = l * expf(mx - m2) + expf(s - m2);
            mx = m2;
        }
        const float inv = 1.0f / l;
        for (int j = 0; j < n_valid; ++j) {
            float s = brow ? brow[j] * 0.6931471805599453f : 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) { s = fmaf(q[c], sK[j * D + c], s); dp = fmaf(g[c], sV[j * D + c], dp); }
            const float p = expf(s - mx) * inv;
            const float ds = p * (dp - Dn);
            pc[(size_t)j * n] = p; dc[(size_t)j * n] = ds;
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] = fmaf(ds, sK[j * D + c], acc[c]);
        }
        for (int j = n_valid; j < n; ++j) { pc[(size_t)j * n] = 0.f; dc[(size_t)j * n] = 0.f; }
#pragma unroll
        for (int c = 0; c < D; ++c)
            if (c < d) Elt<T>::st(dq + c, acc[c] * scale);
    }
}
template <typename T, int D>
__global__ void __launch_bounds__(256)
attn_bwd_kv_kernel(const T* __restrict__ qkv, const T* __restrict__ dO, T* __restrict__ dqkv, const float* __restrict__ sP,
                   const float* __restrict__ sdS, int n, int n_valid, int heads, int d, int ld, int q_off, int k_off, int v_off,
                   int head_stride, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // Q [n_valid][D], dO [n_valid][D]
    float* sQ = lds;
    float* sG = lds + (size_t)n_valid * D;
    const int b = blockIdx.z, hd = blockIdx.y, tid = threadIdx.x, C = heads * d;
    const T* base = qkv + (size_t)b * n * ld;
    for (int i = tid; i < n_valid * D; i += blockDim.x) {
        const int r = i / D, c = i - r * D;
        sQ[i] = c < d ? Elt<T>::ld(base + (size_t)r * ld + q_off + hd * head_stride + c) : 0.f;
        sG[i] = c < d ? Elt<T>::ld(dO + ((size_t)b * n + r) * C + hd * d + c) : 0.f;
    }
    __syncthreads();
    for (int j = blockIdx.x * blockDim.x + tid; j < n; j += gridDim.x * blockDim.x) {
        T* dk = dqkv + ((size_t)b * n + j) * ld + k_off + hd * head_stride;
        T* dv = dqkv + ((size_t)b * n + j) * ld + v_off + hd * head_stride;
        float ak[D], av[D];
#pragma unroll
        for (int c = 0; c < D; ++c) { ak[c] = 0.f; av[c] = 0.f; }
        if (j < n_valid) {
            const float* pr = sP + (((size_t)b * heads + hd) * n + j) * n;          // row j: all queries
            const float* dr = sdS + (((size_t)b * heads + hd) * n + j) * n;
            for (int i = 0; i < n_valid; ++i) {
                const float p = pr[i], ds = dr[i];
#pragma unroll
                for (int c = 0; c < D; ++c) { ak[c] = fmaf(ds, sQ[i * D + c], ak[c]); av[c] = fmaf(p, sG[i * D + c], av[c]); }
            }
        }
#pragma unroll
        for (int c = 0; c < D; ++c)
            if (c < d) { Elt<T>::st(dk + c, ak[c] * scale); Elt<T>::st(dv + c, av[c]); }
    }
}
extern "C" size_t advs_attention_bwd_scratch_bytes(int b, int n, int heads) { return (size_t)b * heads * n * n * 2 * sizeof(float); }

template <typename T, int D>
static int attn_bwd_launch(const void* qkv, const void* out, const void* d_out, void* d_qkv, float* sP, float* sdS, int b, int n,
                           int n_valid, int heads, int d, int ld, int q_off, int k_off, int v_off, int head_stride, hipStream_t st,
                           const float* bias, int bias_mod) {
    const size_t lds = (size_t)2 * n_valid * D * sizeof(float);
    ADVS_REQUIRE(lds <= 160 * 1024, "attention_bwd: %d keys x d %d do not fit the LDS (this path is for short sequences)", n_valid, d);
    static bool attr_set = false;
    if (!attr_set) {
        ADVS_HIP(hipFuncSetAttribute((const void*)attn_bwd_q_kernel<T, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ADVS_HIP(hipFuncSetAttribute((const void*)attn_bwd_kv_kernel<T, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const float scale = (float)(1.0 / sqrt((double)d));
    const dim3 grid(cdiv(n, 256), heads, b);
    attn_bwd_q_kernel<T, D><<<grid, 256, lds, st>>>((const T*)qkv, (const T*)out, (const T*)d_out, (T*)d_qkv, sP, sdS, n, n_valid, heads, d,
                                                    ld, q_off, k_off, v_off, head_stride, scale, bias, bias_mod > 0 ? bias_mod : 1);
    ADVS_CHECK_LAUNCH("attention_bwd (queries)");
    attn_bwd_kv_kernel<T, D><<<grid, 256, lds, st>>>((const T*)qkv, (const T*)d_out, (T*)d_qkv, sP, sdS, n, n_valid, heads, d, ld, q_off,
                                                     k_off, v_off, head_stride, scale);
    ADVS_CHECK_LAUNCH("attention_bwd (keys)");
    return ADVS_OK;
}
// 16-bit dtypes: the MFMA kernels of attention_bwd.hip
int attn_bwd_mfma(const void* qkv, const void* out, const void* d_out, void* d_qkv, void* scratch, int b, int n, int n_valid, int heads,
                  int d, int ld, int q_off, int k_off, int v_off, int head_stride, int dtype, hipStream_t st, const float* bias, int bias_mod);
static int attention_bwd_impl(const void* qkv, const void* out, const void* d_out, void* d_qkv, void* scratch, int b, int n, int n_valid,
                              int heads, int d, int ld, int q_off, int k_off, int v_off, int head_stride, int dtype, void* stream,
                              const float* bias, int bias_mod);

extern "C" int advs_attention_bwd(const void* qkv, const void* out, const void* d_out, void* d_qkv, void* scratch, int b, int n,
                                  int n_valid, int heads, int d, int ld, int q_off, int k_off, int v_off, int head_stride, int dtype,
                                  void* stream) {
    return attention_bwd_impl(qkv, out, d_out, d_qkv, scratch, b, n, n_valid, heads, d, ld, q_off, k_off, v_off, head_stride, dtype, stream,
                              nullptr, 1);
}
// Gradient of advs_attention_bias (Swin's windows: relative position bias + shifted-window mask inside the softmax).
extern "C" int advs_attention_bias_bwd(const void* qkv, const void* out, const void* d_out, void* d_qkv, void* scratch,
                                       const float* bias_log2e, int bias_mod, int b, int n, int heads, int d, int ld, int q_off, int k_off,
                                       int v_off, int head_stride, int dtype, void* stream) {
    ADVS_REQUIRE(bias_log2e && bias_mod > 0, "attention_bias_bwd: bias is required");
    return attention_bwd_impl(qkv, out, d_out, d_qkv, scratch, b, n, n, heads, d, ld, q_off, k_off, v_off, head_stride, dtype, stream,
                              bias_log2e, bias_mod);
}
static int attention_bwd_impl(const void* qkv, const void* out, const void* d_out, void* d_qkv, void* scratch, int b, int n, int n_valid,
                              int heads, int d, int ld, int q_off, int k_off, int v_off, int head_stride, int dtype, void* stream,
                              const float* bias, int bias_mod) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_attention_bwd: unknown dtype code %d", dtype);
    ADVS_REQUIRE(qkv && out && d_out && d_qkv && scratch && b > 0 && n > 0 && n_valid > 0 && n_valid <= n && heads > 0, "attention_bwd: bad args");
    ADVS_REQUIRE(d > 0 && d <= 64, "attention_bwd: d=%d must be in 1..64", d);
#ifdef ADVS_DIAG
    static const bool valu_only = getenv("ADVS_ATTN_BWD_V1") != nullptr;          // A/B knob for tools/ (diagnostic builds only)
#else
    constexpr bool valu_only = false;
#endif
    if (dtype != ADVS_F32 && !valu_only && d % 8 == 0 && ld % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 &&
        head_stride % 8 == 0)
        return attn_bwd_mfma(qkv, out, d_out, d_qkv, scratch, b, n, n_valid, heads, d, ld, q_off, k_off, v_off, head_stride, dtype,
                             (hipStream_t)stream, bias, bias_mod);
    float* sP = (float*)scratch;
    float* sdS = sP + (size_t)b * heads * n * n;
    if (d <= 32) ADVS_SWITCH_T(dtype, return (attn_bwd_launch<T, 32>(qkv, out, d_out, d_qkv, sP, sdS, b, n, n_valid, heads, d, ld, q_off, k_off, v_off, head_stride, (hipStream_t)stream, bias, bias_mod)));
    ADVS_SWITCH_T(dtype, return (attn_bwd_launch<T, 64>(qkv, out, d_out, d_qkv, sP, sdS, b, n, n_valid, heads, d, ld, q_off, k_off, v_off, head_stride, (hipStream_t)stream, bias, bias_mod)));
    return ADVS_ERR_ARG;                    // not reached
}

// ---------------------------------------------------------------------------------------------- head and embedding
// dtok[b][0][:] = d_cls[b][:] (f32 -> T); the other rows of dtok are left as they are (zero: the head reads the CLS row only)
template <typename T>
__global__ void scatter_row0_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, long long row_stride, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    Elt<T>::st(dst + (size_t)b * row_stride * C + c, src[i]);
}
extern "C" int advs_scatter_row0(const float* src, void* dst, int b, long long row_stride, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype) && src && dst && b > 0 && c > 0 && row_stride > 0, "advs_scatter_row0: bad args");
    ADVS_SWITCH_T(dtype, scatter_row0_kernel<T><<<cdiv((long long)b * c, 256), 256, 0, (hipStream_t)stream>>>(src, (T*)dst, b, row_stride, c));
    ADVS_CHECK_LAUNCH("scatter_row0");
    return ADVS_OK;
}
// Gradient of advs_cls_mean_rows_f32 (DINOv2's head input [cls | mean of the patch tokens]): row 0 of image b receives src[b][0..c),
// rows 1 .. np receive src[b][c..2c) / np each; rows beyond (padding) are left as they are (zero).
template <typename T>
__global__ void scatter_cls_mean_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int n_pad, int np, int C) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)B * (np + 1) * C) return;
    const int c = (int)(i % C);
    const long long br = i / C;
    const int r = (int)(br % (np + 1)), b = (int)(br / (np + 1));
    const float v = r == 0 ? src[(size_t)b * 2 * C + c] : src[(size_t)b * 2 * C + C + c] / (float)np;
    Elt<T>::st(dst + ((size_t)b * n_pad + r) * C + c, v);
}
extern "C" int advs_scatter_cls_mean(const float* src, void* dst, int b, int n_pad, int np, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype) && src && dst && b > 0 && c > 0 && np > 0 && n_pad > np, "advs_scatter_cls_mean: bad args");
    ADVS_SWITCH_T(dtype, scatter_cls_mean_kernel<T><<<cdiv((long long)b * (np + 1) * c, 256), 256, 0, (hipStream_t)stream>>>(src, (T*)dst, b, n_pad, np, c));
    ADVS_CHECK_LAUNCH("scatter_cls_mean");
    return ADVS_OK;
}
// inverse of advs_patchify_padded for gradients: dx[b][c][py*ps+ky][px*ps+kx] = dcols[b][row_off + py*gw + px][(c*ps+ky)*ps+kx],
// dcols rows `rows_per_image` apart per image and kpad long (the token-gradient layout: row_off = 1 skips the CLS row).
template <typename T>
__global__ void unpatchify_kernel(const T* __restrict__ dcols, float* __restrict__ dx, int B, int Cin, int H, int W, int ps, int Kp,
                                  int rows_per_image, int row_off) {
    const int gw = W / ps;
    const size_t total = (size_t)B * Cin * H * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t r = i / W;
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % Cin);
        const int b = (int)(r / Cin);
        const int py = y / ps, ky = y - py * ps, px = x / ps, kx = x - px * ps;
        dx[i] = Elt<T>::ld(dcols + ((size_t)b * rows_per_image + row_off + py * gw + px) * Kp + (c * ps + ky) * ps + kx);
    }
}
extern "C" int advs_unpatchify_padded(const void* dcols, float* dx_nchw, int b, int cin, int h, int w, int patch, int kpad,
                                      int rows_per_image, int row_off, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype) && dcols && dx_nchw && b > 0 && cin > 0 && patch > 0 && h % patch == 0 && w % patch == 0, "unpatchify: bad args");
    ADVS_REQUIRE(kpad >= cin * patch * patch && row_off >= 0 && rows_per_image >= row_off + (h / patch) * (w / patch), "unpatchify: bad row layout");
    const size_t total = (size_t)b * cin * h * w;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ADVS_SWITCH_T(dtype, unpatchify_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)dcols, dx_nchw, b, cin, h, w, patch, kpad,
                                                                                   rows_per_image, row_off));
    ADVS_CHECK_LAUNCH("unpatchify");
    return ADVS_OK;
}
