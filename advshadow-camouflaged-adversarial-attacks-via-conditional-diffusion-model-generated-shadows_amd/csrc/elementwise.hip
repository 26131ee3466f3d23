// Elementwise / tiny kernels: packing, layout conversion, the time-embedding path, the
// DDIM update.  HBM-bound streaming work; compiled with -ffp-contract=off so every
// multiply/add rounds exactly like the reference's separate torch ops.
#include "common.h"

// ------------------------------------------------------------------ weight packing
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ out,
                                        int cout, int cin, int rs) {
    size_t total = (size_t)cout * cin * rs;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(i % cin);
        size_t r = i / cin;
        int t = (int)(r % rs);
        int o = (int)(r / rs);
        Elt<T>::st(out + i, w[((size_t)o * cin + c) * rs + t]);
    }
}

extern "C" int advs_pack_conv_weight(const float* w, void* out, int cout, int cin, int r, int s,
                                     int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_pack_conv_weight: unknown dtype code %d", dtype);
    ADVS_REQUIRE(w && out && cout > 0 && cin > 0 && r > 0 && s > 0, "pack_conv_weight: bad args");
    size_t total = (size_t)cout * cin * r * s;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    ADVS_SWITCH_T(dtype, pack_conv_weight_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(w, (T*)out, cout, cin, r * s));
    ADVS_CHECK_LAUNCH("pack_conv_weight");
    return ADVS_OK;
}

// ------------------------------------------------------------------ layout conversion
// One block moves a [32 pixels][32 channels] patch through LDS so both sides coalesce.
template <typename T, bool TO_NHWC>
__global__ void layout_kernel(const void* __restrict__ src, void* __restrict__ dst, int C, int HW) {
    __shared__ float tile[32][33];
    int b = blockIdx.z;
    int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 256 threads: 8 rows per pass
    if (TO_NHWC) {
        const float* x = (const float*)src + (size_t)b * C * HW;
        for (int j = ty; j < 32; j += 8) {
            int c = c0 + j, p = p0 + tx;
            tile[j][tx] = (c < C && p < HW) ? x[(size_t)c * HW + p] : 0.f;
        }
        __syncthreads();
        T* y = (T*)dst + (size_t)b * C * HW;
        for (int j = ty; j < 32; j += 8) {
            int p = p0 + j, c = c0 + tx;
            if (c < C && p < HW) Elt<T>::st(y + (size_t)p * C + c, tile[tx][j]);
        }
    } else {
        const T* x = (const T*)src + (size_t)b * C * HW;
        for (int j = ty; j < 32; j += 8) {
            int p = p0 + j, c = c0 + tx;
            tile[j][tx] = (c < C && p < HW) ? Elt<T>::ld(x + (size_t)p * C + c) : 0.f;
        }
        __syncthreads();
        float* y = (float*)dst + (size_t)b * C * HW;
        for (int j = ty; j < 32; j += 8) {
            int c = c0 + j, p = p0 + tx;
            if (c < C && p < HW) y[(size_t)c * HW + p] = tile[tx][j];
        }
    }
}

extern "C" int advs_nchw_f32_to_nhwc(const float* x, void* y, int b, int c, int h, int w, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_nchw_f32_to_nhwc: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && y && b > 0 && c > 0 && h > 0 && w > 0, "nchw_to_nhwc: bad args");
    dim3 grid(cdiv((long long)h * w, 32), cdiv(c, 32), b);
    ADVS_SWITCH_T(dtype, layout_kernel<T, true><<<grid, 256, 0, (hipStream_t)stream>>>(x, y, c, h * w));
    ADVS_CHECK_LAUNCH("nchw_to_nhwc");
    return ADVS_OK;
}
extern "C" int advs_nhwc_to_nchw_f32(const void* x, float* y, int b, int c, int h, int w, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_nhwc_to_nchw_f32: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && y && b > 0 && c > 0 && h > 0 && w > 0, "nhwc_to_nchw: bad args");
    dim3 grid(cdiv((long long)h * w, 32), cdiv(c, 32), b);
    ADVS_SWITCH_T(dtype, layout_kernel<T, false><<<grid, 256, 0, (hipStream_t)stream>>>(x, y, c, h * w));
    ADVS_CHECK_LAUNCH("nhwc_to_nchw");
    return ADVS_OK;
}

// ------------------------------------------------------------------ small linear (f32)
// One wave per output element row-chunk: y[b][n] = bias[n] + sum_k act(x[b][k]) w[n][k].
// Tiny (<= a few MFLOP per step); lanes stride k so the w row is read coalesced.
__global__ void linear_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ bias, float* __restrict__ y,
                                  int B, int K, int N, int act_in, int act_out) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if (wave >= B * N) return;
    int b = wave / N, n = wave % N;
    const float* xr = x + (size_t)b * K;
    const float* wr = w + (size_t)n * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(apply_act(xr[k], act_in), wr[k], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[(size_t)b * N + n] = apply_act(acc + (bias ? bias[n] : 0.f), act_out);
}

extern "C" int advs_linear_f32(const float* x, const float* w, const float* bias, float* y, int b, int k, int n,
                               int act_in, int act_out, void* stream) {
    ADVS_REQUIRE(x && w && y && b > 0 && k > 0 && n > 0, "linear_f32: bad args");
    long long waves = (long long)b * n;
    linear_f32_kernel<<<cdiv(waves, 4), 256, 0, (hipStream_t)stream>>>(x, w, bias, y, b, k, n, act_in, act_out);
    ADVS_CHECK_LAUNCH("linear_f32");
    return ADVS_OK;
}

// ------------------------------------------------------------------ sinusoidal embedding
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs,
                                          int half, int cos_first, const float* __restrict__ table,
                                          const int64_t* __restrict__ labels, float* __restrict__ out, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    int b = i / half, j = i % half;
    float arg = (float)t[b] * freqs[j];
    float c = cosf(arg), s = sinf(arg);
    float* o = out + (size_t)b * 2 * half;
    float first = cos_first ? c : s, second = cos_first ? s : c;
    if (table && labels) {
        const float* row = table + (size_t)labels[b] * 2 * half;
        first += row[j];
        second += row[half + j];
    }
    o[j] = first;
    o[half + j] = second;
}

extern "C" int advs_timestep_embedding(const int64_t* t, const float* freqs, int half, int cos_first,
                                       const float* emb_table, const int64_t* labels, float* out, int b,
                                       void* stream) {
    ADVS_REQUIRE(t && freqs && out && half > 0 && b > 0, "timestep_embedding: bad args");
    timestep_embedding_kernel<<<cdiv((long long)b * half, 256), 256, 0, (hipStream_t)stream>>>(
        t, freqs, half, cos_first, emb_table, labels, out, b);
    ADVS_CHECK_LAUNCH("timestep_embedding");
    return ADVS_OK;
}

// ------------------------------------------------------------------ DDIM update
// 16 B per lane, grid-stride; x, eps (and eps_u, noise) are read once and x written once:
// algorithmic traffic 12 B/element (20 B with CFG).
__global__ void ddim_step_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                 const float* __restrict__ eps_u, float cfg, const float* __restrict__ noise,
                                 const float* __restrict__ coef, const int32_t* __restrict__ step_counter,
                                 size_t n4, int clip) {
    int step = *step_counter;
    float a_t = coef[3 * step], a_p = coef[3 * step + 1], sigma = coef[3 * step + 2];
    float s1mat = sqrtf(1.0f - a_t), sat = sqrtf(a_t), sap = sqrtf(a_p);
    float c2 = sqrtf((1.0f - a_p) - sigma * sigma);
    bool small = fabsf(cfg) < 0.5f;
    float coeff = small ? cfg : cfg - 1.0f;
    const f32x4* e4 = (const f32x4*)eps;
    const f32x4* u4 = (const f32x4*)eps_u;
    const f32x4* z4 = (const f32x4*)noise;
    f32x4* x4 = (f32x4*)x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 xv = x4[i], ev = e4[i], r;
        f32x4 uv, zv;
        if (eps_u) uv = u4[i];
        if (noise) zv = z4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float e = ev[j];
            if (eps_u) {                       // torch.lerp(eps_u, eps_c, w): fma(coeff, end-start, base)
                float d = e - uv[j];
                e = fmaf(coeff, d, small ? uv[j] : e);
            }
            float x0 = (xv[j] - s1mat * e) / sat;
            if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            float v = sap * x0 + c2 * e;
            v = v + sigma * (noise ? zv[j] : 0.0f);
            r[j] = v;
        }
        x4[i] = r;
    }
}

__global__ void ddim_advance_kernel(int32_t* step_counter, const int64_t* __restrict__ tseq, int nsteps,
                                    int64_t* __restrict__ t_out, int B) {
    int next = *step_counter + 1;
    int64_t t = tseq[next < nsteps ? next : nsteps - 1];
    for (int i = threadIdx.x; i < B; i += blockDim.x) t_out[i] = t;
    __syncthreads();
    if (threadIdx.x == 0) *step_counter = next;
}

extern "C" int advs_ddim_step(float* x, const float* eps, const float* eps_uncond, float cfg_scale,
                              const float* noise, const float* coef, const int64_t* tseq, int nsteps,
                              int32_t* step_counter, int64_t* t_out, int b, size_t per_sample, int clip,
                              void* stream) {
    ADVS_REQUIRE(x && eps && coef && tseq && step_counter && t_out && b > 0 && nsteps > 0, "ddim_step: bad args");
    size_t n = (size_t)b * per_sample;
    ADVS_REQUIRE(n % 4 == 0, "ddim_step: element count %zu not a multiple of 4", n);
    size_t n4 = n / 4;
    int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    ddim_step_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, eps, eps_uncond, cfg_scale, noise, coef,
                                                             step_counter, n4, clip);
    ADVS_CHECK_LAUNCH("ddim_step");
    ddim_advance_kernel<<<1, 256, 0, (hipStream_t)stream>>>(step_counter, tseq, nsteps, t_out, b);
    ADVS_CHECK_LAUNCH("ddim_advance");
    return ADVS_OK;
}

// ------------------------------------------------------------------ uint8 cast
__global__ void to_uint8_kernel(const float* __restrict__ x, uint8_t* __restrict__ y, size_t n, int clamp) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = (x[i] + 1.0f) * 0.5f;
        v = v * 255.0f;
        if (clamp) v = fminf(fmaxf(v, 0.0f), 255.0f);
        // C truncation toward zero, then wrap mod 256 (torch .type(torch.uint8) on CPU)
        long long q = (long long)v;
        y[i] = (uint8_t)(q & 0xff);
    }
}

extern "C" int advs_to_uint8(const float* x, uint8_t* y, size_t n, int clamp, void* stream) {
    ADVS_REQUIRE(x && y && n > 0, "to_uint8: bad args");
    int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    to_uint8_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, n, clamp);
    ADVS_CHECK_LAUNCH("to_uint8");
    return ADVS_OK;
}

// [0,1] float image -> uint8 as torchvision's ToPILImage does (pic.mul(255).byte(): truncation; values are
// clamped first because an out-of-range float->uint8 cast is not portable).
__global__ void unit_to_uint8_kernel(const float* __restrict__ x, uint8_t* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = x[i] * 255.0f;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        y[i] = (uint8_t)(int)v;
    }
}
extern "C" int advs_unit_to_uint8(const float* x, uint8_t* y, size_t n, void* stream) {
    ADVS_REQUIRE(x && y && n > 0, "unit_to_uint8: bad args");
    int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    unit_to_uint8_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, n);
    ADVS_CHECK_LAUNCH("unit_to_uint8");
    return ADVS_OK;
}

// ------------------------------------------------------------------ DDPM posterior step (diff_model.py:361-396)
// GaussianDiffusion.p_sample: x_recon = clamp(sr[t] x - srm1[t] eps); mean = c1[t] x_recon + c2[t] x;
// x <- mean + mask(t != 0) * exp(0.5 logvar[t]) * noise.  coef rows (per STEP, in loop order) hold
// {sr, srm1, c1, c2, mask * exp(0.5 logvar)} as the host computed them with the reference's own f32 torch ops,
// and the file is built with -ffp-contract=off: the result is bit-exact with the reference's op chain.
__global__ void ddpm_posterior_step_kernel(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ noise,
                                           const float* __restrict__ coef, const int32_t* __restrict__ step_counter,
                                           size_t n4, int clip) {
    const int step = *step_counter;
    const float sr = coef[5 * step], srm1 = coef[5 * step + 1], c1 = coef[5 * step + 2], c2 = coef[5 * step + 3],
                sg = coef[5 * step + 4];
    const f32x4* e4 = (const f32x4*)eps;
    const f32x4* z4 = (const f32x4*)noise;
    f32x4* x4 = (f32x4*)x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 xv = x4[i], ev = e4[i], zv = z4[i];
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x0 = sr * xv[j] - srm1 * ev[j];
            if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            const float mean = c1 * x0 + c2 * xv[j];
            r[j] = mean + sg * zv[j];
        }
        x4[i] = r;
    }
}

extern "C" int advs_ddpm_posterior_step(float* x, const float* eps, const float* noise, const float* coef,
                                        const int64_t* tseq, int nsteps, int32_t* step_counter, int64_t* t_out,
                                        int b, size_t per_sample, int clip, void* stream) {
    ADVS_REQUIRE(x && eps && noise && coef && tseq && step_counter && t_out && b > 0 && per_sample > 0 && nsteps > 0,
                 "ddpm_posterior_step: bad args");
    const size_t n = (size_t)b * per_sample;
    ADVS_REQUIRE(n % 4 == 0, "ddpm_posterior_step: element count %zu not a multiple of 4", n);
    const size_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    ddpm_posterior_step_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, eps, noise, coef, step_counter, n4, clip);
    ADVS_CHECK_LAUNCH("ddpm_posterior_step");
    ddim_advance_kernel<<<1, 256, 0, (hipStream_t)stream>>>(step_counter, tseq, nsteps, t_out, b);
    ADVS_CHECK_LAUNCH("ddim_advance");
    return ADVS_OK;
}

// ------------------------------------------------------------------ DDPM ancestral update (model/samples/ddpm.py:86-88)
//   eps' = lerp(eps_u, eps, cfg) (optional);  x = 1/sqrt(alpha) * (x - ((1-alpha)/sqrt(1-alpha_hat)) * eps') + sqrt(beta)*noise
// coef[step] = {alpha, alpha_hat, beta}.  Same device-side step counter / timestep hand-off as advs_ddim_step.
__global__ void ddpm_step_kernel(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ eps_u,
                                 float cfg, const float* __restrict__ noise, const float* __restrict__ coef,
                                 const int32_t* __restrict__ step_counter, size_t n4) {
    const int step = *step_counter;
    const float alpha = coef[3 * step], ahat = coef[3 * step + 1], beta = coef[3 * step + 2];
    const float inv = 1.0f / sqrtf(alpha);
    const float ce = (1.0f - alpha) / sqrtf(1.0f - ahat);
    const float sb = sqrtf(beta);
    const bool small = fabsf(cfg) < 0.5f;
    const float coeff = small ? cfg : cfg - 1.0f;
    const f32x4* e4 = (const f32x4*)eps;
    const f32x4* u4 = (const f32x4*)eps_u;
    const f32x4* z4 = (const f32x4*)noise;
    f32x4* x4 = (f32x4*)x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 xv = x4[i], ev = e4[i], r, uv, zv;
        if (eps_u) uv = u4[i];
        if (noise) zv = z4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float e = ev[j];
            if (eps_u) { const float d = e - uv[j]; e = fmaf(coeff, d, small ? uv[j] : e); }
            const float v = inv * (xv[j] - ce * e);
            r[j] = v + sb * (noise ? zv[j] : 0.0f);
        }
        x4[i] = r;
    }
}

extern "C" int advs_ddpm_step(float* x, const float* eps, const float* eps_uncond, float cfg_scale, const float* noise,
                              const float* coef, const int64_t* tseq, int nsteps, int32_t* step_counter,
                              int64_t* t_out, int b, size_t per_sample, void* stream) {
    ADVS_REQUIRE(x && eps && coef && tseq && step_counter && t_out && b > 0 && nsteps > 0, "ddpm_step: bad args");
    const size_t n = (size_t)b * per_sample;
    ADVS_REQUIRE(n % 4 == 0, "ddpm_step: element count %zu not a multiple of 4", n);
    const size_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    ddpm_step_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, eps, eps_uncond, cfg_scale, noise, coef, step_counter, n4);
    ADVS_CHECK_LAUNCH("ddpm_step");
    ddim_advance_kernel<<<1, 256, 0, (hipStream_t)stream>>>(step_counter, tseq, nsteps, t_out, b);
    ADVS_CHECK_LAUNCH("ddpm_advance");
    return ADVS_OK;
}

// ------------------------------------------------------------------ PLMS multistep combine (model/samples/plms.py:93-107)
// out = lerp(eps_u, eps, cfg) first (optional), then by the number of stored predictions `order`:
//   0 with `next`: (e + next)/2 ; 1: (3e - o1)/2 ; 2: (23e - 16 o1 + 5 o2)/12 ; 3+: (55e - 59 o1 + 37 o2 - 9 o3)/24
// `guided` receives the lerped eps (what the reference appends to old_eps), `out` the combined one.
__global__ void plms_combine_kernel(const float* __restrict__ eps, const float* __restrict__ eps_u, float cfg,
                                    const float* __restrict__ nxt, const float* __restrict__ o1, const float* __restrict__ o2,
                                    const float* __restrict__ o3, int order, float* __restrict__ guided,
                                    float* __restrict__ out, size_t n) {
    const bool small = fabsf(cfg) < 0.5f;
    const float coeff = small ? cfg : cfg - 1.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float e = eps[i];
        if (eps_u) { const float u = eps_u[i], d = e - u; e = fmaf(coeff, d, small ? u : e); }
        if (guided) guided[i] = e;
        float r;
        if (order == 0) r = nxt ? (e + nxt[i]) / 2.0f : e;
        else if (order == 1) r = (3.0f * e - o1[i]) / 2.0f;
        else if (order == 2) r = (23.0f * e - 16.0f * o1[i] + 5.0f * o2[i]) / 12.0f;
        else r = (55.0f * e - 59.0f * o1[i] + 37.0f * o2[i] - 9.0f * o3[i]) / 24.0f;
        out[i] = r;
    }
}
extern "C" int advs_plms_combine(const float* eps, const float* eps_uncond, float cfg_scale, const float* eps_next,
                                 const float* old1, const float* old2, const float* old3, int order, float* guided,
                                 float* out, size_t n, void* stream) {
    ADVS_REQUIRE(eps && out && n > 0 && order >= 0, "plms_combine: bad args");
    ADVS_REQUIRE(order < 1 || old1, "plms_combine: order %d needs old1", order);
    ADVS_REQUIRE(order < 2 || old2, "plms_combine: order %d needs old2", order);
    ADVS_REQUIRE(order < 3 || old3, "plms_combine: order %d needs old3", order);
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    plms_combine_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(eps, eps_uncond, cfg_scale, eps_next, old1, old2, old3, order,
                                                               guided, out, n);
    ADVS_CHECK_LAUNCH("plms_combine");
    return ADVS_OK;
}
