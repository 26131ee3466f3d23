// Spatial resampling and token normalisation of the class-conditional UNet (model/networks/unet.py):
// MaxPool2d(2), bilinear x2 upsample (align_corners=True) fused with the skip concat, LayerNorm.
// All HBM-bound streaming kernels on NHWC activations, 16 bytes per lane.
#include "common.h"

// ---------------------------------------------------------------- MaxPool2d(2)  (block.py:27)
template <typename T>
__global__ void maxpool2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    constexpr int VEC = Elt<T>::VEC;
    const int vpp = C / VEC, Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * vpp;
    const u32x4* xv = (const u32x4*)x;
    u32x4* yv = (u32x4*)y;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const size_t base = (((size_t)b * H + 2 * oy) * W + 2 * ox) * vpp + cv;
        float a[VEC], t[VEC];
        unpack16<T>(xv[base], a);
        unpack16<T>(xv[base + vpp], t);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] = fmaxf(a[j], t[j]);
        unpack16<T>(xv[base + (size_t)W * vpp], t);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] = fmaxf(a[j], t[j]);
        unpack16<T>(xv[base + (size_t)W * vpp + vpp], t);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] = fmaxf(a[j], t[j]);
        yv[i] = pack16<T>(a);
    }
}

extern "C" int advs_maxpool2(const void* x, void* y, int b, int h, int w, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_maxpool2: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && y && b > 0 && h > 1 && w > 1 && h % 2 == 0 && w % 2 == 0, "maxpool2: bad shape");
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "maxpool2: c=%d must be a multiple of %d", c, vec);
    const size_t total = (size_t)b * (h / 2) * (w / 2) * (c / vec);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    ADVS_SWITCH_T(dtype, maxpool2_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, (T*)y, b, h, w, c));
    ADVS_CHECK_LAUNCH("maxpool2");
    return ADVS_OK;
}

// ------------------------------------------- cat([skip, Upsample(x2, bilinear, align_corners)(x)])
// (block.py:66,86-87).  Source index = dst * (in-1)/(out-1), weights as torch's CPU kernel forms them.
// NEAREST: cat([skip, Upsample(x2, nearest)(x)]) of CSPDarkUpBlock (block.py:116,127-128): source = dst >> 1.
template <typename T, bool NEAREST>
__global__ void concat_up_kernel(const T* __restrict__ skip, const T* __restrict__ x, T* __restrict__ y,
                                 int B, int h, int w, int C1, int C2) {
    constexpr int VEC = Elt<T>::VEC;
    const int H = 2 * h, W = 2 * w, v1 = C1 / VEC, v2 = C2 / VEC, vpp = v1 + v2;
    const size_t total = (size_t)B * H * W * vpp;
    const float sh = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sw = w > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const u32x4* sv = (const u32x4*)skip;
    const u32x4* xv = (const u32x4*)x;
    u32x4* yv = (u32x4*)y;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;
        const int ox = (int)(r % W); r /= W;
        const int oy = (int)(r % H);
        const int b = (int)(r / H);
        if (cv < v1) {
            yv[i] = sv[(((size_t)b * H + oy) * W + ox) * v1 + cv];
            continue;
        }
        if (NEAREST) {
            yv[i] = xv[(((size_t)b * h + (oy >> 1)) * w + (ox >> 1)) * v2 + (cv - v1)];
            continue;
        }
        const float fy = sh * oy, fx = sw * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const int c = cv - v1;
        float p00[VEC], p01[VEC], p10[VEC], p11[VEC], o[VEC];
        unpack16<T>(xv[(((size_t)b * h + y0) * w + x0) * v2 + c], p00);
        unpack16<T>(xv[(((size_t)b * h + y0) * w + x1) * v2 + c], p01);
        unpack16<T>(xv[(((size_t)b * h + y1) * w + x0) * v2 + c], p10);
        unpack16<T>(xv[(((size_t)b * h + y1) * w + x1) * v2 + c], p11);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = hy * (hx * p00[j] + lx * p01[j]) + ly * (hx * p10[j] + lx * p11[j]);
        yv[i] = pack16<T>(o);
    }
}

template <bool NEAREST>
static int concat_up_launch(const char* name, const void* skip, const void* x, void* y, int b, int h, int w, int c1, int c2,
                            int dtype, void* stream) {
    ADVS_REQUIRE(skip && x && y && b > 0 && h > 0 && w > 0, "%s: bad args", name);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c1 % vec == 0 && c2 % vec == 0 && c2 > 0, "%s: channels must be multiples of %d", name, vec);
    const size_t total = (size_t)b * 4 * h * w * ((c1 + c2) / vec);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ADVS_SWITCH_T(dtype, concat_up_kernel<T, NEAREST><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)skip, (const T*)x, (T*)y, b, h, w, c1, c2));
    ADVS_CHECK_LAUNCH(name);
    return ADVS_OK;
}

extern "C" int advs_concat_upsample2x(const void* skip, const void* x, void* y, int b, int h, int w, int c1, int c2,
                                      int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_concat_upsample2x: unknown dtype code %d", dtype);
    return concat_up_launch<false>("concat_upsample2x", skip, x, y, b, h, w, c1, c2, dtype, stream);
}
extern "C" int advs_concat_nearest2x(const void* skip, const void* x, void* y, int b, int h, int w, int c1, int c2,
                                     int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_concat_nearest2x: unknown dtype code %d", dtype);
    return concat_up_launch<true>("concat_nearest2x", skip, x, y, b, h, w, c1, c2, dtype, stream);
}

// ---------------------------------------------------------------- LayerNorm over the channel axis
// (attention.py:25,27): one token row per 16-lane group (C <= 2048), eps = 1e-5.
template <typename T>
__global__ void __launch_bounds__(256)
layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                 T* __restrict__ y, long long rows, int C, float eps) {
    constexpr int VEC = Elt<T>::VEC;
    const int l16 = threadIdx.x & 15;
    const int vpr = C / VEC;
    const long long row0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long stride = ((long long)gridDim.x * blockDim.x) >> 4;
    for (long long row = row0; row < rows; row += stride) {            // a 16-lane group shares one row
        const bool live = true;
        const u32x4* xr = (const u32x4*)(x + (size_t)row * C);
        float s = 0.f;
        for (int cv = l16; cv < vpr && live; cv += 16) {
            float f[VEC];
            unpack16<T>(xr[cv], f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s += f[j];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.f;
        for (int cv = l16; cv < vpr && live; cv += 16) {
            float f[VEC];
            unpack16<T>(xr[cv], f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const float d = f[j] - mean; q = fmaf(d, d, q); }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
        if (!live) continue;
        u32x4* yr = (u32x4*)(y + (size_t)row * C);
        for (int cv = l16; cv < vpr; cv += 16) {
            float f[VEC];
            unpack16<T>(xr[cv], f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int c = cv * VEC + j;
                f[j] = (f[j] - mean) * rstd * gamma[c] + beta[c];
            }
            yr[cv] = pack16<T>(f);
        }
    }
}

extern "C" int advs_layernorm(const void* x, const float* gamma, const float* beta, void* y, long long rows, int c,
                              float eps, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_layernorm: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && gamma && beta && y && rows > 0 && c > 0, "layernorm: bad args");
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "layernorm: c=%d must be a multiple of %d", c, vec);
    long long blocks = (rows + 15) / 16;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);
    ADVS_SWITCH_T(dtype, layernorm_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, gamma, beta, (T*)y, rows, c, eps));
    ADVS_CHECK_LAUNCH("layernorm");
    return ADVS_OK;
}
