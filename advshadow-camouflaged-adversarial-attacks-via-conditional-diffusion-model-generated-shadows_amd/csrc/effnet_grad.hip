// EfficientNetV2-S victim run backwards (torchvision efficientnet_v2_s of ASR_fast.py:59-65 in the gradient attack of
// tools/train_shadow.py:177-221): the non-GEMM data-gradient pieces -- SiLU both ways as its own pass (the gradient needs the
// pre-activation, so the attack's forward does not fuse it into the conv epilogue), the strided depthwise gradient, and the
// squeeze-and-excitation block backwards.  HBM-bound single passes over NHWC activations, f32 arithmetic.
#include "common.h"

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_grad(float x) { const float s = sigmoid_f(x); return s * (1.0f + x * (1.0f - s)); }

// y = silu(x) (+ add);   backward: dx = dy * silu'(x) with x the PRE-activation
template <typename T, int MODE>
__global__ void silu_kernel(const T* __restrict__ x, const T* __restrict__ other, T* __restrict__ out, size_t nvec) {
    constexpr int VEC = Elt<T>::VEC;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        float v[VEC], o[VEC];
        unpack16<T>(((const u32x4*)x)[i], v);
        if (other) unpack16<T>(((const u32x4*)other)[i], o);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (MODE == 0) v[e] = v[e] * sigmoid_f(v[e]) + (other ? o[e] : 0.f);
            else v[e] = o[e] * silu_grad(v[e]);
        }
        ((u32x4*)out)[i] = pack16<T>(v);
    }
}
static int silu_launch(const void* x, const void* other, void* out, long long n, int mode, int dtype, void* stream, const char* what) {
    ADVS_REQUIRE(dtype_ok(dtype) && x && out && n > 0 && (mode == 0 || other), "%s: bad args", what);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(n % vec == 0, "%s: n=%lld must be a multiple of %d", what, n, vec);
    const size_t nvec = (size_t)n / vec;
    const int grid = (int)((nvec + 255) / 256 < 16384 ? (nvec + 255) / 256 : 16384);
    if (mode == 0) { ADVS_SWITCH_T(dtype, silu_kernel<T, 0><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, (const T*)other, (T*)out, nvec)); }
    else { ADVS_SWITCH_T(dtype, silu_kernel<T, 1><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, (const T*)other, (T*)out, nvec)); }
    ADVS_CHECK_LAUNCH(what);
    return ADVS_OK;
}
extern "C" int advs_silu(const void* x, const void* add, void* y, long long n, int dtype, void* stream) {
    return silu_launch(x, add, y, n, 0, dtype, stream, "advs_silu");
}
extern "C" int advs_silu_bwd(const void* x_pre, const void* dy, void* dx, long long n, int dtype, void* stream) {
    return silu_launch(x_pre, dy, dx, n, 1, dtype, stream, "advs_silu_bwd");
}

// Data gradient of advs_dwconv2d for stride 1 or 2 ('same' padding k/2): dx has the conv's INPUT size h x w, dy its output size.
//   dx[b][y][x][c] = sum over taps (ky, kx) with (y + p - ky) and (x + p - kx) multiples of the stride of
//                    dy[b][(y + p - ky) / s][(x + p - kx) / s][c] * w[ky*k + kx][c]
template <typename T>
__global__ void __launch_bounds__(256)
dwconv_bwd_strided_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int B, int H, int W, int C, int K,
                          int stride, int Ho, int Wo) {
    constexpr int VEC = Elt<T>::VEC;
    const int vpp = C / VEC, pad = K / 2;
    const size_t total = (size_t)B * H * W * vpp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int ky = 0; ky < K; ++ky) {
            const int ty = y + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int sy = ty / stride;
            if (sy >= Ho) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int tx = x + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int sx = tx / stride;
                if (sx >= Wo) continue;
                float f[VEC];
                unpack16<T>(*((const u32x4*)(dy + (((size_t)b * Ho + sy) * Wo + sx) * C) + cv), f);
                const float* wt = w + (size_t)(ky * K + kx) * C + cv * VEC;
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(f[e], wt[e], acc[e]);
            }
        }
        *((u32x4*)(dx + (((size_t)b * H + y) * W + x) * C) + cv) = pack16<T>(acc);
    }
}
extern "C" int advs_dwconv2d_bwd_strided(const void* dy, const float* w_taps_c, void* dx, int b, int h, int w, int c, int ksize, int stride,
                                         int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_dwconv2d_bwd_strided: unknown dtype code %d", dtype);
    ADVS_REQUIRE(dy && w_taps_c && dx && b > 0 && h > 0 && w > 0 && c > 0, "dwconv2d_bwd_strided: bad args");
    ADVS_REQUIRE((ksize & 1) && ksize >= 1 && ksize <= 7 && (stride == 1 || stride == 2), "dwconv2d_bwd_strided: ksize %d / stride %d unsupported", ksize, stride);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "dwconv2d_bwd_strided: c=%d must be a multiple of %d", c, vec);
    const int ho = (h + 2 * (ksize / 2) - ksize) / stride + 1, wo = (w + 2 * (ksize / 2) - ksize) / stride + 1;
    const size_t total = (size_t)b * h * w * (c / vec);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADVS_SWITCH_T(dtype, dwconv_bwd_strided_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)dy, w_taps_c, (T*)dx, b, h, w, c, ksize, stride, ho, wo));
    ADVS_CHECK_LAUNCH("dwconv2d_bwd_strided");
    return ADVS_OK;
}

// out[b][c] = sum_p a[b][p][c] * bb[b][p][c] (f32): the gradient reaching the squeeze-excitation scale s (y = d * s).
// grid (channel chunks of 32 vectors, b); 256 threads = 32 channel vectors x 8 pixel lanes.
template <typename T>
__global__ void __launch_bounds__(256)
channel_dot_kernel(const T* __restrict__ a, const T* __restrict__ bb, float* __restrict__ out, int HW, int C) {
    constexpr int VEC = Elt<T>::VEC;
    __shared__ float red[8][32][VEC];
    const int vpp = C / VEC, b = blockIdx.y;
    const int cvi = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int cv = blockIdx.x * 32 + cvi;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    if (cv < vpp) {
        for (int p = pl; p < HW; p += 8) {
            float fa[VEC], fb[VEC];
            unpack16<T>(*((const u32x4*)(a + ((size_t)b * HW + p) * C) + cv), fa);
            unpack16<T>(*((const u32x4*)(bb + ((size_t)b * HW + p) * C) + cv), fb);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fmaf(fa[e], fb[e], acc[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[pl][cvi][e] = acc[e];
    __syncthreads();
    if (pl == 0 && cv < vpp) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float s = 0.f;
            for (int k = 0; k < 8; ++k) s += red[k][cvi][e];           // fixed order: bit-reproducible
            out[(size_t)b * C + cv * VEC + e] = s;
        }
    }
}
extern "C" int advs_channel_dot(const void* a, const void* bb, float* out, int b, int hw, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype) && a && bb && out && b > 0 && hw > 0 && c > 0, "advs_channel_dot: bad args");
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "channel_dot: c=%d must be a multiple of %d", c, vec);
    const dim3 grid(cdiv(c / vec, 32), b);
    ADVS_SWITCH_T(dtype, channel_dot_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)a, (const T*)bb, out, hw, c));
    ADVS_CHECK_LAUNCH("channel_dot");
    return ADVS_OK;
}

// Gradient through the squeeze-excitation gate s = sigmoid(z2): out = gs * s * (1 - s) (f32, [b][c] flattened).  The two small
// Linear layers of the block run backwards as advs_linear_f32 on transposed weights, with advs_silu_bwd (f32) between them.
__global__ void sigmoid_gate_bwd_kernel(const float* __restrict__ gs, const float* __restrict__ s, float* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float sv = s[i];
        out[i] = gs[i] * sv * (1.0f - sv);
    }
}
extern "C" int advs_sigmoid_gate_bwd(const float* gs, const float* s, float* out, long long n, void* stream) {
    ADVS_REQUIRE(gs && s && out && n > 0, "advs_sigmoid_gate_bwd: bad args");
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    sigmoid_gate_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(gs, s, out, (size_t)n);
    ADVS_CHECK_LAUNCH("sigmoid_gate_bwd");
    return ADVS_OK;
}

// Gradient at the depthwise conv's pre-activation through y = d * s (d = silu(pre)):
//   out[b][p][c] = (dsc[b][p][c] * s[b][c] + dpooled[b][c] / hw) * silu'(pre[b][p][c])
template <typename T>
__global__ void se_scale_bwd_kernel(const T* __restrict__ dsc, const float* __restrict__ s, const float* __restrict__ dpooled,
                                    const T* __restrict__ pre, T* __restrict__ out, int B, int HW, int C) {
    constexpr int VEC = Elt<T>::VEC;
    const int vpp = C / VEC;
    const size_t total = (size_t)B * HW * vpp;
    const float inv = 1.0f / (float)HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        const int b = (int)(i / ((size_t)HW * vpp));
        float g[VEC], x[VEC];
        unpack16<T>(((const u32x4*)dsc)[i], g);
        unpack16<T>(((const u32x4*)pre)[i], x);
        const float* sp = s + (size_t)b * C + cv * VEC;
        const float* dp = dpooled + (size_t)b * C + cv * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] = fmaf(g[e], sp[e], dp[e] * inv) * silu_grad(x[e]);
        ((u32x4*)out)[i] = pack16<T>(g);
    }
}
extern "C" int advs_se_scale_bwd(const void* dsc, const float* s, const float* dpooled, const void* pre, void* out, int b, int hw, int c,
                                 int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype) && dsc && s && dpooled && pre && out && b > 0 && hw > 0 && c > 0, "advs_se_scale_bwd: bad args");
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "se_scale_bwd: c=%d must be a multiple of %d", c, vec);
    const size_t total = (size_t)b * hw * (c / vec);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADVS_SWITCH_T(dtype, se_scale_bwd_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)dsc, s, dpooled, (const T*)pre, (T*)out, b, hw, c));
    ADVS_CHECK_LAUNCH("se_scale_bwd");
    return ADVS_OK;
}
