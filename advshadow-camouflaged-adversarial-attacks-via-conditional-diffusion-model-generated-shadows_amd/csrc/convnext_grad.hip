// ConvNeXt victim run backwards (the gradient attack of tools/train_shadow.py:177-221 with the timm convnext_base of
// ASR_fast.py:21-26): the data-gradient pieces beyond the Linear layers (advs_conv2d on transposed weights), LayerNorm and GELU
// (vit_grad.hip).  All HBM-bound single passes over NHWC activations, f32 arithmetic.
#include "common.h"

// Data gradient of the depthwise k x k conv (stride 1, 'same' padding) plus the residual stream's gradient:
//   dx[b][y][x][c] = add[b][y][x][c] + sum_{ky,kx} dy[b][y - (ky - p)][x - (kx - p)][c] * w[ky*k + kx][c]
// -- the forward kernel's gather with the taps mirrored.  w is the FORWARD weight layout [k*k][C] f32 (advs_dwconv2d).
template <typename T>
__global__ void __launch_bounds__(256)
dwconv_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ w, const T* __restrict__ add, T* __restrict__ dx,
                  int B, int H, int W, int C, int K) {
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
        if (add) unpack16<T>(*((const u32x4*)(add + (((size_t)b * H + y) * W + x) * C) + cv), acc);
        else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        }
        for (int ky = 0; ky < K; ++ky) {
            const int sy = y - (ky - pad);
            if ((unsigned)sy >= (unsigned)H) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int sx = x - (kx - pad);
                if ((unsigned)sx >= (unsigned)W) continue;
                float f[VEC];
                unpack16<T>(*((const u32x4*)(dy + (((size_t)b * H + sy) * W + sx) * C) + cv), f);
                const float* wt = w + (size_t)(ky * K + kx) * C + cv * VEC;
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(f[e], wt[e], acc[e]);
            }
        }
        *((u32x4*)(dx + (((size_t)b * H + y) * W + x) * C) + cv) = pack16<T>(acc);
    }
}
extern "C" int advs_dwconv2d_bwd(const void* dy, const float* w_taps_c, const void* add, void* dx, int b, int h, int w, int c, int ksize,
                                 int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_dwconv2d_bwd: unknown dtype code %d", dtype);
    ADVS_REQUIRE(dy && w_taps_c && dx && b > 0 && h > 0 && w > 0 && c > 0, "dwconv2d_bwd: bad args");
    ADVS_REQUIRE((ksize & 1) && ksize >= 1 && ksize <= 7, "dwconv2d_bwd: ksize %d unsupported", ksize);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "dwconv2d_bwd: c=%d must be a multiple of %d", c, vec);
    const size_t total = (size_t)b * h * w * (c / vec);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADVS_SWITCH_T(dtype, dwconv_bwd_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)dy, w_taps_c, (const T*)add, (T*)dx, b, h, w, c, ksize));
    ADVS_CHECK_LAUNCH("dwconv2d_bwd");
    return ADVS_OK;
}

// Inverse of advs_space_to_depth2 (the gradient of the 2x2 / stride 2 patch gather in front of a downsampling GEMM):
// y[b][2oy + (q>>1)][2ox + (q&1)][c] = x[b][oy][ox][q*C + c],  x [b][h/2][w/2][4C], y [b][h][w][C].
template <typename T>
__global__ void depth_to_space2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    constexpr int VEC = Elt<T>::VEC;
    const int vpp = C / VEC, Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * 4 * vpp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;
        const int q = (int)(r % 4); r /= 4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        *((u32x4*)(y + (((size_t)b * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C) + cv) = ((const u32x4*)x)[i];
    }
}
extern "C" int advs_depth_to_space2(const void* x, void* y, int b, int h, int w, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_depth_to_space2: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && y && b > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "depth_to_space2: bad shape");
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "depth_to_space2: c=%d must be a multiple of %d", c, vec);
    const size_t total = (size_t)b * h * w * (c / vec);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADVS_SWITCH_T(dtype, depth_to_space2_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, (T*)y, b, h, w, c));
    ADVS_CHECK_LAUNCH("depth_to_space2");
    return ADVS_OK;
}

// Gradient of advs_global_avgpool: out[b][p][c] = g[b][c] / hw  (g f32 [b][c], out in the compute dtype).
template <typename T>
__global__ void avgpool_bwd_kernel(const float* __restrict__ g, T* __restrict__ out, int B, int HW, int C) {
    const size_t total = (size_t)B * HW * C;
    const float inv = 1.0f / (float)HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((size_t)HW * C));
        Elt<T>::st(out + i, g[(size_t)b * C + c] * inv);
    }
}
extern "C" int advs_avgpool_bwd(const float* g, void* out, int b, int hw, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype) && g && out && b > 0 && hw > 0 && c > 0, "advs_avgpool_bwd: bad args");
    const size_t total = (size_t)b * hw * c;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADVS_SWITCH_T(dtype, avgpool_bwd_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(g, (T*)out, b, hw, c));
    ADVS_CHECK_LAUNCH("avgpool_bwd");
    return ADVS_OK;
}
