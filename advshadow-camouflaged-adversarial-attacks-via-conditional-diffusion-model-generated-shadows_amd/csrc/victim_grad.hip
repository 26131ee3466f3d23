// Input-gradient pieces of the victim classifier, for the gradient-based perturbation the reference interleaves with
// the shadow composite (tools/train_shadow.py:177-221 apply_adversarial_perturbation; ddim2/test.py:647-681).
// d loss / d image of the eval-mode ResNet-50 is the forward network run backwards: every conv's data gradient is
// again an advs_conv2d (transposed / flipped weights packed by the host, stride 2 through advs_zero_insert2x);
// what is left are the byte-moving pieces here -- HBM-bound, one 16-byte vector per lane.
#include "common.h"

// ---------------------------------------------------------------- softmax cross-entropy gradient, one wave per row
// out[b][k] = scale * (softmax(logits[b])[k] - [k == label[b]])       (F.cross_entropy, train_shadow.py:207)
__global__ void __launch_bounds__(64)
softmax_ce_grad_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, float* __restrict__ out,
                       int K, float scale) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* row = logits + (size_t)b * K;
    float m = -INFINITY;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, row[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += expf(row[k] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const int lab = (int)labels[b];
    for (int k = lane; k < K; k += 64) out[(size_t)b * K + k] = scale * (expf(row[k] - m) / s - (k == lab ? 1.f : 0.f));
}

extern "C" int advs_softmax_ce_grad(const float* logits, const long long* labels, float* out, int b, int k, float scale,
                                    void* stream) {
    ADVS_REQUIRE(logits && labels && out && b > 0 && k > 0, "softmax_ce_grad: bad args");
    softmax_ce_grad_kernel<<<b, 64, 0, (hipStream_t)stream>>>(logits, labels, out, k, scale);
    ADVS_CHECK_LAUNCH("softmax_ce_grad");
    return ADVS_OK;
}

// ---------------------------------------------------------------- ReLU backward (+ residual-branch sum)
// out = y > 0 ? g + add : 0 ; y is the forward activation AFTER the ReLU; out may alias g.
template <typename T>
__global__ void relu_bwd_kernel(const u32x4* __restrict__ g, const u32x4* __restrict__ add, const u32x4* __restrict__ y,
                                u32x4* __restrict__ out, size_t nvec) {
    constexpr int VEC = Elt<T>::VEC;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        float gv[VEC], yv[VEC], av[VEC];
        unpack16<T>(g[i], gv);
        unpack16<T>(y[i], yv);
        if (add) {
            unpack16<T>(add[i], av);
#pragma unroll
            for (int j = 0; j < VEC; ++j) gv[j] += av[j];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) gv[j] = yv[j] > 0.f ? gv[j] : 0.f;
        out[i] = pack16<T>(gv);
    }
}

extern "C" int advs_relu_bwd(const void* g, const void* add, const void* y, void* out, long long n, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_relu_bwd: unknown dtype code %d", dtype);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(g && y && out && n > 0 && n % vec == 0, "relu_bwd: bad args (n=%lld must be a multiple of %d)", n, vec);
    const size_t nvec = (size_t)n / vec;
    const int grid = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
    ADVS_SWITCH_T(dtype, relu_bwd_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const u32x4*)g, (const u32x4*)add, (const u32x4*)y, (u32x4*)out, nvec));
    ADVS_CHECK_LAUNCH("relu_bwd");
    return ADVS_OK;
}

// ---------------------------------------------------------------- gradient of a stride-2 sampling: zero insertion
// out[b][Y][X][:] = (Y, X both even and Y/2 < h, X/2 < w) ? in[b][Y/2][X/2][:] : 0.  A stride-2 conv's data gradient
// is the stride-1 conv of this with the flipped, transposed weights.
template <typename T>
__global__ void zero_insert2x_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int B, int h, int w, int Ho, int Wo, int vpp) {
    const size_t total = (size_t)B * Ho * Wo * vpp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;
        const int X = (int)(r % Wo); r /= Wo;
        const int Y = (int)(r % Ho);
        const int b = (int)(r / Ho);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (!((X | Y) & 1) && (Y >> 1) < h && (X >> 1) < w) v = in[(((size_t)b * h + (Y >> 1)) * w + (X >> 1)) * vpp + cv];
        out[i] = v;
    }
}

extern "C" int advs_zero_insert2x(const void* in, void* out, int b, int h, int w, int c, int ho, int wo, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_zero_insert2x: unknown dtype code %d", dtype);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(in && out && b > 0 && h > 0 && w > 0 && c > 0 && c % vec == 0, "zero_insert2x: bad args");
    ADVS_REQUIRE(ho >= 2 * h - 1 && ho <= 2 * h && wo >= 2 * w - 1 && wo <= 2 * w, "zero_insert2x: %dx%d is not the stride-2 pre-image of %dx%d", ho, wo, h, w);
    const size_t total = (size_t)b * ho * wo * (c / vec);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ADVS_SWITCH_T(dtype, zero_insert2x_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const u32x4*)in, (u32x4*)out, b, h, w, ho, wo, c / vec));
    ADVS_CHECK_LAUNCH("zero_insert2x");
    return ADVS_OK;
}

// ---------------------------------------------------------------- stride-2 3x3 data gradient without the zero insertion
// A 3x3 / stride 2 / pad 1 conv's data gradient at input pixel (2i+a, 2j+b) only ever sees output pixels (i, j), (i, j+1), (i+1, j),
// (i+1, j+1): parity a = 0 takes row i through tap ky = 1, parity a = 1 takes rows i and i+1 through ky = 2 and ky = 0 (likewise
// columns).  So the gradient of all four parities is ONE 1x1 GEMM over the gathered quadruple, [4 C'] -> [4 C] with the host-packed
// block matrix (9 of its 16 blocks are non-zero: 4 C C' MACs per output pixel quadruple instead of the 9 C C' per pixel = 36 per
// quadruple of the zero-insertion form), followed by a depth-to-space interleave that also applies the ReLU mask.
// out[b][i][j][q*c .. +c) = in[b][i + (q >> 1)][j + (q & 1)][:]  (0 beyond the last row / column)
template <typename T>
__global__ void gather2x2_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int B, int H, int W, int vpp) {
    const size_t total = (size_t)B * H * W * 4 * vpp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;
        const int q = (int)(r % 4); r /= 4;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const int yy = y + (q >> 1), xx = x + (q & 1);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (yy < H && xx < W) v = in[(((size_t)b * H + yy) * W + xx) * vpp + cv];
        out[i] = v;
    }
}
extern "C" int advs_gather2x2(const void* in, void* out, int b, int h, int w, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_gather2x2: unknown dtype code %d", dtype);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(in && out && b > 0 && h > 0 && w > 0 && c > 0 && c % vec == 0, "gather2x2: bad args");
    const size_t total = (size_t)b * h * w * 4 * (c / vec);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ADVS_SWITCH_T(dtype, gather2x2_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const u32x4*)in, (u32x4*)out, b, h, w, c / vec));
    ADVS_CHECK_LAUNCH("gather2x2");
    return ADVS_OK;
}
// out[b][2i + (q >> 1)][2j + (q & 1)][:] = y[same] > 0 ? x[b][i][j][q*c .. +c) : 0   (x [b][h/2][w/2][4c], y / out [b][h][w][c])
template <typename T>
__global__ void depth_to_space2_relu_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ y, u32x4* __restrict__ out, int B, int H,
                                            int W, int vpp) {
    constexpr int VEC = Elt<T>::VEC;
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * 4 * vpp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;
        const int q = (int)(r % 4); r /= 4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const size_t o = (((size_t)b * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * vpp + cv;
        float g[VEC], m[VEC];
        unpack16<T>(x[i], g);
        unpack16<T>(y[o], m);
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
        out[o] = pack16<T>(g);
    }
}
extern "C" int advs_depth_to_space2_relu(const void* x, const void* y, void* out, int b, int h, int w, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_depth_to_space2_relu: unknown dtype code %d", dtype);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(x && y && out && b > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0 && c > 0 && c % vec == 0, "depth_to_space2_relu: bad args");
    const size_t total = (size_t)b * h * w * (c / vec);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ADVS_SWITCH_T(dtype, depth_to_space2_relu_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const u32x4*)x, (const u32x4*)y, (u32x4*)out, b, h, w, c / vec));
    ADVS_CHECK_LAUNCH("depth_to_space2_relu");
    return ADVS_OK;
}

// ---------------------------------------------------------------- AdaptiveAvgPool2d(1) backward fused with the last ReLU
// out[b][p][c] = y[b][p][c] > 0 ? gp[b][c] / HW : 0
template <typename T>
__global__ void avgpool_bwd_relu_kernel(const float* __restrict__ gp, const u32x4* __restrict__ y, u32x4* __restrict__ out,
                                        int B, int HW, int C) {
    constexpr int VEC = Elt<T>::VEC;
    const int vpp = C / VEC;
    const size_t total = (size_t)B * HW * vpp;
    const float inv = 1.0f / (float)HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        const int b = (int)(i / ((size_t)HW * vpp));
        float yv[VEC], gv[VEC];
        unpack16<T>(y[i], yv);
        const float* g = gp + (size_t)b * C + cv * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) gv[j] = yv[j] > 0.f ? g[j] * inv : 0.f;
        out[i] = pack16<T>(gv);
    }
}

extern "C" int advs_avgpool_bwd_relu(const float* gp, const void* y, void* out, int b, int hw, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_avgpool_bwd_relu: unknown dtype code %d", dtype);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(gp && y && out && b > 0 && hw > 0 && c > 0 && c % vec == 0, "avgpool_bwd_relu: bad args");
    const size_t total = (size_t)b * hw * (c / vec);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ADVS_SWITCH_T(dtype, avgpool_bwd_relu_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(gp, (const u32x4*)y, (u32x4*)out, b, hw, c));
    ADVS_CHECK_LAUNCH("avgpool_bwd_relu");
    return ADVS_OK;
}

// ---------------------------------------------------------------- MaxPool2d(3,2,1) backward fused with the stem's ReLU
// Gather form (no atomics, deterministic): input pixel (iy, ix) collects g from every window that contains it and whose
// FIRST maximum in row-major scan order it is (torch's max_pool2d keeps the first maximum: "val > maxval").
// x is the pooled tensor's input, i.e. the stem output after ReLU; pixels with x <= 0 get no gradient (ReLU backward).
template <typename T>
__global__ void maxpool3s2_bwd_relu_kernel(const u32x4* __restrict__ g, const u32x4* __restrict__ x, u32x4* __restrict__ out,
                                           int B, int H, int W, int Ho, int Wo, int vpp) {
    constexpr int VEC = Elt<T>::VEC;
    const size_t total = (size_t)B * H * W * vpp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        float me[VEC], acc[VEC];
        unpack16<T>(x[i], me);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        // windows oy with 2*oy - 1 <= iy <= 2*oy + 1
        const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1, ox0 = ix >> 1, ox1 = (ix + 1) >> 1;
        for (int oy = oy0; oy <= oy1; ++oy) {
            if (oy >= Ho) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                if (ox >= Wo) continue;
                float best[VEC];
                bool mine[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; mine[j] = false; }
                for (int dy = -1; dy <= 1; ++dy) {
                    const int yy = 2 * oy + dy;
                    if ((unsigned)yy >= (unsigned)H) continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int xx = 2 * ox + dx;
                        if ((unsigned)xx >= (unsigned)W) continue;
                        float t[VEC];
                        unpack16<T>(x[(((size_t)b * H + yy) * W + xx) * vpp + cv], t);
                        const bool self = yy == iy && xx == ix;
#pragma unroll
                        for (int j = 0; j < VEC; ++j)
                            if (t[j] > best[j]) { best[j] = t[j]; mine[j] = self; }
                    }
                }
                float gv[VEC];
                unpack16<T>(g[(((size_t)b * Ho + oy) * Wo + ox) * vpp + cv], gv);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += mine[j] ? gv[j] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = me[j] > 0.f ? acc[j] : 0.f;
        out[i] = pack16<T>(acc);
    }
}

extern "C" int advs_maxpool3x3s2_bwd_relu(const void* g, const void* x, void* out, int b, int h, int w, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_maxpool3x3s2_bwd_relu: unknown dtype code %d", dtype);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(g && x && out && b > 0 && h > 0 && w > 0 && c > 0 && c % vec == 0, "maxpool3x3s2_bwd_relu: bad args");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const size_t total = (size_t)b * h * w * (c / vec);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ADVS_SWITCH_T(dtype, maxpool3s2_bwd_relu_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const u32x4*)g, (const u32x4*)x, (u32x4*)out, b, h, w, ho, wo, c / vec));
    ADVS_CHECK_LAUNCH("maxpool3x3s2_bwd_relu");
    return ADVS_OK;
}

// ---------------------------------------------------------------- MaxPool2d(2) backward fused with the ReLU of its input
// VGG's conv-ReLU-pool (ASR_fast.py:33-46 victims): x [b][h][w][c] is the pool's input (a conv output after ReLU), g the
// gradient at [b][h/2][w/2][c].  Windows do not overlap: a pixel gets its window's g iff it is the window's first maximum
// in row-major order (torch keeps the first) and x > 0; rows / columns beyond 2*(h/2), 2*(w/2) get none.
template <typename T>
__global__ void maxpool2_bwd_relu_kernel(const u32x4* __restrict__ g, const u32x4* __restrict__ x, u32x4* __restrict__ out,
                                         int B, int H, int W, int vpp) {
    constexpr int VEC = Elt<T>::VEC;
    const int Ho = H >> 1, Wo = W >> 1;
    const size_t total = (size_t)B * H * W * vpp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        const int oy = iy >> 1, ox = ix >> 1;
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        if (oy < Ho && ox < Wo) {
            float best[VEC], me[VEC], gv[VEC];
            bool mine[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; mine[j] = false; }
            for (int q = 0; q < 4; ++q) {
                const int yy = 2 * oy + (q >> 1), xx = 2 * ox + (q & 1);
                float t[VEC];
                unpack16<T>(x[(((size_t)b * H + yy) * W + xx) * vpp + cv], t);
                const bool self = yy == iy && xx == ix;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    if (self) me[j] = t[j];
                    if (t[j] > best[j]) { best[j] = t[j]; mine[j] = self; }
                }
            }
            unpack16<T>(g[(((size_t)b * Ho + oy) * Wo + ox) * vpp + cv], gv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = (mine[j] && me[j] > 0.f) ? gv[j] : 0.f;
        }
        out[i] = pack16<T>(acc);
    }
}

extern "C" int advs_maxpool2_bwd_relu(const void* g, const void* x, void* out, int b, int h, int w, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_maxpool2_bwd_relu: unknown dtype code %d", dtype);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(g && x && out && b > 0 && h > 1 && w > 1 && c > 0 && c % vec == 0, "maxpool2_bwd_relu: bad args");
    const size_t total = (size_t)b * h * w * (c / vec);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADVS_SWITCH_T(dtype, maxpool2_bwd_relu_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const u32x4*)g, (const u32x4*)x, (u32x4*)out, b, h, w, c / vec));
    ADVS_CHECK_LAUNCH("maxpool2_bwd_relu");
    return ADVS_OK;
}

// ---------------------------------------------------------------- stem conv data gradient: NHWC T -> NCHW f32 image
// dx[b][c][iy][ix] = sum_{r,s,o} g[b][(iy + pad - r) / stride][(ix + pad - s) / stride][o] * w[o][c][r][s]
// over the (r, s) for which both quotients are exact and in range.  thread = one image pixel, all cin (<= 4) channels;
// weights in LDS as [r][s][o][4].
template <typename T>
__global__ void __launch_bounds__(256)
conv_stem_bwd_kernel(const T* __restrict__ g, const float* __restrict__ w, float* __restrict__ dx,
                     int B, int Cin, int H, int W, int Cout, int K, int stride, int pad, int Ho, int Wo) {
    extern __shared__ float sw[];                 // [K*K][Cout][4]
    constexpr int VEC = Elt<T>::VEC;
    for (int i = threadIdx.x; i < K * K * Cout * 4; i += 256) {
        const int c = i & 3, o = (i >> 2) % Cout, rs = (i >> 2) / Cout;
        sw[i] = c < Cin ? w[((size_t)o * Cin + c) * K * K + rs] : 0.f;
    }
    __syncthreads();
    const long long npix = (long long)B * H * W;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long long)gridDim.x * 256) {
        const int b = (int)(pix / ((long long)H * W)), rem = (int)(pix - (long long)b * H * W);
        // which taps exist depends on the pixel's parity (for stride 2): walk a row's even columns first, then its odd
        // ones, so that the lanes of a wave mostly share the tap set instead of masking each other out
        const int iy = rem / W, j = rem - iy * W, he = (W + 1) >> 1;
        const int ix = j < he ? 2 * j : 2 * (j - he) + 1;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < K; ++r) {
            const int ty = iy + pad - r;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int s = 0; s < K; ++s) {
                const int tx = ix + pad - s;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                const u32x4* gp = (const u32x4*)(g + (((size_t)b * Ho + oy) * Wo + ox) * Cout);
                const f32x4* wp = (const f32x4*)(sw + (size_t)(r * K + s) * Cout * 4);
                for (int o8 = 0; o8 < Cout / VEC; ++o8) {
                    float gv[VEC];
                    unpack16<T>(gp[o8], gv);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const f32x4 wv = wp[o8 * VEC + j];
                        acc[0] = fmaf(gv[j], wv[0], acc[0]);
                        acc[1] = fmaf(gv[j], wv[1], acc[1]);
                        acc[2] = fmaf(gv[j], wv[2], acc[2]);
                        acc[3] = fmaf(gv[j], wv[3], acc[3]);
                    }
                }
            }
        }
        for (int c = 0; c < Cin; ++c) dx[(((size_t)b * Cin + c) * H + iy) * W + ix] = acc[c];
    }
}

extern "C" int advs_conv_stem_bwd(const void* g, const float* w_oihw, float* dx_nchw, int b, int cin, int h, int w, int cout,
                                  int ksize, int stride, int pad, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_conv_stem_bwd: unknown dtype code %d", dtype);
    ADVS_REQUIRE(g && w_oihw && dx_nchw && b > 0 && h > 0 && w > 0, "conv_stem_bwd: bad args");
    ADVS_REQUIRE(cin >= 1 && cin <= 4 && cout % 8 == 0 && ksize >= 1 && ksize <= 7 && stride >= 1 && pad >= 0,
                 "conv_stem_bwd: unsupported shape cin=%d cout=%d k=%d", cin, cout, ksize);
    const size_t lds = (size_t)ksize * ksize * cout * 4 * sizeof(float);
    ADVS_REQUIRE(lds <= 65536, "conv_stem_bwd: weights do not fit LDS");
    const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
    const long long npix = (long long)b * h * w;
    const int grid = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
    ADVS_SWITCH_T(dtype, conv_stem_bwd_kernel<T><<<grid, 256, lds, (hipStream_t)stream>>>((const T*)g, w_oihw, dx_nchw, b, cin, h, w, cout, ksize, stride, pad, ho, wo));
    ADVS_CHECK_LAUNCH("conv_stem_bwd");
    return ADVS_OK;
}

// ---------------------------------------------------------------- col2im: the adjoint of advs_im2col_nchw
// dx[b][c][iy][ix] = sum over the (oy, r), (ox, s) with oy*stride + r - pad = iy, ox*stride + s - pad = ix of
// gcol[b][oy][ox][(c*K + r)*K + s];  gcol = advs_conv2d 1x1 of the stem-output gradient with the [Kp][cout] transposed
// stem weight.  Gather form, deterministic.  A workgroup owns a 16x16 tile of image pixels: the column rows that reach
// it (<= ((16 + K - 2) / stride + 1)^2 of them) are copied into LDS with coalesced reads, then every lane sums its taps
// from LDS.  Wave w takes the pixels of parity class (w >> 1, w & 1), so for stride 2 a wave shares one tap set.
constexpr int C2I_T = 16;
template <typename T>
__global__ void __launch_bounds__(256)
col2im_nchw_kernel(const T* __restrict__ gcol, float* __restrict__ dx, int Cin, int H, int W, int K,
                   int stride, int pad, int Ho, int Wo, int Kp, int tiles_x, int tiles_y) {
    extern __shared__ __align__(16) unsigned char c2i_smem[];
    T* cols = (T*)c2i_smem;
    const int KK = Cin * K * K;
    int blk = blockIdx.x;
    const int tx = blk % tiles_x; blk /= tiles_x;
    const int ty = blk % tiles_y;
    const int b = blk / tiles_y;
    const int iy0 = ty * C2I_T, ix0 = tx * C2I_T;
    // output rows / columns whose window touches the tile
    const int lo_y = iy0 + pad - (K - 1), lo_x = ix0 + pad - (K - 1);
    const int oy_lo = lo_y > 0 ? (lo_y + stride - 1) / stride : 0, ox_lo = lo_x > 0 ? (lo_x + stride - 1) / stride : 0;
    int oy_hi = (iy0 + C2I_T - 1 + pad) / stride, ox_hi = (ix0 + C2I_T - 1 + pad) / stride;
    oy_hi = oy_hi < Ho - 1 ? oy_hi : Ho - 1; ox_hi = ox_hi < Wo - 1 ? ox_hi : Wo - 1;
    const int nOy = oy_hi - oy_lo + 1, nOx = ox_hi - ox_lo + 1;
    if (nOy > 0 && nOx > 0) {
        const int total = nOy * nOx * KK;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int e = i % KK, row = i / KK;
            const int oy = oy_lo + row / nOx, ox = ox_lo + row % nOx;
            cols[i] = gcol[(((size_t)b * Ho + oy) * Wo + ox) * Kp + e];
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int iy = iy0 + 2 * (lane >> 3) + (wave >> 1), ix = ix0 + 2 * (lane & 7) + (wave & 1);
    if (iy >= H || ix >= W) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < K; ++r) {
        const int t = iy + pad - r;
        if (t < 0 || t % stride) continue;
        const int oy = t / stride;
        if (oy < oy_lo || oy > oy_hi) continue;
        for (int s = 0; s < K; ++s) {
            const int u = ix + pad - s;
            if (u < 0 || u % stride) continue;
            const int ox = u / stride;
            if (ox < ox_lo || ox > ox_hi) continue;
            const T* gp = cols + ((oy - oy_lo) * nOx + (ox - ox_lo)) * KK + r * K + s;
            for (int c = 0; c < Cin; ++c) acc[c] += Elt<T>::ld(gp + c * K * K);
        }
    }
    for (int c = 0; c < Cin; ++c) dx[(((size_t)b * Cin + c) * H + iy) * W + ix] = acc[c];
}

extern "C" int advs_col2im_nchw(const void* gcol, float* dx_nchw, int b, int cin, int h, int w, int ksize, int stride, int pad,
                                int kp, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_col2im_nchw: unknown dtype code %d", dtype);
    ADVS_REQUIRE(gcol && dx_nchw && b > 0 && cin >= 1 && cin <= 4 && h > 0 && w > 0 && ksize >= 1 && stride >= 1 && pad >= 0, "col2im_nchw: bad args");
    ADVS_REQUIRE(kp >= cin * ksize * ksize, "col2im_nchw: kp=%d must be >= %d", kp, cin * ksize * ksize);
    const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
    ADVS_REQUIRE(ho > 0 && wo > 0, "col2im_nchw: empty output");
    const int nmax = (C2I_T + ksize - 2) / stride + 1;
    const size_t lds = (size_t)nmax * nmax * cin * ksize * ksize * (dtype == ADVS_F32 ? 4 : 2);
    ADVS_REQUIRE(lds <= 160 * 1024, "col2im_nchw: a tile's %d x %d column rows (%zu bytes) do not fit LDS", nmax, nmax, lds);
    const int tiles_x = cdiv(w, C2I_T), tiles_y = cdiv(h, C2I_T);
    const long long blocks = (long long)b * tiles_x * tiles_y;
    ADVS_REQUIRE(blocks < (1ll << 31), "col2im_nchw: too many tiles");
    ADVS_SWITCH_T(dtype, {
        static bool attr_set = false;
        if (!attr_set) {
            ADVS_HIP(hipFuncSetAttribute((const void*)col2im_nchw_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        col2im_nchw_kernel<T><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>((const T*)gcol, dx_nchw, cin, h, w, ksize, stride, pad, ho, wo, kp, tiles_x, tiles_y);
    });
    ADVS_CHECK_LAUNCH("col2im_nchw");
    return ADVS_OK;
}

// ---------------------------------------------------------------- one iterative-gradient update (train_shadow.py:212-216)
// pert = clamp(pert - alpha * sign(grad * mask), -eps, eps);  x_in = x0 + pert     (all NCHW f32; mask [B][mc][H][W], mc = 1 or C)
// grad may hold nsum stacked gradients per image ([B][nsum][C][H][W], summed here: the integrated-gradient variant).
__global__ void iga_step_kernel(const float* __restrict__ x0, const float* __restrict__ grad, const float* __restrict__ mask,
                                float* __restrict__ pert, float* __restrict__ xin, int C, int HW, int mc, int nsum,
                                float alpha, float eps, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const size_t bc = i / HW;
        const int c = (int)(bc % C);
        const size_t b = bc / C;
        float gsum = 0.f;
        for (int k = 0; k < nsum; ++k) gsum += grad[((b * nsum + k) * C + c) * (size_t)HW + p];
        const float gm = gsum * mask[(b * mc + (mc == 1 ? 0 : c)) * (size_t)HW + p];
        const float sgn = gm > 0.f ? 1.f : (gm < 0.f ? -1.f : 0.f);
        const float pv = fminf(fmaxf(pert[i] - alpha * sgn, -eps), eps);
        pert[i] = pv;
        if (xin) xin[i] = x0[i] + pv;
    }
}

extern "C" int advs_iga_step(const float* x0, const float* grad, const float* mask, float* pert, float* xin,
                             int b, int c, int hw, int mask_channels, int nsum, float alpha, float eps, void* stream) {
    ADVS_REQUIRE(x0 && grad && mask && pert && b > 0 && c > 0 && hw > 0 && nsum >= 1, "iga_step: bad args");
    ADVS_REQUIRE(mask_channels == 1 || mask_channels == c, "iga_step: mask must have 1 or %d channels", c);
    const size_t total = (size_t)b * c * hw;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    iga_step_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x0, grad, mask, pert, xin, c, hw, mask_channels, nsum, alpha, eps, total);
    ADVS_CHECK_LAUNCH("iga_step");
    return ADVS_OK;
}

// out = clamp(x0 + pert, 0, 1)   (train_shadow.py:219-220)
__global__ void perturb_clamp01_kernel(const float* __restrict__ x0, const float* __restrict__ pert, float* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = fminf(fmaxf(x0[i] + pert[i], 0.f), 1.f);
}

extern "C" int advs_perturb_clamp01(const float* x0, const float* pert, float* out, long long n, void* stream) {
    ADVS_REQUIRE(x0 && pert && out && n > 0, "perturb_clamp01: bad args");
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    perturb_clamp01_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x0, pert, out, (size_t)n);
    ADVS_CHECK_LAUNCH("perturb_clamp01");
    return ADVS_OK;
}

// out[k] = base + (k / steps) * (x - base), k = 0..steps: the interpolation path of the integrated-gradient variant
// (ddim2/test.py:659-660), stacked as one batch of steps + 1 images.  n = elements of one image.
__global__ void lerp_stack_kernel(const float* __restrict__ base, const float* __restrict__ x, float* __restrict__ out,
                                  int steps, size_t n) {
    const size_t total = n * (size_t)(steps + 1);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i % n;
        const int k = (int)(i / n);
        out[i] = base[e] + ((float)k / (float)steps) * (x[e] - base[e]);
    }
}

extern "C" int advs_lerp_stack(const float* base, const float* x, float* out, int steps, long long n, void* stream) {
    ADVS_REQUIRE(base && x && out && steps >= 1 && n > 0, "lerp_stack: bad args");
    const long long total = n * (steps + 1);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    lerp_stack_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(base, x, out, steps, (size_t)n);
    ADVS_CHECK_LAUNCH("lerp_stack");
    return ADVS_OK;
}
