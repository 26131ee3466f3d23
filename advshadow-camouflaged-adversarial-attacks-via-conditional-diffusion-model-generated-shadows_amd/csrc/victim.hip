// Victim-classifier pieces that are not GEMM-shaped (ResNet-50 of ASR_fast.py:16-20 /
// ddim2/diff_model2.py:19-44): the 7x7 stride-2 stem on the 3-channel image, MaxPool2d(3,2,1) and
// the global average pool.  Everything else in the network is advs_conv2d (BatchNorm folded into
// the packed weights by the host, ReLU / residual in the epilogue) and advs_linear_f32.
#include "common.h"

// ---------------------------------------------------------------- stem: NCHW f32 -> NHWC T
// y = act(conv(x, w[cout][cin][k][k], stride, pad) + bias).  thread = one output pixel x cout/4
// channels; weights broadcast from LDS.
template <typename T>
__global__ void __launch_bounds__(256)
conv_stem_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                 T* __restrict__ y, int B, int Cin, int H, int W, int Cout, int K, int stride, int pad,
                 int Ho, int Wo, int act) {
    extern __shared__ float sw[];                 // [cin*k*k][cout] then bias
    const int KK = Cin * K * K;
    for (int i = threadIdx.x; i < KK * Cout; i += 256) {
        const int k = i / Cout, o = i - k * Cout;
        sw[i] = w[(size_t)o * KK + k];
    }
    for (int i = threadIdx.x; i < Cout; i += 256) sw[KK * Cout + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int px = threadIdx.x & 63, cq = threadIdx.x >> 6, cper = Cout / 4;
    const long long npix = (long long)B * Ho * Wo;
    for (long long pb = (long long)blockIdx.x * 64; pb < npix; pb += (long long)gridDim.x * 64) {
        const long long pix = pb + px;
        if (pix >= npix) continue;
        const int b = (int)(pix / (Ho * Wo)), rem = (int)(pix - (long long)b * Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        T* yo = y + (size_t)pix * Cout + cq * cper;
        for (int c8 = 0; c8 < cper; c8 += 8) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = sw[KK * Cout + cq * cper + c8 + j];
            for (int c = 0; c < Cin; ++c)
                for (int r = 0; r < K; ++r) {
                    const int iy = oy * stride + r - pad;
                    if ((unsigned)iy >= (unsigned)H) continue;
                    const float* xr = x + (((size_t)b * Cin + c) * H + iy) * W;
                    for (int s = 0; s < K; ++s) {
                        const int ix = ox * stride + s - pad;
                        if ((unsigned)ix >= (unsigned)W) continue;
                        const float v = xr[ix];
                        const float* wr = sw + ((c * K + r) * K + s) * Cout + cq * cper + c8;
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
                    }
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = apply_act(acc[j], act);
            *(u32x4*)(yo + c8) = pack16<T>(acc);
            if (sizeof(T) == 4) *(u32x4*)(yo + c8 + 4) = pack16<T>(acc + 4);
        }
    }
}

extern "C" int advs_conv_stem(const float* x_nchw, const float* w_oihw, const float* bias, void* y,
                              int b, int cin, int h, int w, int cout, int ksize, int stride, int pad, int act,
                              int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_conv_stem: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x_nchw && w_oihw && y && b > 0 && h > 0 && w > 0, "conv_stem: bad args");
    ADVS_REQUIRE(cin >= 1 && cin <= 4 && cout % 32 == 0 && ksize >= 1 && ksize <= 7 && stride >= 1 && pad >= 0,
                 "conv_stem: unsupported shape cin=%d cout=%d k=%d", cin, cout, ksize);
    const size_t lds = ((size_t)cin * ksize * ksize * cout + cout) * sizeof(float);
    ADVS_REQUIRE(lds <= 65536, "conv_stem: weights do not fit LDS");
    const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
    const long long npix = (long long)b * ho * wo;
    const int grid = (int)((npix + 63) / 64 < 8192 ? (npix + 63) / 64 : 8192);
    ADVS_SWITCH_T(dtype, conv_stem_kernel<T><<<grid, 256, lds, (hipStream_t)stream>>>(x_nchw, w_oihw, bias, (T*)y, b, cin, h, w, cout, ksize, stride, pad, ho, wo, act));
    ADVS_CHECK_LAUNCH("conv_stem");
    return ADVS_OK;
}

// ---------------------------------------------------------------- stem as a GEMM: im2col of the NCHW f32 image
// y[b][oy][ox][k] = x[b][c][oy*stride + r - pad][ox*stride + s - pad] (0 outside), k = (c*K + r)*K + s < Cin*K*K, zero for
// the padding columns k in [Cin*K*K, Kp).  The stem conv is then advs_conv2d 1x1 over Kp channels with the OIHW weight
// viewed as [cout][Cin*K*K] (zero-padded to Kp): the 7x7 stride-2 conv of ResNet-50 (K = 147) moves from the VALU kernel
// above onto MFMA.  One 16-byte vector of consecutive k per lane.
template <typename T>
__global__ void im2col_nchw_kernel(const float* __restrict__ x, u32x4* __restrict__ y, int B, int Cin, int H, int W, int K,
                                   int stride, int pad, int Ho, int Wo, int Kp) {
    constexpr int VEC = Elt<T>::VEC;
    const int vpp = Kp / VEC, KK = Cin * K * K;
    const size_t total = (size_t)B * Ho * Wo * vpp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kv = (int)(i % vpp);
        size_t rr = i / vpp;
        const int ox = (int)(rr % Wo); rr /= Wo;
        const int oy = (int)(rr % Ho);
        const int b = (int)(rr / Ho);
        float v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int k = kv * VEC + j;
            const int c = k / (K * K), rs = k - c * K * K, r = rs / K, s = rs - r * K;
            const int iy = oy * stride + r - pad, ix = ox * stride + s - pad;
            v[j] = (k < KK && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                       ? x[(((size_t)b * Cin + c) * H + iy) * W + ix] : 0.f;
        }
        y[i] = pack16<T>(v);
    }
}

extern "C" int advs_im2col_nchw(const float* x_nchw, void* y, int b, int cin, int h, int w, int ksize, int stride, int pad,
                                int kp, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_im2col_nchw: unknown dtype code %d", dtype);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(x_nchw && y && b > 0 && cin > 0 && h > 0 && w > 0 && ksize >= 1 && stride >= 1 && pad >= 0, "im2col_nchw: bad args");
    ADVS_REQUIRE(kp >= cin * ksize * ksize && kp % vec == 0, "im2col_nchw: kp=%d must be >= %d and a multiple of %d", kp, cin * ksize * ksize, vec);
    const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
    ADVS_REQUIRE(ho > 0 && wo > 0, "im2col_nchw: empty output");
    const size_t total = (size_t)b * ho * wo * (kp / vec);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADVS_SWITCH_T(dtype, im2col_nchw_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(x_nchw, (u32x4*)y, b, cin, h, w, ksize, stride, pad, ho, wo, kp));
    ADVS_CHECK_LAUNCH("im2col_nchw");
    return ADVS_OK;
}

// ---------------------------------------------------------------- MaxPool2d(3, stride 2, pad 1), NHWC
template <typename T>
__global__ void maxpool3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo) {
    constexpr int VEC = Elt<T>::VEC;
    const int vpp = C / VEC;
    const size_t total = (size_t)B * Ho * Wo * vpp;
    const u32x4* xv = (const u32x4*)x;
    u32x4* yv = (u32x4*)y;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float m[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) m[j] = -INFINITY;
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = 2 * oy + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = 2 * ox + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                float t[VEC];
                unpack16<T>(xv[(((size_t)b * H + iy) * W + ix) * vpp + cv], t);
#pragma unroll
                for (int j = 0; j < VEC; ++j) m[j] = fmaxf(m[j], t[j]);
            }
        }
        yv[i] = pack16<T>(m);
    }
}

extern "C" int advs_maxpool3x3s2(const void* x, void* y, int b, int h, int w, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_maxpool3x3s2: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && y && b > 0 && h > 0 && w > 0, "maxpool3x3s2: bad args");
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "maxpool3x3s2: c=%d must be a multiple of %d", c, vec);
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const size_t total = (size_t)b * ho * wo * (c / vec);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    ADVS_SWITCH_T(dtype, maxpool3s2_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, (T*)y, b, h, w, c, ho, wo));
    ADVS_CHECK_LAUNCH("maxpool3x3s2");
    return ADVS_OK;
}

// ---------------------------------------------------------------- AdaptiveAvgPool2d(1): NHWC T -> f32 [B][C]
template <typename T>
__global__ void __launch_bounds__(256)
global_avgpool_kernel(const T* __restrict__ x, float* __restrict__ y, int HW, int C) {
    __shared__ float part[256];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int p = q; p < HW; p += 4) s += Elt<T>::ld(x + ((size_t)b * HW + p) * C + c);
    part[threadIdx.x] = s;
    __syncthreads();
    if (q == 0 && c < C)
        y[(size_t)b * C + c] = (part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192]) / (float)HW;
}

extern "C" int advs_global_avgpool(const void* x, float* y, int b, int hw, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_global_avgpool: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && y && b > 0 && hw > 0 && c > 0, "global_avgpool: bad args");
    dim3 grid(cdiv(c, 64), b);
    ADVS_SWITCH_T(dtype, global_avgpool_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, y, hw, c));
    ADVS_CHECK_LAUNCH("global_avgpool");
    return ADVS_OK;
}

// ================================================================ ViT token plumbing (HF ViTForImageClassification)
// patchify: NCHW f32 image -> [B][gh][gw][cin*ps*ps] T, channel = c*ps*ps + ky*ps + kx, i.e. the K order of the
// patch-embedding Conv2d(cin, hidden, ps, stride ps) weight viewed as [hidden][cin*ps*ps]; the projection itself
// is then an advs_conv2d 1x1.
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int Cin, int H, int W, int ps, int Kp) {
    const int gh = H / ps, gw = W / ps, K = Cin * ps * ps;         // rows are Kp >= K long, zero beyond K
    const size_t total = (size_t)B * gh * gw * Kp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        size_t r = i / Kp;
        if (k >= K) { Elt<T>::st(y + i, 0.f); continue; }
        const int px = (int)(r % gw); r /= gw;
        const int py = (int)(r % gh);
        const int b = (int)(r / gh);
        const int c = k / (ps * ps), ky = (k / ps) % ps, kx = k % ps;
        Elt<T>::st(y + i, x[(((size_t)b * Cin + c) * H + py * ps + ky) * W + px * ps + kx]);
    }
}
extern "C" int advs_patchify_padded(const float* x_nchw, void* y, int b, int cin, int h, int w, int patch, int kpad, int dtype,
                                    void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_patchify: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x_nchw && y && b > 0 && cin > 0 && patch > 0 && h % patch == 0 && w % patch == 0, "patchify: bad args");
    ADVS_REQUIRE(kpad >= cin * patch * patch, "patchify: row length %d shorter than the %d patch elements", kpad, cin * patch * patch);
    const size_t total = (size_t)b * (h / patch) * (w / patch) * kpad;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ADVS_SWITCH_T(dtype, patchify_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(x_nchw, (T*)y, b, cin, h, w, patch, kpad));
    ADVS_CHECK_LAUNCH("patchify");
    return ADVS_OK;
}
extern "C" int advs_patchify(const float* x_nchw, void* y, int b, int cin, int h, int w, int patch, int dtype, void* stream) {
    return advs_patchify_padded(x_nchw, y, b, cin, h, w, patch, cin * patch * patch, dtype, stream);
}

// tokens[b][0] = cls + pos[0]; tokens[b][1+i] = patches[b][i] + pos[1+i]; rows >= 1+np are zero padding
// (ViTEmbeddings.forward).  tokens is [B][n_pad][C] T, patches [B][np][C] T, cls [C] / pos [1+np][C] f32.
template <typename T>
__global__ void vit_assemble_kernel(const T* __restrict__ patches, const float* __restrict__ cls, const float* __restrict__ pos,
                                    T* __restrict__ tokens, int B, int np, int n_pad, int C) {
    const size_t total = (size_t)B * n_pad * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int t = (int)(r % n_pad);
        const int b = (int)(r / n_pad);
        float v = 0.f;
        if (t == 0) v = cls[c] + pos[c];
        else if (t <= np) v = Elt<T>::ld(patches + ((size_t)b * np + (t - 1)) * C + c) + pos[(size_t)t * C + c];
        Elt<T>::st(tokens + i, v);
    }
}
extern "C" int advs_vit_assemble(const void* patches, const float* cls, const float* pos, void* tokens, int b, int np,
                                 int n_pad, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_vit_assemble: unknown dtype code %d", dtype);
    ADVS_REQUIRE(patches && cls && pos && tokens && b > 0 && np > 0 && n_pad > np && c > 0, "vit_assemble: bad args");
    const size_t total = (size_t)b * n_pad * c;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ADVS_SWITCH_T(dtype, vit_assemble_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)patches, cls, pos, (T*)tokens, b, np, n_pad, c));
    ADVS_CHECK_LAUNCH("vit_assemble");
    return ADVS_OK;
}

// rows [b*row_stride] of a [.][C] T matrix -> f32 [B][C] (the CLS token fed to the classifier head)
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ x, float* __restrict__ y, int B, long long row_stride, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    y[i] = Elt<T>::ld(x + (size_t)b * row_stride * C + c);
}
extern "C" int advs_gather_rows_f32(const void* x, float* y, int b, long long row_stride, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_gather_rows_f32: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && y && b > 0 && c > 0 && row_stride > 0, "gather_rows_f32: bad args");
    ADVS_SWITCH_T(dtype, gather_rows_kernel<T><<<cdiv((long long)b * c, 256), 256, 0, (hipStream_t)stream>>>((const T*)x, y, b, row_stride, c));
    ADVS_CHECK_LAUNCH("gather_rows_f32");
    return ADVS_OK;
}

// ================================================================ ConvNeXt pieces (timm convnext_base, ASR_fast.py:21-26)
// Depthwise k x k convolution (groups = C), stride 1 or 2, 'same' padding k/2, on NHWC activations:
// y[b][oy][ox][c] = bias[c] + sum_t x[b][oy*s + ky - k/2][ox*s + kx - k/2][c] * w[t][c].  A lane owns one 16-byte
// channel vector of one output pixel; the weights are pre-transposed by the host to [k*k][C] f32 so that a tap's
// weights for the lane's channels are contiguous (cached: every pixel of the image re-reads them).  HBM-bound
// (reads x once through L2, writes y once).
template <typename T>
__global__ void __launch_bounds__(256)
dwconv_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y,
              int B, int H, int W, int C, int K, int stride, int Ho, int Wo, int act) {
    constexpr int VEC = Elt<T>::VEC;
    const int vpp = C / VEC, pad = K / 2;
    const size_t total = (size_t)B * Ho * Wo * vpp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = bias ? bias[cv * VEC + e] : 0.f;
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy * stride + ky - pad;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * stride + kx - pad;
                if ((unsigned)ix >= (unsigned)W) continue;
                float f[VEC];
                unpack16<T>(*((const u32x4*)(x + (((size_t)b * H + iy) * W + ix) * C) + cv), f);
                const float* wt = w + (size_t)(ky * K + kx) * C + cv * VEC;
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(f[e], wt[e], acc[e]);
            }
        }
        if (act != ADVS_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = apply_act(acc[e], act);
        }
        *((u32x4*)(y + (((size_t)b * Ho + oy) * Wo + ox) * C) + cv) = pack16<T>(acc);
    }
}

extern "C" int advs_dwconv2d_act(const void* x, const float* w_taps_c, const float* bias, void* y, int b, int h, int w, int c,
                                 int ksize, int stride, int act, int dtype, void* stream);
extern "C" int advs_dwconv2d(const void* x, const float* w_taps_c, const float* bias, void* y, int b, int h, int w, int c,
                             int ksize, int stride, int dtype, void* stream) {
    return advs_dwconv2d_act(x, w_taps_c, bias, y, b, h, w, c, ksize, stride, ADVS_ACT_NONE, dtype, stream);
}
extern "C" int advs_dwconv2d_act(const void* x, const float* w_taps_c, const float* bias, void* y, int b, int h, int w, int c,
                                 int ksize, int stride, int act, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_dwconv2d: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && w_taps_c && y && b > 0 && h > 0 && w > 0 && c > 0, "dwconv2d: bad args");
    ADVS_REQUIRE((ksize & 1) && ksize >= 1 && ksize <= 7 && (stride == 1 || stride == 2), "dwconv2d: ksize %d / stride %d unsupported", ksize, stride);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "dwconv2d: c=%d must be a multiple of %d", c, vec);
    const int ho = (h + 2 * (ksize / 2) - ksize) / stride + 1, wo = (w + 2 * (ksize / 2) - ksize) / stride + 1;
    const size_t total = (size_t)b * ho * wo * (c / vec);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADVS_SWITCH_T(dtype, dwconv_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, w_taps_c, bias, (T*)y, b, h, w, c, ksize, stride, ho, wo, act));
    ADVS_CHECK_LAUNCH("dwconv2d");
    return ADVS_OK;
}

// space-to-depth by 2: [b][h][w][c] -> [b][h/2][w/2][4c], output channel (dy*2 + dx)*c + ch -- the K order of a
// Conv2d(c, cout, 2, stride 2) weight packed as [cout][dy][dx][c] (advs_pack_conv_weight), so the ConvNeXt
// downsampling conv is a 1x1 advs_conv2d on the result.
template <typename T>
__global__ void space_to_depth2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
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
        ((u32x4*)y)[i] = *((const u32x4*)(x + (((size_t)b * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C) + cv);
    }
}

extern "C" int advs_space_to_depth2(const void* x, void* y, int b, int h, int w, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_space_to_depth2: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && y && b > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "space_to_depth2: bad shape");
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "space_to_depth2: c=%d must be a multiple of %d", c, vec);
    const size_t total = (size_t)b * h * w * (c / vec);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADVS_SWITCH_T(dtype, space_to_depth2_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, (T*)y, b, h, w, c));
    ADVS_CHECK_LAUNCH("space_to_depth2");
    return ADVS_OK;
}


// ================================================================ Swin pieces (timm swin_base_patch4_window7_224, ASR_fast.py:27-32)
// Cyclic shift + window partition in one gather (SwinTransformerBlock): tokens of window (wy, wx) are the pixels
// ((wy*win + ty + shift) % H, (wx*win + tx + shift) % W) -- torch.roll(x, -shift) followed by window_partition.
// inverse != 0 runs it backwards (window_reverse + roll(+shift)) and adds the block's residual in the same pass:
// y[b][pixel] = windows[...] + residual[b][pixel].  x / y are [b][h][w][c] on the image side and
// [b * (h/win) * (w/win)][win*win][c] on the window side.  HBM-bound: read + write once.
template <typename T>
__global__ void window_shift_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                    int B, int H, int W, int C, int win, int shift, int inverse) {
    constexpr int VEC = Elt<T>::VEC;
    const int vpp = C / VEC, nwx = W / win, nwy = H / win;
    const size_t total = (size_t)B * H * W * vpp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        size_t r = i / vpp;                          // window-side token index
        const int tx = (int)(r % win); r /= win;
        const int ty = (int)(r % win); r /= win;
        const int wx = (int)(r % nwx); r /= nwx;
        const int wy = (int)(r % nwy);
        const int b = (int)(r / nwy);
        int py = wy * win + ty + shift, px = wx * win + tx + shift;
        if (py >= H) py -= H;
        if (px >= W) px -= W;
        const size_t img = (((size_t)b * H + py) * W + px) * vpp + cv;
        if (!inverse) {
            ((u32x4*)y)[i] = ((const u32x4*)x)[img];
        } else {
            float f[VEC], g[VEC];
            unpack16<T>(((const u32x4*)x)[i], f);
            if (res) {
                unpack16<T>(((const u32x4*)res)[img], g);
#pragma unroll
                for (int e = 0; e < VEC; ++e) f[e] += g[e];
            }
            ((u32x4*)y)[img] = pack16<T>(f);
        }
    }
}

extern "C" int advs_window_shift(const void* x, const void* residual, void* y, int b, int h, int w, int c, int window,
                                 int shift, int inverse, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_window_shift: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && y && b > 0 && h > 0 && w > 0 && window > 0 && h % window == 0 && w % window == 0,
                 "window_shift: %dx%d is not a whole number of %d-pixel windows", h, w, window);
    ADVS_REQUIRE(shift >= 0 && shift < window && (inverse || !residual), "window_shift: bad shift / residual");
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "window_shift: c=%d must be a multiple of %d", c, vec);
    const size_t total = (size_t)b * h * w * (c / vec);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADVS_SWITCH_T(dtype, window_shift_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, (const T*)residual, (T*)y, b, h, w, c, window, shift, inverse));
    ADVS_CHECK_LAUNCH("window_shift");
    return ADVS_OK;
}

// DINOv2 classifier input (Dinov2ForImageClassification.forward): y[b] = [ tokens[b][0] | mean(tokens[b][1 .. np]) ],
// tokens [B][n_pad][C] T -> y [B][2C] f32.  One workgroup per (64 channels, sample).
template <typename T>
__global__ void __launch_bounds__(256)
cls_mean_rows_kernel(const T* __restrict__ x, float* __restrict__ y, int n_pad, int np, int C) {
    __shared__ float part[256];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const T* xb = x + (size_t)b * n_pad * C;
    float s = 0.f;
    if (c < C)
        for (int r = 1 + q; r <= np; r += 4) s += Elt<T>::ld(xb + (size_t)r * C + c);
    part[threadIdx.x] = s;
    __syncthreads();
    if (q == 0 && c < C) {
        y[(size_t)b * 2 * C + c] = Elt<T>::ld(xb + c);
        y[(size_t)b * 2 * C + C + c] = (part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192]) / (float)np;
    }
}
extern "C" int advs_cls_mean_rows_f32(const void* tokens, float* y, int b, int n_pad, int np, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_cls_mean_rows_f32: unknown dtype code %d", dtype);
    ADVS_REQUIRE(tokens && y && b > 0 && np > 0 && np < n_pad && c > 0, "cls_mean_rows: bad args");
    dim3 grid(cdiv(c, 64), b);
    ADVS_SWITCH_T(dtype, cls_mean_rows_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)tokens, y, n_pad, np, c));
    ADVS_CHECK_LAUNCH("cls_mean_rows");
    return ADVS_OK;
}


// squeeze-and-excitation scale (torchvision SqueezeExcitation.forward: scale * input), s is f32 [B][C]
template <typename T>
__global__ void scale_channels_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ y, int B, int HW, int C) {
    constexpr int VEC = Elt<T>::VEC;
    const int vpp = C / VEC;
    const size_t total = (size_t)B * HW * vpp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        const int b = (int)(i / ((size_t)HW * vpp));
        float f[VEC];
        unpack16<T>(((const u32x4*)x)[i], f);
        const float* sp = s + (size_t)b * C + cv * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] *= sp[e];
        ((u32x4*)y)[i] = pack16<T>(f);
    }
}
extern "C" int advs_scale_channels(const void* x, const float* s, void* y, int b, int hw, int c, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_scale_channels: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && s && y && b > 0 && hw > 0 && c > 0, "scale_channels: bad args");
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(c % vec == 0, "scale_channels: c=%d must be a multiple of %d", c, vec);
    const size_t total = (size_t)b * hw * (c / vec);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADVS_SWITCH_T(dtype, scale_channels_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, s, (T*)y, b, hw, c));
    ADVS_CHECK_LAUNCH("scale_channels");
    return ADVS_OK;
}
