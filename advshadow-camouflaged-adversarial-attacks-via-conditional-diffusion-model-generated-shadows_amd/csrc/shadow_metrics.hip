// Shadow composites, PIL-exact resize, PSNR/SSIM and argmax: the stages between the sampler and
// the victim classifier.  All HBM-bound (or tiny) elementwise / stencil kernels; integer paths are
// bit-exact restatements of Pillow's fixed-point arithmetic.
#include "common.h"

// ============================================================================ apply_shadow (float)
// tools/train_shadow.py:242-256,262-266 with the adversarial perturbation absent:
//   m  = [sqrt((X-cx)^2 + (Y-cy)^2) <= r]                (create_shadow_mask, :156-174)
//   mb = GaussianBlur(m, k x k, sigma 0) (BORDER_REFLECT_101), separable taps supplied by the host
//   cm = mb * feature_mask
//   sh = img*(1-cm) + cm*(img*(1-intensity));  out = clamp(img*(1-cm) + sh*cm, 0, 1)
// Traffic: reads C+Cm planes, writes C planes of H*W f32.
struct ShadowP {
    const float* img; const float* fmask; const float* centers; const float* radii; float* out;
    float* shadowed; float* cmask;                // optional [B][C][H][W]: sh (not clamped) and cm, for the gradient attack
    int B, C, H, W, Cm, K;
    float intensity;
    float taps[7];
};

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

__global__ void apply_shadow_kernel(const ShadowP p) {
    const size_t hw = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * hw;
    const int r = p.K / 2;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / hw);
        const int rem = (int)(i - (size_t)b * hw);
        const int y = rem / p.W, x = rem - y * p.W;
        const float cx = p.centers[2 * b], cy = p.centers[2 * b + 1], rad = p.radii[b];
        float mb = 0.f;                                  // column pass over row-filtered values
        for (int j = -r; j <= r; ++j) {
            const int yy = reflect101(y + j, p.H);
            float row = 0.f;
            for (int k = -r; k <= r; ++k) {
                const int xx = reflect101(x + k, p.W);
                const float dx = (float)xx - cx, dy = (float)yy - cy;
                const float d = sqrtf(dx * dx + dy * dy);
                row += p.taps[k + r] * (d <= rad ? 1.f : 0.f);
            }
            mb += p.taps[j + r] * row;
        }
        for (int c = 0; c < p.C; ++c) {
            const float fm = p.fmask[((size_t)b * p.Cm + (p.Cm == 1 ? 0 : c)) * hw + rem];
            const float cm = mb * fm;
            const float v = p.img[((size_t)b * p.C + c) * hw + rem];
            const float keep = v * (1.f - cm);
            const float sh = keep + cm * (v * (1.f - p.intensity));
            const float o = keep + sh * cm;
            if (p.out) p.out[((size_t)b * p.C + c) * hw + rem] = fminf(fmaxf(o, 0.f), 1.f);
            if (p.shadowed) p.shadowed[((size_t)b * p.C + c) * hw + rem] = sh;
            if (p.cmask) p.cmask[((size_t)b * p.C + c) * hw + rem] = cm;
        }
    }
}

static int apply_shadow_launch(const float* img, const float* feature_mask, const float* centers, const float* radii,
                               float* out, float* shadowed, float* cmask, int b, int c, int h, int w, int mask_channels,
                               float intensity, const float* taps, int ntaps, void* stream) {
    ADVS_REQUIRE(img && feature_mask && centers && radii && (out || (shadowed && cmask)) && b > 0 && c > 0 && h > 0 && w > 0, "apply_shadow: bad args");
    ADVS_REQUIRE(mask_channels == 1 || mask_channels == c, "apply_shadow: feature mask must have 1 or %d channels", c);
    ADVS_REQUIRE(ntaps >= 1 && ntaps <= 7 && (ntaps & 1) && taps, "apply_shadow: blur kernel size %d unsupported", ntaps);
    ShadowP p;
    p.img = img; p.fmask = feature_mask; p.centers = centers; p.radii = radii; p.out = out;
    p.shadowed = shadowed; p.cmask = cmask;
    p.B = b; p.C = c; p.H = h; p.W = w; p.Cm = mask_channels; p.K = ntaps; p.intensity = intensity;
    for (int i = 0; i < 7; ++i) p.taps[i] = i < ntaps ? taps[i] : 0.f;
    const size_t total = (size_t)b * h * w;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    apply_shadow_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p);
    ADVS_CHECK_LAUNCH("apply_shadow");
    return ADVS_OK;
}

extern "C" int advs_apply_shadow(const float* img, const float* feature_mask, const float* centers, const float* radii,
                                 float* out, int b, int c, int h, int w, int mask_channels, float intensity,
                                 const float* taps, int ntaps, void* stream) {
    ADVS_REQUIRE(out, "apply_shadow: bad args");
    return apply_shadow_launch(img, feature_mask, centers, radii, out, nullptr, nullptr, b, c, h, w, mask_channels, intensity, taps, ntaps, stream);
}

// The two intermediates the gradient attack needs (train_shadow.py:250-256): shadowed image (not clamped) and the
// combined mask cm, both [b][c][h][w].
extern "C" int advs_apply_shadow_parts(const float* img, const float* feature_mask, const float* centers, const float* radii,
                                       float* shadowed, float* cmask, int b, int c, int h, int w, int mask_channels,
                                       float intensity, const float* taps, int ntaps, void* stream) {
    ADVS_REQUIRE(shadowed && cmask, "apply_shadow_parts: bad args");
    return apply_shadow_launch(img, feature_mask, centers, radii, nullptr, shadowed, cmask, b, c, h, w, mask_channels, intensity, taps, ntaps, stream);
}

// out = clamp(img * (1 - cm) + adv * cm, 0, 1)    (train_shadow.py:262-265)
__global__ void blend_mask_clamp01_kernel(const float* __restrict__ img, const float* __restrict__ adv, const float* __restrict__ cm,
                                          float* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float m = cm[i];
        out[i] = fminf(fmaxf(img[i] * (1.f - m) + adv[i] * m, 0.f), 1.f);
    }
}

extern "C" int advs_blend_mask_clamp01(const float* img, const float* adv, const float* cmask, float* out, long long n, void* stream) {
    ADVS_REQUIRE(img && adv && cmask && out && n > 0, "blend_mask_clamp01: bad args");
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    blend_mask_clamp01_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(img, adv, cmask, out, (size_t)n);
    ADVS_CHECK_LAUNCH("blend_mask_clamp01");
    return ADVS_OK;
}

// ============================================================================ PIL composites (uint8)
// Pillow fixed point: SHIFTFORDIV255(a) = ((a>>8)+a)>>8.
__device__ __forceinline__ unsigned shiftdiv255(unsigned a) { return ((a >> 8) + a) >> 8; }
// Paste.c BLEND8 as Pillow >= 10 computes it (one rounding): DIV255(in1*(255-mask) + in2*mask).
// Checked against the installed Pillow 12.2 by tests/test_gpu_shadow_metrics.py.
__device__ __forceinline__ unsigned blend8(unsigned mask, unsigned in1, unsigned in2) {
    return shiftdiv255(in1 * (255u - mask) + in2 * mask + 128u);
}
// Image.alpha_composite(dst (opaque RGB), src RGBA) -> RGB  (libImaging/AlphaComposite.c, dst alpha 255)
__device__ __forceinline__ void alpha_composite_px(const unsigned* dst, const unsigned* src, unsigned* out) {
    const unsigned sa = src[3];
    if (sa == 0) { out[0] = dst[0]; out[1] = dst[1]; out[2] = dst[2]; return; }
    const unsigned blend = 255u * (255u - sa);
    const unsigned outa255 = sa * 255u + blend;
    const unsigned coef1 = sa * 255u * 255u * 128u / outa255;
    const unsigned coef2 = 255u * 128u - coef1;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = shiftdiv255(src[c] * coef1 + dst[c] * coef2 + (0x80u << 7)) >> 7;
}

// mode 0 (add_shadow.py:57-58): out = Image.composite(alpha_composite(img, layer), img, pmask)
// mode 1 (shadow_for_attack.py:76-93): the layer is first pasted onto (255,255,255,0) through
//   L(layer) & pmask, alpha-composited, then every ELEMENT whose darkening mask is nonzero (dmask: one value per pixel, or one
//   per channel when the mask image has three) is scaled by `factor` in float32, clipped to [0,255] and truncated
//   (shadow_for_attack.py:50-73).  layer / pmask are already on the image's grid (the host pastes at (0,0) as Pillow does).
__global__ void composite_u8_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ layer,
                                    const uint8_t* __restrict__ pmask, const uint8_t* __restrict__ dmask, int dch,
                                    uint8_t* __restrict__ out, size_t npix, int mode, float factor) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        unsigned d[3] = {img[3 * i], img[3 * i + 1], img[3 * i + 2]};
        unsigned s[4] = {layer[4 * i], layer[4 * i + 1], layer[4 * i + 2], layer[4 * i + 3]};
        const unsigned m = pmask[i];
        unsigned o[3];
        if (mode == 0) {
            unsigned c[3];
            alpha_composite_px(d, s, c);
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = blend8(m, d[k], c[k]);
        } else {
            // L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16   (Convert.c L24)
            const unsigned L = (s[0] * 19595u + s[1] * 38470u + s[2] * 7471u + 0x8000u) >> 16;
            const unsigned pm = L & m;
            const unsigned base[4] = {255u, 255u, 255u, 0u};
            unsigned ly[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ly[k] = blend8(pm, base[k], s[k]);
            alpha_composite_px(d, ly, o);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (dmask[dch == 1 ? i : 3 * i + k] != 0) {
                    float f = (float)o[k] * factor;
                    f = fminf(fmaxf(f, 0.f), 255.f);
                    o[k] = (unsigned)f;
                }
            }
        }
        out[3 * i] = (uint8_t)o[0]; out[3 * i + 1] = (uint8_t)o[1]; out[3 * i + 2] = (uint8_t)o[2];
    }
}

extern "C" int advs_composite_u8_masks(const uint8_t* img_hwc, const uint8_t* layer_rgba, const uint8_t* paste_mask,
                                       const uint8_t* dark_mask, int dark_channels, uint8_t* out_hwc, size_t npix, int mode,
                                       float factor, void* stream) {
    ADVS_REQUIRE(img_hwc && layer_rgba && paste_mask && dark_mask && out_hwc && npix > 0, "composite_u8: bad args");
    ADVS_REQUIRE(mode == 0 || mode == 1, "composite_u8: mode %d unknown", mode);
    ADVS_REQUIRE(dark_channels == 1 || dark_channels == 3, "composite_u8: the darkening mask has 1 or 3 channels, not %d", dark_channels);
    const int grid = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
    composite_u8_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(img_hwc, layer_rgba, paste_mask, dark_mask, dark_channels, out_hwc,
                                                              npix, mode, factor);
    ADVS_CHECK_LAUNCH("composite_u8");
    return ADVS_OK;
}
extern "C" int advs_composite_u8(const uint8_t* img_hwc, const uint8_t* layer_rgba, const uint8_t* mask,
                                 uint8_t* out_hwc, size_t npix, int mode, float factor, void* stream) {
    return advs_composite_u8_masks(img_hwc, layer_rgba, mask, mask, 1, out_hwc, npix, mode, factor, stream);
}

// ============================================================================ PIL resize (uint8, 2 passes)
// Pillow's ImagingResample for 8-bit images: coefficients are precomputed on the host exactly as
// precompute_coeffs()/normalize_coeffs_8bpc() do (int32, PRECISION_BITS = 22); one pass is
//   out = clip8( (1<<21) + sum_k in[xmin+k] * coef[k] )  with clip8(v) = clamp(v >> 22, 0, 255).
// `horizontal` selects the axis.  Images are [n][H][W][ch] uint8.
__global__ void resample_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                   const int* __restrict__ bounds, const int* __restrict__ coefs, int ksize,
                                   int n, int inH, int inW, int outH, int outW, int ch, int horizontal) {
    const size_t total = (size_t)n * outH * outW * ch;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ch);
        size_t r = i / ch;
        const int ox = (int)(r % outW); r /= outW;
        const int oy = (int)(r % outH);
        const int b = (int)(r / outH);
        const int o = horizontal ? ox : oy;
        const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
        const int* k = coefs + (size_t)o * ksize;
        int ss = 1 << 21;
        for (int j = 0; j < cnt; ++j) {
            const int iy = horizontal ? oy : lo + j, ix = horizontal ? lo + j : ox;
            ss += (int)in[(((size_t)b * inH + iy) * inW + ix) * ch + c] * k[j];
        }
        ss >>= 22;
        out[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
}

extern "C" int advs_resample_u8(const uint8_t* in, uint8_t* out, const int* bounds, const int* coefs, int ksize,
                                int n, int in_h, int in_w, int out_h, int out_w, int channels, int horizontal,
                                void* stream) {
    ADVS_REQUIRE(in && out && bounds && coefs && ksize > 0 && n > 0 && channels > 0, "resample_u8: bad args");
    ADVS_REQUIRE(horizontal ? in_h == out_h : in_w == out_w, "resample_u8: the other axis must keep its size");
    const size_t total = (size_t)n * out_h * out_w * channels;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    resample_u8_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, bounds, coefs, ksize, n, in_h, in_w, out_h, out_w,
                                                               channels, horizontal);
    ADVS_CHECK_LAUNCH("resample_u8");
    return ADVS_OK;
}

// uint8 HWC -> float NCHW / 255 (transforms.ToTensor, ASR_fast.py:94), optionally (x-mean)/std per channel.
__global__ void u8hwc_to_f32nchw_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int n, int h, int w,
                                        int ch, const float* __restrict__ mean, const float* __restrict__ stdv) {
    const size_t total = (size_t)n * ch * h * w;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        size_t r = i / w;
        const int y = (int)(r % h); r /= h;
        const int c = (int)(r % ch);
        const int b = (int)(r / ch);
        float v = (float)in[(((size_t)b * h + y) * w + x) * ch + c] / 255.0f;
        if (mean) v = (v - mean[c]) / stdv[c];
        out[i] = v;
    }
}
extern "C" int advs_u8hwc_to_f32nchw(const uint8_t* in, float* out, int n, int h, int w, int channels,
                                     const float* mean, const float* stdv, void* stream) {
    ADVS_REQUIRE(in && out && n > 0 && h > 0 && w > 0 && channels > 0, "u8hwc_to_f32nchw: bad args");
    ADVS_REQUIRE((mean == nullptr) == (stdv == nullptr), "u8hwc_to_f32nchw: mean/std must come together");
    const size_t total = (size_t)n * h * w * channels;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    u8hwc_to_f32nchw_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, n, h, w, channels, mean, stdv);
    ADVS_CHECK_LAUNCH("u8hwc_to_f32nchw");
    return ADVS_OK;
}
// float NCHW (x*scale truncated, wrap or clamp) -> uint8 HWC: the tensor->PIL hand-off of save_images
// (utils/utils.py:59-61) kept on the device.
__global__ void u8nchw_to_hwc_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int n, int ch, int h, int w) {
    const size_t total = (size_t)n * ch * h * w;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ch);
        size_t r = i / ch;
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h);
        const int b = (int)(r / h);
        out[i] = in[(((size_t)b * ch + c) * h + y) * w + x];
    }
}
extern "C" int advs_u8_nchw_to_hwc(const uint8_t* in, uint8_t* out, int n, int channels, int h, int w, void* stream) {
    ADVS_REQUIRE(in && out && n > 0 && h > 0 && w > 0 && channels > 0, "u8_nchw_to_hwc: bad args");
    const size_t total = (size_t)n * h * w * channels;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    u8nchw_to_hwc_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, n, channels, h, w);
    ADVS_CHECK_LAUNCH("u8_nchw_to_hwc");
    return ADVS_OK;
}

// ============================================================================ PSNR / SSIM
// PSNR_SSIM_fast.py:21-26 with skimage semantics: per channel, gaussian window sigma 1.5 truncated at
// 3.5 sigma (11 taps, scipy 'reflect' = half-sample symmetric border), each 1-D pass accumulated in
// f64 and stored in f32 (scipy.ndimage on float32 input), sample covariance NP/(NP-1) with
// NP = win_size^2, K1 .01, K2 .03, data_range = max-min of image1, mean over the interior
// (crop (win_size-1)/2) in f64, then mean over channels.  PSNR = 10 log10(R^2 / mse) in f64.
// One workgroup per image; planes up to 64x64 live in LDS.
#define SS_MAX 64
#define SS_R 5
__global__ void __launch_bounds__(256)
psnr_ssim_kernel(const float* __restrict__ im1, const float* __restrict__ im2, double* __restrict__ out,
                 int C, int H, int W, int win) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];   // 6 planes of H*W floats
    float* pl[5];                                 // x, y, xx, yy, xy (then filtered in place)
    for (int k = 0; k < 5; ++k) pl[k] = dyn + (size_t)k * H * W;
    float* tmp = dyn + (size_t)5 * H * W;
    __shared__ double red[256];
    __shared__ float redf[2][256];
    __shared__ double wts[2 * SS_R + 1];
    const int b = blockIdx.x, tid = threadIdx.x, hw = H * W;
    const float* a = im1 + (size_t)b * C * hw;
    const float* q = im2 + (size_t)b * C * hw;
    if (tid == 0) {
        double s = 0.0;
        for (int i = -SS_R; i <= SS_R; ++i) { wts[i + SS_R] = exp(-0.5 / (1.5 * 1.5) * (double)(i * i)); s += wts[i + SS_R]; }
        for (int i = 0; i <= 2 * SS_R; ++i) wts[i] /= s;
    }
    // data range of image1 (all channels) and the squared error
    float mn = INFINITY, mx = -INFINITY;
    double se = 0.0;
    for (int i = tid; i < C * hw; i += 256) {
        const float v = a[i], d = v - q[i];
        mn = fminf(mn, v); mx = fmaxf(mx, v);
        se += (double)(d * d);
    }
    redf[0][tid] = mn; redf[1][tid] = mx; red[tid] = se;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            redf[0][tid] = fminf(redf[0][tid], redf[0][tid + s]);
            redf[1][tid] = fmaxf(redf[1][tid], redf[1][tid + s]);
            red[tid] += red[tid + s];
        }
        __syncthreads();
    }
    const float R = redf[1][0] - redf[0][0];
    const double mse = red[0] / (double)(C * hw);
    __syncthreads();
    const double NP = (double)win * win, cov_norm = NP / (NP - 1.0);
    const float C1 = (0.01f * R) * (0.01f * R), C2 = (0.03f * R) * (0.03f * R);
    const int pad = (win - 1) / 2;
    double ssim_sum = 0.0;
    for (int c = 0; c < C; ++c) {
        for (int i = tid; i < hw; i += 256) {
            const float x = a[c * hw + i], y = q[c * hw + i];
            pl[0][i] = x; pl[1][i] = y; pl[2][i] = x * x; pl[3][i] = y * y; pl[4][i] = x * y;
        }
        __syncthreads();
        for (int k = 0; k < 5; ++k) {
            for (int i = tid; i < hw; i += 256) {              // axis 0 (rows index) first, as scipy does
                const int y = i / W, x = i - y * W;
                double acc = 0.0;
                for (int j = -SS_R; j <= SS_R; ++j) {
                    int yy = y + j;
                    while (yy < 0 || yy >= H) yy = yy < 0 ? -yy - 1 : 2 * H - 1 - yy;
                    acc += wts[j + SS_R] * (double)pl[k][yy * W + x];
                }
                tmp[i] = (float)acc;
            }
            __syncthreads();
            for (int i = tid; i < hw; i += 256) {
                const int y = i / W, x = i - y * W;
                double acc = 0.0;
                for (int j = -SS_R; j <= SS_R; ++j) {
                    int xx = x + j;
                    while (xx < 0 || xx >= W) xx = xx < 0 ? -xx - 1 : 2 * W - 1 - xx;
                    acc += wts[j + SS_R] * (double)tmp[y * W + xx];
                }
                pl[k][i] = (float)acc;
            }
            __syncthreads();
        }
        double part = 0.0;
        for (int i = tid; i < hw; i += 256) {
            const int y = i / W, x = i - y * W;
            if (y < pad || y >= H - pad || x < pad || x >= W - pad) continue;
            const float ux = pl[0][i], uy = pl[1][i];
            const float vx = (float)cov_norm * (pl[2][i] - ux * ux);
            const float vy = (float)cov_norm * (pl[3][i] - uy * uy);
            const float vxy = (float)cov_norm * (pl[4][i] - ux * uy);
            const float A1 = 2.f * ux * uy + C1, A2 = 2.f * vxy + C2;
            const float B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
            part += (double)((A1 * A2) / (B1 * B2));
        }
        red[tid] = part;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
        if (tid == 0) ssim_sum += red[0] / (double)((H - 2 * pad) * (W - 2 * pad));
        __syncthreads();
    }
    if (tid == 0) {
        out[2 * b] = ssim_sum / (double)C;
        out[2 * b + 1] = 10.0 * log10((double)(R * R) / mse);
    }
}

extern "C" int advs_psnr_ssim(const float* img1, const float* img2, double* out_ssim_psnr, int b, int c, int h, int w,
                              int win_size, void* stream) {
    ADVS_REQUIRE(img1 && img2 && out_ssim_psnr && b > 0 && c > 0, "psnr_ssim: bad args");
    ADVS_REQUIRE(h <= SS_MAX && w <= SS_MAX && h > 0 && w > 0, "psnr_ssim: planes up to %dx%d (PSNR_SSIM_fast.py resizes to 64)", SS_MAX, SS_MAX);
    ADVS_REQUIRE(win_size >= 3 && (win_size & 1) && win_size <= h && win_size <= w, "psnr_ssim: bad win_size %d", win_size);
    const int lds = 6 * h * w * (int)sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        ADVS_HIP(hipFuncSetAttribute((const void*)psnr_ssim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     6 * SS_MAX * SS_MAX * (int)sizeof(float)));
        attr_set = true;
    }
    psnr_ssim_kernel<<<b, 256, lds, (hipStream_t)stream>>>(img1, img2, out_ssim_psnr, c, h, w, win_size);
    ADVS_CHECK_LAUNCH("psnr_ssim");
    return ADVS_OK;
}

// ============================================================================ argmax over logits rows
__global__ void argmax_rows_kernel(const float* __restrict__ x, int* __restrict__ out, int rows, int n) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= rows) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const float v = x[(size_t)row * n + i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) out[row] = bi;
}
extern "C" int advs_argmax_rows(const float* x, int* out, int rows, int n, void* stream) {
    ADVS_REQUIRE(x && out && rows > 0 && n > 0, "argmax_rows: bad args");
    argmax_rows_kernel<<<cdiv((long long)rows, 4), 256, 0, (hipStream_t)stream>>>(x, out, rows, n);
    ADVS_CHECK_LAUNCH("argmax_rows");
    return ADVS_OK;
}
