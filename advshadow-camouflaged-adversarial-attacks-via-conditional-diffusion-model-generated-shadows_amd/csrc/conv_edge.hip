// The two thin convolutions at the ends of the eps-predictor.  Neither is GEMM-shaped
// (K = 27 on the way in, N = 3 on the way out), so both are direct, HBM-bound kernels that
// also perform the NCHW <-> NHWC change of layout at the model boundary.
#include "common.h"

// ---------------------------------------------------------------- first conv: NCHW f32 -> NHWC T
// thread = one output pixel x one quarter of the output channels; weights broadcast from LDS.
// Traffic: reads B*cin*H*W*4 B (taps re-read from cache), writes B*H*W*cout*sizeof(T).
template <typename T>
__global__ void __launch_bounds__(256)
conv3x3_first_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                     T* __restrict__ y, int B, int Cin, int H, int W, int Cout) {
    extern __shared__ float sw[];                 // [cin*9][cout] then bias[cout]
    const int K = Cin * 9;
    for (int i = threadIdx.x; i < K * Cout; i += 256) {
        int k = i / Cout, o = i - k * Cout;       // w is [o][c][r][s] -> k = c*9 + r*3 + s
        sw[i] = w[(size_t)o * K + k];
    }
    for (int i = threadIdx.x; i < Cout; i += 256) sw[K * Cout + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int px = threadIdx.x & 63, cq = threadIdx.x >> 6;
    const long long npix = (long long)B * H * W;
    const int cper = Cout / 4;                    // channels per thread (multiple of 8)
    for (long long pb = (long long)blockIdx.x * 64; pb < npix; pb += (long long)gridDim.x * 64) {
        const long long pix = pb + px;
        if (pix >= npix) continue;
        const int b = (int)(pix / (H * W)), rem = (int)(pix - (long long)b * H * W);
        const int oy = rem / W, ox = rem - oy * W;
        float in[36];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int iy = oy + r - 1, ix = ox + s - 1;
                    const bool ok = c < Cin && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    in[c * 9 + r * 3 + s] = ok ? x[(((size_t)b * Cin + c) * H + iy) * W + ix] : 0.f;
                }
        T* yo = y + (size_t)pix * Cout + cq * cper;
        for (int c8 = 0; c8 < cper; c8 += 8) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = sw[K * Cout + cq * cper + c8 + j];
#pragma unroll
            for (int k = 0; k < 36; ++k) {
                if (k < K) {
                    const float v = in[k];
                    const float* wr = sw + k * Cout + cq * cper + c8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
                }
            }
            *(u32x4*)(yo + c8) = pack16<T>(acc);
            if (sizeof(T) == 4) *(u32x4*)(yo + c8 + 4) = pack16<T>(acc + 4);
        }
    }
}

extern "C" int advs_conv3x3_first(const float* x, const float* w, const float* bias, void* y,
                                  int b, int cin, int h, int wd, int cout, int dtype, void* stream) {
    ADVS_REQUIRE(x && w && y && b > 0 && h > 0 && wd > 0, "conv3x3_first: bad args");
    ADVS_REQUIRE(cin >= 1 && cin <= 4, "conv3x3_first: cin=%d must be <= 4", cin);
    ADVS_REQUIRE(cout % 32 == 0 && cout <= 512, "conv3x3_first: cout=%d must be a multiple of 32, <= 512", cout);
    const size_t lds = ((size_t)cin * 9 * cout + cout) * sizeof(float);
    const long long npix = (long long)b * h * wd;
    int grid = (int)((npix + 63) / 64 < 8192 ? (npix + 63) / 64 : 8192);
    if (dtype == ADVS_BF16)
        conv3x3_first_kernel<BF16><<<grid, 256, lds, (hipStream_t)stream>>>(x, w, bias, (BF16*)y, b, cin, h, wd, cout);
    else
        conv3x3_first_kernel<float><<<grid, 256, lds, (hipStream_t)stream>>>(x, w, bias, (float*)y, b, cin, h, wd, cout);
    ADVS_CHECK_LAUNCH("conv3x3_first");
    return ADVS_OK;
}

// ---------------------------------------------------------------- last conv: NHWC T -> NCHW f32
// 16 lanes cooperate on one output pixel, each owning 16-byte channel vectors; partial sums of the
// <= 4 outputs are folded with a 16-lane butterfly.  Traffic: reads B*H*W*cin*sizeof(T) (taps hit
// cache), writes B*cout*H*W*4 B.
template <typename T>
__global__ void __launch_bounds__(256)
conv_last_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                 float* __restrict__ y, int B, int Cin, int H, int W, int Cout, int R) {
    constexpr int VEC = Elt<T>::VEC;
    extern __shared__ float sw[];                 // [tap][cin][4]
    const int taps = R * R;
    for (int i = threadIdx.x; i < taps * Cin * 4; i += 256) {
        const int o = i & 3, c = (i >> 2) % Cin, t = (i >> 2) / Cin;
        sw[i] = o < Cout ? w[((size_t)o * Cin + c) * taps + t] : 0.f;
    }
    __syncthreads();
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;     // 16 pixel groups per block
    const int vpp = Cin / VEC, pad = R / 2;
    const long long npix = (long long)B * H * W;
    for (long long pb = (long long)blockIdx.x * 16; pb < npix; pb += (long long)gridDim.x * 16) {
        const long long pix = pb + grp;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int b = 0, oy = 0, ox = 0;
        const bool live = pix < npix;
        if (live) {
            b = (int)(pix / (H * W));
            const int rem = (int)(pix - (long long)b * H * W);
            oy = rem / W; ox = rem - oy * W;
            for (int t = 0; t < taps; ++t) {
                const int iy = oy + t / R - pad, ix = ox + t % R - pad;
                if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
                const u32x4* src = (const u32x4*)(x + (((size_t)b * H + iy) * W + ix) * Cin);
                for (int cv = l16; cv < vpp; cv += 16) {
                    float f[VEC];
                    unpack16<T>(src[cv], f);
                    const f32x4* wv = (const f32x4*)(sw + ((size_t)t * Cin + cv * VEC) * 4);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const f32x4 ww = wv[j];
                        acc[0] = fmaf(f[j], ww[0], acc[0]); acc[1] = fmaf(f[j], ww[1], acc[1]);
                        acc[2] = fmaf(f[j], ww[2], acc[2]); acc[3] = fmaf(f[j], ww[3], acc[3]);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += __shfl_xor(acc[j], o);
        }
        float mine = l16 == 0 ? acc[0] : l16 == 1 ? acc[1] : l16 == 2 ? acc[2] : acc[3];
        if (live && l16 < Cout)
            y[(((size_t)b * Cout + l16) * H + oy) * W + ox] = mine + (bias ? bias[l16] : 0.f);
    }
}

extern "C" int advs_conv_last(const void* x, const float* w, const float* bias, float* y,
                              int b, int cin, int h, int wd, int cout, int ksize, int dtype, void* stream) {
    ADVS_REQUIRE(x && w && y && b > 0 && h > 0 && wd > 0, "conv_last: bad args");
    ADVS_REQUIRE(cout >= 1 && cout <= 4, "conv_last: cout=%d must be <= 4", cout);
    ADVS_REQUIRE(ksize == 1 || ksize == 3, "conv_last: ksize %d unsupported", ksize);
    const int vec = dtype == ADVS_BF16 ? 8 : 4;
    ADVS_REQUIRE(cin % vec == 0, "conv_last: cin=%d must be a multiple of %d", cin, vec);
    const size_t lds = (size_t)ksize * ksize * cin * 4 * sizeof(float);
    ADVS_REQUIRE(lds <= 65536, "conv_last: cin=%d too large", cin);
    const long long npix = (long long)b * h * wd;
    int grid = (int)((npix + 15) / 16 < 16384 ? (npix + 15) / 16 : 16384);
    if (dtype == ADVS_BF16)
        conv_last_kernel<BF16><<<grid, 256, lds, (hipStream_t)stream>>>((const BF16*)x, w, bias, y, b, cin, h, wd, cout, ksize);
    else
        conv_last_kernel<float><<<grid, 256, lds, (hipStream_t)stream>>>((const float*)x, w, bias, y, b, cin, h, wd, cout, ksize);
    ADVS_CHECK_LAUNCH("conv_last");
    return ADVS_OK;
}
