// The two thin convolutions at the ends of the eps-predictor.  Neither is GEMM-shaped
// (K = 27 on the way in, N = 3 on the way out), so both are direct, HBM-bound kernels that
// also perform the NCHW <-> NHWC change of layout at the model boundary.
#include "common.h"

// ---------------------------------------------------------------- first conv: NCHW f32 -> NHWC T
// cout/8 consecutive lanes own a run of CF_PX consecutive output pixels of one row, 8 channels
// (16 B bf16 / 32 B f32) each, so a wave's stores cover whole contiguous pixel rows; per filter tap the
// lane's 8 weights are read once from LDS ([k][cout]) and reused for the CF_PX pixels; the inputs of a
// pixel are wave-merged broadcast loads.
// Traffic: reads B*cin*H*W*4 B (taps re-read from cache), writes B*H*W*cout*sizeof(T).
#define CF_PX 4
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
    const int lpp = Cout / 8, gpb = 256 / lpp;    // lanes per pixel run, runs per block pass
    const int gl = threadIdx.x / lpp, c8 = (threadIdx.x - gl * lpp) * 8;
    if (gl >= gpb) return;
    const int runs_per_row = (W + CF_PX - 1) / CF_PX;
    const long long nruns = (long long)B * H * runs_per_row;
    for (long long rb = (long long)blockIdx.x * gpb; rb < nruns; rb += (long long)gridDim.x * gpb) {
        const long long run = rb + gl;
        if (run >= nruns) continue;
        const int rr = (int)(run % runs_per_row);
        const long long rowid = run / runs_per_row;
        const int oy = (int)(rowid % H), b = (int)(rowid / H), ox0 = rr * CF_PX;
        float acc[CF_PX][8];
#pragma unroll
        for (int q = 0; q < CF_PX; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[q][j] = sw[K * Cout + c8 + j];
        for (int c = 0; c < Cin; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = oy + r - 1;
                if ((unsigned)iy >= (unsigned)H) continue;
                const float* xr = x + (((size_t)b * Cin + c) * H + iy) * W;
                float in[CF_PX + 2];
#pragma unroll
                for (int q = 0; q < CF_PX + 2; ++q) {
                    const int ix = ox0 + q - 1;
                    in[q] = (unsigned)ix < (unsigned)W ? xr[ix] : 0.f;
                }
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const float* wr = sw + (c * 9 + r * 3 + s) * Cout + c8;
                    float wv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) wv[j] = wr[j];
#pragma unroll
                    for (int q = 0; q < CF_PX; ++q)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[q][j] = fmaf(in[q + s], wv[j], acc[q][j]);
                }
            }
#pragma unroll
        for (int q = 0; q < CF_PX; ++q) {
            if (ox0 + q >= W) continue;
            T* yo = y + ((((size_t)b * H + oy) * W) + ox0 + q) * Cout + c8;
            *(u32x4*)yo = pack16<T>(acc[q]);
            if (sizeof(T) == 4) *(u32x4*)(yo + 4) = pack16<T>(acc[q] + 4);
        }
    }
}

extern "C" int advs_conv3x3_first(const float* x, const float* w, const float* bias, void* y,
                                  int b, int cin, int h, int wd, int cout, int dtype, void* stream) {
    ADVS_REQUIRE(x && w && y && b > 0 && h > 0 && wd > 0, "conv3x3_first: bad args");
    ADVS_REQUIRE(cin >= 1 && cin <= 4, "conv3x3_first: cin=%d must be <= 4", cin);
    ADVS_REQUIRE(cout % 8 == 0 && cout <= 512, "conv3x3_first: cout=%d must be a multiple of 8, <= 512", cout);
    const size_t lds = ((size_t)cin * 9 * cout + cout) * sizeof(float);
    const long long nruns = (long long)b * h * ((wd + CF_PX - 1) / CF_PX);
    const int gpb = 256 / (cout / 8);
    int grid = (int)((nruns + gpb - 1) / gpb < 16384 ? (nruns + gpb - 1) / gpb : 16384);
    ADVS_SWITCH_T(dtype, conv3x3_first_kernel<T><<<grid, 256, lds, (hipStream_t)stream>>>(x, w, bias, (T*)y, b, cin, h, wd, cout));
    ADVS_CHECK_LAUNCH("conv3x3_first");
    return ADVS_OK;
}

// ---------------------------------------------------------------- last conv: NHWC T -> NCHW f32
// 16 lanes cooperate on a run of PX consecutive output pixels of one image row, each lane owning
// 16-byte channel vectors.  Per filter tap the lane's weights are loaded once from LDS and reused for
// the PX pixels; the <= 4 partial outputs are folded with a 16-lane butterfly.
// Traffic: reads B*H*W*cin*sizeof(T) (taps hit cache), writes B*cout*H*W*4 B.
#define CL_PX 8
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
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;     // 16 pixel runs per block
    const int vpp = Cin / VEC, pad = R / 2;
    const int runs_per_row = (W + CL_PX - 1) / CL_PX;
    const long long nruns = (long long)B * H * runs_per_row;
    for (long long rb = (long long)blockIdx.x * 16; rb < nruns; rb += (long long)gridDim.x * 16) {
        const long long run = rb + grp;
        const bool live = run < nruns;
        float acc[CL_PX][4];
#pragma unroll
        for (int p = 0; p < CL_PX; ++p) { acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.f; }
        int b = 0, oy = 0, ox0 = 0;
        if (live) {
            const int rr = (int)(run % runs_per_row);
            const long long rowid = run / runs_per_row;
            oy = (int)(rowid % H); b = (int)(rowid / H); ox0 = rr * CL_PX;
            for (int cv = l16; cv < vpp; cv += 16) {
                for (int t = 0; t < taps; ++t) {
                    const int iy = oy + t / R - pad, dx = t % R - pad;
                    if ((unsigned)iy >= (unsigned)H) continue;
                    f32x4 wv[VEC];
                    const f32x4* wp = (const f32x4*)(sw + ((size_t)t * Cin + cv * VEC) * 4);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) wv[j] = wp[j];
                    const u32x4* rowp = (const u32x4*)(x + (((size_t)b * H + iy) * W) * Cin) + cv;
#pragma unroll
                    for (int p = 0; p < CL_PX; ++p) {
                        const int ix = ox0 + p + dx;
                        if ((unsigned)ix >= (unsigned)W) continue;
                        float f[VEC];
                        unpack16<T>(rowp[(size_t)ix * vpp], f);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            acc[p][0] = fmaf(f[j], wv[j][0], acc[p][0]); acc[p][1] = fmaf(f[j], wv[j][1], acc[p][1]);
                            acc[p][2] = fmaf(f[j], wv[j][2], acc[p][2]); acc[p][3] = fmaf(f[j], wv[j][3], acc[p][3]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
#pragma unroll
            for (int p = 0; p < CL_PX; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[p][j] += __shfl_xor(acc[p][j], o);
        // lane l16 stores (pixel p = l16 & 7, output o = l16 >> 3 and o + 2)
        if (live) {
            const int p = l16 & (CL_PX - 1), ob = l16 >> 3;
            float v0 = 0.f, v2 = 0.f;
#pragma unroll
            for (int q = 0; q < CL_PX; ++q)
                if (q == p) { v0 = ob ? acc[q][1] : acc[q][0]; v2 = ob ? acc[q][3] : acc[q][2]; }
            const int ox = ox0 + p;
            if (ox < W) {
                if (ob < Cout) y[(((size_t)b * Cout + ob) * H + oy) * W + ox] = v0 + (bias ? bias[ob] : 0.f);
                if (ob + 2 < Cout) y[(((size_t)b * Cout + ob + 2) * H + oy) * W + ox] = v2 + (bias ? bias[ob + 2] : 0.f);
            }
        }
    }
}

extern "C" int advs_conv_last(const void* x, const float* w, const float* bias, float* y,
                              int b, int cin, int h, int wd, int cout, int ksize, int dtype, void* stream) {
    ADVS_REQUIRE(x && w && y && b > 0 && h > 0 && wd > 0, "conv_last: bad args");
    ADVS_REQUIRE(cout >= 1 && cout <= 4, "conv_last: cout=%d must be <= 4", cout);
    ADVS_REQUIRE(ksize == 1 || ksize == 3, "conv_last: ksize %d unsupported", ksize);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(cin % vec == 0, "conv_last: cin=%d must be a multiple of %d", cin, vec);
    const size_t lds = (size_t)ksize * ksize * cin * 4 * sizeof(float);
    ADVS_REQUIRE(lds <= 65536, "conv_last: cin=%d too large", cin);
    const long long nruns = (long long)b * h * ((wd + CL_PX - 1) / CL_PX);
    int grid = (int)((nruns + 15) / 16 < 16384 ? (nruns + 15) / 16 : 16384);
    ADVS_SWITCH_T(dtype, conv_last_kernel<T><<<grid, 256, lds, (hipStream_t)stream>>>((const T*)x, w, bias, y, b, cin, h, wd, cout, ksize));
    ADVS_CHECK_LAUNCH("conv_last");
    return ADVS_OK;
}
