// The two thin convolutions at the ends of the eps-predictor.  Neither is GEMM-shaped
// (K = 27 on the way in, N = 3 on the way out), so both are direct, HBM-bound kernels that
// also perform the NCHW <-> NHWC change of layout at the model boundary.
#include "common.h"
#include <type_traits>

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

typedef __attribute__((ext_vector_type(8))) __bf16 mbf16x8;
template <typename T> __device__ __forceinline__ f32x4 mma16x16(const u32x4& a, const u32x4& b, const f32x4& c);
template <> __device__ __forceinline__ f32x4 mma16x16<BF16>(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mbf16x8, a), __builtin_bit_cast(mbf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16x16<F16>(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}


// ---------------------------------------------------------------- first conv on the matrix cores (16-bit modes)
// K = 9 * cin <= 32 is ONE k-step of v_mfma_f32_16x16x32.  The product is computed transposed -- A = weights
// (16 channels x K), B = im2col (K x 16 pixels) -- so a lane ends up with 4 consecutive channels of ITS pixel;
// two tiles whose channel rows interleave (32j + 8g + {0..3} and + {4..7}) give it 8 consecutive channels = one
// 16-byte NHWC store.  The im2col fragment is gathered straight from the NCHW f32 input (8 cached scalar loads
// per lane).  A wave owns a contiguous chunk of FS_TILES 16-pixel tiles of one image and (optionally) leaves
// per-channel (sum, sum of squares) of the ROUNDED outputs of that chunk for the GroupNorm that follows
// (same [row block][channel][2] layout as advs_conv2d's epilogue statistics).
#define FS_TILES 64                              // 1024 pixels per statistics block
template <typename T, int NP>
__global__ void __launch_bounds__(256)
conv_first_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                       T* __restrict__ y, float* __restrict__ stats, int B, int Cin, int H, int W) {
    constexpr int Cout = NP * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4, K = Cin * 9, HW = H * W;
    // A fragments: tile (pair j, half hh), row n <-> channel 32j + 8(n>>2) + 4hh + (n&3); k = 8g .. 8g+7
    u32x4 af[NP][2];
    float bv[NP][8];
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int ch = 32 * j + 8 * (n >> 2) + 4 * hh + (n & 3);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k0 = 8 * g + 2 * q;
                af[j][hh][q] = pack2<T>(k0 < K ? w[(size_t)ch * K + k0] : 0.f, k0 + 1 < K ? w[(size_t)ch * K + k0 + 1] : 0.f);
            }
        }
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) bv[j][i] = bias ? bias[32 * j + 8 * g + i] : 0.f;      // D rows of this lane
    // im2col taps of this lane: k = 8g + q -> (channel, dy, dx)
    int tc[8], tdy[8], tdx[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = 8 * g + q, c = k / 9, r9 = k - 9 * c;
        tc[q] = k < K ? c : -1; tdy[q] = r9 / 3 - 1; tdx[q] = r9 % 3 - 1;
    }
    const int cpi = HW / (16 * FS_TILES) > 0 ? (HW + 16 * FS_TILES - 1) / (16 * FS_TILES) : 1;     // chunks per image
    const int nchunks = B * cpi, tiles_img = HW >> 4;
    for (int chunk = blockIdx.x * 4 + wave; chunk < nchunks; chunk += gridDim.x * 4) {
        const int b = chunk / cpi, ci = chunk - b * cpi;
        const int t0 = ci * FS_TILES, t1 = min(tiles_img, t0 + FS_TILES);
        float ssum[NP][8], ssq[NP][8];
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) { ssum[j][i] = 0.f; ssq[j][i] = 0.f; }
        const float* xb = x + (size_t)b * Cin * HW;
        for (int t = t0; t < t1; ++t) {
            const int p = t * 16 + n, oy = p / W, ox = p - oy * W;
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int iy = oy + tdy[q], ix = ox + tdx[q];
                const bool ok = tc[q] >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const float* src = ok ? xb + ((size_t)tc[q] * H + iy) * W + ix : xb;
                const float val = *src;
                v[q] = ok ? val : 0.f;
            }
            const u32x4 bf = u32x4{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])};
            T* yo = y + ((size_t)b * HW + p) * Cout + 8 * g;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 d0 = mma16x16<T>(af[j][0], bf, z), d1 = mma16x16<T>(af[j][1], bf, z);
                float o[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { o[i] = d0[i] + bv[j][i]; o[4 + i] = d1[i] + bv[j][4 + i]; }
                const u32x4 packed = pack16<T>(o);
                *(u32x4*)(yo + 32 * j) = packed;
                if (stats) {
                    float rr[8];
                    unpack16<T>(packed, rr);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { ssum[j][i] += rr[i]; ssq[j][i] = fmaf(rr[i], rr[i], ssq[j][i]); }
                }
            }
        }
        if (stats) {                              // fold the 16 pixel lanes, fixed order; lane n == 0 writes
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float a = ssum[j][i], q2 = ssq[j][i];
#pragma unroll
                    for (int m = 8; m > 0; m >>= 1) { a += __shfl_xor(a, m); q2 += __shfl_xor(q2, m); }
                    if (n == 0) {
                        float* so = stats + (((size_t)b * cpi + ci) * Cout + 32 * j + 8 * g + i) * 2;
                        so[0] = a; so[1] = q2;
                    }
                }
        }
    }
}

// Pixels per statistics block of the MFMA first conv for this shape, or 0 when that path (or its statistics)
// does not apply: 16-bit dtype, 9 * cin <= 32, cout 64 or 128, w a multiple of 16, h * w a multiple of 1024.
extern "C" int advs_conv_first_stats_rows(int cin, int h, int w, int cout, int dtype) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_conv_first_stats_rows: unknown dtype code %d", dtype);
    if (dtype == ADVS_F32 || cin < 1 || cin * 9 > 32 || (cout != 64 && cout != 128) || w % 16) return 0;
    return ((long long)h * w) % (16 * FS_TILES) == 0 ? 16 * FS_TILES : 0;
}

template <typename T>
static int conv_first_mfma_launch(const float* x, const float* w, const float* bias, void* y, float* stats, int b, int cin,
                                  int h, int wd, int cout, hipStream_t st) {
    const long long hw = (long long)h * wd;
    ADVS_REQUIRE(hw * b < (1ll << 31), "conv3x3_first: too many pixels");
    const long long cpi = (hw + 16 * FS_TILES - 1) / (16 * FS_TILES);
    const long long nchunks = cpi * b;
    const int grid = (int)((nchunks + 3) / 4 < 4096 ? (nchunks + 3) / 4 : 4096);
    if (cout == 128) conv_first_mfma_kernel<T, 4><<<grid, 256, 0, st>>>(x, w, bias, (T*)y, stats, b, cin, h, wd);
    else conv_first_mfma_kernel<T, 2><<<grid, 256, 0, st>>>(x, w, bias, (T*)y, stats, b, cin, h, wd);
    ADVS_CHECK_LAUNCH("conv3x3_first");
    return ADVS_OK;
}

extern "C" int advs_conv3x3_first_stats(const float* x, const float* w, const float* bias, void* y, float* stats,
                                        int b, int cin, int h, int wd, int cout, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_conv3x3_first_stats: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && w && y && b > 0 && h > 0 && wd > 0, "conv3x3_first: bad args");
    ADVS_REQUIRE(cin >= 1 && cin <= 4, "conv3x3_first: cin=%d must be <= 4", cin);
    ADVS_REQUIRE(cout % 8 == 0 && cout <= 512, "conv3x3_first: cout=%d must be a multiple of 8, <= 512", cout);
    const bool mfma = dtype != ADVS_F32 && cin * 9 <= 32 && (cout == 64 || cout == 128) && wd % 16 == 0;
    ADVS_REQUIRE(!stats || advs_conv_first_stats_rows(cin, h, wd, cout, dtype) > 0,
                 "conv3x3_first: statistics are not available for this shape (advs_conv_first_stats_rows == 0)");
    if (mfma) {
        if (dtype == ADVS_BF16) return conv_first_mfma_launch<BF16>(x, w, bias, y, stats, b, cin, h, wd, cout, (hipStream_t)stream);
        return conv_first_mfma_launch<F16>(x, w, bias, y, stats, b, cin, h, wd, cout, (hipStream_t)stream);
    }
    const size_t lds = ((size_t)cin * 9 * cout + cout) * sizeof(float);
    const long long nruns = (long long)b * h * ((wd + CF_PX - 1) / CF_PX);
    const int gpb = 256 / (cout / 8);
    int grid = (int)((nruns + gpb - 1) / gpb < 16384 ? (nruns + gpb - 1) / gpb : 16384);
    ADVS_SWITCH_T(dtype, conv3x3_first_kernel<T><<<grid, 256, lds, (hipStream_t)stream>>>(x, w, bias, (T*)y, b, cin, h, wd, cout));
    ADVS_CHECK_LAUNCH("conv3x3_first");
    return ADVS_OK;
}

extern "C" int advs_conv3x3_first(const float* x, const float* w, const float* bias, void* y,
                                  int b, int cin, int h, int wd, int cout, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_conv3x3_first: unknown dtype code %d", dtype);
    return advs_conv3x3_first_stats(x, w, bias, y, nullptr, b, cin, h, wd, cout, dtype, stream);
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
    const int nb = gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = blockIdx.x & 7;     // XCD-contiguous row bands (see below)
    const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
    for (long long rb = (long long)bid * 16; rb < nruns; rb += (long long)nb * 16) {
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

// ---------------------------------------------------------------- last conv, 16-bit activations
// Same contract, written around v_dot2c_f32_{bf16,f16}: the activations stay packed (no unpack), the
// weights are rounded to T like every other conv's in the 16-bit modes and parked in LDS as channel pairs
// ([tap][channel vector][output][4 x u32]).  16 lanes own a run of 8 output pixels of one row; per filter
// row the run's 10 input vectors are loaded once and serve the three horizontal taps.  The 24 partial sums
// (8 pixels x 3 outputs) are folded over the 16 lanes with a halving butterfly (24 shuffles instead of 96).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
template <typename T> __device__ __forceinline__ float dot2acc(unsigned a, unsigned b, float c);
template <> __device__ __forceinline__ float dot2acc<BF16>(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
template <> __device__ __forceinline__ float dot2acc<F16>(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
}

template <typename T, int R>
__global__ void __launch_bounds__(256)
conv_last16_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                   float* __restrict__ y, int B, int Cin, int H, int W, int Cout, const u32x4* __restrict__ zero) {
    constexpr int TAPS = R * R, PAD = R / 2, PXR = 8;
    extern __shared__ u32x4 swq[];                // [tap][vpp][3]
    const int vpp = Cin / 8;
    for (int i = threadIdx.x; i < TAPS * vpp * 3; i += 256) {
        const int o = i % 3, cv = (i / 3) % vpp, t = i / (3 * vpp);
        u32x4 q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = cv * 8 + 2 * j;
            const float lo = o < Cout ? w[((size_t)o * Cin + c) * TAPS + t] : 0.f;
            const float hi = o < Cout ? w[((size_t)o * Cin + c + 1) * TAPS + t] : 0.f;
            q[j] = pack2<T>(lo, hi);
        }
        swq[i] = q;
    }
    __syncthreads();
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int runs_per_row = (W + PXR - 1) / PXR;
    const int nruns = B * H * runs_per_row;             // < 2^31, checked by the host
    // Workgroups are dealt round-robin to the 8 XCDs.  Give each XCD a contiguous band of rows, so that the two
    // neighbour rows every output row needs are found in that XCD's own L2 instead of being fetched by three XCDs.
    const int nb = gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = blockIdx.x & 7;
    const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
    for (int rb = bid * 16; rb < nruns; rb += nb * 16) {
        const int run = rb + grp;
        const bool live = run < nruns;
        float acc[PXR * 3];
#pragma unroll
        for (int i = 0; i < PXR * 3; ++i) acc[i] = 0.f;
        int b = 0, oy = 0, ox0 = 0;
        if (live) {
            const int rowid = run / runs_per_row, rr = run - rowid * runs_per_row;
            b = rowid / H; oy = rowid - b * H; ox0 = rr * PXR;
            for (int cv = l16; cv < vpp; cv += 16) {
#pragma unroll 1                                  // one filter row's inputs live at a time
                for (int r = 0; r < R; ++r) {
                    const int iy = oy + r - PAD;
                    if ((unsigned)iy >= (unsigned)H) continue;
                    const u32x4* rowp = (const u32x4*)(x + (((size_t)b * H + iy) * W) * Cin) + cv;
                    int opaque = 0;                   // keeps the weight reads inside the loop: 12 live registers
                    asm volatile("" : "+v"(opaque));  // instead of 108, i.e. 5 instead of 2 waves per SIMD
                    u32x4 in[PXR + 2 * PAD];
#pragma unroll
                    for (int q = 0; q < PXR + 2 * PAD; ++q) {
                        // unconditional load through a selected POINTER (zero page outside the row): a conditional
                        // load becomes a branch with its own wait, which serialises the ten loads of the row
                        const int ix = ox0 + q - PAD;
                        in[q] = *((unsigned)ix < (unsigned)W ? rowp + (size_t)ix * vpp : zero);
                    }
#pragma unroll
                    for (int sx = 0; sx < R; ++sx) {
                        const u32x4* wq = swq + ((r * R + sx) * vpp + cv) * 3 + opaque;
                        const u32x4 w0 = wq[0], w1 = wq[1], w2 = wq[2];
#pragma unroll
                        for (int p = 0; p < PXR; ++p)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                acc[p * 3] = dot2acc<T>(in[p + sx][j], w0[j], acc[p * 3]);
                                acc[p * 3 + 1] = dot2acc<T>(in[p + sx][j], w1[j], acc[p * 3 + 1]);
                                acc[p * 3 + 2] = dot2acc<T>(in[p + sx][j], w2[j], acc[p * 3 + 2]);
                            }
                    }
                }
            }
        }
        // halving butterfly: after the steps with lane bits 8, 4, 2 the lane holds the three outputs of pixel
        // p = 4*bit3 + 2*bit2 + bit1 summed over its partners; the last step folds bit 0.
#pragma unroll
        for (int n = 12, m = 8; n >= 3; n >>= 1, m >>= 1) {
            const bool upper = (l16 & m) != 0;
#pragma unroll
            for (int i = 0; i < n; ++i) {
                const float send = upper ? acc[i] : acc[i + n];
                const float keep = upper ? acc[i + n] : acc[i];
                acc[i] = keep + __shfl_xor(send, m);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] += __shfl_xor(acc[i], 1);
        if (live && !(l16 & 1)) {
            const int p = l16 >> 1, ox = ox0 + p;
            if (ox < W)
#pragma unroll
                for (int o = 0; o < 3; ++o)
                    if (o < Cout) y[(((size_t)b * Cout + o) * H + oy) * W + ox] = acc[o] + (bias ? bias[o] : 0.f);
        }
    }
}

// ---------------------------------------------------------------- last conv on the matrix cores
// v_mfma_f32_16x16x32: 16 pixels x 16 output columns (Cout <= 3 of them real) x 32 channels per instruction.
// The A fragment of a lane -- pixel (lane & 15), 8 consecutive channels at 8 * (lane >> 4) -- is exactly one
// 16-byte piece of the NHWC row, so activations go global -> registers -> MFMA with no LDS at all; the taps
// re-read the shifted pixels through L1.  The B fragments (weights, TAPS x Cin/32 of them) are read from LDS
// (36 KiB at Cin = 128; in registers they would cost 144 VGPRs and the occupancy that hides the load latency).  36 MFMAs of 16 cycles per 16 pixels replace ~860
// multi-cycle dot2 instructions per 32.
template <typename T, int R, int KS>
__global__ void __launch_bounds__(256)
conv_last_mfma_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                      float* __restrict__ y, int B, int H, int W, int Cout, const char* __restrict__ zero) {
    constexpr int TAPS = R * R, PAD = R / 2, Cin = KS * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    // B fragments (column n = an output channel, k = 8g .. 8g+7) parked in LDS as [tap][k-step][lane]
    __shared__ u32x4 sbw[TAPS * KS * 64];
    for (int i = threadIdx.x; i < TAPS * KS * 64; i += 256) {
        const int l = i & 63, ks = (i >> 6) % KS, t = (i >> 6) / KS, nn = l & 15, gg = l >> 4;
        u32x4 q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = ks * 32 + gg * 8 + 2 * j;
            const float lo = nn < Cout ? w[((size_t)nn * Cin + c) * TAPS + t] : 0.f;
            const float hi = nn < Cout ? w[((size_t)nn * Cin + c + 1) * TAPS + t] : 0.f;
            q[j] = pack2<T>(lo, hi);
        }
        sbw[i] = q;
    }
    __syncthreads();
    const float bv = (bias && n < Cout) ? bias[n] : 0.f;
    const int tiles_per_row = (W + 15) >> 4;
    const int ntiles = B * H * tiles_per_row;       // < 2^31, checked by the host
    // XCD-contiguous bands of rows (workgroups are dealt round-robin to the 8 XCDs)
    const int nb = gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = blockIdx.x & 7;
    const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
    for (int tile = bid * 4 + wave; tile < ntiles; tile += nb * 4) {
        const int rowid = tile / tiles_per_row, tx = tile - rowid * tiles_per_row;
        const int b = rowid / H, oy = rowid - b * H, ox0 = tx * 16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int iy = oy + r - PAD;
            if ((unsigned)iy >= (unsigned)H) continue;               // wave-uniform: a tile lies in one row
            const char* rowp = (const char*)(x + ((size_t)b * H + iy) * W * Cin) + g * 16;
            u32x4 a[R][KS];
#pragma unroll
            for (int sx = 0; sx < R; ++sx) {
                const int ix = ox0 + n + sx - PAD;
                const char* p = (unsigned)ix < (unsigned)W ? rowp + (size_t)ix * (Cin * 2) : zero;
                const int step = (unsigned)ix < (unsigned)W ? 64 : 0;  // the zero page is 128 bytes: do not walk off it
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) a[sx][ks] = *(const u32x4*)(p + ks * step);
            }
#pragma unroll
            for (int sx = 0; sx < R; ++sx)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = mma16x16<T>(a[sx][ks], sbw[((r * R + sx) * KS + ks) * 64 + lane], acc);
        }
        // D: lane holds column n, pixels 4g .. 4g+3 of the tile
        if (n < Cout) {
            float* yo = y + (((size_t)b * Cout + n) * H + oy) * W + ox0 + 4 * g;
            if ((W & 3) == 0 && ox0 + 4 * g + 3 < W) {
                *(f32x4*)yo = f32x4{acc[0] + bv, acc[1] + bv, acc[2] + bv, acc[3] + bv};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ox0 + 4 * g + i < W) yo[i] = acc[i] + bv;
            }
        }
    }
}

// 3x3 variant that walks DOWN a 16-pixel-wide strip: every input row is loaded once (one centre fragment per
// 32 channels plus the two edge pixels) and kept in registers for the three output rows that use it; the
// horizontally shifted fragments of the left / right taps are made with DPP row shifts (pixel = lane & 15, so
// a one-pixel shift is a one-lane shift inside the 16-lane row; the vacated lane takes the edge pixel).
// 8 loads per 16 output pixels instead of 36, and HBM sees each row once.  Three row slots rotate.
template <typename T, int KS>
__global__ void __launch_bounds__(256, 2)
conv_last_strip_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                       float* __restrict__ y, int B, int H, int W, int Cout, int RC, const char* __restrict__ zero) {
    constexpr int TAPS = 9, Cin = KS * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    __shared__ u32x4 sbw[TAPS * KS * 64];          // B fragments [tap][k-step][lane]: column n, k = 8g .. 8g+7
    for (int i = threadIdx.x; i < TAPS * KS * 64; i += 256) {
        const int l = i & 63, ks = (i >> 6) % KS, t = (i >> 6) / KS, nn = l & 15, gg = l >> 4;
        u32x4 q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = ks * 32 + gg * 8 + 2 * j;
            const float lo = nn < Cout ? w[((size_t)nn * Cin + c) * TAPS + t] : 0.f;
            const float hi = nn < Cout ? w[((size_t)nn * Cin + c + 1) * TAPS + t] : 0.f;
            q[j] = pack2<T>(lo, hi);
        }
        sbw[i] = q;
    }
    __syncthreads();
    const float bv = (bias && n < Cout) ? bias[n] : 0.f;
    const int tiles_per_row = (W + 15) >> 4, chunks = (H + RC - 1) / RC;
    const int nstrips = B * chunks * tiles_per_row;
    const int nb = gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = blockIdx.x & 7;     // XCD-contiguous bands
    const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
    for (int strip = bid * 4 + wave; strip < nstrips; strip += nb * 4) {
        const int bc = strip / tiles_per_row, tx = strip - bc * tiles_per_row;
        const int b = bc / chunks, ch = bc - b * chunks;
        const int y0 = ch * RC, y1 = min(H, y0 + RC), ox0 = tx * 16;
        const int ixc = ox0 + n, ixe = n == 0 ? ox0 - 1 : ox0 + 16;       // centre pixel / edge pixel of this lane
        const bool okc = ixc < W, oke = (unsigned)ixe < (unsigned)W;
        const size_t offc = ((size_t)b * H * W + ixc) * (Cin * 2) + g * 16, offe = ((size_t)b * H * W + ixe) * (Cin * 2) + g * 16;
        u32x4 c[3][KS], e[3][KS];
        auto load_row = [&](auto slot, int iy) {
            constexpr int sl = decltype(slot)::value;
            const bool rowok = (unsigned)iy < (unsigned)H;
            const char* rp = (const char*)x + (size_t)(rowok ? iy : 0) * W * (Cin * 2);
            const char* pc = (rowok && okc) ? rp + offc : zero;
            const char* pe = (rowok && oke) ? rp + offe : zero;
            const int stc = (rowok && okc) ? 64 : 0, ste = (rowok && oke) ? 64 : 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { c[sl][ks] = *(const u32x4*)(pc + ks * stc); e[sl][ks] = *(const u32x4*)(pe + ks * ste); }
        };
        auto row_mma = [&](auto slot, int r, f32x4& acc) {
            constexpr int sl = decltype(slot)::value;
            int opq = 0;                               // keeps the (loop-invariant) weight reads where they are:
            asm volatile("" : "+v"(opq));              // hoisted out of the row loop they would pin 144 registers
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 lft, rgt;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lft[j] = (unsigned)__builtin_amdgcn_update_dpp((int)e[sl][ks][j], (int)c[sl][ks][j], 0x111, 0xf, 0xf, false);  // row_shr:1
                    rgt[j] = (unsigned)__builtin_amdgcn_update_dpp((int)e[sl][ks][j], (int)c[sl][ks][j], 0x101, 0xf, 0xf, false);  // row_shl:1
                }
                const u32x4* bq = sbw + ((r * 3) * KS + ks) * 64 + lane + opq;
                acc = mma16x16<T>(lft, bq[0], acc);
                acc = mma16x16<T>(c[sl][ks], bq[KS * 64], acc);
                acc = mma16x16<T>(rgt, bq[2 * KS * 64], acc);
            }
        };
        auto out_row = [&](int oy, const f32x4& acc) {
            if (n < Cout) {
                float* yo = y + (((size_t)b * Cout + n) * H + oy) * W + ox0 + 4 * g;
                if ((W & 3) == 0 && ox0 + 4 * g + 3 < W) {
                    *(f32x4*)yo = f32x4{acc[0] + bv, acc[1] + bv, acc[2] + bv, acc[3] + bv};
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (ox0 + 4 * g + i < W) yo[i] = acc[i] + bv;
                }
            }
        };
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        // step(top, mid, bot): rows oy-1, oy, oy+1 are resident.  Once the top row's taps are issued its slot is
        // free: row oy+2 is fetched into it, a whole step before it is needed (as the bottom row of step oy+1).
        auto step = [&](auto top, auto mid, auto bot, int oy) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            row_mma(top, 0, acc);
            load_row(top, oy + 2);
            row_mma(mid, 1, acc); row_mma(bot, 2, acc);
            out_row(oy, acc);
        };
        load_row(S0{}, y0 - 1); load_row(S1{}, y0); load_row(S2{}, y0 + 1);
        for (int oy = y0; oy < y1; oy += 3) {
            step(S0{}, S1{}, S2{}, oy);
            if (oy + 1 < y1) step(S1{}, S2{}, S0{}, oy + 1);
            if (oy + 2 < y1) step(S2{}, S0{}, S1{}, oy + 2);
        }
    }
}

template <typename T, int KS>
static int conv_last_strip_launch(const void* x, const float* w, const float* bias, float* y, int b, int h, int wd, int cout,
                                  hipStream_t st) {
    const int rc = h >= 64 ? 32 : (h >= 16 ? 16 : h);
    const long long nstrips = (long long)b * ((h + rc - 1) / rc) * ((wd + 15) / 16);
    ADVS_REQUIRE(nstrips < (1ll << 31) - 4 * 8192 && (long long)b * h * wd * KS * 64 < (1ll << 62), "conv_last: shape out of range");
    const int grid = (int)((nstrips + 3) / 4 < 2048 ? (nstrips + 3) / 4 : 2048);
    conv_last_strip_kernel<T, KS><<<grid, 256, 0, st>>>((const T*)x, w, bias, y, b, h, wd, cout, rc, (const char*)advs_zero_page());
    ADVS_CHECK_LAUNCH("conv_last");
    return ADVS_OK;
}

template <typename T, int R, int KS>
static int conv_last_mfma_launch(const void* x, const float* w, const float* bias, float* y, int b, int h, int wd, int cout,
                                 hipStream_t st) {
    const long long ntiles = (long long)b * h * ((wd + 15) / 16);
    ADVS_REQUIRE(ntiles < (1ll << 31) - 4 * 8192, "conv_last: too many pixels for 32-bit tile indices");
    const int grid = (int)((ntiles + 3) / 4 < 2048 ? (ntiles + 3) / 4 : 2048);     // persistent: 8 workgroups per CU
    conv_last_mfma_kernel<T, R, KS><<<grid, 256, 0, st>>>((const T*)x, w, bias, y, b, h, wd, cout, (const char*)advs_zero_page());
    ADVS_CHECK_LAUNCH("conv_last");
    return ADVS_OK;
}

template <typename T>
static int conv_last16_launch(const void* x, const float* w, const float* bias, float* y, int b, int cin, int h, int wd,
                              int cout, int ksize, hipStream_t st) {
    const size_t lds = (size_t)ksize * ksize * (cin / 8) * 3 * sizeof(u32x4);
    ADVS_REQUIRE(lds <= 65536, "conv_last: cin=%d too large", cin);
    const long long nruns = (long long)b * h * ((wd + 7) / 8);
    ADVS_REQUIRE(nruns < (1ll << 31) - 16 * 16384, "conv_last: too many pixels for 32-bit run indices");
    const int grid = (int)((nruns + 15) / 16 < 16384 ? (nruns + 15) / 16 : 16384);
    if (ksize == 3) conv_last16_kernel<T, 3><<<grid, 256, lds, st>>>((const T*)x, w, bias, y, b, cin, h, wd, cout, (const u32x4*)advs_zero_page());
    else conv_last16_kernel<T, 1><<<grid, 256, lds, st>>>((const T*)x, w, bias, y, b, cin, h, wd, cout, (const u32x4*)advs_zero_page());
    ADVS_CHECK_LAUNCH("conv_last");
    return ADVS_OK;
}

extern "C" int advs_conv_last(const void* x, const float* w, const float* bias, float* y,
                              int b, int cin, int h, int wd, int cout, int ksize, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_conv_last: unknown dtype code %d", dtype);
    ADVS_REQUIRE(x && w && y && b > 0 && h > 0 && wd > 0, "conv_last: bad args");
    ADVS_REQUIRE(cout >= 1 && cout <= 4, "conv_last: cout=%d must be <= 4", cout);
    ADVS_REQUIRE(ksize == 1 || ksize == 3, "conv_last: ksize %d unsupported", ksize);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(cin % vec == 0, "conv_last: cin=%d must be a multiple of %d", cin, vec);
    if (cout <= 3 && dtype != ADVS_F32 && (cin == 128 || cin == 64)) {      // the UNets' widths: matrix-core kernel
        const hipStream_t st = (hipStream_t)stream;
        if (dtype == ADVS_BF16) {
            if (ksize == 3) return cin == 128 ? conv_last_strip_launch<BF16, 4>(x, w, bias, y, b, h, wd, cout, st)
                                              : conv_last_strip_launch<BF16, 2>(x, w, bias, y, b, h, wd, cout, st);
            return cin == 128 ? conv_last_mfma_launch<BF16, 1, 4>(x, w, bias, y, b, h, wd, cout, st)
                              : conv_last_mfma_launch<BF16, 1, 2>(x, w, bias, y, b, h, wd, cout, st);
        }
        if (ksize == 3) return cin == 128 ? conv_last_strip_launch<F16, 4>(x, w, bias, y, b, h, wd, cout, st)
                                          : conv_last_strip_launch<F16, 2>(x, w, bias, y, b, h, wd, cout, st);
        return cin == 128 ? conv_last_mfma_launch<F16, 1, 4>(x, w, bias, y, b, h, wd, cout, st)
                          : conv_last_mfma_launch<F16, 1, 2>(x, w, bias, y, b, h, wd, cout, st);
    }
    if (cout <= 3 && dtype == ADVS_BF16) return conv_last16_launch<BF16>(x, w, bias, y, b, cin, h, wd, cout, ksize, (hipStream_t)stream);
    if (cout <= 3 && dtype == ADVS_F16) return conv_last16_launch<F16>(x, w, bias, y, b, cin, h, wd, cout, ksize, (hipStream_t)stream);
    const size_t lds = (size_t)ksize * ksize * cin * 4 * sizeof(float);
    ADVS_REQUIRE(lds <= 65536, "conv_last: cin=%d too large", cin);
    const long long nruns = (long long)b * h * ((wd + CL_PX - 1) / CL_PX);
    int grid = (int)((nruns + 15) / 16 < 16384 ? (nruns + 15) / 16 : 16384);
    ADVS_SWITCH_T(dtype, conv_last_kernel<T><<<grid, 256, lds, (hipStream_t)stream>>>((const T*)x, w, bias, y, b, cin, h, wd, cout, ksize));
    ADVS_CHECK_LAUNCH("conv_last");
    return ADVS_OK;
}
