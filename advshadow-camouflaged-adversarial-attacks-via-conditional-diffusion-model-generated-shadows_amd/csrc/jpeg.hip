// The lossy part of the JPEG file hop between save_images (utils/utils.py:51-91, Pillow defaults: baseline,
// quality 75, 4:2:0) and preprocess_image / load_image (ASR_fast.py:90-92, PSNR_SSIM_fast.py:21-24), on the
// device: the victim sees these pixels, not the sampler's.  Entropy coding is lossless and is skipped.
// Integer arithmetic exactly as the JPEG library behind Pillow does it (published algorithm of
// libjpeg/libjpeg-turbo; oracle/jpeg.py restates it and is pinned bit for bit against Pillow):
//   kernel 1, one wave per 16x16 MCU: RGB -> YCbCr (16-bit fixed point), 2x2 chroma box filter with the
//             alternating 1,2 bias, then per 8x8 block (4 Y + Cb + Cr) level shift -> forward DCT "islow"
//             -> quantise (round half away from zero, divisor 8q) -> dequantise -> inverse DCT "islow" ->
//             range limit; planes of reconstructed samples go to scratch (1.5 B / pixel);
//   kernel 2, per 4 output pixels: triangle ("fancy") 2x2 chroma upsampling across MCU borders, YCbCr -> RGB.
// HBM-bound byte work: 3 + 1.5 + 1.5 (+ halo re-reads from cache) + 3 bytes per pixel.
#include "common.h"

#define JC_BITS 13
#define JP1_BITS 2
#define JF_0_298631336 2446
#define JF_0_390180644 3196
#define JF_0_541196100 4433
#define JF_0_765366865 6270
#define JF_0_899976223 7373
#define JF_1_175875602 9633
#define JF_1_501321110 12299
#define JF_1_847759065 15137
#define JF_1_961570560 16069
#define JF_2_053119869 16819
#define JF_2_562915447 20995
#define JF_3_072711026 25172

struct JpegQ { unsigned short q[2][64]; };      // luminance, chrominance (natural order)

__device__ __forceinline__ int jdescale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 8-point pass of jfdctint.c; FIRST: rows (outputs scaled up by 4), else columns
template <bool FIRST>
__device__ __forceinline__ void jfdct8(int* d) {
    const int t0 = d[0] + d[7], t7 = d[0] - d[7], t1 = d[1] + d[6], t6 = d[1] - d[6];
    const int t2 = d[2] + d[5], t5 = d[2] - d[5], t3 = d[3] + d[4], t4 = d[3] - d[4];
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    constexpr int n = FIRST ? JC_BITS - JP1_BITS : JC_BITS + JP1_BITS;
    d[0] = FIRST ? (t10 + t11) << JP1_BITS : jdescale(t10 + t11, JP1_BITS);
    d[4] = FIRST ? (t10 - t11) << JP1_BITS : jdescale(t10 - t11, JP1_BITS);
    int z1 = (t12 + t13) * JF_0_541196100;
    d[2] = jdescale(z1 + t13 * JF_0_765366865, n);
    d[6] = jdescale(z1 - t12 * JF_1_847759065, n);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * JF_1_175875602;
    const int a4 = t4 * JF_0_298631336, a5 = t5 * JF_2_053119869, a6 = t6 * JF_3_072711026, a7 = t7 * JF_1_501321110;
    z1 *= -JF_0_899976223; z2 *= -JF_2_562915447;
    z3 = z3 * -JF_1_961570560 + z5; z4 = z4 * -JF_0_390180644 + z5;
    d[7] = jdescale(a4 + z1 + z3, n); d[5] = jdescale(a5 + z2 + z4, n);
    d[3] = jdescale(a6 + z2 + z3, n); d[1] = jdescale(a7 + z1 + z4, n);
}

// one 8-point pass of jidctint.c; FIRST: columns, else rows (+3 bits: the forward transform's factor 8)
template <bool FIRST>
__device__ __forceinline__ void jidct8(int* v) {
    int z1 = (v[2] + v[6]) * JF_0_541196100;
    int t2 = z1 - v[6] * JF_1_847759065;
    int t3 = z1 + v[2] * JF_0_765366865;
    int t0 = (v[0] + v[4]) << JC_BITS, t1 = (v[0] - v[4]) << JC_BITS;
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    t0 = v[7]; t1 = v[5]; t2 = v[3]; t3 = v[1];
    z1 = t0 + t3;
    int z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
    const int z5 = (z3 + z4) * JF_1_175875602;
    t0 *= JF_0_298631336; t1 *= JF_2_053119869; t2 *= JF_3_072711026; t3 *= JF_1_501321110;
    z1 *= -JF_0_899976223; z2 *= -JF_2_562915447;
    z3 = z3 * -JF_1_961570560 + z5; z4 = z4 * -JF_0_390180644 + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    constexpr int n = FIRST ? JC_BITS - JP1_BITS : JC_BITS + JP1_BITS + 3;
    v[0] = jdescale(t10 + t3, n); v[7] = jdescale(t10 - t3, n);
    v[1] = jdescale(t11 + t2, n); v[6] = jdescale(t11 - t2, n);
    v[2] = jdescale(t12 + t1, n); v[5] = jdescale(t12 - t1, n);
    v[3] = jdescale(t13 + t0, n); v[4] = jdescale(t13 - t0, n);
}

#define JFIX(x) ((int)((x) * 65536.0 + 0.5))

// ---- kernel 1: one 64-lane workgroup per MCU
__global__ void __launch_bounds__(64)
jpeg_codec_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ yp, unsigned char* __restrict__ cbp,
                  unsigned char* __restrict__ crp, int H, int W, const JpegQ q) {
    __shared__ int blk[6][64];                   // Y00 Y01 Y10 Y11 Cb Cr, row-major 8x8
    __shared__ unsigned char cfull[2][16][16];   // full-resolution Cb, Cr of the MCU
    const int lane = threadIdx.x;
    const int mcux = W >> 4, mcuy = H >> 4;
    const int m = blockIdx.x;
    const int img = m / (mcux * mcuy), rem = m - img * mcux * mcuy;
    const int my = rem / mcux, mx = rem - my * mcux;
    const unsigned char* s = src + ((size_t)img * H + my * 16) * W * 3 + (size_t)mx * 16 * 3;
    // colour conversion: lane -> 4 pixels of the 16x16 MCU (jccolor.c)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = k * 64 + lane, py = p >> 4, px = p & 15;
        const unsigned char* e = s + ((size_t)py * W + px) * 3;
        const int r = e[0], g = e[1], b = e[2];
        const int yv = (JFIX(0.29900) * r + JFIX(0.58700) * g + JFIX(0.11400) * b + 32768) >> 16;
        const int cb = (-JFIX(0.16874) * r - JFIX(0.33126) * g + JFIX(0.50000) * b + (128 << 16) + 32767) >> 16;
        const int cr = (JFIX(0.50000) * r - JFIX(0.41869) * g - JFIX(0.08131) * b + (128 << 16) + 32767) >> 16;
        blk[(py >> 3) * 2 + (px >> 3)][(py & 7) * 8 + (px & 7)] = yv - 128;
        cfull[0][py][px] = (unsigned char)cb; cfull[1][py][px] = (unsigned char)cr;
    }
    __syncthreads();
    {   // 2x2 box filter, bias 1,2,1,2 along the row (jcsample.c h2v2_downsample)
        const int oy = lane >> 3, ox = lane & 7, bias = 1 + (ox & 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int sum = cfull[c][2 * oy][2 * ox] + cfull[c][2 * oy][2 * ox + 1] + cfull[c][2 * oy + 1][2 * ox] +
                            cfull[c][2 * oy + 1][2 * ox + 1];
            blk[4 + c][lane] = ((sum + bias) >> 2) - 128;
        }
    }
    __syncthreads();
    const int b8 = lane >> 3, l8 = lane & 7;     // lanes 0..47: block b8, line l8 of a 1-D pass
    int d[8];
    if (b8 < 6) {                                // forward rows
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = blk[b8][l8 * 8 + i];
        jfdct8<true>(d);
#pragma unroll
        for (int i = 0; i < 8; ++i) blk[b8][l8 * 8 + i] = d[i];
    }
    __syncthreads();
    if (b8 < 6) {                                // forward columns, quantise, dequantise, inverse columns
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = blk[b8][i * 8 + l8];
        jfdct8<false>(d);
        const unsigned short* qt = q.q[b8 < 4 ? 0 : 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int qv = qt[i * 8 + l8], dv = qv << 3;
            const int a = ((d[i] < 0 ? -d[i] : d[i]) + (dv >> 1)) / dv;     // jcdctmgr.c: round half away from zero
            d[i] = (d[i] < 0 ? -a : a) * qv;
        }
        jidct8<true>(d);
#pragma unroll
        for (int i = 0; i < 8; ++i) blk[b8][i * 8 + l8] = d[i];
    }
    __syncthreads();
    if (b8 < 6) {                                // inverse rows, range limit, store
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = blk[b8][l8 * 8 + i];
        jidct8<false>(d);
        unsigned o[2] = {0u, 0u};
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int v = d[i] + 128; o[i >> 2] |= (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v)) << (8 * (i & 3)); }
        unsigned char* dst;
        if (b8 < 4) dst = yp + ((size_t)img * H + my * 16 + (b8 >> 1) * 8 + l8) * W + mx * 16 + (b8 & 1) * 8;
        else dst = (b8 == 4 ? cbp : crp) + ((size_t)img * (H >> 1) + my * 8 + l8) * (W >> 1) + mx * 8;
        *(uint2*)dst = make_uint2(o[0], o[1]);
    }
}

// ---- kernel 2: thread = 4 consecutive output pixels of one row
__global__ void __launch_bounds__(256)
jpeg_merge_kernel(const unsigned char* __restrict__ yp, const unsigned char* __restrict__ cbp,
                  const unsigned char* __restrict__ crp, unsigned char* __restrict__ dst, int N, int H, int W) {
    const int qw = W >> 2;
    const long long total = (long long)N * H * qw;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xq = (int)(i % qw);
        const long long r = i / qw;
        const int y = (int)(r % H), img = (int)(r / H);
        const int h2 = H >> 1, w2 = W >> 1;
        const int cr_ = y >> 1, nb = (y & 1) ? (cr_ + 1 < h2 ? cr_ + 1 : cr_) : (cr_ > 0 ? cr_ - 1 : 0);
        const int c0 = xq * 2;                   // chroma columns c0, c0+1 (+ neighbours c0-1, c0+2)
        const unsigned char* rows[2][2] = {{cbp + ((size_t)img * h2 + cr_) * w2, cbp + ((size_t)img * h2 + nb) * w2},
                                           {crp + ((size_t)img * h2 + cr_) * w2, crp + ((size_t)img * h2 + nb) * w2}};
        int up[2][4];                            // upsampled Cb, Cr of the 4 pixels (jdsample.c h2v2_fancy_upsample)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            int cs[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int col = c0 - 1 + k;
                col = col < 0 ? 0 : (col >= w2 ? w2 - 1 : col);
                cs[k] = rows[c][0][col] * 3 + rows[c][1][col];
            }
            up[c][0] = (cs[1] * 3 + cs[0] + 8) >> 4; up[c][1] = (cs[1] * 3 + cs[2] + 7) >> 4;
            up[c][2] = (cs[2] * 3 + cs[1] + 8) >> 4; up[c][3] = (cs[2] * 3 + cs[3] + 7) >> 4;
        }
        const unsigned yy = *(const unsigned*)(yp + ((size_t)img * H + y) * W + xq * 4);
        unsigned ow[3] = {0u, 0u, 0u};           // 12 output bytes, little endian
#pragma unroll
        for (int k = 0; k < 4; ++k) {            // jdcolor.c ycc_rgb_convert
            const int Y = (yy >> (8 * k)) & 255, cb = up[0][k] - 128, cr = up[1][k] - 128;
            const int R = Y + ((JFIX(1.40200) * cr + 32768) >> 16);
            const int G = Y + ((-JFIX(0.34414) * cb + 32768 - JFIX(0.71414) * cr) >> 16);
            const int B = Y + ((JFIX(1.77200) * cb + 32768) >> 16);
            const unsigned rgb[3] = {(unsigned)(R < 0 ? 0 : (R > 255 ? 255 : R)), (unsigned)(G < 0 ? 0 : (G > 255 ? 255 : G)),
                                     (unsigned)(B < 0 ? 0 : (B > 255 ? 255 : B))};
#pragma unroll
            for (int c = 0; c < 3; ++c) { const int byte = 3 * k + c; ow[byte >> 2] |= rgb[c] << (8 * (byte & 3)); }
        }
        unsigned* d = (unsigned*)(dst + (((size_t)img * H + y) * W + xq * 4) * 3);
        d[0] = ow[0]; d[1] = ow[1]; d[2] = ow[2];
    }
}

extern "C" size_t advs_jpeg_scratch_bytes(int n, int h, int w) {
    return (size_t)n * h * w * 3 / 2;
}

extern "C" int advs_jpeg_roundtrip_u8(const unsigned char* src, unsigned char* dst, void* scratch, int n, int h, int w,
                                      int quality, void* stream) {
    ADVS_REQUIRE(src && dst && scratch && n > 0 && h > 0 && w > 0, "jpeg_roundtrip: bad args");
    ADVS_REQUIRE(h % 16 == 0 && w % 16 == 0, "jpeg_roundtrip: %dx%d is not a whole number of 16x16 MCUs", h, w);
    ADVS_REQUIRE(quality >= 1 && quality <= 100, "jpeg_roundtrip: quality %d out of 1..100", quality);
    ADVS_REQUIRE((long long)n * (h / 16) * (w / 16) < (1ll << 31), "jpeg_roundtrip: too many MCUs");
    // Annex K tables scaled as jcparam.c does (jpeg_quality_scaling, force_baseline)
    static const unsigned char lum[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                          14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                          49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
    static const unsigned char chr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                                          47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                          99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    JpegQ q;
    for (int i = 0; i < 64; ++i) {
        int a = (lum[i] * scale + 50) / 100, b = (chr[i] * scale + 50) / 100;
        q.q[0][i] = (unsigned short)(a < 1 ? 1 : (a > 255 ? 255 : a));
        q.q[1][i] = (unsigned short)(b < 1 ? 1 : (b > 255 ? 255 : b));
    }
    unsigned char* yp = (unsigned char*)scratch;
    unsigned char* cbp = yp + (size_t)n * h * w;
    unsigned char* crp = cbp + (size_t)n * h * w / 4;
    const int nmcu = n * (h / 16) * (w / 16);
    jpeg_codec_kernel<<<nmcu, 64, 0, (hipStream_t)stream>>>(src, yp, cbp, crp, h, w, q);
    ADVS_CHECK_LAUNCH("jpeg_codec");
    const long long quads = (long long)n * h * (w / 4);
    const int grid = (int)((quads + 255) / 256 < 16384 ? (quads + 255) / 256 : 16384);
    jpeg_merge_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(yp, cbp, crp, dst, n, h, w);
    ADVS_CHECK_LAUNCH("jpeg_merge");
    return ADVS_OK;
}
