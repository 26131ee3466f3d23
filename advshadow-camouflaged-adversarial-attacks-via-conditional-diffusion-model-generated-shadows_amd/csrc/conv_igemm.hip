// Conv2d as an implicit GEMM on the CDNA4 matrix cores.
//
//   y[m][n] = act( sum_k A[m][k] * Wp[n][k] + bias[n] + temb[b(m)][n] + residual[m][n] )
//   m = (b, oy, ox) output pixel, n = output channel, k = (r, s, c) filter tap x input channel.
//
// A is never materialised: every 128-byte K-slab of an A row is one contiguous NHWC channel
// run of one source pixel (or the zero page outside the image), fetched straight into LDS by
// global_load_lds (16 B per lane, per-lane source address = the im2col gather).  The gather
// also performs the channel concat of two sources and the nearest x2 upsample.
//
// Tile BM x BN x 128 B of K per step, NW waves each owning a WM x WN block of 32x32 MFMA
// tiles.  bf16: v_mfma_f32_32x32x16_bf16; f32: v_mfma_f32_32x32x2_f32 (exact f32).
// LDS: double-buffered A and B tiles, rows of 128 B with the 16-byte chunk index XOR-ed by
// (row>>1)&7 so every ds_read_b128 of a fragment is bank-conflict free; since the DMA writes
// LDS linearly the same XOR is applied to the per-lane SOURCE address.
// Epilogue: each wave transposes its accumulators through a private LDS patch so that bias /
// temb / residual are read and y is written as 16-byte vectors along the channel axis
// (whole 128-byte lines), instead of one element per lane.
//
// Roofline: MFMA-bound.  Algorithmic FLOPs per launch = 2*M*N*K.
#include "common.h"

#define SLAB 128                 // bytes of K per row per step

struct ConvKP {
    const char* x1; const char* x2; const char* w;
    const float* bias; const float* temb; const char* res; char* y;
    const char* zero;
    int B, H, W, C1, C2, Cout;
    int R, stride, pad, ups;
    int Ho, Wo, M, K;            // K in elements
    int act, temb_stride;
    int nMt, nNt;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}

template <typename T> struct Mma;
template <> struct Mma<BF16> {
    static constexpr int ESZ = 2;
    // one 16-byte fragment per operand = K of 16 (two lane halves x 8)
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int ESZ = 4;
    // 16 bytes = 4 floats per lane half; float j of both halves forms one K=2 step
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
    }
};

template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64)
conv_igemm_kernel(const ConvKP p) {
    constexpr int ESZ = Mma<T>::ESZ;
    constexpr int VEC = 16 / ESZ;
    constexpr int BKE = SLAB / ESZ;                       // K elements per slab
    constexpr int NWN = BN / WN, NW = (BM / WM) * NWN, NT = NW * 64;
    constexpr int AR = BM * 8 / NT, BR = BN * 8 / NT;     // 16-byte chunks each lane stages per slab
    constexpr int TM = WM / 32, TN = WN / 32;             // MFMA tiles per wave
    constexpr int A_BYTES = BM * SLAB, B_BYTES = BN * SLAB, STAGE = A_BYTES + B_BYTES;
    static_assert(AR * NT == BM * 8 && BR * NT == BN * 8, "tile/wave geometry");
    static_assert(NW * 32 * WN * 4 <= 2 * STAGE, "epilogue patches must fit the staging buffers");
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // [2][A tile | B tile]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware tile order: blocks that share an XCD (id % 8) walk neighbouring tiles,
    // N tiles of one M tile first, so the A rows are re-read from that XCD's L2.
    const int nblk = p.nMt * p.nNt;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = bid / p.nNt, nt = bid - mt * p.nNt;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-lane staging geometry: AR rows of A and BR rows of B per lane, one 16-B chunk each
    const int HoWo = p.Ho * p.Wo;
    const int Cin = p.C1 + p.C2;
    const int HL = p.H << p.ups, WL = p.W << p.ups;
    const int chunk = lane & 7;
    int a_b[AR], a_iy[AR], a_ix[AR], a_csw[AR];
    const char* b_src[BR];
    int b_csw[BR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int row = (wave * AR + j) * 8 + (lane >> 3);
        a_csw[j] = ((chunk ^ ((row >> 1) & 7)) << 4);
        const int m = m0 + row;
        if (m < p.M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_b[j] = b; a_iy[j] = oy * p.stride - p.pad; a_ix[j] = ox * p.stride - p.pad;
        } else {
            a_b[j] = 0; a_iy[j] = -0x40000000; a_ix[j] = 0;      // always out of the image
        }
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int row = (wave * BR + j) * 8 + (lane >> 3);
        b_csw[j] = ((chunk ^ ((row >> 1) & 7)) << 4);
        const int n = n0 + row;
        b_src[j] = (n < p.Cout) ? p.w + ((size_t)n * p.K) * ESZ + b_csw[j] : nullptr;
    }

    // K-slab cursor (uniform): tap (r, s) and channel offset c0 inside the concatenated input
    int kr = 0, ks = 0, c0 = 0;
    const int nk = p.K / BKE;

    auto stage = [&](int buf, int kt) {
        char* la = smem + buf * STAGE;
        char* lb = la + A_BYTES;
        const bool first = c0 < p.C1;
        const char* src = first ? p.x1 : p.x2;
        const int cs = first ? p.C1 : p.C2;
        const int cc = first ? c0 : c0 - p.C1;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const int iy = a_iy[j] + kr, ix = a_ix[j] + ks;
            const bool ok = (unsigned)iy < (unsigned)HL && (unsigned)ix < (unsigned)WL;
            const size_t pix = ((size_t)a_b[j] * p.H + (iy >> p.ups)) * p.W + (ix >> p.ups);
            const char* s = ok ? src + (pix * cs + cc) * ESZ + a_csw[j] : p.zero + a_csw[j];
            glds16(s, la + (wave * AR + j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const char* s = b_src[j] ? b_src[j] + (size_t)kt * SLAB : p.zero + b_csw[j];
            glds16(s, lb + (wave * BR + j) * 1024);
        }
        c0 += BKE;
        if (c0 == Cin) { c0 = 0; if (++ks == p.R) { ks = 0; ++kr; } }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wr = wave / NWN, wc = wave - wr * NWN;
    const int l31 = lane & 31, lh = lane >> 5;
    // fragment read offsets (bytes inside a tile) for k-step s: row*128 + ((2s+lh) ^ sw(row))*16
    int a_off[TM], a_sw[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ra = wr * WM + i * 32 + l31;
        a_off[i] = ra * SLAB; a_sw[i] = (ra >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int rb = wc * WN + j * 32 + l31;
        b_off[j] = rb * SLAB; b_sw[j] = (rb >> 1) & 7;
    }

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            stage(buf ^ 1, kt + 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR + BR) : "memory");   // slab kt landed, kt+1 in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const char* la = smem + buf * STAGE;
        const char* lb = la + A_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const u32x4*)(la + a_off[i] + (((2 * s + lh) ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *(const u32x4*)(lb + b_off[j] + (((2 * s + lh) ^ b_sw[j]) << 4));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's LDS reads retired ...
        __builtin_amdgcn_s_barrier();                            // ... before anyone restages buf
    }

    // ---- epilogue.  C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // Per 32-row strip the wave parks its 32 x WN accumulators in a private LDS patch
    // (f32, row-major) and re-reads it one 16-byte output vector per lane.
    constexpr int LPR = WN / VEC;                 // lanes per patch row
    constexpr int RPI = 64 / LPR;                 // rows per wave-instruction
    float* patch = (float*)smem + wave * (32 * WN);
    T* y = (T*)p.y;
    const T* res = (const T*)p.res;
    const int prow = lane / LPR, pcv = lane - prow * LPR;
    const int n = n0 + wc * WN + pcv * VEC;
    const bool n_ok = n < p.Cout;                 // Cout is a multiple of VEC
    float bias[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) bias[e] = (p.bias && n_ok) ? p.bias[n + e] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * WN + j * 32 + l31] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int row = it * RPI + prow;
            const int m = m0 + wr * WM + i * 32 + row;
            float v[VEC];
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                const f32x4 t = *(const f32x4*)(patch + row * WN + pcv * VEC + e);
                v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
            }
            if (m < p.M && n_ok) {
                const size_t o = (size_t)m * p.Cout + n;
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] += bias[e];
                if (p.temb) {
                    const float* tp = p.temb + (size_t)(m / HoWo) * p.temb_stride + n;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += tp[e];
                }
                if (res) {
                    float rv[VEC];
                    unpack16<T>(*(const u32x4*)(res + o), rv);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += rv[e];
                }
                if (p.act != ADVS_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] = apply_act(v[e], p.act);
                }
                *(u32x4*)(y + o) = pack16<T>(v);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <typename T, int BM, int BN, int WM, int WN>
static int conv_launch(ConvKP& p, hipStream_t st) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int lds = 2 * (BM + BN) * SLAB;
    static bool attr_set = false;
    if (!attr_set) {
        ADVS_HIP(hipFuncSetAttribute((const void*)conv_igemm_kernel<T, BM, BN, WM, WN>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    p.nMt = cdiv(p.M, BM); p.nNt = cdiv(p.Cout, BN);
    conv_igemm_kernel<T, BM, BN, WM, WN><<<p.nMt * p.nNt, NT, lds, st>>>(p);
    ADVS_CHECK_LAUNCH("conv_igemm");
    return ADVS_OK;
}

template <typename T>
static int conv_dispatch(ConvKP& p, int tile, hipStream_t st) {
    if (tile == 0) {
        // measured on MI355X (tools/tune_conv.py, round 1): the 256x256 tile wins (~1.0-1.06 vs
        // ~0.85 PFLOP/s) whenever Cout fills it and M still yields >= 1.5 blocks per CU; the
        // 128x128 tile at two blocks per CU wins everywhere else, including Cout = 128.
        const long long blocks256 = (long long)cdiv(p.M, 256) * cdiv(p.Cout, 256);
        tile = (p.Cout % 256 == 0 && blocks256 >= 384) ? 4 : 1;
    }
    switch (tile) {
        case 1: return conv_launch<T, 128, 128, 64, 64>(p, st);
        case 2: return conv_launch<T, 256, 128, 64, 64>(p, st);
        case 3: return conv_launch<T, 256, 128, 128, 64>(p, st);
        case 4: return conv_launch<T, 256, 256, 128, 64>(p, st);
        default: ADVS_FAIL(ADVS_ERR_ARG, "conv2d: unknown tile id %d", tile);
    }
}

static int g_tile_override = 0;
extern "C" int advs_conv_set_tile(int tile) { g_tile_override = tile; return ADVS_OK; }

extern "C" int advs_conv2d(const advs_conv_args* a, void* stream) {
    ADVS_REQUIRE(a && a->x1 && a->w && a->y, "conv2d: null pointer");
    ADVS_REQUIRE(a->b > 0 && a->h > 0 && a->w_ > 0 && a->c1 > 0 && a->c2 >= 0 && a->cout > 0, "conv2d: bad shape");
    ADVS_REQUIRE(a->ksize == 1 || a->ksize == 3, "conv2d: ksize %d unsupported", a->ksize);
    ADVS_REQUIRE(a->stride == 1 || a->stride == 2, "conv2d: stride %d unsupported", a->stride);
    ADVS_REQUIRE(a->pad >= 0 && a->pad <= 1, "conv2d: pad %d unsupported", a->pad);
    ADVS_REQUIRE((a->c2 == 0) == (a->x2 == nullptr), "conv2d: x2/c2 mismatch");
    const int esz = a->dtype == ADVS_BF16 ? 2 : 4;
    const int bke = SLAB / esz;
    ADVS_REQUIRE(a->c1 % bke == 0 && a->c2 % bke == 0, "conv2d: channels (%d,%d) must be multiples of %d",
                 a->c1, a->c2, bke);
    ADVS_REQUIRE(a->cout % (16 / esz) == 0, "conv2d: cout=%d must be a multiple of %d", a->cout, 16 / esz);
    const void* zero = advs_zero_page();
    ADVS_REQUIRE(zero, "conv2d: advs_init() has not been called on this device");
    ConvKP p;
    p.x1 = (const char*)a->x1; p.x2 = (const char*)a->x2; p.w = (const char*)a->w;
    p.bias = a->bias; p.temb = a->temb; p.res = (const char*)a->residual; p.y = (char*)a->y;
    p.zero = (const char*)zero;
    p.B = a->b; p.H = a->h; p.W = a->w_; p.C1 = a->c1; p.C2 = a->c2; p.Cout = a->cout;
    p.R = a->ksize; p.stride = a->stride; p.pad = a->pad; p.ups = a->upsample ? 1 : 0;
    const int HL = a->h << p.ups, WL = a->w_ << p.ups;
    p.Ho = (HL + 2 * a->pad - a->ksize) / a->stride + 1;
    p.Wo = (WL + 2 * a->pad - a->ksize) / a->stride + 1;
    const long long M = (long long)a->b * p.Ho * p.Wo;
    ADVS_REQUIRE(M > 0 && M < (1ll << 31) - 256, "conv2d: M=%lld out of range", M);
    p.M = (int)M;
    p.K = a->ksize * a->ksize * (a->c1 + a->c2);
    p.act = a->act; p.temb_stride = a->temb_stride > 0 ? a->temb_stride : a->cout;
    const int tile = g_tile_override ? g_tile_override : a->tile;
    if (a->dtype == ADVS_BF16) return conv_dispatch<BF16>(p, tile, (hipStream_t)stream);
    return conv_dispatch<float>(p, tile, (hipStream_t)stream);
}
