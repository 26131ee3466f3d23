// Conv2d as an implicit GEMM on the CDNA4 matrix cores.
//
//   y[m][n] = act( sum_k A[m][k] * Wp[n][k] + bias[n] + temb[b(m)][n] + residual[m][n] )
//   m = (b, oy, ox) output pixel, n = output channel, k = (r, s, c) filter tap x input channel.
//
// A is never materialised: every 128-byte K-slab of an A row is one contiguous NHWC channel
// run of one source pixel, fetched straight into LDS by `buffer_load_dwordx4 ... lds` (16 B per
// lane; the per-lane byte offset IS the im2col gather, and rows outside the image carry an
// out-of-range offset so the buffer bounds check writes zeros).  The gather also performs the
// channel concat of two sources and the nearest x2 upsample.  Offsets are recomputed only when
// the filter tap or the concat source changes; within a tap a slab is one scalar-offset bump.
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
// An optional EXTRA operand (again up to two concatenated sources, [B][Ho][Wo][E]) is appended to K as
// a 1x1 stride-1 tap: y = conv_RxR(x) + conv_1x1(e) in ONE launch -- the residual block's
// `conv2(h) + shortcut(x)` (diff_model.py:102-103) without writing and re-reading the shortcut.
//
// Optionally the epilogue also emits, per WM-row block and channel, the sum and sum of squares of the
// stored values: the GroupNorm that consumes y then needs no statistics pass over the tensor.
//
// Roofline: MFMA-bound.  Algorithmic FLOPs per launch = 2*M*N*K.
#include "conv_common.h"
#include <stdlib.h>

template <typename T, int BM, int BN, int WM, int WN, int NSTAGE, bool ILV, bool MASK = false>
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
    static_assert(NW * 32 * WN * 4 <= NSTAGE * STAGE, "epilogue patches must fit the staging buffers");
    static_assert(NSTAGE == 2 || NSTAGE == 3, "2- or 3-deep LDS ring");
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // [NSTAGE][A tile | B tile]

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
    int a_b[AR], a_iy[AR], a_ix[AR], a_m[AR];
    unsigned a_csw[AR], a_voff[AR], b_voff[BR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int row = (wave * AR + j) * 8 + (lane >> 3);
        a_csw[j] = ((chunk ^ ((row >> 1) & 7)) << 4);
        a_voff[j] = OOB_OFFSET;
        const int m = m0 + row;
        if (m < p.M) {
            const int b = (int)p.dHoWo.div((unsigned)m), rem = m - b * HoWo;
            const int oy = (int)p.dWo.div((unsigned)rem), ox = rem - oy * p.Wo;
            a_b[j] = b; a_iy[j] = oy * p.stride - p.pad; a_ix[j] = ox * p.stride - p.pad; a_m[j] = m;
        } else {
            a_b[j] = 0; a_iy[j] = -0x40000000; a_ix[j] = 0; a_m[j] = -1;      // always out of the image
        }
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int row = (wave * BR + j) * 8 + (lane >> 3);
        const int n = n0 + row;
        b_voff[j] = (n < p.Cout) ? (unsigned)n * (unsigned)p.K * ESZ + ((chunk ^ ((row >> 1) & 7)) << 4) : OOB_OFFSET;
    }
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.x1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x1), 0, p.x2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rse1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.e1 ? p.e1 : p.x1), 0, p.e1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rse2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.e2 ? p.e2 : p.x1), 0, p.e2_bytes, 0x00020000);

    // K-slab cursor (uniform): tap (r, s) and channel offset c0 inside the concatenated input
    int kr = 0, ks = 0, c0 = 0;
    const int nk = p.K / BKE;

    // Staging of one K-slab, split so the main loop can spread the DMA issue over its k-steps:
    // stage_begin (uniform bookkeeping + per-row offsets on a new (tap, source) segment), then AR+BR
    // loads issued by stage_part(q) for q = 0..3, then stage_end (advance the cursor).
    int st_sel = 0;                                       // 0: x1, 1: x2, 2: e1, 3: e2
    unsigned st_soff = 0, st_woff = 0;
    char* st_la = smem;
    auto stage_begin = [&](int buf, int kt) {
        st_la = smem + buf * STAGE;
        if (kr < p.R) {
            const bool first = c0 < p.C1;
            st_sel = first ? 0 : 1;
            if (c0 == 0 || c0 == p.C1) {
                // new (tap, source) segment: per-row byte offset of the source pixel, or out of range
                const unsigned cs = (unsigned)(first ? p.LD1 : p.LD2) * ESZ;
#pragma unroll
                for (int j = 0; j < AR; ++j) {
                    const int iy = a_iy[j] + kr, ix = a_ix[j] + ks;
                    const bool ok = (unsigned)iy < (unsigned)HL && (unsigned)ix < (unsigned)WL;
                    const unsigned pix = (unsigned)((a_b[j] * p.H + (iy >> p.ups)) * p.W + (ix >> p.ups));
                    a_voff[j] = ok ? pix * cs + a_csw[j] : OOB_OFFSET;
                }
            }
            st_soff = (unsigned)(first ? c0 : c0 - p.C1) * ESZ;
        } else {
            // extra operand: the output pixel itself, channels [0, E1) of e1 then [0, E2) of e2
            const bool first = c0 < p.E1;
            st_sel = first ? 2 : 3;
            if (c0 == 0 || c0 == p.E1) {
                const unsigned cs = (unsigned)(first ? p.E1 : p.E2) * ESZ;
#pragma unroll
                for (int j = 0; j < AR; ++j) a_voff[j] = a_m[j] >= 0 ? (unsigned)a_m[j] * cs + a_csw[j] : OOB_OFFSET;
            }
            st_soff = (unsigned)(first ? c0 : c0 - p.E1) * ESZ;
        }
        st_woff = (unsigned)kt * SLAB;
    };
    auto stage_part = [&](int q) {                        // q-th quarter of the slab's loads
        char* lb = st_la + A_BYTES;
#pragma unroll
        for (int j = 0; j < AR; ++j)
            if (j * 4 / AR == q) {
                char* dst = st_la + (wave * AR + j) * 1024;
                if (st_sel == 0) blds16(rs1, a_voff[j], st_soff, dst);
                else if (st_sel == 1) blds16(rs2, a_voff[j], st_soff, dst);
                else if (st_sel == 2) blds16(rse1, a_voff[j], st_soff, dst);
                else blds16(rse2, a_voff[j], st_soff, dst);
            }
#pragma unroll
        for (int j = 0; j < BR; ++j)
            if (j * 4 / BR == q) blds16(rsw, b_voff[j], st_woff, lb + (wave * BR + j) * 1024);
    };
    auto stage_end = [&]() {
        c0 += BKE;
        if (kr < p.R && c0 == Cin) { c0 = 0; if (++ks == p.R) { ks = 0; ++kr; } }
    };
    auto stage = [&](int buf, int kt) {
        stage_begin(buf, kt);
#pragma unroll
        for (int q = 0; q < 4; ++q) stage_part(q);
        stage_end();
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

    if constexpr (NSTAGE == 2 && !ILV) {
        // two buffers, two barriers per slab: slab kt+1 is in flight while slab kt is consumed
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
    } else if constexpr (NSTAGE == 2) {
        // two buffers, two barriers per slab.  The loads of slab kt+1 are issued a quarter at a time
        // BETWEEN the MFMA groups of slab kt (their buffer was released by the barrier that ended
        // iteration kt-1), and the fragments of k-step s+1 are read while step s multiplies, so the
        // wave's DMA issue and LDS latency hide behind its own matrix work.
        stage(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // slab kt (issued during kt-1) landed
            __builtin_amdgcn_s_barrier();
            const char* la = smem + buf * STAGE;
            const char* lb = la + A_BYTES;
            const bool more = kt + 1 < nk;
            if (more) stage_begin(buf ^ 1, kt + 1);
            u32x4 af[2][TM], bf[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[0][i] = *(const u32x4*)(la + a_off[i] + (((0 + lh) ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[0][j] = *(const u32x4*)(lb + b_off[j] + (((0 + lh) ^ b_sw[j]) << 4));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int cur = s & 1, nxt = cur ^ 1;
                if (s < 3) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        af[nxt][i] = *(const u32x4*)(la + a_off[i] + (((2 * (s + 1) + lh) ^ a_sw[i]) << 4));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bf[nxt][j] = *(const u32x4*)(lb + b_off[j] + (((2 * (s + 1) + lh) ^ b_sw[j]) << 4));
                }
                if (more) stage_part(s);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(af[cur][i], bf[cur][j], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
            }
            if (more) stage_end();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's LDS reads retired ...
            __builtin_amdgcn_s_barrier();                            // ... before anyone restages buf
        }
    } else {
        // three buffers, ONE barrier per slab, two slabs in flight: passing the barrier of iteration
        // kt proves every wave's share of slab kt has landed (each waited its own vmcnt first) and
        // every wave has finished reading slab kt-1, whose buffer is restaged right after.
        stage(0, 0);
        if (nk > 1) stage(1, 1);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR + BR) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) stage(buf == 0 ? 2 : buf - 1, kt + 2);   // (kt+2) % 3 == (kt-1) % 3
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
            buf = buf == 2 ? 0 : buf + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // all reads done before the patches reuse LDS
    }

    // ---- epilogue (conv_common.h)
    const int mw = m0 + wr * WM;                               // first row of this wave
    const bool wave_live = mw < p.M;
    const int temb_b = (p.temb && HoWo % WM == 0 && wave_live) ? (int)p.dHoWo.div((unsigned)mw) : -1;
    conv_epilogue<T, WN, TM, TN, MASK>(p, acc, (float*)smem + wave * (32 * WN), lane, n0 + wc * WN,
                                 [&](int lr) { const int m = mw + lr; return m < p.M ? m : -1; },
                                 temb_b, wave_live ? mw / WM : -1);
}

template <typename T, int BM, int BN, int WM, int WN, int NSTAGE = 2, bool ILV = false, bool MASK = false>
static int conv_launch(ConvKP& p, hipStream_t st) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int lds = NSTAGE * (BM + BN) * SLAB;
    static bool attr_set = false;
    if (!attr_set) {
        ADVS_HIP(hipFuncSetAttribute((const void*)conv_igemm_kernel<T, BM, BN, WM, WN, NSTAGE, ILV, MASK>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    p.nMt = cdiv(p.M, BM); p.nNt = cdiv(p.Cout, BN);
    conv_igemm_kernel<T, BM, BN, WM, WN, NSTAGE, ILV, MASK><<<p.nMt * p.nNt, NT, lds, st>>>(p);
    ADVS_CHECK_LAUNCH("conv_igemm");
    return ADVS_OK;
}

static int pick_tile(long long m_img, int cout);
bool conv_halo_eligible(const ConvKP& p);
int conv_halo_dispatch(ConvKP& p, int dtype, hipStream_t st);
int conv_halo_subpixel_dispatch(ConvKP& p, int dtype, hipStream_t st);
bool conv_halo_extra_eligible(const ConvKP& p);
int conv_halo_extra_dispatch(ConvKP& p, int dtype, hipStream_t st);
bool conv_halo2_eligible(const ConvKP& p, int dtype, int tile, int kind);
int conv_halo2_dispatch(ConvKP& p, int dtype, int tile, int kind, hipStream_t st);

// A/B knobs read from the environment exist only in diagnostic builds (make DIAG=1, used by tools/); the shipped library's
// dispatch depends on the call's arguments alone.
#ifdef ADVS_DIAG
#define ADVS_DIAG_ENV(name) (getenv(name) != nullptr)
#define ADVS_DIAG_ENV_LL(name, dflt) (getenv(name) ? atoll(getenv(name)) : (long long)(dflt))
#else
#define ADVS_DIAG_ENV(name) false
#define ADVS_DIAG_ENV_LL(name, dflt) ((long long)(dflt))
#endif

template <typename T>
static int conv_dispatch(ConvKP& p, int tile, hipStream_t st) {
    if (tile == 0) tile = pick_tile((long long)p.Ho * p.Wo, p.Cout);
    if (p.mask) {                       // ReLU-backward epilogue: its own instantiations, the others never read p.mask
        if (tile == 4) return conv_launch<T, 256, 256, 128, 64, 2, true, true>(p, st);
        if (tile == 15) return conv_launch<T, 64, 128, 32, 64, 2, false, true>(p, st);
        if (tile == 16) return conv_launch<T, 64, 64, 32, 32, 2, false, true>(p, st);
        return conv_launch<T, 128, 128, 64, 64, 2, false, true>(p, st);
    }
    switch (tile) {
        // ILV (DMA issue spread between the MFMA groups) pays when both waves of a SIMD belong to one
        // workgroup and so run in lockstep (8-wave tiles); with two 4-wave workgroups per CU the plain
        // loop is faster because the workgroups already interleave each other (tools/tune_conv.py).
        case 1: return conv_launch<T, 128, 128, 64, 64, 2, false>(p, st);
        case 2: return conv_launch<T, 256, 128, 64, 64, 2, true>(p, st);
        case 3: return conv_launch<T, 256, 128, 128, 64, 2, true>(p, st);
        case 4: return conv_launch<T, 256, 256, 128, 64, 2, true>(p, st);
        case 8: return conv_launch<T, 128, 128, 64, 64, 2, true>(p, st);
        case 9: return conv_launch<T, 512, 128, 128, 64, 2, true>(p, st);
        case 5: return conv_launch<T, 256, 128, 64, 64, 3>(p, st);
        case 6: return conv_launch<T, 128, 128, 64, 64, 3>(p, st);
        case 7: return conv_launch<T, 256, 128, 128, 64, 3>(p, st);
        // small maps (the victims' 28x28 .. 7x7 layers): 64-row tiles, so that a launch still has a workgroup for every CU
        case 15: return conv_launch<T, 64, 128, 32, 64, 2, false>(p, st);
        case 16: return conv_launch<T, 64, 64, 32, 32, 2, false>(p, st);
        default: ADVS_FAIL(ADVS_ERR_ARG, "conv2d: unknown tile id %d", tile);
    }
}

static int g_tile_override = 0;
extern "C" int advs_conv_set_tile(int tile) { g_tile_override = tile; return ADVS_OK; }

static int pick_tile(long long m_img, int cout) {
    // measured on MI355X (tools/tune_conv.py, round 1): the 256x256 tile wins (~1.05-1.15 vs
    // ~1.0 PFLOP/s) whenever Cout fills it and the launch still yields >= 1.5 blocks per CU at the
    // headline batch of 32; the 128x128 tile at two blocks per CU wins everywhere else, including Cout = 128.
    // The rule looks at ONE image's pixels (m_img = Ho*Wo), never at the batch: the tile fixes the order of the
    // K summation, so an image must get the same tile whichever batch it is evaluated in (batch-shard equality).
    const long long blocks256 = (long long)cdiv(m_img, 256) * cdiv(cout, 256);
    // Small maps (the victims' 56x56 .. 7x7 layers, the ViTs' token rows): 64-row tiles in 4-wave workgroups -- three to a CU, and a
    // launch of a few thousand rows still has work for every CU.  ResNet-50 attack at batch 32: 396 -> 490 images/s, ViT-B 229 -> 233,
    // VGG16 and the eps-predictor unchanged (tools/_diag sweep, round 2).  64 x 64 tiles below 15 x 15 pixels only: ViT's wide GEMMs
    // (208 rows per image, 768-3072 columns) lose 9 % on them.  (Diagnostic builds read the two thresholds from the environment.)
    static const long long t15_m = ADVS_DIAG_ENV_LL("ADVS_T15_M", 3136);
    static const long long t16_m = ADVS_DIAG_ENV_LL("ADVS_T16_M", 196);
    if (m_img <= t16_m) return 16;
    if (m_img <= t15_m) return 15;
    return (cout % 256 == 0 && blocks256 >= 12) ? 4 : 1;
}
// the 16x16-pixel halo kernel wins wherever it applies (3x3, stride 1, no upsample / extra operand, image a
// multiple of 16: +15..30 % over the per-tap tiles on every such layer of the eps-predictor, tools/tune_conv.py)
// tiles 17-19 (conv_halo2.hip) by the descriptor alone: 16-bit, 3x3 stride 1 pad 1, plain or sub-pixel upsample, optional fused 1x1
static bool halo2_args_ok(const advs_conv_args* a, int tile) {
    if (tile < 17 || tile > 19 || a->dtype == ADVS_F32 || a->relu_mask) return false;
    if (a->norm && (tile != 19 || a->upsample || a->c1 + a->c2 > 384)) return false;     // GroupNorm + SiLU on load: tile 19 only
    if (!(a->ksize == 3 && a->stride == 1 && a->pad == 1 && a->h % 16 == 0 && a->w_ % (tile == 17 ? 32 : 16) == 0)) return false;
    if ((long long)a->b * a->h * a->w_ >= (1ll << 28)) return false;
    if (a->upsample == ADVS_UPSAMPLE_SUBPIXEL) return !a->e1;
    return !a->upsample;
}
static int resolve_tile(const advs_conv_args* a, long long m_img) {
    const int want = g_tile_override ? g_tile_override : a->tile;
    if (a->norm) return halo2_args_ok(a, 19) ? 19 : -1;          // the fused form exists in one kernel only (advs_conv2d rejects -1)
    if (halo2_args_ok(a, want)) return want;
    // Second-generation halo kernels wherever they apply (16-bit, 3x3 stride 1; plain, sub-pixel upsample or fused 1x1 operand).  Measured
    // round 3 (tools/conv_one.py --suite, profiles/round3_halo2.txt): from 128 x 128 maps up the 4-wave form with two workgroups per CU
    // (tile 19) is 4-18 % faster than tile 10 on every layer shape of the eps-predictor, at 64 x 64 the 16 x 32-pixel form (tile 17) by
    // 2-7 %; at 32 x 32 tile 10's 256 workgroups of 8 waves stay ahead.  The rule looks at ONE image's map, never at the batch.
    if (!want && halo2_args_ok(a, 19)) {
        const long long hw = (long long)a->h * a->w_;
        // (sub-pixel upsample convs follow the same rule: in isolation the 16 x 32-pixel tile is 3 % ahead at 128 -> 256 (1084 vs 1116 us),
        // inside the forward 5 % behind (1049 vs 998 us, profiles/round3_c vs round3_b) -- the forward decides)
        if (hw >= 128 * 128) return 19;
        if (hw >= 64 * 64) return a->w_ % 32 == 0 ? 17 : 19;
    }
    if (a->upsample == ADVS_UPSAMPLE_SUBPIXEL) return 12;       // weights are packed per output parity: the halo kernels only
    if (a->relu_mask) return a->tile ? a->tile : pick_tile(m_img, a->cout);
    if (g_tile_override) return g_tile_override;
    if (a->tile) return a->tile;
    // (with a fused 1x1 operand tile 10 becomes 13: the same kernel with one-tap units behind the 3x3 slabs, +3 %)
    const bool no_halo_extra = ADVS_DIAG_ENV("ADVS_NO_HALO_EXTRA");
    if (a->ksize == 3 && a->stride == 1 && a->pad == 1 && !a->upsample && (!a->e1 || !no_halo_extra) && a->h % 16 == 0 && a->w_ % 16 == 0)
        return 10;
    return pick_tile(m_img, a->cout);
}
/* tile id advs_conv2d will use for this descriptor, and the row-block height (rows per stats entry) of a tile id */
extern "C" int advs_conv_resolve_tile(const advs_conv_args* a) {
    if (!a) return 0;
    const int ups = a->upsample ? 1 : 0;
    const long long ho = (((long long)a->h << ups) + 2 * a->pad - a->ksize) / a->stride + 1;
    const long long wo = (((long long)a->w_ << ups) + 2 * a->pad - a->ksize) / a->stride + 1;
    return resolve_tile(a, ho * wo);
}
extern "C" int advs_conv_tile_rows(int tile) {
    switch (tile) {
        case 1: case 2: case 5: case 6: case 8: case 10: case 12: case 13: return 64;
        case 18: case 19: return 256;       // conv_halo2.hip: one entry per workgroup
        case 17: return 512;
        case 3: case 4: case 7: case 9: return 128;
        case 15: case 16: return 32;
        default: return 0;
    }
}

extern "C" int advs_conv2d(const advs_conv_args* a, void* stream) {
    ADVS_REQUIRE(a && a->x1 && a->w && a->y, "conv2d: null pointer");
    ADVS_REQUIRE(dtype_ok(a->dtype), "conv2d: unknown dtype code %d", a->dtype);
    ADVS_REQUIRE(a->b > 0 && a->h > 0 && a->w_ > 0 && a->c1 > 0 && a->c2 >= 0 && a->cout > 0, "conv2d: bad shape");
    ADVS_REQUIRE(a->ksize == 1 || a->ksize == 3, "conv2d: ksize %d unsupported", a->ksize);
    ADVS_REQUIRE(a->stride == 1 || a->stride == 2, "conv2d: stride %d unsupported", a->stride);
    ADVS_REQUIRE(a->pad >= 0 && a->pad <= 1, "conv2d: pad %d unsupported", a->pad);
    ADVS_REQUIRE((a->c2 == 0) == (a->x2 == nullptr), "conv2d: x2/c2 mismatch");
    ADVS_REQUIRE((a->ce1 == 0) == (a->e1 == nullptr) && (a->ce2 == 0) == (a->e2 == nullptr) && (a->e1 || !a->e2),
                 "conv2d: extra operand pointers/channels mismatch");
    ADVS_REQUIRE(!a->e1 || a->stride == 1, "conv2d: the extra 1x1 operand needs stride 1");
    const int esz = a->dtype == ADVS_F32 ? 4 : 2;
    const int bke = SLAB / esz;
    ADVS_REQUIRE(a->c1 % bke == 0 && a->c2 % bke == 0 && a->ce1 % bke == 0 && a->ce2 % bke == 0,
                 "conv2d: channels (%d,%d | %d,%d) must be multiples of %d", a->c1, a->c2, a->ce1, a->ce2, bke);
    ADVS_REQUIRE(a->cout % (16 / esz) == 0, "conv2d: cout=%d must be a multiple of %d", a->cout, 16 / esz);
    ConvKP p;
    p.x1 = (const char*)a->x1; p.x2 = (const char*)a->x2; p.w = (const char*)a->w;
    p.bias = a->bias; p.temb = a->temb; p.res = (const char*)a->residual; p.y = (char*)a->y;
    p.stats = a->stats;
    p.fast_epi = 0;
    p.mask = (const char*)a->relu_mask;
    p.norm = a->norm;
    ADVS_REQUIRE(!a->relu_mask || (!a->stats && a->upsample != ADVS_UPSAMPLE_SUBPIXEL && (a->tile == 0 || a->tile == 1 || a->tile == 4 || a->tile == 15 || a->tile == 16)),
                 "conv2d: relu_mask needs a per-tap tile (0, 1, 4, 15 or 16), no stats, no sub-pixel upsample");
    p.B = a->b; p.H = a->h; p.W = a->w_; p.C1 = a->c1; p.C2 = a->c2; p.Cout = a->cout;
    p.LD1 = a->ld1 > 0 ? a->ld1 : a->c1; p.LD2 = a->ld2 > 0 ? a->ld2 : a->c2;
    ADVS_REQUIRE(p.LD1 <= a->c1 && p.LD2 <= a->c2 && (p.LD1 * esz) % 16 == 0 && (p.LD2 * esz) % 16 == 0 && a->ld1 >= 0 && a->ld2 >= 0,
                 "conv2d: pixel strides (%d,%d) must be 16-byte multiples and <= the channel counts (%d,%d)", a->ld1, a->ld2, a->c1, a->c2);
    p.R = a->ksize; p.stride = a->stride; p.pad = a->pad; p.ups = a->upsample ? 1 : 0;
    const int HL = a->h << p.ups, WL = a->w_ << p.ups;
    p.Ho = (HL + 2 * a->pad - a->ksize) / a->stride + 1;
    p.Wo = (WL + 2 * a->pad - a->ksize) / a->stride + 1;
    const long long M = (long long)a->b * p.Ho * p.Wo;
    ADVS_REQUIRE(M > 0 && M < (1ll << 31) - 256, "conv2d: M=%lld out of range", M);
    p.M = (int)M;
    p.K = a->ksize * a->ksize * (a->c1 + a->c2) + a->ce1 + a->ce2;
    const bool subpixel = a->upsample == ADVS_UPSAMPLE_SUBPIXEL;
    if (subpixel) {
        ADVS_REQUIRE(a->ksize == 3 && a->stride == 1 && a->pad == 1 && !a->e1 && a->h % 16 == 0 && a->w_ % 16 == 0,
                     "conv2d: ADVS_UPSAMPLE_SUBPIXEL needs 3x3 stride 1 pad 1, no extra operand, h and w multiples of 16");
        p.K = 4 * (a->c1 + a->c2);                          // per parity: 2x2 taps; w holds [4][cout][4][c1 + c2]
    }
    p.e1 = (const char*)a->e1; p.e2 = (const char*)a->e2; p.E1 = a->ce1; p.E2 = a->ce2;
    const unsigned long long e1b = (unsigned long long)M * a->ce1 * esz, e2b = (unsigned long long)M * a->ce2 * esz;
    ADVS_REQUIRE(e1b < 0xF0000000ull && e2b < 0xF0000000ull, "conv2d: extra operand exceeds the 32-bit buffer offsets");
    p.e1_bytes = (unsigned)(a->e1 ? e1b : 16); p.e2_bytes = (unsigned)(a->e2 ? e2b : 16);
    const unsigned long long x1b = (unsigned long long)a->b * a->h * a->w_ * p.LD1 * esz;
    const unsigned long long x2b = (unsigned long long)a->b * a->h * a->w_ * p.LD2 * esz;
    const unsigned long long wb = (unsigned long long)a->cout * p.K * esz * (subpixel ? 4 : 1);
    ADVS_REQUIRE(x1b < 0xF0000000ull && x2b < 0xF0000000ull && wb < 0xF0000000ull,
                 "conv2d: a source of %llu bytes exceeds the 32-bit buffer offsets (split the batch)", x1b > x2b ? x1b : x2b);
    p.x1_bytes = (unsigned)x1b; p.x2_bytes = (unsigned)(a->x2 ? x2b : x1b); p.w_bytes = (unsigned)wb;
    p.act = a->act; p.temb_stride = a->temb_stride > 0 ? a->temb_stride : (a->temb_stride < 0 ? 0 : a->cout);    // < 0: one row for every sample
    p.dHoWo.init((unsigned)(p.Ho * p.Wo)); p.dWo.init((unsigned)p.Wo);
    int tile = resolve_tile(a, (long long)p.Ho * p.Wo);
    ADVS_REQUIRE(tile >= 0, "conv2d: norm (GroupNorm + SiLU on load) needs a 16-bit dtype, 3x3 stride 1 pad 1, no upsample, h and w multiples of 16, c1 + c2 <= 384");
    const int h2kind = subpixel ? 1 : (p.e1 ? 2 : 0);
    if (tile >= 17 && tile <= 19 && !conv_halo2_eligible(p, a->dtype, tile, h2kind)) {
        ADVS_REQUIRE(g_tile_override != 0, "conv2d: tile %d (second-generation halo kernel) cannot take this shape", tile);
        tile = subpixel ? 12 : 10;                           // tuning override on a shape it cannot take
    }
    if (tile == 10 && conv_halo_extra_eligible(p)) tile = 13;   // 13: the halo kernel with the fused 1x1 operand
    if (tile == 10 && !conv_halo_eligible(p)) {
        ADVS_REQUIRE(g_tile_override != 0, "conv2d: tile 10 (halo kernel) needs 3x3 stride 1 pad 1, no upsample / extra operand, H and W multiples of 16");
        tile = pick_tile((long long)p.Ho * p.Wo, a->cout);   // tuning override on a shape the halo kernel cannot take
    }
    if (p.stats) {
        const int wm = advs_conv_tile_rows(tile);
        ADVS_REQUIRE(wm > 0 && (p.Ho * p.Wo) % wm == 0, "conv2d: stats need Ho*Wo (%d) to be a multiple of the tile's row block (%d)",
                     p.Ho * p.Wo, wm);
        if (g_tile_override && a->stats_rows != wm) p.stats = nullptr;   // tuning runs: buffer sized for another tile
        else ADVS_REQUIRE(a->stats_rows == wm, "conv2d: stats buffer sized for %d-row blocks but the tile uses %d", a->stats_rows, wm);
    }
    // the halo kernels' fast epilogue (conv_common.h): 16-bit storage, nothing but bias / time embedding / statistics around the GEMM
    p.fast_epi = (tile == 10 || tile == 12 || tile == 13 || (tile >= 17 && tile <= 19)) && a->dtype != ADVS_F32 && !a->residual && !a->relu_mask &&
                 a->act == ADVS_ACT_NONE && !ADVS_DIAG_ENV("ADVS_NO_FAST_EPILOGUE");
    if (tile >= 17 && tile <= 19) return conv_halo2_dispatch(p, a->dtype, tile, h2kind, (hipStream_t)stream);
    if (tile == 12) return conv_halo_subpixel_dispatch(p, a->dtype, (hipStream_t)stream);
    if (tile == 13) return conv_halo_extra_dispatch(p, a->dtype, (hipStream_t)stream);
    if (tile == 10) return conv_halo_dispatch(p, a->dtype, (hipStream_t)stream);
    ADVS_SWITCH_T(a->dtype, return conv_dispatch<T>(p, tile, (hipStream_t)stream));
    return ADVS_ERR_ARG;                    // not reached
}
