// 3x3 stride-1 convolution as an implicit GEMM over a HALO tile (the hot shape of the eps-predictor:
// every ResidualBlock conv, diff_model.py:73,86).
//
// conv_igemm.hip re-fetches the A rows of a tile once per filter tap (9x).  Here a workgroup owns a
// 16x16 block of output pixels of ONE image and stages, per 128-byte channel slab, the 18x18 halo of
// input pixels ONCE (41 KiB); the nine taps then read shifted windows of that patch straight out of
// LDS.  Only the weights (16 KiB per tap) still stream per tap.  Per (slab, 9 taps) the workgroup moves
// 41 + 144 KiB instead of 9 x (32 + 16) KiB, and a wave issues ~2.7 instead of 8 LDS-DMA
// instructions per 16 MFMAs, which is what bounds the per-tap kernel (MI355X_MICROARCH: an LDS-DMA
// piece costs 60-185 issue cycles).
//
// Workgroup: 8 waves (4 along pixels x 2 along channels), each a 64x64 block of 32x32 MFMA tiles;
// output tile 256 pixels x 128 channels.  LDS: A halo double-buffered per slab (2 x 42 KiB), B ring of
// FOUR 16 KiB stages with ONE barrier per tap.  The weights run two taps ahead, so the barrier that
// opens tap t already publishes tap t+1's weights and the first fragments of tap t+1 are read during
// the last MFMA group of tap t: no LDS latency is exposed after a barrier.  The nine taps are
// unrolled so every counted `s_waitcnt vmcnt(N)` is a literal: at the top of tap t everything up to
// W(t+1) must have landed, and the only younger traffic is W(t+2) plus at most two halo pieces.
// Rows of 128 B are XOR-swizzled by (halo pixel >> 1) & 7 on the source address (see conv_igemm.hip).
#include "conv_common.h"
#include <type_traits>

#define HT 16                    // output tile edge
#define HWID 18                  // halo edge
#define HPIX (HWID * HWID)       // 324 halo pixels
#define HPIECES 41               // ceil(324 / 8) DMA pieces of 8 pixels x 128 B
#define HA_STAGE (42 * 1024)     // 41 pieces + one scratch piece for the waves' padding DMAs
#define HB_STAGE (128 * SLAB)    // 16 KiB: 128 output channels x 128 B
#define HNP 6                    // A pieces per wave per slab (8 waves x 6 >= 41)

// MFMA row r of a 32-row tile <-> pixel (tile row g, column idx) of a 2 x 16 pixel strip.  ds_read_b128 serves
// a wave in the fixed lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}; giving each group 16 CONSECUTIVE
// halo pixels makes (pixel & 1, (pixel >> 1) & 7) -- the 16-byte slot after the XOR swizzle -- distinct within
// the group for every tap shift, i.e. the window reads are bank-conflict free.  The natural r -> (r >> 4, r & 15)
// map is 2-way conflicted on every read (a group would straddle two halo rows).
__device__ __forceinline__ void halo_row_map(int r, int& g, int& idx) {
    const int blk = r >> 2;                          // 8 blocks of 4 rows: groups 0 1 1 0 1 0 0 1
    g = (0x96 >> blk) & 1;
    // position of this block among the blocks of its group, times 4
    const int before = __builtin_popcount((g ? 0x96 : 0x69) & ((1 << blk) - 1));
    idx = before * 4 + (r & 3);
}

// NT = 9: the 3x3 convolution.  NT = 4: the SUB-PIXEL form of "nearest x2 upsample, then 3x3" (Upsample,
// diff_model.py:129-140).  Output pixel (2i+a, 2j+b) only ever sees input rows {i-1, i} (a = 0) or {i, i+1} (a = 1),
// and likewise for columns, so each of the four output parities is a 2x2 convolution of the LOW-resolution input
// with weights summed over the taps that coincide (packed per parity by the host: [parity][Cout][2x2][Cin]).
// 4 taps per slab instead of 9: 2.25x fewer MACs than convolving the upsampled image, no upsampled tensor.
// A workgroup then owns (16x16 low-res pixels, one parity, 128 channels) = 256 of the 1024 output pixels above it.
template <int NT> __host__ __device__ constexpr int halo_a_pieces(int t) {
    // halo pieces of the next slab issued during tap t; none in the last two taps (see the wait at the tap top)
    return NT == 9 ? ((t >= 0 && t < HNP) ? 1 : 0) : ((t == 0 || t == 1) ? 3 : 0);
}

// XT: the fused 1x1 operand of ResidualBlock.shortcut (conv_igemm.hip) follows the 3x3 slabs as extra "units" of ONE tap
// each -- the centre tap of the same halo staging, reading e1 / e2 and the weight columns behind the 9*Cin block.  A
// one-tap unit cannot hide its own staging (8 LDS-DMA pieces per 16 MFMAs), but the 3x3 part keeps the halo kernel's rate.
template <typename T, int NT, bool XT = false>
__global__ void __launch_bounds__(512)
conv3x3_halo_kernel(const ConvKP p) {
    static_assert(!XT || NT == 9, "the extra operand is only wired into the 3x3 kernel");
    constexpr int ESZ = Mma<T>::ESZ;
    constexpr int NTW = NT == 9 ? 3 : 2;                         // taps per filter row
    constexpr int BKE = SLAB / ESZ;
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // [2][HA_STAGE] then [4][HB_STAGE]
    char* sA = smem;
    char* sB = smem + 2 * HA_STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5, chunk = lane & 7;

    // ---- workgroup -> (image, pixel tile, channel tile); channel tiles of one pixel tile are neighbours
    // on one XCD so the halo is re-read from that XCD's L2
    const int tiles_x = p.W / HT, tpi = tiles_x * (p.H / HT);
    constexpr int NPAR = NT == 9 ? 1 : 4;                        // output parities per pixel tile
    const int nblk = p.nMt * NPAR * p.nNt;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tl = bid / (NPAR * p.nNt), brem = bid - tl * (NPAR * p.nNt);
    const int par = brem / p.nNt, nt = brem - par * p.nNt;       // parity (a, b) = (par >> 1, par & 1)
    const int r0 = NT == 9 ? 0 : par >> 1, s0 = NT == 9 ? 0 : par & 1;
    const int b = tl / tpi, ti = tl - b * tpi;
    const int ty = ti / tiles_x, tx = ti - ty * tiles_x;
    const int y0 = ty * HT, x0 = tx * HT, n0 = nt * 128;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- staging geometry.  Halo piece q covers halo pixels 8q .. 8q+7 (row-major in the 18x18 patch).
    int apix[HNP];                                   // source pixel index, -1 = outside the image / patch
    unsigned acsw[HNP], a_voff[HNP], a_dst[HNP], b_voff[2];
#pragma unroll
    for (int j = 0; j < HNP; ++j) {
        const int q = wave + 8 * j;
        const int hidx = q * 8 + (lane >> 3);
        const int hy = hidx / HWID, hx = hidx - hy * HWID;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool inb = q < HPIECES && hidx < HPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        apix[j] = inb ? (b * p.H + gy) * p.W + gx : -1;
        acsw[j] = (unsigned)((chunk ^ ((hidx >> 1) & 7)) << 4);
        a_dst[j] = (unsigned)(q < HPIECES ? q : HPIECES) * 1024u;       // padding DMAs land in the scratch piece
        a_voff[j] = OOB_OFFSET;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 8 + (lane >> 3);
        const int n = n0 + row;
        b_voff[j] = (n < p.Cout) ? (unsigned)(par * p.Cout + n) * (unsigned)p.K * ESZ + ((chunk ^ ((row >> 1) & 7)) << 4) : OOB_OFFSET;
    }
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.x1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x1), 0, p.x2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rse1 = __builtin_amdgcn_make_buffer_rsrc((void*)(XT ? p.e1 : p.x1), 0, XT ? p.e1_bytes : 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rse2 = __builtin_amdgcn_make_buffer_rsrc((void*)(XT && p.e2 ? p.e2 : p.x1), 0, XT ? p.e2_bytes : 16, 0x00020000);

    const int Cin = p.C1 + p.C2;
    const int ncs1 = p.C1 / BKE, nunits = Cin / BKE;   // one unit = one 128-byte channel slab (all 9 taps)
    const int ne1 = XT ? p.E1 / BKE : 0, ne = XT ? (p.E1 + p.E2) / BKE : 0;      // one-tap units of the extra operand
    const int gtaps = nunits * NT + ne;                // taps in all

    auto set_a_voff = [&](int unit) {                // byte offsets of the halo pixels in `unit`'s source
        unsigned cs = (unsigned)(unit < ncs1 ? p.LD1 : p.LD2) * ESZ;
        if (XT && unit >= nunits) cs = (unsigned)(unit - nunits < ne1 ? p.E1 : p.E2) * ESZ;
#pragma unroll
        for (int j = 0; j < HNP; ++j) a_voff[j] = apix[j] >= 0 ? (unsigned)apix[j] * cs + acsw[j] : OOB_OFFSET;
    };
    auto issue_A = [&](int unit, int j) {
        char* dst = sA + (unit & 1) * HA_STAGE + a_dst[j];
        if (XT && unit >= nunits) {
            const int e = unit - nunits;
            if (e < ne1) blds16(rse1, a_voff[j], (unsigned)e * SLAB, dst);
            else blds16(rse2, a_voff[j], (unsigned)(e - ne1) * SLAB, dst);
        } else if (unit < ncs1) blds16(rs1, a_voff[j], (unsigned)unit * SLAB, dst);
        else blds16(rs2, a_voff[j], (unsigned)(unit - ncs1) * SLAB, dst);
    };
    auto issue_B = [&](int unit, int t, int j) {     // weights of (slab unit, tap t): K offset t*Cin + unit*BKE
        const unsigned woff = (unsigned)(t * Cin + unit * BKE) * ESZ;
        blds16(rsw, b_voff[j], woff, sB + ((unit * NT + t) & 3) * HB_STAGE + (wave * 2 + j) * 1024);
    };
    auto issue_Bx = [&](int e, int j) {              // weights of extra unit e: K offset NT*Cin + e*BKE, ring slot of its tap
        const unsigned woff = (unsigned)(NT * Cin + e * BKE) * ESZ;
        blds16(rsw, b_voff[j], woff, sB + ((nunits * NT + e) & 3) * HB_STAGE + (wave * 2 + j) * 1024);
    };
    // weights of the tap three ahead of global tap `it` (none past the end)
    auto issue_W3 = [&](int unit, int t, int j) {
        const int g3 = unit * NT + t + 3;
        if (g3 >= gtaps) return;
        if (XT && g3 >= nunits * NT) { issue_Bx(g3 - nunits * NT, j); return; }
        if (t + 3 < NT) issue_B(unit, t + 3, j); else issue_B(unit + 1, t + 3 - NT, j);
    };

    // ---- fragment geometry: wave (wr, wc) owns tile rows 4wr..4wr+3 (x16 px) and channels wc*64..+63
    int hidx0[2], b_off[2], b_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int g, idx;
        halo_row_map(l31, g, idx);
        hidx0[i] = (4 * wr + 2 * i + g) * HWID + idx;
        const int rb = wc * 64 + i * 32 + l31;
        b_off[i] = rb * SLAB; b_sw[i] = (rb >> 1) & 7;
    }

    auto row_to_m = [&](int lr) {
        int g, idx;
        halo_row_map(lr & 31, g, idx);
        const int yy = y0 + 4 * wr + 2 * (lr >> 5) + g, xx = x0 + idx;      // input-grid pixel
        if (NT == 9) return (b * p.H + yy) * p.W + xx;
        return (b * 2 * p.H + 2 * yy + r0) * (2 * p.W) + 2 * xx + s0;       // its parity's output
    };

    // ---- prologue: halo of slab 0, weights of taps 0, 1 and 2
    set_a_voff(0);
#pragma unroll
    for (int j = 0; j < HNP; ++j) issue_A(0, j);
    issue_B(0, 0, 0); issue_B(0, 0, 1);
    issue_B(0, 1, 0); issue_B(0, 1, 1);
    issue_B(0, 2, 0); issue_B(0, 2, 1);

    f32x16 acc[2][2];                                // zero, or (fast epilogue) bias + time embedding
    conv_acc_init<2, 2>(p, acc, lane, n0 + wc * 64, b);

    u32x4 af[2][2], bf[2][2];                        // [k-step parity][tile]; slot 0 is carried across taps
    auto load_frags = [&](int slot, const char* la, const char* lb, int r, int s, int ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hidx = hidx0[i] + r * HWID + s;
            af[slot][i] = *(const u32x4*)(la + hidx * SLAB + (((2 * ks + lh) ^ ((hidx >> 1) & 7)) << 4));
            bf[slot][i] = *(const u32x4*)(lb + b_off[i] + (((2 * ks + lh) ^ b_sw[i]) << 4));
        }
    };

    for (int unit = 0; unit < nunits; ++unit) {
        const bool hn = unit + 1 < nunits + ne;      // a next slab exists: its halo is prefetched during this one
        if (hn && (unit + 1 == ncs1 || unit == 0 || unit + 1 == nunits || unit + 1 == nunits + ne1)) set_a_voff(unit + 1);
        const char* la = sA + (unit & 1) * HA_STAGE;
        auto tap = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int it = unit * NT + t;
            // Outstanding, oldest first: W(t+1) [+halo pieces of tap t-2], W(t+2) [+halo pieces of tap t-1].  Wait for
            // W(t+1).  The last tap of a slab reads the NEXT slab's halo at its end: everything but W(t+2) must be in.
            constexpr int nA = t == NT - 1 ? 0 : halo_a_pieces<NT>(t - 2) + halo_a_pieces<NT>(t - 1);
            if (it + 2 >= gtaps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no W(t+2): nothing may be pending
            else if (hn) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + nA) : "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int r = r0 + t / NTW, s = s0 + t % NTW;
            const char* lb = sB + (it & 3) * HB_STAGE;
            if (it == 0) load_frags(0, la, lb, r, s, 0);              // nothing was carried into the very first tap
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                // this tap's share of the DMA issue, spread between the MFMA groups (weights first), AHEAD of the k-step's fragment
                // prefetch (-0.9 % conv time against the other order, round 2)
                // (3x3: weights in k-steps 1 and 2, the halo piece in 3 -- 1.5 % faster than 0 / 1 / 2, round 2; the issue ORDER inside a
                // tap stays weights, then halo pieces, which is what the counted wait at the tap top assumes)
                constexpr int WK = NT == 9 ? 1 : 0;
                if (ks == WK) issue_W3(unit, t, 0);
                if (ks == WK + 1) issue_W3(unit, t, 1);
                if (NT == 9) {
                    if (ks == 3 && hn && t < HNP) issue_A(unit + 1, t);
                } else if (hn && t < 2) {                             // three halo pieces in each of taps 0 and 1
                    if (ks >= 1) issue_A(unit + 1, 3 * t + ks - 1);
                }
                // (no s_setprio around the MFMAs: with the fast epilogue in place raising the priority costs 2 % of the conv time,
                // profiles/round2_halo512.txt (3))
                if (ks < 3) {
                    load_frags(nxt, la, lb, r, s, ks + 1);
                } else if (t < NT - 1) {                              // first fragments of the next tap, same slab
                    load_frags(nxt, la, sB + ((it + 1) & 3) * HB_STAGE, r0 + (t + 1) / NTW, s0 + (t + 1) % NTW, 0);
                } else if (hn) {                                      // ... or the first tap of the next slab's halo
                    const int nc = (XT && unit + 1 >= nunits) ? 1 : 0;            // an extra unit's only tap is the centre one
                    load_frags(nxt, sA + ((unit + 1) & 1) * HA_STAGE, sB + ((it + 1) & 3) * HB_STAGE, r0 + nc, s0 + nc, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Mma<T>::run(af[cur][i], bf[cur][j], acc[i][j]);
            }
        };
        tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
        tap(std::integral_constant<int, 3>{});
        if constexpr (NT == 9) {
            tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{}); tap(std::integral_constant<int, 6>{});
            tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
        }
    }
    if constexpr (XT) {
        for (int e = 0; e < ne; ++e) {
            const int unit = nunits + e, it = nunits * NT + e;
            const bool hn = e + 1 < ne;
            if (hn && e + 1 == ne1) set_a_voff(unit + 1);
            // this unit's halo was issued during the previous tap, behind W(it+2): everything has to be in
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const char* la = sA + (unit & 1) * HA_STAGE;
            const char* lb = sB + (it & 3) * HB_STAGE;
            if (e > 0) load_frags(0, la, lb, 1, 1, 0);               // unit 0's first fragments came with the last 3x3 tap
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks < 3) load_frags(nxt, la, lb, 1, 1, ks + 1);
                if (ks < 2 && it + 3 < gtaps) issue_Bx(e + 3, ks);
                if (hn) {                                            // the next unit's whole halo rides in this one tap
                    if (ks == 1) { issue_A(unit + 1, 0); issue_A(unit + 1, 1); }
                    if (ks == 2) { issue_A(unit + 1, 2); issue_A(unit + 1, 3); }
                    if (ks == 3) { issue_A(unit + 1, 4); issue_A(unit + 1, 5); }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Mma<T>::run(af[cur][i], bf[cur][j], acc[i][j]);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // every wave is done reading before the patches reuse LDS

    if constexpr (sizeof(T) == 2) {
        if (p.fast_epi) {
            conv_epilogue_fast<T, 64, 2, 2>(p, acc, smem + wave * (32 * 64 * 2), lane, n0 + wc * 64, row_to_m, (tl * NPAR + par) * 4 + wr);
            return;
        }
    }
    conv_epilogue<T, 64, 2, 2>(p, acc, (float*)smem + wave * (32 * 64), lane, n0 + wc * 64, row_to_m,
                               p.temb ? b : -1, (tl * NPAR + par) * 4 + wr);
}

// Can this launch use the halo kernel?  3x3, stride 1, pad 1, no upsample, no extra operand, image a
// multiple of 16 pixels both ways.
bool conv_halo_eligible(const ConvKP& p) {
    return p.R == 3 && p.stride == 1 && p.pad == 1 && p.ups == 0 && p.e1 == nullptr &&
           p.H % HT == 0 && p.W % HT == 0 && p.Ho == p.H && p.Wo == p.W;
}
// ... with the fused 1x1 operand (stride 1: e1 / e2 live on the same pixel grid)?
bool conv_halo_extra_eligible(const ConvKP& p) {
    return p.R == 3 && p.stride == 1 && p.pad == 1 && p.ups == 0 && p.e1 != nullptr &&
           p.H % HT == 0 && p.W % HT == 0 && p.Ho == p.H && p.Wo == p.W;
}
// ... and the sub-pixel form of upsample + 3x3 (weights packed per output parity)?
bool conv_halo_subpixel_eligible(const ConvKP& p) {
    return p.R == 3 && p.stride == 1 && p.pad == 1 && p.ups == 1 && p.e1 == nullptr &&
           p.H % HT == 0 && p.W % HT == 0 && p.Ho == 2 * p.H && p.Wo == 2 * p.W;
}

template <typename T>
static int halo_launch(ConvKP& p, hipStream_t st) {
    constexpr int lds = 2 * HA_STAGE + 4 * HB_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        ADVS_HIP(hipFuncSetAttribute((const void*)conv3x3_halo_kernel<T, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    p.nMt = p.B * (p.H / HT) * (p.W / HT);
    p.nNt = cdiv(p.Cout, 128);
    conv3x3_halo_kernel<T, 9><<<p.nMt * p.nNt, 512, lds, st>>>(p);
    ADVS_CHECK_LAUNCH("conv3x3_halo");
    return ADVS_OK;
}

template <typename T>
static int halo_subpixel_launch(ConvKP& p, hipStream_t st) {
    constexpr int lds = 2 * HA_STAGE + 4 * HB_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        ADVS_HIP(hipFuncSetAttribute((const void*)conv3x3_halo_kernel<T, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    p.nMt = p.B * (p.H / HT) * (p.W / HT);
    p.nNt = cdiv(p.Cout, 128);
    conv3x3_halo_kernel<T, 4><<<p.nMt * 4 * p.nNt, 512, lds, st>>>(p);
    ADVS_CHECK_LAUNCH("conv3x3_halo_subpixel");
    return ADVS_OK;
}

int conv_halo_subpixel_dispatch(ConvKP& p, int dtype, hipStream_t st) {
    ADVS_REQUIRE(conv_halo_subpixel_eligible(p), "conv2d: the sub-pixel upsample conv needs 3x3 stride 1 pad 1, upsample, no extra operand, H and W multiples of 16");
    ADVS_SWITCH_T(dtype, return halo_subpixel_launch<T>(p, st));
    return ADVS_ERR_ARG;                    // not reached
}

template <typename T>
static int halo_extra_launch(ConvKP& p, hipStream_t st) {
    constexpr int lds = 2 * HA_STAGE + 4 * HB_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        ADVS_HIP(hipFuncSetAttribute((const void*)conv3x3_halo_kernel<T, 9, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    p.nMt = p.B * (p.H / HT) * (p.W / HT);
    p.nNt = cdiv(p.Cout, 128);
    conv3x3_halo_kernel<T, 9, true><<<p.nMt * p.nNt, 512, lds, st>>>(p);
    ADVS_CHECK_LAUNCH("conv3x3_halo_extra");
    return ADVS_OK;
}

int conv_halo_extra_dispatch(ConvKP& p, int dtype, hipStream_t st) {
    ADVS_REQUIRE(conv_halo_extra_eligible(p), "conv2d: the halo kernel with the fused 1x1 operand needs 3x3 stride 1 pad 1, no upsample, H and W multiples of 16");
    ADVS_SWITCH_T(dtype, return halo_extra_launch<T>(p, st));
    return ADVS_ERR_ARG;                    // not reached
}

int conv_halo_dispatch(ConvKP& p, int dtype, hipStream_t st) {
    ADVS_REQUIRE(conv_halo_eligible(p), "conv2d: tile 10 (halo kernel) needs 3x3 stride 1 pad 1, no upsample / extra operand, H and W multiples of 16");
    ADVS_SWITCH_T(dtype, return halo_launch<T>(p, st));
    return ADVS_ERR_ARG;                    // not reached
}
