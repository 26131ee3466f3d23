// 3x3 stride-1 convolution over a HALO tile, second generation (16-bit dtypes): the kernel family of conv_halo.hip with
//   (1) the tile geometry as a template parameter, so that one workgroup can own 16 x 32 = 512 output pixels, and
//   (2) LDS images whose swizzle depends on the halo COLUMN only, so that every fragment address of a tap is a precomputed
//       register plus an immediate (conv_halo.hip derives ~35 address VALU per tap from the tap shift).
// Layers: every ResidualBlock conv (diff_model.py:73,86), the Upsample convs in sub-pixel form (diff_model.py:129-140) and the
// fused ResidualBlock.shortcut operand (diff_model.py:89,103).
//
// Why (round 3).  conv_halo.hip's 256-pixel x 128-channel workgroup issues, per tap and wave, 16 MFMAs, 16 ds_read_b128 and 2.7
// LDS-DMA pieces of 60-185 issue cycles each; round 2's ablations put the LDS-DMA issue at 18-24 % and the fragment reads at
// 10-20 % of the launch (profiles/round2_ablation.txt, round2_halo512.txt).  Geometry H2G512: 16 x 32 pixels with 32-channel units
// (64 B per pixel) keeps the 18 x 34 halo double-buffered in 80 KiB; eight waves of 64 pixels x 128 channels (2 x 4 MFMA tiles):
// 12 fragment reads and 1.6 LDS-DMA pieces per 16 MFMAs instead of 16 and 2.7 (-6 % over the layer suite).  Geometry H2G256W4 -- the one
// advs_conv2d picks from 128 x 128 maps up: 16 x 16 pixels, FOUR waves of 64 pixels x 128 channels, 76 KiB of LDS and <= 256 registers, so
// that TWO workgroups share a CU and one's prologue / epilogue / barrier stalls run under the other's MFMAs (-8.5 %; -13..-18 % on the level-0
// shapes whose epilogue is a quarter of the launch).
//
// LDS images.  A: halo pixels row-major [y * HW + x][RB bytes], RB = 16 * NCH; the 16-byte chunk c of pixel (y, x) sits in slot
// c ^ sw(x), sw(x) = (x >> (4 - log2 NCH)) & (NCH - 1).  The 16 lanes of a ds_read_b128 group read 16 CONSECUTIVE columns of one
// row at one chunk: their slots are distinct mod 256 B for every tap shift (conflict-free), a row shift r is the immediate
// r * HW * RB, the column shifts s = 0..2 are TM * 3 precomputed registers and k-step ks is that address ^ (32 * ks).  B (weights of one tap): rows
// [n][RB] swizzled by n the same way.  The LDS-DMA lands bytes lane-linearly, so the swizzle is applied to the SOURCE chunk: NCH
// consecutive lanes fetch one pixel's (one weight row's) RB contiguous bytes in permuted order -- whole 64 / 128 B segments per lane
// group, which is what keeps a piece cheap for the address path.  (First form of this file: a swizzle-free [y][chunk][x] image with
// all-immediate reads, whose DMA lanes each touch a different pixel: 7 % SLOWER than conv_halo.hip at the same geometry -- the
// LDS-DMA's address coalescing, not the address VALU, is what costs; profiles/round3_halo2.txt.)
// Everything else is conv_halo.hip's scheme: weights three taps ahead in a ring of four stages, ONE barrier per tap, the next
// unit's halo riding in the first taps of the current one, counted vmcnt literals (taps unrolled), first fragments of the next tap
// prefetched across the barrier, the shared epilogues of conv_common.h.
#include "conv_common.h"
#include <type_traits>

template <int TX_, int KC_, int WM_, int WN_>
struct H2Geom {
    static constexpr int TY = 16, TX = TX_, KC = KC_, WM = WM_, WN = WN_;
    static constexpr int NW = WM * WN, NTHR = NW * 64;
    static constexpr int MINW = 2;                             // waves per SIMD the register budget must allow (4-wave form: two workgroups per CU)
    static constexpr int HW = TX + 2, HH = TY + 2;             // halo extent
    static constexpr int NCH = KC / 8;                         // 16-byte chunks of one unit per pixel (16-bit elements)
    static constexpr int RB = NCH * 16;                        // bytes of one pixel (one weight row) per unit
    static constexpr int PP = 1024 / RB;                       // pixels (weight rows) per 1 KiB DMA piece
    static constexpr int SWS = NCH == 8 ? 1 : 2;               // swizzle: (x >> SWS) & (NCH - 1)
    static constexpr int A_ROW = HW * RB;                      // bytes of one halo row
    static constexpr int A_PIX = HH * HW;                      // pixels of one halo image
    static constexpr int A_PIECES = (A_PIX + PP - 1) / PP;     // 1 KiB DMA pieces
    static constexpr int A_STAGE = (A_PIECES + 1) * 1024;      // + one scratch piece for the waves' padding DMAs
    static constexpr int B_STAGE = 128 * KC * 2;               // weights of one tap: 128 output channels x KC
    static constexpr int NB = 4;                               // weight ring stages
    static constexpr int NBP = B_STAGE / 1024 / NW;            // weight pieces per wave per tap
    static constexpr int NAP = (A_PIECES + NW - 1) / NW;       // halo pieces per wave per unit
    static constexpr int KS = KC / 16;                         // MFMA k-steps per tap
    static constexpr int WNC = 128 / WN;                       // channels per wave
    static constexpr int TM = (TY * TX / 32) / WM, TN = WNC / 32;
    static constexpr int SPR = TX / 16;                        // 16-pixel strips per tile row
    static constexpr int LDS_MAIN = 2 * A_STAGE + NB * B_STAGE;
    static constexpr int LDS_EPI = NW * 32 * WNC * 4;          // generic epilogue: one f32 patch per wave
    static constexpr int LDS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
    static constexpr int NORM_MAXC = 384;                      // input channels of the GroupNorm-on-load form: an 8-byte (scale, shift) each
    static constexpr int LDS_NORM = LDS_MAIN + NORM_MAXC * 8 > LDS_EPI ? LDS_MAIN + NORM_MAXC * 8 : LDS_EPI;
    static_assert(B_STAGE / 1024 % NW == 0 && NBP >= 1, "whole weight pieces per wave");
    static_assert(TM * WM * 32 == TY * TX && TM >= 1, "pixel tiles divide over the waves");
    static_assert(NAP <= 6, "the halo pieces of a unit ride one per tap in taps 0..5");
    static_assert(NCH == 4 || NCH == 8, "64- or 128-byte pixel rows");
    static_assert(TM == 2 && TY * TX == WM * 64, "a wave owns 64 pixel rows, a workgroup one statistics row block of WM * 64");
};

// MFMA row r of a 32-row tile <-> (strip g, column idx) of two 16-pixel strips (conv_halo.hip: halo_row_map): each ds_read_b128
// lane group {0-3,12-15,20-27} / {4-11,16-19,28-31} gets 16 consecutive columns of one strip.
__device__ __forceinline__ void h2_row_map(int r, int& g, int& idx) {
    const int blk = r >> 2;
    g = (0x96 >> blk) & 1;
    const int before = __builtin_popcount((g ? 0x96 : 0x69) & ((1 << blk) - 1));
    idx = before * 4 + (r & 3);
}

// halo pieces of the next unit issued during tap t (none in the last two taps: the counted wait at the tap top)
template <int NT, int NAP> __host__ __device__ constexpr int h2_a_pieces(int t) {
    if (t < 0) return 0;
    if (NT == 9) return t < NAP ? 1 : 0;
    return t == 0 ? (NAP + 1) / 2 : (t == 1 ? NAP / 2 : 0);       // NT == 4: two taps carry them all
}
template <int NT, int NAP> __host__ __device__ constexpr int h2_a_first(int t) {    // index of the first piece issued in tap t
    int n = 0;
    for (int u = 0; u < t; ++u) n += h2_a_pieces<NT, NAP>(u);
    return n;
}

// NT = 9: 3x3.  NT = 4: sub-pixel form of nearest-x2 + 3x3 (weights [parity][Cout][2x2][Cin], conv_halo.hip).
// XT: fused 1x1 operand (ResidualBlock.shortcut) as one-tap units behind the 3x3 units.
// NORM: GroupNorm + SiLU of the 3x3 input applied in LDS (advs_conv_args.norm; norm_layer + SiLU + Conv2d, diff_model.py:70-73,
// 83-86).  The wave that DMA'd a halo piece transforms it two taps later (its own vmcnt covers its own piece): every lane reads
// back the 16 bytes it fetched, y = silu_fast(fma(x, scale_c, shift_c)) -- gn_apply_kernel's arithmetic, rounded to T the same
// way, so the MFMA sees the bits the two-pass form would have stored -- and writes them in place; lanes whose slot is zero
// padding (outside the image) leave the zeros.  The next unit's pieces ride in taps 0-5 and are transformed in taps 2-7; the
// barrier that opens the last tap publishes them before its closing prefetch reads the next halo.  ~76 VALU issue slots per 8
// elements, once per element per workgroup (plus the halo ring) -- against a separate pass that reads and writes the tensor
// through HBM.  Measured (round 3, level-0 layers, batch 32): +0.10-0.12 ms per 128 input channels on the conv against 0.19 ms
// for the pass it replaces; the cost equals the transform's VALU issue time, and a sched_group_barrier ladder that spreads it
// between the MFMAs changes nothing (20.86 vs 20.74 ms of conv per forward) -- it is issue / power, not placement.  With two
// 128-channel output tiles the transform would run twice per element and lose: the host fuses single-tile layers only.
// The (scale, shift) rows of the image sit in LDS behind the weight ring.
template <typename T, typename G, int NT, bool XT, bool NORM = false>
__global__ void __launch_bounds__(G::NTHR, G::MINW)
conv3x3_halo2_kernel(const ConvKP p) {
    static_assert(!NORM || NT == 9, "the normalising form is the 3x3 kernel's");
    static_assert(sizeof(T) == 2, "16-bit dtypes only");
    static_assert(!XT || NT == 9, "the extra operand is only wired into the 3x3 kernel");
    constexpr int TM = G::TM, TN = G::TN, KS = G::KS, NAP = G::NAP, NBP = G::NBP, NW = G::NW, KC = G::KC;
    constexpr int NTW = NT == 9 ? 3 : 2;
    constexpr int NPAR = NT == 9 ? 1 : 4;
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // [2][A_STAGE] then [NB][B_STAGE]
    char* sA = smem;
    char* sB = smem + 2 * G::A_STAGE;
    char* sN = smem + G::LDS_MAIN;                                  // NORM: [Cin][2] f32 (scale, shift) of this image

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / G::WN, wn = wave - wm * G::WN;

    // ---- workgroup -> (image, pixel tile, parity, channel tile), XCD-aware (channel tiles of a pixel tile share an L2)
    const int tiles_x = p.W / G::TX, tpi = tiles_x * (p.H / G::TY);
    const int nblk = p.nMt * NPAR * p.nNt;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tl = bid / (NPAR * p.nNt), brem = bid - tl * (NPAR * p.nNt);
    const int par = brem / p.nNt, nt = brem - par * p.nNt;
    const int r0 = NT == 9 ? 0 : par >> 1, s0 = NT == 9 ? 0 : par & 1;
    const int b = tl / tpi, ti = tl - b * tpi;
    const int ty = ti / tiles_x, tx = ti - ty * tiles_x;
    const int y0 = ty * G::TY, x0 = tx * G::TX, n0 = nt * 128;

    // ---- staging geometry.  Halo piece q holds pixels PP*q .. PP*q+PP-1 (row-major in the halo patch); NCH consecutive lanes fill one
    // pixel's slots in order, each fetching the chunk that belongs there (slot ^ sw(x)).
    unsigned a_pk[NAP];                              // (pixel index << 3 | source chunk), ~0 = outside the image / the patch
    unsigned b_voff[NBP];
#pragma unroll
    for (int j = 0; j < NAP; ++j) {
        const int q = wave + NW * j;
        const int hidx = q * G::PP + lane / G::NCH;
        const int hy = hidx / G::HW, hx = hidx - hy * G::HW;
        const int c = (lane & (G::NCH - 1)) ^ ((hx >> G::SWS) & (G::NCH - 1));
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool inb = q < G::A_PIECES && hidx < G::A_PIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        a_pk[j] = inb ? ((unsigned)((b * p.H + gy) * p.W + gx) << 3) | (unsigned)c : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
        const int n = (wave * NBP + j) * G::PP + lane / G::NCH;   // weight row of this lane
        const int c = (lane & (G::NCH - 1)) ^ ((n >> G::SWS) & (G::NCH - 1));
        b_voff[j] = (n0 + n < p.Cout) ? (unsigned)(par * p.Cout + n0 + n) * (unsigned)p.K * 2u + (unsigned)c * 16u : OOB_OFFSET;
    }
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.x1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x1), 0, p.x2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rse1 = __builtin_amdgcn_make_buffer_rsrc((void*)(XT ? p.e1 : p.x1), 0, XT ? p.e1_bytes : 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rse2 = __builtin_amdgcn_make_buffer_rsrc((void*)(XT && p.e2 ? p.e2 : p.x1), 0, XT ? p.e2_bytes : 16, 0x00020000);

    const int Cin = p.C1 + p.C2;
    const int ncs1 = p.C1 / KC, nunits = Cin / KC;     // one unit = KC channels (all NT taps)
    const int ne1 = XT ? p.E1 / KC : 0, ne = XT ? (p.E1 + p.E2) / KC : 0;        // one-tap units of the extra operand
    const int gtaps = nunits * NT + ne;

    // byte offset of this lane's halo slot in the source of the unit being staged: pixel stride a_cs (set when the source changes).
    // Derived per issue from a_pk (3 VALU) rather than kept per piece: the registers go to the accumulators' neighbours.
    unsigned a_cs = 0;
    auto set_a_voff = [&](int unit) {
        a_cs = (unsigned)(unit < ncs1 ? p.LD1 : p.LD2) * 2u;
        if (XT && unit >= nunits) a_cs = (unsigned)(unit - nunits < ne1 ? p.E1 : p.E2) * 2u;
    };
    auto issue_A = [&](int unit, int j) {
        const int q = wave + NW * j;
        char* dst = sA + (unit & 1) * G::A_STAGE + (q < G::A_PIECES ? q : G::A_PIECES) * 1024;    // padding DMAs: scratch piece
        const unsigned voff = a_pk[j] != 0xFFFFFFFFu ? (a_pk[j] >> 3) * a_cs + (a_pk[j] & 7u) * 16u : OOB_OFFSET;
        if (XT && unit >= nunits) {
            const int e = unit - nunits;
            if (e < ne1) blds16(rse1, voff, (unsigned)(e * KC * 2), dst);
            else blds16(rse2, voff, (unsigned)((e - ne1) * KC * 2), dst);
        } else if (unit < ncs1) blds16(rs1, voff, (unsigned)(unit * KC * 2), dst);
        else blds16(rs2, voff, (unsigned)((unit - ncs1) * KC * 2), dst);
    };
    auto issue_Bg = [&](int g, unsigned woff, int j) {          // weights of global tap g: K byte offset woff, ring stage g & 3
        blds16(rsw, b_voff[j], woff, sB + (g & (G::NB - 1)) * G::B_STAGE + (wave * NBP + j) * 1024);
    };
    // weights of the tap three ahead of (unit, t) (none past the end)
    auto issue_W3 = [&](int unit, int t, int j) {
        const int g3 = unit * NT + t + 3;
        if (g3 >= gtaps) return;
        if (XT && g3 >= nunits * NT) { issue_Bg(g3, (unsigned)(NT * Cin + (g3 - nunits * NT) * KC) * 2u, j); return; }
        const int u3 = t + 3 < NT ? unit : unit + 1, t3 = t + 3 < NT ? t + 3 : t + 3 - NT;
        issue_Bg(g3, (unsigned)(t3 * Cin + u3 * KC) * 2u, j);
    };

    // NORM: half h (four channels) of this lane's vector of piece j of `unit`'s halo -- this wave's own DMA, already waited for --
    // becomes SiLU(scale * x + shift) in place.  For the transform a lane always works on source chunk (lane % NCH) of pixel
    // (lane / NCH) of the piece, whatever slot the swizzle put it in (slot = chunk ^ sw(x) = the source chunk the DMA lane
    // fetched, a_pk & (NCH - 1)): its eight (scale, shift) pairs then depend on the unit only and live in registers (nco, loaded
    // once per unit) instead of being read from LDS for every vector.  Branch-free, so that a tap stays one scheduling region:
    // a lane on zero padding (outside the image) writes its zeros back, a padding piece (q >= A_PIECES) works on the scratch piece.
    f32x4 nco[4];                                    // (scale, shift) of channels 8 * (lane % NCH) .. +7 of the unit being normalised
    auto norm_coeffs = [&](int unit) {
        const float* co = (const float*)(sN + (unit * KC + (lane & (G::NCH - 1)) * 8) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) nco[i] = *(const f32x4*)(co + 4 * i);
    };
    auto norm_half = [&](int unit, int j, int h, bool on) {
        const int q = wave + NW * j;
        char* slot = sA + (unit & 1) * G::A_STAGE + (q < G::A_PIECES ? q : G::A_PIECES) * 1024 + (lane / G::NCH) * G::RB
                     + (int)(a_pk[j] & (G::NCH - 1)) * 16 + h * 8;
        const u32x2 raw = *(const u32x2*)slot;
        const f32x4 c0 = nco[2 * h], c1 = nco[2 * h + 1];
        float f[4];
        if constexpr (std::is_same<T, BF16>::value) {
            f[0] = __uint_as_float(raw[0] << 16); f[1] = __uint_as_float(raw[0] & 0xffff0000u);
            f[2] = __uint_as_float(raw[1] << 16); f[3] = __uint_as_float(raw[1] & 0xffff0000u);
        } else {
            f[0] = f16_to_f32((unsigned short)(raw[0] & 0xffffu)); f[1] = f16_to_f32((unsigned short)(raw[0] >> 16));
            f[2] = f16_to_f32((unsigned short)(raw[1] & 0xffffu)); f[3] = f16_to_f32((unsigned short)(raw[1] >> 16));
        }
        f[0] = silu_fast_prescaled(fmaf(f[0], c0[0], c0[1]));      // the table's (scale, shift) carry the factor log2(e)
        f[1] = silu_fast_prescaled(fmaf(f[1], c0[2], c0[3]));
        f[2] = silu_fast_prescaled(fmaf(f[2], c1[0], c1[1]));
        f[3] = silu_fast_prescaled(fmaf(f[3], c1[2], c1[3]));
        const bool live = on && a_pk[j] != 0xFFFFFFFFu;       // (`on` false: the next unit is not a 3x3 unit -- its bytes go back unchanged)
        u32x2 out;
        out[0] = live ? pack2<T>(f[0], f[1]) : raw[0];
        out[1] = live ? pack2<T>(f[2], f[3]) : raw[1];
        *(u32x2*)slot = out;
    };

    // ---- fragment geometry: wave (wm, wn) owns MFMA pixel tiles wm*TM .. +TM-1 (two 16-pixel strips each) and channels wn*WNC ..
    // a_addr[i][s]: byte address inside the halo image of tile i's fragment at column shift s0 + s, k-step 0, row shift 0; k-step ks
    // is that address ^ (32 * ks): the chunk index (2 ks + lh) ^ sw occupies bits 4.. of the in-row offset, rows are RB-aligned
    unsigned a_addr[TM][NTW], b_addr[TN];
    {
        int g, idx;
        h2_row_map(l31, g, idx);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int strip = 2 * (wm * TM + i) + g;
            const int sy = strip / G::SPR, sx = (strip - sy * G::SPR) * 16;
#pragma unroll
            for (int s = 0; s < NTW; ++s) {
                const int hx = sx + idx + s0 + s;
                const int sw = (hx >> G::SWS) & (G::NCH - 1);
                a_addr[i][s] = (unsigned)(((sy + r0) * G::HW + hx) * G::RB + ((lh ^ sw) << 4));
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = wn * G::WNC + j * 32 + l31;
            const int sw = (n >> G::SWS) & (G::NCH - 1);
            b_addr[j] = (unsigned)(n * G::RB + ((lh ^ sw) << 4));
        }
    }
    auto row_to_m = [&](int lr) {
        int g, idx;
        h2_row_map(lr & 31, g, idx);
        const int strip = 2 * (wm * TM + (lr >> 5)) + g;
        const int sy = strip / G::SPR, sx = (strip - sy * G::SPR) * 16;
        const int yy = y0 + sy, xx = x0 + sx + idx;                          // input-grid pixel
        if (NT == 9) return (b * p.H + yy) * p.W + xx;
        return (b * 2 * p.H + 2 * yy + r0) * (2 * p.W) + 2 * xx + s0;       // its parity's output
    };

    // ---- prologue: halo of unit 0, weights of taps 0, 1 and 2
    set_a_voff(0);
#pragma unroll
    for (int j = 0; j < NAP; ++j) issue_A(0, j);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int j = 0; j < NBP; ++j)
            if (g < gtaps) {
                if (XT && g >= nunits * NT) issue_Bg(g, (unsigned)(NT * Cin + (g - nunits * NT) * KC) * 2u, j);
                else issue_Bg(g, (unsigned)((g % NT) * Cin + (g / NT) * KC) * 2u, j);
            }

    if constexpr (NORM) {
        // the image's (scale, shift) rows -> LDS, then unit 0's halo pieces (issued ahead of the three weight stages) are transformed
        const float2* tb = (const float2*)p.norm + (size_t)b * Cin;
        for (int c = tid; c < Cin; c += G::NTHR) ((float2*)sN)[c] = tb[c];
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NBP) : "memory");
        __syncthreads();                                             // the table is visible (the weight DMAs drain here too: once per workgroup)
        norm_coeffs(0);
#pragma unroll
        for (int j = 0; j < NAP; ++j) { norm_half(0, j, 0, true); norm_half(0, j, 1, true); }
    }

    f32x16 acc[TM][TN];                              // zero, or (fast epilogue) bias + time embedding
    conv_acc_init<TM, TN>(p, acc, lane, n0 + wn * G::WNC, b);

    u32x4 af[2][TM], bf[2][TN];                      // [k-step parity][tile]; slot 0 is carried across taps
    // fragments of (row shift r, column shift s, k-step ks): a_addr already carries the CURRENT halo buffer's base (it is moved by
    // +-A_STAGE at every unit boundary), `adelta` reaches the other buffer for the prefetch across a unit boundary; `boff` is the
    // weight ring stage of the tap.  Both offsets are kept opaque to the compiler: left to itself it materialises the sums for
    // both halo buffers and all four ring stages (~50 registers) and spills beside the 128 accumulators.
    auto load_frags = [&](int slot, int adelta, int boff, int r, int s, int ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[slot][i] = *(const u32x4*)(sA + ((a_addr[i][s] ^ (unsigned)(32 * ks)) + adelta) + r * G::A_ROW);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[slot][j] = *(const u32x4*)(sB + ((b_addr[j] ^ (unsigned)(32 * ks)) + boff));
    };
    auto mma_step = [&](int cur) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) Mma<T>::run(af[cur][i], bf[cur][j], acc[i][j]);
    };

    // the very first fragments: nothing was carried into tap 0 of unit 0 (its own wait + barrier follow; once per workgroup)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NBP) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0, 0, 0, 0);

    for (int unit = 0; unit < nunits; ++unit) {
        const bool hn = unit + 1 < nunits + ne;      // a next unit exists: its halo is prefetched during this one
        if (hn && (unit + 1 == ncs1 || unit == 0 || unit + 1 == nunits || unit + 1 == nunits + ne1)) set_a_voff(unit + 1);
        int nxt_delta = (unit & 1) ? -G::A_STAGE : G::A_STAGE;   // from this unit's halo buffer to the next one's
        asm volatile("" : "+s"(nxt_delta));
        const bool norm_on = NORM && unit + 1 < nunits;          // the next unit is a 3x3 unit: its halo is normalised in this one
        auto tap = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int it = unit * NT + t;
            // Outstanding, oldest first: W(t+1) [+halo pieces of tap t-2], W(t+2) [+halo pieces of tap t-1].  Wait for W(t+1).
            // The last tap of a unit reads the NEXT unit's halo at its end: everything but W(t+2) must be in.
            // (NORM: the piece of tap t-2 is transformed in this tap, so it must be in as well)
            constexpr int nA = t == NT - 1 ? 0 : (NORM ? 0 : h2_a_pieces<NT, NAP>(t - 2)) + h2_a_pieces<NT, NAP>(t - 1);
            static_assert(h2_a_pieces<NT, NAP>(NT - 2) == 0, "no halo piece may be issued behind W(t+2) of the last tap");
            if (it + 2 >= gtaps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no W(t+2): nothing may be pending
            else if (hn) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + nA) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            constexpr int r = t / NTW, s = t % NTW;                   // tap shift relative to (r0, s0), which a_addr carries
            int lb = (it & (G::NB - 1)) * G::B_STAGE, lbn = ((it + 1) & (G::NB - 1)) * G::B_STAGE;
            asm volatile("" : "+s"(lb), "+s"(lbn));
            // this tap's DMA issues, weights first (the counted wait assumes that order), spread towards the late k-steps; they open
            // their k-step, so that what follows (fragment reads, NORM arithmetic, MFMAs) is ONE scheduling region per k-step
            constexpr int nAt = h2_a_pieces<NT, NAP>(t), a0 = h2_a_first<NT, NAP>(t);
            constexpr int nops = NBP + nAt;
            // NORM: this tap transforms the piece the wave issued two taps ago, half of it beside each of the first two k-steps' MFMAs
            constexpr bool norm_here = NORM && t >= 2 && h2_a_pieces<NT, NAP>(t - 2) > 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
#pragma unroll
                for (int k = 0; k < nops; ++k) {
                    const int at = KS - nops + k < 0 ? 0 : KS - nops + k;
                    if (at != ks) continue;
                    if (k < NBP) issue_W3(unit, t, k);
                    else if (hn) issue_A(unit + 1, a0 + k - NBP);
                }
                if (ks < KS - 1) {
                    load_frags(nxt, 0, lb, r, s, ks + 1);
                } else if (t < NT - 1) {                              // first fragments of the next tap, same unit
                    constexpr int rn = (t + 1) / NTW, sn = (t + 1) % NTW;
                    load_frags(nxt, 0, lbn, rn, sn, 0);
                } else if (hn) {                                      // ... or the first tap of the next unit's halo
                    if (XT && unit + 1 >= nunits) load_frags(nxt, nxt_delta, lbn, 1, 1, 0);   // an extra unit's only tap: the centre
                    else load_frags(nxt, nxt_delta, lbn, 0, 0, 0);
                }
                if constexpr (NORM) {
                    if (t == 1 && ks == 0 && norm_on) norm_coeffs(unit + 1);       // ahead of the first piece's tap (wave-uniform branch)
                }
                if constexpr (norm_here) {
                    if (ks < 2) norm_half(unit + 1, t - 2, ks, norm_on);
                }
                mma_step(cur);
            }
        };
        static_assert(KS % 2 == 0, "fragment slot 0 carries into the next tap");
        tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
        tap(std::integral_constant<int, 3>{});
        if constexpr (NT == 9) {
            tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{}); tap(std::integral_constant<int, 6>{});
            tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int s = 0; s < NTW; ++s) a_addr[i][s] += nxt_delta;                  // the next unit reads the other buffer
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
            int lb = (it & (G::NB - 1)) * G::B_STAGE;
            asm volatile("" : "+s"(lb));
            if (e > 0) load_frags(0, 0, lb, 1, 1, 0);                // unit 0's first fragments came with the last 3x3 tap
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks < KS - 1) load_frags(nxt, 0, lb, 1, 1, ks + 1);
                if (ks == 0 && it + 3 < gtaps) {
#pragma unroll
                    for (int j = 0; j < NBP; ++j) issue_Bg(it + 3, (unsigned)(NT * Cin + (e + 3) * KC) * 2u, j);
                }
                if (hn) {                                            // the next unit's whole halo rides in this one tap
#pragma unroll
                    for (int j = 0; j < NAP; ++j)
                        if (j * KS / NAP == ks) issue_A(unit + 1, j);
                }
                mma_step(cur);
            }
            const int dlt = (unit & 1) ? -G::A_STAGE : G::A_STAGE;  // the next extra unit reads the other halo buffer
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int s = 0; s < NTW; ++s) a_addr[i][s] += dlt;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // padding DMAs included: the epilogue reuses the LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // every wave is done reading before the patches reuse LDS

    // Epilogue statistics: ONE (sum, sum of squares) entry per workgroup and channel.  Each wave leaves its 64 rows' sums at the head of
    // its own (by then idle) patch, and after a barrier 128 threads add the WM waves of their channel up in wave order.  [One entry per
    // wave until round 3: at the level-0 shapes that was 33.5 MB of statistics per launch against 537 MB of y -- written here, read again
    // by the fold -- now a quarter (tiles 18/19) or an eighth (tile 17).]
    const int rb = tl * NPAR + par;
    if (p.fast_epi) {
        char* patch = smem + wave * (32 * G::WNC * 2);
        conv_epilogue_fast<T, G::WNC, TM, TN>(p, acc, patch, lane, n0 + wn * G::WNC, row_to_m, rb, p.stats ? (float2*)patch : nullptr);
    } else {
        float* patch = (float*)smem + wave * (32 * G::WNC);
        conv_epilogue<T, G::WNC, TM, TN>(p, acc, patch, lane, n0 + wn * G::WNC, row_to_m, p.temb ? b : -1, rb, p.stats ? (float2*)patch : nullptr);
    }
    if (!p.stats) return;
    __syncthreads();
    if (tid < 128) {
        const int cw = tid / G::WNC, cc = tid - cw * G::WNC;          // the wave column that owns channel tid, and its slot there
        const int pstride = p.fast_epi ? 32 * G::WNC * 2 : 32 * G::WNC * 4;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < G::WM; ++w) {
            const float2 v = ((const float2*)(smem + (w * G::WN + cw) * pstride))[cc];
            s += v.x;
            q += v.y;
        }
        const int c = n0 + tid;
        if (c < p.Cout) *(float2*)(p.stats + ((size_t)rb * p.Cout + c) * 2) = make_float2(s, q);
    }
}

// ---- host side
typedef H2Geom<32, 32, 8, 1> H2G512;      // tile 17: 16 x 32 pixels, 8 waves x (64 px x 128 ch), 32-channel units
typedef H2Geom<16, 64, 4, 2> H2G256;      // tile 18: conv_halo.hip's geometry (16 x 16 pixels, 8 waves x (64 px x 64 ch), 64-channel units)
typedef H2Geom<16, 32, 4, 1> H2G256W4;    // tile 19: 16 x 16 pixels, 4 waves x (64 px x 128 ch), 32-channel units, two workgroups per CU

static int h2_tx(int tile) { return tile == 17 ? 32 : 16; }

// kind: 0 = 3x3, 1 = sub-pixel upsample (weights packed per parity), 2 = 3x3 + fused 1x1 operand
bool conv_halo2_eligible(const ConvKP& p, int dtype, int tile, int kind) {
    if (tile < 17 || tile > 19 || dtype == ADVS_F32) return false;
    if (!(p.R == 3 && p.stride == 1 && p.pad == 1 && p.H % 16 == 0 && p.W % h2_tx(tile) == 0)) return false;
    if ((long long)p.B * p.H * p.W >= (1ll << 28)) return false;               // (pixel << 3 | chunk) in 32 bits
    if (p.C1 % 64 || p.C2 % 64 || p.E1 % 64 || p.E2 % 64) return false;
    if (kind == 1) return p.ups == 1 && p.e1 == nullptr && p.Ho == 2 * p.H && p.Wo == 2 * p.W;
    if (p.ups != 0 || p.Ho != p.H || p.Wo != p.W) return false;
    return kind == 2 ? p.e1 != nullptr : p.e1 == nullptr;
}

template <typename T, typename G, int NT, bool XT, bool NORM = false>
static int h2_launch(ConvKP& p, hipStream_t st) {
    constexpr int lds = NORM ? G::LDS_NORM : G::LDS;
    static bool attr_set = false;
    if (!attr_set) {
        ADVS_HIP(hipFuncSetAttribute((const void*)conv3x3_halo2_kernel<T, G, NT, XT, NORM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    p.nMt = p.B * (p.H / G::TY) * (p.W / G::TX);
    p.nNt = cdiv(p.Cout, 128);
    conv3x3_halo2_kernel<T, G, NT, XT, NORM><<<p.nMt * (NT == 9 ? 1 : 4) * p.nNt, G::NTHR, lds, st>>>(p);
    ADVS_CHECK_LAUNCH("conv3x3_halo2");
    return ADVS_OK;
}

template <typename T, typename G>
static int h2_kind(ConvKP& p, int kind, hipStream_t st) {
    if (kind == 1) return h2_launch<T, G, 4, false>(p, st);
    if (kind == 2) return h2_launch<T, G, 9, true>(p, st);
    return h2_launch<T, G, 9, false>(p, st);
}

int conv_halo2_dispatch(ConvKP& p, int dtype, int tile, int kind, hipStream_t st) {
    ADVS_REQUIRE(conv_halo2_eligible(p, dtype, tile, kind),
                 "conv2d: tiles 17-19 (second-generation halo kernels) need a 16-bit dtype, 3x3 stride 1 pad 1, H a multiple of 16 and W of %d", h2_tx(tile));
    if (p.norm) {                                    // GroupNorm + SiLU on load: the 4-wave form only
        ADVS_REQUIRE(tile == 19 && kind != 1 && p.C1 + p.C2 <= H2G256W4::NORM_MAXC, "conv2d: norm needs tile 19, no upsample, c1 + c2 <= %d", H2G256W4::NORM_MAXC);
        if (dtype == ADVS_BF16) return kind == 2 ? h2_launch<BF16, H2G256W4, 9, true, true>(p, st) : h2_launch<BF16, H2G256W4, 9, false, true>(p, st);
        return kind == 2 ? h2_launch<F16, H2G256W4, 9, true, true>(p, st) : h2_launch<F16, H2G256W4, 9, false, true>(p, st);
    }
    if (dtype == ADVS_BF16) {
        if (tile == 17) return h2_kind<BF16, H2G512>(p, kind, st);
        if (tile == 18) return h2_kind<BF16, H2G256>(p, kind, st);
        return h2_kind<BF16, H2G256W4>(p, kind, st);
    }
    if (tile == 17) return h2_kind<F16, H2G512>(p, kind, st);
    if (tile == 18) return h2_kind<F16, H2G256>(p, kind, st);
    return h2_kind<F16, H2G256W4>(p, kind, st);
}
