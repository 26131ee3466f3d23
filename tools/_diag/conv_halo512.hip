// 3x3 stride-1 convolution over a HALO tile, 512 pixels x 128 channels per workgroup, ONE wave per SIMD (16-bit dtypes).
// EXPERIMENTAL tile 14: parity-green, selectable (advs_conv_args.tile = 14, or ADVS_HALO512=1 for every eligible conv), NOT what
// advs_conv2d picks by itself -- it is 5-15 % slower than tile 10 (conv_halo.hip).  Kept as the test bed of the form below.
//
// Why it exists: conv_halo.hip gives each of its 8 waves a 64 x 64 block: per K = 16 step a wave reads 2 + 2 fragments for 4
// MFMAs, one ds_read_b128 per MFMA.  Four SIMDs doing that keep the LDS array busy half of every 32-cycle MFMA slot before the
// LDS-DMA staging writes anything into it (MI355X_MICROARCH: ds_read_b128 = 4 LDS cycles per wave instruction, 256 B/clk/CU);
// reading only half the fragments (wrong results) made tile 10's launches 10-20 % shorter (profiles/round2_halo512.txt).  Here a
// wave owns 128 pixels x 128 channels = 4 x 4 MFMA tiles in 256 accumulator registers (AGPRs): 4 + 4 fragments per 16 MFMAs,
// HALF the LDS reads per MFMA; the weights of a tap serve 512 instead of 256 pixels, so LDS-DMA issue per MFMA drops ~40 %.
//
// Workgroup: 4 waves (rows 4w .. 4w+3 of a 16 x 32 pixel tile of ONE image), 512 registers each.  K runs in HALF slabs of 32
// channels (64 bytes per pixel) so that the 18 x 34 halo (612 pixels, 39 KiB) can stay double-buffered: LDS = 2 x 40 KiB
// halo + a ring of four 8 KiB weight stages = 112 KiB.  One barrier per (half slab, tap) = per 32 MFMAs per wave; weights
// run two taps ahead, the next half slab's halo rides in taps 0-4 (two 1 KiB pieces per wave each).  Rows of 64 B are
// XOR-swizzled by (row >> 2) & 3 on the source address: with 16 consecutive rows per ds_read_b128 lane group (h5_row_map)
// every read is bank-conflict free.
//
// Why it loses (same file, five forms measured, 256 x 256 x 128 -> 128 at batch 32, tile 10 = 670-700 us on the same boxes):
// with one wave per SIMD nothing but the wave's own instruction stream fills the matrix pipe, and a 32x32x16 MFMA gap hides
// about five single-issue instructions (MI355X_MICROARCH, instruction costs).
//   * taps unrolled, addresses hoisted by the compiler, LDS-DMA staging: 766 us.  ~110 instructions per 32 MFMAs is within
//     budget, but hipcc emits them as one clump between two runs of 16 back-to-back MFMAs, and each of the four LDS-DMA
//     issues holds the wave 60-185 cycles.
//   * the same with a sched_group_barrier ladder, or with explicit address tables and register staging: 390-980 spilled
//     VGPRs (the unrolled taps' hoisted addresses do not fit beside 256 accumulators); with scratch reloads in the loop 2336 us.
//   * THIS form -- rolled loop, every address derived when used, uniform DMA schedule, ladder: no spills, 750 us; ~300
//     instructions per tap is twice what the gaps hide.
//   * rolled, with the DMA pieces replaced by buffer_load -> ds_write_b128 through registers: 814 us (more instructions still).
// A hand-scheduled instruction stream (precomputed tables in the 200 free VGPRs, one filler per gap) is what this form needs.
#include "conv_common.h"
#include <type_traits>

#define H5_TY 16                     // output tile: 16 rows x 32 columns
#define H5_TX 32
#define H5_HW 34                     // halo width
#define H5_HPIX (18 * H5_HW)         // 612 halo pixels
#define H5_ROWB 64                   // bytes of a half slab per pixel / per weight row
#define H5_PIECES 39                 // ceil(612 / 16) DMA pieces of 16 pixels x 64 B
#define H5_A_STAGE (40 * 1024)       // 39 pieces + one scratch piece for the padding DMAs
#define H5_B_STAGE (128 * H5_ROWB)   // 8 KiB: 128 output channels x 64 B
#define H5_NP 10                     // halo pieces per wave per half slab (4 waves x 10 >= 39)
#define H5_BKE 32                    // channels per half slab

// same lane-group-friendly row map as conv_halo.hip: MFMA row r of a 32-row tile <-> pixel (g, idx) of a 2 x 16 strip
__device__ __forceinline__ void h5_row_map(int r, int& g, int& idx) {
    const int blk = r >> 2;
    g = (0x96 >> blk) & 1;
    const int before = __builtin_popcount((g ? 0x96 : 0x69) & ((1 << blk) - 1));
    idx = before * 4 + (r & 3);
}

template <typename T>
__global__ void __launch_bounds__(256)
conv3x3_halo512_kernel(const ConvKP p) {
    static_assert(sizeof(T) == 2, "16-bit dtypes only");
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // [2][H5_A_STAGE] then [4][H5_B_STAGE]
    char* sA = smem;
    char* sB = smem + 2 * H5_A_STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5, chunk = lane & 3;

    const int tiles_x = p.W / H5_TX, tpi = tiles_x * (p.H / H5_TY);
    const int nblk = p.nMt * p.nNt;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tl = bid / p.nNt, nt = bid - tl * p.nNt;
    const int b = tl / tpi, ti = tl - b * tpi;
    const int ty = ti / tiles_x, tx = ti - ty * tiles_x;
    const int y0 = ty * H5_TY, x0 = tx * H5_TX, n0 = nt * 128;

    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.x1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x1), 0, p.x2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    const int Cin = p.C1 + p.C2;
    const int ncs1 = p.C1 / H5_BKE, nunits = Cin / H5_BKE;       // one unit = one half slab (all 9 taps)
    const int gtaps = nunits * 9;

    // ---- staging.  Everything is derived from (half slab, tap, piece) when it is issued: the loop below is ROLLED, so nothing per
    // tap can be hoisted into registers (the unrolled forms of this kernel spilled: see the header), and every tap issues the same
    // four DMA instructions (two weight pieces, two halo pieces; the ones with nothing to fetch carry an out-of-range offset:
    // zeros into the scratch piece / a retired ring stage), which keeps the counted vmcnt wait one literal.
    // Halo piece q (16 pixels x 64 B) of half slab `unit` into halo buffer unit & 1; dead: 0 or the out-of-range mask
    auto issue_A = [&](int unit, int q, unsigned dead) {
        const int hidx = q * 16 + (lane >> 2);
        const int hy = (hidx * 241) >> 13, hx = hidx - hy * H5_HW;          // / 34, exact below 656
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        // outside the patch or the image -> all ones (sign-bit arithmetic: no branch may split the loop body's scheduling region)
        const unsigned bad = (unsigned)((~(hidx - H5_HPIX) | gy | (p.H - 1 - gy) | gx | (p.W - 1 - gx)) >> 31);
        const bool from1 = unit < ncs1;
        const unsigned cs = (unsigned)(from1 ? p.LD1 : p.LD2) * 2u;
        const unsigned sw = (unsigned)((chunk ^ ((hidx >> 2) & 3)) << 4);
        const unsigned voff = ((unsigned)((b * p.H + gy) * p.W + gx) * cs + sw) | (bad & OOB_OFFSET) | dead;
        const int qd = q < H5_PIECES ? q : H5_PIECES;
        char* dst = sA + (unit & 1) * H5_A_STAGE + qd * 1024;
        const __amdgpu_buffer_rsrc_t rs = from1 ? rs1 : rs2;
        blds16(rs, voff, (unsigned)(from1 ? unit : unit - ncs1) * H5_ROWB, dst);
    };
    unsigned b_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + (lane >> 2);
        const int n = n0 + row;
        b_voff[j] = (n < p.Cout) ? (unsigned)n * (unsigned)p.K * 2u + ((chunk ^ ((row >> 2) & 3)) << 4) : OOB_OFFSET;
    }
    // weights of global tap g (half slab gu, tap gt) into ring stage g & 3
    auto issue_B = [&](int g, int gu, int gt, int j, unsigned dead) {
        const unsigned woff = (unsigned)(gt * Cin + gu * H5_BKE) * 2u;
        blds16(rsw, b_voff[j] | dead, woff, sB + (g & 3) * H5_B_STAGE + (wave * 2 + j) * 1024);
    };

    // ---- fragment geometry: wave w owns tile rows 4w .. 4w+3; MFMA tile i = (row pair i >> 1, column half i & 1)
    int hidx0[4];
    unsigned boff0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int g, idx;
        h5_row_map(l31, g, idx);
        hidx0[i] = (4 * wave + 2 * (i >> 1) + g) * H5_HW + (i & 1) * 16 + idx;
        const int rb = i * 32 + l31;
        boff0[i] = (unsigned)(rb * H5_ROWB + ((lh ^ ((rb >> 2) & 3)) << 4));      // k-step 0; k-step 1 is this ^ 32
    }
    auto row_to_m = [&](int lr) {
        int g, idx;
        h5_row_map(lr & 31, g, idx);
        const int i = lr >> 5;
        const int yy = y0 + 4 * wave + 2 * (i >> 1) + g, xx = x0 + (i & 1) * 16 + idx;
        return (b * p.H + yy) * p.W + xx;
    };

    // ---- prologue: halo of half slab 0, weights of taps 0, 1, 2 -- with padding DMAs so that the loop's uniform count holds from tap 0
    for (int j = 0; j < H5_NP; ++j) issue_A(0, wave + 4 * j, 0u);
    issue_B(0, 0, 0, 0, 0u); issue_B(0, 0, 0, 1, 0u);
    issue_B(1, 0, 1, 0, 0u); issue_B(1, 0, 1, 1, 0u);
    issue_A(1, H5_PIECES, OOB_OFFSET); issue_A(1, H5_PIECES, OOB_OFFSET);
    issue_B(2, 0, 2, 0, 0u); issue_B(2, 0, 2, 1, 0u);
    issue_A(1, H5_PIECES, OOB_OFFSET); issue_A(1, H5_PIECES, OOB_OFFSET);

    f32x16 acc[4][4];
    conv_acc_init<4, 4>(p, acc, lane, n0, b);

    u32x4 af[2][4], bf[2][4];                        // [k-step][tile]
    unsigned aaddr[4];                               // LDS byte address of the A fragments of (tap, k-step 0); k-step 1 is ^ 32
    auto a_addrs = [&](int abuf, int shift) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hidx = hidx0[i] + shift;
            aaddr[i] = (unsigned)(abuf + hidx * H5_ROWB + ((lh ^ ((hidx >> 2) & 3)) << 4));
        }
    };
    auto load_frags = [&](int slot, unsigned bst, unsigned x) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[slot][i] = *(const u32x4*)(sA + (aaddr[i] ^ x));
            bf[slot][i] = *(const u32x4*)(sB + bst + (boff0[i] ^ x));
        }
    };

    // Outstanding at the top of tap g, oldest first: W(g+1), halo pieces of tap g-2, W(g+2), halo pieces of tap g-1: 2 + 2 + 2 may
    // stay in flight.  The halo pieces of the next half slab ride in taps 0-4, so tap 8 (which reads that halo at its end) is covered.
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    a_addrs(0, 0);
    load_frags(0, 0u, 0u);

    int unit = 0, t = 0;
    for (int g = 0; g < gtaps; ++g) {
        // ---- k-step 0: fragments of k-step 1; weights of the tap three ahead
        load_frags(1, (unsigned)(g & 3) * H5_B_STAGE, 32u);
        {
            const int g3 = g + 3, t3 = t + 3 < 9 ? t + 3 : t - 6, u3 = t + 3 < 9 ? unit : unit + 1;
            const unsigned dead = ~(unsigned)((g3 - gtaps) >> 31) & OOB_OFFSET;
            issue_B(g3, u3, t3, 0, dead);
            issue_B(g3, u3, t3, 1, dead);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<T>::run(af[0][i], bf[0][j], acc[i][j]);
        // ---- k-step 1: first fragments of the next tap (the next half slab's halo after tap 8); two halo pieces of the next half slab
        {
            const int tn = t == 8 ? 0 : t + 1, un = t == 8 ? unit + 1 : unit;
            const int rn = (tn * 11) >> 5, sn = tn - 3 * rn;
            a_addrs((un & 1) * H5_A_STAGE, rn * H5_HW + sn);
            load_frags(0, (unsigned)((g + 1) & 3) * H5_B_STAGE, 0u);
            const unsigned dead = ~(unsigned)(((t - 5) & (unit + 1 - nunits)) >> 31) & OOB_OFFSET;     // real: t < 5 and a next half slab
            const int q = dead ? H5_PIECES : wave + 8 * t;             // a padding DMA writes its zeros into the scratch piece
            issue_A(unit + 1, q, dead);
            issue_A(unit + 1, q + 4, dead);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<T>::run(af[1][i], bf[1][j], acc[i][j]);
#pragma unroll
        for (int m = 0; m < 32; ++m) {               // the interleave the scheduler is asked for
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // MFMA
            if ((m & 15) < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // one fragment read
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                      // VALU
            __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);                      // SALU
            if ((m & 7) == 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // one LDS-DMA issue
        }
        t = t == 8 ? 0 : t + 1;
        unit += t == 0 ? 1 : 0;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the padding DMAs too: the epilogue reuses the LDS
    __builtin_amdgcn_s_barrier();

    if (p.fast_epi) {
        conv_epilogue_fast<T, 128, 4, 4>(p, acc, smem + wave * (32 * 128 * 2), lane, n0, row_to_m, tl * 4 + wave);
        return;
    }
    conv_epilogue<T, 128, 4, 4>(p, acc, (float*)smem + wave * (32 * 128), lane, n0, row_to_m, p.temb ? b : -1, tl * 4 + wave);
}

// 3x3, stride 1, pad 1, no upsample, no extra operand, 16-bit, image a multiple of 16 x 32 pixels
bool conv_halo512_eligible(const ConvKP& p, int dtype) {
    return dtype != ADVS_F32 && p.R == 3 && p.stride == 1 && p.pad == 1 && p.ups == 0 && p.e1 == nullptr &&
           p.H % H5_TY == 0 && p.W % H5_TX == 0 && p.Ho == p.H && p.Wo == p.W && p.C1 % H5_BKE == 0 && p.C2 % H5_BKE == 0 && (p.C1 + p.C2) % 64 == 0;
}

template <typename T>
static int halo512_launch(ConvKP& p, hipStream_t st) {
    constexpr int lds = 2 * H5_A_STAGE + 4 * H5_B_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        ADVS_HIP(hipFuncSetAttribute((const void*)conv3x3_halo512_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    p.nMt = p.B * (p.H / H5_TY) * (p.W / H5_TX);
    p.nNt = cdiv(p.Cout, 128);
    conv3x3_halo512_kernel<T><<<p.nMt * p.nNt, 256, lds, st>>>(p);
    ADVS_CHECK_LAUNCH("conv3x3_halo512");
    return ADVS_OK;
}

int conv_halo512_dispatch(ConvKP& p, int dtype, hipStream_t st) {
    ADVS_REQUIRE(conv_halo512_eligible(p, dtype), "conv2d: tile 14 (512-pixel halo kernel) needs a 16-bit dtype, 3x3 stride 1 pad 1, no upsample / extra operand, H a multiple of 16 and W of 32");
    if (dtype == ADVS_BF16) return halo512_launch<BF16>(p, st);
    return halo512_launch<F16>(p, st);
}
