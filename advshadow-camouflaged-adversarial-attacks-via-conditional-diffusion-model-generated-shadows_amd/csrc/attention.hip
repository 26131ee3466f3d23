// Flash-style self-attention on the matrix cores: softmax(q k^T / sqrt(d)) v per (batch, head)
// with an online softmax, never materialising the N x N scores.
//
// Workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries and walks
// the keys in tiles of 64.  Scores are computed TRANSPOSED, S^T = K Q^T, so that in the 32x32
// accumulator layout the query sits on the lane and the keys on the registers: the row max /
// row sum are in-register reductions plus one exchange with lane^32, and the exponentiated
// tile is, register for register, the B operand of the next product O^T += V^T P^T (no LDS
// round trip).  V is transposed once while staging it into LDS.
//
// FLOPs per launch: 4 * B * heads * N^2 * d.
#include "common.h"
#include <stdlib.h>

#define AT_THREADS 256
#define KT 64          // keys per tile

struct AttnP {
    const char* qkv; char* out;
    int B, N, heads, d, ld, q_off, k_off, v_off, head_stride;
    int n_valid;                 // keys >= n_valid are padding: masked out of the softmax
    float scale_log2e;
    const float* bias;           // optional additive score bias [bias_mod][heads][N][N] (already times log2 e), or null
    int bias_mod;                // sequence b uses bias block b % bias_mod (Swin: one block per window position)
};

template <typename T> struct AMma;
template <> struct AMma<BF16> { static constexpr int ESZ = 2; };
template <> struct AMma<F16> { static constexpr int ESZ = 2; };
template <> struct AMma<float> { static constexpr int ESZ = 4; };

template <typename T, int DT>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn_kernel(const AttnP p) {
    constexpr int ESZ = AMma<T>::ESZ;
    constexpr int VEC = 16 / ESZ;
    constexpr int DMAX = DT * 32;
    constexpr int KS = DMAX * ESZ + 16;           // K tile row stride (bytes), padded
    constexpr int VS = KT * ESZ + (ESZ == 2 ? 8 : 16);   // V^T tile row stride (bytes): 34 / 68 dwords spread the rows over the banks
    constexpr int QSTEPS = DMAX * ESZ / 32;       // max 32-byte k-steps over d
    extern __shared__ __attribute__((aligned(16))) char sm[];     // KT*KS + DMAX*VS bytes
    char* sK = sm;
    char* sV = sm + KT * KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool active = q0 < p.N;
    const int d = p.d;
    const int dbytes = d * ESZ;                   // a multiple of 16; a trailing half step is zero-filled
    const int dsteps = (dbytes + 31) / 32;        // 32-byte steps actually used
    const size_t rowb = (size_t)p.ld * ESZ;
    const char* base = p.qkv + (size_t)b * p.N * rowb;
    const char* qp = base + (size_t)(p.q_off + hd * p.head_stride) * ESZ;
    const char* kp = base + (size_t)(p.k_off + hd * p.head_stride) * ESZ;
    const char* vp = base + (size_t)(p.v_off + hd * p.head_stride) * ESZ;

    // Q fragments: lane (query l31, half lh) holds bytes [32*s + 16*lh, +16) of its query row
    u32x4 qf[QSTEPS];
#pragma unroll
    for (int s = 0; s < QSTEPS; ++s) {
        if (q0 + l31 < p.N && s * 32 + lh * 16 < dbytes) qf[s] = *(const u32x4*)(qp + (size_t)(q0 + l31) * rowb + s * 32 + lh * 16);
        else qf[s] = u32x4{0, 0, 0, 0};
    }

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int cpr = dbytes / 16;                  // 16-byte chunks per K/V row
    const int cprp = dsteps * 2;                  // ... of the K tile incl. the zero half step
    const int nvec = KT * cprp;

    // The next tile's K / V vectors travel in registers while the current tile is being consumed: the global
    // latency hides behind the MFMAs instead of sitting between two barriers.
    // K: one 16-byte vector per (key, chunk), stored as it comes.  V: a thread takes FOUR consecutive keys of one
    // chunk and writes them transposed, 4 keys (8 or 16 bytes) per V^T row at a time -- the per-element transposing
    // stores of a single key were 16-way bank conflicted (every chunk's rows start on the same bank).
    constexpr int NV = DMAX * ESZ / 64;           // K vectors per thread per tile (upper bound)
    constexpr int NQ = (DMAX * ESZ + 255) / 256;  // V key-quads per thread per tile: 16 quads x (d*ESZ/16) chunks / 256
    u32x4 kreg[NV], vreg[NQ][4];
    const int nquad = (KT / 4) * cpr;
    // tile-invariant staging geometry of this thread (the divisions by the runtime chunk counts happen once, not per tile)
    int k_key[NV], k_goff[NV], k_loff[NV], v_key0[NQ], v_goff[NQ], v_loff[NQ];   // key < 0: nothing to do
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * AT_THREADS, key = v / cprp, ch = v - key * cprp;
        k_key[i] = v < nvec ? (ch < cpr ? key : key | 0x40000000) : -1;             // bit 30: the zero half-step chunk
        k_goff[i] = ch * 16;
        k_loff[i] = key * KS + ch * 16;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int v = tid + i * AT_THREADS, kq = v / cpr, ch = v - kq * cpr;
        v_key0[i] = v < nquad ? 4 * kq : -1;
        v_goff[i] = ch * 16;
        v_loff[i] = (ESZ == 2) ? (ch * 8) * VS + kq * 8 : (ch * 4) * VS + kq * 16;
    }
    auto fetch = [&](int k0) {
        const bool edge = k0 + KT > p.N;              // only the last tile of a short sequence needs the row checks
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            kreg[i] = u32x4{0, 0, 0, 0};
            const int key = k_key[i];
            if (key >= 0 && !(key & 0x40000000) && (!edge || k0 + key < p.N))
                kreg[i] = *(const u32x4*)(kp + (size_t)(k0 + key) * rowb + k_goff[i]);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = v_key0[i] + j;
                vreg[i][j] = (v_key0[i] >= 0 && (!edge || k0 + key < p.N)) ? *(const u32x4*)(vp + (size_t)(k0 + key) * rowb + v_goff[i])
                                                                            : u32x4{0, 0, 0, 0};
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (k_key[i] >= 0) *(u32x4*)(sK + k_loff[i]) = kreg[i];          // incl. the zero half-step chunk
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (v_key0[i] < 0) continue;
            char* dst = sV + v_loff[i];
            if (ESZ == 2) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {          // rows d = 8 ch + 2w, 2w + 1 get keys 4kq .. 4kq+3 (8 bytes each)
                    // v_perm_b32 picks the low / high halves of two dwords: {hi(a), hi(b)} <- 0x07060302, lo <- 0x05040100
                    const unsigned lo01 = __builtin_amdgcn_perm(vreg[i][1][w], vreg[i][0][w], 0x05040100u);
                    const unsigned lo23 = __builtin_amdgcn_perm(vreg[i][3][w], vreg[i][2][w], 0x05040100u);
                    const unsigned hi01 = __builtin_amdgcn_perm(vreg[i][1][w], vreg[i][0][w], 0x07060302u);
                    const unsigned hi23 = __builtin_amdgcn_perm(vreg[i][3][w], vreg[i][2][w], 0x07060302u);
                    *(u32x2*)(dst + (2 * w) * VS) = u32x2{lo01, lo23};
                    *(u32x2*)(dst + (2 * w + 1) * VS) = u32x2{hi01, hi23};
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    *(u32x4*)(dst + e * VS) = u32x4{vreg[i][0][e], vreg[i][1][e], vreg[i][2][e], vreg[i][3][e]};
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < p.n_valid; k0 += KT) {
        __syncthreads();                          // previous tile fully consumed
        commit();
        __syncthreads();
        if (k0 + KT < p.n_valid) fetch(k0 + KT);
        if (!active) continue;

        // ---- S^T = K Q^T for the two 32-key blocks
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            st[kb] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < QSTEPS; ++s) {
                if (s == 0 || s < dsteps) {                 // step 0 always exists: its C operand folds to the constant 0
                    const u32x4 kf = *(const u32x4*)(sK + (kb * 32 + l31) * KS + s * 32 + lh * 16);
                    if (ESZ == 2) {
                        if constexpr (ESZ == 2) st[kb] = mma16<T>(kf, qf[s], st[kb]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            st[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(kf[j]), __uint_as_float(qf[s][j]),
                                                                          st[kb], 0, 0, 0);
                    }
                }
            }
        }
        // ---- online softmax over this tile's 64 keys (32 in this lane, 32 in lane^32)
        if (k0 + KT > p.n_valid) {                          // only the last tile can hold padded keys (wave-uniform)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= p.n_valid) st[kb][r] = -INFINITY;
                }
        }
        float sc = p.scale_log2e;
        if (p.bias) {                                       // relative-position bias (+ shift mask): scale now, add, and
            const int query = q0 + l31;                     // continue with a unit scale
            const float* bp = p.bias + ((size_t)((b % p.bias_mod) * p.heads + hd) * p.N + (query < p.N ? query : 0)) * p.N;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    st[kb][r] = fmaf(st[kb][r], sc, key < p.N ? bp[key] : 0.f);
                }
            sc = 1.0f;
        }
        float mx = -INFINITY;                               // max of the RAW scores; the scale (> 0) is applied once
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * sc;
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // exp2(-inf) = 0 on the first tile
        float ls = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {                  // one FMA + one v_exp_f32 per score
                st[kb][r] = __builtin_amdgcn_exp2f(fmaf(st[kb][r], sc, -m_new));
                ls += st[kb][r];
            }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

        // ---- O^T += V^T P^T ; P^T registers are the B operand as they stand
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (ESZ == 2) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 pf;
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        if constexpr (ESZ == 2) pf[w] = pack2<T>(st[kb][8 * s2 + 2 * w], st[kb][8 * s2 + 2 * w + 1]);
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        const char* vr = sV + (t * 32 + l31) * VS + (kb * 32 + 16 * s2 + 4 * lh) * 2;
                        const u32x2 lo = *(const u32x2*)vr;            // keys +0..3
                        const u32x2 hi = *(const u32x2*)(vr + 16);     // keys +8..11
                        const u32x4 vf = u32x4{lo[0], lo[1], hi[0], hi[1]};
                        if constexpr (ESZ == 2) o[t] = mma16<T>(vf, pf, o[t]);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        const u32x4 vf = *(const u32x4*)(sV + (t * 32 + l31) * VS + (kb * 32 + 8 * g + 4 * lh) * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(vf[j]), st[kb][4 * g + j], o[t], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (!active || q0 + l31 >= p.N) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    T* orow = (T*)p.out + ((size_t)b * p.N + q0 + l31) * (size_t)(p.heads * d) + hd * d;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int di = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (di < d) Elt<T>::st(orow + di, o[t][r] * inv);
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Second-generation kernel for 16-bit storage without a score bias (every attention of the two UNet lineages and of the ViT
// victims).  The first kernel is VALU-bound on the softmax -- per score one FMA (scale, subtract the row max), one v_exp, one add
// (row sum), one max, half a pack: ~22 issue cycles per 64 scores against 6..12 MFMAs per 64-key tile -- most of all at the small
// head widths of the class-conditional UNet (d = 16 at N = 65 536: 86 % of that network's FLOPs).  Here
//   * q is pre-multiplied by log2(e) / sqrt(d) once, and the running row maximum enters the score MFMA as its INITIAL accumulator
//     (-m), so a score leaves the matrix core ready for v_exp: no per-score FMA;
//   * the maximum lags: a tile is rescaled only when some score exceeds the running maximum by more than 2^8 (T13 of the CDNA
//     guide; P stays <= 256, exact in the f32 accumulators, 8 significant bits in the 16-bit P operand either way);
//   * where the head width leaves spare rows in the 32-row V^T tile (d % 32 != 0), one of them is all ones and the row sum comes
//     out of the P.V product, rescaling included: no per-score add;
//   * keys are staged SUB 64-key tiles at a time: one barrier pair per 64 * SUB keys instead of per 64.
// Exact softmax algebra otherwise (online rescaling, masked padding keys).  FLOPs per launch: 4 * B * heads * N^2 * d.
template <typename T, int DT, int SUB, bool ONES>
__global__ void __launch_bounds__(AT_THREADS, 2)
attn2_kernel(const AttnP p) {
    constexpr int ESZ = 2;
    constexpr int KTS = KT * SUB;                 // keys staged per barrier pair
    constexpr int DMAX = DT * 32;
    constexpr int KS = DMAX * ESZ + 16;           // K tile row stride (bytes), padded
    constexpr int VS = KTS * ESZ + 8;             // V^T tile row stride (bytes)
    constexpr int QSTEPS = DMAX * ESZ / 32;
    constexpr float THR = 8.0f;
    extern __shared__ __attribute__((aligned(16))) char sm[];     // KTS*KS + DMAX*VS bytes
    char* sK = sm;
    char* sV = sm + KTS * KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool active = q0 < p.N;
    const int d = p.d;
    const int dbytes = d * ESZ;
    const int dsteps = (dbytes + 31) / 32;
    const size_t rowb = (size_t)p.ld * ESZ;
    const char* base = p.qkv + (size_t)b * p.N * rowb;
    const char* qp = base + (size_t)(p.q_off + hd * p.head_stride) * ESZ;
    const char* kp = base + (size_t)(p.k_off + hd * p.head_stride) * ESZ;
    const char* vp = base + (size_t)(p.v_off + hd * p.head_stride) * ESZ;
    constexpr bool ones_row = ONES;               // (d & 31) != 0: V^T row d of the last tile is all ones: O^T row d = the softmax denominator
    // (a template parameter since round 3: as a run-time flag it left a conditional add behind every v_exp)

    // Q fragments, pre-scaled: lane (query l31, half lh) holds bytes [32*s + 16*lh, +16) of its query row
    u32x4 qf[QSTEPS];
#pragma unroll
    for (int s = 0; s < QSTEPS; ++s) {
        qf[s] = u32x4{0, 0, 0, 0};
        if (q0 + l31 < p.N && s * 32 + lh * 16 < dbytes) {
            const u32x4 raw = *(const u32x4*)(qp + (size_t)(q0 + l31) * rowb + s * 32 + lh * 16);
            float f[8];
            unpack16<T>(raw, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] *= p.scale_log2e;
            qf[s] = pack16<T>(f);
        }
    }
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;
    f32x16 negm;                                   // -m_run in every register: the C operand of the score MFMA
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    bool first = true;

    const int cpr = dbytes / 16, cprp = dsteps * 2;
    const int nvec = KTS * cprp;
    constexpr int NV = (KTS * DMAX * ESZ / 16 + AT_THREADS - 1) / AT_THREADS;       // K vectors per thread per stage
    constexpr int NQ = ((KTS / 4) * (DMAX * ESZ / 16) + AT_THREADS - 1) / AT_THREADS; // V key-quads per thread per stage
    u32x4 kreg[NV], vreg[NQ][4];
    const int nquad = (KTS / 4) * cpr;
    int k_key[NV], k_goff[NV], k_loff[NV], v_key0[NQ], v_goff[NQ], v_loff[NQ];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * AT_THREADS, key = v / cprp, ch = v - key * cprp;
        k_key[i] = v < nvec ? (ch < cpr ? key : key | 0x40000000) : -1;
        k_goff[i] = ch * 16;
        k_loff[i] = key * KS + ch * 16;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int v = tid + i * AT_THREADS, kq = v / cpr, ch = v - kq * cpr;
        v_key0[i] = v < nquad ? 4 * kq : -1;
        v_goff[i] = ch * 16;
        v_loff[i] = (ch * 8) * VS + kq * 8;
    }
    // the commit below never touches V^T rows >= d: row d becomes the ones row, the rows behind it zeros (once, here) -- their
    // O^T rows are never stored, but uninitialised LDS could feed NaN bit patterns through the P.V MFMA for nothing
    for (int v = tid; v < (DMAX - d) * (KTS / 2); v += AT_THREADS) {
        const int row = d + v / (KTS / 2), kp2 = v - (row - d) * (KTS / 2);
        unsigned fill = 0u;
        if (ones_row && row == d) { T t1; Elt<T>::st(&t1, 1.0f); fill = (unsigned)t1.v * 0x10001u; }
        *(unsigned*)(sV + (size_t)row * VS + kp2 * 4) = fill;
    }
    auto fetch = [&](int k0) {
        const bool edge = k0 + KTS > p.N;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            kreg[i] = u32x4{0, 0, 0, 0};
            const int key = k_key[i];
            if (key >= 0 && !(key & 0x40000000) && (!edge || k0 + key < p.N))
                kreg[i] = *(const u32x4*)(kp + (size_t)(k0 + key) * rowb + k_goff[i]);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = v_key0[i] + j;
                vreg[i][j] = (v_key0[i] >= 0 && (!edge || k0 + key < p.N)) ? *(const u32x4*)(vp + (size_t)(k0 + key) * rowb + v_goff[i])
                                                                            : u32x4{0, 0, 0, 0};
            }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (k_key[i] >= 0) *(u32x4*)(sK + k_loff[i]) = kreg[i];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (v_key0[i] < 0) continue;
            char* dst = sV + v_loff[i];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const unsigned lo01 = __builtin_amdgcn_perm(vreg[i][1][w], vreg[i][0][w], 0x05040100u);
                const unsigned lo23 = __builtin_amdgcn_perm(vreg[i][3][w], vreg[i][2][w], 0x05040100u);
                const unsigned hi01 = __builtin_amdgcn_perm(vreg[i][1][w], vreg[i][0][w], 0x07060302u);
                const unsigned hi23 = __builtin_amdgcn_perm(vreg[i][3][w], vreg[i][2][w], 0x07060302u);
                *(u32x2*)(dst + (2 * w) * VS) = u32x2{lo01, lo23};
                *(u32x2*)(dst + (2 * w + 1) * VS) = u32x2{hi01, hi23};
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < p.n_valid; k0 += KTS) {
        __syncthreads();
        commit();
        __syncthreads();
        if (k0 + KTS < p.n_valid) fetch(k0 + KTS);
        if (!active) continue;
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            const int kt0 = k0 + sub * KT;
            if (kt0 >= p.n_valid) break;                       // wave-uniform
            // ---- S'^T = K (s q)^T - m for the two 32-key blocks (m = 0 before the first tile)
            // (the first k-step takes the block of -m registers as its C operand and writes the scores elsewhere: no 32 moves per tile
            // to initialise the accumulators -- a quarter of the per-tile VALU work at d = 16)
            // (narrow heads only: at d = 64 the 16 extra registers cost a wave per SIMD and 4.6 %; +1-3 % at d = 16 / 32, round 3)
            f32x16 st[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const u32x4 kf0 = *(const u32x4*)(sK + (sub * KT + kb * 32 + l31) * KS + lh * 16);
                if constexpr (DT == 1) {
                    st[kb] = mma16<T>(kf0, qf[0], negm);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[kb][r] = -m_run;
                    st[kb] = mma16<T>(kf0, qf[0], st[kb]);
                }
#pragma unroll
                for (int s = 1; s < QSTEPS; ++s)
                    if (s < dsteps) {
                        const u32x4 kf = *(const u32x4*)(sK + (sub * KT + kb * 32 + l31) * KS + s * 32 + lh * 16);
                        st[kb] = mma16<T>(kf, qf[s], st[kb]);
                    }
            }
            if (kt0 + KT > p.n_valid) {                          // only the last tile can hold padded keys (wave-uniform)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= p.n_valid) st[kb][r] = -INFINITY;
            }
            // one v_max3_f32 per two scores, four independent chains (a single running maximum is a 16-deep dependent chain).  This
            // file is compiled with -fno-honor-nans (Makefile): with IEEE fmaxf the compiler canonicalises every operand first, 49
            // v_max_f32 per tile instead of 16 v_max3_f32; a NaN score would poison its row either way.
            float mxp[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mxp[(r >> 1) & 3] = fmaxf(mxp[(r >> 1) & 3], fmaxf(st[kb][r], st[kb][r + 1]));
            float mx = fmaxf(fmaxf(mxp[0], mxp[1]), fmaxf(mxp[2], mxp[3]));
            // (the two lane halves of a query hold different keys; their maxima are only combined when a rescale happens at all --
            // the decision itself is wave-uniform through __any -- so the cross-half exchange leaves the per-tile dependency chain)
            if (first || __any(mx > THR)) {                    // wave-uniform: raise the running maximum, rescale what was accumulated
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float up = first ? mx : fmaxf(mx, 0.f);
                const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-up);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[kb][r] -= up;
                l_run *= alpha;
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
                m_run += up;
                if constexpr (DT == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) negm[r] = -m_run;
                }
                first = false;
            }
            float ls = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[kb][r] = __builtin_amdgcn_exp2f(st[kb][r]);
                    if (!ones_row) ls += st[kb][r];
                }
            l_run += ls;
            // ---- O^T += V^T P^T ; P^T registers are the B operand as they stand
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 pf;
#pragma unroll
                    for (int w = 0; w < 4; ++w) pf[w] = pack2<T>(st[kb][8 * s2 + 2 * w], st[kb][8 * s2 + 2 * w + 1]);
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        const char* vr = sV + (t * 32 + l31) * VS + (sub * KT + kb * 32 + 16 * s2 + 4 * lh) * 2;
                        const u32x2 lo = *(const u32x2*)vr;
                        const u32x2 hi = *(const u32x2*)(vr + 16);
                        o[t] = mma16<T>(u32x4{lo[0], lo[1], hi[0], hi[1]}, pf, o[t]);
                    }
                }
        }
    }
    if (!active || q0 + l31 >= p.N) return;
    float l_tot;
    if (ones_row) {                                            // row d of O^T: register r of lane half lh with (r&3) + 8(r>>2) + 4 lh = d % 32
        const int dr = d & 31, t = d >> 5;
        float v = 0.f;
#pragma unroll
        for (int tt = 0; tt < DT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (tt == t && (r & 3) + 8 * (r >> 2) + 4 * lh == dr) v = o[tt][r];
        l_tot = v + __shfl_xor(v, 32);                          // only one lane half holds the row: the other contributes 0
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32);
    }
    const float inv = 1.0f / l_tot;
    T* orow = (T*)p.out + ((size_t)b * p.N + q0 + l31) * (size_t)(p.heads * d) + hd * d;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int di = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (di < d) Elt<T>::st(orow + di, o[t][r] * inv);
        }
}

template <typename T, int DT, int SUB, bool ONES>
static int attn2_launch_o(const AttnP& p, hipStream_t st) {
    constexpr size_t lds = (size_t)KT * SUB * (DT * 32 * 2 + 16) + (size_t)DT * 32 * (KT * SUB * 2 + 8);
    static bool attr_set = false;
    if (!attr_set) {
        ADVS_HIP(hipFuncSetAttribute((const void*)attn2_kernel<T, DT, SUB, ONES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid(cdiv(p.N, 128), p.heads, p.B);
    attn2_kernel<T, DT, SUB, ONES><<<grid, AT_THREADS, lds, st>>>(p);
    ADVS_CHECK_LAUNCH("attention");
    return ADVS_OK;
}
template <typename T, int DT, int SUB>
static int attn2_launch(const AttnP& p, hipStream_t st) {
    return (p.d & 31) ? attn2_launch_o<T, DT, SUB, true>(p, st) : attn2_launch_o<T, DT, SUB, false>(p, st);
}

template <typename T, int DT>
static int attn_launch_dt(const AttnP& p, hipStream_t st) {
    constexpr int ESZ = AMma<T>::ESZ;
    constexpr size_t lds = (size_t)KT * (DT * 32 * ESZ + 16) + (size_t)DT * 32 * (KT * ESZ + (ESZ == 2 ? 8 : 16));
    static bool attr_set = false;                 // opt in once per instantiation (> 64 KiB for f32, d = 128)
    if (!attr_set) {
        ADVS_HIP(hipFuncSetAttribute((const void*)attn_kernel<T, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid(cdiv(p.N, 128), p.heads, p.B);
    attn_kernel<T, DT><<<grid, AT_THREADS, lds, st>>>(p);
    ADVS_CHECK_LAUNCH("attention");
    return ADVS_OK;
}

template <typename T>
static int attn_launch(const AttnP& p, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
#ifdef ADVS_DIAG
        static const bool old_kernel = getenv("ADVS_ATTN_V1") != nullptr;      // A/B knob for tools/ (diagnostic builds only)
#else
        constexpr bool old_kernel = false;
#endif
        if (!p.bias && p.d <= 64 && !old_kernel) {
            if (p.d <= 32) return attn2_launch<T, 1, 4>(p, st);
            return attn2_launch<T, 2, 2>(p, st);
        }
    }
    switch ((p.d + 31) / 32) {
        case 1: return attn_launch_dt<T, 1>(p, st);
        case 2: return attn_launch_dt<T, 2>(p, st);
        case 3: return attn_launch_dt<T, 3>(p, st);
        default: return attn_launch_dt<T, 4>(p, st);
    }
}

static int attention_impl(const void* qkv, void* out, int b, int n, int n_valid, int heads, int d, int ld,
                          int q_off, int k_off, int v_off, int head_stride, int dtype, void* stream,
                          const float* bias = nullptr, int bias_mod = 1) {
    ADVS_REQUIRE(qkv && out && b > 0 && n > 0 && heads > 0 && d > 0, "attention: bad args");
    ADVS_REQUIRE(n_valid > 0 && n_valid <= n, "attention: n_valid=%d out of range", n_valid);
    const int vec = dtype == ADVS_F32 ? 4 : 8;
    ADVS_REQUIRE(d % vec == 0 && d <= 128, "attention: d=%d must be a multiple of %d and <= 128", d, vec);
    ADVS_REQUIRE(ld % vec == 0 && q_off % vec == 0 && k_off % vec == 0 && v_off % vec == 0 && head_stride % vec == 0,
                 "attention: offsets must keep 16-byte alignment");
    AttnP p;
    p.qkv = (const char*)qkv; p.out = (char*)out;
    p.B = b; p.N = n; p.heads = heads; p.d = d; p.ld = ld;
    p.q_off = q_off; p.k_off = k_off; p.v_off = v_off; p.head_stride = head_stride; p.n_valid = n_valid;
    p.scale_log2e = (float)(1.4426950408889634 / sqrt((double)d));
    p.bias = bias; p.bias_mod = bias_mod > 0 ? bias_mod : 1;
    ADVS_SWITCH_T(dtype, return attn_launch<T>(p, (hipStream_t)stream));
    return ADVS_ERR_ARG;                    // not reached
}

extern "C" int advs_attention(const void* qkv, void* out, int b, int n, int heads, int d, int ld,
                              int q_off, int k_off, int v_off, int head_stride, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_attention: unknown dtype code %d", dtype);
    return attention_impl(qkv, out, b, n, n, heads, d, ld, q_off, k_off, v_off, head_stride, dtype, stream);
}
// Same with the token axis padded (victims.py: to a multiple of 16): only the first n_valid tokens are real keys
// (ViT: 197 tokens in rows of 208); padded query rows produce values nobody reads.
extern "C" int advs_attention_masked(const void* qkv, void* out, int b, int n, int n_valid, int heads, int d, int ld,
                                     int q_off, int k_off, int v_off, int head_stride, int dtype, void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_attention_masked: unknown dtype code %d", dtype);
    return attention_impl(qkv, out, b, n, n_valid, heads, d, ld, q_off, k_off, v_off, head_stride, dtype, stream);
}

// Same with an additive score bias: softmax(q k^T / sqrt(d) + bias) v, bias [bias_mod][heads][n][n] f32 GIVEN IN
// UNITS OF log2(e) (bias * 1.442695...), sequence i using block i % bias_mod -- Swin's relative position bias plus
// the shifted-window mask of window position i % nW (WindowAttention.forward, timm swin_transformer.py).
extern "C" int advs_attention_bias(const void* qkv, void* out, const float* bias_log2e, int bias_mod, int b, int n,
                                   int heads, int d, int ld, int q_off, int k_off, int v_off, int head_stride, int dtype,
                                   void* stream) {
    ADVS_REQUIRE(dtype_ok(dtype), "advs_attention_bias: unknown dtype code %d", dtype);
    ADVS_REQUIRE(bias_log2e && bias_mod > 0, "attention_bias: bias is required");
    return attention_impl(qkv, out, b, n, n, heads, d, ld, q_off, k_off, v_off, head_stride, dtype, stream, bias_log2e, bias_mod);
}
