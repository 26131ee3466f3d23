// Attention backward on the matrix cores, 16-bit dtypes, head width <= 64 (the ViT victim of the gradient attack,
// victims.py: timm vision_transformer.Attention with fused_attn; 197 tokens in rows of 208, 12 heads of 64).
//
//   S = (s q) k^T,  P = softmax_rows(S),  O = P v.   Given dO:   D_i = dO_i . O_i,   dP = dO v^T,   dS = P o (dP - D),
//   dq = s dS k,    dk = s dS^T q,    dv = P^T dO                                    (s = 1 / sqrt d).
//
// Two kernels, both recompute the scores (nothing of the forward is kept but O):
//   dq kernel  -- a wave owns 32 queries (q, dO fragments in registers), keys stream through the LDS 64 at a time.  Sweep 1: row
//                 maximum and sum -> LSE (log2 domain, written to scratch with D).  Sweep 2: S^T = K q'^T - LSE and
//                 dP^T = V dO^T - D as two MFMA chains whose accumulators START at -LSE / -D (lane-uniform: a lane's column is
//                 its query), dS^T = exp2(S^T) o dP^T, and dq^T += K^T dS^T with dS^T's accumulator registers used as the B
//                 operand as they stand (the key order inside a K = 16 step is permuted identically in the K^T tile, as
//                 attention.hip does for V^T).
//   dkv kernel -- a wave owns 32 keys (k', v fragments in registers), queries stream 64 at a time with their LSE / D:
//                 S = q' k^T - LSE and dP = dO v^T - D start from the per-row values (four ds_read_b128 broadcasts), then
//                 dv^T += dO^T P and dk^T += q'^T dS.
// q is scaled by s log2(e) and rounded BEFORE both kernels use it (as the forward does), so the probabilities the two kernels
// rebuild are the same numbers; dk is rescaled by ln 2 at the end.  Padding: only the first n_valid tokens are keys and
// queries; staged rows beyond them are zero-filled, their outputs are written as zeros.
// FLOPs per launch pair: 18 N^2 d per (image, head) (9 N x N x d products).  f32 inputs keep the VALU kernels of vit_grad.hip.
#include "common.h"
#include <math.h>

#define AB_THREADS 256
#define AB_ROWS 64                       // streamed rows per stage
#define AB_VS (AB_ROWS * 2 + 8)          // transposed tile row stride (bytes)

struct AttnBwdP {
    const char* qkv; const char* o; const char* dO; char* dqkv;
    float* lse; float* dsum;
    const float* bias; int bias_mod;          // optional additive score bias [bias_mod][heads][N][N] f32 in units of log2(e) (Swin)
    int B, N, n_valid, heads, d, ld, q_off, k_off, v_off, head_stride;
    float scale, scale_log2e;
};

template <typename T>
__device__ __forceinline__ u32x4 scaled16(const u32x4& raw, float mul) {
    float f[8];
    unpack16<T>(raw, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= mul;
    return pack16<T>(f);
}

// 64 rows [r0, r0 + 64) of a row-major matrix -> LDS rows of DMAX * 2 + 16 bytes, zero beyond rlimit / dbytes
template <typename T, int DMAX, bool SCALE>
__device__ __forceinline__ void stage_rows(char* sR, const char* src, size_t rowb, int r0, int rlimit, int dbytes, int tid, float mul) {
    constexpr int CPR = DMAX * 2 / 16, KS = DMAX * 2 + 16;
#pragma unroll
    for (int i = 0; i < AB_ROWS * CPR / AB_THREADS; ++i) {
        const int v = tid + i * AB_THREADS, row = v / CPR, ch = v - row * CPR;
        u32x4 x{0, 0, 0, 0};
        if (r0 + row < rlimit && ch * 16 < dbytes) {
            x = *(const u32x4*)(src + (size_t)(r0 + row) * rowb + ch * 16);
            if (SCALE) x = scaled16<T>(x, mul);
        }
        *(u32x4*)(sR + row * KS + ch * 16) = x;
    }
}
// the same rows transposed: LDS row = column of the matrix, 64 row-values along it in natural order
template <typename T, bool SCALE>
__device__ __forceinline__ void stage_transposed(char* sT, const char* src, size_t rowb, int r0, int rlimit, int dbytes, int tid, float mul) {
    const int cpr = dbytes / 16, nquad = (AB_ROWS / 4) * cpr;
    if (tid >= nquad) return;
    const int kq = tid / cpr, ch = tid - kq * cpr;
    u32x4 x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = r0 + 4 * kq + j;
        x[j] = u32x4{0, 0, 0, 0};
        if (row < rlimit) {
            x[j] = *(const u32x4*)(src + (size_t)row * rowb + ch * 16);
            if (SCALE) x[j] = scaled16<T>(x[j], mul);
        }
    }
    char* dst = sT + (ch * 8) * AB_VS + kq * 8;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned lo01 = __builtin_amdgcn_perm(x[1][w], x[0][w], 0x05040100u);
        const unsigned lo23 = __builtin_amdgcn_perm(x[3][w], x[2][w], 0x05040100u);
        const unsigned hi01 = __builtin_amdgcn_perm(x[1][w], x[0][w], 0x07060302u);
        const unsigned hi23 = __builtin_amdgcn_perm(x[3][w], x[2][w], 0x07060302u);
        *(u32x2*)(dst + (2 * w) * AB_VS) = u32x2{lo01, lo23};
        *(u32x2*)(dst + (2 * w + 1) * AB_VS) = u32x2{hi01, hi23};
    }
}
// A fragment of a transposed tile: output rows t * 32 + l31, the 8 K-slots of step (blk, s2) in accumulator-register order
__device__ __forceinline__ u32x4 frag_t(const char* sT, int t, int blk, int s2, int l31, int lh) {
    const char* vr = sT + (t * 32 + l31) * AB_VS + (blk * 32 + 16 * s2 + 4 * lh) * 2;
    const u32x2 lo = *(const u32x2*)vr;
    const u32x2 hi = *(const u32x2*)(vr + 16);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
}

// ---------------------------------------------------------------------------------------------- dq, LSE, D
template <typename T, int DT>
__global__ void __launch_bounds__(AB_THREADS, 2)
attn_bwd_dq_kernel(const AttnBwdP p) {
    constexpr int DMAX = DT * 32, KS = DMAX * 2 + 16, QSTEPS = DMAX * 2 / 32;
    __shared__ __attribute__((aligned(16))) char sK[AB_ROWS * KS];
    __shared__ __attribute__((aligned(16))) char sV[AB_ROWS * KS];
    __shared__ __attribute__((aligned(16))) char sKT[DMAX * AB_VS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32, qi = q0 + l31;
    const bool active = q0 < p.n_valid;                        // wave-uniform: the wave holds at least one real query
    const bool qvalid = qi < p.n_valid;
    const int d = p.d, dbytes = d * 2, dsteps = (dbytes + 31) / 32;
    const size_t rowb = (size_t)p.ld * 2, crow = (size_t)p.heads * d * 2;
    const char* base = p.qkv + (size_t)b * p.N * rowb;
    const char* qp = base + (size_t)(p.q_off + hd * p.head_stride) * 2;
    const char* kp = base + (size_t)(p.k_off + hd * p.head_stride) * 2;
    const char* vp = base + (size_t)(p.v_off + hd * p.head_stride) * 2;
    const char* gp = p.dO + (size_t)b * p.N * crow + (size_t)hd * dbytes;
    const char* op = p.o + (size_t)b * p.N * crow + (size_t)hd * dbytes;
    // this lane's bias row (its query), clamped so that padding lanes read something valid
    const float* brow = p.bias ? p.bias + (((size_t)(b % p.bias_mod) * p.heads + hd) * p.N + (qi < p.N ? qi : p.N - 1)) * (size_t)p.N : nullptr;
    auto add_bias_t = [&](f32x16& st, int key0) {                  // S'^T tile: registers = keys key0 + row(r), column = this lane's query
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int kk = key0 + 8 * a + 4 * lh;
#pragma unroll
            for (int c = 0; c < 4; ++c) st[4 * a + c] += brow[kk + c < p.N ? kk + c : p.N - 1];
        }
    };

    u32x4 qf[QSTEPS], gf[QSTEPS];
    float Dn = 0.f;
#pragma unroll
    for (int s = 0; s < QSTEPS; ++s) {
        qf[s] = u32x4{0, 0, 0, 0};
        gf[s] = u32x4{0, 0, 0, 0};
        if (qvalid && s * 32 + lh * 16 < dbytes) {
            qf[s] = scaled16<T>(*(const u32x4*)(qp + (size_t)qi * rowb + s * 32 + lh * 16), p.scale_log2e);
            gf[s] = *(const u32x4*)(gp + (size_t)qi * crow + s * 32 + lh * 16);
            const u32x4 orw = *(const u32x4*)(op + (size_t)qi * crow + s * 32 + lh * 16);
            float g[8], o[8];
            unpack16<T>(gf[s], g);
            unpack16<T>(orw, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) Dn = fmaf(g[j], o[j], Dn);
        }
    }
    Dn += __shfl_xor(Dn, 32);

    // ---- sweep 1: row maximum and sum of exp2(S') over the keys
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < p.n_valid; k0 += AB_ROWS) {
        __syncthreads();
        stage_rows<T, DMAX, false>(sK, kp, rowb, k0, p.n_valid, dbytes, tid, 1.f);
        __syncthreads();
        if (!active) continue;
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < QSTEPS; ++s)
                if (s == 0 || s < dsteps)
                    st[kb] = mma16<T>(*(const u32x4*)(sK + (kb * 32 + l31) * KS + s * 32 + lh * 16), qf[s], st[kb]);
            if (brow) add_bias_t(st[kb], k0 + kb * 32);
        }
        if (k0 + AB_ROWS > p.n_valid) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= p.n_valid) st[kb][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(mx, fmaxf(st[kb][r], st[kb][r + 1]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m2 = fmaxf(m, mx);
        float ls = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) ls += __builtin_amdgcn_exp2f(st[kb][r] - m2);
        l = l * __builtin_amdgcn_exp2f(m - m2) + ls;
        m = m2;
    }
    l += __shfl_xor(l, 32);
    const float lse = active ? m + __builtin_amdgcn_logf(l) : 0.f;           // v_log_f32 = log2

    // ---- sweep 2: dS^T and dq^T
    f32x16 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;
    for (int k0 = 0; k0 < p.n_valid; k0 += AB_ROWS) {
        __syncthreads();
        stage_rows<T, DMAX, false>(sK, kp, rowb, k0, p.n_valid, dbytes, tid, 1.f);
        stage_rows<T, DMAX, false>(sV, vp, rowb, k0, p.n_valid, dbytes, tid, 1.f);
        stage_transposed<T, false>(sKT, kp, rowb, k0, p.n_valid, dbytes, tid, 1.f);
        __syncthreads();
        if (!active) continue;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (k0 + kb * 32 >= p.n_valid) break;                 // wave-uniform
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = -lse; dp[r] = -Dn; }
#pragma unroll
            for (int s = 0; s < QSTEPS; ++s)
                if (s == 0 || s < dsteps) {
                    st = mma16<T>(*(const u32x4*)(sK + (kb * 32 + l31) * KS + s * 32 + lh * 16), qf[s], st);
                    dp = mma16<T>(*(const u32x4*)(sV + (kb * 32 + l31) * KS + s * 32 + lh * 16), gf[s], dp);
                }
            if (brow) add_bias_t(st, k0 + kb * 32);
            const bool edge = k0 + kb * 32 + 32 > p.n_valid;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pr = __builtin_amdgcn_exp2f(st[r]);
                if (edge && k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= p.n_valid) pr = 0.f;
                st[r] = pr * dp[r];                              // dS^T
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 df;
#pragma unroll
                for (int w = 0; w < 4; ++w) df[w] = pack2<T>(st[8 * s2 + 2 * w], st[8 * s2 + 2 * w + 1]);
#pragma unroll
                for (int t = 0; t < DT; ++t) dq[t] = mma16<T>(frag_t(sKT, t, kb, s2, l31, lh), df, dq[t]);
            }
        }
    }
    if (qi >= p.N) return;
    if (lh == 0) {
        const size_t si = ((size_t)b * p.heads + hd) * p.N + qi;
        p.lse[si] = qvalid ? lse : 0.f;
        p.dsum[si] = qvalid ? Dn : 0.f;
    }
    char* dqrow = p.dqkv + ((size_t)b * p.N + qi) * rowb + (size_t)(p.q_off + hd * p.head_stride) * 2;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int di = t * 32 + 8 * a + 4 * lh;
            if (di < d) {
                const float z = (active && qvalid) ? p.scale : 0.f;
                *(u32x2*)(dqrow + di * 2) = u32x2{pack2<T>(dq[t][4 * a] * z, dq[t][4 * a + 1] * z), pack2<T>(dq[t][4 * a + 2] * z, dq[t][4 * a + 3] * z)};
            }
        }
}

// ---------------------------------------------------------------------------------------------- dk, dv
template <typename T, int DT>
__global__ void __launch_bounds__(AB_THREADS, 2)
attn_bwd_dkv_kernel(const AttnBwdP p) {
    constexpr int DMAX = DT * 32, KS = DMAX * 2 + 16, QSTEPS = DMAX * 2 / 32;
    __shared__ __attribute__((aligned(16))) char sQ[AB_ROWS * KS];
    __shared__ __attribute__((aligned(16))) char sG[AB_ROWS * KS];
    __shared__ __attribute__((aligned(16))) char sQT[DMAX * AB_VS];
    __shared__ __attribute__((aligned(16))) char sGT[DMAX * AB_VS];
    __shared__ __attribute__((aligned(16))) float sL[AB_ROWS];
    __shared__ __attribute__((aligned(16))) float sD[AB_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int kw0 = blockIdx.x * 128 + wave * 32, kj = kw0 + l31;
    const bool active = kw0 < p.n_valid;
    const bool kvalid = kj < p.n_valid;
    const int d = p.d, dbytes = d * 2, dsteps = (dbytes + 31) / 32;
    const size_t rowb = (size_t)p.ld * 2, crow = (size_t)p.heads * d * 2;
    const char* base = p.qkv + (size_t)b * p.N * rowb;
    const char* qp = base + (size_t)(p.q_off + hd * p.head_stride) * 2;
    const char* kp = base + (size_t)(p.k_off + hd * p.head_stride) * 2;
    const char* vp = base + (size_t)(p.v_off + hd * p.head_stride) * 2;
    const char* gp = p.dO + (size_t)b * p.N * crow + (size_t)hd * dbytes;
    const float* lsep = p.lse + ((size_t)b * p.heads + hd) * p.N;
    const float* dsp = p.dsum + ((size_t)b * p.heads + hd) * p.N;
    // bias column of this lane's key: element (query i, key kj) at bcol[i * N]
    const float* bcol = p.bias ? p.bias + ((size_t)(b % p.bias_mod) * p.heads + hd) * (size_t)p.N * p.N + (kj < p.N ? kj : p.N - 1) : nullptr;

    u32x4 kf[QSTEPS], vf[QSTEPS];
#pragma unroll
    for (int s = 0; s < QSTEPS; ++s) {
        kf[s] = u32x4{0, 0, 0, 0};
        vf[s] = u32x4{0, 0, 0, 0};
        if (kvalid && s * 32 + lh * 16 < dbytes) {
            kf[s] = *(const u32x4*)(kp + (size_t)kj * rowb + s * 32 + lh * 16);
            vf[s] = *(const u32x4*)(vp + (size_t)kj * rowb + s * 32 + lh * 16);
        }
    }
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

    for (int i0 = 0; i0 < p.n_valid; i0 += AB_ROWS) {
        __syncthreads();
        stage_rows<T, DMAX, true>(sQ, qp, rowb, i0, p.n_valid, dbytes, tid, p.scale_log2e);
        stage_rows<T, DMAX, false>(sG, gp, crow, i0, p.n_valid, dbytes, tid, 1.f);
        stage_transposed<T, true>(sQT, qp, rowb, i0, p.n_valid, dbytes, tid, p.scale_log2e);
        stage_transposed<T, false>(sGT, gp, crow, i0, p.n_valid, dbytes, tid, 1.f);
        if (tid < AB_ROWS) {
            const int i = i0 + tid;
            sL[tid] = i < p.n_valid ? lsep[i] : 0.f;
            sD[tid] = i < p.n_valid ? dsp[i] : 0.f;
        }
        __syncthreads();
        if (!active) continue;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (i0 + qb * 32 >= p.n_valid) break;                 // wave-uniform
            f32x16 st, dp;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 L = *(const f32x4*)&sL[qb * 32 + 8 * a + 4 * lh];
                const f32x4 Dv = *(const f32x4*)&sD[qb * 32 + 8 * a + 4 * lh];
#pragma unroll
                for (int c = 0; c < 4; ++c) { st[4 * a + c] = -L[c]; dp[4 * a + c] = -Dv[c]; }
            }
#pragma unroll
            for (int s = 0; s < QSTEPS; ++s)
                if (s == 0 || s < dsteps) {
                    st = mma16<T>(*(const u32x4*)(sQ + (qb * 32 + l31) * KS + s * 32 + lh * 16), kf[s], st);
                    dp = mma16<T>(*(const u32x4*)(sG + (qb * 32 + l31) * KS + s * 32 + lh * 16), vf[s], dp);
                }
            if (bcol) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qq = i0 + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    st[r] += bcol[(size_t)(qq < p.N ? qq : p.N - 1) * p.N];
                }
            }
            const bool edge = i0 + qb * 32 + 32 > p.n_valid;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pr = __builtin_amdgcn_exp2f(st[r]);
                if (edge && i0 + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= p.n_valid) pr = 0.f;      // padding query
                st[r] = pr;
                dp[r] *= pr;                                      // dS
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 pf, df;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    pf[w] = pack2<T>(st[8 * s2 + 2 * w], st[8 * s2 + 2 * w + 1]);
                    df[w] = pack2<T>(dp[8 * s2 + 2 * w], dp[8 * s2 + 2 * w + 1]);
                }
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    dv[t] = mma16<T>(frag_t(sGT, t, qb, s2, l31, lh), pf, dv[t]);
                    dk[t] = mma16<T>(frag_t(sQT, t, qb, s2, l31, lh), df, dk[t]);
                }
            }
        }
    }
    if (kj >= p.N) return;
    char* dkrow = p.dqkv + ((size_t)b * p.N + kj) * rowb + (size_t)(p.k_off + hd * p.head_stride) * 2;
    char* dvrow = p.dqkv + ((size_t)b * p.N + kj) * rowb + (size_t)(p.v_off + hd * p.head_stride) * 2;
    const float ksc = (active && kvalid) ? 0.6931471805599453f : 0.f;    // q' = s log2(e) q:  s dS^T q = ln 2 * dS^T q'
    const float vsc = (active && kvalid) ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int di = t * 32 + 8 * a + 4 * lh;
            if (di < d) {
                *(u32x2*)(dkrow + di * 2) = u32x2{pack2<T>(dk[t][4 * a] * ksc, dk[t][4 * a + 1] * ksc), pack2<T>(dk[t][4 * a + 2] * ksc, dk[t][4 * a + 3] * ksc)};
                *(u32x2*)(dvrow + di * 2) = u32x2{pack2<T>(dv[t][4 * a] * vsc, dv[t][4 * a + 1] * vsc), pack2<T>(dv[t][4 * a + 2] * vsc, dv[t][4 * a + 3] * vsc)};
            }
        }
}

template <typename T, int DT>
static int attn_bwd_mfma_launch(const AttnBwdP& p, hipStream_t st) {
    const dim3 grid(cdiv(p.N, 128), p.heads, p.B);
    attn_bwd_dq_kernel<T, DT><<<grid, AB_THREADS, 0, st>>>(p);
    ADVS_CHECK_LAUNCH("attention_bwd (dq)");
    attn_bwd_dkv_kernel<T, DT><<<grid, AB_THREADS, 0, st>>>(p);
    ADVS_CHECK_LAUNCH("attention_bwd (dk, dv)");
    return ADVS_OK;
}

// Called by advs_attention_bwd (vit_grad.hip) for 16-bit dtypes; scratch holds 2 * b * heads * n floats here.
int attn_bwd_mfma(const void* qkv, const void* out, const void* d_out, void* d_qkv, void* scratch, int b, int n, int n_valid, int heads,
                  int d, int ld, int q_off, int k_off, int v_off, int head_stride, int dtype, hipStream_t st, const float* bias, int bias_mod) {
    AttnBwdP p;
    p.bias = bias; p.bias_mod = bias_mod > 0 ? bias_mod : 1;
    p.qkv = (const char*)qkv; p.o = (const char*)out; p.dO = (const char*)d_out; p.dqkv = (char*)d_qkv;
    p.lse = (float*)scratch; p.dsum = p.lse + (size_t)b * heads * n;
    p.B = b; p.N = n; p.n_valid = n_valid; p.heads = heads; p.d = d; p.ld = ld;
    p.q_off = q_off; p.k_off = k_off; p.v_off = v_off; p.head_stride = head_stride;
    p.scale = (float)(1.0 / sqrt((double)d));
    p.scale_log2e = (float)(1.4426950408889634 / sqrt((double)d));
    if (dtype == ADVS_BF16) return d <= 32 ? attn_bwd_mfma_launch<BF16, 1>(p, st) : attn_bwd_mfma_launch<BF16, 2>(p, st);
    return d <= 32 ? attn_bwd_mfma_launch<F16, 1>(p, st) : attn_bwd_mfma_launch<F16, 2>(p, st);
}
