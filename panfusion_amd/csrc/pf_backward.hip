// Training kernels for gfx950 (CDNA4): backward of the EPA block -- the trainable part of the denoiser next to the
// rank-4 LoRA (reference models/pano/modules.py:15-59 under autograd, models/modules/transformer.py:40-161; the
// reference wraps the block in its CheckpointFunction, transformer.py:77-127: the forward is recomputed in backward,
// which is what the host side of these entry points does as well).
//
//   attention backward   two MFMA kernels that recompute P from the forward's log-sum-exp:
//                          k_attn_bwd_dq   a wavefront owns 32 queries and walks the keys   -> dQ
//                          k_attn_bwd_dkv  a wavefront owns 32 keys and walks the queries   -> dK, dV
//                        both use the transposed-product layout of the forward kernel (pf_attention.hip): the owner
//                        index is the MFMA column = the lane, so per-owner scalars (lse, delta) are lane scalars in the
//                        first kernel and the probabilities feed the second product straight from the registers.
//   LayerNorm backward   one row per wavefront, per-block partial sums for gamma / beta (reduced by k_colsum)
//   GEGLU backward, column sums (bias gradients), gradient normalisation (amax -> power-of-two scale).
#include "pf_common.h"
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

namespace pf {

__device__ __forceinline__ float wave_sum_b(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- attention backward ------------------------------------------------------------------------------------------
struct AttnBwdParams {
    const unsigned short *q, *k, *v, *dout, *qt, *kt, *dot;
    unsigned short *dq, *dk, *dv;
    int H, nq, nk;
    int q_ld, k_ld, v_ld, do_ld, qt_ld, kt_ld, dot_ld, dq_ld, dk_ld, dv_ld;
    long q_bs, k_bs, v_bs, do_bs, qt_bs, kt_bs, dot_bs, dq_bs, dk_bs, dv_bs;
    float scale, scale_log2e;
    const float* bias; long bias_ld; const uint8_t* flags; int flags_ld;
    const float *lse, *delta;
    int qsplit;                         // > 1: the keys-stationary kernel splits the query range over qsplit blocks ...
    float *part_k, *part_v;             // ... which leave fp32 partial sums [qsplit][B][nk][H*D], added in order by k_attn_bwd_reduce
};

constexpr float LOG2E = 1.44269504088896340736f;

// A-operand fragment of a TRANSPOSED tensor ([rows = head dim][tokens contiguous]) for a product whose reduction runs
// over 32 tokens starting at t0: the k-slot order (lo = t0 + 16 s2 + 4 hi + 0..3, up = the same + 8) is the permutation
// in which the score registers of a lane enumerate the tokens (forward kernel, O^T += V^T P^T).
__device__ __forceinline__ u16x8 load_t_frag(const unsigned short* row, int t0, int s2, int hi) {
    const unsigned short* pp = row + t0 + 16 * s2 + 4 * hi;
    const u16x4 lo = *reinterpret_cast<const u16x4*>(pp);
    const u16x4 up = *reinterpret_cast<const u16x4*>(pp + 8);
    return u16x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
}

// the same with the tokens >= n_tok read as zero and nothing read past the row (ragged token counts)
__device__ __forceinline__ u16x8 load_t_frag_guard(const unsigned short* row, int t0, int s2, int hi, int n_tok) {
    u16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = t0 + 16 * s2 + 4 * hi + (j & 3) + 8 * (j >> 2);
        v[j] = t < n_tok ? row[t] : static_cast<unsigned short>(0);
    }
    return v;
}

template <typename T, int D, bool TAIL>
__global__ __launch_bounds__(256) void k_attn_bwd_dq(const AttnBwdParams p) {
    constexpr int KS = D / 16, DB = D / 32;
    typedef typename Mfma32<T>::frag frag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= p.nq) return;
    const int h = blockIdx.y;
    const long b = blockIdx.z;
    const int qrow = TAIL ? min(q0 + ql, p.nq - 1) : q0 + ql;
    const unsigned short* qp = p.q + b * p.q_bs + static_cast<long>(qrow) * p.q_ld + h * D;
    const unsigned short* dop = p.dout + b * p.do_bs + static_cast<long>(qrow) * p.do_ld + h * D;
    const unsigned short* kp = p.k + b * p.k_bs + h * D;
    const unsigned short* vp = p.v + b * p.v_bs + h * D;
    const unsigned short* ktp = p.kt + b * p.kt_bs + static_cast<long>(h) * D * p.kt_ld;

    frag qf[KS], dof[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(qp + 16 * s + 8 * hi));
        dof[s] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(dop + 16 * s + 8 * hi));
    }
    const long stat = (b * p.H + h) * p.nq + qrow;
    const float lse = p.lse[stat], delta = p.delta[stat];
    const float c2 = p.scale_log2e;

    f32x16 acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    const uint8_t* flag_row = p.flags ? p.flags + static_cast<long>(q0 >> 5) * p.flags_ld : nullptr;
    const float* bias_row = p.bias ? p.bias + static_cast<long>(qrow) * p.bias_ld : nullptr;
    const int nkt = (p.nk + 31) / 32;
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32;
        const int krow = TAIL ? min(k0 + ql, p.nk - 1) : k0 + ql;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u16x8 kf = *reinterpret_cast<const u16x8*>(kp + static_cast<long>(krow) * p.k_ld + 16 * ks + 8 * hi);
            const u16x8 vf = *reinterpret_cast<const u16x8*>(vp + static_cast<long>(krow) * p.v_ld + 16 * ks + 8 * hi);
            s = Mfma32<T>::run(__builtin_bit_cast(frag, kf), qf[ks], s);          // S^T  [key][query]
            dp = Mfma32<T>::run(__builtin_bit_cast(frag, vf), dof[ks], dp);       // dP^T [key][query]
        }
        float sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = s[r] * c2;
        if (flag_row && flag_row[kt]) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (TAIL && k0 + 8 * g + 4 * hi + 3 >= p.nk) continue;      // (nk % 4 == 0 with a bias: whole quads)
                const float4 bv = *reinterpret_cast<const float4*>(bias_row + k0 + 8 * g + 4 * hi);
                sv[4 * g + 0] += bv.x * LOG2E;
                sv[4 * g + 1] += bv.y * LOG2E;
                sv[4 * g + 2] += bv.z * LOG2E;
                sv[4 * g + 3] += bv.w * LOG2E;
            }
        }
        u16x8 pb[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float pr = exp2f(sv[r] - lse);
            if (TAIL && k0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.nk) pr = 0.f;
            pb[r >> 3][r & 7] = from_f32<T>(pr * (dp[r] - delta));
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const unsigned short* trow = ktp + static_cast<long>(d * 32 + ql) * p.kt_ld;
                const u16x8 a = TAIL ? load_t_frag_guard(trow, k0, s2, hi, p.nk) : load_t_frag(trow, k0, s2, hi);
                acc[d] = Mfma32<T>::run(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, pb[s2]), acc[d]);   // dQ^T += K^T dS^T
            }
    }
    if (TAIL && q0 + ql >= p.nq) return;
    unsigned short* op = p.dq + b * p.dq_bs + static_cast<long>(q0 + ql) * p.dq_ld + h * D;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u16x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(acc[d][4 * g + e] * p.scale);
            *reinterpret_cast<u16x4*>(op + d * 32 + 8 * g + 4 * hi) = w;
        }
}

template <typename T, int D, bool TAIL>
__global__ __launch_bounds__(256) void k_attn_bwd_dkv(const AttnBwdParams p) {
    constexpr int KS = D / 16, DB = D / 32;
    typedef typename Mfma32<T>::frag frag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kl = lane & 31, hi = lane >> 5;
    const int k0 = (blockIdx.x * 4 + wave) * 32;
    if (k0 >= p.nk) return;
    const int h = blockIdx.y;
    const long b = blockIdx.z;
    const int krow = TAIL ? min(k0 + kl, p.nk - 1) : k0 + kl;
    const unsigned short* kp = p.k + b * p.k_bs + static_cast<long>(krow) * p.k_ld + h * D;
    const unsigned short* vp = p.v + b * p.v_bs + static_cast<long>(krow) * p.v_ld + h * D;
    const unsigned short* qp = p.q + b * p.q_bs + h * D;
    const unsigned short* dop = p.dout + b * p.do_bs + h * D;
    const unsigned short* qtp = p.qt + b * p.qt_bs + static_cast<long>(h) * D * p.qt_ld;
    const unsigned short* dotp = p.dot + b * p.dot_bs + static_cast<long>(h) * D * p.dot_ld;

    frag kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        kf[s] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(kp + 16 * s + 8 * hi));
        vf[s] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(vp + 16 * s + 8 * hi));
    }
    const float c2 = p.scale_log2e;
    const float* lsep = p.lse + (b * p.H + h) * p.nq;
    const float* delp = p.delta + (b * p.H + h) * p.nq;

    f32x16 acc_k[DB], acc_v[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_k[d][r] = 0.f; acc_v[d][r] = 0.f; }

    const int kt = k0 >> 5;
    const int nqt = (p.nq + 31) / 32;
    for (int qt = 0; qt < nqt; ++qt) {
        const int q0 = qt * 32;
        const int qrow = TAIL ? min(q0 + kl, p.nq - 1) : q0 + kl;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u16x8 qf = *reinterpret_cast<const u16x8*>(qp + static_cast<long>(qrow) * p.q_ld + 16 * ks + 8 * hi);
            const u16x8 df = *reinterpret_cast<const u16x8*>(dop + static_cast<long>(qrow) * p.do_ld + 16 * ks + 8 * hi);
            s = Mfma32<T>::run(__builtin_bit_cast(frag, qf), kf[ks], s);           // S  [query][key]
            dp = Mfma32<T>::run(__builtin_bit_cast(frag, df), vf[ks], dp);         // dP [query][key]
        }
        // register r <-> query q0 + (r & 3) + 8 (r >> 2) + 4 hi; the lane's column is key k0 + kl
        float sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = s[r] * c2;
        if (p.flags && p.flags[static_cast<long>(qt) * p.flags_ld + kt]) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (!TAIL || (qi < p.nq && k0 + kl < p.nk)) sv[r] += p.bias[static_cast<long>(qi) * p.bias_ld + k0 + kl] * LOG2E;
            }
        }
        u16x8 pp[2], pd[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float lv[4], dv[4];
            if (TAIL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qi = min(q0 + 8 * g + 4 * hi + e, p.nq - 1);
                    lv[e] = lsep[qi];
                    dv[e] = delp[qi];
                }
            } else {
                const float4 l4 = *reinterpret_cast<const float4*>(lsep + q0 + 8 * g + 4 * hi);
                const float4 d4 = *reinterpret_cast<const float4*>(delp + q0 + 8 * g + 4 * hi);
                lv[0] = l4.x; lv[1] = l4.y; lv[2] = l4.z; lv[3] = l4.w;
                dv[0] = d4.x; dv[1] = d4.y; dv[2] = d4.z; dv[3] = d4.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                float pr = exp2f(sv[r] - lv[e]);
                if (TAIL && q0 + 8 * g + 4 * hi + e >= p.nq) pr = 0.f;
                pp[r >> 3][r & 7] = from_f32<T>(pr);
                pd[r >> 3][r & 7] = from_f32<T>(pr * (dp[r] - dv[e]));
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const unsigned short* rdo = dotp + static_cast<long>(d * 32 + kl) * p.dot_ld;
                const unsigned short* rq = qtp + static_cast<long>(d * 32 + kl) * p.qt_ld;
                const u16x8 a_do = TAIL ? load_t_frag_guard(rdo, q0, s2, hi, p.nq) : load_t_frag(rdo, q0, s2, hi);
                const u16x8 a_q = TAIL ? load_t_frag_guard(rq, q0, s2, hi, p.nq) : load_t_frag(rq, q0, s2, hi);
                acc_v[d] = Mfma32<T>::run(__builtin_bit_cast(frag, a_do), __builtin_bit_cast(frag, pp[s2]), acc_v[d]);   // dV^T += dO^T P
                acc_k[d] = Mfma32<T>::run(__builtin_bit_cast(frag, a_q), __builtin_bit_cast(frag, pd[s2]), acc_k[d]);    // dK^T += Q^T dS
            }
    }
    if (TAIL && k0 + kl >= p.nk) return;
    unsigned short* okp = p.dk + b * p.dk_bs + static_cast<long>(k0 + kl) * p.dk_ld + h * D;
    unsigned short* ovp = p.dv + b * p.dv_bs + static_cast<long>(k0 + kl) * p.dv_ld + h * D;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u16x4 wk, wv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wk[e] = from_f32<T>(acc_k[d][4 * g + e] * p.scale);
                wv[e] = from_f32<T>(acc_v[d][4 * g + e]);
            }
            *reinterpret_cast<u16x4*>(okp + d * 32 + 8 * g + 4 * hi) = wk;
            *reinterpret_cast<u16x4*>(ovp + d * 32 + 8 * g + 4 * hi) = wv;
        }
}

// ---- LDS-staged variants (token counts that are multiples of 32) ----------------------------------------------------
// Same math and register layout; the tile every wavefront of the block walks past (32 tokens of the OTHER side: rows and
// their transposes) is fetched once per block with coalesced 16-byte loads, staged through LDS and shared by the four
// waves, double buffered with the next tile's global loads in flight during the MFMAs (one barrier per tile).  The direct
// kernels issue fragment-shaped loads (32 cache lines per instruction) from every wave.
//   row-major tiles [32][D] with rows padded to D + 8 elements (fragment reads: 16 bytes per lane, rows 144 / 80 bytes apart)
//   transposed tiles [D][32] with rows padded to 36 elements (two 8-byte reads per lane, rows 72 bytes apart)
template <int D> struct BwdTile {
    static constexpr int RS = D + 8, TS = 32 + 4;
    static constexpr int ROWMAJ = 32 * RS, TRANS = D * TS;
    static constexpr int CH_R = 32 * (D / 8), CH_T = D * 4;      // 16-byte chunks per row-major / transposed tile
};

__device__ __forceinline__ u16x8 lds_t_frag(const unsigned short* row, int s2, int hi) {
    const unsigned short* pp = row + 16 * s2 + 4 * hi;
    const u16x4 lo = *reinterpret_cast<const u16x4*>(pp);
    const u16x4 up = *reinterpret_cast<const u16x4*>(pp + 8);
    return u16x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
}

template <typename T, int D>
__global__ __launch_bounds__(256) void k_attn_bwd_dkv_lds(const AttnBwdParams p) {
    constexpr int KS = D / 16, DB = D / 32;
    typedef BwdTile<D> TL;
    typedef typename Mfma32<T>::frag frag;
    constexpr int BUF = 2 * TL::ROWMAJ + 2 * TL::TRANS;          // Q rows | dO rows | Q^T | dO^T
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BUF];
    // log-sum-exp and delta of the 32 staged queries travel with the tiles: read straight from global memory inside the
    // step they sat on the critical path of every iteration (a step is about as long as one L2 round trip)
    __shared__ __attribute__((aligned(16))) float lsd[2][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int kl = lane & 31, hi = lane >> 5;
    const int qs = blockIdx.x % p.qsplit;                          // this block's share of the query tiles
    const int k0 = ((blockIdx.x / p.qsplit) * 4 + wave) * 32;
    const bool live = k0 < p.nk;                                   // (a block's trailing waves still help staging)
    const int h = blockIdx.y;
    const long b = blockIdx.z;
    const int krow = live ? k0 + kl : p.nk - 1;
    const unsigned short* kp = p.k + b * p.k_bs + static_cast<long>(krow) * p.k_ld + h * D;
    const unsigned short* vp = p.v + b * p.v_bs + static_cast<long>(krow) * p.v_ld + h * D;
    const unsigned short* qp = p.q + b * p.q_bs + h * D;
    const unsigned short* dop = p.dout + b * p.do_bs + h * D;
    const unsigned short* qtp = p.qt + b * p.qt_bs + static_cast<long>(h) * D * p.qt_ld;
    const unsigned short* dotp = p.dot + b * p.dot_bs + static_cast<long>(h) * D * p.dot_ld;

    frag kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        kf[s] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(kp + 16 * s + 8 * hi));
        vf[s] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(vp + 16 * s + 8 * hi));
    }
    const float c2 = p.scale_log2e;
    const float* lsep = p.lse + (b * p.H + h) * p.nq;
    const float* delp = p.delta + (b * p.H + h) * p.nq;
    f32x16 acc_k[DB], acc_v[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_k[d][r] = 0.f; acc_v[d][r] = 0.f; }

    // staging ownership: row-major chunk c -> (row c / (D/8), 8 channels at (c % (D/8)) * 8); transposed chunk c -> (row c / 4, 8 tokens at (c % 4) * 8)
    constexpr int NR = (TL::CH_R + 255) / 256, NT = (TL::CH_T + 255) / 256;
    u16x8 rq[NR], rdo[NR], rqt[NT], rdot[NT];
    float4 rls = {0.f, 0.f, 0.f, 0.f};
    auto stage_load = [&](int qt_) {
        const int q0 = qt_ * 32;
        if (t < 16) rls = *reinterpret_cast<const float4*>((t < 8 ? lsep : delp) + q0 + 4 * (t & 7));
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = t + 256 * i;
            if (c < TL::CH_R) {
                const int row = c / (D / 8), ch = (c % (D / 8)) * 8;
                rq[i] = *reinterpret_cast<const u16x8*>(qp + static_cast<long>(q0 + row) * p.q_ld + ch);
                rdo[i] = *reinterpret_cast<const u16x8*>(dop + static_cast<long>(q0 + row) * p.do_ld + ch);
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int c = t + 256 * i;
            if (c < TL::CH_T) {
                const int row = c >> 2, tk = (c & 3) * 8;
                rqt[i] = *reinterpret_cast<const u16x8*>(qtp + static_cast<long>(row) * p.qt_ld + q0 + tk);
                rdot[i] = *reinterpret_cast<const u16x8*>(dotp + static_cast<long>(row) * p.dot_ld + q0 + tk);
            }
        }
    };
    auto stage_store = [&](int buf) {
        unsigned short* Q = smem + buf * BUF;
        unsigned short* DO = Q + TL::ROWMAJ;
        unsigned short* QT = DO + TL::ROWMAJ;
        unsigned short* DOT = QT + TL::TRANS;
        if (t < 16) *reinterpret_cast<float4*>(&lsd[buf][(t < 8 ? 0 : 32) + 4 * (t & 7)]) = rls;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = t + 256 * i;
            if (c < TL::CH_R) {
                const int off = (c / (D / 8)) * TL::RS + (c % (D / 8)) * 8;
                *reinterpret_cast<u16x8*>(Q + off) = rq[i];
                *reinterpret_cast<u16x8*>(DO + off) = rdo[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int c = t + 256 * i;
            if (c < TL::CH_T) {
                const int off = (c >> 2) * TL::TS + (c & 3) * 8;          // 8-byte aligned (72-byte rows)
                const u16x8 a = rqt[i], d = rdot[i];
                *reinterpret_cast<u16x4*>(QT + off) = u16x4{a[0], a[1], a[2], a[3]};
                *reinterpret_cast<u16x4*>(QT + off + 4) = u16x4{a[4], a[5], a[6], a[7]};
                *reinterpret_cast<u16x4*>(DOT + off) = u16x4{d[0], d[1], d[2], d[3]};
                *reinterpret_cast<u16x4*>(DOT + off + 4) = u16x4{d[4], d[5], d[6], d[7]};
            }
        }
    };

    const int kt = k0 >> 5;
    const int per = (p.nq / 32 + p.qsplit - 1) / p.qsplit;
    const int qt_begin = qs * per, nqt = min(p.nq / 32, qt_begin + per);
    if (qt_begin < nqt) {
        stage_load(qt_begin);
        stage_store(qt_begin & 1);
    }
    __syncthreads();
    for (int qt = qt_begin; qt < nqt; ++qt) {
        if (qt + 1 < nqt) stage_load(qt + 1);
        if (live) {
            const int q0 = qt * 32;
            const unsigned short* Q = smem + (qt & 1) * BUF;
            const unsigned short* DO = Q + TL::ROWMAJ;
            const unsigned short* QT = DO + TL::ROWMAJ;
            const unsigned short* DOT = QT + TL::TRANS;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u16x8 qf = *reinterpret_cast<const u16x8*>(Q + kl * TL::RS + 16 * ks + 8 * hi);
                const u16x8 df = *reinterpret_cast<const u16x8*>(DO + kl * TL::RS + 16 * ks + 8 * hi);
                s = Mfma32<T>::run(__builtin_bit_cast(frag, qf), kf[ks], s);
                dp = Mfma32<T>::run(__builtin_bit_cast(frag, df), vf[ks], dp);
            }
            float sv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = s[r] * c2;
            if (p.flags && p.flags[static_cast<long>(qt) * p.flags_ld + kt]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qi = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    sv[r] += p.bias[static_cast<long>(qi) * p.bias_ld + k0 + kl] * LOG2E;
                }
            }
            u16x8 pp[2], pd[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 l4 = *reinterpret_cast<const float4*>(&lsd[qt & 1][8 * g + 4 * hi]);
                const float4 d4 = *reinterpret_cast<const float4*>(&lsd[qt & 1][32 + 8 * g + 4 * hi]);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float pr = exp2f(sv[r] - lv[e]);
                    pp[r >> 3][r & 7] = from_f32<T>(pr);
                    pd[r >> 3][r & 7] = from_f32<T>(pr * (dp[r] - dv[e]));
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const u16x8 a_do = lds_t_frag(DOT + (d * 32 + kl) * TL::TS, s2, hi);
                    const u16x8 a_q = lds_t_frag(QT + (d * 32 + kl) * TL::TS, s2, hi);
                    acc_v[d] = Mfma32<T>::run(__builtin_bit_cast(frag, a_do), __builtin_bit_cast(frag, pp[s2]), acc_v[d]);
                    acc_k[d] = Mfma32<T>::run(__builtin_bit_cast(frag, a_q), __builtin_bit_cast(frag, pd[s2]), acc_k[d]);
                }
        }
        if (qt + 1 < nqt) stage_store((qt + 1) & 1);      // that buffer was last read in step qt - 1: every wave passed the barrier since
        __syncthreads();
    }
    if (!live) return;
    if (p.qsplit > 1) {                                            // fp32 partial sums of this query range
        const long row = ((static_cast<long>(qs) * gridDim.z + b) * p.nk + k0 + kl) * (static_cast<long>(p.H) * D) + h * D;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = d * 32 + 8 * g + 4 * hi;
                *reinterpret_cast<float4*>(p.part_k + row + c) = float4{acc_k[d][4 * g] * p.scale, acc_k[d][4 * g + 1] * p.scale,
                                                                        acc_k[d][4 * g + 2] * p.scale, acc_k[d][4 * g + 3] * p.scale};
                *reinterpret_cast<float4*>(p.part_v + row + c) = float4{acc_v[d][4 * g], acc_v[d][4 * g + 1], acc_v[d][4 * g + 2], acc_v[d][4 * g + 3]};
            }
        return;
    }
    unsigned short* okp = p.dk + b * p.dk_bs + static_cast<long>(k0 + kl) * p.dk_ld + h * D;
    unsigned short* ovp = p.dv + b * p.dv_bs + static_cast<long>(k0 + kl) * p.dv_ld + h * D;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u16x4 wk, wv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wk[e] = from_f32<T>(acc_k[d][4 * g + e] * p.scale);
                wv[e] = from_f32<T>(acc_v[d][4 * g + e]);
            }
            *reinterpret_cast<u16x4*>(okp + d * 32 + 8 * g + 4 * hi) = wk;
            *reinterpret_cast<u16x4*>(ovp + d * 32 + 8 * g + 4 * hi) = wv;
        }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void k_attn_bwd_dq_lds(const AttnBwdParams p) {
    constexpr int KS = D / 16, DB = D / 32;
    typedef BwdTile<D> TL;
    typedef typename Mfma32<T>::frag frag;
    constexpr int BUF = 2 * TL::ROWMAJ + TL::TRANS;              // K rows | V rows | K^T
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BUF];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const bool live = q0 < p.nq;
    const int h = blockIdx.y;
    const long b = blockIdx.z;
    const int qrow = live ? q0 + ql : p.nq - 1;
    const unsigned short* qp = p.q + b * p.q_bs + static_cast<long>(qrow) * p.q_ld + h * D;
    const unsigned short* dop = p.dout + b * p.do_bs + static_cast<long>(qrow) * p.do_ld + h * D;
    const unsigned short* kp = p.k + b * p.k_bs + h * D;
    const unsigned short* vp = p.v + b * p.v_bs + h * D;
    const unsigned short* ktp = p.kt + b * p.kt_bs + static_cast<long>(h) * D * p.kt_ld;

    frag qf[KS], dof[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(qp + 16 * s + 8 * hi));
        dof[s] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(dop + 16 * s + 8 * hi));
    }
    const long stat = (b * p.H + h) * p.nq + qrow;
    const float lse = p.lse[stat], delta = p.delta[stat];
    const float c2 = p.scale_log2e;
    f32x16 acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    constexpr int NR = (TL::CH_R + 255) / 256, NT = (TL::CH_T + 255) / 256;
    u16x8 rk[NR], rv[NR], rkt[NT];
    auto stage_load = [&](int kt_) {
        const int k0 = kt_ * 32;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = t + 256 * i;
            if (c < TL::CH_R) {
                const int row = c / (D / 8), ch = (c % (D / 8)) * 8;
                rk[i] = *reinterpret_cast<const u16x8*>(kp + static_cast<long>(k0 + row) * p.k_ld + ch);
                rv[i] = *reinterpret_cast<const u16x8*>(vp + static_cast<long>(k0 + row) * p.v_ld + ch);
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int c = t + 256 * i;
            if (c < TL::CH_T) rkt[i] = *reinterpret_cast<const u16x8*>(ktp + static_cast<long>(c >> 2) * p.kt_ld + k0 + (c & 3) * 8);
        }
    };
    auto stage_store = [&](int buf) {
        unsigned short* K = smem + buf * BUF;
        unsigned short* V = K + TL::ROWMAJ;
        unsigned short* KT = V + TL::ROWMAJ;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = t + 256 * i;
            if (c < TL::CH_R) {
                const int off = (c / (D / 8)) * TL::RS + (c % (D / 8)) * 8;
                *reinterpret_cast<u16x8*>(K + off) = rk[i];
                *reinterpret_cast<u16x8*>(V + off) = rv[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int c = t + 256 * i;
            if (c < TL::CH_T) {
                const int off = (c >> 2) * TL::TS + (c & 3) * 8;
                const u16x8 a = rkt[i];
                *reinterpret_cast<u16x4*>(KT + off) = u16x4{a[0], a[1], a[2], a[3]};
                *reinterpret_cast<u16x4*>(KT + off + 4) = u16x4{a[4], a[5], a[6], a[7]};
            }
        }
    };

    const uint8_t* flag_row = p.flags ? p.flags + static_cast<long>(min(q0, p.nq - 32) >> 5) * p.flags_ld : nullptr;
    const float* bias_row = p.bias ? p.bias + static_cast<long>(qrow) * p.bias_ld : nullptr;
    const int nkt = p.nk / 32;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) stage_load(kt + 1);
        if (live) {
            const int k0 = kt * 32;
            const unsigned short* K = smem + (kt & 1) * BUF;
            const unsigned short* V = K + TL::ROWMAJ;
            const unsigned short* KT = V + TL::ROWMAJ;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u16x8 kf = *reinterpret_cast<const u16x8*>(K + ql * TL::RS + 16 * ks + 8 * hi);
                const u16x8 vf = *reinterpret_cast<const u16x8*>(V + ql * TL::RS + 16 * ks + 8 * hi);
                s = Mfma32<T>::run(__builtin_bit_cast(frag, kf), qf[ks], s);
                dp = Mfma32<T>::run(__builtin_bit_cast(frag, vf), dof[ks], dp);
            }
            float sv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = s[r] * c2;
            if (flag_row && flag_row[kt]) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias_row + k0 + 8 * g + 4 * hi);
                    sv[4 * g + 0] += bv.x * LOG2E;
                    sv[4 * g + 1] += bv.y * LOG2E;
                    sv[4 * g + 2] += bv.z * LOG2E;
                    sv[4 * g + 3] += bv.w * LOG2E;
                }
            }
            u16x8 pb[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) pb[r >> 3][r & 7] = from_f32<T>(exp2f(sv[r] - lse) * (dp[r] - delta));
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const u16x8 a = lds_t_frag(KT + (d * 32 + ql) * TL::TS, s2, hi);
                    acc[d] = Mfma32<T>::run(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, pb[s2]), acc[d]);
                }
        }
        if (kt + 1 < nkt) stage_store((kt + 1) & 1);
        __syncthreads();
    }
    if (!live) return;
    unsigned short* op = p.dq + b * p.dq_bs + static_cast<long>(q0 + ql) * p.dq_ld + h * D;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u16x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(acc[d][4 * g + e] * p.scale);
            *reinterpret_cast<u16x4*>(op + d * 32 + 8 * g + 4 * hi) = w;
        }
}

// dK / dV of a query-split keys-stationary launch: the qsplit fp32 partial sums added in order, written in the 16-bit type.
template <typename T>
__global__ void k_attn_bwd_reduce(const float* __restrict__ part_k, const float* __restrict__ part_v, int qsplit, int B, int nk, int HD,
                                  unsigned short* __restrict__ dk, unsigned short* __restrict__ dv, int dk_ld, int dv_ld, long dk_bs, long dv_bs) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;       // (b, key, quad of channels)
    const int Q4 = HD / 4;
    if (i >= static_cast<long>(B) * nk * Q4) return;
    const int c = static_cast<int>(i % Q4) * 4;
    const long key = (i / Q4) % nk, b = i / (static_cast<long>(Q4) * nk);
    const long slab = static_cast<long>(B) * nk * HD, off = (b * nk + key) * HD + c;
    float4 sk = *reinterpret_cast<const float4*>(part_k + off), sv = *reinterpret_cast<const float4*>(part_v + off);
    for (int q = 1; q < qsplit; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(part_k + q * slab + off), v = *reinterpret_cast<const float4*>(part_v + q * slab + off);
        sk.x += a.x; sk.y += a.y; sk.z += a.z; sk.w += a.w;
        sv.x += v.x; sv.y += v.y; sv.z += v.z; sv.w += v.w;
    }
    *reinterpret_cast<u16x4*>(dk + b * dk_bs + key * dk_ld + c) = u16x4{from_f32<T>(sk.x), from_f32<T>(sk.y), from_f32<T>(sk.z), from_f32<T>(sk.w)};
    *reinterpret_cast<u16x4*>(dv + b * dv_bs + key * dv_ld + c) = u16x4{from_f32<T>(sv.x), from_f32<T>(sv.y), from_f32<T>(sv.z), from_f32<T>(sv.w)};
}

template <typename T>
__global__ void k_attn_delta(const unsigned short* __restrict__ o, const unsigned short* __restrict__ d_o, int B, int H,
                             int D, long nq, int ld, long bs, float* __restrict__ delta) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;       // (b, q, h), h fastest
    if (i >= static_cast<long>(B) * nq * H) return;
    const int h = i % H;
    const long q = (i / H) % nq, b = i / (H * nq);
    const long off = b * bs + q * ld + h * D;
    float acc = 0.f;
    for (int c = 0; c < D; c += 8) {
        float a[8], g[8];
        unpack8<T>(*reinterpret_cast<const u16x8*>(o + off + c), a);
        unpack8<T>(*reinterpret_cast<const u16x8*>(d_o + off + c), g);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += a[j] * g[j];
    }
    delta[(b * H + h) * nq + q] = acc;
}

// ---- LayerNorm backward -------------------------------------------------------------------------------------------
constexpr int LN_BWD_MAX_PARTS = 512;

template <typename TI> __device__ __forceinline__ void load8_any(const void* base, long idx, float (&f)[8]) {
    unpack8<TI>(*reinterpret_cast<const u16x8*>(static_cast<const unsigned short*>(base) + idx), f);
}
template <> __device__ __forceinline__ void load8_any<float>(const void* base, long idx, float (&f)[8]) {
    const float* xp = static_cast<const float*>(base) + idx;
    const float4 a = reinterpret_cast<const float4*>(xp)[0], c = reinterpret_cast<const float4*>(xp)[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = c.x; f[5] = c.y; f[6] = c.z; f[7] = c.w;
}

template <typename TI>
__global__ __launch_bounds__(256) void k_layernorm_bwd(const void* __restrict__ xv, const float* __restrict__ pe, long pe_rows,
                                                       long rows, int C, const float* __restrict__ gamma, float eps,
                                                       const float* __restrict__ dy, const float* __restrict__ dres,
                                                       float* __restrict__ dx, float* __restrict__ partials, int n_part) {
    constexpr int KMAX = 4;                            // octets per lane (C <= 2048)
    __shared__ float flat[4 * 2048];                   // the four waves' column sums, one quantity at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int OCT = C / 8;
    float sg[KMAX][8], sb[KMAX][8];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) { sg[k][j] = 0.f; sb[k][j] = 0.f; }

    for (long row = blockIdx.x * 4L + wave; row < rows; row += 4L * gridDim.x) {
        float v[KMAX][8], g[KMAX][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int oct = lane + 64 * k;
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[k][j] = 0.f; g[k][j] = 0.f; }
            if (oct < OCT) {
                load8_any<TI>(xv, row * C + oct * 8, v[k]);
                if (pe) {
                    const float4* pp = reinterpret_cast<const float4*>(pe + (row % pe_rows) * C + oct * 8);
                    const float4 a = pp[0], c = pp[1];
                    v[k][0] += a.x; v[k][1] += a.y; v[k][2] += a.z; v[k][3] += a.w; v[k][4] += c.x; v[k][5] += c.y; v[k][6] += c.z; v[k][7] += c.w;
                }
                const float4* gp = reinterpret_cast<const float4*>(dy + row * C + oct * 8);
                const float4 a = gp[0], c = gp[1];
                g[k][0] = a.x; g[k][1] = a.y; g[k][2] = a.z; g[k][3] = a.w; g[k][4] = c.x; g[k][5] = c.y; g[k][6] = c.z; g[k][7] = c.w;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[k][j];
            }
        }
        const float mean = wave_sum_b(s) / static_cast<float>(C);
        float qv = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (lane + 64 * k < OCT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[k][j] - mean; qv += d * d; }
            }
        const float rstd = 1.0f / sqrtf(wave_sum_b(qv) / static_cast<float>(C) + eps);
        // xhat in v, g = dy * gamma; row sums of g and g * xhat
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int oct = lane + 64 * k;
            if (oct < OCT) {
                const float4* gp = reinterpret_cast<const float4*>(gamma + oct * 8);
                const float4 a = gp[0], c = gp[1];
                const float gm[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (v[k][j] - mean) * rstd;
                    v[k][j] = xh;
                    sg[k][j] += g[k][j] * xh;
                    sb[k][j] += g[k][j];
                    g[k][j] *= gm[j];
                    s1 += g[k][j];
                    s2 += g[k][j] * xh;
                }
            }
        }
        const float m1 = wave_sum_b(s1) / static_cast<float>(C), m2 = wave_sum_b(s2) / static_cast<float>(C);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int oct = lane + 64 * k;
            if (oct < OCT) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rstd * (g[k][j] - m1 - v[k][j] * m2);
                if (dres) {
                    const float4* rp = reinterpret_cast<const float4*>(dres + row * C + oct * 8);
                    const float4 a = rp[0], c = rp[1];
                    o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w; o[4] += c.x; o[5] += c.y; o[6] += c.z; o[7] += c.w;
                }
                float4* op = reinterpret_cast<float4*>(dx + row * C + oct * 8);
                op[0] = float4{o[0], o[1], o[2], o[3]};
                op[1] = float4{o[4], o[5], o[6], o[7]};
            }
        }
    }
    // the four waves of the block: fixed-order sum through LDS, one partial row per block
    for (int which = 0; which < 2; ++which) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int oct = lane + 64 * k;
            if (oct < OCT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) flat[wave * C + oct * 8 + j] = which == 0 ? sg[k][j] : sb[k][j];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256)
            partials[(static_cast<long>(which) * n_part + blockIdx.x) * C + c] = ((flat[c] + flat[C + c]) + flat[2 * C + c]) + flat[3 * C + c];
    }
}

// ---- GroupNorm (+ SiLU) backward ----------------------------------------------------------------------------------
// y = act(x * scale + shift) with scale = gamma * rstd, shift = beta - mean * scale per (image, channel) as the forward
// left them (pf_groupnorm_stats); x = channel concat of two sources.  Given dy:
//   dz = dy * act'(z),  g = dz * gamma,  dx = rstd * (g - mean_grp(g) - xhat * mean_grp(g * xhat))   (+ dres)
// Three launches: per-(image, pixel chunk) partial sums of (x, x^2, g, g x) per group; one block per image folds them
// into (mean, rstd, mean g, mean g xhat); the apply pass.  gamma / beta take no gradient (the UNet is frozen: only the
// LoRA matrices and the EPA blocks train, PanoGenerator.py:141-160).
__device__ __forceinline__ float act_grad(float z, int act) {
    if (!act) return 1.0f;
    const float sg = 1.0f / (1.0f + __expf(-z));
    return sg * (1.0f + z * (1.0f - sg));
}

template <typename TI>
__global__ __launch_bounds__(256) void k_gn_bwd_partial(const void* __restrict__ x0, int c0, const void* __restrict__ x1, int c1,
                                                        int hw, int groups, int pix_per_chunk, const float* __restrict__ gamma,
                                                        const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                        const float* __restrict__ dy, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [4][pix_par][C]
    const int C = c0 + c1, OCT = C / 8, cpg = C / groups;
    const int OCTB = OCT < 256 ? OCT : 256;
    const int pix_par = 256 / OCTB;
    const int img = blockIdx.y, chunk = blockIdx.x;
    const int p0 = chunk * pix_per_chunk, p1 = min(p0 + pix_per_chunk, hw);
    const int t = threadIdx.x, slot = t / OCTB, olane = t % OCTB;
    for (int ob = 0; ob < OCT; ob += OCTB) {
        const int oct = ob + olane;
        if (slot < pix_par && oct < OCT) {
            const int c = oct * 8;
            float a[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[q][j] = 0.f;
            float sc[8], sh[8], gm[8];
            load8_any<float>(scale, static_cast<long>(img) * C + c, sc);
            load8_any<float>(shift, static_cast<long>(img) * C + c, sh);
            load8_any<float>(gamma, c, gm);
            for (int pp = p0 + slot; pp < p1; pp += pix_par) {
                const long pix = static_cast<long>(img) * hw + pp;
                float v[8], d[8];
                if (c < c0) load8_any<TI>(x0, pix * c0 + c, v); else load8_any<TI>(x1, pix * c1 + (c - c0), v);
                load8_any<float>(dy, pix * C + c, d);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g = d[j] * act_grad(v[j] * sc[j] + sh[j], act) * gm[j];
                    a[0][j] += v[j];
                    a[1][j] += v[j] * v[j];
                    a[2][j] += g;
                    a[3][j] += g * v[j];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) sm[(q * pix_par + slot) * C + c + j] = a[q][j];
        }
    }
    __syncthreads();
    for (int i = t; i < groups * 4; i += 256) {
        const int g = i >> 2, q = i & 3;
        float s = 0.f;
        for (int sl = 0; sl < pix_par; ++sl)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) s += sm[(q * pix_par + sl) * C + c];
        partial[((static_cast<long>(img) * gridDim.x + chunk) * groups + g) * 4 + q] = s;
    }
}

__global__ __launch_bounds__(256) void k_gn_bwd_finalize(const float* __restrict__ partial, int nchunks, int groups, int C, int hw,
                                                         float eps, float* __restrict__ stats) {
    const int img = blockIdx.x;
    for (int g = threadIdx.x; g < groups; g += 256) {
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < nchunks; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(partial + ((static_cast<long>(img) * nchunks + k) * groups + g) * 4);
            a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
        }
        const double cnt = static_cast<double>(hw) * (C / groups);
        const double mean = a[0] / cnt;
        double var = a[1] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + static_cast<double>(eps));
        float* o = stats + (static_cast<long>(img) * groups + g) * 4;
        o[0] = static_cast<float>(mean);
        o[1] = static_cast<float>(rstd);
        o[2] = static_cast<float>(a[2] / cnt);                               // mean g
        o[3] = static_cast<float>(rstd * (a[3] - mean * a[2]) / cnt);        // mean g * xhat
    }
}

template <typename TI>
__global__ __launch_bounds__(256) void k_gn_bwd_apply(const void* __restrict__ x0, int c0, const void* __restrict__ x1, int c1, int hw,
                                                      int groups, const float* __restrict__ gamma, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int act, const float* __restrict__ dy,
                                                      const float* __restrict__ dres, const float* __restrict__ stats,
                                                      float* __restrict__ dx0, float* __restrict__ dx1) {
    const unsigned C = c0 + c1, OCT = C / 8, per_img = static_cast<unsigned>(hw) * OCT, cpg = C / groups;
    const unsigned img = blockIdx.y;
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    if (j >= per_img) return;
    const unsigned p = j / OCT, c = (j - p * OCT) * 8;
    const long pix = static_cast<long>(img) * hw + p;
    float v[8], d[8], sc[8], sh[8], gm[8], o[8];
    if (c < static_cast<unsigned>(c0)) load8_any<TI>(x0, pix * c0 + c, v); else load8_any<TI>(x1, pix * c1 + (c - c0), v);
    load8_any<float>(dy, pix * C + c, d);
    load8_any<float>(scale, static_cast<long>(img) * C + c, sc);
    load8_any<float>(shift, static_cast<long>(img) * C + c, sh);
    load8_any<float>(gamma, c, gm);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float* st = stats + (static_cast<long>(img) * groups + (c + e) / cpg) * 4;      // (groups of >= 1 channel)
        const float g = d[e] * act_grad(v[e] * sc[e] + sh[e], act) * gm[e];
        const float xh = (v[e] - st[0]) * st[1];
        o[e] = st[1] * (g - st[2] - xh * st[3]);
    }
    if (dres) {
        float r[8];
        load8_any<float>(dres, pix * C + c, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += r[e];
    }
    float* dst = (c < static_cast<unsigned>(c0)) ? dx0 + pix * c0 + c : dx1 + pix * c1 + (c - c0);
    reinterpret_cast<float4*>(dst)[0] = float4{o[0], o[1], o[2], o[3]};
    reinterpret_cast<float4*>(dst)[1] = float4{o[4], o[5], o[6], o[7]};
}

// ---- parameter gradients of a trainable GroupNorm (the ControlNet: every parameter trains, PanoGenerator.py:153-157) ----------
// y = act(gamma * xhat + beta), xhat = x * uscale + ushift per (image, channel) (the statistics as pf_groupnorm_stats leaves them
// for gamma = 1, beta = 0), z = x * scale + shift the pre-activation of the affine forward.  With dz = dy * act'(z):
//   dgamma[c] = sum dz * xhat,   dbeta[c] = sum dz        over images and pixels.
// One partial row [dgamma | dbeta] per (image, pixel chunk); the rows are added in a fixed order by the column-sum kernels.
template <typename TI>
__global__ __launch_bounds__(256) void k_gn_param_partial(const void* __restrict__ x0, int c0, const void* __restrict__ x1, int c1,
                                                          int hw, int pix_per_chunk, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ uscale,
                                                          const float* __restrict__ ushift, int act, const float* __restrict__ dy,
                                                          float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [2][pix_par][C]
    const int C = c0 + c1, OCT = C / 8;
    const int OCTB = OCT < 256 ? OCT : 256;
    const int pix_par = 256 / OCTB;
    const int img = blockIdx.y, chunk = blockIdx.x;
    const int p0 = chunk * pix_per_chunk, p1 = min(p0 + pix_per_chunk, hw);
    const int t = threadIdx.x, slot = t / OCTB, olane = t % OCTB;
    for (int ob = 0; ob < OCT; ob += OCTB) {
        const int oct = ob + olane;
        if (slot < pix_par && oct < OCT) {
            const int c = oct * 8;
            float ag[8], ab[8], sc[8], sh[8], us[8], uh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ag[j] = ab[j] = 0.f;
            load8_any<float>(scale, static_cast<long>(img) * C + c, sc);
            load8_any<float>(shift, static_cast<long>(img) * C + c, sh);
            load8_any<float>(uscale, static_cast<long>(img) * C + c, us);
            load8_any<float>(ushift, static_cast<long>(img) * C + c, uh);
            for (int pp = p0 + slot; pp < p1; pp += pix_par) {
                const long pix = static_cast<long>(img) * hw + pp;
                float v[8], d[8];
                if (c < c0) load8_any<TI>(x0, pix * c0 + c, v); else load8_any<TI>(x1, pix * c1 + (c - c0), v);
                load8_any<float>(dy, pix * C + c, d);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float dz = d[j] * act_grad(v[j] * sc[j] + sh[j], act);
                    ag[j] += dz * (v[j] * us[j] + uh[j]);
                    ab[j] += dz;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sm[(0 * pix_par + slot) * C + c + j] = ag[j];
                sm[(1 * pix_par + slot) * C + c + j] = ab[j];
            }
        }
    }
    __syncthreads();
    float* row = partial + (static_cast<long>(img) * gridDim.x + chunk) * 2 * C;
    for (int i = t; i < 2 * C; i += 256) {
        const int q = i / C, c = i - q * C;
        float s = 0.f;
        for (int sl = 0; sl < pix_par; ++sl) s += sm[(q * pix_par + sl) * C + c];
        row[i] = s;
    }
}

// dz = dy * silu'(z): z 16-bit or fp32 (the pre-activation the forward kept), dy / dz fp32.
template <typename S>
__global__ void k_silu_bwd(const void* __restrict__ z, const float* __restrict__ dy, long n, float* __restrict__ dz) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i < n) dz[i] = dy[i] * act_grad(ld_any<S>(z, i), 1);
}

// im2col of a 3x3 / pad 1 convolution (stride 1 or 2): y [n * ho * wo][9][C] = the nine zero-padded taps of x [n][h][w][C]
// (16-bit octets).  The weight gradient of the convolution is then ONE token-reducing GEMM dW [cout][9 C] = dY^T cols.
__global__ void k_im2col3(const u16x8* __restrict__ x, int n, int h, int w, int OCT, int stride, int ho, int wo, u16x8* __restrict__ y) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    const long total = static_cast<long>(n) * ho * wo * 9 * OCT;
    if (i >= total) return;
    const int o = i % OCT;
    const int tap = (i / OCT) % 9;
    const long pix = i / (9L * OCT);
    const int xo = pix % wo, yo = (pix / wo) % ho;
    const long img = pix / (static_cast<long>(ho) * wo);
    const int yy = yo * stride + tap / 3 - 1, xx = xo * stride + tap % 3 - 1;
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = x[((img * h + yy) * w + xx) * OCT + o];
    y[i] = v;
}

// ---- LoRA fold: W' = W + scale * up @ down, every step of a training run ---------------------------------------------------
// (PanoGenerator.py:129-151: rank-4 LoRA on q / k / v / out of every attention; the forward kernels read the folded 16-bit weight.)
// One launch per projection writes, from the fp32 weight and the two thin matrices:
//   out   [N][out_ld]    the folded weight in the 16-bit operand type (a row slice of the packed (q | k | v) buffer),
//   out_t [K][out_t_ld]  its transpose (data-gradient operand of the backward), optional,
//   d_out [r][d_ld]      the down matrix in 16 bit (rows of the group's stacked D), optional,
//   u_out [r][u_ld]      the up matrix transposed in 16 bit (a block of the group's block-diagonal U), optional.
// 64 x 64 tiles; the transpose goes through LDS so that both outputs are written in whole rows.
constexpr int LORA_MAX_RANK = 64;                                  // (lora_rank is a hyperparameter of the reference: PanoGenerator.py:73)
template <typename T>
__global__ __launch_bounds__(256) void k_lora_fold(const float* __restrict__ w, const float* __restrict__ up, const float* __restrict__ down,
                                                   int N, int K, int r, float scale, unsigned short* __restrict__ out, long out_ld,
                                                   unsigned short* __restrict__ out_t, long out_t_ld, unsigned short* __restrict__ d_out,
                                                   long d_ld, unsigned short* __restrict__ u_out, long u_ld) {
    __shared__ unsigned short tile[64][66];
    __shared__ float s_up[64][LORA_MAX_RANK], s_down[LORA_MAX_RANK][64];
    const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64, t = threadIdx.x;
    for (int i = t; i < 64 * r; i += 256) {
        const int a = i / r, j = i - a * r;                         // up [N][r]: row n0 + a
        s_up[a][j] = (up && n0 + a < N) ? up[static_cast<long>(n0 + a) * r + j] : 0.f;
        const int jj = i / 64, b = i - jj * 64;                     // down [r][K]: column k0 + b
        s_down[jj][b] = (down && k0 + b < K) ? down[static_cast<long>(jj) * K + k0 + b] : 0.f;
    }
    __syncthreads();
    const int kl = t & 63;
    float wv[16];                                                  // the thread's 16 weights in flight before the first is used
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = n0 + (t >> 6) + 4 * i, k = k0 + kl;
        wv[i] = (n < N && k < K) ? w[static_cast<long>(n) * K + k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int nl = (t >> 6) + 4 * i, n = n0 + nl, k = k0 + kl;
        unsigned short v = 0;
        if (n < N && k < K) {
            float acc = 0.f;
            for (int j = 0; j < r; ++j) acc += s_up[nl][j] * s_down[j][kl];
            v = from_f32<T>(wv[i] + scale * acc);
            out[static_cast<long>(n) * out_ld + k] = v;
        }
        tile[nl][kl] = v;
    }
    if (out_t) {
        __syncthreads();
        const int nl = t & 63;
        for (int kk = t >> 6; kk < 64; kk += 4)
            if (n0 + nl < N && k0 + kk < K) out_t[static_cast<long>(k0 + kk) * out_t_ld + n0 + nl] = tile[nl][kk];
    }
    if (d_out && blockIdx.y == 0)
        for (int i = t; i < 64 * r; i += 256) {
            const int j = i / 64, b = i - j * 64;
            if (k0 + b < K) d_out[static_cast<long>(j) * d_ld + k0 + b] = from_f32<T>(s_down[j][b]);
        }
    if (u_out && blockIdx.x == 0)
        for (int i = t; i < 64 * r; i += 256) {
            const int j = i / 64, a = i - j * 64;
            if (n0 + a < N) u_out[static_cast<long>(j) * u_ld + n0 + a] = from_f32<T>(s_up[a][j]);
        }
}

// ---- data movement of the backward pass ---------------------------------------------------------------------------
// zero insertion (data gradient of a stride-2 convolution = stride-1 convolution of the zero-stuffed gradient with the
// flipped kernel): y[n][2i][2j] = x[n][i][j], zero elsewhere; 16-bit octets.
__global__ void k_zero_insert2(const u16x8* __restrict__ x, int n, int h, int w, int OCT, u16x8* __restrict__ y) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    const long total = static_cast<long>(n) * 4 * h * w * OCT;
    if (i >= total) return;
    const int o = i % OCT;
    const long pix = i / OCT;
    const int xx = pix % (2 * w), yy = (pix / (2 * w)) % (2 * h);
    const long img = pix / (4L * h * w);
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!(xx & 1) && !(yy & 1)) v = x[((img * h + (yy >> 1)) * w + (xx >> 1)) * OCT + o];
    y[i] = v;
}

// nearest x2 up-sampling backward: y[n][i][j] = sum of the 2x2 block of x [n][2h][2w][C], fp32 quads.
__global__ void k_sum2x2(const float4* __restrict__ x, int n, int h, int w, int Q, float4* __restrict__ y) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    const long total = static_cast<long>(n) * h * w * Q;
    if (i >= total) return;
    const int q = i % Q;
    const long pix = i / Q;
    const int xx = pix % w, yy = (pix / w) % h;
    const long img = pix / (static_cast<long>(h) * w);
    const float4* b = x + ((img * 2 * h + 2 * yy) * 2 * w + 2 * xx) * Q + q;
    const float4 a0 = b[0], a1 = b[Q], a2 = b[2L * w * Q], a3 = b[2L * w * Q + Q];
    y[i] = float4{(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w)};
}

// backward of the circular width pad (utils/pano.py:74-99): dx[.., j] = dy[.., j + pad] + the wrapped copies
// (left margin column i is a copy of column w - pad + i, right margin column w + pad + i a copy of column i);
// backward of the crop: dy placed at column offset `crop`, zero margins.  fp32 quads of the channel axis.
__global__ void k_pad_width_bwd(const float4* __restrict__ dy, long rows, int w, int pad, int Q, float4* __restrict__ dx) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= rows * w * Q) return;
    const int q = i % Q;
    const int j = (i / Q) % w;
    const long r = i / (static_cast<long>(Q) * w);
    const float4* row = dy + r * (w + 2 * pad) * Q + q;
    float4 a = row[static_cast<long>(j + pad) * Q];
    if (j >= w - pad) { const float4 b = row[static_cast<long>(j - (w - pad)) * Q]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    if (j < pad) { const float4 b = row[static_cast<long>(w + pad + j) * Q]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    dx[i] = a;
}

__global__ void k_crop_width_bwd(const float4* __restrict__ dy, long rows, int w, int crop, int Q, float4* __restrict__ dx) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= rows * w * Q) return;
    const int q = i % Q;
    const int j = (i / Q) % w;
    const long r = i / (static_cast<long>(Q) * w);
    float4 a = {0.f, 0.f, 0.f, 0.f};
    if (j >= crop && j < w - crop) a = dy[(r * (w - 2 * crop) + (j - crop)) * Q + q];
    dx[i] = a;
}

// ---- token transpose ----------------------------------------------------------------------------------------------
// x [B][T][C] -> y [B][C][T] for 16-bit elements: 64 x 64 tiles through LDS, 8-byte accesses on both sides
// (the generic permute kernel moves single elements: one side of it is strided by the row length).
__global__ __launch_bounds__(256) void k_transpose16(const unsigned short* __restrict__ x, long T, int C, unsigned short* __restrict__ y) {
    __shared__ unsigned short tile[64][64 + 4];
    const long b = blockIdx.z;
    const long t0 = blockIdx.x * 64L;
    const int c0 = blockIdx.y * 64;
    const unsigned short* xb = x + b * T * C;
    unsigned short* yb = y + b * T * C;
    const int q = threadIdx.x & 15, r = threadIdx.x >> 4;          // 16 quads per 64-element row, 16 rows per pass
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long t = t0 + r + 16 * p;
        const int c = c0 + 4 * q;
        u16x4 v = {0, 0, 0, 0};
        if (t < T && c + 3 < C) v = *reinterpret_cast<const u16x4*>(xb + t * C + c);
        else if (t < T) { for (int e = 0; e < 4; ++e) if (c + e < C) v[e] = xb[t * C + c + e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[4 * q + e][r + 16 * p] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = c0 + r + 16 * p;
        const long t = t0 + 4 * q;
        if (c >= C) continue;
        const unsigned short* src = &tile[r + 16 * p][4 * q];
        if (t + 3 < T && (T & 3) == 0) *reinterpret_cast<u16x4*>(yb + static_cast<long>(c) * T + t) = u16x4{src[0], src[1], src[2], src[3]};
        else { for (int e = 0; e < 4; ++e) if (t + e < T) yb[static_cast<long>(c) * T + t + e] = src[e]; }
    }
}

// ---- GEGLU backward -----------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_geglu_bwd(const unsigned short* __restrict__ u, const unsigned short* __restrict__ dg, long total_oct,
                            int inner, unsigned short* __restrict__ du) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= total_oct) return;
    const int OCT = inner / 8;
    const long row = i / OCT;
    const int c = (i % OCT) * 8;
    float a[8], g[8], d[8], da[8], dgate[8];
    unpack8<T>(*reinterpret_cast<const u16x8*>(u + row * 2 * inner + c), a);
    unpack8<T>(*reinterpret_cast<const u16x8*>(u + row * 2 * inner + inner + c), g);
    unpack8<T>(*reinterpret_cast<const u16x8*>(dg + row * inner + c), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float cdf = 0.5f * (1.0f + erff(g[j] * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * __expf(-0.5f * g[j] * g[j]);
        da[j] = d[j] * g[j] * cdf;
        dgate[j] = d[j] * a[j] * (cdf + g[j] * pdf);
    }
    *reinterpret_cast<u16x8*>(du + row * 2 * inner + c) = pack8<T>(da);
    *reinterpret_cast<u16x8*>(du + row * 2 * inner + inner + c) = pack8<T>(dgate);
}

// ---- column sums --------------------------------------------------------------------------------------------------
constexpr int COLSUM_SLABS = 64;

template <typename S>
__global__ __launch_bounds__(256) void k_colsum_partial(const void* __restrict__ x, long rows, int N, long ld, int slabs,
                                                        float* __restrict__ part) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    const long per = (rows + slabs - 1) / slabs, r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float acc = 0.f;
    if (col < N)
        for (long r = r0 + ty; r < r1; r += 4) acc += ld_any<S>(x, r * ld + col);
    red[ty][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ty == 0 && col < N)
        part[static_cast<long>(blockIdx.y) * N + col] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ void k_colsum_final(const float* __restrict__ part, int slabs, int N, float* __restrict__ out) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= N) return;
    float acc = 0.f;
    for (int s = 0; s < slabs; ++s) acc += part[static_cast<long>(s) * N + col];
    out[col] = acc;
}

// ---- weighted column sums: out[r][c] = sum_t w[r][t] * x[t][c] -------------------------------------------------------------
// The token-reducing half of the LoRA gradients (d_up = dY^T P, d_down = Q^T X with P = X down^T, Q = dY up: R columns, <= 16 per launch):
// x is read ONCE, row-major as the backward holds it -- no transposed copy, no 64-row MFMA tile for a 4-row product.
// Partial sums per (row slab, 4 row phases) in a fixed order, then a final pass that also applies the gradient-normalisation
// factor and lays the result out per LoRA pair.
struct WcsBlocks { int n; int row0[4], rows[4], col0[4], cols[4]; };

template <typename T16, int RQ>                                     // R = 4 RQ weight rows (compile time: the row loop must not branch)
__global__ __launch_bounds__(256) void k_wcolsum_partial(const unsigned short* __restrict__ x, long T, int C, long ld,
                                                         const float* __restrict__ w, long w_ld, int slabs, float* __restrict__ part) {
    constexpr int CH = 256, R = 4 * RQ;                            // rows per staged chunk of the weights
    __shared__ __attribute__((aligned(16))) float sw[CH][R];       // w^T of the chunk: one ds_read_b128 per four weights, broadcast
    __shared__ float red[3][R][128];
    const int c2 = threadIdx.x & 63, ph = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = blockIdx.x * 128 + 2 * c2;
    const long per = (T + slabs - 1) / slabs, r0 = blockIdx.y * per, r1 = min(T, r0 + per);
    float a0[R], a1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a0[r] = a1[r] = 0.f;
    for (long base = r0; base < r1; base += CH) {
        const int nrow = static_cast<int>(min(static_cast<long>(CH), r1 - base));
        __syncthreads();
        for (int i = threadIdx.x; i < R * CH; i += 256) {
            const int r = i / CH, j = i - r * CH;                  // (consecutive threads: consecutive rows of one weight row)
            sw[j][r] = j < nrow ? w[r * w_ld + base + j] : 0.f;
        }
        __syncthreads();
        if (col < C) {
            const unsigned short* xp = x + base * ld + col;
            // eight rows of x in flight per thread; rows past the chunk carry zero weights and are clamped to a valid address
            for (int j0 = ph; j0 < nrow; j0 += 32) {
                unsigned v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const unsigned*>(xp + static_cast<long>(min(j0 + 4 * u, nrow - 1)) * ld);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = min(j0 + 4 * u, CH - 1);
                    const float x0 = to_f32<T16>(static_cast<unsigned short>(v[u] & 0xffffu)), x1 = to_f32<T16>(static_cast<unsigned short>(v[u] >> 16));
#pragma unroll
                    for (int q = 0; q < RQ; ++q) {
                        const float4 wv = *reinterpret_cast<const float4*>(&sw[j][4 * q]);
                        a0[4 * q] += wv.x * x0; a1[4 * q] += wv.x * x1;
                        a0[4 * q + 1] += wv.y * x0; a1[4 * q + 1] += wv.y * x1;
                        a0[4 * q + 2] += wv.z * x0; a1[4 * q + 2] += wv.z * x1;
                        a0[4 * q + 3] += wv.w * x0; a1[4 * q + 3] += wv.w * x1;
                    }
                }
            }
        }
    }
    if (ph > 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) { red[ph - 1][r][2 * c2] = a0[r]; red[ph - 1][r][2 * c2 + 1] = a1[r]; }
    }
    __syncthreads();
    if (ph == 0 && col < C) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float* o = part + (static_cast<long>(blockIdx.y) * R + r) * C + col;
            o[0] = ((a0[r] + red[0][r][2 * c2]) + red[1][r][2 * c2]) + red[2][r][2 * c2];
            o[1] = ((a1[r] + red[0][r][2 * c2 + 1]) + red[1][r][2 * c2 + 1]) + red[2][r][2 * c2 + 1];
        }
    }
}

// out: blocks.n == 0 -> [R][C]; else the listed blocks one after the other, block b as [cols_b][rows_b] (transposed: the
// [N_i][rank] layout of a LoRA up matrix), everything outside the blocks dropped.  scale: host factor * (optional) device scalar.
__global__ void k_wcolsum_final(const float* __restrict__ part, int slabs, int R, int C, const float* __restrict__ dev_scale, float host_scale,
                                WcsBlocks blocks, float* __restrict__ out) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= static_cast<long>(R) * C) return;
    const int r = static_cast<int>(i / C), c = static_cast<int>(i - static_cast<long>(r) * C);
    long dst = i;
    if (blocks.n > 0) {
        dst = -1;
        long off = 0;
        for (int b = 0; b < blocks.n; ++b) {
            if (r >= blocks.row0[b] && r < blocks.row0[b] + blocks.rows[b] && c >= blocks.col0[b] && c < blocks.col0[b] + blocks.cols[b])
                dst = off + static_cast<long>(c - blocks.col0[b]) * blocks.rows[b] + (r - blocks.row0[b]);
            off += static_cast<long>(blocks.rows[b]) * blocks.cols[b];
        }
        if (dst < 0) return;
    }
    // partial sums: one region per chunk of <= 16 weight rows, [slab][rows of the chunk][C] each
    const int chunk = r >> 4, rc = r & 15, Rc = min(16, R - 16 * chunk);
    const float* pc = part + static_cast<long>(slabs) * 16 * chunk * C;
    float acc = 0.f;
    for (int s = 0; s < slabs; ++s) acc += pc[(static_cast<long>(s) * Rc + rc) * C + c];
    out[dst] = acc * host_scale * (dev_scale ? dev_scale[0] : 1.0f);
}

// ---- gradient normalisation ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_amax(const float* __restrict__ x, long n, unsigned* __restrict__ state) {
    __shared__ float red[4];
    float m = 0.f;
    const long n4 = n / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    auto take = [&](float a) { a = fabsf(a); m = (a > m || a != a) ? a : m; };      // NaN propagates into the maximum (-> scale 1 downstream)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
        const float4 v = x4[i];
        take(v.x); take(v.y); take(v.z); take(v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) take(x[4 * n4 + threadIdx.x]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(m, o);
        m = (w > m || w != w) ? w : m;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {                            // one atomic per block (4096 per launch serialised on one address before)
#pragma unroll
        for (int i = 1; i < 4; ++i) m = (red[i] > m || red[i] != red[i]) ? red[i] : m;
        atomicMax(state, __float_as_uint(m));          // |x| >= 0: the bit pattern orders like the value (NaN > inf)
    }
}

__global__ void k_pow2_scale(float* state) {
    const float amax = state[0];
    float inv = 1.f, fwd = 1.f;
    if (amax > 0.f && amax < INFINITY) {               // (false for NaN)
        const int e = static_cast<int>(floorf(log2f(amax)));
        const int ec = max(-120, min(120, e));
        inv = exp2f(static_cast<float>(-ec));
        fwd = exp2f(static_cast<float>(ec));
    }
    state[1] = inv;
    state[2] = fwd;
}

template <typename SO>
__global__ void k_scale_f32(const float* __restrict__ x, long n, const float* __restrict__ scale, void* __restrict__ y) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    st_any<SO>(y, i, x[i] * scale[0]);
}

}  // namespace pf

using namespace pf;

extern "C" pf_status pf_attention_delta(const void* out, const void* dout, int dtype, int B, int H, int D, long nq,
                                        int ld, long bs, float* delta, void* stream) {
    PF_REQUIRE(out && dout && delta && B > 0 && H > 0 && nq > 0, "pf_attention_delta: bad arguments");
    PF_REQUIRE(D % 8 == 0 && ld % 8 == 0 && bs % 8 == 0 && aligned16(out) && aligned16(dout), "pf_attention_delta: D, ld, bs must be multiples of 8, pointers 16-byte aligned");
    const long total = static_cast<long>(B) * nq * H;
    PF_DISPATCH_16(dtype, "pf_attention_delta",
        hipLaunchKernelGGL(k_attn_delta<T>, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream),
                           static_cast<const unsigned short*>(out), static_cast<const unsigned short*>(dout), B, H, D, nq, ld, bs, delta));
    PF_CHECK_LAUNCH("pf_attention_delta");
    return PF_OK;
}

// LDS-staged kernels: whole 32-token tiles, 16-byte rows for the cooperative tile loads.  PF_ATTN_BWD_IMPL=direct: A/B switch
static bool attn_bwd_lds(const pf_attn_bwd_desc* d) {
    static const bool want_lds = [] { const char* e = getenv("PF_ATTN_BWD_IMPL"); return !(e && e[0] == 'd'); }();
    const bool tail = d->nq % 32 != 0 || d->nk % 32 != 0;
    return want_lds && !tail && d->qt_ld % 8 == 0 && d->kt_ld % 8 == 0 && d->dot_ld % 8 == 0 && d->qt_bs % 8 == 0 && d->kt_bs % 8 == 0 &&
           d->dot_bs % 8 == 0;
}

// Query split of the keys-stationary launch: towards ~512 blocks, at least 8 query tiles (256 queries) per block.
static int attn_bwd_qsplit(const pf_attn_bwd_desc* d) {
    static const int on = [] { const char* e = getenv("PF_ATTN_BWD_QSPLIT"); return e ? atoi(e) : 1; }();
    if (!on || !attn_bwd_lds(d) || (d->H * d->D) % 4 != 0) return 1;
    const long base = cdiv(d->nk, 128) * d->H * d->B;
    if (base >= 384) return 1;
    const long qs = std::min<long>(cdiv(512, base), (d->nq / 32) / 8);
    return static_cast<int>(std::max<long>(1, qs));
}

extern "C" size_t pf_attention_bwd_workspace_size(const pf_attn_bwd_desc* d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->nq <= 0 || d->nk <= 0 || (d->D != 32 && d->D != 64)) return 0;
    const int qs = attn_bwd_qsplit(d);
    return qs > 1 ? static_cast<size_t>(2) * qs * d->B * d->nk * d->H * d->D * sizeof(float) : 0;
}

extern "C" pf_status pf_attention_bwd(const pf_attn_bwd_desc* d, void* stream) {
    PF_REQUIRE(d, "pf_attention_bwd: null descriptor");
    PF_REQUIRE(d->q && d->k && d->v && d->dout && d->qt && d->kt && d->dot && d->dq && d->dk && d->dv && d->lse && d->delta,
               "pf_attention_bwd: null pointer");
    PF_REQUIRE(d->D == 32 || d->D == 64, "pf_attention_bwd: head dim %d unsupported (32 or 64)", d->D);
    PF_REQUIRE(d->B > 0 && d->H > 0 && d->nq > 0 && d->nk > 0, "pf_attention_bwd: bad sizes");
    PF_REQUIRE(d->q_ld % 8 == 0 && d->k_ld % 8 == 0 && d->v_ld % 8 == 0 && d->do_ld % 8 == 0, "pf_attention_bwd: row-major leading dimensions must be multiples of 8");
    PF_REQUIRE(d->qt_ld % 4 == 0 && d->kt_ld % 4 == 0 && d->dot_ld % 4 == 0 && d->qt_ld >= d->nq && d->dot_ld >= d->nq && d->kt_ld >= d->nk,
               "pf_attention_bwd: transposed leading dimensions must be multiples of 4 and cover the token count");
    PF_REQUIRE(d->dq_ld % 4 == 0 && d->dk_ld % 4 == 0 && d->dv_ld % 4 == 0, "pf_attention_bwd: output leading dimensions must be multiples of 4");
    PF_REQUIRE(d->q_bs % 8 == 0 && d->k_bs % 8 == 0 && d->v_bs % 8 == 0 && d->do_bs % 8 == 0 && d->qt_bs % 4 == 0 && d->kt_bs % 4 == 0 &&
               d->dot_bs % 4 == 0 && d->dq_bs % 4 == 0 && d->dk_bs % 4 == 0 && d->dv_bs % 4 == 0, "pf_attention_bwd: batch strides misaligned");
    PF_REQUIRE(aligned16(d->q) && aligned16(d->k) && aligned16(d->v) && aligned16(d->dout) && aligned16(d->qt) && aligned16(d->kt) &&
               aligned16(d->dot) && aligned16(d->dq) && aligned16(d->dk) && aligned16(d->dv) && aligned16(d->lse) && aligned16(d->delta),
               "pf_attention_bwd: pointers must be 16-byte aligned");
    PF_REQUIRE((d->bias == nullptr) == (d->flags == nullptr), "pf_attention_bwd: bias and flags come together");
    if (d->bias) {
        PF_REQUIRE(d->nk % 4 == 0 && d->bias_ld % 4 == 0 && d->bias_ld >= d->nk && aligned16(d->bias), "pf_attention_bwd: bias needs nk %% 4 == 0 and an aligned ld >= nk");
        PF_REQUIRE(d->flags_ld >= (d->nk + 31) / 32, "pf_attention_bwd: flags_ld too small");
    }
    AttnBwdParams p;
    auto u16 = [](const void* v) { return static_cast<const unsigned short*>(v); };
    p.q = u16(d->q); p.k = u16(d->k); p.v = u16(d->v); p.dout = u16(d->dout); p.qt = u16(d->qt); p.kt = u16(d->kt); p.dot = u16(d->dot);
    p.dq = static_cast<unsigned short*>(d->dq); p.dk = static_cast<unsigned short*>(d->dk); p.dv = static_cast<unsigned short*>(d->dv);
    p.H = d->H; p.nq = d->nq; p.nk = d->nk;
    p.q_ld = d->q_ld; p.k_ld = d->k_ld; p.v_ld = d->v_ld; p.do_ld = d->do_ld; p.qt_ld = d->qt_ld; p.kt_ld = d->kt_ld; p.dot_ld = d->dot_ld;
    p.dq_ld = d->dq_ld; p.dk_ld = d->dk_ld; p.dv_ld = d->dv_ld;
    p.q_bs = d->q_bs; p.k_bs = d->k_bs; p.v_bs = d->v_bs; p.do_bs = d->do_bs; p.qt_bs = d->qt_bs; p.kt_bs = d->kt_bs; p.dot_bs = d->dot_bs;
    p.dq_bs = d->dq_bs; p.dk_bs = d->dk_bs; p.dv_bs = d->dv_bs;
    p.scale = d->scale; p.scale_log2e = d->scale * LOG2E;
    p.bias = d->bias; p.bias_ld = d->bias_ld; p.flags = d->flags; p.flags_ld = d->flags_ld;
    p.lse = d->lse; p.delta = d->delta;
    hipStream_t st = as_stream(stream);
    const bool tail = d->nq % 32 != 0 || d->nk % 32 != 0;        // ragged token counts: guarded loads, masked tails
    const bool lds = attn_bwd_lds(d);
    // keys-stationary launch with few blocks (the 128 text keys of a cross-attention: H x B blocks walking all queries; one
    // panorama sample): the query range is split over several blocks, fp32 partial sums added in a fixed order afterwards
    const int qsplit = d->workspace ? attn_bwd_qsplit(d) : 1;
    p.qsplit = qsplit;
    p.part_k = p.part_v = nullptr;
    if (qsplit > 1) {
        const size_t half = static_cast<size_t>(qsplit) * d->B * d->nk * d->H * d->D * sizeof(float);
        PF_REQUIRE(d->workspace_bytes >= 2 * half && aligned16(d->workspace), "pf_attention_bwd: workspace too small or misaligned (pf_attention_bwd_workspace_size)");
        p.part_k = static_cast<float*>(d->workspace);
        p.part_v = p.part_k + half / sizeof(float);
    }
    const dim3 block(256), gq(cdiv(d->nq, 128), d->H, d->B), gk(cdiv(d->nk, 128) * qsplit, d->H, d->B);
#define PF_BWD(DD)                                                                                                          \
    do {                                                                                                                    \
        if (lds) { hipLaunchKernelGGL((k_attn_bwd_dq_lds<T, DD>), gq, block, 0, st, p); hipLaunchKernelGGL((k_attn_bwd_dkv_lds<T, DD>), gk, block, 0, st, p); } \
        else if (tail) { hipLaunchKernelGGL((k_attn_bwd_dq<T, DD, true>), gq, block, 0, st, p); hipLaunchKernelGGL((k_attn_bwd_dkv<T, DD, true>), gk, block, 0, st, p); } \
        else { hipLaunchKernelGGL((k_attn_bwd_dq<T, DD, false>), gq, block, 0, st, p); hipLaunchKernelGGL((k_attn_bwd_dkv<T, DD, false>), gk, block, 0, st, p); } \
    } while (0)
    PF_DISPATCH_16(d->dtype, "pf_attention_bwd", if (d->D == 64) PF_BWD(64); else PF_BWD(32));
#undef PF_BWD
    if (qsplit > 1) {
        const long total = static_cast<long>(d->B) * d->nk * (d->H * d->D / 4);
        PF_DISPATCH_16(d->dtype, "pf_attention_bwd",
            hipLaunchKernelGGL(k_attn_bwd_reduce<T>, dim3(cdiv(total, 256)), dim3(256), 0, st, p.part_k, p.part_v, qsplit, d->B, d->nk, d->H * d->D,
                               p.dk, p.dv, p.dk_ld, p.dv_ld, p.dk_bs, p.dv_bs));
    }
    PF_CHECK_LAUNCH("pf_attention_bwd");
    return PF_OK;
}

extern "C" int pf_layernorm_bwd_parts(long rows) {
    return static_cast<int>(std::max(1L, std::min<long>(LN_BWD_MAX_PARTS, cdiv(rows, 4))));
}

extern "C" pf_status pf_layernorm_bwd(const void* x, const float* pe, long pe_rows, int dtype, long rows, int C,
                                      const float* gamma, float eps, const float* dy, const float* dres, float* dx,
                                      float* partials, void* stream) {
    PF_REQUIRE(x && gamma && dy && dx && partials && rows > 0, "pf_layernorm_bwd: bad arguments");
    PF_REQUIRE(C > 0 && C % 8 == 0 && C <= 2048, "pf_layernorm_bwd: C=%d must be a multiple of 8, at most 2048", C);
    PF_REQUIRE(!pe || pe_rows > 0, "pf_layernorm_bwd: pe_rows must be positive");
    PF_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma) && (!pe || aligned16(pe)) && (!dres || aligned16(dres)),
               "pf_layernorm_bwd: pointers must be 16-byte aligned");
    const int parts = pf_layernorm_bwd_parts(rows);
    hipStream_t st = as_stream(stream);
    const dim3 grid(parts), block(256);
    if (dtype == PF_F32) hipLaunchKernelGGL(k_layernorm_bwd<float>, grid, block, 0, st, x, pe, pe_rows, rows, C, gamma, eps, dy, dres, dx, partials, parts);
    else if (dtype == PF_F16) hipLaunchKernelGGL(k_layernorm_bwd<F16>, grid, block, 0, st, x, pe, pe_rows, rows, C, gamma, eps, dy, dres, dx, partials, parts);
    else if (dtype == PF_BF16) hipLaunchKernelGGL(k_layernorm_bwd<Bf16>, grid, block, 0, st, x, pe, pe_rows, rows, C, gamma, eps, dy, dres, dx, partials, parts);
    else PF_REQUIRE(false, "pf_layernorm_bwd: unsupported dtype %d", dtype);
    PF_CHECK_LAUNCH("pf_layernorm_bwd");
    return PF_OK;
}

extern "C" pf_status pf_geglu_bwd(const void* u, const void* dg, int dtype, long rows, int inner, void* du, void* stream) {
    PF_REQUIRE(u && dg && du && rows > 0 && inner > 0 && inner % 8 == 0, "pf_geglu_bwd: bad arguments");
    PF_REQUIRE(aligned16(u) && aligned16(dg) && aligned16(du), "pf_geglu_bwd: pointers must be 16-byte aligned");
    const long total = rows * (inner / 8);
    PF_DISPATCH_16(dtype, "pf_geglu_bwd",
        hipLaunchKernelGGL(k_geglu_bwd<T>, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), static_cast<const unsigned short*>(u),
                           static_cast<const unsigned short*>(dg), total, inner, static_cast<unsigned short*>(du)));
    PF_CHECK_LAUNCH("pf_geglu_bwd");
    return PF_OK;
}

static int colsum_slabs(long rows) { return static_cast<int>(std::max(1L, std::min<long>(COLSUM_SLABS, cdiv(rows, 64)))); }

extern "C" size_t pf_colsum_workspace_size(long rows, int N) {
    if (rows <= 0 || N <= 0) return 0;
    return static_cast<size_t>(colsum_slabs(rows)) * N * sizeof(float);
}

extern "C" pf_status pf_colsum(const void* x, int dtype, long rows, int N, long ld, float* out, void* workspace,
                               size_t workspace_bytes, void* stream) {
    PF_REQUIRE(x && out && workspace && rows > 0 && N > 0 && ld >= N, "pf_colsum: bad arguments");
    PF_REQUIRE(workspace_bytes >= pf_colsum_workspace_size(rows, N), "pf_colsum: workspace too small (%zu < %zu bytes)", workspace_bytes,
               pf_colsum_workspace_size(rows, N));
    const int slabs = colsum_slabs(rows);
    const dim3 grid(cdiv(N, 64), slabs), block(256);
    hipStream_t st = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    if (dtype == PF_F32) hipLaunchKernelGGL(k_colsum_partial<AnyF32>, grid, block, 0, st, x, rows, N, ld, slabs, part);
    else if (dtype == PF_F16) hipLaunchKernelGGL(k_colsum_partial<AnyF16>, grid, block, 0, st, x, rows, N, ld, slabs, part);
    else if (dtype == PF_BF16) hipLaunchKernelGGL(k_colsum_partial<AnyBf16>, grid, block, 0, st, x, rows, N, ld, slabs, part);
    else PF_REQUIRE(false, "pf_colsum: unsupported dtype %d", dtype);
    hipLaunchKernelGGL(k_colsum_final, dim3(cdiv(N, 256)), dim3(256), 0, st, part, slabs, N, out);
    PF_CHECK_LAUNCH("pf_colsum");
    return PF_OK;
}

static int wcolsum_slabs(long T) { return static_cast<int>(std::max(1L, std::min<long>(64, T / 256))); }

extern "C" size_t pf_weighted_colsum_workspace_size(long T, int C, int R) {
    if (T <= 0 || C <= 0 || R <= 0) return 0;
    return static_cast<size_t>(wcolsum_slabs(T)) * R * C * sizeof(float);
}

extern "C" pf_status pf_weighted_colsum(const void* x, int dtype, long n_tok, int C, long ld, const float* w, int R, long w_ld,
                                        const float* dev_scale, float host_scale, const int* blocks, int n_blocks, float* out,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    PF_REQUIRE(x && w && out && workspace && n_tok > 0 && C > 0, "pf_weighted_colsum: bad arguments");
    PF_REQUIRE(R > 0 && R % 4 == 0 && w_ld >= n_tok, "pf_weighted_colsum: R = %d must be a positive multiple of 4 and w_ld >= T", R);
    PF_REQUIRE(C % 2 == 0 && ld % 2 == 0 && ld >= C && (reinterpret_cast<uintptr_t>(x) & 3) == 0, "pf_weighted_colsum: x needs an even width / row stride and 4-byte alignment");
    PF_REQUIRE(n_blocks >= 0 && n_blocks <= 4 && (n_blocks == 0 || blocks), "pf_weighted_colsum: at most 4 output blocks");
    PF_REQUIRE(workspace_bytes >= pf_weighted_colsum_workspace_size(n_tok, C, R), "pf_weighted_colsum: workspace too small");
    WcsBlocks b;
    b.n = n_blocks;
    for (int i = 0; i < 4; ++i) {
        b.row0[i] = i < n_blocks ? blocks[4 * i] : 0; b.rows[i] = i < n_blocks ? blocks[4 * i + 1] : 0;
        b.col0[i] = i < n_blocks ? blocks[4 * i + 2] : 0; b.cols[i] = i < n_blocks ? blocks[4 * i + 3] : 0;
        PF_REQUIRE(i >= n_blocks || (b.row0[i] >= 0 && b.rows[i] > 0 && b.row0[i] + b.rows[i] <= R && b.col0[i] >= 0 && b.cols[i] > 0 && b.col0[i] + b.cols[i] <= C),
                   "pf_weighted_colsum: block %d outside [R][C]", i);
    }
    const int slabs = wcolsum_slabs(n_tok);
    float* part = static_cast<float*>(workspace);
    hipStream_t st = as_stream(stream);
    const dim3 grid(cdiv(C, 128), slabs), block(256);
    // weight rows in chunks of <= 16 (the kernel keeps 2 accumulators per row in registers): x is re-read once per chunk
#define PF_WCS(RQ) hipLaunchKernelGGL((k_wcolsum_partial<T, RQ>), grid, block, 0, st, static_cast<const unsigned short*>(x), n_tok, C, ld, \
                                      w + static_cast<long>(r0) * w_ld, w_ld, slabs, part + static_cast<long>(slabs) * r0 * C)
    for (int r0 = 0; r0 < R; r0 += 16) {
        const int Rc = std::min(16, R - r0);
        PF_DISPATCH_16(dtype, "pf_weighted_colsum",
            if (Rc == 4) PF_WCS(1); else if (Rc == 8) PF_WCS(2); else if (Rc == 12) PF_WCS(3); else PF_WCS(4));
    }
#undef PF_WCS
    hipLaunchKernelGGL(k_wcolsum_final, dim3(cdiv(static_cast<long>(R) * C, 256)), dim3(256), 0, st, part, slabs, R, C, dev_scale, host_scale, b, out);
    PF_CHECK_LAUNCH("pf_weighted_colsum");
    return PF_OK;
}

extern "C" pf_status pf_amax_f32(const float* x, long n, float* state, int reset, void* stream) {
    PF_REQUIRE(x && state && n > 0, "pf_amax_f32: bad arguments");
    hipStream_t st = as_stream(stream);
    if (reset && hipMemsetAsync(state, 0, sizeof(float), st) != hipSuccess) {
        pf::set_error("pf_amax_f32: clearing the state failed");
        return PF_ERR_LAUNCH;
    }
    PF_REQUIRE(aligned16(x), "pf_amax_f32: x must be 16-byte aligned");
    const long blocks = std::min<long>(512, cdiv(n, 1024));
    hipLaunchKernelGGL(k_amax, dim3(std::max<long>(1, blocks)), dim3(256), 0, st, x, n, reinterpret_cast<unsigned*>(state));
    PF_CHECK_LAUNCH("pf_amax_f32");
    return PF_OK;
}

extern "C" pf_status pf_pow2_scale(float* state, void* stream) {
    PF_REQUIRE(state, "pf_pow2_scale: null state");
    hipLaunchKernelGGL(k_pow2_scale, dim3(1), dim3(1), 0, as_stream(stream), state);
    PF_CHECK_LAUNCH("pf_pow2_scale");
    return PF_OK;
}

extern "C" pf_status pf_scale_f32(const float* x, long n, const float* state, int index, int out_dtype, void* y, void* stream) {
    PF_REQUIRE(x && state && y && n > 0 && index >= 0 && index < 4, "pf_scale_f32: bad arguments");
    const dim3 grid(cdiv(n, 256)), block(256);
    hipStream_t st = as_stream(stream);
    if (out_dtype == PF_F32) hipLaunchKernelGGL(k_scale_f32<AnyF32>, grid, block, 0, st, x, n, state + index, y);
    else if (out_dtype == PF_F16) hipLaunchKernelGGL(k_scale_f32<AnyF16>, grid, block, 0, st, x, n, state + index, y);
    else if (out_dtype == PF_BF16) hipLaunchKernelGGL(k_scale_f32<AnyBf16>, grid, block, 0, st, x, n, state + index, y);
    else PF_REQUIRE(false, "pf_scale_f32: unsupported dtype %d", out_dtype);
    PF_CHECK_LAUNCH("pf_scale_f32");
    return PF_OK;
}

static int gn_bwd_chunks(int hw, int* pix_per_chunk) {
    int ppc = 64;
    while (ppc > 8 && cdiv(hw, ppc) < 16) ppc >>= 1;
    *pix_per_chunk = ppc;
    return static_cast<int>(cdiv(hw, ppc));
}

extern "C" size_t pf_groupnorm_bwd_workspace_size(int n_img, int hw, int groups) {
    if (n_img <= 0 || hw <= 0 || groups <= 0) return 0;
    int ppc;
    const int chunks = gn_bwd_chunks(hw, &ppc);
    return (static_cast<size_t>(n_img) * chunks * groups * 4 + static_cast<size_t>(n_img) * groups * 4) * sizeof(float);
}

extern "C" pf_status pf_groupnorm_bwd(const void* x0, int c0, const void* x1, int c1, int dtype, int n_img, int hw, int groups,
                                      float eps, const float* gamma, const float* scale, const float* shift, int act,
                                      const float* dy, const float* dres, float* dx0, float* dx1, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    PF_REQUIRE(x0 && gamma && scale && shift && dy && dx0 && workspace && n_img > 0 && hw > 0, "pf_groupnorm_bwd: bad arguments");
    if (!x1) c1 = 0;
    PF_REQUIRE(c1 == 0 || dx1, "pf_groupnorm_bwd: two sources need two gradient outputs");
    const int C = c0 + c1;
    PF_REQUIRE(c0 > 0 && c0 % 8 == 0 && c1 % 8 == 0 && groups > 0 && C % groups == 0, "pf_groupnorm_bwd: channels (%d,%d) must be multiples of 8 and divide into %d groups", c0, c1, groups);
    PF_REQUIRE(n_img <= 65535, "pf_groupnorm_bwd: at most 65535 images per call");
    PF_REQUIRE(act == 0 || act == 1, "pf_groupnorm_bwd: act must be 0 (none) or 1 (SiLU)");
    PF_REQUIRE(workspace_bytes >= pf_groupnorm_bwd_workspace_size(n_img, hw, groups), "pf_groupnorm_bwd: workspace too small");
    PF_REQUIRE(aligned16(x0) && (!x1 || aligned16(x1)) && aligned16(dy) && aligned16(dx0) && (!dx1 || aligned16(dx1)) && aligned16(gamma) &&
               aligned16(scale) && aligned16(shift) && (!dres || aligned16(dres)) && aligned16(workspace), "pf_groupnorm_bwd: pointers must be 16-byte aligned");
    int ppc;
    const int chunks = gn_bwd_chunks(hw, &ppc);
    float* partial = static_cast<float*>(workspace);
    float* stats = partial + static_cast<size_t>(n_img) * chunks * groups * 4;
    const int OCT = C / 8, OCTB = OCT < 256 ? OCT : 256, pix_par = 256 / OCTB;
    const size_t smem = static_cast<size_t>(4) * pix_par * C * sizeof(float);
    PF_REQUIRE(smem <= 64 * 1024, "pf_groupnorm_bwd: %d channels exceed the 64 KiB staging buffer", C);
    hipStream_t st = as_stream(stream);
    const dim3 g1(chunks, n_img), g3(cdiv(static_cast<long>(hw) * OCT, 256), n_img), block(256);
#define PF_GNB(TI) do {                                                                                                     \
        hipLaunchKernelGGL(k_gn_bwd_partial<TI>, g1, block, smem, st, x0, c0, x1, c1, hw, groups, ppc, gamma, scale, shift, act, dy, partial); \
        hipLaunchKernelGGL(k_gn_bwd_finalize, dim3(n_img), block, 0, st, partial, chunks, groups, C, hw, eps, stats);             \
        hipLaunchKernelGGL(k_gn_bwd_apply<TI>, g3, block, 0, st, x0, c0, x1, c1, hw, groups, gamma, scale, shift, act, dy, dres, stats, dx0, dx1); \
    } while (0)
    if (dtype == PF_F32) PF_GNB(float);
    else if (dtype == PF_F16) PF_GNB(F16);
    else if (dtype == PF_BF16) PF_GNB(Bf16);
    else PF_REQUIRE(false, "pf_groupnorm_bwd: unsupported dtype %d", dtype);
#undef PF_GNB
    PF_CHECK_LAUNCH("pf_groupnorm_bwd");
    return PF_OK;
}

extern "C" size_t pf_groupnorm_param_grads_workspace_size(int n_img, int hw, int C) {
    if (n_img <= 0 || hw <= 0 || C <= 0) return 0;
    int ppc;
    const long rows = static_cast<long>(n_img) * gn_bwd_chunks(hw, &ppc);
    return static_cast<size_t>(rows) * 2 * C * sizeof(float) + pf_colsum_workspace_size(rows, 2 * C);
}

extern "C" pf_status pf_groupnorm_param_grads(const void* x0, int c0, const void* x1, int c1, int dtype, int n_img, int hw,
                                              const float* scale, const float* shift, const float* unit_scale, const float* unit_shift,
                                              int act, const float* dy, float* dgamma_dbeta, void* workspace, size_t workspace_bytes,
                                              void* stream) {
    PF_REQUIRE(x0 && scale && shift && unit_scale && unit_shift && dy && dgamma_dbeta && workspace && n_img > 0 && hw > 0,
               "pf_groupnorm_param_grads: bad arguments");
    if (!x1) c1 = 0;
    const int C = c0 + c1;
    PF_REQUIRE(c0 > 0 && c0 % 8 == 0 && c1 % 8 == 0, "pf_groupnorm_param_grads: channels (%d,%d) must be multiples of 8", c0, c1);
    PF_REQUIRE(n_img <= 65535, "pf_groupnorm_param_grads: at most 65535 images per call");
    PF_REQUIRE(act == 0 || act == 1, "pf_groupnorm_param_grads: act must be 0 (none) or 1 (SiLU)");
    PF_REQUIRE(workspace_bytes >= pf_groupnorm_param_grads_workspace_size(n_img, hw, C), "pf_groupnorm_param_grads: workspace too small");
    PF_REQUIRE(aligned16(x0) && (!x1 || aligned16(x1)) && aligned16(dy) && aligned16(scale) && aligned16(shift) && aligned16(unit_scale) &&
               aligned16(unit_shift) && aligned16(workspace), "pf_groupnorm_param_grads: pointers must be 16-byte aligned");
    int ppc;
    const int chunks = gn_bwd_chunks(hw, &ppc);
    const long rows = static_cast<long>(n_img) * chunks;
    float* partial = static_cast<float*>(workspace);
    const int OCT = C / 8, OCTB = OCT < 256 ? OCT : 256, pix_par = 256 / OCTB;
    const size_t smem = static_cast<size_t>(2) * pix_par * C * sizeof(float);
    PF_REQUIRE(smem <= 64 * 1024, "pf_groupnorm_param_grads: %d channels exceed the 64 KiB staging buffer", C);
    hipStream_t st = as_stream(stream);
    const dim3 grid(chunks, n_img), block(256);
#define PF_GNP(TI) hipLaunchKernelGGL(k_gn_param_partial<TI>, grid, block, smem, st, x0, c0, x1, c1, hw, ppc, scale, shift, unit_scale, unit_shift, act, dy, partial)
    if (dtype == PF_F32) PF_GNP(float);
    else if (dtype == PF_F16) PF_GNP(F16);
    else if (dtype == PF_BF16) PF_GNP(Bf16);
    else PF_REQUIRE(false, "pf_groupnorm_param_grads: unsupported dtype %d", dtype);
#undef PF_GNP
    PF_CHECK_LAUNCH("pf_groupnorm_param_grads");
    const size_t used = static_cast<size_t>(rows) * 2 * C * sizeof(float);
    return pf_colsum(partial, PF_F32, rows, 2 * C, 2 * C, dgamma_dbeta, reinterpret_cast<char*>(workspace) + used, workspace_bytes - used, stream);
}

extern "C" pf_status pf_silu_bwd(const void* z, int dtype, const float* dy, long n, float* dz, void* stream) {
    PF_REQUIRE(z && dy && dz && n > 0, "pf_silu_bwd: bad arguments");
    const dim3 grid(cdiv(n, 256)), block(256);
    hipStream_t st = as_stream(stream);
    if (dtype == PF_F32) hipLaunchKernelGGL(k_silu_bwd<AnyF32>, grid, block, 0, st, z, dy, n, dz);
    else if (dtype == PF_F16) hipLaunchKernelGGL(k_silu_bwd<AnyF16>, grid, block, 0, st, z, dy, n, dz);
    else if (dtype == PF_BF16) hipLaunchKernelGGL(k_silu_bwd<AnyBf16>, grid, block, 0, st, z, dy, n, dz);
    else PF_REQUIRE(false, "pf_silu_bwd: unsupported dtype %d", dtype);
    PF_CHECK_LAUNCH("pf_silu_bwd");
    return PF_OK;
}

extern "C" pf_status pf_im2col3(const void* x, int dtype, int n, int h, int w, int C, int stride, void* y, void* stream) {
    PF_REQUIRE(x && y && x != y && n > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0, "pf_im2col3: bad arguments");
    PF_REQUIRE(dtype == PF_F16 || dtype == PF_BF16, "pf_im2col3: 16-bit tensors only");
    PF_REQUIRE(stride == 1 || stride == 2, "pf_im2col3: stride must be 1 or 2");
    PF_REQUIRE(aligned16(x) && aligned16(y), "pf_im2col3: pointers must be 16-byte aligned");
    const int ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
    const long total = static_cast<long>(n) * ho * wo * 9 * (C / 8);
    PF_REQUIRE(cdiv(total, 256) < (1L << 31), "pf_im2col3: tensor too large for one launch");
    hipLaunchKernelGGL(k_im2col3, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), static_cast<const u16x8*>(x), n, h, w, C / 8, stride, ho, wo,
                       static_cast<u16x8*>(y));
    PF_CHECK_LAUNCH("pf_im2col3");
    return PF_OK;
}

extern "C" pf_status pf_lora_fold(const float* w, const float* up, const float* down, int N, int K, int r, float scale, int dtype,
                                  void* out, long out_ld, void* out_t, long out_t_ld, void* d_out, long d_ld, void* u_out, long u_ld,
                                  void* stream) {
    PF_REQUIRE(w && out && N > 0 && K > 0, "pf_lora_fold: bad arguments");
    PF_REQUIRE(r >= 0 && r <= LORA_MAX_RANK && (r == 0 || (up && down)), "pf_lora_fold: rank %d unsupported (0..%d, with both matrices)", r, LORA_MAX_RANK);
    PF_REQUIRE(out_ld >= K && (!out_t || out_t_ld >= N) && (!d_out || d_ld >= K) && (!u_out || u_ld >= N), "pf_lora_fold: leading dimensions too small");
    PF_REQUIRE(cdiv(N, 64) <= 65535, "pf_lora_fold: too many rows");
    const dim3 grid(cdiv(K, 64), cdiv(N, 64)), block(256);
    PF_DISPATCH_16(dtype, "pf_lora_fold",
        hipLaunchKernelGGL(k_lora_fold<T>, grid, block, 0, as_stream(stream), w, up, down, N, K, r, scale, static_cast<unsigned short*>(out), out_ld,
                           static_cast<unsigned short*>(out_t), out_t_ld, static_cast<unsigned short*>(d_out), d_ld,
                           static_cast<unsigned short*>(u_out), u_ld));
    PF_CHECK_LAUNCH("pf_lora_fold");
    return PF_OK;
}

extern "C" pf_status pf_zero_insert2(const void* x, int dtype, int n, int h, int w, int C, void* y, void* stream) {
    PF_REQUIRE(x && y && x != y && n > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0, "pf_zero_insert2: bad arguments");
    PF_REQUIRE(dtype == PF_F16 || dtype == PF_BF16, "pf_zero_insert2: 16-bit tensors only");
    PF_REQUIRE(aligned16(x) && aligned16(y), "pf_zero_insert2: pointers must be 16-byte aligned");
    const long total = static_cast<long>(n) * 4 * h * w * (C / 8);
    hipLaunchKernelGGL(k_zero_insert2, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), static_cast<const u16x8*>(x), n, h, w, C / 8,
                       static_cast<u16x8*>(y));
    PF_CHECK_LAUNCH("pf_zero_insert2");
    return PF_OK;
}

extern "C" pf_status pf_sum2x2(const float* x, int n, int h, int w, int C, float* y, void* stream) {
    PF_REQUIRE(x && y && x != y && n > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "pf_sum2x2: bad arguments");
    PF_REQUIRE(aligned16(x) && aligned16(y), "pf_sum2x2: pointers must be 16-byte aligned");
    const long total = static_cast<long>(n) * h * w * (C / 4);
    hipLaunchKernelGGL(k_sum2x2, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(x), n, h, w, C / 4,
                       reinterpret_cast<float4*>(y));
    PF_CHECK_LAUNCH("pf_sum2x2");
    return PF_OK;
}

extern "C" pf_status pf_pad_width_bwd(const float* dy, int n, int h, int w, int C, int pad, float* dx, void* stream) {
    PF_REQUIRE(dy && dx && dy != dx && n > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "pf_pad_width_bwd: bad arguments");
    PF_REQUIRE(pad >= 0 && pad <= w, "pf_pad_width_bwd: pad must be in [0, w]");
    PF_REQUIRE(aligned16(dy) && aligned16(dx), "pf_pad_width_bwd: pointers must be 16-byte aligned");
    const long rows = static_cast<long>(n) * h, total = rows * w * (C / 4);
    hipLaunchKernelGGL(k_pad_width_bwd, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(dy), rows, w, pad,
                       C / 4, reinterpret_cast<float4*>(dx));
    PF_CHECK_LAUNCH("pf_pad_width_bwd");
    return PF_OK;
}

extern "C" pf_status pf_crop_width_bwd(const float* dy, int n, int h, int w, int C, int crop, float* dx, void* stream) {
    PF_REQUIRE(dy && dx && dy != dx && n > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "pf_crop_width_bwd: bad arguments");
    PF_REQUIRE(crop >= 0 && 2 * crop < w, "pf_crop_width_bwd: crop too large");
    PF_REQUIRE(aligned16(dy) && aligned16(dx), "pf_crop_width_bwd: pointers must be 16-byte aligned");
    const long rows = static_cast<long>(n) * h, total = rows * w * (C / 4);
    hipLaunchKernelGGL(k_crop_width_bwd, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(dy), rows, w, crop,
                       C / 4, reinterpret_cast<float4*>(dx));
    PF_CHECK_LAUNCH("pf_crop_width_bwd");
    return PF_OK;
}

extern "C" pf_status pf_transpose_tokens(const void* x, int dtype, int B, long T, int C, void* y, void* stream) {
    PF_REQUIRE(x && y && x != y && B > 0 && T > 0 && C > 0, "pf_transpose_tokens: bad arguments");
    PF_REQUIRE(dtype == PF_F16 || dtype == PF_BF16, "pf_transpose_tokens: 16-bit tensors only");
    PF_REQUIRE(C % 4 == 0 && aligned16(x) && aligned16(y), "pf_transpose_tokens: C must be a multiple of 4, pointers 16-byte aligned");
    PF_REQUIRE(B <= 65535 && cdiv(C, 64) <= 65535, "pf_transpose_tokens: too many batches / channel tiles");
    hipLaunchKernelGGL(k_transpose16, dim3(cdiv(T, 64), cdiv(C, 64), B), dim3(256), 0, as_stream(stream), static_cast<const unsigned short*>(x), T, C,
                       static_cast<unsigned short*>(y));
    PF_CHECK_LAUNCH("pf_transpose_tokens");
    return PF_OK;
}
