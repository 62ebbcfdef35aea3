// Flash attention on MFMA for gfx950 (CDNA4): UNet self / text cross attention (head dim 64)
// and the EPA pano<->view cross attention (head dim 32) with its sparse correspondence bias.
//
//   out[b][i][h*D+:] = softmax_j(scale * q_i.k_j + bias[i][j]) v_j
//
// One wavefront owns 32 query rows and walks the keys in tiles of 32 with an online softmax.
// Both products are computed TRANSPOSED so that a lane owns one query column:
//   S^T = K Q^T  : v_mfma_f32_32x32x16 with A = K tile (rows = keys), B = Q^T.  Lane (q, hi)
//                  then holds 16 of the 32 scores of query q -> row max / row sum are 15 local
//                  ops + one exchange with lane^32, no LDS.
//   O^T += V^T P^T: A = V^T tile (rows = head dim, keys contiguous -- the V projection is written
//                  transposed by the GEMM), B = P^T straight from the score registers: the k-slot
//                  order of the MFMA is a permutation of the keys that is applied identically to
//                  both operands, so no cross-lane movement of P is needed.
// The additive bias is a dense fp32 table of (mask + 1) that is zero for 98-99 % of its entries;
// a byte flag per 32x32 tile says whether the tile has to be read at all.
//
// Replaces xformers.ops.memory_efficient_attention (models/modules/transformer.py:57-74) and the
// diffusers AttnProcessor (baddbmm + softmax + bmm) inside unet.*.attentions[j]
// (models/pano/MVGenModel.py:104,116,185,190,227,241).
#include "pf_common.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

#ifndef PF_ATTN_BUFLOAD
#define PF_ATTN_BUFLOAD 1
#endif
namespace pf {

struct AttnParams {
    const unsigned short* q; const unsigned short* k; const unsigned short* vt; unsigned short* out;
    int H, nq, nk;
    int q_ld, k_ld, vt_ld, o_ld;
    long q_bs, k_bs, vt_bs, o_bs;
    float scale_log2e;
    const float* bias; long bias_ld; const uint8_t* flags; int flags_ld;
    float* lse;
    int pp_role;                 // k_attention_pp: how a wave finds its phase group (PF_ATTENTION_PP_ROLE, see the kernel)
    int xcd_map;                 // k_attention_lds: 1 = heads pinned to XCDs (PF_ATTENTION_XCD, default), 0 = plain block order
    int pp_prio;                 // k_attention_pp: wave priorities (PF_ATTENTION_PP_PRIO): 0 none, 1 group B static 1, 2 raised inside matrix segments
    int steady2;                 // k_attention_lds (non-pipelined form): 1 = branch-free two-tile steady-state loop (PF_ATTENTION_STEADY2, default)
    int split_tiles;             // k_attention_lds (non-pipelined form): > 0 = the key range is split over gridDim.y workgroups of split_tiles (EVEN) key tiles each;
    long split_o, split_lse;     // split s writes its NORMALISED output at out + s * split_o and its log-sum-exp at lse + s * split_lse (k_attention_merge combines)
};

template <typename T, int D>
__global__ __launch_bounds__(256) void k_attention(const AttnParams p) {
    constexpr int KS = D / 16;      // k-slabs of the QK^T product
    constexpr int DB = D / 32;      // 32-row blocks of O^T
    typedef typename Mfma32<T>::frag frag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= p.nq) return;
    const int h = blockIdx.y;
    const long b = blockIdx.z;
    const unsigned short* qp = p.q + b * p.q_bs + h * D;
    const unsigned short* kp = p.k + b * p.k_bs + h * D;
    const unsigned short* vp = p.vt + b * p.vt_bs + static_cast<long>(h) * D * p.vt_ld;
    const int qrow = min(q0 + ql, p.nq - 1);

    frag qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        qf[s] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(qp + static_cast<long>(qrow) * p.q_ld + 16 * s + 8 * hi));

    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = (p.nk + 31) / 32;
    u16x8 kcur[KS], knext[KS];
    auto load_k = [&](int kt, u16x8 (&dst)[KS]) {
        const int krow = min(kt * 32 + ql, p.nk - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s)
            dst[s] = *reinterpret_cast<const u16x8*>(kp + static_cast<long>(krow) * p.k_ld + 16 * s + 8 * hi);
    };
    load_k(0, kcur);
    const uint8_t* flag_row = p.flags ? p.flags + static_cast<long>(q0 >> 5) * p.flags_ld : nullptr;
    const float* bias_row = p.bias ? p.bias + static_cast<long>(qrow) * p.bias_ld : nullptr;

    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = Mfma32<T>::run(__builtin_bit_cast(frag, kcur[ks]), qf[ks], s);
        if (kt + 1 < nkt) load_k(kt + 1, knext);

        // V^T fragments for this key tile (issued early, consumed after the softmax)
        u16x8 vf[2][DB];
        const bool tail = k0 + 32 > p.nk;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const unsigned short* vrow = vp + static_cast<long>(d * 32 + ql) * p.vt_ld + k0 + 16 * s2 + 4 * hi;
                const u16x4 lo = *reinterpret_cast<const u16x4*>(vrow);
                const u16x4 up = *reinterpret_cast<const u16x4*>(vrow + 8);
                u16x8 v = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
                if (tail) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int key = k0 + 16 * s2 + 4 * hi + (j & 3) + 8 * (j >> 2);
                        if (key >= p.nk) v[j] = 0;
                    }
                }
                vf[s2][d] = v;
            }

        float sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = s[r] * p.scale_log2e;
        if (flag_row && flag_row[kt]) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key = k0 + 8 * g + 4 * hi;
                if (key + 3 < p.nk) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias_row + key);
                    sv[4 * g + 0] += bv.x * 1.44269504088896340736f;
                    sv[4 * g + 1] += bv.y * 1.44269504088896340736f;
                    sv[4 * g + 2] += bv.z * 1.44269504088896340736f;
                    sv[4 * g + 3] += bv.w * 1.44269504088896340736f;
                }
            }
        }
        if (tail) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (key >= p.nk) sv[r] = -INFINITY;
            }
        }
        float mt = sv[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sv[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = exp2f(m_run - m_new);
        float ls = 0.f;
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pv[r] = exp2f(sv[r] - m_new);
            ls += pv[r];
        }
        ls += __shfl_xor(ls, 32);
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

        u16x8 pb[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[s2][j] = from_f32<T>(pv[8 * s2 + j]);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int d = 0; d < DB; ++d)
                o[d] = Mfma32<T>::run(__builtin_bit_cast(frag, vf[s2][d]), __builtin_bit_cast(frag, pb[s2]), o[d]);

        if (kt + 1 < nkt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kcur[ks] = knext[ks];
        }
    }

    if (q0 + ql < p.nq) {
        const float inv = 1.0f / l_run;
        unsigned short* op = p.out + b * p.o_bs + static_cast<long>(q0 + ql) * p.o_ld + h * D;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u16x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(o[d][4 * g + e] * inv);
                *reinterpret_cast<u16x4*>(op + d * 32 + 8 * g + 4 * hi) = w;
            }
        if (p.lse && hi == 0) p.lse[(b * p.H + h) * p.nq + q0 + ql] = m_run + __log2f(l_run);
    }
}

// ---- LDS-staged variant (default) ----------------------------------------------------------------
// Same math and register layout as k_attention, but the K tile [64 keys][D] and the V^T tile
// [D][64 keys] of a step are fetched ONCE per workgroup with coalesced 16-byte loads (a key row /
// 8 keys of a V^T row per lane), staged through LDS and shared by the 4 wavefronts; double buffered
// with the next tile's global loads in flight during the MFMAs (one barrier per 64 keys).  The direct
// variant issues fragment-shaped loads (32 different cache lines per instruction) from every wave.
//   K rows are 2*D bytes; the 16-B chunk index is XOR-swizzled so ds_read_b128 fragment reads are
//   conflict free (128-B rows: ^(row>>1)&7, 64-B rows: ^(row>>2)&3).
//   V^T rows are padded to 136 B (34 banks): the 32 d-rows read by a half-wave with ds_read_b64 hit
//   distinct bank pairs.
// One online-softmax update per 64 keys.
// PIPE: scores of tile j+1 held in registers while the softmax of tile j runs (three LDS buffers, two
// waves per SIMD at D = 64).  !PIPE: one score array, two LDS buffers, OCC waves per SIMD -- the
// MFMA / vector-ALU overlap then comes from the other waves of the SIMD instead of from inside a wave.
template <typename T, int D, bool BIAS, bool PIPE = true, int OCC = (D == 64 ? 2 : 3), bool MSUM = false>
__global__ __launch_bounds__(256, OCC) void k_attention_lds(const AttnParams p) {
    // (256, 2): at most 256 registers per lane -> the MFMA accumulators live in the VGPR file; with the
    // default budget the compiler parks S and O in AGPRs and pays ~145 v_accvgpr moves per key tile.
    constexpr int KS = D / 16, DB = D / 32, KT = 64;
    constexpr int KCHUNKS = D / 8;                     // 16-B chunks per K row
    constexpr int VROW = KT + 4;                       // elements per padded V^T row (136 B)
    constexpr int K_ELEMS = KT * D, V_ELEMS = D * VROW;
    constexpr int KCH = KT * KCHUNKS / 256, VCH = D * (KT / 8) / 256;
    typedef typename Mfma32<T>::frag frag;
    constexpr int NBUF = PIPE ? 3 : 2;
    __shared__ __attribute__((aligned(16))) unsigned short smem[NBUF * (K_ELEMS + V_ELEMS)];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    // 1-D grid, XCD-aware (round 5): block id l runs on XCD l % 8; when the (batch, head) pairs divide by 8, pair g lives on XCD g % 8
    // with its query blocks back to back there -- a head's K / V^T is fetched into ONE L2 instead of eight
    const int nqb = (p.nq + 127) / 128, BH = static_cast<int>(gridDim.x) / nqb;
    int qb, bh;
    if ((BH & 7) == 0 && p.xcd_map) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bh = xcd + 8 * (idx / nqb);
        qb = idx - (idx / nqb) * nqb;
    } else {
        bh = blockIdx.x / nqb;
        qb = blockIdx.x - bh * nqb;
    }
    const int q0 = (qb * 4 + wave) * 32;
    const int h = bh % p.H;
    const long b = bh / p.H;
    const unsigned short* qp = p.q + b * p.q_bs + h * D;
    const unsigned short* kp = p.k + b * p.k_bs + h * D;
    const unsigned short* vp = p.vt + b * p.vt_bs + static_cast<long>(h) * D * p.vt_ld;
    const int qrow = min(q0 + ql, p.nq - 1);

    frag qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        qf[s] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(qp + static_cast<long>(qrow) * p.q_ld + 16 * s + 8 * hi));

    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;              // running max of the RAW scores (q.k), running sum
    // MSUM (round 5, learnt on k_attention_pp below): the softmax denominator as a third accumulator block osum^T += 1 P^T on the
    // MATRIX pipe (A operand = ones; every row is sum_k P[q][k] over both lane halves) instead of 16 packed adds + a lane exchange per
    // key tile on the vector side, which is the side that bounds the kernel; the sum is then over the same 16-bit P the numerator uses
    // (the lse output of the training forward keeps the fp32 sums: MSUM is never launched with p.lse).
    f32x16 osum;
    u16x8 ones8;
    if constexpr (MSUM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) osum[r] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones8[e] = from_f32<T>(1.0f);
        asm volatile("" : "+v"(ones8));
    }
    const float c2 = p.scale_log2e;                    // softmax(scale * s) = exp2(c2 * s - c2 * max)

    auto k_off = [](int row, int chunk) {
        const int sw = (KCHUNKS == 8) ? ((row >> 1) & 7) : ((row >> 2) & 3);
        return row * D + ((chunk ^ sw) << 3);
    };

    // key tiles [jb, nkt) of this workgroup: all of them, or split blockIdx.y of a split key range (an EVEN number of tiles per split: the LDS buffer
    // of tile j is j & 1 in the non-pipelined form, which is the only one launched with a split)
    const int nkt_all = (p.nk + KT - 1) / KT;
    const int jb = p.split_tiles > 0 ? static_cast<int>(blockIdx.y) * p.split_tiles : 0;
    const int nkt = p.split_tiles > 0 ? min(nkt_all, jb + p.split_tiles) : nkt_all;
    u16x8 kreg[KCH], vreg[VCH];
    // staging ownership (fixed per thread): K chunk c = t + 256 i -> (row c / KCHUNKS, chunk c % KCHUNKS);
    // V^T chunk c -> (d = c >> 3, 8 keys starting at (c & 7) * 8)
    // Full tiles are fetched through buffer descriptors (base = this (batch, head)'s K / V^T, SGPR) + a per-thread 32-bit offset
    // computed once + a per-tile SCALAR offset: no 64-bit vector pointer arithmetic per key tile (four v_lshl_add_u64 and eight
    // address registers in the global-load form, in a kernel bound by each wave's instruction issue).  -DPF_ATTN_BUFLOAD=0: A/B build.
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    auto uniform_ptr = [](const unsigned short* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
        const unsigned hi_ = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
        return reinterpret_cast<unsigned short*>(static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi_) << 32));
    };
    // (the launcher checks that the slices fit 32-bit offsets; the register-pipelined EPA form measured 3-5 % SLOWER with it on its
    // view-query direction and keeps the global loads)
    constexpr bool bufload = PF_ATTN_BUFLOAD != 0 && !PIPE;
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(kp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(vp), 0, 0x7FFFFFFF, 0x00020000);
    unsigned kvoff[KCH], vvoff[VCH];                   // bytes
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const int c = t + 256 * i;
        kvoff[i] = static_cast<unsigned>((c / KCHUNKS) * p.k_ld + (c % KCHUNKS) * 8) * 2u;
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
        const int c = t + 256 * i;
        vvoff[i] = static_cast<unsigned>((c >> 3) * p.vt_ld + (c & 7) * 8) * 2u;
    }
    auto stage_load = [&](int j, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const int k0 = j * KT;
        if constexpr (!TAIL && bufload) {
            const int sk = __builtin_amdgcn_readfirstlane(k0 * p.k_ld * 2), sv_ = __builtin_amdgcn_readfirstlane(k0 * 2);
#pragma unroll
            for (int i = 0; i < KCH; ++i) kreg[i] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_k, kvoff[i], sk, 0));
#pragma unroll
            for (int i = 0; i < VCH; ++i) vreg[i] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_v, vvoff[i], sv_, 0));
            return;
        }
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = t + 256 * i, row = c / KCHUNKS, chunk = c % KCHUNKS;
            if (!TAIL || k0 + row < p.nk)
                kreg[i] = *reinterpret_cast<const u16x8*>(kp + static_cast<long>(k0 + row) * p.k_ld + chunk * 8);
            else
                kreg[i] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int c = t + 256 * i, d = c >> 3, key0 = k0 + (c & 7) * 8;
            if (!TAIL || key0 + 8 <= p.vt_ld) {
                u16x8 v = *reinterpret_cast<const u16x8*>(vp + static_cast<long>(d) * p.vt_ld + key0);
                if (TAIL) {                            // padding of V^T may hold anything (0 * NaN = NaN)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (key0 + e >= p.nk) v[e] = 0;
                }
                vreg[i] = v;
            } else {
                vreg[i] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    };
    auto stage_store = [&](int buf) {
        unsigned short* Ks = smem + buf * (K_ELEMS + V_ELEMS);
        unsigned short* Vs = Ks + K_ELEMS;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = t + 256 * i;
            *reinterpret_cast<u16x8*>(Ks + k_off(c / KCHUNKS, c % KCHUNKS)) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int c = t + 256 * i;
            unsigned short* dst = Vs + (c >> 3) * VROW + (c & 7) * 8;     // 8-byte aligned (136-B rows)
            const u16x8 v = vreg[i];
            *reinterpret_cast<u16x4*>(dst) = u16x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<u16x4*>(dst + 4) = u16x4{v[4], v[5], v[6], v[7]};
        }
    };

    const uint8_t* flag_row = BIAS ? p.flags + static_cast<long>(min(q0, p.nq - 1) >> 5) * p.flags_ld : nullptr;
    const float* bias_row = BIAS ? p.bias + static_cast<long>(qrow) * p.bias_ld : nullptr;

    // Software pipeline over key tiles (three LDS buffers, tiles j and j+1 resident):
    //   scores(j+1) = K_{j+1} Q^T  (8 MFMAs)  ||  softmax of scores(j) (vector ALU)  ->  O^T += V_j^T P^T (8 MFMAs)
    // so the matrix pipe has independent work while the softmax of the previous tile runs; without it
    // a wave alternates strictly between MFMA and VALU phases and, with two waves per SIMD, both pipes
    // idle a good part of the time (measured: MFMA 30 %, VALU 55-65 % busy).
    auto scores = [&](int j, int buf, float (&sv)[2][16], auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const int k0 = j * KT;
        const unsigned short* Ks = smem + buf * (K_ELEMS + V_ELEMS);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#ifdef PF_ATTN_ABL_NOLDS          /* timing-only ablation (make attn_ablate): fragments from registers instead of LDS */
                u16x8 kf = {1, 2, 3, 4, 5, 6, 7, static_cast<unsigned short>(lane + ks)};
                asm volatile("" : "+v"(kf));
#else
                const u16x8 kf = *reinterpret_cast<const u16x8*>(Ks + k_off(hh * 32 + ql, 2 * ks + hi));
#endif
                s = Mfma32<T>::run(__builtin_bit_cast(frag, kf), qf[ks], s);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[hh][r] = s[r];
            const int kb = k0 + hh * 32;
            if (BIAS && (!TAIL || kb < p.nk) && flag_row[kb >> 5]) {
                const float inv_c = 1.44269504088896340736f / c2;      // additive bias in units of the raw score
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int key = kb + 8 * g + 4 * hi;
                    if (!TAIL || key + 3 < p.nk) {
                        const float4 bv = *reinterpret_cast<const float4*>(bias_row + key);
                        sv[hh][4 * g + 0] += bv.x * inv_c;
                        sv[hh][4 * g + 1] += bv.y * inv_c;
                        sv[hh][4 * g + 2] += bv.z * inv_c;
                        sv[hh][4 * g + 3] += bv.w * inv_c;
                    }
                }
            }
            if (TAIL) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.nk) sv[hh][r] = -INFINITY;
            }
        }
    };
    auto softmax_pv = [&](int buf, float (&sv)[2][16]) {
        const unsigned short* Vs = smem + buf * (K_ELEMS + V_ELEMS) + K_ELEMS;
        float mt = -INFINITY;                            // (a constant first operand: no canonicalising v_max of the first two scores)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sv[hh][r]);
#ifdef PF_ATTN_BPERMUTE
        mt = fmaxf(mt, __shfl_xor(mt, 32));
#else
        {   // lanes l <-> l + 32 by v_permlane32_swap (gfx950) instead of ds_bpermute: no LDS round trip in the in-order wave
            float a_ = mt, b_ = mt;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
            mt = fmaxf(a_, b_);
        }
#endif
        {   // Deferred rescale: the running maximum is advanced (and O / l rescaled by exp2(c2 * (old - new))) only when
            // some row of the wavefront grew by more than 2^DEFER_LOG2 in the exponent; otherwise the STALE maximum stays
            // the reference point and this tile's probabilities are bounded by 2^DEFER_LOG2 instead of 1 -- exact
            // arithmetic either way, the final normalisation divides by the sum taken with the same reference.  On the
            // benchmark's data the branch is taken for the first tile or two of a row: 33 vector instructions less per
            // key tile (self-attention 64^2 +3 %, panorama 8192^2 +11 %, EPA +4 %, same-box A/B).  The branch is
            // wave-uniform (ballot); PF_ATTN_EAGER_RESCALE builds the unconditional form.
#ifndef PF_ATTN_EAGER_RESCALE
            constexpr float DEFER_LOG2 = 8.0f;
            const float m_new = fmaxf(m_run, mt);
            const bool grow = (m_new - m_run) * c2 > DEFER_LOG2;             // (first tile: -inf -> always)
            if (__builtin_amdgcn_ballot_w64(grow) != 0)
#else
            const float m_new = fmaxf(m_run, mt);
#endif
            {
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                if constexpr (MSUM) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) osum[r] *= alpha;
                }
            }
        }
        // p = exp2(c2 * s - c2 * max), two scores per instruction (v_pk_fma_f32), two running sums
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        const f32x2 c22 = {c2, c2}, mc2 = {m_run * c2, m_run * c2};
        f32x2 ls2 = {0.f, 0.f};
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 x = f32x2{sv[hh][r], sv[hh][r + 1]} * c22 - mc2;
                const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                sv[hh][r] = e[0];
                sv[hh][r + 1] = e[1];
                if constexpr (!MSUM) ls2 += e;
            }
        if constexpr (!MSUM) {
        float ls = ls2[0] + ls2[1];
#ifdef PF_ATTN_BPERMUTE
        ls += __shfl_xor(ls, 32);
#else
        {
            float a_ = ls, b_ = ls;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
            ls = a_ + b_;
        }
#endif
        l_run += ls;
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u16x8 pb;
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[e] = from_f32<T>(sv[hh][8 * s2 + e]);
#pragma unroll
                for (int d = 0; d < DB; ++d) {
#ifdef PF_ATTN_ABL_NOLDS
                    u16x8 vf = {1, 2, 3, 4, 5, 6, 7, static_cast<unsigned short>(lane + d)};
                    asm volatile("" : "+v"(vf));
#else
                    const unsigned short* vrow = Vs + (d * 32 + ql) * VROW + hh * 32 + 16 * s2 + 4 * hi;
                    const u16x4 lo = *reinterpret_cast<const u16x4*>(vrow);
                    const u16x4 up = *reinterpret_cast<const u16x4*>(vrow + 8);
                    const u16x8 vf = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
#endif
                    o[d] = Mfma32<T>::run(__builtin_bit_cast(frag, vf), __builtin_bit_cast(frag, pb), o[d]);
                }
                if constexpr (MSUM) osum = Mfma32<T>::run(__builtin_bit_cast(frag, ones8), __builtin_bit_cast(frag, pb), osum);
            }
    };

    const std::true_type TAILY;
    const std::false_type FULL;
    const bool ragged = (p.nk % KT) != 0 && nkt == nkt_all;      // only the last tile (of the last split) can be partial
    auto load_tile = [&](int j) { if (j + 1 == nkt && ragged) stage_load(j, TAILY); else stage_load(j, FULL); };
    auto score_tile = [&](int j, float (&sv)[2][16]) { if (j + 1 == nkt && ragged) scores(j, j % 3, sv, TAILY); else scores(j, j % 3, sv, FULL); };
    load_tile(jb);
    stage_store(0);                                    // (jb is even: buffer jb & 1; the pipelined form is never split, jb = 0)
    if constexpr (!PIPE) {
        __syncthreads();
        float sv[2][16];
        int j = jb;
        // steady state, two tiles per trip: tile j + 1 exists and is full -> no tail variants, no branches, and the LDS buffer of a tile
        // (j & 1) is a compile-time offset.  The generic loop below carried both variants of stage_load / scores behind scalar
        // compares and branches plus per-tile buffer selects: ~45 of its ~225 instructions per key tile, in a kernel that is bound by
        // each wave's in-order issue (section 3.4 of DESIGN.md).  Same arithmetic in the same order: bit-identical results.
        {
            const int n_steady = nkt - 1 - (ragged ? 1 : 0);
            auto body = [&](int jj, auto buf_tag) __attribute__((always_inline)) {
                constexpr int B = decltype(buf_tag)::value;
                stage_load(jj + 1, FULL);
                scores(jj, B, sv, FULL);
                softmax_pv(B, sv);
                stage_store(B ^ 1);
                __syncthreads();
            };
            if (p.steady2)
            for (; j + 2 <= n_steady; j += 2) {
                body(j, std::integral_constant<int, 0>());
                body(j + 1, std::integral_constant<int, 1>());
            }
        }
        for (; j < nkt; ++j) {
            if (j + 1 < nkt) load_tile(j + 1);          // global -> registers while this tile is processed
            if (j + 1 == nkt && ragged) scores(j, j & 1, sv, TAILY); else scores(j, j & 1, sv, FULL);
            softmax_pv(j & 1, sv);
            if (j + 1 < nkt) stage_store((j + 1) & 1);  // held tile j-1: every wave passed the last barrier after using it
            __syncthreads();
        }
    } else {
    if (nkt > 1) { load_tile(1); stage_store(1); }
    __syncthreads();
    float s_cur[2][16], s_next[2][16];
    score_tile(0, s_cur);
    auto rotate = [&]() {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_cur[hh][r] = s_next[hh][r];
    };
    int j = 0;
    // steady state: tiles j+1 and j+2 exist and are full -> a branch-free body.  Unrolled six times: the
    // LDS buffer of a tile (j % 3) becomes a compile-time offset and the two score arrays swap roles
    // instead of being copied (290 -> 197 instructions per tile).  Measured: +1..5 % only -- with two
    // waves per SIMD the loop is bound by the latency of its dependent chain (LDS -> 4 chained MFMAs ->
    // max -> lane exchange -> exp -> convert -> 4 chained MFMAs), not by issue slots; pinning the score
    // MFMAs between groups of VALU instructions (sched_group_barrier) lost 15-25 % on this body.
    const int n_steady = nkt - 2 - (ragged ? 1 : 0);
    auto body = [&](int jj, auto buf_tag, float (&sc)[2][16], float (&sn)[2][16]) {
        constexpr int B = decltype(buf_tag)::value;
        stage_load(jj + 2, FULL);
        scores(jj + 1, (B + 1) % 3, sn, FULL);
        softmax_pv(B, sc);
        stage_store((B + 2) % 3);
        __syncthreads();
    };
    const std::integral_constant<int, 0> B0;
    const std::integral_constant<int, 1> B1;
    const std::integral_constant<int, 2> B2;
    if constexpr (!BIAS)                               // (the EPA instantiations spill with six copies of the body)
    for (; j + 6 <= n_steady; j += 6) {
        body(j, B0, s_cur, s_next);
        body(j + 1, B1, s_next, s_cur);
        body(j + 2, B2, s_cur, s_next);
        body(j + 3, B0, s_next, s_cur);
        body(j + 4, B1, s_cur, s_next);
        body(j + 5, B2, s_next, s_cur);
    }
    for (; j < n_steady; ++j) {
        stage_load(j + 2, FULL);
        scores(j + 1, (j + 1) % 3, s_next, FULL);
        softmax_pv(j % 3, s_cur);
        stage_store((j + 2) % 3);
        __syncthreads();
        rotate();
    }
    for (; j < nkt; ++j) {
        if (j + 2 < nkt) load_tile(j + 2);             // global -> registers, written to LDS at the end of the step
        if (j + 1 < nkt) score_tile(j + 1, s_next);
        softmax_pv(j % 3, s_cur);
        if (j + 2 < nkt) stage_store((j + 2) % 3);     // buffer (j+2)%3 held tile j-1: every wave passed the last barrier after using it
        __syncthreads();
        rotate();
    }
    }

    if constexpr (MSUM) l_run = osum[0];                // (every row of osum is the row sum)
    if (q0 + ql < p.nq) {
        const float inv = 1.0f / l_run;
        unsigned short* op = p.out + b * p.o_bs + static_cast<long>(q0 + ql) * p.o_ld + h * D;
        float* lsep = p.lse;
        if (p.split_tiles > 0) { op += blockIdx.y * p.split_o; lsep += blockIdx.y * p.split_lse; }
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u16x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(o[d][4 * g + e] * inv);
                *reinterpret_cast<u16x4*>(op + d * 32 + 8 * g + 4 * hi) = w;
            }
        // (m_run is a reference point in raw-score units, not necessarily the maximum: deferred rescale)
        if (p.lse && hi == 0) lsep[(b * p.H + h) * p.nq + q0 + ql] = m_run * c2 + __log2f(l_run);
    }
}

// Combine of a split key range (AttnParams.split_tiles): out[b][q][c] = sum_s w_s part[s][b][q][c] / sum_s w_s, w_s = exp2(lse[s][b][h][q] - max_s lse),
// in split order (fixed: no atomics).  part [S][B][nq][C] 16-bit (normalised partial outputs), lse [S][B][H][nq] fp32 (log2 domain).  One thread = 8 channels.
template <typename T>
__global__ __launch_bounds__(256) void k_attention_merge(const unsigned short* __restrict__ part, const float* __restrict__ lse, int S, int B, int H, int D, int nq,
                                                          unsigned short* __restrict__ out, int o_ld, long o_bs) {
    const int C8 = H * D / 8;
    const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x, total = static_cast<long>(B) * nq * C8;
    if (i >= total) return;
    const int c8 = static_cast<int>(i % C8);
    const long bq = i / C8;
    const int q = static_cast<int>(bq % nq), b = static_cast<int>(bq / nq), h = c8 * 8 / D;
    const long so = static_cast<long>(B) * nq * H * D, sl = static_cast<long>(B) * H * nq;
    const float* lp = lse + (static_cast<long>(b) * H + h) * nq + q;
    float m = -INFINITY;
    for (int s = 0; s < S; ++s) m = fmaxf(m, lp[s * sl]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wsum = 0.f;
    const unsigned short* pp = part + bq * (H * D) + c8 * 8;
    for (int s = 0; s < S; ++s) {
        const float w = __builtin_amdgcn_exp2f(lp[s * sl] - m);
        const u16x8 v = *reinterpret_cast<const u16x8*>(pp + s * so);
        wsum += w;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += w * to_f32<T>(v[e]);
    }
    const float inv = 1.0f / wsum;
    u16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = from_f32<T>(acc[e] * inv);
    *reinterpret_cast<u16x8*>(out + b * o_bs + static_cast<long>(q) * o_ld + c8 * 8) = r;
}


// ---- 8-wave ping-pong variant (round 5; D = 64, no bias: the UNet self-attentions) -------------------------------------
// MI355X_MICROARCH.md, "Two waves per SIMD": a 512-thread workgroup places waves w and w + 4 on one SIMD.  The key walk is cut
// into PHASES separated by s_barrier; in every phase one wave of a SIMD runs its MATRIX segment (O^T += V_j^T P_j^T, then
// S_{j+1}^T = K_{j+1} Q^T: 16 MFMAs, fragments already in registers) while its partner runs its VECTOR segment (online softmax
// of its own S_j, conversion to the 16-bit P operand, LDS -> register prefetch of the fragments of its next matrix segment) --
// the two pipes of the SIMD work side by side by construction instead of by the accident of three drifting waves
// (k_attention_lds: 26-31 % matrix-pipe busy, profiles/r4k_attn_ablate.txt).  Group A = waves 0..3, group B = waves 4..7, B runs
// half a period behind A:
//     phase 2j      A: vector_j            B: matrix_{j-1}
//     phase 2j + 1  A: matrix_j            B: vector_j
// K [64 keys][64] and V^T [64][64 keys] tiles reach LDS by LDS-DMA (buffer_load ... lds, no registers, no ds_write): every wave
// moves one 1-KB piece of each, into rings of NB = L + 1 tiles.  "Set j" = (K_{j+1+L}, V_{j+L}) is requested by ALL waves at the
// start of phase 2j -- into the slots last read (register prefetch) in phases 2j - 2 / 2j - 1 -- and must have landed when
// phase 2(j+L) begins: 2 L phases.  At the end of phase 2j + 1 a wave waits for set j + 1 - L with a COUNTED vmcnt(2 (L - 1)): its
// loads retire in order and nothing else is in flight in the loop.  (First version: L = 1, two slots, vmcnt(0) -- correct and 15-35 %
// SLOWER than k_attention_lds: one period of look-ahead is less than the L2 / fabric latency under load, every tile waited for its
// DMA; profiles/r5c_attn_pp_first.txt.)  Rows are 128 B with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7 (applied to the SOURCE offset, the
// DMA writes lane-linear): conflict-free ds_read_b128 for both tiles.  To make a V^T fragment ONE 16-byte read, tile row i of K
// holds key pi(i) = i with bits 2 and 3 swapped: the score registers of lane (q, hi) then cover, per 16-key slab, the 8 CONSECUTIVE
// keys 16 s + 8 hi .. + 7 -- exactly one chunk of a V^T row (the MFMA k-slot order is free as long as P and V^T agree).
// Needs nk % 8 == 0 (whole chunks are either keys or zero fill); everything else stays on k_attention_lds.
template <typename T>
__global__ __launch_bounds__(512, 1) void k_attention_pp(const AttnParams p) {
    constexpr int D = 64, KS = 4, DB = 2, KT = 64, TILE = KT * D;
    constexpr int L = 3, NB = L + 2;                                            // look-ahead in tiles, ring slots per operand
    typedef typename Mfma32<T>::frag frag;
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * NB * TILE];  // K tiles [NB], V^T tiles [NB]: 80 KB
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ql = lane & 31, hi = lane >> 5;
    // 1-D grid, XCD-aware: block id l runs on XCD l % 8 (speed only).  When the (batch, head) pairs divide by 8, pair g lives on XCD
    // g % 8 with its query blocks back to back there -- a head's K / V^T (1 MB at 4096 keys) is fetched into ONE L2, not eight.
    const int nqb = (p.nq + 255) / 256, BH = static_cast<int>(gridDim.x) / nqb;
    int qb, bh;
    if ((BH & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bh = xcd + 8 * (idx / nqb);
        qb = idx - (idx / nqb) * nqb;
    } else {
        bh = blockIdx.x / nqb;
        qb = blockIdx.x - bh * nqb;
    }
    const int q0 = (qb * 8 + wave) * 32;
    const int h = bh % p.H;
    const long b = bh / p.H;
    const unsigned short* qp = p.q + b * p.q_bs + h * D;
    const unsigned short* kp = p.k + b * p.k_bs + h * D;
    const unsigned short* vp = p.vt + b * p.vt_bs + static_cast<long>(h) * D * p.vt_ld;
    const int qrow = min(q0 + ql, p.nq - 1);

    frag qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        qf[s] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(qp + static_cast<long>(qrow) * p.q_ld + 16 * s + 8 * hi));

    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    // The softmax denominator comes out of the MATRIX pipe: a third accumulator block osum^T += 1 P^T (A operand = ones) -- every row of
    // it is sum_k P[q][k] over both lane halves, so the vector segment loses its 16 packed adds and one lane exchange per key tile (it is
    // the VALU-issue-bound side: profiles/r5i_attn_pp_ablate.txt), and the sum is taken over the SAME 16-bit P the numerator uses.
    f32x16 osum;
#pragma unroll
    for (int r = 0; r < 16; ++r) osum[r] = 0.f;
    u16x8 ones8;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones8[e] = from_f32<T>(1.0f);
    asm volatile("" : "+v"(ones8));                                             // (kept in registers, not rematerialised per MFMA)
    const frag ones = __builtin_bit_cast(frag, ones8);
    float m_run = -INFINITY;
    const float c2 = p.scale_log2e;
    const int nkt = (p.nk + KT - 1) / KT;
    const bool ragged = (p.nk % KT) != 0;

    // ---- LDS-DMA staging: wave w moves rows 8 w .. 8 w + 7 of a tile (1 KB), lane l -> row 8 w + l / 8, physical chunk l % 8
    constexpr unsigned OOB = 0x80000000u;
    auto uniform_ptr = [](const unsigned short* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
        const unsigned hi32 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
        return reinterpret_cast<unsigned short*>(static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi32) << 32));
    };
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(kp), 0, __builtin_amdgcn_readfirstlane(((p.nk - 1) * p.k_ld + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(vp), 0, __builtin_amdgcn_readfirstlane(D * p.vt_ld * 2), 0x00020000);
    const int srow = wave * 8 + (lane >> 3);
    const int lchunk = (lane & 7) ^ ((srow >> 1) & 7);                           // logical chunk that belongs in this lane's physical slot
    const int krow = (srow & ~12) | ((srow & 4) << 1) | ((srow & 8) >> 1);       // pi: tile row srow holds key krow of the tile
    const unsigned kvoff0 = static_cast<unsigned>(krow * p.k_ld + lchunk * 8) * 2u;
    const unsigned vvoff0 = static_cast<unsigned>(srow * p.vt_ld + lchunk * 8) * 2u;
    const unsigned kstep = static_cast<unsigned>(KT * p.k_ld) * 2u;
    auto lds_dma = [&](const __amdgpu_buffer_rsrc_t& r, unsigned short* dst, unsigned voff) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
    };
    auto dma_k = [&](int j) __attribute__((always_inline)) {                     // K tile j -> K slot j % NB
        if (j >= nkt) return;
        unsigned voff = kvoff0 + static_cast<unsigned>(j) * kstep;
        if (ragged && j == nkt - 1 && j * KT + krow >= p.nk) voff = OOB;
        lds_dma(rs_k, smem + (j % NB) * TILE + wave * 512, voff);
    };
    auto dma_v = [&](int j) __attribute__((always_inline)) {                     // V^T tile j -> V slot j % NB
        if (j >= nkt) return;
        unsigned voff = vvoff0 + static_cast<unsigned>(j) * (KT * 2u);
        if (ragged && j == nkt - 1 && j * KT + lchunk * 8 >= p.nk) voff = OOB;
        lds_dma(rs_v, smem + (NB + j % NB) * TILE + wave * 512, voff);
    };
    // counted wait: the L youngest sets (2 pieces each) may stay in flight while they are all real, i.e. while `youngest` has its K tile
    auto wait_sets = [&](int youngest) __attribute__((always_inline)) {
        static_assert(L == 3, "the counted wait is written for L = 3");
        if (youngest + 1 + L < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto frag_off = [&](int row, int chunk) __attribute__((always_inline)) { return row * D + ((chunk ^ ((row >> 1) & 7)) << 3); };

    frag kf[2][KS], vf[DB][2][2];
    auto load_kfrags = [&](int j) __attribute__((always_inline)) {
        const unsigned short* Ks = smem + (j % NB) * TILE;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#if defined(PF_ATTN_PP_ABL) && (PF_ATTN_PP_ABL & 2)
                { u16x8 z = {1, 2, 3, 4, 5, 6, 7, static_cast<unsigned short>(lane + ks)}; asm volatile("" : "+v"(z)); kf[hh][ks] = __builtin_bit_cast(frag, z); }
#else
                kf[hh][ks] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(Ks + frag_off(hh * 32 + ql, 2 * ks + hi)));
#endif
    };
    auto load_vfrags = [&](int j) __attribute__((always_inline)) {
        const unsigned short* Vs = smem + (NB + j % NB) * TILE;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#if defined(PF_ATTN_PP_ABL) && (PF_ATTN_PP_ABL & 2)
                    { u16x8 z = {1, 2, 3, 4, 5, 6, 7, static_cast<unsigned short>(lane + d)}; asm volatile("" : "+v"(z)); vf[d][hh][s2] = __builtin_bit_cast(frag, z); }
#else
                    vf[d][hh][s2] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(Vs + frag_off(d * 32 + ql, 4 * hh + 2 * s2 + hi)));
#endif
    };
    float sv[2][16];
    u16x8 pb[2][2];
    auto mm_qk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s = Mfma32<T>::run(kf[hh][ks], qf[ks], s);
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[hh][r] = s[r];
        }
    };
    // the 32-key half hh of the tile: 4 + 2 MFMAs; `between` runs behind the first three (a DMA piece: its issue then overlaps the matrix
    // pipe's work instead of standing in front of it -- at the head of the segment the two pieces cost 270 clocks)
    auto mm_pv = [&](auto hh_tag, auto between) __attribute__((always_inline)) {
        constexpr int hh = decltype(hh_tag)::value;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int d = 0; d < DB; ++d)
                o[d] = Mfma32<T>::run(vf[d][hh][s2], __builtin_bit_cast(frag, pb[hh][s2]), o[d]);
            osum = Mfma32<T>::run(ones, __builtin_bit_cast(frag, pb[hh][s2]), osum);
            if (s2 == 0) {
                __builtin_amdgcn_sched_barrier(0);
                between();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // score register r of half hh <-> key  k0 + 32 hh + 16 (g >> 1) + 8 hi + 4 (g & 1) + e,  g = r >> 2, e = r & 3  (pi above)
    auto mask_tail = [&](int j) __attribute__((always_inline)) {
        if (!(ragged && j == nkt - 1)) return;
        asm volatile("; ragged last tile" ::: "memory");                         // (a real branch: if-converted, the 32 selects ran on every tile)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int g = r >> 2, e = r & 3;
                if (j * KT + 32 * hh + 16 * (g >> 1) + 8 * hi + 4 * (g & 1) + e >= p.nk) sv[hh][r] = -INFINITY;
            }
    };
    // The exchange between lanes l and l + 32 (the two halves of a query's scores): v_permlane32_swap (gfx950) on two copies of the
    // value -- a = [x_lo, x_lo], b = [x_hi, x_hi] afterwards -- instead of ds_bpermute: the timing build showed the vector segment at
    // 1400-1900 clocks per key tile against 630 for the 16 MFMAs (profiles/r5g_attn_pp_timing.txt); it is VALU-ISSUE bound (33 v_exp at
    // quarter rate, ~100 other vector instructions at 4 clocks each) and every LDS round trip (bpermute -> s_waitcnt lgkmcnt(0)) stalls
    // the in-order wave on top of that.  (The builtin __builtin_amdgcn_permlane32_swap of this hipcc returns the SAME register for
    // both results -- inline assembly.)
    auto xchg32 = [&](float x, float& lo, float& hi_) __attribute__((always_inline)) {
        float a_ = x, b_ = x;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
        lo = a_;
        hi_ = b_;
    };
    auto softmax = [&]() __attribute__((always_inline)) {                        // sv -> pb (16-bit P), running max / sum, deferred rescale of O
        float mt = sv[0][0];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#if defined(PF_ATTN_PP_ABL) && (PF_ATTN_PP_ABL & 4)
                if (r & 7) continue;                                             // timing-only: 4 of the 32 maxima
#endif
                mt = fmaxf(mt, sv[hh][r]);
            }
        {
            float x0, x1;
            xchg32(mt, x0, x1);
            mt = fmaxf(x0, x1);
        }
        {
            constexpr float DEFER_LOG2 = 8.0f;                                    // (see k_attention_lds)
            const float m_new = fmaxf(m_run, mt);
            const bool grow = (m_new - m_run) * c2 > DEFER_LOG2;
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
                m_run = m_new;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) osum[r] *= alpha;
            }
        }
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        const f32x2 c22 = {c2, c2}, mc2 = {m_run * c2, m_run * c2};
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 x = f32x2{sv[hh][r], sv[hh][r + 1]} * c22 - mc2;
#if defined(PF_ATTN_PP_ABL) && (PF_ATTN_PP_ABL & 1)
                const f32x2 e = x * x;                                           // timing-only: what the 32 v_exp_f32 per key tile cost
#else
                const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
#endif
                sv[hh][r] = e[0];
                sv[hh][r + 1] = e[1];
            }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#if defined(PF_ATTN_PP_ABL) && (PF_ATTN_PP_ABL & 8)
                { pb[hh][s2] = __builtin_bit_cast(u16x8, f32x4{sv[hh][8 * s2], sv[hh][8 * s2 + 1], sv[hh][8 * s2 + 2], sv[hh][8 * s2 + 3]}); }   // timing-only: no conversion
#else
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[hh][s2][e] = from_f32<T>(sv[hh][8 * s2 + e]);
#endif
        // P is FINISHED here, in the vector segment: without this pin the compiler sinks the 32 v_exp / 16 v_cvt_pk (pure register
        // work whose only users are the MFMAs behind the barrier) into the matrix segment, next to the MFMAs of the same wave --
        // the two segments of a wave would swap their contents and the partner waves' pipes collide instead of interleaving
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) asm volatile("" : "+v"(pb[hh][s2]));
    };
#ifdef PF_ATTN_PP_TIMING      /* debug build (make attn_pp_timing): per-wave clock totals of the vector segments, the matrix segments and the
                               * barrier waits, written over the lse output -- tools/attn_bench.py --pp-timing prints the per-tile averages */
    unsigned long long tm_vec = 0, tm_mat = 0, tm_bar = 0, tm_last = __builtin_amdgcn_s_memtime();
#define PF_PP_MARK(acc) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tm_last; tm_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PF_PP_MARK(acc) do { } while (0)
#endif
    auto barrier = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- which phase group?  The two waves of a SIMD must be in DIFFERENT groups (one runs its matrix segment while the other runs
    // its vector segment); how the 8 waves of a workgroup land on the 4 SIMDs is the dispatcher's business, so the default asks the
    // hardware: every wave publishes its SIMD id (HW_REG_HW_ID bits 5:4), and of the waves that share a SIMD the lowest-numbered
    // is group A, the next group B, alternating.  pp_role 0 / 1 / 2 are the fixed guesses wave >> 2, wave & 1, (wave >> 1) & 1 (A/B).
    int group_b;
    if (p.pp_role == 0) group_b = wave >> 2;
    else if (p.pp_role == 1) group_b = wave & 1;
    else if (p.pp_role == 2) group_b = (wave >> 1) & 1;
    else {
        int* simd_of = reinterpret_cast<int*>(smem);                             // (the ring is not in use yet)
        const int my_simd = (__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4)) & 3;   // HW_REG_HW_ID, offset 4, 2 bits: SIMD_ID
        if (lane == 0) simd_of[wave] = my_simd;
        __syncthreads();
        int rank_on_simd = 0;
        for (int w = 0; w < 8; ++w) rank_on_simd += (w < wave && simd_of[w] == my_simd) ? 1 : 0;
        group_b = __builtin_amdgcn_readfirstlane(rank_on_simd & 1);
        __syncthreads();                                                         // (before the DMA overwrites the table)
    }

    // ---- prologue: K_0, then sets -L .. 0 = (K_1, V_0) .. (K_{L+1}, V_L) in the loop's request order; only K_0 is awaited here (S_0
    // needs nothing else) -- the loop's counted waits take the rest as they would any set (a 1024-key head is 16 tiles: waiting for
    // nine tiles up front cost the 32^2 level a third of its time, profiles/r5k_attn_pp_prio.txt)
    dma_k(0);
#pragma unroll
    for (int i = 0; i <= L; ++i) { dma_k(i + 1); dma_v(i); }
    if (L + 1 < nkt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // (2 (L + 1) younger pieces, all real)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_assert(L == 3, "prologue wait count");
    barrier();
    load_kfrags(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mm_qk();
    // (no second barrier: the first overwrite of K slot 0 is set 1's K_{L+2} -> slot (L + 2) % NB = 0, requested in the first matrix
    // segment, which every wave enters through a barrier of the loop below)

    // matrix_j of a wave: request set j + 1 = (K_{j+2+L}, V_{j+1+L}) into the slots of K_j / V_{j-1} (last read in matrix_{j-1}: by A in
    // phase 2j - 1, by B in phase 2j), read the fragments of V_j and K_{j+1} (= set j - L), 12 + 8 MFMAs.
    auto matrix_segment = [&](int j) __attribute__((always_inline)) {
        if (p.pp_prio == 2) __builtin_amdgcn_s_setprio(2);                       // MFMA issue wins the SIMD's arbitration against the partner's VALU stream
        load_vfrags(j);
        mm_pv(std::integral_constant<int, 0>(), [&]() __attribute__((always_inline)) { dma_k(j + 2 + L); });
        __builtin_amdgcn_sched_barrier(0);                                       // (the K fragments are requested behind the first half of
        if (j + 1 < nkt) load_kfrags(j + 1);                                     // the PV product: its V^T registers are free by then, and the
        __builtin_amdgcn_sched_barrier(0);                                       // second half covers the LDS latency)
        mm_pv(std::integral_constant<int, 1>(), [&]() __attribute__((always_inline)) { dma_v(j + 1 + L); });
        if (j + 1 < nkt) mm_qk();
        if (p.pp_prio == 2) __builtin_amdgcn_s_setprio(0);
    };
    if (p.pp_prio == 1 && group_b) __builtin_amdgcn_s_setprio(1);               // (MI355X_MICROARCH.md, two waves per SIMD, item 4: the younger half loses every arbitration)
    if (!group_b) {
        // group A: vector_j in phase 2 j, matrix_j in phase 2 j + 1
        for (int j = 0; j < nkt; ++j) {
            mask_tail(j);
            softmax();
            wait_sets(j);                                                        // set j - L (V_j, K_{j+1}) landed (mine): I have requested up to set j
            PF_PP_MARK(tm_vec);
            barrier();
            PF_PP_MARK(tm_bar);
            matrix_segment(j);
#ifdef PF_ATTN_PP_TIMING
            asm volatile("s_nop 0" :: "v"(sv[0][0]), "v"(sv[1][15]), "v"(o[0][0]), "v"(o[1][15]));   // (the MFMA results: the stamp waits for them)
#endif
            PF_PP_MARK(tm_mat);
            barrier();
            PF_PP_MARK(tm_bar);
        }
    } else {
        // group B: half a period behind -- vector_j in phase 2 j + 1, matrix_j in phase 2 j + 2
        wait_sets(0);                                                            // my pieces of set -L (K_1, V_0): group A reads them in phase 1 (found by the
        barrier();                                                               // bit-reproducibility-under-contention test on k_attention_pp2, same prologue)
        for (int j = 0; j < nkt; ++j) {
            mask_tail(j);
            softmax();
            PF_PP_MARK(tm_vec);
            barrier();
            PF_PP_MARK(tm_bar);
            matrix_segment(j);
            wait_sets(j + 1);                                                    // set j + 1 - L (V_{j+1}, K_{j+2}: A reads them next phase) landed (mine)
#ifdef PF_ATTN_PP_TIMING
            asm volatile("s_nop 0" :: "v"(sv[0][0]), "v"(sv[1][15]), "v"(o[0][0]), "v"(o[1][15]));
#endif
            PF_PP_MARK(tm_mat);
            if (j + 1 < nkt) barrier();                                          // (A executes 2 nkt barriers in its loop, B 1 + 2 nkt - 1)
            PF_PP_MARK(tm_bar);
        }
    }

    if (q0 + ql < p.nq) {
        const float l_run = osum[0];                                              // (every row of osum is the row sum)
        const float inv = 1.0f / l_run;
        unsigned short* op = p.out + b * p.o_bs + static_cast<long>(q0 + ql) * p.o_ld + h * D;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u16x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(o[d][4 * g + e] * inv);
                *reinterpret_cast<u16x4*>(op + d * 32 + 8 * g + 4 * hi) = w;
            }
        if (p.lse && hi == 0) p.lse[(b * p.H + h) * p.nq + q0 + ql] = m_run * c2 + __log2f(l_run);
    }
#ifdef PF_ATTN_PP_TIMING
    if (p.lse && lane == 0 && q0 + 8 <= p.nq) {
        float* dst = p.lse + (b * p.H + h) * p.nq + q0;
        dst[0] = static_cast<float>(tm_vec) / nkt; dst[1] = static_cast<float>(tm_mat) / nkt; dst[2] = static_cast<float>(tm_bar) / nkt;
        dst[3] = static_cast<float>(group_b); dst[4] = static_cast<float>((__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4)) & 3);
        dst[5] = static_cast<float>(wave);
    }
#endif
}

#ifndef PF_PP2_MSUM
#define PF_PP2_MSUM 0          /* 1: row sums as ones-operand MFMAs (20 more registers: the kernel then spills) */
#endif
// ---- ping-pong, TWO key tiles per phase (round 5, second structure) -------------------------------------------------------------------
// tools/ubench/valu_rate.hip settled what the hardware can do (profiles/r5t_valu_rate.txt): on one SIMD a wave issuing 16 MFMAs (520 clocks) and
// a partner issuing the softmax's vector mix (606 clocks alone) run SIDE BY SIDE at 527 / 711 -- the pipes do overlap.  k_attention_pp's phases
// carried ~450 clocks of fixed cost each (barrier skew, DMA issue, the first exposed LDS read) against ~640 of work; here a phase covers a PAIR
// of 64-key tiles (40 MFMAs / ~1150 clocks of vector issue), so the fixed costs are paid once per 128 keys.  Same DMA rings (tile slots, six
// per operand = three pairs), same K-row permutation, same row sums on the matrix pipe; fragments go through four 16-register buffers in a
// fixed software pipeline (the reads of the next half tile are issued in front of the MFMAs of the current one).  nk % 128 == 0 (every
// UNet self-attention), no bias, no lse.
template <typename T>
__global__ __launch_bounds__(512, 1) void k_attention_pp2(const AttnParams p) {
    constexpr bool MSUM = PF_PP2_MSUM;
    constexpr int D = 64, KS = 4, DB = 2, KT = 64, TILE = KT * D;
    constexpr int LP = 1, NB = 2 * (LP + 2);                                    // look-ahead in PAIRS; ring slots per operand in tiles
    typedef typename Mfma32<T>::frag frag;
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * NB * TILE + 8 * KS * 64 * 8];  // 96 KB of rings + 32 KB: the waves' Q fragments
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ql = lane & 31, hi = lane >> 5;
    const int nqb = (p.nq + 255) / 256, BH = static_cast<int>(gridDim.x) / nqb;
    int qb, bh;
    if ((BH & 7) == 0) {                                                         // (batch, head) pairs pinned to XCDs, as in k_attention_lds
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bh = xcd + 8 * (idx / nqb);
        qb = idx - (idx / nqb) * nqb;
    } else {
        bh = blockIdx.x / nqb;
        qb = blockIdx.x - bh * nqb;
    }
    const int q0 = (qb * 8 + wave) * 32;
    const int h = bh % p.H;
    const long b = bh / p.H;
    const unsigned short* qp = p.q + b * p.q_bs + h * D;
    const unsigned short* kp = p.k + b * p.k_bs + h * D;
    const unsigned short* vp = p.vt + b * p.vt_bs + static_cast<long>(h) * D * p.vt_ld;
    const int qrow = min(q0 + ql, p.nq - 1);

    // Q fragments live in LDS in fragment layout (a wave re-reads its four 16-byte fragments per key tile): 16 registers the spilling
    // first version kept in scratch -- and every scratch reload is a VMEM load whose compiler-inserted vmcnt(0) also waits for the
    // LDS-DMA pieces just requested (1881 us against 1228 for k_attention_lds at 64^2)
    unsigned short* const qs = smem + 2 * NB * TILE + wave * (KS * 64 * 8) + lane * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s)
        *reinterpret_cast<u16x8*>(qs + s * 64 * 8) = *reinterpret_cast<const u16x8*>(qp + static_cast<long>(qrow) * p.q_ld + 16 * s + 8 * hi);
    f32x16 o[DB], osum;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; osum[r] = 0.f; }
    u16x8 ones8 = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (MSUM) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ones8[e] = from_f32<T>(1.0f);
        asm volatile("" : "+v"(ones8));
    }
    const frag ones = __builtin_bit_cast(frag, ones8);
    float m_run = -INFINITY, l_run = 0.f;
    const float c2 = p.scale_log2e;
    const int nkt = p.nk / KT, npair = nkt / 2;                                  // (launcher: nk % 128 == 0)

    auto uniform_ptr = [](const unsigned short* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
        const unsigned hi32 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
        return reinterpret_cast<unsigned short*>(static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi32) << 32));
    };
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(kp), 0, __builtin_amdgcn_readfirstlane(((p.nk - 1) * p.k_ld + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(vp), 0, __builtin_amdgcn_readfirstlane(D * p.vt_ld * 2), 0x00020000);
    const int srow = wave * 8 + (lane >> 3);
    const int lchunk = (lane & 7) ^ ((srow >> 1) & 7);
    const int krow = (srow & ~12) | ((srow & 4) << 1) | ((srow & 8) >> 1);       // pi (see k_attention_pp)
    const unsigned kvoff0 = static_cast<unsigned>(krow * p.k_ld + lchunk * 8) * 2u;
    const unsigned vvoff0 = static_cast<unsigned>(srow * p.vt_ld + lchunk * 8) * 2u;
    const unsigned kstep = static_cast<unsigned>(KT * p.k_ld) * 2u;
    auto lds_dma = [&](const __amdgpu_buffer_rsrc_t& r, unsigned short* dst, unsigned voff) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
    };
    auto dma_k = [&](int j) __attribute__((always_inline)) {                     // K tile j -> K slot j % NB
        if (j < nkt) lds_dma(rs_k, smem + (j % NB) * TILE + wave * 512, kvoff0 + static_cast<unsigned>(j) * kstep);
    };
    auto dma_v = [&](int j) __attribute__((always_inline)) {                     // V^T tile j -> V slot j % NB
        if (j < nkt) lds_dma(rs_v, smem + (NB + j % NB) * TILE + wave * 512, vvoff0 + static_cast<unsigned>(j) * (KT * 2u));
    };
    // set s = (K pair s + 1 + LP, V pair s + LP), 4 pieces per wave; the LP youngest sets may stay in flight while `youngest` is whole
    auto wait_sets = [&](int youngest) __attribute__((always_inline)) {
        static_assert(LP == 1, "the counted wait is written for LP = 1");
        if (youngest + 1 + LP < npair) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto frag_off = [&](int row, int chunk) __attribute__((always_inline)) { return row * D + ((chunk ^ ((row >> 1) & 7)) << 3); };
    // fragments of one 32-key half of a tile: V^T (d 0 | d 1) x (slab 0 | slab 1), or K slabs 0..3
    // Fragment addresses from ONE per-lane register: row (32 r + ql), logical chunk (even c + hi) sits at element
    // (32 r + ql) 64 + (((c + hi) ^ swz) << 3) = 2048 r + (lane_base ^ (c << 3)),  lane_base = 64 ql + ((swz ^ hi) << 3)  (c even, swz = (ql >> 1) & 7):
    // an XOR with an immediate per read (volatile: left to itself the compiler hoists all eight XORs out of the loop into eight registers
    // -- registers this kernel does not have: with them it spilled a lane constant and reloaded it through VMEM inside the loop).
    const int lane_base = (64 * ql + ((((ql >> 1) & 7) ^ hi) << 3)) * 2;        // bytes
    auto frag_at = [&](const unsigned short* tile, int r32, int c_even) __attribute__((always_inline)) {
        int off;
        asm volatile("v_xor_b32 %0, %2, %1" : "=v"(off) : "v"(lane_base), "s"(c_even << 4));
        return __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(reinterpret_cast<const char*>(tile) + 4096 * r32 + off));
    };
    auto ld_v = [&](int j, int hh, frag (&f)[4]) __attribute__((always_inline)) {
        const unsigned short* Vs = smem + (NB + j % NB) * TILE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int d = 0; d < DB; ++d) f[2 * s2 + d] = frag_at(Vs, d, 4 * hh + 2 * s2);
    };
    auto ld_k = [&](int j, int hh, frag (&f)[4]) __attribute__((always_inline)) {
        const unsigned short* Ks = smem + (j % NB) * TILE;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) f[ks] = frag_at(Ks, hh, 2 * ks);
    };
    float sv[2][2][16];                                                          // [tile of the pair][32-key half][score register]
    u16x8 pb[2][2][2];                                                           // 16-bit P: [tile][half][16-key slab]
    auto pv = [&](frag (&f)[4], int u, int hh) __attribute__((always_inline)) {  // 4 + 2 MFMAs
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            o[0] = Mfma32<T>::run(f[2 * s2], __builtin_bit_cast(frag, pb[u][hh][s2]), o[0]);
            o[1] = Mfma32<T>::run(f[2 * s2 + 1], __builtin_bit_cast(frag, pb[u][hh][s2]), o[1]);
            if constexpr (MSUM) osum = Mfma32<T>::run(ones, __builtin_bit_cast(frag, pb[u][hh][s2]), osum);
        }
    };
    auto qk = [&](frag (&f0)[4], frag (&f1)[4], int u) __attribute__((always_inline)) {   // both halves of a tile: two interleaved chains of 4
        f32x16 s0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s1 = s0;
        frag qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(qs + ks * 64 * 8));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            s0 = Mfma32<T>::run(f0[ks], qf[ks], s0);
            s1 = Mfma32<T>::run(f1[ks], qf[ks], s1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { sv[u][0][r] = s0[r]; sv[u][1][r] = s1[r]; }
    };
    auto xchg32 = [&](float x, float& lo, float& hi_) __attribute__((always_inline)) {
        float a_ = x, b_ = x;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
        lo = a_;
        hi_ = b_;
    };
    auto softmax = [&]() __attribute__((always_inline)) {                        // the pair's 128 keys at once: one maximum, one rescale decision
        float mt = sv[0][0][0];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sv[u][hh][r]);
        {
            float x0, x1;
            xchg32(mt, x0, x1);
            mt = fmaxf(x0, x1);
        }
        {
            constexpr float DEFER_LOG2 = 8.0f;                                    // (see k_attention_lds)
            const float m_new = fmaxf(m_run, mt);
            const bool grow = (m_new - m_run) * c2 > DEFER_LOG2;
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                if constexpr (MSUM) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) osum[r] *= alpha;
                } else l_run *= alpha;
            }
        }
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        const f32x2 c22 = {c2, c2}, mc2 = {m_run * c2, m_run * c2};
        f32x2 ls2 = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 x = f32x2{sv[u][hh][r], sv[u][hh][r + 1]} * c22 - mc2;
                    const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                    sv[u][hh][r] = e[0];
                    sv[u][hh][r + 1] = e[1];
                    if constexpr (!MSUM) ls2 += e;
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pb[u][hh][s2][e] = from_f32<T>(sv[u][hh][8 * s2 + e]);
                    asm volatile("" : "+v"(pb[u][hh][s2]));                      // P is finished in the vector segment (see k_attention_pp)
                }
            }
        if constexpr (!MSUM) {
            float x0, x1;
            xchg32(ls2[0] + ls2[1], x0, x1);
            l_run += x0 + x1;
        }
    };
    auto barrier = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
#define PF_SB() __builtin_amdgcn_sched_barrier(0)

    // phase groups from the hardware SIMD id (see k_attention_pp)
    int group_b;
    {
        int* simd_of = reinterpret_cast<int*>(smem);
        const int my_simd = (__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4)) & 3;
        if (lane == 0) simd_of[wave] = my_simd;
        __syncthreads();
        int rank_on_simd = 0;
        for (int w = 0; w < 8; ++w) rank_on_simd += (w < wave && simd_of[w] == my_simd) ? 1 : 0;
        group_b = __builtin_amdgcn_readfirstlane(rank_on_simd & 1);
        __syncthreads();
    }

    frag fa[4], fb[4], fc[4], fd[4];
    // ---- prologue: K pair 0, then sets -LP .. 0 in the loop's request order; only K pair 0 is awaited here
    dma_k(0);
    dma_k(1);
#pragma unroll
    for (int s = -LP; s <= 0; ++s) {
        dma_k(2 * (s + 1 + LP));
        dma_k(2 * (s + 1 + LP) + 1);
        dma_v(2 * (s + LP));
        dma_v(2 * (s + LP) + 1);
    }
    if (1 + LP < npair) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // (4 (LP + 1) younger pieces, all real)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_assert(LP == 1, "prologue wait count");
    barrier();
    ld_k(0, 0, fa); ld_k(0, 1, fb); ld_k(1, 0, fc); ld_k(1, 1, fd);
    qk(fa, fb, 0);
    qk(fc, fd, 1);

    // matrix segment of pair i: tiles a = 2 i, a + 1: O^T += V^T P^T (+ row sums), then S^T of pair i + 1; requests set i + 1 between the MFMAs.
    // Fragment buffers rotate fa -> fb -> ...: the reads of the next half tile are issued in front of the MFMAs of the current one.
    auto matrix_segment = [&](int i) __attribute__((always_inline)) {
        const int a = 2 * i, c = 2 * i + 2;
        const bool more = i + 1 < npair;
        __builtin_amdgcn_s_setprio(2);
        ld_v(a, 0, fa);
        ld_v(a, 1, fb);
        PF_SB();
        pv(fa, 0, 0);
        PF_SB();
        dma_k(2 * (i + 2 + LP));
        ld_v(a + 1, 0, fc);
        PF_SB();
        pv(fb, 0, 1);
        PF_SB();
        dma_k(2 * (i + 2 + LP) + 1);
        ld_v(a + 1, 1, fd);
        PF_SB();
        pv(fc, 1, 0);
        PF_SB();
        dma_v(2 * (i + 1 + LP));
        if (more) { ld_k(c, 0, fa); ld_k(c, 1, fb); }
        PF_SB();
        pv(fd, 1, 1);
        PF_SB();
        dma_v(2 * (i + 1 + LP) + 1);
        if (more) {
            ld_k(c + 1, 0, fc);
            PF_SB();
            qk(fa, fb, 0);
            PF_SB();
            ld_k(c + 1, 1, fd);
            PF_SB();
            qk(fc, fd, 1);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    if (!group_b) {
        for (int i = 0; i < npair; ++i) {                                        // group A: vector_i in phase 2 i, matrix_i in phase 2 i + 1
            softmax();
            wait_sets(i);                                                        // set i - LP (K pair i + 1, V pair i) landed (mine)
            barrier();
            matrix_segment(i);
            barrier();
        }
    } else {
        wait_sets(0);                                                            // my pieces of set -LP (K pair 1, V pair 0): group A reads them in phase 1
        barrier();                                                               // group B: half a period behind
        for (int i = 0; i < npair; ++i) {
            softmax();
            barrier();
            matrix_segment(i);
            wait_sets(i + 1);                                                    // set i + 1 - LP landed (mine): A reads it next phase
            if (i + 1 < npair) barrier();
        }
    }
#undef PF_SB

    if (q0 + ql < p.nq) {
        const float inv = 1.0f / (MSUM ? osum[0] : l_run);                       // (every row of osum is the row sum)
        unsigned short* op = p.out + b * p.o_bs + static_cast<long>(q0 + ql) * p.o_ld + h * D;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u16x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = from_f32<T>(o[d][4 * g + e] * inv);
                *reinterpret_cast<u16x4*>(op + d * 32 + 8 * g + 4 * hi) = w;
            }
    }
}

static bool use_lds_attention() {
    static int v = -1;                     // PF_ATTENTION_IMPL=direct selects the no-LDS kernel (A/B switch)
    if (v < 0) { const char* e = getenv("PF_ATTENTION_IMPL"); v = (e && e[0] == 'd') ? 0 : 1; }
    return v == 1;
}

// A/B switches.  PF_ATTENTION_OCC (D = 64): 3 = one score array, three waves per SIMD (default), 2 = register-pipelined
// scores, two waves per SIMD.  PF_ATTENTION_OCC32 (D = 32): 4 (default since round 6: 122 registers, no scratch; -0.2 ... -0.35 ms per step on three boxes,
// profiles/r6o_ab_epa_occupancy.txt) | 5 = not pipelined at that occupancy, 3 = register-pipelined scores at three waves per SIMD (rounds 3-5).
static int attention_occupancy(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Split of the key range (round 6): a launch of the biased D = 32 kernel (EPA) whose query blocks fill less than ~0.95 of the chip's workgroup slots
// (256 CUs x 4) with a long key walk is cut into S splits of an even number of key tiles, at least MIN_TILES each, aiming at ~1.9 rounds of
// workgroups.  PF_ATTENTION_SPLIT=0 disables it (read per call: the tests run both).
struct AttnSplit { int S, tiles; size_t o_bytes, bytes; };
static AttnSplit attention_split(const pf_attn_desc* d) {
    AttnSplit r{1, 0, 0, 0};
    if (!attention_occupancy("PF_ATTENTION_SPLIT", 1) || d->D != 32 || !d->bias || d->lse || d->o_ld % 8 != 0 || d->o_bs % 8 != 0) return r;
    static const int occ32 = attention_occupancy("PF_ATTENTION_OCC32", 4), min_tiles = attention_occupancy("PF_ATTENTION_SPLIT_MIN_TILES", 8);
    if (occ32 != 4 && occ32 != 5) return r;                                   // (the register-pipelined form walks three LDS buffers: never split)
    const long blocks = cdiv(d->nq, 128) * d->H * d->B, slots = 256L * occ32;
    const int nkt = (d->nk + 63) / 64;
    if (blocks * 20 >= slots * 19) return r;
    int S = static_cast<int>((slots * 19 / 10 + blocks / 2) / blocks);
    S = std::min(S, 8);
    S = std::min(S, nkt / std::max(min_tiles, 2));
    if (S < 2) return r;
    int tiles = static_cast<int>(cdiv(nkt, S));
    tiles += tiles & 1;
    S = static_cast<int>(cdiv(nkt, tiles));
    if (S < 2) return r;
    r.S = S; r.tiles = tiles;
    r.o_bytes = static_cast<size_t>(S) * d->B * d->nq * d->H * d->D * 2;
    r.bytes = r.o_bytes + static_cast<size_t>(S) * d->B * d->H * d->nq * sizeof(float);
    return r;
}

}  // namespace pf

using namespace pf;

extern "C" size_t pf_attention_workspace_size(const pf_attn_desc* d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->nq <= 0 || d->nk <= 0) return 0;
    const AttnSplit sp = attention_split(d);
    return sp.S > 1 ? sp.bytes : 0;
}

extern "C" pf_status pf_attention(const pf_attn_desc* d, void* stream) {
    PF_REQUIRE(d, "pf_attention: null descriptor");
    PF_REQUIRE(d->q && d->k && d->vt && d->out, "pf_attention: null pointer");
    PF_REQUIRE(d->D == 32 || d->D == 64, "pf_attention: head dim %d unsupported (32 or 64)", d->D);
    PF_REQUIRE(d->B > 0 && d->H > 0 && d->nq > 0 && d->nk > 0, "pf_attention: bad sizes");
    PF_REQUIRE(d->q_ld % 8 == 0 && d->k_ld % 8 == 0 && d->o_ld % 4 == 0 && d->vt_ld % 4 == 0,
               "pf_attention: leading dimensions must be multiples of 8 (q,k) / 4 (out, vt)");
    PF_REQUIRE(d->vt_ld >= ((d->nk + 31) / 32) * 32, "pf_attention: vt_ld=%d must cover nk=%d rounded up to 32", d->vt_ld, d->nk);
    PF_REQUIRE(d->q_bs % 8 == 0 && d->k_bs % 8 == 0 && d->vt_bs % 4 == 0 && d->o_bs % 4 == 0, "pf_attention: batch strides misaligned");
    PF_REQUIRE(aligned16(d->q) && aligned16(d->k) && aligned16(d->vt) && aligned16(d->out), "pf_attention: pointers must be 16-byte aligned");
    PF_REQUIRE((d->bias == nullptr) == (d->flags == nullptr), "pf_attention: bias and flags come together");
    if (d->bias) {
        PF_REQUIRE(d->nk % 4 == 0 && d->bias_ld % 4 == 0 && d->bias_ld >= d->nk && aligned16(d->bias), "pf_attention: bias needs nk %% 4 == 0 and an aligned ld >= nk");
        PF_REQUIRE(d->flags_ld >= (d->nk + 31) / 32, "pf_attention: flags_ld too small");
    }
    AttnParams p;
    p.q = static_cast<const unsigned short*>(d->q); p.k = static_cast<const unsigned short*>(d->k);
    p.vt = static_cast<const unsigned short*>(d->vt); p.out = static_cast<unsigned short*>(d->out);
    p.H = d->H; p.nq = d->nq; p.nk = d->nk;
    p.q_ld = d->q_ld; p.k_ld = d->k_ld; p.vt_ld = d->vt_ld; p.o_ld = d->o_ld;
    p.q_bs = d->q_bs; p.k_bs = d->k_bs; p.vt_bs = d->vt_bs; p.o_bs = d->o_bs;
    p.scale_log2e = d->scale * 1.44269504088896340736f;
    p.bias = d->bias; p.bias_ld = d->bias_ld; p.flags = d->flags; p.flags_ld = d->flags_ld;
    p.lse = d->lse;
    p.split_tiles = 0; p.split_o = 0; p.split_lse = 0;
    static const int env_role = attention_occupancy("PF_ATTENTION_PP_ROLE", 3), env_prio = attention_occupancy("PF_ATTENTION_PP_PRIO", 2),
                     env_xcd = attention_occupancy("PF_ATTENTION_XCD", 1),       // (read once; PF_ATTENTION_PP below is read per call)
                     env_steady2 = attention_occupancy("PF_ATTENTION_STEADY2", 1);
    p.steady2 = env_steady2;
    p.pp_role = env_role;
    p.pp_prio = env_prio;
    p.xcd_map = env_xcd;
    const dim3 grid1(static_cast<unsigned>(cdiv(d->nq, 128) * d->H * d->B));      // k_attention_lds: 1-D, decoded in the kernel
    dim3 grid(cdiv(d->nq, 128), d->H, d->B), block(256);
    hipStream_t st = as_stream(stream);
    // (k_attention_lds addresses K / V^T tiles with 32-bit byte offsets inside one (batch, head) slice)
    const bool fits32 = !PF_ATTN_BUFLOAD || (static_cast<long>(d->nk + 64) * d->k_ld * 2 < (1L << 31) && static_cast<long>(d->vt_ld) * (d->D + 1) * 2 < (1L << 31));
    const bool lds = use_lds_attention() && d->vt_ld % 8 == 0 && d->vt_bs % 8 == 0 && fits32;
    // The experimental ping-pong kernels (PF_ATTENTION_PP = 1 / 2) address K and V^T with 32-bit byte offsets inside a (batch, head) slice.
    // Evaluated HERE, outside PF_DISPATCH_16: a preprocessor conditional inside that macro's argument had swallowed the K-offset guard
    // of the one-tile form -- no directives inside macro arguments.
    const bool pp_fits = static_cast<long>(d->nk) * d->k_ld * 2 < (1L << 31) && static_cast<long>(d->vt_ld) * 64 * 2 < (1L << 31);
#ifdef PF_ATTN_PP_TIMING
    const bool pp_lse_ok = true;                                                   // (the timing build writes its stamps over lse)
#else
    const bool pp_lse_ok = !d->lse;                                                // (lse = training forward: exact fp32 row sums there)
#endif
    const bool pp1_ok = pp_lse_ok && d->nk % 8 == 0 && d->nk >= 128 && pp_fits;
    const bool pp2_ok = !d->lse && d->nk % 128 == 0 && d->nk >= 256 && pp_fits;
    PF_DISPATCH_16(d->dtype, "pf_attention",
        if (lds) {
            static const int occ64 = attention_occupancy("PF_ATTENTION_OCC", 3), occ32 = attention_occupancy("PF_ATTENTION_OCC32", 4);
            static const int msum = attention_occupancy("PF_ATTENTION_MSUM", 1);      // 0: softmax row sums on the vector ALU (A/B)
            static const int msum32 = attention_occupancy("PF_ATTENTION_MSUM32", 0);  // the same for the EPA (D = 32, bias) kernel: 168 registers + 3 spilled
            const int pingpong = attention_occupancy("PF_ATTENTION_PP", 0);                // (read per call: the tests switch it)      // 0: k_attention_lds for the D = 64 self-attentions too (A/B)
            if (d->D == 64) {
                if (d->bias) {                 // (not pipelined it needs 169 registers at three waves per SIMD)
                    hipLaunchKernelGGL((k_attention_lds<T, 64, true>), grid1, block, 0, st, p);
                } else if (pingpong == 2 && pp2_ok) {
                    hipLaunchKernelGGL((k_attention_pp2<T>), dim3(static_cast<unsigned>(cdiv(d->nq, 256) * d->H * d->B)), dim3(512), 0, st, p);
                } else if (pingpong && pp1_ok) {
                    hipLaunchKernelGGL((k_attention_pp<T>), dim3(static_cast<unsigned>(cdiv(d->nq, 256) * d->H * d->B)), dim3(512), 0, st, p);
                } else {
                    if (occ64 == 2) hipLaunchKernelGGL((k_attention_lds<T, 64, false>), grid1, block, 0, st, p);
                    else if (msum && !d->lse && d->nk >= 256) hipLaunchKernelGGL((k_attention_lds<T, 64, false, false, 3, true>), grid1, block, 0, st, p);   // (77 text keys: the 4 extra MFMAs of two tiles cost 2 %)
                    else hipLaunchKernelGGL((k_attention_lds<T, 64, false, false, 3>), grid1, block, 0, st, p);
                }
            } else {
                if (d->bias) {
                    const AttnSplit sp = attention_split(d);
                    if (sp.S > 1 && d->workspace && d->workspace_bytes >= sp.bytes && aligned16(d->workspace)) {
                        // split key range: S workgroups per query block write normalised partial outputs + their log-sum-exps, one more launch combines them
                        AttnParams ps = p;
                        ps.out = static_cast<unsigned short*>(d->workspace);
                        ps.lse = reinterpret_cast<float*>(static_cast<char*>(d->workspace) + sp.o_bytes);
                        ps.o_ld = d->H * 32; ps.o_bs = static_cast<long>(d->nq) * d->H * 32;
                        ps.split_tiles = sp.tiles;
                        ps.split_o = static_cast<long>(d->B) * d->nq * d->H * 32;
                        ps.split_lse = static_cast<long>(d->B) * d->H * d->nq;
                        const dim3 grid_s(grid1.x, static_cast<unsigned>(sp.S));
                        if (occ32 == 5) hipLaunchKernelGGL((k_attention_lds<T, 32, true, false, 5>), grid_s, block, 0, st, ps);
                        else hipLaunchKernelGGL((k_attention_lds<T, 32, true, false, 4>), grid_s, block, 0, st, ps);
                        const long total = static_cast<long>(d->B) * d->nq * (d->H * 32 / 8);
                        hipLaunchKernelGGL((k_attention_merge<T>), dim3(static_cast<unsigned>(cdiv(total, 256))), dim3(256), 0, st, ps.out, ps.lse, sp.S, d->B, d->H, 32,
                                           d->nq, p.out, d->o_ld, d->o_bs);
                    }
                    else if (occ32 == 4) hipLaunchKernelGGL((k_attention_lds<T, 32, true, false, 4>), grid1, block, 0, st, p);
                    else if (occ32 == 5) hipLaunchKernelGGL((k_attention_lds<T, 32, true, false, 5>), grid1, block, 0, st, p);
                    else if (msum32 && !d->lse) hipLaunchKernelGGL((k_attention_lds<T, 32, true, true, 3, true>), grid1, block, 0, st, p);
                    else hipLaunchKernelGGL((k_attention_lds<T, 32, true>), grid1, block, 0, st, p);
                } else {
                    if (occ32 == 4) hipLaunchKernelGGL((k_attention_lds<T, 32, false, false, 4>), grid1, block, 0, st, p);
                    else if (occ32 == 5) hipLaunchKernelGGL((k_attention_lds<T, 32, false, false, 5>), grid1, block, 0, st, p);
                    else hipLaunchKernelGGL((k_attention_lds<T, 32, false>), grid1, block, 0, st, p);
                }
            }
        } else {
            if (d->D == 64) hipLaunchKernelGGL((k_attention<T, 64>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((k_attention<T, 32>), grid, block, 0, st, p);
        });
    PF_CHECK_LAUNCH("pf_attention");
    return PF_OK;
}
