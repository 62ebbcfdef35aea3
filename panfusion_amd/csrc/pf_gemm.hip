// MFMA GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
//   out[m][n] = sum_k A[m][k] * W[n][k]  (+ bias[n]) (+ rowvec[img(m)][n]) (+ residual[m][n])
//
// A is never materialised for convolutions: row m = (image, yo, xo) and K-block kb = (tap, 64
// channels) address 128 contiguous bytes of the NHWC activation (zero padding, optional nearest
// x2 upsampling, optional channel concat of two sources are folded into the address).
//
// Two tile kernels share the fragment mapping, the LDS image and the block epilogue:
//   k_conv_gemm   256 threads = 4 wavefronts (2 x 2), tile (32*MREP) x (32*NREP), two LDS slots, 2 blocks per CU;
//   k_conv_gemm8  512 threads = 8 wavefronts (4 x 2), tile 256 x (32*NREP), three LDS slots, persistent over tiles.
// K-step 64, v_mfma_f32_16x16x32 (bf16 or f16) with fp32 accumulation.  The weight tile is the MFMA "A"
// operand and the activation tile the "B" operand, so every lane ends up with 4 CONSECUTIVE output channels
// of one output row (float4 bias loads, 8-byte LDS staging writes).  Operands go global -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds) through scalar buffer descriptors: zero padding and ragged tiles are offsets
// beyond num_records.  LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row>>1)&7 (applied to
// the per-lane SOURCE offset, the DMA writes lane-linear), which makes the ds_read_b128 fragment reads conflict
// free.  Tile ids are remapped so that the tiles sharing an activation row-panel run on one XCD (private L2).
// Epilogue (epilogue_fast): bias / per-image row vector / residual / GEGLU on the fp32 accumulators, one
// rounding, 16-bit tile staged through LDS and written as whole 16-byte row segments.
//
// Replaces cuDNN / cuBLAS behind diffusers Conv2d / Linear (reference call sites
// models/pano/MVGenModel.py:86-144,174-198,224-294; models/modules/transformer.py:8-74).
#include "pf_common.h"
#include "pf_gemm_params.h"
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <type_traits>

namespace pf {


template <typename T> struct Mfma;
template <> struct Mfma<Bf16> {
    typedef __attribute__((ext_vector_type(8))) __bf16 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    // accumulate IN PLACE in the AGPR half of the register file (one wave per SIMD, > 256 registers):
    // with the builtin hipcc copies every AGPR accumulator to a temporary before each MFMA
    static __device__ __forceinline__ void acc_agpr(frag a, frag b, f32x4& c) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
};
template <> struct Mfma<F16> {
    typedef __attribute__((ext_vector_type(8))) _Float16 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void acc_agpr(frag a, frag b, f32x4& c) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
};

__device__ __forceinline__ int lds_off(int row, int chunk) {   // in 16-bit elements
    return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3);
}


// GroupNorm moments of the layer's OUTPUT as a by-product of the epilogue (the consumer's statistics pass re-read
// the whole tensor: 210 MB per fp32 stream tensor at 64 x 64 x 320 x 40 views).  A lane holds 4 consecutive
// columns of the rows (lane & 15) + 16 i of its wavefront's fragment rows: per-lane sums over i, then an
// all-reduce over the 16 lanes of a DPP row (row_ror 8 / 4 / 2 / 1: VALU only, no LDS), and lane 0 of every row
// writes its 4 columns.  One partial row per (wavefront row group): [part][2][N], part = first row / gn_rows --
// fixed summation order, no atomics: results do not depend on scheduling.
__device__ __forceinline__ float row16_allreduce(float v) {
#define PF_ROR(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xF, 0xF, false))
    v += PF_ROR(8);
    v += PF_ROR(4);
    v += PF_ROR(2);
    v += PF_ROR(1);
#undef PF_ROR
    return v;
}
// Moments are kept per column PAIR (2 c, 2 c + 1): every GroupNorm group of the UNets / VAE has an even number of channels
// and pairs never straddle a group.  They are computed in a phase of their own BEFORE the block epilogue, column block by
// column block, while nothing but the accumulators is live -- folded into the epilogue's own loops the 20 + 20 running sums
// pushed the 8-wave kernel from 244 registers into scratch, and every reload drained the DMA queue of the next tile
// (K loop + 15 %, epilogue x 1.8: profiles/archive/r3c_gemm_gn.txt).  The phase re-reads the tile's residual (the epilogue proper
// reads it again, from L2); bias and the image's time-embedding row are one float4 each per column block -- an image is
// whole runs of gn_rows rows (gn_rows_for), so a wavefront's rows belong to ONE image.
template <typename T, int MREP, int NREP, int RES>            // RES: 0 none, 1 fp32 residual, 2 16-bit residual
__device__ __forceinline__ void gn_moments_phase_r(const GemmParams& p, long bz, const f32x4 (&acc)[MREP][NREP],
                                                   int m0, int n0, int row_base, int col_base, int lane) {
    const int cq4 = 4 * (lane >> 4), rl = lane & 15;
    const int NP = p.N >> 1;
    const int mw = m0 + row_base;                                 // first row of this wavefront
    if (mw >= p.M) return;
    float* base = p.gn_partial + gn_part(p, bz, mw) * 2 * NP;
    const float* rv = p.rowvec ? p.rowvec + static_cast<long>(mw / p.rows_per_img) * p.rowvec_ld : nullptr;
    const int nb = n0 + col_base + cq4;                           // (whole N tiles: gn_rows_for) column of block j: nb + 16 j
    // operands are requested in batches (a few memory latencies for the whole phase, not one per column block): bias + row
    // vector of all column blocks first, then the residual of HALF the fragment rows at a time (40 registers; all 80 at once
    // sent the kernel back to scratch)
    float4 c[NREP];
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
        c[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nb + 16 * j) : float4{0.f, 0.f, 0.f, 0.f};
        if (rv) {
            const float4 r = *reinterpret_cast<const float4*>(rv + nb + 16 * j);
            c[j].x += r.x; c[j].y += r.y; c[j].z += r.z; c[j].w += r.w;
        }
    }
    constexpr bool res32 = RES == 1, res16 = RES == 2;
    float sm[NREP][2], sq[NREP][2];
#pragma unroll
    for (int j = 0; j < NREP; ++j) { sm[j][0] = sm[j][1] = 0.f; sq[j][0] = sq[j][1] = 0.f; }
    constexpr int HR = MREP >= 2 ? MREP / 2 : 1;                  // fragment rows per batch
#pragma unroll
    for (int i0 = 0; i0 < MREP; i0 += HR) {
        float4 r32[res32 ? HR : 1][NREP];
        u16x4 r16[res16 ? HR : 1][NREP];
        if constexpr (res32) {
#pragma unroll
            for (int ii = 0; ii < HR; ++ii) {
                const float* rp = static_cast<const float*>(p.residual) + bz * p.res_bs + static_cast<long>(min(mw + (i0 + ii) * 16 + rl, p.M - 1)) * p.res_ld + nb;
#pragma unroll
                for (int j = 0; j < NREP; ++j) r32[ii][j] = *reinterpret_cast<const float4*>(rp + 16 * j);
            }
        } else if constexpr (res16) {
#pragma unroll
            for (int ii = 0; ii < HR; ++ii) {
                const unsigned short* rp = static_cast<const unsigned short*>(p.residual) + bz * p.res_bs + static_cast<long>(min(mw + (i0 + ii) * 16 + rl, p.M - 1)) * p.res_ld + nb;
#pragma unroll
                for (int j = 0; j < NREP; ++j) r16[ii][j] = *reinterpret_cast<const u16x4*>(rp + 16 * j);
            }
        }
#pragma unroll
        for (int ii = 0; ii < HR; ++ii) {
            const int i = i0 + ii;
            const float live = mw + i * 16 + rl < p.M ? 1.f : 0.f;        // (rows past M: clamped copies)
#pragma unroll
            for (int j = 0; j < NREP; ++j) {
                float x0 = acc[i][j][0] + c[j].x, x1 = acc[i][j][1] + c[j].y, x2 = acc[i][j][2] + c[j].z, x3 = acc[i][j][3] + c[j].w;
                if constexpr (res32) { x0 += r32[ii][j].x; x1 += r32[ii][j].y; x2 += r32[ii][j].z; x3 += r32[ii][j].w; }
                else if constexpr (res16) { x0 += to_f32<T>(r16[ii][j][0]); x1 += to_f32<T>(r16[ii][j][1]); x2 += to_f32<T>(r16[ii][j][2]); x3 += to_f32<T>(r16[ii][j][3]); }
                x0 *= live; x1 *= live; x2 *= live; x3 *= live;
                sm[j][0] += x0 + x1; sm[j][1] += x2 + x3;
                sq[j][0] += x0 * x0 + x1 * x1; sq[j][1] += x2 * x2 + x3 * x3;
            }
        }
        __builtin_amdgcn_sched_barrier(0);                        // (one batch of residual registers at a time)
    }
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
        const float s0 = row16_allreduce(sm[j][0]), s1 = row16_allreduce(sm[j][1]);
        const float q0 = row16_allreduce(sq[j][0]), q1 = row16_allreduce(sq[j][1]);
        if (rl == 0) {
            *reinterpret_cast<float2*>(base + ((nb + 16 * j) >> 1)) = float2{s0, s1};
            *reinterpret_cast<float2*>(base + NP + ((nb + 16 * j) >> 1)) = float2{q0, q1};
        }
    }
}

template <typename T, int MREP, int NREP>
__device__ __forceinline__ void gn_moments_phase(const GemmParams& p, long bz, const f32x4 (&acc)[MREP][NREP],
                                                 int m0, int n0, int row_base, int col_base, int lane) {
    if (!p.residual) gn_moments_phase_r<T, MREP, NREP, 0>(p, bz, acc, m0, n0, row_base, col_base, lane);
    else if (p.res_f32) gn_moments_phase_r<T, MREP, NREP, 1>(p, bz, acc, m0, n0, row_base, col_base, lane);
    else gn_moments_phase_r<T, MREP, NREP, 2>(p, bz, acc, m0, n0, row_base, col_base, lane);
}

// Block epilogue.  All arithmetic (bias, per-image row vector, residual, GEGLU) runs in the MFMA
// fragment layout on the fp32 accumulators -- one rounding to 16 bit -- and the finished 16-bit tile is
// staged through LDS so that it leaves as whole 16-byte row segments (a fragment store touches 16
// different rows with 8 bytes each).  fp32 output, ragged N or an unaligned out_ld take the direct
// per-fragment store.  smem16: the block's LDS (the operand ring is dead after the K loop).
struct NoOp { __device__ void operator()() const {} };

// Staged block epilogue, specialised at compile time for the operand mix of the layer (MODE):
//   0 bias | 1 bias + per-image row vector (time embedding) | 2 bias + residual | 3 bias + GEGLU pairing.
// Straight-line code: rows beyond M / quads beyond N are CLAMPED to the last valid one instead of being
// predicated (their results are dropped by the bounds check of the copy-out), the operands of two
// 16-row fragment groups are requested one group ahead of the arithmetic, and nothing in the loop
// waits for memory more than once per group.  The finished 16-bit tile goes through LDS in HALVES row
// slices (every wave stages MREP / HALVES of its fragment rows per slice) and leaves as 16-byte row segments.
template <typename T, int MREP, int NREP, int NT, int BM, int BN, int HALVES, int MODE>
__device__ __forceinline__ void epilogue_fast(const GemmParams& p, long bz, f32x4 (&acc)[MREP][NREP], unsigned short* smem16,
                                              int m0, int n0, int row_base, int col_base, int lane, int t) {
    constexpr int SLD = (MODE == 3 ? BN / 2 : BN) + 8;            // 16-bit elements per staged row (GEGLU: half as many columns)
    constexpr int BMH = BM / HALVES;                              // staged rows per slice
    constexpr int WR = 16 * MREP, WRH = WR / HALVES, IH = MREP / HALVES;
    constexpr int G = 2, NG = MREP / G;                           // fragment rows per operand group
    static_assert(MREP % HALVES == 0 && IH % G == 0, "slices are whole operand groups");
    const int cq = 4 * (lane >> 4), rl = lane & 15;
    const int n_store = MODE == 3 ? p.N >> 1 : p.N;
    int ncl[NREP];
#pragma unroll
    for (int j = 0; j < NREP; ++j) ncl[j] = min(n0 + col_base + j * 16 + cq, p.N - 4);
    float4 bias[NREP];
#pragma unroll
    for (int j = 0; j < NREP; ++j) bias[j] = float4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NREP; ++j) bias[j] = *reinterpret_cast<const float4*>(p.bias + ncl[j]);
    }
    const unsigned short* resp = MODE == 2 ? static_cast<const unsigned short*>(p.residual) + bz * p.res_bs : nullptr;
    // MODE 1 (requires rows_per_img >= BM: the tile touches two images at most): both candidate row
    // vectors are fetched up front, a row picks one by comparing its offset with the image boundary
    float4 rv0[NREP], rv1[NREP];
    int boundary = 0;                                             // first tile row of the second image
    if (MODE == 1) {
        const int img0 = m0 / p.rows_per_img, img_last = (p.M - 1) / p.rows_per_img;
        boundary = (img0 + 1) * p.rows_per_img - m0;
        const float* r0 = p.rowvec + static_cast<long>(img0) * p.rowvec_ld;
        const float* r1 = p.rowvec + static_cast<long>(min(img0 + 1, img_last)) * p.rowvec_ld;
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            rv0[j] = *reinterpret_cast<const float4*>(r0 + ncl[j]);
            rv1[j] = *reinterpret_cast<const float4*>(r1 + ncl[j]);
        }
    }
    u16x4 res[2][G][NREP];                                        // residual, requested one group ahead (MODE 2)
    auto request = [&](auto g_tag, auto buf_tag) __attribute__((always_inline)) {
        constexpr int g = decltype(g_tag)::value, bf = decltype(buf_tag)::value;
        if (MODE != 2) return;
#pragma unroll
        for (int ii = 0; ii < G; ++ii) {
            const int mc = min(m0 + row_base + (g * G + ii) * 16 + rl, p.M - 1);
            const unsigned short* rp = resp + static_cast<long>(mc) * p.res_ld;
#pragma unroll
            for (int j = 0; j < NREP; ++j) res[bf][ii][j] = *reinterpret_cast<const u16x4*>(rp + ncl[j]);
        }
    };
    auto arithmetic = [&](auto g_tag, auto buf_tag) __attribute__((always_inline)) {
        constexpr int g = decltype(g_tag)::value, bf = decltype(buf_tag)::value;
#pragma unroll
        for (int ii = 0; ii < G; ++ii) {
            constexpr int i0 = g * G;
            const int i = i0 + ii, h = i0 / IH;
            const int r = row_base / HALVES + (i - h * IH) * 16 + rl;         // staged row <-> tile row row_base + i*16 + rl
#pragma unroll
            for (int j = 0; j < NREP; ++j) {
                const int c = col_base + j * 16 + cq;
                float v[4] = {acc[i][j][0] + bias[j].x, acc[i][j][1] + bias[j].y, acc[i][j][2] + bias[j].z, acc[i][j][3] + bias[j].w};
                if (MODE == 1) {
                    const bool second = row_base + i * 16 + rl >= boundary;
                    v[0] += second ? rv1[j].x : rv0[j].x; v[1] += second ? rv1[j].y : rv0[j].y;
                    v[2] += second ? rv1[j].z : rv0[j].z; v[3] += second ? rv1[j].w : rv0[j].w;
                }
                if (MODE == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += to_f32<T>(res[bf][ii][j][e]);
                }
                if (MODE == 3) {
                    typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
                    u16x2 w2;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float gt = v[2 * e + 1];
                        w2[e] = from_f32<T>(geglu_value(v[2 * e], gt));
                    }
                    *reinterpret_cast<u16x2*>(smem16 + r * SLD + (c >> 1)) = w2;
                } else {
                    u16x4 w4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w4[e] = from_f32<T>(v[e]);
                    *reinterpret_cast<u16x4*>(smem16 + r * SLD + c) = w4;
                }
            }
        }
    };
    unsigned short* outp = static_cast<unsigned short*>(p.out) + bz * p.out_bs;
    auto copy_out = [&](int h) __attribute__((always_inline)) {
        constexpr int CPR = MODE == 3 ? BN / 16 : BN / 8;         // 16-byte chunks per tile row
        const int nbase = MODE == 3 ? n0 >> 1 : n0;
#pragma unroll
        for (int k = 0; k < (BMH * CPR + NT - 1) / NT; ++k) {
            const int q = t + k * NT, r = q / CPR, c8 = (q - r * CPR) * 8;
            const int m = m0 + (r / WRH) * WR + h * WRH + r % WRH;
            if (q < BMH * CPR && m < p.M && nbase + c8 < n_store) {
                const u16x8 x = *reinterpret_cast<const u16x8*>(smem16 + r * SLD + c8);
                *reinterpret_cast<u16x8*>(outp + out_row(p, bz, m) * p.out_ld + nbase + c8) = x;
            }
        }
    };
    auto group = [&](auto g_tag) __attribute__((always_inline)) {
        constexpr int g = decltype(g_tag)::value;
        if constexpr (g + 1 < NG) request(std::integral_constant<int, g + 1>(), std::integral_constant<int, (g + 1) & 1>());
        arithmetic(g_tag, std::integral_constant<int, g & 1>());
        if constexpr (((g + 1) * G) % IH == 0) {                  // slice complete
            constexpr int h = (g * G) / IH;
            stamp(p, 6 + 3 * h);
            __syncthreads();
            stamp(p, 7 + 3 * h);
            copy_out(h);
            stamp(p, 8 + 3 * h);
            if constexpr (h + 1 < HALVES) __syncthreads();        // slice read out before the next one is staged
        }
    };
    request(std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
    group(std::integral_constant<int, 0>());
    if constexpr (NG > 1) group(std::integral_constant<int, 1>());
    if constexpr (NG > 2) group(std::integral_constant<int, 2>());
    if constexpr (NG > 3) group(std::integral_constant<int, 3>());
    static_assert(NG <= 4, "unrolled by hand");
}

// fp32 block epilogue of the mixed-precision scheme (the residual stream is fp32): bias (+ fp32 residual)
// on the accumulators, stored straight from the fragments -- a lane owns 4 consecutive fp32 columns
// (16 bytes), the four column groups of a fragment make a 64-byte row segment, so no LDS staging is
// needed.  Straight-line like epilogue_fast: rows / quads beyond the edge are clamped for the loads and
// only the store is predicated; the residual of a group of two fragment rows is requested one group ahead.
template <typename T, int MREP, int NREP, bool RES, bool PAIR = false>
__device__ __forceinline__ void epilogue_f32(const GemmParams& p, long bz, f32x4 (&acc)[MREP][NREP],
                                             int m0, int n0, int row_base, int col_base, int lane) {
    constexpr int G = 1, NG = MREP / G;                           // (two fp32 row groups in flight would spill)
    const int cq = 4 * (lane >> 4), rl = lane & 15;
    int ncl[NREP];
    bool nok[NREP];
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
        const int n4 = n0 + col_base + j * 16 + cq;
        nok[j] = n4 < p.N;
        ncl[j] = min(n4, p.N - 4);
    }
    float4 bias[NREP];
#pragma unroll
    for (int j = 0; j < NREP; ++j) bias[j] = float4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NREP; ++j) bias[j] = *reinterpret_cast<const float4*>(p.bias + ncl[j]);
    }
    const float* resp = RES ? static_cast<const float*>(p.residual) + bz * p.res_bs : nullptr;
    float* outp = static_cast<float*>(p.out) + bz * p.out_bs;
    unsigned short* outp16 = static_cast<unsigned short*>(p.out) + bz * p.out_bs;    // PAIR: [M][per 32 columns: hi(32) | lo(32)] 16-bit
    float4 res[2][G][NREP];
    auto request = [&](int g, int bf) __attribute__((always_inline)) {
        if (!RES) return;
#pragma unroll
        for (int ii = 0; ii < G; ++ii) {
            const int mc = min(m0 + row_base + (g * G + ii) * 16 + rl, p.M - 1);
            const float* rp = resp + static_cast<long>(mc) * p.res_ld;
#pragma unroll
            for (int j = 0; j < NREP; ++j) res[bf][ii][j] = *reinterpret_cast<const float4*>(rp + ncl[j]);
        }
    };
    request(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) request(g + 1, (g + 1) & 1);
#pragma unroll
        for (int ii = 0; ii < G; ++ii) {
            const int i = g * G + ii;
            const int m = m0 + row_base + i * 16 + rl;
            float* op = outp + out_row(p, bz, min(m, p.M - 1)) * p.out_ld;
#pragma unroll
            for (int j = 0; j < NREP; ++j) {
                float4 v = float4{acc[i][j][0] + bias[j].x, acc[i][j][1] + bias[j].y, acc[i][j][2] + bias[j].z, acc[i][j][3] + bias[j].w};
                if (RES) {
                    const float4 r = res[g & 1][ii][j];
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if (PAIR) {
                    if (m < p.M && nok[j]) {
                        unsigned short* o16 = outp16 + out_row(p, bz, m) * p.out_ld + pair_off(ncl[j]);
                        const float f[4] = {v.x, v.y, v.z, v.w};
                        u16x4 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            hi[e] = from_f32<T>(f[e]);
                            lo[e] = from_f32<T>(f[e] - to_f32<T>(hi[e]));
                        }
                        *reinterpret_cast<u16x4*>(o16) = hi;
                        *reinterpret_cast<u16x4*>(o16 + 32) = lo;
                    }
                } else if (m < p.M && nok[j]) *reinterpret_cast<float4*>(op + ncl[j]) = v;
            }
        }
    }
}

// fp32 tile WITH the GroupNorm moments of the finished output (bias + residual included) from the same registers: column block by
// column block (j outermost), so that only the 4 running sums of ONE block are live next to the two residual buffers (4 fragment rows
// each: 32 registers against the 40 of epilogue_f32; 256 registers, no scratch) -- the separate moment phase above has to read the
// residual tile a second time, which cost a residual layer as much as the consumer's statistics pass saves
// (profiles/archive/r3d_gemm_gn.txt).  (Row group by row group with all 20 sums live -- the store order of epilogue_f32 -- spills 10-73
// registers in the persistent kernel, whichever way the bias and the residual buffer are arranged.)  Whole N tiles (gn_rows_for);
// rows past M are clamped and masked.
template <typename T, int MREP, int NREP, bool RES>
__device__ __forceinline__ void epilogue_f32_stats(const GemmParams& p, long bz, f32x4 (&acc)[MREP][NREP],
                                                   int m0, int n0, int row_base, int col_base, int lane) {
    const int cq = 4 * (lane >> 4), rl = lane & 15;
    const int NP = p.N >> 1;
    const int mw = m0 + row_base;                                 // first row of this wavefront
    if (mw >= p.M) return;
    float* base = p.gn_partial + gn_part(p, bz, mw) * 2 * NP;
    const float* resp = RES ? static_cast<const float*>(p.residual) + bz * p.res_bs : nullptr;
    float* outp = static_cast<float*>(p.out) + bz * p.out_bs;
    const int nb = n0 + col_base + cq;
    float4 res[2][MREP];
    auto request = [&](int j, int bf) __attribute__((always_inline)) {
        if (!RES) return;
#pragma unroll
        for (int i = 0; i < MREP; ++i)
            res[bf][i] = *reinterpret_cast<const float4*>(resp + static_cast<long>(min(mw + i * 16 + rl, p.M - 1)) * p.res_ld + nb + 16 * j);
    };
    request(0, 0);
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
        if (j + 1 < NREP) request(j + 1, (j + 1) & 1);
        const float4 b = p.bias ? *reinterpret_cast<const float4*>(p.bias + nb + 16 * j) : float4{0.f, 0.f, 0.f, 0.f};
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            const int m = mw + i * 16 + rl;
            float4 v = float4{acc[i][j][0] + b.x, acc[i][j][1] + b.y, acc[i][j][2] + b.z, acc[i][j][3] + b.w};
            if (RES) {
                const float4 r = res[j & 1][i];
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            const bool live = m < p.M;
            if (live) *reinterpret_cast<float4*>(outp + static_cast<long>(m) * p.out_ld + nb + 16 * j) = v;
            const float lf = live ? 1.f : 0.f;
            v.x *= lf; v.y *= lf; v.z *= lf; v.w *= lf;
            s0 += v.x + v.y; s1 += v.z + v.w;
            q0 += v.x * v.x + v.y * v.y; q1 += v.z * v.z + v.w * v.w;
        }
        s0 = row16_allreduce(s0); s1 = row16_allreduce(s1);
        q0 = row16_allreduce(q0); q1 = row16_allreduce(q1);
        if (rl == 0) {
            *reinterpret_cast<float2*>(base + ((nb + 16 * j) >> 1)) = float2{s0, s1};
            *reinterpret_cast<float2*>(base + NP + ((nb + 16 * j) >> 1)) = float2{q0, q1};
        }
    }
}

// Block epilogue.  All arithmetic (bias, per-image row vector, residual, GEGLU) runs in the MFMA
// fragment layout on the fp32 accumulators -- one rounding to 16 bit.  16-bit outputs with 16-byte
// aligned rows take one of the staged specialisations above; everything else (fp32 output, ragged or
// unaligned rows, row vector AND residual together) takes the per-fragment store.  smem16: the LDS
// region the staged tile may use.  HALVES > 1: staged in row slices through a smaller region;
// after_ring() runs right after the barrier that retires the operand ring -- the persistent kernel
// requests the next tile's first stages there, into ring slots the staging region does not overlap.
template <typename T, int MREP, int NREP, int NT, int BM, int BN, int HALVES = 1, bool STATS = false, typename F = NoOp>
__device__ __forceinline__ void epilogue_tile(const GemmParams& p, long bz, f32x4 (&acc)[MREP][NREP],
                                              unsigned short* smem16, int m0, int n0, int row_base, int col_base,
                                              int lane, int t, F after_ring = F()) {
    if constexpr (STATS) {
        // GroupNorm-moment instantiation of the kernel (pf_conv_desc.gn_partial; a separate instantiation so that the
        // default kernels keep their code and register allocation untouched).  Ahead of the ring barrier and of the next
        // tile's DMA requests: the phase's own loads would otherwise queue behind them in vmcnt order (it cost 8 k clocks
        // per tile there), and the waves that arrive early spend their barrier wait on it.
        // (fp32 tiles with an fp32 residual produce their moments inside the store loop instead: epilogue_f32_stats)
        if (!(p.out_f32 && p.residual)) gn_moments_phase<T, MREP, NREP>(p, bz, acc, m0, n0, row_base, col_base, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                              // every wave is done reading the operand ring
    stamp(p, 4);
    after_ring();
    stamp(p, 5);
    if constexpr (STATS) {
        if (p.out_f32 && p.residual) {                            // (gn_rows_for admitted it: fp32 residual, aligned rows, no row vector)
            epilogue_f32_stats<T, MREP, NREP, true>(p, bz, acc, m0, n0, row_base, col_base, lane);
            return;
        }
    }
    const int n_store = p.geglu ? p.N >> 1 : p.N;
    if (p.split_out && !p.rowvec && (p.N & 3) == 0 && (p.out_ld & 3) == 0 && p.N >= 4) {
        if (!p.residual) { epilogue_f32<T, MREP, NREP, false, true>(p, bz, acc, m0, n0, row_base, col_base, lane); return; }
        if (p.res_f32 && (p.res_ld & 3) == 0) { epilogue_f32<T, MREP, NREP, true, true>(p, bz, acc, m0, n0, row_base, col_base, lane); return; }
    }
    if (p.out_f32 && !p.geglu && !p.rowvec && (p.N & 3) == 0 && (p.out_ld & 3) == 0 && p.N >= 4) {
        if (!p.residual) { epilogue_f32<T, MREP, NREP, false>(p, bz, acc, m0, n0, row_base, col_base, lane); return; }
        if (p.res_f32 && (p.res_ld & 3) == 0) { epilogue_f32<T, MREP, NREP, true>(p, bz, acc, m0, n0, row_base, col_base, lane); return; }
    }
    const bool staged = !p.out_f32 && !p.res_f32 && !p.split_out && (p.out_ld & 7) == 0 && (n_store & 7) == 0 && (p.N & 3) == 0;
    if (staged && !(p.rowvec && p.residual) && !(p.geglu && (p.rowvec || p.residual))) {
        // GEGLU halves the columns: the whole 256 x 80 tile fits the staging slot at once -- one barrier pair instead of two
        // (the FF1 tile of a K = 320 layer spends 3 x 1.2 k of its 24.6 k clocks waiting at them)
        if (p.geglu) epilogue_fast<T, MREP, NREP, NT, BM, BN, 1, 3>(p, bz, acc, smem16, m0, n0, row_base, col_base, lane, t);
        else if (p.rowvec && p.rows_per_img < BM) goto generic;
        else if (p.rowvec) epilogue_fast<T, MREP, NREP, NT, BM, BN, HALVES, 1>(p, bz, acc, smem16, m0, n0, row_base, col_base, lane, t);
        else if (p.residual) epilogue_fast<T, MREP, NREP, NT, BM, BN, HALVES, 2>(p, bz, acc, smem16, m0, n0, row_base, col_base, lane, t);
        else epilogue_fast<T, MREP, NREP, NT, BM, BN, HALVES, 0>(p, bz, acc, smem16, m0, n0, row_base, col_base, lane, t);
        return;
    }
generic:
    const int cq = 4 * (lane >> 4), rl = lane & 15;
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = m0 + row_base + i * 16 + rl;
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            const int n4 = n0 + col_base + j * 16 + cq;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (m < p.M && n4 < p.N) epilogue_store<T>(p, bz, m, n4, v);
        }
    }
}

// Split-K combine INSIDE the GEMM launch (no second kernel: the 8 x 8 / 16 x 16 levels and the whole panorama branch paid
// one ~8 us launch per split layer, 200+ per step, most of them on the latency-bound chains the branch joins wait for).
// Every K-slice block publishes its fp32 slab and draws a ticket; the block that draws the last one combines the slabs IN
// SPLIT ORDER (the same sums as k_splitk_reduce: results do not depend on arrival order) and runs the epilogue.
// Inter-workgroup visibility (cdna_hip_programming.md section 6 Guideline 16, write-through form): sc1 slab stores ->
// s_waitcnt vmcnt(0) in every wave -> barrier -> lane 0: relaxed agent-scope fetch_add; the last arriver: lane 0
// agent-scope acquire fence (drops this CU's L1) -> barrier -> plain loads.  Nobody waits for anybody:
// no residency assumption, no deadlock.  The last arriver zeroes the counter for the next launch that is handed the slot.

template <typename T, int BM, int BN, int NT>
__device__ __forceinline__ void splitk_finish(const GemmParams& p, long bz, int tile_lin, int m0, int n0, int* flag, int t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's slab stores have left the wave
    __syncthreads();                                              // (also: every wave is done with the operand ring -> `flag` may live there)
    int* ticket = p.tickets + (bz * p.mtiles * p.ntiles + tile_lin);
    if (t == 0) {                                                 // (write-through slabs + drained stores: no release fence)
        const int drawn = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = drawn == p.splits - 1 ? 1 : 0;
    }
    __syncthreads();
    const bool last = *flag != 0;
    __syncthreads();                                              // (flag read by everyone before the ring is reused)
    if (!last) return;
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    constexpr int QPR = BN / 4;
    const int Ms = p.M - p.m_begin;
    const long slab = static_cast<long>(p.batch) * Ms * p.N;
    for (int q = t; q < BM * QPR; q += NT) {
        const int r = q / QPR, m = m0 + r, n4 = n0 + (q - r * QPR) * 4;
        if (m >= p.M || n4 >= p.N) continue;
        const float* src = p.partial + (bz * Ms + (m - p.m_begin)) * p.N + n4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < p.splits; ++s) {
            const float4 x = *reinterpret_cast<const float4*>(src + s * slab);
            v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
        }
        epilogue_store<T>(p, bz, m, n4, v);
    }
    if (t == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// STAGES (round 6): ring depth.  2 = the original form, two blocks per CU, ONE K step of DMA look-ahead and a full drain per step -- every
// step then costs a whole L2 round trip (~1 us per 64-wide step measured on the panorama branch's small-M layers: M 1024 N 1280 K 1280
// ran 20 steps in 20.8 us, 0.1 of peak).  4 = one block per CU (115 KB), stage it+3 requested while stage it is multiplied and awaited with a
// COUNTED s_waitcnt that leaves the two younger stages in flight: the launch-latency-bound problems whose grid is one round of the chip
// anyway (180 launches per step; the floor of every sharded rank).
template <typename T, int MREP, int NREP, bool STATS = false, bool S3 = false, int STAGES = 2>
__global__ __launch_bounds__(256, STAGES > 2 ? 1 : 2) void k_conv_gemm(const GemmParams p) {
    constexpr int BM = 32 * MREP, BN = 32 * NREP;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short* As = smem;                       // [STAGES][BM][64]
    unsigned short* Bs = smem + STAGES * BM * 64;    // [STAGES][BN][64]

    // XCD-aware, bijective remap of the 1-D tile id (8 XCDs, block b runs on XCD b % 8).
    const int ntile_total = p.mtiles * p.ntiles;
    int tid_lin = blockIdx.x;
    {
        const int q = ntile_total / 8, r = ntile_total % 8;
        const int xcd = tid_lin % 8, idx = tid_lin / 8;
        tid_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = tid_lin % p.ntiles, tile_m = tid_lin / p.ntiles;
    const int m0 = p.m_begin + tile_m * BM, n0 = tile_n * BN;
    const long bz = blockIdx.z;
    const unsigned short* a0 = p.a0 + bz * p.a_bs;
    const unsigned short* a1 = p.a1 ? p.a1 + bz * p.a_bs : nullptr;
    const unsigned short* wg = p.w + bz * p.w_bs;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // staging ownership: global_load_lds_dwordx4 writes LDS at (wave-uniform base + 16 * lane), i.e.
    // lane l of wave w fills physical chunk l%8 of row 8w + l/8 of a 32-row pass.  The XOR swizzle
    // therefore moves to the SOURCE address: the lane fetches the logical chunk that belongs in its
    // physical slot (same 128-B row segment, coalescing intact).  (row>>1)&7 only depends on lrow.
    const int chunk = t & 7, lrow = t >> 3;
    const int lchunk8 = (chunk ^ ((lrow >> 1) & 7)) * 8;

    // per-thread staging rows: output pixel -> top-left input coordinate
    const int pad_y = p.subpix ? 1 - (static_cast<int>(bz) >> 1) : p.pad, pad_x = p.subpix ? 1 - (static_cast<int>(bz) & 1) : p.pad;   // (sub-pixel phase: pf_gemm_params.h)
    int a_img[MREP], a_y[MREP], a_x[MREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = m0 + i * 32 + lrow;
        if (m < p.M) {
            const int img = m / p.rows_per_img, rem = m - img * p.rows_per_img;
            const int yo = rem / p.w_out;
            a_img[i] = img;
            a_y[i] = yo * p.stride - pad_y;
            a_x[i] = (rem - yo * p.w_out + p.crop) * p.stride - pad_x;
        } else {
            a_img[i] = 0; a_y[i] = -(1 << 20); a_x[i] = 0;     // never in range
        }
        if (p.fastseg) seg_pack(a_img[i], a_y[i], a_x[i], p.h_in, p.w_in);      // (pf_gemm_params.h: a_img = pixel index, a_y = validity bits from here on)
    }
    // DMA addressing as in k_conv_gemm8: scalar buffer descriptors + per-thread byte offset (VGPR) +
    // per-stage scalar offset; out-of-range offsets (zero padding, ragged tiles) read as zero.
    constexpr unsigned OOB = 0x80000000u;
    auto uniform_ptr = [](const unsigned short* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
        const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
        return reinterpret_cast<unsigned short*>(static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi) << 32));
    };
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wg), 0, __builtin_amdgcn_readfirstlane(p.w_bytes), 0x00020000);
    unsigned w_off[NREP];                                         // bytes
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
        const int n = n0 + j * 32 + lrow;
        w_off[j] = n < p.N ? static_cast<unsigned>(n * p.K + lchunk8) * 2u : OOB;
    }
    const int Ctot = p.c0 + p.c1;
    const int Hl = p.h_in << p.up, Wl = (p.w_in + 2 * p.wrap) << p.up;

    // K walk (wave-uniform state): K-block kb = (tap, source, 64-channel block); no divisions in the loop
    const int nkb = p.K / 64;
    const int kb0 = blockIdx.y * p.kb_per_split;
    const int kb1 = min(nkb, kb0 + p.kb_per_split);
    int kg = kb0 * 64;
    int tap = kg / Ctot, cc = kg - tap * Ctot;
    unsigned a_off[MREP];                                         // bytes, or OOB
    bool seg1 = false;
    auto set_segment = [&]() {
        const int ky = p.ksize == 3 ? tap / 3 : p.ksize == 2 ? tap >> 1 : 0, kx = p.ksize == 3 ? tap - 3 * ky : p.ksize == 2 ? tap & 1 : 0;
        seg1 = __builtin_amdgcn_readfirstlane(cc >= p.c0 ? 1 : 0) != 0;
        const int ld = seg1 ? p.a1_ld : p.a0_ld;
        if (p.fastseg) {
            const unsigned sel = (1u << ky) | (16u << kx);
            const int tap_pix = ky * p.w_in + kx;
#pragma unroll
            for (int i = 0; i < MREP; ++i) {
                const unsigned off = ((static_cast<unsigned>(a_img[i]) + static_cast<unsigned>(tap_pix)) * static_cast<unsigned>(ld) + static_cast<unsigned>(lchunk8)) * 2u;
                a_off[i] = (static_cast<unsigned>(a_y[i]) & sel) == sel ? off : OOB;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            const int yi = a_y[i] + ky, xi = a_x[i] + kx;             // xi: column in the (virtually wrap-padded, upsampled) input
            const bool ok = yi >= 0 && yi < Hl && xi >= 0 && xi < Wl;
            int sx = (xi >> p.up) - p.wrap;                            // source column: the padding is circular
            sx += sx < 0 ? p.w_in : 0;
            sx -= sx >= p.w_in ? p.w_in : 0;
            const int pix = (a_img[i] * p.h_in + (yi >> p.up)) * p.w_in + sx;
            a_off[i] = ok ? static_cast<unsigned>(pix * ld + lchunk8) * 2u : OOB;
        }
    };
    set_segment();

    auto lds_dma = [&](const __amdgpu_buffer_rsrc_t& r, unsigned short* dst, unsigned voff, int soff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
    };
    auto dma_stage = [&](int buf) {
        const int soff_a = __builtin_amdgcn_readfirstlane((seg1 ? cc - p.c0 : cc) * 2);
        const int soff_w = __builtin_amdgcn_readfirstlane(kg * 2);
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(seg1 ? a1 : a0), 0, __builtin_amdgcn_readfirstlane(seg1 ? p.a1_bytes : p.a0_bytes), 0x00020000);
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            unsigned short* dst = As + buf * BM * 64 + (i * 32 + wave * 8) * 64;
            lds_dma(rs_a, dst, a_off[i], soff_a);
        }
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            unsigned short* dst = Bs + buf * BN * 64 + (j * 32 + wave * 8) * 64;
            lds_dma(rs_w, dst, w_off[j], soff_w);
        }
        kg += 64;
        cc += 64;
        if (cc == Ctot) { cc = 0; ++tap; set_segment(); }
        else if (cc == p.c0) set_segment();
    };

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef typename Mfma<T>::frag frag;
    const int frow = lane & 15, fchunk = lane >> 4;
    auto compute = [&](int buf) {
        if constexpr (S3) {
            // split-precision K block: slab 0 = (W_hi, A_hi), slab 1 = (W_lo, A_lo) of the same 32 channels
            frag af[2][MREP], bf[2][NREP];
#pragma unroll
            for (int slab = 0; slab < 2; ++slab) {
#pragma unroll
                for (int i = 0; i < MREP; ++i)
                    af[slab][i] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(As + buf * BM * 64 + lds_off(wm * 16 * MREP + i * 16 + frow, slab * 4 + fchunk)));
#pragma unroll
                for (int j = 0; j < NREP; ++j)
                    bf[slab][j] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(Bs + buf * BN * 64 + lds_off(wn * 16 * NREP + j * 16 + frow, slab * 4 + fchunk)));
            }
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) {
                    acc[i][j] = Mfma<T>::run(bf[0][j], af[0][i], acc[i][j]);
                    acc[i][j] = Mfma<T>::run(bf[0][j], af[1][i], acc[i][j]);
                    acc[i][j] = Mfma<T>::run(bf[1][j], af[0][i], acc[i][j]);
                }
            return;
        }
#pragma unroll
        for (int slab = 0; slab < 2; ++slab) {
            frag af[MREP], bf[NREP];
#pragma unroll
            for (int i = 0; i < MREP; ++i) {
                const int brow = wm * 16 * MREP + i * 16 + frow;
                u16x8 v = *reinterpret_cast<const u16x8*>(As + buf * BM * 64 + lds_off(brow, slab * 4 + fchunk));
                af[i] = __builtin_bit_cast(frag, v);
            }
#pragma unroll
            for (int j = 0; j < NREP; ++j) {
                const int brow = wn * 16 * NREP + j * 16 + frow;
                u16x8 v = *reinterpret_cast<const u16x8*>(Bs + buf * BN * 64 + lds_off(brow, slab * 4 + fchunk));
                bf[j] = __builtin_bit_cast(frag, v);
            }
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) acc[i][j] = Mfma<T>::run(bf[j], af[i], acc[i][j]);
        }
    };

    stamp(p, 0);
    if constexpr (STAGES == 2) {
        dma_stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp(p, 1);
        for (int kb = kb0, it = 0; kb < kb1; ++kb, ++it) {
            if (kb + 1 < kb1) dma_stage((it + 1) & 1);            // in flight during the MFMAs
            compute(it & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's DMA has landed ...
            __syncthreads();                                      // ... before anyone reads the tile
        }
    } else {
        // Deep ring: stages it+1 .. it+STAGES-1 are in flight while stage it is multiplied.  Every wave issues exactly L = MREP + NREP DMA
        // instructions per stage and loads retire in order among themselves, so "at most (STAGES - 2) L outstanding" means stage it+1
        // has landed whatever the younger ones are doing; the last STAGES - 2 steps (nothing left to request) drain fully.
        constexpr int L = MREP + NREP, KEEP = (STAGES - 2) * L;
        const int n_it = kb1 - kb0;
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < n_it) dma_stage(s);
        if (n_it >= STAGES - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp(p, 1);
        int slot = 0, fill = STAGES - 1;                          // slot of stage it; slot stage it+STAGES-1 goes to (= the one step it-1 read)
        for (int it = 0; it < n_it; ++it) {
            const bool more = it + STAGES - 1 < n_it;
            if (more) dma_stage(fill);
            compute(slot);
            if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                      // stage it+1 complete on every wave; slot of stage it free
            slot = slot == STAGES - 1 ? 0 : slot + 1;
            fill = fill == STAGES - 1 ? 0 : fill + 1;
        }
    }
    stamp(p, 2);

    if (p.splits > 1) {          // fp32 slab straight from the fragments (64-byte row segments)
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            const int m = m0 + wm * 16 * MREP + i * 16 + (lane & 15);
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NREP; ++j) {
                const int n4 = n0 + wn * 16 * NREP + j * 16 + 4 * (lane >> 4);
                if (n4 >= p.N) continue;
                slab_store(p, ((static_cast<long>(blockIdx.y) * p.batch + bz) * (p.M - p.m_begin) + (m - p.m_begin)) * p.N + n4, acc[i][j]);
            }
        }
        if (p.tickets) splitk_finish<T, BM, BN, 256>(p, bz, tid_lin, m0, n0, reinterpret_cast<int*>(smem), t);
        return;
    }
    epilogue_tile<T, MREP, NREP, 256, BM, BN, 1, STATS>(p, bz, acc, smem, m0, n0, wm * 16 * MREP, wn * 16 * NREP, lane, t);
    stamp(p, 3);
}

// ---- 8-wave, 3-stage ring variant (the large layers) ---------------------------------------------
// 512 threads = 8 wavefronts (4 x 2), block tile 256 x (32*NREP) in {256x160, 256x128}, one block per
// CU (2 waves per SIMD), persistent: a block walks tiles blockIdx.x, + gridDim.x, ...
// K loop: fragment registers are double buffered and the single raw s_barrier of a K step sits between its
// two 32-k halves.  The last fragments of stage `it` are requested in the first half, so its slot is retired by
// that barrier and stage it+3 is requested into it right behind the barrier: TWO K steps of DMA look-ahead with
// three slots.  Stage it+1 is awaited with a COUNTED s_waitcnt vmcnt(L) (L = DMA instructions of one stage) so
// that stage it+2 stays in flight -- never a full drain inside a tile.  Every wave issues exactly
// L = 4 + ceil(BN/64) DMA instructions per stage (the 32-row remainder pass of the 160-row weight tile is issued
// by the lower 32 lanes of all 8 waves), so one immediate serves all.
// Tile boundary: right after the barrier that retires the ring the block decodes its next tile and requests
// that tile's first two stages into slots 0 / 1; the epilogue runs meanwhile in two row slices through slot 2.
template <typename T, int NREP, int NW, bool STATS = false, bool S3 = false, int BM_ = 256>
__global__ __launch_bounds__(64 * NW, BM_ == 128 ? 2 : 1) void k_conv_gemm8(const GemmParams p) {
    // NW = 8: 4 x 2 waves, 64x80 per wave, two waves per SIMD.  NW = 4, BM_ = 256: 2 x 2 waves, 128x80 per wave, ONE wave
    // per SIMD with the whole register file (160 accumulator + 104 fragment registers): twice the MFMAs per
    // LDS read / DMA piece / barrier, and no second wave competing for the issue port.
    // NW = 4, BM_ = 128 (round 5, "two tiles in flight"): 2 x 2 waves with the SAME 64x80 wave tile and K-step schedule, block tile
    // 128 x (32*NREP), a TWO-slot ring (74 KB) so that TWO blocks share a CU: one block's ramp / epilogue overlaps the other's K loop
    // (the 8-wave block idles its CU's matrix pipes for 10-50 % of every tile there).  One K step of DMA look-ahead instead of two --
    // a block that waits for its stage leaves the SIMDs to its neighbour.
    constexpr int NT = 64 * NW, WMG = NW / 2, BM = BM_, BN = 32 * NREP, MREP = BM / (16 * WMG), STAGES = BM_ == 128 ? 2 : 3;
    constexpr bool ONEWAVE = NW == 4 && BM_ == 256;
    static_assert(BM_ == 256 || (BM_ == 128 && NW == 4), "tile shapes: 256 rows (8 or 4 waves) or 128 rows (4 waves, two blocks per CU)");
    constexpr int RPP = NT / 8;                      // tile rows staged by one DMA pass of the block
    constexpr int STAGE = (BM + BN) * 64;            // 16-bit elements per ring slot
    constexpr int BFULL = BN / RPP;                  // full DMA passes of the weight tile
    constexpr bool BHALF = (BN % RPP) != 0;          // + one half pass (BN = 160 with 64-row passes)
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];

    const int ntile_total = p.mtiles * p.ntiles;
    const long bz = blockIdx.z;
    const unsigned short* a0 = p.a0 + bz * p.a_bs;
    const unsigned short* a1 = p.a1 ? p.a1 + bz * p.a_bs : nullptr;
    const unsigned short* wg = p.w + bz * p.w_bs;
    int m0 = 0, n0 = 0;                              // origin of the tile being multiplied (set_tile)
    const int pad_y = p.subpix ? 1 - (static_cast<int>(bz) >> 1) : p.pad, pad_x = p.subpix ? 1 - (static_cast<int>(bz) & 1) : p.pad;   // (sub-pixel phase: pf_gemm_params.h)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int chunk = t & 7, lrow = t >> 3;                       // DMA passes of RPP rows: row = pass*RPP + lrow
    const int lchunk8 = (chunk ^ ((lrow >> 1) & 7)) * 8;
    const int hrow = wave * 4 + (lane >> 3);                      // half pass (lanes 0..31): row = BFULL*64 + hrow
    const int hchunk8 = (chunk ^ ((hrow >> 1) & 7)) * 8;

    constexpr int APASS = BM / RPP;                  // DMA passes of the activation tile
    int a_img[APASS], a_y[APASS], a_x[APASS];
    // DMA addressing: buffer descriptors (SGPR) + a per-thread 32-bit byte offset (VGPR) + a per-stage
    // scalar byte offset (soffset), so that a piece costs no vector ALU work per K step -- the kernel is
    // bound by instruction issue, not by the matrix pipe (4 issue slots of a SIMD per 16-clock MFMA,
    // shared by two waves).  Zero padding / ragged tiles: an offset beyond num_records reads as zero.
    constexpr unsigned OOB = 0x80000000u;                         // launcher guarantees tensors < 2 GiB
    // descriptors are built from readfirstlane'd words: hipcc otherwise treats a selected descriptor as
    // divergent and wraps every DMA in a waterfall loop (cdna_hip_programming.md T20)
    auto uniform_ptr = [](const unsigned short* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
        const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
        return reinterpret_cast<unsigned short*>(static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi) << 32));
    };
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wg), 0, __builtin_amdgcn_readfirstlane(p.w_bytes), 0x00020000);
    // (the activation descriptor is rebuilt from scalars at the start of every stage: carried across
    // the loop as a variable it would live in VGPRs again)
    unsigned w_off[BFULL + 1];                                    // bytes
#ifdef PF_GEMM_BDIRECT        /* experiment: weight fragments straight from L1 / L2 into registers, no weight tile in LDS.
                               * Correct (68 GEMM / conv tests) and 2x SLOWER (3x3 convs 850 -> 440 TF/s): a fragment-shaped load touches 16
                               * half-used cache lines per instruction, four waves fetch the same lines, and the prefetch distance is
                               * half a K step.  Kept as a compile-time record of the measurement. */
    unsigned wb_off[NREP];                                        // bytes: (n of fragment j, this lane's k chunk)
    int kw = 0;                                                   // K offset (elements) of the stage being multiplied
#endif
    const int Ctot = p.c0 + p.c1;
    const int Hl = p.h_in << p.up, Wl = (p.w_in + 2 * p.wrap) << p.up;

    // K walk (wave-uniform): K-block = (tap, source, 64-channel block).  Per-thread activation offsets
    // change only when the tap or the source changes ("segment"); inside a segment only soffset moves.
    const int nkb = p.K / 64;
    const int kb0 = blockIdx.y * p.kb_per_split;
    const int kb1 = min(nkb, kb0 + p.kb_per_split);
    const int n_it = kb1 - kb0;
    int kg = 0, tap = 0, cc = 0;
    unsigned a_off[APASS];                                        // bytes, or OOB
    bool seg1 = false;                                            // current source is a1
    auto set_segment = [&]() __attribute__((always_inline)) {
        const int ky = p.ksize == 3 ? tap / 3 : p.ksize == 2 ? tap >> 1 : 0, kx = p.ksize == 3 ? tap - 3 * ky : p.ksize == 2 ? tap & 1 : 0;
        seg1 = __builtin_amdgcn_readfirstlane(cc >= p.c0 ? 1 : 0) != 0;
        const int ld = seg1 ? p.a1_ld : p.a0_ld;
        if (p.fastseg) {                                               // (pf_gemm_params.h: a_img = pixel index, a_y = validity bits)
            const unsigned sel = (1u << ky) | (16u << kx);
            const int tap_pix = ky * p.w_in + kx;
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const unsigned off = ((static_cast<unsigned>(a_img[i]) + static_cast<unsigned>(tap_pix)) * static_cast<unsigned>(ld) + static_cast<unsigned>(lchunk8)) * 2u;
                a_off[i] = (static_cast<unsigned>(a_y[i]) & sel) == sel ? off : OOB;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int yi = a_y[i] + ky, xi = a_x[i] + kx;             // xi: column in the (virtually wrap-padded, upsampled) input
            const bool ok = yi >= 0 && yi < Hl && xi >= 0 && xi < Wl;
            int sx = (xi >> p.up) - p.wrap;                            // source column: the padding is circular
            sx += sx < 0 ? p.w_in : 0;
            sx -= sx >= p.w_in ? p.w_in : 0;
            const int pix = (a_img[i] * p.h_in + (yi >> p.up)) * p.w_in + sx;
            a_off[i] = ok ? static_cast<unsigned>(pix * ld + lchunk8) * 2u : OOB;
        }
    };
    // Persistent over tiles: a block walks tiles blockIdx.x, + gridDim.x, ...; all per-tile state is set here.
    // (XCD-aware, bijective remap of the tile id: block b and all its tiles live on XCD b % 8.)
    auto set_tile = [&](int tile) __attribute__((always_inline)) {
        int tid_lin = tile;
        {
            const int q = ntile_total >> 3, r = ntile_total & 7;
            const int xcd = tid_lin & 7, idx = tid_lin >> 3;
            tid_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        const int tile_m = static_cast<unsigned>(tid_lin) / static_cast<unsigned>(p.ntiles), tile_n = tid_lin - tile_m * p.ntiles;
        m0 = p.m_begin + tile_m * BM;
        n0 = tile_n * BN;
        {
            // output row -> (image, y, x): two divisions for the first DMA pass, then the host-computed
            // advance of RPP rows with one carry per level (this runs once per tile on every thread)
            const unsigned m = static_cast<unsigned>(m0 + lrow);
            int img = static_cast<int>(m / static_cast<unsigned>(p.rows_per_img));
            const unsigned rem = m - static_cast<unsigned>(img) * static_cast<unsigned>(p.rows_per_img);
            int yo = static_cast<int>(rem / static_cast<unsigned>(p.w_out));
            int xo = static_cast<int>(rem) - yo * p.w_out;
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const bool ok = m0 + i * RPP + lrow < p.M;
                a_img[i] = ok ? img : 0;
                a_y[i] = ok ? yo * p.stride - pad_y : -(1 << 20);
                a_x[i] = (xo + p.crop) * p.stride - pad_x;
                if (p.fastseg) seg_pack(a_img[i], a_y[i], a_x[i], p.h_in, p.w_in);
                xo += p.adv_x;
                if (xo >= p.w_out) { xo -= p.w_out; ++yo; }
                yo += p.adv_y;
                if (yo >= p.h_out) { yo -= p.h_out; ++img; }
                img += p.adv_img;
            }
        }
#pragma unroll
        for (int j = 0; j < BFULL; ++j) {
            const int n = n0 + j * RPP + lrow;
            w_off[j] = n < p.N ? static_cast<unsigned>(n * p.K + lchunk8) * 2u : OOB;
        }
        {
            const int n = n0 + BFULL * RPP + hrow;
            w_off[BFULL] = (BHALF && n < p.N) ? static_cast<unsigned>(n * p.K + hchunk8) * 2u : OOB;
        }
        kg = kb0 * 64;
        tap = kg / Ctot;
        cc = kg - tap * Ctot;
        set_segment();
#ifdef PF_GEMM_BDIRECT
        kw = kb0 * 64;
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            const int n = n0 + (wave & 1) * 16 * NREP + j * 16 + (lane & 15);
            wb_off[j] = n < p.N ? static_cast<unsigned>(n * p.K + (lane >> 4) * 8) * 2u : OOB;
        }
#endif
    };

    // One stage = NPIECE DMA instructions per wave: pieces 0..3 the activation passes, then the weight
    // passes, issued one at a time between MFMAs (a burst of 8 waves x 7 KB stalls every wave at issue
    // for ~900 clocks: the CU's vector-memory pipe moves ~64 B/clk).
    constexpr int NPIECE = APASS + BFULL + (BHALF ? 1 : 0);
    int st_soff_a = 0, st_soff_w = 0;
    auto stage_begin = [&]() __attribute__((always_inline)) {
        st_soff_a = __builtin_amdgcn_readfirstlane((seg1 ? cc - p.c0 : cc) * 2);
        st_soff_w = __builtin_amdgcn_readfirstlane(kg * 2);
        return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(seg1 ? a1 : a0), 0,
                                                 __builtin_amdgcn_readfirstlane(seg1 ? p.a1_bytes : p.a0_bytes), 0x00020000);
    };
    auto stage_end = [&]() __attribute__((always_inline)) {
        kg += 64;
        cc += 64;
        if (cc == Ctot) { cc = 0; ++tap; set_segment(); }
        else if (cc == p.c0) set_segment();
    };
    auto lds_dma = [&](const __amdgpu_buffer_rsrc_t& r, unsigned short* dst, unsigned voff, int soff) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
    };
    auto dma_piece = [&](const __amdgpu_buffer_rsrc_t& rs_a, int slot, int k) __attribute__((always_inline)) {
        unsigned short* As = smem + slot * STAGE;
        unsigned short* Bs = As + BM * 64;
        if (k < APASS) {
            unsigned short* dst = As + (k * RPP + wave * 8) * 64;
            lds_dma(rs_a, dst, a_off[k], st_soff_a);
#ifdef PF_GEMM_BDIRECT
        } else if (true) {
#endif
        } else if (k < APASS + BFULL) {
            const int j = k - APASS;
            lds_dma(rs_w, Bs + (j * RPP + wave * 8) * 64, w_off[j], st_soff_w);
        } else if (BHALF) {
            if (lane < 32) lds_dma(rs_w, Bs + (BFULL * RPP + wave * 4) * 64, w_off[BFULL], st_soff_w);
        }
    };
    auto dma_stage = [&](int slot) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs_a = stage_begin();
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) dma_piece(rs_a, slot, k);
        stage_end();
    };

    f32x4 acc[MREP][NREP];

    typedef typename Mfma<T>::frag frag;
    const int frow = lane & 15, fchunk = lane >> 4;
    // Fragment registers are double buffered (f0 = first 32 k of a stage, f1 = second) and the barrier
    // sits BETWEEN the two halves of a step, so the matrix pipe always has 20 MFMAs queued while LDS
    // reads are in flight (measured: with both halves' reads exposed the K step took 2300 clocks
    // without any DMA, against 1280 of pure MFMA issue):
    //   f1 <- LDS(stage it, k 32..63) | MFMA(f0) | wait DMA(it+1), barrier | f0 <- LDS(stage it+1, k 0..31)
    //   | MFMA(f1) interleaved with the DMA pieces of stage it+2
    frag fa0[MREP], fb0[NREP], fa1[MREP], fb1[NREP];
#ifdef PF_ABL_DUMMY_B
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
        u16x8 z = {1, 2, 3, 4, 5, 6, 7, static_cast<unsigned short>(lane)};
        asm volatile("" : "+v"(z));
        fb0[j] = __builtin_bit_cast(frag, z);
        fb1[j] = __builtin_bit_cast(frag, z);
    }
#endif
    auto load_frags_a = [&](int slot, int slab, frag (&fa)[MREP]) __attribute__((always_inline)) {
        const unsigned short* As = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < MREP; ++i)
            fa[i] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(As + lds_off(wm * 16 * MREP + i * 16 + frow, slab * 4 + fchunk)));
    };
    auto load_frags_b = [&](int slot, int slab, frag (&fb)[NREP], int kadd) __attribute__((always_inline)) {
#ifdef PF_GEMM_BDIRECT
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        const int soff = __builtin_amdgcn_readfirstlane((kw + kadd + slab * 32) * 2);
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wb_off[j], soff, 0);
            fb[j] = __builtin_bit_cast(frag, v);
        }
        return;
#endif
        const unsigned short* Bs = smem + slot * STAGE + BM * 64;
#ifdef PF_ABL_DUMMY_B         /* timing-only: the weight-fragment reads are issued and waited for, but the MFMAs keep their first operands */
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            u16x8 v = *reinterpret_cast<const u16x8*>(Bs + lds_off(wn * 16 * NREP + j * 16 + frow, slab * 4 + fchunk));
            asm volatile("" :: "v"(v));
        }
        (void)fb;
#else
#pragma unroll
        for (int j = 0; j < NREP; ++j)
            fb[j] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(Bs + lds_off(wn * 16 * NREP + j * 16 + frow, slab * 4 + fchunk)));
#endif
    };
    auto load_frags = [&](int slot, int slab, frag (&fa)[MREP], frag (&fb)[NREP], int kadd) __attribute__((always_inline)) {   // kadd: 64 for the NEXT stage (PF_GEMM_BDIRECT)
        load_frags_a(slot, slab, fa);
#ifndef PF_ABL_NOLDS_B        /* timing-only: what the weight fragments' share of the LDS reads costs */
        load_frags_b(slot, slab, fb, kadd);
#endif
    };

    auto issue_prologue = [&]() __attribute__((always_inline)) {
        dma_stage(0);
        if (n_it > 1) dma_stage(1);
    };
    const std::true_type YES;
    const std::false_type NO;
    int cur = 0, nx1 = 1, nx2 = 2;
#ifdef PF_GEMM_TIMELINE
    int tl_tile = 0, tl_step = 0;
#endif
    // One K step.  MORE: a next stage exists (wait for it, barrier, prefetch its first fragments);
    // DMA: a stage three ahead exists (issue its pieces between the MFMAs of the second half).  The flags
    // are compile-time so that the steady-state body is branch free and the compiler's s_waitcnt
    // insertion sees exact counts (a conditional around the prefetch made it drain lgkmcnt to 0 right
    // after issuing it, exposing the LDS latency once per step).
    // (-DPF_ABL_NOBARRIER / NODMA / NOLDS: timing-only ablations of this loop -- wrong results -- for tools/gpu_gemm_ablate.sh)
    auto step_l = [&](auto more_tag, auto dma_tag, auto wait_tag, auto flead_tag) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_tag)::value, DMA = decltype(dma_tag)::value;
        constexpr int WAIT = decltype(wait_tag)::value;           // DMA instructions that may stay in flight at the mid-step wait
        constexpr int NM = MREP * NREP, LEAD0 = NM - 1 - 2 * (NPIECE - 1), LEAD = LEAD0 < 4 ? LEAD0 : 4, FLEAD = decltype(flead_tag)::value;   // fragment requests go out after FLEAD MFMAs:
        // at every s_waitcnt lgkmcnt the only outstanding LDS reads are then the ones being waited for
        // (the compiler drains to 0, it does not count), and they were issued >= NM - LEAD MFMAs earlier.
#pragma unroll
        for (int idx = 0; idx < NM; ++idx) {
            // (S3: weight-fragment-major order -- the fragment fb0[j] of THIS step was requested j-th during the previous step's last
            // group, so the later ones have time to arrive)
            const int fi = S3 ? idx % MREP : idx / NREP, fj = S3 ? idx / MREP : idx % NREP;
            if constexpr (ONEWAVE) Mfma<T>::acc_agpr(fb0[fj], fa0[fi], acc[fi][fj]);
            else acc[fi][fj] = Mfma<T>::run(fb0[fj], fa0[fi], acc[fi][fj]);
#ifdef PF_GEMM_FLEAD2          /* two bursts: A fragments after FLEAD MFMAs, B fragments after FLEAD2 (A/B build) */
            if (idx == FLEAD - 1) { __builtin_amdgcn_sched_barrier(0); load_frags_a(cur, 1, fa1); __builtin_amdgcn_sched_barrier(0); }
            if (idx == PF_GEMM_FLEAD2 - 1) { __builtin_amdgcn_sched_barrier(0); load_frags_b(cur, 1, fb1, 0); __builtin_amdgcn_sched_barrier(0); }
#else
            if (idx == FLEAD - 1) {
                __builtin_amdgcn_sched_barrier(0);
#ifndef PF_ABL_NOLDS
                load_frags(cur, 1, fa1, fb1, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
        // (S3: hipcc sank 15 of the first group's 20 MFMAs below the wait -- register-only instructions move across an inline-asm
        // s_waitcnt -- so the wait found the nine f1 reads it had just issued: pinned)
        if constexpr (S3) __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) {
            // my pieces of the next stage have landed (the stage after it may still be in flight: WAIT), and my
            // reads of the current slot have returned -- it is refilled right after the barrier
#ifdef PF_GEMM_BDIRECT
            if constexpr (true) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
            if constexpr (WAIT == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
            else if constexpr (WAIT == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else if constexpr (WAIT == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
            else if constexpr (WAIT == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if constexpr (WAIT == 9) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
            else if constexpr (WAIT == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
            else if constexpr (WAIT == 13) asm volatile("s_waitcnt vmcnt(13) lgkmcnt(0)" ::: "memory");
            else static_assert(WAIT == 0, "add the immediate");
#ifndef PF_ABL_NOBARRIER
            __builtin_amdgcn_s_barrier();                         // next stage complete; the current slot no longer read
#endif
            asm volatile("" ::: "memory");
        }
        __amdgpu_buffer_rsrc_t rs_a = rs_w;
        if constexpr (DMA) rs_a = stage_begin();
        if constexpr (S3) {
            // Split-precision step: the stage holds [hi(32) | lo(32)] of 32 channels for both operands, f0 = (W_hi, A_hi),
            // f1 = (W_lo, A_lo).  The first half multiplied W_hi A_hi; here W_lo A_hi (group X, fa0's last use -> the next stage's fa0
            // is requested behind it) and W_hi A_lo (group Y, weight-fragment-major: fb0[j] is re-requested as soon as its four
            // MFMAs are issued).  W_lo A_lo (2^-22) is not formed.  60 MFMAs per 52 KB stage instead of 40: the K' = 3 K walk of
            // rounds 2-3 moved [hi] twice and ran 3 stages per 64 channels, this one 2.
#pragma unroll
            for (int idx = 0; idx < NM; ++idx) {
                acc[idx / NREP][idx % NREP] = Mfma<T>::run(fb1[idx % NREP], fa0[idx / NREP], acc[idx / NREP][idx % NREP]);
                if constexpr (DMA) {
                    const int k = idx - LEAD;
                    if (k >= 0 && (k & 1) == 0 && (k >> 1) < NPIECE) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma_piece(rs_a, cur, k >> 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (MORE) { __builtin_amdgcn_sched_barrier(0); load_frags_a(nx1, 0, fa0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int j = 0; j < NREP; ++j) {
#pragma unroll
                for (int i = 0; i < MREP; ++i) acc[i][j] = Mfma<T>::run(fb0[j], fa1[i], acc[i][j]);
                if (MORE) {
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned short* Bn = smem + nx1 * STAGE + BM * 64;
                    fb0[j] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(Bn + lds_off(wn * 16 * NREP + j * 16 + frow, fchunk)));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
        for (int idx = 0; idx < NM; ++idx) {
            if constexpr (ONEWAVE) Mfma<T>::acc_agpr(fb1[idx % NREP], fa1[idx / NREP], acc[idx / NREP][idx % NREP]);
            else acc[idx / NREP][idx % NREP] = Mfma<T>::run(fb1[idx % NREP], fa1[idx / NREP], acc[idx / NREP][idx % NREP]);
#ifdef PF_GEMM_FLEAD2
            if (MORE && idx == FLEAD - 1) { __builtin_amdgcn_sched_barrier(0); load_frags_a(nx1, 0, fa0); __builtin_amdgcn_sched_barrier(0); }
            if (MORE && idx == PF_GEMM_FLEAD2 - 1) { __builtin_amdgcn_sched_barrier(0); load_frags_b(nx1, 0, fb0, 64); __builtin_amdgcn_sched_barrier(0); }
#else
            if (MORE && idx == FLEAD - 1) {
                __builtin_amdgcn_sched_barrier(0);
#ifndef PF_ABL_NOLDS
                load_frags(nx1, 0, fa0, fb0, 64);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            if constexpr (DMA) {
                const int k = idx - LEAD;                         // pieces after MFMA LEAD, LEAD+2, ...
                if (k >= 0 && (k & 1) == 0 && (k >> 1) < NPIECE) {
                    __builtin_amdgcn_sched_barrier(0);
#ifndef PF_ABL_NODMA
#ifdef PF_ABL_NODMA_B         /* timing-only: activation pieces only */
                    if ((k >> 1) < APASS)
#endif
                    dma_piece(rs_a, cur, k >> 1);                 // stage it+3 into the slot retired at this step's barrier
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        }
        static_assert(LEAD >= 0 && LEAD + 2 * (NPIECE - 1) < NM, "DMA pieces must fit behind the MFMAs of the second half");
        if constexpr (DMA) stage_end();
        cur = cur == STAGES - 1 ? 0 : cur + 1;
        nx1 = nx1 == STAGES - 1 ? 0 : nx1 + 1;
        nx2 = nx2 == STAGES - 1 ? 0 : nx2 + 1;
#ifdef PF_GEMM_BDIRECT
        kw += 64;
#endif
#ifdef PF_GEMM_TIMELINE
        PF_TL(p, tl_tile == 1 && tl_step < 8, 15 + tl_step);
        ++tl_step;
#endif
    };
#ifndef PF_GEMM_FLEAD
#define PF_GEMM_FLEAD 4
#endif
    auto step = [&](auto more_tag, auto dma_tag, auto wait_tag) __attribute__((always_inline)) { step_l(more_tag, dma_tag, wait_tag, std::integral_constant<int, PF_GEMM_FLEAD>()); };

    stamp(p, 0);
#ifdef PF_GEMM_SETPRIO
    // static priority for the later-dispatched half of the block (MI355X_MICROARCH.md, "two waves per SIMD", item 4):
    // measured neutral here (-DPF_GEMM_SETPRIO=1 / 3, same-box A/B round 2: 3x3 convs 915 -> 891 / 912 TF/s, step 14.85 vs 14.76):
    // both waves of a SIMD run the same MFMA-dense stream, there is no loader / computer pairing for a priority to help
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(PF_GEMM_SETPRIO);
#endif
    int tile = blockIdx.x;
    set_tile(tile);
    issue_prologue();
    bool first = true;
    for (;;) {
        PF_TL(p, tl_tile == 1, 12);
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (first && n_it > 1) {                          // stage 0 landed, stage 1 may still fly
#ifdef PF_GEMM_BDIRECT
            if constexpr (true) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            if constexpr (NPIECE == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
#endif
            else if constexpr (NPIECE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (NPIECE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NPIECE == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if constexpr (NPIECE == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
            else if constexpr (NPIECE == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {                                          // later tiles: the epilogue's stores sit behind the DMAs in vmcnt
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (first) stamp(p, 1);
        first = false;
        PF_TL(p, tl_tile == 1, 13);
        if constexpr (STAGES == 3) { if (n_it > 2) dma_stage(2); }   // third stage in flight (slot 2 staged the previous tile's epilogue)
        load_frags(0, 0, fa0, fb0, 0);
        cur = 0; nx1 = 1; nx2 = 2 % STAGES;
        PF_TL(p, tl_tile == 1, 14);
        // The slot of stage `it` is retired at the mid-step barrier of step `it` (its last fragments are
        // requested in the first half), and stage it+3 is requested into it right behind that barrier: with
        // three slots a stage has TWO K steps to arrive, the mid-step wait leaves the younger one in flight.
        const std::integral_constant<int, NPIECE> W1;
        const std::integral_constant<int, 0> W0;
        if constexpr (STAGES == 3) {
            for (int it = 0; it + 3 < n_it; ++it) step(YES, YES, W1);
            if (n_it >= 3) step(YES, NO, W1);
            if (n_it >= 2) step(YES, NO, W0);
            step(NO, NO, W0);
        } else {
            // two slots: stage it+2 goes into the slot of stage it right behind the mid-step barrier of step it and is awaited
            // (full drain: nothing younger is in flight) at the mid-step wait of step it+1
            for (int it = 0; it + 2 < n_it; ++it) step(YES, YES, W0);
            if (n_it >= 2) step(YES, NO, W0);
            step(NO, NO, W0);
        }
        if constexpr (ONEWAVE) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // inline-asm MFMA results -> vector ALU reads
        stamp(p, 2);

        const int em0 = m0, en0 = n0;
        const int next = tile + static_cast<int>(gridDim.x);
        const bool has_next = next < ntile_total;
        if (p.splits > 1) {          // fp32 slab straight from the fragments (64-byte row segments)
#pragma unroll
            for (int i = 0; i < MREP; ++i) {
                const int m = em0 + wm * 16 * MREP + i * 16 + (lane & 15);
                if (m >= p.M) continue;
#pragma unroll
                for (int j = 0; j < NREP; ++j) {
                    const int n4 = en0 + wn * 16 * NREP + j * 16 + 4 * (lane >> 4);
                    if (n4 >= p.N) continue;
                    slab_store(p, ((static_cast<long>(blockIdx.y) * p.batch + bz) * (p.M - p.m_begin) + (m - p.m_begin)) * p.N + n4, acc[i][j]);
                }
            }
            if (p.tickets) splitk_finish<T, BM, BN, NT>(p, bz, tile, em0, en0, reinterpret_cast<int*>(smem + (STAGES - 1) * STAGE), t);
            if (!has_next) break;
            __syncthreads();                              // every wave is done reading the operand ring
            set_tile(next);
            issue_prologue();
        } else {
            // epilogue in two 128-row halves through ring slot 2; the next tile's first two stages are
            // requested into slots 0 / 1 right after the barrier that frees the ring, so their latency
            // hides behind the epilogue
            // (laundered lane / thread ids: the epilogue's per-thread addressing would otherwise be hoisted out
            // of the tile loop and stay live -- in registers the K loop has none to spare of)
            int e_lane = lane, e_t = t;
            asm volatile("" : "+v"(e_lane), "+v"(e_t));
            if constexpr (STAGES == 3) {
                epilogue_tile<T, MREP, NREP, NT, BM, BN, 2, STATS>(p, bz, acc, smem + 2 * STAGE, em0, en0, wm * 16 * MREP, wn * 16 * NREP,
                                                            e_lane, e_t, [&]() __attribute__((always_inline)) { if (has_next) { set_tile(next); issue_prologue(); } });
            } else {
                // two slots: the whole 128-row tile is staged through the (dead) ring at once; the next tile's first stages are
                // requested only when every wave has read its rows back out -- the neighbour block owns the matrix pipes meanwhile
                epilogue_tile<T, MREP, NREP, NT, BM, BN, 1, STATS>(p, bz, acc, smem, em0, en0, wm * 16 * MREP, wn * 16 * NREP, e_lane, e_t);
                if (has_next) {
                    __syncthreads();
                    set_tile(next);
                    issue_prologue();
                }
            }
            PF_TL(p, tl_tile == 1, 23);
            if (!has_next) break;
        }
        tile = next;
#ifdef PF_GEMM_TIMELINE
        ++tl_tile; tl_step = 0;
#endif
    }
    stamp(p, 3);
}

// Split-K second pass: sum the fp32 slabs in split order (deterministic) and run the epilogue.
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_reduce(const GemmParams p) {
    const long quads = static_cast<long>(p.N / 4);
    const int Ms = p.M - p.m_begin;
    const long total = static_cast<long>(p.batch) * Ms * quads;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total) return;
    const int n4 = static_cast<int>(i % quads) * 4;
    const long bm = i / quads;
    const int ml = static_cast<int>(bm % Ms), m = p.m_begin + ml;
    const long bz = bm / Ms;
    const long slab = static_cast<long>(p.batch) * Ms * p.N;
    const float* src = p.partial + (bz * Ms + ml) * p.N + n4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.splits; ++s) {
        const float4 x = *reinterpret_cast<const float4*>(src + s * slab);
        v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
    }
    epilogue_store<T>(p, bz, m, n4, v);
}

// diagnostics buffer (device) + its capacity, see pf_debug_gemm_profile: published / read as ONE atomic
// snapshot so that concurrent launches from other host threads see either the old or the new pair
struct ProfState { unsigned long long* buf; long blocks; };
static std::atomic<const ProfState*> g_prof_state{nullptr};
static inline ProfState prof_snapshot() {
    const ProfState* s = g_prof_state.load(std::memory_order_acquire);
    return s ? *s : ProfState{nullptr, 0};
}


static int tuning(const char* name, int dflt);
template <typename T, int MREP, int NREP, bool STATS, bool S3, int STAGES>
static pf_status launch_ring(const GemmParams& p, int batch, hipStream_t st) {
    constexpr int BM = 32 * MREP, BN = 32 * NREP;
    const size_t smem = static_cast<size_t>(STAGES) * (BM + BN) * 64 * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_gemm<T, MREP, NREP, STATS, S3, STAGES>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_conv_gemm<T, MREP, NREP, STATS, S3, STAGES>), dim3(p.mtiles * p.ntiles, p.splits, batch), dim3(256), smem, st, p);
    return PF_OK;
}

template <typename T, int MREP, int NREP, bool STATS = false, bool S3 = false>
static pf_status launch_s(const GemmParams& gp, int batch, hipStream_t st) {
    constexpr int BM = 32 * MREP, BN = 32 * NREP;
    GemmParams p = gp;
    p.mtiles = static_cast<int>(cdiv(p.M - p.m_begin, BM));
    p.ntiles = static_cast<int>(cdiv(p.N, BN));
    const ProfState ps = prof_snapshot();
    p.prof = (ps.buf && static_cast<long>(p.mtiles) * p.ntiles * p.splits * batch <= ps.blocks) ? ps.buf : nullptr;
    // a grid of at most one block per CU runs the four-slot ring (one block per CU: nothing to share the CU with anyway)
    static const int deep_max = tuning("PF_GEMM_DEEP_RING", 1) ? tuning("PF_GEMM_DEEP_RING_MAX_BLOCKS", 256) : 0;
    const long blocks = static_cast<long>(p.mtiles) * p.ntiles * p.splits * batch;
    // (instantiated for the plain kernels only: the moment / split-precision variants keep the two-slot form -- compile time)
    if constexpr (!STATS && !S3) {
        if (blocks <= deep_max && p.K / 64 / p.splits >= 3) launch_ring<T, MREP, NREP, false, false, 4>(p, batch, st);
        else launch_ring<T, MREP, NREP, false, false, 2>(p, batch, st);
    } else {
        launch_ring<T, MREP, NREP, STATS, S3, 2>(p, batch, st);
    }
    PF_CHECK_LAUNCH("pf_conv_gemm");
    if (p.splits > 1 && !p.tickets) {
        const long total = static_cast<long>(batch) * (p.M - p.m_begin) * (p.N / 4);
        hipLaunchKernelGGL((k_splitk_reduce<T>), dim3(cdiv(total, 256)), dim3(256), 0, st, p);
        PF_CHECK_LAUNCH("pf_conv_gemm (split-K reduce)");
    }
    return PF_OK;
}

template <typename T, int MREP, int NREP>
static pf_status launch(const GemmParams& gp, int batch, hipStream_t st) {
    if (gp.s3) return gp.gn_partial ? launch_s<T, MREP, NREP, true, true>(gp, batch, st) : launch_s<T, MREP, NREP, false, true>(gp, batch, st);
    return gp.gn_partial ? launch_s<T, MREP, NREP, true>(gp, batch, st) : launch_s<T, MREP, NREP, false>(gp, batch, st);
}

static int tuning(const char* name, int dflt);
template <typename T, int NREP, int NW, bool STATS = false, bool S3 = false, int BM = 256>
static pf_status launch8w_s(const GemmParams& gp, int batch, hipStream_t st) {
    constexpr int BN = 32 * NREP, STAGES = BM == 128 ? 2 : 3;
    GemmParams p = gp;
    p.mtiles = static_cast<int>(cdiv(p.M - p.m_begin, BM));
    p.ntiles = static_cast<int>(cdiv(p.N, BN));
    {
        const int rpp = 8 * NW, rem = rpp % p.rows_per_img;       // rows of one DMA pass, as (images, rows, columns)
        p.adv_img = rpp / p.rows_per_img;
        p.adv_y = rem / p.w_out;
        p.adv_x = rem % p.w_out;
    }
    const ProfState ps = prof_snapshot();
    p.prof = (ps.buf && static_cast<long>(p.mtiles) * p.ntiles * p.splits * batch <= ps.blocks) ? ps.buf : nullptr;
    const size_t smem = static_cast<size_t>(STAGES) * (BM + BN) * 64 * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_gemm8<T, NREP, NW, STATS, S3, BM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
        attr_set = true;
    }
    // persistent over tiles: one block per CU walks tiles b, b + grid, ... (a multiple of 8 keeps a tile on
    // the XCD its id maps to); PF_GEMM8_PERSIST=0 launches one block per tile
    static const int cap1 = tuning("PF_GEMM8_PERSIST", 256);
    const int cap = BM == 128 ? 2 * cap1 : cap1;                  // (two resident 128-row blocks per CU)
    int grid = p.mtiles * p.ntiles;
    if (cap > 0) {
        int gx = std::max(1, cap / (p.splits * batch));           // blockIdx.y / z multiply the resident blocks
        if (gx >= 8) gx = gx / 8 * 8;
        grid = std::min(grid, gx);
    }
    hipLaunchKernelGGL((k_conv_gemm8<T, NREP, NW, STATS, S3, BM>), dim3(grid, p.splits, batch), dim3(64 * NW), smem, st, p);
    PF_CHECK_LAUNCH("pf_conv_gemm (8-wave)");
    if (p.splits > 1 && !p.tickets) {
        const long total = static_cast<long>(batch) * (p.M - p.m_begin) * (p.N / 4);
        hipLaunchKernelGGL((k_splitk_reduce<T>), dim3(cdiv(total, 256)), dim3(256), 0, st, p);
        PF_CHECK_LAUNCH("pf_conv_gemm (split-K reduce)");
    }
    return PF_OK;
}

template <typename T, int NREP, int NW, int BM = 256>
static pf_status launch8w(const GemmParams& gp, int batch, hipStream_t st) {
    if constexpr (NW == 8 || BM == 128) {
        if (gp.s3) return gp.gn_partial ? launch8w_s<T, NREP, NW, true, true, BM>(gp, batch, st) : launch8w_s<T, NREP, NW, false, true, BM>(gp, batch, st);
        if (gp.gn_partial) return launch8w_s<T, NREP, NW, true, false, BM>(gp, batch, st);
    }
    return launch8w_s<T, NREP, NW, false, false, BM>(gp, batch, st);
}

static int tuning(const char* name, int dflt);
template <typename T, int NREP>
static pf_status launch8(const GemmParams& gp, int batch, hipStream_t st, int bm = 256) {
    if (bm == 128) return launch8w<T, NREP, 4, 128>(gp, batch, st);       // two 128-row blocks per CU (round 5)
    // NW = 4 (one 128x80 wave per SIMD, accumulators in AGPRs) is implemented and correct but measured slower
    // (K step 2520 vs 2222 clocks, epilogue 2x): a lone in-order wave exposes every lgkmcnt / vmcnt / barrier wait.
    // PF_GEMM8_WAVES=4 selects it (A/B: it moves 28 % fewer fragment bytes through the LDS).
    static const int nw = tuning("PF_GEMM8_WAVES", 8);
    if (nw == 4 && !gp.s3) return launch8w<T, NREP, 4>(gp, batch, st);       // (the A/B instantiation has no split-precision variant)
    return launch8w<T, NREP, 8>(gp, batch, st);
}

static int tuning(const char* name, int dflt) {     // A/B switches for benchmarking (read once)
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Tile shape + split-K plan of one problem (shared by the launcher and the workspace query).
struct GemmPlan { int mrep, nrep, splits, kb_per_split; bool big; int m_split; int tail_splits, tail_kb; int bm = 256; };
static GemmPlan plan_gemm_small(long M, int N, int K, int batch, bool allow_split, bool s3);
static GemmPlan plan_gemm0(long M, int N, int K, int batch, bool allow_split, bool s3);
// Which plans of the 8-wave kernel run as 128-row blocks, two per CU (k_conv_gemm8<..., BM_ = 128>): PF_GEMM_BM128 = 0 none,
// 1 all of them (A/B), 2 (default) the measured rule: the 128-row blocks win where the tile is ramp / epilogue-bound -- short K
// (K <= PF_GEMM_BM128_MAXK: isolated +7 % at K = 320, +15...22 % at K = 640, +12...16 % at K = 1280; long-K convolutions lose
// 3-5 % to the shorter DMA look-ahead and the doubled weight traffic, profiles/r5a_gemm_bm128.txt): same-box step A/B 62.06 -> 61.39 /
// 61.36 ms (profiles/r5b_ab_gemm_bm128.txt).  PF_GEMM_BM128_ONEROUND=1 also takes every problem that is ONE round of 256-row tiles
// (a CU then runs a single ramp + K loop + epilogue with nothing to overlap -- the per-rank GEMMs of the sharded layouts): measured
// neutral to slightly slower on the simulated ranks (8 ranks 14.35 -> 14.48 ms, 4 ranks 19.9 -> 20.4), so off.
static GemmPlan plan_gemm(long M, int N, int K, int batch, bool allow_split, bool s3 = false) {
    GemmPlan g = plan_gemm0(M, N, K, batch, allow_split, s3);
    static const int mode = tuning("PF_GEMM_BM128", 2);
    static const int max_k = tuning("PF_GEMM_BM128_MAXK", 1280);
    static const int one_round = tuning("PF_GEMM_BM128_ONEROUND", 0);
    if (!g.big || g.m_split > 0) return g;
    if (mode == 1) g.bm = 128;
    else if (mode == 2) {
        const long tiles256 = cdiv(M, 256) * cdiv(N, 32 * g.nrep) * batch * g.splits;
        if (K <= max_k || (one_round && tiles256 <= 256)) g.bm = 128;
    }
    return g;
}
static GemmPlan plan_gemm0(long M, int N, int K, int batch, bool allow_split, bool s3) {
    static const int big_min_tiles = tuning("PF_GEMM8_MIN_TILES", 128);   // 0 disables the 8-wave kernel
    const int nrep = (N % 160 == 0) ? 5 : 4;
    const long tiles256 = cdiv(M, 256) * cdiv(N, 32 * nrep) * batch;
    // one 8-wave block per CU: take it only when the tiles fill their rounds of 256 CUs to >= 60 % overall (320
    // tiles would idle for 37 % of the launch; the 4-wave kernel's 2 blocks per CU degrade more gracefully).
    // The threshold was flat at the single-GPU sizes in round 2 (17.4-17.5 steps/s from 50 to 95 %) and was set to 60 on the smaller
    // per-rank GEMMs of 2 / 4 / 8 ranks (tools/sim_rank.py: 37.1 -> 33.9 ms per step at 2 ranks from 88 to 60 %).  Round 6, last sweep
    // on the final kernels (profiles/r6n_ab_plan_knobs.txt): 30-50 read -0.2 ... -0.5 ms per step on three boxes, cfg 4 -0.9 ms, the
    // simulated ranks unchanged (half-filled rounds of the panorama's 64 x 128 level and of the 8 x 8 level now take the persistent kernel): 30.
    const long rounds = cdiv(tiles256, 256);
    static const int fill_pct = tuning("PF_GEMM8_FILL", 30);
    const bool filled = tiles256 * 100 >= rounds * 256 * fill_pct;
    // a badly filled last round of a LONG-K layer (320 tiles of a 16x16-level 3x3 conv = 1.25 rounds) is better spent on a
    // split-K tail launch than on a second full round: the tail split below goes first when the fill is under
    // PF_GEMM8_TAIL_FIRST % (same-box A/B on the mixed scheme: 69.7 -> 68.7 ms per step)
    static const int tail_first_pct = tuning("PF_GEMM8_TAIL_FIRST", 80);
    static const int big_min_k = tuning("PF_GEMM8_MIN_K", 0);      // A/B: least K for the 8-wave kernel (measured neutral)
    // Tail split: whole rounds of 256 tiles run unsplit; the tile rows left over (a badly filled last round)
    // become a second launch whose K range is split so that it fills the chip once more.  320 tiles then
    // cost 1.25 rounds instead of 2.
    static const int tail_on = tuning("PF_GEMM_TAIL_SPLIT", 1);
    GemmPlan tail;
    bool tail_ok = false;
    if (big_min_tiles > 0 && tail_on && allow_split && batch == 1 && N % 4 == 0 && tiles256 > 256 && K >= big_min_k && K / 64 >= tuning("PF_GEMM_TAIL_MINKB", 40)) {
        const long ntl = cdiv(N, 32 * nrep), mt = cdiv(M, 256);
        const long rows1 = (tiles256 / 256) * 256 / ntl;            // tile rows of the unsplit launch
        const long tiles2 = (mt - rows1) * ntl;
        long sp = tiles2 > 0 ? (256 + tiles2 / 2) / tiles2 : 1;
        if (sp > (K / 64) / 8) sp = (K / 64) / 8;
        if (rows1 > 0 && tiles2 > 0 && sp >= 2 && rows1 * ntl * 100 >= (tiles256 / 256) * 256 * 90) {
            tail.big = true; tail.mrep = 8; tail.nrep = nrep; tail.splits = 1; tail.kb_per_split = K / 64;
            tail.m_split = static_cast<int>(rows1 * 256);
            tail.tail_kb = static_cast<int>(cdiv(K / 64, sp));
            tail.tail_splits = static_cast<int>(cdiv(K / 64, tail.tail_kb));
            tail_ok = true;
        }
    }
    if (tail_ok && tiles256 * 100 < rounds * 256 * tail_first_pct) return tail;
    if (big_min_tiles > 0 && tiles256 >= big_min_tiles && filled && K >= big_min_k) {
        GemmPlan g;
        g.big = true; g.mrep = 8; g.nrep = nrep; g.splits = 1; g.kb_per_split = K / 64; g.m_split = 0;
        return g;
    }
    if (tail_ok) return tail;
    // long-K layers with few output tiles (the 8x8 level, the panorama's inner levels): 256-row tiles
    // re-read the weight panel 2-4x less often than the 64-row tiles of the small kernel; split K so
    // that one round of blocks covers the chip
    const int nkb_all = K / 64;
    if (big_min_tiles > 0 && allow_split && N % 4 == 0 && tiles256 >= 32 && tiles256 <= 128 && nkb_all >= 32) {
        long sp = 256 / tiles256;
        if (sp > nkb_all / 16) sp = nkb_all / 16;
        if (sp >= 2) {
            GemmPlan g;
            g.big = true; g.mrep = 8; g.nrep = nrep; g.m_split = 0;
            g.kb_per_split = static_cast<int>(cdiv(nkb_all, sp));
            g.splits = static_cast<int>(cdiv(nkb_all, g.kb_per_split));
            return g;
        }
    }
    return plan_gemm_small(M, N, K, batch, allow_split, s3);
}
static GemmPlan plan_gemm_small(long M, int N, int K, int batch, bool allow_split, bool s3) {
    GemmPlan g;
    g.big = false; g.m_split = 0;
    // 160-wide N tiles when they divide N exactly (all UNet widths are multiples of 160), 128-wide
    // otherwise; 64-row M tiles when 128-row tiles would not fill the 256 CUs (2 blocks per CU).
    g.nrep = (N % 160 == 0) ? 5 : 4;
    const long ntiles = cdiv(N, 32 * g.nrep);
    const long tiles128 = cdiv(M, 128) * ntiles * batch;
    g.mrep = tiles128 < 512 ? 2 : 4;
    const long tiles = cdiv(M, 32 * g.mrep) * ntiles * batch;
    const int nkb = K / 64;
    g.splits = 1;
    g.kb_per_split = nkb;
    // split-K when the grid cannot fill the chip and K is long: aim at ~640 blocks, >= 6 K-blocks each
    static const int split_min_kb = tuning("PF_GEMM_SPLIT_MINKB", 12);   // least K depth (64-blocks) for split-K on the 4-wave kernel
    // Round 6: a grid of <= 256 blocks runs the four-slot ring, one block per CU (launch_s): there the split aims at ONE round of the chip
    // (fewer fp32 slabs for the reduce kernel: 128 tiles x 2 K slices instead of x 5) and a K slice may be as short as 4 steps.
    static const int deep = tuning("PF_GEMM_DEEP_RING", 1);
    if (deep && !s3 && tiles <= 256) {                               // (the split-precision kernels keep the two-slot ring and its plan)
        long s = allow_split && N % 4 == 0 && nkb >= split_min_kb ? 256 / tiles : 1;
        if (s > nkb / 4) s = nkb / 4;
        if (s > 32) s = 32;
        if (s > 1) {
            g.kb_per_split = static_cast<int>(cdiv(nkb, s));
            g.splits = static_cast<int>(cdiv(nkb, g.kb_per_split));
        }
        return g;
    }
    if (allow_split && N % 4 == 0 && tiles <= 320 && nkb >= split_min_kb) {
        long s = (640 + tiles - 1) / tiles;
        if (s > nkb / 6) s = nkb / 6;
        if (s > 32) s = 32;
        if (s > 1) {
            g.kb_per_split = static_cast<int>(cdiv(nkb, s));
            g.splits = static_cast<int>(cdiv(nkb, g.kb_per_split));
        }
    }
    return g;
}

// descriptor -> kernel parameters (no validation here)
static void params_from_desc(const pf_conv_desc* d, GemmParams& p) {
    const int c1 = d->a1 ? d->c1 : 0;
    const int Ctot = d->c0 + c1;
    p.a0 = static_cast<const unsigned short*>(d->a0);
    p.a1 = static_cast<const unsigned short*>(d->a1);
    p.c0 = d->c0; p.c1 = c1; p.a0_ld = d->a0_ld; p.a1_ld = d->a1 ? d->a1_ld : 0;
    p.h_in = d->h_in; p.w_in = d->w_in; p.h_out = d->h_out; p.w_out = d->w_out;
    p.ksize = d->ksize; p.stride = d->stride; p.pad = d->pad; p.up = d->upsample;
    p.wrap = d->wrap_pad; p.crop = d->crop;
    p.w = static_cast<const unsigned short*>(d->w);
    p.rows_per_img = d->h_out * d->w_out;
    p.M = d->n_img * p.rows_per_img; p.N = d->n_out; p.K = d->ksize * d->ksize * Ctot;
    p.bias = d->bias; p.rowvec = d->rowvec; p.rowvec_ld = d->rowvec_ld;
    p.residual = d->residual; p.res_ld = d->res_ld;
    p.res_f32 = d->residual != nullptr && d->res_dtype == PF_F32;
    p.out = d->out; p.out_ld = d->out_ld; p.out_f32 = d->out_dtype == PF_F32;
    p.geglu = d->epilogue == PF_EPILOGUE_GEGLU;
    p.split_out = d->epilogue == PF_EPILOGUE_SPLIT;
    p.a_bs = d->a_bstride; p.w_bs = d->w_bstride; p.out_bs = d->out_bstride; p.res_bs = d->res_bstride;
    p.mtiles = p.ntiles = 0;
    p.prof = nullptr;
    p.s3 = d->split3 != 0;
    p.batch = d->batch;
    p.gn_partial = nullptr; p.gn_rows = 0;
    p.m_begin = 0; p.splits = 1; p.kb_per_split = 0; p.partial = nullptr; p.tickets = nullptr;
    p.a0_bytes = p.a1_bytes = p.w_bytes = 0; p.adv_img = p.adv_y = p.adv_x = 0;
    p.subpix = 0;
    p.fastseg = (d->upsample == 0 || d->subpixel) && d->wrap_pad == 0 && tuning("PF_CONV_FASTSEG", 1) ? 1 : 0;
    if (d->subpixel) {
        // nearest x2 + 3x3 conv == four 2x2 convolutions on the low-resolution grid, one per output phase (blockIdx.z): 4 Cin
        // instead of 9 Cin MACs per output value.  Everything below is the LOW-resolution problem; out_row() scatters the rows.
        p.subpix = 1; p.ksize = 2; p.up = 0; p.pad = 0;
        p.h_out = d->h_out / 2; p.w_out = d->w_out / 2;
        p.rows_per_img = p.h_out * p.w_out;
        p.M = d->n_img * p.rows_per_img; p.K = 4 * Ctot;
        p.crop = d->crop / 2;
        p.batch = 4; p.a_bs = 0; p.out_bs = 0; p.res_bs = 0; p.w_bs = static_cast<long>(p.N) * p.K;
    }
}
static inline int eff_batch(const pf_conv_desc* d) { return d->subpixel ? 4 : d->batch; }

// Rows per GroupNorm-moment part (pf_conv_desc.gn_partial) of this problem under plan g: the fragment rows of one
// wavefront -- or 0 where the moments cannot be produced: split K (the reduce kernel writes the output), a batch,
// images that are not whole parts, or an operand mix that takes the per-fragment (generic) epilogue.
static int gn_rows_for(const GemmParams& p, const GemmPlan& g, int batch) {
    if ((batch != 1 && !p.subpix) || g.splits > 1 || g.m_split > 0 || p.geglu || p.split_out) return 0;   // (the four sub-pixel phases of an image are contiguous runs: gn_part)
    // Layers with a residual are left to the consumer's statistics pass by default.  Round 3: the moment phase has to read the
    // residual tile a second time, which costs an HBM-bound layer as much as that pass saves (fp32-residual linear at
    // 163840 x 320: 141 -> 171 us, the pass it replaces 42 us; VAE decode 107 -> 122 ms; profiles/archive/r3d_gemm_gn.txt).  Round 4: fp32
    // tiles with an fp32 residual form their moments inside the store loop instead (epilogue_f32_stats: no second read, no extra
    // registers) -- and the step still does not move: 63.17 vs 63.33 / 63.54 ms on one box, 61.23 / 61.40 vs 61.00 on another
    // (profiles/r4ah_ab_gn_moments_residual.txt): ~30 statistics passes of 10-55 us leave the critical stream, the column-block-major
    // store order of the fused epilogue gives as much back.  Off; PF_GN_EPILOGUE_RES=1 enables it (tests run both).
    // Without a residual (resnet conv1 -> norm2, the up-sampling conv) the phase costs 2-3 us against a 25-40 us pass.
    if (p.residual && tuning("PF_GN_EPILOGUE_RES", 0) == 0) return 0;
    if (g.big && tuning("PF_GEMM8_WAVES", 8) == 4) return 0;       // (the one-wave-per-SIMD A/B instantiation has no moment variant)
    const int rows = g.big ? 64 : 16 * g.mrep;
    const int BM = g.big ? 256 : 32 * g.mrep;
    if (p.rows_per_img % rows != 0 || p.N % (32 * g.nrep) != 0) return 0;      // whole runs per image, whole N tiles
    if (p.out_f32) {
        const bool ok = !p.rowvec && (p.out_ld & 3) == 0 && (!p.residual || (p.res_f32 && (p.res_ld & 3) == 0));
        return ok ? rows : 0;
    }
    const bool staged = !p.res_f32 && (p.out_ld & 7) == 0 && (p.N & 7) == 0 && !(p.rowvec && p.residual) &&
                        !(p.rowvec && p.rows_per_img < BM);
    return staged ? rows : 0;
}


// Which problems take the 32x32x16 kernel of pf_gemm32.hip (256 x 320 tiles, one persistent block per CU): plain (not split-precision)
// layers with N a multiple of 320 and a long K whose tiles fill whole rounds of 256 CUs -- or whole rounds plus a tail that a split-K
// launch spreads over the chip once more (640 tiles = 2 rounds + 128 tiles x 2 K slices; 320 = 1 round + 64 x 4).  OFF by default
// (PF_GEMM32=1 enables it; PF_GEMM32_MINK is the least K, PF_GEMM32_K1=1 also admits 1x1 layers): its K loop needs 1680 clocks per 32-wide
// stage against the 16x16 kernel's 2140 per equal-FLOP step, and on N(0,1) operands it is no faster -- the chip is POWER capped there
// (zero operands: +7 %), v_mfma_f32_32x32x16 draws ~13 % more per FLOP than v_mfma_f32_16x16x32 on random data, and the step is 0.6 ms
// slower with it because of the split-K tails its 2.5 / 1.25 rounds need (profiles/r6_gemm32_power_cap.txt).
struct Plan32 { bool use; int m_split, tail_splits, tail_kb; };
static Plan32 plan32(const GemmParams& p, int batch, bool allow_split) {
    Plan32 r{false, 0, 0, 0};
    static const int min_k = tuning("PF_GEMM32_MINK", 2560), k1 = tuning("PF_GEMM32_K1", 0);
    const int on = tuning("PF_GEMM32", 0);                         // (read per call: the tests switch it on for their own launches)
    if (!on || p.s3 || p.subpix || batch != 1 || p.N % 320 != 0 || p.K < min_k || p.K % 64 != 0 || p.geglu) return r;
    if (p.ksize != 3 && !k1) return r;
    const long ntl = p.N / 320, mt = cdiv(p.M, 256), tiles = mt * ntl;
    const long full = tiles / 256 * 256, rest = tiles - full;
    if (rest == 0) { r.use = true; return r; }
    if (full == 0 || !allow_split || p.N % 4 != 0) return r;
    const long rows1 = full / ntl;                                  // tile rows of the unsplit launch (ntl divides 256)
    long sp = (256 + rest / 2) / rest;
    const int nkb = p.K / 64;
    if (sp > nkb / 8) sp = nkb / 8;
    if (sp < 2 || rest * sp > 256) return r;
    r.use = true;
    r.m_split = static_cast<int>(rows1 * 256);
    r.tail_kb = static_cast<int>(cdiv(nkb, sp));
    r.tail_splits = static_cast<int>(cdiv(nkb, r.tail_kb));
    return r;
}
// Rows per GroupNorm-moment run of the 32x32 kernel (a wave's 64 rows), or 0: a split-K tail (the reduce kernel writes those rows), a
// residual (left to the consumer's pass, see gn_rows_for), a row vector over images that are not whole runs, pair output.
static int gn_rows32(const GemmParams& p, const Plan32& g) {
    if (!g.use || g.m_split > 0 || p.residual || p.split_out || p.geglu) return 0;
    if (p.rows_per_img % 64 != 0) return 0;
    const bool ok = p.out_f32 ? (p.out_ld & 3) == 0 : (p.out_ld & 7) == 0;
    return ok ? 64 : 0;
}

}  // namespace pf

using namespace pf;

extern "C" pf_status pf_conv_gemm(const pf_conv_desc* d, void* stream) {
    PF_REQUIRE(d, "pf_conv_gemm: null descriptor");
    PF_REQUIRE(d->a0 && d->w && d->out, "pf_conv_gemm: null pointer");
    PF_REQUIRE(d->dtype == PF_BF16 || d->dtype == PF_F16, "pf_conv_gemm: dtype must be PF_BF16 or PF_F16");
    PF_REQUIRE(d->out_dtype == d->dtype || d->out_dtype == PF_F32, "pf_conv_gemm: out_dtype must equal dtype or be PF_F32");
    PF_REQUIRE(!d->residual || d->res_dtype == d->dtype || d->res_dtype == PF_F32, "pf_conv_gemm: res_dtype must equal dtype or be PF_F32");
    const int c1 = d->a1 ? d->c1 : 0;
    const int Ctot = d->c0 + c1;
    PF_REQUIRE(d->c0 > 0 && d->c0 % 64 == 0 && c1 % 64 == 0, "pf_conv_gemm: channel counts (%d,%d) must be multiples of 64", d->c0, c1);
    PF_REQUIRE(d->ksize == 1 || d->ksize == 3, "pf_conv_gemm: ksize must be 1 or 3");
    PF_REQUIRE(d->stride == 1 || d->stride == 2, "pf_conv_gemm: stride must be 1 or 2");
    PF_REQUIRE(d->pad == 0 || d->pad == 1, "pf_conv_gemm: pad must be 0 or 1");
    PF_REQUIRE(d->upsample == 0 || d->upsample == 1, "pf_conv_gemm: upsample must be 0 or 1");
    PF_REQUIRE(d->n_img > 0 && d->h_in > 0 && d->w_in > 0 && d->h_out > 0 && d->w_out > 0, "pf_conv_gemm: bad spatial sizes");
    // n_out that is not a multiple of 4 is allowed for a bare product (no epilogue operands) when
    // out_ld leaves room for the 4-wide store; the extra columns are written as zeros.
    PF_REQUIRE(d->n_out > 0 && (d->n_out % 4 == 0 || (!d->bias && !d->rowvec && !d->residual && d->out_ld >= (d->n_out + 3) / 4 * 4)),
               "pf_conv_gemm: n_out=%d must be a multiple of 4 (or no epilogue operands and a padded out_ld)", d->n_out);
    PF_REQUIRE(d->a0_ld >= d->c0 && d->a0_ld % 8 == 0, "pf_conv_gemm: a0_ld must be >= c0 and a multiple of 8");
    PF_REQUIRE(!d->a1 || (d->a1_ld >= c1 && d->a1_ld % 8 == 0), "pf_conv_gemm: a1_ld must be >= c1 and a multiple of 8");
    PF_REQUIRE(d->epilogue == PF_EPILOGUE_GEGLU || (d->out_ld >= d->n_out && d->out_ld % 4 == 0), "pf_conv_gemm: out_ld must be >= n_out and a multiple of 4");
    PF_REQUIRE(!d->residual || (d->res_ld >= d->n_out && d->res_ld % 4 == 0), "pf_conv_gemm: res_ld must be >= n_out and a multiple of 4");
    PF_REQUIRE(!d->rowvec || (d->rowvec_ld >= d->n_out && d->rowvec_ld % 4 == 0), "pf_conv_gemm: rowvec_ld must be >= n_out and a multiple of 4");
    PF_REQUIRE(aligned16(d->a0) && aligned16(d->w) && aligned16(d->out) && (!d->a1 || aligned16(d->a1)) &&
               (!d->bias || aligned16(d->bias)) && (!d->rowvec || aligned16(d->rowvec)) &&
               (!d->residual || aligned16(d->residual)), "pf_conv_gemm: pointers must be 16-byte aligned");
    PF_REQUIRE(d->batch >= 1, "pf_conv_gemm: batch must be >= 1");
    PF_REQUIRE(d->epilogue == PF_EPILOGUE_NONE || d->epilogue == PF_EPILOGUE_GEGLU || d->epilogue == PF_EPILOGUE_SPLIT,
               "pf_conv_gemm: unknown epilogue %d", d->epilogue);
    if (d->epilogue == PF_EPILOGUE_SPLIT)
        PF_REQUIRE(d->n_out % 32 == 0 && d->out_dtype == d->dtype && d->out_ld >= 2 * d->n_out && !d->rowvec,
                   "pf_conv_gemm: SPLIT epilogue needs n_out %% 32 == 0 (hi / lo interleave per 32 columns), a 16-bit output with out_ld >= 2 n_out, no row vector");
    if (d->split3)
        PF_REQUIRE(!d->a1 && d->c0 % 64 == 0, "pf_conv_gemm: a split-precision walk (split3) takes ONE source of 2 x channels = c0 (hi / lo interleaved per 32)");
    if (d->epilogue == PF_EPILOGUE_GEGLU)
        PF_REQUIRE(d->n_out % 4 == 0 && d->out_dtype == d->dtype && !d->residual && d->out_ld >= d->n_out / 2 && d->out_ld % 2 == 0,
                   "pf_conv_gemm: GEGLU epilogue needs n_out %% 4 == 0, 16-bit output, no residual, out_ld >= n_out/2");
    PF_REQUIRE(d->wrap_pad >= 0 && d->wrap_pad <= 2 && d->crop >= 0 && d->crop <= 2 && d->wrap_pad <= d->w_in,
               "pf_conv_gemm: wrap_pad and crop must be in 0..2");
    {   // the output size must be what the conv arithmetic produces
        const int hl = d->h_in << d->upsample, wl = (d->w_in + 2 * d->wrap_pad) << d->upsample;
        const int ho = (hl + 2 * d->pad - d->ksize) / d->stride + 1, wo = (wl + 2 * d->pad - d->ksize) / d->stride + 1 - 2 * d->crop;
        // one more zero row / column at the bottom / right is allowed (pixels past the input read as zero anyway):
        // diffusers Downsample2D(padding=0) = F.pad(x, (0, 1, 0, 1)) + conv3x3 stride 2 of the VAE encoder
        const int ho1 = (hl + 2 * d->pad + 1 - d->ksize) / d->stride + 1, wo1 = (wl + 2 * d->pad + 1 - d->ksize) / d->stride + 1 - 2 * d->crop;
        PF_REQUIRE((d->h_out == ho && d->w_out == wo) || (d->h_out == ho1 && d->w_out == wo1),
                   "pf_conv_gemm: output size (%d,%d) does not match (%d,%d) [or (%d,%d) with a trailing zero row / column]",
                   d->h_out, d->w_out, ho, wo, ho1, wo1);
    }
    if (d->subpixel)
        PF_REQUIRE(d->ksize == 3 && d->upsample == 1 && d->stride == 1 && d->pad == 1 && d->batch == 1 && !d->residual && !d->rowvec &&
                   d->epilogue == PF_EPILOGUE_NONE && d->h_out % 2 == 0 && d->w_out % 2 == 0 && d->crop % 2 == 0 && d->wrap_pad <= 1,
                   "pf_conv_gemm: subpixel serves nearest x2 + 3x3 stride-1 pad-1 convolutions (bias only; weights packed [4][n_out][2][2][c0 + c1])");
    const int batch = eff_batch(d);
    GemmParams p;
    params_from_desc(d, p);
    Plan32 g32 = plan32(p, batch, d->workspace != nullptr);
    if (g32.use && d->gn_partial && gn_rows32(p, g32) == 0) g32.use = false;     // (moments asked for: the plan that can emit them)
    if (g32.use) {
        if (d->gn_partial) {
            PF_REQUIRE(aligned16(d->gn_partial), "pf_conv_gemm: gn_partial must be 16-byte aligned");
            p.gn_partial = d->gn_partial;
            p.gn_rows = 64;
        }
        const long npix = static_cast<long>(d->n_img) * d->h_in * d->w_in;
        const long a0b = ((npix - 1) * p.a0_ld + p.c0) * 2, a1b = p.a1 ? ((npix - 1) * p.a1_ld + p.c1) * 2 : 0;
        const long wb = static_cast<long>(p.N) * p.K * 2;
        PF_REQUIRE(a0b < (1L << 31) && a1b < (1L << 31) && wb < (1L << 31),
                   "pf_conv_gemm: each operand must be smaller than 2 GiB (32-bit buffer offsets)");
        p.a0_bytes = static_cast<unsigned>(a0b); p.a1_bytes = static_cast<unsigned>(a1b); p.w_bytes = static_cast<unsigned>(wb);
        p.splits = 1; p.kb_per_split = p.K / 64;
        {
            const ProfState ps = prof_snapshot();
            p.prof = (ps.buf && ps.blocks >= 256) ? ps.buf : nullptr;
        }
        hipStream_t st32 = as_stream(stream);
        if (g32.m_split == 0) return launch_gemm32(p, d->dtype, 1, st32);
        const size_t need = static_cast<size_t>(g32.tail_splits) * (p.M - g32.m_split) * p.N * sizeof(float);
        PF_REQUIRE(d->workspace_bytes >= need && aligned16(d->workspace),
                   "pf_conv_gemm: workspace of %zu bytes (16-byte aligned) needed, got %zu", need, d->workspace_bytes);
        GemmParams p1 = p, p2 = p;
        p1.M = g32.m_split;
        p2.m_begin = g32.m_split; p2.splits = g32.tail_splits; p2.kb_per_split = g32.tail_kb;
        p2.partial = static_cast<float*>(d->workspace);
        pf_status s1 = launch_gemm32(p1, d->dtype, 1, st32);
        if (s1 != PF_OK) return s1;
        s1 = launch_gemm32(p2, d->dtype, 1, st32);
        if (s1 != PF_OK) return s1;
        const long total = static_cast<long>(p2.M - p2.m_begin) * (p2.N / 4);
        PF_DISPATCH_16(d->dtype, "pf_conv_gemm", hipLaunchKernelGGL((k_splitk_reduce<T>), dim3(cdiv(total, 256)), dim3(256), 0, st32, p2));
        PF_CHECK_LAUNCH("pf_conv_gemm (split-K reduce)");
        return PF_OK;
    }
    GemmPlan g = plan_gemm(p.M, p.N, p.K, batch, d->workspace != nullptr, p.s3 != 0);
    if (d->gn_partial) {
        const int r = gn_rows_for(p, g, batch);
        PF_REQUIRE(r > 0 && aligned16(d->gn_partial), "pf_conv_gemm: gn_partial given but this problem cannot emit GroupNorm moments (ask pf_conv_gemm_gn_rows first)");
        p.gn_partial = d->gn_partial;
        p.gn_rows = r;
    }
    {   // extents for the buffer descriptors of the 8-wave kernel (32-bit offsets, < 2 GiB)
        const long npix = static_cast<long>(d->n_img) * d->h_in * d->w_in;
        const long a0b = ((npix - 1) * p.a0_ld + p.c0) * 2, a1b = p.a1 ? ((npix - 1) * p.a1_ld + p.c1) * 2 : 0;
        const long wb = static_cast<long>(p.N) * p.K * 2;      // (per batch element / sub-pixel phase: the kernels offset the base by w_bs)
        PF_REQUIRE(a0b < (1L << 31) && a1b < (1L << 31) && wb < (1L << 31),
                   "pf_conv_gemm: each operand must be smaller than 2 GiB (32-bit buffer offsets)");
        p.a0_bytes = static_cast<unsigned>(a0b); p.a1_bytes = static_cast<unsigned>(a1b); p.w_bytes = static_cast<unsigned>(wb);
    }
    p.splits = g.splits; p.kb_per_split = g.kb_per_split;
    p.partial = static_cast<float*>(d->workspace);
    // In-launch combine: implemented, bit-identical (tests/test_gpu_kernels.py), and OFF by default -- it does not pay at these
    // tile sizes: the last-arriving workgroup reads splits x 164 KB of slabs through one CU (~65 GB/s per block cross-XCD,
    // MI355X_MICROARCH.md "handoff-payload") while the second kernel spreads the same reads over the whole chip: the step went
    // 65.3 -> 67.8 ms with write-through slabs, -> 72.9 ms with a release fence per workgroup (profiles/archive/r3k_ab_splitk.txt,
    // r3l_ab_splitk.txt).  PF_SPLITK_INKERNEL=1 enables it for A/B.
    if (d->tickets && tuning("PF_SPLITK_INKERNEL", 0) && d->workspace_bytes < (1ULL << 32)) {      // (32-bit slab offsets)
        // enough zeroed counters for every (batch, tile) of the split launch?  (tile counts of split plans: <= 320 of the
        // 4-wave kernel, <= 255 of the 8-wave one)
        const int bm = g.big ? g.bm : 32 * g.mrep, bn = 32 * g.nrep;
        const long rows = g.big && g.m_split > 0 ? p.M - g.m_split : p.M;
        const long need = cdiv(rows, bm) * cdiv(p.N, bn) * batch;
        PF_REQUIRE(d->n_tickets >= need && (reinterpret_cast<uintptr_t>(d->tickets) & 3) == 0,
                   "pf_conv_gemm: %ld arrival counters needed, %d given", need, d->n_tickets);
        p.tickets = d->tickets;
    }
    p.m_begin = 0;
    hipStream_t st = as_stream(stream);
    if (g.big && g.m_split > 0) {           // two launches: full rounds unsplit, then the tail rows with split K
        const size_t need = static_cast<size_t>(g.tail_splits) * (p.M - g.m_split) * p.N * sizeof(float);
        PF_REQUIRE(d->workspace_bytes >= need && aligned16(d->workspace),
                   "pf_conv_gemm: workspace of %zu bytes (16-byte aligned) needed, got %zu", need, d->workspace_bytes);
        GemmParams p1 = p, p2 = p;
        p1.M = g.m_split; p1.splits = 1; p1.kb_per_split = p.K / 64;
        p2.m_begin = g.m_split; p2.splits = g.tail_splits; p2.kb_per_split = g.tail_kb;
        PF_DISPATCH_16(d->dtype, "pf_conv_gemm",
            if (g.nrep == 5) { pf_status s1 = launch8<T, 5>(p1, 1, st, g.bm); if (s1 != PF_OK) return s1; return launch8<T, 5>(p2, 1, st, g.bm); }
            else { pf_status s1 = launch8<T, 4>(p1, 1, st, g.bm); if (s1 != PF_OK) return s1; return launch8<T, 4>(p2, 1, st, g.bm); });
    }
    if (p.splits > 1) {
        const size_t need = static_cast<size_t>(p.splits) * batch * p.M * p.N * sizeof(float);
        PF_REQUIRE(d->workspace_bytes >= need && aligned16(d->workspace),
                   "pf_conv_gemm: workspace of %zu bytes (16-byte aligned) needed, got %zu", need, d->workspace_bytes);
    }
    if (g.big) {
        PF_DISPATCH_16(d->dtype, "pf_conv_gemm",
            if (g.nrep == 5) return launch8<T, 5>(p, batch, st, g.bm);
            else return launch8<T, 4>(p, batch, st, g.bm));
    }
    PF_DISPATCH_16(d->dtype, "pf_conv_gemm",
        if (g.nrep == 5) return g.mrep == 2 ? launch<T, 2, 5>(p, batch, st) : launch<T, 4, 5>(p, batch, st);
        else return g.mrep == 2 ? launch<T, 2, 4>(p, batch, st) : launch<T, 4, 4>(p, batch, st));
    return PF_OK;
}

extern "C" pf_status pf_debug_gemm_profile(void* device_buffer, long capacity_blocks) {
    // (states are leaked on purpose: a few bytes per call of a diagnostics switch, never freed under a reader)
    const ProfState* s = device_buffer ? new ProfState{static_cast<unsigned long long*>(device_buffer), capacity_blocks} : nullptr;
    g_prof_state.store(s, std::memory_order_release);
    return PF_OK;
}

extern "C" int pf_conv_gemm_gn_rows(const pf_conv_desc* d) {
    if (!d || d->batch < 1 || d->n_out < 1 || d->n_img < 1) return 0;
    GemmParams p;
    params_from_desc(d, p);
    if (d->subpixel) return gn_rows_for(p, plan_gemm(p.M, p.N, p.K, 4, true, p.s3 != 0), 4);
    {
        const Plan32 g32 = plan32(p, d->batch, true);
        if (g32.use) {
            const int r = gn_rows32(p, g32);
            if (r > 0) return r;                                    // else: the 16x16 kernels' plan below serves a launch that asks for moments
        }
    }
    return gn_rows_for(p, plan_gemm(p.M, p.N, p.K, d->batch, true, p.s3 != 0), d->batch);
}

extern "C" int pf_conv_gemm_kernel_id(const pf_conv_desc* d) {
    if (!d || d->batch < 1 || d->n_out < 1 || d->n_img < 1) return -1;
    GemmParams p;
    params_from_desc(d, p);
    if (plan32(p, eff_batch(d), true).use) return 2;
    return plan_gemm(p.M, p.N, p.K, eff_batch(d), true, p.s3 != 0).big ? 1 : 0;
}

extern "C" size_t pf_conv_gemm_workspace_size(const pf_conv_desc* d) {
    if (!d || d->batch < 1 || d->n_out < 1) return 0;
    const int c1 = d->a1 ? d->c1 : 0;
    if (d->subpixel) {                                              // the four phase problems on the low-resolution grid (params_from_desc)
        GemmParams p;
        params_from_desc(d, p);
        const GemmPlan g = plan_gemm(p.M, p.N, p.K, 4, true, p.s3 != 0);
        return g.splits > 1 ? static_cast<size_t>(g.splits) * 4 * p.M * p.N * sizeof(float) : 0;
    }
    const long M = static_cast<long>(d->n_img) * d->h_out * d->w_out;
    const int K = d->ksize * d->ksize * (d->c0 + c1);
    size_t need32 = 0;
    {
        GemmParams p;
        params_from_desc(d, p);
        const Plan32 g32 = plan32(p, d->batch, true);
        if (g32.use) {
            need32 = g32.m_split > 0 ? static_cast<size_t>(g32.tail_splits) * (M - g32.m_split) * d->n_out * sizeof(float) : 0;
            if (gn_rows32(p, g32) > 0 || p.residual) return need32;      // (no launch of this problem falls back to the 16x16 plan)
        }
    }
    // (a launch that asks for GroupNorm moments the 32x32 plan cannot emit takes the 16x16 plan: room for either)
    const GemmPlan g = plan_gemm(M, d->n_out, K, d->batch, true, d->split3 != 0);
    if (g.big && g.m_split > 0) return std::max(need32, static_cast<size_t>(g.tail_splits) * (M - g.m_split) * d->n_out * sizeof(float));
    return std::max(need32, g.splits > 1 ? static_cast<size_t>(g.splits) * d->batch * M * d->n_out * sizeof(float) : static_cast<size_t>(0));
}
