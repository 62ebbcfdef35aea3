// Weight-stationary linear layers for the token streams of the two finest UNet levels (gfx950 / CDNA4).
//
//   out[m][n] = sum_k a[m][k] * w[n][k]  (+ bias[n]) (+ residual[m][n])        K = 320, N a multiple of 320   (64^2 level, every mode)
//                                                                                K = 640, N a multiple of 256   (32^2 level: 16-bit and GEGLU outputs; of 128: 16-bit)
//                                                                                K = 1280, N a multiple of 128  (16^2 level: 16-bit and GEGLU outputs)
//
// Why a second GEMM structure: at K = 320 the tile kernel of pf_gemm.hip (256 x 160 tile, K loop of 5 steps) spends
// 6.4 k clocks of a 32 k-clock tile in the matrix pipes (profiles/r4a_timeline.txt): 260 KB of operands per tile enter
// through a pipeline that only ramps up, and the epilogue (11 - 16 k clocks) runs with the matrix pipes idle.  Here the roles
// are turned round: a workgroup keeps a 320-channel slab of the WEIGHTS in registers for its whole life (8 wavefronts:
// four hold 48 output channels, four hold 32, so that every SIMD serves 48 + 32 = 80 channels -- 120 / 80 registers per
// lane) and streams 64-token tiles of the activation matrix through a 3-slot LDS ring (40 KB per slot, LDS-DMA with the
// XOR-swizzled source offsets of pf_gemm.hip, one tile = the FULL K).  There is no K loop to pipeline, the only operand
// traffic is the activation stream (12.5 B per clock and CU against ~24 for the tile kernel's two operands), and every
// wavefront runs its own epilogue on 32-token slices between its MFMA bursts: the partner wavefront of the SIMD covers it.
//
// Grid: N / 320 channel blocks x S token ranges, block b on XCD b % 8; the channel blocks of one token range share an XCD
// (one L2 fill of the activation tiles).  Epilogue modes: 16-bit out | fp32 out + fp32 residual (the residual stream of the
// mixed scheme) | GEGLU pairing (transformer.py:8-21) | q | k | v in ONE launch with V written transposed [C][keys]
// (what pf_attention reads).
//
// Replaces cuBLAS behind the nn.Linear layers of diffusers' BasicTransformerBlock / the reference's
// models/modules/transformer.py:57-74 (to_q / to_k / to_v / to_out), :8-38 (GEGLU FeedForward) at C = 320.
#include "pf_common.h"
#include <atomic>
#include <stdlib.h>
#include <algorithm>

namespace pf {

struct LwsParams {
    const unsigned short* a; int a_ld;
    const unsigned short* w;
    const float* bias;
    const float* residual; int res_ld;
    void* out; int out_ld;
    unsigned short* out_vt; int vt_ld; int rows_per_batch; long vt_bs;
    const float* ln_gamma; const float* ln_beta; float ln_eps; unsigned short* ln_out; int ln_ld;   // mode LWS_F32_LN
    int M, N;
    int nblocks, splits_per_xcd, ntiles;
    int flat_splits;              // > 0: block b -> (token range b / nblocks, channel block b % nblocks) without the XCD grouping
    unsigned a_bytes;
};

template <typename T> struct LwsMfma;
template <> struct LwsMfma<Bf16> {
    typedef __attribute__((ext_vector_type(8))) __bf16 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct LwsMfma<F16> {
    typedef __attribute__((ext_vector_type(8))) _Float16 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};


// Two shapes of the same kernel (K = reduction width, BM = tokens per ring slot, CHB = output channels per workgroup):
//   K = 320: 64-token tiles, 320 channels per workgroup (4 waves x 48 + 4 x 32)      -- the 64^2 level
//   K = 640: 32-token tiles, 256 channels per workgroup (8 waves x 32: the slab is 2 x 20 fragments = 160 registers) -- the 32^2 level
//   K = 1280: 16-token tiles, 128 channels per workgroup (8 waves x 16: 1 x 40 fragments)                            -- the 16^2 level
// Either way a ring slot is 40 KB and a wavefront moves 5 one-KB pieces of it.
constexpr int LWS_STAGES = 3, LWS_PIECES = 5;
constexpr int LWS_STAGE_ELEMS = 64 * 320;                         // 16-bit elements per ring slot (40 KB) = BM * K for both shapes
constexpr int LWS_STG_BYTES = 4096;                               // per-wave staging region (<= 48 channel rows x 80 B)
enum { LWS_16 = 0, LWS_F32 = 1, LWS_GEGLU = 2, LWS_QKV = 3, LWS_F32_LN = 4, LWS_VT = 5 };
constexpr int LWS_LN_BYTES = 2 * 8 * 32 * 8;                       // LayerNorm exchange: [half parity][wave][token] (mean, M2) fp32

// The body of one wavefront: NB output-channel sub-blocks of 16 (cb = first channel inside the channel block).
constexpr int lws_waitcnt_vm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }   // gfx9 s_waitcnt immediate: vmcnt(n) only

template <typename T, int MODE, int NB, bool COUNTED, int LWS_K, int LWS_BM, int CHB>
__device__ __forceinline__ void lws_wave(const LwsParams& p, unsigned short* smem, unsigned char* stg, float2* lnx, int nblk, int cb,
                                         int t_lo, int t_hi, int wave, int lane) {
    typedef typename LwsMfma<T>::frag frag;
    constexpr int LWS_KS = LWS_K / 32, LWS_KB = LWS_K / 64;
    static_assert(LWS_BM * LWS_K == LWS_STAGE_ELEMS && LWS_KB * (LWS_BM / 8) == 8 * LWS_PIECES, "a ring slot is 40 KB = 8 waves x 5 pieces");
    static_assert(MODE != LWS_F32_LN || CHB == 320, "the LayerNorm epilogue is written for the 48 / 32-channel wave split");
    constexpr int PB = LWS_BM >= 32 ? 2 : 1, HALVES = LWS_BM >= 32 ? LWS_BM / 32 : 1;   // an epilogue slice = PB blocks of 16 tokens
    static_assert(PB == 2 || MODE == LWS_16 || MODE == LWS_GEGLU || MODE == LWS_F32 || MODE == LWS_VT, "16-token tiles: plain, GEGLU and transposed outputs only");
    const int frow = lane & 15, fchunk = lane >> 4, cq = 4 * fchunk;
    const int n_base = nblk * CHB + cb;                           // first output channel of this wavefront

    // ---- DMA addressing: a tile is KB images [BM rows][64 k] of 128-byte rows; a piece = 8 rows of one image.  Piece q = 5 wave + i of
    // a wave is (K block, row group) = (q / (BM / 8), q % (BM / 8)): at BM = 64 rows [8 wave, 8 wave + 8) of each of the 5 K blocks.
    constexpr int RG = LWS_BM / 8;
    const int chunk = lane & 7;
    auto piece_row = [&](int i) { return ((wave * LWS_PIECES + i) % RG) * 8 + (lane >> 3); };
    auto piece_kb = [&](int i) { return (wave * LWS_PIECES + i) / RG; };
    auto uniform_ptr = [](const unsigned short* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
        const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
        return reinterpret_cast<unsigned short*>(static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi) << 32));
    };
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.a), 0, __builtin_amdgcn_readfirstlane(p.a_bytes), 0x00020000);
    auto dma_tile = [&](int tile, int slot) {                     // rows past M: an offset beyond num_records reads as zero
        unsigned short* dst = smem + slot * LWS_STAGE_ELEMS;
        const int soff0 = __builtin_amdgcn_readfirstlane(tile * LWS_BM * p.a_ld * 2);
#pragma unroll
        for (int i = 0; i < LWS_PIECES; ++i) {
            const int lrow = piece_row(i), kb = __builtin_amdgcn_readfirstlane(piece_kb(i));
            const unsigned src_lane = static_cast<unsigned>(lrow * p.a_ld + ((chunk ^ ((lrow >> 1) & 7)) << 3)) * 2u;   // bytes
            const unsigned voff = tile * LWS_BM + lrow < p.M ? src_lane : 0x80000000u;
            const int rg8 = __builtin_amdgcn_readfirstlane(((wave * LWS_PIECES + i) % RG) * 8);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(dst + kb * LWS_BM * 64 + rg8 * 64), 16,
                                                     voff, soff0 + kb * 128, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                        // (program order of the vector-memory operations is what the counted wait counts)
    };
    if (t_lo < t_hi) dma_tile(t_lo, 0);
    if (t_lo + 1 < t_hi) dma_tile(t_lo + 1, 1);

    // ---- the weight slab of this wavefront: NB x 10 MFMA "A" fragments, loaded once
    frag wf[NB][LWS_KS];
    {
        const unsigned short* wp = p.w + static_cast<long>(n_base + frow) * LWS_K + fchunk * 8;
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int ks = 0; ks < LWS_KS; ++ks)
                wf[j][ks] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(wp + j * 16 * LWS_K + ks * 32));
    }
    float4 bias[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
        bias[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n_base + j * 16 + cq) : float4{0.f, 0.f, 0.f, 0.f};

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // first tiles + weights (once)
    // ---- epilogue of 32 tokens x NB*16 channels (rows m_half .. m_half + 31); the accumulators were initialised with the bias
    auto epilogue = [&](f32x4 (&acc)[2][NB], int m_half) {
#ifdef PF_LWS_ABL_NOEPI
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int j = 0; j < NB; ++j) asm volatile("" :: "v"(acc[pb][j]));
        return;
#endif
        if constexpr (MODE == LWS_F32_LN) {
            // out = a w^T + bias + residual (fp32 stream) AND ln_out = LayerNorm(out) gamma + beta in 16 bit: the row is complete inside
            // the workgroup (N == 320), so the LayerNorm that follows every attention output projection of a transformer block costs
            // no pass of its own (it read the 210 MB stream tensor back and wrote 105 MB).  Statistics: every wavefront forms the mean
            // and the centred sum of squares of ITS channels of a token (two passes over registers, a 4-lane exchange each), the 8
            // partial (mean, M2) pairs meet in LDS and are combined by Chan's formula -- no E[x^2] - mean^2 cancellation.
            const float* rp = p.residual;
            float* op = static_cast<float*>(p.out);
            float mean[2], rstd[2];
            float2* slot_x = lnx + ((m_half >> 5) & 1) * 8 * 32;      // [wave][token of the half]
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int m = m_half + pb * 16 + frow;
                const long mc = m < p.M ? m : p.M - 1;
                float4 r[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    r[j] = rp ? *reinterpret_cast<const float4*>(rp + mc * p.res_ld + n_base + j * 16 + cq) : float4{0.f, 0.f, 0.f, 0.f};
                float s1 = 0.f;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc[pb][j][0] += r[j].x; acc[pb][j][1] += r[j].y; acc[pb][j][2] += r[j].z; acc[pb][j][3] += r[j].w;
                    if (m < p.M) *reinterpret_cast<float4*>(op + static_cast<long>(m) * p.out_ld + n_base + j * 16 + cq) =
                        float4{acc[pb][j][0], acc[pb][j][1], acc[pb][j][2], acc[pb][j][3]};
                    s1 += (acc[pb][j][0] + acc[pb][j][1]) + (acc[pb][j][2] + acc[pb][j][3]);
                }
                s1 += __shfl_xor(s1, 16);
                s1 += __shfl_xor(s1, 32);
                const float mw = s1 * (1.0f / (NB * 16));
                float m2 = 0.f;
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = acc[pb][j][e] - mw; m2 += d * d; }
                m2 += __shfl_xor(m2, 16);
                m2 += __shfl_xor(m2, 32);
                if (fchunk == 0) slot_x[wave * 32 + pb * 16 + frow] = float2{mw, m2};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // (every wave runs the same halves: the barrier count is uniform)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                float2 px[8];
#pragma unroll
                for (int wv = 0; wv < 8; ++wv) px[wv] = slot_x[wv * 32 + pb * 16 + frow];
                float mu = 0.f;
#pragma unroll
                for (int wv = 0; wv < 8; ++wv) mu += px[wv].x * (wv < 4 ? 48.0f : 32.0f);
                mu *= (1.0f / 320.0f);
                float M2 = 0.f;
#pragma unroll
                for (int wv = 0; wv < 8; ++wv) { const float d = px[wv].x - mu; M2 += px[wv].y + (wv < 4 ? 48.0f : 32.0f) * d * d; }
                mean[pb] = mu;
                rstd[pb] = __builtin_amdgcn_rsqf(M2 * (1.0f / 320.0f) + p.ln_eps);
            }
            constexpr int RS = NB * 32 + 16;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float4 g = *reinterpret_cast<const float4*>(p.ln_gamma + cb + j * 16 + cq);
                const float4 bt = *reinterpret_cast<const float4*>(p.ln_beta + cb + j * 16 + cq);
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    u16x4 w4;
                    w4[0] = from_f32<T>((acc[pb][j][0] - mean[pb]) * rstd[pb] * g.x + bt.x);
                    w4[1] = from_f32<T>((acc[pb][j][1] - mean[pb]) * rstd[pb] * g.y + bt.y);
                    w4[2] = from_f32<T>((acc[pb][j][2] - mean[pb]) * rstd[pb] * g.z + bt.z);
                    w4[3] = from_f32<T>((acc[pb][j][3] - mean[pb]) * rstd[pb] * g.w + bt.w);
                    *reinterpret_cast<u16x4*>(stg + (pb * 16 + frow) * RS + (j * 16 + cq) * 2) = w4;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int q = i * 64 + lane, row = q / (2 * NB), cc = q - row * 2 * NB;
                const u16x8 x = *reinterpret_cast<const u16x8*>(stg + row * RS + cc * 16);
                const int m = m_half + row;
                if (m < p.M) *reinterpret_cast<u16x8*>(p.ln_out + static_cast<long>(m) * p.ln_ld + cb + cc * 8) = x;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (MODE == LWS_F32) {
            const float* rp = p.residual;
            float* op = static_cast<float*>(p.out);
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const int m = m_half + pb * 16 + frow;
                const long mc = m < p.M ? m : p.M - 1;
                float4 r[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    r[j] = rp ? *reinterpret_cast<const float4*>(rp + mc * p.res_ld + n_base + j * 16 + cq) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float4 v = float4{acc[pb][j][0] + r[j].x, acc[pb][j][1] + r[j].y, acc[pb][j][2] + r[j].z, acc[pb][j][3] + r[j].w};
                    if (m < p.M) *reinterpret_cast<float4*>(op + static_cast<long>(m) * p.out_ld + n_base + j * 16 + cq) = v;
                }
            }
        } else if constexpr (MODE == LWS_GEGLU) {
            // columns interleaved (value, gate): 2 outputs per lane and sub-block -> staged rows of NB*8 outputs; 48-byte rows for both
            // wave kinds (a 32-byte row stride puts rows r, r + 4, ... on the same banks)
            constexpr int RS = 48;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const unsigned lo = from_f32<T>(geglu_value(acc[pb][j][0], acc[pb][j][1]));
                    const unsigned hi = from_f32<T>(geglu_value(acc[pb][j][2], acc[pb][j][3]));
                    *reinterpret_cast<unsigned*>(stg + (pb * 16 + frow) * RS + (j * 8 + (cq >> 1)) * 2) = lo | (hi << 16);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unsigned short* op = static_cast<unsigned short*>(p.out);
            const int ncol = (n_base >> 1);
#pragma unroll
            for (int i = 0; i < (16 * PB * NB + 63) / 64; ++i) {
                const int q = i * 64 + lane, row = q / NB, cc = q - row * NB;
                if (q < 16 * PB * NB) {
                    const u16x8 x = *reinterpret_cast<const u16x8*>(stg + row * RS + cc * 16);
                    const int m = m_half + row;
                    if (m < p.M) *reinterpret_cast<u16x8*>(op + static_cast<long>(m) * p.out_ld + ncol + cc * 8) = x;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // staging region read out before the next slice overwrites it
        } else if constexpr (MODE == LWS_VT) {
            // the whole output transposed (the V projection of a self-attention at K = 640 / 1280): staged [channel][16 PB tokens], leaves as
            // 32 PB-byte key runs of out_vt[b][channel][key]
            constexpr int RS = PB * 32 + 16, CH = 2 * PB;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        *reinterpret_cast<unsigned short*>(stg + (j * 16 + cq + e) * RS + (pb * 16 + frow) * 2) = from_f32<T>(acc[pb][j][e]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int bidx = m_half / p.rows_per_batch, key0 = m_half - bidx * p.rows_per_batch;
            unsigned short* op = p.out_vt + bidx * p.vt_bs + static_cast<long>(n_base) * p.vt_ld + key0;
#pragma unroll
            for (int i = 0; i < (NB * 16 * CH + 63) / 64; ++i) {
                const int q = i * 64 + lane, ch = q / CH, cc = q - ch * CH;
                if (q < NB * 16 * CH) {
                    const u16x8 x = *reinterpret_cast<const u16x8*>(stg + ch * RS + cc * 16);
                    if (m_half + cc * 8 < p.M) *reinterpret_cast<u16x8*>(op + static_cast<long>(ch) * p.vt_ld + cc * 8) = x;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == LWS_QKV && nblk == 2) {
            // V transposed: staged [channel][32 tokens] (64 B of tokens in 80-byte rows), leaves as 64-byte key runs of out_vt[b][channel][key]
            constexpr int RS = 80;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        *reinterpret_cast<unsigned short*>(stg + (j * 16 + cq + e) * RS + (pb * 16 + frow) * 2) = from_f32<T>(acc[pb][j][e]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int bidx = m_half / p.rows_per_batch, key0 = m_half - bidx * p.rows_per_batch;
            unsigned short* op = p.out_vt + bidx * p.vt_bs + static_cast<long>(cb) * p.vt_ld + key0;
#pragma unroll
            for (int i = 0; i < NB; ++i) {                         // NB*16 channel rows x 4 chunks = NB x 64 lanes
                const int q = i * 64 + lane, ch = q >> 2, cc = q & 3;
                const u16x8 x = *reinterpret_cast<const u16x8*>(stg + ch * RS + cc * 16);
                if (m_half + cc * 8 < p.M) *reinterpret_cast<u16x8*>(op + static_cast<long>(ch) * p.vt_ld + cc * 8) = x;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            // 16-bit rows: staged [32 tokens][NB*16 channels] in rows padded by 16 bytes (an unpadded 96- / 64-byte stride puts the 16
            // token rows of a ds_write_b64 on 4 / 2 distinct bank groups: 45 % of the LDS cycles of the first version were conflicts),
            // leaves as 16-byte chunks
            constexpr int RS = NB * 32 + 16;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    u16x4 w4;
                    w4[0] = from_f32<T>(acc[pb][j][0]); w4[1] = from_f32<T>(acc[pb][j][1]);
                    w4[2] = from_f32<T>(acc[pb][j][2]); w4[3] = from_f32<T>(acc[pb][j][3]);
                    *reinterpret_cast<u16x4*>(stg + (pb * 16 + frow) * RS + (j * 16 + cq) * 2) = w4;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unsigned short* op = static_cast<unsigned short*>(p.out);
#pragma unroll
            for (int i = 0; i < (PB * NB + 1) / 2; ++i) {          // 16 PB rows x 2 NB chunks = PB NB x 32 lanes
                const int q = i * 64 + lane, row = q / (2 * NB), cc = q - row * 2 * NB;
                if (q < 16 * PB * 2 * NB) {
                    const u16x8 x = *reinterpret_cast<const u16x8*>(stg + row * RS + cc * 16);
                    const int m = m_half + row;
                    if (m < p.M) *reinterpret_cast<u16x8*>(op + static_cast<long>(m) * p.out_ld + n_base + cc * 8) = x;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };

    // Two ways of hiding the epilogue's vector-ALU work behind matrix work were built and measured on the GEGLU launch (the only one
    // that is not HBM-bound: MFMA busy 41 %, VALU busy 52 %, their sum the whole time) -- both neutral, both removed again:
    // (1) the 32-channel wave of a SIMD running its epilogues half a tile late, so that one wave of the SIMD is in the matrix pipe
    // while the other is in the vector ALU (361 vs 348 us); (2) two accumulator sets with the epilogue of half h - 1 in the
    // scheduling region of the MFMAs of half h, asked for as {1 MFMA, 4 VALU} groups (364 us).  profiles/r4_lws_notes.txt.
    f32x4 acc[2][NB];                                             // [token block][channel sub-block]
    int slot = 0;
    for (int tile = t_lo; tile < t_hi; ++tile) {
        // This tile's 5 DMA pieces were waited for in the PREVIOUS iteration (below, ahead of its last epilogue slice; the first two
        // tiles before the loop).  COUNTED = false (PF_LWS_COUNTED=0): full drain here instead, for A/B.
        if constexpr (!COUNTED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef PF_LWS_ABL_NOBAR   /* -DPF_LWS_ABL_*: timing-only ablation builds (wrong results), make -C panfusion_amd/csrc lws_ablate */
        __builtin_amdgcn_s_barrier();                             // tile landed (all waves' pieces); slot (tile + 2) % 3 no longer read
#endif
        asm volatile("" ::: "memory");
#ifndef PF_LWS_ABL_NODMA
        if (tile + 2 < t_hi) dma_tile(tile + 2, slot >= 1 ? slot - 1 : 2);
#endif
        const unsigned short* As = smem + slot * LWS_STAGE_ELEMS;
        const int m_tile = tile * LWS_BM;
#pragma unroll
        for (int half = 0; half < HALVES; ++half) {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[pb][j] = f32x4{bias[j].x, bias[j].y, bias[j].z, bias[j].w};
            // activation fragments: tokens half*32 + pb*16 + frow, K slab ks = (K block ks >> 1, 32-k half ks & 1)
            auto afrag = [&](int pb, int ks) {
                const int row = half * 32 + pb * 16 + frow;
                const int c = (ks & 1) * 4 + fchunk;
#ifdef PF_LWS_ABL_NOLDS
                u16x8 z = {1, 2, 3, 4, 5, 6, 7, static_cast<unsigned short>(lane + ks)};
                asm volatile("" : "+v"(z));
                return __builtin_bit_cast(frag, z);
#else
                return __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(As + (ks >> 1) * LWS_BM * 64 + row * 64 + ((c ^ ((row >> 1) & 7)) << 3)));
#endif
            };
            // software pipeline over steps of two K slabs: the 4 fragment reads of step s + 1 are issued BEFORE the 4 NB MFMAs
            // of step s and pinned there (left alone, hipcc sinks every read to just ahead of its first use and waits for it:
            // one exposed LDS round trip per K slab)
            frag af[2][2][2];                                     // [buffer][slab of the step][token block]
            auto load_step = [&](int st, int buf) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) af[buf][h2][pb] = afrag(pb, 2 * st + h2);
            };
            load_step(0, 0);
#pragma unroll
            for (int st = 0; st < LWS_KS / 2; ++st) {
                if (st + 1 < LWS_KS / 2) load_step(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int pb = 0; pb < PB; ++pb)
                            acc[pb][j] = LwsMfma<T>::run(wf[j][2 * st + h2], af[st & 1][h2][pb], acc[pb][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (COUNTED) {
                // Wait for the NEXT tile's pieces (issued one iteration ago) here, where the only younger LOADS are the 5 pieces just
                // issued for tile + 2: loads retire in order among themselves, so "at most 5 operations outstanding" implies the next
                // tile has landed -- whatever the stores of the epilogues are doing (stores and loads retire out of order with respect
                // to each other on the one vmcnt of gfx9: a count that included the epilogues' stores let a tile be read before it
                // had arrived, once in ~40 launches at 16-token tiles).  The older stores had a slice of matrix work to complete.
                if (half == HALVES - 1) {
                    if (tile + 2 < t_hi) __builtin_amdgcn_s_waitcnt(lws_waitcnt_vm(LWS_PIECES));
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            epilogue(acc, m_tile + half * 32);
        }
        slot = slot == LWS_STAGES - 1 ? 0 : slot + 1;
    }
}

template <typename T, int MODE, bool COUNTED, int K, int CHB = (K == 320 ? 320 : K == 640 ? 256 : 128)>
__global__ __launch_bounds__(512, 1) void k_linear_ws(const LwsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // block -> (token range, channel block): the channel blocks of one token range sit on one XCD (block b runs on XCD b % 8) --
    // unless that would leave too many CUs without a workgroup (20 channel blocks: 160 of 256), then plainly b -> (b / nblocks, b % nblocks)
    int s, S, nblk;
    if (p.flat_splits > 0) {
        s = blockIdx.x / p.nblocks; nblk = blockIdx.x - s * p.nblocks; S = p.flat_splits;
        if (s >= S) return;
    } else {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int sl = idx / p.nblocks;
        nblk = idx - sl * p.nblocks;
        if (sl >= p.splits_per_xcd) return;
        s = xcd * p.splits_per_xcd + sl; S = 8 * p.splits_per_xcd;
    }
    const int t_lo = static_cast<int>(static_cast<long>(p.ntiles) * s / S), t_hi = static_cast<int>(static_cast<long>(p.ntiles) * (s + 1) / S);
    unsigned char* stg = reinterpret_cast<unsigned char*>(smem + LWS_STAGES * LWS_STAGE_ELEMS) + wave * LWS_STG_BYTES;
    float2* lnx = reinterpret_cast<float2*>(reinterpret_cast<unsigned char*>(smem + LWS_STAGES * LWS_STAGE_ELEMS) + 8 * LWS_STG_BYTES);
    if constexpr (K == 320) {                                     // waves w and w + 4 share a SIMD: 48 + 32 channels each
        if (wave < 4) lws_wave<T, MODE, 3, COUNTED, 320, 64, 320>(p, smem, stg, lnx, nblk, wave * 48, t_lo, t_hi, wave, lane);
        else lws_wave<T, MODE, 2, COUNTED, 320, 64, 320>(p, smem, stg, lnx, nblk, 192 + (wave - 4) * 32, t_lo, t_hi, wave, lane);
    } else if constexpr (K == 1280) {                             // 16-token tiles, 16 channels per wave (1 x 40 fragments = 160 registers)
        lws_wave<T, MODE, 1, COUNTED, 1280, 16, 128>(p, smem, stg, lnx, nblk, wave * 16, t_lo, t_hi, wave, lane);
    } else if constexpr (CHB == 256) {                            // K = 640: 32 channels per wave (160 registers of weights)
        lws_wave<T, MODE, 2, COUNTED, 640, 32, 256>(p, smem, stg, lnx, nblk, wave * 32, t_lo, t_hi, wave, lane);
    } else {                                                      // K = 640, N a multiple of 128 only (to_q: N = 640): 16 channels per wave
        lws_wave<T, MODE, 1, COUNTED, 640, 32, 128>(p, smem, stg, lnx, nblk, wave * 16, t_lo, t_hi, wave, lane);
    }
}

template <typename T, int MODE, bool COUNTED, int K, int CHB>
static pf_status lws_launch_c(const LwsParams& p, hipStream_t st) {
    const size_t smem = static_cast<size_t>(LWS_STAGES) * LWS_STAGE_ELEMS * 2 + 8 * LWS_STG_BYTES + LWS_LN_BYTES;
    // per (instantiation, device): the attribute is a property of the function ON a device; the status is checked (a failed
    // call used to surface as an opaque launch failure).  std::atomic: concurrent first launches both set it, harmlessly.
    static std::atomic<unsigned long long> attr_devs{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ULL << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_ws<T, MODE, COUNTED, K, CHB>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
        PF_REQUIRE(e == hipSuccess, "pf_linear_ws: the device does not grant %zu bytes of LDS per workgroup (%s)", smem, hipGetErrorString(e));
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_linear_ws<T, MODE, COUNTED, K, CHB>), dim3(256), dim3(512), smem, st, p);
    PF_CHECK_LAUNCH("pf_linear_ws");
    return PF_OK;
}
template <typename T, int MODE, int K = 320, int CHB = (K == 320 ? 320 : K == 640 ? 256 : 128)>
static pf_status lws_launch(const LwsParams& p, hipStream_t st) {
    static const bool counted = !(getenv("PF_LWS_COUNTED") && atoi(getenv("PF_LWS_COUNTED")) == 0);
    return counted ? lws_launch_c<T, MODE, true, K, CHB>(p, st) : lws_launch_c<T, MODE, false, K, CHB>(p, st);
}

}  // namespace pf

extern "C" int pf_linear_ws_supported(long M, int N, int K, int mode) {
    if (K == 640) {               // 256-channel workgroups: 16-bit and GEGLU outputs (FF1, q | k of the 32^2 level);
        if (N <= 0 || M < 32) return 0;                                        // 128-channel workgroups: 16-bit output (to_q: N = 640).  (fp32 + residual
        if (N % 256 == 0 && N / 256 <= 32 && (mode == PF_LWS_16 || mode == PF_LWS_GEGLU)) return 1;   // was built and measured: 98 vs 89 us for the tile kernel)
        return N % 128 == 0 && N / 128 <= 32 && (mode == PF_LWS_16 || mode == PF_LWS_VT);
    }
    if (K == 1280)                // 16-token tiles, 128-channel workgroups: 16-bit and GEGLU outputs (q | k, to_q, FF1 of the 16^2 level)
        return N > 0 && N % 128 == 0 && N / 128 <= 128 && M >= 16 && (mode == PF_LWS_16 || mode == PF_LWS_GEGLU || mode == PF_LWS_VT);
    if (K != 320 || N <= 0 || N % 320 != 0 || M < 64) return 0;
    const int nb = N / 320;
    if (nb > 32) return 0;
    if (mode == PF_LWS_QKV && nb != 3) return 0;
    if (mode == PF_LWS_F32_LN && nb != 1) return 0;
    return 1;
}

extern "C" pf_status pf_linear_ws(const pf_linear_ws_desc* d, void* stream) {
    using namespace pf;
    PF_REQUIRE(d != nullptr, "pf_linear_ws: null descriptor");
    PF_REQUIRE(d->a && d->w && (d->out || d->mode == PF_LWS_VT), "pf_linear_ws: null operand");
    PF_REQUIRE(pf_linear_ws_supported(d->M, d->N, d->K, d->mode), "pf_linear_ws: needs K == 320, N a multiple of 320 (<= 32 blocks; q|k|v: N == 960), M >= 64 -- or K == 640 with N a multiple of 256 (16-bit / GEGLU output) or of 128 (16-bit output), or K == 1280 with N a multiple of 128 (16-bit / GEGLU output) (got M %ld N %d K %d mode %d)",
               static_cast<long>(d->M), d->N, d->K, d->mode);
    PF_REQUIRE(d->mode >= PF_LWS_16 && d->mode <= PF_LWS_VT, "pf_linear_ws: unknown mode %d", d->mode);
    PF_REQUIRE(d->a_ld >= d->K && d->a_ld % 8 == 0 && aligned16(d->a) && aligned16(d->w) && (!d->out || aligned16(d->out)), "pf_linear_ws: operands must be 16-byte aligned, a_ld a multiple of 8");
    PF_REQUIRE(static_cast<long>(d->M) * d->a_ld * 2 < (2L << 30), "pf_linear_ws: activation matrix must be smaller than 2 GiB");
    PF_REQUIRE(!d->bias || aligned16(d->bias), "pf_linear_ws: bias must be 16-byte aligned");
    const int n_store = d->mode == PF_LWS_GEGLU ? d->N / 2 : d->mode == PF_LWS_QKV ? 640 : d->N;
    const bool f32out = d->mode == PF_LWS_F32 || d->mode == PF_LWS_F32_LN;
    PF_REQUIRE(d->mode == PF_LWS_VT || (d->out_ld >= n_store && d->out_ld % (f32out ? 4 : 8) == 0), "pf_linear_ws: out_ld %d does not hold %d columns in 16-byte chunks", d->out_ld, n_store);
    if (f32out)
        PF_REQUIRE(!d->residual || (aligned16(d->residual) && d->res_ld >= d->N && d->res_ld % 4 == 0), "pf_linear_ws: residual must be 16-byte aligned fp32 rows");
    else
        PF_REQUIRE(!d->residual, "pf_linear_ws: a residual needs mode PF_LWS_F32");
    if (d->mode == PF_LWS_F32_LN)
        PF_REQUIRE(d->N == 320 && d->ln_gamma && d->ln_beta && d->ln_out && aligned16(d->ln_gamma) && aligned16(d->ln_beta) && aligned16(d->ln_out) &&
                   d->ln_ld >= 320 && d->ln_ld % 8 == 0 && d->ln_eps > 0.f,
                   "pf_linear_ws: the LayerNorm mode needs N == 320 (a whole row per workgroup), gamma / beta / ln_out 16-byte aligned");
    if (d->mode == PF_LWS_VT) {
        const int tok = d->K == 1280 ? 16 : d->K == 640 ? 32 : 64;
        PF_REQUIRE(d->out_vt && aligned16(d->out_vt) && d->rows_per_batch > 0 && d->rows_per_batch % tok == 0 && d->M % d->rows_per_batch == 0 &&
                   d->vt_ld >= d->rows_per_batch && d->vt_ld % 8 == 0 && d->vt_bs % 8 == 0,
                   "pf_linear_ws: the transposed mode needs out_vt with 16-byte aligned key runs and batches of a multiple of %d tokens", tok);
    }
    if (d->mode == PF_LWS_QKV)
        PF_REQUIRE(d->out_vt && aligned16(d->out_vt) && d->rows_per_batch > 0 && d->rows_per_batch % 64 == 0 && d->M % d->rows_per_batch == 0 &&
                   d->vt_ld >= d->rows_per_batch && d->vt_ld % 8 == 0 && d->vt_bs % 8 == 0,
                   "pf_linear_ws: q|k|v mode needs out_vt with 16-byte aligned key runs and batches of a multiple of 64 tokens");
    LwsParams p;
    p.a = static_cast<const unsigned short*>(d->a); p.a_ld = d->a_ld;
    p.w = static_cast<const unsigned short*>(d->w);
    p.bias = d->bias; p.residual = d->residual; p.res_ld = d->res_ld;
    p.out = d->out; p.out_ld = d->out_ld;
    p.out_vt = static_cast<unsigned short*>(d->out_vt); p.vt_ld = d->vt_ld; p.rows_per_batch = d->rows_per_batch; p.vt_bs = d->vt_bs;
    p.ln_gamma = d->ln_gamma; p.ln_beta = d->ln_beta; p.ln_eps = d->ln_eps; p.ln_out = static_cast<unsigned short*>(d->ln_out); p.ln_ld = d->ln_ld;
    p.M = d->M; p.N = d->N;
    const bool k640 = d->K == 640, k1280 = d->K == 1280;
    const bool chb128 = k640 && (d->mode == PF_LWS_VT || !(d->N % 256 == 0 && (d->mode == PF_LWS_16 || d->mode == PF_LWS_GEGLU)));
    p.nblocks = d->N / (chb128 || k1280 ? 128 : k640 ? 256 : 320);
    p.splits_per_xcd = 32 / p.nblocks;
    p.flat_splits = 8 * p.splits_per_xcd * p.nblocks >= 200 ? 0 : std::max(1, 256 / p.nblocks);
    p.ntiles = static_cast<int>(cdiv(d->M, k1280 ? 16 : k640 ? 32 : 64));
    p.a_bytes = static_cast<unsigned>(static_cast<long>(d->M) * d->a_ld * 2);
    hipStream_t st = as_stream(stream);
#define PF_LWS_MODE(MODE) PF_DISPATCH_16(d->dtype, "pf_linear_ws", return (lws_launch<T, MODE>(p, st)))
    if (k1280 && d->mode == PF_LWS_VT) PF_DISPATCH_16(d->dtype, "pf_linear_ws", return (lws_launch<T, LWS_VT, 1280>(p, st)));
    if (chb128 && d->mode == PF_LWS_VT) PF_DISPATCH_16(d->dtype, "pf_linear_ws", return (lws_launch<T, LWS_VT, 640, 128>(p, st)));
    if (k1280) {
        if (d->mode == PF_LWS_GEGLU) PF_DISPATCH_16(d->dtype, "pf_linear_ws", return (lws_launch<T, LWS_GEGLU, 1280>(p, st)));
        PF_DISPATCH_16(d->dtype, "pf_linear_ws", return (lws_launch<T, LWS_16, 1280>(p, st)));
    }
    if (chb128) PF_DISPATCH_16(d->dtype, "pf_linear_ws", return (lws_launch<T, LWS_16, 640, 128>(p, st)));
    if (k640) {
        if (d->mode == PF_LWS_GEGLU) PF_DISPATCH_16(d->dtype, "pf_linear_ws", return (lws_launch<T, LWS_GEGLU, 640>(p, st)));
        PF_DISPATCH_16(d->dtype, "pf_linear_ws", return (lws_launch<T, LWS_16, 640>(p, st)));
    }
    switch (d->mode) {
        case PF_LWS_16: PF_LWS_MODE(LWS_16);
        case PF_LWS_F32: PF_LWS_MODE(LWS_F32);
        case PF_LWS_GEGLU: PF_LWS_MODE(LWS_GEGLU);
        case PF_LWS_F32_LN: PF_LWS_MODE(LWS_F32_LN);
        case PF_LWS_VT: PF_LWS_MODE(LWS_VT);
        default: PF_LWS_MODE(LWS_QKV);
    }
#undef PF_LWS_MODE
    return PF_OK;
}
