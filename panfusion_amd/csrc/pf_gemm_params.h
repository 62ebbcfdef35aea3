// Shared by the tile GEMM kernels (pf_gemm.hip: 16x16x32 MFMA tiles; pf_gemm32.hip: 32x32x16 MFMA tiles): kernel parameters, phase stamps,
// the per-quad generic epilogue and the split-K slab store.
#pragma once
#include "pf_common.h"

namespace pf {

struct GemmParams {
    const unsigned short* a0; const unsigned short* a1;
    int c0, c1, a0_ld, a1_ld;
    int h_in, w_in, h_out, w_out;
    int ksize, stride, pad, up;
    int wrap, crop;              // virtual circular padding of the input WIDTH by `wrap` columns (pre-upsample) and output columns
                                 // cropped by `crop` on both sides: pad_pano -> conv -> unpad_pano of the panorama branch without
                                 // the padded copies (utils/pano.py:74-105, MVGenModel.py:98-144,224-294)
    const unsigned short* w;
    int M, N, K;                 // K = ksize*ksize*(c0+c1); a launch covers output rows [m_begin, M)
    int m_begin;
    int rows_per_img;
    const float* bias; const float* rowvec; int rowvec_ld;
    const void* residual; int res_ld; int res_f32;   // residual: 16-bit T, or fp32 (the fp32 residual stream)
    void* out; int out_ld; int out_f32; int geglu;
    int split_out;               // 16-bit pair output [M][hi(N) | lo(N)] (PF_EPILOGUE_SPLIT): the A operand of a split-precision GEMM
    long a_bs, w_bs, out_bs, res_bs;
    int mtiles, ntiles;
    int splits, kb_per_split;      // split-K: blockIdx.y walks K-blocks [y*kb_per_split, ...)
    float* partial;                // [split][batch][M][N] fp32 when splits > 1
    int* tickets;                  // splits > 1: arrival counters, one per (batch, tile), all zero between launches; NULL = the
                                   // slabs are combined by a second kernel (k_splitk_reduce)
    int batch;
    unsigned a0_bytes, a1_bytes, w_bytes;   // extents for the buffer descriptors of the 8-wave kernel (< 2 GiB)
    int adv_img, adv_y, adv_x;              // (image, row, column) advance of one DMA pass of output rows (8-wave kernel)
    unsigned long long* prof;      // diagnostics (pf_debug_gemm_profile): 4 s_memtime stamps per block, or NULL
    int s3;                        // split-precision walk (pf_conv_desc.split3): every 64-element K block of A holds [hi(32) | lo(32)] of 32
                                   // channels, of W [W_hi(32) | W_lo(32)]: a K step multiplies W_hi A_hi + W_hi A_lo + W_lo A_hi
    float* gn_partial;             // [M / gn_rows][2][N / 2] fp32 per-column-pair (sum, sum of squares) of the finished output over
    int gn_rows;                   // the gn_rows fragment rows of one wavefront (GroupNorm moments of the NEXT layer), or NULL
    int fastseg;                   // 1: no upsampling and no circular wrap -- the kernels keep, per staged tile row, the linear index of its top-left input pixel and
                                   // one validity bit per tap row / tap column (seg_pack) instead of (image, y, x), and the per-tap source offsets are an add, a
                                   // multiply-add and a select (set_segment used to be ~20 dependent vector instructions in front of the next stage's DMA: 5-7 % of a
                                   // 3x3 convolution, profiles/r6_seg_ablate.txt).  PF_CONV_FASTSEG=0: the general form everywhere (A/B)
    int subpix;                    // nearest x2 upsampling + 3x3 convolution as FOUR 2x2 convolutions on the low-resolution grid (pf_conv_desc.subpixel):
                                   // blockIdx.z = output phase (a, b) = (z >> 1, z & 1); taps of phase a read input rows y - 1 + a, y + a (columns alike): pad = 1 - phase;
                                   // weights [4][N][2][2][C] (w_bs = N K), a_bs = out_bs = 0; M / h_out / w_out / rows_per_img are those of the LOW-resolution grid and
                                   // the result of low-resolution pixel (q = img h + y, x) goes to output row (2 q + a) 2 w_out + 2 x + b (out_row)
};

// fastseg: (image, top-left input row y, column x) of a staged tile row -> (linear pixel index, validity bits: bit k = input row y + k exists, bit 4 + k = column x + k
// exists, k < 3).  Rows past M carry y = -(1 << 20): no bit set.
__device__ __forceinline__ void seg_pack(int& img_pix, int& y_msk, int x, int h_in, int w_in) {
    const int img = img_pix, y = y_msk;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        m |= (static_cast<unsigned>(y + k) < static_cast<unsigned>(h_in) ? 1u : 0u) << k;
        m |= (static_cast<unsigned>(x + k) < static_cast<unsigned>(w_in) ? 16u : 0u) << k;
    }
    img_pix = static_cast<int>((static_cast<unsigned>(img) * static_cast<unsigned>(h_in) + static_cast<unsigned>(y)) * static_cast<unsigned>(w_in) + static_cast<unsigned>(x));
    y_msk = static_cast<int>(m);
}

// Row of the output tensor that holds the result of GEMM row m (identity except for the sub-pixel phases of an upsampling convolution).
__device__ __forceinline__ long out_row(const GemmParams& p, long bz, int m) {
    if (!p.subpix) return m;
    const int q = m / p.w_out, x = m - q * p.w_out;
    return (2L * q + (bz >> 1)) * (2 * p.w_out) + 2 * x + (bz & 1);
}
// Index of the GroupNorm-moment run that holds GEMM row m (runs of gn_rows rows; sub-pixel phases: the four phases of an image are contiguous).
__device__ __forceinline__ long gn_part(const GemmParams& p, long bz, int m) {
    if (!p.subpix) return m / p.gn_rows;
    const int img = m / p.rows_per_img, rem = m - img * p.rows_per_img, ppi = p.rows_per_img / p.gn_rows;
    return (static_cast<long>(img) * 4 + bz) * ppi + rem / p.gn_rows;
}

// phase stamp of wave 0 / lane 0 of a block: [block][4] = kernel entry, first tile landed, K loop done, exit
#ifdef PF_GEMM_TIMELINE       /* debug build (make timeline): wave 0 of every block stamps the phases of ITS SECOND TILE (steady state of the
                               * persistent loop) into slots 12..: tile begin, operands landed + barrier, stage 2 requested + first fragments,
                               * after every K step (<= 8), after the epilogue -- tools/gemm_bench.py --timeline prints the differences */
#define PF_TL(p, cond, slot) do { if ((cond) && (p).prof && threadIdx.x == 0) { const long b_ = blockIdx.x + static_cast<long>(gridDim.x) * (blockIdx.y + static_cast<long>(gridDim.y) * blockIdx.z); (p).prof[b_ * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define PF_TL(p, cond, slot) do { } while (0)
#endif
__device__ __forceinline__ void stamp(const GemmParams& p, int slot) {
    if (p.prof && threadIdx.x == 0) {
        const long b = blockIdx.x + static_cast<long>(gridDim.x) * (blockIdx.y + static_cast<long>(gridDim.y) * blockIdx.z);
        p.prof[b * 32 + slot] = __builtin_amdgcn_s_memtime();
    }
}

// Epilogue of one output row m, 4 consecutive columns n4..n4+3 (fp32 accumulators v): bias, per-image
// row vector, residual, then either a plain 16-bit / fp32 store or the GEGLU pairing
// (columns interleaved (value, gate): out[m][n4/2 + {0,1}] = value * gelu(gate), transformer.py:8-21).
template <typename T>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, long bz, int m, int n4, float (&v)[4]) {
    if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n4);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (p.rowvec) {
        const int img = m / p.rows_per_img;
        const float4 b = *reinterpret_cast<const float4*>(p.rowvec + static_cast<long>(img) * p.rowvec_ld + n4);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (p.residual && p.res_f32) {
        const float4 r = *reinterpret_cast<const float4*>(static_cast<const float*>(p.residual) + bz * p.res_bs + static_cast<long>(m) * p.res_ld + n4);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    } else if (p.residual) {
        const u16x4 r = *reinterpret_cast<const u16x4*>(static_cast<const unsigned short*>(p.residual) + bz * p.res_bs + static_cast<long>(m) * p.res_ld + n4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += to_f32<T>(r[e]);
    }
    if (p.geglu) {
        unsigned short* o = static_cast<unsigned short*>(p.out) + bz * p.out_bs + out_row(p, bz, m) * p.out_ld + (n4 >> 1);
        typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
        u16x2 w2;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float g = v[2 * e + 1];
            w2[e] = from_f32<T>(geglu_value(v[2 * e], g));
        }
        *reinterpret_cast<u16x2*>(o) = w2;
    } else if (p.split_out) {
        unsigned short* o = static_cast<unsigned short*>(p.out) + bz * p.out_bs + out_row(p, bz, m) * p.out_ld + n4;
        u16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = from_f32<T>(v[e]);
            lo[e] = from_f32<T>(v[e] - to_f32<T>(hi[e]));
        }
        *reinterpret_cast<u16x4*>(o + (pair_off(n4) - n4)) = hi;
        *reinterpret_cast<u16x4*>(o + (pair_off(n4) - n4) + 32) = lo;
    } else if (p.out_f32) {
        float* o = static_cast<float*>(p.out) + bz * p.out_bs + out_row(p, bz, m) * p.out_ld + n4;
        *reinterpret_cast<float4*>(o) = float4{v[0], v[1], v[2], v[3]};
    } else {
        unsigned short* o = static_cast<unsigned short*>(p.out) + bz * p.out_bs + out_row(p, bz, m) * p.out_ld + n4;
        u16x4 w4;
#pragma unroll
        for (int e = 0; e < 4; ++e) w4[e] = from_f32<T>(v[e]);
        *reinterpret_cast<u16x4*>(o) = w4;
    }
}

// fp32 slab store of a split-K partial: WRITE-THROUGH (sc1) when the slabs are combined inside the launch -- the bytes leave
// the XCD's L2 with the store itself, so publishing needs no agent-scope release fence (buffer_wbl2 writes back EVERY dirty
// line of the L2, the concurrently running branch's outputs included: with one fence per K-slice workgroup the step lost
// 4.4 ms, profiles/archive/r3k_ab_splitk.txt).
__device__ __forceinline__ void slab_store(const GemmParams& p, long elem, const f32x4& v) {
    if (p.tickets) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        const unsigned long long a = reinterpret_cast<unsigned long long>(p.partial);
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a >> 32));
        float* base = reinterpret_cast<float*>(static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi) << 32));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xFFFFFFFFu, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, static_cast<int>(static_cast<unsigned>(elem * 4)), 0, /* sc1 */ 16);
    } else {
        *reinterpret_cast<float4*>(p.partial + elem) = float4{v[0], v[1], v[2], v[3]};
    }
}

// pf_gemm32.hip: the 32x32x16-MFMA tile kernel (256 x 320 block) on output rows [m_begin, M); split-K slabs are combined by the caller
pf_status launch_gemm32(const GemmParams& gp, int dtype, int batch, hipStream_t st);

}  // namespace pf
