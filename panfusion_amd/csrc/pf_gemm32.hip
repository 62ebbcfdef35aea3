// Implicit-GEMM convolution / GEMM on v_mfma_f32_32x32x16 wave tiles for gfx950 (CDNA4) -- the long-K 3x3 convolutions of the
// 64 x 64 and 32 x 32 UNet levels (reference call sites models/pano/MVGenModel.py:102-144,174-198,224-277: diffusers ResnetBlock2D
// conv1 / conv2 and Upsample2D.conv behind cuDNN).
//
//   out[m][n] = sum_k A[m][k] * W[n][k]  (+ bias[n]) (+ rowvec[img(m)][n]) (+ residual[m][n])       (same contract as pf_gemm.hip)
//
// Why a second tile kernel (round 6).  The 16x16x32 kernel of pf_gemm.hip (256 x 160 block, 64 x 80 wave tiles) needs 18 ds_read_b128
// and 40 MFMA issues per wave and 64-wide K stage; inside its K loop the matrix pipes are 60 % occupied: a wave's
// in-order issue stream carries 40 MFMAs + 18 fragment reads + 7 LDS-DMA pieces per 1280 clocks of matrix time, and
// v_mfma_f32_16x16x32 itself issues at ~17 clocks instead of 16 (MI355X_MICROARCH.md, cycle constants).  Here:
//   * block tile 256 pixels x 320 channels, 512 threads = 8 waves (4 x 2), wave tile 64 x 160 = 2 x 5 MFMA tiles of 32 x 32, fp32
//     accumulators = 160 VGPRs; per 32-wide K stage a wave issues 20 MFMAs (640 clocks of its SIMD's matrix pipe, 1280 with its partner
//     wave) against 14 ds_read_b128 and 5 LDS-DMA instructions: half the MFMA issues, 22 % fewer fragment bytes and 31 % less DMA
//     ingest per FLOP than the 16x16 kernel;
//   * K stages are 32 wide (LDS rows of 64 B), four ring slots of 36 KB: stage it+4 is requested into the slot of stage it right
//     behind the mid-step barrier of step it -- THREE K steps of DMA look-ahead (the 64-wide three-slot ring of pf_gemm.hip has two
//     steps of twice the length: the same time);
//   * the weight tile is the MFMA A operand (rows = output channels), the activation tile the B operand (columns = pixels): a lane
//     ends up with one pixel (lane & 31) and, per 32-channel MFMA tile, four quads of 4 consecutive channels 8 g + 4 (lane >> 5);
//   * LDS rows: 64 B, the 16-byte chunk index XOR-swizzled with (row >> 2) & 3 on the DMA's SOURCE side (the DMA writes lane-linear):
//     a ds_read_b128 fragment read (16-lane groups {0-3, 12-15, 20-27} ... of MI355X_MICROARCH.md, LDS table) is conflict free;
//   * epilogue straight from the fragments: fp32 rows as 16-byte quads; 16-bit rows after a v_permlane32_swap that gives every lane
//     8 consecutive channels (16 bytes).  No LDS staging: the ring is free for the next tile's first four stages at once.
// Addressing (scalar buffer descriptors + per-thread byte offsets + per-stage scalar offsets, zero padding as out-of-range offsets,
// stride 2, fused nearest x2 upsampling, channel concat of two sources, virtual circular padding), the XCD-aware tile order, split K
// over blockIdx.y and the GroupNorm-moment by-product follow pf_gemm.hip.
#include "pf_common.h"
#include "pf_gemm_params.h"
#include <stdlib.h>
#include <type_traits>

namespace pf {

namespace g32 {
constexpr int BM = 256, BN = 320, BK = 32, SLOTS = 4, NT = 512;
constexpr int STAGE = (BM + BN) * BK;            // 16-bit elements per ring slot (36 864 bytes)
constexpr int NPIECE = 5;                        // LDS-DMA instructions per wave and stage: 2 activation passes, 2 weight passes, 1 half pass
constexpr int MI = 2, NJ = 5;                    // 32 x 32 MFMA tiles of a wave: 64 pixels x 160 channels
}


namespace g32 {

__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {       // a[lanes 32..63] <-> b[lanes 0..31]
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xor16_sum(float v) {                      // v + (the value 16 lanes away inside a 32-lane half)
    return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
__device__ __forceinline__ float row16_sum(float v) {                      // all-reduce over the 16 lanes of a DPP row
#define PF_ROR(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xF, 0xF, false))
    v += PF_ROR(8);
    v += PF_ROR(4);
    v += PF_ROR(2);
    v += PF_ROR(1);
#undef PF_ROR
    return v;
}

// Fragment layout of a wave tile (64 pixels x 160 channels): acc[i][j][4 g + e] = pixel mw + 32 i + (lane & 31),
// channel nw + 32 j + 8 g + 4 (lane >> 5) + e.

// GroupNorm moments of the finished fp32 values (bias / time-embedding row already in the accumulators): per column PAIR the sum and
// the sum of squares over the wave's 64 rows -> gn_partial[mw / 64][2][N / 2] (pf_conv_desc.gn_partial with R = 64; fixed order).
__device__ __forceinline__ void gn_moments(const GemmParams& p, const f32x16 (&acc)[MI][NJ], int mw, int nw, int lane) {
    if (mw >= p.M) return;
    const int NP = p.N >> 1;
    float* base = p.gn_partial + static_cast<long>(mw / 64) * 2 * NP;
    const int cq = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const float x0 = acc[i][j][4 * g], x1 = acc[i][j][4 * g + 1], x2 = acc[i][j][4 * g + 2], x3 = acc[i][j][4 * g + 3];
                s0 += x0 + x1; s1 += x2 + x3;
                q0 += x0 * x0 + x1 * x1; q1 += x2 * x2 + x3 * x3;
            }
            s0 = xor16_sum(row16_sum(s0)); s1 = xor16_sum(row16_sum(s1));
            q0 = xor16_sum(row16_sum(q0)); q1 = xor16_sum(row16_sum(q1));
            if ((lane & 31) == 0) {
                const int n = nw + 32 * j + 8 * g + cq;
                *reinterpret_cast<float2*>(base + (n >> 1)) = float2{s0, s1};
                *reinterpret_cast<float2*>(base + NP + (n >> 1)) = float2{q0, q1};
            }
        }
    }
}

// fp32 rows (the mixed scheme's residual streams): out = acc (+ fp32 residual), 16-byte quads straight from the fragments.  The residual
// of a unit (j, i) = four quads is requested two units ahead (three buffers: 8 loads per lane in flight).
template <bool RES>
__device__ __forceinline__ void store_f32(const GemmParams& p, long bz, const f32x16 (&acc)[MI][NJ], int mw, int nw, int lane) {
    const int rl = lane & 31, cq = 4 * (lane >> 5);
    const float* resp = RES ? static_cast<const float*>(p.residual) + bz * p.res_bs : nullptr;
    float* outp = static_cast<float*>(p.out) + bz * p.out_bs;
    bool live[MI];
    long roff[MI], ooff[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mw + 32 * i + rl;
        live[i] = m < p.M;
        const int mc = min(m, p.M - 1);
        roff[i] = static_cast<long>(mc) * p.res_ld + nw + cq;
        ooff[i] = static_cast<long>(mc) * p.out_ld + nw + cq;
    }
    constexpr int NU = NJ * MI;
    float4 r[3][4];
    auto request = [&](auto u_tag) __attribute__((always_inline)) {
        constexpr int u = decltype(u_tag)::value;
        if constexpr (RES && u < NU) {
            constexpr int j = u / MI, i = u % MI;
#pragma unroll
            for (int g = 0; g < 4; ++g) r[u % 3][g] = *reinterpret_cast<const float4*>(resp + roff[i] + 32 * j + 8 * g);
        }
    };
    auto finish = [&](auto u_tag) __attribute__((always_inline)) {
        constexpr int u = decltype(u_tag)::value, j = u / MI, i = u % MI;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v = float4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            if constexpr (RES) { const float4 x = r[u % 3][g]; v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w; }
            if (live[i]) *reinterpret_cast<float4*>(outp + ooff[i] + 32 * j + 8 * g) = v;
        }
    };
#define PF_U(n) std::integral_constant<int, n>()
    request(PF_U(0)); request(PF_U(1));
    request(PF_U(2)); finish(PF_U(0));  request(PF_U(3)); finish(PF_U(1));  request(PF_U(4)); finish(PF_U(2));
    request(PF_U(5)); finish(PF_U(3));  request(PF_U(6)); finish(PF_U(4));  request(PF_U(7)); finish(PF_U(5));
    request(PF_U(8)); finish(PF_U(6));  request(PF_U(9)); finish(PF_U(7));  finish(PF_U(8)); finish(PF_U(9));
    static_assert(NU == 10, "unrolled by hand");
}

// 16-bit rows: out = round16(acc (+ 16-bit residual)).  The four quads of a unit (j, i) are rounded, then v_permlane32_swap between
// lanes l and l + 32 turns (quad g, quad g + 2) into 8 consecutive channels per lane: two 16-byte stores per unit (lanes < 32: channels
// 0-7 and 8-15 of the 32-channel MFMA tile, lanes >= 32: 16-23 and 24-31).  A residual is loaded in that store layout (two 16-byte
// loads, two units ahead) and brought into the fragment layout by the same swap (an involution).
template <typename T, bool RES>
__device__ __forceinline__ void store_16(const GemmParams& p, long bz, const f32x16 (&acc)[MI][NJ], int mw, int nw, int lane) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const int rl = lane & 31, hq = (lane >> 5) * 16;
    const unsigned short* resp = RES ? static_cast<const unsigned short*>(p.residual) + bz * p.res_bs : nullptr;
    unsigned short* outp = static_cast<unsigned short*>(p.out) + bz * p.out_bs;
    bool live[MI];
    long roff[MI], ooff[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mw + 32 * i + rl;
        live[i] = m < p.M;
        const int mc = min(m, p.M - 1);
        roff[i] = static_cast<long>(mc) * p.res_ld + nw + hq;
        ooff[i] = static_cast<long>(mc) * p.out_ld + nw + hq;
    }
    constexpr int NU = NJ * MI;
    u32x4 r[3][2];
    auto request = [&](auto u_tag) __attribute__((always_inline)) {
        constexpr int u = decltype(u_tag)::value;
        if constexpr (RES && u < NU) {
            constexpr int j = u / MI, i = u % MI;
            r[u % 3][0] = *reinterpret_cast<const u32x4*>(resp + roff[i] + 32 * j);
            r[u % 3][1] = *reinterpret_cast<const u32x4*>(resp + roff[i] + 32 * j + 8);
        }
    };
    auto finish = [&](auto u_tag) __attribute__((always_inline)) {
        constexpr int u = decltype(u_tag)::value, j = u / MI, i = u % MI;
        float v[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[g][e] = acc[i][j][4 * g + e];
        if constexpr (RES) {
            // store layout -> fragment layout: chunk c = (quad c, quad c + 2) after the swap
            unsigned q[4][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                q[c][0] = r[u % 3][c][0]; q[c][1] = r[u % 3][c][1]; q[c + 2][0] = r[u % 3][c][2]; q[c + 2][1] = r[u % 3][c][3];
                swap32(q[c][0], q[c + 2][0]);
                swap32(q[c][1], q[c + 2][1]);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v[g][0] += to_f32<T>(static_cast<unsigned short>(q[g][0] & 0xFFFFu)); v[g][1] += to_f32<T>(static_cast<unsigned short>(q[g][0] >> 16));
                v[g][2] += to_f32<T>(static_cast<unsigned short>(q[g][1] & 0xFFFFu)); v[g][3] += to_f32<T>(static_cast<unsigned short>(q[g][1] >> 16));
            }
        }
        unsigned q[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u16x4 w4 = {from_f32<T>(v[g][0]), from_f32<T>(v[g][1]), from_f32<T>(v[g][2]), from_f32<T>(v[g][3])};
            const u32x2 w2 = __builtin_bit_cast(u32x2, w4);
            q[g][0] = w2[0]; q[g][1] = w2[1];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            swap32(q[c][0], q[c + 2][0]);
            swap32(q[c][1], q[c + 2][1]);
        }
        if (live[i]) {
            *reinterpret_cast<u32x4*>(outp + ooff[i] + 32 * j) = u32x4{q[0][0], q[0][1], q[2][0], q[2][1]};
            *reinterpret_cast<u32x4*>(outp + ooff[i] + 32 * j + 8) = u32x4{q[1][0], q[1][1], q[3][0], q[3][1]};
        }
    };
    request(PF_U(0)); request(PF_U(1));
    request(PF_U(2)); finish(PF_U(0));  request(PF_U(3)); finish(PF_U(1));  request(PF_U(4)); finish(PF_U(2));
    request(PF_U(5)); finish(PF_U(3));  request(PF_U(6)); finish(PF_U(4));  request(PF_U(7)); finish(PF_U(5));
    request(PF_U(8)); finish(PF_U(6));  request(PF_U(9)); finish(PF_U(7));  finish(PF_U(8)); finish(PF_U(9));
#undef PF_U
}

}  // namespace g32

template <typename T, bool STATS>
__global__ __launch_bounds__(512, 1) void k_conv_gemm32(const GemmParams p) {
    using namespace g32;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];

    const int ntile_total = p.mtiles * p.ntiles;
    const long bz = blockIdx.z;
    const unsigned short* a0 = p.a0 + bz * p.a_bs;
    const unsigned short* a1 = p.a1 ? p.a1 + bz * p.a_bs : nullptr;
    const unsigned short* wg = p.w + bz * p.w_bs;
    int m0 = 0, n0 = 0;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // DMA passes of 128 tile rows: thread -> (row lrow, 16-byte chunk t & 3); the chunk a lane FETCHES is the one its lane-linear
    // LDS position holds under the swizzle
#ifdef PF_ABL_FULLLINE       /* timing-only (wrong results): the same bytes per DMA instruction as 8 rows x 128 contiguous bytes instead of 16 rows x 64,
                               * the scalar offset advancing 128 bytes per stage: every 128-byte line is requested ONCE instead of by two consecutive stages */
    const int lrow = (t >> 3) * 2;
    const int lchunk8 = (t & 7) * 8;
    const int hrow = wave * 8 + (lane >> 3) * 2;
    const int hchunk8 = (lane & 7) * 8;
#else
    const int lrow = t >> 2;
    const int lchunk8 = ((t & 3) ^ ((lrow >> 2) & 3)) * 8;
    const int hrow = wave * 8 + (lane >> 2);                      // half pass (lanes 0..31 of every wave): weight rows 256 + hrow
    const int hchunk8 = ((lane & 3) ^ ((hrow >> 2) & 3)) * 8;
#endif

    int a_img[2], a_y[2], a_x[2];
    constexpr unsigned OOB = 0x80000000u;                         // launcher guarantees tensors < 2 GiB
    auto uniform_ptr = [](const unsigned short* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
        const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
        return reinterpret_cast<unsigned short*>(static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi) << 32));
    };
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wg), 0, __builtin_amdgcn_readfirstlane(p.w_bytes), 0x00020000);
    unsigned w_off[3];
    const int Ctot = p.c0 + p.c1;
    const int Hl = p.h_in << p.up, Wl = (p.w_in + 2 * p.wrap) << p.up;

    // K walk in 32-element blocks (wave-uniform): block = (tap, source, 32 channels).  kb_per_split counts 64-element blocks.
    const int nkb = p.K / BK;
    const int kb0 = blockIdx.y * p.kb_per_split * 2;
    const int kb1 = min(nkb, kb0 + p.kb_per_split * 2);
    const int n_it = kb1 - kb0;
    int kg = 0, tap = 0, cc = 0;
    unsigned a_off[2];
    bool seg1 = false;
    auto set_segment = [&]() __attribute__((always_inline)) {
        const int ky = p.ksize == 3 ? tap / 3 : 0, kx = p.ksize == 3 ? tap - 3 * ky : 0;
        seg1 = __builtin_amdgcn_readfirstlane(cc >= p.c0 ? 1 : 0) != 0;
        const int ld = seg1 ? p.a1_ld : p.a0_ld;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int yi = a_y[i] + ky, xi = a_x[i] + kx;
            // (no short-circuit: hipcc keeps && as nested exec-masked branches and sinks the uniform K-walk updates into them)
            const bool ok = (static_cast<unsigned>(yi) < static_cast<unsigned>(Hl)) & (static_cast<unsigned>(xi) < static_cast<unsigned>(Wl));
            int sx = (xi >> p.up) - p.wrap;
            sx += sx < 0 ? p.w_in : 0;
            sx -= sx >= p.w_in ? p.w_in : 0;
            const int pix = (a_img[i] * p.h_in + (yi >> p.up)) * p.w_in + sx;
            a_off[i] = ok ? static_cast<unsigned>(pix * ld + lchunk8) * 2u : OOB;
        }
    };
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
            const unsigned m = static_cast<unsigned>(m0 + lrow);
            int img = static_cast<int>(m / static_cast<unsigned>(p.rows_per_img));
            const unsigned rem = m - static_cast<unsigned>(img) * static_cast<unsigned>(p.rows_per_img);
            int yo = static_cast<int>(rem / static_cast<unsigned>(p.w_out));
            int xo = static_cast<int>(rem) - yo * p.w_out;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = m0 + i * 128 + lrow < p.M;
                a_img[i] = ok ? img : 0;
                a_y[i] = ok ? yo * p.stride - p.pad : -(1 << 20);
                a_x[i] = (xo + p.crop) * p.stride - p.pad;
                xo += p.adv_x;
                if (xo >= p.w_out) { xo -= p.w_out; ++yo; }
                yo += p.adv_y;
                if (yo >= p.h_out) { yo -= p.h_out; ++img; }
                img += p.adv_img;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + j * 128 + lrow;
            w_off[j] = n < p.N ? static_cast<unsigned>(n * p.K + lchunk8) * 2u : OOB;
        }
        {
            const int n = n0 + 256 + hrow;
            w_off[2] = n < p.N ? static_cast<unsigned>(n * p.K + hchunk8) * 2u : OOB;
        }
        // (integer division runs on the vector ALU: without the readfirstlane hipcc treats the whole K walk as divergent and puts
        // every stage_end() under exec masks)
        kg = kb0 * BK;
        tap = __builtin_amdgcn_readfirstlane(kg / Ctot);
        cc = __builtin_amdgcn_readfirstlane(kg - tap * Ctot);
        set_segment();
    };

    int st_soff_a = 0, st_soff_w = 0;
    auto stage_begin = [&]() __attribute__((always_inline)) {
#ifdef PF_ABL_FULLLINE
        st_soff_a = __builtin_amdgcn_readfirstlane(((seg1 ? cc - p.c0 : cc) * 4) % ((seg1 ? p.c1 : p.c0) * 2));
        st_soff_w = __builtin_amdgcn_readfirstlane((kg * 4) % (p.K * 2));
#else
        st_soff_a = __builtin_amdgcn_readfirstlane((seg1 ? cc - p.c0 : cc) * 2);
        st_soff_w = __builtin_amdgcn_readfirstlane(kg * 2);
#endif
        return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(seg1 ? a1 : a0), 0,
                                                 __builtin_amdgcn_readfirstlane(seg1 ? p.a1_bytes : p.a0_bytes), 0x00020000);
    };
    auto stage_end = [&]() __attribute__((always_inline)) {
        kg = __builtin_amdgcn_readfirstlane(kg + BK);
        cc = __builtin_amdgcn_readfirstlane(cc + BK);
        tap = __builtin_amdgcn_readfirstlane(tap);
        if (cc == Ctot) { cc = 0; ++tap; set_segment(); }
        else if (cc == p.c0) set_segment();
    };
    auto lds_dma = [&](const __amdgpu_buffer_rsrc_t& r, unsigned short* dst, unsigned voff, int soff) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
    };
    auto dma_piece = [&](const __amdgpu_buffer_rsrc_t& rs_a, int slot, int k) __attribute__((always_inline)) {
        unsigned short* As = smem + slot * STAGE;
        unsigned short* Ws = As + BM * BK;
        if (k < 2) lds_dma(rs_a, As + (k * 128 + wave * 16) * BK, a_off[k], st_soff_a);
        else if (k < 4) lds_dma(rs_w, Ws + ((k - 2) * 128 + wave * 16) * BK, w_off[k - 2], st_soff_w);
        else if (lane < 32) lds_dma(rs_w, Ws + (256 + wave * 8) * BK, w_off[2], st_soff_w);
    };
    auto dma_stage = [&](int slot) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs_a = stage_begin();
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) dma_piece(rs_a, slot, k);
        stage_end();
    };

    f32x16 acc[MI][NJ];
    typedef typename Mfma32<T>::frag frag;
#ifdef PF_ABL_MFMA16          /* timing-only (wrong results): every v_mfma_f32_32x32x16 replaced by two v_mfma_f32_16x16x32 on the same fragment
                               * registers -- the same FLOPs and operand traffic in the other instruction shape: what the MFMA shape costs under the power cap */
    auto mfma = [&](frag a, frag b, f32x16& c) __attribute__((always_inline)) {
        f32x4 lo = {c[0], c[1], c[2], c[3]}, hi = {c[8], c[9], c[10], c[11]};
        if constexpr (std::is_same<T, F16>::value) {
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, lo, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, hi, 0, 0, 0);
        } else {
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, lo, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, hi, 0, 0, 0);
        }
        c[0] = lo[0]; c[1] = lo[1]; c[2] = lo[2]; c[3] = lo[3]; c[8] = hi[0]; c[9] = hi[1]; c[10] = hi[2]; c[11] = hi[3];
    };
#else
    auto mfma = [&](frag a, frag b, f32x16& c) __attribute__((always_inline)) { c = Mfma32<T>::run(a, b, c); };
#endif
    // fragment reads: lane -> tile row (lane & 31), 16-byte chunk 2 s + (lane >> 5) of the 64-byte row, swizzled
    const int rowl = lane & 31, kh = lane >> 5, sw = (rowl >> 2) & 3;
    const int fa_base0 = (wm * 64 + rowl) * BK + ((kh ^ sw) << 3);                     // + slot * STAGE + i * 32 * BK; s = 1: ^ 16
    const int fw_base0 = BM * BK + (wn * 160 + rowl) * BK + ((kh ^ sw) << 3);
    frag fa0[MI], fw0[NJ], fa1[MI], fw1[NJ];
    auto load_frags = [&](int slot, auto s_tag, frag (&fa)[MI], frag (&fw)[NJ]) __attribute__((always_inline)) {
        constexpr int s = decltype(s_tag)::value;
        const unsigned short* ap = smem + slot * STAGE + (s ? (fa_base0 ^ 16) : fa_base0);
        const unsigned short* wp = smem + slot * STAGE + (s ? (fw_base0 ^ 16) : fw_base0);
#ifndef PF_ABL_NOLDS
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(ap + i * 32 * BK));
#pragma unroll
        for (int j = 0; j < NJ; ++j) fw[j] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(wp + j * 32 * BK));
#endif
    };
    const std::integral_constant<int, 0> S0;
    const std::integral_constant<int, 1> S1;
#ifdef PF_ABL_NOLDS           /* timing-only: no fragment reads; the MFMAs run on (real, fixed) weight values fetched once */
#pragma unroll
    for (int i = 0; i < MI; ++i) { fa0[i] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(wg + (lane + 64 * i) * 8)); fa1[i] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(wg + (lane + 64 * (i + 2)) * 8)); }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { fw0[j] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(wg + (lane + 64 * (j + 4)) * 8)); fw1[j] = __builtin_bit_cast(frag, *reinterpret_cast<const u16x8*>(wg + (lane + 64 * (j + 9)) * 8)); }
#endif

    auto issue_prologue = [&]() __attribute__((always_inline)) {
        dma_stage(0);
        if (n_it > 1) dma_stage(1);
        if (n_it > 2) dma_stage(2);
        if (n_it > 3) dma_stage(3);
    };
    const std::true_type YES;
    const std::false_type NO;
    int cur = 0;
    // One K step (see the header).  MORE: a next stage exists (wait for it, barrier, prefetch its first fragments); DMA: stage it+4
    // exists (its pieces go out between the MFMAs of the second half, into the slot retired by this step's barrier); WAIT: DMA
    // instructions that may stay in flight at the mid-step wait (the stages behind it+1).
    auto step = [&](auto more_tag, auto dma_tag, auto wait_tag) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_tag)::value, DMA = decltype(dma_tag)::value;
        constexpr int WAIT = decltype(wait_tag)::value;
        const int nxt = (cur + 1) & 3;
#pragma unroll
        for (int idx = 0; idx < MI * NJ; ++idx) {
            const int j = idx / MI, i = idx % MI;
            mfma(fw0[j], fa0[i], acc[i][j]);
            if (idx == 1) {
                __builtin_amdgcn_sched_barrier(0);
                load_frags(cur, S1, fa1, fw1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) {
            if constexpr (WAIT == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if constexpr (WAIT == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
            else if constexpr (WAIT == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
            else static_assert(WAIT == 0, "add the immediate");
#ifndef PF_ABL_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
        }
        __amdgpu_buffer_rsrc_t rs_a = rs_w;
        if constexpr (DMA) rs_a = stage_begin();
#pragma unroll
        for (int idx = 0; idx < MI * NJ; ++idx) {
            const int j = idx / MI, i = idx % MI;
            mfma(fw1[j], fa1[i], acc[i][j]);
            if (MORE && idx == 1) {
                __builtin_amdgcn_sched_barrier(0);
                load_frags(nxt, S0, fa0, fw0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DMA) {
                if (idx >= 2 && (idx - 2) % 2 == 0 && (idx - 2) / 2 < NPIECE - 1) {       // pieces 0..3 behind MFMAs 2, 4, 6, 8
                    __builtin_amdgcn_sched_barrier(0);
#ifndef PF_ABL_NODMA
                    dma_piece(rs_a, cur, (idx - 2) / 2);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (idx == MI * NJ - 1) {
                    __builtin_amdgcn_sched_barrier(0);
#ifndef PF_ABL_NODMA
                    dma_piece(rs_a, cur, NPIECE - 1);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if constexpr (DMA) stage_end();
        cur = nxt;
    };

    stamp(p, 0);
    // diagnostics (pf_debug_gemm_profile, tools/gemm_bench.py --phases): wave 0 accumulates the shader clocks of its K loops (slot 4),
    // of everything between a K loop's end and the next tile's operands having landed (slot 5: ring barrier, next tile's DMA issue,
    // epilogue, landing wait) and the tile count (slot 6)
    unsigned long long tk = 0, te = 0, tmark = 0;
    int tcount = 0;
    const bool prof = p.prof != nullptr;
    int tile = blockIdx.x;
    set_tile(tile);
    issue_prologue();
    bool first = true;
    const int cq = 4 * (lane >> 5);
    // epilogue plan (uniform): which operands ride in the accumulators, and which store path the layer's operand mix takes
    const bool fold_bias = p.bias != nullptr && p.splits == 1;
    const bool fold_rv = p.rowvec != nullptr && p.splits == 1 && (p.rows_per_img & 63) == 0;
    const bool plain = !p.geglu && !p.split_out && (p.rowvec == nullptr || fold_rv) && p.splits == 1;
    const bool fast32 = plain && p.out_f32 && (p.out_ld & 3) == 0 && (!p.residual || (p.res_f32 && (p.res_ld & 3) == 0));
    const bool fast16 = plain && !p.out_f32 && (p.out_ld & 7) == 0 && (!p.residual || (!p.res_f32 && (p.res_ld & 7) == 0));
    for (;;) {
        // The accumulators start from the bias (+ the image's time-embedding row where a wave's 64 rows belong to one image): the
        // epilogue then only adds a residual, rounds and stores.  (Split-K slabs start from zero: the reduce kernel adds them.)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 b = float4{0.f, 0.f, 0.f, 0.f};
                const int n = n0 + wn * 160 + 32 * j + 8 * g + cq;
                if (fold_bias) b = *reinterpret_cast<const float4*>(p.bias + n);
                if (fold_rv) {
                    const int img = min(m0 + wm * 64, p.M - 1) / p.rows_per_img;
                    const float4 x = *reinterpret_cast<const float4*>(p.rowvec + static_cast<long>(img) * p.rowvec_ld + n);
                    b.x += x.x; b.y += x.y; b.z += x.z; b.w += x.w;
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) { acc[i][j][4 * g] = b.x; acc[i][j][4 * g + 1] = b.y; acc[i][j][4 * g + 2] = b.z; acc[i][j][4 * g + 3] = b.w; }
            }
        if (first && n_it > 3) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");      // stage 0 landed; 1..3 may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // (later tiles: the epilogue's stores share the counter)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (first) stamp(p, 1);
        if (prof) { const unsigned long long now = __builtin_amdgcn_s_memtime(); if (!first) te += now - tmark; tmark = now; }
        first = false;
        load_frags(0, S0, fa0, fw0);
        cur = 0;
        {
            const std::integral_constant<int, 10> W10;
            const std::integral_constant<int, 5> W5;
            const std::integral_constant<int, 0> W0;
            // later tiles wait with vmcnt(0) above, so every stage of the prologue has landed there; the counted waits stay correct
            for (int it = 0; it + 4 < n_it; ++it) step(YES, YES, W10);
            if (n_it >= 4) step(YES, NO, W10);
            if (n_it >= 3) step(YES, NO, W5);
            if (n_it >= 2) step(YES, NO, W0);
            step(NO, NO, W0);
        }
        stamp(p, 2);
        if (prof) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tk += now - tmark; tmark = now; ++tcount; }

        const int em0 = m0, en0 = n0;
        const int next = tile + static_cast<int>(gridDim.x);
        const bool has_next = next < ntile_total;
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                          // every wave is done reading the operand ring
        if (has_next) { set_tile(next); issue_prologue(); }
        __builtin_amdgcn_sched_barrier(0);
        const int rl = lane & 31;
        if (p.splits > 1) {                                       // fp32 slab straight from the fragments
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = em0 + wm * 64 + i * 32 + rl;
                if (m >= p.M) continue;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n4 = en0 + wn * 160 + j * 32 + 8 * g + cq;
                        if (n4 >= p.N) continue;
                        const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        slab_store(p, ((static_cast<long>(blockIdx.y) * p.batch + bz) * (p.M - p.m_begin) + (m - p.m_begin)) * p.N + n4, v);
                    }
            }
        } else {
            const int mw = em0 + wm * 64, nw = en0 + wn * 160;
            // (laundered lane id: the epilogue's per-thread addressing would otherwise be hoisted out of the tile loop and stay live --
            // in registers the K loop has none to spare of)
            int e_lane = lane;
            asm volatile("" : "+v"(e_lane));
            if constexpr (STATS) gn_moments(p, acc, mw, nw, e_lane);
            if (fast32) {
                if (p.residual) store_f32<true>(p, bz, acc, mw, nw, e_lane);
                else store_f32<false>(p, bz, acc, mw, nw, e_lane);
            } else if (fast16) {
                if (p.residual) store_16<T, true>(p, bz, acc, mw, nw, e_lane);
                else store_16<T, false>(p, bz, acc, mw, nw, e_lane);
            } else {
                const int rl = e_lane & 31, cq = 4 * (e_lane >> 5);
                // everything else (pair output, a row vector over images that are not whole 64-row runs, mixed residual / output
                // types): the per-quad generic store of pf_gemm.hip, minus the operands the accumulators already carry
                GemmParams q = p;
                if (fold_bias) q.bias = nullptr;
                if (fold_rv) q.rowvec = nullptr;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int m = mw + i * 32 + rl;
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n4 = nw + j * 32 + 8 * g + cq;
                            float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                            if (m < p.M && n4 < p.N) epilogue_store<T>(q, bz, m, n4, v);
                        }
                }
            }
        }
        if (!has_next) break;
        tile = next;
    }
    stamp(p, 3);
    if (prof && threadIdx.x == 0) {
        const long b = blockIdx.x + static_cast<long>(gridDim.x) * (blockIdx.y + static_cast<long>(gridDim.y) * blockIdx.z);
        p.prof[b * 32 + 4] = tk;
        p.prof[b * 32 + 5] = te + (__builtin_amdgcn_s_memtime() - tmark);
        p.prof[b * 32 + 6] = static_cast<unsigned long long>(tcount);
        p.prof[b * 32 + 7] = static_cast<unsigned long long>(n_it);
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------

static int tuning32(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <typename T, bool STATS>
static pf_status launch32_s(const GemmParams& gp, int batch, hipStream_t st) {
    using namespace g32;
    GemmParams p = gp;
    p.mtiles = static_cast<int>(cdiv(p.M - p.m_begin, BM));
    p.ntiles = static_cast<int>(cdiv(p.N, BN));
    {
        const int rpp = 128, rem = rpp % p.rows_per_img;           // rows of one DMA pass, as (images, rows, columns)
        p.adv_img = rpp / p.rows_per_img;
        p.adv_y = rem / p.w_out;
        p.adv_x = rem % p.w_out;
    }
    const size_t smem = static_cast<size_t>(SLOTS) * STAGE * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_gemm32<T, STATS>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
        if (e != hipSuccess) { set_error("pf_conv_gemm (32x32): hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return PF_ERR_LAUNCH; }
        attr_set = true;
    }
    static const int cap = tuning32("PF_GEMM32_PERSIST", 256);
    int grid = p.mtiles * p.ntiles;
    if (cap > 0) {
        int gx = std::max(1, cap / (p.splits * batch));
        if (gx >= 8) gx = gx / 8 * 8;
        grid = std::min(grid, gx);
    }
    hipLaunchKernelGGL((k_conv_gemm32<T, STATS>), dim3(grid, p.splits, batch), dim3(NT), smem, st, p);
    PF_CHECK_LAUNCH("pf_conv_gemm (32x32)");
    return PF_OK;
}

// Launches the 32x32 kernel on [gp.m_begin, gp.M) (the split-K reduce, if any, is the caller's: pf_gemm.hip owns k_splitk_reduce).
pf_status launch_gemm32(const GemmParams& gp, int dtype, int batch, hipStream_t st) {
    PF_DISPATCH_16(dtype, "pf_conv_gemm", return gp.gn_partial ? launch32_s<T, true>(gp, batch, st) : launch32_s<T, false>(gp, batch, st));
    return PF_OK;
}

}  // namespace pf
