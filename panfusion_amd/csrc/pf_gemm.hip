// MFMA GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
//   out[m][n] = sum_k A[m][k] * W[n][k]  (+ bias[n]) (+ rowvec[img(m)][n]) (+ residual[m][n])
//
// A is never materialised for convolutions: row m = (image, yo, xo) and K-block kb = (tap, 64
// channels) address 128 contiguous bytes of the NHWC activation (zero padding, optional nearest
// x2 upsampling, optional channel concat of two sources are folded into the address).
//
// Structure: 256 threads = 4 wavefronts (2 x 2), block tile (32*MREP) x (32*NREP), K-step 64,
// v_mfma_f32_16x16x32 (bf16 or f16) with fp32 accumulation.  The weight tile is the MFMA "A"
// operand and the activation tile the "B" operand, so every lane ends up with 4 CONSECUTIVE
// output channels of one output row (8-byte stores, float4 bias loads).  Global -> registers ->
// LDS staging, double buffered (one barrier per K-step, next tile's loads in flight during the
// MFMAs).  LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row>>1)&7 which
// makes the ds_read_b128 fragment reads and the ds_write_b128 staging writes conflict free.
// Workgroup ids are remapped so that the tiles sharing an activation row-panel run on one XCD
// (private L2 per XCD).
//
// Replaces cuDNN / cuBLAS behind diffusers Conv2d / Linear (reference call sites
// models/pano/MVGenModel.py:86-144,174-198,224-294; models/modules/transformer.py:8-74).
#include "pf_common.h"
#include <stdlib.h>

namespace pf {

struct GemmParams {
    const unsigned short* a0; const unsigned short* a1;
    int c0, c1, a0_ld, a1_ld;
    int h_in, w_in, h_out, w_out;
    int ksize, stride, pad, up;
    const unsigned short* w;
    int M, N, K;                 // K = ksize*ksize*(c0+c1)
    int rows_per_img;
    const float* bias; const float* rowvec; int rowvec_ld;
    const unsigned short* residual; int res_ld;
    void* out; int out_ld; int out_f32; int geglu;
    long a_bs, w_bs, out_bs, res_bs;
    int mtiles, ntiles;
    const unsigned short* zeros;   // >= 16 zero bytes in global memory (source of padded rows / ragged tiles)
    int splits, kb_per_split;      // split-K: blockIdx.y walks K-blocks [y*kb_per_split, ...)
    float* partial;                // [split][batch][M][N] fp32 when splits > 1
    int batch;
};

template <typename T> struct Mfma;
template <> struct Mfma<Bf16> {
    typedef __attribute__((ext_vector_type(8))) __bf16 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<F16> {
    typedef __attribute__((ext_vector_type(8))) _Float16 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

__device__ __forceinline__ int lds_off(int row, int chunk) {   // in 16-bit elements
    return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3);
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
    if (p.residual) {
        const u16x4 r = *reinterpret_cast<const u16x4*>(p.residual + bz * p.res_bs + static_cast<long>(m) * p.res_ld + n4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += to_f32<T>(r[e]);
    }
    if (p.geglu) {
        unsigned short* o = static_cast<unsigned short*>(p.out) + bz * p.out_bs + static_cast<long>(m) * p.out_ld + (n4 >> 1);
        typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
        u16x2 w2;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float g = v[2 * e + 1];
            w2[e] = from_f32<T>(v[2 * e] * (0.5f * g * (1.0f + erff(g * 0.70710678118654752440f))));
        }
        *reinterpret_cast<u16x2*>(o) = w2;
    } else if (p.out_f32) {
        float* o = static_cast<float*>(p.out) + bz * p.out_bs + static_cast<long>(m) * p.out_ld + n4;
        *reinterpret_cast<float4*>(o) = float4{v[0], v[1], v[2], v[3]};
    } else {
        unsigned short* o = static_cast<unsigned short*>(p.out) + bz * p.out_bs + static_cast<long>(m) * p.out_ld + n4;
        u16x4 w4;
#pragma unroll
        for (int e = 0; e < 4; ++e) w4[e] = from_f32<T>(v[e]);
        *reinterpret_cast<u16x4*>(o) = w4;
    }
}

template <typename T, int MREP, int NREP>
__global__ __launch_bounds__(256, 2) void k_conv_gemm(const GemmParams p) {
    constexpr int BM = 32 * MREP, BN = 32 * NREP;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short* As = smem;                       // [2][BM][64]
    unsigned short* Bs = smem + 2 * BM * 64;         // [2][BN][64]

    // XCD-aware, bijective remap of the 1-D tile id (8 XCDs, block b runs on XCD b % 8).
    const int ntile_total = p.mtiles * p.ntiles;
    int tid_lin = blockIdx.x;
    {
        const int q = ntile_total / 8, r = ntile_total % 8;
        const int xcd = tid_lin % 8, idx = tid_lin / 8;
        tid_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = tid_lin % p.ntiles, tile_m = tid_lin / p.ntiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long bz = blockIdx.z;
    const unsigned short* a0 = p.a0 + bz * p.a_bs;
    const unsigned short* a1 = p.a1 ? p.a1 + bz * p.a_bs : nullptr;
    const unsigned short* wg = p.w + bz * p.w_bs;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // staging ownership: global_load_lds_dwordx4 writes LDS at (wave-uniform base + 16 * lane), i.e.
    // lane l of wave w fills physical chunk l%8 of row 8w + l/8 of a 32-row pass.  The XOR swizzle
    // therefore moves to the SOURCE address: the lane fetches the logical chunk that belongs in its
    // physical slot (same 128-B row segment, coalescing intact).  (row>>1)&7 only depends on lrow.
    const int chunk = t & 7, lrow = t >> 3;
    const int lchunk8 = (chunk ^ ((lrow >> 1) & 7)) * 8;

    // per-thread staging rows: output pixel -> top-left input coordinate
    int a_img[MREP], a_y[MREP], a_x[MREP], a_pix[MREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = m0 + i * 32 + lrow;
        if (m < p.M) {
            const int img = m / p.rows_per_img, rem = m - img * p.rows_per_img;
            const int yo = rem / p.w_out;
            a_img[i] = img;
            a_y[i] = yo * p.stride - p.pad;
            a_x[i] = (rem - yo * p.w_out) * p.stride - p.pad;
        } else {
            a_img[i] = 0; a_y[i] = -(1 << 20); a_x[i] = 0;     // never in range
        }
    }
    int w_off[NREP];
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
        const int n = n0 + j * 32 + lrow;
        w_off[j] = n < p.N ? n * p.K + lchunk8 : -1;
    }
    const int Ctot = p.c0 + p.c1;
    const int Hl = p.h_in << p.up, Wl = p.w_in << p.up;

    // K walk (wave-uniform state): K-block kb = (tap, 64-channel block cc); no divisions in the loop
    const int nkb = p.K / 64;
    const int kb0 = blockIdx.y * p.kb_per_split;
    const int kb1 = min(nkb, kb0 + p.kb_per_split);
    int kg = kb0 * 64;
    int tap = kg / Ctot, cc = kg - tap * Ctot;
    auto set_tap = [&](int tp) {
        const int ky = p.ksize == 3 ? tp / 3 : 0, kx = p.ksize == 3 ? tp - 3 * ky : 0;
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            const int yi = a_y[i] + ky, xi = a_x[i] + kx;
            const bool ok = yi >= 0 && yi < Hl && xi >= 0 && xi < Wl;
            a_pix[i] = ok ? (a_img[i] * p.h_in + (yi >> p.up)) * p.w_in + (xi >> p.up) : -1;
        }
    };
    set_tap(tap);

    auto dma_stage = [&](int buf) {
        const unsigned short* src;
        int ld, coff;
        if (cc < p.c0) { src = a0; ld = p.a0_ld; coff = cc; } else { src = a1; ld = p.a1_ld; coff = cc - p.c0; }
        coff += lchunk8;
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            const unsigned short* g = a_pix[i] >= 0 ? src + (static_cast<long>(a_pix[i]) * ld + coff) : p.zeros;
            unsigned short* dst = As + buf * BM * 64 + (i * 32 + wave * 8) * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            const unsigned short* g = w_off[j] >= 0 ? wg + (static_cast<long>(w_off[j]) + kg) : p.zeros;
            unsigned short* dst = Bs + buf * BN * 64 + (j * 32 + wave * 8) * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        kg += 64;
        cc += 64;
        if (cc == Ctot) { cc = 0; ++tap; set_tap(tap); }
    };

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef typename Mfma<T>::frag frag;
    const int frow = lane & 15, fchunk = lane >> 4;
    auto compute = [&](int buf) {
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

    dma_stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kb = kb0, it = 0; kb < kb1; ++kb, ++it) {
        if (kb + 1 < kb1) dma_stage((it + 1) & 1);            // in flight during the MFMAs
        compute(it & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's DMA has landed ...
        __syncthreads();                                      // ... before anyone reads the tile
    }

    // epilogue: lane holds out[m][n4 .. n4+3], m = lane&15, n4 = 4*(lane>>4)
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = m0 + wm * 16 * MREP + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            const int n4 = n0 + wn * 16 * NREP + j * 16 + 4 * (lane >> 4);
            if (n4 >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.splits > 1) {
                float* o = p.partial + ((static_cast<long>(blockIdx.y) * p.batch + bz) * p.M + m) * p.N + n4;
                *reinterpret_cast<float4*>(o) = float4{v[0], v[1], v[2], v[3]};
            } else {
                epilogue_store<T>(p, bz, m, n4, v);
            }
        }
    }
}

// Split-K second pass: sum the fp32 slabs in split order (deterministic) and run the epilogue.
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_reduce(const GemmParams p) {
    const long quads = static_cast<long>(p.N / 4);
    const long total = static_cast<long>(p.batch) * p.M * quads;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total) return;
    const int n4 = static_cast<int>(i % quads) * 4;
    const long bm = i / quads;
    const int m = static_cast<int>(bm % p.M);
    const long bz = bm / p.M;
    const long slab = static_cast<long>(p.batch) * p.M * p.N;
    const float* src = p.partial + (bz * p.M + m) * p.N + n4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.splits; ++s) {
        const float4 x = *reinterpret_cast<const float4*>(src + s * slab);
        v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
    }
    epilogue_store<T>(p, bz, m, n4, v);
}

static const unsigned short* zero_page() {
    static unsigned short* z = nullptr;          // 256 zero bytes, created on the first launch (before any capture)
    if (!z) {
        if (hipMalloc(&z, 256) != hipSuccess) return nullptr;
        (void)hipMemset(z, 0, 256);
    }
    return z;
}

template <typename T, int MREP, int NREP>
static pf_status launch(const GemmParams& gp, int batch, hipStream_t st) {
    constexpr int BM = 32 * MREP, BN = 32 * NREP;
    GemmParams p = gp;
    p.zeros = zero_page();
    PF_REQUIRE(p.zeros, "pf_conv_gemm: zero page allocation failed");
    p.mtiles = static_cast<int>(cdiv(p.M, BM));
    p.ntiles = static_cast<int>(cdiv(p.N, BN));
    const size_t smem = static_cast<size_t>(2) * (BM + BN) * 64 * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_gemm<T, MREP, NREP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_conv_gemm<T, MREP, NREP>), dim3(p.mtiles * p.ntiles, p.splits, batch), dim3(256), smem, st, p);
    PF_CHECK_LAUNCH("pf_conv_gemm");
    if (p.splits > 1) {
        const long total = static_cast<long>(batch) * p.M * (p.N / 4);
        hipLaunchKernelGGL((k_splitk_reduce<T>), dim3(cdiv(total, 256)), dim3(256), 0, st, p);
        PF_CHECK_LAUNCH("pf_conv_gemm (split-K reduce)");
    }
    return PF_OK;
}

// Tile shape + split-K plan of one problem (shared by the launcher and the workspace query).
struct GemmPlan { int mrep, nrep, splits, kb_per_split; };
static GemmPlan plan_gemm(long M, int N, int K, int batch, bool allow_split) {
    GemmPlan g;
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
    if (allow_split && N % 4 == 0 && tiles <= 320 && nkb >= 12) {
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

}  // namespace pf

using namespace pf;

extern "C" pf_status pf_conv_gemm(const pf_conv_desc* d, void* stream) {
    PF_REQUIRE(d, "pf_conv_gemm: null descriptor");
    PF_REQUIRE(d->a0 && d->w && d->out, "pf_conv_gemm: null pointer");
    PF_REQUIRE(d->dtype == PF_BF16 || d->dtype == PF_F16, "pf_conv_gemm: dtype must be PF_BF16 or PF_F16");
    PF_REQUIRE(d->out_dtype == d->dtype || d->out_dtype == PF_F32, "pf_conv_gemm: out_dtype must equal dtype or be PF_F32");
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
    PF_REQUIRE(d->epilogue == PF_EPILOGUE_NONE || d->epilogue == PF_EPILOGUE_GEGLU, "pf_conv_gemm: unknown epilogue %d", d->epilogue);
    if (d->epilogue == PF_EPILOGUE_GEGLU)
        PF_REQUIRE(d->n_out % 4 == 0 && d->out_dtype == d->dtype && !d->residual && d->out_ld >= d->n_out / 2 && d->out_ld % 2 == 0,
                   "pf_conv_gemm: GEGLU epilogue needs n_out %% 4 == 0, 16-bit output, no residual, out_ld >= n_out/2");
    {   // the output size must be what the conv arithmetic produces
        const int hl = d->h_in << d->upsample, wl = d->w_in << d->upsample;
        const int ho = (hl + 2 * d->pad - d->ksize) / d->stride + 1, wo = (wl + 2 * d->pad - d->ksize) / d->stride + 1;
        PF_REQUIRE(ho == d->h_out && wo == d->w_out, "pf_conv_gemm: output size (%d,%d) does not match (%d,%d)", d->h_out, d->w_out, ho, wo);
    }
    GemmParams p;
    p.a0 = static_cast<const unsigned short*>(d->a0);
    p.a1 = static_cast<const unsigned short*>(d->a1);
    p.c0 = d->c0; p.c1 = c1; p.a0_ld = d->a0_ld; p.a1_ld = d->a1 ? d->a1_ld : 0;
    p.h_in = d->h_in; p.w_in = d->w_in; p.h_out = d->h_out; p.w_out = d->w_out;
    p.ksize = d->ksize; p.stride = d->stride; p.pad = d->pad; p.up = d->upsample;
    p.w = static_cast<const unsigned short*>(d->w);
    p.rows_per_img = d->h_out * d->w_out;
    p.M = d->n_img * p.rows_per_img; p.N = d->n_out; p.K = d->ksize * d->ksize * Ctot;
    p.bias = d->bias; p.rowvec = d->rowvec; p.rowvec_ld = d->rowvec_ld;
    p.residual = static_cast<const unsigned short*>(d->residual); p.res_ld = d->res_ld;
    p.out = d->out; p.out_ld = d->out_ld; p.out_f32 = d->out_dtype == PF_F32;
    p.geglu = d->epilogue == PF_EPILOGUE_GEGLU;
    p.a_bs = d->a_bstride; p.w_bs = d->w_bstride; p.out_bs = d->out_bstride; p.res_bs = d->res_bstride;
    p.mtiles = p.ntiles = 0;
    p.zeros = nullptr;
    p.batch = d->batch;
    const GemmPlan g = plan_gemm(p.M, p.N, p.K, d->batch, d->workspace != nullptr);
    p.splits = g.splits; p.kb_per_split = g.kb_per_split;
    p.partial = static_cast<float*>(d->workspace);
    if (p.splits > 1) {
        const size_t need = static_cast<size_t>(p.splits) * d->batch * p.M * p.N * sizeof(float);
        PF_REQUIRE(d->workspace_bytes >= need && aligned16(d->workspace),
                   "pf_conv_gemm: workspace of %zu bytes (16-byte aligned) needed, got %zu", need, d->workspace_bytes);
    }
    hipStream_t st = as_stream(stream);
    PF_DISPATCH_16(d->dtype, "pf_conv_gemm",
        if (g.nrep == 5) return g.mrep == 2 ? launch<T, 2, 5>(p, d->batch, st) : launch<T, 4, 5>(p, d->batch, st);
        else return g.mrep == 2 ? launch<T, 2, 4>(p, d->batch, st) : launch<T, 4, 4>(p, d->batch, st));
    return PF_OK;
}

extern "C" size_t pf_conv_gemm_workspace_size(const pf_conv_desc* d) {
    if (!d || d->batch < 1 || d->n_out < 1) return 0;
    const int c1 = d->a1 ? d->c1 : 0;
    const long M = static_cast<long>(d->n_img) * d->h_out * d->w_out;
    const int K = d->ksize * d->ksize * (d->c0 + c1);
    const GemmPlan g = plan_gemm(M, d->n_out, K, d->batch, true);
    return g.splits > 1 ? static_cast<size_t>(g.splits) * d->batch * M * d->n_out * sizeof(float) : 0;
}
