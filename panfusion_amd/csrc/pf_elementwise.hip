// HBM-bound kernels of the denoising path for gfx950: GroupNorm statistics / apply(+SiLU),
// LayerNorm(+positional encoding), GEGLU, timestep features, circular width pad / crop, layout
// converters, CFG+DDIM update, and the two 4-channel boundary convolutions.
//
// All activations are NHWC 16-bit, moved as 16-byte (8-element) vectors; statistics are fp32
// (fp64 for the final GroupNorm moments); reductions are wavefront (64-lane) shuffles.
//
// Reference call sites: diffusers ResnetBlock2D / Transformer2DModel driven from
// models/pano/MVGenModel.py:98-294; models/modules/transformer.py:8-38,151-162;
// utils/pano.py:74-105; models/pano/PanoGenerator.py:253-269; DDIMScheduler.step
// (models/pano/PanFusion.py:159-162).
#include "pf_common.h"
#include <stdlib.h>
#include <math.h>
#include <algorithm>

namespace pf {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// 8 consecutive channels of a 16-bit (T = Bf16 / F16) or fp32 (F32In) tensor as floats.  `ptr` is typed
// in units of the element, the octet must be 16-byte (16-bit) / 32-byte (fp32) aligned.
struct F32In {};
template <typename TI> struct In8 {
    typedef unsigned short elem;
    static __device__ __forceinline__ void load(const elem* ptr, float (&f)[8]) {
        unpack8<TI>(*reinterpret_cast<const u16x8*>(ptr), f);
    }
};
template <> struct In8<F32In> {
    typedef float elem;
    static __device__ __forceinline__ void load(const elem* ptr, float (&f)[8]) {
        const float4 a = reinterpret_cast<const float4*>(ptr)[0], b = reinterpret_cast<const float4*>(ptr)[1];
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
};
template <typename TI> __device__ __forceinline__ float elem_to_f32(typename In8<TI>::elem v) { return to_f32<TI>(v); }
template <> __device__ __forceinline__ float elem_to_f32<F32In>(float v) { return v; }
__device__ __forceinline__ void store8_f32(float* ptr, const float (&f)[8]) {
    reinterpret_cast<float4*>(ptr)[0] = float4{f[0], f[1], f[2], f[3]};
    reinterpret_cast<float4*>(ptr)[1] = float4{f[4], f[5], f[6], f[7]};
}

// ---- GroupNorm statistics ----------------------------------------------------------------------
// Stage 1: per (image, pixel chunk) block -> per-group (sum, sumsq) partials, deterministic order.
// Chunks are small (8..64 pixels, >= ~2000 blocks at the benchmark sizes) and every thread keeps
// four independent 16-byte loads in flight: the pass is HBM-bound, not latency-bound.
template <typename TI>
__global__ __launch_bounds__(256) void k_gn_partial(const typename In8<TI>::elem* __restrict__ x0, int c0,
                             const typename In8<TI>::elem* __restrict__ x1, int c1, int hw, int groups,
                             int pix_per_chunk, float* __restrict__ partial, int wimg, int wrap) {
    // wrap > 0: the moments of the tensor circularly padded by `wrap` columns (image width wimg) without building it -- the
    // first and last `wrap` columns count twice (GroupNorm inside pad_pano .. unpad_pano, MVGenModel.py:110-115)
    typedef typename In8<TI>::elem elem;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = c0 + c1, OCT = C / 8, cpg = C / groups;
    const int OCTB = OCT < 256 ? OCT : 256;
    const int pix_par = 256 / OCTB;
    float* csum = sm;                    // [pix_par][C]
    float* csq = sm + pix_par * C;       // [pix_par][C]
    const int img = blockIdx.y, chunk = blockIdx.x;
    const int p0 = chunk * pix_per_chunk;
    const int p1 = min(p0 + pix_per_chunk, hw);
    const int t = threadIdx.x;
    const int slot = t / OCTB, olane = t % OCTB;
    for (int ob = 0; ob < OCT; ob += OCTB) {
        const int oct = ob + olane;
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
        if (slot < pix_par && oct < OCT) {
            const int c = oct * 8;
            const elem* base;
            int ld, cc;
            if (c < c0) { base = x0; ld = c0; cc = c; } else { base = x1; ld = c1; cc = c - c0; }
            base += static_cast<long>(img) * hw * ld + cc;
            for (int p = p0 + slot; p < p1; p += 4 * pix_par) {
                float v[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pp = p + u * pix_par;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[u][j] = 0.f;
                    if (pp < p1) In8<TI>::load(base + static_cast<long>(pp) * ld, v[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float wgt = 1.f;
                    if (wrap > 0) {
                        const int col = (p + u * pix_par) % wimg;
                        wgt = (col < wrap || col >= wimg - wrap) ? 2.f : 1.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float f = v[u][j];
                        s[j] += f * wgt;
                        q[j] += f * f * wgt;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                csum[slot * C + c + j] = s[j];
                csq[slot * C + c + j] = q[j];
            }
        }
    }
    __syncthreads();
    for (int g = t; g < groups; g += 256) {
        float s = 0.f, q = 0.f;
        for (int sl = 0; sl < pix_par; ++sl)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
                s += csum[sl * C + c];
                q += csq[sl * C + c];
            }
        long o = ((static_cast<long>(img) * gridDim.x + chunk) * groups + g) * 2;
        partial[o] = s;
        partial[o + 1] = q;
    }
}

// Stage 2: chunk partials -> fp64 moments per group, folded with gamma/beta into per-(image, channel) scale / shift.
// grid (image, group block of GPB groups): 256 / GPB lanes share a group (4 independent loads in flight each),
// combined in a fixed order.  (Round 3: one block per image walked the chunks serially -- 24 us per launch on the
// 2-image panorama branch, 0.9 ms per step in 122 launches.)
constexpr int GN_GPB = 8;
__global__ __launch_bounds__(256) void k_gn_finalize(const float* __restrict__ partial, int nchunks, int groups, int C,
                              int hw, float eps, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float* __restrict__ scale,
                              float* __restrict__ shift) {
    __shared__ double red[2][256];
    __shared__ float g_mean[GN_GPB], g_rstd[GN_GPB];
    const int img = blockIdx.x, t = threadIdx.x;
    const int cpg = C / groups;
    constexpr int lpg = 256 / GN_GPB;
    const int gl = t / lpg, l = t % lpg, g = blockIdx.y * GN_GPB + gl;
    double s = 0.0, q = 0.0;
    if (g < groups) {
        const float* base = partial + (static_cast<long>(img) * nchunks * groups + g) * 2;
        int k = l;
        for (; k + 3 * lpg < nchunks; k += 4 * lpg) {
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float2*>(base + static_cast<long>(k + u * lpg) * groups * 2);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += v[u].x; q += v[u].y; }
        }
        for (; k < nchunks; k += lpg) {
            const float2 v = *reinterpret_cast<const float2*>(base + static_cast<long>(k) * groups * 2);
            s += v.x;
            q += v.y;
        }
    }
    red[0][t] = s;
    red[1][t] = q;
    __syncthreads();
    if (g < groups && l == 0) {
        for (int j = 1; j < lpg; ++j) { s += red[0][t + j]; q += red[1][t + j]; }
        const double cnt = static_cast<double>(hw) * cpg;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        g_mean[gl] = static_cast<float>(mean);
        g_rstd[gl] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    }
    __syncthreads();
    const int cbeg = blockIdx.y * GN_GPB * cpg, cend = min(C, cbeg + GN_GPB * cpg);
    for (int c = cbeg + t; c < cend; c += 256) {
        const int gg = (c - cbeg) / cpg;
        const float sc = g_rstd[gg] * gamma[c];
        scale[static_cast<long>(img) * C + c] = sc;
        shift[static_cast<long>(img) * C + c] = beta[c] - g_mean[gg] * sc;
    }
}

// Small tensors (the panorama branch's inner levels, the 8 x 8 / 16 x 16 view levels): statistics AND scale / shift in ONE launch,
// one block per (image, group) walking its hw x C/groups elements directly -- the two-stage form costs such a tensor two launches
// on a latency-bound chain (~75 of them per denoiser pass) for a few microseconds of work each.
template <typename TI>
__global__ __launch_bounds__(256) void k_gn_stats_direct(const typename In8<TI>::elem* __restrict__ x0, int c0,
                                  const typename In8<TI>::elem* __restrict__ x1, int c1, int hw, int groups, int hw_counted, float eps,
                                  const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ scale,
                                  float* __restrict__ shift, int wimg, int wrap) {
    typedef typename In8<TI>::elem elem;
    __shared__ double red[2][4];
    __shared__ float stat[2];
    const int img = blockIdx.x, g = blockIdx.y, t = threadIdx.x;
    const int C = c0 + c1, cpg = C / groups, cb = g * cpg;
    const long total = static_cast<long>(hw) * cpg;
    float s = 0.f, q = 0.f;
    for (long e = t; e < total; e += 256) {
        const int p = static_cast<int>(e / cpg), c = cb + static_cast<int>(e - static_cast<long>(p) * cpg);
        float v;
        if (c < c0) v = elem_to_f32<TI>(x0[(static_cast<long>(img) * hw + p) * c0 + c]);
        else v = elem_to_f32<TI>(x1[(static_cast<long>(img) * hw + p) * c1 + (c - c0)]);
        float wgt = 1.f;
        if (wrap > 0) {
            const int col = p % wimg;
            wgt = (col < wrap || col >= wimg - wrap) ? 2.f : 1.f;
        }
        s += v * wgt;
        q += v * v * wgt;
    }
    double ds = wave_sum(s), dq = wave_sum(q);                     // (fp32 inside a wave's 64 partial sums, fp64 across)
    if ((t & 63) == 0) { red[0][t >> 6] = ds; red[1][t >> 6] = dq; }
    __syncthreads();
    if (t == 0) {
        ds = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        dq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double cnt = static_cast<double>(hw_counted) * cpg;
        const double mean = ds / cnt;
        double var = dq / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[0] = static_cast<float>(mean);
        stat[1] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    }
    __syncthreads();
    for (int c = cb + t; c < cb + cpg; c += 256) {
        const float sc = stat[1] * gamma[c];
        scale[static_cast<long>(img) * C + c] = sc;
        shift[static_cast<long>(img) * C + c] = beta[c] - stat[0] * sc;
    }
}

// The same from the per-column-PAIR moments a GEMM epilogue left behind (pf_conv_desc.gn_partial): source s holds
// [n_img * ppi_s][2][c_s / 2] (sum, sum of squares of columns (2 k, 2 k + 1) over runs of hw / ppi_s rows); the channel
// concat (x0 | x1) is the concat of the two column ranges; groups hold an even number of channels and c0 is even, so a pair
// never straddles a group or the two sources.  grid (image, block of gpb groups).  The block's threads are laid out as
// (slice of the parts) x (pair): with few pairs per block (VAE: 2 per group) and thousands of parts per image (512 x 512 / 64)
// a thread per pair walking all parts took 250 us per call -- the host picks gpb so that 256 / (gpb * pairs per group)
// slices share the walk (4 loads in flight each), fp64 accumulation, fixed combination order.
__global__ __launch_bounds__(256) void k_gn_finalize_cols(const float* __restrict__ part0, int c0, int ppi0,
                                   const float* __restrict__ part1, int c1, int ppi1, int groups, int gpb, int hw, float eps,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ scale, float* __restrict__ shift) {
    constexpr int PMAX = 512;                         // pairs of one block (gpb * pairs per group)
    __shared__ double red_s[256], red_q[256];
    __shared__ double chs[PMAX], chq[PMAX];
    __shared__ float g_mean[GN_GPB], g_rstd[GN_GPB];
    const int img = blockIdx.x, t = threadIdx.x;
    const int C = c0 + c1, cpg = C / groups, ppg = cpg / 2;       // pairs per group
    const int P0 = c0 / 2, P1 = c1 / 2;
    const int pbeg = blockIdx.y * gpb * ppg, pend = min(C / 2, pbeg + gpb * ppg);
    const int npairs = pend - pbeg;
    const int slices = npairs <= 128 ? 256 / npairs : 1;
    for (int p0 = 0; p0 < npairs; p0 += 256) {                    // (one round unless a block holds more than 256 pairs)
        const int pl = slices > 1 ? t % npairs : p0 + t, slice = slices > 1 ? t / npairs : 0;
        double s = 0.0, q = 0.0;
        if (pl < npairs && slice < slices) {
            const int pr = pbeg + pl;
            const bool first = pr < P0;
            const int pp = first ? pr : pr - P0, Ps = first ? P0 : P1, ppi = first ? ppi0 : ppi1;
            const float* base = (first ? part0 : part1) + static_cast<long>(img) * ppi * 2 * Ps + pp;
            int k = slice;
            for (; k + 3 * slices < ppi; k += 4 * slices) {
                float a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = base[static_cast<long>(k + u * slices) * 2 * Ps];
                    b[u] = base[static_cast<long>(k + u * slices) * 2 * Ps + Ps];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { s += a[u]; q += b[u]; }
            }
            for (; k < ppi; k += slices) {
                s += base[static_cast<long>(k) * 2 * Ps];
                q += base[static_cast<long>(k) * 2 * Ps + Ps];
            }
        }
        if (slices > 1) {
            red_s[t] = s;
            red_q[t] = q;
            __syncthreads();
            if (t < npairs) {
                double ss = 0.0, qq = 0.0;
                for (int sl = 0; sl < slices; ++sl) { ss += red_s[sl * npairs + t]; qq += red_q[sl * npairs + t]; }
                chs[t] = ss;
                chq[t] = qq;
            }
        } else if (pl < npairs) {
            chs[pl] = s;
            chq[pl] = q;
        }
    }
    __syncthreads();
    if (t < gpb && blockIdx.y * gpb + t < groups) {
        double s = 0.0, q = 0.0;
        for (int j = 0; j < ppg; ++j) { s += chs[t * ppg + j]; q += chq[t * ppg + j]; }
        const double cnt = static_cast<double>(hw) * cpg;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        g_mean[t] = static_cast<float>(mean);
        g_rstd[t] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    }
    __syncthreads();
    const int cbeg = blockIdx.y * gpb * cpg, cend = min(C, cbeg + gpb * cpg);
    for (int c = cbeg + t; c < cend; c += 256) {
        const int gg = (c - cbeg) / cpg;
        const float sc = g_rstd[gg] * gamma[c];
        scale[static_cast<long>(img) * C + c] = sc;
        shift[static_cast<long>(img) * C + c] = beta[c] - g_mean[gg] * sc;
    }
}

// y = act(x * scale + shift) (scale == nullptr: identity), x = channel concat of two sources, 16-bit or fp32.
// OUT 0: 16-bit T [pix][C];  OUT 1: split pair [pix][per 32 channels: hi(32) | lo(32)] with hi = round16(v), lo = round16(v - hi)
// (the A operand of a split-precision GEMM, engine.py);  OUT 2: fp32 [pix][C].
template <typename TI, typename T, int OUT>
__global__ __launch_bounds__(256) void k_scale_shift_act(const typename In8<TI>::elem* __restrict__ x0, int c0,
                                  const typename In8<TI>::elem* __restrict__ x1, int c1, int hw,
                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                  int act, void* __restrict__ yv, unsigned short* __restrict__ raw_pair) {
    typedef typename In8<TI>::elem elem;
    // raw_pair (optional): the UN-normalised input as the split pair (same layout) -- the A operand of the resnet's
    // split-precision shortcut GEMM, written from the registers that already hold the fp32 input (its own pass re-read 629 MB
    // per 960-channel decoder resnet at 64 x 64)
    // grid (octet pairs of one image, image): 32-bit index arithmetic only, two octet loads in flight per thread
    const unsigned C = c0 + c1, OCT = C / 8, per_img = static_cast<unsigned>(hw) * OCT;
    const unsigned img = blockIdx.y;
    const unsigned j0 = (blockIdx.x * 256u + threadIdx.x) * 2u;
    if (j0 >= per_img) return;
    float v[2][8];
    unsigned cc[2];
    long pix[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned j = min(j0 + u, per_img - 1);
        const unsigned p = j / OCT, c = (j - p * OCT) * 8;
        cc[u] = c;
        pix[u] = static_cast<long>(img) * hw + p;
        const elem* src = (c < static_cast<unsigned>(c0)) ? x0 + pix[u] * c0 + c : x1 + pix[u] * c1 + (c - c0);
        In8<TI>::load(src, v[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (j0 + u >= per_img) break;
        if (raw_pair) {
            unsigned short* y = raw_pair + pix[u] * (2 * C) + pair_off(cc[u]);
            const u16x8 hi = pack8<T>(v[u]);
            float lo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) lo[j] = v[u][j] - to_f32<T>(hi[j]);
            *reinterpret_cast<u16x8*>(y) = hi;
            *reinterpret_cast<u16x8*>(y + 32) = pack8<T>(lo);
        }
        float f[8];
        if (scale) {
            const float4* sc = reinterpret_cast<const float4*>(scale + static_cast<long>(img) * C + cc[u]);
            const float4* sh = reinterpret_cast<const float4*>(shift + static_cast<long>(img) * C + cc[u]);
            float4 s0 = sc[0], s1 = sc[1], h0 = sh[0], h1 = sh[1];
            const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = v[u][j] * sv[j] + hv[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = v[u][j];
        }
        if (act) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = f[j] * __builtin_amdgcn_rcpf(1.0f + __expf(-f[j]));   // (v_exp + v_rcp: 1-2 ulp; the
                                                   // libm expf + IEEE division made the pass ALU-bound on tensors too large for the Infinity Cache)
        }
        if (OUT == 0) {
            *reinterpret_cast<u16x8*>(static_cast<unsigned short*>(yv) + pix[u] * C + cc[u]) = pack8<T>(f);
        } else if (OUT == 1) {
            unsigned short* y = static_cast<unsigned short*>(yv) + pix[u] * (2 * C) + pair_off(cc[u]);
            const u16x8 hi = pack8<T>(f);
            float lo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) lo[j] = f[j] - to_f32<T>(hi[j]);
            *reinterpret_cast<u16x8*>(y) = hi;
            *reinterpret_cast<u16x8*>(y + 32) = pack8<T>(lo);
        } else {
            store8_f32(static_cast<float*>(yv) + pix[u] * C + cc[u], f);
        }
    }
}

// ---- LayerNorm (+ PE) ------------------------------------------------------------------------
// One wavefront normalises ROWS rows at a time; all their 16-byte loads are issued before the first
// reduction (a 640-byte row per wave in flight is too little to cover HBM latency: 2.9 TB/s measured with
// one row per wave).  A row lives in registers (<= 4 octets per lane, C <= 2048).
template <typename TI, typename T, int ROWS>
__global__ __launch_bounds__(256) void k_layernorm(const typename In8<TI>::elem* __restrict__ x, const float* __restrict__ pe,
                            long pe_rows, long rows, int C, const float* __restrict__ gamma,
                            const float* __restrict__ beta, float eps, unsigned short* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long row0 = (blockIdx.x * static_cast<long>(blockDim.x >> 6) + (threadIdx.x >> 6)) * ROWS;
    if (row0 >= rows) return;
    const int OCT = C / 8;
    constexpr int KMAX = ROWS == 1 ? 4 : 1;            // octets per lane per row (the multi-row form: C <= 512)
    float raw[ROWS][KMAX][8];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int oct = lane + 64 * k;
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[r][k][j] = 0.f;
            if (oct < OCT && row0 + r < rows) In8<TI>::load(x + (row0 + r) * C + oct * 8, raw[r][k]);
        }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const long row = row0 + r;
        if (row >= rows) break;
        float (&v)[KMAX][8] = raw[r];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int oct = lane + 64 * k;
            if (oct < OCT) {
                if (pe) {
                    const float4* pp = reinterpret_cast<const float4*>(pe + (row % pe_rows) * C + oct * 8);
                    float4 a = pp[0], b = pp[1];
                    v[k][0] += a.x; v[k][1] += a.y; v[k][2] += a.z; v[k][3] += a.w;
                    v[k][4] += b.x; v[k][5] += b.y; v[k][6] += b.z; v[k][7] += b.w;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[k][j];
            }
        }
        const float mean = wave_sum(s) / static_cast<float>(C);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int oct = lane + 64 * k;
            if (oct < OCT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { float d = v[k][j] - mean; q += d * d; }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / static_cast<float>(C) + eps);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int oct = lane + 64 * k;
            if (oct < OCT) {
                const float4* gp = reinterpret_cast<const float4*>(gamma + oct * 8);
                const float4* bp = reinterpret_cast<const float4*>(beta + oct * 8);
                float4 g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1];
                const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (v[k][j] - mean) * rstd * gv[j] + bv[j];
                *reinterpret_cast<u16x8*>(y + row * C + oct * 8) = pack8<T>(f);
            }
        }
    }
}

// ---- GEGLU -----------------------------------------------------------------------------------
template <typename T>
__global__ void k_geglu(const unsigned short* __restrict__ in, long total_oct, int inner,
                        unsigned short* __restrict__ out) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= total_oct) return;
    const int OCT = inner / 8;
    const long row = i / OCT;
    const int c = (i % OCT) * 8;
    u16x8 a = *reinterpret_cast<const u16x8*>(in + row * 2 * inner + c);
    u16x8 g = *reinterpret_cast<const u16x8*>(in + row * 2 * inner + inner + c);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float gv = to_f32<T>(g[j]);
        float gelu = 0.5f * gv * (1.0f + erff(gv * 0.70710678118654752440f));
        f[j] = to_f32<T>(a[j]) * gelu;
    }
    *reinterpret_cast<u16x8*>(out + row * inner + c) = pack8<T>(f);
}

template <typename T>
__global__ void k_silu(const unsigned short* __restrict__ x, long n, unsigned short* __restrict__ y) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    float v = to_f32<T>(x[i]);
    y[i] = from_f32<T>(v / (1.0f + expf(-v)));
}


// diffusers Timesteps(flip_sin_to_cos=True, freq_shift=0): [cos(t f_i) | sin(t f_i)], fp32 math.
template <typename T>
__global__ void k_timestep_features(const int64_t* __restrict__ t, long t_stride, int n, int dim,
                                    unsigned short* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= n * half) return;
    const int r = i / half, k = i % half;
    const float e = (-9.210340371976184f * static_cast<float>(k)) / static_cast<float>(half);
    const float arg = static_cast<float>(t[r * t_stride]) * expf(e);
    out[static_cast<long>(r) * dim + k] = from_f32<T>(cosf(arg));
    out[static_cast<long>(r) * dim + half + k] = from_f32<T>(sinf(arg));
}

// ---- circular width pad / crop on NHWC ------------------------------------------------------
__global__ void k_shift_width(const u16x8* __restrict__ x, int h_rows, int w_in, int w_out, int oct,
                              int offset, u16x8* __restrict__ y) {
    // y[row][xo][:] = x[row][(xo + offset) mod w_in][:]; row = img*h + yy
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    long total = static_cast<long>(h_rows) * w_out * oct;
    if (i >= total) return;
    const int o = i % oct;
    const long px = i / oct;
    const int xo = px % w_out;
    const long row = px / w_out;
    int xi = (xo + offset) % w_in;
    if (xi < 0) xi += w_in;
    y[i] = x[(row * w_in + xi) * oct + o];
}

// circular pad of the innermost (width) axis of a row-major [rows][w] array (NCHW tensors)
template <typename E>
__global__ void k_pad_rows(const E* __restrict__ x, long rows, int w, int wo, int pad, E* __restrict__ y) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= rows * wo) return;
    const int xo = i % wo;
    const long r = i / wo;
    int xi = (xo - pad) % w;
    if (xi < 0) xi += w;
    y[i] = x[r * w + xi];
}

// ---- layout converters ------------------------------------------------------------------------

// y = a + b element-wise; a / y of type SA, b of type SB (fp32 stream + 16-bit ControlNet residual, ...)
template <typename SA, typename SB>
__global__ void k_add(const void* __restrict__ a, const void* __restrict__ b, long n, void* __restrict__ y) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    st_any<SA>(y, i, ld_any<SA>(a, i) + ld_any<SB>(b, i));
}

// out index enumerates the DESTINATION (coalesced writes).
template <typename SI, typename SO, bool TO_NHWC>
__global__ void k_permute(const void* x, int n, int C, long hw, void* y) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    long total = static_cast<long>(n) * C * hw;
    if (i >= total) return;
    long src;
    if (TO_NHWC) {
        int c = i % C; long p = (i / C) % hw; long b = i / (C * hw);
        src = (b * C + c) * hw + p;
    } else {
        long p = i % hw; int c = (i / hw) % C; long b = i / (hw * C);
        src = (b * hw + p) * C + c;
    }
    st_any<SO>(y, i, ld_any<SI>(x, src));
}

// ---- CFG + DDIM ------------------------------------------------------------------------------
// one element of the CFG merge + DDIM update; explicit fused multiply-adds so that both kernels below round identically
// whatever contraction the compiler would pick per loop copy
__device__ __forceinline__ float cfg_ddim_value(float x, float u, float c, float g, float sa, float sb, float sap, float sbp) {
    const float eps = __builtin_fmaf(g, c - u, u);
    const float x0 = __builtin_fmaf(-sb, eps, x) / sa;
    return __builtin_fmaf(sap, x0, sbp * eps);
}

__global__ void k_cfg_ddim(const float* __restrict__ x, const float* __restrict__ eu,
                           const float* __restrict__ ec, float g, float sa, float sb, float sap,
                           float sbp, long rows, int W, int roll, float* __restrict__ out) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= rows * W) return;
    const int w = i % W;
    const long r = i / W;
    const float xn = cfg_ddim_value(x[i], eu[i], ec[i], g, sa, sb, sap, sbp);
    int wo = (w + roll) % W;
    if (wo < 0) wo += W;
    out[r * W + wo] = xn;
}

// Loop-state update in ONE launch (round 5: no torch.cat / copy_ / fill_ kernels between two denoiser calls): one block per row,
// the row is updated into LDS at its rolled position and leaves in order, so `out` may alias `x` for ANY roll; a second copy
// `out2` (the CFG pair's other half -- the denoiser reads [x ; x]) and the next step's timestep words ride along.
__global__ __launch_bounds__(256) void k_cfg_ddim_rows(const float* x, const float* __restrict__ eu,
                                                       const float* __restrict__ ec, float g, float sa, float sb, float sap,
                                                       float sbp, int W, int roll, float* out, float* out2,
                                                       long long* tstep, int n_tstep, long long t_next) {
    extern __shared__ float row[];
    const long base = static_cast<long>(blockIdx.x) * W;
    for (int w = threadIdx.x; w < W; w += 256) {
        int wo = w + roll;
        wo -= wo >= W ? W : 0;
        row[wo] = cfg_ddim_value(x[base + w], eu[base + w], ec[base + w], g, sa, sb, sap, sbp);
    }
    __syncthreads();
    for (int w = threadIdx.x; w < W; w += 256) {
        const float v = row[w];
        out[base + w] = v;
        if (out2) out2[base + w] = v;
    }
    if (tstep && blockIdx.x == 0)
        for (int i = threadIdx.x; i < n_tstep; i += 256) tstep[i] = t_next;
}

// ---- token + position embedding gather (CLIP text encoder, transformers CLIPTextEmbeddings) -------------
// out[b][t][:] = tok[ids[b][t]][:] + pos[t][:] for t < L, zeros for the padding rows L <= t < Lp.
template <typename SO>
__global__ void k_embed_tokens(const int64_t* __restrict__ ids, int B, int L, int Lp, int C, long vocab,
                               const float* __restrict__ tok, const float* __restrict__ pos, void* __restrict__ out) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    const long total = static_cast<long>(B) * Lp * C;
    if (i >= total) return;
    const int c = i % C;
    const int t = (i / C) % Lp;
    const long b = i / (static_cast<long>(C) * Lp);
    float v = 0.f;
    if (t < L) {
        long id = ids[b * L + t];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // torch would raise; never read out of bounds
        v = tok[id * C + c] + pos[static_cast<long>(t) * C + c];
    }
    st_any<SO>(out, i, v);
}

// ---- row softmax (VAE mid-block attention: one head of width 512, scores through the GEMM kernel) -------
// p[r][j] = exp(scale * (s[r][j] - max_j s[r][j])) / sum, fp32 scores -> 16-bit probabilities.  One block
// per row; rows of up to 256 * PT columns live in registers between the three sweeps, longer rows are
// re-read from memory.
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();                                   // red[] may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

template <typename T, int PT>
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ s, long s_ld, int n, float scale,
                                                      unsigned short* __restrict__ p, long p_ld) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    const float* src = s + row * s_ld;
    unsigned short* dst = p + row * p_ld;
    const int t = threadIdx.x;
    const bool cached = n <= 256 * PT;
    float v[PT];
    float m = -INFINITY;
    if (cached) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int j = t + 256 * i;
            v[i] = j < n ? src[j] : -INFINITY;
            m = fmaxf(m, v[i]);
        }
    } else {
        for (int j = t; j < n; j += 256) m = fmaxf(m, src[j]);
    }
    m = block_reduce(m, true, red);
    const float c = scale * 1.44269504088896340736f;   // exp(scale * (x - m)) = exp2(c * x - c * m)
    float sum = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            v[i] = exp2f(c * (v[i] - m));              // exp2f(-inf) = 0 for the padding lanes
            sum += v[i];
        }
    } else {
        for (int j = t; j < n; j += 256) sum += exp2f(c * (src[j] - m));
    }
    sum = block_reduce(sum, false, red);
    const float inv = 1.0f / sum;
    if (cached) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int j = t + 256 * i;
            if (j < n) dst[j] = from_f32<T>(v[i] * inv);
        }
    } else {
        for (int j = t; j < n; j += 256) dst[j] = from_f32<T>(exp2f(c * (src[j] - m)) * inv);
    }
}

// models/modules/utils.py:9-15 tensor_to_image: fp32 NCHW image in [-1, 1] -> uint8 NHWC, (x / 2 + 0.5).clamp(0, 1) * 255
// rounded half-to-even like torch.round.
__global__ void k_tensor_to_image(const float* __restrict__ x, int n, int C, long hw, uint8_t* __restrict__ y) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    const long total = static_cast<long>(n) * C * hw;
    if (i >= total) return;
    const int c = i % C;
    const long p = (i / C) % hw, b = i / (C * hw);
    float v = x[(b * C + c) * hw + p] / 2.0f + 0.5f;
    v = fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f;
    y[i] = static_cast<uint8_t>(nearbyintf(v));
}

// ---- boundary convolutions ---------------------------------------------------------------------
// conv_in: x fp32 NCHW [n][cin][h][w] -> y NHWC 16-bit; weights fp32 [3][3][cin][cout].
template <typename T, bool OUT_F32>
__global__ void k_conv_in(const float* __restrict__ x, int n, int cin, int h, int w,
                          const float* __restrict__ wgt, const float* __restrict__ bias, int cout,
                          int wrap, void* __restrict__ y) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    const int OCT = cout / 8;
    long total = static_cast<long>(n) * h * w * OCT;
    if (i >= total) return;
    const int oct = i % OCT;
    const long pix = i / OCT;
    const int xx = pix % w, yy = (pix / w) % h, b = pix / (static_cast<long>(w) * h);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias ? bias[oct * 8 + j] : 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int yi = yy + ky - 1;
        if (yi < 0 || yi >= h) continue;
        for (int kx = 0; kx < 3; ++kx) {
            int xi = xx + kx - 1;
            if (wrap) xi = (xi + w) % w;
            else if (xi < 0 || xi >= w) continue;
            for (int c = 0; c < cin; ++c) {
                const float v = x[((static_cast<long>(b) * cin + c) * h + yi) * w + xi];
                const float4* wp = reinterpret_cast<const float4*>(wgt + (static_cast<long>((ky * 3 + kx) * cin + c)) * cout + oct * 8);
                float4 w0 = wp[0], w1 = wp[1];
                acc[0] += v * w0.x; acc[1] += v * w0.y; acc[2] += v * w0.z; acc[3] += v * w0.w;
                acc[4] += v * w1.x; acc[5] += v * w1.y; acc[6] += v * w1.z; acc[7] += v * w1.w;
            }
        }
    }
    if (OUT_F32) store8_f32(static_cast<float*>(y) + pix * cout + oct * 8, acc);
    else *reinterpret_cast<u16x8*>(static_cast<unsigned short*>(y) + pix * cout + oct * 8) = pack8<T>(acc);
}

// The same for the UNets' 4 latent channels: a thread owns one output-channel octet of FOUR horizontally adjacent pixels.  The
// generic kernel above re-reads its 72 weight quads (9 taps x 4 channels x 32 B) for every pixel -- 73 KB per wavefront from
// L1 / L2, 7.5 GB per launch for 210 MB of output; here they are read once per four pixels and the 72 input values of the
// 3 x 6 patch are requested up front.  w % 4 == 0.
// Round 5: the 36 x cout weights (46 KB at 320 channels) are staged ONCE per block in LDS and a block walks n_iter x 256
// (pixel group, octet) items (n_iter <= CONV_IN4_ITER, fewer when that would leave CUs without a block): read from global memory per item they do not fit the 32 KB L1 -- every wavefront streamed all of
// them from L2 again (1.2 GB per launch for 210 MB of output, 196 us at 40 x 64 x 64).  LDS layout [tap, c][half][octet] float4:
// consecutive lanes (octets) read consecutive 16 bytes.
constexpr int CONV_IN4_ITER = 8;
template <typename T, bool OUT_F32>
__global__ __launch_bounds__(256, 3) void k_conv_in4(const float* __restrict__ x, int n, int h, int w,
                                                const float* __restrict__ wgt, const float* __restrict__ bias, int cout,
                                                int wrap, void* __restrict__ y, int n_iter) {
    constexpr int CIN = 4, PX = 4;
    extern __shared__ __attribute__((aligned(16))) float4 wl4[];                 // [36][2][OCT]
    const int OCT = cout / 8, wq = w / PX;
    const long total = static_cast<long>(n) * h * wq * OCT;
    for (int s4 = threadIdx.x; s4 < 36 * 2 * OCT; s4 += 256) {
        const int tc = s4 / (2 * OCT), r = s4 - tc * 2 * OCT;
        wl4[(tc * 2 + (r & 1)) * OCT + (r >> 1)] = reinterpret_cast<const float4*>(wgt)[s4];
    }
    __syncthreads();
    for (int it = 0; it < n_iter; ++it) {
    const long i = (blockIdx.x * static_cast<long>(n_iter) + it) * 256 + threadIdx.x;
    if (i >= total) continue;
    const int oct = i % OCT;
    const long pg = i / OCT;
    const int x0 = (pg % wq) * PX, yy = (pg / wq) % h, b = pg / (static_cast<long>(wq) * h);
    const long plane = static_cast<long>(h) * w;
    const float* xb = x + static_cast<long>(b) * CIN * plane;
    float v[3][PX + 2][CIN];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int j = 0; j < PX + 2; ++j) {
            const int yi = yy + ky - 1;
            int xi = x0 + j - 1;
            bool ok = yi >= 0 && yi < h;
            if (wrap) xi = xi < 0 ? xi + w : (xi >= w ? xi - w : xi);
            else ok = ok && xi >= 0 && xi < w;
            const long off = static_cast<long>(ok ? yi : yy) * w + (ok ? xi : x0);
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float t = xb[c * plane + off];
                v[ky][j][c] = ok ? t : 0.f;
            }
        }
    typedef __attribute__((ext_vector_type(2))) float f32x2;       // two output channels per v_pk_fma_f32
    f32x2 acc2[PX][4];
#pragma unroll
    for (int q = 0; q < PX; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[q][j] = bias ? f32x2{bias[oct * 8 + 2 * j], bias[oct * 8 + 2 * j + 1]} : f32x2{0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const int tc = (ky * 3 + kx) * CIN + c;
                const float4 w0 = wl4[(tc * 2) * OCT + oct], w1 = wl4[(tc * 2 + 1) * OCT + oct];
                const f32x2 wa = {w0.x, w0.y}, wb = {w0.z, w0.w}, wc = {w1.x, w1.y}, wd = {w1.z, w1.w};
#pragma unroll
                for (int q = 0; q < PX; ++q) {
                    const float t = v[ky][q + kx][c];
                    const f32x2 tt = {t, t};
                    acc2[q][0] += tt * wa; acc2[q][1] += tt * wb; acc2[q][2] += tt * wc; acc2[q][3] += tt * wd;
                }
            }
    const long pix0 = (static_cast<long>(b) * h + yy) * w + x0;
    float acc[PX][8];
#pragma unroll
    for (int q = 0; q < PX; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[q][2 * j] = acc2[q][j][0]; acc[q][2 * j + 1] = acc2[q][j][1]; }
#pragma unroll
    for (int q = 0; q < PX; ++q) {
        if (OUT_F32) store8_f32(static_cast<float*>(y) + (pix0 + q) * cout + oct * 8, acc[q]);
        else *reinterpret_cast<u16x8*>(static_cast<unsigned short*>(y) + (pix0 + q) * cout + oct * 8) = pack8<T>(acc[q]);
    }
    }
}

// conv_out: x NHWC 16-bit [n][h][w][cin] -> y fp32 NCHW [n][cout<=8][h][w]; weights fp32
// [cout][3][3][cin].  LPP lanes per output pixel (the smallest power of two >= cin / 8, lanes over the channel octets of
// each tap), 64 / LPP pixels per wavefront: at the VAE's 128 channels a whole wavefront per pixel left 48 of 64 lanes idle
// (12.8 ms of a 107 ms decode in three launches, profiles/archive/r3h_vae_kernels.txt).
template <typename TI, int LPP>
__global__ void k_conv_out(const typename In8<TI>::elem* __restrict__ x, int n, int cin, int h, int w,
                           const float* __restrict__ wgt, const float* __restrict__ bias, int cout,
                           int wrap, float* __restrict__ y) {
    constexpr int PPW = 64 / LPP;                                 // pixels per wavefront
    const int lane = threadIdx.x & 63, sub = lane % LPP;
    const long pix = (blockIdx.x * static_cast<long>(blockDim.x >> 6) + (threadIdx.x >> 6)) * PPW + lane / LPP;
    const long npix = static_cast<long>(n) * h * w;
    const bool live = pix < npix;
    const long pc = live ? pix : npix - 1;
    const int xx = pc % w, yy = (pc / w) % h, b = pc / (static_cast<long>(w) * h);
    const int OCT = cin / 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int yi = yy + tap / 3 - 1;
        int xi = xx + tap % 3 - 1;
        if (wrap) xi = (xi + w) % w;
        if (yi < 0 || yi >= h || xi < 0 || xi >= w) continue;   // (uniform per pixel = per LPP-lane group)
        const typename In8<TI>::elem* src = x + ((static_cast<long>(b) * h + yi) * w + xi) * cin;
        for (int oct = sub; oct < OCT; oct += LPP) {
            float f[8];
            In8<TI>::load(src + oct * 8, f);
#pragma unroll
            for (int co = 0; co < 8; ++co) {
                if (co >= cout) break;
                const float4* wp = reinterpret_cast<const float4*>(wgt + (static_cast<long>(co) * 9 + tap) * cin + oct * 8);
                float4 w0 = wp[0], w1 = wp[1];
                acc[co] += f[0] * w0.x + f[1] * w0.y + f[2] * w0.z + f[3] * w0.w +
                           f[4] * w1.x + f[5] * w1.y + f[6] * w1.z + f[7] * w1.w;
            }
        }
    }
#pragma unroll
    for (int co = 0; co < 8; ++co) {
        if (co >= cout) break;
        float s = acc[co];
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (sub == 0 && live) y[((static_cast<long>(b) * cout + co) * h + yy) * w + xx] = s + (bias ? bias[co] : 0.f);
    }
}


// conv_out with its GroupNorm-apply + SiLU folded in (the UNets' head, fp32 streams): x fp32 NHWC [n][h][w][cin] (un-normalised),
// scale / shift [n][cin] (pf_groupnorm_*), weights fp32 [3][3][cin][4] (cout padded to 4), y fp32 NCHW [n][cout <= 4][h][w].
// The two-launch form wrote the activated tensor (210 MB at 40 x 64 x 64 x 320) only for k_conv_out to read it nine times
// through L2 (0.1 + 0.3 ms per denoiser pass on the view branch's stream).  Here a block owns an 8 x 32 tile of output pixels:
// per 32-channel chunk it normalises + activates the 10 x 34 halo tile ONCE into LDS (zero outside the image: the padding is
// applied to the ACTIVATED tensor; circular in width for the panorama), then every thread accumulates its pixel's 9 x 32 x 4
// products from LDS (ds_read_b128, pixel stride 36 floats: conflict free) against wavefront-uniform weights (scalar loads).
constexpr int COG_TH = 8, COG_TW = 32, COG_CH = 32, COG_PS = COG_CH + 4;
__global__ __launch_bounds__(256, 3) void k_conv_out_gn(const float* __restrict__ x, int cin, int h, int w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                   const float* __restrict__ wt, const float* __restrict__ bias, int cout,
                                                   int wrap, float* __restrict__ y) {
    constexpr int HP = COG_TH + 2, WP = COG_TW + 2, NPIX = HP * WP, NQ = COG_CH / 4;
    constexpr int FILL = (NPIX * NQ + 255) / 256;
    __shared__ __attribute__((aligned(16))) float tile[NPIX * COG_PS];
    __shared__ __attribute__((aligned(16))) float4 wl[9 * COG_CH];   // this chunk's weights [tap][c] (cout quad): broadcast reads
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const int t = threadIdx.x, tx = t % COG_TW, ty = t / COG_TW;
    const int x0 = blockIdx.x * COG_TW, y0 = blockIdx.y * COG_TH, img = blockIdx.z;
    const float* xb = x + static_cast<long>(img) * h * w * cin;
    const float* scb = scale + static_cast<long>(img) * cin;
    const float* shb = shift + static_cast<long>(img) * cin;
    const int q = t % NQ;                                          // this thread's channel quad of a chunk (256 % NQ == 0)
    // halo pixels this thread fills (the same for every chunk): source offset in pixels, or -1 = zero padding
    int src[FILL];
#pragma unroll
    for (int k = 0; k < FILL; ++k) {
        const int pixel = (t + 256 * k) / NQ;
        const int py = pixel / WP, px = pixel - py * WP;
        const int gy = y0 + py - 1;
        int gx = x0 + px - 1;
        if (wrap) gx = gx < 0 ? gx + w : (gx >= w ? gx - w : gx);
        const bool ok = pixel < NPIX && gy >= 0 && gy < h && gx >= 0 && gx < w;
        src[k] = ok ? gy * w + gx : -1;
    }
    f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
    for (int c0 = 0; c0 < cin; c0 += COG_CH) {
        float4 v[FILL];
#pragma unroll
        for (int k = 0; k < FILL; ++k)
            v[k] = src[k] >= 0 ? *reinterpret_cast<const float4*>(xb + static_cast<long>(src[k]) * cin + c0 + 4 * q) : float4{0.f, 0.f, 0.f, 0.f};
        const float4 sc = *reinterpret_cast<const float4*>(scb + c0 + 4 * q), sh = *reinterpret_cast<const float4*>(shb + c0 + 4 * q);
        // (9 x 32 = 288 weight quads: one per thread + 32 more)
        const float4 wv_fill = reinterpret_cast<const float4*>(wt)[static_cast<long>(t / COG_CH) * cin + c0 + t % COG_CH];
        const float4 wv_fill2 = reinterpret_cast<const float4*>(wt)[static_cast<long>(8) * cin + c0 + t % COG_CH];
        if (c0) __syncthreads();                                   // every thread is done reading the previous chunk
        wl[t] = wv_fill;
        if (t < COG_CH) wl[8 * COG_CH + t] = wv_fill2;
#pragma unroll
        for (int k = 0; k < FILL; ++k) {
            const int pixel = (t + 256 * k) / NQ;
            if (pixel >= NPIX) break;
            float f[4] = {v[k].x * sc.x + sh.x, v[k].y * sc.y + sh.y, v[k].z * sc.z + sh.z, v[k].w * sc.w + sh.w};
            if (act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = f[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-f[e]));      // (as k_scale_shift_act)
            }
            const bool ok = src[k] >= 0;
            *reinterpret_cast<float4*>(tile + pixel * COG_PS + 4 * q) = ok ? float4{f[0], f[1], f[2], f[3]} : float4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        // 36 groups of (tap, 8 channels): the 2 activation + 8 weight quads of group g + 1 are requested before the 16 packed
        // FMAs of group g are issued (the scheduling barriers keep hipcc from sinking every read next to its use, which left
        // one LDS round trip exposed per two FMAs)
        constexpr int NG = 9 * (COG_CH / 8);
        float4 av[2][2], wq[2][8];
        auto load_group = [&](int g, int buf) __attribute__((always_inline)) {
            const int tap = g / (COG_CH / 8), c8 = g % (COG_CH / 8);
            const float* a = tile + ((ty + tap / 3) * WP + tx + tap % 3) * COG_PS + 8 * c8;
            av[buf][0] = *reinterpret_cast<const float4*>(a);
            av[buf][1] = *reinterpret_cast<const float4*>(a + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) wq[buf][e] = wl[tap * COG_CH + 8 * c8 + e];      // wavefront-uniform address: an LDS broadcast
        };
        load_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load_group(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int b = g & 1;
            const float ae[8] = {av[b][0].x, av[b][0].y, av[b][0].z, av[b][0].w, av[b][1].x, av[b][1].y, av[b][1].z, av[b][1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc01 += f32x2{ae[e], ae[e]} * f32x2{wq[b][e].x, wq[b][e].y};
                acc23 += f32x2{ae[e], ae[e]} * f32x2{wq[b][e].z, wq[b][e].w};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int gy = y0 + ty, gx = x0 + tx;
    if (gy < h && gx < w) {
        const float r[4] = {acc01[0], acc01[1], acc23[0], acc23[1]};
        for (int co = 0; co < cout; ++co)
            y[((static_cast<long>(img) * cout + co) * h + gy) * w + gx] = r[co] + (bias ? bias[co] : 0.f);
    }
}

}  // namespace pf

using namespace pf;

// Dispatch an INPUT dtype (16-bit or fp32) to the loader tag TI.
#define PF_DISPATCH_IN(dtype, name, ...)                                          \
    do {                                                                          \
        if ((dtype) == PF_BF16) { using TI = pf::Bf16; __VA_ARGS__; }             \
        else if ((dtype) == PF_F16) { using TI = pf::F16; __VA_ARGS__; }          \
        else if ((dtype) == PF_F32) { using TI = pf::F32In; __VA_ARGS__; }        \
        else { pf::set_error("%s: dtype must be PF_BF16, PF_F16 or PF_F32", name); return PF_ERR_ARG; } \
    } while (0)

// pixels per statistics chunk: ~2048 blocks in flight, 8..64 pixels each, at most 256 chunks per image
static int gn_pixels_per_chunk(int n_img, int hw) {
    long ppc = static_cast<long>(n_img) * hw / 2048;
    if (ppc > 64) ppc = 64;
    if (ppc < 8) ppc = 8;
    const long floor_ppc = cdiv(hw, 256);
    if (ppc < floor_ppc) ppc = floor_ppc;
    if (ppc > hw) ppc = hw;
    return static_cast<int>(ppc);
}

extern "C" size_t pf_groupnorm_workspace_size(int n_img, int hw, int C) {
    (void)C;
    const int ppc = gn_pixels_per_chunk(n_img, hw);
    const long nchunks = cdiv(hw, ppc);
    return static_cast<size_t>(n_img) * nchunks * 64 * 2 * sizeof(float);   // up to 64 groups
}

static pf_status groupnorm_stats_impl(const void* x0, int c0, const void* x1, int c1, int dtype,
                                      int n_img, int hw, int groups, float eps, const float* gamma,
                                      const float* beta, float* scale, float* shift, void* workspace,
                                      size_t ws_bytes, void* stream, int wimg, int wrap);
extern "C" pf_status pf_groupnorm_stats(const void* x0, int c0, const void* x1, int c1, int dtype,
                                        int n_img, int hw, int groups, float eps, const float* gamma,
                                        const float* beta, float* scale, float* shift, void* workspace,
                                        size_t ws_bytes, void* stream) {
    return groupnorm_stats_impl(x0, c0, x1, c1, dtype, n_img, hw, groups, eps, gamma, beta, scale, shift, workspace, ws_bytes, stream, 0, 0);
}
extern "C" pf_status pf_groupnorm_stats_wrap(const void* x0, int c0, const void* x1, int c1, int dtype,
                                             int n_img, int h, int w, int wrap_pad, int groups, float eps, const float* gamma,
                                             const float* beta, float* scale, float* shift, void* workspace,
                                             size_t ws_bytes, void* stream) {
    PF_REQUIRE(h > 0 && w > 0 && wrap_pad >= 0 && 2 * wrap_pad <= w, "pf_groupnorm_stats_wrap: bad image size / padding");
    return groupnorm_stats_impl(x0, c0, x1, c1, dtype, n_img, h * w, groups, eps, gamma, beta, scale, shift, workspace, ws_bytes, stream, w, wrap_pad);
}
static pf_status groupnorm_stats_impl(const void* x0, int c0, const void* x1, int c1, int dtype,
                                      int n_img, int hw, int groups, float eps, const float* gamma,
                                      const float* beta, float* scale, float* shift, void* workspace,
                                      size_t ws_bytes, void* stream, int wimg, int wrap) {
    const int C = c0 + (x1 ? c1 : 0);
    if (!x1) c1 = 0;
    PF_REQUIRE(x0 && gamma && beta && scale && shift && workspace, "pf_groupnorm_stats: null pointer");
    PF_REQUIRE(n_img > 0 && hw > 0 && groups > 0 && groups <= 64, "pf_groupnorm_stats: bad sizes");
    PF_REQUIRE(C % 8 == 0 && c0 % 8 == 0 && C % groups == 0, "pf_groupnorm_stats: C=%d must be a multiple of 8 and of groups=%d", C, groups);
    PF_REQUIRE(aligned16(x0) && (!x1 || aligned16(x1)), "pf_groupnorm_stats: inputs must be 16-byte aligned");
    PF_REQUIRE(dtype != PF_F32 || (c0 % 4 == 0 && c1 % 4 == 0), "pf_groupnorm_stats: fp32 sources need c0, c1 %% 4 == 0");
    PF_REQUIRE(ws_bytes >= pf_groupnorm_workspace_size(n_img, hw, C), "pf_groupnorm_stats: workspace too small");
    hipStream_t st = as_stream(stream);
    const int hw_counted = wrap > 0 ? hw / wimg * (wimg + 2 * wrap) : hw;         // pixels of the virtually padded tensor
    static const long direct_max = getenv("PF_GN_DIRECT_MAX") ? atol(getenv("PF_GN_DIRECT_MAX")) : 32768;   // elements per (image, group); 0 = off (A/B)
    if (static_cast<long>(hw) * (C / groups) <= direct_max && static_cast<long>(n_img) * hw * C <= (4L << 20)) {
        PF_DISPATCH_IN(dtype, "pf_groupnorm_stats",
            hipLaunchKernelGGL(k_gn_stats_direct<TI>, dim3(n_img, groups), dim3(256), 0, st,
                               static_cast<const In8<TI>::elem*>(x0), c0, static_cast<const In8<TI>::elem*>(x1), c1,
                               hw, groups, hw_counted, eps, gamma, beta, scale, shift, wimg, wrap));
        PF_CHECK_LAUNCH("pf_groupnorm_stats (direct)");
        return PF_OK;
    }
    const int ppc = gn_pixels_per_chunk(n_img, hw);
    const int nchunks = static_cast<int>(cdiv(hw, ppc));
    const int OCT = C / 8, OCTB = OCT < 256 ? OCT : 256, pix_par = 256 / OCTB;
    const size_t smem = static_cast<size_t>(2) * pix_par * C * sizeof(float);
    PF_REQUIRE(smem <= 64 * 1024, "pf_groupnorm_stats: C=%d too large", C);
    float* partial = static_cast<float*>(workspace);
    PF_DISPATCH_IN(dtype, "pf_groupnorm_stats",
        hipLaunchKernelGGL(k_gn_partial<TI>, dim3(nchunks, n_img), dim3(256), smem, st,
                           static_cast<const In8<TI>::elem*>(x0), c0, static_cast<const In8<TI>::elem*>(x1), c1,
                           hw, groups, ppc, partial, wimg, wrap));
    hipLaunchKernelGGL(k_gn_finalize, dim3(n_img, cdiv(groups, GN_GPB)), dim3(256), 0, st, partial, nchunks, groups, C, hw_counted, eps,
                       gamma, beta, scale, shift);
    PF_CHECK_LAUNCH("pf_groupnorm_stats");
    return PF_OK;
}

extern "C" pf_status pf_groupnorm_from_partials(const float* part0, int c0, int rows0, const float* part1, int c1, int rows1,
                                                int n_img, int hw, int groups, float eps, const float* gamma,
                                                const float* beta, float* scale, float* shift, void* stream) {
    if (!part1) { c1 = 0; rows1 = rows0; }
    const int C = c0 + c1;
    PF_REQUIRE(part0 && gamma && beta && scale && shift, "pf_groupnorm_from_partials: null pointer");
    PF_REQUIRE(n_img > 0 && hw > 0 && groups > 0 && groups <= 64 && C % groups == 0 && c0 > 0,
               "pf_groupnorm_from_partials: bad sizes");
    PF_REQUIRE(rows0 > 0 && rows1 > 0 && hw % rows0 == 0 && hw % rows1 == 0,
               "pf_groupnorm_from_partials: an image (%d rows) must be whole runs of %d / %d rows", hw, rows0, rows1);
    PF_REQUIRE((C / groups) % 2 == 0 && c0 % 2 == 0 && c1 % 2 == 0, "pf_groupnorm_from_partials: groups and sources must hold even numbers of channels (moments are per column pair)");
    const int ppi = std::max(hw / rows0, hw / rows1);
    int gpb = ppi <= 32 ? 8 : ppi <= 128 ? 4 : ppi <= 512 ? 2 : 1;       // fewer groups per block = more slices walking the parts
    while (gpb > 1 && gpb * (C / groups) / 2 > 512) gpb >>= 1;
    PF_REQUIRE(gpb * (C / groups) / 2 <= 512, "pf_groupnorm_from_partials: C=%d too large for %d groups", C, groups);
    hipLaunchKernelGGL(k_gn_finalize_cols, dim3(n_img, cdiv(groups, gpb)), dim3(256), 0, as_stream(stream),
                       part0, c0, hw / rows0, part1, c1, hw / rows1, groups, gpb, hw, eps, gamma, beta, scale, shift);
    PF_CHECK_LAUNCH("pf_groupnorm_from_partials");
    return PF_OK;
}

static pf_status scale_shift_act_impl(const void* x0, int c0, const void* x1, int c1, int dtype,
                                      int n_img, int hw, const float* scale, const float* shift,
                                      int act, int out_dtype, int out_split, void* y, void* raw_pair, void* stream);
extern "C" pf_status pf_scale_shift_act(const void* x0, int c0, const void* x1, int c1, int dtype,
                                        int n_img, int hw, const float* scale, const float* shift,
                                        int act, int out_dtype, int out_split, void* y, void* stream) {
    return scale_shift_act_impl(x0, c0, x1, c1, dtype, n_img, hw, scale, shift, act, out_dtype, out_split, y, nullptr, stream);
}
extern "C" pf_status pf_scale_shift_act_pair(const void* x0, int c0, const void* x1, int c1, int n_img, int hw,
                                             const float* scale, const float* shift, int act, int out_dtype, void* y,
                                             void* raw_pair, void* stream) {
    PF_REQUIRE(raw_pair && aligned16(raw_pair), "pf_scale_shift_act_pair: raw_pair must be a 16-byte aligned pointer");
    return scale_shift_act_impl(x0, c0, x1, c1, PF_F32, n_img, hw, scale, shift, act, out_dtype, 0, y, raw_pair, stream);
}
static pf_status scale_shift_act_impl(const void* x0, int c0, const void* x1, int c1, int dtype,
                                      int n_img, int hw, const float* scale, const float* shift,
                                      int act, int out_dtype, int out_split, void* y, void* raw_pair, void* stream) {
    if (!x1) c1 = 0;
    const int C = c0 + c1;
    PF_REQUIRE(x0 && y, "pf_scale_shift_act: null pointer");
    PF_REQUIRE((scale == nullptr) == (shift == nullptr), "pf_scale_shift_act: scale and shift must both be given or both be NULL");
    PF_REQUIRE(C % 8 == 0 && c0 % 8 == 0 && n_img > 0 && hw > 0, "pf_scale_shift_act: bad sizes");
    PF_REQUIRE(aligned16(x0) && aligned16(y) && (!x1 || aligned16(x1)) && (!scale || (aligned16(scale) && aligned16(shift))),
               "pf_scale_shift_act: pointers must be 16-byte aligned");
    PF_REQUIRE(out_dtype == PF_BF16 || out_dtype == PF_F16 || (out_dtype == PF_F32 && !out_split),
               "pf_scale_shift_act: out_dtype must be 16-bit (optionally split) or PF_F32");
    PF_REQUIRE(dtype == PF_F32 || out_dtype == PF_F32 || dtype == out_dtype, "pf_scale_shift_act: 16-bit input and output types must agree");
    PF_REQUIRE(!raw_pair || (dtype == PF_F32 && out_dtype != PF_F32 && !out_split), "pf_scale_shift_act_pair: fp32 sources, plain 16-bit output");
    PF_REQUIRE(!(raw_pair || out_split) || C % 32 == 0, "pf_scale_shift_act: a split pair interleaves hi / lo per 32 channels: C=%d must be a multiple of 32", C);
    const long per_img = static_cast<long>(hw) * (C / 8);
    PF_REQUIRE(per_img < (1L << 31) && n_img <= 65535, "pf_scale_shift_act: image too large / too many images");
    const dim3 grid(cdiv(per_img, 512), n_img), block(256);
    hipStream_t st = as_stream(stream);
#define PF_SSA(TI, T, OUT) hipLaunchKernelGGL((k_scale_shift_act<TI, T, OUT>), grid, block, 0, st,                      \
                                              static_cast<const In8<TI>::elem*>(x0), c0,                                \
                                              static_cast<const In8<TI>::elem*>(x1), c1, hw, scale, shift, act, y,       \
                                              static_cast<unsigned short*>(raw_pair))
    const int out_kind = out_dtype == PF_F32 ? 2 : (out_split ? 1 : 0);
    // T = the 16-bit type on whichever side has one (bf16 when both sides are fp32: unused)
    const int t16 = out_dtype != PF_F32 ? out_dtype : (dtype != PF_F32 ? dtype : PF_BF16);
    PF_DISPATCH_16(t16, "pf_scale_shift_act",
        if (dtype == PF_F32) {
            if (out_kind == 0) PF_SSA(F32In, T, 0); else if (out_kind == 1) PF_SSA(F32In, T, 1); else PF_SSA(F32In, T, 2);
        } else {
            if (out_kind == 0) PF_SSA(T, T, 0); else if (out_kind == 1) PF_SSA(T, T, 1); else PF_SSA(T, T, 2);
        });
#undef PF_SSA
    PF_CHECK_LAUNCH("pf_scale_shift_act");
    return PF_OK;
}

extern "C" pf_status pf_layernorm(const void* x, const float* pe, long pe_rows, int dtype, long rows,
                                  int C, const float* gamma, const float* beta, float eps, int out_dtype, void* y,
                                  void* stream) {
    PF_REQUIRE(x && gamma && beta && y && rows > 0, "pf_layernorm: bad arguments");
    PF_REQUIRE(C % 8 == 0 && C <= 2048, "pf_layernorm: C=%d must be a multiple of 8 and <= 2048", C);
    PF_REQUIRE(!pe || pe_rows > 0, "pf_layernorm: pe_rows must be > 0 with pe");
    PF_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta) && (!pe || aligned16(pe)),
               "pf_layernorm: pointers must be 16-byte aligned");
    PF_REQUIRE(dtype == PF_F32 || dtype == out_dtype, "pf_layernorm: a 16-bit input must have the output's type");
    hipStream_t st = as_stream(stream);
    unsigned short* yo = static_cast<unsigned short*>(y);
#define PF_LN(TI, T) do {                                                                                                \
        if (C <= 512 && rows >= 4096)      /* narrow rows: 4 rows per wavefront in flight */                             \
            hipLaunchKernelGGL((k_layernorm<TI, T, 4>), dim3(cdiv(rows, 16)), dim3(256), 0, st,                          \
                               static_cast<const In8<TI>::elem*>(x), pe, pe_rows, rows, C, gamma, beta, eps, yo);        \
        else                                                                                                             \
            hipLaunchKernelGGL((k_layernorm<TI, T, 1>), dim3(cdiv(rows, 4)), dim3(256), 0, st,                           \
                               static_cast<const In8<TI>::elem*>(x), pe, pe_rows, rows, C, gamma, beta, eps, yo);        \
    } while (0)
    PF_DISPATCH_16(out_dtype, "pf_layernorm",
        if (dtype == PF_F32) PF_LN(F32In, T); else PF_LN(T, T));
#undef PF_LN
    PF_CHECK_LAUNCH("pf_layernorm");
    return PF_OK;
}

__global__ void k_vae_sample(const float* __restrict__ mom, const float* __restrict__ eps, int n, int L, long hw, float scale,
                             float* __restrict__ z) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;      // NCHW index
    if (i >= static_cast<long>(n) * L * hw) return;
    const long p = i % hw;
    const int c = (i / hw) % L;
    const long b = i / (hw * L);
    const float* m = mom + (b * hw + p) * 2 * L;
    const float logvar = fminf(fmaxf(m[L + c], -30.f), 20.f);
    z[i] = (m[c] + expf(0.5f * logvar) * eps[i]) * scale;
}

__global__ void k_axpby(const float* __restrict__ x, const float* __restrict__ y, float a, float b, long n, float* __restrict__ out) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i];
}

extern "C" pf_status pf_axpby(const float* x, const float* y, float a, float b, long n, float* out, void* stream) {
    PF_REQUIRE(x && y && out && n > 0, "pf_axpby: bad arguments");
    hipLaunchKernelGGL(k_axpby, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), x, y, a, b, n, out);
    PF_CHECK_LAUNCH("pf_axpby");
    return PF_OK;
}

extern "C" pf_status pf_vae_sample(const float* moments, const float* eps, int n, int L, long hw, float scale, float* z, void* stream) {
    PF_REQUIRE(moments && eps && z && n > 0 && L > 0 && hw > 0, "pf_vae_sample: bad arguments");
    const long total = static_cast<long>(n) * L * hw;
    hipLaunchKernelGGL(k_vae_sample, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), moments, eps, n, L, hw, scale, z);
    PF_CHECK_LAUNCH("pf_vae_sample");
    return PF_OK;
}

extern "C" pf_status pf_geglu(const void* in, int dtype, long rows, int inner, void* out, void* stream) {
    PF_REQUIRE(in && out && rows > 0 && inner > 0 && inner % 8 == 0, "pf_geglu: bad arguments");
    PF_REQUIRE(aligned16(in) && aligned16(out), "pf_geglu: pointers must be 16-byte aligned");
    const long total = rows * (inner / 8);
    PF_DISPATCH_16(dtype, "pf_geglu",
        hipLaunchKernelGGL(k_geglu<T>, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream),
                           static_cast<const unsigned short*>(in), total, inner, static_cast<unsigned short*>(out)));
    PF_CHECK_LAUNCH("pf_geglu");
    return PF_OK;
}

extern "C" pf_status pf_timestep_features_strided(const int64_t* t, long t_stride, int n, int dim, int out_dtype, void* out,
                                                  void* stream) {
    PF_REQUIRE(t && out && n > 0 && dim > 0 && dim % 2 == 0 && t_stride >= 0, "pf_timestep_features: bad arguments");
    PF_DISPATCH_16(out_dtype, "pf_timestep_features",
        hipLaunchKernelGGL(k_timestep_features<T>, dim3(cdiv(static_cast<long>(n) * (dim / 2), 256)), dim3(256),
                           0, as_stream(stream), t, t_stride, n, dim, static_cast<unsigned short*>(out)));
    PF_CHECK_LAUNCH("pf_timestep_features");
    return PF_OK;
}

extern "C" pf_status pf_timestep_features(const int64_t* t, int n, int dim, int out_dtype, void* out,
                                          void* stream) {
    return pf_timestep_features_strided(t, 1, n, dim, out_dtype, out, stream);
}

extern "C" pf_status pf_silu(const void* x, int dtype, long n, void* y, void* stream) {
    PF_REQUIRE(x && y && n > 0, "pf_silu: bad arguments");
    PF_DISPATCH_16(dtype, "pf_silu",
        hipLaunchKernelGGL(k_silu<T>, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream),
                           static_cast<const unsigned short*>(x), n, static_cast<unsigned short*>(y)));
    PF_CHECK_LAUNCH("pf_silu");
    return PF_OK;
}

extern "C" pf_status pf_add(const void* a, int dtype_a, const void* b, int dtype_b, long n, void* y, void* stream) {
    PF_REQUIRE(a && b && y && n > 0, "pf_add: bad arguments");
    const dim3 grid(cdiv(n, 256)), block(256);
    hipStream_t st = as_stream(stream);
#define PF_ADD(SA, SB) hipLaunchKernelGGL((k_add<SA, SB>), grid, block, 0, st, a, b, n, y)
#define PF_ADD_B(SA)                                                                  \
    do {                                                                              \
        if (dtype_b == PF_F32) PF_ADD(SA, AnyF32);                                    \
        else if (dtype_b == PF_BF16) PF_ADD(SA, AnyBf16);                             \
        else if (dtype_b == PF_F16) PF_ADD(SA, AnyF16);                               \
        else PF_REQUIRE(false, "pf_add: unknown dtype_b %d", dtype_b);               \
    } while (0)
    if (dtype_a == PF_F32) PF_ADD_B(AnyF32);
    else if (dtype_a == PF_BF16) PF_ADD_B(AnyBf16);
    else if (dtype_a == PF_F16) PF_ADD_B(AnyF16);
    else PF_REQUIRE(false, "pf_add: unknown dtype_a %d", dtype_a);
#undef PF_ADD_B
#undef PF_ADD
    PF_CHECK_LAUNCH("pf_add");
    return PF_OK;
}

static pf_status shift_width(const void* x, int dtype, int n, int h, int w_in, int w_out, int C, int offset,
                             void* y, void* stream, const char* who) {
    PF_REQUIRE(x && y && n > 0 && h > 0 && w_in > 0 && w_out > 0, "%s: bad sizes", who);
    PF_REQUIRE(dtype == PF_BF16 || dtype == PF_F16 || dtype == PF_F32, "%s: unknown dtype", who);
    if (dtype == PF_F32) C *= 2;                         // whole pixels are moved: an fp32 channel = two 16-bit words
    PF_REQUIRE(C % 8 == 0 && aligned16(x) && aligned16(y), "%s: 16-byte pixels and alignment required", who);
    const long total = static_cast<long>(n) * h * w_out * (C / 8);
    hipLaunchKernelGGL(k_shift_width, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream),
                       static_cast<const u16x8*>(x), n * h, w_in, w_out, C / 8, offset, static_cast<u16x8*>(y));
    PF_CHECK_LAUNCH(who);
    return PF_OK;
}

extern "C" pf_status pf_pad_width(const void* x, int dtype, int n, int h, int w, int C, int pad, void* y, void* stream) {
    PF_REQUIRE(pad >= 0 && pad <= w, "pf_pad_width: pad must be in [0, w]");
    return shift_width(x, dtype, n, h, w, w + 2 * pad, C, -pad, y, stream, "pf_pad_width");
}

extern "C" pf_status pf_crop_width(const void* x, int dtype, int n, int h, int w, int C, int crop, void* y, void* stream) {
    PF_REQUIRE(crop >= 0 && 2 * crop < w, "pf_crop_width: crop too large");
    return shift_width(x, dtype, n, h, w, w - 2 * crop, C, crop, y, stream, "pf_crop_width");
}

static pf_status pad_rows(const void* x, int elem_bytes, long rows, int w, int wo, int shift, void* y,
                          void* stream, const char* who) {
    PF_REQUIRE(x && y && x != y && rows > 0 && w > 0, "%s: bad arguments", who);
    PF_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "%s: element size must be 2 or 4 bytes", who);
    const long total = rows * wo;
    if (elem_bytes == 2)
        hipLaunchKernelGGL(k_pad_rows<unsigned short>, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream),
                           static_cast<const unsigned short*>(x), rows, w, wo, shift, static_cast<unsigned short*>(y));
    else
        hipLaunchKernelGGL(k_pad_rows<unsigned>, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream),
                           static_cast<const unsigned*>(x), rows, w, wo, shift, static_cast<unsigned*>(y));
    PF_CHECK_LAUNCH(who);
    return PF_OK;
}

extern "C" pf_status pf_pad_width_rows(const void* x, int elem_bytes, long rows, int w, int pad, void* y, void* stream) {
    PF_REQUIRE(pad >= 0 && pad <= w, "pf_pad_width_rows: pad must be in [0, w]");
    return pad_rows(x, elem_bytes, rows, w, w + 2 * pad, pad, y, stream, "pf_pad_width_rows");
}

extern "C" pf_status pf_roll_width_rows(const void* x, int elem_bytes, long rows, int w, int shift, void* y, void* stream) {
    // torch.roll(x, shift, -1): y[..., (i + shift) mod w] = x[..., i]
    return pad_rows(x, elem_bytes, rows, w, w, ((shift % w) + w) % w, y, stream, "pf_roll_width_rows");
}

template <bool TO_NHWC>
static pf_status permute(const void* x, int sd, int n, int C, int h, int w, int dd, void* y, void* stream,
                         const char* who) {
    PF_REQUIRE(x && y && n > 0 && C > 0 && h > 0 && w > 0, "%s: bad arguments", who);
    const long hw = static_cast<long>(h) * w, total = hw * n * C;
    dim3 grid(cdiv(total, 256)), block(256);
    hipStream_t st = as_stream(stream);
#define PF_PERM(SI, SO) hipLaunchKernelGGL((k_permute<SI, SO, TO_NHWC>), grid, block, 0, st, x, n, C, hw, y)
    if (sd == PF_F32 && dd == PF_F32) PF_PERM(AnyF32, AnyF32);
    else if (sd == PF_F32 && dd == PF_BF16) PF_PERM(AnyF32, AnyBf16);
    else if (sd == PF_F32 && dd == PF_F16) PF_PERM(AnyF32, AnyF16);
    else if (sd == PF_BF16 && dd == PF_F32) PF_PERM(AnyBf16, AnyF32);
    else if (sd == PF_F16 && dd == PF_F32) PF_PERM(AnyF16, AnyF32);
    else if (sd == PF_BF16 && dd == PF_BF16) PF_PERM(AnyBf16, AnyBf16);
    else if (sd == PF_F16 && dd == PF_F16) PF_PERM(AnyF16, AnyF16);
    else PF_REQUIRE(false, "%s: unsupported dtype pair %d -> %d", who, sd, dd);
#undef PF_PERM
    PF_CHECK_LAUNCH(who);
    return PF_OK;
}

extern "C" pf_status pf_nchw_to_nhwc(const void* x, int sd, int n, int C, int h, int w, int dd, void* y, void* stream) {
    return permute<true>(x, sd, n, C, h, w, dd, y, stream, "pf_nchw_to_nhwc");
}
extern "C" pf_status pf_nhwc_to_nchw(const void* x, int sd, int n, int C, int h, int w, int dd, void* y, void* stream) {
    return permute<false>(x, sd, n, C, h, w, dd, y, stream, "pf_nhwc_to_nchw");
}

extern "C" pf_status pf_cfg_ddim_step(const float* x, const float* eu, const float* ec, float g, float sa,
                                      float sb, float sap, float sbp, long rows, int W, int roll, float* out,
                                      void* stream) {
    PF_REQUIRE(x && eu && ec && out && rows > 0 && W > 0, "pf_cfg_ddim_step: bad arguments");
    PF_REQUIRE(out != x || roll % W == 0, "pf_cfg_ddim_step: in-place update requires roll == 0");
    hipLaunchKernelGGL(k_cfg_ddim, dim3(cdiv(rows * W, 256)), dim3(256), 0, as_stream(stream), x, eu, ec, g, sa,
                       sb, sap, sbp, rows, W, roll, out);
    PF_CHECK_LAUNCH("pf_cfg_ddim_step");
    return PF_OK;
}

extern "C" pf_status pf_cfg_ddim_step_pair(const float* x, const float* eu, const float* ec, float g, float sa, float sb, float sap,
                                           float sbp, long rows, int W, int roll, float* out, float* out2,
                                           int64_t* tstep, int n_tstep, int64_t t_next, void* stream) {
    PF_REQUIRE(x && eu && ec && out && rows > 0 && W > 0, "pf_cfg_ddim_step_pair: bad arguments");
    PF_REQUIRE(rows < (1L << 31) && W <= 16384, "pf_cfg_ddim_step_pair: rows=%ld must be < 2^31 and W=%d <= 16384 (one row per block, staged in LDS)", rows, W);
    PF_REQUIRE(out2 != x && out2 != out && out != eu && out != ec, "pf_cfg_ddim_step_pair: out2 must be a buffer of its own, out must not alias the predictions");
    PF_REQUIRE(!tstep || n_tstep > 0, "pf_cfg_ddim_step_pair: n_tstep must be positive with tstep");
    int r = roll % W;
    if (r < 0) r += W;
    hipLaunchKernelGGL(k_cfg_ddim_rows, dim3(static_cast<unsigned>(rows)), dim3(256), static_cast<size_t>(W) * sizeof(float), as_stream(stream),
                       x, eu, ec, g, sa, sb, sap, sbp, W, r, out, out2, reinterpret_cast<long long*>(tstep), n_tstep, static_cast<long long>(t_next));
    PF_CHECK_LAUNCH("pf_cfg_ddim_step_pair");
    return PF_OK;
}

extern "C" pf_status pf_embed_tokens(const int64_t* ids, int B, int L, int Lp, int C, long vocab, const float* tok,
                                     const float* pos, int out_dtype, void* out, void* stream) {
    PF_REQUIRE(ids && tok && pos && out && B > 0 && L > 0 && Lp >= L && C > 0 && vocab > 0, "pf_embed_tokens: bad arguments");
    const long total = static_cast<long>(B) * Lp * C;
    const dim3 grid(cdiv(total, 256)), block(256);
    hipStream_t st = as_stream(stream);
    if (out_dtype == PF_F32) hipLaunchKernelGGL(k_embed_tokens<AnyF32>, grid, block, 0, st, ids, B, L, Lp, C, vocab, tok, pos, out);
    else if (out_dtype == PF_F16) hipLaunchKernelGGL(k_embed_tokens<AnyF16>, grid, block, 0, st, ids, B, L, Lp, C, vocab, tok, pos, out);
    else if (out_dtype == PF_BF16) hipLaunchKernelGGL(k_embed_tokens<AnyBf16>, grid, block, 0, st, ids, B, L, Lp, C, vocab, tok, pos, out);
    else PF_REQUIRE(false, "pf_embed_tokens: unsupported out_dtype %d", out_dtype);
    PF_CHECK_LAUNCH("pf_embed_tokens");
    return PF_OK;
}

extern "C" pf_status pf_softmax_rows(const float* scores, long rows, int n, long scores_ld, float scale, int out_dtype,
                                     void* probs, long probs_ld, void* stream) {
    PF_REQUIRE(scores && probs && rows > 0 && n > 0, "pf_softmax_rows: bad arguments");
    PF_REQUIRE(scores_ld >= n && probs_ld >= n, "pf_softmax_rows: leading dimensions must cover n=%d", n);
    PF_REQUIRE(rows < (1L << 31), "pf_softmax_rows: too many rows");
    PF_DISPATCH_16(out_dtype, "pf_softmax_rows",
        hipLaunchKernelGGL((k_softmax_rows<T, 40>), dim3(static_cast<unsigned>(rows)), dim3(256), 0, as_stream(stream),
                           scores, scores_ld, n, scale, static_cast<unsigned short*>(probs), probs_ld));
    PF_CHECK_LAUNCH("pf_softmax_rows");
    return PF_OK;
}

extern "C" pf_status pf_tensor_to_image(const float* x, int n, int C, int h, int w, uint8_t* y, void* stream) {
    PF_REQUIRE(x && y && n > 0 && C > 0 && h > 0 && w > 0, "pf_tensor_to_image: bad arguments");
    const long total = static_cast<long>(n) * C * h * w;
    hipLaunchKernelGGL(k_tensor_to_image, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), x, n, C,
                       static_cast<long>(h) * w, y);
    PF_CHECK_LAUNCH("pf_tensor_to_image");
    return PF_OK;
}

extern "C" pf_status pf_conv_in(const float* x, int n, int cin, int h, int w, const float* wgt,
                                const float* bias, int cout, int wrap, int out_dtype, void* y, void* stream) {
    PF_REQUIRE(x && wgt && y && n > 0 && cin > 0 && h > 0 && w > 0, "pf_conv_in: bad arguments");
    PF_REQUIRE(cout % 8 == 0 && aligned16(wgt) && aligned16(y), "pf_conv_in: cout %% 8 and 16-byte alignment required");
    const long total = static_cast<long>(n) * h * w * (cout / 8);
    if (cin == 4 && w % 4 == 0) {                                 // the UNets' latent input: four pixels per thread
        const long total4 = total / 4;
        const size_t wlds = static_cast<size_t>(36) * cout * sizeof(float);
        PF_REQUIRE(wlds <= 64 * 1024, "pf_conv_in: cout = %d too wide for the weight tile in LDS", cout);
        // items per block: up to CONV_IN4_ITER x 256, but at least ~3 blocks per CU (the panorama's 2 x 64 x 128 latent is 320 blocks)
        const int n_iter = static_cast<int>(std::max<long>(1, std::min<long>(CONV_IN4_ITER, total4 / (256L * 768))));
        const dim3 grid4(cdiv(total4, 256L * n_iter));
        if (out_dtype == PF_F32)
            hipLaunchKernelGGL((k_conv_in4<Bf16, true>), grid4, dim3(256), wlds, as_stream(stream), x, n, h, w,
                               wgt, bias, cout, wrap, y, n_iter);
        else
            PF_DISPATCH_16(out_dtype, "pf_conv_in",
                hipLaunchKernelGGL((k_conv_in4<T, false>), grid4, dim3(256), wlds, as_stream(stream), x, n, h, w,
                                   wgt, bias, cout, wrap, y, n_iter));
        PF_CHECK_LAUNCH("pf_conv_in");
        return PF_OK;
    }
    if (out_dtype == PF_F32)
        hipLaunchKernelGGL((k_conv_in<Bf16, true>), dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), x, n, cin, h, w,
                           wgt, bias, cout, wrap, y);
    else
        PF_DISPATCH_16(out_dtype, "pf_conv_in",
            hipLaunchKernelGGL((k_conv_in<T, false>), dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), x, n, cin, h, w,
                               wgt, bias, cout, wrap, y));
    PF_CHECK_LAUNCH("pf_conv_in");
    return PF_OK;
}

extern "C" pf_status pf_conv_out(const void* x, int dtype, int n, int cin, int h, int w, const float* wgt,
                                 const float* bias, int cout, int wrap, float* y, void* stream) {
    PF_REQUIRE(x && wgt && y && n > 0 && h > 0 && w > 0, "pf_conv_out: bad arguments");
    PF_REQUIRE(cin % 8 == 0 && cout > 0 && cout <= 8, "pf_conv_out: cin %% 8 == 0 and cout <= 8 required");
    PF_REQUIRE(aligned16(x) && aligned16(wgt), "pf_conv_out: 16-byte alignment required");
    const long npix = static_cast<long>(n) * h * w;
    const int oct = cin / 8;
    // lanes per output pixel: 8 (every lane walks cin / 64 octets of each tap; the 8 pixels of a wavefront read the SAME weight
    // addresses in one instruction, and no lane idles: at the UNets' 320 channels the power-of-two rule gave 64 lanes, 40 of them
    // busy, one pixel per wavefront, 0.5 ms per denoiser pass).  PF_CONV_OUT_LPP: A/B.
    static const int lpp_env = getenv("PF_CONV_OUT_LPP") ? atoi(getenv("PF_CONV_OUT_LPP")) : 0;
    const int lpp = lpp_env ? lpp_env : (oct <= 16 ? 8 : oct <= 64 ? 8 : 16);     // lanes per output pixel
#define PF_CONV_OUT(L) hipLaunchKernelGGL((k_conv_out<TI, L>), dim3(cdiv(npix, 4 * (64 / L))), dim3(256), 0, as_stream(stream), \
                                          static_cast<const In8<TI>::elem*>(x), n, cin, h, w, wgt, bias, cout, wrap, y)
    PF_DISPATCH_IN(dtype, "pf_conv_out",
        if (lpp == 8) PF_CONV_OUT(8); else if (lpp == 16) PF_CONV_OUT(16); else if (lpp == 32) PF_CONV_OUT(32); else PF_CONV_OUT(64));
#undef PF_CONV_OUT
    PF_CHECK_LAUNCH("pf_conv_out");
    return PF_OK;
}

extern "C" pf_status pf_conv_out_gn(const float* x, int n, int cin, int h, int w, const float* scale, const float* shift, int act,
                                    const float* wgt_t, const float* bias, int cout, int wrap, float* y, void* stream) {
    PF_REQUIRE(x && scale && shift && wgt_t && y && n > 0 && h > 0 && w > 0, "pf_conv_out_gn: bad arguments");
    PF_REQUIRE(cin % COG_CH == 0 && cout > 0 && cout <= 4, "pf_conv_out_gn: cin %% %d == 0 and cout <= 4 required", COG_CH);
    PF_REQUIRE(aligned16(x) && aligned16(wgt_t) && aligned16(scale) && aligned16(shift), "pf_conv_out_gn: 16-byte alignment required");
    PF_REQUIRE(n <= 65535 && cdiv(h, COG_TH) <= 65535, "pf_conv_out_gn: too many images / rows for one launch");
    hipLaunchKernelGGL(k_conv_out_gn, dim3(cdiv(w, COG_TW), cdiv(h, COG_TH), n), dim3(256), 0, as_stream(stream),
                       x, cin, h, w, scale, shift, act, wgt_t, bias, cout, wrap, y);
    PF_CHECK_LAUNCH("pf_conv_out_gn");
    return PF_OK;
}
