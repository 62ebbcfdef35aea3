// Micro-benchmark: issue cost (shader clocks per wave64 instruction) of the vector instructions a flash-attention softmax is made of,
// on ONE wave per SIMD and on TWO (the partner running the same stream), measured with s_memtime around unrolled dependent-free runs.
// DESIGN.md section 3.4 prices the attention kernels' vector segment with these numbers.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// every body: 64 independent-ish instructions over 8 rotating registers (no two consecutive ones depend on each other)
#define BODY_EXP32   REP8(asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
#define BODY_EXP16   REP8(asm volatile("v_exp_f16 %0, %0\n\tv_exp_f16 %1, %1\n\tv_exp_f16 %2, %2\n\tv_exp_f16 %3, %3\n\tv_exp_f16 %4, %4\n\tv_exp_f16 %5, %5\n\tv_exp_f16 %6, %6\n\tv_exp_f16 %7, %7" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
#define BODY_RCP32   REP8(asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\tv_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
#define BODY_FMA32   REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %4\n\tv_fma_f32 %3, %3, %4, %5\n\tv_fma_f32 %4, %4, %5, %6\n\tv_fma_f32 %5, %5, %6, %7\n\tv_fma_f32 %6, %6, %7, %0\n\tv_fma_f32 %7, %7, %0, %1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
#define BODY_MAX3    REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %1, %1, %2, %3\n\tv_max3_f32 %2, %2, %3, %4\n\tv_max3_f32 %3, %3, %4, %5\n\tv_max3_f32 %4, %4, %5, %6\n\tv_max3_f32 %5, %5, %6, %7\n\tv_max3_f32 %6, %6, %7, %0\n\tv_max3_f32 %7, %7, %0, %1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
#define BODY_CVTPK   REP8(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n\tv_cvt_pk_f16_f32 %1, %1, %2\n\tv_cvt_pk_f16_f32 %2, %2, %3\n\tv_cvt_pk_f16_f32 %3, %3, %4\n\tv_cvt_pk_f16_f32 %4, %4, %5\n\tv_cvt_pk_f16_f32 %5, %5, %6\n\tv_cvt_pk_f16_f32 %6, %6, %7\n\tv_cvt_pk_f16_f32 %7, %7, %0" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
#define BODY_PKFMA32 REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %1, %1, %2, %3\n\tv_pk_fma_f32 %2, %2, %3, %4\n\tv_pk_fma_f32 %3, %3, %4, %5\n\tv_pk_fma_f32 %4, %4, %5, %6\n\tv_pk_fma_f32 %5, %5, %6, %7\n\tv_pk_fma_f32 %6, %6, %7, %0\n\tv_pk_fma_f32 %7, %7, %0, %1" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));)
#define BODY_PKADD32 REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %1, %1, %2\n\tv_pk_add_f32 %2, %2, %3\n\tv_pk_add_f32 %3, %3, %4\n\tv_pk_add_f32 %4, %4, %5\n\tv_pk_add_f32 %5, %5, %6\n\tv_pk_add_f32 %6, %6, %7\n\tv_pk_add_f32 %7, %7, %0" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));)
#define BODY_PKFMA16 REP8(asm volatile("v_pk_fma_f16 %0, %0, %1, %2\n\tv_pk_fma_f16 %1, %1, %2, %3\n\tv_pk_fma_f16 %2, %2, %3, %4\n\tv_pk_fma_f16 %3, %3, %4, %5\n\tv_pk_fma_f16 %4, %4, %5, %6\n\tv_pk_fma_f16 %5, %5, %6, %7\n\tv_pk_fma_f16 %6, %6, %7, %0\n\tv_pk_fma_f16 %7, %7, %0, %1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
#define BODY_SWAP32  REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %1, %2\n\tv_permlane32_swap_b32 %3, %4\n\tv_permlane32_swap_b32 %5, %6\n\tv_permlane32_swap_b32 %7, %0" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)

#define KERNEL(NAME, BODY)                                                                                   \
    __global__ void NAME(const float* in, float* out, unsigned long long* clk, int iters) {                  \
        float r[8];                                                                                          \
        f32x2 p[8];                                                                                          \
        for (int i = 0; i < 8; ++i) { r[i] = in[threadIdx.x + 64 * i] * 0.01f; p[i] = f32x2{r[i], -r[i]}; } \
        __syncthreads();                                                                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
        for (int it = 0; it < iters; ++it) { BODY }                                                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
        float s = 0;                                                                                         \
        for (int i = 0; i < 8; ++i) s += r[i] + p[i][0] + p[i][1];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                      \
        if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;       \
    }
KERNEL(k_exp32, BODY_EXP32)
KERNEL(k_exp16, BODY_EXP16)
KERNEL(k_rcp32, BODY_RCP32)
KERNEL(k_fma32, BODY_FMA32)
KERNEL(k_max3, BODY_MAX3)
KERNEL(k_cvtpk, BODY_CVTPK)
KERNEL(k_pkfma32, BODY_PKFMA32)
KERNEL(k_pkadd32, BODY_PKADD32)
KERNEL(k_pkfma16, BODY_PKFMA16)
KERNEL(k_swap32, BODY_SWAP32)

// Do the matrix pipe and the vector ALU of ONE SIMD overlap?  Waves 0..3 of a 512-thread workgroup (one per SIMD) issue 32x32x16 MFMAs on
// four independent accumulators, waves 4..7 (their SIMD partners) a vector stream (fma / exp mix of a softmax); each group is timed alone
// (the other group idles at the barrier) and together.
typedef __attribute__((ext_vector_type(8))) _Float16 hfrag;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k_overlap(const float* in, float* out, unsigned long long* clk, int iters, int mode) {   // mode 1: MFMA group only, 2: VALU group only, 3: both
    const int wave = threadIdx.x >> 6;
    float r[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { r[i] = in[(threadIdx.x & 63) + 64 * i] * 0.01f; p[i] = f32x2{r[i], -r[i]}; }
    hfrag a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(r[e] * 0.1f); b[e] = (_Float16)(r[7 - e] * 0.1f); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 3], 0, 0, 0);
            }
    } else if (mode & 2) {
        for (int it = 0; it < iters; ++it) {      // 64 fma + 32 exp per iteration (the softmax mix of one key tile, roughly)
            BODY_FMA32
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();       // (after the accumulators are read: the MFMAs have drained)
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* in; float* out; unsigned long long* clk;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMallocManaged(&clk, 4096 * 8);
    hipMemset(in, 0x3c, 4096 * 4);
    const int iters = 200;
    struct { const char* name; void (*fn)(const float*, float*, unsigned long long*, int); } tests[] = {
        {"v_exp_f32", k_exp32}, {"v_exp_f16", k_exp16}, {"v_rcp_f32", k_rcp32}, {"v_fma_f32", k_fma32}, {"v_max3_f32", k_max3},
        {"v_cvt_pk_f16_f32", k_cvtpk}, {"v_pk_fma_f32", k_pkfma32}, {"v_pk_add_f32", k_pkadd32}, {"v_pk_fma_f16", k_pkfma16},
        {"v_permlane32_swap_b32", k_swap32}};
    printf("%-24s %22s %22s\n", "instruction (wave64)", "clocks, 1 wave / SIMD", "clocks, 2 waves / SIMD");
    for (auto& t : tests) {
        double res[2];
        for (int w = 0; w < 2; ++w) {
            const int threads = w == 0 ? 256 : 512;                       // 4 or 8 waves in ONE workgroup on one CU
            hipLaunchKernelGGL(t.fn, dim3(256), dim3(threads), 0, 0, in, out, clk, iters);
            hipDeviceSynchronize();
            double s = 0; const int n = 256 * threads / 64;
            for (int i = 0; i < n; ++i) s += (double)clk[i];
            res[w] = s / n / (iters * 64.0);
        }
        printf("%-24s %22.2f %22.2f\n", t.name, res[0], res[1]);
    }
    printf("\nmatrix pipe and vector ALU of one SIMD (512 threads: waves 0-3 MFMA 32x32x16 x16 per iteration, waves 4-7 64 v_fma + 32 v_exp per iteration)\n");
    double alone[2] = {0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
        hipLaunchKernelGGL(k_overlap, dim3(256), dim3(512), 0, 0, in, out, clk, iters, mode);
        hipDeviceSynchronize();
        double m = 0, v = 0;
        for (int bl = 0; bl < 256; ++bl)
            for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)clk[bl * 8 + w];
        m /= 256 * 4 * (double)iters; v /= 256 * 4 * (double)iters;
        if (mode == 1) alone[0] = m;
        if (mode == 2) alone[1] = v;
        printf("  mode %d (%s): MFMA waves %7.1f clocks per 16 MFMAs, vector waves %7.1f clocks per 96 instructions\n", mode,
               mode == 1 ? "MFMA group alone" : mode == 2 ? "vector group alone" : "both", mode & 1 ? m : 0.0, mode & 2 ? v : 0.0);
    }
    printf("  (perfect overlap: both = max of the two alone; no overlap: both = their sum %.1f)\n", alone[0] + alone[1]);
    return 0;
}
