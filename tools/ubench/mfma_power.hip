// Micro-benchmark: what the POWER CAP leaves of the dense 16-bit MFMA peak.  256 workgroups x 8 waves (two per SIMD) issue nothing but
// v_mfma_f32_32x32x16 (or v_mfma_f32_16x16x32) on register-resident operands -- no LDS, no memory traffic -- with operands that are zero,
// constant, or N(0,1) fp16 / bf16 values.  Reports TF/s, shader clocks per MFMA and the effective clock (s_memtime clocks / wall time):
// the chip clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"), so the same instruction stream runs slower on random operands.
// This is the ceiling the conv / GEMM kernels of pf_gemm.hip / pf_gemm32.hip are priced against in DESIGN.md section 5.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 fragh;
typedef __attribute__((ext_vector_type(8))) __bf16 fragb;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

template <bool BF>
__global__ __launch_bounds__(512, 1) void k32(const unsigned short* in, float* out, unsigned long long* clk, int iters) {
    u16x8 a[2], b[5];
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const u16x8*>(in + (threadIdx.x * 7 + i) * 8);
    for (int j = 0; j < 5; ++j) b[j] = *reinterpret_cast<const u16x8*>(in + (threadIdx.x * 7 + 2 + j) * 8);
    f32x16 acc[10];
    for (int i = 0; i < 10; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                if constexpr (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fragb, b[i / 2]), __builtin_bit_cast(fragb, a[i % 2]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fragh, b[i / 2]), __builtin_bit_cast(fragh, a[i % 2]), acc[i], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 10; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <bool BF>
__global__ __launch_bounds__(512, 1) void k16(const unsigned short* in, float* out, unsigned long long* clk, int iters) {
    u16x8 a[4], b[5];
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const u16x8*>(in + (threadIdx.x * 9 + i) * 8);
    for (int j = 0; j < 5; ++j) b[j] = *reinterpret_cast<const u16x8*>(in + (threadIdx.x * 9 + 4 + j) * 8);
    f32x4 acc[20];
    for (int i = 0; i < 20; ++i) acc[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                if constexpr (BF) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(fragb, b[i % 5]), __builtin_bit_cast(fragb, a[i / 5]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fragh, b[i % 5]), __builtin_bit_cast(fragh, a[i / 5]), acc[i], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 20; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); return (u + 0x7FFF + ((u >> 16) & 1)) >> 16; }
static float gauss() { float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX; return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }

int main() {
    const int threads = 512, blocks = 256;
    const size_t n = (size_t)threads * 9 * 8;
    unsigned short* d; float* o; unsigned long long* c;
    hipMalloc(&d, n * 2); hipMalloc(&o, blocks * threads * 4); hipMalloc(&c, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[] = {"zeros", "ones", "N(0,1) scaled 1/64", "N(0,1)"};
    printf("%-10s %-5s %-20s %9s %9s %12s %8s\n", "mfma", "type", "operands", "ms", "TF/s", "clk/MFMA*", "GHz");
    for (int which = 0; which < 2; ++which)
        for (int bf = 0; bf < 2; ++bf)
            for (int data = 0; data < 4; ++data) {
                std::vector<unsigned short> h(n);
                srand(1);
                for (auto& v : h) {
                    const float f = data == 0 ? 0.f : data == 1 ? 1.f : data == 2 ? gauss() / 64 : gauss();
                    v = bf ? f2b(f) : f2h(f);
                }
                hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
                float ms = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (which == 0) { if (bf) hipLaunchKernelGGL(k32<true>, dim3(blocks), dim3(threads), 0, 0, d, o, c, iters); else hipLaunchKernelGGL(k32<false>, dim3(blocks), dim3(threads), 0, 0, d, o, c, iters); }
                    else { if (bf) hipLaunchKernelGGL(k16<true>, dim3(blocks), dim3(threads), 0, 0, d, o, c, iters); else hipLaunchKernelGGL(k16<false>, dim3(blocks), dim3(threads), 0, 0, d, o, c, iters); }
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                std::vector<unsigned long long> hc(blocks);
                hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost);
                double clk = 0; for (auto v : hc) clk += (double)v; clk /= blocks;
                const double nm = which == 0 ? 20.0 : 40.0;                 // MFMAs per wave and iteration (same FLOPs)
                const double fl = (double)blocks * 8 * iters * nm * (which == 0 ? 32768.0 : 16384.0);
                // clk/MFMA*: clocks of the SIMD's matrix pipe per instruction = wave clocks / (2 waves x MFMAs per wave)
                printf("%-10s %-5s %-20s %9.2f %9.0f %12.1f %8.2f\n", which == 0 ? "32x32x16" : "16x16x32", bf ? "bf16" : "fp16", names[data], ms, fl / ms / 1e9,
                       clk / (iters * nm * 2), clk / (ms * 1e6));
            }
    return 0;
}
