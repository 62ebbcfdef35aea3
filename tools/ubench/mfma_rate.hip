// Micro-benchmark: sustained MFMA rate of v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16 with 1 or 2 waves per
// SIMD, independent accumulators, random operands (DVFS is data dependent).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 frag;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ void k16(const unsigned short* in, float* out, int iters) {
    frag a[4], b[5];
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const frag*>(in + (threadIdx.x * 9 + i) * 8);
    for (int j = 0; j < 5; ++j) b[j] = *reinterpret_cast<const frag*>(in + (threadIdx.x * 9 + 4 + j) * 8);
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[i % 5], a[i % 4], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void k32(const unsigned short* in, float* out, int iters) {
    frag a[2], b[5];
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const frag*>(in + (threadIdx.x * 9 + i) * 8);
    for (int j = 0; j < 5; ++j) b[j] = *reinterpret_cast<const frag*>(in + (threadIdx.x * 9 + 4 + j) * 8);
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[i % 5], a[i % 2], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int threads = 512, blocks = 256 * 4;
    std::vector<unsigned short> h(threads * 9 * 8);
    srand(1);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.1f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    unsigned short* d; float* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, blocks * threads * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int thr : {256, 512}) {
        for (int which = 0; which < 2; ++which) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k16<20>, dim3(blocks), dim3(thr), 0, 0, d, o, iters);
                else hipLaunchKernelGGL(k32<5>, dim3(blocks), dim3(thr), 0, 0, d, o, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                // flops per wave per iter: 16x16x32: 40 mfma x 16384; 32x32x16: 20 mfma x 32768  (same)
                double fl = (double)blocks * (thr / 64) * iters * 40.0 * 16384.0;
                if (rep) printf("%s  %d waves/CU (blocks of %d thr, 4 blocks/CU queued)  %.2f ms  %.0f TF/s\n", which ? "32x32x16" : "16x16x32", thr / 64, thr, ms, fl / ms / 1e9);
            }
        }
    }
    return 0;
}
