// Developer microbenchmark (GPU box): do vector instructions run under fp16 MFMAs (v_mfma_f32_32x32x16_f16) on gfx950?
// A wave issues blocks of 4 independent MFMAs with V FP32 ops (v_min3-like fminf chains or FMAs) between them; 1, 2 or 3
// waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_f16_valu.hip -o mfma_f16_valu && ./mfma_f16_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(768) void kern(float* out, long long* cyc, int iters, float seed) {
    f32x16 a0, a1, a2, a3;
    for (int i = 0; i < 16; ++i) a0[i] = a1[i] = a2[i] = a3[i] = 0.f;
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) {
        x[i] = (_Float16)(seed + threadIdx.x * 1e-3f);
        y[i] = (_Float16)1.0f;
    }
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = seed * (i + 1);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a3, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < V; ++k) f[k & 7] = __builtin_fmaf(f[k & 7], 1.0000001f, 1e-7f);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(int waves_per_simd) {
    const int iters = 20000, blocks = 256, threads = 256 * waves_per_simd;
    float* out;
    long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipLaunchKernelGGL((kern<V>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<V>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // 32x32x16 MFMA = 32768 flop
    printf("FP32 VALU ops per 4 MFMAs: %3d, waves/SIMD %d: %.3f ms, %.0f TFLOP/s fp16 matrix\n", V, waves_per_simd, ms,
           32768.0 * 4 * iters * blocks * (threads / 64) / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 3; ++w) {
        run<0>(w);
        run<8>(w);
        run<16>(w);
        run<32>(w);
        run<64>(w);
    }
    return 0;
}
