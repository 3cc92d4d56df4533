// Developer microbenchmark (GPU box): does FP64 vector arithmetic run under FP64 MFMAs on gfx950?
// A wave issues blocks of 4 independent v_mfma_f64_16x16x4_f64 with V FP64 FMAs (or FP32 FMAs) between them; one or two
// waves per SIMD.  Prints cycles per MFMA.   hipcc --offload-arch=gfx950 -O3 mfma_f64_valu.hip -o mfma_f64_valu && ./mfma_f64_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int V, bool F32>
__global__ __launch_bounds__(512) void kern(double* out, long long* cyc, int iters, double seed) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = seed + threadIdx.x * 1e-9, y = 1.0 + 1e-12;
    double v[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = seed * (i + 1);
        f[i] = (float)seed * (i + 1);
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if (F32) f[k & 7] = __builtin_fmaf(f[k & 7], 1.0000001f, 1e-7f);
            else v[k & 7] = __builtin_fma(v[k & 7], 1.0000000001, 1e-10);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, bool F32>
void run(int waves_per_simd) {
    const int iters = 20000, blocks = 256, threads = 256 * waves_per_simd;
    double* out;
    long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipLaunchKernelGGL((kern<V, F32>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.5);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<V, F32>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.5);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (long long c : h) avg += (double)c;
    avg /= blocks;
    // MFMAs per SIMD = waves_per_simd * iters * 4
    printf("%s VALU ops per 4 MFMAs: %3d, waves/SIMD %d: %.1f shader cycles per MFMA slot (per SIMD), %.3f ms, %.1f TFLOP/s FP64 matrix\n",
           F32 ? "FP32" : "FP64", V, waves_per_simd, avg / (iters * 4.0 * waves_per_simd), ms,
           2048.0 * 4 * iters * blocks * (threads / 64) / (ms * 1e-3) / 1e12);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0, false>(w);
        run<8, false>(w);
        run<16, false>(w);
        run<32, false>(w);
        run<64, false>(w);
        run<16, true>(w);
        run<64, true>(w);
    }
    return 0;
}
