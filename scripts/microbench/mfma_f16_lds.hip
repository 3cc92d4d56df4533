// Developer microbenchmark (GPU box): what an LDS read costs a SIMD that is busy with fp16 MFMAs on gfx950.
// A wave issues blocks of 4 independent v_mfma_f32_32x32x16_f16 with R ds_read_b128 between them (the candidate kernel's ratio
// is 4 reads per 4 MFMAs); 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_f16_lds.hip -o mfma_f16_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R>
__global__ __launch_bounds__(512) void kern(float* out, int iters, float seed) {
    __shared__ f32x4 tile[2048];  // 32 KB
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) tile[i] = f32x4{seed, seed, seed, seed};
    __syncthreads();
    f32x16 a0, a1, a2, a3;
    for (int i = 0; i < 16; ++i) a0[i] = a1[i] = a2[i] = a3[i] = 0.f;
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) {
        x[i] = (_Float16)(seed + threadIdx.x * 1e-3f);
        y[i] = (_Float16)1.0f;
    }
    f32x4 acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a3, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const f32x4 v = tile[(lane + 64 * ((it + k) & 31)) & 2047];
            acc[k & 3] += v[k & 3];  // (one cheap use per read keeps it alive)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + acc[0] + acc[1] + acc[2] + acc[3];
}

template <int R>
void run(int waves_per_simd) {
    const int iters = 20000, blocks = 256, threads = 256 * waves_per_simd;
    float* out;
    (void)hipMalloc(&out, sizeof(float) * blocks * threads);
    hipLaunchKernelGGL((kern<R>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kern<R>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_ms_floor = 4.0 * iters * waves_per_simd * 32.0 / 2.4e6;  // 32 cycles an MFMA at 2.4 GHz
    printf("ds_read_b128 per 4 MFMAs: %2d, waves/SIMD %d: %.3f ms (bare MFMA time at 2.4 GHz %.3f), %.0f TFLOP/s\n", R,
           waves_per_simd, ms, mfma_ms_floor, 32768.0 * 4 * iters * blocks * (threads / 64) / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>(w);
        run<2>(w);
        run<4>(w);
        run<8>(w);
    }
    return 0;
}
