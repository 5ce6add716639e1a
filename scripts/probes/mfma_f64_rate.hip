// Calibration probe: issue cost (shader cycles, s_memtime) of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950, independent and dependent chains, one wave per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_rate scripts/probes/mfma_f64_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(long long *out, double seed, int reps) {
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = seed + threadIdx.x, y = seed * 0.5;
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {          // 4 independent accumulators
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {          // one dependent chain
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    }
    long long t2 = __builtin_readcyclecounter();
    double f0 = x, f1 = y, f2 = x + 1, f3 = y + 1;
    for (int r = 0; r < reps; r++) {          // 4 independent FMA chains
        f0 = fma(f0, x, y); f1 = fma(f1, x, y); f2 = fma(f2, x, y); f3 = fma(f3, x, y);
    }
    long long t3 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {          // one dependent FMA chain
        f0 = fma(f0, x, y); f0 = fma(f0, x, y); f0 = fma(f0, x, y); f0 = fma(f0, x, y);
    }
    long long t4 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; }
    if (a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1 + f2 + f3 == 12345.678) out[4] = 1;
}
int main() {
    long long *d, h[5]; hipMalloc(&d, 64);
    for (int waves : {1, 2, 4, 8, 12}) {
        const int reps = 1000;
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, 1.5, reps);
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, 1.5, reps);
        hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
        printf("waves per workgroup %2d (one CU): cycles per instruction -- mfma_f64_16x16x4 independent %.1f  dependent %.1f | v_fma_f64 independent %.1f  dependent %.1f\n", waves,
               h[0] / (4.0 * reps), h[1] / (4.0 * reps), h[2] / (4.0 * reps), h[3] / (4.0 * reps));
    }
    return 0;
}
