// Do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap?   hipcc --offload-arch=gfx950 -O3 -o probe mfma_valu_probe.hip
// One workgroup per CU, 8 waves = 2 per SIMD (waves w and w+4 share SIMD w).  mode 0: waves 0-3 issue MFMAs, waves 4-7 idle;
// mode 1: waves 4-7 issue VALU work, waves 0-3 idle; mode 2: both; mode 3: every wave alternates 4 MFMAs / 32 VALU ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KIND>   // VALU flavour: 0 = v_fma_f32, 1 = v_exp_f32, 2 = v_cvt f32->f16->f32
__global__ __launch_bounds__(512) void probe(int mode, int iters, float* out, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool do_mfma = (mode == 0 || mode == 2) ? wave < 4 : (mode == 3);
    const bool do_valu = (mode == 1 || mode == 2) ? wave >= 4 : (mode == 3);
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(1.0f - i * 0.01f); }
    floatx16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = lane * 0.01f + i * 0.001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (KIND == 0) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
                else if (KIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.0f;
                else v[i] = (float)(_Float16)v[i] * 0.999f;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
static void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, mode, 10, out, cyc);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("%-10s mode %d: %8.1f us   cycles/iter wave0 %6.1f wave4 %6.1f\n", name, mode, ms * 1e3, (double)h[0] / iters, (double)h[4] / iters);
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    printf("per iteration: 4 independent MFMA 32x32x16 f16 (4 x 32 = 128 MFMA-pipe cycles) and / or 32 VALU ops per wave\n");
    printf("mode 0 = MFMA waves only, 1 = VALU waves only, 2 = MFMA waves + VALU waves on the same SIMDs, 3 = every wave does both\n");
    run<0>("v_fma_f32"); run<1>("v_exp_f32"); run<2>("cvt f16");
    return 0;
}
