// How many independent VALU instructions hide under one v_mfma_f32_32x32x16_f16, with one or two waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip && ./issue_probe
// Every wave loops over {MFMA, K fillers} x 4 (four independent accumulators, fillers on independent registers, order pinned
// with sched_barrier); prints shader cycles per MFMA of wave 0 for K = 0..12 and filler kinds fma / exp2 / cvt_pk+fma_mix.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

template <int K, int KIND, int DEP>
__global__ __launch_bounds__(512) void probe(int iters, float* out, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(1.0f - i * 0.01f); }
    floatx16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = lane * 0.01f + i * 0.001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[DEP ? 0 : j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[DEP ? 0 : j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float& x = v[(j * K + i) & 15];
                if (KIND == 0) x = __builtin_fmaf(x, 0.999f, 0.001f);
                else if (KIND == 1) x = __builtin_amdgcn_exp2f(x);
                else {
                    float2v p = {x, x + 1.0f};
                    union { half2v h; unsigned u; } c; c.h = __builtin_convertvector(p, half2v);
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(x) : "v"(c.u), "v"(x));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int K, int KIND, int DEP>
static void run(int threads) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((probe<K, KIND, DEP>), dim3(256), dim3(threads), 0, 0, 10, out, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<K, KIND, DEP>), dim3(256), dim3(threads), 0, 0, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (iters * 4.0);
    printf("waves/SIMD %d  kind %d  dep %d  K %2d: %6.1f cycles per MFMA per wave  (%6.1f per MFMA on the SIMD), %7.1f us\n", threads / 256, KIND, DEP, K, per,
           per / (threads / 256), ms * 1e3);
    hipFree(out); hipFree(cyc);
}

template <int KIND, int DEP>
static void sweep(int threads) {
    run<0, KIND, DEP>(threads); run<2, KIND, DEP>(threads); run<4, KIND, DEP>(threads); run<5, KIND, DEP>(threads);
    run<6, KIND, DEP>(threads); run<8, KIND, DEP>(threads); run<12, KIND, DEP>(threads);
}
int main() {
    for (int threads : {256, 512}) { sweep<0, 0>(threads); sweep<1, 0>(threads); sweep<2, 0>(threads); sweep<0, 1>(threads); }
    return 0;
}
