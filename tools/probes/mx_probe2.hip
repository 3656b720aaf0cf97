// Scale-block membership probe for v_mfma_scale_f32_32x32x64_f8f6f4: A is one-hot in operand position (lane half hp,
// byte q) for all rows, B is all ones, and exactly one lane half hs carries A-scale 2^1 (all 32 lanes of that half).
// out[row][col] = 2 if position (hp, q) belongs to the K block scaled by lanes of half hs, else 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int int8v __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void mx_kernel(const int* a, const int* b, const int* sa, const int* sb, float* out) {
    int8v A, B;
    for (int i = 0; i < 8; ++i) { A[i] = a[threadIdx.x * 8 + i]; B[i] = b[threadIdx.x * 8 + i]; }
    floatx16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = c[i];
}
int main() {
    int *da, *db, *dsa, *dsb; float* dout;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dout, 4096);
    std::vector<float> out(1024);
    std::vector<uint8_t> b(2048, 0x38);   // all ones
    hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
    for (int side = 0; side < 2; ++side)          // 0: probe the A scale, 1: probe the B scale (roles swapped)
        for (int hs = 0; hs < 2; ++hs) {
            printf("%c-scale on lane half %d: operand positions (half, byte) it scales:\n", side ? 'B' : 'A', hs);
            for (int hp = 0; hp < 2; ++hp) {
                printf("   half %d bytes:", hp);
                for (int q = 0; q < 32; ++q) {
                    std::vector<uint8_t> a(2048, 0);
                    for (int l = 0; l < 64; ++l) if ((l >> 5) == hp) a[l * 32 + q] = 0x38;
                    std::vector<int> s1(64, 127), s0(64, 127);
                    for (int l = 0; l < 64; ++l) if ((l >> 5) == hs) s1[l] = 128;
                    if (side == 0) { hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
                                     hipMemcpy(dsa, s1.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, s0.data(), 256, hipMemcpyHostToDevice); }
                    else { hipMemcpy(db, a.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(da, b.data(), 2048, hipMemcpyHostToDevice);
                           hipMemcpy(dsb, s1.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsa, s0.data(), 256, hipMemcpyHostToDevice); }
                    hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
                    hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
                    printf(" %g", out[0]);
                }
                printf("\n");
            }
        }
    // which rows does ONE lane's A scale touch?  A, B all ones; lane L scale 2^1 -> rows whose output != 64
    for (int L : {0, 5, 32, 37}) {
        std::vector<uint8_t> a(2048, 0x38);
        std::vector<int> s1(64, 127), s0(64, 127); s1[L] = 128;
        hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsa, s1.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, s0.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
        hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
        printf("A-scale 2^1 on lane %d only -> rows with out != 64:", L);
        for (int l = 0; l < 64; l += 32) for (int r = 0; r < 16; ++r) if (out[l * 16 + r] != 64.f) printf(" row %d (=%g)", (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), out[l * 16 + r]);
        printf("\n");
    }
    return 0;
}
