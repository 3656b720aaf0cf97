// Layout probe for v_mfma_scale_f32_32x32x64_f8f6f4 (gfx950): which (lane, byte) holds A[i][k] / B[k][j], and how the
// per-lane E8M0 scale bytes apply.  Not part of the product; build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mx_probe tools/probes/mx_probe.hip && /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
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

static uint8_t e4m3(int v) {   // small integers -4..4 exactly
    static const uint8_t mag[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
    return (uint8_t)(mag[v < 0 ? -v : v] | (v < 0 ? 0x80 : 0));
}
int Aval(int i, int k) { return (i + 2 * k) % 5 - 2; }
int Bval(int k, int j) { return (3 * k + j) % 7 - 3; }

// candidate k index held by (lane half h, byte q of the 32 operand bytes)
int kmap(int cand, int h, int q) {
    switch (cand) {
        case 0: return 32 * h + q;                                  // half h holds 32 consecutive k
        case 1: return 16 * h + (q % 16) + 32 * (q / 16);           // two K=32 steps, each split 16/16 across halves
        case 2: return 8 * h + (q % 8) + 16 * (q / 8);              // four K=16 steps, each split 8/8
        default: return 4 * h + (q % 4) + 8 * (q / 4);
    }
}

int main() {
    int *da, *db, *dsa, *dsb; float* dout;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dout, 64 * 16 * 4);
    std::vector<float> out(1024);
    for (int cand = 0; cand < 4; ++cand)
        for (int smode = 0; smode < 3; ++smode) {
            std::vector<uint8_t> a(64 * 32), b(64 * 32);
            std::vector<int> sa(64), sb(64);
            for (int l = 0; l < 64; ++l) {
                const int h = l >> 5, r = l & 31;
                for (int q = 0; q < 32; ++q) { const int k = kmap(cand, h, q); a[l * 32 + q] = e4m3(Aval(r, k)); b[l * 32 + q] = e4m3(Bval(k, r)); }
                // smode 0: all scales 2^0.  1: A scale of lane-half 1 = 2^1.  2: B scale of lane-half 1 = 2^2 (byte 0 of the VGPR)
                sa[l] = 127 + (smode == 1 && h == 1 ? 1 : 0);
                sb[l] = 127 + (smode == 2 && h == 1 ? 2 : 0);
            }
            hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
            hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
            hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
            // expectation under "scale of lane-half h applies to the k values that half holds"
            double err = 0, ref_norm = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                    const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    double want = 0;
                    for (int h = 0; h < 2; ++h)
                        for (int q = 0; q < 32; ++q) {
                            const int k = kmap(cand, h, q);
                            double s = 1.0;
                            if (smode == 1 && h == 1) s = 2.0;
                            if (smode == 2 && h == 1) s = 4.0;
                            want += s * Aval(row, k) * Bval(k, col);
                        }
                    err += fabs(out[l * 16 + r] - want); ref_norm += fabs(want);
                }
            printf("cand %d smode %d: sum|err| = %.1f (sum|ref| = %.1f)%s\n", cand, smode, err, ref_norm, err == 0 ? "  <== MATCH" : "");
            if (cand == 0 && smode > 0) {      // decompose: out = x0*S0 + x1*S1 with S_h = sum over the k values of lane-half h
                for (int l : {0, 1, 33}) for (int r : {0, 5}) {
                    const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    double S[2] = {0, 0};
                    for (int h = 0; h < 2; ++h) for (int q = 0; q < 32; ++q) { const int k = kmap(0, h, q); S[h] += Aval(row, k) * Bval(k, col); }
                    printf("   row %2d col %2d: S0 = %6.0f S1 = %6.0f  out = %8.1f\n", row, col, S[0], S[1], out[l * 16 + r]);
                }
            }
        }
    return 0;
}
