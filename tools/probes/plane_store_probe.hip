// Is the bilinear kernel's STORE pattern what holds it at 2.4 TB/s?  1.6 GB written (a) linearly, 16 B per lane (fill), (b) in the
// blocked plane layout [C/32][rows][hi 64 B | lo 64 B] with the bilinear kernel's thread mapping (16 lanes = the 16 channel
// groups of one pixel: 4 planes x 4 pixels per wave instruction, hi and lo halves of a line by two instructions), (c) the same
// layout with a lane mapping that writes whole 128-B lines per instruction (lane = (pixel, 16-B chunk of the line)).
//   hipcc --offload-arch=gfx950 -O3 -o bin/plane_store_probe plane_store_probe.hip && bin/plane_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_fill(uint4* out, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n16; i += step) out[i] = make_uint4(1, 2, 3, 4);
}
// one workgroup per output row of Wc pixels, C channels (c8 = C / 8 groups): the bilinear kernel's mapping
__global__ __launch_bounds__(256) void k_planes(char* out, int rows_per_img_total, int Wc, int c8, size_t plane_rows) {
    const size_t orow = (size_t)blockIdx.x * Wc;
    for (int i = threadIdx.x; i < Wc * c8; i += 256) {
        const int x = i / c8, c = (i - x * c8) * 8;
        const size_t o = ((size_t)(c >> 5) * plane_rows + orow + x) * 128 + (c & 31) * 2;
        *reinterpret_cast<uint4*>(out + o) = make_uint4(1, 2, 3, 4);
        *reinterpret_cast<uint4*>(out + o + 64) = make_uint4(5, 6, 7, 8);
    }
}
// same bytes, whole lines per instruction: lane l of a wave -> pixel l >> 3 of an 8-pixel group, chunk l & 7 of the 128-B line
__global__ __launch_bounds__(256) void k_lines(char* out, int rows_per_img_total, int Wc, int c8, size_t plane_rows) {
    const size_t orow = (size_t)blockIdx.x * Wc;
    const int nblk = c8 / 4;
    for (int i = threadIdx.x; i < Wc * nblk * 8; i += 256) {
        const int chunk = i & 7, pg = i >> 3;             // pg = (pixel, block)
        const int x = pg / nblk, blk = pg - x * nblk;
        const size_t o = ((size_t)blk * plane_rows + orow + x) * 128 + chunk * 16;
        *reinterpret_cast<uint4*>(out + o) = make_uint4(1, 2, 3, 4);
    }
}
int main() {
    const int n = 16, Hc = 384, Wc = 512, C = 128, c8 = C / 8;
    const size_t plane_rows = (size_t)n * Hc * Wc, bytes = plane_rows * C * 4;
    char* out; hipMalloc(&out, bytes + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-44s %7.1f us  %5.2f TB/s\n", name, best * 1e3, bytes / best / 1e9);
    };
    time("linear fill, 16 B per lane", [&] { hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (uint4*)out, bytes / 16); });
    time("plane layout, bilinear thread mapping", [&] { hipLaunchKernelGGL(k_planes, dim3(n * Hc), dim3(256), 0, 0, out, 0, Wc, c8, plane_rows); });
    time("plane layout, whole lines per instruction", [&] { hipLaunchKernelGGL(k_lines, dim3(n * Hc), dim3(256), 0, 0, out, 0, Wc, c8, plane_rows); });
    return 0;
}
