// What does ONE workgroup per CU get out of the two ways of moving a GEMM operand tile into LDS?  The small-grid GEMMs (SLAM scale)
// spend 0.45 - 0.5 us per 24-KB K tile (128 + 64 rows x 128 B) with every CU holding one 4-wave workgroup; 0.18 us of that is MFMA
// issue.  Modes, same bytes, same 3-deep pipeline, no arithmetic:
//   0  global_load_lds_dwordx4 (LDS-DMA, what the kernels use): 6 x 1 KB per wave and tile
//   1  global_load_dwordx4 -> VGPR -> ds_write_b128 (register staged): 6 x 16 B per lane and tile
//   2  global_load_dwordx4 only (no LDS write): the L2 -> CU path by itself
// Each workgroup walks its own sequence of tiles (fresh bytes every tile, like weights).  Prints KB / us per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o bin/operand_path_probe operand_path_probe.hip && bin/operand_path_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define TILE_BYTES (192 * 128)

template <int MODE, int STAGES = 3, bool MIXED = false, bool PRODUCER = false>
__global__ __launch_bounds__(256) void probe(const char* src, int tiles, unsigned long long* stamps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * tiles * TILE_BYTES;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint4 keep = make_uint4(0, 0, 0, 0);
    auto issue = [&](int t, int stage) {
        const char* g = base + (size_t)t * TILE_BYTES;
        char* l = smem + stage * TILE_BYTES;
        if (PRODUCER && wave != 0) return;
#pragma unroll
        for (int s = 0; s < (PRODUCER ? 24 : 6); ++s) {
            const int slot = PRODUCER ? s : wave + 4 * s;                  // 24 slots of 1 KB (PRODUCER: wave 0 issues all of them)
            if (MODE == 0) {
                // MIXED: slots 0..15 (the activation rows of a 128x64 tile) come from ONE region every workgroup shares (L2-hot),
                // slots 16..23 (the weight rows) are fresh bytes
                const char* gs = (MIXED && slot < 16) ? src + (size_t)(t & 7) * TILE_BYTES : g;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gs + slot * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(l + slot * 1024), 16, 0, 0);
            }
        }
    };
    if (MODE == 0) {
        for (int q = 0; q < STAGES - 1; ++q) if (q < tiles) issue(q, q);
        for (int t = 0; t < tiles; ++t) {
            int rem = tiles - 1 - t; if (rem > STAGES - 2) rem = STAGES - 2;
            if (PRODUCER) {
                if (rem == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                else if (rem == 1) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(48)\n\ts_barrier" ::: "memory");
            } else if (rem == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
            else if (rem == 2) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(18)\n\ts_barrier" ::: "memory");
            if (t + STAGES - 1 < tiles) issue(t + STAGES - 1, (t + STAGES - 1) % STAGES);
            // consume: one LDS read per lane so the tile is "used"
            keep.x ^= *reinterpret_cast<const unsigned*>(smem + (t % STAGES) * TILE_BYTES + tid * 4);
        }
    } else {
        uint4 r[2][6];
        auto ld = [&](int t, int b) {
            const char* g = base + (size_t)t * TILE_BYTES;
#pragma unroll
            for (int s = 0; s < 6; ++s) r[b][s] = *reinterpret_cast<const uint4*>(g + (wave + 4 * s) * 1024 + lane * 16);
        };
        ld(0, 0);
        for (int t = 0; t < tiles; t += 2) {
            if (t + 1 < tiles) ld(t + 1, 1);
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                if (MODE == 1) *reinterpret_cast<uint4*>(smem + (t % STAGES) * TILE_BYTES + (wave + 4 * s) * 1024 + lane * 16) = r[0][s];
                else { keep.x ^= r[0][s].x; keep.y ^= r[0][s].w; }
            }
            __syncthreads();
            if (MODE == 1) keep.x ^= *reinterpret_cast<const unsigned*>(smem + (t % STAGES) * TILE_BYTES + tid * 4);
            if (t + 2 < tiles) ld(t + 2, 0);
            if (t + 1 < tiles) {
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    if (MODE == 1) *reinterpret_cast<uint4*>(smem + ((t + 1) % STAGES) * TILE_BYTES + (wave + 4 * s) * 1024 + lane * 16) = r[1][s];
                    else { keep.x ^= r[1][s].x; keep.y ^= r[1][s].w; }
                }
                __syncthreads();
                if (MODE == 1) keep.x ^= *reinterpret_cast<const unsigned*>(smem + ((t + 1) % STAGES) * TILE_BYTES + tid * 4);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (keep.x == 0x12345678u && keep.y == 0x9abcdef0u) sink[tid] = 1.f;      // never true: keeps the loads alive
    if (tid == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

template <int MODE, int STAGES = 3, bool MIXED = false, bool PRODUCER = false>
static void run(const char* name, const char* src, int wgs, int tiles, unsigned long long* stamps, float* sink) {
    hipFuncSetAttribute((const void*)probe<MODE, STAGES, MIXED, PRODUCER>, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * TILE_BYTES);
    std::vector<double> med;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL((probe<MODE, STAGES, MIXED, PRODUCER>), dim3(wgs), dim3(256), STAGES * TILE_BYTES, 0, src, tiles, stamps, sink);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * wgs);
        hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * wgs, hipMemcpyDeviceToHost);
        std::vector<double> us(wgs);
        for (int i = 0; i < wgs; ++i) us[i] = (double)(h[2 * i + 1] - h[2 * i]) / 100.0;      // 100 MHz
        std::sort(us.begin(), us.end());
        med.push_back(us[wgs / 2]);
    }
    std::sort(med.begin(), med.end());
    const double t = med[2];
    printf("%-44s %4d WGs x %3d tiles: %7.2f us per workgroup = %6.3f us per 24-KB tile = %6.1f KB/us per CU\n", name, wgs, tiles, t, t / tiles,
           tiles * (TILE_BYTES / 1024.0) / t);
}

int main() {
    const int wgs = 256, max_tiles = 32;
    char* src; unsigned long long* stamps; float* sink;
    hipMalloc(&src, (size_t)wgs * max_tiles * TILE_BYTES);
    hipMemset(src, 1, (size_t)wgs * max_tiles * TILE_BYTES);
    hipMalloc(&stamps, sizeof(unsigned long long) * 2 * wgs);
    hipMalloc(&sink, 4096);
    for (int tiles : {4, 11, 32}) {
        run<0>("LDS-DMA (global_load_lds_dwordx4)", src, wgs, tiles, stamps, sink);
        run<1>("register staged (global_load + ds_write_b128)", src, wgs, tiles, stamps, sink);
        run<2>("global_load_dwordx4 only", src, wgs, tiles, stamps, sink);
    }
    printf("-- mixed source: 16 KB of every tile from a region all workgroups share (activations), 8 KB fresh (weights)\n");
    for (int tiles : {11, 32}) {
        run<0, 3, true>("LDS-DMA mixed, 3 stages", src, wgs, tiles, stamps, sink);
        run<0, 4, true>("LDS-DMA mixed, 4 stages", src, wgs, tiles, stamps, sink);
        run<0, 5, true>("LDS-DMA mixed, 5 stages", src, wgs, tiles, stamps, sink);
    }
    run<0, 5, false>("LDS-DMA fresh, 5 stages", src, wgs, 32, stamps, sink);
    printf("-- one producer wave issues all 24 DMA instructions of a tile (the other three only meet it at the barrier)\n");
    for (int tiles : {11, 32}) run<0, 3, true, true>("LDS-DMA mixed, 3 stages, one producer wave", src, wgs, tiles, stamps, sink);
    for (int w : {64, 128}) {
        run<0>("LDS-DMA (global_load_lds_dwordx4)", src, w, 32, stamps, sink);
        run<1>("register staged (global_load + ds_write_b128)", src, w, 32, stamps, sink);
    }
    return 0;
}
