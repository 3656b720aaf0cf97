// boundary_probe: what does ONE dependent kernel boundary cost on this box, and which property of a launch moves it?
//
// the SLAM-scale chains of this library advance at ~8.5 us per dispatch and the shortest kernels read
// 4.7 - 5.7 us in a rocprofv3 timeline, while MI355X_MICROARCH.md's `boundary` row says 1.45 us between trivial dependent
// 256-workgroup kernels.  This probe times chains of N dependent launches on one stream (host wall between two stream
// synchronisations, and a HIP-event pair around the chain; eager and as a captured hipGraph) and sweeps, one factor at a time:
//   grid (1 / 256 / 1024 workgroups), block (64 / 256 / 512 threads), kernarg (8 B vs a 448-B by-value struct like GemmParams),
//   dynamic LDS (0 / 80 / 160 KiB behind hipFuncSetAttribute), what the kernel does (nothing; load what the predecessor stored
//   -> store: one dependent memory round trip; that plus a block reduction and a second store = a LayerNorm-shaped body),
//   dirty bytes the predecessor leaves (0 / 1 / 8 / 32 MB), an event pair around every launch, the stream kind (null stream,
//   hipStreamCreate, non-blocking; tools/boundary_probe.py adds torch's current stream inside a torch process next to
//   libsta_mi355.so).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/boundary_probe tools/probes/boundary_probe.hip            (standalone)
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared -DBP_SHARED -o tools/probes/bin/libboundary_probe.so ...       (for the runner)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return -1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));   // a native vector: HIP's float4 struct made hipcc stage the value in LDS (promote-alloca) and read the
                                                        // dispatch packet for the flat thread id - a 12-us scalar load from host memory per wave (first probe run)
struct BigArg { unsigned long long v[56]; };        // 448 B by value (GemmParams is ~430 B)

extern __shared__ char dyn_lds[];

__global__ void k_empty(const float* in, float* out) {
    if (in == (const float*)1) out[0] = 1.0f;       // never true: the kernel touches no memory
}
__global__ void k_empty_big(BigArg a) {
    if (a.v[55] == 0x1234567ull) ((float*)a.v[0])[0] = 1.0f;
}
__global__ void k_lds(const float* in, float* out) {
    if (in == (const float*)1) { dyn_lds[threadIdx.x] = 1; __syncthreads(); out[0] = dyn_lds[0]; }
}
// one dependent memory round trip: every thread loads 16 B the PREDECESSOR stored and stores 16 B the successor loads
__global__ void k_copy(const f4* in, f4* out, int n4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { f4 v = in[i]; v.x += 1.0f; out[i] = v; }
}
// the same bytes as 4-B accesses (thread t of a block: floats t, t + 256, t + 512, t + 768 of the block's 4 KB)
__global__ void k_copy_dw(const float* in, float* out, int n) {
    const int b = blockIdx.x * 1024;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = in[b + k * 256 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) out[b + k * 256 + threadIdx.x] = v[k] + 1.0f;
}
// 16-B loads only (the store never happens), 16-B stores only
__global__ void k_load16(const f4* in, f4* out, int n4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { f4 v = in[i]; if (v.x == 123456.75f) out[i] = v; }
}
__global__ void k_store16(const f4* in, f4* out, int n4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { f4 v = {(float)i, 1.f, 2.f, 3.f}; out[i] = v; }
}
// the first run's accident, kept as a row: HIP's float4 struct -> the value is staged in LDS and the kernel reads the AQL dispatch
// packet (host memory) for its flat thread id
__global__ void k_copy_hipfloat4(const float4* in, float4* out, int n4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) { float4 v = in[i]; v.x += 1.0f; out[i] = v; }
}
// LayerNorm-shaped: one block per row of `cols` floats: load, block reduction (mean, variance), second pass, store
__global__ void k_rowstat(const float* in, float* out, int cols) {
    __shared__ float red[2][16];
    const float* r = in + (size_t)blockIdx.x * cols;
    float s = 0.f, q = 0.f;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) { const float v = r[c]; s += v; q += v * v; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    s = 0.f; q = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { s += red[0][w]; q += red[1][w]; }
    const float mean = s / cols, rstd = rsqrtf(fmaxf(q / cols - mean * mean, 0.f) + 1e-6f);
    for (int c = threadIdx.x; c < cols; c += blockDim.x) out[(size_t)blockIdx.x * cols + c] = (r[c] - mean) * rstd * 0.5f + 1.0f;
}
// predecessor that leaves `bytes` dirty in the L2s (grid-stride fill)
__global__ void k_dirty(float4* buf, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) buf[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

struct Cfg {
    const char* name;
    int body;          // 0 empty, 1 empty + 448-B kernarg, 2 dynamic LDS, 3 copy, 4 rowstat, 5-9 copy variants
    int grid, block; int lds; size_t dirty; int events;
};

struct Res { double wall_us, ev_us, enq_us, graph_us; };

static float *g_a = nullptr, *g_b = nullptr, *g_d = nullptr;
static const int ROWS = 256, COLS = 1024;
static const size_t DIRTY_MAX = (size_t)32 << 20;

static void launch_one(const Cfg& c, hipStream_t st, int i) {
    const float* in = (i & 1) ? g_b : g_a; float* out = (i & 1) ? g_a : g_b;
    if (c.dirty) hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, st, (float4*)g_d, c.dirty / 16);
    switch (c.body) {
    case 0: hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), 0, st, in, out); break;
    case 1: { BigArg a; memset(&a, 0, sizeof a); a.v[0] = (unsigned long long)out; hipLaunchKernelGGL(k_empty_big, dim3(c.grid), dim3(c.block), 0, st, a); break; }
    case 2: hipLaunchKernelGGL(k_lds, dim3(c.grid), dim3(c.block), c.lds, st, in, out); break;
    case 3: hipLaunchKernelGGL(k_copy, dim3(c.grid), dim3(c.block), 0, st, (const f4*)in, (f4*)out, ROWS * COLS / 4); break;
    case 4: hipLaunchKernelGGL(k_rowstat, dim3(c.grid), dim3(c.block), 0, st, in, out, COLS); break;
    case 5: hipLaunchKernelGGL(k_copy_dw, dim3(c.grid), dim3(c.block), 0, st, in, out, ROWS * COLS); break;
    case 6: hipLaunchKernelGGL(k_load16, dim3(c.grid), dim3(c.block), 0, st, (const f4*)in, (f4*)out, ROWS * COLS / 4); break;
    case 7: hipLaunchKernelGGL(k_store16, dim3(c.grid), dim3(c.block), 0, st, (const f4*)in, (f4*)out, ROWS * COLS / 4); break;
    case 8: hipLaunchKernelGGL(k_copy, dim3(c.grid), dim3(c.block), 0, st, (const f4*)g_d, (f4*)out, ROWS * COLS / 4); break;   // source never written by the chain
    case 9: hipLaunchKernelGGL(k_copy, dim3(c.grid), dim3(c.block), 0, st, (const f4*)in, (f4*)g_d + (i & 1) * (ROWS * COLS / 4), ROWS * COLS / 4); break;   // destination never read by the chain
    case 10: hipLaunchKernelGGL(k_copy_hipfloat4, dim3(c.grid), dim3(c.block), 0, st, (const float4*)in, (float4*)out, ROWS * COLS / 4); break;
    }
}

static int run_cfg(const Cfg& c, hipStream_t st, int n, Res* r) {
    using clk = std::chrono::steady_clock;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<hipEvent_t> per;
    if (c.events) { per.resize(2 * n); for (auto& e : per) CK(hipEventCreate(&e)); }
    for (int i = 0; i < 20; ++i) launch_one(c, st, i);
    CK(hipStreamSynchronize(st));
    double best_wall = 1e30, best_ev = 1e30, best_enq = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipStreamSynchronize(st));
        auto t0 = clk::now();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < n; ++i) {
            if (c.events) CK(hipEventRecord(per[2 * i], st));
            launch_one(c, st, i);
            if (c.events) CK(hipEventRecord(per[2 * i + 1], st));
        }
        CK(hipEventRecord(e1, st));
        auto t1 = clk::now();
        CK(hipStreamSynchronize(st));
        auto t2 = clk::now();
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double wall = std::chrono::duration<double, std::micro>(t2 - t0).count() / n;
        const double enq = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
        if (wall < best_wall) best_wall = wall;
        if (ms * 1e3 / n < best_ev) best_ev = ms * 1e3 / n;
        if (enq < best_enq) best_enq = enq;
    }
    r->wall_us = best_wall; r->ev_us = best_ev; r->enq_us = best_enq; r->graph_us = -1;
    // the same chain as a captured graph (no host launch cost between the nodes); the null stream cannot be captured
    if (st != nullptr && !c.events) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < n; ++i) launch_one(c, st, i);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        double bg = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms * 1e3 / n < bg) bg = ms * 1e3 / n;
        }
        r->graph_us = bg;
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    for (auto& e : per) (void)hipEventDestroy(e);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

static int setup() {
    if (g_a) return 0;
    CK(hipMalloc((void**)&g_a, (size_t)ROWS * COLS * 4)); CK(hipMalloc((void**)&g_b, (size_t)ROWS * COLS * 4)); CK(hipMalloc((void**)&g_d, DIRTY_MAX));
    CK(hipMemset(g_a, 0, (size_t)ROWS * COLS * 4)); CK(hipMemset(g_b, 0, (size_t)ROWS * COLS * 4));
    CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

// One sweep on `st` (nullptr = the null stream).  Prints one row per configuration; `tag` names the stream kind.
extern "C" int bp_sweep(void* stream, const char* tag, int n) {
    if (setup()) return -1;
    hipStream_t st = (hipStream_t)stream;
    const Cfg cfgs[] = {
        {"empty 256 WG x 256 thr, 16-B kernarg", 0, 256, 256, 0, 0, 0},
        {"empty   1 WG x 256 thr", 0, 1, 256, 0, 0, 0},
        {"empty 1024 WG x 256 thr", 0, 1024, 256, 0, 0, 0},
        {"empty 256 WG x  64 thr", 0, 256, 64, 0, 0, 0},
        {"empty 256 WG x 512 thr", 0, 256, 512, 0, 0, 0},
        {"empty 256 WG x 256 thr, 448-B kernarg", 1, 256, 256, 0, 0, 0},
        {"empty 256 WG x 512 thr, dyn LDS 0 (attr 160K set)", 2, 256, 512, 0, 0, 0},
        {"empty 256 WG x 512 thr, dyn LDS 80 KiB", 2, 256, 512, 80 * 1024, 0, 0},
        {"empty 256 WG x 512 thr, dyn LDS 160 KiB", 2, 256, 512, 160 * 1024, 0, 0},
        {"copy 1 MB (load predecessor's stores -> store), 256 WG", 3, 256, 256, 0, 0, 0},
        {"copy 1 MB as 4-B accesses", 5, 256, 256, 0, 0, 0},
        {"16-B loads only of the predecessor's buffer (1 MB)", 6, 256, 256, 0, 0, 0},
        {"16-B stores only (1 MB)", 7, 256, 256, 0, 0, 0},
        {"copy 1 MB, source a buffer the chain never writes", 8, 256, 256, 0, 0, 0},
        {"copy 1 MB, destination a buffer the chain never reads", 9, 256, 256, 0, 0, 0},
        {"copy 1 MB through HIP float4 (LDS staging + dispatch-packet read)", 10, 256, 256, 0, 0, 0},
        {"rowstat 256 rows x 1024 (LayerNorm-shaped), 256 WG", 4, 256, 256, 0, 0, 0},
        {"empty 256 WG behind a 1-MB dirty fill (pair)", 0, 256, 256, 0, (size_t)1 << 20, 0},
        {"empty 256 WG behind an 8-MB dirty fill (pair)", 0, 256, 256, 0, (size_t)8 << 20, 0},
        {"empty 256 WG behind a 32-MB dirty fill (pair)", 0, 256, 256, 0, (size_t)32 << 20, 0},
        {"empty 256 WG, event pair around every launch", 0, 256, 256, 0, 0, 1},
        {"rowstat 256 x 1024, event pair around every launch", 4, 256, 256, 0, 0, 1},
    };
    printf("# stream: %s   chain of %d dependent launches; us per launch (pairs: per fill + kernel)\n", tag, n);
    printf("%-62s %9s %9s %9s %9s\n", "configuration", "wall", "events", "enqueue", "graph");
    for (const Cfg& c : cfgs) {
        Res r;
        if (run_cfg(c, st, n, &r)) return -1;
        if (r.graph_us >= 0) printf("%-62s %9.2f %9.2f %9.2f %9.2f\n", c.name, r.wall_us, r.ev_us, r.enq_us, r.graph_us);
        else printf("%-62s %9.2f %9.2f %9.2f %9s\n", c.name, r.wall_us, r.ev_us, r.enq_us, "-");
    }
    fflush(stdout);
    return 0;
}

#ifndef BP_SHARED
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000;
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    if (bp_sweep(nullptr, "null stream (standalone process)", n)) return 1;
    if (bp_sweep(s1, "hipStreamCreate (standalone process)", n)) return 1;
    if (bp_sweep(s2, "hipStreamCreateWithFlags(NonBlocking) (standalone process)", n)) return 1;
    return 0;
}
#endif
