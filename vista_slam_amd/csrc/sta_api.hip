// C-ABI implementation of the MI355X STA frontend (see include/sta_mi355.h).
// Host orchestration only: every FLOP runs in the hand-written gfx950 kernels of gemm.h,
// attention.h and elementwise.h.  No torch, no BLAS, no CPU fallback.
#include "../../include/sta_mi355.h"
#ifdef STA_TEST_HOOKS
#include "../../include/sta_mi355_debug.h"      // the test-hooks build exports these too (STA_API = default visibility comes from the declarations)
#endif
#include "gemm.h"
#include "gemm2.h"
#include "conv3h.h"
#include "attention.h"
#include "elementwise.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";
static int set_err(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return -1;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_err("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define CHK(x) do { int r_ = (x); if (r_ != 0) return r_; } while (0)
// Every entry point runs on the handle's device and RESTORES the caller's current HIP device on return (torch keeps its
// own notion of the current device; a library that leaves another one selected silently redirects the caller's next ops).
struct DevScope {
    int prev = -1; bool ok = true;
    explicit DevScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
    }
    ~DevScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define DEV_SCOPE(dev) DevScope dev_scope_(dev); do { if (!dev_scope_.ok) return set_err("hipSetDevice(%d) failed", (int)(dev)); } while (0)
#define REQUIRE(c, ...) do { if (!(c)) return set_err(__VA_ARGS__); } while (0)

// Development builds (python -m vista_slam_amd.build with STA_DEV_FAST=1; never shipped): the single-product f16 forms of the
// MFMA kernels are not instantiated - half the compile time while iterating on the f16x3 path.
#ifdef STA_DEV_FAST
#define STA_F16ONLY(x) return set_err("development build (-DSTA_DEV_FAST): precision f16 kernels are not compiled in")
#else
#define STA_F16ONLY(x) x
#endif

// ------------------------------------------------------------------------------------------ types
// fp16 planes.  Activations / weights use the K-tile-blocked layout [cols/32][rp rows][hi32|lo32]
// (lo == hi + 32 in f16x3, lo == nullptr in f16); rp == 0 marks the row-major Q/K/V^T buffers.
struct Planes { f16* hi = nullptr; f16* lo = nullptr; int64_t rp = 0; bool mx = false; };   // mx: rows in the f16mx format (DPT buffers carry it explicitly)
static inline Planes slice_rows(const Planes& p, int64_t r0) {
    Planes q = p; const int64_t es = p.lo ? 64 : 32;
    q.hi = p.hi + r0 * es; if (p.lo) q.lo = q.hi + 32;
    return q;
}
// layer classes of the precision policy: the DPT head's convolutions may run "f16 main product + one block-scaled fp8 correction
// MFMA" (the f16mx arithmetic, sta_common.h; precision f16x3h); the transformer and the pose head never do (round 3: the
// all-layers form failed the stress goldens and was retired)
enum { CLS_NONE = 0, CLS_HEAD = 16 /* DPT head convolutions */, CLS_FC2 = 32 /* mlp.fc2 of both transformers (precision f16x3m) */ };
struct Lin { Planes w; Planes wmx; float* bias = nullptr; int N = 0, K = 0; int cls = CLS_NONE; };   // wmx: second packed copy in the f16mx row format (head only)
struct LNp { float* g = nullptr; float* b = nullptr; };
struct EncBlk { LNp n1, n2; Lin qkv, proj, fc1, fc2; };
struct DecBlk { LNp n1, n2, n3, ny; Lin qkv, proj, cq, ckv, cproj, fc1, fc2; };
struct RCU { Lin c1, c2; };
struct Refine { Lin out; RCU u1, u2; };
struct F32Lin { float* w = nullptr; float* b = nullptr; };

enum SlotKind { SK_F32, SK_W_ID, SK_W_CONV, SK_W_CONVT, SK_B_CONVT, SK_DROP };
struct Slot {
    std::vector<int64_t> shape;
    SlotKind kind = SK_DROP;
    float* dst32 = nullptr;          // SK_F32 / SK_B_CONVT destination (+offset applied)
    f16* dst_hi = nullptr; f16* dst_lo = nullptr;   // packed destination (+row offset applied)
    f16* dst_mx = nullptr;           // f16mx copy of the same weight (transformer linears)
    int reps = 1;                    // SK_B_CONVT: k*k
    int64_t N = 0, K = 0, n_off = 0; // packed-weight geometry: rows of the whole Lin, contraction length, row offset
    bool loaded = false;
    bool loaded_tiny = false;        // the tensor last loaded into this slot was all below 2^-12 (sta_load_tensor)
};

static const int KSTAMP_WG = 2048, KSTAMP_LAUNCHES = 512;      // in-model stamps (tools/model_stamps.py): workgroups kept per launch, launches per dump
static const int64_t SKBUF_ELEMS = (int64_t)4 << 20;   // 16 MiB: M*N of the largest split-K plane-epilogue GEMM
// Execution context of ONE caller stream: everything a call writes between its kernels.  The handle keeps one context per
// stream it has been called on (stream_ctx), so calls enqueued on DIFFERENT streams never share scratch memory and may run
// concurrently on the GPU - e.g. sta_encode of keyframe i+1 on a second stream under sta_regress_views of keyframe i
// (independent in OnlineSLAM.step: slam.py:258 vs :263-277).  Weights, tables and the zero page are read-only and shared.
static const int MAX_STREAM_CTX = 8;
struct StreamCtx {
    hipStream_t st = nullptr; uint64_t last_use = 0;      // last_use: the handle's use_clock at the latest call (recycling order)
    char* ws = nullptr; int64_t ws_cap = 0;   // bump-allocated workspace, sized by the dry planning pass of the call
    float* skbuf = nullptr;   // fp32 partial sums of split-K GEMMs with the plane / QKV epilogue (SKBUF_ELEMS floats)
    float* slab = nullptr;    // slab split-K of the small-M in-place residual GEMMs (GemmParams::slab), same size
    // split-phase keyframe scheduler (sta_regress_views_begin / _finish): the call pending on this stream
    bool rv_open = false; int rv_k = 0, rv_H = 0, rv_W = 0;
    float* rv_conf = nullptr;         // pinned host copy of the k pose confidences (async D2H in begin, read in finish)
    hipEvent_t rv_ev = nullptr;       // recorded behind that copy
    // side lane of the DPT head (dpt_impl): an internal second stream for the branches of the head that do not lie on its
    // critical chain, with its own split-K scratch; forked from and joined back into `st` inside the call
    hipStream_t side = nullptr; float* side_skbuf = nullptr; hipEvent_t side_ev[4] = {nullptr, nullptr, nullptr, nullptr};
};

struct sta_handle {
    sta_config cfg;
    int device = 0;
    int prec = STA_PREC_F16X3;
    bool deterministic = false;   // sta_set_deterministic: no ATOMIC split-K (the slab forms have a fixed summation order anyway)
    int mx_mask = 0;          // CLS_HEAD when the DPT head runs in the f16mx arithmetic (precision f16x3h), else 0
    bool finalized = false;
    std::unordered_map<std::string, Slot> slots;
    int n_loaded = 0;
    std::vector<void*> allocs;
    int64_t weight_bytes = 0;
    // weights
    Lin patch; std::vector<EncBlk> enc; Lin dec_embed; float* pose_tok = nullptr;
    std::vector<DecBlk> dec; LNp dec_norm, enc_norm;
    Lin act0_0, act0_1, act1_0, act1_1, act2_0, act3_0, act3_1;
    Lin rn[4]; Refine ref[4];      // ref[0] = refinenet1 ... ref[3] = refinenet4
    Lin head0, head2; F32Lin head4;
    float head4_scale[4] = {1.f, 1.f, 1.f, 1.f};   // per output row of head.4: the power of two that brings its largest |weight| into [0.5, 1) - the fused tail contracts fp16 hi / lo planes of the SCALED rows (sta_finalize_weights)
    F32Lin pm0, pm1, pm2, pt, pr, pc;
    // staging + workspace
    float* stage = nullptr; int64_t stage_elems = 0;
    uint64_t use_clock = 0;
    std::vector<StreamCtx> ctx; StreamCtx* cur = nullptr;     // per-stream scratch (stream_ctx); cur = the context of the running call
    f16* zero_page = nullptr;
    unsigned long long* range = nullptr;   // the two range counters (sta_range_report): 16 B of device memory, per handle
    int small_grid_mode = 0;  // tools/tile_table.py only (sta_set_gemm_variant 10 / 11): 1 = never the small-grid family, 2 = 4x the product threshold
    int opt[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // experiment switches (sta_debug_set_option; 0 = product behaviour)
    int tail_hint = 0;      // decode_impl: the last tail_hint rows of every dense GEMM are pose-token rows (GemmParams::m_tail)
    int gemm_variant = 0;   // tests / tools: 0 auto, 1..4 forced GEMM families, 8 = conv3h wherever legal, 9 = auto WITHOUT conv3h (A/B)
    // rope table
    float* rope_tab = nullptr; int rope_P = 0;
    // sta_decode_pos, for the duration of the call: the decoder's QKV epilogues rotate by the identity table and rope_planes_kernel
    // rotates Q / K by the caller's positions afterwards (decode_impl)
    bool rope_foreign = false; const int* rope_pos = nullptr; const float* rope_ident = nullptr;
    // timing
    bool timing = false; hipEvent_t ev[5]; bool ev_ok = false;
    // per-launch timing of the dominant kernel (gemm_kernel<*, A_DENSE, EPI_F32>) for the roofline report
    unsigned long long* clk_buf = nullptr;   // {shader cycles, 100 MHz ticks} summed over sampled workgroups of the timed GEMMs
    bool ktime = false; std::vector<hipEvent_t> kev; int kn = 0; std::vector<double> kflops, kbytes; std::vector<int> kvar;
    int kfilter[4] = {-1, -1, -1, -1};    // mode 3: {epilogue, A-loader, tile family, mx} of the one kernel symbol that is timed
    int kevery = 1, kseen = 0;            // mode 3: every kevery-th matching launch carries the event pair (an event pair costs ~9 us of
                                          // dispatch, tools/probes/boundary_probe.hip: 11.4 vs 2.7 us per launch)
    bool ktime_all = false; std::vector<int> kshape;   // sta_kernel_timing(h, 2): every GEMM / conv launch is timed; {M, N, K, EPI, AMODE, mx} per record
    unsigned long long* kstamp = nullptr; bool kstamp_on = false;   // sta_kernel_timing(h, 4): mode 2 + in-kernel stamps (GemmParams::stamps) of every launch, KSTAMP_WG workgroups x 4 per record
    // f3 input-step tables (one cached geometry)
    int pre_key[6] = {0, 0, 0, 0, 0, 0}; int* pre_tab = nullptr; int64_t pre_cap = 0; int pre_meta[12] = {0};
    bool dry = false;   // planning pass: run the orchestration without launching to size the workspace
    int lanes_mode = STA_LANES_AUTO;   // sta_set_side_lanes
    std::vector<hipStream_t> pipe_streams; int pipe_verified = 0;   // sta_pipeline_streams: library-owned streams probed to overlap pairwise
    int lane = 0;       // 1 while dpt_impl enqueues on the context's side stream (launch_gemm then hands out the side lane's split-K scratch)
    // sta_reserve / sta_alloc_stats: device allocations (hipMalloc / hipFree / hipHostMalloc / stream and event creation count as
    // one each) and device-wide synchronisations the COMPUTE entry points made since sta_create - after sta_reserve neither moves
    int tiny_tensors = 0;     // packed (MFMA-operand) weight tensors whose every value is below 2^-12: their fp16 planes are subnormal (sta_range_report adds them to counts[0])
    int64_t n_alloc = 0, n_devsync = 0;
    bool reserve_only = false;      // inside sta_reserve: plan_and_run sizes and allocates, launches nothing
};

static int dalloc(sta_handle* h, void** p, int64_t bytes) {
    HIPCHK(hipMalloc(p, (size_t)(bytes > 0 ? bytes : 16)));
    h->allocs.push_back(*p);
    h->weight_bytes += bytes;
    return 0;
}

struct Bump {
    char* base; int64_t cap; int64_t off = 0; bool overflow = false; int64_t peak = 0;
    void* take(int64_t bytes) {
        int64_t a = (off + 255) & ~int64_t(255);
        off = a + bytes;
        if (off > peak) peak = off;
        if (off > cap) { overflow = true; return base; }
        return base + a;
    }
    void rewind(int64_t mark) { off = mark; }
    Planes planes(int64_t elems, bool split) {      // row-major pair of planes (Q / K / V^T buffers)
        Planes p; p.hi = (f16*)take(elems * 2); p.lo = split ? (f16*)take(elems * 2) : nullptr; return p;
    }
    Planes act(int64_t rows, int64_t cols, bool split) {   // blocked activation planes [ceil(cols/32)][rows][32 (+32)]
        Planes p; p.rp = rows;
        cols = (cols + 31) & ~int64_t(31);                 // (a 16-column test tensor still occupies whole 32-column blocks)
        p.hi = (f16*)take(rows * cols * (split ? 4 : 2) + 256);
        p.lo = split ? p.hi + 32 : nullptr;
        return p;
    }
};

// The context of stream `st` becomes the current one (created on the first call on that stream: 2 x 16 MiB of split-K scratch;
// the workspace grows with the first call of a shape).  At most MAX_STREAM_CTX contexts per handle: one more stream RECYCLES the
// least recently used context with no split-phase call pending (behind a device synchronisation - work of the old stream may
// still be using that scratch, and the old stream itself may be gone).
static int stream_ctx(sta_handle* h, hipStream_t st) {
    for (auto& c : h->ctx) if (c.st == st) { h->cur = &c; c.last_use = ++h->use_clock; return 0; }
    if ((int)h->ctx.size() >= MAX_STREAM_CTX) {
        StreamCtx* lru = nullptr;
        for (auto& c : h->ctx) if (!c.rv_open && (!lru || c.last_use < lru->last_use)) lru = &c;
        REQUIRE(lru, "all %d scratch contexts of this handle have a split-phase scheduler call pending (sta_regress_views_begin without _finish)", MAX_STREAM_CTX);
        HIPCHK(hipDeviceSynchronize()); h->n_devsync++;
        lru->st = st; lru->last_use = ++h->use_clock;
        h->cur = lru;
        return 0;
    }
    StreamCtx c; c.st = st; c.last_use = ++h->use_clock;
    HIPCHK(hipMalloc((void**)&c.skbuf, (size_t)SKBUF_ELEMS * 4));
    if (hipMalloc((void**)&c.slab, (size_t)SKBUF_ELEMS * 4) != hipSuccess) { hipFree(c.skbuf); return set_err("split-K slab alloc failed"); }
    h->n_alloc += 2;
    h->ctx.push_back(c);          // (reserve()d in sta_create: pointers into the vector stay valid)
    h->cur = &h->ctx.back();
    return 0;
}
static int ensure_ws(sta_handle* h, int64_t bytes, hipStream_t st) {
    CHK(stream_ctx(h, st));
    StreamCtx& c = *h->cur;
    // every path that takes this stream's workspace comes through here (plan_and_run, the kernel-level test entry points):
    // phase B of a pending scheduler call still reads what phase A left in it
    REQUIRE(!c.rv_open, "a scheduler call begun with sta_regress_views_begin is pending on this stream: finish (or abort) it before the next call on the same stream");
    if (bytes <= c.ws_cap) return 0;
    HIPCHK(hipDeviceSynchronize()); h->n_devsync++;
    if (c.ws) { HIPCHK(hipFree(c.ws)); h->n_alloc++; }
    c.ws = nullptr; c.ws_cap = 0;
    // a first call of a shape grows the workspace with 12.5 % of slack; sta_reserve takes the exact maximum of its plans
    int64_t want = h->reserve_only ? bytes : bytes + (bytes >> 3) + (1 << 20);
    HIPCHK(hipMalloc((void**)&c.ws, (size_t)want)); h->n_alloc++;
    c.ws_cap = want;
    return 0;
}

// zero the padding of V^T buffers (keys >= ntok must contribute 0): one fill when the planes sit back to back in the workspace
static int zero_planes(const Planes* const* pl, int n, int64_t elems, bool split, hipStream_t st) {
    const int64_t each = elems * 2 * (split ? 2 : 1);
    bool contiguous = true;
    for (int i = 0; i < n; ++i) {
        if (split && (char*)pl[i]->lo != (char*)pl[i]->hi + elems * 2) contiguous = false;
        if (i > 0 && (char*)pl[i]->hi != (char*)pl[i - 1]->hi + each) contiguous = false;
    }
    if (contiguous) { HIPCHK(hipMemsetAsync(pl[0]->hi, 0, (size_t)(each * n), st)); return 0; }
    for (int i = 0; i < n; ++i) {
        HIPCHK(hipMemsetAsync(pl[i]->hi, 0, (size_t)elems * 2, st));
        if (split) HIPCHK(hipMemsetAsync(pl[i]->lo, 0, (size_t)elems * 2, st));
    }
    return 0;
}
static Bump cur_bump(sta_handle* h) { return Bump{h->cur->ws, h->cur->ws_cap}; }

// side lane of the current context (created on first use: one stream, 16 MiB of split-K scratch, four events)
static int ensure_side(sta_handle* h) {
    StreamCtx& c = *h->cur;
    // every resource on its own: a failure half way leaves what exists in place for the next attempt (nothing is leaked twice)
    if (!c.side_skbuf) { HIPCHK(hipMalloc((void**)&c.side_skbuf, (size_t)SKBUF_ELEMS * 4)); h->n_alloc++; }
    for (auto& e : c.side_ev) if (!e) { HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->n_alloc++; }
    if (!c.side) { HIPCHK(hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking)); h->n_alloc++; }
    return 0;
}
static inline float* lane_skbuf(sta_handle* h) { return h->lane ? h->cur->side_skbuf : h->cur->skbuf; }

// ------------------------------------------------------------------------------------------ schema
static int make_lin(sta_handle* h, Lin& L, int N, int K, bool bias = true, bool mx = false) {
    L.N = N; L.K = K;
    CHK(dalloc(h, (void**)&L.w.hi, (int64_t)N * K * 4 + 256));     // blocked [K/32][N][hi32|lo32]
    L.w.lo = L.w.hi + 32; L.w.rp = N;
    if (mx) { CHK(dalloc(h, (void**)&L.wmx.hi, (int64_t)N * K * 4 + 256)); L.wmx.lo = L.wmx.hi + 32; L.wmx.rp = N; }   // [K/32][N][hi32 | hi8 x32 | lo8 x32]
    if (bias) CHK(dalloc(h, (void**)&L.bias, (int64_t)N * 4));
    return 0;
}
static int make_ln(sta_handle* h, LNp& n, int C) {
    CHK(dalloc(h, (void**)&n.g, C * 4)); CHK(dalloc(h, (void**)&n.b, C * 4)); return 0;
}
static void slot_w(sta_handle* h, const std::string& name, std::vector<int64_t> shape, SlotKind k, Lin& L, int row_off = 0) {
    Slot s; s.shape = std::move(shape); s.kind = k;
    s.dst_hi = L.w.hi; s.dst_lo = L.w.lo; s.dst_mx = L.wmx.hi; s.N = L.N; s.K = L.K; s.n_off = row_off;
    h->slots[name] = s;
}
static void slot_f32(sta_handle* h, const std::string& name, std::vector<int64_t> shape, float* dst) {
    Slot s; s.shape = std::move(shape); s.kind = SK_F32; s.dst32 = dst; h->slots[name] = s;
}
static void slot_drop(sta_handle* h, const std::string& name, std::vector<int64_t> shape) {
    Slot s; s.shape = std::move(shape); s.kind = SK_DROP; h->slots[name] = s;
}
static int reg_linear(sta_handle* h, const std::string& name, Lin& L, int N, int K, bool mx = false, int cls = CLS_NONE) {
    CHK(make_lin(h, L, N, K, true, mx));
    L.cls = cls;
    slot_w(h, name + ".weight", {N, K}, SK_W_ID, L);
    slot_f32(h, name + ".bias", {N}, L.bias);
    return 0;
}
static int reg_ln(sta_handle* h, const std::string& name, LNp& n, int C) {
    CHK(make_ln(h, n, C));
    slot_f32(h, name + ".weight", {C}, n.g); slot_f32(h, name + ".bias", {C}, n.b);
    return 0;
}
// conv weight [Co,Ci,k,k] -> packed [Co][k][k][Ci]
static int reg_conv(sta_handle* h, const std::string& name, Lin& L, int Co, int Ci, int k, bool bias) {
    CHK(make_lin(h, L, Co, Ci * k * k, bias, true)); L.cls = CLS_HEAD;
    slot_w(h, name + ".weight", {Co, Ci, k, k}, k == 1 ? SK_W_ID : SK_W_CONV, L);
    if (bias) slot_f32(h, name + ".bias", {Co}, L.bias);
    return 0;
}
static int reg_convt(sta_handle* h, const std::string& name, Lin& L, int C, int k) {
    CHK(make_lin(h, L, k * k * C, C, true, true)); L.cls = CLS_HEAD;
    slot_w(h, name + ".weight", {C, C, k, k}, SK_W_CONVT, L);
    Slot s; s.shape = {C}; s.kind = SK_B_CONVT; s.dst32 = L.bias; s.reps = k * k;
    h->slots[name + ".bias"] = s;
    return 0;
}
static int reg_f32lin(sta_handle* h, const std::string& name, F32Lin& L, int N, int K, std::vector<int64_t> wshape) {
    CHK(dalloc(h, (void**)&L.w, (int64_t)N * K * 4)); CHK(dalloc(h, (void**)&L.b, N * 4));
    slot_f32(h, name + ".weight", std::move(wshape), L.w); slot_f32(h, name + ".bias", {N}, L.b);
    return 0;
}

static int build_schema(sta_handle* h) {
    const sta_config& c = h->cfg;
    const int E = c.enc_embed_dim, D = c.dec_embed_dim, P = c.patch_size, R = c.mlp_ratio;
    char nm[256];
    CHK(dalloc(h, (void**)&h->pose_tok, D * 4));
    slot_f32(h, "init_pose_token", {1, 1, D}, h->pose_tok);
    CHK(make_lin(h, h->patch, E, 3 * P * P));
    slot_w(h, "patch_embed.proj.weight", {E, 3, P, P}, SK_W_ID, h->patch);
    slot_f32(h, "patch_embed.proj.bias", {E}, h->patch.bias);
    h->enc.resize(c.enc_depth);
    for (int i = 0; i < c.enc_depth; ++i) {
        EncBlk& b = h->enc[i];
        snprintf(nm, sizeof nm, "enc_blocks.%d.", i); std::string p(nm);
        CHK(reg_ln(h, p + "norm1", b.n1, E));
        CHK(reg_linear(h, p + "attn.qkv", b.qkv, 3 * E, E));
        CHK(reg_linear(h, p + "attn.proj", b.proj, E, E));
        CHK(reg_ln(h, p + "norm2", b.n2, E));
        CHK(reg_linear(h, p + "mlp.fc1", b.fc1, R * E, E));
        CHK(reg_linear(h, p + "mlp.fc2", b.fc2, E, R * E, true, CLS_FC2));
    }
    CHK(reg_ln(h, "enc_norm", h->enc_norm, E));   // only applied by _encode_image(normalize=True) (sta_model.py:172-173); the forward / SLAM paths pass False
    CHK(reg_linear(h, "decoder_embed", h->dec_embed, D, E));
    h->dec.resize(c.dec_depth);
    for (int i = 0; i < c.dec_depth; ++i) {
        DecBlk& b = h->dec[i];
        snprintf(nm, sizeof nm, "dec_block.%d.", i); std::string p(nm);
        CHK(reg_ln(h, p + "norm1", b.n1, D));
        CHK(reg_linear(h, p + "attn.qkv", b.qkv, 3 * D, D));
        CHK(reg_linear(h, p + "attn.proj", b.proj, D, D));
        CHK(reg_linear(h, p + "cross_attn.projq", b.cq, D, D));
        // projk + projv packed as one [2D, D] GEMM
        CHK(make_lin(h, b.ckv, 2 * D, D));
        slot_w(h, p + "cross_attn.projk.weight", {D, D}, SK_W_ID, b.ckv, 0);
        slot_f32(h, p + "cross_attn.projk.bias", {D}, b.ckv.bias);
        slot_w(h, p + "cross_attn.projv.weight", {D, D}, SK_W_ID, b.ckv, D);
        slot_f32(h, p + "cross_attn.projv.bias", {D}, b.ckv.bias + D);
        CHK(reg_linear(h, p + "cross_attn.proj", b.cproj, D, D));
        CHK(reg_ln(h, p + "norm2", b.n2, D));
        CHK(reg_ln(h, p + "norm3", b.n3, D));
        CHK(reg_linear(h, p + "mlp.fc1", b.fc1, R * D, D));
        CHK(reg_linear(h, p + "mlp.fc2", b.fc2, D, R * D, true, CLS_FC2));
        CHK(reg_ln(h, p + "norm_y", b.ny, D));
    }
    CHK(reg_ln(h, "dec_norm", h->dec_norm, D));
    const std::string dp = "downstream_head_pts.dpt.";
    const int F = 256, L0 = 96, L1 = 192, L2 = 384, L3 = 768, LD[4] = {L0, L1, L2, L3};
    for (int k = 0; k < 4; ++k) {
        snprintf(nm, sizeof nm, "scratch.layer%d_rn", k + 1);
        CHK(reg_conv(h, dp + nm, h->rn[k], F, LD[k], 3, false));
        snprintf(nm, sizeof nm, "scratch.layer_rn.%d.weight", k);       // alias of the same tensor (dpt_block.py:70-75)
        slot_drop(h, dp + nm, {F, LD[k], 3, 3});
    }
    for (int r = 1; r <= 4; ++r) {
        Refine& rf = h->ref[r - 1];
        snprintf(nm, sizeof nm, "scratch.refinenet%d.", r); std::string p = dp + nm;
        CHK(reg_conv(h, p + "out_conv", rf.out, F, F, 1, true));
        if (r == 4) {   // refinenet4.resConfUnit1 is never executed (single-input fusion, dpt_block.py:196-204)
            for (const char* cn : {"conv1", "conv2"}) {
                slot_drop(h, p + "resConfUnit1." + cn + ".weight", {F, F, 3, 3});
                slot_drop(h, p + "resConfUnit1." + cn + ".bias", {F});
            }
        } else {
            CHK(reg_conv(h, p + "resConfUnit1.conv1", rf.u1.c1, F, F, 3, true));
            CHK(reg_conv(h, p + "resConfUnit1.conv2", rf.u1.c2, F, F, 3, true));
        }
        CHK(reg_conv(h, p + "resConfUnit2.conv1", rf.u2.c1, F, F, 3, true));
        CHK(reg_conv(h, p + "resConfUnit2.conv2", rf.u2.c2, F, F, 3, true));
    }
    CHK(reg_conv(h, dp + "head.0", h->head0, F / 2, F, 3, true));
    CHK(reg_conv(h, dp + "head.2", h->head2, 128, F / 2, 3, true));
    CHK(reg_f32lin(h, dp + "head.4", h->head4, 4, 128, {4, 128, 1, 1}));
    CHK(reg_conv(h, dp + "act_postprocess.0.0", h->act0_0, L0, E, 1, true));
    CHK(reg_convt(h, dp + "act_postprocess.0.1", h->act0_1, L0, 4));
    CHK(reg_conv(h, dp + "act_postprocess.1.0", h->act1_0, L1, D, 1, true));
    CHK(reg_convt(h, dp + "act_postprocess.1.1", h->act1_1, L1, 2));
    CHK(reg_conv(h, dp + "act_postprocess.2.0", h->act2_0, L2, D, 1, true));
    CHK(reg_conv(h, dp + "act_postprocess.3.0", h->act3_0, L3, D, 1, true));
    CHK(reg_conv(h, dp + "act_postprocess.3.1", h->act3_1, L3, L3, 3, true));
    const int Hd = 512;
    CHK(reg_f32lin(h, "head_pose_s.mlp.0", h->pm0, Hd, D, {Hd, D}));
    CHK(reg_f32lin(h, "head_pose_s.mlp.2", h->pm1, Hd, Hd, {Hd, Hd}));
    CHK(reg_f32lin(h, "head_pose_s.mlp.4", h->pm2, Hd, Hd, {Hd, Hd}));
    CHK(reg_f32lin(h, "head_pose_s.fc_t", h->pt, 3, Hd, {3, Hd}));
    CHK(reg_f32lin(h, "head_pose_s.fc_conf.0", h->pc, 1, Hd, {1, Hd}));
    CHK(reg_f32lin(h, "head_pose_s.fc_rot", h->pr, 9, Hd, {9, Hd}));
    return 0;
}

// ------------------------------------------------------------------------------------------ API: lifecycle
static int mask_of_precision(int prec) {
    return prec == STA_PREC_F16X3H ? CLS_HEAD : (prec == STA_PREC_F16X3M ? (CLS_HEAD | CLS_FC2) : 0);
}
static bool known_precision(int p) { return p == STA_PREC_F16 || p == STA_PREC_F16X3 || p == STA_PREC_F16X3H || p == STA_PREC_F16X3M; }

extern "C" void sta_default_config(sta_config* c) {
    c->patch_size = 16; c->enc_embed_dim = 1024; c->enc_depth = 24; c->enc_num_heads = 16;
    c->dec_embed_dim = 768; c->dec_depth = 12; c->dec_num_heads = 12; c->mlp_ratio = 4;
    c->rope_base = 100.0f; c->ln_eps = 1e-6f; c->precision = STA_PREC_F16X3H;
}

extern "C" const char* sta_last_error(void) { return g_err; }
extern "C" const char* sta_version(void) { return "sta_mi355 0.1 (gfx950)"; }

extern "C" int sta_create(const sta_config* cfg, int device, sta_handle** out) {
    REQUIRE(cfg && out, "sta_create: null argument");
    REQUIRE(cfg->patch_size == 16, "patch_size must be 16");
    REQUIRE(cfg->enc_embed_dim == 64 * cfg->enc_num_heads, "encoder head_dim must be 64");
    REQUIRE(cfg->dec_embed_dim == 64 * cfg->dec_num_heads, "decoder head_dim must be 64");
    REQUIRE(cfg->enc_embed_dim % 128 == 0 && cfg->enc_embed_dim <= 1024, "enc_embed_dim must be a multiple of 128, <= 1024");
    REQUIRE(cfg->dec_embed_dim % 128 == 0 && cfg->dec_embed_dim <= 1024, "dec_embed_dim must be a multiple of 128, <= 1024");
    REQUIRE(cfg->dec_depth > 9, "dec_depth must be > 9 (heads/dpt_head.py:102)");
    REQUIRE(known_precision(cfg->precision), "unknown precision %d", cfg->precision);
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    DEV_SCOPE(device);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "this library is built for gfx950 only, device reports %s", prop.gcnArchName);
    sta_handle* h = new sta_handle();
    h->cfg = *cfg; h->device = device; h->prec = cfg->precision; h->mx_mask = mask_of_precision(cfg->precision);
#ifdef STA_BENCH_EXPERIMENTS
    for (int i = 0; i < 8; ++i) {            // experiment builds only (never shipped): switches from the environment, STA_OPT0 .. STA_OPT7
        char nm[16]; snprintf(nm, sizeof nm, "STA_OPT%d", i);
        if (const char* v = getenv(nm)) h->opt[i] = atoi(v);
    }
#endif
    h->ctx.reserve(MAX_STREAM_CTX);
    if (build_schema(h) != 0) { sta_destroy(h); return -1; }
    h->stage_elems = (int64_t)cfg->mlp_ratio * cfg->enc_embed_dim * cfg->enc_embed_dim;
    int64_t big = (int64_t)768 * 768 * 9;
    if (big > h->stage_elems) h->stage_elems = big;
    if (hipMalloc((void**)&h->stage, (size_t)h->stage_elems * 4) != hipSuccess) { sta_destroy(h); return set_err("staging alloc failed"); }
    if (hipMalloc((void**)&h->zero_page, 256) != hipSuccess || hipMemset(h->zero_page, 0, 256) != hipSuccess) { sta_destroy(h); return set_err("zero page alloc failed"); }
    if (hipMalloc((void**)&h->range, 16) != hipSuccess || hipMemset(h->range, 0, 16) != hipSuccess) { sta_destroy(h); return set_err("range counter alloc failed"); }
    *out = h;
    return 0;
}

extern "C" int sta_destroy(sta_handle* h) {
    if (!h) return 0;
    DevScope dev_scope_(h->device);
    hipDeviceSynchronize();
    for (void* p : h->allocs) hipFree(p);
    if (h->stage) hipFree(h->stage);
    for (auto& c : h->ctx) {
        if (c.ws) hipFree(c.ws); if (c.skbuf) hipFree(c.skbuf); if (c.slab) hipFree(c.slab);
        if (c.rv_conf) hipHostFree(c.rv_conf); if (c.rv_ev) hipEventDestroy(c.rv_ev);
        if (c.side) hipStreamDestroy(c.side); if (c.side_skbuf) hipFree(c.side_skbuf);
        for (auto& e : c.side_ev) if (e) hipEventDestroy(e);
    }
    for (hipStream_t s : h->pipe_streams) if (s) hipStreamDestroy(s);
    if (h->rope_tab) hipFree(h->rope_tab);
    if (h->zero_page) hipFree(h->zero_page);
    if (h->range) hipFree(h->range);
    if (h->pre_tab) hipFree(h->pre_tab);
    if (h->clk_buf) hipFree(h->clk_buf);
    if (h->kstamp) hipFree(h->kstamp);
    if (h->ev_ok) for (auto& e : h->ev) hipEventDestroy(e);
    for (auto& e : h->kev) hipEventDestroy(e);
    delete h;
    return 0;
}

extern "C" int sta_set_precision(sta_handle* h, int precision) {
    REQUIRE(h, "null handle");
    REQUIRE(known_precision(precision), "unknown precision %d", precision);
    h->prec = precision; h->mx_mask = mask_of_precision(precision);
    return 0;
}
extern "C" int sta_set_deterministic(sta_handle* h, int on) {
    REQUIRE(h, "null handle");
    h->deterministic = on != 0;
    return 0;
}
extern "C" int sta_set_side_lanes(sta_handle* h, int mode) {
    REQUIRE(h && (mode == STA_LANES_AUTO || mode == STA_LANES_OFF || mode == STA_LANES_ON), "sta_set_side_lanes: bad argument");
    h->lanes_mode = mode;
    return 0;
}
#ifdef STA_TEST_HOOKS      // libsta_mi355_test.so only (include/sta_mi355_debug.h)
extern "C" int sta_set_gemm_variant(sta_handle* h, int variant) {
    REQUIRE(h && ((variant >= 0 && variant <= 4) || (variant >= 8 && variant <= 11)), "bad gemm variant");
    h->small_grid_mode = variant == 10 ? 1 : (variant == 11 ? 2 : 0);     // measurement tool only; per handle
    h->gemm_variant = variant >= 10 ? 0 : variant;
    return 0;
}
#endif
extern "C" int sta_range_report(sta_handle* h, unsigned long long counts[2], int reset) {
    REQUIRE(h && counts, "null argument");
    DEV_SCOPE(h->device);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(counts, h->range, 16, hipMemcpyDeviceToHost));
    if (reset) HIPCHK(hipMemset(h->range, 0, 16));
    counts[0] += (unsigned long long)h->tiny_tensors;      // a property of the loaded weights: survives a reset
    return 0;
}
extern "C" int sta_num_expected_tensors(const sta_handle* h) { return h ? (int)h->slots.size() : -1; }
extern "C" int sta_num_loaded_tensors(const sta_handle* h) { return h ? h->n_loaded : -1; }
extern "C" int64_t sta_workspace_bytes(const sta_handle* h) {
    if (!h) return -1;
    int64_t b = 0;
    for (const auto& c : h->ctx) b += c.ws_cap;
    return b;
}
extern "C" int64_t sta_weight_bytes(const sta_handle* h) { return h ? h->weight_bytes : -1; }

extern "C" int sta_load_tensor(sta_handle* h, const char* name, const void* host_ptr,
                               const int64_t* shape, int ndim, int dtype) {
    REQUIRE(h && name && host_ptr && shape, "sta_load_tensor: null argument");
    REQUIRE(dtype == STA_DTYPE_F32, "only fp32 source tensors are supported");
    DEV_SCOPE(h->device);
    auto it = h->slots.find(name);
    REQUIRE(it != h->slots.end(), "unexpected key in state_dict: %s", name);
    Slot& s = it->second;
    REQUIRE((int)s.shape.size() == ndim, "size mismatch for %s: ndim %d vs expected %d", name, ndim, (int)s.shape.size());
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        REQUIRE(shape[i] == s.shape[i], "size mismatch for %s: dim %d is %lld, expected %lld", name, i, (long long)shape[i], (long long)s.shape[i]);
        n *= shape[i];
    }
    if (s.kind == SK_W_ID || s.kind == SK_W_CONV || s.kind == SK_W_CONVT) {
        // the LOW end of the fp16 range: a weight is carried as hi = fp16(w), lo = fp16(w - hi), both with an absolute floor of 2^-25.
        // Mixed-magnitude tensors do not care (the floor is 1e-6 of a typical 0.02 weight), but a tensor whose EVERY value is below
        // 2^-12 would enter the MFMAs as subnormals with a few bits each - counted as a range event (reported, not silent) unless
        // the tensor is all zeros.  (head.4, the one layer the outlier statistics push there, is fp32 and pre-scaled: sta_finalize_weights.)
        const float* w = static_cast<const float*>(host_ptr);
        float mx = 0.f;
        for (int64_t i = 0; i < n; ++i) { const float a = fabsf(w[i]); mx = a > mx ? a : mx; }
        if (!s.loaded_tiny && mx > 0.f && mx < 0x1p-12f) { s.loaded_tiny = true; h->tiny_tensors++; }
        else if (s.loaded_tiny && !(mx > 0.f && mx < 0x1p-12f)) { s.loaded_tiny = false; h->tiny_tensors--; }
    }
    if (s.kind != SK_DROP) {
        REQUIRE(n <= h->stage_elems, "tensor %s too large for staging", name);
        if (s.kind == SK_F32) {
            HIPCHK(hipMemcpy(s.dst32, host_ptr, (size_t)n * 4, hipMemcpyHostToDevice));
        } else {
            HIPCHK(hipMemcpy(h->stage, host_ptr, (size_t)n * 4, hipMemcpyHostToDevice));
            int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
            if (s.kind == SK_B_CONVT) {
                int tot = (int)n * s.reps;
                expand_bias_kernel<<<(tot + 255) / 256, 256>>>(h->stage, s.dst32, (int)n, s.reps);
            } else {
                int mode = s.kind == SK_W_ID ? 0 : (s.kind == SK_W_CONV ? 1 : 2);
                int d0 = (int)s.shape[0], d1 = ndim > 1 ? (int)s.shape[1] : 1, d2 = ndim > 2 ? (int)s.shape[2] : 1, d3 = ndim > 3 ? (int)s.shape[3] : 1;
                repack_weight_kernel<<<blocks, 256>>>(h->stage, s.dst_hi, s.dst_lo, n, mode, d0, d1, d2, d3, s.N, s.K, s.n_off, 0, h->range);
                if (s.dst_mx) repack_weight_kernel<<<blocks, 256>>>(h->stage, s.dst_mx, s.dst_mx + 32, n, mode, d0, d1, d2, d3, s.N, s.K, s.n_off, 1, h->range);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipDeviceSynchronize());
        }
    }
    if (!s.loaded) { s.loaded = true; h->n_loaded++; }
    return 0;
}

extern "C" int sta_finalize_weights(sta_handle* h) {
    REQUIRE(h, "null handle");
    for (auto& kv : h->slots)
        REQUIRE(kv.second.loaded, "missing key in state_dict: %s", kv.first.c_str());
    DEV_SCOPE(h->device);
    HIPCHK(hipDeviceSynchronize());
    {   // head.4 (1x1 conv 128 -> 4) enters the fused DPT tail as fp16 hi / lo planes (head_epilogue_t, gemm2.h).  Its values can be far
        // below fp16's normal range (a checkpoint whose DPT feature maps are large has correspondingly small head.4 weights: the outlier
        // goldens scale them by 1/300 -> 1e-5, subnormal in fp16, 1e-3 relative error) - so they are split AFTER an exact power-of-two
        // scaling into [0.5, 1) and the result is scaled back
        std::vector<float> w4(4 * 128);
        HIPCHK(hipMemcpy(w4.data(), h->head4.w, w4.size() * 4, hipMemcpyDeviceToHost));
        for (int o = 0; o < 4; ++o) {
            float mx = 0.f;
            for (int c = 0; c < 128; ++c) mx = fmaxf(mx, fabsf(w4[o * 128 + c]));
            int e = 0;
            if (mx > 0.f && std::isfinite(mx)) (void)frexpf(mx, &e);        // mx = f * 2^e, f in [0.5, 1)
            if (e > 100) e = 100; if (e < -100) e = -100;
            h->head4_scale[o] = ldexpf(1.0f, -e);
        }
    }
    h->finalized = true;
    return 0;
}

#include "sta_launch.inc"     // tile-family cost model, launch_gemm, the per-op launch wrappers

#include "sta_forward.inc"    // encode_impl / decode_impl / pose_impl / dpt_impl

// ------------------------------------------------------------------------------------------ API: compute
static int check_ready(sta_handle* h, int B, int H, int W) {
    REQUIRE(h, "null handle");
    REQUIRE(h->finalized, "weights not finalized (call sta_finalize_weights)");
    REQUIRE(B > 0, "batch must be positive");
    REQUIRE(H % 16 == 0 && W % 16 == 0 && H > 0 && W > 0, "Input image size (%dx%d) is not a multiple of patch size (16)", H, W);
    return 0;
}

// Two-pass execution: a dry planning pass sizes the workspace exactly (same allocation sequence,
// no launches), then the real pass runs.  Steady state: the plan fits, nothing is allocated.
template <class F>
static int plan_and_run(sta_handle* h, hipStream_t st, F&& body) {
    CHK(stream_ctx(h, st));          // this stream's scratch context (the dry pass already reads h->cur)
    REQUIRE(!h->cur->rv_open, "a split-phase scheduler call (sta_regress_views_begin) is pending on this stream: its workspace is live "
                              "until sta_regress_views_finish; run other calls on another stream");
    h->dry = true;
    Bump plan{nullptr, INT64_MAX};
    int r = body(plan);
    h->dry = false;
    if (r != 0) return r;
    CHK(ensure_ws(h, plan.peak + 4096, st));
    if (h->reserve_only) return ensure_side(h);      // sta_reserve: the plan is allocated (and the side lane with it), nothing runs
    Bump ws = cur_bump(h);
    return body(ws);
}

extern "C" int sta_encode(sta_handle* h, const float* img_dev, int B, int H, int W, float* feat_dev, void* stream) {
    REQUIRE(h, "null handle");
    DEV_SCOPE(h->device);
    CHK(check_ready(h, B, H, W));
    REQUIRE(img_dev && feat_dev, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
    const int hp = H / 16, wp = W / 16;
    CHK(ensure_rope(h, hp > wp ? hp : wp));
    const void* imgs[1] = {img_dev};
    return plan_and_run(h, st, [&](Bump& ws) { return encode_impl(h, ws, imgs, false, 1, B, H, W, feat_dev, st); });
}

extern "C" int sta_encode_u8hwc(sta_handle* h, const uint8_t* img_dev, int B, int H, int W, float* feat_dev, void* stream) {
    REQUIRE(h, "null handle");
    DEV_SCOPE(h->device);
    CHK(check_ready(h, B, H, W));
    REQUIRE(img_dev && feat_dev, "null device pointer");
    REQUIRE(((uintptr_t)img_dev & 15) == 0, "u8 HWC image must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int hp = H / 16, wp = W / 16;
    CHK(ensure_rope(h, hp > wp ? hp : wp));
    const void* imgs[1] = {img_dev};
    return plan_and_run(h, st, [&](Bump& ws) { return encode_impl(h, ws, imgs, true, 1, B, H, W, feat_dev, st); });
}

extern "C" int sta_encoder_norm(sta_handle* h, const float* feat_dev, int64_t rows, float* out_dev, void* stream) {
    REQUIRE(h && h->finalized, "handle not ready");
    REQUIRE(feat_dev && out_dev && rows > 0 && rows < (int64_t)1 << 31, "bad argument");
    DEV_SCOPE(h->device);
    Planes none;
    return run_ln(h, feat_dev, (int)rows, h->cfg.enc_embed_dim, h->enc_norm, none, nullptr, nullptr, out_dev, (hipStream_t)stream);
}

extern "C" int sta_decode(sta_handle* h, const float* feat1, const float* feat2, int B, int hp, int wp,
                          float* const* out1, float* const* out2, void* stream) {
    REQUIRE(h, "null handle");
    DEV_SCOPE(h->device);
    CHK(check_ready(h, B, hp * 16, wp * 16));
    REQUIRE(feat1 && feat2, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
    CHK(ensure_rope(h, hp > wp ? hp : wp));
    const int N = hp * wp, D = h->cfg.dec_embed_dim;
    const int64_t xbytes = (int64_t)2 * B * (N + 1) * D * 4;
    return plan_and_run(h, st, [&](Bump& ws) {
        float* x = (float*)ws.take(xbytes);
        return decode_impl(h, ws, feat1, feat2, B, hp, wp, x, out1, out2, true, st);
    });
}

// _decode_stereo with positions that are NOT the patch grid (a crop of a larger grid, a permuted token order): the reference rotates
// q / k by whatever positions it is handed (sta_blocks.py:134-137 self-attention, :196-199 cross-attention: q by the own side's
// positions, k by the other side's).  The positions become a [2B][N][2] int32 table in the workspace and every QKV epilogue of the
// call rotates by the IDENTITY (a table of (1, 0) rows), after which rope_planes_kernel rotates the Q / K buffers in place, row by row from
// that table (decode_impl) - the throughput form's epilogues stay untouched.
extern "C" int sta_decode_pos(sta_handle* h, const float* feat1, const float* feat2, const int64_t* pos1, const int64_t* pos2,
                              int B, int N, int pos_max, float* const* out1, float* const* out2, void* stream) {
    REQUIRE(h, "null handle");
    DEV_SCOPE(h->device);
    CHK(check_ready(h, B, 16, 16));
    REQUIRE(feat1 && feat2 && pos1 && pos2, "null device pointer");
    REQUIRE(N > 0 && pos_max >= 0 && pos_max < (1 << 20), "bad argument (N %d, pos_max %d)", N, pos_max);
    hipStream_t st = (hipStream_t)stream;
    CHK(ensure_rope(h, pos_max + 1));
    const int D = h->cfg.dec_embed_dim;
    const int64_t xbytes = (int64_t)2 * B * (N + 1) * D * 4;
    struct Scope { sta_handle* h; ~Scope() { h->rope_foreign = false; h->rope_pos = nullptr; h->rope_ident = nullptr; } } scope{h};
    h->rope_foreign = true;
    return plan_and_run(h, st, [&](Bump& ws) {
        float* x = (float*)ws.take(xbytes);
        int* rp = (int*)ws.take((int64_t)2 * B * N * 2 * 4);
        const int64_t n_ident = (int64_t)(N + 2) * 16;            // the grid form of the call is 1 x N: table rows 0 .. N + 1
        float2* ident = (float2*)ws.take(n_ident * 8);
        if (!h->dry) {
            const int64_t n = (int64_t)B * N * 2;
            hipLaunchKernelGGL(rope_pos_table_kernel, dim3((unsigned)((2 * n + n_ident + 255) / 256)), dim3(256), 0, st, pos1, pos2, n, pos_max, rp, ident, n_ident);
            HIPCHK(hipGetLastError());
            h->rope_pos = rp; h->rope_ident = (const float*)ident;
        }
        return decode_impl(h, ws, feat1, feat2, B, 1, N, x, out1, out2, true, st);
    });
}

extern "C" int sta_head_pose(sta_handle* h, const float* tok, int B, int64_t tok_stride, float* pose, float* conf, void* stream) {
    REQUIRE(h && h->finalized, "handle not ready");
    REQUIRE(tok && pose && conf && B > 0, "bad argument");
    DEV_SCOPE(h->device);
    REQUIRE(tok_stride % 4 == 0, "tok_stride must be a multiple of 4 floats");
    hipStream_t st = (hipStream_t)stream;
    return plan_and_run(h, st, [&](Bump& ws) { return pose_impl(h, ws, tok, B, tok_stride, pose, conf, st); });
}

extern "C" int sta_head_pts(sta_handle* h, const float* enc_feat, int64_t enc_bstride,
                            const float* hook1, int64_t hook1_bstride, const float* hook2, int64_t hook2_bstride,
                            const float* hook3, int64_t hook3_bstride, int B, int H, int W,
                            float* pts, float* conf, void* stream) {
    REQUIRE(h, "null handle");
    DEV_SCOPE(h->device);
    CHK(check_ready(h, B, H, W));
    REQUIRE(enc_feat && hook1 && hook2 && hook3 && pts && conf, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
    return plan_and_run(h, st, [&](Bump& ws) {
        return dpt_impl(h, ws, enc_feat, enc_bstride, hook1, hook1_bstride, hook2, hook2_bstride, hook3, hook3_bstride,
                        B, H, W, pts, conf, B, nullptr, nullptr, st);
    });
}

static int forward_pair_any(sta_handle* h, const void* img_a, const void* img_b, bool u8hwc, int B, int H, int W,
                            float* const pts[2], float* const conf[2], float* const pose[2], float* const pose_conf[2],
                            void* stream) {
    REQUIRE(h, "null handle");
    DEV_SCOPE(h->device);
    CHK(check_ready(h, B, H, W));
    REQUIRE(img_a && img_b && pts && conf && pose && pose_conf, "null argument");
    hipStream_t st = (hipStream_t)stream;
    const sta_config& c = h->cfg;
    const int hp = H / 16, wp = W / 16, N = hp * wp, Np = N + 1, E = c.enc_embed_dim, D = c.dec_embed_dim, S = 2 * B;
    CHK(ensure_rope(h, hp > wp ? hp : wp));
    if (h->timing && !h->ev_ok) { for (auto& e : h->ev) HIPCHK(hipEventCreate(&e)); h->ev_ok = true; }
    const int dd = c.dec_depth;
    const int hidx[3] = {dd * 2 / 4, dd * 3 / 4, dd};   // hooks [d/2+1, 3d/4+1, d+1] - 1 (dpt_head.py:112)
    // the whole batch of pairs, every launch on stream `st`
    auto run_slice = [&](Bump& ws, const void* ia, const void* ib, int Bs, float* const p[2], float* const cf[2],
                         float* const po[2], float* const pc[2], hipStream_t st, bool rec) -> int {
        const int Ss = 2 * Bs;
        const int64_t feat_b = (int64_t)Ss * N * E * 4, x_b = (int64_t)Ss * Np * D * 4;
        float* feat = (float*)ws.take(feat_b);
        float* x = (float*)ws.take(x_b);
        float* hk[3] = {(float*)ws.take(x_b), (float*)ws.take(x_b), (float*)ws.take(x_b)};
        const int64_t mark = ws.off;
        if (rec) HIPCHK(hipEventRecord(h->ev[0], st));
        const void* imgs[2] = {ia, ib};
        CHK(encode_impl(h, ws, imgs, u8hwc, 2, Bs, H, W, feat, st));
        if (rec) HIPCHK(hipEventRecord(h->ev[1], st));
        ws.rewind(mark);
        // hooks in the decoder's own row order (decode_impl): [Ss*N patch rows | Ss pose rows]
        std::vector<float*> w1(dd + 1, nullptr);
        for (int k = 0; k < 3; ++k) w1[hidx[k]] = hk[k];
        CHK(decode_impl(h, ws, feat, feat + (size_t)Bs * N * E, Bs, hp, wp, x, w1.data(), nullptr, false, st));
        if (rec) HIPCHK(hipEventRecord(h->ev[2], st));
        // pose heads read the pose token of the dec_norm'ed last layer (sta_model.py:273,277)
        ws.rewind(mark);
        const float* ptok = hk[2] + (size_t)Ss * N * D;
        CHK(pose_impl(h, ws, ptok, 2 * Bs, (int64_t)D, po[0], pc[0], st, po[1], pc[1], Bs));       // both sides: the 2 Bs pose rows are consecutive
        if (rec) HIPCHK(hipEventRecord(h->ev[3], st));
        ws.rewind(mark);
        CHK(dpt_impl(h, ws, feat, (int64_t)N * E, hk[0], (int64_t)N * D, hk[1], (int64_t)N * D, hk[2], (int64_t)N * D,
                     Ss, H, W, p[0], cf[0], Bs, p[1], cf[1], st));
        if (rec) HIPCHK(hipEventRecord(h->ev[4], st));
        return 0;
    };
    return plan_and_run(h, st, [&](Bump& ws) -> int {
        return run_slice(ws, img_a, img_b, B, pts, conf, pose, pose_conf, st, h->timing && !h->dry);
    });
}

extern "C" int sta_forward_pair(sta_handle* h, const float* img_a, const float* img_b, int B, int H, int W,
                                float* const pts[2], float* const conf[2], float* const pose[2], float* const pose_conf[2],
                                void* stream) {
    return forward_pair_any(h, img_a, img_b, false, B, H, W, pts, conf, pose, pose_conf, stream);
}
extern "C" int sta_forward_pair_u8hwc(sta_handle* h, const uint8_t* img_a, const uint8_t* img_b, int B, int H, int W,
                                      float* const pts[2], float* const conf[2], float* const pose[2], float* const pose_conf[2],
                                      void* stream) {
    REQUIRE((((uintptr_t)img_a | (uintptr_t)img_b) & 15) == 0, "u8 HWC images must be 16-byte aligned");
    return forward_pair_any(h, img_a, img_b, true, B, H, W, pts, conf, pose, pose_conf, stream);
}

#include "sta_rows.inc"     // input step, keyframe scheduler, post-STA reductions, output step

extern "C" int sta_kernel_timing(sta_handle* h, int enable) {
    REQUIRE(h, "null handle");
    DEV_SCOPE(h->device);
    if (enable && !h->clk_buf) { HIPCHK(hipMalloc((void**)&h->clk_buf, 16)); }
    if (h->clk_buf) HIPCHK(hipMemset(h->clk_buf, 0, 16));
    if (enable == 4) {            // mode 2 + in-kernel stamps of every GEMM / convolution launch
        if (!h->kstamp) HIPCHK(hipMalloc((void**)&h->kstamp, (size_t)KSTAMP_LAUNCHES * KSTAMP_WG * 32));
        HIPCHK(hipMemset(h->kstamp, 0, (size_t)KSTAMP_LAUNCHES * KSTAMP_WG * 32));
    }
    h->kstamp_on = enable == 4;
    h->ktime = enable != 0; h->ktime_all = enable == 2 || enable == 4; h->kn = 0;
    if (enable != 3) { h->kfilter[0] = h->kfilter[1] = h->kfilter[2] = h->kfilter[3] = -1; h->kevery = 1; }
    h->kseen = 0;
    return 0;
}
extern "C" int sta_kernel_timing_filter(sta_handle* h, int epilogue, int a_mode, int family, int mx, int every) {
    REQUIRE(h && every >= 1, "sta_kernel_timing_filter: null handle or every < 1");
    h->kfilter[0] = epilogue; h->kfilter[1] = a_mode; h->kfilter[2] = family; h->kfilter[3] = mx;
    h->kevery = every; h->kseen = 0;
    return 0;
}
extern "C" int sta_kernel_clock_read(sta_handle* h, float* ghz_out) {
    REQUIRE(h && ghz_out, "null argument");
    REQUIRE(h->clk_buf, "kernel timing was never enabled");
    DEV_SCOPE(h->device);
    HIPCHK(hipDeviceSynchronize());
    unsigned long long hc[2] = {0, 0};
    HIPCHK(hipMemcpy(hc, h->clk_buf, 16, hipMemcpyDeviceToHost));
    *ghz_out = hc[1] ? (float)((double)hc[0] / ((double)hc[1] / 100e6) * 1e-9) : 0.f;
    return 0;
}
extern "C" int sta_kernel_timing_read(sta_handle* h, int variant, int* launches, double* total_ms, double* total_flops, double* total_bytes) {
    REQUIRE(h && launches && total_ms && total_flops && total_bytes, "null argument");
    DEV_SCOPE(h->device);
    double ms = 0, fl = 0, by = 0;
    int cnt = 0;
    for (int i = 0; i < h->kn; ++i) {
        if (variant > 0 && h->kvar[i] != variant) continue;
        ++cnt;
        HIPCHK(hipEventSynchronize(h->kev[2 * i + 1]));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, h->kev[2 * i], h->kev[2 * i + 1]));
        ms += t; fl += h->kflops[i]; by += h->kbytes[i];
    }
    *launches = cnt; *total_ms = ms; *total_flops = fl; *total_bytes = by;
    return 0;
}

#ifdef STA_TEST_HOOKS
extern "C" int sta_kernel_timing_dump(sta_handle* h, int cap, double* flops, float* ms, int* variant, int* n_out) {
    REQUIRE(h && flops && ms && variant && n_out, "null argument");
    DEV_SCOPE(h->device);
    int n = 0;
    for (int i = 0; i < h->kn && n < cap; ++i, ++n) {
        HIPCHK(hipEventSynchronize(h->kev[2 * i + 1]));
        HIPCHK(hipEventElapsedTime(&ms[n], h->kev[2 * i], h->kev[2 * i + 1]));
        flops[n] = h->kflops[i]; variant[n] = h->kvar[i];
    }
    *n_out = n;
    return 0;
}
#endif

extern "C" int sta_kernel_timing_dump_shapes(sta_handle* h, int cap, int* shape6, float* ms, int* variant, int* n_out) {
    REQUIRE(h && shape6 && ms && variant && n_out, "null argument");
    DEV_SCOPE(h->device);
    int n = 0;
    for (int i = 0; i < h->kn && n < cap; ++i, ++n) {
        HIPCHK(hipEventSynchronize(h->kev[2 * i + 1]));
        HIPCHK(hipEventElapsedTime(&ms[n], h->kev[2 * i], h->kev[2 * i + 1]));
        for (int q = 0; q < 6; ++q) shape6[6 * n + q] = h->kshape[6 * i + q];
        variant[n] = h->kvar[i];
    }
    *n_out = n;
    return 0;
}

#ifdef STA_TEST_HOOKS
// In-kernel stamps of the launches recorded since sta_kernel_timing(h, 4): per launch out6 = {workgroups, span us, median entry ->
// first K tile, median main loop, median epilogue, spread of the exits} (100-MHz stamps of every workgroup, GemmParams::stamps).
extern "C" int sta_kernel_stamps_dump(sta_handle* h, int cap, double* out6, int* n_out) {
    REQUIRE(h && out6 && n_out && h->kstamp, "stamps were never enabled (sta_kernel_timing(h, 4))");
    DEV_SCOPE(h->device);
    HIPCHK(hipDeviceSynchronize());
    const int n = h->kn < cap ? (h->kn < KSTAMP_LAUNCHES ? h->kn : KSTAMP_LAUNCHES) : cap;
    std::vector<unsigned long long> hs((size_t)KSTAMP_WG * 4);
    for (int i = 0; i < n; ++i) {
        HIPCHK(hipMemcpy(hs.data(), h->kstamp + (size_t)i * KSTAMP_WG * 4, (size_t)KSTAMP_WG * 32, hipMemcpyDeviceToHost));
        std::vector<double> pro, loop, epi;
        unsigned long long t0min = ~0ull, t3min = ~0ull, t3max = 0;
        for (int b = 0; b < KSTAMP_WG; ++b) {
            const unsigned long long* q = &hs[(size_t)b * 4];
            if (!q[3] || !q[0]) continue;
            pro.push_back((double)(q[1] - q[0]) * 0.01); loop.push_back((double)(q[2] - q[1]) * 0.01); epi.push_back((double)(q[3] - q[2]) * 0.01);
            if (q[0] < t0min) t0min = q[0]; if (q[3] < t3min) t3min = q[3]; if (q[3] > t3max) t3max = q[3];
        }
        double* o = out6 + (size_t)i * 6;
        if (pro.empty()) { for (int k = 0; k < 6; ++k) o[k] = 0; continue; }
        auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        o[0] = (double)pro.size(); o[1] = (double)(t3max - t0min) * 0.01; o[2] = med(pro); o[3] = med(loop); o[4] = med(epi); o[5] = (double)(t3max - t3min) * 0.01;
    }
    *n_out = n;
    return 0;
}
#endif

extern "C" int sta_enable_stage_timing(sta_handle* h, int on) { REQUIRE(h, "null handle"); h->timing = on != 0; return 0; }
extern "C" int sta_get_stage_ms(sta_handle* h, float ms[4]) {
    REQUIRE(h && h->ev_ok, "stage timing not recorded");
    HIPCHK(hipEventSynchronize(h->ev[4]));
    for (int i = 0; i < 4; ++i) HIPCHK(hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]));
    return 0;
}

extern "C" int sta_rope2d_inplace_dtype(void* tokens_dev, int dtype, int64_t stride_b, int64_t stride_n, const int64_t* pos_dev,
                                        int B, int N, int Hh, int D, float base, float fwd, void* stream) {
    REQUIRE(tokens_dev && pos_dev, "null device pointer");
    REQUIRE(D % 4 == 0, "token dim must be multiple of 4");
    REQUIRE(dtype == STA_DTYPE_F32 || dtype == STA_DTYPE_F16 || dtype == STA_DTYPE_F64, "rope_2d: unsupported token dtype %d", dtype);
    int64_t total = (int64_t)B * N * Hh * (D / 2);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == STA_DTYPE_F32) hipLaunchKernelGGL(rope2d_inplace_kernel<float>, grid, block, 0, st, (float*)tokens_dev, stride_b, stride_n, pos_dev, B, N, Hh, D, base, fwd);
    else if (dtype == STA_DTYPE_F16) hipLaunchKernelGGL(rope2d_inplace_kernel<f16>, grid, block, 0, st, (f16*)tokens_dev, stride_b, stride_n, pos_dev, B, N, Hh, D, base, fwd);
    else hipLaunchKernelGGL(rope2d_inplace_kernel<double>, grid, block, 0, st, (double*)tokens_dev, stride_b, stride_n, pos_dev, B, N, Hh, D, base, fwd);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int sta_rope2d_inplace(float* tokens_dev, int64_t stride_b, int64_t stride_n, const int64_t* pos_dev,
                                  int B, int N, int Hh, int D, float base, float fwd, void* stream) {
    return sta_rope2d_inplace_dtype(tokens_dev, STA_DTYPE_F32, stride_b, stride_n, pos_dev, B, N, Hh, D, base, fwd, stream);
}

#include "sta_bench.inc"    // FLOP model (product) + GEMM / attention micro-benchmark entry points (STA_TEST_HOOKS)

#ifdef STA_TEST_HOOKS
#include "sta_debug.inc"   // kernel-level test entry points: libsta_mi355_test.so only
#endif
