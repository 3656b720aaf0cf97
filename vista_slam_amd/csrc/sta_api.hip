// C-ABI implementation of the MI355X STA frontend (see include/sta_mi355.h).
// Host orchestration only: every FLOP runs in the hand-written gfx950 kernels of gemm.h,
// attention.h and elementwise.h.  No torch, no BLAS, no CPU fallback.
#include "../../include/sta_mi355.h"
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
enum { CLS_NONE = 0, CLS_HEAD = 16 /* DPT head convolutions */ };
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
        HIPCHK(hipDeviceSynchronize());
        lru->st = st; lru->last_use = ++h->use_clock;
        h->cur = lru;
        return 0;
    }
    StreamCtx c; c.st = st; c.last_use = ++h->use_clock;
    HIPCHK(hipMalloc((void**)&c.skbuf, (size_t)SKBUF_ELEMS * 4));
    if (hipMalloc((void**)&c.slab, (size_t)SKBUF_ELEMS * 4) != hipSuccess) { hipFree(c.skbuf); return set_err("split-K slab alloc failed"); }
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
    HIPCHK(hipDeviceSynchronize());
    if (c.ws) HIPCHK(hipFree(c.ws));
    c.ws = nullptr; c.ws_cap = 0;
    int64_t want = bytes + (bytes >> 3) + (1 << 20);
    HIPCHK(hipMalloc((void**)&c.ws, (size_t)want));
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
    if (!c.side_skbuf) HIPCHK(hipMalloc((void**)&c.side_skbuf, (size_t)SKBUF_ELEMS * 4));
    for (auto& e : c.side_ev) if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!c.side) HIPCHK(hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking));
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
        CHK(reg_linear(h, p + "mlp.fc2", b.fc2, E, R * E));
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
        CHK(reg_linear(h, p + "mlp.fc2", b.fc2, D, R * D));
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
    return prec == STA_PREC_F16X3H ? CLS_HEAD : 0;
}

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
    REQUIRE(cfg->precision == STA_PREC_F16 || cfg->precision == STA_PREC_F16X3 || cfg->precision == STA_PREC_F16X3H, "unknown precision %d", cfg->precision);
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
    REQUIRE(precision == STA_PREC_F16 || precision == STA_PREC_F16X3 || precision == STA_PREC_F16X3H, "unknown precision %d", precision);
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
    h->finalized = true;
    return 0;
}

// ------------------------------------------------------------------------------------------ launch helpers
template <bool SPLIT, int AMODE, int EPI, int BM, int BN, int WMS, int WNS, int NSTG = 2, bool MX = false>
static int launch_gemm2(const GemmParams& p, hipStream_t st, int dev = 0) {
    static unsigned attr_done = 0;        // one bit per device: the attribute is set on the current device's copy of the function
    constexpr int smem = gemm2_smem_bytes<SPLIT, BM, BN>(NSTG);
    if (!(attr_done >> (dev & 31) & 1u)) {
        HIPCHK(hipFuncSetAttribute((const void*)gemm2_kernel<SPLIT, AMODE, EPI, BM, BN, WMS, WNS, 0, NSTG, MX>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done |= 1u << (dev & 31);
    }
    int tm = (p.M - p.m_tail + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int ntail = p.m_tail > 0 ? (p.N + 31) / 32 : 0;     // skinny tail blocks (gemm2_tail), first in the grid
    hipLaunchKernelGGL((gemm2_kernel<SPLIT, AMODE, EPI, BM, BN, WMS, WNS, 0, NSTG, MX>), dim3((unsigned)(tm * tn * ks + ntail)), dim3(WMS * WNS * 64), smem, st, p);
    return 0;
}

template <bool SPLIT, int EPI, int BM, int BN, bool MX, int WMS = 4, int WNS = 4>
static int launch_conv3h(const GemmParams& p, hipStream_t st, int dev) {
    static unsigned attr_done = 0;        // one bit per device
    constexpr int smem = conv3h_smem_bytes<SPLIT, BM, BN>();
    static_assert(smem <= 160 * 1024, "conv3h: LDS budget");
    auto kern = conv3h_kernel<SPLIT, EPI, BM, BN, WMS, WNS, MX>;
    if (!(attr_done >> (dev & 31) & 1u)) {
        HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done |= 1u << (dev & 31);
    }
    const int n_img = p.M / (p.Ho * p.Wo);
    const int tiles = n_img * ((p.Ho + BM / 32 - 1) / (BM / 32)) * ((p.Wo + 31) / 32) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(WMS * WNS * 64), smem, st, p);
    return 0;
}

static inline bool auto_family(const sta_handle* h) { return h->gemm_variant == 0 || h->gemm_variant == 9; }

// The ONE small-grid predicate (SLAM scale: 224x224, batch 1..8 -> M = 196..3200 rows; the DPT levels with <= 12288 pixels
// at any scale): below it the 128x64 split-K family runs (launch_gemm), in-place residual GEMMs hand their K slices to
// resid_ln_kernel (gemm_resid_ln), and the specialised throughput epilogues / the paired launch are not used (gemm_f32,
// gemm_qkv_pair).  Threshold measured (profiles/r03_tile_table.txt): with fewer than 192 tiles of 192x128 the 128x64 family is
// 15-40 % faster (162 tiles: 36 vs 45 us; 136: 113 vs 144 us; 128: 113 vs 133 us), from 216 tiles on it is slower.
static inline bool small_grid_m(int mode, int64_t M, int N) {
    if (mode == 1) return false;
    const int64_t t = ((M + 191) / 192) * (int64_t)((N + 127) / 128);
    if (mode == 2) return M <= 2560 || t < 512;
    return M <= 640 || t < 192;
}
// K slices of a small-grid GEMM with tiles_r output tiles: about one workgroup per CU and never more than 256 of them while that
// still splits (256 / tiles_r >= 2: the grid stays "lone" and runs the 8-wave form, launch_gemm); between 128 and 256 tiles a
// GEMM whose K slices would need a finisher kernel of their own (QKV, plane epilogues) is not split at all - the finisher is a
// dispatch (5 - 7 us) for a K loop that two slices shorten by less -, one whose slices are summed by the LayerNorm kernel that
// follows anyway (in-place residual GEMMs) takes two.  Measured against ceil(256 / tiles_r) everywhere: encode -2 %, 5-edge
// scheduler -1.7 %, one pair @224 +3.5 %, one pair @512x384 +1 %.  h->opt[1] == 1: the old rule (A/B).
static inline int sg_slices(const sta_handle* h, int tiles_r, bool own_finisher = false) {
    if (h->opt[1] != 1) { const int f = 256 / tiles_r; if (f >= 2) return f; if (own_finisher) return 1; }
    return (256 + tiles_r - 1) / tiles_r;
}
static inline bool small_grid(const sta_handle* h, int64_t M, int N) { return small_grid_m(h->small_grid_mode, M, N); }

// Tile-family choice = a quantisation-aware cost model calibrated on profiles/r03_tile_table.txt (tools/tile_table.py: every
// GEMM / convolution shape of the forward at B in {1,2,4,8} x {224x224, 384x512}, in-model HIP-event durations under every forced
// family; tests/test_tile_table.py replays the table through this function).  Every family saturates at the same 300-400
// algorithmic TFLOP/s (the chip is power-bound on this instruction mix), so a launch costs
//     rounds x tile area / rate,   rounds = ceil(tiles / 256 CUs),
// `rate` = the family's measured advantage on that class of GEMM at a full grid, x a penalty when the last round is < 90 %
// full (one-workgroup-per-CU families lose more there than 192x128, whose second resident workgroup evens the CUs out):
//   5: 192x128, 8 waves, two workgroups per CU (one block's HBM-bound epilogue hides under the other's main loop): rate 1;
//   3: 192x256, 12 waves, one workgroup per CU, 30 % less L2->LDS traffic per FLOP: +2 % mlp.fc1 (GELU), +4 % the long-K
//      in-place-residual GEMMs (mlp.fc2), +6 % the Cout = 256 convolutions with K >= 1728;
//   2: 256x256, 16 waves: +5 % mlp.fc1, +10 % those convolutions, -10 % the in-place-residual epilogue;
//   8: halo-tiled 3x3 convolution (conv3h.h; 8 x 32-pixel tiles of one image): +12 % at Cout = 256 (x1.47 from three
//      well-filled rounds on), equal at Cout = 128
//      (head.0: ties go to the halo form, which moves 1.3x instead of 6.5x the algorithmic bytes), -2.5 % with the fused head epilogue below 2M pixels, equal from there on (the benchmark's 3.1M: 1.25x instead of 6.6x the bytes);
//   6: small-grid family (predicate above);  1: 128x128 register-staged kernel: N not a multiple of 128;
//   7 (paired 192x128 launch) is chosen by gemm_qkv_pair.
struct FamilyQuery { int amode, epi; int64_t M; int N, K; int split, cstride, Ho, Wo; int sg_mode; };
static int pick_family(const FamilyQuery& q) {
    const bool sg = small_grid_m(q.sg_mode, q.M, q.N) && q.N % 64 == 0;
    if (q.N % 128 != 0) return sg ? 6 : 1;
    if (sg) return 6;
    auto cost = [](int64_t tiles, int area, double rate, double pen) {
        const int64_t rounds = (tiles + 255) / 256;
        const double fill = (double)tiles / (double)(rounds * 256);
        return (double)rounds * area / (rate * (fill < 0.9 ? pen : 1.0));
    };
    auto tiles = [&](int bm, int bn) { return ((q.M + bm - 1) / bm) * (int64_t)((q.N + bn - 1) / bn); };
    int best = 5;
    double cbest = cost(tiles(192, 128), 192 * 128, 1.0, 0.96);
    auto consider = [&](int fam, double c) { if (c <= cbest) { cbest = c; best = fam; } };      // ties: the later (larger) family
    if (q.split && q.N % 256 == 0 && q.epi != EPI_QKV) {
        double r3 = 0, r2 = 0;
        if (q.amode == A_DENSE && q.epi == EPI_GELU) { r3 = 1.02; r2 = 1.05; }
        else if (q.amode == A_DENSE && q.epi == EPI_F32R && q.K >= 2048) { r3 = 1.04; r2 = 0.9; }
        else if (q.amode == A_CONV3 && q.N == 256 && q.K >= 1728) { r3 = 1.06; r2 = 1.10; }
        else if (q.amode == A_DENSE && q.epi == EPI_F16) { r3 = 1.0; r2 = 1.0; }      // 1x1 convolutions of the head: quantisation only
        if (r3 > 0) { consider(3, cost(tiles(192, 256), 192 * 256, r3, 0.93)); consider(2, cost(tiles(256, 256), 256 * 256, r2, 0.93)); }
    }
    if (q.amode == A_CONV3 && (q.epi == EPI_F16 || q.epi == EPI_HEAD) && q.cstride == 1 && (q.N == 128 || q.N == 256) &&
        q.Wo >= 32 && q.M >= 16384 && q.Ho > 0) {
        const int64_t imgs = q.M / ((int64_t)q.Ho * q.Wo);
        const int64_t t8 = imgs * ((q.Ho + 7) / 8) * ((q.Wo + 31) / 32);
        if (q.N == 256) {
            // +12 % per tile against 192x128 in general (a single round: equal to 192x256 within the +-3 % the boxes differ by,
            // which keeps its choice there), 1.47x from three well-filled rounds on (196608 pixels: 539 vs 576 / 640 us)
            const int64_t rounds = (t8 + 255) / 256;
            const double fill = (double)t8 / (double)(rounds * 256);
            consider(8, (double)rounds * 256 * 256 / (fill >= 0.7 && rounds >= 3 ? 1.47 : 1.12));
        } else {
            consider(8, cost(t8, 256 * q.N, q.epi == EPI_HEAD ? (q.M >= (1 << 21) ? 1.0 : 0.975) : 1.0, 0.96));
        }
    }
    return best;
}
#ifdef STA_TEST_HOOKS
extern "C" int sta_debug_pick_family(int amode, int epi, long long M, int N, int K, int split, int cstride, int Ho, int Wo) {
    FamilyQuery q{amode, epi, M, N, K, split, cstride, Ho, Wo, 0};
    return pick_family(q);
}
#endif

// slab_ks_out: K slices a slab GEMM wrote (0: it did not take the slab path) - the caller's finisher sums exactly those
template <int AMODE, int EPI>
static int launch_gemm(sta_handle* h, const GemmParams& p_in, hipStream_t st, int* slab_ks_out = nullptr) {
    GemmParams p = p_in;
    p.range = h->range;
    int slab_ks = 0;
    if (slab_ks_out) *slab_ks_out = 0;
    p.zero_page = h->zero_page;
    REQUIRE(p.K % GEMM_BK == 0, "GEMM K=%d must be a multiple of %d", p.K, GEMM_BK);
    REQUIRE(p.M > 0 && p.N > 0, "empty GEMM");
    if (AMODE == A_CONV3) REQUIRE(p.Cin % GEMM_BK == 0, "conv Cin=%d must be a multiple of %d", p.Cin, GEMM_BK);
    // the LDS-DMA loaders address a K tile with 32-bit byte offsets (128 B per row of a row block)
    REQUIRE((AMODE != A_DENSE || (int64_t)p.M * 128 < ((int64_t)1 << 32)) && (int64_t)p.N * 128 < ((int64_t)1 << 32),
            "GEMM with M=%d, N=%d exceeds the 32-bit row-offset range of the DMA loaders", p.M, p.N);
    if (AMODE == A_CONV3) REQUIRE((int64_t)p.a_rp * 128 < ((int64_t)1 << 32), "convolution input of %lld pixels exceeds the 32-bit offset range of the tap loader", (long long)p.a_rp);
    if (h->dry) return 0;
    const bool split = h->prec != STA_PREC_F16;
    // Row tail hint (decode_impl: the 2B pose-token rows after the 2B x N patch rows).  Tile rules below look at the
    // patch rows; the tail is kept only where a gemm2 family tiles them exactly (checked after the family is chosen).
    p.m_tail = 0;
    if (AMODE == A_DENSE && h->tail_hint > 0 && h->tail_hint <= 32 && p.M > h->tail_hint && !p.mx && p.N % 128 == 0 && h->gemm_variant != 1) {
        if (!small_grid(h, p.M - h->tail_hint, p.N)) p.m_tail = h->tail_hint;   // throughput scale only (the small-grid family splits K instead)
    }
    const int M_all = p.M;
    p.M -= p.m_tail;
    // Tile family (pick_family above: cost model calibrated on the measured table); a forced family never displaces the
    // small-grid one, whose split-K plumbing the callers rely on
    const FamilyQuery fq{AMODE, EPI, p.M, p.N, p.K, split ? 1 : 0, p.cstride, p.Ho, p.Wo, h->small_grid_mode};
    int variant = pick_family(fq);
    if (h->gemm_variant == 9 && variant == 8) { FamilyQuery f2 = fq; f2.Wo = 0; variant = pick_family(f2); }     // A/B: no halo kernel
    // Small grids (SLAM scale: 224x224, batch 1..8 -> M = 196..3200 rows; the coarse DPT levels at any scale): 128x64 tiles,
    // 3-stage DMA ring, and split-K so that ~256 workgroups stream the weights once instead of 16-64 workgroups looping over
    // all of K.  The K slices go to fp32 SLABS that the next kernel sums (resid_ln_kernel / qkv_finish_kernel /
    // splitk_finish_kernel: fixed order, bit-reproducible) - the product path.  Only an in-place residual GEMM whose caller
    // passed no slab (forced tile families, N > 1024) still adds its slices with fp32 atomics, and not in deterministic mode.
    if (variant == 6) {
        const int tiles_r = ((p.M + 127) / 128) * (p.N / 64);
        const int tiles = h->deterministic ? (1 << 30) : tiles_r;    // deterministic: no ATOMIC split-K (the slab forms below have a fixed order)
        if (AMODE == A_DENSE && EPI == EPI_F32 && p.resid == p.C32 && p.slab) {
            // the caller finishes the GEMM in the LayerNorm kernel that follows (gemm_resid_ln): slices store partial tiles
            // to slabs instead of atomically adding to the residual stream (the device-scope fp32 atomics of 256 workgroups
            // cost more than the 4-32 K tiles of a slice); fixed summation order -> also taken in deterministic mode
            // (swept at M = 196 / 394 / 1970: a target of 128 / 192 / 256 / 384 / 512 workgroups -> encode 2.50 / 2.40 / 2.38 /
            // 2.49 / 2.55 ms: one workgroup per CU; more slices cost more slab traffic than their shorter K loops save)
            int ks = tiles_r < 256 ? sg_slices(h, tiles_r) : 1;
            const int max_ks = p.K / 128;                 // keep >= 4 K tiles per slice
            if (ks > max_ks) ks = max_ks;
            while (ks > 1 && (int64_t)ks * p.M * p.N > SKBUF_ELEMS) --ks;
            if (ks > 1) { p.ksplit = ks; slab_ks = ks; } else p.slab = nullptr;
        } else if (AMODE == A_DENSE && EPI == EPI_F32 && p.resid == p.C32 && tiles < 256) {
            int ks = (256 + tiles - 1) / tiles;
            const int max_ks = p.K / 128;                 // keep >= 4 K tiles per slice
            if (ks > max_ks) ks = max_ks;
            if (ks > 1) p.ksplit = ks;
        }
        // QKV-epilogue GEMMs (attn.qkv, cross_attn.projq / projk|projv) on small grids: K slices to slabs, qkv_finish_kernel
        // applies bias + RoPE and writes Q / K / V^T (un-split attn.qkv at M = 196: 96 workgroups x 32 K tiles = 25 us)
        if (AMODE == A_DENSE && EPI == EPI_QKV && tiles_r <= 192 && p.K >= 512) {
            int ks = sg_slices(h, tiles_r, true);
            const int max_ks = p.K / 256;                 // keep >= 8 K tiles per slice
            if (ks > max_ks) ks = max_ks;
            while (ks > 1 && (int64_t)ks * p.M * p.N > SKBUF_ELEMS) --ks;
            if (ks > 1) { p.ksplit = ks; p.skbuf = lane_skbuf(h); }
        }
        // plane-epilogue GEMMs / convs on tiny grids (DPT levels at SLAM scale: 16-64 workgroups looping over K = 2304 ..
        // 6912): split K into fp32 partial sums, a finishing kernel applies bias / activation / residuals.  Worth two
        // extra tiny launches only when the K loop is long and the grid leaves most of the chip idle (swept with atomics: <= 96 /
        // 160 / 256 tiles -> DPT 1.00 / 0.85 / 0.83 ms per view; with slabs, 5-edge scheduler: <= 192 / 256 tiles -> 6.04 / 5.89 ms).  An in-kernel fix-up (last slice finishes the tile behind a
        // device-scope fence + ticket) was 1.7x SLOWER than this: the fence writes back / invalidates the XCD's L2.
        if (EPI == EPI_F16 && tiles_r <= 256 && p.K >= 1024 && p.N % 4 == 0 && (int64_t)p.M * p.N <= SKBUF_ELEMS) {
            int ks = sg_slices(h, tiles_r, true);
            const int max_ks = p.K / 256;                 // keep >= 8 K tiles per slice
            if (ks > max_ks) ks = max_ks;
            while (ks > 1 && (int64_t)ks * p.M * p.N > SKBUF_ELEMS) --ks;      // one slab per K slice
            if (ks > 1) { p.ksplit = ks; p.skbuf = lane_skbuf(h); }
        }
    }
    // forced families (tests / tools): 1 = 128x128 register-staged, 2 = 256x256, 3 = 192x256 (both wherever N % 256 == 0 and
    // the epilogue is not the RoPE one), 4 = 192x128 everywhere
    if ((h->gemm_variant == 2 || h->gemm_variant == 3) && variant != 6 && p.N % 128 == 0)
        variant = (p.N % 256 == 0 && EPI != EPI_QKV) ? h->gemm_variant : 5;
    if (h->gemm_variant == 4 && variant != 6 && p.N % 128 == 0) variant = 5;
    if (h->gemm_variant == 1) variant = 1;
    // 8 forced (tests): the halo-tiled 3x3 convolution wherever it is legal (stride 1, Cout 128 / 256)
    if (AMODE == A_CONV3 && (EPI == EPI_F16 || EPI == EPI_HEAD) && p.cstride == 1 && (p.N == 128 || p.N == 256) && h->gemm_variant == 8) variant = 8;
    if (variant != 6) { p.ksplit = 1; slab_ks = 0; }
    if (slab_ks == 0) p.slab = nullptr;
    if (slab_ks_out) *slab_ks_out = slab_ks;
    p.M = M_all;
    {
        const int bm_v = variant == 2 ? 256 : ((variant == 3 || variant == 5) ? 192 : 0);
        constexpr bool tail_epi = EPI == EPI_F32 || EPI == EPI_F32R || EPI == EPI_GELU || EPI == EPI_QKV;   // gemm2_body: HAS_TAIL
        if (p.m_tail && (bm_v == 0 || (p.M - p.m_tail) % bm_v != 0 || !tail_epi)) p.m_tail = 0;
    }
    constexpr bool MX_EPI = EPI == EPI_F16 || EPI == EPI_CONVT || EPI == EPI_HEAD;    // the DPT head's epilogues: only they have f16mx kernels
    REQUIRE(MX_EPI || !p.mx, "internal: f16mx arithmetic outside the DPT head");
    if (p.mx && variant == 1) variant = 5;     // no f16mx form of the register-staged kernel (use_mx() already requires N % 64 == 0)
    // per-launch HIP-event timing (bench / tools): every launch (mode 2), or only the launches of ONE kernel symbol
    // (mode 3, sta_kernel_timing_filter: the event pairs break back-to-back dispatch, ~3.5 us each, so the timed region of
    // bench.py carries them on the dominant kernel only)
    bool timed = h->ktime && (h->ktime_all || (h->kfilter[0] == EPI && h->kfilter[1] == AMODE && h->kfilter[2] == variant && h->kfilter[3] == p.mx));
    if (timed && !h->ktime_all && (h->kseen++ % h->kevery) != 0) timed = false;      // mode 3: a 1-in-kevery sample of the symbol's launches
    if (timed) {
        if ((int)h->kev.size() < 2 * (h->kn + 1)) {
            hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
            h->kev.push_back(a); h->kev.push_back(b);
        }
        if ((int)h->kflops.size() <= h->kn) h->kflops.resize(h->kn + 1);
        h->kflops[h->kn] = 2.0 * p.M * p.N * p.K;
        if ((int)h->kshape.size() < 6 * (h->kn + 1)) h->kshape.resize(6 * (h->kn + 1));
        { int* q = &h->kshape[6 * h->kn]; q[0] = p.M; q[1] = p.N; q[2] = p.K; q[3] = EPI; q[4] = AMODE; q[5] = p.mx; }
        if ((int)h->kbytes.size() <= h->kn) h->kbytes.resize(h->kn + 1);
        // algorithmic bytes: A and W planes (2 B x planes) read once, C written once (+ residual read)
        h->kbytes[h->kn] = (split ? 4.0 : 2.0) * ((double)p.M * p.K + (double)p.N * p.K) + 4.0 * p.M * p.N * (p.resid ? 2.0 : 1.0);
        HIPCHK(hipEventRecord(h->kev[2 * h->kn], st));
        h->kn++;
    }
    if (timed && variant == 5) p.clk_dbg = h->clk_buf;   // effective-clock probe (bench only)
    if (timed && h->kstamp_on && h->kn <= KSTAMP_LAUNCHES) p.stamps = h->kstamp + (size_t)(h->kn - 1) * KSTAMP_WG * 4;
    if constexpr (AMODE == A_CONV3 && (EPI == EPI_F16 || EPI == EPI_HEAD)) {
        if (variant == 8) {
            REQUIRE((int64_t)p.Ho * p.Wo > 0 && p.M % (p.Ho * p.Wo) == 0, "internal: conv3h needs whole images");
            if (p.N == 128) {
                if (p.mx) CHK((launch_conv3h<true, EPI, 256, 128, true>(p, st, h->device)));
                else if (split) CHK((launch_conv3h<true, EPI, 256, 128, false>(p, st, h->device)));
                else STA_F16ONLY(CHK((launch_conv3h<false, EPI, 256, 128, false>(p, st, h->device))));
            } else if constexpr (EPI == EPI_F16) {
                if (p.mx) CHK((launch_conv3h<true, EPI, 256, 256, true>(p, st, h->device)));
                else if (split) CHK((launch_conv3h<true, EPI, 256, 256, false>(p, st, h->device)));
                else STA_F16ONLY(CHK((launch_conv3h<false, EPI, 256, 256, false>(p, st, h->device))));
            }
        }
    }
    if (variant == 2 || variant == 3) REQUIRE(EPI != EPI_QKV, "internal: the RoPE epilogue exists on the 192x128 / 128x64 / 128x128 tiles only");
    if (variant == 8) {
    } else
    if constexpr (EPI == EPI_HEAD) {      // exists for the 192x128 family only (conv3_head checks the shape)
        REQUIRE(variant == 5 && p.N == 128, "internal: fused head epilogue on a tile family without it");
        if (p.mx) { if constexpr (MX_EPI) CHK((launch_gemm2<true, AMODE, EPI, 192, 128, 2, 4, 2, true>(p, st, h->device))); }
        else if (split) CHK((launch_gemm2<true, AMODE, EPI, 192, 128, 2, 4>(p, st, h->device)));
        else STA_F16ONLY(CHK((launch_gemm2<false, AMODE, EPI, 192, 128, 2, 4>(p, st, h->device))));
    } else
    if (variant == 2) {
      if constexpr (EPI != EPI_QKV) {
        if (p.mx) { if constexpr (MX_EPI) CHK((launch_gemm2<true, AMODE, EPI, 256, 256, 4, 4, 2, true>(p, st, h->device))); }
        else if (split) CHK((launch_gemm2<true, AMODE, EPI, 256, 256, 4, 4>(p, st, h->device)));
        else STA_F16ONLY(CHK((launch_gemm2<false, AMODE, EPI, 256, 256, 4, 4>(p, st, h->device))));
      }
    } else if (variant == 3) {
      if constexpr (EPI != EPI_QKV) {
        if (p.mx) { if constexpr (MX_EPI) CHK((launch_gemm2<true, AMODE, EPI, 192, 256, 3, 4, 2, true>(p, st, h->device))); }
        else if (split) CHK((launch_gemm2<true, AMODE, EPI, 192, 256, 3, 4>(p, st, h->device)));
        else STA_F16ONLY(CHK((launch_gemm2<false, AMODE, EPI, 192, 256, 3, 4>(p, st, h->device))));
      }
    } else if (variant == 5) {
        if (p.mx) { if constexpr (MX_EPI) CHK((launch_gemm2<true, AMODE, EPI, 192, 128, 2, 4, 2, true>(p, st, h->device))); }
        else if (split) CHK((launch_gemm2<true, AMODE, EPI, 192, 128, 2, 4>(p, st, h->device)));
        else STA_F16ONLY(CHK((launch_gemm2<false, AMODE, EPI, 192, 128, 2, 4>(p, st, h->device))));
    } else if (variant == 6) {
        // A grid of <= 256 workgroups leaves every workgroup alone on its CU: with 4 waves (one per SIMD) the barrier, the DMA
        // issue, the fragment reads and the MFMAs of a K tile simply add up (ablations, tools/ring_ablate.py: 0.12 + 0.14 + 0.09 +
        // 0.19 = 0.54 us per K tile) - the same tile on 8 waves (two per SIMD, wave tile 32x32) lets one wave's MFMAs run under
        // the other's issue and waits: encode -3.8 %, one pair @224 +4.1 %.  Larger grids (two workgroups per CU already) keep
        // the 4-wave form, whose 64x32 wave tile reads 25 % fewer fragments (8 waves there: -1.2 ... -1.4 %).
        const int sg_grid = ((p.M + 127) / 128) * (p.N / 64) * (p.ksplit > 1 ? p.ksplit : 1);
        const bool lone = sg_grid <= (h->opt[2] > 1 ? h->opt[2] : 256) && h->opt[2] != 1;
        if (p.mx) {
            if constexpr (MX_EPI) {
                if (lone) CHK((launch_gemm2<true, AMODE, EPI, 128, 64, 4, 2, 3, true>(p, st, h->device)));
                else CHK((launch_gemm2<true, AMODE, EPI, 128, 64, 2, 2, 3, true>(p, st, h->device)));
            }
        } else if (split) {
            if (lone) CHK((launch_gemm2<true, AMODE, EPI, 128, 64, 4, 2, 3>(p, st, h->device)));
            else CHK((launch_gemm2<true, AMODE, EPI, 128, 64, 2, 2, 3>(p, st, h->device)));
        } else STA_F16ONLY(CHK((launch_gemm2<false, AMODE, EPI, 128, 64, 2, 2, 3>(p, st, h->device))));
        if (EPI == EPI_QKV && p.ksplit > 1) {
            const int64_t nthr = (int64_t)p.M * (p.nq + p.nk) + (int64_t)((p.M + 3) / 4) * p.nv;
            const int blocks = (int)((nthr + 255) / 256);
            if (split) hipLaunchKernelGGL(qkv_finish_kernel<true>, dim3(blocks), dim3(256), 0, st, p);
            else hipLaunchKernelGGL(qkv_finish_kernel<false>, dim3(blocks), dim3(256), 0, st, p);
            HIPCHK(hipGetLastError());
        }
        if (EPI == EPI_F16 && p.ksplit > 1) {
            const int64_t n4 = (int64_t)p.M * (p.N / 4);
            const int blocks = (int)((n4 + 255) / 256);
            if (split) hipLaunchKernelGGL(splitk_finish_kernel<true>, dim3(blocks), dim3(256), 0, st, p.skbuf, p.ksplit, p.bias, p.M, p.N, p.act, p.R1_hi, p.R2_hi, p.C_hi, p.c_rp, p.r_mx, p.c_mx, p.range);
            else hipLaunchKernelGGL(splitk_finish_kernel<false>, dim3(blocks), dim3(256), 0, st, p.skbuf, p.ksplit, p.bias, p.M, p.N, p.act, p.R1_hi, p.R2_hi, p.C_hi, p.c_rp, 0, 0, p.range);
            HIPCHK(hipGetLastError());
        }
    } else {
        int tm = (p.M + GEMM_BM - 1) / GEMM_BM, tn = (p.N + GEMM_BN - 1) / GEMM_BN;
        dim3 grid((unsigned)(tm * tn));
        if (split) {
            static unsigned attr_done = 0;
            if (!(attr_done >> (h->device & 31) & 1u)) {
                HIPCHK(hipFuncSetAttribute((const void*)gemm_kernel<true, AMODE, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_smem_bytes<true>()));
                attr_done |= 1u << (h->device & 31);
            }
            hipLaunchKernelGGL((gemm_kernel<true, AMODE, EPI>), grid, dim3(256), gemm_smem_bytes<true>(), st, p);
        } else {
            STA_F16ONLY(hipLaunchKernelGGL((gemm_kernel<false, AMODE, EPI>), grid, dim3(256), gemm_smem_bytes<false>(), st, p));
        }
    }
    HIPCHK(hipGetLastError());
    if (timed) { HIPCHK(hipEventRecord(h->kev[2 * (h->kn - 1) + 1], st)); if ((int)h->kvar.size() < h->kn) h->kvar.resize(h->kn); h->kvar[h->kn - 1] = variant; }
    return 0;
}

// mx: A is in the f16mx row format and the f16mx copy of the weight is used (transformer linears in precision f16mx)
static GemmParams gp_dense(const Planes& A, int lda, const Lin& W, int M, bool mx = false) {
    GemmParams p; memset(&p, 0, sizeof p);
    p.A_hi = A.hi; p.A_lo = A.lo; p.lda = lda; p.a_rp = A.rp;
    p.B_hi = mx ? W.wmx.hi : W.w.hi; p.B_lo = mx ? W.wmx.lo : W.w.lo; p.bias = W.bias;
    p.mx = mx ? 1 : 0;
    p.M = M; p.N = W.N; p.K = W.K;
    return p;
}

// out fp32 = A*W^T + bias (+resid), optional row remap
// an f16mx kernel exists for every tile family with N % 64 == 0 (the one exception on the path: act_postprocess[0], N = 96)
static bool use_mx(const sta_handle* h, const Lin& W) { return (h->mx_mask & W.cls) != 0 && W.wmx.hi != nullptr && W.N % 64 == 0; }

static int gemm_f32(sta_handle* h, const Planes& A, const Lin& W, int M, float* out, int ldc,
                    const float* resid, hipStream_t st) {
    GemmParams p = gp_dense(A, W.K, W, M, use_mx(h, W));
    p.C32 = out; p.ldc = ldc; p.resid = resid; p.ldr = ldc;
    // throughput-scale in-place residual GEMMs (attn.proj, mlp.fc2, cross_attn.proj): specialised epilogue
    if (resid == out && !small_grid(h, M, W.N))
        return launch_gemm<A_DENSE, EPI_F32R>(h, p, st);
    return launch_gemm<A_DENSE, EPI_F32>(h, p, st);
}
// c_mx: the output planes feed an f16mx GEMM (mlp.fc1 -> GELU -> mlp.fc2)
static int gemm_f16(sta_handle* h, const Planes& A, const Lin& W, int M, const Planes& out, int act, hipStream_t st, bool c_mx = false) {
    REQUIRE(h->dry || !c_mx || act != ACT_GELU, "internal: the GELU epilogue (mlp.fc1) has no f16mx output form");
    GemmParams p = gp_dense(A, W.K, W, M, use_mx(h, W));
    p.C_hi = out.hi; p.C_lo = out.lo; p.ldc16 = W.N; p.act = act; p.c_rp = out.rp; p.c_mx = c_mx ? 1 : 0;
    REQUIRE(h->dry || (A.rp >= M && out.rp >= M), "internal: plane rows mismatch in gemm_f16");
    REQUIRE(h->dry || !A.mx || p.mx, "internal: f16mx input rows for a GEMM without an f16mx kernel");
    if (act == ACT_GELU && M > 640) return launch_gemm<A_DENSE, EPI_GELU>(h, p, st);   // mlp.fc1 at throughput scale: compile-time activation, no residual
                                                                                      // planes (small M keeps the generic epilogue and its split-K path)
    return launch_gemm<A_DENSE, EPI_F16>(h, p, st);
}
static int run_ln(sta_handle* h, const float* x, int M, int C, const LNp& a, const Planes& oa, const LNp* b, const Planes* ob,
                  float* o32, hipStream_t st, const float* slab, int nslab);
// x += A W^T + b (attn.proj, mlp.fc2, cross_attn.proj) followed by the LayerNorm(s) of x the next GEMM(s) read (la == nullptr:
// none).  Throughput scale: the in-place GEMM, then the LayerNorm kernel.  Small-M regime (SLAM scale): the K slices store
// partial tiles to slabs and the LayerNorm kernel adds them into x before normalising - the same two dispatches without
// the atomics epilogue (13-15 us -> 8 us per GEMM at M = 196), and bit-reproducible.
static int gemm_resid_ln(sta_handle* h, const Planes& A, const Lin& W, int M, float* x, int ld, const LNp* la, const Planes* oa,
                         const LNp* lb, const Planes* ob, hipStream_t st) {
    static const LNp no_ln = {nullptr, nullptr};
    static const Planes no_planes;
    const bool small = small_grid(h, M, W.N) && W.N % 64 == 0 && W.N <= 1024 && auto_family(h) && ld == W.N;
    if (small) {
        GemmParams p = gp_dense(A, W.K, W, M, use_mx(h, W));
        p.C32 = x; p.ldc = ld; p.resid = x; p.ldr = ld;
        p.slab = h->cur->slab;
        int ks = 0;
        CHK((launch_gemm<A_DENSE, EPI_F32>(h, p, st, &ks)));
        if (ks > 1) return run_ln(h, x, M, W.N, la ? *la : no_ln, oa ? *oa : no_planes, lb, ob, nullptr, st, p.slab, ks);
    } else {
        CHK(gemm_f32(h, A, W, M, x, ld, x, st));
    }
    if (la) return run_ln(h, x, M, W.N, *la, *oa, lb, ob, nullptr, st, nullptr, 0);
    return 0;
}
struct QKVOut { Planes q, k, vt; int npad; };
static int gp_qkv(sta_handle* h, GemmParams& p, const Planes& A, const Lin& W, int M, int nq, int nk, int nv,
                  const QKVOut& o, int ntok, int heads, int wp, int has_pose, int pose_base = 0) {
    p = gp_dense(A, W.K, W, M, use_mx(h, W));
    p.Q_hi = o.q.hi; p.Q_lo = o.q.lo; p.K_hi = o.k.hi; p.K_lo = o.k.lo; p.Vt_hi = o.vt.hi; p.Vt_lo = o.vt.lo;
    p.nq = nq; p.nk = nk; p.nv = nv; p.ntok = ntok; p.npad = o.npad; p.heads = heads; p.wp = wp; p.has_pose_tok = has_pose;
    p.pose_base = pose_base;
    REQUIRE(!(pose_base > 0 && has_pose), "internal: pose_base and has_pose_tok are two layouts of the same thing");
    p.rope_tab = h->rope_tab;
    p.ntok_magic = ntok > 1 ? (unsigned)((1ull << 32) / (unsigned)ntok + 1) : 0u;
    p.wp_magic = wp > 1 ? (unsigned)((1ull << 32) / (unsigned)wp + 1) : 0u;
    REQUIRE(nq + nk + nv == W.N, "qkv segment mismatch");
    return 0;
}
static int gemm_qkv(sta_handle* h, const Planes& A, const Lin& W, int M, int nq, int nk, int nv,
                    const QKVOut& o, int ntok, int heads, int wp, int has_pose, hipStream_t st, int pose_base = 0) {
    GemmParams p;
    CHK(gp_qkv(h, p, A, W, M, nq, nk, nv, o, ntok, heads, wp, has_pose, pose_base));
    return launch_gemm<A_DENSE, EPI_QKV>(h, p, st);
}
// Two QKV-epilogue GEMMs that do not depend on each other as ONE launch (gemm2_pair_kernel) when both run on the 192x128
// family at throughput scale; otherwise two launches.  Decoder: attn.qkv on norm1(x) + cross_attn.projk|projv on norm_y.
static bool qkv_pair_one_launch(const sta_handle* h, const GemmParams& pa, const GemmParams& pb) {
    auto big = [h](const GemmParams& p) { return p.N % 128 == 0 && !small_grid(h, p.M, p.N); };
    return h->prec != STA_PREC_F16 && !pa.mx && !pb.mx && big(pa) && big(pb) && pa.K == pb.K && pa.M == pb.M && auto_family(h);
}
// Side lanes of the current context (dpt_impl, decode_impl): inside one call the library forks an internal second stream for the
// launches that are off the call's critical chain.  sta_set_side_lanes(h, mode) is the application's switch (include/sta_mi355.h):
//   STA_LANES_OFF / STA_LANES_ON: what they say (results are bit-identical either way);
//   STA_LANES_AUTO (default): on, unless the application itself is overlapping calls on several streams - another scratch context
//     of this handle was used within its last 8 context switches - because the chip is then filled ACROSS calls and the extra
//     internal streams only compete for the runtime's few hardware queues (bench.py slam_replay, three caller streams: 229
//     keyframes/s without side lanes, 189 with them); and off when GPU_MAX_HW_QUEUES is set in the environment (measured with 8:
//     every fork / join between streams on different hardware queues cost ~0.4 ms; the lanes are tuned for the runtime default).
// The whole-model timing modes (stage timing, per-launch timing of every GEMM, stamps) always run one lane.
static bool lanes_on(const sta_handle* h) {
    if (h->dry || h->timing || h->ktime_all || h->kstamp_on) return false;     // (the one-kernel timing mode of bench.py stays on the product path: its event pairs sit on each launch's own stream)
    if (h->lanes_mode == STA_LANES_OFF || h->opt[6] == 1) return false;
    if (h->lanes_mode == STA_LANES_ON || h->opt[6] == 2) return true;
    static const bool queues_overridden = getenv("GPU_MAX_HW_QUEUES") != nullptr;
    if (queues_overridden) return false;
    for (const auto& c : h->ctx) if (&c != h->cur && c.last_use + 8 > h->use_clock) return false;
    return true;
}
struct Lane { sta_handle* h; int v; Lane(sta_handle* h_, int v_) : h(h_), v(h_->lane) { h->lane = v_; } ~Lane() { h->lane = v; } };
// Leaves a function that forked the side lane: the caller's stream waits for whatever the side lane still has in flight - also
// on the error paths (an early return must not leave side-lane kernels writing a workspace the next call reuses).
struct LaneJoin {
    sta_handle* h; hipStream_t st; bool armed;
    ~LaneJoin() {
        if (!armed || !h->cur || !h->cur->side) return;
        if (hipEventRecord(h->cur->side_ev[3], h->cur->side) == hipSuccess) (void)hipStreamWaitEvent(st, h->cur->side_ev[3], 0);
    }
};
static int gemm_qkv_pair(sta_handle* h, const GemmParams& pa_in, const GemmParams& pb_in, hipStream_t st) {
    GemmParams pa = pa_in, pb = pb_in;
    pa.range = pb.range = h->range;
    if (h->dry) return 0;
    if (!qkv_pair_one_launch(h, pa, pb)) {
        CHK((launch_gemm<A_DENSE, EPI_QKV>(h, pa, st)));
        return launch_gemm<A_DENSE, EPI_QKV>(h, pb, st);
    }
    pa.zero_page = pb.zero_page = h->zero_page;
    pa.ksplit = pb.ksplit = 1;
    // pose-token rows as skinny tail blocks (GemmParams::m_tail), same rule as launch_gemm
    pa.m_tail = pb.m_tail = (h->tail_hint > 0 && h->tail_hint <= 32 && pa.M > h->tail_hint && (pa.M - h->tail_hint) % 192 == 0 &&
                             !small_grid(h, pa.M - h->tail_hint, pa.N) && !small_grid(h, pb.M - h->tail_hint, pb.N)) ? h->tail_hint : 0;
    const int ta = ((pa.M - pa.m_tail + 191) / 192) * (pa.N / 128) + (pa.m_tail ? pa.N / 32 : 0);
    const int tb = ((pb.M - pb.m_tail + 191) / 192) * (pb.N / 128) + (pb.m_tail ? pb.N / 32 : 0);
    bool timed = h->ktime && (h->ktime_all || (h->kfilter[0] == EPI_QKV && h->kfilter[1] == A_DENSE && h->kfilter[2] == 7 && h->kfilter[3] == 0));
    if (timed && !h->ktime_all && (h->kseen++ % h->kevery) != 0) timed = false;
    if (timed) {      // one record: M x (Na + Nb) x K, tile family id 7 = gemm2_pair_kernel
        if ((int)h->kev.size() < 2 * (h->kn + 1)) {
            hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
            h->kev.push_back(a); h->kev.push_back(b);
        }
        if ((int)h->kflops.size() <= h->kn) h->kflops.resize(h->kn + 1);
        h->kflops[h->kn] = 2.0 * pa.M * (pa.N + pb.N) * pa.K;
        if ((int)h->kshape.size() < 6 * (h->kn + 1)) h->kshape.resize(6 * (h->kn + 1));
        { int* q = &h->kshape[6 * h->kn]; q[0] = pa.M; q[1] = pa.N + pb.N; q[2] = pa.K; q[3] = EPI_QKV; q[4] = A_DENSE; q[5] = 0; }
        if ((int)h->kbytes.size() <= h->kn) h->kbytes.resize(h->kn + 1);
        h->kbytes[h->kn] = 4.0 * (2.0 * pa.M * pa.K + (double)(pa.N + pb.N) * pa.K) + 4.0 * pa.M * (pa.N + pb.N);
        HIPCHK(hipEventRecord(h->kev[2 * h->kn], st));
        h->kn++;
        if (h->kstamp_on && h->kn <= KSTAMP_LAUNCHES && ta + tb <= KSTAMP_WG) {
            pa.stamps = h->kstamp + (size_t)(h->kn - 1) * KSTAMP_WG * 4; pb.stamps = pa.stamps + (size_t)ta * 4;
        }
    }
    static unsigned attr_done = 0;      // one bit per device
    constexpr int smem = gemm2_smem_bytes<true, 192, 128>(2);
    auto kern = gemm2_pair_kernel<true, A_DENSE, EPI_QKV, 192, 128, 2, 4>;
    if (!(attr_done >> (h->device & 31) & 1u)) { HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem)); attr_done |= 1u << (h->device & 31); }
    hipLaunchKernelGGL(kern, dim3((unsigned)(ta + tb)), dim3(512), smem, st, pa, pb, ta);
    HIPCHK(hipGetLastError());
    if (timed) { HIPCHK(hipEventRecord(h->kev[2 * (h->kn - 1) + 1], st)); if ((int)h->kvar.size() < h->kn) h->kvar.resize(h->kn); h->kvar[h->kn - 1] = 7; }
    return 0;
}
static int gemm_convt(sta_handle* h, const Planes& A, const Lin& W, int nimg, int hh, int ww, int k, int cout,
                      const Planes& out, hipStream_t st) {
    GemmParams p = gp_dense(A, W.K, W, nimg * hh * ww, use_mx(h, W));
    REQUIRE(h->dry || A.mx == (p.mx != 0), "internal: ConvT input format mismatch");
    p.C_hi = out.hi; p.C_lo = out.lo; p.ct_k = k; p.ct_cout = cout; p.ct_h = hh; p.ct_w = ww; p.c_rp = out.rp; p.c_mx = out.mx ? 1 : 0;
    return launch_gemm<A_DENSE, EPI_CONVT>(h, p, st);
}
// 3x3 conv, pad 1, NHWC planes
static int conv3(sta_handle* h, const Planes& in, int nimg, int Hi, int Wi, int Cin, const Lin& W, int stride,
                 bool relu_in, int act, const Planes& out, const Planes* r1, const Planes* r2, hipStream_t st) {
    GemmParams p; memset(&p, 0, sizeof p);
    p.A_hi = in.hi; p.A_lo = in.lo; p.a_rp = in.rp;
    p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.cstride = stride; p.relu_in = relu_in ? 1 : 0;
    p.Ho = (Hi + 2 - 3) / stride + 1; p.Wo = (Wi + 2 - 3) / stride + 1;
    const bool mx = use_mx(h, W);
    REQUIRE(h->dry || in.mx == mx, "internal: conv input format mismatch");
    p.mx = mx ? 1 : 0;
    p.B_hi = mx ? W.wmx.hi : W.w.hi; p.B_lo = mx ? W.wmx.lo : W.w.lo; p.bias = W.bias;
    p.M = nimg * p.Ho * p.Wo; p.N = W.N; p.K = W.K;
    REQUIRE(W.K == 9 * Cin, "conv weight K mismatch");
    p.C_hi = out.hi; p.C_lo = out.lo; p.ldc16 = W.N; p.act = act; p.c_rp = out.rp; p.c_mx = out.mx ? 1 : 0;
    REQUIRE(h->dry || ((int64_t)nimg * Hi * Wi == in.rp && out.rp == p.M), "internal: conv plane rows mismatch");
    if (r1) { p.R1_hi = r1->hi; p.R1_lo = r1->lo; p.r_mx = r1->mx ? 1 : 0; REQUIRE(h->dry || r1->rp == out.rp, "internal: residual rows mismatch"); }
    if (r2) { p.R2_hi = r2->hi; p.R2_lo = r2->lo; REQUIRE(h->dry || (r2->rp == out.rp && (!r1 || r1->mx == r2->mx)), "internal: residual rows / format mismatch"); p.r_mx = r2->mx ? 1 : 0; }
    return launch_gemm<A_CONV3, EPI_F16>(h, p, st);
}

// head.2 (3x3 conv 128 -> 128) + ReLU + head.4 (1x1 conv 128 -> 4) + point-map / confidence activations as ONE kernel
// (EPI_HEAD): true when it was launched; false = the caller runs conv3 + head_final_kernel (small grids, forced tile families).
static bool conv3_head_ok(sta_handle* h, const Lin& W, int64_t M) {
    return W.N == 128 && ((auto_family(h) && !small_grid(h, M, W.N)) || h->gemm_variant == 8);
}
static int conv3_head(sta_handle* h, const Planes& in, int nimg, int Hi, int Wi, int Cin, const Lin& W, const F32Lin& W4,
                      float* ptsA, float* confA, int nA, float* ptsB, float* confB, hipStream_t st) {
    GemmParams p; memset(&p, 0, sizeof p);
    p.A_hi = in.hi; p.A_lo = in.lo; p.a_rp = in.rp;
    p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.cstride = 1; p.relu_in = 0; p.Ho = Hi; p.Wo = Wi;
    const bool mx = use_mx(h, W);
    REQUIRE(h->dry || in.mx == mx, "internal: conv input format mismatch");
    p.mx = mx ? 1 : 0;
    p.B_hi = mx ? W.wmx.hi : W.w.hi; p.B_lo = mx ? W.wmx.lo : W.w.lo; p.bias = W.bias;
    p.M = nimg * Hi * Wi; p.N = W.N; p.K = W.K;
    REQUIRE(W.K == 9 * Cin && W.N == 128, "conv weight shape mismatch (fused head)");
    REQUIRE(h->dry || (int64_t)nimg * Hi * Wi == in.rp, "internal: conv plane rows mismatch");
    p.hw4 = W4.w; p.hb4 = W4.b; p.hptsA = ptsA; p.hconfA = confA; p.hptsB = ptsB; p.hconfB = confB;
    p.hsplit = (int64_t)(nA < nimg ? nA : nimg) * Hi * Wi;
    return launch_gemm<A_CONV3, EPI_HEAD>(h, p, st);
}

static int run_ln(sta_handle* h, const float* x, int M, int C, const LNp& a, const Planes& oa,
                  const LNp* b, const Planes* ob, float* o32, hipStream_t st,
                  const float* slab = nullptr, int nslab = 0) {
    if (h->dry) return 0;
    LnParams p; memset(&p, 0, sizeof p);
    p.range = h->range;
    p.slab = slab; p.nslab = nslab; p.xw = const_cast<float*>(x);     // slab split-K: x += sum of the slices first (x is the residual stream)
    p.x = x; p.ldx = C; p.M = M; p.C = C; p.eps = h->cfg.ln_eps;
    p.g1 = a.g; p.b1 = a.b; p.o1_hi = oa.hi; p.o1_lo = oa.lo;
    if (b) { p.g2 = b->g; p.b2 = b->b; p.o2_hi = ob->hi; p.o2_lo = ob->lo; }
    p.o32 = o32; p.ldo32 = C;
    if (slab) {          // residual-GEMM finish + LayerNorm, one block per row
        REQUIRE(C <= 1024 && C % 4 == 0, "internal: slab LayerNorm needs C <= 1024");
        if (h->prec != STA_PREC_F16) hipLaunchKernelGGL(resid_ln_kernel<true>, dim3(M), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(resid_ln_kernel<false>, dim3(M), dim3(256), 0, st, p);
        HIPCHK(hipGetLastError());
        return 0;
    }
    dim3 grid((M + 3) / 4);
    if (h->prec != STA_PREC_F16) hipLaunchKernelGGL(ln_kernel<true>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(ln_kernel<false>, grid, dim3(256), 0, st, p);
    HIPCHK(hipGetLastError());
    return 0;
}

// pose: token index nq (== nk) of the buffers is the pose token (decoder row order, see decode_impl / AttnParams::pose)
static int run_attn(sta_handle* h, const QKVOut& qkv, const Planes& out, int ldo, int S, int heads,
                    int nq, int nk, int kv_shift, hipStream_t st, bool pose = false) {
    if (h->dry) return 0;
    REQUIRE(!pose || (nq == nk && nq + 1 <= qkv.npad), "internal: pose-token attention needs nq == nk < npad");
    AttnParams p; memset(&p, 0, sizeof p);
    p.range = h->range;
    // the pose query: 2 = one more row of the last query block when that block has spare rows (nq = 196: rows 196..255 of the
    // second block are dead anyway - free, and no latency-bound side path at SLAM scale: 12.8 vs 26.1 us for 10 x 12 heads);
    // 1 = wave-per-(sequence, head) side blocks when the patch queries fill their blocks exactly (nq = 768)
    p.pose = pose ? (nq % 128 != 0 ? 2 : 1) : 0;
    p.Q_hi = qkv.q.hi; p.Q_lo = qkv.q.lo; p.K_hi = qkv.k.hi; p.K_lo = qkv.k.lo; p.Vt_hi = qkv.vt.hi; p.Vt_lo = qkv.vt.lo;
    p.O_hi = out.hi; p.O_lo = out.lo; p.ldo = ldo;
    p.S = S; p.heads = heads; p.nq = nq; p.nk = nk; p.npad = qkv.npad; p.kv_shift = kv_shift;
    p.scale_log2e = 0.125f * 1.44269504088896340736f;
    const int npose = p.pose == 1 ? S * heads : 0;
    REQUIRE(!pose || (int64_t)(qkv.npad + 8 + 256) * 4 <= attn_smem_bytes<false>(), "internal: pose-query scratch exceeds the LDS allocation");
    REQUIRE(!pose || qkv.npad % 64 == 0, "internal: pose-query path needs npad % 64 == 0");
    dim3 grid((unsigned)(((nq + (p.pose == 2 ? 1 : 0) + 127) / 128) * heads * S + npose));
    // small grids with <= 4 key tiles: 4 LDS stages, every K / V^T tile requested up front (attention.h; one workgroup per CU then,
    // which a grid of <= 256 workgroups has anyway)
    p.prefetch = (nk <= ATT_PREFETCH_TILES * ATT_KV && grid.x <= 256 && h->opt[5] != 1) ? 1 : 0;
    const int stages = p.prefetch ? ATT_PREFETCH_TILES : 2;
    if (h->prec != STA_PREC_F16) {
        static unsigned attr_done = 0;      // one bit per device
        if (!(attr_done >> (h->device & 31) & 1u)) { hipFuncSetAttribute((const void*)attn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_smem_bytes<true>(ATT_PREFETCH_TILES)); attr_done |= 1u << (h->device & 31); }
        hipLaunchKernelGGL(attn_kernel<true>, grid, dim3(256), attn_smem_bytes<true>(stages), st, p);
    } else {
        STA_F16ONLY(hipLaunchKernelGGL(attn_kernel<false>, grid, dim3(256), attn_smem_bytes<false>(stages), st, p));
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int run_rows_to_planes(sta_handle* h, const float* x, int64_t bstride, int nb, int rows, int C, const Planes& o, hipStream_t st, int64_t obstride = 0, bool mx = false) {
    if (h->dry) return 0;
    int64_t total4 = (int64_t)nb * rows * C / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 8192) blocks = 8192;
    if (h->prec != STA_PREC_F16) hipLaunchKernelGGL(rows_to_planes_kernel<true>, dim3(blocks), dim3(256), 0, st, x, bstride, rows, C, total4, o.hi, o.lo, obstride, o.rp, mx ? 1 : 0, h->range);
    else hipLaunchKernelGGL(rows_to_planes_kernel<false>, dim3(blocks), dim3(256), 0, st, x, bstride, rows, C, total4, o.hi, o.lo, obstride, o.rp, 0, h->range);
    HIPCHK(hipGetLastError());
    return 0;
}

static int run_up2(sta_handle* h, const Planes& in, int n, int Hi, int Wi, int C, int Hc, int Wc, const Planes& out, hipStream_t st) {
    if (h->dry) return 0;
    REQUIRE(in.mx == out.mx && C % 8 == 0, "internal: bilinear format mismatch");
    // four output rows per workgroup from 64 rows on (shared taps: half the tap loads, a third of the L2 fetches); one row per
    // workgroup below that, where the grid would no longer fill the chip
    const bool quad = Hc >= 64 && (int64_t)n * ((Hc + 3) / 4) >= 512 && h->opt[7] != 1;
    const int blocks = quad ? n * ((Hc + 3) / 4) : n * Hc;
    if (h->prec != STA_PREC_F16) {
        if (quad) hipLaunchKernelGGL((bilinear_up2_kernel<true, 4>), dim3(blocks), dim3(256), 0, st, in.hi, in.lo, n, Hi, Wi, C, Hc, Wc, out.hi, out.lo, in.mx ? 1 : 0, h->range);
        else hipLaunchKernelGGL((bilinear_up2_kernel<true, 1>), dim3(blocks), dim3(256), 0, st, in.hi, in.lo, n, Hi, Wi, C, Hc, Wc, out.hi, out.lo, in.mx ? 1 : 0, h->range);
    } else {
        STA_F16ONLY(hipLaunchKernelGGL((bilinear_up2_kernel<false, 1>), dim3(blocks), dim3(256), 0, st, in.hi, in.lo, n, Hi, Wi, C, Hc, Wc, out.hi, out.lo, 0, h->range));
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int ensure_rope(sta_handle* h, int P) {
    if (P <= h->rope_P && h->rope_tab) return 0;
    // cos/sin of pos * base^(-d/16), pos = -1 .. P-1, fp32 like the reference python path (pos_embed.py:127-146)
    std::vector<float> tab((size_t)(P + 1) * 32);
    for (int pi = 0; pi <= P; ++pi)
        for (int d = 0; d < 16; ++d) {
            float inv_freq = 1.0f / powf(h->cfg.rope_base, (float)d / 16.0f);
            float f = (float)(pi - 1) * inv_freq;
            tab[((size_t)pi * 16 + d) * 2 + 0] = cosf(f);
            tab[((size_t)pi * 16 + d) * 2 + 1] = sinf(f);
        }
    HIPCHK(hipDeviceSynchronize());
    if (h->rope_tab) HIPCHK(hipFree(h->rope_tab));
    HIPCHK(hipMalloc((void**)&h->rope_tab, tab.size() * 4));
    HIPCHK(hipMemcpy(h->rope_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    h->rope_P = P;
    return 0;
}

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------ encoder
// imgs: nsets pointers of B images each -> feat [nsets*B, N, E] (fp32, caller memory = residual stream)
static int encode_impl(sta_handle* h, Bump& ws, const void* const* imgs, bool u8hwc, int nsets, int B, int H, int W,
                       float* feat, hipStream_t st) {
    const sta_config& c = h->cfg;
    const bool split = h->prec != STA_PREC_F16;
    const int E = c.enc_embed_dim, Hh = c.enc_num_heads, hp = H / 16, wp = W / 16, N = hp * wp;
    const int n = nsets * B, M = n * N, npad = rup(N, 64);
    Planes patches = ws.act(M, 768, split);
    Planes lnp = ws.act(M, E, split);
    Planes ao = ws.act(M, E, split);
    Planes f1 = ws.act(M, (int64_t)E * c.mlp_ratio, split);
    QKVOut qkv; qkv.npad = npad;
    int64_t hsz = (int64_t)n * Hh * npad * 64;
    qkv.q = ws.planes(hsz, split); qkv.k = ws.planes(hsz, split); qkv.vt = ws.planes(hsz, split);
    if (h->dry) return 0;
    REQUIRE(!ws.overflow, "internal: encode workspace overflow");
    { const Planes* z[1] = {&qkv.vt}; CHK(zero_planes(z, 1, hsz, split, st)); }
    for (int sidx = 0; sidx < nsets; ++sidx) {
        int64_t total = (int64_t)B * N * 48;
        int blocks = (int)((total + 255) / 256);
        const int64_t row0 = (int64_t)sidx * B * N;
        if (u8hwc) {
            const int b16 = (int)(((int64_t)B * N * 16 + 255) / 256);
            if (split) hipLaunchKernelGGL(patch_gather_u8hwc_kernel<true>, dim3(b16), dim3(256), 0, st, (const uint8_t*)imgs[sidx], B, H, W, patches.hi, patches.lo, row0, (int64_t)M, h->range);
            else hipLaunchKernelGGL(patch_gather_u8hwc_kernel<false>, dim3(b16), dim3(256), 0, st, (const uint8_t*)imgs[sidx], B, H, W, patches.hi, patches.lo, row0, (int64_t)M, h->range);
        } else if (split) hipLaunchKernelGGL(patch_gather_kernel<true>, dim3(blocks), dim3(256), 0, st, (const float*)imgs[sidx], B, H, W, patches.hi, patches.lo, row0, (int64_t)M, h->range);
        else hipLaunchKernelGGL(patch_gather_kernel<false>, dim3(blocks), dim3(256), 0, st, (const float*)imgs[sidx], B, H, W, patches.hi, patches.lo, row0, (int64_t)M, h->range);
        HIPCHK(hipGetLastError());
    }
    CHK(gemm_f32(h, patches, h->patch, M, feat, E, nullptr, st));
    // every in-place residual GEMM is issued together with the LayerNorm that reads its result (gemm_resid_ln)
    if (c.enc_depth > 0) CHK(run_ln(h, feat, M, E, h->enc[0].n1, lnp, nullptr, nullptr, nullptr, st));
    for (int i = 0; i < c.enc_depth; ++i) {
        const EncBlk& b = h->enc[i];
        CHK(gemm_qkv(h, lnp, b.qkv, M, E, E, E, qkv, N, Hh, wp, 0, st));
        CHK(run_attn(h, qkv, ao, E, n, Hh, N, N, 0, st));
        CHK(gemm_resid_ln(h, ao, b.proj, M, feat, E, &b.n2, &lnp, nullptr, nullptr, st));
        CHK(gemm_f16(h, lnp, b.fc1, M, f1, ACT_GELU, st));
        if (i + 1 < c.enc_depth) CHK(gemm_resid_ln(h, f1, b.fc2, M, feat, E, &h->enc[i + 1].n1, &lnp, nullptr, nullptr, st));
        else CHK(gemm_resid_ln(h, f1, b.fc2, M, feat, E, nullptr, nullptr, nullptr, nullptr, st));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ decoder
// Row order of the decoder's residual stream x (fp32, [2B*N + 2B, D]) and of every plane buffer derived from it:
//     rows [0, 2B*N)        patch tokens, sequence-major (sequence s = side * B + b, token t: row s*N + t)
//     rows [2B*N, 2B*N+2B)  the pose tokens of the 2B sequences
// (the reference prepends the pose token to every sequence, sta_model.py:206-213: M = 2B x (N + 1) interleaved rows.  Token
// order is immaterial to every layer - linears and LayerNorm are row-wise, attention is permutation-equivariant - and with
// the pose rows LAST the patch rows tile exactly: at 512x384, B = 8: 12288 = 64 x 192 rows + a 16-row tail that the GEMMs
// serve with skinny tail blocks (GemmParams::m_tail) and the attention kernel with its pose path (AttnParams::pose),
// instead of a 65th tile row, a 7th query block and a 13th key tile everywhere.)
// Outputs through the `want` tables, after layer i (i = 0: decoder input):
//   ref_layout:  want1[i] / want2[i] = [B, N+1, D] per side in the reference's token order (pose token first), or NULL
//   otherwise:   want1[i] = the whole x in the row order above (internal consumers: sta_forward_pair, sta_regress_views)
struct TailHint { sta_handle* h; TailHint(sta_handle* h_, int t) : h(h_) { h->tail_hint = t; } ~TailHint() { h->tail_hint = 0; } };
static int decode_impl(sta_handle* h, Bump& ws, const float* feat1, const float* feat2, int B, int hp, int wp,
                       float* x, float* const* want1, float* const* want2, bool ref_layout, hipStream_t st) {
    const sta_config& c = h->cfg;
    const bool split = h->prec != STA_PREC_F16;
    const int E = c.enc_embed_dim, D = c.dec_embed_dim, Hh = c.dec_num_heads;
    const int N = hp * wp, Np = N + 1, S = 2 * B, M = S * Np, Mp = S * N, npad = rup(Np, 64);
    Planes fp = ws.act((int64_t)S * N, E, split);
    Planes a1 = ws.act(M, D, split);
    Planes ay = ws.act(M, D, split);
    Planes ao = ws.act(M, D, split);
    Planes f1 = ws.act(M, (int64_t)D * c.mlp_ratio, split);
    QKVOut qkv; qkv.npad = npad;
    int64_t hsz = (int64_t)S * Hh * npad * 64;
    qkv.q = ws.planes(hsz, split); qkv.k = ws.planes(hsz, split);
    QKVOut cqkv; cqkv.npad = npad;          // cross attention: its K / V^T are produced while the self-attention set is live
    cqkv.q = ws.planes(hsz, split); cqkv.k = ws.planes(hsz, split);
    qkv.vt = ws.planes(hsz, split); cqkv.vt = ws.planes(hsz, split);      // back to back: one fill zeroes both paddings
    if (h->dry) return 0;
    REQUIRE(!ws.overflow, "internal: decode workspace overflow");
    { const Planes* z[2] = {&qkv.vt, &cqkv.vt}; CHK(zero_planes(z, 2, hsz, split, st)); }

    Planes fp2 = slice_rows(fp, (int64_t)B * N);
    if (feat2 == feat1 + (size_t)B * N * E) {          // both sides in one buffer (sta_forward_pair, the scheduler): one launch
        CHK(run_rows_to_planes(h, feat1, (int64_t)N * E, 2 * B, N, E, fp, st));
    } else {
        CHK(run_rows_to_planes(h, feat1, (int64_t)N * E, B, N, E, fp, st));
        CHK(run_rows_to_planes(h, feat2, (int64_t)N * E, B, N, E, fp2, st));
    }
    CHK(gemm_f32(h, fp, h->dec_embed, Mp, x, D, nullptr, st));
    float* xpose = x + (size_t)Mp * D;
    hipLaunchKernelGGL(fill_pose_token_kernel, dim3((S * D + 255) / 256), dim3(256), 0, st, xpose, S, 1, D, h->pose_tok);
    HIPCHK(hipGetLastError());
    auto emit = [&](int idx, const float* src) -> int {
        if (!ref_layout) {
            if (want1 && want1[idx] && want1[idx] != src) HIPCHK(hipMemcpyAsync(want1[idx], src, (size_t)M * D * 4, hipMemcpyDeviceToDevice, st));
            return 0;
        }
        for (int side = 0; side < 2; ++side) {
            float* dst = side == 0 ? (want1 ? want1[idx] : nullptr) : (want2 ? want2[idx] : nullptr);
            if (!dst) continue;
            const int64_t total4 = (int64_t)B * Np * D / 4;
            int blocks = (int)((total4 + 255) / 256); if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(emit_tokens_kernel, dim3(blocks), dim3(256), 0, st, src, side * B, B, N, D, (int64_t)Mp, dst);
            HIPCHK(hipGetLastError());
        }
        return 0;
    };
    CHK(emit(0, x));
    TailHint tail(h, S);                 // every dense GEMM below: the last S rows are the pose-token rows
    // norm1(x) and norm_y(x) from one read: y of one side == x of the other (sta_model.py:231-235); qkv and projk|projv:
    // one class, one plane format.  Layer i+1's pair is issued with layer i's mlp.fc2 (gemm_resid_ln).
    if (c.dec_depth > 0) CHK(run_ln(h, x, M, D, h->dec[0].n1, a1, &h->dec[0].ny, &ay, nullptr, st));
    LaneJoin join_on_exit{h, st, false};          // armed by the first layer that forks
    for (int i = 0; i < c.dec_depth; ++i) {
        const DecBlk& b = h->dec[i];
        // self-attention q,k,v and the cross-attention k,v of the OTHER side depend only on the layer input: one launch.
        // (They write disjoint buffers: qkv / ckv_out.)
        // Below the paired launch's scale (two launches + two split-K finishers) the cross-attention K / V run on the context's
        // SIDE stream under the self-attention chain (qkv, attention, proj + norm2, cross q) and join before the cross attention.
        bool forked = false;
        {
            GemmParams pq, pkv;
            CHK(gp_qkv(h, pq, a1, b.qkv, M, D, D, D, qkv, N, Hh, wp, 0, Mp));
            CHK(gp_qkv(h, pkv, ay, b.ckv, M, 0, D, D, cqkv, N, Hh, wp, 0, Mp));
            if (lanes_on(h) && !qkv_pair_one_launch(h, pq, pkv)) {
                CHK(ensure_side(h));
                join_on_exit.armed = true;
                hipEvent_t* ev = h->cur->side_ev + 2 * (i & 1);
                HIPCHK(hipEventRecord(ev[0], st));
                HIPCHK(hipStreamWaitEvent(h->cur->side, ev[0], 0));
                {
                    Lane lane(h, 1);
                    CHK((launch_gemm<A_DENSE, EPI_QKV>(h, pkv, h->cur->side)));
                }
                HIPCHK(hipEventRecord(ev[1], h->cur->side));
                CHK((launch_gemm<A_DENSE, EPI_QKV>(h, pq, st)));
                forked = true;
            } else {
                CHK(gemm_qkv_pair(h, pq, pkv, st));
            }
        }
        CHK(run_attn(h, qkv, ao, D, S, Hh, N, N, 0, st, true));
        CHK(gemm_resid_ln(h, ao, b.proj, M, x, D, &b.n2, &a1, nullptr, nullptr, st));
        CHK(gemm_qkv(h, a1, b.cq, M, D, 0, 0, cqkv, N, Hh, wp, 0, st, Mp));
        if (forked) HIPCHK(hipStreamWaitEvent(st, h->cur->side_ev[2 * (i & 1) + 1], 0));
        CHK(run_attn(h, cqkv, ao, D, S, Hh, N, N, B, st, true));
        CHK(gemm_resid_ln(h, ao, b.cproj, M, x, D, &b.n3, &a1, nullptr, nullptr, st));
        CHK(gemm_f16(h, a1, b.fc1, M, f1, ACT_GELU, st));
        if (i + 1 < c.dec_depth) {
            const DecBlk& nb = h->dec[i + 1];
            CHK(gemm_resid_ln(h, f1, b.fc2, M, x, D, &nb.n1, &a1, &nb.ny, &ay, st));
            CHK(emit(i + 1, x));
        } else {   // final_x[-1] = dec_norm(final_x[-1])  (sta_model.py:241-242)
            CHK(gemm_resid_ln(h, f1, b.fc2, M, x, D, nullptr, nullptr, nullptr, nullptr, st));
            const bool wanted = (want1 && want1[i + 1]) || (ref_layout && want2 && want2[i + 1]);
            if (wanted) {
                Planes none;
                float* dst = ref_layout ? x : want1[i + 1];          // x is dead after the last layer: normalise it in place
                CHK(run_ln(h, x, M, D, h->dec_norm, none, nullptr, nullptr, dst, st));
                CHK(emit(i + 1, dst));
            }
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ pose head
// pose2 / conf2 (optional): the last B - split samples write there (the two sides of a pair: one set of four launches)
static int pose_impl(sta_handle* h, Bump& ws, const float* tok, int B, int64_t stride, float* pose, float* conf, hipStream_t st,
                     float* pose2 = nullptr, float* conf2 = nullptr, int split = 0) {
    const int D = h->cfg.dec_embed_dim, Hd = 512;
    float* f0 = (float*)ws.take((int64_t)B * Hd * 4);
    float* f1 = (float*)ws.take((int64_t)B * Hd * 4);
    if (h->dry) return 0;
    REQUIRE(!ws.overflow, "internal: pose workspace overflow");
    PoseParams p;
    p.tok = tok; p.tok_stride = stride; p.D = D; p.Hd = Hd;
    p.w0 = h->pm0.w; p.b0 = h->pm0.b; p.w1 = h->pm1.w; p.b1 = h->pm1.b; p.w2 = h->pm2.w; p.b2 = h->pm2.b;
    p.wt = h->pt.w; p.bt = h->pt.b; p.wr = h->pr.w; p.br = h->pr.b; p.wc = h->pc.w; p.bc = h->pc.b;
    p.pose = pose; p.conf = conf; p.pose2 = pose2; p.conf2 = conf2; p.split = split;
    dim3 grid(Hd / 4, B);
    hipLaunchKernelGGL(pose_layer_kernel, grid, dim3(256), 0, st, tok, stride, p.w0, p.b0, f0, D, Hd, 1);
    hipLaunchKernelGGL(pose_layer_kernel, grid, dim3(256), 0, st, f0, (int64_t)Hd, p.w1, p.b1, f1, Hd, Hd, 1);
    hipLaunchKernelGGL(pose_layer_kernel, grid, dim3(256), 0, st, f1, (int64_t)Hd, p.w2, p.b2, f0, Hd, Hd, 1);
    hipLaunchKernelGGL(pose_final_kernel, dim3(B), dim3(256), 0, st, p, f0);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------ DPT head
struct RcuTmp { Planes t, y; };
static int run_rcu(sta_handle* h, const Planes& x, int n, int Hh, int Ww, const RCU& u, const Planes& tmp,
                   const Planes& out, const Planes* extra, hipStream_t st) {
    // out = x + conv2(relu(conv1(relu(x)))) [+ extra]   (dpt_block.py:121-142, 196-204)
    CHK(conv3(h, x, n, Hh, Ww, 256, u.c1, 1, true, ACT_RELU, tmp, nullptr, nullptr, st));
    CHK(conv3(h, tmp, n, Hh, Ww, 256, u.c2, 1, false, ACT_NONE, out, &x, extra, st));
    return 0;
}

// outputs: first nA images -> (ptsA, confA), remaining -> (ptsB, confB)
static int dpt_impl(sta_handle* h, Bump& ws, const float* enc, int64_t enc_bs,
                    const float* h1, int64_t h1_bs, const float* h2, int64_t h2_bs, const float* h3, int64_t h3_bs,
                    int n, int H, int W, float* ptsA, float* confA, int nA, float* ptsB, float* confB, hipStream_t st) {
    const sta_config& c = h->cfg;
    const bool split = h->prec != STA_PREC_F16;
    const int E = c.enc_embed_dim, D = c.dec_embed_dim, hp = H / 16, wp = W / 16, N = hp * wp;
    const int M = n * N;
    // precision f16mx: every DPT buffer is in the f16mx row format, except the input of act_postprocess[0]
    // (N = 96: no f16mx kernel for that one GEMM, it reads f16x3 rows and WRITES f16mx rows)
    const bool dmx = (h->mx_mask & CLS_HEAD) != 0;
    auto act = [&](int64_t rows, int64_t cols, bool mx) { Planes q = ws.act(rows, cols, split); q.mx = mx; return q; };
    // Two lanes (round 4).  The head is a chain - level 3 (1/32 scale) -> refinenet4 -> refinenet3 -> refinenet2 -> refinenet1 -> head -
    // with three side branches feeding it: the reassembly of levels 2, 1, 0 (rows -> planes, act_postprocess, layer_rn) and the first
    // convolution of each refinenet's resConfUnit1, which reads only that level.  The 14 side-branch launches run on the context's
    // SIDE stream under the chain: fork at entry, one event per level (2, 1, 0) that the chain waits on right before it consumes
    // that level.  At SLAM scale every kernel of the head is a fraction of a round of workgroups and a launch costs as much as the
    // kernel (tools/model_stamps.py: 10 - 25 us of event time around 5 - 17 us of work): 5-edge scheduler call -5.8 %, one pair
    // @512x384 +2.3 %; at the benchmark's 8 pairs the side kernels fill the chain's partial rounds (+0.8 %).  Same kernels, same
    // split-K slices (the side lane has its own scratch): bit-identical to the one-lane order (tests/test_gpu_parity.py).
    // sta_debug_set_option(h, 6, 1) switches the lane off (A/B: tools/ab_option.py 6 1 0); the whole-model timing modes (stage timing, per-launch timing of every GEMM, stamps) run one lane.
    const bool two = lanes_on(h);
    hipStream_t sb = st;                                                  // the side branches' stream
    if (two) {
        CHK(ensure_side(h));
        sb = h->cur->side;
        HIPCHK(hipEventRecord(h->cur->side_ev[3], st));
        HIPCHK(hipStreamWaitEvent(sb, h->cur->side_ev[3], 0));
    }
    LaneJoin join_on_exit{h, st, two};
    Planes t0 = act(M, E, use_mx(h, h->act0_0)), t1 = act(M, D, dmx);
    Planes t2 = act(M, D, dmx), t3 = act(M, D, dmx);
    // act_postprocess (dpt_block.py:356-410)
    Planes a0 = act(M, 96, dmx), l0 = act((int64_t)M * 16, 96, dmx);
    Planes a1 = act(M, 192, dmx), l1 = act((int64_t)M * 4, 192, dmx);
    Planes l2 = act(M, 384, dmx);
    Planes a3 = act(M, 768, dmx);
    const int h3s = (hp - 1) / 2 + 1, w3s = (wp - 1) / 2 + 1;
    Planes l3 = act((int64_t)n * h3s * w3s, 768, dmx);
    // layer_rn (3x3, no bias) -> 256 channels at 4x, 2x, 1x, 1/2x
    const int Hs[4] = {4 * hp, 2 * hp, hp, h3s}, Ws[4] = {4 * wp, 2 * wp, wp, w3s};
    const int Cs[4] = {96, 192, 384, 768};
    Planes lin[4] = {l0, l1, l2, l3}, r[4], c1o[4];          // c1o[k]: relu(conv1(relu(r[k]))) of refinenet k's resConfUnit1 (k < 3)
    for (int k = 0; k < 4; ++k) {
        r[k] = act((int64_t)n * Hs[k] * Ws[k], 256, dmx);
        if (k < 3) c1o[k] = act((int64_t)n * Hs[k] * Ws[k], 256, dmx);
    }
    REQUIRE(!ws.overflow, "internal: dpt workspace overflow (stage 1)");
    {   // level 3 opens the chain
        CHK(run_rows_to_planes(h, h3, h3_bs, n, N, D, t3, st, 0, t3.mx));
        CHK(gemm_f16(h, t3, h->act3_0, M, a3, ACT_NONE, st, a3.mx));
        CHK(conv3(h, a3, n, hp, wp, 768, h->act3_1, 2, false, ACT_NONE, l3, nullptr, nullptr, st));
        CHK(conv3(h, l3, n, Hs[3], Ws[3], Cs[3], h->rn[3], 1, false, ACT_NONE, r[3], nullptr, nullptr, st));
    }
    {   // side branches, in the order the chain consumes them: level 2, 1, 0
        Lane lane(h, two ? 1 : 0);
        for (int k = 2; k >= 0; --k) {
            if (k == 2) {
                CHK(run_rows_to_planes(h, h2, h2_bs, n, N, D, t2, sb, 0, t2.mx));
                CHK(gemm_f16(h, t2, h->act2_0, M, l2, ACT_NONE, sb, l2.mx));
            } else if (k == 1) {
                CHK(run_rows_to_planes(h, h1, h1_bs, n, N, D, t1, sb, 0, t1.mx));
                CHK(gemm_f16(h, t1, h->act1_0, M, a1, ACT_NONE, sb, a1.mx));
                CHK(gemm_convt(h, a1, h->act1_1, n, hp, wp, 2, 192, l1, sb));
            } else {
                CHK(run_rows_to_planes(h, enc, enc_bs, n, N, E, t0, sb, 0, t0.mx));
                CHK(gemm_f16(h, t0, h->act0_0, M, a0, ACT_NONE, sb, a0.mx));
                CHK(gemm_convt(h, a0, h->act0_1, n, hp, wp, 4, 96, l0, sb));
            }
            CHK(conv3(h, lin[k], n, Hs[k], Ws[k], Cs[k], h->rn[k], 1, false, ACT_NONE, r[k], nullptr, nullptr, sb));
            CHK(conv3(h, r[k], n, Hs[k], Ws[k], 256, h->ref[k].u1.c1, 1, true, ACT_RELU, c1o[k], nullptr, nullptr, sb));
            if (two) HIPCHK(hipEventRecord(h->cur->side_ev[k], sb));
        }
    }
    // refinenet4 .. refinenet1.  out_conv (1x1) commutes with the bilinear upsample (both linear, the
    // interpolation weights sum to 1), so it runs BEFORE the x2 upsample at 1/4 of the FLOPs.
    Planes path;   // upsampled output of the previous stage
    int ph = 0, pw = 0;
    for (int k = 3; k >= 0; --k) {
        const Refine& rf = h->ref[k];
        const int hh = Hs[k], ww = Ws[k];
        const int64_t el = (int64_t)n * hh * ww;
        Planes tmp = act(el, 256, dmx), cur = r[k];
        if (k < 3) {
            REQUIRE(ph == hh && pw == ww, "internal: refinenet size mismatch %dx%d vs %dx%d", ph, pw, hh, ww);
            Planes sum = act(el, 256, dmx);
            REQUIRE(!ws.overflow, "internal: dpt workspace overflow (fusion)");
            if (two) HIPCHK(hipStreamWaitEvent(st, h->cur->side_ev[k], 0));
            // path + RCU1(layer) = path + r[k] + conv2(c1o[k])   (dpt_block.py:121-142, 196-204)
            CHK(conv3(h, c1o[k], n, hh, ww, 256, rf.u1.c2, 1, false, ACT_NONE, sum, &r[k], &path, st));
            cur = sum;
        }
        Planes y = act(el, 256, dmx), z = act(el, 256, dmx);
        REQUIRE(!ws.overflow, "internal: dpt workspace overflow (rcu2)");
        CHK(run_rcu(h, cur, n, hh, ww, rf.u2, tmp, y, nullptr, st));
        CHK(gemm_f16(h, y, rf.out, n * hh * ww, z, ACT_NONE, st, z.mx));
        // upsample x2 (align_corners) ; refinenet4 output is cropped to the layers[2] size (dpt_head.py:58)
        int oh = 2 * hh, ow = 2 * ww;
        if (k == 3) { if (oh > Hs[2]) oh = Hs[2]; if (ow > Ws[2]) ow = Ws[2]; }
        Planes up = act((int64_t)n * oh * ow, 256, dmx);
        REQUIRE(!ws.overflow, "internal: dpt workspace overflow (up)");
        CHK(run_up2(h, z, n, hh, ww, 256, oh, ow, up, st));
        path = up; ph = oh; pw = ow;
    }
    // head: 3x3 256->128, up x2, 3x3 128->128 + ReLU, 1x1 128->4 + postprocess (dpt_block.py:316-324)
    Planes h0 = act((int64_t)n * ph * pw, 128, dmx);
    Planes h0u = act((int64_t)n * H * W, 128, dmx);
    const bool fused_tail = conv3_head_ok(h, h->head2, (int64_t)n * H * W);     // the [pixels,128] map of head.2 stays on chip
    Planes h2o; if (!fused_tail) h2o = act((int64_t)n * H * W, 128, dmx);
    REQUIRE(!ws.overflow, "internal: dpt workspace overflow (head)");
    REQUIRE(2 * ph == H && 2 * pw == W, "internal: head size mismatch");
    CHK(conv3(h, path, n, ph, pw, 256, h->head0, 1, false, ACT_NONE, h0, nullptr, nullptr, st));
    CHK(run_up2(h, h0, n, ph, pw, 128, H, W, h0u, st));
    if (fused_tail) return conv3_head(h, h0u, n, H, W, 128, h->head2, h->head4, ptsA, confA, nA, ptsB, confB, st);
    CHK(conv3(h, h0u, n, H, W, 128, h->head2, 1, false, ACT_RELU, h2o, nullptr, nullptr, st));
    for (int part = 0; part < 2 && !h->dry; ++part) {
        int i0 = part == 0 ? 0 : nA, cnt = part == 0 ? (nA < n ? nA : n) : n - nA;
        if (cnt <= 0) continue;
        float* pp = part == 0 ? ptsA : ptsB; float* cp = part == 0 ? confA : confB;
        int64_t npix = (int64_t)cnt * H * W;
        const int64_t pix0 = (int64_t)i0 * H * W;
        int blocks = (int)((npix * 16 + 255) / 256); if (blocks > 16384) blocks = 16384;
        if (split) hipLaunchKernelGGL(head_final_kernel<true>, dim3(blocks), dim3(256), 0, st, h2o.hi, h2o.lo, pix0, h2o.rp, npix, h->head4.w, h->head4.b, pp, cp, h2o.mx ? 1 : 0);
        else hipLaunchKernelGGL(head_final_kernel<false>, dim3(blocks), dim3(256), 0, st, h2o.hi, h2o.lo, pix0, h2o.rp, npix, h->head4.w, h->head4.b, pp, cp, 0);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

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
