// MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T + bias[N] )
//
// * v_mfma_f32_32x32x16_f16, fp32 accumulate.  SPLIT=true issues three products per tile step
//   (Ah*Wh + Al*Wh + Ah*Wl) on 2-term fp16 splits of both operands: fp32-class accuracy at 1/3
//   of the fp16 MFMA rate (gfx950 has no TF32/XF32 MFMA; the reference runs TF32).
// * 128x128x32 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles (64 accumulator regs).
// * A and W are both K-contiguous, staged global -> registers -> LDS (double-buffered, one barrier
//   per K step); LDS rows are 64 B with the 16-B chunk index XOR-swizzled by (row>>2)&3 so the
//   fragment ds_read_b128 is bank-conflict free.
// * A loader variants: dense rows, or 3x3 implicit-GEMM gather over an NHWC image (pad 1,
//   stride 1/2, optional ReLU on load).
// * Epilogues: fp32 (+bias, +residual, row remap), fp16 planes (+bias, GELU/ReLU, up to two
//   residual plane sets), QKV (+bias, 2-D RoPE via lane shuffle, head-major Q/K and transposed V),
//   transposed-conv pixel scatter.
//
// Replaces: every nn.Linear / nn.Conv2d / nn.ConvTranspose2d on the STA path
// (sta_blocks.py:73-79,132,146,193-195,207; dpt_block.py:20-77,94-112,178-186,316-324,356-410)
// and the curope rotary kernel (pos_embed/curope/kernels.cu:17-82) fused into the QK epilogue.
#pragma once
#include "sta_common.h"

enum { A_DENSE = 0, A_CONV3 = 1 };
enum { EPI_F32 = 0, EPI_F16 = 1, EPI_QKV = 2, EPI_CONVT = 3,
       EPI_GELU = 4,     // mlp.fc1: fp16-plane epilogue with the activation fixed at compile time and no residual planes
       EPI_F32R = 5,     // attn.proj / mlp.fc2 at throughput scale: fp32 output added IN PLACE to the residual stream,
                         // no row remap, no split-K - the common case of EPI_F32 without its per-element flag branches
       EPI_HEAD = 6 };   // DPT head tail at throughput scale: 3x3 conv 128->128 + ReLU (head.2/.3) whose 192x128 tile holds every
                         // channel of its pixels, so the 1x1 conv 128->4 (head.4) and the point-map / confidence activations
                         // (postprocess.py:10-62) run in the epilogue and the [pixels,128] map never reaches HBM (gemm2.h)
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2 };

struct GemmParams {
    // ---- A operand
    const f16* A_hi; const f16* A_lo; int lda;          // blocked planes (A_lo == A_hi + 32 in f16x3); lda unused
    int64_t a_rp;                                        // rows of the A planes (pixels of the conv input)
    int Hi, Wi, Cin, Ho, Wo, cstride, relu_in;           // conv3x3: NHWC [nimg,Hi,Wi,Cin] -> [nimg,Ho,Wo,*]
    // ---- B operand (weights, [N,K] planes) and bias
    const f16* B_hi; const f16* B_lo; const float* bias;
    int M, N, K;
    int m_tail;                                          // gemm2 families: the last m_tail (<= 32) rows are computed by skinny tail
                                                         //     blocks (one 32-row MFMA tile per wave, K split over the waves) instead
                                                         //     of a (BM-row) tile row of their own; (M - m_tail) % BM == 0
    int mx;                                              // A and B are f16mx rows (sta_common.h); dense GEMMs only
    // ---- EPI_F32
    float* C32; int ldc; const float* resid; int ldr;
    int ksplit;                                          // >1: split-K, every slice atomically adds into C32 (which already
                                                         //     holds the residual); bias is added by slice 0 only
    float* slab = nullptr;                               // split-K without atomics: slice s stores its partial tile (slice 0 with
                                                         //     the bias) to slab[s][M][N]; ln_kernel adds the slices to the residual
                                                         //     stream and applies the LayerNorm that follows (small-M regime)
    // ---- EPI_F16
    f16* C_hi; f16* C_lo; int ldc16; int act;            // blocked output planes with c_rp rows (ldc16 unused)
    int c_mx;                                            // EPI_F16 / EPI_CONVT output in the f16mx row format (consumer = f16mx GEMM)
    int r_mx;                                            // residual planes R1 / R2 are f16mx rows
    int64_t c_rp;
    unsigned long long* range;                           // the handle's range counters (sta_common.h RangeAcc::flush)
    float* skbuf = nullptr;                              // EPI_F16 split-K: fp32 partial tiles [ksplit][M,N], one slab per K slice (no
                                                         //     atomics); splitk_finish_kernel sums them, applies bias / activation /
                                                         //     residual planes and writes the planes
    const f16* R1_hi; const f16* R1_lo; const f16* R2_hi; const f16* R2_lo;
    // ---- EPI_QKV
    f16* Q_hi; f16* Q_lo; f16* K_hi; f16* K_lo; f16* Vt_hi; f16* Vt_lo;
    int nq, nk, nv, ntok, npad, heads, wp, has_pose_tok;
    int pose_base;                                       // > 0: rows >= pose_base are the pose tokens of sequences row - pose_base,
                                                         //     stored at token index ntok of the Q / K / V^T buffers (decoder row order
                                                         //     [S x ntok patch rows | S pose rows]; has_pose_tok must be 0)
    unsigned ntok_magic, wp_magic;                       // floor(2^32/d)+1 (0 when d == 1): exact n/d for n*d < 2^32
    const float* rope_tab;                               // [(pos+1)][16][2] cos,sin ; pos = -1 .. P-1
    // ---- EPI_CONVT
    int ct_k, ct_cout, ct_h, ct_w;
    // ---- EPI_HEAD: head.4 weights [4][128] / bias [4] (fp32); pixels [0, hsplit) -> (hptsA, hconfA), the rest -> (hptsB, hconfB)
    const float* hw4; const float* hb4; float* hptsA; float* hconfA; float* hptsB; float* hconfB; int64_t hsplit;
    float hw4_scale[4] = {1.f, 1.f, 1.f, 1.f};           // powers of two, one per output row: head_epilogue_t splits hw4[o][:] * hw4_scale[o] into fp16 planes and divides the sums by it
    // ---- misc
    const f16* zero_page;                                // >= 64 B of zeros (OOB taps of the direct-to-LDS conv loader)
    unsigned long long* clk_dbg = nullptr;               // bench only: block 0 stores {shader cycles, 100 MHz ticks} of its lifetime
    unsigned long long* stamps = nullptr;                // bench only (sta_bench_gemm_stamps): every workgroup stores 4 x s_memrealtime (100 MHz):
                                                         // kernel entry, first K tile landed, main loop done, epilogue done
};

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 32
#define GEMM_TILE_BYTES (128 * 64)   // one 128 x 32 fp16 tile

template <bool SPLIT>
constexpr int gemm_smem_bytes() { return 2 * (SPLIT ? 4 : 2) * GEMM_TILE_BYTES; }

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

__device__ __forceinline__ uint4 relu_pair_hi(uint4 hi, uint4& lo) {
    // relu on (hi+lo): the sign of hi decides (hi==0 implies lo==0).
    H8 a, b; a.u = hi; b.u = lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bool neg = a.e[i] < (f16)0;
        a.e[i] = neg ? (f16)0 : a.e[i];
        b.e[i] = neg ? (f16)0 : b.e[i];
    }
    lo = b.u;
    return a.u;
}

__device__ __forceinline__ int fast_div(int n, int d, unsigned magic) {
    return magic ? (int)__umulhi((unsigned)n, magic) : n;   // magic == 0 encodes d == 1
}
// QKV epilogue: GEMM row -> (sequence, token index in the Q / K / V^T buffers, is it the pose token)
__device__ __forceinline__ void qkv_row_token(const GemmParams& p, int row, int& s, int& t, bool& pose) {
    if (p.pose_base > 0 && row >= p.pose_base) { s = row - p.pose_base; t = p.ntok; pose = true; }
    else { s = fast_div(row, p.ntok, p.ntok_magic); t = row - s * p.ntok; pose = p.has_pose_tok && t == 0; }
}
// table row of the RoPE cos/sin table for token t along the y (xpart == 0) or x axis; row 0 == position -1 (pose token)
__device__ __forceinline__ int qkv_rope_pos(const GemmParams& p, int t, bool pose, int xpart) {
    if (pose) return 0;
    const int tt = p.has_pose_tok ? t - 1 : t;
    const int ty = fast_div(tt, p.wp, p.wp_magic);
    return (xpart ? tt - ty * p.wp : ty) + 1;
}
// 8-byte store of 4 halves at 2-byte alignment (a V^T run starts at any token index): the copy carries the destination's real
// alignment (an assignment through an under-aligned vector typedef is what -Walign-mismatch warned about); gfx950 global
// memory takes unaligned dwordx2 stores, and hipcc emits one (checked in the ISA of qkv_finish_kernel)
__device__ __forceinline__ void store8_a2(f16* dst, const uint2& v) { __builtin_memcpy(dst, &v, 8); }

// QKV epilogue of one 32x32 accumulator tile: +bias, 2-D RoPE on q/k (pair partner = lane^16, cos/sin
// from the table), head-major Q/K stores, and V written TRANSPOSED ([d][token]) - a lane owns one d
// column and 4 consecutive tokens per register group, i.e. one 8-byte run of V^T per group.
template <bool SPLIT>
__device__ __forceinline__ void epilogue_qkv_tile(const GemmParams& p, const floatx16& acc, int row0, int col, int lane, char* wave_lds = nullptr) {
    const int lhi = lane >> 5;
    const bool col_ok = col < p.N;
    RangeAcc ra;                      // never flushed: q / k / v are linear maps of LayerNorm outputs (bounded by sqrt(C) x the gains x
                                      // the weights), the range report (sta_common.h) covers the unnormalised tensors instead
    const float bv = (p.bias != nullptr && col_ok) ? p.bias[col] : 0.f;
    const int cbase = (col - (lane & 31)) & ~63;
    int seg = 0, cc = cbase;
    if (cbase >= p.nq + p.nk) { seg = 2; cc = cbase - p.nq - p.nk; }
    else if (cbase >= p.nq) { seg = 1; cc = cbase - p.nq; }
    const int head = cc >> 6, dcol = col - cbase, xpart = (dcol >> 5) & 1;
    if (seg == 2) {
        // V^T through the wave's LDS scratch (QKV_LDS_BYTES, gemm2_body): a lane owns ONE d column and 16 tokens, so written
        // straight from the accumulators a store instruction scatters 64 x 8 B into 64 different 128-B lines (measured: the V
        // third of this epilogue cost 1.6 % of the whole step).  Transposed in LDS ([d][32 tokens], 80-B rows) the same tile
        // leaves as 2 x 2 instructions of 16 B per lane: 64 contiguous bytes per d row.  Fast path: the 32 rows are patch /
        // token rows of ONE sequence and 16-B aligned in V^T; everything else (sequence boundaries, M tail, pose rows) below.
        if (wave_lds != nullptr && __all(col_ok) && row0 + 32 <= (p.pose_base > 0 ? p.pose_base : p.M)) {
            const int s0 = fast_div(row0, p.ntok, p.ntok_magic), t0 = row0 - s0 * p.ntok;      // wave-uniform
            if (t0 + 32 <= p.ntok && (t0 & 7) == 0) {
                char* Lh = wave_lds; char* Ll = wave_lds + 32 * 80;
                const int d = lane & 31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    H4 ph, pl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[g * 4 + e] + bv;
                        if (SPLIT) split_f16(v, ph.e[e], pl.e[e], ra); else { ph.e[e] = to_f16_sat(v, ra); pl.e[e] = (f16)0; }
                    }
                    *reinterpret_cast<uint2*>(Lh + d * 80 + (8 * g + 4 * lhi) * 2) = ph.u;
                    if (SPLIT) *reinterpret_cast<uint2*>(Ll + d * 80 + (8 * g + 4 * lhi) * 2) = pl.u;
                }
                const int dbase = dcol - d;                      // first d of this 32-column tile (0 or 32): wave-uniform
                const size_t obase = ((size_t)(s0 * p.heads + head) * 64 + dbase) * p.npad + t0;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int dr = (lane >> 2) + 16 * half, c = lane & 3;
                    const size_t o = obase + (size_t)dr * p.npad + c * 8;
                    *reinterpret_cast<uint4*>(p.Vt_hi + o) = *reinterpret_cast<const uint4*>(Lh + dr * 80 + c * 16);
                    if (SPLIT) *reinterpret_cast<uint4*>(p.Vt_lo + o) = *reinterpret_cast<const uint4*>(Ll + dr * 80 + c * 16);
                }
                return;
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int rowg = row0 + 8 * g + 4 * lhi;
            const int rc = rowg < p.M ? rowg : p.M - 1;
            int s, t; bool pose_row;
            qkv_row_token(p, rc, s, t, pose_row);
            f16 hh[4], ll[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = acc[g * 4 + e] + bv;
                if (SPLIT) split_f16(v, hh[e], ll[e], ra); else { hh[e] = to_f16_sat(v, ra); ll[e] = (f16)0; }
            }
            const size_t o = ((size_t)(s * p.heads + head) * 64 + dcol) * p.npad + t;
            if (col_ok && rowg + 3 < p.M && t + 3 < p.ntok) {      // 4 tokens of one sequence: one run
                H4 ph, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) { ph.e[e] = hh[e]; pl.e[e] = ll[e]; }
                store8_a2(p.Vt_hi + o, ph.u);
                if (SPLIT) store8_a2(p.Vt_lo + o, pl.u);
            } else if (col_ok) {                                      // sequence boundary / M tail
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = rowg + e;
                    if (row < p.M) {
                        int s2, t2; bool pr2;
                        qkv_row_token(p, row, s2, t2, pr2);
                        const size_t o2 = ((size_t)(s2 * p.heads + head) * 64 + dcol) * p.npad + t2;
                        p.Vt_hi[o2] = hh[e];
                        if (SPLIT) p.Vt_lo[o2] = ll[e];
                    }
                }
            }
        }
        return;
    }
    f16* const dh = seg == 0 ? p.Q_hi : p.K_hi;
    f16* const dl = seg == 0 ? p.Q_lo : p.K_lo;
    // Fast path: the 32 rows are consecutive patch tokens of ONE sequence (every tile of the decoder's [patch rows | pose rows]
    // order and of the encoder at 768 tokens).  (sequence, token, grid position) of row0 are wave-uniform - one division on
    // the scalar unit instead of two magic divisions per element - and row k of the tile is token t0 + k; the rotated values
    // go through the wave's LDS scratch ([token][32 d] rows of 64 B per plane) and leave as 16 B per lane (2 x 2 store
    // instructions instead of 32 of 2 B per lane).
    if (wave_lds != nullptr && __all(col_ok) && row0 + 32 <= (p.pose_base > 0 ? p.pose_base : p.M)) {
        const int s0 = fast_div(row0, p.ntok, p.ntok_magic), t0 = row0 - s0 * p.ntok;
        if (t0 + 32 <= p.ntok && !(p.has_pose_tok && t0 == 0) && p.wp >= 11) {
            const int tt0 = p.has_pose_tok ? t0 - 1 : t0;                  // index in the patch grid
            const int y0 = fast_div(tt0, p.wp, p.wp_magic), x0 = tt0 - y0 * p.wp;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float2 cs8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = half * 8 + q, k = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    int x = x0 + k, y = y0;                                    // x0 < wp, k <= 31, wp >= 11: at most three wraps
#pragma unroll
                    for (int w = 0; w < 3; ++w) { const int over = x >= p.wp ? 1 : 0; x -= over * p.wp; y += over; }
                    const int pos = (xpart ? x : y) + 1;                       // table row 0 == position -1
                    cs8[q] = *reinterpret_cast<const float2*>(p.rope_tab + ((size_t)pos * 16 + (lane & 15)) * 2);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = half * 8 + q, k = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const float2 cs = cs8[q];
                    float v = acc[r] + bv;
                    const float other = __shfl_xor(v, 16);
                    v = (lane & 16) ? (v * cs.x + other * cs.y) : (v * cs.x - other * cs.y);
                    if (SPLIT) {
                        f16 h, l; split_f16(v, h, l, ra);
                        *reinterpret_cast<f16*>(wave_lds + k * 80 + (lane & 31) * 2) = h;
                        *reinterpret_cast<f16*>(wave_lds + 32 * 80 + k * 80 + (lane & 31) * 2) = l;
                    } else *reinterpret_cast<f16*>(wave_lds + k * 80 + (lane & 31) * 2) = to_f16_sat(v, ra);
                }
            }
            const int dbase = dcol - (lane & 31);
            const size_t obase = ((size_t)(s0 * p.heads + head) * p.npad + t0) * 64 + dbase;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int k = (lane >> 2) + 16 * half, c = lane & 3;
                *reinterpret_cast<uint4*>(dh + obase + (size_t)k * 64 + c * 8) = *reinterpret_cast<const uint4*>(wave_lds + k * 80 + c * 16);
                if (SPLIT) *reinterpret_cast<uint4*>(dl + obase + (size_t)k * 64 + c * 8) = *reinterpret_cast<const uint4*>(wave_lds + 32 * 80 + k * 80 + c * 16);
            }
            return;
        }
    }
    // General path (sequence boundaries, M tail, pose rows, small-grid family): cos / sin pairs in two batches of 8 before their
    // stores (a table load issued behind a store waits for that store's acknowledgement; all 16 at once would push the 192x128
    // kernel past 128 VGPRs = one workgroup per CU; hoisting the per-tile bias loads as well and an LDS copy of the table: 0.0 %)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float2 cs8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = half * 8 + q;
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            int s, t; bool pose_row;
            qkv_row_token(p, row < p.M ? row : p.M - 1, s, t, pose_row);
            const int pos = qkv_rope_pos(p, t, pose_row, xpart);               // table row 0 == position -1 (pose token)
            cs8[q] = *reinterpret_cast<const float2*>(p.rope_tab + ((size_t)pos * 16 + (lane & 15)) * 2);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = half * 8 + q;
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            int s, t; bool pose_row;
            qkv_row_token(p, row < p.M ? row : p.M - 1, s, t, pose_row);
            const float2 cs = cs8[q];
            float v = acc[r] + bv;
            const float other = __shfl_xor(v, 16);
            v = (lane & 16) ? (v * cs.x + other * cs.y) : (v * cs.x - other * cs.y);
            if (col_ok && row < p.M) {
                const size_t o = ((size_t)(s * p.heads + head) * p.npad + t) * 64 + dcol;
                if (SPLIT) { f16 h, l; split_f16(v, h, l, ra); dh[o] = h; dl[o] = l; }
                else dh[o] = to_f16_sat(v, ra);
            }
        }
    }
}

// Epilogue of one 32x32 MFMA accumulator tile.  C/D layout of v_mfma_f32_32x32x16: this lane holds
// column `col` (= tile col + lane&31) and rows row0 + (r&3) + 8*(r>>2) + 4*(lane>>5), r = 0..15.
// The tile column origin is a multiple of 32 and (for EPI_QKV) segment/head boundaries are multiples
// of 64, so segment, head and the RoPE half (y for d<32, x for d>=32) are wave-uniform.
// mlim >= 0: rows >= mlim are not stored (halo-tiled convolutions: a 32-row MFMA tile = 32 pixels of ONE image row, the rest
// of the tile lies beyond the row's end); default: the GEMM's M.
#define QKV_LDS_BYTES (2 * 32 * 80)      // per-wave LDS scratch of the V^T transpose in epilogue_qkv_tile
#define EPI_LDS_BYTES 5120               // per-wave LDS scratch of the epilogues (>= QKV_LDS_BYTES, >= 32 x 144 for the plane tiles)
template <bool SPLIT, int EPI>
__device__ __forceinline__ void epilogue_tile(const GemmParams& p, const floatx16& acc, int row0, int col, int lane,
                                              int kslice = 0, int mlim = -1, char* wave_lds = nullptr) {
    const bool first_slice = kslice == 0;
    const int M_ = mlim >= 0 ? mlim : p.M;
    RangeAcc ra;                      // range report (sta_common.h): one flush per tile - for the plane epilogues of the DPT head
                                      // (no normalisation layers) and the generic EPI_F16; mlp.fc1's GELU tile (EPI_GELU) is a
                                      // function of a LayerNorm output and stays uncounted (its compares are dead code)
    if (EPI == EPI_QKV && p.ksplit <= 1) { epilogue_qkv_tile<SPLIT>(p, acc, row0, col, lane, wave_lds); return; }
    const int lhi = lane >> 5;
    const bool col_ok = col < p.N;
    const float bv = (p.bias != nullptr && col_ok && first_slice) ? p.bias[col] : 0.f;
    // Split-K partial tile (small-M regime) -> the fp32 slab of this K slice, straight-line: qkv_finish_kernel / splitk_finish_kernel /
    // resid_ln_kernel sum the slices.  (Left in the generic per-element loop below, the slab store sat behind ~40 scalar
    // instructions and three uniform branches PER ELEMENT - row remap division, activation / residual / format flags -: stamps
    // inside the kernel showed 4.5 us of epilogue in an 8-us workgroup for a 32-KB store that takes 1.1 us by itself,
    // tools/gemm_stamps.py, tools/probes/store_probe.hip.)
    if ((EPI == EPI_QKV || EPI == EPI_F16 || EPI == EPI_F32) && p.ksplit > 1 && (EPI != EPI_F32 || p.slab != nullptr)) {
        float* const o = (EPI == EPI_F32 ? p.slab : p.skbuf) + ((size_t)kslice * p.M + row0 + 4 * lhi) * p.N + col;
        const float b = EPI == EPI_F32 ? bv : 0.f;              // slice 0 of an in-place residual GEMM carries the bias; the finishers add theirs
        if (row0 + 32 <= p.M && __all(col_ok)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * p.N] = acc[r] + b;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int roff = (r & 3) + 8 * (r >> 2);
                if (col_ok && row0 + 4 * lhi + roff < p.M) o[(size_t)roff * p.N] = acc[r] + b;
            }
        }
        return;
    }
    if (EPI == EPI_F32R && row0 + 32 <= M_ && __all(col_ok)) {
        // interior tile of the in-place residual epilogue.  ALL 16 residual loads first, then the 16 stores: written as
        // `*c = v + *c` per element the compiler must keep load r+1 behind store r (it cannot prove ldc != 0), and on gfx9 loads
        // and stores retire through ONE in-order counter (vmcnt): the data of a load cannot be used before every store issued
        // ahead of it has been acknowledged - 16 dependent (store-ack, load) round trips per 32x32 tile: most of the 25-36 us
        // this epilogue used to expose per launch (tools/isa_serial_scan.py finds the pattern in the built code).
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = p.C32[(size_t)(row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * p.ldc + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) p.C32[(size_t)(row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * p.ldc + col] = (acc[r] + bv) + old[r];
        return;
    }
    if (EPI == EPI_GELU && SPLIT && wave_lds != nullptr && row0 + 32 <= M_ && __all(col_ok)) {
        // interior tile of mlp.fc1 through the wave's LDS scratch: a lane owns ONE column and 16 rows, so straight from the
        // accumulators the tile leaves as 32 store instructions of 2 B per lane.  Staged as [row][hi 64 B | lo 64 B] (exactly
        // one 128-B row block of the blocked plane layout) it leaves as 4 instructions of 16 B per lane: 8 whole lines each.
        constexpr int RS = 144;                               // LDS row stride (128 B + 16: 16-B aligned reads, no 2^k stride)
        const int c = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const float v = gelu_erf(acc[r] + bv);
            if (p.c_mx) {     // consumer = mlp.fc2 in the f16mx arithmetic (precision f16x3m): [hi 64 B | 32 byte pairs] - same row size
                f16 h; unsigned short pr; split_mx1<false>(v, ra, h, pr);
                *reinterpret_cast<f16*>(wave_lds + k * RS + c * 2) = h;
                *reinterpret_cast<unsigned short*>(wave_lds + k * RS + 64 + c * 2) = pr;
            } else {
                f16 h, l; split_f16(v, h, l, ra);
                *reinterpret_cast<f16*>(wave_lds + k * RS + c * 2) = h;
                *reinterpret_cast<f16*>(wave_lds + k * RS + 64 + c * 2) = l;
            }
        }
        const size_t o0 = blk_off<SPLIT>(row0, col - c, p.c_rp);       // first element of the tile's first row block (64 elements per row)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int k = (lane >> 3) + 8 * it, ch = lane & 7;
            *reinterpret_cast<uint4*>(p.C_hi + o0 + (size_t)k * 64 + ch * 8) = *reinterpret_cast<const uint4*>(wave_lds + k * RS + ch * 16);
        }
        return;
    }
    if (EPI == EPI_GELU && row0 + 32 <= M_ && __all(col_ok)) {
        // interior tile of the other hot epilogue (every tile when M, N are tile multiples, as at bench scale): no per-element
        // bounds predicate -> no exec-mask save / branch per element
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const float v = gelu_erf(acc[r] + bv);
            const size_t o = blk_off<SPLIT>(row, col, p.c_rp);
            if (SPLIT && p.c_mx) store_mx1<false>(p.C_hi, o, v, ra);
            else if (SPLIT) { f16 h, l; split_f16(v, h, l, ra); p.C_hi[o] = h; p.C_hi[o + 32] = l; }
            else p.C_hi[o] = to_f16_sat(v, ra);
        }
        return;
    }
    // ---- generic tile.  Written as PHASES over the 16 accumulator rows of the lane, with every wave-uniform flag (activation,
    // residual planes, output format) tested once per phase: left inside one per-element loop those flags compiled to three
    // uniform branches and ~40 scalar instructions around EVERY store (stamps: 4.5 - 7 us of epilogue per workgroup).
    // Everything the epilogue READS from global memory (residual stream, residual planes) is loaded before its first store:
    // interleaved, every load would wait for the acknowledgement of the store before it (in-order vmcnt, see above).
    if (EPI == EPI_F32R || EPI == EPI_F32) {
        const bool interior = row0 + 32 <= M_ && __all(col_ok);
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ok[r] = interior || (col_ok && row0 + 4 * lhi + (r & 3) + 8 * (r >> 2) < M_);
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r] + bv;
        float* const c = p.C32 + (size_t)(row0 + 4 * lhi) * p.ldc + col;
        if (EPI == EPI_F32 && p.ksplit > 1) {           // atomic split-K (forced tile families only: the product path uses slabs)
#pragma unroll
            for (int r = 0; r < 16; ++r) if (ok[r]) unsafeAtomicAdd(c + (size_t)((r & 3) + 8 * (r >> 2)) * p.ldc, v[r]);       // hardware global_atomic_add_f32
            return;
        }
        const float* const rs = EPI == EPI_F32R ? c : (p.resid ? p.resid + (size_t)(row0 + 4 * lhi) * p.ldr + col : nullptr);
        if (rs) {
            float pre[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pre[r] = ok[r] ? rs[(size_t)((r & 3) + 8 * (r >> 2)) * (EPI == EPI_F32R ? p.ldc : p.ldr)] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += pre[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) if (ok[r]) c[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = v[r];
        return;
    }
    // ---- plane epilogues (DPT head, mlp.fc1 edge tiles) and the ConvT scatter: one loop over the lane's 16 rows with the flags
    // made branch-free where that is free (ReLU as a floor that is -inf otherwise; absent residual planes read as 0) and the
    // output format tested once, outside the loop.  The callers put a scheduling barrier between a wave's tiles
    // (STA_EPI_TILE_FENCE): left free, hipcc overlaps the tiles' phases and the kernel needs 50+ more VGPRs.
    const bool interior = row0 + 32 <= M_ && __all(col_ok);
    if (EPI == EPI_F16 && SPLIT && wave_lds != nullptr && interior) {
        // Interior tile of a plane epilogue (DPT head) through the wave's LDS scratch (round 4).  A lane owns ONE column and 16
        // rows, so straight from the accumulators the tile left as 32 store instructions of 2 B per lane and every residual
        // plane arrived as 32 loads of 2 B per lane - in-kernel stamps: 33 - 44 us of epilogue per workgroup next to 118 - 126 us
        // of main loop in the 256-channel convolutions.  The tile's 32 rows are consecutive rows of the blocked planes: one
        // row block = [hi 64 B | lo / pair 64 B], i.e. the tile is ONE contiguous 4-KiB piece of the output and of each residual
        // plane.  Phase 1 (column domain): act(acc + bias) as fp32 into LDS [row][32].  Phase 2 (row domain, two tasks per lane =
        // 8 channels of one row each): 2 x ds_read_b128, the residual planes as 16-B loads, the adds, the split, 16-B stores.
        float* const L = reinterpret_cast<float*>(wave_lds);              // [32][32] fp32 = 4 KiB
        const float floor_ = p.act == ACT_RELU ? 0.f : -INFINITY;
        const int c = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = acc[r] + bv;
            if (p.act == ACT_GELU) x = gelu_erf(x);
            L[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 32 + c] = fmaxf(x, floor_);
        }
        // first element of the tile's first row block (64 elements per row); the tile's column origin is wave-uniform
        const size_t o0 = blk_off<true>(row0, __builtin_amdgcn_readfirstlane(col - c), p.c_rp);
        // every residual load of the tile before its first store (a load issued behind a store waits for that store's
        // acknowledgement); the two residual planes one after the other, so that only one of them occupies registers at a time
        float v[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int task = lane + 64 * t, row = task >> 2, q = task & 3;
            const float4 a = *reinterpret_cast<const float4*>(L + row * 32 + q * 8), b = *reinterpret_cast<const float4*>(L + row * 32 + q * 8 + 4);
            v[t][0] = a.x; v[t][1] = a.y; v[t][2] = a.z; v[t][3] = a.w; v[t][4] = b.x; v[t][5] = b.y; v[t][6] = b.z; v[t][7] = b.w;
        }
        auto add_plane = [&](const f16* R) {
            uint4 hh[2], ll[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int task = lane + 64 * t;
                const size_t o = o0 + (size_t)(task >> 2) * 64 + (task & 3) * 8;
                hh[t] = ldg16(R + o); ll[t] = ldg16(R + o + 32);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                H8 h8, l8; h8.u = hh[t]; l8.u = ll[t];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (p.r_mx) v[t][e] += (float)h8.e[e] + __builtin_amdgcn_cvt_f32_bf8(reinterpret_cast<const unsigned short*>(&l8)[e], 1) * (1.0f / (float)(1 << STA_MX_A_SLO));
                    else v[t][e] += (float)h8.e[e] + (float)l8.e[e];
                }
            }
        };
        if (p.R1_hi) add_plane(p.R1_hi);
        if (p.R2_hi) add_plane(p.R2_hi);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int task = lane + 64 * t;
            const size_t o = o0 + (size_t)(task >> 2) * 64 + (task & 3) * 8;
            if (p.c_mx) {
                const MX4 m0 = split_mx4<false>(v[t], ra), m1 = split_mx4<false>(v[t] + 4, ra);
                *reinterpret_cast<uint4*>(p.C_hi + o) = make_uint4(m0.hi.x, m0.hi.y, m1.hi.x, m1.hi.y);
                *reinterpret_cast<uint4*>(p.C_hi + o + 32) = make_uint4(m0.pairs.x, m0.pairs.y, m1.pairs.x, m1.pairs.y);
            } else {
                H8 oh, ol;
#pragma unroll
                for (int e = 0; e < 8; ++e) split_f16(v[t][e], oh.e[e], ol.e[e], ra);
                *reinterpret_cast<uint4*>(p.C_hi + o) = oh.u;
                *reinterpret_cast<uint4*>(p.C_hi + o + 32) = ol.u;
            }
        }
        ra.flush(p.range);
        return;
    }
    bool ok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ok[r] = interior || (col_ok && row0 + 4 * lhi + (r & 3) + 8 * (r >> 2) < M_);
    float pre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = 0.f;
    if (EPI == EPI_F16 && (p.R1_hi != nullptr || p.R2_hi != nullptr)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (!ok[r]) continue;
            const size_t o = blk_off<SPLIT>(row0 + 4 * lhi + (r & 3) + 8 * (r >> 2), col, p.c_rp);
            if (SPLIT && p.r_mx) {
                if (p.R1_hi) pre[r] = load_mx_act(p.R1_hi, o);
                if (p.R2_hi) pre[r] += load_mx_act(p.R2_hi, o);
            } else {
                if (p.R1_hi) pre[r] = (float)p.R1_hi[o] + (SPLIT ? (float)p.R1_hi[o + 32] : 0.f);
                if (p.R2_hi) pre[r] += (float)p.R2_hi[o] + (SPLIT ? (float)p.R2_hi[o + 32] : 0.f);
            }
        }
    }
    const float floor_ = (EPI != EPI_GELU && p.act == ACT_RELU) ? 0.f : -INFINITY;
    const bool gelu = EPI == EPI_GELU || p.act == ACT_GELU;
    auto value = [&](int r) {
        float x = acc[r] + bv;
        if (gelu) x = gelu_erf(x);            // (EPI_F16 with GELU: un-split mlp.fc1 of tiny test shapes only)
        return fmaxf(x, floor_) + pre[r];
    };
    if (EPI == EPI_GELU || EPI == EPI_F16) {
        if (SPLIT && p.c_mx) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (ok[r]) store_mx1<false>(p.C_hi, blk_off<SPLIT>(row0 + 4 * lhi + (r & 3) + 8 * (r >> 2), col, p.c_rp), value(r), ra);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!ok[r]) continue;
                const size_t o = blk_off<SPLIT>(row0 + 4 * lhi + (r & 3) + 8 * (r >> 2), col, p.c_rp);
                const float x = value(r);
                if (SPLIT) { f16 h, l; split_f16(x, h, l, ra); p.C_hi[o] = h; p.C_hi[o + 32] = l; }
                else p.C_hi[o] = to_f16_sat(x, ra);
            }
        }
    } else {  // EPI_CONVT: pixel scatter
        const int g = col / p.ct_cout, co = col - g * p.ct_cout;
        const int dy = g / p.ct_k, dx = g - dy * p.ct_k;
        const int hw = p.ct_h * p.ct_w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (!ok[r]) continue;
            const int row = row0 + 4 * lhi + (r & 3) + 8 * (r >> 2);
            const int img = row / hw, rem = row - img * hw;
            const int y = rem / p.ct_w, x = rem - y * p.ct_w;
            const size_t opix = ((size_t)img * (p.ct_h * p.ct_k) + (y * p.ct_k + dy)) * (p.ct_w * p.ct_k) + (x * p.ct_k + dx);
            const size_t o = blk_off<SPLIT>(opix, co, p.c_rp);
            const float v = acc[r] + bv;
            if (SPLIT && p.c_mx) store_mx1<false>(p.C_hi, o, v, ra);
            else if (SPLIT) { f16 h, l; split_f16(v, h, l, ra); p.C_hi[o] = h; p.C_hi[o + 32] = l; }
            else p.C_hi[o] = to_f16_sat(v, ra);
        }
    }
    if (EPI != EPI_GELU) ra.flush(p.range);
}
// between the tiles of one wave's epilogue: nothing moves across (keeps ONE tile's temporaries live; see epilogue_tile)
#define STA_EPI_TILE_FENCE() __builtin_amdgcn_sched_barrier(0)

// Second half of a split-K GEMM with the QKV epilogue (small-M regime): sums the K-slice slabs skbuf[s][M,N], adds the
// bias, rotates Q / K (RoPE pairs (d, d+16) sit in lanes l and l^16 of a wave: a wave is one 64-column head of one row) and
// writes head-major Q / K and V^T exactly like epilogue_qkv_tile.  Threads [0, M*(nq+nk)): one Q/K element each;
// threads after that: one V column x 4 consecutive rows each (one 8-byte run of V^T).
template <bool SPLIT>
__global__ __launch_bounds__(256) void qkv_finish_kernel(const GemmParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nqk = p.nq + p.nk;
    const int64_t n_qk = (int64_t)p.M * nqk;
    const int lane = threadIdx.x & 63;
    if (i < n_qk) {                                     // n_qk is a multiple of 64: the branch is wave-uniform
        const int row = (int)(i / nqk), c = (int)(i - (int64_t)row * nqk);
        float v = p.bias ? p.bias[c] : 0.f;
        for (int s = 0; s < p.ksplit; ++s) v += p.skbuf[((size_t)s * p.M + row) * p.N + c];
        const int seg = c >= p.nq ? 1 : 0, cc = c - (seg ? p.nq : 0);
        const int head = cc >> 6, dcol = cc & 63, xpart = (dcol >> 5) & 1;
        int s_, t; bool pose_row;
        qkv_row_token(p, row, s_, t, pose_row);
        const int pos = qkv_rope_pos(p, t, pose_row, xpart);
        const float2 cs = *reinterpret_cast<const float2*>(p.rope_tab + ((size_t)pos * 16 + (lane & 15)) * 2);
        const float other = __shfl_xor(v, 16);
        v = (lane & 16) ? (v * cs.x + other * cs.y) : (v * cs.x - other * cs.y);
        f16* const dh = seg == 0 ? p.Q_hi : p.K_hi;
        f16* const dl = seg == 0 ? p.Q_lo : p.K_lo;
        const size_t o = ((size_t)(s_ * p.heads + head) * p.npad + t) * 64 + dcol;
        if (SPLIT) { f16 h, l; split_f16(v, h, l, p.range); dh[o] = h; dl[o] = l; }
        else dh[o] = to_f16_sat(v, p.range);
        return;
    }
    const int64_t j = i - n_qk;
    const int m4 = (p.M + 3) >> 2;
    if (j >= (int64_t)m4 * p.nv) return;
    const int g = (int)(j / p.nv), c = (int)(j - (int64_t)g * p.nv), col = nqk + c;
    const int head = c >> 6, dcol = c & 63;
    const float bv = p.bias ? p.bias[col] : 0.f;
    f16 hh[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = g * 4 + e < p.M ? g * 4 + e : p.M - 1;
        float v = bv;
        for (int s = 0; s < p.ksplit; ++s) v += p.skbuf[((size_t)s * p.M + row) * p.N + col];
        if (SPLIT) split_f16(v, hh[e], ll[e], p.range); else { hh[e] = to_f16_sat(v, p.range); ll[e] = (f16)0; }
    }
    const int row0 = g * 4;
    int s0, t0; bool pr0;
    qkv_row_token(p, row0 < p.M ? row0 : p.M - 1, s0, t0, pr0);
    if (row0 + 3 < p.M && t0 + 3 < p.ntok) {
        const size_t o = ((size_t)(s0 * p.heads + head) * 64 + dcol) * p.npad + t0;
        H4 ph, pl;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ph.e[e] = hh[e]; pl.e[e] = ll[e]; }
        store8_a2(p.Vt_hi + o, ph.u);
        if (SPLIT) store8_a2(p.Vt_lo + o, pl.u);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = row0 + e;
            if (row < p.M) {
                int s2, t2; bool pr2;
                qkv_row_token(p, row, s2, t2, pr2);
                const size_t o2 = ((size_t)(s2 * p.heads + head) * 64 + dcol) * p.npad + t2;
                p.Vt_hi[o2] = hh[e];
                if (SPLIT) p.Vt_lo[o2] = ll[e];
            }
        }
    }
}

template <bool SPLIT, int AMODE, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int STAGE = 2 * NPL * GEMM_TILE_BYTES;     // A planes then B planes
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- tile coordinates: N tiles fastest inside a group of 8 M tiles so that blocks that are
    // co-resident share A rows / W rows in L2 (XCD-level locality comes from the group size).
    const int tiles_n = (p.N + GEMM_BN - 1) / GEMM_BN;
    const int bid = blockIdx.x;
    const int bm = bid / tiles_n, bn = bid % tiles_n;
    const int m0 = bm * GEMM_BM, n0 = bn * GEMM_BN;

    // ---- per-thread staging assignment: 2 chunks (16 B) of A and of B per plane per K tile
    int a_row[2], a_kc[2];
    const f16* a_ptr_hi[2]; const f16* a_ptr_lo[2];
    const f16* b_ptr_hi[2]; const f16* b_ptr_lo[2];
    int cv_img[2], cv_y[2], cv_x[2]; bool cv_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int c = tid + 256 * i;
        int row = c >> 2, kc = c & 3;
        a_row[i] = row; a_kc[i] = kc;
        int gm = m0 + row;
        if (AMODE == A_DENSE) {
            int gmc = gm < p.M ? gm : p.M - 1;
            a_ptr_hi[i] = p.A_hi + (size_t)gmc * (SPLIT ? 64 : 32) + kc * 8;
            a_ptr_lo[i] = SPLIT ? a_ptr_hi[i] + 32 : nullptr;
        } else {
            cv_ok[i] = gm < p.M;
            int gmc = cv_ok[i] ? gm : 0;
            int hw = p.Ho * p.Wo;
            cv_img[i] = gmc / hw;
            int rem = gmc - cv_img[i] * hw;
            cv_y[i] = (rem / p.Wo) * p.cstride - 1;
            cv_x[i] = (rem % p.Wo) * p.cstride - 1;
            a_ptr_hi[i] = nullptr; a_ptr_lo[i] = nullptr;
        }
        int gn = n0 + row;
        int gnc = gn < p.N ? gn : p.N - 1;
        b_ptr_hi[i] = p.B_hi + (size_t)gnc * 64 + kc * 8;          // weights: [K/32][N][hi32|lo32]
        b_ptr_lo[i] = SPLIT ? b_ptr_hi[i] + 32 : nullptr;
    }

    uint4 ra_hi[2], ra_lo[2], rb_hi[2], rb_lo[2];
    const int nkt = p.K / GEMM_BK;

    const size_t a_kstride = (size_t)p.a_rp * (SPLIT ? 64 : 32), b_kstride = (size_t)p.N * 64;
    auto load_tile = [&](int kt) {
        const int k0 = kt * GEMM_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (AMODE == A_DENSE) {
                ra_hi[i] = ldg16(a_ptr_hi[i] + kt * a_kstride);
                if (SPLIT) ra_lo[i] = ldg16(a_ptr_lo[i] + kt * a_kstride);
            } else {
                int tap = k0 / p.Cin;
                int c0 = k0 - tap * p.Cin;
                int ky = tap / 3, kx = tap - ky * 3;
                int yi = cv_y[i] + ky, xi = cv_x[i] + kx;
                bool ok = cv_ok[i] && yi >= 0 && yi < p.Hi && xi >= 0 && xi < p.Wi;
                uint4 z = make_uint4(0, 0, 0, 0);
                ra_hi[i] = z; if (SPLIT) ra_lo[i] = z;
                if (ok) {
                    size_t off = ((size_t)(c0 >> 5) * p.a_rp + (size_t)(cv_img[i] * p.Hi + yi) * p.Wi + xi) * (SPLIT ? 64 : 32) + a_kc[i] * 8;
                    ra_hi[i] = ldg16(p.A_hi + off);
                    if (SPLIT) ra_lo[i] = ldg16(p.A_hi + off + 32);
                    if (p.relu_in) {
                        uint4 lo = SPLIT ? ra_lo[i] : z;
                        ra_hi[i] = relu_pair_hi(ra_hi[i], lo);
                        if (SPLIT) ra_lo[i] = lo;
                    }
                }
            }
            rb_hi[i] = ldg16(b_ptr_hi[i] + kt * b_kstride);
            if (SPLIT) rb_lo[i] = ldg16(b_ptr_lo[i] + kt * b_kstride);
        }
    };
    auto store_tile = [&](int stage) {
        char* sA = smem + stage * STAGE;
        char* sB = sA + NPL * GEMM_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int off = lds_off(a_row[i], a_kc[i]);
            *reinterpret_cast<uint4*>(sA + off) = ra_hi[i];
            if (SPLIT) *reinterpret_cast<uint4*>(sA + GEMM_TILE_BYTES + off) = ra_lo[i];
            *reinterpret_cast<uint4*>(sB + off) = rb_hi[i];
            if (SPLIT) *reinterpret_cast<uint4*>(sB + GEMM_TILE_BYTES + off) = rb_lo[i];
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + NPL * GEMM_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 a_hi[2], a_lo[2], b_hi[2], b_lo[2];
            const int chunk = ks * 2 + lhi;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                int ra = wm * 64 + t * 32 + l31;
                int rb = wn * 64 + t * 32 + l31;
                a_hi[t] = *reinterpret_cast<const half8*>(sA + lds_off(ra, chunk));
                b_hi[t] = *reinterpret_cast<const half8*>(sB + lds_off(rb, chunk));
                if (SPLIT) {
                    a_lo[t] = *reinterpret_cast<const half8*>(sA + GEMM_TILE_BYTES + lds_off(ra, chunk));
                    b_lo[t] = *reinterpret_cast<const half8*>(sB + GEMM_TILE_BYTES + lds_off(rb, chunk));
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (SPLIT) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[i], b_hi[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[i], b_lo[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            { epilogue_tile<SPLIT, EPI>(p, acc[i][j], m0 + wm * 64 + i * 32, n0 + wn * 64 + j * 32 + l31, lane); STA_EPI_TILE_FENCE(); }
}
