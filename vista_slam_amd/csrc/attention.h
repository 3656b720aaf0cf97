// Flash-style fused attention for gfx950, head_dim 64, fp16 MFMA / fp32 softmax.
//
// Replaces xformers memory_efficient_attention (sta_blocks.py:143) and the naive
// softmax(QK^T*scale)V of CrossAttention (sta_blocks.py:201-205) - the N'xN' score matrix is
// never materialised.
//
// Layout (written by the QKV GEMM epilogue): Q,K [S, heads, npad, 64] fp16 planes (RoPE already
// applied), V^T [S, heads, 64, npad] fp16 planes.  Output O planes [S*nq, ldo], col = head*64+d.
//
// One block = 4 waves = 128 query rows of one (sequence, head); each wave owns 32 queries.
// The score tile is computed TRANSPOSED (S^T = K Q^T, O^T = V^T P^T) so that every lane owns one
// query column: row max / row sum are in-lane reductions plus one lane^32 exchange, the online
// softmax rescale is a per-lane scalar, and the P^T B-operand of the second MFMA is built from the
// accumulator registers with one v_permlane32_swap per packed register pair (the accumulator layout
// interleaves 4-key groups between the two lane halves; the swap restores the natural k order, so the
// V^T A-operand is a single contiguous 16-B fragment read, exactly like K).
// K / V^T tiles (64 keys, hi + lo planes = 32 KiB) are double-buffered in LDS and filled by LDS-DMA
// (global_load_lds_dwordx4, source-side XOR swizzle): tile i+1 streams in while tile i is computed, no
// staging registers and no ds_write.  Measured (MI355X, f16x3, 16 x 16 heads x 768^2): 170 -> 140 us per
// launch vs the register-staged / two-ds_read_b64 version (VALU instructions per tile 660 -> 310).
// Cross attention = same kernel with kv_shift selecting the other view's K/V.
#pragma once
#include "sta_common.h"
#include <type_traits>

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_base);   // gemm2.h

struct AttnParams {
    const f16* Q_hi; const f16* Q_lo; const f16* K_hi; const f16* K_lo; const f16* Vt_hi; const f16* Vt_lo;
    f16* O_hi; f16* O_lo; int ldo;
    int S, heads, nq, nk, npad;
    int kv_shift;            // K/V come from sequence (s + kv_shift) % S
    float scale_log2e;       // head_dim^-0.5 * log2(e)
    unsigned long long* range;   // the handle's range counters (sta_common.h)
    int prefetch;            // 1: nk <= 4 key tiles and the launch carries 4 LDS stages - ALL K / V^T tiles are requested up front (small grids)
    int pose;                // decoder: token index nk (== nq) of Q / K / V^T is the pose token.  As a KEY it is folded into the
                             // initial online-softmax state of every query (no 13th key tile for one key); as a QUERY it is served by
                             // the pose blocks (pose == 1: one wave per (sequence, head), plain fp32 dot products; nq % 128 == 0) or
                             // rides as query nq in the last query block's spare rows (pose == 2); its output row is S*nq + s
};

#define ATT_KV 64
#define ATT_TILE_BYTES (64 * 128)   // 64 rows x 64 fp16

template <bool SPLIT>
constexpr int attn_smem_bytes(int stages = 2) { return stages * (SPLIT ? 4 : 2) * ATT_TILE_BYTES; }
#define ATT_PREFETCH_TILES 4

// The pose-token query of one (sequence, head) per workgroup: 1 x (nk + 1) scores, softmax and 1 x 64 output as fp32 dot products
// (an MFMA tile would carry 31 dead queries through every key tile).  Phase 1: thread = key (K rows are 128 contiguous bytes),
// phase 2: block-wide max / sum, probabilities parked in LDS, phase 3: lane = d (V^T rows are contiguous along the keys).
template <bool SPLIT>
__device__ __forceinline__ void attn_pose_query(const AttnParams& p, char* smem) {
    // One workgroup per (sequence, head); its four waves split the keys (a single wave walking all 769 keys was a 58-us
    // dependent chain - longer than the whole kernel at B <= 4).  Phase 1: thread = key; phase 2: block-wide max / sum;
    // phase 3: lane = d, wave w sums over its quarter of the keys; wave partials meet in LDS.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sh = blockIdx.x;
    const int s = sh / p.heads, h = sh - s * p.heads;
    const int skv = (s + p.kv_shift) % p.S;
    const size_t qoff = ((size_t)(s * p.heads + h) * p.npad + p.nq) * 64;
    const size_t koff = (size_t)(skv * p.heads + h) * p.npad * 64;
    const size_t voff = (size_t)(skv * p.heads + h) * 64 * p.npad;
    float* pl = reinterpret_cast<float*>(smem);               // [npad] scores, then probabilities
    float* red = pl + p.npad;                                 // [4] wave maxima, [4] wave sums, [4][64] partial outputs
    float q[64];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        H8 a, b; a.u = ldg16(p.Q_hi + qoff + c * 8);
        if (SPLIT) b.u = ldg16(p.Q_lo + qoff + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[c * 8 + e] = (float)a.e[e] + (SPLIT ? (float)b.e[e] : 0.f);
    }
    const int nkeys = p.nk + 1;
    float mx = -INFINITY;
    for (int j = tid; j < p.npad; j += 256) {
        float t = -INFINITY;
        if (j < nkeys) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                H8 a, b; a.u = ldg16(p.K_hi + koff + (size_t)j * 64 + c * 8);
                if (SPLIT) b.u = ldg16(p.K_lo + koff + (size_t)j * 64 + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(q[c * 8 + e], (float)a.e[e] + (SPLIT ? (float)b.e[e] : 0.f), acc);
            }
            t = acc * p.scale_log2e;
        }
        pl[j] = t;
        mx = fmaxf(mx, t);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < p.npad; j += 256) {
        const float e = __builtin_amdgcn_exp2f(pl[j] - mx);       // exp2(-inf) = 0 for the padding
        pl[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    // wave w: keys [w * npad / 4, (w + 1) * npad / 4) (npad % 64 == 0: whole 16-key groups), two 8-key chunks in flight
    const f16* vh = p.Vt_hi + voff + (size_t)lane * p.npad;
    const f16* vl = SPLIT ? p.Vt_lo + voff + (size_t)lane * p.npad : nullptr;
    const int k0 = wave * (p.npad >> 2), k1 = k0 + (p.npad >> 2);
    float o = 0.f;
    for (int c0 = k0; c0 < k1; c0 += 16) {                         // V^T columns >= nkeys are zero (memset) and their p is 0
        H8 a[2], b[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { a[u].u = ldg16(vh + c0 + u * 8); if (SPLIT) b[u].u = ldg16(vl + c0 + u * 8); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = c0 + u * 8;
            const float4 p0 = *reinterpret_cast<const float4*>(pl + c), p1 = *reinterpret_cast<const float4*>(pl + c + 4);
            const float pe[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) o = __builtin_fmaf(pe[e], (float)a[u].e[e] + (SPLIT ? (float)b[u].e[e] : 0.f), o);
        }
    }
    red[8 + wave * 64 + lane] = o;
    __syncthreads();
    if (wave != 0) return;
    o = (red[8 + lane] + red[8 + 64 + lane]) + (red[8 + 128 + lane] + red[8 + 192 + lane]);
    o /= (red[4] + red[5]) + (red[6] + red[7]);
    const int64_t orow = (int64_t)p.S * p.nq + s, orows = (int64_t)p.S * p.nq + p.S;
    const size_t oo = blk_off<SPLIT>(orow, h * 64 + lane, orows);
    if (SPLIT) { f16 hh, ll; split_f16(o, hh, ll, p.range); p.O_hi[oo] = hh; p.O_hi[oo + 32] = ll; }
    else p.O_hi[oo] = to_f16_sat(o, p.range);
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int STAGE = 2 * NPL * ATT_TILE_BYTES;   // K planes then V^T planes
    // pose blocks first in the grid (short: they end while the first round of query blocks is still running)
    const int npose_blocks = p.pose == 1 ? p.S * p.heads : 0;
    const int nqe = p.nq + (p.pose == 2 ? 1 : 0);       // pose == 2: the pose query rides in the last query block's spare rows
    if ((int)blockIdx.x < npose_blocks) { attn_pose_query<SPLIT>(p, smem); return; }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    // 1-D grid, XCD-aware: block b runs on XCD b%8 (observed dispatch); give each XCD a contiguous range
    // of logical ids so the query blocks of one (sequence, head) share that XCD's L2 copy of K/V.
    const int nqb = (nqe + 127) / 128;
    const int nwg = nqb * p.heads * p.S;
    int logical;
    {
        const int bid = blockIdx.x - npose_blocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int qb = logical % nqb;
    const int h = (logical / nqb) % p.heads, s = logical / (nqb * p.heads);
    const int skv = (s + p.kv_shift) % p.S;
    const int q0 = qb * 128 + wave * 32;

    const size_t qoff = (size_t)(s * p.heads + h) * p.npad * 64;
    const size_t koff = (size_t)(skv * p.heads + h) * p.npad * 64;
    const size_t voff = (size_t)(skv * p.heads + h) * 64 * p.npad;

    // ---- Q fragments (B operand: col = query, 8 consecutive d per lane-half per k-step)
    half8 qf_hi[4], qf_lo[4];
    {
        int qrow = q0 + l31; if (qrow > nqe - 1) qrow = nqe - 1;
        const f16* qp = p.Q_hi + qoff + (size_t)qrow * 64 + lhi * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { H8 t; t.u = ldg16(qp + kk * 16); qf_hi[kk] = t.h; }
        if (SPLIT) {
            const f16* ql = p.Q_lo + qoff + (size_t)qrow * 64 + lhi * 8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { H8 t; t.u = ldg16(ql + kk * 16); qf_lo[kk] = t.h; }
        }
    }

    // ---- K / V^T tiles go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write).
    // A plane tile = 64 rows x 128 B = 8 slots of 1 KiB (8 rows); wave w moves slots 2w, 2w+1 of every plane tile.
    // The DMA writes lane-linear (lane l -> row l>>3, chunk l&7 of its slot); the bank swizzle chunk ^ (row>>1)&7
    // is applied to the SOURCE chunk each lane fetches (same involution the fragment reads apply).
    int ksrc_l[2], vsrc_l[2];                   // per-lane element offsets inside a K tile / V^T tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (2 * wave + i) * 8 + (lane >> 3);
        const int sch = (lane & 7) ^ ((row >> 1) & 7);
        ksrc_l[i] = row * 64 + sch * 8;         // K row = key (64 d = 128 B)
        vsrc_l[i] = row * p.npad + sch * 8;     // V^T row = d, columns = keys
    }
    auto issue_tile = [&](int stage, int kv0) {
        char* sK = smem + stage * STAGE + (2 * wave) * 1024;
        char* sV = sK + NPL * ATT_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t ko = koff + (size_t)kv0 * 64 + ksrc_l[i];
            const size_t vo = voff + (size_t)kv0 + vsrc_l[i];
            glds16(p.K_hi + ko, sK + i * 1024);
            glds16(p.Vt_hi + vo, sV + i * 1024);
            if (SPLIT) {
                glds16(p.K_lo + ko, sK + ATT_TILE_BYTES + i * 1024);
                glds16(p.Vt_lo + vo, sV + ATT_TILE_BYTES + i * 1024);
            }
        }
    };

    floatx16 oacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;       // running max in scaled (log2) units
    const bool wave_active = q0 < nqe;         // decoder: 769 = 6 x 128 + 1 queries -> the last block has one live wave

    // ---- the pose token as a key (index nk): initial online-softmax state m = s_p, l = 1, O = v_p in fp32
    if (p.pose && wave_active) {
        float sp = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            H8 a, b; a.u = ldg16(p.K_hi + koff + (size_t)p.nk * 64 + kk * 16 + lhi * 8);
            if (SPLIT) b.u = ldg16(p.K_lo + koff + (size_t)p.nk * 64 + kk * 16 + lhi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float kv = (float)a.e[e] + (SPLIT ? (float)b.e[e] : 0.f);
                const float qv = (float)qf_hi[kk][e] + (SPLIT ? (float)qf_lo[kk][e] : 0.f);
                sp = __builtin_fmaf(qv, kv, sp);
            }
        }
        sp += __shfl_xor(sp, 32);
        m_run = sp * p.scale_log2e;
        l_run = lhi == 0 ? 1.f : 0.f;              // the two lane halves of a query add their l at the end
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t o = voff + (size_t)(d * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * p.npad + p.nk;
                oacc[d][r] = (float)p.Vt_hi[o] + (SPLIT ? (float)p.Vt_lo[o] : 0.f);
            }
    }

    // per-lane fragment byte offset inside a plane tile: row l31 (+32 rows = +4096 B: same swizzle since 16 & 7 == 0),
    // chunk c = 2*kk + lhi -> (c ^ swz) << 4  (identical for K rows = keys and V^T rows = d)
    int foff_l[4];
    {
        const int swz = (l31 >> 1) & 7;
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) foff_l[c2] = l31 * 128 + (((c2 * 2 + lhi) ^ swz) << 4);
    }

    // one KV tile held in LDS stage `stage_c` (compile-time: the stage / t / d displacements fold into the ds_read
    // offset field); TAIL = last, partly valid tile (keys >= nk masked to -inf)
    auto tile_body = [&](auto stage_c, auto tail_c, int kv0, int stage_rt = 0) {
        constexpr int STG = decltype(stage_c)::value;
        constexpr bool TAIL = decltype(tail_c)::value;
        const char* sK = smem + STG * STAGE + stage_rt * STAGE;     // stage_rt: the prefetch-all schedule's runtime stage (0 in the double-buffered loop)
        const char* sV = sK + NPL * ATT_TILE_BYTES;

        // ---- S^T = K Q^T  (rows = keys, cols = queries)
        floatx16 sacc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[t][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                half8 kf = *reinterpret_cast<const half8*>(sK + t * 4096 + foff_l[kk]);
                if (SPLIT) {
                    half8 kl = *reinterpret_cast<const half8*>(sK + ATT_TILE_BYTES + t * 4096 + foff_l[kk]);
                    sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qf_hi[kk], sacc[t], 0, 0, 0);
                    sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf_lo[kk], sacc[t], 0, 0, 0);
                }
                sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf_hi[kk], sacc[t], 0, 0, 0);
            }
        }

        // ---- online softmax (per lane = one query; this lane holds 32 of the tile's 64 keys:
        //      register r of sacc[t] <-> key t*32 + (r&3) + 8*(r>>2) + 4*lhi)
        if (TAIL) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (key >= p.nk) sacc[t][r] = -INFINITY;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx * p.scale_log2e);          // scale > 0: max commutes with it
        // raw v_exp_f32: arguments are <= 0 (or -inf), results in [0,1]; no denormal/overflow handling needed
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        const float neg_m = -m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[t][r], p.scale_log2e, neg_m));
                sacc[t][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {     // running max unchanged for every query of the wave: nothing to rescale
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }

        // ---- O^T += V^T P^T.  The accumulator layout gives lane-half lhi the keys {8m + 4lhi .. +3}; one
        // v_permlane32_swap per packed register pair regroups them so that half 0 holds the 8 consecutive keys
        // 16w .. 16w+7 and half 1 holds 16w+8 .. 16w+15 = the natural k order of the MFMA, i.e. the V^T A-operand is
        // one contiguous 16-B fragment read (same swizzle as K).
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                union { half8 h; unsigned u[4]; } ph, pl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = sacc[t][8 * w + j];
                    const f16 hh = (f16)pv;
                    ph.h[j] = hh;
                    if (SPLIT) pl.h[j] = (f16)(pv - (float)hh);
                }
                // u[0..1] = X (registers 8w..8w+3), u[2..3] = Y (8w+4..8w+7): swap X.high-lanes <-> Y.low-lanes
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    auto r1 = __builtin_amdgcn_permlane32_swap(ph.u[e], ph.u[2 + e], false, false);
                    ph.u[e] = r1[0]; ph.u[2 + e] = r1[1];
                    if (SPLIT) {
                        auto r2 = __builtin_amdgcn_permlane32_swap(pl.u[e], pl.u[2 + e], false, false);
                        pl.u[e] = r2[0]; pl.u[2 + e] = r2[1];
                    }
                }
                // after the swap: half 0 = [own X, partner X] = keys 16w+0..7; half 1 = [partner Y, own Y] = 16w+8..15,
                // in registers (u0,u1 | u2,u3) = k slots (0..3 | 4..7)
                const int c2 = t * 2 + w;            // 16-key group of the tile -> chunks 2*c2 + lhi
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    half8 vf = *reinterpret_cast<const half8*>(sV + d * 4096 + foff_l[c2]);
                    if (SPLIT) {
                        half8 vl = *reinterpret_cast<const half8*>(sV + ATT_TILE_BYTES + d * 4096 + foff_l[c2]);
                        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph.h, oacc[d], 0, 0, 0);
                        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pl.h, oacc[d], 0, 0, 0);
                    }
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, ph.h, oacc[d], 0, 0, 0);
                }
            }
    };
    using std::integral_constant;
    // Full tiles run in a loop whose body has ONE version of the tile code per stage; the partly valid last tile (decoder:
    // 769 = 12 x 64 + 1 keys) is peeled.  (With the tail test inside the loop hipcc merged the two bodies' O accumulators
    // through 16 v_mov_b64 per tile.)
    auto step = [&](auto stage_c, auto tail_c, int it, int ntiles_) {
        constexpr int STG = decltype(stage_c)::value;
        const int kv0 = it * ATT_KV;
        if (it + 1 < ntiles_) issue_tile(STG ^ 1, kv0 + ATT_KV);      // stage STG^1 was released by the last barrier
        if (wave_active) tile_body(stage_c, tail_c, kv0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    const int ntiles = (p.nk + ATT_KV - 1) / ATT_KV, nfull = p.nk / ATT_KV;
    using F = integral_constant<bool, false>; using T = integral_constant<bool, true>;
    if (p.prefetch) {
        // Small grids (SLAM scale: 196 keys = 4 tiles, <= 256 workgroups): one tile's arithmetic (~0.7 us) is shorter than the
        // latency of the next tile's DMA (~2 us), so the double-buffered loop below waits ~1.3 us per tile on a chain of four.
        // With 4 LDS stages every tile is requested up front: one latency, then four tiles of arithmetic.  The DMAs complete in
        // order, so tile t has landed once at most (ntiles - 1 - t) x DMAS instructions are outstanding.
        constexpr int DMAS = 4 * NPL;                   // global_load_lds instructions per tile and wave
        for (int t = 0; t < ntiles; ++t) issue_tile(t, t * ATT_KV);
        for (int t = 0; t < ntiles; ++t) {
            switch (ntiles - 1 - t) {
                case 3: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * DMAS) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * DMAS) : "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 * DMAS) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
            __syncthreads();
            if (wave_active) {
                if (t >= nfull) tile_body(integral_constant<int, 0>{}, T{}, t * ATT_KV, t);
                else tile_body(integral_constant<int, 0>{}, F{}, t * ATT_KV, t);
            }
        }
    } else {
        issue_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int it = 0;
        for (; it + 1 < nfull; it += 2) {
            step(integral_constant<int, 0>{}, F{}, it, ntiles);
            step(integral_constant<int, 1>{}, F{}, it + 1, ntiles);
        }
        if (it < nfull) { step(integral_constant<int, 0>{}, F{}, it, ntiles); ++it; }
        if (it < ntiles) {                                                  // the tail tile, in whichever stage it landed
            if (it & 1) step(integral_constant<int, 1>{}, T{}, it, ntiles);
            else step(integral_constant<int, 0>{}, T{}, it, ntiles);
        }
    }

    // ---- normalise and store: lane owns query q, d = dt*32 + (r&3) + 8*(r>>2) + 4*lhi.
    // Straight from the registers a store instruction would scatter 64 x 8 B over 32 rows; the wave stages its 32 queries x 64 d
    // as [query][column block][hi 64 B | lo 64 B] (= the row blocks of the output planes) in LDS and writes whole 128-B lines.
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int64_t orows = (int64_t)p.S * p.nq + (p.pose ? p.S : 0);
    if constexpr (SPLIT) {
        // LDS image per wave: 64 rows of 128 B, row R = d * 32 + query = [hi 4 chunks | lo 4 chunks] of 16 B (8 d each), chunk c of
        // row R stored at position c ^ (R & 7).  Round 6: the first form ([query][d block] rows of 144 B, ds_write_b64 straight from
        // the accumulator registers) put lanes l, l+4, l+8, l+12 of every 16-lane store group on one bank pair (288-B query stride
        // = 8 banks mod 32: 4-way) and two to three lanes of a b128 read group on one slot - ALL of the kernel's bank conflicts
        // (SQ_LDS_BANK_CONFLICT = 13 % of its LDS cycles, profiles/r05_lds_util_summary.txt; the K / V^T fragment reads are
        // conflict-free by the swizzle above).  Now one v_permlane32_swap per register pair first gives every lane a WHOLE 16-B
        // chunk (the same exchange the P^T operand uses: half 0 gets d 16j .. 16j+7, half 1 d 16j+8 .. 16j+15), so the tile is
        // written with ds_write_b128 - 8 consecutive lanes = 8 consecutive queries = 8 different positions of the same chunk:
        // one 128-B bank row per 8-lane group - and read back with ds_read_b128 whose four 16-lane groups each cover the 16
        // slots of a 256-B bank row exactly once (rows of equal parity differ in their chunk set, the XOR permutes inside it).
        __syncthreads();                                // every wave is done with the K / V^T stages (all waves reach this point)
        char* const wl = smem + wave * (64 * 128);
        RangeAcc ra;        // never flushed (dead code): a convex combination of V rows stays inside V's range
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int rowb = (d * 32 + l31) * 128, sw = l31 & 7;
#pragma unroll
            for (int j = 0; j < 2; ++j) {               // register groups g = 2j, 2j+1 -> chunks 2j (lane half 0), 2j+1 (half 1)
                union { uint4 u4; unsigned u[4]; } ch, cl;
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    H4 oh, ol;
#pragma unroll
                    for (int e = 0; e < 4; ++e) split_f16(oacc[d][(2 * j + gg) * 4 + e] * inv, oh.e[e], ol.e[e], ra);
                    ch.u[2 * gg] = oh.u.x; ch.u[2 * gg + 1] = oh.u.y; cl.u[2 * gg] = ol.u.x; cl.u[2 * gg + 1] = ol.u.y;
                }
                // u[0..1] = X (group 2j: d 16j + 4 lhi + 0..3), u[2..3] = Y (group 2j+1: d 16j + 8 + 4 lhi + 0..3); after the swap
                // half 0 = [own X, partner X] = d 16j .. 16j+7, half 1 = [partner Y, own Y] = d 16j+8 .. 16j+15
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    auto r1 = __builtin_amdgcn_permlane32_swap(ch.u[e], ch.u[2 + e], false, false);
                    ch.u[e] = r1[0]; ch.u[2 + e] = r1[1];
                    auto r2 = __builtin_amdgcn_permlane32_swap(cl.u[e], cl.u[2 + e], false, false);
                    cl.u[e] = r2[0]; cl.u[2 + e] = r2[1];
                }
                const int c = 2 * j + lhi;
                *reinterpret_cast<uint4*>(wl + rowb + ((c ^ sw) << 4)) = ch.u4;
                *reinterpret_cast<uint4*>(wl + rowb + (((4 + c) ^ sw) << 4)) = cl.u4;
            }
        }
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int qq = (lane >> 3) + 8 * it, ch = lane & 7;
                const int q = q0 + qq;
                if (q < nqe) {
                    const int64_t orow = q < p.nq ? (int64_t)s * p.nq + q : (int64_t)p.S * p.nq + s;
                    const size_t o = blk_off<true>(orow, h * 64 + d * 32, orows);
                    *reinterpret_cast<uint4*>(p.O_hi + o + ch * 8) = *reinterpret_cast<const uint4*>(wl + (d * 32 + qq) * 128 + ((ch ^ (qq & 7)) << 4));
                }
            }
    } else {
        const int q = q0 + l31;
        if (q < nqe) {
            RangeAcc ra;
            const int64_t orow = q < p.nq ? (int64_t)s * p.nq + q : (int64_t)p.S * p.nq + s;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dcol = d * 32 + 8 * g + 4 * lhi;
                    const size_t o = blk_off<false>(orow, h * 64 + dcol, orows);    // blocked planes [ldo/32][S*nq][32]
                    H4 oh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) oh.e[e] = to_f16_sat(oacc[d][g * 4 + e] * inv, ra);
                    *reinterpret_cast<uint2*>(p.O_hi + o) = oh.u;
                }
        }
    }
}
