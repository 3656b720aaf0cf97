// Throughput-scale flash attention for gfx950 (head_dim 64, f16 / f16x3 MFMA, fp32 softmax): the software-pipelined form of
// attention.h (same math, same operand layout, same AttnParams; attention.h keeps serving small grids).
//
// attention.h runs S^T MFMAs -> softmax VALU -> PV MFMAs strictly in sequence in every wave; its component ablation showed
// that the parts ADD UP (MFMA-busy 0.36 at full clock): the two waves of a SIMD fall into phase, so their matrix and vector
// phases collide instead of interleaving.  Here every wave overlaps them itself (two score tiles live):
//
//     per 64-key tile i:    sync | DMA tile i+2 |  S^T(i+1) MFMAs  ||  softmax(i) VALU  |  PV(i) MFMAs || P(i) fp16 hi/lo + permlane
//
// Both halves of the tile body are 24 MFMAs (x3 products) plus 16 fragment reads plus ~100-170 VALU that do not depend on
// them, and a sched_group_barrier pattern asks hipcc to issue them interleaved (one MFMA, a read, a handful of VALU) instead
// of clustered.  The body is ONE basic block: unconditional O rescale, the DMA of a clamped tile index and a discarded
// S^T(ntiles) instead of branches; the partly valid last tile is peeled.
// * workgroup = 8 waves x 32 queries = 256 queries of one (sequence, head); three 32-KiB K / V^T stages (96 KiB -> one
//   workgroup per CU, two waves per SIMD); tile i+2 is DMA'd right after the one barrier of tile i: a DMA round trip has a
//   whole tile of compute to land, and K / V^T stream through L2 -> LDS once per 256 queries (attention.h: per 128).
// * S^T = K Q^T (lane = query): row max / sum in-lane + one v_permlane32_swap exchange of the lane halves; P^T fragments
//   from the accumulator registers (packed cvt, v_fma_mix for the fp16 residual, v_permlane32_swap regrouping); keys >= nk
//   masked in the last tile; the pose token (AttnParams::pose) is folded into the initial state as a key and served by the
//   pose blocks (attention.h) as a query.
// Replaces xformers memory_efficient_attention / CrossAttention's softmax(QK^T)V (sta_blocks.py:143,201-205) at throughput
// scale (>= 256 workgroups of 256 queries).
#pragma once
#include "attention.h"

#define ATT2_STAGES 3
template <bool SPLIT>
constexpr int attn2_smem_bytes() { return ATT2_STAGES * 2 * (SPLIT ? 2 : 1) * ATT_TILE_BYTES; }

typedef float float2v __attribute__((ext_vector_type(2)));
typedef f16 half2v __attribute__((ext_vector_type(2)));
// max over the two lane halves of a query (lane l and l ^ 32 hold the two key halves of one query)
__device__ __forceinline__ float att2_halfmax(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // r[0] = the low half's value, r[1] = the high half's, in all lanes
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// (p0, p1) -> packed fp16 hi pair and packed fp16 residual pair (p - hi): v_cvt_pk_f16_f32, 2 x v_fma_mix_f32, v_cvt_pk_f16_f32
template <bool SPLIT>
__device__ __forceinline__ void att2_split2(float p0, float p1, unsigned& hi, unsigned& lo) {
    float2v a = {p0, p1};
    union { half2v v; unsigned u; } h, l;
    h.v = __builtin_convertvector(a, half2v);
    hi = h.u;
    if (SPLIT) {
        float d0, d1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(h.u), "v"(p0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(h.u), "v"(p1));
        float2v d = {d0, d1};
        l.v = __builtin_convertvector(d, half2v);
        lo = l.u;
    }
}

// Slot schedule of attn2_kernel's tile body (see there): items emitted after MFMA slot s = [first[s], first[s+1]).
#define ATT2_N_ITEMS (8 + 2 + 32 + 4 * 6 + 1)
struct Att2Sched {
    int first[49];
    constexpr Att2Sched() : first{} {
        int cost[ATT2_N_ITEMS] = {};
        int n = 0;
        for (int k = 0; k < 8; ++k) cost[n++] = 2;               // A
        cost[n++] = 3; cost[n++] = 4;                            // B
        for (int k = 0; k < 16; ++k) cost[n++] = 3;              // C t=0
        for (int g = 0; g < 2; ++g) { for (int j = 0; j < 4; ++j) cost[n++] = 4; cost[n++] = 2; cost[n++] = 2; }
        for (int k = 0; k < 16; ++k) cost[n++] = 3;              // C t=1
        for (int g = 2; g < 4; ++g) { for (int j = 0; j < 4; ++j) cost[n++] = 4; cost[n++] = 2; cost[n++] = 2; }
        cost[n++] = 2;                                           // L
        int it = 0, cum = 0;
        for (int s = 0; s < 48; ++s) {
            first[s] = it;
            const int target = (s + 1) * 5;                      // instructions issued by the end of slot s
            // the O-rescale branch sits after slot 4: A and B must be complete there
            while (it < ATT2_N_ITEMS && (cum + cost[it] <= target || (s == 4 && it < 10))) { cum += cost[it]; ++it; }
        }
        first[48] = ATT2_N_ITEMS;
    }
};
static constexpr Att2Sched ATT2_SCHED{};
static constexpr int IT_B = 8, IT_C0 = 10, IT_E0 = 26, IT_E1 = 32, IT_C1 = 38, IT_E2 = 54, IT_E3 = 60, IT_L = 66;
static_assert(ATT2_SCHED.first[5] >= IT_C0 && ATT2_SCHED.first[5] <= IT_C0 + 4, "A and B are done when the rescale branch is reached (after slot 4)");
static_assert(ATT2_SCHED.first[24] >= IT_E1 && ATT2_SCHED.first[30] >= IT_C1 && ATT2_SCHED.first[36] >= IT_E3 && ATT2_SCHED.first[42] >= IT_L,
              "P fragments of group g must be complete before PV slot 6g");
static_assert(ATT2_SCHED.first[29] <= IT_E2 && ATT2_SCHED.first[35] <= IT_E3, "a P-fragment buffer is rewritten only after the MFMAs that read it have issued");
static_assert(ATT2_SCHED.first[47] == ATT2_N_ITEMS, "every item is emitted");

template <int I, int N, class F>
__device__ __forceinline__ void att2_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); att2_static_for<I + 1, N>(f); }
}

// ABL (tools only, 0 in the product): component ablations of the tile body - bit 0: no MFMAs, bit 1: no VALU work items,
// bit 2: no fragment reads, bit 3: no per-tile DMA / wait / barrier.  Results are garbage; the timing says which part bounds.
template <bool SPLIT, int ABL = 0>
__global__ __launch_bounds__(512, 2) void attn2_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int STAGE = 2 * NPL * ATT_TILE_BYTES;   // K planes then V^T planes
    const int npose_blocks = p.pose ? (p.S * p.heads + 7) / 8 : 0;
    if ((int)blockIdx.x < npose_blocks) { attn_pose_query<SPLIT>(p, smem); return; }
    static_assert(SPLIT, "attn2_kernel: f16x3 only (precision f16 keeps attention.h)");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nqb = (p.nq + 255) / 256;
    const int nwg = nqb * p.heads * p.S;
    int logical;
    {
        const int bid = blockIdx.x - npose_blocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int qb = logical % nqb;
    const int h = (logical / nqb) % p.heads, s = logical / (nqb * p.heads);
    const int skv = (s + p.kv_shift) % p.S;
    const int q0 = qb * 256 + wave * 32;

    const size_t qoff = (size_t)(s * p.heads + h) * p.npad * 64;
    const size_t koff = (size_t)(skv * p.heads + h) * p.npad * 64;
    const size_t voff = (size_t)(skv * p.heads + h) * 64 * p.npad;

    // ---- Q fragments (B operand: col = query, 8 consecutive d per lane half and k step)
    half8 qf_hi[4], qf_lo[4];
    {
        int qrow = q0 + l31; if (qrow > p.nq - 1) qrow = p.nq - 1;
        const f16* qp = p.Q_hi + qoff + (size_t)qrow * 64 + lhi * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { H8 t; t.u = ldg16(qp + kk * 16); qf_hi[kk] = t.h; }
        if (SPLIT) {
            const f16* ql = p.Q_lo + qoff + (size_t)qrow * 64 + lhi * 8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { H8 t; t.u = ldg16(ql + kk * 16); qf_lo[kk] = t.h; }
        }
    }

    // ---- K / V^T tiles: global -> LDS by DMA; a plane tile = 8 slots of 1 KiB (8 rows), wave w moves slot w of every plane tile
    const int drow = wave * 8 + (lane >> 3);
    const int dsch = (lane & 7) ^ ((drow >> 1) & 7);
    const int ksrc_l = drow * 64 + dsch * 8, vsrc_l = drow * p.npad + dsch * 8;
    auto issue_tile = [&](int stage, int kv0) {
        char* sK = smem + stage * STAGE + wave * 1024;
        char* sV = sK + NPL * ATT_TILE_BYTES;
        const size_t ko = koff + (size_t)kv0 * 64 + ksrc_l;
        const size_t vo = voff + (size_t)kv0 + vsrc_l;
        glds16(p.K_hi + ko, sK);
        glds16(p.Vt_hi + vo, sV);
        if (SPLIT) {
            glds16(p.K_lo + ko, sK + ATT_TILE_BYTES);
            glds16(p.Vt_lo + vo, sV + ATT_TILE_BYTES);
        }
    };
    constexpr int GPT = 2 * NPL;                 // DMA instructions per wave per tile

    floatx16 oacc[2], sacc[2][2];                // sacc[b]: the score tile being consumed / the one being produced (b alternates)
    float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    // ---- the pose token as a key (index nk): initial state m = s_p, l = 1, O = v_p in fp32 (attention.h)
    if (p.pose) {
        float sp = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            H8 a, b; a.u = ldg16(p.K_hi + koff + (size_t)p.nk * 64 + kk * 16 + lhi * 8);
            if (SPLIT) b.u = ldg16(p.K_lo + koff + (size_t)p.nk * 64 + kk * 16 + lhi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                sp = __builtin_fmaf((float)qf_hi[kk][e] + (SPLIT ? (float)qf_lo[kk][e] : 0.f), (float)a.e[e] + (SPLIT ? (float)b.e[e] : 0.f), sp);
        }
        sp += __shfl_xor(sp, 32);
        m_run = sp * p.scale_log2e;
        l_run = lhi == 0 ? 1.f : 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t o = voff + (size_t)(d * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * p.npad + p.nk;
                oacc[d][r] = (float)p.Vt_hi[o] + (SPLIT ? (float)p.Vt_lo[o] : 0.f);
            }
    }

    int foff_l[4];                                // per-lane fragment byte offsets inside a plane tile (attention.h)
    {
        const int swz = (l31 >> 1) & 7;
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) foff_l[c2] = l31 * 128 + (((c2 * 2 + lhi) ^ swz) << 4);
    }
    using std::integral_constant;
    const int ntiles = (p.nk + ATT_KV - 1) / ATT_KV;
    const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ------------------------------------------------------------------------------------------------------------
    // The tile body as 48 MFMA SLOTS (24 of S^T(i+1), then 24 of PV(i)), each followed by its share of the VALU WORK ITEMS of
    // tile i (in dependency order, ~5 instructions per slot, all done by slot 40) and by the fragment reads of the MFMAs three
    // to six slots ahead; a sched_barrier(0) after every slot pins that order (hipcc on its own clusters the 48 MFMAs and
    // then runs the ~200 VALU instructions with the matrix pipe idle).
    //   items:  A0-7   running-max candidates (4 scores each)          B0,B1  row max over the lane halves, m_new, alpha
    //           C0-31  p = exp2(s*scale - m), row sum                   E(g)   fp16 hi / residual of 16-key group g + regrouping
    //           L      l = l*alpha + sum
    //   order:  A, B | (rare branch: O rescale) | C0-15, E(0), E(1), C16-31, E(2), E(3), L
    //   deadlines: E(g) before PV slot 6g (its first MFMA) - met by the uniform rate (asserted below).
    // The running max is only raised when some query's maximum grew by more than 2^8 (deferred rescale: p <= 256 is exact
    // enough for the fp16 hi + residual pair and far from fp32 overflow in l and O), so the O rescale is a rare branch.
    int st = 0;                                    // LDS stage of tile i
    float m_new = 0.f, alpha = 1.f, neg_m = 0.f, mx = 0.f, psum = 0.f;
    union PFrag { half8 h; unsigned u[4]; };
    PFrag ph[2], pl[2];                            // P fragments of PV groups g (buffer g & 1)
    half8 kfb[2][2], klb[2][2], vfb[2][2], vlb[2][2];   // K fragment pairs ([kk & 1][t]), V^T fragment pairs ([group & 1][d])

    auto tile_iter = [&](int i, auto cur_c, auto tail_c) {
        constexpr int CUR = decltype(cur_c)::value, NXT = 1 - CUR;
        constexpr bool TAIL = decltype(tail_c)::value;
        const int st1 = st == ATT2_STAGES - 1 ? 0 : st + 1, st2 = st1 == ATT2_STAGES - 1 ? 0 : st1 + 1;
        // tile i+1 (issued one tile ago) has landed; every wave is past PV(i-1): its stage (= st2) takes tile i+2
        if (!(ABL & 8)) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const int nxt = i + 2 < ntiles ? i + 2 : ntiles - 1;       // (clamped: a redundant reload instead of a branch)
        if (!(ABL & 8)) issue_tile(st2, nxt * ATT_KV);
        const char* sK = smem + st1 * STAGE;                        // K of tile i+1 (i+1 == ntiles: stale data, result discarded)
        const char* sV = smem + st * STAGE + NPL * ATT_TILE_BYTES;  // V^T of tile i
        const int kv0 = i * ATT_KV;
        if (TAIL) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (key >= p.nk) sacc[CUR][t][r] = -INFINITY;
                }
        }
#define ATT2_SC(e) sacc[CUR][(e) >> 4][(e) & 15]
        // ---- one VALU work item (IT is a compile-time constant)
        auto item = [&](auto it_c) {
            constexpr int IT = decltype(it_c)::value;
            if constexpr (IT < IT_B) {                               // A: running-max candidates
                const float a = fmaxf(fmaxf(ATT2_SC(4 * IT), ATT2_SC(4 * IT + 1)), fmaxf(ATT2_SC(4 * IT + 2), ATT2_SC(4 * IT + 3)));
                mx = IT == 0 ? a : fmaxf(mx, a);
            } else if constexpr (IT == IT_B) {
                mx = att2_halfmax(mx) * p.scale_log2e;
            } else if constexpr (IT == IT_B + 1) {
                // deferred rescale (THR = 8 in log2 units): keep the old reference max unless some query outgrew it by 2^8
                const bool grow = mx > m_run + 8.0f;
                m_new = grow ? mx : m_run;                          // (m_run == -inf on the first tile without a pose key: grows)
                alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                neg_m = -m_new;
                psum = 0.f;
            } else if constexpr ((IT >= IT_C0 && IT < IT_E0) || (IT >= IT_C1 && IT < IT_E2)) {     // C: one score
                constexpr int e = IT < IT_E0 ? IT - IT_C0 : 16 + IT - IT_C1;
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(ATT2_SC(e), p.scale_log2e, neg_m));
                ATT2_SC(e) = pv;
                psum += pv;
            } else if constexpr (IT == IT_L) {
                l_run = l_run * alpha + psum;
                m_run = m_new;
            } else {                                                 // E(g): six items per group
                constexpr int q = IT < IT_C1 ? IT - IT_E0 : 12 + IT - IT_E2;
                constexpr int g = q / 6, j = q % 6, e0 = (g >> 1) * 16 + 8 * (g & 1);
                if constexpr (j < 4) att2_split2<SPLIT>(ATT2_SC(e0 + 2 * j), ATT2_SC(e0 + 2 * j + 1), ph[g & 1].u[j], pl[g & 1].u[j]);
                else {
                    constexpr int e = j - 4;                         // regroup to the natural k order (attention.h)
                    auto r1 = __builtin_amdgcn_permlane32_swap(ph[g & 1].u[e], ph[g & 1].u[2 + e], false, false);
                    ph[g & 1].u[e] = r1[0]; ph[g & 1].u[2 + e] = r1[1];
                    auto r2 = __builtin_amdgcn_permlane32_swap(pl[g & 1].u[e], pl[g & 1].u[2 + e], false, false);
                    pl[g & 1].u[e] = r2[0]; pl[g & 1].u[2 + e] = r2[1];
                }
            }
        };
        auto kread = [&](auto kk_c, auto t_c) {                      // K fragment pair of key half t, k step kk
            constexpr int kk = decltype(kk_c)::value, t = decltype(t_c)::value;
            if (ABL & 4) { asm volatile("" : "+v"(kfb[kk & 1][t]), "+v"(klb[kk & 1][t])); return; }
            kfb[kk & 1][t] = *reinterpret_cast<const half8*>(sK + t * 4096 + foff_l[kk]);
            klb[kk & 1][t] = *reinterpret_cast<const half8*>(sK + ATT_TILE_BYTES + t * 4096 + foff_l[kk]);
        };
        auto vread = [&](auto g_c, auto d_c) {                       // V^T fragment pair of PV group g (chunk c2 = g), output half d
            constexpr int g = decltype(g_c)::value, d = decltype(d_c)::value;
            if (ABL & 4) { asm volatile("" : "+v"(vfb[g & 1][d]), "+v"(vlb[g & 1][d])); return; }
            vfb[g & 1][d] = *reinterpret_cast<const half8*>(sV + d * 4096 + foff_l[g]);
            vlb[g & 1][d] = *reinterpret_cast<const half8*>(sV + ATT_TILE_BYTES + d * 4096 + foff_l[g]);
        };
        // MFMA order: consecutive MFMAs alternate between the two accumulators of the phase (key halves t in S^T, output halves d
        // in PV), so an MFMA never waits for the result of the one right before it
        auto slot = [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            if constexpr (S < 24) {                                  // S^T(i+1): k step kk = S / 6, product (S % 6) / 2, key half t = S % 2
                constexpr int kk = S / 6, pr = (S % 6) / 2, t = S % 2;
                if constexpr (S % 6 == 0 && kk + 1 < 4) kread(integral_constant<int, kk + 1>{}, integral_constant<int, 0>{});
                if constexpr (S % 6 == 2 && kk + 1 < 4) kread(integral_constant<int, kk + 1>{}, integral_constant<int, 1>{});
                if constexpr (S == 20) { vread(integral_constant<int, 0>{}, integral_constant<int, 0>{}); }
                if constexpr (S == 22) { vread(integral_constant<int, 0>{}, integral_constant<int, 1>{}); }
                const floatx16& c = (kk == 0 && pr == 0) ? zero16 : sacc[NXT][t];
                if constexpr ((ABL & 1) != 0) { asm volatile("" : "+v"(sacc[NXT][t]) : "v"(klb[kk & 1][t]), "v"(kfb[kk & 1][t])); }
                else if constexpr (pr == 0) sacc[NXT][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(klb[kk & 1][t], qf_hi[kk], c, 0, 0, 0);
                else if constexpr (pr == 1) sacc[NXT][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfb[kk & 1][t], qf_lo[kk], c, 0, 0, 0);
                else sacc[NXT][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfb[kk & 1][t], qf_hi[kk], c, 0, 0, 0);
            } else {                                                 // PV(i): group g = P / 6, product (P % 6) / 2, output half d = P % 2
                constexpr int P = S - 24, g = P / 6, pr = (P % 6) / 2, d = P % 2;
                if constexpr (P % 6 == 0 && g + 1 < 4) vread(integral_constant<int, g + 1>{}, integral_constant<int, 0>{});
                if constexpr (P % 6 == 2 && g + 1 < 4) vread(integral_constant<int, g + 1>{}, integral_constant<int, 1>{});
                if constexpr ((ABL & 1) != 0) { asm volatile("" : "+v"(oacc[d]) : "v"(vlb[g & 1][d]), "v"(vfb[g & 1][d]), "v"(ph[g & 1].h), "v"(pl[g & 1].h)); }
                else if constexpr (pr == 0) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlb[g & 1][d], ph[g & 1].h, oacc[d], 0, 0, 0);
                else if constexpr (pr == 1) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfb[g & 1][d], pl[g & 1].h, oacc[d], 0, 0, 0);
                else oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfb[g & 1][d], ph[g & 1].h, oacc[d], 0, 0, 0);
            }
            if constexpr (!(ABL & 2)) att2_static_for<ATT2_SCHED.first[S], ATT2_SCHED.first[S + 1]>(item);
            __builtin_amdgcn_sched_barrier(0);
        };
        kread(integral_constant<int, 0>{}, integral_constant<int, 0>{});
        kread(integral_constant<int, 0>{}, integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        att2_static_for<0, 5>(slot);
        if (!__all(alpha == 1.0f)) {                                 // rare after the first tiles (deferred rescale)
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }
        att2_static_for<5, 48>(slot);
        st = st1;
    };
    const bool has_tail = (p.nk % ATT_KV) != 0;

    // ---- prologue: tiles 0 and 1 in flight, S^T(0) as soon as tile 0 has landed
    issue_tile(0, 0);
    issue_tile(1, ntiles > 1 ? ATT_KV : 0);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(GPT) : "memory");
    {
        const char* sK = smem;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[0][t][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                half8 kf = *reinterpret_cast<const half8*>(sK + t * 4096 + foff_l[kk]);
                half8 kl = *reinterpret_cast<const half8*>(sK + ATT_TILE_BYTES + t * 4096 + foff_l[kk]);
                sacc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qf_hi[kk], sacc[0][t], 0, 0, 0);
                sacc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf_lo[kk], sacc[0][t], 0, 0, 0);
                sacc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf_hi[kk], sacc[0][t], 0, 0, 0);
            }
        }
    }
    using B0 = integral_constant<int, 0>; using B1 = integral_constant<int, 1>;
    using F = integral_constant<bool, false>; using T = integral_constant<bool, true>;
    int i = 0;
    for (; i + 2 < ntiles; i += 2) { tile_iter(i, B0{}, F{}); tile_iter(i + 1, B1{}, F{}); }
    if (i + 2 == ntiles) {                          // two tiles left
        tile_iter(i, B0{}, F{});
        if (has_tail) tile_iter(i + 1, B1{}, T{}); else tile_iter(i + 1, B1{}, F{});
    } else {                                        // one tile left
        if (has_tail) tile_iter(i, B0{}, T{}); else tile_iter(i, B0{}, F{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped reloads must not outlive the workgroup's LDS allocation

    // ---- normalise and store: lane owns query q, d = dt*32 + (r&3) + 8*(r>>2) + 4*lhi
    const int64_t orows = (int64_t)p.S * p.nq + (p.pose ? p.S : 0);
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (q < p.nq) {
        const int64_t orow = (int64_t)s * p.nq + q;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dcol = d * 32 + 8 * g + 4 * lhi;
                const size_t o = blk_off<SPLIT>(orow, h * 64 + dcol, orows);
                if (SPLIT && p.o_mx) {
                    const float y[4] = {oacc[d][g * 4] * inv, oacc[d][g * 4 + 1] * inv, oacc[d][g * 4 + 2] * inv, oacc[d][g * 4 + 3] * inv};
                    store_mx4(p.O_hi, o, split_mx4<false>(y));
                    continue;
                }
                H4 oh, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = oacc[d][g * 4 + e] * inv;
                    if (SPLIT) split_f16(v, oh.e[e], ol.e[e]); else oh.e[e] = to_f16_sat(v);
                }
                *reinterpret_cast<uint2*>(p.O_hi + o) = oh.u;
                if (SPLIT) *reinterpret_cast<uint2*>(p.O_hi + o + 32) = ol.u;
            }
    }
}
