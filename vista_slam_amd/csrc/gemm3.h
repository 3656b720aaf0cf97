// One-wave-per-SIMD MFMA GEMM for gfx950: 256x256x32 block tile, 4 waves (2x2), each wave owns a
// 128x128 output tile = 4x4 MFMA 32x32 tiles = 256 accumulator registers and uses the whole 512-entry
// register file of its SIMD.  Same operands / epilogues / LDS image as gemm2.h; what changes is the
// schedule:
//   * no second wave competes for the matrix pipe: the wave's own instruction stream keeps it busy
//     (48 MFMAs per K step in f16x3, 16 ds_read_b128 + 8 LDS-DMA issues to hide in their shadow);
//   * fragment registers are double-buffered: the reads for K step s+1 are issued before the MFMAs of
//     step s, and the reads for the first step of the NEXT K tile are issued right after the barrier
//     but BEFORE the last 16 MFMAs of the current tile, so neither LDS latency nor the barrier skew
//     is exposed;
//   * LDS traffic per K tile drops from 192 KiB (8 waves x 64-wide tiles) to 128 KiB.
#pragma once
#include "gemm2.h"

template <bool SPLIT, int AMODE, int EPI>
__global__ __launch_bounds__(256, 1) void gemm3_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, NW = 4, MT = 4, NT = 4;
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;
    constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
    constexpr int SA = BM / 16 / NW, SB = BN / 16 / NW;     // 4 + 4 one-KiB DMA slots per wave per plane

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int t;
    {
        const int bid = blockIdx.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    int bm, bn;
    {
        const int band = t / (4 * tiles_n);
        const int hb = tiles_m - band * 4 < 4 ? tiles_m - band * 4 : 4;
        const int local = t - band * 4 * tiles_n;
        bm = band * 4 + local % hb;
        bn = local / hb;
    }
    const int m0 = bm * BM, n0 = bn * BN;

    // ---- per-lane DMA sources (row-major operands; chunk swizzle on the source side)
    const int row_in = lane >> 2;
    const int src_chunk = (lane & 3) ^ ((row_in >> 2) & 3);
    size_t a_off[SA], b_off[SB];
    int cv_img[SA], cv_y[SA], cv_x[SA]; bool cv_ok[SA];
#pragma unroll
    for (int s = 0; s < SA; ++s) {
        const int row = 16 * (wave + NW * s) + row_in;
        const int gm = m0 + row;
        if (AMODE == A_DENSE) {
            const int gmc = gm < p.M ? gm : p.M - 1;
            a_off[s] = (size_t)gmc * p.lda + src_chunk * 8;
        } else {
            cv_ok[s] = gm < p.M;
            const int gmc = cv_ok[s] ? gm : 0;
            const int hw = p.Ho * p.Wo;
            cv_img[s] = gmc / hw;
            const int rem = gmc - cv_img[s] * hw;
            cv_y[s] = (rem / p.Wo) * p.cstride - 1;
            cv_x[s] = (rem % p.Wo) * p.cstride - 1;
            a_off[s] = 0;
        }
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        const int row = 16 * (wave + NW * s) + row_in;
        const int gn = n0 + row;
        const int gnc = gn < p.N ? gn : p.N - 1;
        b_off[s] = (size_t)gnc * p.K + src_chunk * 8;
    }

    // DMA of one K tile, split in 4 groups (A_hi, A_lo, B_hi, B_lo planes; 4 one-KiB slots each) so the
    // issue can be spread between MFMA groups.  part: 0 = A plane 0, 1 = A plane 1, 2 = B plane 0, 3 = B plane 1.
    auto issue_part = [&](int kt, int stage, int part) {
        const int k0 = kt * GEMM_BK;
        char* sA = smem + stage * STAGE;
        char* sB = sA + NPL * A_PLANE;
        if (part < 2) {
            if (part == 1 && !SPLIT) return;
            int tap = 0, c0 = 0, ky = 0, kx = 0;
            if (AMODE == A_CONV3) { tap = k0 / p.Cin; c0 = k0 - tap * p.Cin; ky = tap / 3; kx = tap - ky * 3; }
            const f16* base = part == 0 ? p.A_hi : p.A_lo;
#pragma unroll
            for (int s = 0; s < SA; ++s) {
                char* dst = sA + part * A_PLANE + (wave + NW * s) * 1024;
                if (AMODE == A_DENSE) {
                    glds16(base + a_off[s] + k0, dst);
                } else {
                    const int yi = cv_y[s] + ky, xi = cv_x[s] + kx;
                    const bool ok = cv_ok[s] && yi >= 0 && yi < p.Hi && xi >= 0 && xi < p.Wi;
                    const size_t off = ((size_t)(cv_img[s] * p.Hi + yi) * p.Wi + xi) * p.Cin + c0 + src_chunk * 8;
                    glds16(ok ? base + off : p.zero_page, dst);
                }
            }
        } else {
            if (part == 3 && !SPLIT) return;
            const f16* base = part == 2 ? p.B_hi : p.B_lo;
#pragma unroll
            for (int s = 0; s < SB; ++s)
                glds16(base + b_off[s] + k0, sB + (part - 2) * B_PLANE + (wave + NW * s) * 1024);
        }
    };
    auto issue_tile = [&](int kt, int stage) {
#pragma unroll
        for (int part = 0; part < 4; ++part) issue_part(kt, stage, part);
    };

    struct Frag { half8 a_hi[MT], a_lo[MT], b_hi[NT], b_lo[NT]; };
    auto load_frag = [&](Frag& f, int stage, int ks) {
        const char* sA = smem + stage * STAGE;
        const char* sB = sA + NPL * A_PLANE;
        const int chunk = ks * 2 + lhi;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int off = lds_off(wm * 128 + i * 32 + l31, chunk);
            f.a_hi[i] = *reinterpret_cast<const half8*>(sA + off);
            if (SPLIT) f.a_lo[i] = *reinterpret_cast<const half8*>(sA + A_PLANE + off);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int off = lds_off(wn * 128 + j * 32 + l31, chunk);
            f.b_hi[j] = *reinterpret_cast<const half8*>(sB + off);
            if (SPLIT) f.b_lo[j] = *reinterpret_cast<const half8*>(sB + B_PLANE + off);
        }
        if (AMODE == A_CONV3) {
            if (p.relu_in) {   // relu(hi + lo): the sign of hi decides
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const short8 neg = f.a_hi[i] < (half8)(f16)0;
                    f.a_hi[i] = __builtin_bit_cast(half8, (short8)(__builtin_bit_cast(short8, f.a_hi[i]) & ~neg));
                    if (SPLIT) f.a_lo[i] = __builtin_bit_cast(half8, (short8)(__builtin_bit_cast(short8, f.a_lo[i]) & ~neg));
                }
            }
        }
    };

    floatx16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // MFMAs of one K step on rows [i0, i1) of the wave tile (product-major: 16 accumulators apart)
    auto mma_rows = [&](const Frag& f, int i0, int i1) {
        if (SPLIT) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (i >= i0 && i < i1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a_lo[i], f.b_hi[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (i >= i0 && i < i1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a_hi[i], f.b_lo[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if (i >= i0 && i < i1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a_hi[i], f.b_hi[j], acc[i][j], 0, 0, 0);
    };

    const int nkt = p.K / GEMM_BK;
    Frag f0, f1;
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_frag(f0, 0, 0);

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        // K step 1 fragments: in flight under the MFMAs of K step 0
        load_frag(f1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        // K step 0, one row of 32x32 tiles at a time, with the next tile's DMA issued in the gaps
        // (stage cur^1 was released by the barrier of iteration kt-1)
        mma_rows(f0, 0, 1);
        if (more) issue_part(kt + 1, cur ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(f0, 1, 2);
        if (more) issue_part(kt + 1, cur ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(f0, 2, 3);
        if (more) issue_part(kt + 1, cur ^ 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(f0, 3, 4);
        if (more) issue_part(kt + 1, cur ^ 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(f1, 0, MT - 1);                   // first 3/4 of K step 1
        __builtin_amdgcn_sched_barrier(0);
        // every LDS read of stage `cur` has completed (consumed above); wait for this wave's DMA of the
        // next tile, then meet the other waves: stage cur^1 is complete, stage cur is free.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (more) load_frag(f0, cur ^ 1, 0);       // latency covered by the remaining MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(f1, MT - 1, MT);
        __builtin_amdgcn_sched_barrier(0);
    }

#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
            epilogue_tile<SPLIT, EPI>(p, acc[i][j], m0 + wm * 128 + i * 32, n0 + wn * 128 + j * 32 + l31, lane);
}
