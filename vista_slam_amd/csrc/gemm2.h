// Large-tile MFMA GEMM / implicit-GEMM conv for gfx950 (the throughput kernel).
//
// Same math, operands, epilogues and GemmParams as gemm.h; different machine mapping:
//  * BM x BN x 32 block tile with BM = 256, BN in {256, 128}, 8 waves (512 threads, 2 waves per SIMD,
//    one workgroup per CU).  256x256: waves 2(M) x 4(N), each 128x64 = 4x2 MFMA 32x32 tiles
//    (128 accumulator registers).  Per K step a wave issues 12 ds_read_b128 for 24 (split: 3 products)
//    v_mfma_f32_32x32x16_f16, i.e. 48 MFMAs (1536 matrix-pipe cycles) per barrier - 4x the work per
//    barrier and half the L2->LDS bytes per FLOP of the 128x128 kernel.
//  * Operands go global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write):
//    each wave instruction fills 1 KiB = 16 rows x 64 B.  LDS stays lane-linear as the DMA requires;
//    the bank-conflict swizzle (16-B chunk ^= (row>>2)&3) is applied to the per-lane SOURCE address
//    and again on the fragment read (same involution on both sides).
//  * Two LDS stages (2 x 64 KiB for the split 256x256 tile): DMA of tile k+1 overlaps the MFMAs of
//    tile k; one vmcnt(0) + barrier per K step.
//  * 3x3 conv taps that fall outside the image read a zero page (the DMA cannot synthesise zeros);
//    the RCU's input ReLU is applied on the fragment registers.
//  * Block id -> tile: bijective XCD remap (consecutive logical tiles share an XCD's L2) and
//    band-major order (4 M-tiles x all N-tiles) so co-resident blocks reuse A row panels / W panels.
#pragma once
#include "gemm.h"

typedef short short8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

template <bool SPLIT, int BM, int BN>
constexpr int gemm2_smem_bytes() { return 2 * (SPLIT ? 2 : 1) * (BM + BN) * 64; }

// ABL (bench-only ablations, 0 in the product): 1 = no DMA inside the K loop, 2 = DMA + barriers only
// (no LDS reads, no MFMA), 3 = MFMA on stale registers (no LDS reads).
template <bool SPLIT, int AMODE, int EPI, int BM, int BN, int WAVES_M, int WAVES_N, int ABL = 0>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm2_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MT = WM / 32, NT = WN / 32;
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;
    constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
    // 1-KiB DMA slots (16 rows) per plane: slot g = wave + NW*s, valid while g < rows/16 (BM = 192 gives
    // waves 0-3 two A slots and waves 4-7 one; the guard is wave-uniform)
    constexpr int NSA = BM / 16, NSB = BN / 16;
    constexpr int SA = (NSA + NW - 1) / NW, SB = (NSB + NW - 1) / NW;
    static_assert(BM % 32 == 0 && BN % 32 == 0 && WM % 32 == 0 && WN % 32 == 0, "tile/wave mismatch");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- block id -> tile (XCD-aware, band-major)
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

    // ---- per-lane DMA source bookkeeping
    const int row_in = lane >> 2;
    const int src_chunk = (lane & 3) ^ ((row_in >> 2) & 3);
    const f16* a_src_hi[SA]; const f16* a_src_lo[SA];
    const f16* b_src_hi[SB]; const f16* b_src_lo[SB];
    int cv_img[SA], cv_y[SA], cv_x[SA]; bool cv_ok[SA];
#pragma unroll
    for (int s = 0; s < SA; ++s) {
        const int row = 16 * (wave + NW * s) + row_in;
        const int gm = m0 + row;
        if (AMODE == A_DENSE) {
            const int gmc = gm < p.M ? gm : p.M - 1;
            const size_t rs = (ABL & 4) ? 32 : p.lda;   // bench-only: K-tile-blocked source layout [K/32][M][32]
            a_src_hi[s] = p.A_hi + (size_t)gmc * rs + src_chunk * 8;
            a_src_lo[s] = SPLIT ? p.A_lo + (size_t)gmc * rs + src_chunk * 8 : nullptr;
        } else {
            cv_ok[s] = gm < p.M;
            const int gmc = cv_ok[s] ? gm : 0;
            const int hw = p.Ho * p.Wo;
            cv_img[s] = gmc / hw;
            const int rem = gmc - cv_img[s] * hw;
            cv_y[s] = (rem / p.Wo) * p.cstride - 1;
            cv_x[s] = (rem % p.Wo) * p.cstride - 1;
            a_src_hi[s] = nullptr; a_src_lo[s] = nullptr;
        }
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        const int row = 16 * (wave + NW * s) + row_in;
        const int gn = n0 + row;
        const int gnc = gn < p.N ? gn : p.N - 1;
        const size_t rsb = (ABL & 4) ? 32 : p.K;
        b_src_hi[s] = p.B_hi + (size_t)gnc * rsb + src_chunk * 8;
        b_src_lo[s] = SPLIT ? p.B_lo + (size_t)gnc * rsb + src_chunk * 8 : nullptr;
    }

    auto issue_tile = [&](int kt, int stage) {
        const int k0 = kt * GEMM_BK;
        const size_t ka = (ABL & 4) ? (size_t)kt * p.M * 32 : (size_t)k0, kb = (ABL & 4) ? (size_t)kt * p.N * 32 : (size_t)k0;
        char* sA = smem + stage * STAGE;
        char* sB = sA + NPL * A_PLANE;
        int tap = 0, c0 = 0, ky = 0, kx = 0;
        if (AMODE == A_CONV3) { tap = k0 / p.Cin; c0 = k0 - tap * p.Cin; ky = tap / 3; kx = tap - ky * 3; }
#pragma unroll
        for (int s = 0; s < SA; ++s) {
            if (NSA % NW != 0 && wave + NW * s >= NSA) continue;
            char* dst = sA + (wave + NW * s) * 1024;
            if (AMODE == A_DENSE) {
                glds16(a_src_hi[s] + ka, dst);
                if (SPLIT) glds16(a_src_lo[s] + ka, dst + A_PLANE);
            } else {
                const int yi = cv_y[s] + ky, xi = cv_x[s] + kx;
                const bool ok = cv_ok[s] && yi >= 0 && yi < p.Hi && xi >= 0 && xi < p.Wi;
                const size_t off = ((size_t)(cv_img[s] * p.Hi + yi) * p.Wi + xi) * p.Cin + c0 + src_chunk * 8;
                glds16(ok ? p.A_hi + off : p.zero_page, dst);
                if (SPLIT) glds16(ok ? p.A_lo + off : p.zero_page, dst + A_PLANE);
            }
        }
#pragma unroll
        for (int s = 0; s < SB; ++s) {
            if (NSB % NW != 0 && wave + NW * s >= NSB) continue;
            char* dst = sB + (wave + NW * s) * 1024;
            glds16(b_src_hi[s] + kb, dst);
            if (SPLIT) glds16(b_src_lo[s] + kb, dst + B_PLANE);
        }
    };

    floatx16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = p.K / GEMM_BK;
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt && (ABL & 3) != 1) issue_tile(kt + 1, cur ^ 1);
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + NPL * A_PLANE;
        half8 a_hi[MT], a_lo[MT], b_hi[NT], b_lo[NT];
        if ((ABL & 3) == 3) {
#pragma unroll
            for (int i = 0; i < MT; ++i) { a_hi[i] = (half8)(f16)(0.001f * (lane + i)); a_lo[i] = a_hi[i]; asm volatile("" : "+v"(a_hi[i]), "+v"(a_lo[i])); }
#pragma unroll
            for (int j = 0; j < NT; ++j) { b_hi[j] = (half8)(f16)(0.002f * (lane + j)); b_lo[j] = b_hi[j]; asm volatile("" : "+v"(b_hi[j]), "+v"(b_lo[j])); }
        }
#pragma unroll
        for (int ks = 0; ks < 2 && (ABL & 3) != 2; ++ks) {
            const int chunk = ks * 2 + lhi;
#pragma unroll
            for (int i = 0; i < MT && (ABL & 3) != 3; ++i) {
                const int ra = wm * WM + i * 32 + l31;
                a_hi[i] = *reinterpret_cast<const half8*>(sA + lds_off(ra, chunk));
                if (SPLIT) a_lo[i] = *reinterpret_cast<const half8*>(sA + A_PLANE + lds_off(ra, chunk));
                if (AMODE == A_CONV3) {
                    if (p.relu_in) {   // relu(hi + lo): the sign of hi decides
                        const short8 neg = a_hi[i] < (half8)(f16)0;
                        a_hi[i] = __builtin_bit_cast(half8, (short8)(__builtin_bit_cast(short8, a_hi[i]) & ~neg));
                        if (SPLIT) a_lo[i] = __builtin_bit_cast(half8, (short8)(__builtin_bit_cast(short8, a_lo[i]) & ~neg));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NT && (ABL & 3) != 3; ++j) {
                const int rb = wn * WN + j * 32 + l31;
                b_hi[j] = *reinterpret_cast<const half8*>(sB + lds_off(rb, chunk));
                if (SPLIT) b_lo[j] = *reinterpret_cast<const half8*>(sB + B_PLANE + lds_off(rb, chunk));
            }
            // product-major order: consecutive MFMAs hit different accumulators (MT*NT apart)
            if (SPLIT) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[i], b_hi[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[i], b_lo[j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
            epilogue_tile<SPLIT, EPI>(p, acc[i][j], m0 + wm * WM + i * 32, n0 + wn * WN + j * 32 + l31, lane);
}
