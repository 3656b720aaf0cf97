// Halo-tiled 3x3 convolution (stride 1, pad 1) for gfx950: the throughput form of the DPT head's convolutions
// (ResidualConvUnit_custom, layer_rn, head.0 / head.2: heads/dpt_block.py:20-77,121-142,316-324).
//
// The implicit-GEMM loader of gemm2.h (A_CONV3) DMA's every input pixel of a tile once per tap - nine times per output
// tile - and the counters showed the re-reads reaching the fabric (3-6.6x the algorithmic bytes, round 2).  Here a
// workgroup owns a 2-D pixel tile of TR x 32 outputs (TR = BM / 32: one 32-row MFMA tile = 32 pixels of ONE image row) and,
// per 32-channel block, DMA's its (TR + 2) x 34 input HALO into LDS once; the nine taps are nine LDS offsets into it:
//     K loop:  for channel block cb:  for tap (ky, kx):  acc += halo[(ty + ky, tx + kx), cb] . W[tap, cb]^T
//  * A-operand DMA per output tile: (TR + 2) * 34 / (TR * 32) = 1.33x the input (256-pixel tile) instead of 9x; the
//    global->LDS stream per K step drops from (A 32 KiB + B) to (A 4.8 KiB + B): what bounds these kernels.
//  * LDS image of the halo: one 128-B row [hi32 | lo32] per halo pixel hp = hy * 34 + hx, 16-B chunk index XOR
//    (hp >> 1) & 7 - the same involution as the GEMM tiles, keyed on the HALO pixel: a fragment read touches 32
//    consecutive halo pixels of one halo row, i.e. 16 distinct hp mod 16 per ds_read_b128 lane group = all 64 banks once.
//  * Two halo stages (channel block cb + 1 streams in, spread over the K steps of cb) and two weight stages (step s + 1 under
//    step s); one vmcnt(0) + barrier per K step; a K step is one tap (256-column tiles) or two (128-column tiles, round 6);
//    out-of-image halo pixels read the zero page.
//  * 16 waves (4 x 4), one workgroup per CU: 256 x 128 tile (wave 64 x 32) for Cout = 128 - its epilogue can be the fused DPT
//    tail (EPI_HEAD) - and 256 x 256 (wave 64 x 64) for Cout = 256.  f16 / f16x3 / f16mx arithmetic as in gemm2.h.
//  * Epilogue: the ordinary plane epilogue per 32-pixel row segment (bias, ReLU, residual planes), bounded to the image.
#pragma once
#include "gemm2.h"
#include <type_traits>

#define C3H_PW 34     // halo width: 32 output columns + 2

template <bool SPLIT, int BM>
constexpr int conv3h_halo_bytes() { return ((BM / 32 + 2) * C3H_PW * (SPLIT ? 128 : 64) + 1023) / 1024 * 1024; }
// taps per K step (= per barrier): two for the 128-column tiles (a weight tile of one tap is only 16 KiB and carries 256 MFMA clocks per
// wave - a barrier + a DMA round trip per tap is what the period of those kernels was made of), one for the 256-column tiles (two
// 64-KiB weight stages would not fit next to the two halo stages)
template <int BN>
constexpr int conv3h_tps() { return BN == 128 ? 2 : 1; }
template <bool SPLIT, int BM, int BN>
constexpr int conv3h_smem_bytes() { return 2 * conv3h_halo_bytes<SPLIT, BM>() + 2 * conv3h_tps<BN>() * BN * (SPLIT ? 128 : 64); }

// (A 3-stage weight ring with counted vmcnt, an 8-wave 256 x 128 tile, Cout = 256 as two 128-column tiles and a persistent tile loop
// were measured and dropped: -0.3 %, equal, -4 %, equal; LABNOTES.md.)
template <bool SPLIT, int EPI, int BM, int BN, int WAVES_M, int WAVES_N, bool MX>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv3h_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MT = WM / 32, NT = WN / 32, TR = BM / 32;
    constexpr int RB = SPLIT ? 128 : 64, RPS = 1024 / RB, CPR = RB / 16;
    constexpr int A_ES = SPLIT ? 64 : 32;
    constexpr int HALO = conv3h_halo_bytes<SPLIT, BM>(), B_TILE = BN * RB;
    constexpr int HPIX = (TR + 2) * C3H_PW;
    constexpr int NHS = (HPIX + RPS - 1) / RPS;            // 1-KiB DMA slots of one halo
    constexpr int TPS = conv3h_tps<BN>();                  // taps per K step
    // the next channel block's halo streams in over the K steps of this one, HPS slots per step, one per wave.  TPS == 1: nine steps.
    // TPS == 2: a step may hold tap 8 of block cb AND tap 0 of cb + 1, so (a) the halo of cb + 1 has to be complete one step early and
    // (b) stage (cb + 1) & 1 may only be overwritten by a step whose FIRST tap belongs to cb (no tap of that step reads cb - 1): the steps
    // with first tap 0..7 of cb, of which every block has exactly four (taps 0,2,4,6 or 1,3,5,7)
    constexpr int NISS = TPS == 1 ? 9 : 4;
    constexpr int HPS = (NHS + NISS - 1) / NISS;
    constexpr int NSB = BN / RPS, SB = (NSB + NW - 1) / NW;
    static_assert((NW & (NW - 1)) == 0 && HPS <= NW && WM % 32 == 0 && WN % 32 == 0 && (!MX || SPLIT), "conv3h tile / wave mismatch");
    static_assert(EPI == EPI_F16 || EPI == EPI_HEAD, "conv3h: plane epilogue or the fused DPT tail");
    // fused DPT tail: the MFMAs run with their operands swapped - the accumulator tiles come out TRANSPOSED (lane = pixel, register =
    // channel), which is the layout head.4's contraction over the channels wants as an MFMA operand (head_epilogue_t, gemm2.h)
    constexpr bool TRN = EPI == EPI_HEAD && NT == 1;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lhi = lane >> 5;
    unsigned long long st0 = 0, st1 = 0, st2 = 0;        // tools only (GemmParams::stamps, as in gemm2_body)
    if (p.stamps) st0 = __builtin_amdgcn_s_memrealtime();

    // ---- block id -> (pixel tile, N tile): XCD-contiguous ranges, bands of 4 pixel tiles x all N tiles (as gemm2.h)
    const int tiles_x = (p.Wo + 31) >> 5, tiles_y = (p.Ho + TR - 1) / TR;
    const int n_img = p.M / (p.Ho * p.Wo);
    const int tiles_m = n_img * tiles_y * tiles_x, tiles_n = (p.N + BN - 1) / BN;
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
    const int img = bm / (tiles_y * tiles_x);
    const int trem = bm - img * (tiles_y * tiles_x);
    const int y0 = (trem / tiles_x) * TR, x0 = (trem % tiles_x) * 32;
    const int n0 = bn * BN;

    // ---- DMA bookkeeping.  Lane l of a 1-KiB slot fills LDS row (l / CPR), chunk position (l % CPR) with the source chunk
    // (l % CPR) ^ swizzle(row).
    const int row_in = lane / CPR, c_lds = lane % CPR;
    unsigned b_src[SB];
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        const int row = RPS * (wave + NW * s) + row_in;
        const int sw = SPLIT ? (row >> 1) & 7 : (row >> 2) & 3;
        const int gn = n0 + row;
        const int gnc = gn < p.N ? gn : p.N - 1;
        b_src[s] = ((unsigned)gnc * 64 + (c_lds ^ sw) * 8) * 2u;              // weights are always [hi32|lo32]
    }
    const size_t b_kstride = (size_t)p.N * 64;
    const int cblocks = p.Cin >> 5;
    const int nkt = 9 * cblocks;
    char* const sH = smem;
    char* const sB = smem + 2 * HALO;

    // weight K tile (tap, cb): K order of the packed weights is (ky, kx, ci), the loop runs cb outermost - the tile's base is a running
    // byte offset (next tap: + cblocks K tiles; next channel block: back to tap 0, + 1 K tile) instead of a 64-bit multiply per step
    const size_t b_tap_bytes = (size_t)cblocks * b_kstride * 2, b_cb_bytes = b_kstride * 2;
    size_t b_off = 0;                                          // byte offset of the NEXT tile issue_b will be asked for
    int b_tap = 0;                                             // ... and its tap (tiles are requested in K-loop order)
    auto issue_b = [&](int slot) {                             // -> weight slot `slot` of 2 * TPS (stage * TPS + tap within the step)
        const char* base = reinterpret_cast<const char*>(p.B_hi) + b_off;
        if (b_tap == 8) { b_off = b_off - 8 * b_tap_bytes + b_cb_bytes; b_tap = 0; } else { b_off += b_tap_bytes; ++b_tap; }
#pragma unroll
        for (int s = 0; s < SB; ++s) {
            if (NSB % NW != 0 && wave + NW * s >= NSB) continue;
            unsigned o = b_src[s];
            asm volatile("" : "+v"(o));
            glds16(base + o, sB + slot * B_TILE + (wave + NW * s) * 1024);
        }
    };
    auto issue_halo = [&](int j, int cb, int stage) {          // halo slot j (RPS halo pixels) of channel block cb
        const int hp = j * RPS + row_in;
        const int sw = SPLIT ? (hp >> 1) & 7 : (hp >> 2) & 3;
        const int hy = hp / C3H_PW, hx = hp - hy * C3H_PW;
        const int yi = y0 - 1 + hy, xi = x0 - 1 + hx;
        const bool ok = hp < HPIX && yi >= 0 && yi < p.Hi && xi >= 0 && xi < p.Wi;
        const size_t pix = (size_t)(img * p.Hi + yi) * p.Wi + xi;
        glds16(ok ? p.A_hi + ((size_t)cb * p.a_rp + pix) * A_ES + (c_lds ^ sw) * 8 : p.zero_page, sH + stage * HALO + j * 1024);
    };

    floatx16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int j = wave; j < NHS; j += NW) issue_halo(j, 0, 0);
#pragma unroll
    for (int u = 0; u < TPS; ++u) if (u < nkt) issue_b(u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.stamps) st1 = __builtin_amdgcn_s_memrealtime();

    // The K loop exists twice, with the input ReLU of resConfUnit*.conv1 resolved at COMPILE time (round 6): tested per fragment
    // inside the loop, the wave-uniform flag was six branches per K step - basic-block boundaries hipcc schedules nothing across
    // (every fragment read waited right in front of its MFMAs), in the kernels with no ReLU at all (head.0, the fused tail) too.
    auto k_loop = [&](auto relu_c) {
    constexpr bool RELU = decltype(relu_c)::value;
    int cb = 0, tap = 0;                                       // first tap of the step
    for (int kt = 0; kt < nkt; kt += TPS) {
        const int cur = (kt / TPS) & 1;
#pragma unroll
        for (int u = 0; u < TPS; ++u) if (kt + TPS + u < nkt) issue_b((cur ^ 1) * TPS + u);
        if (cb + 1 < cblocks && (TPS == 1 || tap < 8)) {       // halo of the next channel block: HPS slots per issuing step, one per wave
            const int is = TPS == 1 ? tap : tap >> 1;
            const int q = (wave - is * HPS) & (NW - 1);
            const int j = is * HPS + q;
            if (q < HPS && j < NHS) issue_halo(j, cb + 1, (cb & 1) ^ 1);
        }
        auto do_tap = [&](int cb, int tap, int slot) {
        const int hs = cb & 1;
        const int ky = tap / 3, kx = tap - ky * 3;
        const char* hA = sH + hs * HALO;
        const char* bB = sB + slot * B_TILE;
        int hp[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) hp[i] = (wm * MT + i + ky) * C3H_PW + l31 + kx;
        if constexpr (MX) {
            typedef int int4v __attribute__((ext_vector_type(4)));
            typedef int int8v __attribute__((ext_vector_type(8)));
            half8 ah[MT], bh[NT];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int chunk = ks * 2 + lhi;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    ah[i] = *reinterpret_cast<const half8*>(hA + lds2_off<true>(hp[i], chunk));
                    if (RELU) {
                        union { half8 h; unsigned u[4]; } tt; tt.h = ah[i];
#pragma unroll
                        for (int w = 0; w < 4; ++w) { const unsigned sgn = (tt.u[w] >> 15) & 0x00010001u; tt.u[w] &= ~((sgn << 16) - sgn); }
                        ah[i] = tt.h;
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) bh[j] = *reinterpret_cast<const half8*>(bB + lds2_off<true>(wn * WN + j * 32 + l31, chunk));
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = TRN ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
            union U8 { struct { int4v x, y; } q; int8v v; };
            U8 a8[MT], b8[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                a8[i].q.x = *reinterpret_cast<const int4v*>(hA + lds2_off<true>(hp[i], 4 + 2 * lhi));
                a8[i].q.y = *reinterpret_cast<const int4v*>(hA + lds2_off<true>(hp[i], 5 + 2 * lhi));
                if (RELU) {
#pragma unroll
                    for (int w = 0; w < 8; ++w) { const unsigned u = (unsigned)a8[i].v[w]; const unsigned sgn = (u >> 7) & 0x00010001u; a8[i].v[w] = (int)(u & ~((sgn << 16) - sgn)); }
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int rb = wn * WN + j * 32 + l31;
                b8[j].q.x = *reinterpret_cast<const int4v*>(bB + lds2_off<true>(rb, 4 + 2 * lhi));
                b8[j].q.y = *reinterpret_cast<const int4v*>(bB + lds2_off<true>(rb, 5 + 2 * lhi));
            }
            constexpr int sc_a = 127 - STA_MX_A_SLO, sc_b = 127 - STA_MX_W_SHI;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = TRN ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[j].v, a8[i].v, acc[i][j], 0 /* A: e4m3 weights */, 1 /* B: e5m2 */, 0, sc_b, 0, sc_a)
                                   : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i].v, b8[j].v, acc[i][j], 1 /* A: e5m2 */, 0 /* B: e4m3 */, 0, sc_a, 0, sc_b);
        } else {
            half8 a_hi[MT], a_lo[MT], b_hi[NT], b_lo[NT];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int chunk = ks * 2 + lhi;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    a_hi[i] = *reinterpret_cast<const half8*>(hA + lds2_off<SPLIT>(hp[i], chunk));
                    if (SPLIT) a_lo[i] = *reinterpret_cast<const half8*>(hA + lds2_off<SPLIT>(hp[i], 4 + chunk));
                    if (RELU) {              // relu(hi + lo): the sign of hi decides (packed-half integer form, gemm2.h)
                        union { half8 h; unsigned u[4]; } ah, al;
                        ah.h = a_hi[i]; al.h = a_lo[i];
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const unsigned sgn = (ah.u[w] >> 15) & 0x00010001u;
                            const unsigned m = (sgn << 16) - sgn;
                            ah.u[w] &= ~m;
                            if (SPLIT) al.u[w] &= ~m;
                        }
                        a_hi[i] = ah.h; if (SPLIT) a_lo[i] = al.h;
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int rb = wn * WN + j * 32 + l31;
                    b_hi[j] = *reinterpret_cast<const half8*>(bB + lds2_off<SPLIT>(rb, chunk));
                    if (SPLIT) b_lo[j] = *reinterpret_cast<const half8*>(bB + lds2_off<SPLIT>(rb, 4 + chunk));
                }
                if (SPLIT) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = TRN ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b_hi[j], a_lo[i], acc[i][j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[i], b_hi[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = TRN ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b_lo[j], a_hi[i], acc[i][j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[i], b_lo[j], acc[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = TRN ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b_hi[j], a_hi[i], acc[i][j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
            }
        }
        };      // do_tap
#pragma unroll
        for (int u = 0; u < TPS; ++u) {
            if (u > 0 && kt + u >= nkt) break;
            int c = cb, t = tap + u;
            if (t >= 9) { t -= 9; ++c; }
            do_tap(c, t, cur * TPS + u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tap += TPS;
        if (tap >= 9) { tap -= 9; ++cb; }
    }
    };
    if (EPI != EPI_HEAD && p.relu_in) k_loop(std::integral_constant<bool, true>{});
    else k_loop(std::integral_constant<bool, false>{});

    if (p.stamps) { asm volatile("" ::"v"(acc[MT - 1][NT - 1][15]), "v"(acc[0][0][0]) : "memory"); st2 = __builtin_amdgcn_s_memrealtime(); }
    // ---- epilogue: MFMA tile (i, j) of this wave = the 32 pixels (y0 + wm*MT + i, x0 .. x0 + 31) x 32 channels
    const int cols_valid = p.Wo - x0 < 32 ? p.Wo - x0 : 32;
    if constexpr (EPI == EPI_HEAD) {
        if constexpr (BN == 128) {
            const int rows_valid = p.Ho - y0 < TR ? p.Ho - y0 : TR;
            if constexpr (TRN) head_epilogue_t<BM, MT, WM, WAVES_N>(p, acc, (int64_t)(img * p.Ho + y0) * p.Wo + x0, p.Wo, rows_valid, cols_valid, wm, wn, tid, smem);
            else head_epilogue<BM, MT, NT, WM, WAVES_N>(p, acc, (int64_t)(img * p.Ho + y0) * p.Wo + x0, p.Wo, rows_valid, cols_valid, wm, wn, tid, smem);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int yy = y0 + wm * MT + i;
                const int row0 = (img * p.Ho + yy) * p.Wo + x0;
                // (the stages are free: every wave is past the last barrier of the K loop - 4 KiB of LDS scratch per wave)
                { epilogue_tile<SPLIT, EPI>(p, acc[i][j], row0, n0 + wn * WN + j * 32 + l31, lane, 0, yy < p.Ho ? row0 + cols_valid : row0, smem + wave * 4096); STA_EPI_TILE_FENCE(); }
            }
    }
    if (p.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0 && blockIdx.x < 2048) {
            unsigned long long* o = p.stamps + (size_t)blockIdx.x * 4;
            o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memrealtime();
        }
    }
}
