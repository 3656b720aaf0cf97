// Large-tile MFMA GEMM / implicit-GEMM conv for gfx950 (the throughput kernel family).
//
// Same math, epilogues and GemmParams as gemm.h; different machine mapping:
//  * BM x BN x 32 block tile (256x256, 256x128, 192x256, 192x128), 8 waves (512 threads).  The 192x128
//    tile needs 80 KiB of LDS and ~110 VGPRs, so TWO workgroups share a CU: one block's HBM-bound
//    epilogue and barrier bubbles are covered by the other's main loop (measured: the best family
//    in-model, sta_api.hip cost model).
//  * Operands go global -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no ds_write).
//    Activations and weights live in the K-tile-blocked layout  [K/32][rows][hi32|lo32]  (blk_off in
//    sta_common.h): one K tile of 8 consecutive rows is 1 KiB of CONTIGUOUS memory, i.e. every DMA
//    wave-instruction reads 8 full 128-B lines (row-major operands gave half-line 64-B pieces and
//    25 % less DMA throughput, round-1 micro-benchmark, now tools/gemm_tiles.py).
//  * LDS image (f16x3): rows of 128 B = [hi 32 halves | lo 32 halves]; the 16-B chunk index is XORed
//    with (row>>1)&7 so the fragment ds_read_b128 is bank-conflict free.  The DMA writes lane-linear,
//    so the XOR is applied to the per-lane SOURCE chunk (same involution on both sides).
//    (f16: rows of 64 B, chunk ^ (row>>2)&3, weights still read from the interleaved layout.)
//  * Two LDS stages: the DMA of tile k+1 overlaps the MFMAs of tile k; one vmcnt(0)+barrier per K tile.
//  * 3x3 conv taps outside the image read a zero page; the RCU's input ReLU is applied on fragments.
//  * Block id -> tile: bijective XCD remap + band-major order (4 M-tiles x all N-tiles) so the blocks
//    resident on one XCD share A / W panels in that XCD's L2.
#pragma once
#include "gemm.h"
#include <type_traits>

typedef short short8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

template <bool SPLIT, int BM, int BN>
constexpr int gemm2_smem_bytes(int nstg = 2) { return nstg * (SPLIT ? 2 : 1) * (BM + BN) * 64; }

// LDS byte offset of 16-B chunk `chunk` of tile row `row` (SPLIT: 8 chunks/row = 4 hi + 4 lo)
template <bool SPLIT>
__device__ __forceinline__ int lds2_off(int row, int chunk) {
    return SPLIT ? row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4) : row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

// ABL (bench-only ablation bit mask, 0 in the product): 1 = no DMA inside the K loop, 2 = no LDS fragment
// reads (MFMA on stale registers), 4 = no MFMA (fragments kept alive); barriers always stay.
// NSTG: LDS stages.  2 = issue tile k+1, compute k, drain, barrier.  >2 = ring: the DMA runs NSTG-1 K tiles
// ahead and stays in flight across the barrier (counted vmcnt + raw s_barrier in one asm statement) -
// no gain on the big throughput shapes (DMA-throughput bound) but it is what makes the small-M /
// split-K family (SLAM-scale GEMMs, a handful of K tiles per block) latency-tolerant.
// EPI_HEAD epilogue of one workgroup (BN == 128 == N: the tile holds all channels of its BM pixels; a wave holds 32 of them
// for MT x 32 pixels).  Per 32x32 accumulator tile: v = relu(acc + bias), the four head.4 partial dot products of the lane's
// channel, then a halving butterfly over the 32 channel lanes (62 shuffles for 64 (pixel, output) sums instead of 320);
// the four waves of a pixel row meet in LDS, and BM threads apply the activations and store pts / conf.
// Tile row r -> pixel pix_base + (r >> 5) * rstride + (r & 31), stored when (r >> 5) < rows_valid, (r & 31) < cols_valid and
// the pixel index is < p.M (linear tiles: rstride 32; halo tiles: rstride = image width, conv3h.h).
// NT accumulator tiles per wave along the channels (a lane then owns NT channels: their products are summed before the butterfly).
template <int BM, int MT, int NT, int WM, int WAVES_N>
__device__ __forceinline__ void head_epilogue(const GemmParams& p, const floatx16 (&acc)[MT][NT], int64_t pix_base, int rstride, int rows_valid,
                                              int cols_valid, int wm, int wn, int tid, char* smem) {
    float* red = reinterpret_cast<float*>(smem);                 // [WAVES_N][BM][4]
    const int lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    float bv[NT], w4[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = (wn * NT + j) * 32 + l31;
        bv[j] = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int o = 0; o < 4; ++o) w4[j][o] = p.hw4[o * 128 + col];
    }
    __syncthreads();                                             // every wave has left the operand stages
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float part[64];                                          // [r][o]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float v = fmaxf(acc[i][j][r] + bv[j], 0.f);
#pragma unroll
                for (int o = 0; o < 4; ++o) part[r * 4 + o] = j == 0 ? v * w4[j][o] : __builtin_fmaf(v, w4[j][o], part[r * 4 + o]);
            }
        }
#pragma unroll
        for (int m = 16, n = 64; m >= 1; m >>= 1, n >>= 1) {     // after the step a lane keeps n/2 sums: the upper half if its bit is set
            const bool up = (l31 & m) != 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                if (k < n / 2) {
                    const float send = up ? part[k] : part[k + n / 2];
                    const float keep = up ? part[k + n / 2] : part[k];
                    part[k] = keep + __shfl_xor(send, m);
                }
            }
        }
        // lane l31 now holds sums 2*l31, 2*l31+1 of [r][o]: r = l31 >> 1, o = (l31 & 1) * 2 + {0, 1}
        const int r = l31 >> 1, row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        *reinterpret_cast<float2*>(red + ((size_t)wn * BM + row) * 4 + (l31 & 1) * 2) = make_float2(part[0], part[1]);
    }
    __syncthreads();
    if (tid < BM) {
        const int64_t pix = pix_base + (int64_t)(tid >> 5) * rstride + (tid & 31);
        if (pix < p.M && (tid >> 5) < rows_valid && (tid & 31) < cols_valid) {
            float4 a = *reinterpret_cast<const float4*>(red + (size_t)tid * 4);
#pragma unroll
            for (int w = 1; w < WAVES_N; ++w) {
                const float4 b = *reinterpret_cast<const float4*>(red + ((size_t)w * BM + tid) * 4);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            const float x = a.x + p.hb4[0], y = a.y + p.hb4[1], z = a.z + p.hb4[2], c = a.w + p.hb4[3];
            const float d = sqrtf(x * x + y * y + z * z);
            const float sc = expm1f(d) / fmaxf(d, 1e-8f);
            const bool second = pix >= p.hsplit;
            const int64_t q = second ? pix - p.hsplit : pix;
            float* pts = (second ? p.hptsB : p.hptsA) + q * 3;
            pts[0] = x * sc; pts[1] = y * sc; pts[2] = z * sc;
            (second ? p.hconfB : p.hconfA)[q] = 1.0f + expf(c);
        }
    }
}

// EPI_HEAD for TRANSPOSED accumulators (conv3h.h, round 6): the halo kernel issues its MFMAs with the operands swapped when its
// epilogue is the fused DPT tail, so an accumulator tile holds (32 channels) x (32 pixels) - lane = pixel, register r = channel
// (r & 3) + 8 (r >> 2) + 4 lhi of the wave's 32-channel block, the same interleave the attention kernel's score tile has for its
// keys.  head.4 (1x1 conv 128 -> 4) is then ONE more contraction over the channels, on the matrix pipe: relu(acc + bias) becomes the
// B operand (k = channel, n = pixel; two fp16 planes, regrouped to the natural k order by a v_permlane32_swap per register pair
// exactly like P^T in attention.h), the A operand is head.4's weight [4 outputs, zero-padded to 32][32 channels] as fp16 hi / lo:
// 6 MFMAs per 32-pixel tile instead of 64 FMAs + a 62-shuffle butterfly on the accumulator-holding waves (7.4 us of a 51-us
// workgroup with nothing resident beside it to hide them).  Lanes 0-31 end up with the four partial outputs of pixel l31 over the
// wave's 32 channels; the four channel waves meet in LDS and BM threads apply the activations as before.
template <int BM, int MT, int WM, int WAVES_N>
__device__ __forceinline__ void head_epilogue_t(const GemmParams& p, const floatx16 (&acc)[MT][1], int64_t pix_base, int rstride, int rows_valid,
                                                int cols_valid, int wm, int wn, int tid, char* smem) {
    float* red = reinterpret_cast<float*>(smem);                 // [WAVES_N][BM][4]
    const int lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = p.bias ? p.bias[wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] : 0.f;
    // A fragments of head.4: lane l31 = output row (valid below 4), 8 consecutive channels 16 w + 8 lhi .. + 7 of the wave's block
    half8 wh[2], wl[2];
    {
        RangeAcc rw;
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = l31 < 4 ? p.hw4[l31 * 128 + wn * 32 + 16 * w + 8 * lhi + e] * p.hw4_scale[l31 & 3] : 0.f;       // (exact: a power of two into [0.5, 1))
                f16 h_, l_; split_f16(x, h_, l_, rw);
                wh[w][e] = h_; wl[w][e] = l_;
            }
    }
    __syncthreads();                                             // every wave has left the operand stages
    RangeAcc ra;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        floatx16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            union { half8 h; unsigned u[4]; } ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = fmaxf(acc[i][0][8 * w + e] + bv[8 * w + e], 0.f);
                f16 h_, l_; split_f16(v, h_, l_, ra);
                ph.h[e] = h_; pl.h[e] = l_;
            }
            // u[0..1] = registers 8w .. 8w+3 (channels 16w + 4 lhi + 0..3), u[2..3] = 8w+4 .. 8w+7 (16w + 8 + 4 lhi + 0..3): after the
            // swap half 0 holds channels 16w .. 16w+7 and half 1 holds 16w+8 .. 16w+15, in k order (attention.h, P^T)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                auto r1 = __builtin_amdgcn_permlane32_swap(ph.u[e], ph.u[2 + e], false, false);
                ph.u[e] = r1[0]; ph.u[2 + e] = r1[1];
                auto r2 = __builtin_amdgcn_permlane32_swap(pl.u[e], pl.u[2 + e], false, false);
                pl.u[e] = r2[0]; pl.u[2 + e] = r2[1];
            }
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[w], ph.h, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[w], pl.h, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[w], ph.h, o, 0, 0, 0);
        }
        // rows 0..3 of the result tile = registers 0..3 of lane half 0: the four outputs of pixel l31
        if (lhi == 0) *reinterpret_cast<float4*>(red + ((size_t)wn * BM + wm * WM + i * 32 + l31) * 4) =
            make_float4(o[0] / p.hw4_scale[0], o[1] / p.hw4_scale[1], o[2] / p.hw4_scale[2], o[3] / p.hw4_scale[3]);
    }
    ra.flush(p.range);
    __syncthreads();
    if (tid < BM) {
        const int64_t pix = pix_base + (int64_t)(tid >> 5) * rstride + (tid & 31);
        if (pix < p.M && (tid >> 5) < rows_valid && (tid & 31) < cols_valid) {
            float4 a = *reinterpret_cast<const float4*>(red + (size_t)tid * 4);
#pragma unroll
            for (int w = 1; w < WAVES_N; ++w) {
                const float4 b = *reinterpret_cast<const float4*>(red + ((size_t)w * BM + tid) * 4);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            const float x = a.x + p.hb4[0], y = a.y + p.hb4[1], z = a.z + p.hb4[2], c = a.w + p.hb4[3];
            const float d = sqrtf(x * x + y * y + z * z);
            const float sc = expm1f(d) / fmaxf(d, 1e-8f);
            const bool second = pix >= p.hsplit;
            const int64_t q = second ? pix - p.hsplit : pix;
            float* pts = (second ? p.hptsB : p.hptsA) + q * 3;
            pts[0] = x * sc; pts[1] = y * sc; pts[2] = z * sc;
            (second ? p.hconfB : p.hconfA)[q] = 1.0f + expf(c);
        }
    }
}

// Skinny tail of a dense GEMM (GemmParams::m_tail <= 32 rows after a whole number of BM-row tiles; the decoder's M = 2B x
// (768 patch tokens) + 2B pose tokens is 64 tiles of 192 rows + 16 rows).  A tile row of their own would cost every N tile a
// BM-row tile for 16 rows - and, worse, a whole extra round of the 512 resident slots (65 x 24 tiles = 3.05 rounds).
// Tail block tb owns the 32 columns [32 tb, 32 tb + 32): its NW waves split K between them, every wave runs one
// 32x32 MFMA tile straight from global memory (the operands are L2-hot: the main tiles stream the same weights), the
// partial tiles meet in LDS and wave 0 runs the ordinary epilogue.  ~2-4 us of work per block, dispatched first.
// MX (round 6, mlp.fc2 in the f16mx arithmetic): the operands are f16mx rows - per K tile two fp16 MFMAs on the hi halves + ONE
// block-scaled fp8 MFMA on the 32 pair bytes of the lane's half (bytes [64 + 32 lhi, 64 + 32 lhi + 32) of the row block: chunks
// 4 + 2 lhi, 5 + 2 lhi - exactly what compute_tile's MX branch reads from LDS).
template <bool SPLIT, int EPI, int NW, bool MX = false>
__device__ __forceinline__ void gemm2_tail(const GemmParams& p, const int tb, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int A_ES = SPLIT ? 64 : 32;
    const int row0 = p.M - p.m_tail, col0 = tb * 32;
    const int arow = row0 + l31 < p.M ? row0 + l31 : p.M - 1;
    const int brow = col0 + l31 < p.N ? col0 + l31 : p.N - 1;
    const f16* ap = p.A_hi + (size_t)arow * A_ES + lhi * 8;
    const f16* bp = p.B_hi + (size_t)brow * 64 + lhi * 8;
    const size_t a_kstride = (size_t)p.a_rp * A_ES, b_kstride = (size_t)p.N * 64;
    const int nkt_all = p.K / GEMM_BK;
    const int kt0 = (int)((int64_t)wave * nkt_all / NW), kt1 = (int)((int64_t)(wave + 1) * nkt_all / NW);
    struct Frag { uint4 ah[2], al[2], bh[2], bl[2]; };
    auto load = [&](int kt, Frag& f) {
        const f16* a = ap + kt * a_kstride;
        const f16* b = bp + kt * b_kstride;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f.ah[ks] = ldg16(a + ks * 16); f.bh[ks] = ldg16(b + ks * 16);
            if (MX) {           // al / bl: the lane half's 32 pair bytes (two 16-B chunks), ap / bp already carry + lhi * 8 elements
                f.al[ks] = ldg16(a - lhi * 8 + 32 + lhi * 16 + ks * 8); f.bl[ks] = ldg16(b - lhi * 8 + 32 + lhi * 16 + ks * 8);
            } else if (SPLIT) { f.al[ks] = ldg16(a + 32 + ks * 16); f.bl[ks] = ldg16(b + 32 + ks * 16); }
        }
    };
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto mma = [&](const Frag& f) {
        if constexpr (MX) {
            typedef int int8v __attribute__((ext_vector_type(8)));
            union U8 { struct { uint4 x, y; } q; int8v v; } a8, b8;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                H8 ah, bh; ah.u = f.ah[ks]; bh.u = f.bh[ks];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bh.h, acc, 0, 0, 0);
            }
            a8.q.x = f.al[0]; a8.q.y = f.al[1]; b8.q.x = f.bl[0]; b8.q.y = f.bl[1];
            constexpr int sc_a = 127 - STA_MX_A_SLO, sc_b = 127 - STA_MX_W_SHI;
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8.v, b8.v, acc, 1 /* A: e5m2 */, 0 /* B: e4m3 */, 0, sc_a, 0, sc_b);
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            H8 ah, al, bh, bl; ah.u = f.ah[ks]; bh.u = f.bh[ks];
            if (SPLIT) {
                al.u = f.al[ks]; bl.u = f.bl[ks];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, bh.h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bl.h, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bh.h, acc, 0, 0, 0);
        }
    };
    // one K tile in flight per wave (32 VGPRs of fragments: the tail must stay under the register budget of the main loop it
    // shares a kernel with - two 192x128 workgroups per CU need <= 128); the NW waves of the block and the other resident
    // blocks cover the load latency
#pragma unroll 1
    for (int kt = kt0; kt < kt1; ++kt) {
        Frag f;
        load(kt, f);
        mma(f);
    }
    float* red = reinterpret_cast<float*>(smem);    // [NW][16][64]
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 1
    for (int w = 1; w < NW; ++w) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(w * 16 + r) * 64 + lane];
    }
    epilogue_tile<SPLIT, EPI>(p, acc, row0, col0 + l31, lane, 0);
}

#ifndef STA_RING_ABL
#define STA_RING_ABL 0      // probe builds (tools/ring_ablate.py): the ablation mask below applied to the ring family inside the product flow
#endif
template <bool SPLIT, int AMODE, int EPI, int BM, int BN, int WAVES_M, int WAVES_N, int ABL_ = 0, int NSTG = 2, bool MX = false>
__device__ __forceinline__ void gemm2_body(const GemmParams& p, const int block_id_in) {
    constexpr int ABL = ABL_ | (NSTG > 2 ? STA_RING_ABL : 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MT = WM / 32, NT = WN / 32;
    constexpr int RB = SPLIT ? 128 : 64;                     // LDS row bytes
    constexpr int RPS = 1024 / RB;                           // rows per 1-KiB DMA slot (8 or 16)
    constexpr int CPR = RB / 16;                             // 16-B chunks per row (8 or 4)
    constexpr int A_TILE = BM * RB, B_TILE = BN * RB, STAGE = A_TILE + B_TILE;
    constexpr int NSA = BM / RPS, NSB = BN / RPS;
    constexpr int SA = (NSA + NW - 1) / NW, SB = (NSB + NW - 1) / NW;
    constexpr int A_ES = SPLIT ? 64 : 32;                    // global elements per activation row block
    static_assert(BM % 32 == 0 && BN % 32 == 0 && WM % 32 == 0 && WN % 32 == 0, "tile/wave mismatch");
    // fused DPT tail (EPI_HEAD, BN == 128 == N, one 32-channel block per wave): the MFMAs run with their operands swapped, the
    // accumulator tiles come out TRANSPOSED (lane = pixel, register = channel) - what head_epilogue_t contracts on the matrix pipe
    constexpr bool TRN = EPI == EPI_HEAD && BN == 128 && NT == 1;

    // skinny tail blocks come first in the grid (launch_gemm2 adds them): short, they overlap the first round of tiles
    int block_id = block_id_in;
    constexpr bool HAS_TAIL = AMODE == A_DENSE && NSTG == 2 && ABL_ == 0 && BM >= 192 &&
                              (EPI == EPI_F32 || EPI == EPI_F32R || ((EPI == EPI_GELU || EPI == EPI_QKV) && !MX));
    if constexpr (HAS_TAIL) {
        if (p.m_tail > 0) {
            const int ntail = (p.N + 31) >> 5;
            if (block_id < ntail) { gemm2_tail<SPLIT, EPI, NW, MX>(p, block_id, smem); return; }
            block_id -= ntail;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lhi = lane >> 5;
    unsigned long long clk0 = 0, rt0 = 0;
    if (p.clk_dbg) { clk0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.stamps) st0 = __builtin_amdgcn_s_memrealtime();

    // ---- block id -> (tile, K slice) (XCD-aware, band-major; the slices of one tile are consecutive ids)
    const int tiles_m = (p.M - (HAS_TAIL ? p.m_tail : 0) + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int nwg = tiles_m * tiles_n * ksplit;
    int t;
    {
        const int bid = block_id, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int kslice = t % ksplit;
    t /= ksplit;
    int bm, bn;
    {
        const int band = t / (4 * tiles_n);
        const int hb = tiles_m - band * 4 < 4 ? tiles_m - band * 4 : 4;
        const int local = t - band * 4 * tiles_n;
        bm = band * 4 + local % hb;
        bn = local / hb;
    }
    const int m0 = bm * BM, n0 = bn * BN;

    // ---- per-lane DMA source bookkeeping.  Lane l of a slot fills LDS row (l / CPR), chunk (l % CPR);
    // it must fetch source chunk (l % CPR) ^ swizzle(row).
    const int row_in = lane / CPR, c_lds = lane % CPR;
    unsigned a_src[SA], b_src[SB];         // BYTE offsets of this lane's 16 B within K tile 0 (32-bit: the K-tile base is
                                           // wave-uniform, so the DMA uses the SGPR-base + VGPR-offset address form)
    // 3x3 taps: per slot the pixel index of tap (0,0) (may lie outside the image), a 9-bit mask of the taps that fall inside it,
    // and the lane's byte offset inside a row block - tap t of channel block cb is then ONE 32-bit add away (round 4: the
    // address of every tap used to be rebuilt from (image, y, x) with two exec-masked bounds branches and 64-bit multiplies per
    // slot plus a scalar division per K tile: ~400 issue cycles per K tile next to 512 cycles of f16mx MFMAs)
    int cv_pix0[SA]; unsigned cv_mask[SA], cv_coff[SA];
#pragma unroll
    for (int s = 0; s < SA; ++s) {
        const int row = RPS * (wave + NW * s) + row_in;                      // tile row
        const int sw = SPLIT ? (row >> 1) & 7 : (row >> 2) & 3;
        const int chunk = c_lds ^ sw;                                         // source chunk of the row block
        const int gm = m0 + row;
        if (AMODE == A_DENSE) {
            const int gmc = gm < p.M ? gm : p.M - 1;
            a_src[s] = ((unsigned)gmc * A_ES + chunk * 8) * 2u;
        } else {
            const bool rok = gm < p.M;
            const int gmc = rok ? gm : 0;
            const int hw = p.Ho * p.Wo;
            const int img = gmc / hw;
            const int rem = gmc - img * hw;
            const int y0 = (rem / p.Wo) * p.cstride - 1, x0 = (rem % p.Wo) * p.cstride - 1;
            cv_pix0[s] = (img * p.Hi + y0) * p.Wi + x0;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yi = y0 + t / 3, xi = x0 + t % 3;
                if (rok && yi >= 0 && yi < p.Hi && xi >= 0 && xi < p.Wi) mk |= 1u << t;
            }
            cv_mask[s] = mk;
            cv_coff[s] = (unsigned)chunk * 16u;
            a_src[s] = 0;
        }
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        const int row = RPS * (wave + NW * s) + row_in;
        const int sw = SPLIT ? (row >> 1) & 7 : (row >> 2) & 3;
        const int gn = n0 + row;
        const int gnc = gn < p.N ? gn : p.N - 1;
        b_src[s] = ((unsigned)gnc * 64 + (c_lds ^ sw) * 8) * 2u;             // weights are always [hi32|lo32]
    }
    const size_t a_kstride = (size_t)p.a_rp * A_ES, b_kstride = (size_t)p.N * 64;

    int cv_tap = 0, cv_cb = 0;             // conv: (tap, channel block) of the NEXT K tile to be issued (set below, once kt0 is known)
    // operand bases of the NEXT K tile to be issued (tiles are issued in increasing kt): advanced by one K tile per call instead of
    // a 64-bit multiply per operand and tile
    const char* a_run = nullptr; const char* b_run = nullptr;
    auto issue_tile = [&](int kt, int stage) {
        char* sA = smem + stage * STAGE;
        char* sB = sA + A_TILE;
        // conv: K tile kt = (tap, channel block); the tiles are issued in increasing kt, so the pair is carried as a counter
        // (cv_tap / cv_cb, set for kt0 before the first call) instead of being re-derived by a division per call
        int tap_off = 0; const char* a_cb = nullptr;
        if (AMODE == A_CONV3) {
            const int ky = cv_tap >= 6 ? 2 : (cv_tap >= 3 ? 1 : 0), kx = cv_tap - 3 * ky;
            tap_off = ky * p.Wi + kx;
            a_cb = reinterpret_cast<const char*>(p.A_hi) + (size_t)cv_cb * p.a_rp * (A_ES * 2);
        }
#pragma unroll
        for (int s = 0; s < SA; ++s) {
            if (NSA % NW != 0 && wave + NW * s >= NSA) continue;
            char* dst = sA + (wave + NW * s) * 1024;
            if (AMODE == A_DENSE) {
                unsigned o = a_src[s];
                asm volatile("" : "+v"(o));      // opaque: keeps the 32-bit offset form (hipcc would hoist base + offset into a VGPR pair)
                glds16(a_run + o, dst);
            } else {
                const unsigned off = (unsigned)(cv_pix0[s] + tap_off) * (unsigned)(A_ES * 2) + cv_coff[s];     // (pixels x 128 B < 2^32: checked by the launcher)
                const bool ok = (cv_mask[s] >> cv_tap) & 1u;
                glds16(ok ? static_cast<const void*>(a_cb + off) : static_cast<const void*>(p.zero_page), dst);
            }
        }
        if (AMODE == A_CONV3) { if (++cv_cb == (p.Cin >> 5)) { cv_cb = 0; ++cv_tap; } }
#pragma unroll
        for (int s = 0; s < SB; ++s) {
            if (NSB % NW != 0 && wave + NW * s >= NSB) continue;
            unsigned o = b_src[s];
            asm volatile("" : "+v"(o));
            glds16(b_run + o, sB + (wave + NW * s) * 1024);
        }
        a_run += a_kstride * 2; b_run += b_kstride * 2;
    };

    floatx16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // K tiles of this block's slice
    const int nkt_all = p.K / GEMM_BK;
    const int kt0 = (int)((int64_t)kslice * nkt_all / ksplit);
    const int nkt = (int)((int64_t)(kslice + 1) * nkt_all / ksplit) - kt0;
    if (AMODE == A_CONV3) { const int cblocks = p.Cin >> 5; cv_tap = kt0 / cblocks; cv_cb = kt0 - cv_tap * cblocks; }
    a_run = reinterpret_cast<const char*>(p.A_hi + kt0 * a_kstride); b_run = reinterpret_cast<const char*>(p.B_hi + kt0 * b_kstride);
    constexpr bool RING = NSTG > 2;
    constexpr int GPW = SA + SB;                   // DMA instructions per wave per K tile
    static_assert(!RING || (NSA % NW == 0 && NSB % NW == 0), "ring needs the same DMA count in every wave");
    static_assert(NSTG >= 2 && NSTG <= 5, "2..5 stages");
    if (RING) {
#pragma unroll
        for (int s0 = 0; s0 < NSTG - 1; ++s0)
            if (s0 < nkt && !(ABL & 1)) issue_tile(kt0 + s0, s0);
    } else {
        issue_tile(kt0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (p.stamps) st1 = __builtin_amdgcn_s_memrealtime();
    }

    // STAGGER (ABL bit 3, experiment): the two waves that share a SIMD (w and w + NW/2) issue their DMA at
    // different points of the K tile - the older half before K step 0, the younger half between the two
    // K steps - so that one of them always has MFMAs to feed the matrix pipe while the other sits in the
    // (slow, back-pressured) LDS-DMA issue.
    const bool late_dma = (ABL & 8) && !RING && wave >= NW / 2;
    // the arithmetic of ONE K tile held in LDS stage `cur` (kt: its index in this block's slice, for the STAGGER experiment)
    // relu_c: the input ReLU of resConfUnit*.conv1 (implicit-GEMM convolutions) as a COMPILE-time constant - the main loop below
    // exists once per value (round 6: as a wave-uniform runtime flag it was a branch per fragment inside the K tile, and hipcc
    // schedules nothing across those basic-block boundaries)
    auto compute_tile = [&](const int cur, const int kt, auto relu_c) {
        constexpr bool RELU = AMODE == A_CONV3 && decltype(relu_c)::value;
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + A_TILE;
        if (MX) {
            // precision f16mx (sta_common.h): per K tile and accumulator 2 f16 MFMAs (hi x hi, K steps 0 / 1) + ONE
            // block-scaled fp8 MFMA that carries both correction products.  Operand semantics of
            // v_mfma_scale_f32_32x32x64_f8f6f4 as decoded on hardware (tools/probes/mx_probe*.hip): byte q of lane
            // (row, half) of A meets byte q of lane (col, half) of B; scale block b = bytes [16b, 16b+16) of both lane
            // halves, scaled by the E8M0 byte of lane (row, half b).  The second half of every row block holds the
            // (hi8, lo8) / (lo8, hi8) byte pairs, so a lane's operand is simply 32 contiguous bytes = 16 k, and all
            // blocks carry the one combined scale 2^-15.
            typedef int int4v __attribute__((ext_vector_type(4)));
            typedef int int8v __attribute__((ext_vector_type(8)));
            half8 ah[MT], bh[NT];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int chunk = ks * 2 + lhi;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    ah[i] = *reinterpret_cast<const half8*>(sA + lds2_off<true>(wm * WM + i * 32 + l31, chunk));
                    if (RELU) {      // relu on the f16 hi fragment (packed sign masks, see the f16x3 path)
                        union { half8 h; unsigned u[4]; } t; t.h = ah[i];
#pragma unroll
                        for (int w = 0; w < 4; ++w) { const unsigned sgn = (t.u[w] >> 15) & 0x00010001u; t.u[w] &= ~((sgn << 16) - sgn); }
                        ah[i] = t.h;
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) bh[j] = *reinterpret_cast<const half8*>(sB + lds2_off<true>(wn * WN + j * 32 + l31, chunk));
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
                const int ra = wm * WM + i * 32 + l31;
                a8[i].q.x = *reinterpret_cast<const int4v*>(sA + lds2_off<true>(ra, 4 + 2 * lhi));
                a8[i].q.y = *reinterpret_cast<const int4v*>(sA + lds2_off<true>(ra, 5 + 2 * lhi));
                if (RELU) {          // a pair (hi8, lo8) is 16 bits: zero it when hi8 is negative
#pragma unroll
                    for (int w = 0; w < 8; ++w) { const unsigned u = (unsigned)a8[i].v[w]; const unsigned sgn = (u >> 7) & 0x00010001u; a8[i].v[w] = (int)(u & ~((sgn << 16) - sgn)); }
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int rb = wn * WN + j * 32 + l31;
                b8[j].q.x = *reinterpret_cast<const int4v*>(sB + lds2_off<true>(rb, 4 + 2 * lhi));
                b8[j].q.y = *reinterpret_cast<const int4v*>(sB + lds2_off<true>(rb, 5 + 2 * lhi));
            }
            constexpr int sc_a = 127 - STA_MX_A_SLO, sc_b = 127 - STA_MX_W_SHI;       // 2^-11 * 2^-4 = the shared 2^-15
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = TRN ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[j].v, a8[i].v, acc[i][j], 0 /* A: e4m3 weights */, 1 /* B: e5m2 */, 0, sc_b, 0, sc_a)
                                    : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i].v, b8[j].v, acc[i][j], 1 /* A: e5m2 */, 0 /* B: e4m3 */, 0, sc_a, 0, sc_b);
            return;
        }
        half8 a_hi[MT], a_lo[MT], b_hi[NT], b_lo[NT];
        if (ABL & 2) {
#pragma unroll
            for (int i = 0; i < MT; ++i) { a_hi[i] = (half8)(f16)(0.001f * (lane + i)); a_lo[i] = a_hi[i]; asm volatile("" : "+v"(a_hi[i]), "+v"(a_lo[i])); }
#pragma unroll
            for (int j = 0; j < NT; ++j) { b_hi[j] = (half8)(f16)(0.002f * (lane + j)); b_lo[j] = b_hi[j]; asm volatile("" : "+v"(b_hi[j]), "+v"(b_lo[j])); }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 2 + lhi;
#pragma unroll
            for (int i = 0; i < MT && !(ABL & 2); ++i) {
                const int ra = wm * WM + i * 32 + l31;
                a_hi[i] = *reinterpret_cast<const half8*>(sA + lds2_off<SPLIT>(ra, chunk));
                if (SPLIT) a_lo[i] = *reinterpret_cast<const half8*>(sA + lds2_off<SPLIT>(ra, 4 + chunk));
                if (AMODE == A_CONV3) {
                    if (RELU) {   // relu(hi + lo): the sign of hi decides.  Packed-half integer form, 5 VALU per
                        // 32-bit word for both planes (the vector compare scalarises to ~11 per word):
                        // s = sign bits at bit 0 / 16, m = 0xFFFF in every negative half, x &= ~m
                        union { half8 h; unsigned u[4]; } ah, al;
                        ah.h = a_hi[i]; al.h = a_lo[i];
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const unsigned sgn = (ah.u[w] >> 15) & 0x00010001u;
                            const unsigned m = (sgn << 16) - sgn;        // 0xFFFF per set sign (mod 2^32)
                            ah.u[w] &= ~m;
                            if (SPLIT) al.u[w] &= ~m;
                        }
                        a_hi[i] = ah.h; if (SPLIT) a_lo[i] = al.h;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NT && !(ABL & 2); ++j) {
                const int rb = wn * WN + j * 32 + l31;
                b_hi[j] = *reinterpret_cast<const half8*>(sB + lds2_off<SPLIT>(rb, chunk));
                if (SPLIT) b_lo[j] = *reinterpret_cast<const half8*>(sB + lds2_off<SPLIT>(rb, 4 + chunk));
            }
            if (ABL & 4) {      // keep the fragment loads alive, skip the matrix work
#pragma unroll
                for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(a_hi[i]), "v"(a_lo[i]));
#pragma unroll
                for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(b_hi[j]), "v"(b_lo[j]));
                continue;
            }
            // product-major order: consecutive MFMAs hit different accumulators (MT*NT apart)
            if (SPLIT && (ABL & 16)) {
                // bench-only what-if (results are NOT a GEMM): the two correction products of both K steps replaced by
                // ONE block-scaled fp8 MFMA (32x32x64, 2x rate) per accumulator per K tile - the instruction mix of an
                // "f16 + MX-fp8 corrections" scheme, to price it before building it (operands: whatever bits are there)
                if (ks == 1) {
                    typedef int int8v __attribute__((ext_vector_type(8)));
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            union { struct { half8 x, y; } h; int8v v; } ua, ub;
                            ua.h.x = a_lo[i]; ua.h.y = a_hi[i]; ub.h.x = b_hi[j]; ub.h.y = b_lo[j];
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ua.v, ub.v, acc[i][j], 0, 0, 0, 0x7f, 0, 0x7f);
                        }
                }
            } else if (SPLIT) {
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
            if ((ABL & 8) && ks == 0 && late_dma && kt + 1 < nkt) issue_tile(kt0 + kt + 1, cur ^ 1);
        }
    };
    auto main_loop = [&](auto relu_c) {
    if (RING) {
        // Steady state: every step waits for the same number of younger tiles and issues one more - a fixed wait immediate, an
        // unconditional issue, the stage indices as wrapping counters.  (As ONE loop with the wait count, the issue condition and
        // kt % NSTG decided per step the control flow alone was ~10 branch instructions per K tile, several of them taken - part
        // of the 0.12 us a K tile costs with the DMA, the fragment reads and the MFMAs ablated, tools/ring_ablate.py.)  The
        // last NSTG - 1 tiles (nothing left to issue, fewer tiles outstanding) are peeled.
        int cur = 0, nxt = NSTG - 1, kt = 0;
        for (; kt + NSTG - 1 < nkt; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NSTG - 2) * GPW) : "memory");      // tile kt landed; every wave finished tile kt-1
            if (p.stamps && kt == 0) st1 = __builtin_amdgcn_s_memrealtime();
            if (!(ABL & 1)) issue_tile(kt0 + kt + NSTG - 1, nxt);                                    // into the stage tile kt-1 left
            compute_tile(cur, kt, relu_c);
            cur = cur + 1 == NSTG ? 0 : cur + 1;
            nxt = nxt + 1 == NSTG ? 0 : nxt + 1;
        }
        for (; kt < nkt; ++kt) {
            const int rem = nkt - 1 - kt;                                  // younger tiles of this wave still outstanding: 0 .. NSTG-2
            if (rem == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(GPW) : "memory");
            else if (rem == 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * GPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * GPW) : "memory");
            if (p.stamps && kt == 0) st1 = __builtin_amdgcn_s_memrealtime();
            compute_tile(cur, kt, relu_c);
            cur = cur + 1 == NSTG ? 0 : cur + 1;
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt && !(ABL & 1) && !late_dma) issue_tile(kt0 + kt + 1, cur ^ 1);
            compute_tile(cur, kt, relu_c);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    };
    if constexpr (AMODE == A_CONV3) {
        if (p.relu_in) main_loop(std::integral_constant<bool, true>{});
        else main_loop(std::integral_constant<bool, false>{});
    } else main_loop(std::integral_constant<bool, false>{});

    if (p.stamps) {      // after the LAST MFMA has delivered (the stamp is scalar code: without the data dependence it is scheduled early)
        asm volatile("" ::"v"(acc[MT - 1][NT - 1][15]), "v"(acc[0][0][0]) : "memory");
        st2 = __builtin_amdgcn_s_memrealtime();
    }
    if constexpr (EPI == EPI_HEAD) {
        if constexpr (BN == 128) {
            if constexpr (TRN) head_epilogue_t<BM, MT, WM, WAVES_N>(p, acc, m0, 32, BM / 32, 32, wm, wn, tid, smem);
            else head_epilogue<BM, MT, NT, WM, WAVES_N>(p, acc, m0, 32, BM / 32, 32, wm, wn, tid, smem);
        }
    } else {
        // EPI_QKV / EPI_GELU / EPI_F16: a private LDS scratch per wave for the epilogue's transposes (the stage buffers are free)
        // (two stages: every wave is past the last barrier of the main loop.  The ring form has no barrier behind its last K tile:
        //  one here, so that no wave still reads fragments from the memory another wave's epilogue is about to reuse)
        char* const wave_lds = ((EPI == EPI_QKV || EPI == EPI_GELU || EPI == EPI_F16) && NW * EPI_LDS_BYTES <= NSTG * STAGE) ? smem + wave * EPI_LDS_BYTES : nullptr;
        if (RING && wave_lds != nullptr) __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                { epilogue_tile<SPLIT, EPI>(p, acc[i][j], m0 + wm * WM + i * 32, n0 + wn * WN + j * 32 + l31, lane, kslice, -1, wave_lds); if (EPI == EPI_F16 || EPI == EPI_CONVT) STA_EPI_TILE_FENCE(); }
    }
    if (p.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the epilogue's stores acknowledged
        if (tid == 0 && block_id_in < 2048) {            // (the in-model dump keeps 2048 workgroups per launch)
            unsigned long long* o = p.stamps + (size_t)block_id_in * 4;
            o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memrealtime();
        }
    }
    if (p.clk_dbg && tid == 0 && (block_id & 63) == 0) {      // effective shader clock = cycles / (ticks / 100 MHz)
        atomicAdd(p.clk_dbg, __builtin_readcyclecounter() - clk0);
        atomicAdd(p.clk_dbg + 1, __builtin_amdgcn_s_memrealtime() - rt0);
    }
}

template <bool SPLIT, int AMODE, int EPI, int BM, int BN, int WAVES_M, int WAVES_N, int ABL = 0, int NSTG = 2, bool MX = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm2_kernel(const GemmParams p) {
    gemm2_body<SPLIT, AMODE, EPI, BM, BN, WAVES_M, WAVES_N, ABL, NSTG, MX>(p, blockIdx.x);
}

// Two independent GEMMs of the same tile family in ONE launch: blocks [0, tiles_a) work on `pa`, the rest on `pb`.
// Used for the decoder's attn.qkv (on norm1(x)) and cross_attn.projk|projv (on norm_y(other side)), which depend only on
// the layer input: 1170 + 780 tiles of 192x128 are 3 + 2 rounds of the 512 resident slots as two launches, 4 as one.
template <bool SPLIT, int AMODE, int EPI, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm2_pair_kernel(const GemmParams pa, const GemmParams pb, const int tiles_a) {
    if ((int)blockIdx.x < tiles_a) gemm2_body<SPLIT, AMODE, EPI, BM, BN, WAVES_M, WAVES_N>(pa, blockIdx.x);
    else gemm2_body<SPLIT, AMODE, EPI, BM, BN, WAVES_M, WAVES_N>(pb, blockIdx.x - tiles_a);
}
