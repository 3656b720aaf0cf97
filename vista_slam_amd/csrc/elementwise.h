// HBM-bound helper kernels of the STA path (gfx950): LayerNorm -> fp16 planes, patch gather,
// token/plane conversion, bilinear x2 (align_corners), final 1x1 conv + pointmap postprocess,
// pose head (MLP + 3x3 polar rotation), curope-compatible in-place RoPE, weight repacking.
// All reductions are wave64 shuffles; all loads/stores are 8-16 B per lane.
#pragma once
#include "sta_common.h"

// ---------------------------------------------------------------------------------------------
// LayerNorm (eps inside sqrt, affine) over fp32 rows -> fp16 hi/lo planes; up to two affine sets
// from one read of x (decoder norm1 + norm_y share their input: sta_blocks.py:226-229).
// One wave per row, C <= 1024, C % 4 == 0.
struct LnParams {
    const float* x; int ldx; int M; int C; float eps;
    const float* g1; const float* b1; f16* o1_hi; f16* o1_lo;
    const float* g2; const float* b2; f16* o2_hi; f16* o2_lo;   // optional (g2 == nullptr)
    float* o32; int ldo32;                                       // optional fp32 output with set 1
    // optional first half of a slab split-K residual GEMM (GemmParams::slab): x[row] += sum_s slab[s][row]; x is rewritten,
    // then normalised as usual (g1 == nullptr: only the add)
    const float* slab; int nslab; float* xw;
    unsigned long long* range;      // the handle's range counters (sta_common.h)
};

// normalised values n[4] of columns idx..idx+3 of `row` -> affine set 1 (fp32 copy and / or planes) and optional set 2
// The affine parameters come in registers (LnAffine, loaded by the caller together with the row, BEFORE its first store: a load
// issued behind a store waits for that store's acknowledgement - loads and stores retire through one in-order vmcnt on gfx9).
struct LnAffine { float4 g1, b1, g2, b2; };
__device__ __forceinline__ LnAffine ln_load_affine(const LnParams& p, int idx) {
    LnAffine a;
    a.g1 = *reinterpret_cast<const float4*>(p.g1 + idx); a.b1 = *reinterpret_cast<const float4*>(p.b1 + idx);
    if (p.g2) { a.g2 = *reinterpret_cast<const float4*>(p.g2 + idx); a.b2 = *reinterpret_cast<const float4*>(p.b2 + idx); }
    else { a.g2 = a.g1; a.b2 = a.b1; }
    return a;
}
template <bool SPLIT>
__device__ __forceinline__ void ln_store4(const LnParams& p, int row, int idx, const float n[4], const LnAffine& af) {
    RangeAcc ra;       // never flushed (dead code): a normalised row times the gains cannot leave the fp16 range; what can go wrong is a
                       // non-finite ROW (inf / NaN in the residual stream), which ln_kernel / resid_ln_kernel count from the row statistics
    {
        const float4 g = af.g1, b = af.b1;
        float y[4] = {n[0] * g.x + b.x, n[1] * g.y + b.y, n[2] * g.z + b.z, n[3] * g.w + b.w};
        if (p.o32) *reinterpret_cast<float4*>(p.o32 + (size_t)row * p.ldo32 + idx) = make_float4(y[0], y[1], y[2], y[3]);
        if (p.o1_hi) {
            const size_t o = blk_off<SPLIT>(row, idx, p.M);
            H4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { if (SPLIT) split_f16(y[e], h.e[e], l.e[e], ra); else h.e[e] = to_f16_sat(y[e], ra); }
            *reinterpret_cast<uint2*>(p.o1_hi + o) = h.u;
            if (SPLIT) *reinterpret_cast<uint2*>(p.o1_hi + o + 32) = l.u;
        }
    }
    if (p.g2) {
        const float4 g = af.g2, b = af.b2;
        float y[4] = {n[0] * g.x + b.x, n[1] * g.y + b.y, n[2] * g.z + b.z, n[3] * g.w + b.w};
        const size_t o = blk_off<SPLIT>(row, idx, p.M);
        H4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { if (SPLIT) split_f16(y[e], h.e[e], l.e[e], ra); else h.e[e] = to_f16_sat(y[e], ra); }
        *reinterpret_cast<uint2*>(p.o2_hi + o) = h.u;
        if (SPLIT) *reinterpret_cast<uint2*>(p.o2_hi + o + 32) = l.u;
    }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void ln_kernel(const LnParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const float* xr = p.x + (size_t)row * p.ldx;
    float4 v[4];
    LnAffine af[4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int idx = (i * 64 + lane) * 4;
        if (idx < p.C) {
            v[i] = *reinterpret_cast<const float4*>(xr + idx);
            af[i] = ln_load_affine(p, idx);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(sum) / (float)p.C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int idx = (i * 64 + lane) * 4;
        if (idx < p.C) {
            float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)p.C + p.eps);
    if (lane == 0 && !(fabsf(mean) <= 3.0e38f && rstd <= 3.0e38f && rstd > 0.f)) atomicAdd(p.range, 1ull);   // non-finite row, or a variance that overflowed fp32 (rstd == 0: the row would silently become pure bias)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int idx = (i * 64 + lane) * 4;
        if (idx < p.C) {
            float n[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd};
            ln_store4<SPLIT>(p, row, idx, n, af[i]);
        }
    }
}

// Second half of a slab split-K residual GEMM (GemmParams::slab) + the LayerNorm that follows it (small-M regime):
// x[row] += sum_s slab[s][row] (slice 0 carries the bias), x is rewritten, then normalised like ln_kernel (g1 == nullptr:
// only the add).  One block per row, one float4 per thread (C <= 1024): 1 + nslab independent loads per thread.
template <bool SPLIT>
__global__ __launch_bounds__(256) void resid_ln_kernel(const LnParams p) {
    const int row = blockIdx.x, idx = threadIdx.x * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool on = idx < p.C;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    LnAffine af;
    if (on && p.g1) af = ln_load_affine(p, idx);
    if (on) {
        v = *reinterpret_cast<const float4*>(p.x + (size_t)row * p.ldx + idx);
        for (int s = 0; s < p.nslab; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(p.slab + ((size_t)s * p.M + row) * p.C + idx);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        *reinterpret_cast<float4*>(p.xw + (size_t)row * p.ldx + idx) = v;
    }
    if (!p.g1) return;
    __shared__ float red[2][4];
    float sum = wave_sum((v.x + v.y) + (v.z + v.w));
    if (lane == 0) red[0][wave] = sum;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)p.C;
    float sq = 0.f;
    if (on) { float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean; sq = (a * a + b * b) + (c * c + d * d); }
    sq = wave_sum(sq);
    if (lane == 0) red[1][wave] = sq;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)p.C + p.eps);
    if (threadIdx.x == 0 && !(fabsf(mean) <= 3.0e38f && rstd <= 3.0e38f && rstd > 0.f)) atomicAdd(p.range, 1ull);   // non-finite row, or a variance that overflowed fp32 (rstd == 0: the row would silently become pure bias)
    if (on) {
        float n[4] = {(v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd};
        ln_store4<SPLIT>(p, row, idx, n, af);
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 rows [nb, rows, C] (batch stride bstride floats) -> contiguous fp16 planes [nb*rows, C].
template <bool SPLIT>
__global__ void rows_to_planes_kernel(const float* x, int64_t bstride, int rows, int C, int64_t total4,
                                      f16* o_hi, f16* o_lo, int64_t obstride /* output batch stride in rows; 0 = rows */,
                                      int64_t orows /* rows of the blocked output planes; 0 = row-major [.,C] (Q/K buffers) */,
                                      int mx, /* blocked output in the f16mx row format */ unsigned long long* rng) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    const int c4 = C / 4;
    for (; i < total4; i += step) {
        int64_t r = i / c4; int c = (int)(i - r * c4) * 4;
        int64_t b = r / rows; int rr = (int)(r - b * rows);
        float4 v = *reinterpret_cast<const float4*>(x + b * bstride + (int64_t)rr * C + c);
        float y[4] = {v.x, v.y, v.z, v.w};
        H4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { if (SPLIT) split_f16(y[e], h.e[e], l.e[e], rng); else h.e[e] = to_f16_sat(y[e], rng); }
        const int64_t orow = obstride ? b * obstride + rr : r;
        if (SPLIT && mx && orows) {
            store_mx4(o_hi, blk_off<SPLIT>(orow, c, orows), split_mx4<false>(y, rng));
        } else if (orows) {
            const size_t o = blk_off<SPLIT>(orow, c, orows);
            *reinterpret_cast<uint2*>(o_hi + o) = h.u;
            if (SPLIT) *reinterpret_cast<uint2*>(o_hi + o + 32) = l.u;
        } else {
            *reinterpret_cast<uint2*>(o_hi + orow * C + c) = h.u;
            if (SPLIT) *reinterpret_cast<uint2*>(o_lo + orow * C + c) = l.u;
        }
    }
}

// planes [nb, rows(+pad), C] -> fp32 [nb, rows, C]  (test/debug taps only)
__global__ void planes_to_f32_kernel(const f16* hi, const f16* lo, int64_t ibstride_rows, int rows, int C,
                                     int64_t total, float* out, int64_t irows /* blocked planes with irows rows; 0 = row-major */,
                                     int mx = 0 /* blocked f16mx rows: value = hi + lo8 * 2^-11 */) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        int64_t r = i / C; int c = (int)(i - r * C);
        int64_t b = r / rows; int rr = (int)(r - b * rows);
        const int64_t srow = ibstride_rows ? b * ibstride_rows + rr : r;
        const size_t src = irows ? (lo ? blk_off<true>(srow, c, irows) : blk_off<false>(srow, c, irows)) : (size_t)(srow * C + c);
        float v = (float)hi[src];
        if (mx && lo) v = load_mx_act(hi, src); else if (lo) v += (float)lo[src];
        out[i] = v;
    }
}

// fp32 V [nb, rows, 64] -> transposed planes Vt [nb, 64, npad]  (test/debug only)
__global__ void pack_vt_kernel(const float* v, int nb, int rows, int npad, f16* hi, f16* lo, unsigned long long* rng) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)nb * rows * 64;
    if (i >= total) return;
    int d = (int)(i % 64); int64_t t = i / 64; int r = (int)(t % rows); int64_t b = t / rows;
    f16 h, l; split_f16(v[i], h, l, rng);
    int64_t o = (b * 64 + d) * npad + r;
    hi[o] = h; if (lo) lo[o] = l;
}

// ---------------------------------------------------------------------------------------------
// Patch gather for the 16x16/16 patch-embed conv == GEMM (patch_embed.py:17-27):
// img NCHW fp32 [n,3,H,W] -> planes [n*hp*wp, 768], K order (c, ky, kx) == conv weight flatten.
// One thread = one (token, c, ky) row of 16 pixels (64 B in, 32 B out per plane).
template <bool SPLIT>
__global__ void patch_gather_kernel(const float* img, int n, int H, int W, f16* o_hi, f16* o_lo, int64_t row0, int64_t orows, unsigned long long* rng) {
    const int hp = H / 16, wp = W / 16;
    const int64_t total = (int64_t)n * hp * wp * 48;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        int ck = (int)(i % 48); int64_t tok = i / 48;
        int c = ck / 16, ky = ck % 16;
        int px = (int)(tok % wp); int64_t t2 = tok / wp; int py = (int)(t2 % hp); int b = (int)(t2 / hp);
        const float* src = img + (((int64_t)b * 3 + c) * H + py * 16 + ky) * W + px * 16;
        H8 h0, h1, l0, l1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = *reinterpret_cast<const float4*>(src + q * 4);
            float y[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f16 hh, ll;
                if (SPLIT) split_f16(y[e], hh, ll, rng); else { hh = to_f16_sat(y[e], rng); ll = (f16)0; }
                int k = q * 4 + e;
                if (k < 8) { h0.e[k] = hh; l0.e[k] = ll; } else { h1.e[k - 8] = hh; l1.e[k - 8] = ll; }
            }
        }
        const size_t o = blk_off<SPLIT>(row0 + tok, ck * 16, orows);
        *reinterpret_cast<uint4*>(o_hi + o) = h0.u; *reinterpret_cast<uint4*>(o_hi + o + 8) = h1.u;
        if (SPLIT) { *reinterpret_cast<uint4*>(o_hi + o + 32) = l0.u; *reinterpret_cast<uint4*>(o_hi + o + 40) = l1.u; }
    }
}

// Same gather straight from camera-format input: uint8 HWC [n,H,W,3] (the step before the
// path).  One thread = one (token, ky) row = 16 pixels x 3 interleaved channels = 48 CONTIGUOUS bytes
// (3 x 16-B loads); the ImgNorm of the reference, ToTensor + Normalize(0.5, 0.5)
// (vista_slam/utils/image.py:13, datasets/slam_images_only.py:19,30): (u/255 - 0.5)/0.5 in fp32 with the
// same operation order, is fused here, so the result is bit-identical to feeding the normalised
// fp32 NCHW image.
template <bool SPLIT>
__global__ void patch_gather_u8hwc_kernel(const uint8_t* img, int n, int H, int W, f16* o_hi, f16* o_lo, int64_t row0, int64_t orows, unsigned long long* rng) {
    const int hp = H / 16, wp = W / 16;
    const int64_t total = (int64_t)n * hp * wp * 16;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        const int ky = (int)(i % 16); const int64_t tok = i / 16;
        const int px = (int)(tok % wp); const int64_t t2 = tok / wp; const int py = (int)(t2 % hp); const int b = (int)(t2 / hp);
        const uint8_t* src = img + (((int64_t)b * H + py * 16 + ky) * W + px * 16) * 3;   // 48 B, 16-B aligned (W % 16 == 0)
        union { uint4 v[3]; uint8_t e[48]; } raw;
        raw.v[0] = ldg16(src); raw.v[1] = ldg16(src + 16); raw.v[2] = ldg16(src + 32);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            H8 h0, h1, l0, l1;
#pragma unroll
            for (int kx = 0; kx < 16; ++kx) {
                const float a = (float)raw.e[kx * 3 + c] / 255.0f;
                const float v = (a - 0.5f) / 0.5f;
                f16 hh, ll;
                if (SPLIT) split_f16(v, hh, ll, rng); else { hh = to_f16_sat(v, rng); ll = (f16)0; }
                if (kx < 8) { h0.e[kx] = hh; l0.e[kx] = ll; } else { h1.e[kx - 8] = hh; l1.e[kx - 8] = ll; }
            }
            const size_t o = blk_off<SPLIT>(row0 + tok, (c * 16 + ky) * 16, orows);
            *reinterpret_cast<uint4*>(o_hi + o) = h0.u; *reinterpret_cast<uint4*>(o_hi + o + 8) = h1.u;
            if (SPLIT) { *reinterpret_cast<uint4*>(o_hi + o + 32) = l0.u; *reinterpret_cast<uint4*>(o_hi + o + 40) = l1.u; }
        }
    }
}

// x[s, 0, :] = token  (pose token prepend, sta_model.py:206-213)
__global__ void fill_pose_token_kernel(float* x, int S, int ntok, int D, const float* tok) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S * D) { int s = i / D, d = i - s * D; x[(size_t)s * ntok * D + d] = tok[d]; }
}

// Decoder rows -> the reference's token order: out[b, 0, :] = pose row of sequence s0 + b, out[b, 1 + t, :] = patch row
// (s0 + b) * N + t  (x: [S*N patch rows | S pose rows], pose_row0 = S*N; sta_model.py:206-213 prepends the pose token)
__global__ void emit_tokens_kernel(const float* x, int s0, int B, int N, int D, int64_t pose_row0, float* out) {
    const int d4 = D / 4;
    const int64_t total = (int64_t)B * (N + 1) * d4, step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const int c = (int)(i % d4); const int64_t r = i / d4;
        const int t = (int)(r % (N + 1)); const int b = (int)(r / (N + 1));
        const int64_t src = t == 0 ? pose_row0 + s0 + b : (int64_t)(s0 + b) * N + (t - 1);
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(x + src * D)[c];
    }
}

// sta_decode_pos: the two views' int64 [B, N, 2] (y, x) positions -> one int32 table [2B][N][2], clamped to the RoPE table's range
// [-1, pos_max] (the shim passes the true maximum, so nothing is clamped on that path); the tail of the grid fills `ident`, a cos / sin
// table whose every row is the identity rotation (1, 0): the QKV epilogues of the call rotate by it, i.e. not at all
__global__ __launch_bounds__(256) void rope_pos_table_kernel(const int64_t* pos1, const int64_t* pos2, int64_t n, int pos_max, int* out,
                                                             float2* ident, int64_t n_ident) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * n) {
        int64_t v = i < n ? pos1[i] : pos2[i - n];
        v = v < -1 ? -1 : (v > pos_max ? pos_max : v);
        out[i] = (int)v;
    } else if (i - 2 * n < n_ident) ident[i - 2 * n] = make_float2(1.f, 0.f);
}

// sta_decode_pos: 2-D RoPE (pos_embed.py:169-185; curope kernels.cu:17-82) applied IN PLACE to a head-major Q or K buffer
// [S][heads][npad][64] of fp16 planes, with the position of every token looked up in `pos` ([S][ntok][2]; token index ntok = the pose
// token, position -1).  One thread per rotation pair (d, d + 16) of the y half (d < 32) or the x half of a head.  The general form of
// what the QKV epilogue does on the patch grid - off the throughput path on purpose: the epilogue of the grid form stays free of a
// per-row table lookup (a first version with the lookup inside the epilogue put the 192x128 QKV kernel on 256 B of scratch: x0.6).
template <bool SPLIT>
__global__ __launch_bounds__(256) void rope_planes_kernel(f16* hi, f16* lo, int S, int heads, int npad, int ntok, const int* pos,
                                                          const float* tab, unsigned long long* rng) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)S * heads * (ntok + 1) * 32;
    if (i >= total) return;
    const int j = (int)(i & 31), xp = j >> 4, f = j & 15;
    int64_t r = i >> 5;
    const int t = (int)(r % (ntok + 1)); r /= ntok + 1;
    const int hd = (int)(r % heads), s = (int)(r / heads);
    const int ps = t < ntok ? pos[((int64_t)s * ntok + t) * 2 + xp] : -1;
    const float2 cs = *reinterpret_cast<const float2*>(tab + ((size_t)(ps + 1) * 16 + f) * 2);
    const int64_t o = (((int64_t)s * heads + hd) * npad + t) * 64 + xp * 32 + f;
    float v0 = (float)hi[o], v1 = (float)hi[o + 16];
    if (SPLIT) { v0 += (float)lo[o]; v1 += (float)lo[o + 16]; }
    const float r0 = v0 * cs.x - v1 * cs.y, r1 = v1 * cs.x + v0 * cs.y;
    RangeAcc ra;
    if (SPLIT) { split_f16(r0, hi[o], lo[o], ra); split_f16(r1, hi[o + 16], lo[o + 16], ra); }
    else { hi[o] = to_f16_sat(r0, ra); hi[o + 16] = to_f16_sat(r1, ra); }
    ra.flush(rng);
}

// ---------------------------------------------------------------------------------------------
// Bilinear x2 upsample, align_corners=True (dpt_block.py:215-216,320), NHWC fp16 planes.
// Output may be cropped to (Hc,Wc) <= (2Hi,2Wi) (dpt_head.py:58); interpolation ratios always use
// the full (2Hi,2Wi) grid.  One workgroup = NR consecutive output rows of one image: the rows' taps / weights are block-uniform
// and the per-element index math is 32-bit; one thread = 8 channels of one output COLUMN, for all NR rows, per step.
// NR = 4 (round 4): four consecutive output rows read four consecutive input rows (align_corners x2: rows 2m .. 2m+3 <- m-1 .. m+2),
// so a thread fetches 4 rows x 2 columns of taps ONCE for 4 outputs - 16 tap loads instead of 32 - and an input row is
// fetched from L2 by ~1.5 workgroups instead of 4.  The kernel was bound by exactly that traffic: 1.6 GB of (re-)fetched
// input + 1.6 GB of output through the L2 <-> CU fabric in 720 us = 4.4 TB/s, the copy rate of this chip; its loads cost
// 40 us alone and 250 us next to the stores (round-3 ablation), and re-ordering them around the stores (software pipeline,
// round 4) changed nothing.
template <bool SPLIT, int NR>
__global__ __launch_bounds__(256) void bilinear_up2_kernel(const f16* i_hi, const f16* i_lo, int n, int Hi, int Wi, int C,
                                                           int Hc, int Wc, f16* o_hi, f16* o_lo, int mx /* input and output are f16mx rows */, unsigned long long* rng) {
    const int c8 = C / 8;
    const float ry = Hi > 1 ? (float)(Hi - 1) / (float)(2 * Hi - 1) : 0.f;
    const float rx = Wi > 1 ? (float)(Wi - 1) / (float)(2 * Wi - 1) : 0.f;
    // XCD-aware row order: consecutive workgroups run on different XCDs (block b -> XCD b % 8), and neighbouring output
    // rows read the same input rows - so give every XCD a contiguous band of output rows (its L2 then fetches an input
    // row once instead of once per XCD: the counters showed 5x the algorithmic read bytes with the plain order)
    int bid;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = blockIdx.x % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + blockIdx.x / 8;
    }
    const int groups = (Hc + NR - 1) / NR;
    const int b = bid / groups, yg = (bid - b * groups) * NR;
    // input rows of the group: [ybase, ybase + NR] at most (NR + 1 rows: the regular x2 pattern needs NR, a crop / tiny image may differ)
    int y0r[NR], y1r[NR]; float fyr[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int y = yg + k < Hc ? yg + k : Hc - 1;
        const float sy = ry * y;
        y0r[k] = (int)sy; y1r[k] = y0r[k] + (y0r[k] < Hi - 1); fyr[k] = sy - y0r[k];
    }
    const int ybase = y0r[0];
    bool regular = NR == 4 && yg + 3 < Hc;                   // the interior pairing (0,1) (1,2) (1,2) (2,3)
    if (NR == 4) {
        const int pa[4] = {0, 1, 1, 2}, pb[4] = {1, 2, 2, 3};
#pragma unroll
        for (int k = 0; k < NR; ++k) regular = regular && y0r[k] - ybase == pa[k & 3] && y1r[k] - ybase == pb[k & 3];
    }
    constexpr int NIN = NR + 1;                              // rows ybase .. ybase + NR cover every tap of the group
    const int64_t irows = (int64_t)n * Hi * Wi, orows = (int64_t)n * Hc * Wc;
    const bool pow2 = (c8 & (c8 - 1)) == 0;
    const int sh = __ffs(c8) - 1;
    const int per_row = Wc * c8;
    for (int i = threadIdx.x; i < per_row; i += 256) {
        const int x = pow2 ? i >> sh : i / c8;
        const int c = (i - x * c8) * 8;
        const float sx = rx * x;
        const int x0 = (int)sx, x1 = x0 + (x0 < Wi - 1);
        const float fx = sx - x0;
        // taps of the NIN input rows (rows past the last one the group uses are not loaded)
        float row_top[NIN][8];                               // (1 - fx) * v(x0) + fx * v(x1) of every input row: the horizontal lerp is shared by the output rows
        const int last_in = y1r[NR - 1] - ybase;
#pragma unroll
        for (int r = 0; r < NIN; ++r) {
            if (r > last_in) continue;
            const int yr = ybase + r < Hi ? ybase + r : Hi - 1;
            const size_t rb = ((size_t)b * Hi + yr) * Wi;
            const size_t oa = blk_off<SPLIT>(rb + x0, c, irows), ob = blk_off<SPLIT>(rb + x1, c, irows);
            H8 a, b_, al, bl;
            a.u = ldg16(i_hi + oa); b_.u = ldg16(i_hi + ob);
            if (SPLIT) { al.u = ldg16(i_hi + oa + 32); bl.u = ldg16(i_hi + ob + 32); }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v0 = (float)a.e[e], v1 = (float)b_.e[e];
                if (SPLIT && mx) {       // the second field holds (hi8, lo8) byte pairs: value = hi + lo8 * 2^-11
                    constexpr float KL = 1.0f / (float)(1 << STA_MX_A_SLO);
                    v0 += __builtin_amdgcn_cvt_f32_bf8(reinterpret_cast<const unsigned short*>(&al)[e], 1) * KL;
                    v1 += __builtin_amdgcn_cvt_f32_bf8(reinterpret_cast<const unsigned short*>(&bl)[e], 1) * KL;
                } else if (SPLIT) { v0 += (float)al.e[e]; v1 += (float)bl.e[e]; }
                row_top[r][e] = (1.f - fx) * v0 + fx * v1;   // same association as ATen upsample_bilinear2d: w0*v00 + w1*v01
            }
        }
        // output row k = lerp of input rows (ra, rb) of the group.  Interior groups of the x2 pattern pair them as
        // (0,1) (1,2) (1,2) (2,3) - compile-time register indices; any other group (first rows of an image, crops) selects by value.
        auto emit = [&](int k, const float (&top)[8], const float (&bot)[8]) {
            const float fy = fyr[k];
            H8 oh, ol;
            float vout[8];
            RangeAcc ra;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (1.f - fy) * top[e] + fy * bot[e];          // h0*(...) + h1*(...)
                vout[e] = v;
                if (SPLIT && mx) continue;
                if (SPLIT) split_f16(v, oh.e[e], ol.e[e], ra); else oh.e[e] = to_f16_sat(v, ra);
            }
            const size_t o = blk_off<SPLIT>(((size_t)b * Hc + yg + k) * Wc + x, c, orows);
            if (SPLIT && mx) {
                const MX4 m0 = split_mx4<false>(vout, ra), m1 = split_mx4<false>(vout + 4, ra);     // two 16-B stores per lane
                *reinterpret_cast<uint4*>(o_hi + o) = make_uint4(m0.hi.x, m0.hi.y, m1.hi.x, m1.hi.y);
                *reinterpret_cast<uint4*>(o_hi + o + 32) = make_uint4(m0.pairs.x, m0.pairs.y, m1.pairs.x, m1.pairs.y);
                ra.flush(rng);
                return;
            }
            ra.flush(rng);
            *reinterpret_cast<uint4*>(o_hi + o) = oh.u;
            if (SPLIT) *reinterpret_cast<uint4*>(o_hi + o + 32) = ol.u;
        };
        if (NR == 4 && regular) {
            emit(0, row_top[0], row_top[1]);
            emit(1, row_top[1], row_top[NR >= 2 ? 2 : 0]);
            emit(2, row_top[1], row_top[NR >= 2 ? 2 : 0]);
            emit(3, row_top[NR >= 2 ? 2 : 0], row_top[NR >= 3 ? 3 : 0]);
        } else {
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                if (yg + k >= Hc) continue;
                const int ra_ = y0r[k] - ybase, rb_ = y1r[k] - ybase;
                float top[8], bot[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    top[e] = 0.f; bot[e] = 0.f;
#pragma unroll
                    for (int r = 0; r < NIN; ++r) { if (r == ra_) top[e] = row_top[r][e]; if (r == rb_) bot[e] = row_top[r][e]; }
                }
                emit(k, top, bot);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Final 1x1 conv 128 -> 4 (dpt_block.py:323) fused with the pointmap / confidence activations
// (postprocess.py:10-62, modes ('exp',-inf,inf) and ('exp',1,inf)):
//   pts = xyz / max(|xyz|,1e-8) * expm1(|xyz|),  conf = 1 + exp(c)
// Input planes [npix, 128] (ReLU already applied by the producing conv).  16 lanes per pixel.
template <bool SPLIT>
__global__ __launch_bounds__(256) void head_final_kernel(const f16* i_hi, const f16* i_lo, int64_t pix0, int64_t irows, int64_t npix,
                                                         const float* w /*[4][128]*/, const float* bias /*[4]*/,
                                                         float* pts, float* conf, int mx = 0 /* input planes are f16mx rows */) {
    const int sub = threadIdx.x & 15;
    int64_t pix = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t step = ((int64_t)gridDim.x * blockDim.x) >> 4;
    float wv[4][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[o][e] = w[o * 128 + sub * 8 + e];
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2], b3 = bias[3];
    const int64_t npix_r = (npix + 3) / 4 * 4;   // keep whole waves in the shuffle
    for (; pix < npix_r; pix += step) {
        const bool ok = pix < npix;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const size_t o = blk_off<SPLIT>(pix0 + pix, sub * 8, irows);
            H8 a; a.u = ldg16(i_hi + o);
            H8 al; if (SPLIT) al.u = ldg16(i_hi + o + 32);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = (float)a.e[e];
                if (SPLIT && mx) v += __builtin_amdgcn_cvt_f32_bf8(reinterpret_cast<const unsigned short*>(&al)[e], 1) * (1.0f / (float)(1 << STA_MX_A_SLO));
                else if (SPLIT) v += (float)al.e[e];
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o] += v * wv[o][e];
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            acc[o] += __shfl_xor(acc[o], 1); acc[o] += __shfl_xor(acc[o], 2);
            acc[o] += __shfl_xor(acc[o], 4); acc[o] += __shfl_xor(acc[o], 8);
        }
        if (ok && sub == 0) {
            float x = acc[0] + b0, y = acc[1] + b1, z = acc[2] + b2, c = acc[3] + b3;
            float d = sqrtf(x * x + y * y + z * z);
            float sc = expm1f(d) / fmaxf(d, 1e-8f);
            pts[pix * 3 + 0] = x * sc; pts[pix * 3 + 1] = y * sc; pts[pix * 3 + 2] = z * sc;
            conf[pix] = 1.0f + expf(c);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Pose head (heads/pose_head.py:109-120): MLP 768->512->512->512 (ReLU), fc_t(3), fc_rot(9),
// fc_conf(1)+sigmoid, rotation = closest SO(3) matrix to the row-normalised 3x3 (pose_head.py:38-57
// computes the same matrix through torch.svd: R = V diag(1,1,det) U^T).  fp32 VALU (1.85 MFLOP/sample,
// latency-only; must not run in half: lu/svd fail in half in the reference too).  One block/sample.
struct PoseParams {
    const float* tok; int64_t tok_stride; int D; int Hd;
    const float *w0, *b0, *w1, *b1, *w2, *b2, *wt, *bt, *wr, *br, *wc, *bc;
    float* pose; float* conf;
    float* pose2; float* conf2; int split;      // samples >= split (when pose2 != nullptr) go to pose2 / conf2, re-indexed from 0
};

__device__ inline void jacobi_eig3(double A[3][3], double V[3][3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off < 1e-40) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 3; ++k) { double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; ++k) { double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
    }
}

// R = closest rotation to M (3x3): M = L S W^T  ->  R = L diag(1,1,det(L W^T)) W^T, realised with
// cross products so that it stays well defined when the smallest singular value vanishes.
__device__ inline void nearest_rotation(const double M[3][3], double R[3][3]) {
    double B[3][3], W[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += M[k][i] * M[k][j]; B[i][j] = s; }
    jacobi_eig3(B, W);
    int idx[3] = {0, 1, 2};
    double ev[3] = {B[0][0], B[1][1], B[2][2]};
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (ev[idx[b]] > ev[idx[a]]) { int t = idx[a]; idx[a] = idx[b]; idx[b] = t; }
    double w0[3], w1[3], l0[3], l1[3];
    for (int k = 0; k < 3; ++k) { w0[k] = W[k][idx[0]]; w1[k] = W[k][idx[1]]; }
    for (int i = 0; i < 3; ++i) { l0[i] = 0; l1[i] = 0; for (int k = 0; k < 3; ++k) { l0[i] += M[i][k] * w0[k]; l1[i] += M[i][k] * w1[k]; } }
    double n0 = sqrt(l0[0] * l0[0] + l0[1] * l0[1] + l0[2] * l0[2]);
    for (int i = 0; i < 3; ++i) l0[i] /= n0;
    double dp = l0[0] * l1[0] + l0[1] * l1[1] + l0[2] * l1[2];
    for (int i = 0; i < 3; ++i) l1[i] -= dp * l0[i];
    double n1 = sqrt(l1[0] * l1[0] + l1[1] * l1[1] + l1[2] * l1[2]);
    for (int i = 0; i < 3; ++i) l1[i] /= n1;
    double l2[3] = {l0[1] * l1[2] - l0[2] * l1[1], l0[2] * l1[0] - l0[0] * l1[2], l0[0] * l1[1] - l0[1] * l1[0]};
    double w2[3] = {w0[1] * w1[2] - w0[2] * w1[1], w0[2] * w1[0] - w0[0] * w1[2], w0[0] * w1[1] - w0[1] * w1[0]};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = l0[i] * w0[j] + l1[i] * w1[j] + l2[i] * w2[j];
}

// one MLP layer: out[b, n] = relu(w[n,:] . in[b,:] + bias[n]); one wave per (sample, neuron) so the
// 1.85 MFLOP head spreads over the chip instead of serialising on B blocks.
__global__ __launch_bounds__(256) void pose_layer_kernel(const float* in, int64_t in_stride, const float* w, const float* bias,
                                                         float* out, int K, int N, int relu) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (n >= N) return;
    const float4* wr = reinterpret_cast<const float4*>(w + (size_t)n * K);
    const float4* xr = reinterpret_cast<const float4*>(in + (size_t)b * in_stride);
    float s = 0.f;
    for (int k = lane; k < K / 4; k += 64) { float4 a = wr[k], x = xr[k]; s += (a.x * x.x + a.y * x.y) + (a.z * x.z + a.w * x.w); }
    s = wave_sum(s);
    if (lane == 0) { s += bias[n]; out[(size_t)b * N + n] = relu ? fmaxf(s, 0.f) : s; }
}

// heads fc_t(3) / fc_rot(9) / fc_conf(1) + rotation orthogonalisation + 4x4 packing; one block per sample
__global__ __launch_bounds__(256) void pose_final_kernel(const PoseParams p, const float* feat) {
    __shared__ float outv[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* in = feat + (size_t)b * p.Hd;
    for (int n = wave; n < 13; n += 4) {
        const float* wr = n < 3 ? p.wt + (size_t)n * p.Hd : (n < 12 ? p.wr + (size_t)(n - 3) * p.Hd : p.wc);
        // (16 B per lane and load: with one element per iteration the 13 rows were 32 dependent load round trips - 21 us of kernel)
        float s = 0.f;
        if ((p.Hd & 255) == 0) {
            for (int k = lane * 4; k < p.Hd; k += 256) {
                const float4 w4 = *reinterpret_cast<const float4*>(wr + k), x4 = *reinterpret_cast<const float4*>(in + k);
                s += (w4.x * x4.x + w4.y * x4.y) + (w4.z * x4.z + w4.w * x4.w);
            }
        } else {
            for (int k = lane; k < p.Hd; k += 64) s += wr[k] * in[k];
        }
        s = wave_sum(s);
        if (lane == 0) outv[n] = s + (n < 3 ? p.bt[n] : (n < 12 ? p.br[n - 3] : p.bc[0]));
    }
    __syncthreads();
    if (tid == 0) {
        // m (3x3 row-major from fc_rot) -> normalise rows -> R = nearest rotation of normalised m
        double M[3][3], R[3][3];
        for (int i = 0; i < 3; ++i) {
            float a = outv[3 + i * 3], c = outv[4 + i * 3], d = outv[5 + i * 3];
            float nrm = fmaxf(sqrtf(a * a + c * c + d * d), 1e-12f);   // F.normalize eps
            M[i][0] = a / nrm; M[i][1] = c / nrm; M[i][2] = d / nrm;
        }
        nearest_rotation(M, R);
        const bool second = p.pose2 != nullptr && b >= p.split;
        float* o = second ? p.pose2 + (size_t)(b - p.split) * 16 : p.pose + (size_t)b * 16;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) o[i * 4 + j] = (float)R[i][j]; o[i * 4 + 3] = outv[i]; }
        o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
        (second ? p.conf2 + (b - p.split) : p.conf + b)[0] = 1.0f / (1.0f + expf(-outv[12]));
    }
}

// ---------------------------------------------------------------------------------------------
// the reductions that consume the path's output every accepted pair, fused into one pass.
//  * estimate_intrinsic_from_pts3d (vista_slam/utils/slam_utils.py:8-79): confidence-weighted least
//    squares focal  fx = sum(w*xz*u)/sum(w*xz^2), fy likewise, u = col - W/2, v = row - H/2,
//    w = clamp(conf, 1e-6), xz = nan_to_num(X/Z, 0), yz = nan_to_num(Y/Z, 0); shared or per-image K.
//  * depths = pts[...,2] (slam.py:185) and conf.mean() per image (pose_graph.py:37).
// HBM-bound (16 B/pixel in, 4 B out).  Partial sums are fp64 per block (deterministic two-stage
// reduction: block partials -> finalise kernel), which is at least as accurate as torch's fp32 sums.
// tr != 0: the maps are portrait frames in image orientation [H,W] and the reference would see them TRANSPOSED
// ([W,H] views, utils/misc.py:60-61,81): u = row - H/2 pairs with X, v = col - W/2 with Y, principal point (H/2, W/2).
__global__ __launch_bounds__(256) void intrinsics_partial_kernel(const float* pts, const float* conf, int B, int H, int W,
                                                                 float* depth, double* partial /*[B][nblk][5]*/, int nblk, int tr) {
    const int b = blockIdx.y;
    const int64_t hw = (int64_t)H * W;
    const float cx = W / 2.0f, cy = H / 2.0f;
    double s[5] = {0, 0, 0, 0, 0};   // fx_num, fx_den, fy_num, fy_den, conf_sum
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
        const float* p3 = pts + ((int64_t)b * hw + i) * 3;
        const float X = p3[0], Y = p3[1], Z = p3[2];
        const float c = conf[(int64_t)b * hw + i];
        if (depth) depth[(int64_t)b * hw + i] = Z;
        const float w = fmaxf(c, 1e-6f);
        float xz = X / Z, yz = Y / Z;
        if (!isfinite(xz)) xz = 0.f;
        if (!isfinite(yz)) yz = 0.f;
        const float col = (float)(i % W) - cx, row = (float)(i / W) - cy;
        const float u = tr ? row : col, v = tr ? col : row;
        s[0] += (double)(w * xz * u); s[1] += (double)(w * xz * xz);
        s[2] += (double)(w * yz * v); s[3] += (double)(w * yz * yz);
        s[4] += (double)c;
    }
    __shared__ double red[4][5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double v = s[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) partial[((int64_t)b * nblk + blockIdx.x) * 5 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// shared: 0 = one K per image, 1 = one K over all B images, g >= 2 = one K per group of g consecutive images
// (g = 2: the two views of a pair, slam.py:184 shared_intrinsic=True) -> K [B/g,3,3].
__global__ void intrinsics_final_kernel(const double* partial, int B, int nblk, int H, int W, int shared,
                                        float* K /*[3,3] or [B,3,3] or [B/g,3,3]*/, float* conf_mean /*[B] or null*/, int tr) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;          // one wave: lanes stride over the block partials, then a butterfly
    const int lane = threadIdx.x;
    const float pcx = (tr ? H : W) / 2.0f, pcy = (tr ? W : H) / 2.0f;
    double tot[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b) {
        double s[5] = {0, 0, 0, 0, 0};
        for (int j = lane; j < nblk; j += 64) for (int k = 0; k < 5; ++k) s[k] += partial[((int64_t)b * nblk + j) * 5 + k];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o);
        }
        if (lane != 0) continue;                                // (every lane holds the sums; lane 0 writes)
        if (conf_mean) conf_mean[b] = (float)(s[4] / ((double)H * W));
        for (int k = 0; k < 5; ++k) tot[k] += s[k];
        if (shared >= 2 && (b + 1) % shared == 0) {
            float* Kb = K + (b / shared) * 9;
            Kb[0] = (float)(tot[0] / tot[1]); Kb[1] = 0.f; Kb[2] = pcx;
            Kb[3] = 0.f; Kb[4] = (float)(tot[2] / tot[3]); Kb[5] = pcy;
            Kb[6] = 0.f; Kb[7] = 0.f; Kb[8] = 1.f;
            for (int k = 0; k < 5; ++k) tot[k] = 0;
        }
        if (!shared) {
            float* Kb = K + b * 9;
            Kb[0] = (float)(s[0] / s[1]); Kb[1] = 0.f; Kb[2] = pcx;
            Kb[3] = 0.f; Kb[4] = (float)(s[2] / s[3]); Kb[5] = pcy;
            Kb[6] = 0.f; Kb[7] = 0.f; Kb[8] = 1.f;
        }
    }
    if (shared == 1 && lane == 0) {
        K[0] = (float)(tot[0] / tot[1]); K[1] = 0.f; K[2] = pcx;
        K[3] = 0.f; K[4] = (float)(tot[2] / tot[3]); K[5] = pcy;
        K[6] = 0.f; K[7] = 0.f; K[8] = 1.f;
    }
}

// estimate_scale_with_depth_and_confidence (slam_utils.py:168-190): s = sum(w Di Dj) / sum(w Di Di),
// w = clamp(ci*cj, 1e-6).  Single block (n ~ 5e4), fp64 partials.
__global__ __launch_bounds__(1024) void scale_estimate_kernel(const float* Di, const float* Dj, const float* ci, const float* cj,
                                                               int64_t n, float* s_out) {
    double num = 0, den = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float w = fmaxf(ci[i] * cj[i], 1e-6f);
        num += (double)(w * Di[i] * Dj[i]); den += (double)(w * Di[i] * Di[i]);
    }
    __shared__ double rn[16], rd[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { num += __shfl_xor(num, o); den += __shfl_xor(den, o); }
    if (lane == 0) { rn[wave] = num; rd[wave] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { a += rn[k]; b += rd[k]; }
        *s_out = (float)(a / b);
    }
}

// ---------------------------------------------------------------------------------------------
// curope-compatible in-place 2-D RoPE on tokens (B,N,Hh,D) of type T = f16 / float / double (kernels.cu:17-82,101): one
// thread per (token, head, pair); the rotation is evaluated in fp32 for every T, like the reference kernel (its shared
// memory and cos / sin are float); accurate sinf/cosf/powf (the reference CUDA build uses fast-math variants).
template <class T>
__global__ void rope2d_inplace_kernel(T* tok, int64_t sb, int64_t sn, const int64_t* pos,
                                      int B, int N, int Hh, int D, float base, float fwd) {
    const int Q = D / 4;
    const int64_t total = (int64_t)B * N * Hh * 2 * Q;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int d = (int)(i % Q); int64_t t = i / Q;
    int xy = (int)(t % 2); t /= 2;
    int h = (int)(t % Hh); t /= Hh;
    int n = (int)(t % N); int b = (int)(t / N);
    T* base_p = tok + b * sb + n * sn + (int64_t)h * D + xy * 2 * Q;
    const float pp = (float)pos[((int64_t)b * N + n) * 2 + xy];
    const float ang = fwd * pp / powf(base, (float)d / (float)Q);
    const float c = cosf(ang), s = sinf(ang);
    const float u = (float)base_p[d], v = (float)base_p[d + Q];
    base_p[d] = (T)(u * c - v * s);
    base_p[d + Q] = (T)(v * c + u * s);
}

// ---------------------------------------------------------------------------------------------
// Weight repack: fp32 source -> fp16 hi/lo planes with an index permutation.
//   mode 0: identity ([N,K] linear weights, patch-embed conv flatten)
//   mode 1: conv [Co,Ci,kh,kw] -> [Co][kh][kw][Ci]
//   mode 2: transposed conv [Ci,Co,k,k] -> [(dy*k+dx)*Co+co][Ci]
__global__ void repack_weight_kernel(const float* src, f16* hi, f16* lo, int64_t total, int mode,
                                     int d0, int d1, int d2, int d3, int64_t N, int64_t K, int64_t n_off, int mx, unsigned long long* rng) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        int64_t si = i;
        if (mode == 1) {          // dst index i = ((co*kh + ky)*kw + kx)*Ci + ci ; src [co][ci][ky][kx]
            int ci = (int)(i % d1); int64_t t = i / d1; int kx = (int)(t % d3); t /= d3; int ky = (int)(t % d2); int co = (int)(t / d2);
            si = (((int64_t)co * d1 + ci) * d2 + ky) * d3 + kx;
        } else if (mode == 2) {   // dst i = ((dy*k+dx)*Co + co)*Ci + ci ; src [ci][co][dy][dx], d0=Ci d1=Co d2=d3=k
            int ci = (int)(i % d0); int64_t t = i / d0; int co = (int)(t % d1); t /= d1; int dx = (int)(t % d3); int dy = (int)(t / d3);
            si = (((int64_t)ci * d1 + co) * d2 + dy) * d3 + dx;
        }
        f16 h, l; split_f16(src[si], h, l, rng);
        // logical packed index i = n*K + k  ->  blocked [K/32][N][hi32|lo32]
        const int64_t nl = i / K, k = i - nl * K, n = nl + n_off;
        const size_t o = ((size_t)(k >> 5) * N + n) * 64 + (k & 31);
        if (mx) store_mx1<true>(hi, o, src[si], rng);          // f16mx weight rows: [hi f16 | e4m3(hi*2^4) | e4m3(lo*2^15)]
        else { hi[o] = h; hi[o + 32] = l; }
    }
}

__global__ void expand_bias_kernel(const float* b, float* out, int cout, int reps) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cout * reps) out[i] = b[i % cout];
}

// Gather of up to 64 equally-typed chunks (16-byte units) in one launch: blockIdx.y = chunk.  Used by the keyframe
// scheduler (sta_regress_views) to batch per-view feature tensors that live at unrelated addresses.
struct GatherChunks { const uint4* src[64]; uint4* dst[64]; int64_t n16[64]; };
__global__ __launch_bounds__(256) void gather_chunks_kernel(GatherChunks g) {
    const uint4* s = g.src[blockIdx.y];
    uint4* d = g.dst[blockIdx.y];
    const int64_t n = g.n16[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}

// Row (e): the compact per-pair record a SLAM consumer reads (slam.py:165-185) - per view pose 4x4, pose confidence,
// depth = pts[..., 2], confidence map - packed for the one all-gather of a step (vista_slam_amd/parallel.py).
// Row b of out = [view 0: pose16 | pose_conf | depth HW | conf HW][view 1: the same]; blockIdx.y = b * 2 + view.
struct PackCompact { const float* pts[2]; const float* conf[2]; const float* pose[2]; const float* pose_conf[2]; float* out; int64_t hw; int64_t row_stride; };
__global__ __launch_bounds__(256) void pack_compact_kernel(PackCompact g) {
    const int b = blockIdx.y >> 1, v = blockIdx.y & 1;
    float* o = g.out + (size_t)b * g.row_stride + (size_t)v * (17 + 2 * g.hw);
    const float* pts = g.pts[v] + (size_t)b * g.hw * 3;
    const float* cf = g.conf[v] + (size_t)b * g.hw;
    if (blockIdx.x == 0 && threadIdx.x < 17) o[threadIdx.x] = threadIdx.x < 16 ? g.pose[v][(size_t)b * 16 + threadIdx.x] : g.pose_conf[v][b];
    o += 17;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * g.hw; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = i < g.hw ? pts[i * 3 + 2] : cf[i - g.hw];
}

// ---------------------------------------------------------------------------------------------------------
// f3 input step: crop -> LANCZOS rescale -> centre crop -> ImgNorm / ImgGray, fused
// (base_view_graph_dataset.py:171-225, cropping.py:54-81, slam_images_only.py:19-33).  The rescale reproduces Pillow's
// 8-bit resampler bit for bit: separable, horizontal pass into a uint8 intermediate, 22-bit fixed-point coefficients
// (tables built on the host in double exactly like Resample.c precompute_coeffs), int32 accumulation from 2^21,
// arithmetic shift, clamp.  Only the rows / columns the final crop keeps are computed.  HBM-bound, tiny.
struct PreParams {
    const uint8_t* src; int Ws;            // source frame [Hs,Ws,3]
    int l, t;                              // first crop origin in the source
    int ow, oh;                            // final size
    const int* bh; const int* kh; int ksh; // horizontal bounds [ow][2] / coeffs [ow][ksh] (columns l2 .. l2+ow of the rescaled image)
    const int* bv; const int* kv; int ksv; // vertical   bounds [oh][2] / coeffs [oh][ksv] (rows t2 .. t2+oh), ymin relative to y_first
    int y_first, y_rows;                   // rows of the cropped source the vertical pass reads
    uint8_t* tmp;                          // [y_rows][ow][4]
    uint8_t* out_u8; float* out_rgb; float* out_gray;
};
__device__ __forceinline__ int pre_clip8(int acc) { acc >>= 22; return acc < 0 ? 0 : (acc > 255 ? 255 : acc); }

__global__ __launch_bounds__(256) void pre_horizontal_kernel(PreParams p) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, yr = blockIdx.y;
    if (x >= p.ow) return;
    const int x0 = p.bh[2 * x], n = p.bh[2 * x + 1];
    const int* k = p.kh + (size_t)x * p.ksh;
    const uint8_t* row = p.src + ((size_t)(p.t + p.y_first + yr) * p.Ws + p.l + x0) * 3;
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    for (int i = 0; i < n; ++i) {
        const int w = k[i];
        a0 += row[3 * i] * w; a1 += row[3 * i + 1] * w; a2 += row[3 * i + 2] * w;
    }
    uchar4 o; o.x = (uint8_t)pre_clip8(a0); o.y = (uint8_t)pre_clip8(a1); o.z = (uint8_t)pre_clip8(a2); o.w = 0;
    reinterpret_cast<uchar4*>(p.tmp)[(size_t)yr * p.ow + x] = o;
}

__global__ __launch_bounds__(256) void pre_vertical_kernel(PreParams p) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= p.ow) return;
    const int y0 = p.bv[2 * y], n = p.bv[2 * y + 1];
    const int* k = p.kv + (size_t)y * p.ksv;
    const uchar4* col = reinterpret_cast<const uchar4*>(p.tmp) + (size_t)y0 * p.ow + x;
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    for (int i = 0; i < n; ++i) {
        const int w = k[i];
        const uchar4 v = col[(size_t)i * p.ow];
        a0 += v.x * w; a1 += v.y * w; a2 += v.z * w;
    }
    const int r = pre_clip8(a0), g = pre_clip8(a1), b = pre_clip8(a2);
    const size_t pix = (size_t)y * p.ow + x, hw = (size_t)p.oh * p.ow;
    if (p.out_u8) { p.out_u8[pix * 3] = (uint8_t)r; p.out_u8[pix * 3 + 1] = (uint8_t)g; p.out_u8[pix * 3 + 2] = (uint8_t)b; }
    // ToTensor: x/255 ; Normalize(0.5,0.5): (v-0.5)/0.5 ; Grayscale: 0.2989 R + 0.587 G + 0.114 B  (fp32, no contraction)
    {
#pragma clang fp contract(off)      // torch evaluates mul and add as separate fp32 ops: no FMA here
        const float fr = (float)r / 255.0f, fg = (float)g / 255.0f, fb = (float)b / 255.0f;
        if (p.out_rgb) {
            p.out_rgb[pix] = (fr - 0.5f) / 0.5f; p.out_rgb[hw + pix] = (fg - 0.5f) / 0.5f; p.out_rgb[2 * hw + pix] = (fb - 0.5f) / 0.5f;
        }
        if (p.out_gray) {
            const float m0 = 0.2989f * fr, m1 = 0.587f * fg, m2 = 0.114f * fb;
            const float s01 = m0 + m1;
            p.out_gray[pix] = s01 + m2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// f4 output step: the world point cloud of OnlineSLAM.save_data_all (slam.py:396-408):
//   local = K^-1 [x,y,1]^T * (depth * scale)   (compute_local_pointclouds, slam_utils.py:82-121)
//   world = pose [local,1]^T ; keep conf > thres ; colour = (img + 1) / 2.
// Stable stream compaction in view-major / row-major order (= torch boolean-mask order): per-block counts ->
// single-block exclusive scan -> emit.  Emits fp32 points / colours and/or packed little-endian PLY vertex records
// (3 x float64 + 3 x uint8 = 27 bytes, Open3D's binary layout for a coloured cloud).
struct CloudParams {
    const float* depth; const float* scale; const float* K; const float* pose; const float* conf; const float* img;
    int N, H, W; float thres;
    int* counts; int64_t* offs; int nblk;          // offs[nblk] = total
    float* pts; float* col; uint8_t* rec;
};
__global__ __launch_bounds__(256) void cloud_count_kernel(CloudParams p) {
    const int64_t total = (int64_t)p.N * p.H * p.W, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool keep = i < total && p.conf[i] > p.thres;
    const unsigned long long b = __ballot(keep);
    __shared__ int wc[4];
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) p.counts[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}
__global__ __launch_bounds__(1024) void cloud_scan_kernel(CloudParams p) {
    __shared__ int64_t part[1024];
    const int per = (p.nblk + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, p.nblk);
    int64_t s = 0;
    for (int i = lo; i < hi; ++i) s += p.counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int64_t run = 0; for (int i = 0; i < 1024; ++i) { const int64_t v = part[i]; part[i] = run; run += v; } p.offs[p.nblk] = run; }
    __syncthreads();
    int64_t run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { p.offs[i] = run; run += p.counts[i]; }
}
__global__ __launch_bounds__(256) void cloud_emit_kernel(CloudParams p) {
    const int64_t hw = (int64_t)p.H * p.W, total = p.N * hw, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool keep = i < total && p.conf[i] > p.thres;
    const unsigned long long b = __ballot(keep);
    __shared__ int wc[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wc[wave] = __popcll(b);
    __syncthreads();
    if (!keep) return;
    int rank = __popcll(b & ((1ull << lane) - 1));
    for (int w = 0; w < wave; ++w) rank += wc[w];
    const int64_t o = p.offs[blockIdx.x] + rank;
    const int n = (int)(i / hw), pix = (int)(i - n * hw), y = pix / p.W, x = pix - y * p.W;
    const float* K = p.K + n * 9;
    // 3x3 inverse by the adjugate (torch.inverse uses LU: same value to fp32 rounding)
    const float a = K[0], bb = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], k = K[8];
    const float A = e * k - f * h, B = -(d * k - f * g), C = d * h - e * g;
    const float idet = 1.0f / (a * A + bb * B + c * C);
    const float i00 = A * idet, i01 = -(bb * k - c * h) * idet, i02 = (bb * f - c * e) * idet;
    const float i10 = B * idet, i11 = (a * k - c * g) * idet, i12 = -(a * f - c * d) * idet;
    const float i20 = C * idet, i21 = -(a * h - bb * g) * idet, i22 = (a * e - bb * d) * idet;
    const float z = p.depth[i] * p.scale[n];
    const float fx = (float)x, fy = (float)y;
    const float lx = (i00 * fx + i01 * fy + i02) * z, ly = (i10 * fx + i11 * fy + i12) * z, lz = (i20 * fx + i21 * fy + i22) * z;
    const float* P = p.pose + n * 16;
    const float wx = P[0] * lx + P[1] * ly + P[2] * lz + P[3];
    const float wy = P[4] * lx + P[5] * ly + P[6] * lz + P[7];
    const float wz = P[8] * lx + P[9] * ly + P[10] * lz + P[11];
    float cr = 0.f, cg = 0.f, cb = 0.f;
    if (p.img) {
        const float* im = p.img + (int64_t)n * 3 * hw + pix;
        cr = (im[0] + 1.0f) / 2.0f; cg = (im[hw] + 1.0f) / 2.0f; cb = (im[2 * hw] + 1.0f) / 2.0f;
    }
    if (p.pts) { p.pts[o * 3] = wx; p.pts[o * 3 + 1] = wy; p.pts[o * 3 + 2] = wz; }
    if (p.col) { p.col[o * 3] = cr; p.col[o * 3 + 1] = cg; p.col[o * 3 + 2] = cb; }
    if (p.rec) {
        uint8_t* r = p.rec + o * 27;
        const double dv[3] = {(double)wx, (double)wy, (double)wz};
        const uint8_t* src = reinterpret_cast<const uint8_t*>(dv);
#pragma unroll
        for (int q = 0; q < 24; ++q) r[q] = src[q];
        r[24] = (uint8_t)rintf(fminf(fmaxf(cr, 0.f), 1.f) * 255.f);
        r[25] = (uint8_t)rintf(fminf(fmaxf(cg, 0.f), 1.f) * 255.f);
        r[26] = (uint8_t)rintf(fminf(fmaxf(cb, 0.f), 1.f) * 255.f);
    }
}

// pp.mat2SE3 (slam.py:166): [B,4,4] rigid transforms -> [B,7] = (tx,ty,tz, qx,qy,qz,qw), unit quaternion with the
// largest-magnitude component computed first (Shepperd) and qw >= 0.
__global__ void mat_to_se3_kernel(const float* pose, int B, float* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* P = pose + b * 16;
    const float m00 = P[0], m01 = P[1], m02 = P[2], m10 = P[4], m11 = P[5], m12 = P[6], m20 = P[8], m21 = P[9], m22 = P[10];
    float qx, qy, qz, qw;
    const float tr = m00 + m11 + m22;
    if (tr > 0.f) { const float s = sqrtf(tr + 1.f) * 2.f; qw = 0.25f * s; qx = (m21 - m12) / s; qy = (m02 - m20) / s; qz = (m10 - m01) / s; }
    else if (m00 > m11 && m00 > m22) { const float s = sqrtf(1.f + m00 - m11 - m22) * 2.f; qw = (m21 - m12) / s; qx = 0.25f * s; qy = (m01 + m10) / s; qz = (m02 + m20) / s; }
    else if (m11 > m22) { const float s = sqrtf(1.f + m11 - m00 - m22) * 2.f; qw = (m02 - m20) / s; qx = (m01 + m10) / s; qy = 0.25f * s; qz = (m12 + m21) / s; }
    else { const float s = sqrtf(1.f + m22 - m00 - m11) * 2.f; qw = (m10 - m01) / s; qx = (m02 + m20) / s; qy = (m12 + m21) / s; qz = 0.25f * s; }
    const float nrm = rsqrtf(qx * qx + qy * qy + qz * qz + qw * qw) * (qw < 0.f ? -1.f : 1.f);
    float* o = out + b * 7;
    o[0] = P[3]; o[1] = P[7]; o[2] = P[11]; o[3] = qx * nrm; o[4] = qy * nrm; o[5] = qz * nrm; o[6] = qw * nrm;
}

// Second half of a split-K GEMM / conv with the fp16-plane epilogue (small SLAM-scale grids): every K slice stored its
// fp32 partial tile to its own slab skbuf[s][M,N] (no atomics, fixed summation order); this sums them, applies bias, activation and the residual planes exactly like
// epilogue_tile<EPI_F16> and writes the blocked output planes.  One thread = 4 consecutive columns of one row.
template <bool SPLIT>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* skbuf, int ks, const float* bias, int M, int N, int act,
                                                            const f16* R1, const f16* R2, f16* C, int64_t c_rp, int r_mx, int c_mx, unsigned long long* rng) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n4 = N >> 2;
    if (i >= (int64_t)M * n4) return;
    const int row = (int)(i / n4), col = (int)(i - (int64_t)row * n4) * 4;
    float4 a = *reinterpret_cast<const float4*>(skbuf + (size_t)row * N + col);
    for (int s = 1; s < ks; ++s) {
        const float4 b = *reinterpret_cast<const float4*>(skbuf + ((size_t)s * M + row) * N + col);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    const size_t o = blk_off<SPLIT>(row, col, c_rp);
    H4 oh, ol;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e] + (bias ? bias[col + e] : 0.f);
        if (act == 1) x = gelu_erf(x);
        else if (act == 2) x = fmaxf(x, 0.f);
        if (SPLIT && r_mx) {          // residual planes in the f16mx row format
            if (R1) x += load_mx_act(R1, o + e);
            if (R2) x += load_mx_act(R2, o + e);
        } else {
            if (R1) { x += (float)R1[o + e]; if (SPLIT) x += (float)R1[o + 32 + e]; }
            if (R2) { x += (float)R2[o + e]; if (SPLIT) x += (float)R2[o + 32 + e]; }
        }
        v[e] = x;
        if (SPLIT) split_f16(x, oh.e[e], ol.e[e], rng); else oh.e[e] = to_f16_sat(x, rng);
    }
    if (SPLIT && c_mx) { store_mx4(C, o, split_mx4<false>(v, rng)); return; }
    *reinterpret_cast<uint2*>(C + o) = oh.u;
    if (SPLIT) *reinterpret_cast<uint2*>(C + o + 32) = ol.u;
}
