// Shared device helpers for the MI355X (gfx950 / CDNA4) STA frontend kernels.
// Wave = 64 lanes everywhere; no CUDA/other-arch paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef f16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define STA_F16_MAX 65504.0f

// Range report (sta_range_report): how many values did not fit the fp16 planes (|x| > 65504) and how many fp8 correction
// bytes of the f16mx arithmetic saturated (activations: e5m2, +-57344; weights: e4m3, |w| > 28), since the last reset.  The planes SATURATE instead of
// producing inf; a non-zero count says the result is no longer backed by the parity goldens.  Rare path: one compare per value.
// A writer tracks the largest magnitude it stored in a RangeAcc (ONE v_max_f32 per value, no compare, no branch) and
// flushes once per tile / row / call: the counters count (lane, flush) events with at least one out-of-range value, not values.
// The running maximum is kept on the BIT PATTERN of |x| (v_and + v_max_u32): as unsigned integers NaN (0x7fc00000) > inf
// (0x7f800000) > every finite value, so a NaN is sticky and counts as a range event too (a float max would drop it, and the
// clamp below turns it into -65504: silent).  A non-finite ROW of a residual stream is also counted by the LayerNorm kernels.
// The two counters live in the handle (sta_handle::range, 16 B of device memory): every kernel that reports gets the pointer
// (GemmParams::range, LnParams::range, AttnParams::range or a trailing `rng` argument).
#define STA_F16_MAX_BITS 0x477fe000u      // bits of 65504.0f
#define STA_E5M2_MAX_BITS 0x47600000u     // bits of 57344.0f
struct RangeAcc {
    unsigned amax = 0u;     // bits of the largest |x| written as an fp16 (hi, residual) pair
    unsigned amax8 = 0u;    // the same for values whose fp8 (e5m2) copy was written (f16mx activation rows: saturates beyond 57344)
    bool w8 = false;        // a WEIGHT e4m3 byte saturated (packing at load time: exact check)
    __device__ __forceinline__ void flush(unsigned long long* rng) {
        if (__builtin_expect(amax > STA_F16_MAX_BITS, 0)) atomicAdd(rng, 1ull);
        if (__builtin_expect(amax8 > STA_E5M2_MAX_BITS || w8, 0)) atomicAdd(rng + 1, 1ull);
        amax = amax8 = 0u; w8 = false;
    }
};
__device__ __forceinline__ float sat_f16_range(float x, RangeAcc& a) {
    const unsigned ax = __float_as_uint(x) & 0x7fffffffu;
    a.amax = ax > a.amax ? ax : a.amax;
    return fminf(fmaxf(x, -STA_F16_MAX), STA_F16_MAX);
}

// 2-term fp16 split: x ~= hi + lo with |lo| <= ulp(hi)/2  (~21 significant bits while lo is a
// normal fp16, absolute floor 2^-25 below that).  Saturates instead of producing inf.
__device__ __forceinline__ void split_f16(float x, f16& hi, f16& lo, RangeAcc& ra) {
    x = sat_f16_range(x, ra);
    hi = (f16)x;
    lo = (f16)(x - (float)hi);
}
__device__ __forceinline__ void split_f16(float x, f16& hi, f16& lo, unsigned long long* rng) { RangeAcc ra; split_f16(x, hi, lo, ra); ra.flush(rng); }
__device__ __forceinline__ f16 to_f16_sat(float x, RangeAcc& ra) { return (f16)sat_f16_range(x, ra); }
__device__ __forceinline__ f16 to_f16_sat(float x, unsigned long long* rng) { RangeAcc ra; const f16 r = to_f16_sat(x, ra); ra.flush(rng); return r; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// exact-erf GELU (nn.GELU default, sta_blocks.py:60)
// erfc(|z|) by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32-rounding class) - 14 VALU ops
// instead of the ~45 of the branchy libm erff; evaluated on the erfc side so the negative tail has no
// 1 + erf cancellation.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float ec = poly * __expf(-az * az);          // erfc(|z|)
    return 0.5f * x * (z >= 0.f ? 2.0f - ec : ec);
}

__device__ __forceinline__ uint4 ldg16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint2 ldg8(const void* p) { return *reinterpret_cast<const uint2*>(p); }

// K-tile-blocked plane layout of every fp16 activation / weight tensor:  [cols/32][rows][hi32 | lo32]
// (f16 mode: [cols/32][rows][32]).  Element (row, col): hi at blk_off, lo 32 elements further.
// One K tile (32 columns) of consecutive rows is contiguous memory - the unit the GEMM DMA moves.
template <bool SPLIT>
__device__ __forceinline__ size_t blk_off(int64_t row, int col, int64_t rows) {
    return ((size_t)(col >> 5) * rows + row) * (SPLIT ? 64 : 32) + (col & 31);
}

// ---------------------------------------------------------------------------------------------------------
// Precision f16mx ("f16 main product + block-scaled fp8 corrections"): x ~= hi + lo as in f16x3, but the two correction
// products Al*Bh + Ah*Bl are issued as ONE v_mfma_scale_f32_32x32x64_f8f6f4 (2x the f16 MFMA rate) on fp8 copies.
// Row block of 32 k (128 B: same size and the same blocked layout as f16x3):
//     [ hi f16 x32 (64 B) | 32 byte pairs (64 B) ]
//   activation pair k = ( e5m2(hi * 2^0),  e5m2(lo * 2^11) )        (|lo| <= 2^-11 |hi|; e5m2 since round 3, see cvt2_fp8)
//   weight     pair k = ( e4m3(lo * 2^15), e4m3(hi * 2^4)  )        (|w| << 1) - the OPPOSITE order, so that byte q of
//   an activation row meets byte q of a weight row as hi8 x lo8 / lo8 x hi8, i.e. exactly the two correction products;
//   both carry the same combined scale 2^-(0+15) = 2^-(11+4) = 2^-15, applied by the instruction's E8M0 block scales.
// Relative error of a GEMM ~2e-5 with e5m2 activation bytes, ~1e-5 with e4m3 (f16x3: 9e-7, one-product f16: 3e-4) - measured in tests/test_gpu_kernels.py.
#define STA_MX_A_SHI 0
#define STA_MX_A_SLO 11
#define STA_MX_W_SHI 4
#define STA_MX_W_SLO 15
static_assert(STA_MX_A_SHI + STA_MX_W_SLO == STA_MX_A_SLO + STA_MX_W_SHI, "both correction products must share one scale");

// fp8 formats of the correction bytes: WEIGHTS e4m3 (|w * 2^4| <= 448, 3 mantissa bits: static, small values); ACTIVATIONS
// e5m2 ("bf8", +-57344 - the fp16 planes' own range - at 2 mantissa bits: round 3, after the outlier goldens showed DPT feature
// maps past +-448 saturating the e4m3 copy and the head's error growing from 2e-5 to 3e-4).  The MFMA takes one format per
// operand (cbsz = 1: A is bf8, blgp = 0: B is fp8).
#define STA_MX_A_MAX 57344.f
#define STA_MX_W_MAX 448.f
__device__ __forceinline__ float clamp_e4m3(float x) { return fminf(fmaxf(x, -STA_MX_W_MAX), STA_MX_W_MAX); }
__device__ __forceinline__ float clamp_e5m2(float x) { return fminf(fmaxf(x, -STA_MX_A_MAX), STA_MX_A_MAX); }
// (first, second) -> two fp8 bytes (RNE, saturating) in the low / high 16 bits of `old`: OCP e4m3 (weights) or e5m2 (activations)
template <bool WEIGHT>
__device__ __forceinline__ int cvt2_fp8(float first, float second, int old, bool high) {
    if (WEIGHT) return high ? __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(first), clamp_e4m3(second), old, true)
                            : __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(first), clamp_e4m3(second), old, false);
    return high ? __builtin_amdgcn_cvt_pk_bf8_f32(clamp_e5m2(first), clamp_e5m2(second), old, true)
                : __builtin_amdgcn_cvt_pk_bf8_f32(clamp_e5m2(first), clamp_e5m2(second), old, false);
}
struct MX4 { uint2 hi; uint2 pairs; };
template <bool WEIGHT>
__device__ __forceinline__ MX4 split_mx4(const float y[4], RangeAcc& ra) {
    constexpr float KHI = WEIGHT ? (float)(1 << STA_MX_W_SHI) : (float)(1 << STA_MX_A_SHI);
    constexpr float KLO = WEIGHT ? (float)(1 << STA_MX_W_SLO) : (float)(1 << STA_MX_A_SLO);
    union { uint2 u; f16 e[4]; } h;
    float a[4], b[4];            // first / second byte of every pair
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = sat_f16_range(y[e], ra);
        h.e[e] = (f16)x;
        const float hf = (float)h.e[e], lf = x - hf;
        a[e] = WEIGHT ? lf * KLO : hf * KHI;
        b[e] = WEIGHT ? hf * KHI : lf * KLO;
        // activations: |lo * 2^11| <= |hi|, so the pair saturates iff |hi| > 57344; weights (load time): exact check
        if (WEIGHT) ra.w8 |= fabsf(a[e]) > STA_MX_W_MAX || fabsf(b[e]) > STA_MX_W_MAX;
    }
    if (!WEIGHT) ra.amax8 = ra.amax;      // (this accumulator only ever sees f16mx rows)
    MX4 r; r.hi = h.u;
    int w0 = cvt2_fp8<WEIGHT>(a[0], b[0], 0, false); w0 = cvt2_fp8<WEIGHT>(a[1], b[1], w0, true);
    int w1 = cvt2_fp8<WEIGHT>(a[2], b[2], 0, false); w1 = cvt2_fp8<WEIGHT>(a[3], b[3], w1, true);
    r.pairs = make_uint2((unsigned)w0, (unsigned)w1);
    return r;
}
template <bool WEIGHT>
__device__ __forceinline__ MX4 split_mx4(const float y[4], unsigned long long* rng) { RangeAcc ra; const MX4 r = split_mx4<WEIGHT>(y, ra); ra.flush(rng); return r; }
// o = blk_off<true>(row, col, rows) with col % 4 == 0: hi at base + o, the byte pairs in the second half of the row block
__device__ __forceinline__ void store_mx4(f16* base, size_t o, const MX4& v) {
    *reinterpret_cast<uint2*>(base + o) = v.hi;
    *reinterpret_cast<uint2*>(base + o + 32) = v.pairs;          // +64 B, 2 B per element: same offset arithmetic as the lo plane
}
// value of an f16mx ACTIVATION element: hi + lo8 * 2^-11 (readers outside the GEMMs: residual adds, bilinear, head)
__device__ __forceinline__ float load_mx_act(const f16* base, size_t o) {
    const int pair = reinterpret_cast<const unsigned short*>(base)[o + 32];
    return (float)base[o] + __builtin_amdgcn_cvt_f32_bf8(pair, 1) * (1.0f / (float)(1 << STA_MX_A_SLO));
}
template <bool WEIGHT>
__device__ __forceinline__ void split_mx1(float x, RangeAcc& ra, f16& h, unsigned short& pair) {     // scalar variant (column-per-lane epilogues)
    constexpr float KHI = WEIGHT ? (float)(1 << STA_MX_W_SHI) : (float)(1 << STA_MX_A_SHI);
    constexpr float KLO = WEIGHT ? (float)(1 << STA_MX_W_SLO) : (float)(1 << STA_MX_A_SLO);
    x = sat_f16_range(x, ra);
    h = (f16)x; const float hf = (float)h, lf = x - hf;
    if (WEIGHT) ra.w8 |= fabsf(lf * KLO) > STA_MX_W_MAX || fabsf(hf * KHI) > STA_MX_W_MAX; else ra.amax8 = ra.amax;
    const int b = WEIGHT ? cvt2_fp8<true>(lf * KLO, hf * KHI, 0, false) : cvt2_fp8<false>(hf * KHI, lf * KLO, 0, false);
    pair = (unsigned short)(b & 0xFFFF);
}
template <bool WEIGHT>
__device__ __forceinline__ void store_mx1(f16* base, size_t o, float x, RangeAcc& ra) {
    f16 h; unsigned short pair;
    split_mx1<WEIGHT>(x, ra, h, pair);
    base[o] = h;
    reinterpret_cast<unsigned short*>(base)[o + 32] = pair;
}
template <bool WEIGHT>
__device__ __forceinline__ void store_mx1(f16* base, size_t o, float x, unsigned long long* rng) { RangeAcc ra; store_mx1<WEIGHT>(base, o, x, ra); ra.flush(rng); }

union H8 { uint4 u; half8 h; f16 e[8]; };
union H4 { uint2 u; half4 h; f16 e[4]; };
