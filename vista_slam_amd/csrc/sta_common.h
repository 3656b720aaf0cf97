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

// 2-term fp16 split: x ~= hi + lo with |lo| <= ulp(hi)/2  (~21 significant bits while lo is a
// normal fp16, absolute floor 2^-25 below that).  Saturates instead of producing inf.
__device__ __forceinline__ void split_f16(float x, f16& hi, f16& lo) {
    x = fminf(fmaxf(x, -STA_F16_MAX), STA_F16_MAX);
    hi = (f16)x;
    lo = (f16)(x - (float)hi);
#ifdef STA_EMU_LO_MANT
    // experiment only (never defined in the product build): keep STA_EMU_LO_MANT explicit mantissa bits of the
    // residual plane, i.e. emulate an fp8-class `lo` to measure what a 2-unit (f16 + MX-fp8 correction) scheme would cost
    {
        unsigned u = __float_as_uint(x - (float)hi);
        const int drop = 23 - STA_EMU_LO_MANT;
        u += 1u << (drop - 1);
        u &= ~((1u << drop) - 1u);
        lo = (f16)__uint_as_float(u);
    }
#endif
}
__device__ __forceinline__ f16 to_f16_sat(float x) {
    return (f16)fminf(fmaxf(x, -STA_F16_MAX), STA_F16_MAX);
}

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

union H8 { uint4 u; half8 h; f16 e[8]; };
union H4 { uint2 u; half4 h; f16 e[4]; };
