// exact_reciprocal.h -- 1.0f / x, correctly rounded, in three vector instructions instead of the eleven of the
// compiler's IEEE division, for the x it is proven for.  (The reference's slab test multiplies by 1 / direction,
// src/intersect_kernel.cl:21-61: the reciprocals are part of the arithmetic contract, bit for bit.)
//
//     y0 = v_rcp_f32(x)            (1 ulp)
//     e  = fma(-x, y0, 1)
//     y1 = fma(y0, e, y0)
//
// y1 == RN(1 / x) for EVERY float x whose biased exponent is 1 ... 252, both signs: checked exhaustively on the
// MI355X against the compiler's division (tests/test_hip_parity.py::test_fast_reciprocal_is_exact_for_every_float,
// tests/reciprocal_check.hip: all 2^32 bit patterns, seconds).  Outside that range -- zeros and subnormals (v_rcp_f32
// flushes them), |x| >= 2^126 (the result is subnormal), infinities, NaNs -- the short form is wrong, and the caller
// must divide: reciprocals_are_short() says whether three numbers all qualify.
#pragma once

namespace ocrt {

#ifdef __HIPCC__
__device__ __forceinline__ float short_reciprocal(float x) {
	const float y0 = __builtin_amdgcn_rcpf(x);
	const float e = __builtin_fmaf(-x, y0, 1.0f);
	return __builtin_fmaf(y0, e, y0);
}

// x, y and z are normal numbers (class tests: one instruction each) below 2^126 in magnitude.
__device__ __forceinline__ bool reciprocals_are_short(float x, float y, float z) {
	constexpr int NORMAL = (1 << 3) | (1 << 8);  // negative normal, positive normal
	const float largest = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x), __builtin_fabsf(y)), __builtin_fabsf(z));
	return __builtin_amdgcn_classf(x, NORMAL) & __builtin_amdgcn_classf(y, NORMAL) & __builtin_amdgcn_classf(z, NORMAL) &
	       (largest < 0x1p126f);
}
#endif

}  // namespace ocrt
