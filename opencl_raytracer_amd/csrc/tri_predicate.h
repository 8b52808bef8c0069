// tri_predicate.h -- the parametric half of the reference's triangle test (src/intersect_kernel.cl:92-103) as a
// PREDICATE, for rays that only ask "is there an accepted triangle" (the ambient-occlusion pass: reference :237-255
// looks at scene_intersect's bool and nothing else).
//
// The reference decides with two IEEE divisions by the triangle's D = uv*uv - uu*vv:
//     s = X / D;  t = Y / D;  accept unless  s < -0.00001f | s > 1.00001 | t < -0.00001f | s + t > 1.00001
// (X = uv*wv - vv*wu, Y = uv*wu - uu*wv; 1.00001 is a double, the compare is exact: s > 0x3F800053 as a float).
// A correctly rounded division is 11 vector instructions on gfx950; two of them are a quarter of a triangle test.
// Here the decision is taken on  s' = X * inv_d, t' = Y * inv_d  (inv_d = RN(1 / D), made on the host next to D:
// TriRec::inv_d) wherever s', t' and s' + t' are farther from the thresholds than the two forms can differ, and
// by the reference's own divisions everywhere else -- so every decision is the reference's.
//
// How far they can differ.  D and 1/D normal (else inv_d is NaN: see below), RN = round to nearest:
//     s' = RN(X * RN(1/D))  = (X/D)(1 + d1)(1 + d2),   s = RN(X / D) = (X/D)(1 + d3),   |di| <= 2^-24
// (+- 2^-149 where a result is subnormal), so |s' - s| <= |X/D| * 3.01 * 2^-24 < |X/D| * 1.8e-7: below 1.5e-6 while
// |X/D| <= 8, and beyond 8 both are on the same side of every threshold (same sign, magnitude > 7.9).  The sums:
// |RN(s' + t') - RN(s + t)| <= |s' - s| + |t' - t| + 2^-23 * max|sum| < 5e-6 while |X/D|, |Y/D| <= 8; with one of
// them beyond 8 the reference rejects (s or t below -7.9, s above 7.9, or t above 7.9 and then either s < -0.00001 or
// s + t > 7.8) and so does the out-zone test below.  The bands are 1e-5 / 3e-5 wide on either side: 6x the bounds.
//     IN  (accept):  min(s', t') >= 0         and  max(s' + t', s') <= 0.99998
//     OUT (reject):  min(s', t') <  -0.00002  or   max(s' + t', s') >  1.00004
//     neither: the exact divisions (about one test in 10^4).
// IN implies 0 <= s', t' <= 1: the bounds apply, s >= -1.5e-6, t likewise, s <= 0.99998 + 1.5e-6, s + t <= 0.99998 +
// 5e-6: accepted by the reference.  OUT: whichever of the four comparisons holds, the reference's counterpart holds
// by the bounds (or by the magnitude argument beyond 8).
// NaN and infinity: v_min_f32 / v_max_f32 return the other operand for a quiet NaN (IEEE minNum / maxNum; arithmetic
// results are quiet), a NaN compares false.  X or Y infinite or NaN gives the same class in both forms (a product by
// a finite non-zero inv_d); walking through the cases -- s' NaN: max(NaN, NaN) = NaN, neither zone, exact path; t'
// NaN alone: zones decided by s' only, and the reference's t and s + t comparisons are all false too; +-infinity: OUT
// exactly where the reference rejects.  A triangle whose D is zero, subnormal, beyond 1e30 or not a number gets
// inv_d = NaN on the host: s' and t' are NaN, every such test takes the exact path.
// Checked on the device over 2^32 triples crowded around the thresholds and zone edges (tests/tri_predicate_check.hip,
// tests/test_hip_parity.py::test_triangle_zones_never_contradict_the_divisions), on the host for what pack_scene stores
// (tests/tri_inverse_check.cc), and by every golden frame: a wrong decision is an occlusion count off by one.
#pragma once

namespace ocrt {

// The zones, in units of s and t.
constexpr float TRI_IN_LOW = 0.0f, TRI_IN_HIGH = 0.99998f, TRI_OUT_LOW = -0.00002f, TRI_OUT_HIGH = 1.00004f;
// The reference's thresholds: -0.00001f, and the smallest float above the double 1.00001.
constexpr float TRI_LOW = -0.00001f;
constexpr unsigned int TRI_HIGH_BITS = 0x3F800053u;

#ifdef __HIPCC__
// The reference's decision, by its own arithmetic.
__device__ __forceinline__ bool tri_accepts_exact(float X, float Y, float D) {
	const float high = __uint_as_float(TRI_HIGH_BITS);
	const float s = X / D;
	const float t = Y / D;
	return !((s < TRI_LOW) | (s > high) | (t < TRI_LOW) | ((s + t) > high));
}

// 0 = rejected, 1 = accepted, 2 = too close to a threshold to say (tri_accepts_exact decides).
__device__ __forceinline__ unsigned int tri_zone(float X, float Y, float inv_d, float in_low = TRI_IN_LOW, float in_high = TRI_IN_HIGH,
                                                 float out_low = TRI_OUT_LOW, float out_high = TRI_OUT_HIGH) {
	const float s = X * inv_d, t = Y * inv_d;
	const float low = __builtin_fminf(s, t), high = __builtin_fmaxf(s + t, s);
	const bool in = (low >= in_low) & (high <= in_high);
	const bool out = (low < out_low) | (high > out_high);
	return in ? 1u : out ? 0u : 2u;
}
#endif

// Host side: what goes into TriRec::inv_d.
#ifdef __HIPCC__
__host__ __device__
#endif
inline float tri_inverse_d(float D) {
	const float magnitude = D < 0.0f ? -D : D;
	if (!(magnitude >= 1.0e-30f && magnitude <= 1.0e30f))  // (false for a NaN too)
		return __builtin_nanf("");
	return 1.0f / D;
}

}  // namespace ocrt
