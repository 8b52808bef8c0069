// kernels/first_generation.hip.h -- the first-generation walk (every lane on its own under a wave scheduler): A/B build only
// (part of the one translation unit kernels.hip; see its head for the passes and the arithmetic contract)
#pragma once
#include "common.hip.h"

namespace ocrt {

#ifdef OCRT_DEBUG_KNOBS  // ---- first generation (A/B build only): every lane walks on its own under a wave scheduler ----
// ---------------------------------------------------------------------------
// Wave-scheduled traversal.
//
// A lane walks nodes (T), has hit leaves pending their triangle test (L) or has
// no ray (I).  Instead of letting every lane run its own nested loops -- where
// the wave pays for the longest ray and a triangle test runs with a handful of
// live lanes -- the wave picks, per iteration and with scalar ballots only, the
// one body worth running: refill idle lanes from the ray queue, run the triangle
// test for the lanes with a pending leaf, or advance the walking lanes by one
// node.  Each body is straight-line and predicated, so exec-mask bookkeeping
// stays out of the hot loop.  A lane keeps up to TWO pending leaves (a FIFO, so
// the reference's ascending leaf order of the tests is preserved) and goes on
// walking while the second slot is free: lanes rarely block on a triangle test,
// and the tests run with more lanes at once.
// ---------------------------------------------------------------------------
struct Pending {
	uint32_t first, second;  // leaf indices in the order they were met; NONE = free
};
__device__ __forceinline__ bool can_walk(const Pending &p, uint32_t i, uint32_t count) { return p.second == NONE && i < count; }
// Thresholds (KernelParams::refill_min / leaf_min, 16 each): refill once that many
// lanes are idle, run the triangle tests once that many leaves are pending.


// Largest magnitude for which (b - o) cannot overflow.  A ray is "regular" when
// its origin and its reciprocal direction are finite and within it (so no
// inf * 0, no inf - inf); for any other ray the reference's own select-based
// slab test is used instead of the min/max form.
constexpr float REGULAR_LIMIT = 1.0e37f;

__device__ __forceinline__ bool ray_is_regular(const Ray &r) {
	return fabsf(r.ox) <= REGULAR_LIMIT && fabsf(r.oy) <= REGULAR_LIMIT && fabsf(r.oz) <= REGULAR_LIMIT &&
	       fabsf(r.ix) <= REGULAR_LIMIT && fabsf(r.iy) <= REGULAR_LIMIT && fabsf(r.iz) <= REGULAR_LIMIT;
}

// min/max form of the slab test.  For a regular ray against a regular box
// (finite, lo <= hi) no NaN can arise, (lo-o)*inv and (hi-o)*inv are ordered by
// the sign of inv (IEEE rounding is monotonic), and the reference's chain of
// early-outs (src/intersect_kernel.cl:21-61) reduces to
//   max(near) <= min(far)  &&  max(near) < max_distance  &&  min(far) > 0,
// the same comparisons on the same values.  With below = pred(max_distance) and
// tiny = the smallest positive float, that is  max(near, tiny) <= min(far, below).
__device__ __forceinline__ bool slab_hit_regular(const float4 lo, const float4 hi, const Ray &r, float below) {
	const float x0 = (lo.x - r.ox) * r.ix, x1 = (hi.x - r.ox) * r.ix;
	const float y0 = (lo.y - r.oy) * r.iy, y1 = (hi.y - r.oy) * r.iy;
	const float z0 = (lo.z - r.oz) * r.iz, z1 = (hi.z - r.oz) * r.iz;
	const float tiny = __uint_as_float(1u);
	const float t_near = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tiny));
	const float t_far = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), below));
	return t_near <= t_far;
}

// One node for a lane in state T: box hit -> next node in pre-order (and the leaf,
// if it is one, becomes pending); miss -> skip the subtree.  Inner nodes carry
// leaf == NONE, so no leaf/inner branch is needed.
template <bool REGULAR>
__device__ __forceinline__ void node_step(__amdgpu_buffer_rsrc_t nodes, const Ray &r, float max_distance,
                                          float below, uint32_t &i, Pending &pending) {
	const float4 lo = load_f4(nodes, i * 32u);
	const float4 hi = load_f4(nodes, i * 32u + 16u);
	const bool hit = REGULAR ? slab_hit_regular(lo, hi, r, below) : slab_hit(lo, hi, r, max_distance);
	const uint32_t leaf = hit ? __float_as_uint(hi.w) : NONE;  // NONE unless a leaf's box was hit
	const bool empty = pending.first == NONE;                  // (this lane walks, so `second` is free)
	pending.second = empty ? NONE : leaf;
	pending.first = empty ? leaf : pending.first;
	i += hit ? 1u : __float_as_uint(lo.w);
}

// Advances the lanes in state T by one node: the min/max slab form when every
// walking lane's ray is regular (the common case), the reference's own form otherwise.
__device__ __forceinline__ void advance_walkers(const SceneViews &scene, const Ray &r, bool regular, float max_distance,
                                                float below, uint32_t count, uint32_t &i, Pending &pending) {
	const bool walking_lane = can_walk(pending, i, count);
	const bool all_regular = wave_ballot(walking_lane && !regular) == 0ull;
	if (walking_lane) {
		if (all_regular)
			node_step<true>(scene.nodes, r, max_distance, below, i, pending);
		else
			node_step<false>(scene.nodes, r, max_distance, below, i, pending);
	}
}

#endif  // OCRT_DEBUG_KNOBS

}  // namespace ocrt
