// kernels.hip -- gfx950 ray-casting kernels (the replacement for reference
// src/intersect_kernel.cl).
//
// Arithmetic contract: IEEE binary32 + - * / sqrt in the
// reference's operation order, NO fma contraction (this file is compiled with
// -ffp-contract=off), IEEE maxNum/minNum for max/min, double-literal
// comparisons folded to their exact float thresholds.  Anything else (data
// layout, traversal order, where a value is computed) is free and is chosen
// for the CDNA4 wave64 machine.
//
// Work decomposition: one 64-lane wavefront per 8x8 tile of sub-pixels, in two ray passes and a finishing sweep --
// three kernels per frame (round 3: six), captured once per host as a hipGraph and replayed.
//   primary_kernel  every lane casts its primary ray (closest hit) and computes
//                   the smooth normal and head-light term.  Sub-pixels that need
//                   no ambient occlusion are final; the tile's other hits leave a TAG in the image and
//                   are ballot-compacted into the tile's slots of the hit list (tile_base: sized by what is hit).
//                   Its TAIL is the ordering step: the last workgroup of each XCD group to finish sorts the
//                   group's non-empty tiles by AO cost class (counting sort, the costly blocks first).
//   ao_kernel       persistent workgroups claim runs of (tile, table direction) units in
//                   that order -- a tile at a time, whose directions the four waves take
//                   from a cursor in LDS.  A wave rebuilds the tile's tangent frames in its
//                   LDS slice and casts one packet of 64 any-hit rays per
//                   direction -- one table direction from the tile's neighbouring
//                   surface points -- that stop at the first accepted triangle and walk only
//                   the interval of the node array their segments can reach (entry_kernel, once per upload);
//                   occlusion counts are LDS atomics, flushed to a per-hit counter
//                   when the claim is done.
//   finish_kernel / finish_wide_kernel
//                   value * (1 - occluded / n) into the tagged sub-pixels and the supersample box filter +
//                   quantisation of the same sweep (n = 1: a thread per pixel; n >= 2: along the sub-pixel rows).
//   (on demand)     entry_kernel: the walk intervals, once per upload; occluded_sum_kernel: the frame's occlusion
//                   total when the statistics are asked for; resize_kernel: a box filter on its own.
// Why not one fused launch (it was, see profiles/r01_notes.md): cost per tile
// varies 30x (background vs model, 29 rays per hit sub-pixel), so the frame used
// to end on a long tail of half-empty CUs.  With the tiles' costs known after the
// primary pass, claiming the costly blocks first packs them almost perfectly.
//
// How rays walk the tree: the 64 rays of a wave share ONE node index ("shared
// walk", see walk_collect below) -- nodes and triangles arrive by
// scalar loads, boxes are tested out of SGPRs, nothing diverges and nothing is
// gathered.  The first generation, in which every lane walked on its own under a
// wave scheduler, is only compiled into the A/B build (-DOCRT_DEBUG_KNOBS, where
// OCRT_NO_SHARED_WALK=1 selects it); the product library does not contain it.
//
// What bounds it: the scene (19 MB) is cache-resident, HBM traffic is negligible;
// the walk is bound by vector-instruction issue (11 to 17 per node and primary packet, 12 per any-hit packet) and the
// latency of the one scalar load per pair of nodes (scenes beyond the caches: by that latency alone) -- see DESIGN.md section 5 and
// profiles/r0*_notes.md for the counters and the microbenchmarks.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_types.h"
#include "exact_reciprocal.h"
#include "tri_predicate.h"

namespace ocrt {

namespace {

struct Ray {
	float ox, oy, oz;
	float dx, dy, dz;
	float ix, iy, iz;  // 1.0f / d, hoisted out of the per-node slab test
};

// What the fast form of the shared walk keeps per lane: t = fma(plane, i, oi).  An infinite reciprocal (a zero
// direction component) is replaced by +-2^100: with inf the fma would be inf - inf = NaN for every box, the axis
// would drop out of the test and the ray would "hit" every box the other two axes allow (a third of the frame's
// ambient-occlusion rays start on an axis-aligned ground plane, three of their 28 directions have a zero
// component).  The reference's test on such an axis says "the origin's coordinate lies in the box's slab"
// ((b - o) * inf is +-inf by the sign of b - o, NaN -- dropped -- for b == o); 2^100 (b' - o) has the sign of
// b' - o, which the outward margin keeps on the conservative side (padded_bound: the argument holds for every finite
// reciprocal), and is either <= 0 or far above any max_distance.  (|o|, |b'| <= ~1e6: no overflow.)
struct WalkRay {
	float ix, iy, iz;
	float oix, oiy, oiz;  // -(o * i), rounded once
};
__device__ __forceinline__ float walk_reciprocal(float i) {
	return fabsf(i) == __builtin_inff() ? copysignf(0x1.0p+100f, i) : i;
}
// `scale`: 1 for the plain form (t in ray units); the SCALED form of the node test (walk_collect<true>) measures t in
// units of the ray's max_distance, scale = KernelParams::walk_scale ~ 1 / max_distance.
// `tame`: no reciprocal is infinite (ray_is_tame held for the packet).
__device__ __forceinline__ WalkRay make_walk_ray(const Ray &r, float scale, bool tame = false) {
	WalkRay w;
	w.ix = (tame ? r.ix : walk_reciprocal(r.ix)) * scale;
	w.iy = (tame ? r.iy : walk_reciprocal(r.iy)) * scale;
	w.iz = (tame ? r.iz : walk_reciprocal(r.iz)) * scale;
	w.oix = -(r.ox * w.ix);
	w.oiy = -(r.oy * w.iy);
	w.oiz = -(r.oz * w.iz);
	return w;
}

struct Hit {
	float distance;
	uint32_t leaf;
	float s, t;
	float px, py, pz;
};

// The reference keeps, among the nearest accepted triangles, the first in ITS leaf order
// (`best.distance > distance`, strict, src/intersect_kernel.cl:107).  The tree walked here
// may list the leaves in another order (walk_tree.h), so the tie is decided by the leaf
// number itself.  (`best` starts at distance +inf: a hit at +inf or NaN never replaces it.)
__device__ __forceinline__ bool nearer(float distance, uint32_t leaf, const Hit &best) {
	return best.distance > distance || (best.distance == distance && leaf < best.leaf && distance < __builtin_inff());
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
	return (ax * bx + ay * by) + az * bz;
}

__device__ __forceinline__ Ray make_ray(float ox, float oy, float oz, float dx, float dy, float dz) {
	Ray r;
	r.ox = ox; r.oy = oy; r.oz = oz;
	r.dx = dx; r.dy = dy; r.dz = dz;
	// 1 / direction: the short form where every lane's three components allow it (exact_reciprocal.h: the same bits as
	// the division, proven for every such float), the division for the packet otherwise (a zero component, say)
	if (__builtin_amdgcn_ballot_w64(!reciprocals_are_short(dx, dy, dz)) == 0ull) {
		r.ix = short_reciprocal(dx);
		r.iy = short_reciprocal(dy);
		r.iz = short_reciprocal(dz);
	} else {
		r.ix = 1.0f / dx;
		r.iy = 1.0f / dy;
		r.iz = 1.0f / dz;
	}
	return r;
}

// Slab test, reference src/intersect_kernel.cl:21-61.  The reference's early
// returns only skip work; evaluating everything and AND-ing the same
// comparisons (kept in their original `a > b` polarity for NaN) is identical.
__device__ __forceinline__ bool slab_hit(const float4 lo, const float4 hi, const Ray &r, float max_distance) {
	const bool px = r.ix >= 0.0f, py = r.iy >= 0.0f, pz = r.iz >= 0.0f;
	float t_min = ((px ? lo.x : hi.x) - r.ox) * r.ix;
	float t_max = ((px ? hi.x : lo.x) - r.ox) * r.ix;
	const float ty_min = ((py ? lo.y : hi.y) - r.oy) * r.iy;
	const float ty_max = ((py ? hi.y : lo.y) - r.oy) * r.iy;
	bool miss = (t_min > ty_max) | (ty_min > t_max);
	t_min = fmaxf(t_min, ty_min);
	t_max = fminf(t_max, ty_max);
	const float tz_min = ((pz ? lo.z : hi.z) - r.oz) * r.iz;
	const float tz_max = ((pz ? hi.z : lo.z) - r.oz) * r.iz;
	miss |= (t_min > tz_max) | (tz_min > t_max);
	t_min = fmaxf(t_min, tz_min);
	t_max = fminf(t_max, tz_max);
	return !miss & (t_min < max_distance) & (t_max > 0.0f);
}

constexpr uint32_t NONE = 0xFFFFFFFFu;

// Orders this wave's LDS writes before its later LDS reads.  A wave executes in
// lockstep and the LDS unit serves one wave's requests in order, so only the
// compiler must be kept from reordering across this point.
__device__ __forceinline__ void wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// The lane number recomputed (two instructions) where it is needed, opaque to the optimiser (which would otherwise
// compute it once and hold it in a register across the walks).
__device__ __forceinline__ uint32_t fresh_lane() {
	uint32_t lane;
	asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
	return lane;
}

// Position of this lane among the set bits of `mask` below it.
__device__ __forceinline__ uint32_t rank_in(unsigned long long mask) {
	return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

// The lanes' predicate as a 64-bit mask, straight from the compare (HIP's __ballot goes through an int).
__device__ __forceinline__ unsigned long long wave_ballot(bool predicate) { return __builtin_amdgcn_ballot_w64(predicate); }

// What the shared walk's fast form needs of a ray: a finite origin within the limit
// and reciprocal directions that are numbers (infinite is fine -- a zero direction
// component -- as long as not all three are).  Its slab test picks near and far by
// the sign of the reciprocal like the reference's does (`inv >= 0 ? lo : hi`), so
// (b - o) * inf behaves there exactly as in the reference: -inf / +inf order
// themselves, and the NaN of 0 * inf is dropped by maxNum / minNum here as it is
// dropped by the reference's `t_min > ty_max` comparisons and fmax / fmin updates.
// With all three reciprocals infinite nothing would be left to compare (the
// reference then rejects on `NaN < max_distance`), hence the exclusion.
// `origin_limit` (KernelParams): the magnitude up to which the outward margin of the padded walk boxes covers the
// rounding of the fma form; a finite reciprocal must stay below 1e30 so that o * inv cannot overflow (scene_pack.cc,
// padded_bound).
constexpr float RECIPROCAL_LIMIT = 1.0e30f;
__device__ __forceinline__ bool ray_is_selectable(const Ray &r, float origin_limit) {
	const bool origin_ok = fabsf(r.ox) <= origin_limit && fabsf(r.oy) <= origin_limit && fabsf(r.oz) <= origin_limit;
	const float ax = fabsf(r.ix), ay = fabsf(r.iy), az = fabsf(r.iz), inf = __builtin_inff();
	// a number on every axis (NaN fails every comparison), either infinite or small enough, and the reciprocal of a
	// unit vector's component (the margins convert an underflow in t to plane units with |inv| >= 1/2)
	const bool numbers = (ax <= RECIPROCAL_LIMIT || ax == inf) && (ay <= RECIPROCAL_LIMIT || ay == inf) &&
	                     (az <= RECIPROCAL_LIMIT || az == inf) && fminf(fminf(ax, ay), az) >= 0.5f;
	const bool some_finite = ax <= RECIPROCAL_LIMIT || ay <= RECIPROCAL_LIMIT || az <= RECIPROCAL_LIMIT;
	return origin_ok && numbers && some_finite;
}

// The common case in nine instructions: a finite origin within the limit and direction components that are numbers of
// magnitude 2^-99 ... 2.  Every reciprocal is then a normal number in [0.5, 2^99] (< RECIPROCAL_LIMIT): the short
// reciprocal has the division's bits (exact_reciprocal.h: exponents 1 ... 252), ray_is_selectable holds and there is no
// infinite reciprocal for the walk to replace.  (NaN fails: v_cmp_o for the direction, `<=` for the origin.)
__device__ __forceinline__ bool ray_is_tame(float ox, float oy, float oz, float dx, float dy, float dz, float origin_limit) {
	const float smallest = fminf(fminf(fabsf(dx), fabsf(dy)), fabsf(dz)), largest = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
	const bool numbers = !__builtin_isunordered(dx, dy) && !__builtin_isunordered(dz, dz);
	return numbers && smallest >= 0x1.0p-99f && largest <= 2.0f && fabsf(ox) <= origin_limit && fabsf(oy) <= origin_limit &&
	       fabsf(oz) <= origin_limit;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 128-bit loads through a buffer descriptor (wave-uniform base + 32-bit per-lane
// byte offset): one instruction per float4, out-of-range offsets return 0
// instead of faulting, and -- unlike a plain pointer load -- the compiler cannot
// split off the .w lane and sink it behind the box test (which it did, adding a
// second dependent memory round trip per node).
__device__ __forceinline__ float4 load_f4(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_offset) {
	const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int) byte_offset, 0, 0);
	return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ void normalize3(float &x, float &y, float &z) {
	const float l = sqrtf(dot3(x, y, z, x, y, z));
	x = x / l;
	y = y / l;
	z = z / l;
}

// Maps a rank-local tile row to the global tile row under the band partition.
__device__ __forceinline__ uint32_t global_tile_row(const Partition &p, uint32_t local_row) {
	const uint32_t band_local = local_row / p.band_tile_rows;
	const uint32_t within = local_row - band_local * p.band_tile_rows;
	return (band_local * p.nranks + p.rank) * p.band_tile_rows + within;
}

}  // namespace

// Triangle test for a pending leaf, straight-line.  Same operations and order
// as the reference (src/intersect_kernel.cl:65-114) on the precomputed TriRec;
// the early returns become one accumulated predicate so that all L lanes stay
// converged.  `x > 1.00001` (double literal) == `x > 0x3F800053`.
struct TriResult {
	bool accepted;
	float s, t, distance;
	float px, py, pz;
};

template <bool CLOSEST>
__device__ __forceinline__ TriResult tri_eval(const float4 q0, const float4 q1, const float4 q2, const float4 q3,
                                              const Ray &r) {
	const float tax = q0.x, tay = q0.y, taz = q0.z;
	const float ux = q0.w, uy = q1.x, uz = q1.y;
	const float vx = q1.z, vy = q1.w, vz = q2.x;
	const float nx = q2.y, ny = q2.z, nz = q2.w;
	const float uu = q3.x, uv = q3.y, vv = q3.z, D = q3.w;
	TriResult out;
	out.accepted = false;
	out.s = out.t = 0.0f;
	out.px = out.py = out.pz = 0.0f;
	out.distance = 0.0f;
	// reject: |b| < 1e-6, r < 0, s < -1e-5, s > 1.00001, t < -1e-5, s + t > 1.00001 -- in the
	// reference's order; the wave stops as soon as none of its lanes is left in the running
	const float a = -dot3(nx, ny, nz, r.ox - tax, r.oy - tay, r.oz - taz);
	const float b = dot3(nx, ny, nz, r.dx, r.dy, r.dz);
	const float rr = a / b;
	bool reject = (fabsf(b) < 0.000001f) | (rr < 0.0f);
	if (wave_ballot(!reject) == 0ull)
		return out;
	const float ipx = r.ox + rr * r.dx, ipy = r.oy + rr * r.dy, ipz = r.oz + rr * r.dz;
	const float wx = ipx - tax, wy = ipy - tay, wz = ipz - taz;
	const float wu = dot3(ux, uy, uz, wx, wy, wz);
	const float wv = dot3(wx, wy, wz, vx, vy, vz);
	const float slack_hi = __uint_as_float(0x3F800053u);
	const float s = (uv * wv - vv * wu) / D;
	reject |= (s < -0.00001f) | (s > slack_hi);
	if (wave_ballot(!reject) == 0ull)
		return out;
	const float t = (uv * wu - uu * wv) / D;
	reject |= (t < -0.00001f) | ((s + t) > slack_hi);
	out.accepted = !reject;
	out.s = s;
	out.t = t;
	out.px = ipx; out.py = ipy; out.pz = ipz;
	if (CLOSEST) {
		const float ex = ipx - r.ox, ey = ipy - r.oy, ez = ipz - r.oz;
		out.distance = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
	}
	return out;
}

// The any-hit form of the test: the same plane half (a, b, r = a / b, the point, wu, wv -- the reference's operations
// in the reference's order), then the parametric half as a predicate on products by TriRec::inv_d, with the
// reference's two divisions only for the lanes too close to a threshold to be decided that way (tri_predicate.h:
// the decision is the reference's in every case; 9 vector instructions instead of 27).
__device__ __forceinline__ bool tri_any_hit(const float4 q0, const float4 q1, const float4 q2, const float4 q3, float inv_d,
                                            const Ray &r) {
	const float tax = q0.x, tay = q0.y, taz = q0.z;
	const float ux = q0.w, uy = q1.x, uz = q1.y;
	const float vx = q1.z, vy = q1.w, vz = q2.x;
	const float nx = q2.y, ny = q2.z, nz = q2.w;
	const float uu = q3.x, uv = q3.y, vv = q3.z, D = q3.w;
	const float a = -dot3(nx, ny, nz, r.ox - tax, r.oy - tay, r.oz - taz);
	const float b = dot3(nx, ny, nz, r.dx, r.dy, r.dz);
	const float rr = a / b;
	const bool reject = (fabsf(b) < 0.000001f) | (rr < 0.0f);
	if (wave_ballot(!reject) == 0ull)
		return false;
	const float ipx = r.ox + rr * r.dx, ipy = r.oy + rr * r.dy, ipz = r.oz + rr * r.dz;
	const float wx = ipx - tax, wy = ipy - tay, wz = ipz - taz;
	const float wu = dot3(ux, uy, uz, wx, wy, wz);
	const float wv = dot3(wx, wy, wz, vx, vy, vz);
	const float X = uv * wv - vv * wu, Y = uv * wu - uu * wv;  // the numerators of s and t
	const unsigned int zone = tri_zone(X, Y, inv_d);
	bool accepted = zone == 1u;
	if (wave_ballot(!reject & (zone == 2u)) != 0ull)  // (about one test in 10^4)
		accepted = zone == 2u ? tri_accepts_exact(X, Y, D) : accepted;
	return accepted & !reject;
}

// Leaf records are 96 bytes: the leaf's own box (float4 0, 1), then the triangle (float4 2..5).
constexpr uint32_t LEAF_BYTES = 96u, LEAF_TRI_OFFSET = 32u, LEAF_F4 = 6u, LEAF_TRI_F4 = 2u;

template <bool CLOSEST>
__device__ __forceinline__ TriResult tri_test(__amdgpu_buffer_rsrc_t tris, uint32_t leaf, const Ray &r) {
	const uint32_t at = leaf * LEAF_BYTES + LEAF_TRI_OFFSET;
	const float4 q0 = load_f4(tris, at), q1 = load_f4(tris, at + 16u);
	const float4 q2 = load_f4(tris, at + 32u), q3 = load_f4(tris, at + 48u);
	return tri_eval<CLOSEST>(q0, q1, q2, q3, r);
}

struct SceneViews {
	__amdgpu_buffer_rsrc_t nodes;  // NodeRec[node_count]
	__amdgpu_buffer_rsrc_t tris;   // TriRec[tri_count]
};

__device__ __forceinline__ SceneViews make_views(const float4 *nodes_ptr, const float4 *tris_ptr, const KernelParams &P) {
	// descriptors are built from kernel arguments only, so they live in SGPRs
	SceneViews scene;
	scene.nodes = __builtin_amdgcn_make_buffer_rsrc((void *) nodes_ptr, 0, (int) (P.node_count * 32u), 0x00020000);
	scene.tris = __builtin_amdgcn_make_buffer_rsrc((void *) tris_ptr, 0, (int) (P.tri_count * LEAF_BYTES), 0x00020000);
	return scene;
}

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

// ---------------------------------------------------------------------------
// Shared walk.  The 64 rays of a wave visit the union of their nodes together:
// one wave-uniform node index `at`, so a node (and a leaf's triangle) arrives by
// scalar loads and the box is tested out of SGPRs; no per-lane index, no gathers,
// no scheduling.  Needs sibling subtrees to tile their parent's index range (KernelParams::shared_walk;
// checked at upload for the uploaded binary tree and again for the rebuilt, possibly wider one).
//
// Fast form (`exact` false: regular scene with nested boxes, regular rays): every
// live lane tests every visited box.  A lane that missed an ancestor also misses
// each box nested in it -- (lo-o)*inv and (hi-o)*inv are monotone in lo and hi, so
// near can only grow and far only shrink -- hence a lane accepts exactly the
// triangles of its own walk, in the same ascending leaf order.
//
// Exact form: each lane also keeps `mine`, the next node of its OWN walk, tests a
// box only when at == mine, and uses the reference's select-based slab test: lane
// by lane that is the reference's walk (src/intersect_kernel.cl:184-213) of the node
// array the kernel was given, whatever the boxes and rays hold.  For damaged scene
// arrays that array is the uploaded one, so this IS the reference walk.  On a regular,
// nested scene it may be the rebuilt tree (scene_pack.cc); the form is then only taken by
// packets holding a ray that is not "selectable" -- a NaN direction or origin (zero-length
// vertex normals), all three reciprocals infinite -- and such a ray fails the slab test
// at the ROOT of any tree (`NaN < max_distance` is false, reference :60), so it tests no
// triangle in either tree, while its selectable neighbours get, lane by lane, the
// monotone slab test on nested boxes, for which the tree does not matter.
// tests/test_hip_parity.py::test_zero_normals_on_a_rebuilt_tree covers it.
// `at` never overtakes a live lane's `mine` because subtree ranges nest.
// ---------------------------------------------------------------------------
// One node of the exact form for a lane: the reference's slab test where the lane's own walk stands (`mine`).
__device__ __forceinline__ bool exact_box(const float4 lo, const float4 hi, const Ray &ray, float max_distance, bool alive,
                                          uint32_t at, uint32_t skip, uint32_t &mine) {
	const bool here = alive && mine == at;
	const bool box = here && slab_hit(lo, hi, ray, max_distance);
	mine = here ? (box ? at + 1u : at + skip) : mine;
	return box;
}

// Per packet: which lanes have a reciprocal direction >= 0 on each axis (the reference's
// `inv >= 0 ? lo : hi` choice of the near plane, made once instead of at every node).
struct SignMasks {
	unsigned long long x, y, z;
};
__device__ __forceinline__ SignMasks sign_masks(const Ray &r) {
	SignMasks m;
	m.x = wave_ballot(r.ix >= 0.0f);
	m.y = wave_ballot(r.iy >= 0.0f);
	m.z = wave_ballot(r.iz >= 0.0f);
	return m;
}

// A node record by one scalar load (asm for the reason given at walk_collect: a plain load through
// nodes_ptr next to that loop makes the compiler keep the pointer in VGPRs).
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32x8 scalar_load_node(const float4 *nodes_ptr, uint32_t at) {
	u32x8 r;
	const uint32_t offset = at * 32u;
	asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(nodes_ptr), "s"(offset));
	return r;
}

// The node steps of the fast form, hand-scheduled.  From byte offset `at` on it walks the packet through the
// PADDED copy of the tree (scene_pack.cc, pad_walk_boxes): per node a conservative slab test on values fetched by a
// scalar load -- t = fma(plane, inv, oi) with oi = -(o * inv) rounded once, near/far planes picked by the sign of
// inv, max(.., tiny = bit pattern 1), min(.., below), near <= far -- and a scalar decision: some live lane hit ->
// first child, nobody -> skip the subtree.  The outward margin of the boxes makes up for the fma's rounding
// (proof at padded_bound), so a lane passes every box the reference's own test (src/intersect_kernel.cl:21-61)
// would let it pass, and possibly a few more: the walk only finds CANDIDATE leaves, the exact test on the leaf's
// own box is the caller's (exact_leaf_gate).  Zero direction components enter with +-2^100 for the infinite
// reciprocal (WalkRay); should a NaN still arise, v_max3 / v_min3 drop it (IEEE maxNum / minNum, the kernel runs
// with IEEE mode on) -- one constraint fewer, conservative.
//
// Primary packets are sign-coherent -- every live lane's reciprocal direction has the same sign on each axis -- but on
// the image's centre lines: the near and far plane of each axis are then known when the loop is entered and the test is
// 6 v_fma + max + min + max3 + min3 + cmp = 11 vector instructions; the loop exists once per sign octant
// (OCRT_WALK_COHERENT).  Mixed packets select per lane with v_cndmask on the sign masks: 17 (the first generation of
// this loop computed (b - o) * inv exactly: 23).  The any-hit rays of the ambient-occlusion pass, whose max_distance is
// one number per frame, take the SCALED form of the test on centre / half-extent records: 12, one loop for every packet
// (OCRT_TEST_CE_SCALED below).  What an instruction costs here (tools/microbench/
// valu_rate_probe.hip, 8 waves per SIMD): ~2.3 cycles per SIMD for v_fma / v_mul / v_add / v_mov on registers, ~4.2
// for everything else (min / max / max3 / cmp / cndmask, v_pk_fma_f32, and alone also an fma with a scalar operand):
// the 11-instruction test runs at 35.7 cycles, the 9-instruction one at 28.3, a v_pk_fma_f32 version with 8 at 35.9.
//
// One 64-byte load fetches a node and its pre-order successor: after a hit on an inner node its first child is
// tested straight from s[56:63].  The array ends in two END records whose infinite box every live lane "hits" and
// whose leaf field says WALK_END, so the loop needs no bounds check.  Scalar instructions per node: load, wait,
// s_and (sets SCC), branch, add = 5 on a miss.
// At a leaf hit by fewer than `batch_below` lanes it does not stop but appends the (lane, leaf) pairs to the
// wave's list in LDS (entry = leaf | lane << 26 at index waiting + rank of the lane among the hitters) and walks
// on; `leaf_stops` counts the leaves some lane hit.
// Returns 0: walk over; 1: `leaf` is hit by many lanes (`hit`), test it now, `at` is on it; 2: 64 or more pairs
// are waiting, run a batch, `at` is on the leaf appended last.
// Scratch: s[42:63], v56-v62; only scalar outputs, so that the compiler knows the results to be wave-uniform.
// The pointer operand must not be dereferenced by plain loads elsewhere in the same kernel: the compiler then
// keeps it in VGPRs and cannot hand it to this operand.
// gfx950 hazards checked by hand (the assembler inserts nothing inside inline asm): v_cmp writes VCC -> s_and_b64
// reads it (SALU reads of VALU-written SGPRs are interlocked); s_mov_b64 exec -> ds_write_b32 / following VALU
// (EXEC writes by SALU are interlocked for vector and LDS instructions); s_load -> s_waitcnt lgkmcnt(0) before the
// first use (also drains the ds_write of an append, harmless); no v_readlane / v_div_fmas / VMEM-with-SGPR-address
// consumers of VALU-written SGPRs in here.  The build fails if the kernels using this loop spill vector registers
// or leave 8 waves per SIMD (tools/check_kernel_resources.py).
#define OCRT_TEST_COHERENT(NX, NY, NZ, FX, FY, FZ) \
	"\tv_fma_f32 v56, " NX ", %[ix], %[oix]\n"     \
	"\tv_fma_f32 v57, " NY ", %[iy], %[oiy]\n"     \
	"\tv_fma_f32 v58, " NZ ", %[iz], %[oiz]\n"     \
	"\tv_fma_f32 v59, " FX ", %[ix], %[oix]\n"     \
	"\tv_fma_f32 v60, " FY ", %[iy], %[oiy]\n"     \
	"\tv_fma_f32 v61, " FZ ", %[iz], %[oiz]\n"     \
	"\tv_max_f32 v58, 1, v58\n"                    \
	"\tv_min_f32 v61, %[below], v61\n"             \
	"\tv_max3_f32 v56, v56, v57, v58\n"            \
	"\tv_min3_f32 v59, v59, v60, v61\n"            \
	"\tv_cmp_le_f32 vcc, v56, v59\n"
#define OCRT_TEST_MIXED(LX, LY, LZ, HX, HY, HZ)    \
	"\tv_fma_f32 v56, " LX ", %[ix], %[oix]\n"     \
	"\tv_fma_f32 v57, " HX ", %[ix], %[oix]\n"     \
	"\tv_fma_f32 v58, " LY ", %[iy], %[oiy]\n"     \
	"\tv_fma_f32 v59, " HY ", %[iy], %[oiy]\n"     \
	"\tv_fma_f32 v60, " LZ ", %[iz], %[oiz]\n"     \
	"\tv_fma_f32 v61, " HZ ", %[iz], %[oiz]\n"     \
	"\tv_cndmask_b32 v62, v57, v56, %[px]\n"       \
	"\tv_cndmask_b32 v56, v56, v57, %[px]\n"       \
	"\tv_cndmask_b32 v57, v59, v58, %[py]\n"       \
	"\tv_cndmask_b32 v58, v58, v59, %[py]\n"       \
	"\tv_cndmask_b32 v59, v61, v60, %[pz]\n"       \
	"\tv_cndmask_b32 v60, v60, v61, %[pz]\n"       \
	"\tv_max_f32 v59, 1, v59\n"                    \
	"\tv_min_f32 v60, %[below], v60\n"             \
	"\tv_max3_f32 v62, v62, v57, v59\n"            \
	"\tv_min3_f32 v56, v56, v58, v60\n"            \
	"\tv_cmp_le_f32 vcc, v62, v56\n"
// The SCALED form (any-hit rays, whose max_distance is one number per frame): the reciprocals carry a factor
// ~ 1 / max_distance, so "t < max_distance" reads "t' <= 1" and both limits fit the CLAMP modifier of the z-axis fmas
// (clamp to [0, 1]): near = max3(x, y, clamp(z)), far = min3(x, y, clamp(z)), hit iff near < far.  The comparison is
// strict so that a box behind the origin on z (far clamped to 0, near >= 0) fails; why no pair the reference accepts is
// lost to that: scene_pack.cc, padded_bound ("The scaled form").
// It reads the CENTRE / HALF-EXTENT copy of the walk array (scene_pack.cc, ce_record: c in the lo fields, e in the hi
// fields; the copy lies behind the plane form's records and their END records):
// t_c = fma(c, inv, oi), near = fma(-e, |inv|, t_c), far = fma(e, |inv|, t_c) -- right for either sign of inv, so ONE loop
// serves every any-hit packet: 9 v_fma + max3 + min3 + cmp = 12 vector instructions, nine of them of the fast class.
// (Rounds 2-3 walked the plane-form records here too: a loop per sign octant, 6 fma + max3 + min3 + cmp = 9 per node, and
// a select form of 15 -- nine of them of the slow class -- for packets whose rays disagree on a sign, a third of the
// bunny's model packets.  The one loop measures 1.3 ... 4.4 % faster per frame on every workload, although coherent
// packets execute three instructions more per node: fast-class fmas, one array in the caches, a ninth of the code.)
// Conservative like the plane form (the half-extent carries the rounding of t_c: proof at ce_record).
#define OCRT_TEST_CE_SCALED(CX, CY, CZ, EX, EY, EZ)       \
	"\tv_fma_f32 v56, " CX ", %[ix], %[oix]\n"           \
	"\tv_fma_f32 v57, " CY ", %[iy], %[oiy]\n"           \
	"\tv_fma_f32 v58, " CZ ", %[iz], %[oiz]\n"           \
	"\tv_fma_f32 v59, -" EX ", |%[ix]|, v56\n"           \
	"\tv_fma_f32 v56, " EX ", |%[ix]|, v56\n"            \
	"\tv_fma_f32 v60, -" EY ", |%[iy]|, v57\n"           \
	"\tv_fma_f32 v57, " EY ", |%[iy]|, v57\n"            \
	"\tv_fma_f32 v61, -" EZ ", |%[iz]|, v58 clamp\n"     \
	"\tv_fma_f32 v58, " EZ ", |%[iz]|, v58 clamp\n"      \
	"\tv_max3_f32 v59, v59, v60, v61\n"                  \
	"\tv_min3_f32 v56, v56, v57, v58\n"                  \
	"\tv_cmp_lt_f32 vcc, v59, v56\n"
// (LEAF: the s-register holding the node's leaf field; NEXT: where the walk goes on after an append)
#define OCRT_WALK_LEAF(LEAF, NOW, NEXT)                 \
	"\ts_cmp_eq_u32 " LEAF ", -2\n"                     \
	"\ts_cbranch_scc1 .Lw_over_%=\n"                    \
	"\ts_bcnt1_i32_b64 s46, s[44:45]\n"                 \
	"\ts_add_u32 %[stops], %[stops], 1\n"               \
	"\ts_cmp_ge_u32 s46, %[batch_below]\n"              \
	"\ts_cbranch_scc1 " NOW "\n"                        \
	"\tv_mbcnt_lo_u32_b32 v56, s44, 0\n"                \
	"\tv_mbcnt_hi_u32_b32 v56, s45, v56\n"              \
	"\tv_add_u32 v56, %[waiting], v56\n"                \
	"\tv_lshl_add_u32 v56, v56, 2, %[list]\n"           \
	"\tv_or_b32 v57, " LEAF ", %[tag]\n"                \
	"\ts_mov_b64 s[42:43], exec\n"                      \
	"\ts_mov_b64 exec, s[44:45]\n"                      \
	"\tds_write_b32 v56, v57\n"                         \
	"\ts_mov_b64 exec, s[42:43]\n"                      \
	"\ts_add_u32 %[waiting], %[waiting], s46\n"         \
	"\ts_cmp_ge_u32 %[waiting], 64\n"                   \
	"\ts_cbranch_scc1 .Lw_full_%=\n"                    \
	"\ts_branch " NEXT "\n"
// PF_B: what the loop does between the tests of a pair, once `a` is known to be hit.  OCRT_PF_SUCCESSORS touches -- with
// one-dword scalar loads nobody reads -- both places the walk can go to after `b`: the line behind the pair and b's skip
// target.  One of the two is the next load, which then finds its line in the scalar cache or on its way instead of
// starting a round trip of its own: the walk is a chain of dependent loads, and this takes the test of `b` out of the
// chain.  It pays where packets mostly descend (the bunny's model tiles: ambient-occlusion pass -2 %, frames in flight
// -2.5 ... -3.7 %) and costs where they mostly miss (the interior scene: +1 ... 2 %; the useless one of the two loads is
// waited for by the next s_waitcnt all the same) -- so a render host can be told which form to launch
// (DeviceRenderer::setAoPrefetch; a frame ring measures both on its scene at upload).  Forms that touch the skip target of
// `a` before its test, or the next line alone, measured worse on one side or the other (profiles/r04_notes.md).
#define OCRT_PF_NONE ""
// Only where `b` is an inner node: then every path from here leads to the loop's next load and its s_waitcnt lgkmcnt(0),
// which also waits for these two.  Behind a leaf the loop may be LEFT (a leaf stop, a full list) -- with a load still on
// its way to s47, a register the compiler is free to use again the moment the asm block ends: it would be overwritten
// whenever the load lands.  (That was the first form of this; a 20 k-triangle height field showed it, the bunny did not.)
#define OCRT_PF_SUCCESSORS                                    \
	"\ts_cmp_lg_u32 s63, -1\n"                               \
	"\ts_cbranch_scc1 .Lw_no_pf_%=\n"                        \
	"\ts_add_u32 s46, %[at], 32\n"                           \
	"\ts_load_dword s47, %[base], s46\n"                     \
	"\ts_add_u32 s46, %[at], s59\n"                          \
	"\ts_load_dword s47, %[base], s46\n"                     \
	".Lw_no_pf_%=:\n"
// HEAD: what is checked while a pair is being fetched.  OCRT_HEAD_END: the walk is over once `at` has left the range it
// was given (the any-hit walks of a tile run through ONE subtree, its entry: ao_kernel) -- two scalar instructions per
// pair, in the shadow of the load (which is waited for on the way out too: its sixteen registers are the compiler's again
// once the block ends; the records behind any range are there to be read, END records at the latest).
#define OCRT_HEAD_NONE ""
#define OCRT_HEAD_END "\ts_cmp_ge_u32 %[at], %[end]\n\ts_cbranch_scc1 .Lw_over_wait_%=\n"
#define OCRT_WALK_ASM(HEAD, TEST_A, TEST_B, PF_B)           \
	"\ts_branch .Lw_node_%=\n"                              \
	".Lw_miss_a_%=:\n"                                      \
	"\ts_add_u32 %[at], %[at], s51\n"                       \
	".Lw_node_%=:\n"                                        \
	"\ts_load_dwordx16 s[48:63], %[base], %[at]\n"          \
	HEAD                                                    \
	"\ts_waitcnt lgkmcnt(0)\n"                              \
	TEST_A                                                  \
	"\ts_and_b64 s[44:45], vcc, %[alive]\n"                 \
	"\ts_cbranch_scc0 .Lw_miss_a_%=\n"                      \
	"\ts_cmp_lg_u32 s55, -1\n"                              \
	"\ts_cbranch_scc1 .Lw_leaf_a_%=\n"                      \
	".Lw_next_b_%=:\n"                                      \
	"\ts_add_u32 %[at], %[at], 32\n"                        \
	PF_B                                                    \
	TEST_B                                                  \
	"\ts_and_b64 s[44:45], vcc, %[alive]\n"                 \
	"\ts_cbranch_scc0 .Lw_miss_b_%=\n"                      \
	"\ts_cmp_lg_u32 s63, -1\n"                              \
	"\ts_cbranch_scc1 .Lw_leaf_b_%=\n"                      \
	".Lw_next_a_%=:\n"                                      \
	"\ts_add_u32 %[at], %[at], 32\n"                        \
	"\ts_branch .Lw_node_%=\n"                              \
	".Lw_miss_b_%=:\n"                                      \
	"\ts_add_u32 %[at], %[at], s59\n"                       \
	"\ts_branch .Lw_node_%=\n"                              \
	".Lw_leaf_a_%=:\n"                                      \
	OCRT_WALK_LEAF("s55", ".Lw_now_a_%=", ".Lw_next_b_%=")  \
	".Lw_now_a_%=:\n"                                       \
	"\ts_mov_b32 %[leaf], s55\n"                            \
	"\ts_branch .Lw_now_%=\n"                               \
	".Lw_leaf_b_%=:\n"                                      \
	OCRT_WALK_LEAF("s63", ".Lw_now_b_%=", ".Lw_next_a_%=")  \
	".Lw_now_b_%=:\n"                                       \
	"\ts_mov_b32 %[leaf], s63\n"                            \
	".Lw_now_%=:\n"                                         \
	"\ts_mov_b64 %[hit], s[44:45]\n"                        \
	"\ts_mov_b32 %[status], 1\n"                            \
	"\ts_branch .Lw_out_%=\n"                               \
	".Lw_full_%=:\n"                                        \
	"\ts_mov_b32 %[status], 2\n"                            \
	"\ts_branch .Lw_out_%=\n"                               \
	".Lw_over_wait_%=:\n"                                   \
	"\ts_waitcnt lgkmcnt(0)\n"                              \
	".Lw_over_%=:\n"                                        \
	"\ts_mov_b32 %[status], 0\n"                            \
	".Lw_out_%=:\n"
#define OCRT_WALK_CLOBBERS                                                                                              \
	"s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", \
	    "s59", "s60", "s61", "s62", "s63", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "vcc", "scc", "memory"
// node a = s[48:55] (lo.xyz, skip bytes, hi.xyz, leaf), node b = s[56:63]; X/Y/Z: "P" = reciprocal >= 0 on that axis
// (near plane lo), "N" = negative (near plane hi)
#define OCRT_NEAR_P(LO, HI) LO
#define OCRT_NEAR_N(LO, HI) HI
#define OCRT_FAR_P(LO, HI) HI
#define OCRT_FAR_N(LO, HI) LO
#define OCRT_WALK_COHERENT(TEST, X, Y, Z, PF)                                                                               \
	asm volatile(OCRT_WALK_ASM(OCRT_HEAD_NONE, TEST(OCRT_NEAR_##X("s48", "s52"), OCRT_NEAR_##Y("s49", "s53"),              \
	                                OCRT_NEAR_##Z("s50", "s54"), OCRT_FAR_##X("s48", "s52"),                               \
	                                OCRT_FAR_##Y("s49", "s53"), OCRT_FAR_##Z("s50", "s54")),                               \
	                           TEST(OCRT_NEAR_##X("s56", "s60"), OCRT_NEAR_##Y("s57", "s61"),                              \
	                                OCRT_NEAR_##Z("s58", "s62"), OCRT_FAR_##X("s56", "s60"),                               \
	                                OCRT_FAR_##Y("s57", "s61"), OCRT_FAR_##Z("s58", "s62")), PF)                           \
	             : [at] "+s"(at), [waiting] "+s"(waiting), [stops] "+s"(leaf_stops), [hit] "=&s"(hit_mask),               \
	               [leaf] "=&s"(leaf), [status] "=&s"(status)                                                            \
	             : [base] "s"(walk_ptr), [alive] "s"(alive_mask), [below] "s"(below), [batch_below] "s"(batch_below),     \
	               [list] "s"(list_lds_address), [tag] "v"(lane_tag), [ix] "v"(ray.ix), [iy] "v"(ray.iy), [iz] "v"(ray.iz), \
	               [oix] "v"(ray.oix), [oiy] "v"(ray.oiy), [oiz] "v"(ray.oiz)                                             \
	             : OCRT_WALK_CLOBBERS)

#define OCRT_WALK_MIXED(TEST, PF)                                                                                         \
	asm volatile(OCRT_WALK_ASM(OCRT_HEAD_NONE, TEST("s48", "s49", "s50", "s52", "s53", "s54"), TEST("s56", "s57", "s58", "s60", "s61", "s62"), PF) \
	             : [at] "+s"(at), [waiting] "+s"(waiting), [stops] "+s"(leaf_stops), [hit] "=&s"(hit_mask),               \
	               [leaf] "=&s"(leaf), [status] "=&s"(status)                                                            \
	             : [base] "s"(walk_ptr), [alive] "s"(alive_mask), [below] "s"(below), [batch_below] "s"(batch_below),     \
	               [px] "s"(sign.x), [py] "s"(sign.y), [pz] "s"(sign.z), [list] "s"(list_lds_address), [tag] "v"(lane_tag), \
	               [ix] "v"(ray.ix), [iy] "v"(ray.iy), [iz] "v"(ray.iz), [oix] "v"(ray.oix), [oiy] "v"(ray.oiy),          \
	               [oiz] "v"(ray.oiz)                                                                                    \
	             : OCRT_WALK_CLOBBERS)
#define OCRT_WALK_MIXED_CE(TEST, PF)                                                                                      \
	asm volatile(OCRT_WALK_ASM(OCRT_HEAD_END, TEST("s48", "s49", "s50", "s52", "s53", "s54"), TEST("s56", "s57", "s58", "s60", "s61", "s62"), PF) \
	             : [at] "+s"(at), [waiting] "+s"(waiting), [stops] "+s"(leaf_stops), [hit] "=&s"(hit_mask),               \
	               [leaf] "=&s"(leaf), [status] "=&s"(status)                                                            \
	             : [base] "s"(walk_ptr), [alive] "s"(alive_mask), [end] "s"(walk_end), [batch_below] "s"(batch_below),    \
	               [list] "s"(list_lds_address), [tag] "v"(lane_tag),                                                      \
	               [ix] "v"(ray.ix), [iy] "v"(ray.iy), [iz] "v"(ray.iz), [oix] "v"(ray.oix), [oiy] "v"(ray.oiy),          \
	               [oiz] "v"(ray.oiz)                                                                                    \
	             : OCRT_WALK_CLOBBERS)
#define OCRT_WALK_SWITCH(COHERENT_TEST, MIXED_TEST, PF)                 \
	switch (variant) {                                                  \
	case 0u: OCRT_WALK_COHERENT(COHERENT_TEST, N, N, N, PF); break;     \
	case 1u: OCRT_WALK_COHERENT(COHERENT_TEST, P, N, N, PF); break;     \
	case 2u: OCRT_WALK_COHERENT(COHERENT_TEST, N, P, N, PF); break;     \
	case 3u: OCRT_WALK_COHERENT(COHERENT_TEST, P, P, N, PF); break;     \
	case 4u: OCRT_WALK_COHERENT(COHERENT_TEST, N, N, P, PF); break;     \
	case 5u: OCRT_WALK_COHERENT(COHERENT_TEST, P, N, P, PF); break;     \
	case 6u: OCRT_WALK_COHERENT(COHERENT_TEST, N, P, P, PF); break;     \
	case 7u: OCRT_WALK_COHERENT(COHERENT_TEST, P, P, P, PF); break;     \
	default: MIXED_TEST; break;                                         \
	}

// `variant`: 0..7 = sign octant of a coherent packet (bit 0: x reciprocals >= 0, bit 1: y, bit 2: z), 8 = mixed.
// SCALED: `ray` was made with the frame's walk_scale and `below` is not looked at (the limit is 1.0).
constexpr uint32_t WALK_MIXED = 8u;
template <bool SCALED, bool PREFETCH = false>
__device__ __forceinline__ uint32_t walk_collect(uint32_t variant, const float4 *walk_ptr, uint32_t &at, const WalkRay &ray,
                                                 const SignMasks &sign, float below, unsigned long long alive_mask,
                                                 unsigned long long &hit_mask, uint32_t &leaf, uint32_t &waiting,
                                                 uint32_t &leaf_stops, uint32_t list_lds_address, uint32_t lane_tag,
                                                 uint32_t batch_below, uint32_t walk_end = 0xFFFFFFFFu) {
	(void) walk_end;  // (the any-hit loop's: byte offset behind the subtree it walks)
	uint32_t status;
	if (SCALED) {
		// one loop for every any-hit packet, whatever its rays' signs (the caller starts `at` in the centre / half-extent
		// copy of the array)
		(void) sign;
		(void) variant;
		if (PREFETCH) {
			OCRT_WALK_MIXED_CE(OCRT_TEST_CE_SCALED, OCRT_PF_SUCCESSORS);
		} else {
			OCRT_WALK_MIXED_CE(OCRT_TEST_CE_SCALED, OCRT_PF_NONE);
		}
	} else {
		// (the primary pass keeps the plane form and its loop per sign octant: its packets are coherent but for the
		// image's centre lines, and 11 instructions beat 14: 1-5 % of the pass)
		OCRT_WALK_SWITCH(OCRT_TEST_COHERENT, OCRT_WALK_MIXED(OCRT_TEST_MIXED, OCRT_PF_NONE), OCRT_PF_NONE)
	}
	return status;
}

// Which loop a packet takes: its sign octant if every live lane agrees on every axis, else WALK_MIXED.
__device__ __forceinline__ uint32_t walk_variant(const SignMasks &sign, unsigned long long alive_mask) {
	const unsigned long long x = sign.x & alive_mask, y = sign.y & alive_mask, z = sign.z & alive_mask;
	bool coherent = (x == 0ull || x == alive_mask) && (y == 0ull || y == alive_mask) && (z == 0ull || z == alive_mask);
#ifdef OCRT_ALWAYS_MIXED
	coherent = false;
#endif
	return coherent ? (x != 0ull ? 1u : 0u) | (y != 0ull ? 2u : 0u) | (z != 0ull ? 4u : 0u) : WALK_MIXED;
}

// The exact test on a candidate leaf's OWN box (uploaded, unpadded), the reference's gate of the triangle test
// (src/intersect_kernel.cl:189,195).  Only packets of the fast form get here -- regular boxes, selectable rays --,
// for which the reference's chain of comparisons folds into max(near, tiny) <= min(far, below) with the near / far
// plane picked by the sign of the reciprocal and IEEE maxNum / minNum dropping the NaN of 0 * inf (the
// first generation of walk_collect applied exactly this arithmetic to every node).  `below` is the largest float
// under the ray kind's max_distance.
__device__ __forceinline__ bool exact_leaf_gate(const float4 lo, const float4 hi, const Ray &r, float below) {
	const float x0 = (lo.x - r.ox) * r.ix, x1 = (hi.x - r.ox) * r.ix;
	const float y0 = (lo.y - r.oy) * r.iy, y1 = (hi.y - r.oy) * r.iy;
	const float z0 = (lo.z - r.oz) * r.iz, z1 = (hi.z - r.oz) * r.iz;
	const bool px = r.ix >= 0.0f, py = r.iy >= 0.0f, pz = r.iz >= 0.0f;
	const float tiny = __uint_as_float(1u);
	const float t_near = fmaxf(fmaxf(px ? x0 : x1, py ? y0 : y1), fmaxf(pz ? z0 : z1, tiny));
	const float t_far = fminf(fminf(px ? x1 : x0, py ? y1 : y0), fminf(pz ? z1 : z0, below));
	return t_near <= t_far;
}

// Triangle tests of an any-hit packet waiting to be run 64 at a time (LDS, one per wave).
struct LeafBatch {
	unsigned int entry[128];       // leaf | owning lane << 26; up to 63 waiting + 64 appended at one leaf
	unsigned int occluded_bits[2];  // lanes whose ray was found occluded by the batch just run
};

// The same for a closest-hit packet (primary rays).  The reference keeps the first hit in
// leaf order among the nearest (`best.distance > distance`, strict): that is the minimum of
// (distance, leaf) in lexicographic order, so the tests may run in any order and on any lane
// if each ray's minimum of key = distance bits << 32 | leaf is kept -- here by LDS atomics.
// The winner's barycentrics and hit point are recomputed by the ray's own lane at the end.
struct ClosestBatch {
	unsigned int entry[128];
	unsigned long long best_key[64];
	unsigned int hit_bits[2];  // rays with an accepted triangle, whatever its distance (reference :108-113)
};
constexpr unsigned long long KEY_NONE = ~0ull;
constexpr uint32_t INF_BITS = 0x7F800000u;

// Any-hit shared walk of one packet (AO): a lane leaves at its first accepted
// triangle and bumps *occluded (reference :251 only uses the boolean).
// EXACT walks `nodes_ptr` (exact boxes); the fast form walks `walk_ptr` (padded boxes) and gates every candidate
// leaf with its own box, the head of its leaf record (scalar loads for a leaf tested on the spot; in a batch each lane
// loads its pair's record relative to the same scalar base: no buffer descriptor held across the walk).
template <bool EXACT, bool PREFETCH = false>
__device__ __forceinline__ void shared_walk_any_hit(const float4 *__restrict__ nodes_ptr, const float4 *__restrict__ walk_ptr,
                                                    const float4 *__restrict__ tris_ptr,
                                                    uint32_t count, const Ray &ray_in, const float (&frame)[12][64], uint32_t h,
                                                    float max_distance, float below, float walk_scale, bool alive, bool tame,
                                                    unsigned int *occluded, LeafBatch &batch, uint32_t batch_below,
                                                    unsigned long long *prof, uint32_t entry_begin = 0u, uint32_t entry_end = 0xFFFFFFFFu) {
	(void) prof;  // (-DOCRT_STAMPS builds: time in the node loop / in batches, loop entries, batches, leaf stops)
	// the live lanes as a scalar mask: the node steps then need no per-lane bookkeeping at all
	unsigned long long alive_mask = wave_ballot(alive);
	// Registers held across the walk are scarce (64 per lane at 8 waves per SIMD, 7 of them the loop's own): the
	// ray's origin stays in the tile's LDS table (frame[0..2][h], where setup_ray took it from) and is read again
	// where a triangle or a leaf's own box is tested, and the lane number is recomputed where it is needed.
	auto with_origin = [&]() {
		Ray r = ray_in;
		r.ox = frame[0][h]; r.oy = frame[1][h]; r.oz = frame[2][h];
		return r;
	};
	if (!EXACT) {
		// The triangle tests are not run where the walk meets them -- a leaf is hit by 15 of the 64
		// rays on average -- but collected as (ray, leaf) pairs and run 64 at a time, each lane taking
		// ANY pair: it fetches that ray from its owner (cross-lane reads) and the triangle by a
		// gather.  An any-hit ray only needs "some accepted triangle", so neither the order of the
		// tests nor who computes them matters, and every test is the same arithmetic on the same
		// operands as before.  A ray found occluded leaves the walk after the batch instead of at
		// the leaf, which only lets it ride along a little longer.
		uint32_t waiting = 0u;  // pairs in batch.entry (wave-uniform)
		uint32_t leaf_stops = 0u;  // (not used by this pass)
		auto run_batch = [&](uint32_t n) {
#ifdef OCRT_STAMPS
			const unsigned long long tb0 = __builtin_amdgcn_s_memrealtime();
#endif
			wave_lds_sync();
			const uint32_t lane = fresh_lane();
			const uint32_t pair = batch.entry[lane < n ? lane : 0u];
			const int owner = (int) (pair >> 26);
			const Ray ray = with_origin();
			Ray theirs;
			theirs.ox = __shfl(ray.ox, owner); theirs.oy = __shfl(ray.oy, owner); theirs.oz = __shfl(ray.oz, owner);
			theirs.dx = __shfl(ray.dx, owner); theirs.dy = __shfl(ray.dy, owner); theirs.dz = __shfl(ray.dz, owner);
			theirs.ix = __shfl(ray.ix, owner); theirs.iy = __shfl(ray.iy, owner); theirs.iz = __shfl(ray.iz, owner);
			if (lane < n) {
				// the pair is a candidate of the padded walk: the leaf's own box decides whether the reference tests it
				const uint32_t pair_leaf = pair & 0x03FFFFFFu;
				const float4 *rec = (const float4 *) ((const char *) tris_ptr + pair_leaf * LEAF_BYTES);  // (scalar base + 32-bit lane offset)
				const float4 lo = rec[0], hi = rec[1];
				const bool gate = exact_leaf_gate(lo, hi, theirs, below);
#ifdef OCRT_STAMPS
				prof[5] += n;  // candidate pairs / pairs whose own box passes
				prof[6] += (unsigned long long) __popcll(wave_ballot(gate));
#endif
				if (gate) {
					if (tri_any_hit(rec[2], rec[3], rec[4], rec[5], hi.w, theirs))
						atomicOr(&batch.occluded_bits[owner >> 5], 1u << (owner & 31));
				}
			}
			wave_lds_sync();
			const uint32_t bits = batch.occluded_bits[lane >> 5];
			if (alive && ((bits >> (lane & 31u)) & 1u)) {
				atomicAdd(occluded, 1u);  // once per ray, however many of its pairs were accepted
				alive = false;
			}
			wave_lds_sync();
			if (lane < 2u)
				batch.occluded_bits[lane] = 0u;
			alive_mask = wave_ballot(alive);
#ifdef OCRT_STAMPS
			prof[1] += __builtin_amdgcn_s_memrealtime() - tb0;
			prof[3] += 1;
#endif
		};
		const uint32_t list_lds_address = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (uintptr_t) &batch.entry[0]);  // (low half of the flat address; scalar)
		const SignMasks sign{ 0ull, 0ull, 0ull };  // (not looked at by the any-hit loop)
		const uint32_t variant = WALK_MIXED;
		// byte offset of the node: the walk reads the centre / half-extent copy of the records, which lies behind the plane
		// form's and its two END records (scene_pack.cc, make_walk_array: 2 * (count + 2) * 32 < 2^32) -- and of that copy
		// only the tile's ENTRY subtree [entry_begin, entry_end): the deepest node under which every leaf lies that a ray
		// of this tile can reach (entry_kernel)
		const uint32_t copy = (count + 2u) * 32u;
		uint32_t at = copy + entry_begin;
		const uint32_t whole = count * 32u;
		const uint32_t end = copy + (entry_end < whole ? entry_end : whole);
		const WalkRay walk_ray = tame ? make_walk_ray(with_origin(), walk_scale, true) : make_walk_ray(with_origin(), walk_scale);  // (wave-uniform)
		const uint32_t lane_tag = fresh_lane() << 26;
		while (alive_mask != 0ull && at < end) {
			uint32_t leaf = 0u;
			unsigned long long hit_mask = 0ull;
#ifdef OCRT_STAMPS
			const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#endif
			const uint32_t status = walk_collect<true, PREFETCH>(variant, walk_ptr, at, walk_ray, sign, below, alive_mask, hit_mask, leaf,
			                                           waiting, leaf_stops, list_lds_address, lane_tag, batch_below, end);
#ifdef OCRT_STAMPS
			prof[0] += __builtin_amdgcn_s_memrealtime() - tw0;
			prof[2] += 1;
#endif
			if (status == 0u)
				break;
			if (status == 1u) {
				// enough of the packet is at this leaf: test it here, box and triangle out of SGPRs
				const float4 *rec = tris_ptr + LEAF_F4 * leaf;
				const float4 lo = rec[0], hi = rec[1], q0 = rec[2], q1 = rec[3], q2 = rec[4], q3 = rec[5];
				const Ray ray = with_origin();
				if (((hit_mask >> fresh_lane()) & 1ull) && exact_leaf_gate(lo, hi, ray, below)) {
					if (tri_any_hit(q0, q1, q2, q3, hi.w, ray)) {
						atomicAdd(occluded, 1u);
						alive = false;
					}
				}
				alive_mask = wave_ballot(alive);
			} else {
				run_batch(64u);
				waiting -= 64u;
				const uint32_t me = fresh_lane();
				if (me < waiting)  // the pairs beyond the batch move to the front
					batch.entry[me] = batch.entry[64u + me];
			}
			at += 32u;
		}
		if (waiting != 0u)
			run_batch(waiting);
#ifdef OCRT_STAMPS
		prof[4] += leaf_stops;
#endif
		return;
	}
	const Ray ray = ray_in;
	uint32_t mine = 0u;
	uint32_t at = 0u;
	while (at < count) {
		const u32x8 node = scalar_load_node(nodes_ptr, at);
		const float4 lo = make_float4(__uint_as_float(node[0]), __uint_as_float(node[1]), __uint_as_float(node[2]), 0.0f);
		const float4 hi = make_float4(__uint_as_float(node[4]), __uint_as_float(node[5]), __uint_as_float(node[6]), 0.0f);
		const uint32_t skip = node[3], leaf = node[7];
		const bool here = alive && mine == at;
		const bool box = here && slab_hit(lo, hi, ray, max_distance);
		mine = here ? (box ? at + 1u : at + skip) : mine;
		const unsigned long long hit_mask = wave_ballot(box);
		if (hit_mask != 0ull && leaf != NONE) {
			const float4 *tri = tris_ptr + LEAF_F4 * leaf + LEAF_TRI_F4;
			const float4 q0 = tri[0], q1 = tri[1], q2 = tri[2], q3 = tri[3];
			if (box) {
				const TriResult tr = tri_eval<false>(q0, q1, q2, q3, ray);
				if (tr.accepted) {
					atomicAdd(occluded, 1u);
					alive = false;
				}
			}
			if (wave_ballot(alive) == 0ull)
				break;
		}
		at = (uint32_t) __builtin_amdgcn_readfirstlane((int) (at + (hit_mask != 0ull ? 1u : skip)));
	}
}


// Tile <-> workgroup mapping shared by the passes.  A workgroup of the primary
// pass covers 2x2 tiles (16x16 sub-pixels).  Workgroups b and b+8 share an XCD
// and its L2 (MI355X_MICROARCH.md, dispatch is round-robin over XCDs), so the
// image is cut into vertical strips KernelParams::strip_tiles wide (two by default),
// strips are dealt round-robin to the 8 XCD groups, and each group walks its strips
// top to bottom, row by row: neighbouring workgroups of a group touch the same BVH
// region, while every group still sees the whole image height.  Strips of two tiles
// balance best and are right while the scene lives in the caches; a scene far beyond
// the L2s gets wider ones, so that an XCD's rays mostly meet geometry that only this
// XCD needs (with 16-pixel strips all eight fetch the same nodes from HBM).

// Kernel arguments that are READ AGAIN from the kernel-argument segment where they are used -- one scalar load each
// (asm volatile: the compiler can neither hoist it out of a loop nor merge it with another) -- instead of being held in
// scalar registers, and spilled from them to VGPR lanes, across the walks (see AoArgs).
template <uint32_t OFFSET>
__device__ __forceinline__ uint32_t cold_u32() {
	uint32_t v;
	asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "i"(OFFSET));
	return v;
}
template <uint32_t OFFSET>
__device__ __forceinline__ unsigned long long cold_u64() {
	unsigned long long v;
	asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "i"(OFFSET));
	return v;
}

// The primary pass's arguments as one block (see AoArgs for why): the walk holds the two pointers at the head,
// node_count, primary_below and batch_below in registers; what the tile's epilogue and the kernel's tail need is read
// again there.
struct PrimaryArgs {
	const float4 *walk_ptr, *tris_ptr;
	const float4 *nodes_ptr, *shade;
	float *image;
	HitRec *hits;
	uint32_t *occluded_of, *tile_hits, *order;
	const uint32_t *tile_base;  // first slot of each tile in the hit list (DeviceRenderer: a prefix sum of the tiles' hit counts)
	FrameCounters *counters;
	KernelParams P;
};
#define OCRT_PCOLD_U32(FIELD) cold_u32<(uint32_t) offsetof(PrimaryArgs, FIELD)>()
#define OCRT_PCOLD_PTR(TYPE, FIELD) ((TYPE) cold_u64<(uint32_t) offsetof(PrimaryArgs, FIELD)>())

// A hit sub-pixel that still waits for its ambient-occlusion factor holds, in the float image, a TAG instead of a value:
// a negative quiet NaN whose low six bits are the sub-pixel's slot in its tile's part of the hit list (no value the
// path computes is a NaN: the head-light term is clamped to [0, 1]).  The finishing kernel puts the value there.
constexpr uint32_t PENDING_TAG = 0xFFC00000u;
__device__ __forceinline__ bool is_pending(uint32_t bits) { return (bits & 0xFFFFFFC0u) == PENDING_TAG; }

// SHARED: the shared walk.  (The A/B build also instantiates the first generation, SHARED = false; two instantiations,
// so that its per-lane state stays out of the default path's register budget.)
// One tile of the primary pass, one wave.
template <bool SHARED>
__device__ __forceinline__ void primary_tile(const PrimaryArgs &A, ClosestBatch *closest_batches, uint32_t tile_x, uint32_t local_row) {
	const KernelParams &P = A.P;  // (fields used BEFORE or IN the walk only; the epilogue reads its own again)
	const float4 *__restrict__ const walk_ptr = A.walk_ptr, *__restrict__ const tris_ptr = A.tris_ptr;
	const float4 *__restrict__ const nodes_ptr = A.nodes_ptr;  // (exact form and first-generation walk only)
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	const SceneViews scene = make_views(nodes_ptr, tris_ptr, P);
	const uint32_t tile = local_row * P.tiles_x + tile_x;
	const uint32_t tile_y = global_tile_row(P.part, local_row);
	const uint32_t x = tile_x * TILE_W + (lane & 7u);
	const uint32_t y = tile_y * TILE_H + (lane >> 3);
	const bool active = x < P.width && y < P.height;
	// the float image holds this rank's bands only, one after the other: row `local_y` of it is image row `y`
	const uint32_t local_y = local_row * TILE_H + (lane >> 3);
	const uint32_t count = P.node_count;

	// reference src/intersect_kernel.cl:279-295
	float dx = ((float) x + 0.5f) / P.a - P.half_w;
	float dy = -(((float) y + 0.5f) / P.a - P.half_h);
	float dz = -1.0f;
	normalize3(dx, dy, dz);
	const Ray ray = make_ray(0.0f, 0.0f, 2.0f, dx, dy, dz);
	Hit best;
	best.distance = __builtin_inff();
	best.leaf = 0;
	best.s = best.t = 0.0f;
	best.px = best.py = best.pz = 0.0f;
	bool hit = false;
	uint32_t leaf_stops = 0u;  // leaves the tile's shared walk stopped at: how dense the geometry is along these rays
	if (SHARED) {
		const bool exact = !P.fast_walk || wave_ballot(active && !ray_is_selectable(ray, P.origin_limit)) != 0ull;
		// closest hit = minimum of (distance, reference leaf), see nearer(); reference :106-112
		auto leaf_test = [&](uint32_t leaf, bool box) {
			const float4 *tri = tris_ptr + LEAF_F4 * leaf + LEAF_TRI_F4;
			const float4 q0 = tri[0], q1 = tri[1], q2 = tri[2], q3 = tri[3];
			if (box) {
				const TriResult tr = tri_eval<true>(q0, q1, q2, q3, ray);
				if (tr.accepted) {
					hit = true;
					if (nearer(tr.distance, leaf, best)) {
						best.distance = tr.distance;
						best.leaf = leaf;
						best.s = tr.s;
						best.t = tr.t;
						best.px = tr.px; best.py = tr.py; best.pz = tr.pz;
					}
				}
			}
		};
		if (!exact) {
			// Leaves hit by few lanes are collected and tested 64 pairs at a time (see ClosestBatch).
			ClosestBatch &cb = closest_batches[wave];
			cb.best_key[lane] = KEY_NONE;
			if (lane < 2u)
				cb.hit_bits[lane] = 0u;
			unsigned long long my_key = KEY_NONE;  // from the leaves tested on the spot
			uint32_t waiting = 0u;
			auto key_of = [](float distance, uint32_t leaf) {
				return ((unsigned long long) __float_as_uint(distance) << 32) | leaf;
			};
			auto run_batch = [&](uint32_t n) {
				wave_lds_sync();
				const uint32_t pair = cb.entry[lane < n ? lane : 0u];
				const int owner = (int) (pair >> 26);
				Ray theirs = ray;  // (all primary rays start at the eye)
				theirs.dx = __shfl(ray.dx, owner); theirs.dy = __shfl(ray.dy, owner); theirs.dz = __shfl(ray.dz, owner);
				theirs.ix = __shfl(ray.ix, owner); theirs.iy = __shfl(ray.iy, owner); theirs.iz = __shfl(ray.iz, owner);
				if (lane < n) {
					// a candidate of the padded walk: the leaf's own box decides whether the reference tests it (:189)
					const uint32_t leaf = pair & 0x03FFFFFFu;
					const float4 lo = load_f4(scene.tris, leaf * LEAF_BYTES), hi = load_f4(scene.tris, leaf * LEAF_BYTES + 16u);
					if (exact_leaf_gate(lo, hi, theirs, P.primary_below)) {
						const TriResult tr = tri_test<true>(scene.tris, leaf, theirs);
						if (tr.accepted) {
							atomicMin(&cb.best_key[owner], key_of(tr.distance, leaf));
							atomicOr(&cb.hit_bits[owner >> 5], 1u << (owner & 31));
						}
					}
				}
			};
			const unsigned long long alive_mask = wave_ballot(active);
			const SignMasks sign = sign_masks(ray);
			const uint32_t variant = walk_variant(sign, alive_mask);
			const uint32_t first = 0u;  // (the plane-form records)
			const WalkRay walk_ray = make_walk_ray(ray, 1.0f);
			const uint32_t list_lds_address = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (uintptr_t) &cb.entry[0]);  // (low half of the flat address; scalar)
			const uint32_t end = first + count * 32u;
			uint32_t at = first;  // byte offset
			while (alive_mask != 0ull && at < end) {
				uint32_t leaf = 0u;
				unsigned long long hit_mask = 0ull;
				const uint32_t status = walk_collect<false>(variant, walk_ptr, at, walk_ray, sign, P.primary_below, alive_mask, hit_mask,
				                                            leaf, waiting, leaf_stops, list_lds_address, lane << 26, P.batch_below);
				if (status == 0u)
					break;
				if (status == 1u) {
					const float4 *rec = tris_ptr + LEAF_F4 * leaf;
					const float4 lo = rec[0], hi = rec[1], q0 = rec[2], q1 = rec[3], q2 = rec[4], q3 = rec[5];
					if (((hit_mask >> lane) & 1ull) && exact_leaf_gate(lo, hi, ray, P.primary_below)) {
						const TriResult tr = tri_eval<true>(q0, q1, q2, q3, ray);
						if (tr.accepted) {
							hit = true;
							const unsigned long long key = key_of(tr.distance, leaf);
							my_key = key < my_key ? key : my_key;
						}
					}
				} else {
					run_batch(64u);
					waiting -= 64u;
					if (lane < waiting)  // the pairs beyond the batch move to the front
						cb.entry[lane] = cb.entry[64u + lane];
				}
				at += 32u;
			}
			if (waiting != 0u)
				run_batch(waiting);
			wave_lds_sync();
			const unsigned long long batched = cb.best_key[lane];
			const unsigned long long key = batched < my_key ? batched : my_key;
			hit = hit || ((cb.hit_bits[lane >> 5] >> (lane & 31u)) & 1u);
			// the nearest hit's barycentrics and position: the same test once more, on the ray's own lane.  (A
			// distance of +inf or NaN never satisfies the reference's `best.distance > distance`: `best` stays as it is.)
			if (hit && (uint32_t) (key >> 32) < INF_BITS) {
				const uint32_t leaf = (uint32_t) key;
				const TriResult tr = tri_test<true>(scene.tris, leaf, ray);
				best.distance = tr.distance;
				best.leaf = leaf;
				best.s = tr.s;
				best.t = tr.t;
				best.px = tr.px; best.py = tr.py; best.pz = tr.pz;
			}
		} else {
			uint32_t mine = 0u;
			uint32_t at = 0u;
			while (at < count) {
				const u32x8 node = scalar_load_node(nodes_ptr, at);
				const float4 lo = make_float4(__uint_as_float(node[0]), __uint_as_float(node[1]), __uint_as_float(node[2]), 0.0f);
				const float4 hi = make_float4(__uint_as_float(node[4]), __uint_as_float(node[5]), __uint_as_float(node[6]), 0.0f);
				const uint32_t skip = node[3], leaf = node[7];
				const bool box = exact_box(lo, hi, ray, 100000.0f, active, at, skip, mine);
				const bool any = wave_ballot(box) != 0ull;
				if (any && leaf != NONE) {
					leaf_test(leaf, box);
					++leaf_stops;
				}
				at = (uint32_t) __builtin_amdgcn_readfirstlane((int) (at + (any ? 1u : skip)));
			}
		}
	}
#ifdef OCRT_DEBUG_KNOBS
	if (!SHARED) {
		const bool regular = P.scene_regular && ray_is_regular(ray);
		uint32_t i = active ? 0u : count;
		Pending pending = { NONE, NONE };
		for (;;) {
			const unsigned long long walking = wave_ballot(can_walk(pending, i, count));
			const unsigned long long leaves = wave_ballot(pending.first != NONE);
			if (leaves != 0ull && ((uint32_t) __popcll(leaves) >= P.leaf_min || walking == 0ull)) {
				if (pending.first != NONE) {
					const TriResult tr = tri_test<true>(scene.tris, pending.first, ray);
					// closest hit = minimum of (distance, reference leaf), see nearer(); reference :106-112
					if (tr.accepted) {
						hit = true;
						if (nearer(tr.distance, pending.first, best)) {
							best.distance = tr.distance;
							best.leaf = pending.first;
							best.s = tr.s;
							best.t = tr.t;
							best.px = tr.px; best.py = tr.py; best.pz = tr.pz;
						}
					}
					pending.first = pending.second;
					pending.second = NONE;
				}
				continue;
			}
			if (walking == 0ull)
				break;
			advance_walkers(scene, ray, regular, 100000.0f, P.primary_below, count, i, pending);
			if ((uint32_t) __popcll(wave_ballot(pending.first != NONE)) < P.leaf_min)
				advance_walkers(scene, ray, regular, 100000.0f, P.primary_below, count, i, pending);
		}
	}
#endif

	// smooth normal and head-light term, reference :296-304
	float value = 0.0f;
	float nx = 0.0f, ny = 0.0f, nz = 0.0f;
	if (hit) {
		const float4 *const shade = OCRT_PCOLD_PTR(const float4 *, shade);
		const float4 n0 = shade[3 * (size_t) best.leaf + 0];
		const float4 n1 = shade[3 * (size_t) best.leaf + 1];
		const float4 n2 = shade[3 * (size_t) best.leaf + 2];
		const float b0 = 1.0f - best.s - best.t, b1 = best.s, b2 = best.t;
		nx = (n0.x * b0 + n1.x * b1) + n2.x * b2;
		ny = (n0.y * b0 + n1.y * b1) + n2.y * b2;
		nz = (n0.z * b0 + n1.z * b1) + n2.z * b2;
		normalize3(nx, ny, nz);
		value = 1.0f;
		if (OCRT_PCOLD_U32(P.shading))
			value = fminf(fmaxf(-dot3(nx, ny, nz, dx, dy, dz), 0.0f), 1.0f);
	}
	const bool want_ao = OCRT_PCOLD_U32(P.ao_mode) != (uint32_t) AO_NONE && OCRT_PCOLD_U32(P.ao_dirs) > 0u;
	const uint32_t image_width = OCRT_PCOLD_U32(P.width);
	// the tile's hits go into the tile's own 64 slots of the hit list, compacted
	const unsigned long long hit_mask = wave_ballot(hit);
	const uint32_t hit_count = (uint32_t) __popcll(hit_mask);
	const uint32_t slot_in_tile = rank_in(hit_mask);
	if (active)  // final already, or the tag that says which slot will bring the ambient-occlusion factor
		OCRT_PCOLD_PTR(float *, image)[(size_t) local_y * image_width + x] = (hit && want_ao) ? __uint_as_float(PENDING_TAG | slot_in_tile) : value;
	if (lane == 0u) {
		// hit count, and above it the tile's AO cost class 1..64 for the ordering step: its 28 AO packets
		// walk about as far as the primary packet did (correlation 0.8-0.9, tools/analysis/packet_union.cc).
		// The hit count does not predict the cost at all: a sparse tile's packets mix several directions
		// and walk as many nodes as a full tile's.
		uint32_t cost = SHARED ? leaf_stops : hit_count;
		cost = cost < 1u ? 1u : cost;
		cost = cost > 64u ? 64u : cost;
		// (a store that is coherent across the device: the workgroup that orders the group's tiles at the end of this very
		// kernel reads the word with a load of the same kind -- primary_kernel's tail)
		__hip_atomic_store(&OCRT_PCOLD_PTR(uint32_t *, tile_hits)[tile], hit_count | ((want_ao && hit_count) ? cost << 8 : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	// The hit list holds a tile's hits at tile_base[tile] ..., in the order of the lanes.  (tile_base is the exclusive
	// prefix sum of the tiles' hit counts -- a function of scene, options and the fixed camera, counted once per upload by
	// a pass of this kernel that has no hit list yet: `hits` is null then and nothing is recorded.)
	HitRec *const hit_list = OCRT_PCOLD_PTR(HitRec *, hits);
	if (hit && want_ao && hit_list) {
		HitRec rec;
		rec.ox = best.px; rec.oy = best.py; rec.oz = best.pz;
		rec.value = value;
		rec.nx = nx; rec.ny = ny; rec.nz = nz;
		rec.pixel = local_y * image_width + x;  // index into this rank's band image
		const size_t slot = (size_t) OCRT_PCOLD_PTR(const uint32_t *, tile_base)[tile] + slot_in_tile;
		hit_list[slot] = rec;
		OCRT_PCOLD_PTR(uint32_t *, occluded_of)[slot] = 0u;
	}
}

// ---------------------------------------------------------------------------
// Ordering step of one XCD group, run by the LAST workgroup of the primary pass that finishes in the group (below):
// blocks of 64 neighbouring tiles sorted by their AO cost (sum of the tiles' cost classes >> KernelParams::cost_shift,
// capped: the costly blocks share the top key and keep their spatial order, the cheap ones follow by cost --
// scene_pack.cc says why); the tiles of a block stay together and in spatial order (counting sort, one wave per
// block).  Also sums the group's hit sub-pixels.  Every tile word is read with a device-coherent load: the words were
// written by other workgroups of this kernel.
// ---------------------------------------------------------------------------
struct OrderScratch {
	unsigned int bucket[65];  // non-empty tiles per key, then the keys' write cursors
	unsigned int cost_total;  // sum of the group's tiles' cost classes
	unsigned int hit_total;   // hit sub-pixels of the group
	unsigned int last;        // (primary_kernel's tail: this workgroup is the group's last)
};
__device__ __forceinline__ void order_group(const uint32_t *__restrict__ tile_hits, uint32_t *__restrict__ order,
                                            FrameCounters *__restrict__ counters, uint32_t tiles_x, uint32_t local_tile_rows,
                                            uint32_t strip_tiles, bool no_sort, uint32_t cost_shift, uint32_t group,
                                            OrderScratch &scratch, uint32_t waves) {
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t strips = (tiles_x + strip_tiles - 1u) / strip_tiles;
	const uint32_t strips_here = (strips + XCD_GROUPS - 1u - group) >> 3;
	const uint32_t tiles_here = strips_here * strip_tiles * local_tile_rows;  // incl. possible columns past the image
	// this group's segment of `order` starts where the previous groups' capacity ends
	uint32_t segment = 0u;
	for (uint32_t g = 0; g < group; ++g)
		segment += ((strips + XCD_GROUPS - 1u - g) >> 3) * strip_tiles * local_tile_rows;
	if (threadIdx.x < 65u)
		scratch.bucket[threadIdx.x] = 0u;
	if (threadIdx.x == 0u) {
		scratch.cost_total = 0u;
		scratch.hit_total = 0u;
	}
	__syncthreads();
	// tile e of the group: strip (e / (strip_tiles * rows)), then row-major across the strip; returns its word (0: nothing there)
	auto word_of = [&](uint32_t e, uint32_t &tile) -> uint32_t {
		if (e >= tiles_here)
			return 0u;
		const uint32_t per_strip = strip_tiles * local_tile_rows;
		const uint32_t strip_index = e / per_strip, within = e - strip_index * per_strip;
		const uint32_t local_row = within / strip_tiles;
		const uint32_t tile_x = strip_tiles * (group + XCD_GROUPS * strip_index) + (within - local_row * strip_tiles);
		tile = local_row * tiles_x + tile_x;
		if (tile_x >= tiles_x)
			return 0u;
		return __hip_atomic_load(&tile_hits[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	};
	// One wave per block of 64 spatially consecutive tiles (a strip wide, 64 / strip_tiles high).  key: the block's cost, 1..64.
	const uint32_t n_blocks = (tiles_here + 63u) >> 6;
	auto block_key = [&](uint32_t block, uint32_t &tile, uint32_t &word, unsigned long long &work_mask, uint32_t &cost) -> uint32_t {
		word = word_of(block * 64u + lane, tile);
		const uint32_t cls = word >> 8;
		work_mask = wave_ballot(cls != 0u);
		cost = cls;
		for (int offset = 32; offset >= 1; offset >>= 1)
			cost += (uint32_t) __shfl_xor((int) cost, offset);
		const uint32_t key = no_sort ? 1u : 1u + (cost >> cost_shift);
		return key > 64u ? 64u : key;
	};
	for (uint32_t block = wave; block < n_blocks; block += waves) {
		uint32_t tile = 0u, word, cost;
		unsigned long long work_mask;
		const uint32_t key = block_key(block, tile, word, work_mask, cost);
		uint32_t hit_sum = word & 0xFFu;
		for (int offset = 32; offset >= 1; offset >>= 1)
			hit_sum += (uint32_t) __shfl_xor((int) hit_sum, offset);
		if (lane == 0u) {
			if (work_mask != 0ull) {
				atomicAdd(&scratch.bucket[key], (uint32_t) __popcll(work_mask));
				atomicAdd(&scratch.cost_total, cost);
			}
			if (hit_sum)
				atomicAdd(&scratch.hit_total, hit_sum);
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		// exclusive prefix over descending keys: costly regions first, so that the frame ends on short claims
		uint32_t running = 0u;
		for (int k = 64; k >= 1; --k) {
			const uint32_t n = scratch.bucket[k];
			scratch.bucket[k] = running;
			running += n;
		}
		counters->queue[group].work_tiles = running;
		counters->queue[group].cost_sum = scratch.cost_total;
		counters->queue[group].hits = scratch.hit_total;
		counters->queue[group].head = 0u;
	}
	__syncthreads();
	for (uint32_t block = wave; block < n_blocks; block += waves) {
		uint32_t tile = 0u, word, cost;
		unsigned long long work_mask;
		const uint32_t key = block_key(block, tile, word, work_mask, cost);
		if (work_mask == 0ull)
			continue;
		uint32_t base = 0u;
		if (lane == 0u)
			base = atomicAdd(&scratch.bucket[key], (uint32_t) __popcll(work_mask));
		base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
		// entry = tile (26 bits: at most 2^32 sub-pixels per frame) | hit count - 1 (6 bits); inside a block
		// the tiles keep their spatial order
		if (word >> 8)
			order[segment + base + rank_in(work_mask)] = tile | (((word & 0xFFu) - 1u) << 26);
	}
	if (threadIdx.x == 0) {
		counters->queue[group].tick_ordered = (unsigned long long) __builtin_amdgcn_s_memrealtime();
#ifdef OCRT_TAIL  // (the AO pass starts right after this kernel: its waves' end times are counted from here)
		atomicMax(&counters->stamp[7], __builtin_amdgcn_s_memrealtime());
#endif
	}
}

// ---------------------------------------------------------------------------
// Pass 1: primary rays, then -- the last workgroup of each XCD group -- the group's ordering step.  Four waves per
// workgroup, one tile each; they meet once, at the end.
//
// Why the ordering step lives here and not in a kernel of its own: it is 8 workgroups of work, and as a kernel it cost
// a frame that shares its GPU ~0.2 ms of waiting (a launch boundary on either side, and workgroups of 1024 threads that
// need 16 free wave slots on one CU while other frames' persistent passes hold them).  The hand-over inside the kernel:
// every wave's tile word is a device-coherent store (primary_tile), drained (s_waitcnt vmcnt(0)) before the workgroup's
// barrier; then ONE returning atomic per workgroup on the group's `done` counter -- whoever takes it to the number of
// the group's workgroups is the last, reads the words with device-coherent loads and puts the counter back to 0 for the
// next frame.  (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores drained before the counter, sc1 loads after it.)
// ---------------------------------------------------------------------------
#ifndef OCRT_PRIMARY_WAVES
#define OCRT_PRIMARY_WAVES 4
#endif
constexpr uint32_t PRIMARY_WAVES = OCRT_PRIMARY_WAVES;  // 4, 8 or 16: a workgroup covers a block of tiles 2 wide and PRIMARY_WAVES / 2 high
constexpr uint32_t PRIMARY_ROWS = PRIMARY_WAVES / 2u;

template <bool SHARED>
__global__ __launch_bounds__(64 * PRIMARY_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void primary_kernel(PrimaryArgs A) {
	__shared__ ClosestBatch closest_batches[PRIMARY_WAVES];
	__shared__ OrderScratch scratch;
	const uint32_t wave = threadIdx.x >> 6;
	if (blockIdx.x == 0u && threadIdx.x == 0u) {
		FrameCounters *const counters = A.counters;
		counters->tick_begin = __builtin_amdgcn_s_memrealtime();
		// the sums the LATER kernels of this frame add to (nobody touches them before this kernel has ended)
		counters->occluded = 0ull;
		counters->tick_ao_end = 0ull;
	}
	const uint32_t group = blockIdx.x & (XCD_GROUPS - 1u), seq = blockIdx.x >> 3;
	{
		const uint32_t strip_tiles = A.P.strip_tiles, columns = strip_tiles >> 1;  // (a workgroup is two tiles wide)
		const uint32_t strips = (A.P.tiles_x + strip_tiles - 1u) / strip_tiles;
		const uint32_t row_blocks = (A.P.local_tile_rows + PRIMARY_ROWS - 1u) / PRIMARY_ROWS;
		const uint32_t strips_here = (strips + XCD_GROUPS - 1u - group) >> 3;
		const uint32_t per_strip = row_blocks * columns;
		const uint32_t strip_index = seq / per_strip;
		const uint32_t rest = seq - strip_index * per_strip;
		const uint32_t row_block = rest / columns;
		const uint32_t tile_x = strip_tiles * (group + XCD_GROUPS * strip_index) + 2u * (rest - row_block * columns) + (wave & 1u);
		const uint32_t local_row = PRIMARY_ROWS * row_block + (wave >> 1);
		if (seq < strips_here * per_strip && tile_x < A.P.tiles_x && local_row < A.P.local_tile_rows)
			primary_tile<SHARED>(A, closest_batches, tile_x, local_row);
	}
	// ---- the tail: is this the group's last workgroup? ----
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's tile word has left)
	__syncthreads();
	if (threadIdx.x == 0u) {
		FrameCounters *const counters = OCRT_PCOLD_PTR(FrameCounters *, counters);
		const uint32_t before = __hip_atomic_fetch_add(&counters->queue[group].done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		scratch.last = before + 1u == gridDim.x / XCD_GROUPS ? 1u : 0u;
	}
	__syncthreads();
	if (scratch.last == 0u)
		return;
	FrameCounters *const counters = OCRT_PCOLD_PTR(FrameCounters *, counters);
	order_group(OCRT_PCOLD_PTR(const uint32_t *, tile_hits), OCRT_PCOLD_PTR(uint32_t *, order), counters, OCRT_PCOLD_U32(P.tiles_x),
	            OCRT_PCOLD_U32(P.local_tile_rows), OCRT_PCOLD_U32(P.strip_tiles), OCRT_PCOLD_U32(P.debug_no_sort) != 0u,
	            OCRT_PCOLD_U32(P.cost_shift), group, scratch, PRIMARY_WAVES);
	if (threadIdx.x == 0u)
		__hip_atomic_store(&counters->queue[group].done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// xorshift128 of the RANDOM hemisphere sampler, reference src/intersect_kernel.cl:128-152.
struct Rng {
	uint32_t x, y, z, w;
};
__device__ __forceinline__ uint32_t rng_next(Rng &v) {
	const uint32_t t = v.x ^ (v.x << 11u);
	v.x = v.y;
	v.y = v.z;
	v.z = v.w;
	return v.w = v.w ^ (v.w >> 19u) ^ (t ^ (t >> 8u));
}
__device__ __forceinline__ Rng rng_seed(uint32_t seed) {
	Rng v;
	v.x = (123456789u ^ seed) * 88675123u;
	v.y = (362436069u ^ seed) * 123456789u;
	v.z = (521288629u ^ seed) * 362436069u;
	v.w = (88675123u ^ seed) * 521288629u;
	rng_next(v);
	return v;
}
__device__ __forceinline__ float rng_float(Rng &v) { return 2.32830643653869629E-10f * rng_next(v); }

// ---------------------------------------------------------------------------
// Pass 2: ambient occlusion.  Persistent, independent waves; one tile at a time.
// ---------------------------------------------------------------------------
constexpr uint32_t AO_WAVES = AO_WORKGROUP_WAVES;  // (workgroups per CU: DeviceRenderer::aoWorkgroups, 8 for a host alone)

// LDS slice of one wave: the tile's hit table, structure of arrays and lane-major
// so that consecutive hits sit in consecutive banks.
struct TileShared {
	float frame[12][64];  // origin xyz, basis_x xyz, basis_y xyz, basis_z xyz
	unsigned int occluded[64];
	unsigned int pixel[64];  // RANDOM mode: the sub-pixel's image index seeds its generator
	LeafBatch batch;
};


// MODE is AO_UNIFORM or AO_RANDOM: two instantiations, so that the RANDOM sampler's
// code and registers stay out of the default path.
#ifdef OCRT_STAMPS
#define OCRT_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define OCRT_STAMP_ADD(slot, value) stamp_acc[slot] += (unsigned long long) (value)
#else
#define OCRT_STAMP(var)
#define OCRT_STAMP_ADD(slot, value)
#endif

// The pass's arguments as ONE block.  The walks leave the kernel some 40 scalar registers for everything it holds across
// them (80 per wave at 8 waves per SIMD, 22 of them the node loop's own and 14 its operands), and what does not fit is
// spilled to VGPR lanes: v_writelane / v_readlane -- VECTOR instructions, the resource the pass is bound by (round 3:
// 33 per packet and 9 per leaf stop or batch, a tenth of the pass's vector instructions).  So only what every leaf stop
// needs is held in registers (the two pointers at the head, node_count, ao_below, batch_below); every other argument is
// READ AGAIN from the kernel-argument segment where it is used -- one scalar load (asm volatile: the compiler can neither
// hoist it out of a loop nor merge it with another) that hits the scalar cache and costs no vector issue slot.
struct AoArgs {
	const float4 *walk_ptr, *tris_ptr;
	const float4 *nodes_ptr, *ao_table;
	const HitRec *hits;
	uint32_t *occluded_of;
	const uint32_t *order;
	const uint32_t *tile_base;  // first slot of each tile in the hit list
	const uint2 *tile_entry;    // per tile, 1 + ao_dirs byte ranges of the walk records: what its any-hit rays have to walk (entry_kernel)
	FrameCounters *counters;
	KernelParams P;
};
#define OCRT_COLD_U32(FIELD) cold_u32<(uint32_t) offsetof(AoArgs, FIELD)>()
#define OCRT_COLD_F32(FIELD) __uint_as_float(cold_u32<(uint32_t) offsetof(AoArgs, FIELD)>())
#define OCRT_COLD_PTR(TYPE, FIELD) ((TYPE) cold_u64<(uint32_t) offsetof(AoArgs, FIELD)>())

// PREFETCH: the node loop touches a pair's two successors ahead of time (OCRT_PF_SUCCESSORS): a launch-time choice.
template <int MODE, bool SHARED, bool PREFETCH = false>
__global__ __launch_bounds__(64 * AO_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void ao_kernel(AoArgs A) {
	__shared__ TileShared shared_tiles[AO_WAVES];
	__shared__ unsigned int wg_claim[4];  // the workgroup's current claim: first unit, units per wave, end (if dealt by cursor), cursor
	const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) threadIdx.x) >> 6;  // (scalar)
	TileShared &sh = shared_tiles[wave];
	const float4 *__restrict__ const walk_ptr = A.walk_ptr, *__restrict__ const tris_ptr = A.tris_ptr;
	const uint32_t count = A.P.node_count;
#ifdef OCRT_DEBUG_KNOBS  // (the first-generation walk of the A/B build uses the arguments freely: its register budget is nobody's concern)
	const KernelParams &P = A.P;
	const SceneViews scene = make_views(A.nodes_ptr, A.tris_ptr, A.P);
#endif

#ifndef OCRT_STAMPS
	unsigned long long *walk_prof = nullptr;
#endif
#ifdef OCRT_STAMPS
	unsigned long long stamp_acc[6] = { 0, 0, 0, 0, 0, 0 };
	const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
	unsigned long long t_last_claim = t_begin;
	unsigned long long walk_prof_store[7] = { 0, 0, 0, 0, 0, 0, 0 };
	unsigned long long *walk_prof = walk_prof_store;
#endif
	// Workgroups b and b+8 share an XCD: start with that group's queue, then help the others.
	const uint32_t home = blockIdx.x & (XCD_GROUPS - 1u);
	for (uint32_t turn = 0; turn < XCD_GROUPS; ++turn) {
		const uint32_t group = (home + turn) & (XCD_GROUPS - 1u);
		uint32_t segment = 0u;
		{
			const uint32_t strip_tiles = OCRT_COLD_U32(P.strip_tiles);
			const uint32_t strips = (OCRT_COLD_U32(P.tiles_x) + strip_tiles - 1u) / strip_tiles, rows = OCRT_COLD_U32(P.local_tile_rows);
			for (uint32_t g = 0; g < group; ++g)
				segment += ((strips + XCD_GROUPS - 1u - g) >> 3) * strip_tiles * rows;
		}
		// the group's work in units of (tile, table direction), tile-major
		uint32_t units, claim_max;
		{
		FrameCounters *const counters = OCRT_COLD_PTR(FrameCounters *, counters);
		const uint32_t ao_dirs = OCRT_COLD_U32(P.ao_dirs);
		// (loads through a re-read pointer are vector loads -- the compiler cannot know the memory to be constant --: what they
		// return is made scalar again by hand)
		const uint32_t queued_tiles = (uint32_t) __builtin_amdgcn_readfirstlane((int) counters->queue[group].work_tiles);
		units = queued_tiles * ao_dirs;
		// A wave's largest claim.  A quarter of a tile's directions, so that the workgroup's four waves take ONE tile
		// together (the best locality, and the finest balance); whole tiles per wave where packets are cheap and
		// plentiful -- a tile's mean cost class (the leaves its primary packet stopped at) below 8 and 512 or more
		// units per wave: 16+ samples per pixel -- because there the ~12 us of set-up per claim (hit records,
		// tangent frames) weigh more than the locality.  Swept per workload: profiles/r02_notes.md.
		claim_max = OCRT_COLD_U32(P.ao_claim_max);
		if (claim_max == 0u) {
			const uint32_t tiles = queued_tiles, cost = (uint32_t) __builtin_amdgcn_readfirstlane((int) counters->queue[group].cost_sum);
			const uint32_t claim_div = OCRT_COLD_U32(P.ao_claim_div);
			const bool cheap_and_plenty = cost < 8u * tiles && units >= 512u * claim_div;
			const uint32_t quarter = (ao_dirs + AO_WAVES - 1u) / AO_WAVES;
			// ... and less than a quarter where work is scarce (one GPU's share of a frame split eight ways holds 8
			// units per wave): half a quarter below 24 units per wave, a third below 12 -- the frame then ends when its
			// heaviest tile does, and more waves should share that one (tools/partition_probe.py: -13 % at 1/8).  Not
			// where other frames run beside this one: what a pass leaves idle at its end is theirs, and the smaller
			// claims only cost (an eighth of the headline frame, six frames in flight: 0.25 ms with them, 0.21 without).
			const uint32_t per_wave = OCRT_COLD_U32(P.shared_device) ? 24u : units / claim_div;
			claim_max = cheap_and_plenty ? ao_dirs : per_wave < 12u ? quarter / 3u : per_wave < 24u ? (quarter + 1u) / 2u : quarter;
			claim_max = claim_max < 1u ? 1u : claim_max;
		}
		}
		for (;;) {
			OCRT_STAMP(t_claim);
			// The WORKGROUP claims (thread 0: a plain load first -- most visits to a foreign group find its queue
			// drained, and a load does not queue up behind the other workgroups' atomics --, then one returning
			// atomic), and its four waves share the claim: they then work on the same tile, or on neighbouring
			// ones, at the same time, and share its nodes in the CU's scalar cache (63 % of the scalar loads of the
			// per-wave claims missed it, profiles/r02_notes.md) and its hit records in L2.  A claim is four times
			// claim_max units, to the end of the queue; how the waves divide it is decided below.  (Guided self-scheduling -- the share
			// shrinking to 1/ao_guide of what is left per wave of the group -- is kept behind OCRT_AO_GUIDE: the
			// single directions it hands out at the end cost a claim each, two barriers and a tile set-up, and
			// lengthened the pass by 3-5 %; the costly tiles are claimed first anyway: order_group.)
			uint32_t per_wave = 0u, first = units;  // (wave 0's, scalar; in registers until the siblings are done with the last claim)
			if (wave == 0u) {
				FrameCounters *const counters = OCRT_COLD_PTR(FrameCounters *, counters);
				uint32_t seen = 0u;
				if (fresh_lane() == 0u)
					seen = __hip_atomic_load(&counters->queue[group].head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				seen = (uint32_t) __builtin_amdgcn_readfirstlane((int) seen);
				if (seen < units) {
					const uint32_t guide = OCRT_COLD_U32(P.ao_guide);
					per_wave = guide ? (units - seen) / guide : claim_max;
					per_wave = per_wave < 1u ? 1u : per_wave > claim_max ? claim_max : per_wave;
					uint32_t got = 0u;
					if (fresh_lane() == 0u)
						got = atomicAdd(&counters->queue[group].head, per_wave * AO_WAVES);
					first = (uint32_t) __builtin_amdgcn_readfirstlane((int) got);
				}
			}
			__syncthreads();  // (everybody is done with the previous claim: its words in LDS, the cursor among them, are free)
			if (wave == 0u && fresh_lane() == 0u) {
				// A claim that lies in ONE tile (the rule: four quarters of a tile's directions) is not dealt out in fixed
				// quarters: the four waves take its directions from a cursor in LDS, a packet's worth at a time, so that they
				// finish within a packet of each other -- with fixed quarters a wave spent 12-18 % of its life waiting for the
				// slowest sibling at the barrier (profiles/r02_notes.md; what the cursor buys and where it does not:
				// profiles/r03_notes.md).  Other claims (whole tiles per wave, the short ones of scarce work) keep fixed shares.
				// wg_claim[2] = the claim's end, 0 for fixed shares.
				const uint32_t end = first + per_wave * AO_WAVES < units ? first + per_wave * AO_WAVES : units;
				const uint32_t ao_dirs = OCRT_COLD_U32(P.ao_dirs);
				const bool one_tile = SHARED && first < units && first / ao_dirs == (end - 1u) / ao_dirs && ao_dirs < 0x8000u;
				wg_claim[0] = first;
				wg_claim[1] = per_wave;
				wg_claim[2] = one_tile ? end : 0u;
				// the cursor, in directions of the tile: end << 16 | next (both 0 for fixed shares: nothing to take)
				const uint32_t base = first - first / ao_dirs * ao_dirs;
				wg_claim[3] = one_tile ? (base + (end - first)) << 16 | base : 0u;
			}
			__syncthreads();
			const uint32_t wg_claimed = (uint32_t) __builtin_amdgcn_readfirstlane((int) wg_claim[0]);
			const uint32_t want = (uint32_t) __builtin_amdgcn_readfirstlane((int) wg_claim[1]);
			if (wg_claimed >= units)
				break;  // (the same for all four waves)
			// fixed shares: this wave's quarter; a claim with a cursor: one job, the tile, for every wave
			const bool dealt_by_cursor = __builtin_amdgcn_readfirstlane((int) wg_claim[2]) != 0;
			const uint32_t claimed = dealt_by_cursor ? wg_claimed : wg_claimed + wave * want;
			if (claimed >= units)
				continue;  // nothing left for this wave; it meets the others again at the next claim
			const uint32_t claim_end = dealt_by_cursor ? claimed + 1u : claimed + want < units ? claimed + want : units;
			OCRT_STAMP(t_claimed);
			OCRT_STAMP_ADD(0, t_claimed - t_claim);
#ifdef OCRT_STAMPS
			t_last_claim = t_claimed;
#endif
			for (uint32_t unit = claimed; unit < claim_end;) {
				OCRT_STAMP(t_job);
				// job = (tile, direction range); a claim that runs over a tile's last direction goes on in
				// the next tile.  Neighbouring claims work on the same tile: its hit records are shared in L2.
				uint32_t tile_index, dir0, n_dirs;
				{
					const uint32_t ao_dirs = OCRT_COLD_U32(P.ao_dirs);
					tile_index = unit / ao_dirs;
					dir0 = unit - tile_index * ao_dirs;
					n_dirs = ao_dirs - dir0 < claim_end - unit ? ao_dirs - dir0 : claim_end - unit;
				}
				unit += n_dirs;
				const uint32_t entry = (uint32_t) __builtin_amdgcn_readfirstlane((int) OCRT_COLD_PTR(const uint32_t *, order)[segment + tile_index]);
				const uint32_t tile = entry & 0x03FFFFFFu;
				const uint32_t hit_count = (entry >> 26) + 1u;
				uint32_t total = hit_count * n_dirs;  // the rays of this piece of the job: directions dir0 ...
				// A claim dealt by the cursor: the next piece of the tile's directions -- enough to fill a packet, twice that
				// at most (64 >> floor(log2(hit_count)) directions) -- or nothing, if the siblings have taken them all.
				auto take_from_cursor = [&]() {
					const uint32_t chunk = 64u >> (31u - (uint32_t) __builtin_clz(hit_count));
					// one LDS atomic, issued by lane 0 alone: the word holds the claim's end above its cursor (directions of the tile)
					uint32_t word, scratch;
					unsigned long long saved;
					asm volatile("s_mov_b64 %[saved], exec\n\t"
					             "s_mov_b64 exec, 1\n\t"
					             "v_mov_b32 %[scratch], %[address]\n\t"
					             "v_mov_b32 %[word], %[chunk]\n\t"
					             "ds_add_rtn_u32 %[word], %[scratch], %[word]\n\t"
					             "s_waitcnt lgkmcnt(0)\n\t"
					             "s_mov_b64 exec, %[saved]"
					             : [word] "=&v"(word), [scratch] "=&v"(scratch), [saved] "=&s"(saved)
					             : [address] "s"((uint32_t) (uintptr_t) &wg_claim[3]), [chunk] "s"(chunk)
					             : "memory");
					word = (uint32_t) __builtin_amdgcn_readfirstlane((int) word);
					const uint32_t at = word & 0xFFFFu, end = word >> 16;
					const uint32_t left = at < end ? end - at : 0u;
					dir0 = at;
					total = hit_count * (left < chunk ? left : chunk);
				};
				if (SHARED && __builtin_amdgcn_readfirstlane((int) wg_claim[2]) != 0) {
					take_from_cursor();
					if (total == 0u)
						continue;  // (not even the tile's table is needed)
				}

				// ---- the tile's tangent frames -> this wave's LDS slice (reference :215-236) ----
				{
				const uint32_t lane = fresh_lane();  // (recomputed where it is needed: no register held across the walks)
				if (lane < hit_count) {
					const size_t slot = (size_t) OCRT_COLD_PTR(const uint32_t *, tile_base)[tile] + lane;
					const float4 *const hits = OCRT_COLD_PTR(const float4 *, hits);
					const float4 q0 = hits[2 * slot];
					const float4 q1 = hits[2 * slot + 1];
					float nx = q1.x, ny = q1.y, nz = q1.z;
					// p = point + normal * (1.0f / 100000.0f)
					const float eps = 1.0f / 100000.0f;
					sh.frame[0][lane] = q0.x + nx * eps;
					sh.frame[1][lane] = q0.y + ny * eps;
					sh.frame[2][lane] = q0.z + nz * eps;
					sh.pixel[lane] = __float_as_uint(q1.w);
					if (MODE == AO_RANDOM)
						normalize3(nx, ny, nz);  // hemisphere_sampler normalises once more, reference :155
					// tangent frame: the smallest |component| of the normal is replaced by 1
					float hx = nx, hy = ny, hz = nz;
					const float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
					if (ax <= ay && ax <= az)
						hx = 1.0f;
					else if (ay <= ax && ay <= az)
						hy = 1.0f;
					else if (az <= ax && az <= ay)
						hz = 1.0f;
					// basis_x = normalize(cross(h, basis_y)), basis_z = normalize(cross(basis_x, basis_y))
					float bxx = hy * nz - hz * ny, bxy = hz * nx - hx * nz, bxz = hx * ny - hy * nx;
					normalize3(bxx, bxy, bxz);
					float bzx = bxy * nz - bxz * ny, bzy = bxz * nx - bxx * nz, bzz = bxx * ny - bxy * nx;
					normalize3(bzx, bzy, bzz);
					sh.frame[3][lane] = bxx; sh.frame[4][lane] = bxy; sh.frame[5][lane] = bxz;
					sh.frame[6][lane] = nx;  sh.frame[7][lane] = ny;  sh.frame[8][lane] = nz;
					sh.frame[9][lane] = bzx; sh.frame[10][lane] = bzy; sh.frame[11][lane] = bzz;
				}
				sh.occluded[lane] = 0u;
				if (lane < 2u)
					sh.batch.occluded_bits[lane] = 0u;
				}
				wave_lds_sync();
				OCRT_STAMP(t_frames);
				OCRT_STAMP_ADD(1, t_frames - t_job);

				// ---- the tile's hit_count * ao_dirs any-hit rays (reference :237-255).  Queue
				// order is direction-major, so neighbouring lanes cast the same table direction
				// from neighbouring pixels. ----
				uint32_t h = 0;
				Ray ray;
				bool tame = false;  // (wave-uniform) every ray of the packet set up last is "tame": ray_is_tame
#ifdef OCRT_DEBUG_KNOBS
				uint32_t next = 0u;  // wave-uniform queue head
				uint32_t i = count;
				Pending pending = { NONE, NONE };
				bool regular = true;
#endif

				// ray number `item` of the job -> this lane
				// `whole` (wave-uniform): the tile is full and the 64 rays are one table direction, `shared_dir`
				auto setup_ray = [&](uint32_t item, bool whole, const float4 shared_dir) {
					uint32_t k;
					float xs = shared_dir.x, ys = shared_dir.y, zs = shared_dir.z;
					bool along_normal = false;
					if (whole) {
						k = item >> 6;
						h = item & 63u;
					} else {
						k = item / hit_count;
						h = item - k * hit_count;
						if (MODE == AO_UNIFORM) {
							const float4 dir = OCRT_COLD_PTR(const float4 *, ao_table)[dir0 + k];
							xs = dir.x; ys = dir.y; zs = dir.z;
						}
					}
					if (MODE != AO_UNIFORM) {
						// RANDOM (reference :153-183, :257-276): ray 0 goes along the normal, ray
						// j >= 1 uses draws 2j-2 and 2j-1 of the sub-pixel's generator.  Device libm
						// rounds differently from the host's: this mode is outside the bit-exact contract.
						const uint32_t j = dir0 + k;
						along_normal = j == 0u;
						// the generator is seeded with the sub-pixel's index in the WHOLE image (reference :169, :279-281)
						const uint32_t local_y = sh.pixel[h] / A.P.width, x = sh.pixel[h] - local_y * A.P.width;
						const uint32_t y = global_tile_row(A.P.part, local_y / TILE_H) * TILE_H + (local_y & (TILE_H - 1u));
						Rng rng = rng_seed(536870923u * (y * A.P.width + x));
						for (uint32_t skip = 1; skip < j; ++skip) {
							rng_next(rng);
							rng_next(rng);
						}
						const float xi1 = rng_float(rng);
						const float xi2 = rng_float(rng);
						const float theta = acosf(sqrtf(1.0f - xi1));
						const float phi = (float) (2.0 * (double) xi2);
						xs = sinf(theta) * cospif(phi);
						ys = cosf(theta);
						zs = sinf(theta) * sinpif(phi);
					}
					// ray_dir = basis_x * xs + basis_y * ys + basis_z * zs, lane by lane
					float rx = (sh.frame[3][h] * xs + sh.frame[6][h] * ys) + sh.frame[9][h] * zs;
					float ry = (sh.frame[4][h] * xs + sh.frame[7][h] * ys) + sh.frame[10][h] * zs;
					float rz = (sh.frame[5][h] * xs + sh.frame[8][h] * ys) + sh.frame[11][h] * zs;
					if (MODE == AO_RANDOM) {
						normalize3(rx, ry, rz);
						if (along_normal) {
							// the un-normalised shading normal itself (:263), kept in the hit record
							const float4 q1 = ((const float4 *) A.hits)[2 * ((size_t) A.tile_base[tile] + h) + 1];
							rx = q1.x; ry = q1.y; rz = q1.z;
						}
					}
					const float ox = sh.frame[0][h], oy = sh.frame[1][h], oz = sh.frame[2][h];
					// (only the lanes with a ray are here: the ballot is over the packet's live lanes)
					tame = MODE == AO_UNIFORM && wave_ballot(!ray_is_tame(ox, oy, oz, rx, ry, rz, OCRT_COLD_F32(P.origin_limit))) == 0ull;
					if (tame) {
						ray.ox = ox; ray.oy = oy; ray.oz = oz;
						ray.dx = rx; ray.dy = ry; ray.dz = rz;
						ray.ix = short_reciprocal(rx); ray.iy = short_reciprocal(ry); ray.iz = short_reciprocal(rz);
					} else {  // (rare: a zero or tiny direction component, a NaN from a zero-length normal)
						ray.ox = ox; ray.oy = oy; ray.oz = oz;
						ray.dx = rx; ray.dy = ry; ray.dz = rz;
						ray.ix = 1.0f / rx; ray.iy = 1.0f / ry; ray.iz = 1.0f / rz;
					}
#ifdef OCRT_DEBUG_KNOBS
					regular = P.scene_regular && P.ao_regular && ray_is_regular(ray);
#endif
				};

#ifdef OCRT_DEBUG_KNOBS
				// Every lane walks on its own; idle lanes are refilled from the job's rays while
				// next < total.
				auto walk_individually = [&]() {
					for (;;) {
						const bool idle_lane = pending.first == NONE && !(i < count);
						const unsigned long long walking = wave_ballot(can_walk(pending, i, count));
						const uint32_t n_leaves = (uint32_t) __popcll(wave_ballot(pending.first != NONE));
						const unsigned long long idle_mask = wave_ballot(idle_lane);
						const uint32_t idle = (uint32_t) __popcll(idle_mask);
						if (next < total && idle >= P.refill_min) {
							const uint32_t item = next + rank_in(idle_mask);
							if (idle_lane && item < total) {
								setup_ray(item, false, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
								i = 0u;
							}
							next += idle;
							continue;
						}
						if (n_leaves != 0u && (n_leaves >= P.leaf_min || walking == 0ull)) {
							if (pending.first != NONE) {
								const TriResult tr = tri_test<false>(scene.tris, pending.first, ray);
								pending.first = pending.second;
								pending.second = NONE;
								if (tr.accepted) {
									atomicAdd(&sh.occluded[h], 1u);
									i = count;  // any-hit: the reference walks on but only uses the boolean (:251)
									pending.first = NONE;
								}
							}
							continue;
						}
						if (walking == 0ull)
							break;
						advance_walkers(scene, ray, regular, P.ao_max_distance, P.ao_below, count, i, pending);
						// a second node straight away while few leaves are pending: halves the scheduling overhead
						if ((uint32_t) __popcll(wave_ballot(pending.first != NONE)) < P.leaf_min)
							advance_walkers(scene, ray, regular, P.ao_max_distance, P.ao_below, count, i, pending);
					}
				};
#endif

#ifdef OCRT_DEBUG_KNOBS
				if (!SHARED)
					walk_individually();
#endif
				if (SHARED) {
					// shared walks (see walk_collect) of 64 consecutive rays of the job at a time; a lane
					// leaves at its first accepted triangle
					const bool scene_fast = OCRT_COLD_U32(P.fast_walk) && OCRT_COLD_U32(P.ao_regular) && OCRT_COLD_F32(P.walk_scale) > 0.0f;
#ifdef OCRT_STAMPS
					uint32_t job_exact = 0u;
#endif
					do {  // (once per piece: fixed shares are one piece, the cursor hands out the others)
					OCRT_STAMP_ADD(5, (total + 63u) / 64u);
					for (uint32_t base = 0u; base < total; base += 64u) {
						const uint32_t lane = fresh_lane();
						bool alive = base + lane < total;
						// a full tile's packet is one table direction: the entry comes by a scalar load
						const bool whole = hit_count == 64u;
						float4 shared_dir = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
						if (whole && MODE == AO_UNIFORM)
							shared_dir = OCRT_COLD_PTR(const float4 *, ao_table)[dir0 + (base >> 6)];
						// the interval of the walk array this packet has to walk (entry_kernel): the one of its table direction,
						// or the tile's.  Two words, made scalar by hand like every load through a re-read pointer.
						uint32_t entry_begin, entry_end;
						{
							const uint32_t stride = OCRT_COLD_U32(P.entry_stride);  // (1: this frame keeps the tiles' own intervals only)
							const uint32_t which = (whole && MODE == AO_UNIFORM && stride > 1u) ? 1u + dir0 + (base >> 6) : 0u;
							const uint2 range = OCRT_COLD_PTR(const uint2 *, tile_entry)[(size_t) tile * stride + which];
							entry_begin = (uint32_t) __builtin_amdgcn_readfirstlane((int) range.x);
							entry_end = (uint32_t) __builtin_amdgcn_readfirstlane((int) range.y);
						}
						if (alive)
							setup_ray(base + lane, whole, shared_dir);
						const bool exact = !scene_fast || (!tame && wave_ballot(alive && !ray_is_selectable(ray, OCRT_COLD_F32(P.origin_limit))) != 0ull);
#ifdef OCRT_STAMPS
						job_exact += exact ? 1u : 0u;
#endif
						if (exact)
							shared_walk_any_hit<true>(OCRT_COLD_PTR(const float4 *, nodes_ptr), walk_ptr, tris_ptr, count, ray, sh.frame, h,
							                          OCRT_COLD_F32(P.ao_max_distance), A.P.ao_below, 0.0f, alive, false, &sh.occluded[h], sh.batch,
							                          A.P.batch_below, walk_prof);
						else
							shared_walk_any_hit<false, PREFETCH>(nullptr, walk_ptr, tris_ptr, count, ray, sh.frame, h,
							                           0.0f, A.P.ao_below, OCRT_COLD_F32(P.walk_scale), alive, tame, &sh.occluded[h], sh.batch,
							                           A.P.batch_below, walk_prof, entry_begin, entry_end);
					}
					take_from_cursor();  // (fixed shares: the cursor holds nothing)
					} while (total != 0u);
#ifdef OCRT_STAMPS
					if (fresh_lane() == 0u && job_exact) {
						atomicAdd(&A.counters->stamp[63], (unsigned long long) job_exact);  // packets that took the exact form
						atomicAdd(&A.counters->stamp[64], (__builtin_amdgcn_s_memrealtime() - t_frames));  // ... and the time of the jobs holding them
					}
#endif
				}
				wave_lds_sync();
				OCRT_STAMP(t_walked);
				OCRT_STAMP_ADD(2, t_walked - t_frames);
				OCRT_STAMP_ADD(4, 1);

				// ---- this job's share of the occlusion counts ----
				{
					const uint32_t lane = fresh_lane();
					if (lane < hit_count) {
						const uint32_t occluded = sh.occluded[lane];
						if (occluded)
							atomicAdd(&OCRT_COLD_PTR(uint32_t *, occluded_of)[(size_t) OCRT_COLD_PTR(const uint32_t *, tile_base)[tile] + lane], occluded);
					}
				}
				wave_lds_sync();
				OCRT_STAMP(t_flushed);
				OCRT_STAMP_ADD(3, t_flushed - t_walked);
#ifdef OCRT_STAMPS
				if (fresh_lane() == 0u) {  // jobs by duration: bucket k holds those of 2^k .. 2^(k+1) microseconds
					const unsigned long long us = (t_flushed - t_job) / 100ull;
					const int bucket = us == 0ull ? 0 : 63 - __builtin_clzll(us);
					atomicAdd(&A.counters->stamp[49 + (bucket > 15 ? 15 : bucket)], 1ull);
				}
#endif
			}
		}
	}
	if (wave == 0u && fresh_lane() == 0u)  // (nothing kept across the pass: one clock read and one atomic per workgroup)
		atomicMax(&OCRT_COLD_PTR(FrameCounters *, counters)->tick_ao_end, (unsigned long long) __builtin_amdgcn_s_memrealtime());
#ifdef OCRT_TAIL  // minimal: nothing is kept across the pass, one load and two atomics when the wave ends
	if (fresh_lane() == 0u) {
		const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
		const unsigned long long origin = __hip_atomic_load(&A.counters->stamp[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const unsigned long long end_bucket = (t_end - origin) / 5000ull;  // 0.05 ms
		atomicAdd(&A.counters->stamp[10 + (end_bucket > 31 ? 31 : end_bucket)], 1ull);
		atomicMax(&A.counters->stamp[8], t_end);
	}
#endif
#ifdef OCRT_STAMPS
	if (fresh_lane() == 0u) {
		const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
		for (int k = 0; k < 6; ++k)
			atomicAdd(&A.counters->stamp[k], stamp_acc[k]);           // claim, frames, walks, flush (10 ns ticks); jobs, packets
		atomicAdd(&A.counters->stamp[6], t_end - t_begin);            // sum of wave lifetimes
		atomicMin(&A.counters->stamp[7], t_begin);                     // first start
		atomicMax(&A.counters->stamp[8], t_end);                       // last end
		if (stamp_acc[4])
			atomicAdd(&A.counters->stamp[9], 1ull);                    // waves that got any work
		for (int k = 0; k < 7; ++k)
			atomicAdd(&A.counters->stamp[42 + k], walk_prof_store[k]);  // time in the node loop, in batches; loop entries, batches, leaf stops
		// when this wave ended, counted from the first wave's start (settled long before any wave ends), 0.1 ms buckets:
		// how the occupancy decays towards the end of the launch
		(void) t_last_claim;
		const unsigned long long first = __hip_atomic_load(&A.counters->stamp[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const unsigned long long end_bucket = (t_end - (first < t_begin ? first : t_begin)) / 10000ull;
		atomicAdd(&A.counters->stamp[10 + (end_bucket > 31 ? 31 : end_bucket)], 1ull);
	}
#endif
}

// ---------------------------------------------------------------------------
// Once per upload (camera, scene and options are fixed, so a tile's hit points are the same in every frame): for every
// tile, WHERE in the walk array its ambient-occlusion packets have to walk.  An any-hit ray starts at one of the tile's hit
// points (+ normal * 1e-5) and is at most AO_MAX_DISTANCE long: the reference's slab test (src/intersect_kernel.cl:21-61:
// t_near < max_distance, t_far > 0, t_near <= t_far) only passes for a box that holds a point o + t d with 0 <= t <=
// max_distance, a point of the SEGMENT the ray covers.  So a leaf whose box stays clear of the box around the segments of
// a set of rays (grown by a margin for the roundings: 1 % of the distance + 2^-20 of the coordinates) is tested by none of
// them, in any tree.  The walk array is the tree in pre-order with skip offsets, so a walk can start at ANY record and
// stop at any other: it visits what lies between in the usual way.  An interval [begin, end) for a region: from the root
// down, `begin` moves to the first child that meets the region whenever that child follows clear ones or is the only one
// that meets it (its parent's test and the clear subtrees are skipped), `end` moves to the end of the last child that
// meets it.  Where exactly one child meets the region at every level this is the deepest subtree that holds everything
// reachable; below that it trims both flanks.
// Per tile, 1 + ao_dirs intervals: [0] for the tile's hit points grown by the distance on every side -- any ray from the
// tile: packets of tiles that are not full (several table directions in one packet) and the RANDOM mode --, [1 + k] for
// the 64 rays of table direction k, a full tile's packet: the box around 64 segments, a third of the other's volume or
// less.  One wave per tile: lanes = hits while the origins and tangent frames (reference :225-236) go to LDS, then lanes
// = intervals, each going down the tree on its own; speed is nobody's concern here.
// Node tests per packet against walking the whole array (tools/analysis/ao_packets.cc, rows TRIM and PDIR): the bunny's
// plane 7.4 -> 5.8, its model tiles 104.8 -> 91.2 (AO_MAX_DISTANCE is a fifth of the model), the interior scene 38.2 -> 21.2.
// ---------------------------------------------------------------------------
constexpr uint32_t ENTRY_WAVES = 4;
__global__ __launch_bounds__(64 * ENTRY_WAVES) void entry_kernel(const NodeRec *__restrict__ walk, const HitRec *__restrict__ hits,
                                                                 const uint32_t *__restrict__ tile_hits, const uint32_t *__restrict__ tile_base,
                                                                 const float4 *__restrict__ ao_table, uint2 *__restrict__ tile_entry,
                                                                 uint32_t tiles, uint32_t stride, int32_t uniform_table,
                                                                 float max_distance) {
	__shared__ float origin[ENTRY_WAVES][3][64];
	__shared__ float frame[ENTRY_WAVES][9][64];  // basis_x, basis_y (the normal), basis_z
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const uint32_t tile = blockIdx.x * ENTRY_WAVES + wave;
	if (tile >= tiles)
		return;  // (a whole wave: the waves of a workgroup never meet at a barrier)
	const uint32_t hit_count = tile_hits[tile] & 0xFFu;
	uint2 *const out = tile_entry + (size_t) tile * stride;
	const uint32_t whole = walk[0].skip / (uint32_t) sizeof(NodeRec);
	if (lane < hit_count) {
		const HitRec rec = hits[(size_t) tile_base[tile] + lane];
		const float nx = rec.nx, ny = rec.ny, nz = rec.nz;
		const float eps = 1.0f / 100000.0f;
		origin[wave][0][lane] = rec.ox + nx * eps;
		origin[wave][1][lane] = rec.oy + ny * eps;
		origin[wave][2][lane] = rec.oz + nz * eps;
		float hx = nx, hy = ny, hz = nz;
		const float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
		if (ax <= ay && ax <= az)
			hx = 1.0f;
		else if (ay <= ax && ay <= az)
			hy = 1.0f;
		else if (az <= ax && az <= ay)
			hz = 1.0f;
		float bxx = hy * nz - hz * ny, bxy = hz * nx - hx * nz, bxz = hx * ny - hy * nx;
		normalize3(bxx, bxy, bxz);
		float bzx = bxy * nz - bxz * ny, bzy = bxz * nx - bxx * nz, bzz = bxx * ny - bxy * nx;
		normalize3(bzx, bzy, bzz);
		frame[wave][0][lane] = bxx; frame[wave][1][lane] = bxy; frame[wave][2][lane] = bxz;
		frame[wave][3][lane] = nx;  frame[wave][4][lane] = ny;  frame[wave][5][lane] = nz;
		frame[wave][6][lane] = bzx; frame[wave][7][lane] = bzy; frame[wave][8][lane] = bzz;
	}
	wave_lds_sync();
	const float inf = __builtin_inff();
	for (uint32_t j = lane; j < stride; j += 64u) {
		// ---- the region of interval j ----
		float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
		bool odd = !(max_distance > 0.0f) || hit_count == 0u;  // nothing can be said: the whole array it is
		const bool one_direction = j != 0u && uniform_table != 0 && hit_count == 64u;
		float4 table = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (one_direction)
			table = ao_table[j - 1u];
		for (uint32_t h = 0u; h < hit_count; ++h) {
			const float o[3] = { origin[wave][0][h], origin[wave][1][h], origin[wave][2][h] };
			for (int k = 0; k < 3; ++k) {
				float a = o[k], b = o[k];
				float margin = fabsf(o[k]) * 0x1.0p-20f + 1.0e-30f;
				if (one_direction) {
					// ray_dir = basis_x * xs + basis_y * ys + basis_z * zs (reference :246), its length 1 up to roundings
					const float d = (frame[wave][k][h] * table.x + frame[wave][3 + k][h] * table.y) + frame[wave][6 + k][h] * table.z;
					b = o[k] + d * (max_distance * 1.01f);
					margin += max_distance * 0.01f;
				} else {
					a = o[k] - max_distance * 1.01f;
					b = o[k] + max_distance * 1.01f;
				}
				odd = odd || !(fabsf(a) < inf) || !(fabsf(b) < inf);
				lo[k] = fminf(lo[k], fminf(a, b) - margin);
				hi[k] = fmaxf(hi[k], fmaxf(a, b) + margin);
			}
		}
		// ---- its interval (node indices; the walk array keeps byte offsets) ----
		const auto skip_of = [&](uint32_t n) { return walk[n].skip / (uint32_t) sizeof(NodeRec); };
		const auto meets = [&](uint32_t n) {
			const NodeRec box = walk[n];
			return !(box.lo[0] > hi[0] || box.hi[0] < lo[0] || box.lo[1] > hi[1] || box.hi[1] < lo[1] || box.lo[2] > hi[2] || box.hi[2] < lo[2]);
		};
		uint32_t begin = 0u, end = whole ? whole : 1u;
		if (!odd && (j == 0u || one_direction)) {
			uint32_t n = 0u;
			while (skip_of(n) > 1u) {  // left
				uint32_t first = 0u, index = 0u, others = 0u;
				for (uint32_t c = n + 1u; c < n + skip_of(n); c += skip_of(c)) {
					if (first)
						others += meets(c) ? 1u : 0u;
					else {
						++index;
						if (meets(c))
							first = c;
					}
				}
				if (first == 0u) {  // (no child meets it: nothing under this node can be reached)
					n += skip_of(n);
					break;
				}
				if (index == 1u && others != 0u)
					break;
				n = first;
			}
			begin = n;
			n = 0u;
			while (skip_of(n) > 1u) {  // right
				uint32_t last = 0u;
				for (uint32_t c = n + 1u; c < n + skip_of(n); c += skip_of(c))
					if (meets(c))
						last = c;
				if (last == 0u) {
					end = n;
					break;
				}
				end = last + skip_of(last);
				n = last;
			}
			if (begin > end)
				begin = end;
		}
		// (a tile that is not full never looks at its per-direction intervals: they are filled with interval 0's rule
		// all the same -- the whole array here, harmless -- so that every word of the table is defined)
		out[j] = make_uint2(begin * (uint32_t) sizeof(NodeRec), end * (uint32_t) sizeof(NodeRec));
	}
}

// Pass 3, the frame's last kernel: value *= 1 - hits / n (reference :256 and :305-307) for the sub-pixels that wait
// for it, and the supersample box filter + 8-bit quantisation (reference src/ray_tracer.cc:3-16) in the same sweep
// over the float image.  One thread per OUTPUT pixel of this rank's bands: it visits its n x n sub-pixels in the
// reference's order (ssY-major, ssX-minor), replaces every pending tag (primary_tile) by
// value * (1 - occluded / n_dirs) -- value and count from the tile's slot of the hit list --, WRITES THAT BACK (the float
// image is what `download` hands out, reference src/opencl_host.cc:150-153) and sums.  `out` may be null (a frame
// without the device resize).  (The frame's occlusion TOTAL is no business of the frame: until round 4 this kernel summed
// it -- per lane, wave, workgroup, then one atomic per workgroup on ONE address: 8 640 of them at 1080p, which took
// longer than the rest of the kernel, 48 us -> 12 us without, 0.29 -> 0.07 ms at 4K.  The counts stay in the hit list
// until the host's next frame, and whoever asks for the statistic has them summed then: occluded_sum_kernel.)
// Band layout as in resize_kernel below.
__global__ __launch_bounds__(256) void finish_kernel(float *__restrict__ image, const HitRec *__restrict__ hits,
                                                     const uint32_t *__restrict__ occluded_of,
                                                     const uint32_t *__restrict__ tile_base, unsigned char *__restrict__ out,
                                                     uint32_t width, uint32_t height, uint32_t total_width, uint32_t n,
                                                     uint32_t tiles_x, Partition part, uint32_t rows_per_band, uint32_t ao_divisor) {
	const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t j = blockIdx.y;
	const uint32_t band_local = j / rows_per_band;
	const uint32_t y = (band_local * part.nranks + part.rank) * rows_per_band + (j - band_local * rows_per_band);
	if (x >= width)
		return;
	const float divisor = (float) ao_divisor;
	float total = 0.0f;
	for (uint32_t sy = 0; sy < n; ++sy) {
		const uint32_t row_index = j * n + sy;
		float *row = image + (size_t) row_index * total_width + (size_t) x * n;
		for (uint32_t sx = 0; sx < n; ++sx) {
			float v = row[sx];
			const uint32_t bits = __float_as_uint(v);
			if (is_pending(bits)) {
				const uint32_t column = x * n + sx;
				const size_t slot = (size_t) tile_base[(size_t) (row_index / TILE_H) * tiles_x + column / TILE_W] + (bits & 63u);
				v = hits[slot].value * (1.0f - ((float) occluded_of[slot] / divisor));
				row[sx] = v;
			}
			total += v;
		}
	}
	if (out)
		out[(size_t) j * width + x] = y < height ? (unsigned char) ((total / (float) (n * n)) * 255.0f) : (unsigned char) 0;
}

// The same pass for supersampled frames (n >= 2).  With one thread per output pixel a wave's 64 lanes read 64 places
// 4 n bytes apart -- and, through the tags, 64 tiles' parts of the hit list -- with every load: at `-s 64` (n = 8: a pixel is a
// whole tile) that is 64 cache lines per instruction and the pass runs at the L1's line rate, 3.85 ms for 3.8 GB
// (profiles/r04_notes.md, section 13).  Here a workgroup takes `pixels_per_block` neighbouring output pixels of one row:
// its threads sweep the n sub-pixel rows ALONG the rows (a wave reads 256 contiguous bytes of the image and the slots of
// eight neighbouring tiles), replace the tags as above, write back, and leave the values in LDS; then one thread per
// output pixel adds its n x n values up in the reference's order (ssY-major, ssX-minor: src/ray_tracer.cc:7-13) -- the
// same additions in the same order as finish_kernel's, so the same bits.  The cells of a pixel are n * n | 1 floats
// apart (odd: the adding threads do not meet in a bank).
// RESOLVE = false: the box filter alone, of an image that holds no tags any more (a resize on its own: launch_resize).
constexpr uint32_t FINISH_CELL_FLOATS = 4160u;  // 64 pixels of 8 x 8 sub-pixels and their padding
template <bool RESOLVE>
__global__ __launch_bounds__(256) void finish_wide_kernel(float *__restrict__ image, const HitRec *__restrict__ hits,
                                                          const uint32_t *__restrict__ occluded_of,
                                                          const uint32_t *__restrict__ tile_base, unsigned char *__restrict__ out,
                                                          uint32_t width, uint32_t height, uint32_t total_width, uint32_t n,
                                                          uint32_t tiles_x, Partition part, uint32_t rows_per_band, uint32_t ao_divisor,
                                                          uint32_t pixels_per_block) {
	__shared__ float cell[FINISH_CELL_FLOATS];
	const uint32_t x0 = blockIdx.x * pixels_per_block;
	const uint32_t pixels = width - x0 < pixels_per_block ? width - x0 : pixels_per_block;
	const uint32_t columns = pixels * n;
	const uint32_t stride = (n * n) | 1u;
	const uint32_t j = blockIdx.y;
	const uint32_t band_local = j / rows_per_band;
	const uint32_t y = (band_local * part.nranks + part.rank) * rows_per_band + (j - band_local * rows_per_band);
	const float divisor = (float) ao_divisor;
	for (uint32_t sy = 0; sy < n; ++sy) {
		const uint32_t row_index = j * n + sy;
		float *row = image + (size_t) row_index * total_width + (size_t) x0 * n;
		const uint32_t *bases = tile_base + (size_t) (row_index / TILE_H) * tiles_x;
		for (uint32_t c = threadIdx.x; c < columns; c += 256u) {
			float v = row[c];
			const uint32_t bits = __float_as_uint(v);
			if (RESOLVE && is_pending(bits)) {
				const size_t slot = (size_t) bases[(x0 * n + c) / TILE_W] + (bits & 63u);
				v = hits[slot].value * (1.0f - ((float) occluded_of[slot] / divisor));
				row[c] = v;
			}
			const uint32_t p = c / n;
			cell[p * stride + sy * n + (c - p * n)] = v;
		}
	}
	__syncthreads();
	if (out && threadIdx.x < pixels) {
		const float *cells = cell + threadIdx.x * stride;
		float total = 0.0f;
		for (uint32_t i = 0; i < n * n; ++i)
			total += cells[i];
		out[(size_t) j * width + x0 + threadIdx.x] = y < height ? (unsigned char) ((total / (float) (n * n)) * 255.0f) : (unsigned char) 0;
	}
}

// The occlusion counts of a frame, summed: RenderStats::ao_occluded, on demand (DeviceRenderer::stats) -- the counts are
// in the hit list from the end of the ambient-occlusion pass until the host's next primary pass clears them slot by slot.
// Grid-stride, per lane / wave / workgroup, one atomic per workgroup (at most 256) onto a total the launcher has zeroed.
__global__ __launch_bounds__(256) void occluded_sum_kernel(const uint32_t *__restrict__ occluded_of, size_t slots,
                                                           FrameCounters *__restrict__ counters) {
	__shared__ unsigned long long block_total;
	if (threadIdx.x == 0)
		block_total = 0ull;
	__syncthreads();
	unsigned long long mine = 0ull;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t) gridDim.x * blockDim.x)
		mine += occluded_of[i];
	for (int offset = 32; offset > 0; offset >>= 1)
		mine += (unsigned long long) __shfl_down((long long) mine, offset);
	if ((threadIdx.x & 63u) == 0u && mine)
		atomicAdd(&block_total, mine);
	__syncthreads();
	if (threadIdx.x == 0 && block_total)
		atomicAdd(&counters->occluded, block_total);
}

// Supersample box filter + 8-bit quantisation on the device: one thread per
// output pixel, ssY-major / ssX-minor float summation and truncating store,
// exactly reference src/ray_tracer.cc:3-16.  Works on this rank's bands only: both the
// float image and the 8-bit buffer hold them back to back, so local output row j is
// the box filter of local sub-pixel rows j*n .. j*n+n-1; it is global row
// (band_local * nranks + rank) * rows_per_band + j % rows_per_band, and rows past the
// image's height are written as 0.
__global__ __launch_bounds__(256) void resize_kernel(const float *__restrict__ tmp, unsigned char *__restrict__ out,
                                                     uint32_t width, uint32_t height, uint32_t total_width, uint32_t n,
                                                     Partition part, uint32_t rows_per_band) {
	const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t j = blockIdx.y;
	const uint32_t band_local = j / rows_per_band;
	const uint32_t y = (band_local * part.nranks + part.rank) * rows_per_band + (j - band_local * rows_per_band);
	if (x >= width)
		return;
	unsigned char q = 0;
	if (y < height) {
		float total = 0.0f;
		for (uint32_t sy = 0; sy < n; ++sy) {
			const float *row = tmp + (size_t) (j * n + sy) * total_width + (size_t) x * n;
			for (uint32_t sx = 0; sx < n; ++sx)
				total += row[sx];
		}
		q = (unsigned char) ((total / (float) (n * n)) * 255.0f);
	}
	out[(size_t) j * width + x] = q;
}

// ---- host-callable launchers (keeps the launch syntax inside this TU) ----
// Makes the runtime load this library's code object for the current device now (it is otherwise loaded at the first
// launch): called from a warm-up thread while the CPU still builds the scene.
void preload_kernels() {
	hipFuncAttributes attr;
	(void) hipFuncGetAttributes(&attr, (const void *) primary_kernel<true>);
	(void) hipFuncGetAttributes(&attr, (const void *) ao_kernel<AO_UNIFORM, true, true>);
	(void) hipGetLastError();
}

#if defined(OCRT_STAMPS) || defined(OCRT_TAIL)
// (instrumented builds only: the debug stamps are sums and need zeroing; the frame's own counters do not -- device_types.h)
__global__ __launch_bounds__(256) void clear_stamps_kernel(FrameCounters *counters) {
	for (uint32_t i = threadIdx.x; i < sizeof(counters->stamp) / sizeof(counters->stamp[0]); i += blockDim.x)
		counters->stamp[i] = 0ull;
}
#endif

// The frame is three kernels (two without ambient occlusion): primary pass (+ ordering step in its tail), the
// ambient-occlusion pass, the finishing kernel (AO factor + box filter + quantisation).
void launch_primary(const SceneBuffers &scene, float *image, void *hits, void *occluded_of, void *tile_hits, void *order,
                    const void *tile_base, void *counters, const KernelParams &P, void *stream) {
	hipStream_t s = (hipStream_t) stream;
#if defined(OCRT_STAMPS) || defined(OCRT_TAIL)
	hipLaunchKernelGGL(clear_stamps_kernel, dim3(1), dim3(256), 0, s, (FrameCounters *) counters);
#endif
	if (P.tiles_x * P.local_tile_rows == 0)
		return;
	const uint32_t strips = (P.tiles_x + P.strip_tiles - 1u) / P.strip_tiles, row_blocks = (P.local_tile_rows + PRIMARY_ROWS - 1u) / PRIMARY_ROWS;
	const uint32_t blocks = XCD_GROUPS * ((strips + XCD_GROUPS - 1u) >> 3) * row_blocks * (P.strip_tiles >> 1);
	auto launch = [&](auto kernel) {
		PrimaryArgs args;
		args.walk_ptr = (const float4 *) scene.walk;
		args.tris_ptr = (const float4 *) scene.tris;
		args.nodes_ptr = (const float4 *) scene.nodes;
		args.shade = (const float4 *) scene.shade;
		args.image = image;
		args.hits = (HitRec *) hits;
		args.occluded_of = (uint32_t *) occluded_of;
		args.tile_hits = (uint32_t *) tile_hits;
		args.order = (uint32_t *) order;
		args.tile_base = (const uint32_t *) tile_base;
		args.counters = (FrameCounters *) counters;
		args.P = P;
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * PRIMARY_WAVES), 0, s, args);
	};
#ifdef OCRT_DEBUG_KNOBS
	if (!P.shared_walk)
		return launch(primary_kernel<false>);
#endif
	launch(primary_kernel<true>);
}

// Fills `tile_entry` (1 + ao_dirs intervals of two words per tile) for the frame whose hit list is in `hits`: once per
// upload (entry_kernel).
void launch_entries(const SceneBuffers &scene, const void *hits, const void *tile_hits, const void *tile_base, void *tile_entry,
                    const KernelParams &P, void *stream) {
	const uint32_t tiles = P.tiles_x * P.local_tile_rows;
	if (tiles == 0)
		return;
	hipLaunchKernelGGL(entry_kernel, dim3((tiles + ENTRY_WAVES - 1u) / ENTRY_WAVES), dim3(64 * ENTRY_WAVES), 0, (hipStream_t) stream,
	                   (const NodeRec *) scene.walk, (const HitRec *) hits, (const uint32_t *) tile_hits, (const uint32_t *) tile_base,
	                   (const float4 *) scene.ao_table, (uint2 *) tile_entry, tiles, P.entry_stride, P.ao_mode == AO_UNIFORM && scene.ao_table ? 1 : 0,
	                   P.ao_max_distance);
}

void launch_ao(const SceneBuffers &scene, void *hits, void *occluded_of, void *order, const void *tile_base, const void *tile_entry,
               void *counters, const KernelParams &params, uint32_t workgroups, bool prefetch, void *stream, void *event_before_ao,
               void *event_after_ao) {
	if (params.tiles_x * params.local_tile_rows == 0 || params.ao_mode == AO_NONE || params.ao_dirs == 0)
		return;
	hipStream_t s = (hipStream_t) stream;
	// persistent grid: what the chip holds, or one wave per (tile, direction) when the image is small
	const uint32_t tiles = params.tiles_x * params.local_tile_rows;
	const uint64_t units = (uint64_t) tiles * params.ao_dirs;
	uint32_t ao_blocks = workgroups;  // (DeviceRenderer::aoWorkgroups: 8 per CU for a host alone on its GPU)
	if ((units + AO_WAVES - 1) / AO_WAVES < ao_blocks)
		ao_blocks = (uint32_t) ((units + AO_WAVES - 1) / AO_WAVES);
	KernelParams P = params;
	const uint32_t waves_per_group = (ao_blocks * AO_WAVES + XCD_GROUPS - 1u) / XCD_GROUPS;
	P.ao_guide = P.ao_guide * (waves_per_group ? waves_per_group : 1u);
	P.ao_claim_div = waves_per_group ? waves_per_group : 1u;  // (the waves of a group, for ao_kernel's claim rule)
	auto launch = [&](auto kernel) {
		// (the events bracket the ao_kernel launch alone: its duration is the one the roofline is quoted for)
		if (event_before_ao)
			(void) hipEventRecord((hipEvent_t) event_before_ao, s);
		AoArgs args;
		args.walk_ptr = (const float4 *) scene.walk;
		args.tris_ptr = (const float4 *) scene.tris;
		args.nodes_ptr = (const float4 *) scene.nodes;
		args.ao_table = (const float4 *) scene.ao_table;
		args.hits = (const HitRec *) hits;
		args.occluded_of = (uint32_t *) occluded_of;
		args.order = (const uint32_t *) order;
		args.tile_base = (const uint32_t *) tile_base;
		args.tile_entry = (const uint2 *) tile_entry;
		args.counters = (FrameCounters *) counters;
		args.P = P;
		hipLaunchKernelGGL(kernel, dim3(ao_blocks), dim3(64 * AO_WAVES), 0, s, args);
		if (event_after_ao)
			(void) hipEventRecord((hipEvent_t) event_after_ao, s);
	};
#ifdef OCRT_DEBUG_KNOBS
	if (!P.shared_walk) {
		if (P.ao_mode == AO_UNIFORM)
			launch(ao_kernel<AO_UNIFORM, false>);
		else
			launch(ao_kernel<AO_RANDOM, false>);
	} else
#endif
	if (P.ao_mode == AO_UNIFORM) {
		if (prefetch)
			launch(ao_kernel<AO_UNIFORM, true, true>);
		else
			launch(ao_kernel<AO_UNIFORM, true, false>);
	} else
		launch(ao_kernel<AO_RANDOM, true>);
}

// `out`: this rank's 8-bit bands (local_out_rows x out_width), or null for a frame without the device resize.
void launch_finish(float *image, const void *hits, const void *occluded_of, const void *tile_base, unsigned char *out,
                   const KernelParams &P, uint32_t out_width, uint32_t n, uint32_t local_out_rows, void *stream) {
	if (local_out_rows == 0 || out_width == 0 || n == 0)
		return;
	const bool has_ao = P.ao_mode != AO_NONE && P.ao_dirs > 0;
	if (!has_ao && !out)
		return;  // (nothing pending, nothing to filter)
	const uint32_t rows_per_band = P.part.band_tile_rows * TILE_H / n;
	const uint32_t cells_per_pixel = (n * n) | 1u;
	if (n >= 2u && cells_per_pixel <= FINISH_CELL_FLOATS) {
		// supersampled: a workgroup per run of output pixels, swept along the sub-pixel rows (finish_wide_kernel)
		uint32_t pixels_per_block = FINISH_CELL_FLOATS / cells_per_pixel;
		pixels_per_block = pixels_per_block > 256u ? 256u : pixels_per_block > 16u ? pixels_per_block & ~15u : pixels_per_block;
		hipLaunchKernelGGL(finish_wide_kernel<true>, dim3((out_width + pixels_per_block - 1u) / pixels_per_block, local_out_rows), dim3(256), 0,
		                   (hipStream_t) stream, image, (const HitRec *) hits, (const uint32_t *) occluded_of, (const uint32_t *) tile_base,
		                   out, out_width, P.height / n, P.width, n, P.tiles_x, P.part, rows_per_band,
		                   P.ao_divisor ? P.ao_divisor : 1u, pixels_per_block);
		return;
	}
	hipLaunchKernelGGL(finish_kernel, dim3((out_width + 255) / 256, local_out_rows), dim3(256), 0, (hipStream_t) stream, image,
	                   (const HitRec *) hits, (const uint32_t *) occluded_of, (const uint32_t *) tile_base, out,
	                   out_width, P.height / n,
	                   P.width, n, P.tiles_x, P.part, rows_per_band, P.ao_divisor ? P.ao_divisor : 1u);
}

// counters->occluded = the sum of the hit list's `slots` occlusion counts (stream-ordered: after the frames enqueued so far).
void launch_occluded_sum(const void *occluded_of, size_t slots, void *counters, void *stream) {
	FrameCounters *const c = (FrameCounters *) counters;
	(void) hipMemsetAsync(&c->occluded, 0, sizeof c->occluded, (hipStream_t) stream);
	if (slots == 0)
		return;
	const size_t blocks = (slots + 4095) / 4096;  // (16 counts per thread at least)
	hipLaunchKernelGGL(occluded_sum_kernel, dim3((unsigned) (blocks < 256 ? blocks : 256)), dim3(256), 0, (hipStream_t) stream,
	                   (const uint32_t *) occluded_of, slots, c);
}

void launch_resize(const float *tmp, unsigned char *out, const KernelParams &P, uint32_t out_width, uint32_t n,
                   uint32_t local_out_rows, void *stream) {
	if (local_out_rows == 0 || out_width == 0 || n == 0)
		return;
	const uint32_t rows_per_band = P.part.band_tile_rows * TILE_H / n;
	const uint32_t cells_per_pixel = (n * n) | 1u;
	if (n >= 2u && cells_per_pixel <= FINISH_CELL_FLOATS) {
		// supersampled: along the sub-pixel rows, like the frame's own finishing sweep (no tags to resolve, nothing written back)
		uint32_t pixels_per_block = FINISH_CELL_FLOATS / cells_per_pixel;
		pixels_per_block = pixels_per_block > 256u ? 256u : pixels_per_block > 16u ? pixels_per_block & ~15u : pixels_per_block;
		hipLaunchKernelGGL(finish_wide_kernel<false>, dim3((out_width + pixels_per_block - 1u) / pixels_per_block, local_out_rows), dim3(256), 0,
		                   (hipStream_t) stream, const_cast<float *>(tmp), (const HitRec *) nullptr, (const uint32_t *) nullptr,
		                   (const uint32_t *) nullptr, out, out_width, P.height / n, P.width, n, P.tiles_x, P.part, rows_per_band, 1u,
		                   pixels_per_block);
		return;
	}
	hipLaunchKernelGGL(resize_kernel, dim3((out_width + 255) / 256, local_out_rows), dim3(256), 0,
	                   (hipStream_t) stream, tmp, out, out_width, P.height / n, P.width, n, P.part, rows_per_band);
}

}  // namespace ocrt
