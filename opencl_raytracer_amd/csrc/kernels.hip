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
// Work decomposition: two passes per frame.
//   primary_kernel  one 64-lane wavefront per 8x8 tile of sub-pixels: closest-hit
//                   primary rays, smooth normal and head-light term.  Sub-pixels
//                   that need no ambient occlusion are final; every other hit is
//                   appended (ballot-compacted per wave, one atomic per wave) to a
//                   hit list in HBM.
//   ao_kernel       persistent: 4 workgroups of 8 waves per CU copy the first
//                   levels of the BVH into LDS once; then every wave, on its own,
//                   claims batches of (64 consecutive hits) x (~7 directions)
//                   from its XCD group's queue -- stealing from the other groups
//                   once it is empty -- rebuilds the 64 tangent frames in its LDS
//                   slice and drains the batch's any-hit rays (direction-major:
//                   the lanes of a wave cast one table direction from neighbouring
//                   surface points); occlusion counts are LDS atomics, flushed to
//                   a per-hit counter in HBM when the batch is done.
//   resolve_kernel  one thread per hit: value * (1 - occluded / n) -> image.
// The split exists for load balance: cost per tile varies 30x (background vs
// model, 29 rays per hit sub-pixel), and with fused tiles the frame ended on a
// tail of half-empty CUs.  After compaction every batch is full and small
// (about 8 rays per lane), and no wave idles before the frame's last batches.
// The hit list is segmented by XCD group so that each XCD's L2 sees the same
// part of the scene in both passes.
//
// What bounds it (profiles/r01_notes.md): the scene (12 MB) lives in L2 and HBM
// traffic is negligible.  Every node visit is a dependent 32-byte gather; the
// vector L1 looks up about one cache line per clock per CU and divergent lanes
// each need their own line, so the walk is bound by L1 gather rate and latency.
// Hence: the top of the tree in LDS (3x lower latency, separate bandwidth), as
// few VALU instructions per visit as the arithmetic contract allows (a
// conservative packed-FMA box test on enlarged boxes for the walk, the exact
// test only where the reference's result depends on it: at the leaves), and a
// wave-level scheduler that keeps the lanes busy.
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace ocrt {

namespace {

struct Ray {
	float ox, oy, oz;
	float dx, dy, dz;
	float ix, iy, iz;  // 1.0f / d, hoisted out of the per-node slab test
	float cx, cy, cz;  // -(o * inv): lets the conservative walk test be one fma per plane
};

struct Hit {
	float distance;
	uint32_t leaf;
	float s, t;
	float px, py, pz;
};

constexpr uint32_t NONE = 0xFFFFFFFFu;

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
	return (ax * bx + ay * by) + az * bz;
}

__device__ __forceinline__ void normalize3(float &x, float &y, float &z) {
	const float l = sqrtf(dot3(x, y, z, x, y, z));
	x = x / l;
	y = y / l;
	z = z / l;
}

__device__ __forceinline__ Ray make_ray(float ox, float oy, float oz, float dx, float dy, float dz) {
	Ray r;
	r.ox = ox; r.oy = oy; r.oz = oz;
	r.dx = dx; r.dy = dy; r.dz = dz;
	r.ix = 1.0f / dx;
	r.iy = 1.0f / dy;
	r.iz = 1.0f / dz;
	r.cx = -(ox * r.ix);
	r.cy = -(oy * r.iy);
	r.cz = -(oz * r.iz);
	return r;
}

// ---------------------------------------------------------------------------
// Slab tests
// ---------------------------------------------------------------------------

// Exact form, reference src/intersect_kernel.cl:21-61.  The reference's early
// returns only skip work; evaluating everything and combining the same
// comparisons (kept in their original `a > b` polarity for NaN) is identical.
__device__ __forceinline__ bool slab_hit_exact(const float4 lo, const float4 hi, const Ray &r, float max_distance) {
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

// Largest magnitude for which (b - o) cannot overflow.  A ray is "regular" when
// its origin is within it and its reciprocal direction is finite and non-zero;
// a box is regular when it is finite, within the limit and lo <= hi (checked on
// the host, KernelParams::scene_regular).
constexpr float REGULAR_LIMIT = 1.0e37f;

__device__ __forceinline__ bool ray_is_regular(const Ray &r) {
	return fabsf(r.ox) <= REGULAR_LIMIT && fabsf(r.oy) <= REGULAR_LIMIT && fabsf(r.oz) <= REGULAR_LIMIT &&
	       fabsf(r.ix) <= REGULAR_LIMIT && fabsf(r.iy) <= REGULAR_LIMIT && fabsf(r.iz) <= REGULAR_LIMIT &&
	       r.ix != 0.0f && r.iy != 0.0f && r.iz != 0.0f;
}

// May this ray use the enlarged walk boxes?  Origin within the bound the margin
// was sized for, reciprocal direction in [2^-60, 2^100] (no overflow of o*inv,
// no underflow of margin*inv).
__device__ __forceinline__ bool ray_is_walkable(const Ray &r, float origin_limit) {
	const float inv_lo = 8.6736174e-19f, inv_hi = 1.2676506e30f;
	return fabsf(r.ox) <= origin_limit && fabsf(r.oy) <= origin_limit && fabsf(r.oz) <= origin_limit &&
	       fabsf(r.ix) >= inv_lo && fabsf(r.ix) <= inv_hi && fabsf(r.iy) >= inv_lo && fabsf(r.iy) <= inv_hi &&
	       fabsf(r.iz) >= inv_lo && fabsf(r.iz) <= inv_hi;
}

// min/max form.  For a regular ray against a regular box no NaN can arise,
// (lo-o)*inv and (hi-o)*inv are ordered by the sign of inv (IEEE rounding is
// monotonic), and the reference's chain of early-outs reduces to
//   max(near) <= min(far)  &&  max(near) < max_distance  &&  min(far) > 0
// -- the same comparisons on the same values.  With below = pred(max_distance)
// (max_distance > 0) and tiny = the smallest positive float this is
//   max(near, tiny) <= min(far, below).
// The same monotonicity makes the test conservative under box enlargement:
// lo' <= lo, hi' >= hi can only widen [near, far].
__device__ __forceinline__ bool slab_hit_regular(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                                 const Ray &r, float below) {
	const float x0 = (lox - r.ox) * r.ix, x1 = (hix - r.ox) * r.ix;
	const float y0 = (loy - r.oy) * r.iy, y1 = (hiy - r.oy) * r.iy;
	const float z0 = (loz - r.oz) * r.iz, z1 = (hiz - r.oz) * r.iz;
	const float tiny = __uint_as_float(1u);
	const float t_near = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tiny));
	const float t_far = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), below));
	return t_near <= t_far;
}

// ---------------------------------------------------------------------------
// Scene access: 128-bit loads through buffer descriptors (wave-uniform base in
// SGPRs + 32-bit per-lane byte offset).  One instruction per 16 bytes, offsets
// past the end return 0 instead of faulting, and -- unlike a pointer load --
// hipcc cannot split a lane off and sink it behind the box test (it did, which
// cost a second dependent memory round trip per node).
// ---------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 load_u4(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_offset) {
	return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int) byte_offset, 0, 0);
}
__device__ __forceinline__ float4 load_f4(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_offset) {
	const u32x4 v = load_u4(rsrc, byte_offset);
	return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

struct SceneViews {
	__amdgpu_buffer_rsrc_t nodes;   // NodeRec[node_count]      exact boxes
	__amdgpu_buffer_rsrc_t wnodes;  // WalkNodeRec[node_count]  enlarged boxes
	__amdgpu_buffer_rsrc_t tris;    // TriRec[tri_count]
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Conservative slab test of the walk: t = fma(plane, inv, -(o*inv)) per plane
// (three packed FMAs for the six planes) on a box that the host enlarged by
// m = 2^-19 * S on every side, S >= every |box coordinate| and |ray origin|.
//
// Claim: whenever the exact test (slab_hit_regular on the exact box) passes,
// this one passes.  Per plane and for inv > 0 (inv < 0 mirrors), with u = 2^-24:
//   here   U = fl(lo'*inv + c),  c = fl(-o*inv),  lo' <= lo - m
//          U <= (lo - m - o)*inv + |o|*inv*u + |U|*u
//   exact  T = fl(fl(lo - o)*inv) >= (lo - o)*inv - |lo - o|*inv*(2u + u^2)
// so U <= T as soon as m >= u*(4|o| + 3|lo| + m) ~ 7uS, and m = 32uS.  The far
// plane is symmetric (U >= T).  Hence [near, far] here contains the exact
// interval and  max(near, tiny) <= min(far, below)  is implied.  Sub-normal
// results would break the relative-error model; |inv| >= 2^-60 keeps m*inv far
// above them.
__device__ __forceinline__ bool slab_hit_walk(const u32x4 a, const u32x4 b, const Ray &r, float below) {
	const f32x2 px = { __uint_as_float(a.x), __uint_as_float(a.y) };
	const f32x2 py = { __uint_as_float(a.z), __uint_as_float(a.w) };
	const f32x2 pz = { __uint_as_float(b.x), __uint_as_float(b.y) };
	const f32x2 tx = __builtin_elementwise_fma(px, (f32x2){ r.ix, r.ix }, (f32x2){ r.cx, r.cx });
	const f32x2 ty = __builtin_elementwise_fma(py, (f32x2){ r.iy, r.iy }, (f32x2){ r.cy, r.cy });
	const f32x2 tz = __builtin_elementwise_fma(pz, (f32x2){ r.iz, r.iz }, (f32x2){ r.cz, r.cz });
	const float tiny = __uint_as_float(1u);
	const float t_near = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), tiny));
	const float t_far = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), below));
	return t_near <= t_far;
}

// Cursor of a walking lane.
//   walk mode : i indexes the top-first walk array; entries below `top` are read
//               from LDS, the rest from global memory.  While inside a cut-off
//               body, `end` is the body's end and `ret` the top entry to resume at.
//   exact mode: i indexes the original pre-order NodeRec array (limit = node count).
// A lane is done when i >= limit and it is not inside a body.
struct Cursor {
	uint32_t i, end, ret, limit;
};
__device__ __forceinline__ bool cursor_alive(const Cursor &c) { return c.i < c.limit || c.end != NONE; }
__device__ __forceinline__ void cursor_finish(Cursor &c) {
	c.i = c.limit;
	c.end = NONE;
}

// One node for a lane in state T on the enlarged boxes: box hit -> first child
// (next entry; a hit leaf becomes pending; a hit portal enters its body); miss ->
// skip the subtree.
__device__ __forceinline__ void node_step_walk(const SceneViews &scene, const uint4 *__restrict__ top, uint32_t top_lds,
                                               const Ray &r, float below, Cursor &c, uint32_t &pending) {
	u32x4 a, b;
	if (c.i < top_lds) {
		const uint4 la = top[2u * c.i], lb = top[2u * c.i + 1u];
		a = (u32x4){ la.x, la.y, la.z, la.w };
		b = (u32x4){ lb.x, lb.y, lb.z, lb.w };
	} else {
		a = load_u4(scene.wnodes, c.i * 32u);
		b = load_u4(scene.wnodes, c.i * 32u + 16u);
	}
	const bool hit = slab_hit_walk(a, b, r, below);
	const uint32_t span = b.z, kind = b.w >> WALK_KIND_SHIFT, payload = b.w & WALK_PAYLOAD_MASK;
	const bool portal = kind == WALK_PORTAL;
	pending = (hit && kind == WALK_LEAF) ? payload : NONE;
	if (hit && portal) {
		c.ret = c.i + 1u;
		c.i = payload;
		c.end = payload + span;
	} else {
		c.i += (hit || portal) ? 1u : span;
	}
	if (c.i == c.end) {
		c.i = c.ret;
		c.end = NONE;
	}
}

// Same on the exact boxes with the reference's own test (irregular rays, scenes
// without a walk array).
__device__ __forceinline__ void node_step_exact(const SceneViews &scene, const Ray &r, float max_distance, Cursor &c,
                                                uint32_t &pending) {
	const float4 lo = load_f4(scene.nodes, c.i * 32u);
	const float4 hi = load_f4(scene.nodes, c.i * 32u + 16u);
	const bool hit = slab_hit_exact(lo, hi, r, max_distance);
	pending = hit ? __float_as_uint(hi.w) : NONE;  // inner nodes carry NONE
	c.i += hit ? 1u : __float_as_uint(lo.w);
}

// Advances the T lanes: WALK_STEPS nodes for the lanes that walk the enlarged
// boxes (the common case); when none of those is walking, one node for the lanes
// on the exact boxes.
template <int WALK_STEPS>
__device__ __forceinline__ void advance_walkers(const SceneViews &scene, const uint4 *__restrict__ top,
                                                uint32_t top_lds, const Ray &r, bool walkable, float max_distance,
                                                float below, Cursor &c, uint32_t &pending) {
	bool walking_lane = pending == NONE && cursor_alive(c);
	if (__ballot(walking_lane && walkable) != 0ull) {
#pragma unroll
		for (int step = 0; step < WALK_STEPS; ++step) {
			if (walking_lane && walkable)
				node_step_walk(scene, top, top_lds, r, below, c, pending);
			walking_lane = pending == NONE && cursor_alive(c);
		}
	} else if (walking_lane) {
		node_step_exact(scene, r, max_distance, c, pending);
	}
}

// Triangle test for a pending leaf, straight-line: the leaf's exact box gate
// (reference :189,:195 -- the leaf is a node of the walk) followed by the plane
// hit + parametric (s,t) test (reference :65-114) on the precomputed TriRec.
// The early returns of the reference become one accumulated predicate so that
// all L lanes stay converged.  `x > 1.00001` (double literal) == `x > 0x3F800053`.
struct TriResult {
	bool accepted;
	float s, t, distance;
	float px, py, pz;
};

template <bool CLOSEST>
__device__ __forceinline__ TriResult tri_test(const SceneViews &scene, uint32_t leaf, const Ray &r, bool regular,
                                              float below) {
	const uint32_t base = leaf * (uint32_t) sizeof(TriRec);
	const float4 q0 = load_f4(scene.tris, base), q1 = load_f4(scene.tris, base + 16u);
	const float4 q2 = load_f4(scene.tris, base + 32u), q3 = load_f4(scene.tris, base + 48u);
	const float4 b0 = load_f4(scene.tris, base + 64u), b1 = load_f4(scene.tris, base + 80u);
	// A lane on the walk array may have reached this leaf through its enlarged box:
	// apply the exact one.  A lane on the exact array already passed it.
	const bool box_ok = !regular || slab_hit_regular(b0.x, b0.y, b0.z, b1.x, b1.y, b1.z, r, below);
	const float tax = q0.x, tay = q0.y, taz = q0.z;
	const float ux = q0.w, uy = q1.x, uz = q1.y;
	const float vx = q1.z, vy = q1.w, vz = q2.x;
	const float nx = q2.y, ny = q2.z, nz = q2.w;
	const float uu = q3.x, uv = q3.y, vv = q3.z, D = q3.w;
	const float a = -dot3(nx, ny, nz, r.ox - tax, r.oy - tay, r.oz - taz);
	const float b = dot3(nx, ny, nz, r.dx, r.dy, r.dz);
	const float rr = a / b;
	const float ipx = r.ox + rr * r.dx, ipy = r.oy + rr * r.dy, ipz = r.oz + rr * r.dz;
	const float wx = ipx - tax, wy = ipy - tay, wz = ipz - taz;
	const float wu = dot3(ux, uy, uz, wx, wy, wz);
	const float wv = dot3(wx, wy, wz, vx, vy, vz);
	const float slack_hi = __uint_as_float(0x3F800053u);
	const float s = (uv * wv - vv * wu) / D;
	const float t = (uv * wu - uu * wv) / D;
	// reject: |b| < 1e-6, r < 0, s < -1e-5, s > 1.00001, t < -1e-5, s + t > 1.00001
	const bool reject = (fabsf(b) < 0.000001f) | (rr < 0.0f) | (s < -0.00001f) | (s > slack_hi) | (t < -0.00001f) |
	                    ((s + t) > slack_hi);
	TriResult out;
	out.accepted = box_ok & !reject;
	out.s = s;
	out.t = t;
	out.px = ipx; out.py = ipy; out.pz = ipz;
	out.distance = 0.0f;
	if (CLOSEST) {
		const float ex = ipx - r.ox, ey = ipy - r.oy, ez = ipz - r.oz;
		out.distance = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
	}
	return out;
}

// Maps a rank-local tile row to the global tile row under the band partition.
__device__ __forceinline__ uint32_t global_tile_row(const Partition &p, uint32_t local_row) {
	const uint32_t band_local = local_row / p.band_tile_rows;
	const uint32_t within = local_row - band_local * p.band_tile_rows;
	return (band_local * p.nranks + p.rank) * p.band_tile_rows + within;
}

}  // namespace

// Wave scheduler thresholds.  A lane is in state T (walking nodes), L (a hit
// leaf is pending its triangle test) or I (no ray).  Per iteration the wave
// runs, chosen with scalar ballots only, ONE straight-line predicated body:
// refill the I lanes from the ray queue, test the L lanes' triangles, or advance
// the T lanes.  This keeps the wave from paying for its longest ray and from
// running a 150-instruction triangle test for two lanes.
constexpr uint32_t REFILL_MIN = 16;  // refill once this many lanes are idle ...
constexpr uint32_t LEAF_MIN = 16;    // ... test triangles once this many leaves are pending

__device__ __forceinline__ SceneViews make_views(const float4 *nodes_ptr, const float4 *wnodes_ptr,
                                                 const float4 *tris_ptr, const KernelParams &P) {
	// descriptors are built from kernel arguments only, so they live in SGPRs
	SceneViews scene;
	scene.nodes = __builtin_amdgcn_make_buffer_rsrc((void *) nodes_ptr, 0, (int) (P.node_count * 32u), 0x00020000);
	scene.wnodes = __builtin_amdgcn_make_buffer_rsrc((void *) wnodes_ptr, 0, (int) (P.node_count * 32u), 0x00020000);
	scene.tris = __builtin_amdgcn_make_buffer_rsrc((void *) tris_ptr, 0,
	                                               (int) (P.tri_count * (uint32_t) sizeof(TriRec)), 0x00020000);
	return scene;
}

// ---------------------------------------------------------------------------
// Pass 1: primary rays.  Four independent waves per workgroup, one tile each.
// ---------------------------------------------------------------------------
constexpr uint32_t PRIMARY_WAVES = 4;

template <int WALK_STEPS>
__global__ __launch_bounds__(64 * PRIMARY_WAVES) void primary_kernel(
    const float4 *__restrict__ nodes_ptr, const float4 *__restrict__ wnodes_ptr, const float4 *__restrict__ tris_ptr,
    const float4 *__restrict__ shade, float *__restrict__ image, HitRec *__restrict__ hits,
    uint32_t *__restrict__ occluded_of, FrameCounters *__restrict__ counters,
    const uint32_t *__restrict__ group_offset, KernelParams P) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	const SceneViews scene = make_views(nodes_ptr, wnodes_ptr, tris_ptr, P);

	// Workgroup -> 2x2 tiles (16x16 sub-pixels).  Workgroups b and b+8 share an
	// XCD and its L2 (MI355X_MICROARCH.md, dispatch is round-robin over XCDs), so
	// the image is cut into vertical strips two tiles wide, strips are dealt
	// round-robin to the 8 XCD groups, and each group walks its strips top to
	// bottom: neighbouring workgroups of a group touch the same BVH region, while
	// every group still sees the whole image height.
	const uint32_t group = blockIdx.x & 7u, seq = blockIdx.x >> 3;
	const uint32_t strips = (P.tiles_x + 1u) >> 1;
	const uint32_t row_pairs = (P.local_tile_rows + 1u) >> 1;
	const uint32_t strips_here = (strips + 7u - group) >> 3;
	if (seq >= strips_here * row_pairs)
		return;
	const uint32_t strip_index = seq / row_pairs;
	const uint32_t row_pair = seq - strip_index * row_pairs;
	const uint32_t tile_x = 2u * (group + 8u * strip_index) + (wave & 1u);
	const uint32_t local_row = 2u * row_pair + (wave >> 1);
	if (tile_x >= P.tiles_x || local_row >= P.local_tile_rows)
		return;  // the waves of a workgroup never synchronise
	const uint32_t tile_y = global_tile_row(P.part, local_row);
	const uint32_t x = tile_x * TILE_W + (lane & 7u);
	const uint32_t y = tile_y * TILE_H + (lane >> 3);
	const bool active = x < P.width && y < P.height;
	const uint32_t count = P.node_count;
	const unsigned long long lanes_below = (1ull << lane) - 1ull;

	// reference src/intersect_kernel.cl:279-295
	float dx = ((float) x + 0.5f) / P.a - P.half_w;
	float dy = -(((float) y + 0.5f) / P.a - P.half_h);
	float dz = -1.0f;
	normalize3(dx, dy, dz);
	const Ray ray = make_ray(0.0f, 0.0f, 2.0f, dx, dy, dz);
	const bool walkable = P.walk_ok && ray_is_regular(ray) && ray_is_walkable(ray, P.origin_limit);
	Hit best;
	best.distance = __builtin_inff();
	best.leaf = 0;
	best.s = best.t = 0.0f;
	best.px = best.py = best.pz = 0.0f;
	bool hit = false;
	Cursor cur;
	cur.limit = walkable ? P.top_count : count;
	cur.i = active ? 0u : cur.limit;
	cur.end = NONE;
	cur.ret = 0u;
	uint32_t pending = NONE;
	for (;;) {
		const unsigned long long walking = __ballot(pending == NONE && cursor_alive(cur));
		const unsigned long long leaves = __ballot(pending != NONE);
		if (leaves != 0ull && ((uint32_t) __popcll(leaves) >= LEAF_MIN || walking == 0ull)) {
			if (pending != NONE) {
				const TriResult tr = tri_test<true>(scene, pending, ray, walkable, P.primary_below);
				// closest hit: strict '>' in ascending leaf order, reference :106-112
				if (tr.accepted) {
					hit = true;
					if (best.distance > tr.distance) {
						best.distance = tr.distance;
						best.leaf = pending;
						best.s = tr.s;
						best.t = tr.t;
						best.px = tr.px; best.py = tr.py; best.pz = tr.pz;
					}
				}
				pending = NONE;
			}
			continue;
		}
		if (walking == 0ull)
			break;
		advance_walkers<WALK_STEPS>(scene, nullptr, 0u, ray, walkable, 100000.0f, P.primary_below, cur, pending);
	}

	// smooth normal and head-light term, reference :296-304
	float value = 0.0f;
	float nx = 0.0f, ny = 0.0f, nz = 0.0f;
	if (hit) {
		const float4 n0 = shade[3 * (size_t) best.leaf + 0];
		const float4 n1 = shade[3 * (size_t) best.leaf + 1];
		const float4 n2 = shade[3 * (size_t) best.leaf + 2];
		const float b0 = 1.0f - best.s - best.t, b1 = best.s, b2 = best.t;
		nx = (n0.x * b0 + n1.x * b1) + n2.x * b2;
		ny = (n0.y * b0 + n1.y * b1) + n2.y * b2;
		nz = (n0.z * b0 + n1.z * b1) + n2.z * b2;
		normalize3(nx, ny, nz);
		value = 1.0f;
		if (P.shading)
			value = fminf(fmaxf(-dot3(nx, ny, nz, dx, dy, dz), 0.0f), 1.0f);
	}
	const bool want_ao = P.ao_mode == AO_UNIFORM && P.ao_dirs > 0;
	if (active && !(hit && want_ao))
		image[(size_t) y * P.width + x] = value;  // final already

	// append the hits of this wave to the hit list: one returning atomic per wave
	const unsigned long long hit_mask = __ballot(hit);
	const uint32_t hit_count = (uint32_t) __popcll(hit_mask);
	if (hit_count == 0u)
		return;
	uint32_t base = 0u;
	if (lane == 0u) {
		atomicAdd(&counters->primary_hits, hit_count);
		if (want_ao)
			base = group_offset[group] + atomicAdd(&counters->hit_count[group], hit_count);
	}
	base = (uint32_t) __shfl((int) base, 0);
	if (hit && want_ao) {
		HitRec rec;
		rec.ox = best.px; rec.oy = best.py; rec.oz = best.pz;
		rec.value = value;
		rec.nx = nx; rec.ny = ny; rec.nz = nz;
		rec.pixel = y * P.width + x;
		const uint32_t slot = base + (uint32_t) __popcll(hit_mask & lanes_below);
		hits[slot] = rec;
		occluded_of[slot] = 0u;
	}
}

// ---------------------------------------------------------------------------
// Pass 2: ambient occlusion.  Persistent workgroups; waves work independently.
// ---------------------------------------------------------------------------
constexpr uint32_t AO_WAVES = 8;
constexpr uint32_t AO_BLOCKS_PER_CU = 4;

// LDS slice of one wave: the tangent frames of the batch's hits (structure of
// arrays, lane-major: consecutive hits sit in consecutive banks).
struct WaveShared {
	float frame[12][64];  // origin xyz, basis_x xyz, basis_y xyz, basis_z xyz
	unsigned int occluded[64];
};
struct AoShared {
	uint4 top[2 * WALK_TOP_CAPACITY];  // first levels of the walk array
	WaveShared wave[AO_WAVES];
};
static_assert(sizeof(AoShared) <= 40960, "four workgroups must fit the 160 KB of a CU");

// Orders this wave's LDS writes before its later LDS reads.  A wave executes in
// lockstep and the LDS unit serves one wave's requests in order, so only the
// compiler must be kept from reordering across this point.
__device__ __forceinline__ void wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int WALK_STEPS>
__global__ __launch_bounds__(64 * AO_WAVES) __attribute__((amdgpu_waves_per_eu(4, 6))) void ao_kernel(
    const float4 *__restrict__ nodes_ptr, const float4 *__restrict__ wnodes_ptr, const float4 *__restrict__ tris_ptr,
    const float4 *__restrict__ ao_table, const HitRec *__restrict__ hits, uint32_t *__restrict__ occluded_of,
    FrameCounters *__restrict__ counters, const uint32_t *__restrict__ group_offset, KernelParams P) {
	__shared__ AoShared sh;
	const uint32_t lane = threadIdx.x & 63u;
	WaveShared &mine = sh.wave[threadIdx.x >> 6];
	const SceneViews scene = make_views(nodes_ptr, wnodes_ptr, tris_ptr, P);
	const uint32_t top_lds = P.walk_ok ? P.top_lds : 0u;
	for (uint32_t e = threadIdx.x; e < 2u * top_lds; e += 64u * AO_WAVES) {
		const u32x4 v = load_u4(scene.wnodes, e * 16u);
		sh.top[e] = make_uint4(v.x, v.y, v.z, v.w);
	}
	__syncthreads();  // the only workgroup-wide synchronisation: waves are independent from here on

	const uint32_t count = P.node_count;
	const unsigned long long lanes_below = (1ull << lane) - 1ull;
	// Workgroups b and b+8 share an XCD: start with that group's queue, then help the others.
	const uint32_t home = blockIdx.x & (XCD_GROUPS - 1u);
	for (uint32_t turn = 0; turn < XCD_GROUPS; ++turn) {
		const uint32_t group = (home + turn) & (XCD_GROUPS - 1u);
		const uint32_t group_hits = counters->hit_count[group];  // produced by the primary pass
		const uint32_t group_batches = ((group_hits + 63u) >> 6) * P.batches_per_hits;
		for (;;) {
			// ---- claim a batch: hits [first, first + n) of the group, directions [dir0, dir0 + n_dirs) ----
			uint32_t batch = 0u;
			if (lane == 0u)
				batch = atomicAdd(&counters->queue_head[group], 1u);
			batch = (uint32_t) __shfl((int) batch, 0);
			if (batch >= group_batches)
				break;
			const uint32_t hit_block = batch / P.batches_per_hits;
			const uint32_t dir0 = (batch - hit_block * P.batches_per_hits) * P.dirs_per_batch;
			const uint32_t n_dirs = P.ao_dirs - dir0 < P.dirs_per_batch ? P.ao_dirs - dir0 : P.dirs_per_batch;
			const uint32_t first = group_offset[group] + (hit_block << 6);
			const uint32_t n = group_hits - (hit_block << 6) < 64u ? group_hits - (hit_block << 6) : 64u;

			// ---- the batch's tangent frames -> this wave's LDS slice ----
			if (lane < n) {
				const float4 q0 = ((const float4 *) hits)[2 * (size_t) (first + lane)];
				const float4 q1 = ((const float4 *) hits)[2 * (size_t) (first + lane) + 1];
				const float nx = q1.x, ny = q1.y, nz = q1.z;
				// p = point + normal * (1.0f / 100000.0f), reference :215
				const float eps = 1.0f / 100000.0f;
				mine.frame[0][lane] = q0.x + nx * eps;
				mine.frame[1][lane] = q0.y + ny * eps;
				mine.frame[2][lane] = q0.z + nz * eps;
				// tangent frame (reference :224-236): smallest |component| of the normal replaced by 1
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
				mine.frame[3][lane] = bxx; mine.frame[4][lane] = bxy; mine.frame[5][lane] = bxz;
				mine.frame[6][lane] = nx;  mine.frame[7][lane] = ny;  mine.frame[8][lane] = nz;
				mine.frame[9][lane] = bzx; mine.frame[10][lane] = bzy; mine.frame[11][lane] = bzz;
			}
			mine.occluded[lane] = 0u;
			wave_lds_sync();

			// ---- the batch's n * n_dirs any-hit rays (reference :237-255), direction-major ----
			const uint32_t total = n * n_dirs;
			uint32_t next = 0u;  // wave-uniform queue head
			Ray ray;
			bool walkable = true;
			Cursor cur;
			cur.limit = P.top_count;
			cur.i = cur.limit;
			cur.end = NONE;
			cur.ret = 0u;
			uint32_t pending = NONE;
			uint32_t h = 0;
			for (;;) {
				const bool walking_lane = pending == NONE && cursor_alive(cur);
				const unsigned long long walking = __ballot(walking_lane);
				const unsigned long long leaves = __ballot(pending != NONE);
				const uint32_t n_leaves = (uint32_t) __popcll(leaves);
				const uint32_t idle = 64u - (uint32_t) __popcll(walking) - n_leaves;
				if (next < total && idle >= REFILL_MIN) {
					const bool idle_lane = !walking_lane && pending == NONE;
					const unsigned long long idle_mask = __ballot(idle_lane);
					const uint32_t item = next + (uint32_t) __popcll(idle_mask & lanes_below);
					if (idle_lane && item < total) {
						const uint32_t k = item / n;
						h = item - k * n;
						const float4 dir = ao_table[dir0 + k];
						// ray_dir = basis_x * xs + basis_y * ys + basis_z * zs, lane by lane
						const float rx = (mine.frame[3][h] * dir.x + mine.frame[6][h] * dir.y) + mine.frame[9][h] * dir.z;
						const float ry = (mine.frame[4][h] * dir.x + mine.frame[7][h] * dir.y) + mine.frame[10][h] * dir.z;
						const float rz = (mine.frame[5][h] * dir.x + mine.frame[8][h] * dir.y) + mine.frame[11][h] * dir.z;
						ray = make_ray(mine.frame[0][h], mine.frame[1][h], mine.frame[2][h], rx, ry, rz);
						walkable = P.walk_ok && P.ao_regular && ray_is_regular(ray) && ray_is_walkable(ray, P.origin_limit);
						cur.limit = walkable ? P.top_count : count;
						cur.i = 0u;
						cur.end = NONE;
					}
					next += idle;
					continue;
				}
				if (n_leaves != 0u && (n_leaves >= LEAF_MIN || walking == 0ull)) {
					if (pending != NONE) {
						const TriResult tr = tri_test<false>(scene, pending, ray, walkable, P.ao_below);
						if (tr.accepted) {
							atomicAdd(&mine.occluded[h], 1u);
							cursor_finish(cur);  // any-hit: the reference walks on but only uses the boolean (:251)
						}
						pending = NONE;
					}
					continue;
				}
				if (walking == 0ull)
					break;  // nothing walking, nothing pending, and then next >= total (idle == 64 would have refilled)
				advance_walkers<WALK_STEPS>(scene, sh.top, top_lds, ray, walkable, P.ao_max_distance, P.ao_below, cur,
				                            pending);
			}
			wave_lds_sync();
			// ---- flush this batch's occlusion counts ----
			if (lane < n) {
				const uint32_t occluded = mine.occluded[lane];
				if (occluded)
					atomicAdd(&occluded_of[first + lane], occluded);
			}
			wave_lds_sync();
		}
	}
}

// Pass 3: value *= 1 - hits / n (reference :256 and :305-307), one thread per hit-list slot.
__global__ __launch_bounds__(256) void resolve_kernel(const HitRec *__restrict__ hits,
                                                      const uint32_t *__restrict__ occluded_of,
                                                      FrameCounters *__restrict__ counters, float *__restrict__ image,
                                                      const uint32_t *__restrict__ group_offset, KernelParams P) {
	__shared__ unsigned int block_total;
	if (threadIdx.x == 0)
		block_total = 0u;
	__syncthreads();
	const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
	// which group's segment is this slot in, and is it occupied?
	uint32_t group = 0u;
	for (uint32_t g = 1; g < XCD_GROUPS; ++g)
		if (slot >= group_offset[g])
			group = g;
	const bool valid = slot < group_offset[XCD_GROUPS] && slot - group_offset[group] < counters->hit_count[group];
	if (valid) {
		const uint32_t occluded = occluded_of[slot];
		const HitRec rec = hits[slot];
		image[rec.pixel] = rec.value * (1.0f - ((float) occluded / (float) P.ao_dirs));
		atomicAdd(&block_total, occluded);
	}
	__syncthreads();
	if (threadIdx.x == 0 && block_total)
		atomicAdd(&counters->occluded, (unsigned long long) block_total);
}

// Supersample box filter + 8-bit quantisation on the device: one thread per
// output pixel, ssY-major / ssX-minor float summation and truncating store,
// exactly reference src/ray_tracer.cc:3-16.  Works on this rank's bands only:
// local output row j of the compact band buffer is global row
// (band_local * nranks + rank) * rows_per_band + j % rows_per_band.
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
			const float *row = tmp + (size_t) (y * n + sy) * total_width + (size_t) x * n;
			for (uint32_t sx = 0; sx < n; ++sx)
				total += row[sx];
		}
		q = (unsigned char) ((total / (float) (n * n)) * 255.0f);
	}
	out[(size_t) j * width + x] = q;
}

// ---- host-callable launchers (keeps the launch syntax inside this TU) ----
void launch_primary(const void *nodes, const void *wnodes, const void *tris, const void *shade, float *image,
                    void *hits, void *occluded_of, void *counters, const void *group_offset, const KernelParams &P,
                    void *stream) {
	if (P.tiles_x * P.local_tile_rows == 0)
		return;
	const uint32_t strips = (P.tiles_x + 1u) >> 1, row_pairs = (P.local_tile_rows + 1u) >> 1;
	const uint32_t blocks = 8u * ((strips + 7u) >> 3) * row_pairs;
#define OCRT_LAUNCH(K)                                                                                              \
	hipLaunchKernelGGL(primary_kernel<K>, dim3(blocks), dim3(64 * PRIMARY_WAVES), 0, (hipStream_t) stream,           \
	                   (const float4 *) nodes, (const float4 *) wnodes, (const float4 *) tris, (const float4 *) shade, \
	                   image, (HitRec *) hits, (uint32_t *) occluded_of, (FrameCounters *) counters,                 \
	                   (const uint32_t *) group_offset, P)
	switch (P.variant) {  // debug knob OCRT_KERNEL_VARIANT: walk steps per scheduling decision
	case 11: OCRT_LAUNCH(1); break;
	case 14: OCRT_LAUNCH(4); break;
	default: OCRT_LAUNCH(2); break;
	}
#undef OCRT_LAUNCH
}

void launch_ao(const void *nodes, const void *wnodes, const void *tris, const void *ao_table, float *image,
               const void *hits, void *occluded_of, void *counters, const void *group_offset, const KernelParams &P,
               uint32_t max_hits, uint32_t compute_units, void *stream) {
	if (max_hits == 0 || P.ao_mode != AO_UNIFORM || P.ao_dirs == 0)
		return;
	// persistent grid: what the chip holds, or fewer when the frame cannot have that many batches
	const unsigned long long max_batches = (unsigned long long) ((max_hits + 63u) / 64u) * P.batches_per_hits;
	uint32_t blocks = compute_units * AO_BLOCKS_PER_CU;
	if ((max_batches + AO_WAVES - 1) / AO_WAVES < blocks)
		blocks = (uint32_t) ((max_batches + AO_WAVES - 1) / AO_WAVES);
#define OCRT_LAUNCH(K)                                                                                            \
	hipLaunchKernelGGL(ao_kernel<K>, dim3(blocks), dim3(64 * AO_WAVES), 0, (hipStream_t) stream,                   \
	                   (const float4 *) nodes, (const float4 *) wnodes, (const float4 *) tris,                     \
	                   (const float4 *) ao_table, (const HitRec *) hits, (uint32_t *) occluded_of,                 \
	                   (FrameCounters *) counters, (const uint32_t *) group_offset, P)
	switch (P.variant) {
	case 11: OCRT_LAUNCH(1); break;
	case 14: OCRT_LAUNCH(4); break;
	default: OCRT_LAUNCH(2); break;
	}
#undef OCRT_LAUNCH
	hipLaunchKernelGGL(resolve_kernel, dim3((max_hits + 255) / 256), dim3(256), 0, (hipStream_t) stream,
	                   (const HitRec *) hits, (const uint32_t *) occluded_of, (FrameCounters *) counters, image,
	                   (const uint32_t *) group_offset, P);
}

void launch_resize(const float *tmp, unsigned char *out, const KernelParams &P, uint32_t out_width, uint32_t n,
                   uint32_t local_out_rows, void *stream) {
	if (local_out_rows == 0 || out_width == 0 || n == 0)
		return;
	const uint32_t rows_per_band = P.part.band_tile_rows * TILE_H / n;
	hipLaunchKernelGGL(resize_kernel, dim3((out_width + 255) / 256, local_out_rows), dim3(256), 0,
	                   (hipStream_t) stream, tmp, out, out_width, P.height / n, P.width, n, P.part, rows_per_band);
}

}  // namespace ocrt
