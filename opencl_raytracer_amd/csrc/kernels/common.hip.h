// kernels/common.hip.h -- rays, small wave helpers, the triangle tests, buffer views
// (part of the one translation unit kernels.hip; see its head for the passes and the arithmetic contract)
#pragma once

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

// dot / cross / length / normalize of the reference kernel's float4 values (src/intersect_kernel.cl:65-127, 215-236,
// 284-304), which OpenCL leaves to the implementation (OpenCL 1.2 section 7.4).  The product fixes them to the IEEE
// definitions of SURVEY.md 8a-0.3 -- what the reference's own host code uses (include/vec3.h:30-36,73-75) and the only
// ones a CPU can reproduce.  The test-only build -DOCRT_OCML_BUILTINS calls ROCm's OWN library functions instead
// (oracle/ocl_builtins.cl: fused multiply-add chains in dot and cross, v * rsqrt(dot(v, v)) in normalize, a scaled
// square root in length), linked in as bitcode: that build must reproduce the reference kernel compiled against that
// library for this GPU bit for bit (tests/test_ocml_pin.py) -- the HIP path checked against the reference itself.
#ifdef OCRT_OCML_BUILTINS
typedef float ocl_f4 __attribute__((ext_vector_type(4)));
extern "C" __device__ float ocl_dot(ocl_f4, ocl_f4);
extern "C" __device__ ocl_f4 ocl_cross(ocl_f4, ocl_f4);
extern "C" __device__ ocl_f4 ocl_normalize(ocl_f4);
extern "C" __device__ float ocl_length(ocl_f4);
extern "C" __device__ float ocl_sin(float);
extern "C" __device__ float ocl_cos(float);
extern "C" __device__ float ocl_cospi(float);
extern "C" __device__ float ocl_sinpi(float);
extern "C" __device__ float ocl_acos(float);
#define OCRT_SIN ocl_sin
#define OCRT_COS ocl_cos
#define OCRT_COSPI ocl_cospi
#define OCRT_SINPI ocl_sinpi
#define OCRT_ACOS ocl_acos
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
	return ocl_dot(ocl_f4{ ax, ay, az, 0.0f }, ocl_f4{ bx, by, bz, 0.0f });
}
__device__ __forceinline__ void cross3(float ax, float ay, float az, float bx, float by, float bz, float &cx, float &cy, float &cz) {
	const ocl_f4 c = ocl_cross(ocl_f4{ ax, ay, az, 0.0f }, ocl_f4{ bx, by, bz, 0.0f });
	cx = c.x; cy = c.y; cz = c.z;
}
__device__ __forceinline__ float length3(float x, float y, float z) { return ocl_length(ocl_f4{ x, y, z, 0.0f }); }
#else
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
	return (ax * bx + ay * by) + az * bz;
}
__device__ __forceinline__ void cross3(float ax, float ay, float az, float bx, float by, float bz, float &cx, float &cy, float &cz) {
	cx = ay * bz - az * by;
	cy = az * bx - ax * bz;
	cz = ax * by - ay * bx;
}
__device__ __forceinline__ float length3(float x, float y, float z) { return sqrtf(dot3(x, y, z, x, y, z)); }
#define OCRT_SIN sinf
#define OCRT_COS cosf
#define OCRT_COSPI cospif
#define OCRT_SINPI sinpif
#define OCRT_ACOS acosf
#endif

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
#ifdef OCRT_OCML_BUILTINS
	const ocl_f4 n = ocl_normalize(ocl_f4{ x, y, z, 0.0f });
	x = n.x; y = n.y; z = n.z;
#else
	const float l = sqrtf(dot3(x, y, z, x, y, z));
	x = x / l;
	y = y / l;
	z = z / l;
#endif
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
		out.distance = length3(ex, ey, ez);
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

// The closest-hit walk's form: a candidate is accepted or rejected as in tri_any_hit (the reference's decision, by products),
// and what is kept of an accepted one is the reference's distance alone (src/intersect_kernel.cl:104-105) -- the
// barycentrics are wanted for ONE triangle per ray, the nearest, and the walk's epilogue runs the reference's test on that
// one (primary_tile): two divisions per ray instead of two per candidate.
struct Candidate {
	bool accepted;
	float distance;
};
__device__ __forceinline__ Candidate tri_candidate(const float4 q0, const float4 q1, const float4 q2, const float4 q3, float inv_d,
                                                   const Ray &r) {
	const float tax = q0.x, tay = q0.y, taz = q0.z;
	const float ux = q0.w, uy = q1.x, uz = q1.y;
	const float vx = q1.z, vy = q1.w, vz = q2.x;
	const float nx = q2.y, ny = q2.z, nz = q2.w;
	const float uu = q3.x, uv = q3.y, vv = q3.z, D = q3.w;
	Candidate out = { false, 0.0f };
	const float a = -dot3(nx, ny, nz, r.ox - tax, r.oy - tay, r.oz - taz);
	const float b = dot3(nx, ny, nz, r.dx, r.dy, r.dz);
	const float rr = a / b;
	const bool reject = (fabsf(b) < 0.000001f) | (rr < 0.0f);
	if (wave_ballot(!reject) == 0ull)
		return out;
	const float ipx = r.ox + rr * r.dx, ipy = r.oy + rr * r.dy, ipz = r.oz + rr * r.dz;
	const float wx = ipx - tax, wy = ipy - tay, wz = ipz - taz;
	const float wu = dot3(ux, uy, uz, wx, wy, wz);
	const float wv = dot3(wx, wy, wz, vx, vy, vz);
	const float X = uv * wv - vv * wu, Y = uv * wu - uu * wv;  // the numerators of s and t
	const unsigned int zone = tri_zone(X, Y, inv_d);
	bool accepted = zone == 1u;
	if (wave_ballot(!reject & (zone == 2u)) != 0ull)
		accepted = zone == 2u ? tri_accepts_exact(X, Y, D) : accepted;
	out.accepted = accepted & !reject;
	if (wave_ballot(out.accepted) == 0ull)
		return out;
	out.distance = length3(ipx - r.ox, ipy - r.oy, ipz - r.oz);
	return out;
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

__device__ __forceinline__ SceneViews make_views(const float4 *nodes_ptr, const float4 *tris_ptr, uint32_t node_count, uint32_t tri_count) {
	// descriptors are built from kernel arguments only, so they live in SGPRs
	SceneViews scene;
	scene.nodes = __builtin_amdgcn_make_buffer_rsrc((void *) nodes_ptr, 0, (int) (node_count * 32u), 0x00020000);
	scene.tris = __builtin_amdgcn_make_buffer_rsrc((void *) tris_ptr, 0, (int) (tri_count * LEAF_BYTES), 0x00020000);
	return scene;
}
__device__ __forceinline__ SceneViews make_views(const float4 *nodes_ptr, const float4 *tris_ptr, const KernelParams &P) {
	return make_views(nodes_ptr, tris_ptr, P.node_count, P.tri_count);
}

}  // namespace ocrt
