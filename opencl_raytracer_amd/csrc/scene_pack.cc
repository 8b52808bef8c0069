#include "scene_pack.h"

#include "tri_predicate.h"
#include "walk_tree.h"

#include <cmath>
#include <cstdlib>
#include <future>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>

namespace ocrt {

namespace {

// Debug knobs are environment variables that only the A/B build reads (make EXTRA_DEFS=-DOCRT_DEBUG_KNOBS): the
// product library never looks at the environment for them.  None changes results.
const char *debug_knob(const char *name) {
#ifdef OCRT_DEBUG_KNOBS
	return std::getenv(name);
#else
	(void) name;
	return nullptr;
#endif
}

}  // namespace

PackedScene pack_scene(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes,
                       const std::vector<Vec3f> &aabbs, const std::vector<Vec3f> &vertices,
                       const std::vector<Vec3f> &vnormals) {
	if (nodes.empty())
		throw std::invalid_argument("upload: empty node array");
	const size_t count = nodes.size();
	if (nodes[0] != count)
		throw std::invalid_argument("upload: nodes[0] must equal the node count");
	if (aabbs.size() != 2 * count)
		throw std::invalid_argument("upload: aabbs must hold a (min,max) pair per node");
	if (faces.size() % 3 != 0)
		throw std::invalid_argument("upload: faces must hold 3 vertex ids per triangle");
	if (vnormals.size() != vertices.size())
		throw std::invalid_argument("upload: one normal per vertex expected");
	const size_t tri_count = faces.size() / 3;
	// the kernels address nodes (32 B) and leaf records (96 B) with 32-bit byte offsets
	if (count >= (1u << 27) || tri_count >= (1u << 25))
		throw std::invalid_argument("upload: scene too large for 32-bit device offsets");
	for (uint32_t v : faces)
		if (v >= vertices.size())
			throw std::invalid_argument("upload: face references a vertex out of range");

	PackedScene out;
	out.nodes.resize(count);
	out.regular = true;
	size_t leaf = 0;
	for (size_t i = 0; i < count; ++i) {
		const uint32_t skip = nodes[i];
		if (skip == 0 || i + skip > count)
			throw std::invalid_argument("upload: subtree size runs past the node array");
		NodeRec &n = out.nodes[i];
		for (unsigned k = 0; k < 3; ++k) {
			n.lo[k] = aabbs[2 * i][k];
			n.hi[k] = aabbs[2 * i + 1][k];
			// negated comparisons so that NaN counts as irregular
			if (!(std::fabs(n.lo[k]) <= 1.0e37f) || !(std::fabs(n.hi[k]) <= 1.0e37f) || !(n.lo[k] <= n.hi[k]))
				out.regular = false;
		}
		n.skip = skip;
		n.leaf = 0xFFFFFFFFu;
		if (skip == 1) {
			if (leaf >= tri_count)
				throw std::invalid_argument("upload: more leaves than triangles");
			n.leaf = (uint32_t) leaf++;
		}
	}
	if (leaf != tri_count)
		throw std::invalid_argument("upload: leaf count differs from triangle count");
	// children of inner node i are i+1 and i+1+skip(i+1)
	out.binary_tree = true;
	out.nested = true;
	for (size_t i = 0; i < count; ++i) {
		const NodeRec &p = out.nodes[i];
		if (p.skip == 1)
			continue;
		const size_t first = i + 1, second = first + out.nodes[first].skip;
		// The reference finds a leaf's triangle by adding (subtree size + 1) / 2 for every subtree it
		// skips (src/intersect_kernel.cl:191): only a full binary tree gives that a meaning.
		if (second >= i + p.skip || second + out.nodes[second].skip != i + p.skip)
			throw std::invalid_argument("upload: the node array is not a binary tree");
		for (size_t c : { first, second })
			for (unsigned k = 0; k < 3; ++k)
				if (!(p.lo[k] <= out.nodes[c].lo[k]) || !(out.nodes[c].hi[k] <= p.hi[k]))
					out.nested = false;
	}

	// The walk may use any tree over the same leaves (walk_tree.h): take a binned-SAH one when it is cheaper
	// than what was uploaded.  (Damaged arrays keep their tree: the exact form of the walk follows it.)  The rebuild
	// is half of what an upload costs the CPU and independent of the triangle records packed below: it runs beside them.
	std::future<std::vector<NodeRec>> rebuilding;
	if (out.regular && out.nested && !debug_knob("OCRT_KEEP_TREE")) {
		double threshold = 0.5;
		if (const char *env = debug_knob("OCRT_CONTRACT"))  // area ratio above which an inner node is dropped
			threshold = std::atof(env);
		const std::vector<NodeRec> &uploaded = out.nodes;  // (not touched until the future is collected)
		rebuilding = std::async(std::launch::async, [&uploaded, threshold, tri_count]() -> std::vector<NodeRec> {
			std::vector<NodeRec> rebuilt = contract_walk_tree(rebuild_walk_tree(uploaded), threshold);
			if (rebuilt.empty() || !(tree_cost(rebuilt) < tree_cost(uploaded)))
				return {};
			// The flags above were computed on the uploaded array; the kernels walk this one, so it has to
			// earn them again: an inner node may have more than two children here (contraction), what the
			// shared walk needs is that sibling subtrees tile their parent's index range, every leaf index
			// appears once, and boxes are regular and nested.
			bool regular = true, nested = true, ranges = rebuilt.size() >= 1 && rebuilt[0].skip == rebuilt.size();
			std::vector<unsigned char> seen(tri_count, 0);
			size_t leaves = 0;
			for (size_t i = 0; ranges && i < rebuilt.size(); ++i) {
				const NodeRec &p = rebuilt[i];
				for (unsigned k = 0; k < 3; ++k)
					if (!(std::fabs(p.lo[k]) <= 1.0e37f) || !(std::fabs(p.hi[k]) <= 1.0e37f) || !(p.lo[k] <= p.hi[k]))
						regular = false;
				if (p.skip == 0 || i + p.skip > rebuilt.size()) {
					ranges = false;
				} else if (p.skip == 1) {
					if (p.leaf >= tri_count || seen[p.leaf]++)
						ranges = false;
					++leaves;
				} else {
					if (p.leaf != 0xFFFFFFFFu)
						ranges = false;
					size_t c = i + 1;
					while (ranges && c < i + p.skip) {
						const NodeRec &child = rebuilt[c];
						if (child.skip == 0 || c + child.skip > i + p.skip)
							ranges = false;
						for (unsigned k = 0; k < 3; ++k)
							if (!(p.lo[k] <= child.lo[k]) || !(child.hi[k] <= p.hi[k]))
								nested = false;
						c += child.skip;
					}
				}
			}
			if (ranges && leaves == tri_count && regular && nested)
				return rebuilt;
			return {};
		});
	}

#ifdef OCRT_OCML_BUILTINS
	// (test-only build, kernels/common.hip.h: the triangle records' cross and dot products as ROCm's OpenCL library
	// computes them on the device -- fused multiply-add chains over the four lanes of a float4 whose w is 0, see
	// oracle/ocl_builtins.cl -- reproduced here with std::fmaf, which is exact)
	const auto lib_dot = [](const Vec3f &a, const Vec3f &b) {
		return std::fmaf(0.0f, 0.0f, std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)));
	};
	const auto lib_cross = [](const Vec3f &a, const Vec3f &b) {
		Vec3f c;
		c.x = std::fmaf(a.y, b.z, b.y * (-a.z));
		c.y = std::fmaf(a.z, b.x, b.z * (-a.x));
		c.z = std::fmaf(a.x, b.y, b.x * (-a.y));
		return c;
	};
#else
	const auto lib_dot = [](const Vec3f &a, const Vec3f &b) { return a.dot(b); };
	const auto lib_cross = [](const Vec3f &a, const Vec3f &b) { return a.cross(b); };
#endif
	out.tris.resize(tri_count);
	out.shade.resize(tri_count);
	for (size_t t = 0; t < tri_count; ++t) {
		const uint32_t i0 = faces[3 * t], i1 = faces[3 * t + 1], i2 = faces[3 * t + 2];
		const Vec3f ta = vertices[i0];
		const Vec3f u = vertices[i1] - ta;
		const Vec3f v = vertices[i2] - ta;
		const Vec3f n = lib_cross(u, v);
		TriRec &r = out.tris[t];
		for (unsigned k = 0; k < 3; ++k) {
			r.ta[k] = ta[k];
			r.u[k] = u[k];
			r.v[k] = v[k];
			r.n[k] = n[k];
		}
		r.pad0 = 0.0f;
		r.uu = lib_dot(u, u);
		r.uv = lib_dot(u, v);
		r.vv = lib_dot(v, v);
		r.D = r.uv * r.uv - r.uu * r.vv;
		r.inv_d = tri_inverse_d(r.D);
		ShadeRec &s = out.shade[t];
		const Vec3f *src[3] = { &vnormals[i0], &vnormals[i1], &vnormals[i2] };
		float *dst[3] = { s.n0, s.n1, s.n2 };
		for (unsigned c = 0; c < 3; ++c) {
			dst[c][0] = src[c]->x;
			dst[c][1] = src[c]->y;
			dst[c][2] = src[c]->z;
			dst[c][3] = 0.0f;
		}
	}
	if (rebuilding.valid()) {
		std::vector<NodeRec> rebuilt = rebuilding.get();
		if (!rebuilt.empty()) {
			out.nodes.swap(rebuilt);
			out.rebuilt = true;
		}
	}
	// the leaf's own box, exactly as uploaded, next to its triangle (leaf indices are the same in either tree)
	for (const NodeRec &n : out.nodes)
		if (n.skip == 1)
			for (unsigned k = 0; k < 3; ++k) {
				out.tris[n.leaf].lo[k] = n.lo[k];
				out.tris[n.leaf].hi[k] = n.hi[k];
			}
	return out;
}

// The fast form of the shared walk tests a box with t = fma(b', inv, oi), oi = -(o * inv) rounded once, instead of
// the reference's fl(fl(b - o) * inv) (src/intersect_kernel.cl:21-61).  It only has to be CONSERVATIVE: whenever the
// reference's test passes for a (ray, box) pair, the fma test on the padded box b' must pass too; what it says about
// other pairs does not matter (a candidate leaf is gated by the exact test on its own box).
//
// Near plane, inv > 0 (the far plane and inv < 0 are mirror images).  Let s = fl(b - o), E = fl(s * inv) the
// reference's value, F = fl(b' * inv + oi) ours, u = 2^-24.  As reals,
//     b' * inv + oi  <=  (b' - o) * inv + u |o| inv + 2^-150          (oi = -fl(o * inv), 2^-150 if it underflows)
//     s * inv        >=  (b - o) * inv - u |b - o| inv                 (fl of a difference has relative error <= u)
// so  b - b' >= u (|b| + 2 |o|) + 2^-150  implies  b' * inv + oi <= s * inv, and rounding being monotone, F <= E:
// the conservative near value never exceeds the reference's, the far value never falls below it.
// |o| for a pair that PASSES the reference's test: t_near < max_distance and t_far > 0 put the origin within
// max_distance (|d| <= 1 per axis) of the box on every axis, |o_k| <= B_k + D with B_k the box's largest magnitude on
// axis k and D the ambient-occlusion rays' max_distance; the primary rays all start at the camera.  Hence per box and
// axis  O_k = max(|camera_k|, B_k + D),  capped by the global origin_limit the kernel checks per packet (rays beyond it,
// and rays with a reciprocal direction that is neither infinite nor below 1e30 -- o * inv could overflow -- take the
// exact form).  Margin used: 1.5 u (|b| + 2 O_k), rounded outward.
// An infinite inv (a zero direction component) is replaced by +-2^100 for the walk (kernels.hip, WalkRay): the
// argument above holds for every finite inv, so 2^100 (b' - o) is <= 0 whenever the reference's (b - o) * inf is -inf
// or NaN (no constraint), and where the reference gives +inf (the origin's coordinate outside the slab: reject) ours
// may be anything.  (Extent <= 1e6 keeps 2^100 * 1e6 finite.)
//
// The scaled form (kernels.hip, OCRT_TEST_COHERENT_SCALED; any-hit rays of one max_distance D per frame) replaces inv
// by inv' = fl(inv * r), r = walk_scale_for(D) ~ 1 / D, and the two limits "t_near < D", "t_far > 0" by the fma's
// clamp to [0, 1] on the z axis, compared STRICTLY:  max3(near'_x, near'_y, clamp near'_z) < min3(far'_x, far'_y,
// clamp far'_z).  (Strictly, because a box behind the origin on z has its negative far'_z clamped to 0, and 0 <= 0
// would let every such box through whose x and y slabs contain the origin -- a line through the whole scene.)
// Write Y = b' * inv' + oi' as a real, oi' = -fl(o * inv'), F' = fl(Y), X = s * inv as a real (E = fl(X)).  The
// argument above never used what inv is; with a further margin d on every plane it reads
//     near:  Y <= s * inv' - d |inv'|,      far:  Y >= s * inv' + d |inv'|,
// the underflow term being 2^-150 / |inv'| <= 2^-128 in plane units (|inv| >= 1/2: directions are unit vectors up to
// rounding, ray_is_selectable checks it; r >= 2^-21).  d = 14u D is enough (u D r >= u (1 - 2^-21), |inv'| >= r / 2):
//  * far, reference E > 0: s * inv' > 0, so Y >= d |inv'| >= 6.9u and F' >= 6.9u > 0.
//  * near, reference E < D: s * inv < D, s * inv' < D r (1 + u) <= 1 by the choice of r, Y <= 1 - 6.9u, F' < 1.
//  * near_i against far_j (any two planes), reference E_i <= E_j (and E_j > 0): if E_i <= 0, F'_i <= 0 < F'_j.  If
//    E_j >= 1.01 D, F'_j >= fl(1.01 D r (1 - 3u)) > 1 > F'_i (a clamped far'_z is 1, still above).  Otherwise
//    0 < E_i <= E_j < 1.01 D; as reals X_i <= X_j (1 + 2.1u), and inv'_i / inv_i, inv'_j / inv_j differ by at most
//    (1 +- u)^2 -- the one thing the scaling adds:
//        Y_j - Y_i >= r [(1 - u) (X_j + d |inv_j|) - (1 + u) ((1 + 2.1u) X_j - d |inv_i|)]
//                  >= r (1 - u) d |inv_j| - 4.2u r X_j  >  (1 - u) 7u (1 - 2^-21) - 4.3u  >  2.2u,
//    (r X_j < 1.02), and both values being below 1.05 their roundings move them by less than 1.05u each: F'_i < F'_j.
//  * a clamp never breaks this: clamping is monotone, F'_near < 1 and F'_far > 0 keep a clamped partner apart.
// Margin used for it: 16u * 1.001 D.  Infinite reciprocals: as above, 2^100 r stays finite and keeps the sign of b' - o.
float padded_bound(float b, float origin_bound, bool upper, float scaled_reach) {
	const double margin = 1.5 * std::ldexp(std::fabs((double) b) + 2.0 * (double) origin_bound, -24) +
	                      16.0 * std::ldexp((double) scaled_reach, -24) + 1.0e-38;
	const double moved = upper ? (double) b + margin : (double) b - margin;
	float f = (float) moved;  // round to nearest, then make sure it lies outside
	if (upper ? (double) f < moved : (double) f > moved)
		f = std::nextafterf(f, upper ? std::numeric_limits<float>::infinity() : -std::numeric_limits<float>::infinity());
	return f;
}

float walk_scale_for(float max_distance) {
	if (!(max_distance >= 0x1.0p-20f && max_distance <= 0x1.0p+20f))
		return 0.0f;
	const double want = (1.0 - std::ldexp(1.0, -22)) / (double) max_distance;
	float r = (float) want;
	while ((double) r * (double) max_distance * (1.0 + std::ldexp(1.0, -23)) > 1.0)
		r = std::nextafterf(r, 0.0f);
	return r;
}

// The scaled reciprocals meet origins up to `origin_limit`: oi' = -(o * inv') must stay finite for the largest
// stand-in of an infinite reciprocal (2^100 * scale) and for the largest finite one the packets admit (1e30 * scale);
// an overflow to +-inf would make every fma on that axis +-inf and the ray reject real boxes -- not conservative.
// (Scenes of extent 1e6 with a max_distance below ~0.01: the any-hit rays then take the exact form.)
bool walk_scale_usable(float scale, float origin_limit) {
	if (!(scale > 0.0f))
		return false;
	const double worst = (double) origin_limit * (double) scale;
	const double room = 0.25 * (double) std::numeric_limits<float>::max();
	return worst * std::ldexp(1.0, 100) < room && worst * 1.0e30 < room;
}

// A padded walk record (lo', hi': already conservative for the plane form, padded_bound) as centre c and half-extent e
// (stored in the lo / hi fields).  The kernel computes t_c = fl(c * inv + oi) and then near = fl(t_c - e |inv|),
// far = fl(t_c + e |inv|), each by ONE fma.  With T(x) = x * inv + oi as a real (oi as the kernel rounded it), the plane
// form's values are fl(T(lo')) and fl(T(hi')) (for inv > 0; mirrored otherwise).  |t_c - T(c)| <= u |T(c)| <=
// u (|c| + |o| (1 + u)) |inv|, hence
//     e >= max(c - lo', hi' - c) + u (|c| + 1.01 O)      (O >= |o|: the bound padded_bound was given for this box and axis)
// gives  t_c - e |inv| <= T(c) - (c - lo') |inv| = T(lo')  and  t_c + e |inv| >= T(hi')  as reals, and rounding being
// monotone the fma's results lie at or outside the plane form's: whatever passes there passes here.  The z axis' clamp
// is monotone too.  Nothing depends on the magnitude of inv (2^100 for an infinite reciprocal included: the bound scales
// with |inv|); e is computed in double and rounded up.  tests/walk_margin_check.cc throws the same pairs at this form.
void padded_centre_extent(float padded_lo, float padded_hi, float origin_bound, float *centre, float *half_extent) {
	const double lo = padded_lo, hi = padded_hi;
	const float c = (float) (0.5 * lo + 0.5 * hi);
	const double half = std::fmax((double) c - lo, hi - (double) c);
	const double want = (half + std::ldexp(std::fabs((double) c) + 1.01 * (double) origin_bound, -24)) * (1.0 + 1.0e-12) + 1.0e-38;
	float e = (float) want;
	if ((double) e < want)
		e = std::nextafterf(e, std::numeric_limits<float>::infinity());
	*centre = c;
	*half_extent = e;
}

static NodeRec ce_record(const NodeRec &padded, const double origin_bound[3]) {
	NodeRec r = padded;
	for (unsigned k = 0; k < 3; ++k)
		padded_centre_extent(padded.lo[k], padded.hi[k], (float) origin_bound[k], &r.lo[k], &r.hi[k]);
	return r;
}

WalkArray make_walk_array(const PackedScene &scene, float ao_max_distance, bool for_a_stream) {
	WalkArray out;
	const std::vector<NodeRec> &nodes = scene.nodes;
	if (!(scene.regular && scene.nested) || nodes.empty())
		return out;
	float extent = 0.0f;
	for (unsigned k = 0; k < 3; ++k)  // boxes are nested: the root's box bounds every coordinate
		extent = std::fmax(extent, std::fmax(std::fabs(nodes[0].lo[k]), std::fabs(nodes[0].hi[k])));
	if (!(extent <= 1.0e6f))
		return out;  // (o * inv must not overflow for |inv| <= 1e30: such scenes keep the exact form)
	// reach of the rays whose origin is not the camera; no ambient occlusion, or an unusable distance: the scene itself
	const bool ao_bounded = ao_max_distance > 0.0f && ao_max_distance <= extent;
	const double reach = ao_bounded ? (double) ao_max_distance * 1.001 : 2.0 * (double) extent;
	// ray origins: the camera at (0, 0, 2) and hit points, which lie in the root box up to the triangle test's slack
	out.origin_limit = 2.0f * extent + 4.0f;
	out.ao_scale = walk_scale_for(ao_max_distance);
	if (!walk_scale_usable(out.ao_scale, out.origin_limit))
		out.ao_scale = 0.0f;
	const double camera[3] = { 0.0, 0.0, 2.0 };  // reference src/intersect_kernel.cl:284
	// (both copies of the records are made in one sweep, the sweep cut into slices for a few threads: an upload's CPU
	// time is the walk-tree rebuild and this)
	if (2 * (nodes.size() + 4) * sizeof(NodeRec) >= (size_t) 1 << 32)
		out.ao_scale = 0.0f;  // (no room for the centre / half-extent copy below 2^32 bytes: the any-hit rays of such a scene take the exact form)
	const bool with_ce = out.ao_scale > 0.0f;
	const float scaled_reach_of = out.ao_scale > 0.0f ? ao_max_distance * 1.001f : 0.0f;
	// The plane-form records -- the closest-hit walk's, i.e. the primary rays' -- list every node's children nearest to
	// the camera first (walk_tree.h, nearest_children_first); the centre / half-extent copy -- the any-hit walk's -- keeps
	// the builder's order (its rays start on the surfaces: measured 4.5 % slower on the bunny in the camera's order).
	// Without that copy both walks read the one array, in the camera's order.
	std::vector<NodeRec> by_camera;
	// The closest-hit walk PRUNES: a lane with a hit at distance d does not enter a box of this copy whose near distance
	// exceeds d (1 + 1e-5) + prune_margin (kernels/primary.hip.h, far_limit).  That is right if every hit the reference's
	// triangle test can accept lies INSIDE its leaf's box of this copy (a point of the ray inside a box is not nearer than
	// the box) -- so the leaves' boxes are grown here by how far outside the box around its vertices an accepted hit can lie:
	//   * the test accepts s, t in [-1e-5, 1.00001] (src/intersect_kernel.cl:93-103): the point a + s u + t v lies up to
	//     1e-5 (|u| + |v|) outside the triangle;
	//   * s and t are float quotients by Cramer's rule, X / D and Y / D with X = uv wv - vv wu: with k = uu vv / |D| (1 / sin^2
	//     of the angle between u and v), r = the longer over the shorter of |u|, |v| and rho = how far outside [0, 1] the
	//     point's true parameters lie, |s_float - s_true| <= 41 * 2^-24 * k * r * rho (three-term dot products, the two
	//     products and the difference of X, the same for D, the division; |w| <= rho (|u| + |v|)); with eta = 128 * 2^-24 * k * r
	//     < 1/4 a point that passes has rho <= (1 + 1e-5) / (1 - eta) < 4/3 and lies within (1e-5 + 4/3 eta) (|u| + |v|) of
	//     the triangle: grown by (2e-5 + 4 eta) (|u| + |v|) -- used up to eta = 1/32 --, plus 32 * 2^-24 of the coordinates (the hit point itself is
	//     rounded);
	//   * a triangle with |n| < 9.9e-7 is never accepted (|dot(n, d)| < 1e-6 for every unit d, :80) and needs nothing;
	// and the inner boxes are the unions of their children's again.  In a scene with a triangle that can be accepted and
	// whose eta is 1/4 or more (needles, long slivers: Cramer's rule makes their accepted region fuzzy by whole edge
	// lengths) the closest-hit walk does NOT prune (prune_margin = +inf; both interior stand-ins are such scenes).  What
	// is left for prune_margin otherwise: the rounding of the two distances compared, 1e-5 of the largest coordinate.
	//   * the triangles left over -- needles, long slivers: Cramer's rule makes their accepted region fuzzy by a good part of
	//     their edge lengths or more -- get no box that could promise anything.  They are taken out of this copy's tree and put into
	//     a small tree of their own (their own boxes), which becomes the FIRST child of the root: the walk meets them before
	//     anything else, and while it is in there (`unpruned_bytes`) no lane's limit is lowered -- they are tested as the
	//     reference tests them.  (Both interior stand-ins have a few dozen such faces among 75 000.)
	bool prunable = for_a_stream;
	out.unpruned_bytes = 0;
	if (!prunable) {
		by_camera = nodes;  // (a one-shot host: nothing re-ordered, nothing grown, nothing pruned)
		out.prune_margin = std::numeric_limits<float>::infinity();
	} else {
		const auto never_accepted = [&](const TriRec &t) {
			return std::sqrt((double) t.n[0] * t.n[0] + (double) t.n[1] * t.n[1] + (double) t.n[2] * t.n[2]) < 9.9e-7;
		};
		// how far a leaf's box has to grow; +inf: no bound
		const auto growth_of = [&](uint32_t leaf) {
			if (leaf >= scene.tris.size())
				return std::numeric_limits<double>::infinity();
			const TriRec &t = scene.tris[leaf];
			if (never_accepted(t))
				return 0.0;
			const double uu = t.uu, vv = t.vv, area2 = std::fabs((double) t.D);
			const double lu = std::sqrt(uu), lv = std::sqrt(vv);
			double coordinate = 0.0;
			for (unsigned k = 0; k < 3; ++k)
				coordinate = std::fmax(coordinate, std::fabs((double) t.ta[k]));
			const double k = area2 > 0.0 ? uu * vv / area2 : std::numeric_limits<double>::infinity();
			const double r = lu > lv ? lu / lv : lv / lu;
			const double eta = 128.0 * std::ldexp(1.0, -24) * k * r;
			// (the bound holds up to eta = 1/4; but a box grown by a good part of its triangle's size is entered by packets that
			// have no business with it -- the HARDER interior stand-in's 17-unit slivers at eta = 0.2 grew by 27 units and its
			// primary pass from 0.098 to 0.156 ms --: from 1/32 on a face counts as one without a bound and keeps its own box)
			if (!(eta < 1.0 / 32.0))  // (NaN too)
				return std::numeric_limits<double>::infinity();
			const double grow = (2e-5 + 4.0 * eta) * (lu + lv) + 32.0 * std::ldexp(1.0, -24) * (coordinate + lu + lv + 2.0);
			return grow < 1e30 ? grow : std::numeric_limits<double>::infinity();
		};
		// (once per triangle)
		std::vector<float> growth(scene.tris.size());
		for (size_t t = 0; t < scene.tris.size(); ++t) {
			const double g = growth_of((uint32_t) t);
			growth[t] = g < 1e30 ? std::nextafterf((float) g, std::numeric_limits<float>::infinity()) : std::numeric_limits<float>::infinity();
		}
		const auto grown_by = [&](uint32_t leaf) { return leaf < growth.size() ? (double) growth[leaf] : std::numeric_limits<double>::infinity(); };
		// the leaves without a bound: out of the tree (inner nodes left with one child go too), into a tree of their own
		std::vector<NodeRec> loose;
		for (const NodeRec &n : nodes)
			if (n.skip == 1 && !(grown_by(n.leaf) < 1e30))
				loose.push_back(n);
		std::vector<NodeRec> rest;  // the tree of the others
		if (loose.empty()) {
			rest = nodes;
		} else {
			const size_t total = nodes.size();
			std::vector<uint32_t> size(total, 0);
			std::vector<char> keep(total, 0);
			for (size_t i = total; i-- > 0;) {
				if (nodes[i].skip == 1) {
					keep[i] = grown_by(nodes[i].leaf) < 1e30;
					size[i] = keep[i] ? 1u : 0u;
					continue;
				}
				uint32_t below = 0, kids = 0;
				for (size_t c = i + 1; c < i + nodes[i].skip && c < total; c += nodes[c].skip ? nodes[c].skip : 1) {
					below += size[c];
					kids += size[c] != 0;
				}
				keep[i] = kids >= 2;
				size[i] = below + (keep[i] ? 1u : 0u);
			}
			rest.reserve(size[0]);
			for (size_t i = 0; i < total; ++i)
				if (keep[i]) {
					rest.push_back(nodes[i]);
					rest.back().skip = nodes[i].skip == 1 ? 1u : size[i];
				}
		}
		// the others' boxes grown, leaves first, the inner ones as the unions of their children's
		double largest = 2.0;  // (the camera's z)
		for (size_t i = rest.size(); i-- > 0;) {
			NodeRec &n = rest[i];
			if (n.skip == 1) {
				const double grow = grown_by(n.leaf);
				if (!(grow < 1e30)) {
					prunable = false;
					break;
				}
				for (unsigned k = 0; k < 3; ++k) {
					n.lo[k] = std::nextafterf((float) ((double) n.lo[k] - grow), -std::numeric_limits<float>::infinity());
					n.hi[k] = std::nextafterf((float) ((double) n.hi[k] + grow), std::numeric_limits<float>::infinity());
					largest = std::fmax(largest, std::fmax(std::fabs((double) n.lo[k]), std::fabs((double) n.hi[k])));
				}
			} else {
				for (size_t c = i + 1; c < i + n.skip && c < rest.size(); c += rest[c].skip ? rest[c].skip : 1)
					for (unsigned k = 0; k < 3; ++k) {
						n.lo[k] = std::fmin(n.lo[k], rest[c].lo[k]);
						n.hi[k] = std::fmax(n.hi[k], rest[c].hi[k]);
					}
			}
		}
		const bool reorder = !debug_knob("OCRT_KEEP_CHILD_ORDER");  // (debug knob)
		if (prunable && loose.empty()) {
			by_camera = reorder ? nearest_children_first(rest, camera) : rest;
		} else if (prunable) {
			std::vector<NodeRec> own_tree = loose.size() == 1 ? loose : rebuild_walk_tree(loose);
			if (rest.size() >= 3 && rest[0].skip == rest.size() && !own_tree.empty() &&
			    (rest.size() + own_tree.size() + nodes.size() + 8) * sizeof(NodeRec) < (size_t) 1 << 32) {
				if (reorder)
					rest = nearest_children_first(rest, camera);
				by_camera.clear();
				by_camera.push_back(rest[0]);
				by_camera[0].skip = (uint32_t) (rest.size() + own_tree.size());
				for (unsigned k = 0; k < 3; ++k) {  // (the root holds the faces' own tree too)
					by_camera[0].lo[k] = std::fmin(by_camera[0].lo[k], own_tree[0].lo[k]);
					by_camera[0].hi[k] = std::fmax(by_camera[0].hi[k], own_tree[0].hi[k]);
				}
				by_camera.insert(by_camera.end(), own_tree.begin(), own_tree.end());
				by_camera.insert(by_camera.end(), rest.begin() + 1, rest.end());
				out.unpruned_bytes = (uint32_t) ((1 + own_tree.size()) * sizeof(NodeRec));
			} else {
				prunable = false;  // (nothing but such faces, or a tree this cannot be done to)
			}
		}
		out.prune_margin = prunable && largest < 1e30 ? (float) (1e-5 * largest) + 1e-30f : std::numeric_limits<float>::infinity();
		if (!prunable) {  // (nothing is pruned: the boxes as they were, every leaf where it was)
			by_camera = debug_knob("OCRT_KEEP_CHILD_ORDER") ? nodes : nearest_children_first(nodes, camera);
			out.unpruned_bytes = 0;
		}
	}
	// (the primary rays' records: a node or two more or fewer than the others -- inner nodes go with the faces taken out of
	// their tree, the faces' own tree brings some --, two END records, then the other copy)
	const size_t records = by_camera.size() + 2;
	out.primary_bytes = (uint32_t) (by_camera.size() * sizeof(NodeRec));
	out.nodes.resize(with_ce ? records + nodes.size() + 2 : records);
	auto padded = [&](const NodeRec &n, double origin[3]) {
		NodeRec w = n;
		for (unsigned k = 0; k < 3; ++k) {
			const double box = std::fmax(std::fabs((double) n.lo[k]), std::fabs((double) n.hi[k]));
			origin[k] = std::fmin((double) out.origin_limit, std::fmax(camera[k], box + reach));
			w.lo[k] = padded_bound(n.lo[k], (float) origin[k], false, scaled_reach_of);
			w.hi[k] = padded_bound(n.hi[k], (float) origin[k], true, scaled_reach_of);
		}
		w.skip = n.skip * (uint32_t) sizeof(NodeRec);  // (node count < 2^27, checked at pack time)
		return w;
	};
	auto pad_slice = [&](size_t from, size_t to) {
		for (size_t i = from; i < to; ++i) {
			double origin[3];
			if (i < by_camera.size())
				out.nodes[i] = padded(by_camera[i], origin);
			if (with_ce && i < nodes.size())
				out.nodes[records + i] = ce_record(padded(nodes[i], origin), origin);
		}
	};
	{
		const size_t longest = std::max(nodes.size(), by_camera.size());
		const size_t slices = longest > 16384 ? 4 : 1, per = (longest + slices - 1) / slices;
		std::vector<std::future<void>> running;
		for (size_t k = 1; k < slices; ++k)
			running.push_back(std::async(std::launch::async, pad_slice, k * per, std::min(longest, (k + 1) * per)));
		pad_slice(0, std::min(longest, per));
		for (auto &r : running)
			r.get();
	}
	// The record behind the last node ends the walk: every live ray "hits" its box (an infinite slab) and its
	// leaf field says END.  One more, because a node is fetched together with its successor.
	NodeRec end{};
	for (unsigned k = 0; k < 3; ++k) {
		end.lo[k] = -std::numeric_limits<float>::infinity();
		end.hi[k] = std::numeric_limits<float>::infinity();
	}
	end.skip = 0;
	end.leaf = WALK_END;
	out.nodes[by_camera.size()] = end;
	out.nodes[by_camera.size() + 1] = end;
	// The same records once more in CENTRE / HALF-EXTENT form, behind the END records, for the packets whose rays do not
	// agree on the sign of their direction (kernels.hip, OCRT_TEST_CE_SCALED): with t_c = fma(c, inv, oi) the two planes
	// of an axis are fma(-e, |inv|, t_c) and fma(e, |inv|, t_c) whatever the sign of inv -- nine fmas and no selects.
	// Only the scaled form (the any-hit rays) uses it.  Why it stays conservative: ce_record() above.
	if (with_ce) {
		out.ce_offset = (uint32_t) (records * sizeof(NodeRec));
		NodeRec ce_end = end;  // centre 0, half-extent +inf: near = -inf, far = +inf for every finite ray
		for (unsigned k = 0; k < 3; ++k) {
			ce_end.lo[k] = 0.0f;
			ce_end.hi[k] = std::numeric_limits<float>::infinity();
		}
		out.nodes[records + nodes.size()] = ce_end;
		out.nodes[records + nodes.size() + 1] = ce_end;
	}
	// (two records of slack behind everything: a scalar load that touches the line behind the last pair stays in the array)
	out.nodes.resize(out.nodes.size() + 2, NodeRec{});
	return out;
}

void prepare_walk_array(PackedScene &scene, float ao_max_distance, bool for_a_stream) {
	scene.walk = std::make_shared<const WalkArray>(make_walk_array(scene, ao_max_distance, for_a_stream));
	scene.walk_max_distance = ao_max_distance;
	scene.walk_for_a_stream = for_a_stream;
}

float kernel_float(float v) {
	if (!std::isfinite(v))
		return v;
	std::ostringstream ss;
	ss << v;
	return std::strtof(ss.str().c_str(), nullptr);
}

std::vector<float> uniform_ao_table(unsigned int rings, int alpha_min, int alpha_max) {
	std::vector<float> table;
	const float degrees = (float) (M_PI / 180);
	const float amin = (float) alpha_min * degrees;
	const float amax = (float) alpha_max * degrees;
	for (unsigned int ring = 0; ring < rings; ++ring) {
		const float step = amax / rings;
		const float elevation = (step * ring) + amin;
		const unsigned int ray_count = (unsigned int) ((2.0f * M_PI * std::cos(elevation)) / step);
		const float theta = (float) (M_PI_2 - elevation);
		for (unsigned int k = 0; k <= ray_count; ++k) {
			// The reference hands an angle that already contains 2*pi to
			// cospi/sinpi; cospi(x) is cos(pi * x) evaluated in double.
			const float phi = (float) ((2.0f * M_PI * k) / ray_count);
			table.push_back(std::sin(theta) * (float) std::cos(M_PI * (double) phi));
			table.push_back(std::cos(theta));
			table.push_back(std::sin(theta) * (float) std::sin(M_PI * (double) phi));
			table.push_back(0.0f);
		}
	}
	return table;
}

uint32_t band_tile_rows_for(unsigned int grid) {
	if (grid == 0)
		grid = 1;
	uint32_t a = TILE_H, b = grid;
	while (b) {
		const uint32_t t = a % b;
		a = b;
		b = t;
	}
	return grid / a;  // lcm(TILE_H, grid) / TILE_H
}

uint32_t local_tile_rows_for(uint32_t total_height, const Partition &part) {
	const uint32_t tile_rows = (total_height + TILE_H - 1) / TILE_H;
	const uint32_t bands = (tile_rows + part.band_tile_rows - 1) / part.band_tile_rows;
	uint32_t mine = 0;
	for (uint32_t b = part.rank; b < bands; b += part.nranks) {
		const uint32_t first = b * part.band_tile_rows;
		const uint32_t last = first + part.band_tile_rows < tile_rows ? first + part.band_tile_rows : tile_rows;
		// Keep the band's full tile-row count so the kernel's local->global map
		// stays a closed formula; rows past the image are masked off per lane.
		(void) last;
		mine += part.band_tile_rows;
	}
	return mine;
}

SceneFacts scene_facts(const PackedScene &scene, const WalkArray &walk) {
	SceneFacts f;
	f.regular = scene.regular;
	f.nested = scene.nested;
	f.binary_tree = scene.binary_tree;
	f.has_walk = !walk.nodes.empty();
	f.origin_limit = walk.origin_limit;
	f.ao_scale = walk.ao_scale;
	f.prune_margin = walk.prune_margin;
	f.unpruned_bytes = walk.unpruned_bytes;
	f.primary_bytes = walk.primary_bytes;
	f.ce_offset = walk.ce_offset;
	return f;
}

KernelParams make_kernel_params(const RayTracer &rt, uint32_t node_count, uint32_t tri_count, uint32_t ao_dirs,
                                const Partition &part, const SceneFacts *facts) {
	KernelParams p{};
	p.width = rt.totalWidth;
	p.height = rt.totalHeight;
	const float focal = kernel_float(rt.options.focalLength);
	const int longest = (int) (p.width > p.height ? p.width : p.height);
	p.a = focal * (float) longest;
	p.half_w = (float) (int) p.width / (2.0f * p.a);
	p.half_h = (float) (int) p.height / (2.0f * p.a);
	p.node_count = node_count;
	p.tri_count = tri_count;
	p.shading = rt.options.enableShading ? 1 : 0;
	p.ao_mode = AO_NONE;
	if (rt.options.enableAO && rt.options.aoNumSamples > 0)
		p.ao_mode = rt.options.aoMethod == RayTracer::AmbientOcclusionMethod::UNIFORM ? AO_UNIFORM : AO_RANDOM;
	p.ao_max_distance = kernel_float(rt.options.aoMaxDistance);
	p.ao_dirs = ao_dirs;
	p.ao_divisor = p.ao_mode == AO_RANDOM ? ao_dirs - 1 : ao_dirs;
	p.origin_limit = facts ? facts->origin_limit : 0.0f;
	p.prune_margin = facts && !debug_knob("OCRT_NO_PRUNE") ? facts->prune_margin : __builtin_inff();  // (debug knob)
	p.unpruned_bytes = facts ? facts->unpruned_bytes : 0u;
	p.primary_walk_bytes = facts ? facts->primary_bytes : 0u;
	p.walk_ce_bytes = facts ? facts->ce_offset : 0u;
	// (the array's margins hold for the max_distance it was made for: the renderer re-makes it when that changes)
	p.walk_scale = (facts && facts->ao_scale > 0.0f && facts->ao_scale == walk_scale_for(p.ao_max_distance) &&
	                walk_scale_usable(facts->ao_scale, facts->origin_limit) &&
	                !debug_knob("OCRT_NO_SCALED_WALK")) ? facts->ao_scale : 0.0f;  // (debug knob)
	p.fast_walk = (facts && facts->has_walk && !debug_knob("OCRT_FORCE_EXACT_WALK")) ? 1 : 0;  // (debug knob)
	p.scene_regular = (facts && facts->regular) ? 1 : 0;
	p.scene_nested = (facts && facts->nested && !debug_knob("OCRT_FORCE_EXACT_WALK")) ? 1 : 0;  // (debug knob)
	p.shared_walk = (facts && facts->binary_tree && !debug_knob("OCRT_NO_SHARED_WALK")) ? 1 : 0;  // (debug knob)
	p.debug_no_sort = debug_knob("OCRT_NO_SORT") ? 1 : 0;
	const char *refill_min = debug_knob("OCRT_REFILL_MIN"), *leaf_min = debug_knob("OCRT_LEAF_MIN");
	p.refill_min = refill_min ? (uint32_t) std::atoi(refill_min) : 16u;
	p.leaf_min = leaf_min ? (uint32_t) std::atoi(leaf_min) : 16u;
	const char *guide = debug_knob("OCRT_AO_GUIDE");  // debug knob
	p.ao_guide = guide && std::atoi(guide) > 0 ? (uint32_t) std::atoi(guide) : 0u;  // (0: claims never shrink, the default)
	const char *claim_max = debug_knob("OCRT_AO_CLAIM_MAX");  // debug knob
	p.ao_claim_max = claim_max && std::atoi(claim_max) > 0 ? (uint32_t) std::atoi(claim_max) : 0u;
	p.ao_claim_div = 1u;
	const char *batch_below = debug_knob("OCRT_BATCH_BELOW");  // debug knob
	// (32 until round 3; with the coarser ordering key below 40 ... 56 measure alike and 1-2 % better, profiles/r03_notes.md)
	p.batch_below = batch_below ? (uint32_t) std::atoi(batch_below) : 48u;
	const char *cost_shift = debug_knob("OCRT_COST_SHIFT");  // debug knob
	// Ordering key of a block of 64 tiles = 1 + (sum of its tiles' cost classes >> 2), capped at 64: every block whose
	// tiles average four leaf stops or more -- the model -- shares the top key and is claimed first IN SPATIAL ORDER
	// (neighbouring claims walk the same part of the tree: the scalar cache and the XCD's L2 see it again), the cheap
	// blocks after them by cost.  A finer key for the costly blocks (>> 5 until round 3: costliest first) shortened the
	// tail of a pass that runs alone by a little and cost the locality: -3 % ... -7 % per frame on every workload with
	// frames in flight, -6 ... -10 % for the pass alone at 600 x 600 -s 4 and on the interior scene, +0.5 % for the
	// headline pass alone (profiles/r03_notes.md).
	p.cost_shift = cost_shift ? (uint32_t) std::atoi(cost_shift) & 31u : 2u;
	p.ao_regular = (p.ao_max_distance > 0.0f && std::isfinite(p.ao_max_distance)) ? 1 : 0;
	p.primary_below = std::nextafterf(100000.0f, 0.0f);
	p.ao_below = std::nextafterf(p.ao_max_distance, -std::numeric_limits<float>::infinity());
	p.tiles_x = (p.width + TILE_W - 1) / TILE_W;
	p.strip_tiles = 2u;  // (DeviceRenderer::adopt widens the strips for scenes far beyond the L2s)
	p.entry_stride = 1u + ao_dirs;  // (... and keeps the tiles' own walk intervals only where the table per direction would be too large)
	p.part = part;
	p.local_tile_rows = local_tile_rows_for(p.height, part);
	return p;
}

}  // namespace ocrt
