// prune_check.cc -- CPU check (no GPU) of what the closest-hit walk's pruning rests on (kernels/primary.hip.h, far_limit;
// scene_pack.cc, make_walk_array): in the copy of the walk records the primary rays read,
//   * every hit the reference's triangle test accepts (src/intersect_kernel.cl:65-114, restated here operation for
//     operation) lies at or behind the near distance of its leaf's box -- so a box that begins behind a lane's hit
//     holds nothing nearer --, checked for rays through the bunny from the reference's camera, with the rays shaken so that
//     hits in the test's slack zone (s, t a little outside [0, 1]) occur;
//   * both copies hold the same leaves, skip counts that tile the array, children nearest to the camera first in the one;
//   * faces no box can promise anything about (needles, long slivers) lie at the head of that copy, where no limit is
//     lowered; a face that can never be hit (|n| < 1e-6) does not matter.
// Exit code 0 = fine.   usage: prune_check bunny.off interior_hard.off
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "bvh.h"
#include "mesh.h"
#include "scene_pack.h"

using namespace ocrt;

struct Accepted {
	bool ok;
	double distance;
};
// the reference's test on a TriRec, float operations in its order (the build: -ffp-contract=off)
static Accepted reference_triangle(const TriRec &t, const float o[3], const float d[3]) {
	const float w0[3] = { o[0] - t.ta[0], o[1] - t.ta[1], o[2] - t.ta[2] };
	const float a = -((t.n[0] * w0[0] + t.n[1] * w0[1]) + t.n[2] * w0[2]);
	const float b = (t.n[0] * d[0] + t.n[1] * d[1]) + t.n[2] * d[2];
	if (std::fabs(b) < 0.000001f)
		return { false, 0 };
	const float r = a / b;
	if (r < 0.0f)
		return { false, 0 };
	const float ip[3] = { o[0] + r * d[0], o[1] + r * d[1], o[2] + r * d[2] };
	const float w[3] = { ip[0] - t.ta[0], ip[1] - t.ta[1], ip[2] - t.ta[2] };
	const float wu = (t.u[0] * w[0] + t.u[1] * w[1]) + t.u[2] * w[2];
	const float wv = (w[0] * t.v[0] + w[1] * t.v[1]) + w[2] * t.v[2];
	const float s = (t.uv * wv - t.vv * wu) / t.D;
	if (s < -0.00001f || (double) s > 1.00001)
		return { false, 0 };
	const float q = (t.uv * wu - t.uu * wv) / t.D;
	if (q < -0.00001f || (double) (s + q) > 1.00001)
		return { false, 0 };
	const float e[3] = { ip[0] - o[0], ip[1] - o[1], ip[2] - o[2] };
	return { true, (double) std::sqrt((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) };
}

// near distance of a box along a ray, in double (the walk's own value is a lower bound of it: walk_margin_check.cc)
static bool box_near(const NodeRec &n, const float o[3], const float d[3], double *near_out) {
	double near = 0.0, far = std::numeric_limits<double>::infinity();
	for (int k = 0; k < 3; ++k) {
		if (d[k] == 0.0f) {
			if (o[k] < n.lo[k] || o[k] > n.hi[k])
				return false;
			continue;
		}
		const double a = ((double) n.lo[k] - o[k]) / d[k], b = ((double) n.hi[k] - o[k]) / d[k];
		near = std::fmax(near, std::fmin(a, b));
		far = std::fmin(far, std::fmax(a, b));
	}
	*near_out = near;
	return near <= far;
}

static PackedScene pack(Mesh &m) {
	compute_vertex_normals(&m);
	BVH bvh(BVH::Method::CUT_LONGEST_AXIS);
	bvh.buildBVH(m);
	const auto sorted = sort_faces_by_leaf_order(m, bvh);
	return pack_scene(sorted, bvh.nodes, bvh.aabbs, m.vertices, m.vnormals);
}

int main(int argc, char **argv) {
	int bad = 0;
	auto expect = [&](bool ok, const char *what) {
		if (!ok) {
			std::printf("FAILED: %s\n", what);
			++bad;
		}
	};
	if (argc < 2)
		return 2;
	for (int which = 1; which < argc; ++which) {
	Mesh bunny;
	load_off_mesh(argv[which], &bunny);
	const PackedScene scene = pack(bunny);
	const WalkArray walk = make_walk_array(scene, 0.2f);
	const size_t count = scene.nodes.size();
	expect(!walk.nodes.empty() && walk.ce_offset != 0 && walk.nodes.size() >= walk.ce_offset / sizeof(NodeRec) + count + 2, "the mesh has both copies of the walk records");
	expect(std::isfinite(walk.prune_margin) && walk.prune_margin > 0.0f && walk.prune_margin < 1e-3f, "the mesh is pruned, by a small margin");
	if (bad)
		return 1;
	const NodeRec *by_camera = walk.nodes.data(), *any_hit = (const NodeRec *) ((const char *) walk.nodes.data() + walk.ce_offset);
	// the faces no box promises anything about lie in [1, loose_end) of the primary rays' copy (none in the bunny, a few dozen
	// in the interior stand-ins); that copy may be a few nodes shorter than the other (END records fill it up)
	const size_t loose_end = walk.unpruned_bytes / sizeof(NodeRec);
	const size_t camera_count = walk.primary_bytes / sizeof(NodeRec);
	std::printf("prune_check: %s: %zu nodes, %zu in the primary rays' copy, the first %zu hold the faces without a bound, margin %g\n", argv[which], count,
	            camera_count, loose_end, (double) walk.prune_margin);
	expect(which == 1 ? loose_end == 0 : loose_end > 2, which == 1 ? "the bunny has no face without a bound" : "the second mesh has faces without a bound");
	// structure: the same leaves in both, skips that tile, the nearer child first
	{
		std::vector<uint32_t> seen[2];
		seen[0].assign(scene.tris.size(), 0);
		seen[1].assign(scene.tris.size(), 0);
		bool tiles = true, ordered = true;
		const double eye[3] = { 0, 0, 2 };
		auto outside2 = [&](const NodeRec &n) {  // (of the UNPADDED box: the padded one is monotone in it up to the padding)
			double s = 0;
			for (int k = 0; k < 3; ++k) {
				const double dd = eye[k] < n.lo[k] ? n.lo[k] - eye[k] : eye[k] > n.hi[k] ? eye[k] - n.hi[k] : 0.0;
				s += dd * dd;
			}
			return s;
		};
		for (int copy = 0; copy < 2; ++copy) {
			const NodeRec *nodes = copy ? any_hit : by_camera;
			const size_t here = copy ? count : camera_count;
			for (size_t i = 0; i < here; ++i) {
				const size_t skip = nodes[i].skip / sizeof(NodeRec);
				tiles = tiles && skip >= 1 && i + skip <= here;
				if (skip == 1) {
					if (nodes[i].leaf < scene.tris.size())
						++seen[copy][nodes[i].leaf];
					continue;
				}
				size_t c = i + 1, inside = 0;
				double before = -1.0;
				while (c < i + skip) {
					const size_t cs = nodes[c].skip / sizeof(NodeRec);
					if (cs == 0) { tiles = false; break; }
					if (copy == 0 && !(i == 0 && c < loose_end) && !(i >= 1 && i < loose_end)) {  // (the loose faces' subtree comes first whatever its distance)
						const double now = outside2(nodes[c]);
						ordered = ordered && now >= before - 1e-3 * (1.0 + before);  // (the records are padded: equal up to that)
						before = now;
					}
					inside += cs;
					c += cs;
				}
				tiles = tiles && inside + 1 == skip;
			}
		}
		bool same = true;
		for (size_t t = 0; t < scene.tris.size(); ++t)
			same = same && seen[0][t] == 1 && seen[1][t] == 1;
		expect(tiles, "skip counts tile both copies");
		expect(same, "both copies hold every leaf once");
		expect(ordered, "the primary rays' copy lists children nearest to the camera first");
	}
	// the invariant: an accepted hit is not nearer than its leaf's box in the primary rays' copy
	{
		std::vector<const NodeRec *> leaf_box(scene.tris.size(), nullptr);
		std::vector<char> loose(scene.tris.size(), 0);
		for (size_t i = 0; i < camera_count; ++i)
			if (by_camera[i].skip / sizeof(NodeRec) == 1 && by_camera[i].leaf < scene.tris.size()) {
				leaf_box[by_camera[i].leaf] = &by_camera[i];
				loose[by_camera[i].leaf] = i >= 1 && i < loose_end;
			}
		std::mt19937 rng(20261005);
		std::uniform_real_distribution<float> shake(-1.0f, 1.0f);
		unsigned long long hits = 0, slack_hits = 0, violations = 0;
		double worst = 0.0;
		const float o[3] = { 0.0f, 0.0f, 2.0f };
		for (int ray = 0; ray < 600; ++ray) {
			// towards a vertex of the mesh, shaken by a few 1e-6: through edges and corners, where the slack zone is
			const Vec3f &v = bunny.vertices[rng() % bunny.vertices.size()];
			float d[3] = { v.x - o[0] + 3e-6f * shake(rng), v.y - o[1] + 3e-6f * shake(rng), v.z - o[2] + 3e-6f * shake(rng) };
			const float len = std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
			for (float &x : d)
				x /= len;
			for (size_t t = 0; t < scene.tris.size(); ++t) {
				const Accepted h = reference_triangle(scene.tris[t], o, d);
				if (!h.ok || loose[t])  // (a face without a bound promises nothing: the walk meets it before any limit is lowered)
					continue;
				++hits;
				double near = 0.0;
				// (the reference tests a triangle only if the ray meets its leaf's own box; a hit accepted here without
				// that is none of the walk's business -- but most are tested, and all of those must obey)
				NodeRec own{};
				for (int k = 0; k < 3; ++k) {
					own.lo[k] = scene.tris[t].lo[k];
					own.hi[k] = scene.tris[t].hi[k];
				}
				double own_near;
				if (!box_near(own, o, d, &own_near))
					continue;
				if (!leaf_box[t] || !box_near(*leaf_box[t], o, d, &near)) {
					++violations;  // (the grown box must be met whenever the leaf's own is)
					continue;
				}
				slack_hits += own_near > h.distance;  // (the hit lies in front of the leaf's OWN box: the case the growth is for)
				if (near > h.distance * (1.0 + 1e-5) + walk.prune_margin) {
					++violations;
					worst = std::fmax(worst, near - h.distance);
				}
			}
		}
		std::printf("prune_check: %llu accepted hits, %llu of them in front of their leaf's own box, %llu violations (worst %.3g)\n", hits, slack_hits,
		            violations, worst);
		expect(hits > 1000, "the rays hit something");
		expect(violations == 0, "no accepted hit lies in front of its leaf's box in the primary rays' copy");
	}
	}
	// a needle among ordinary triangles goes where nothing is pruned; a degenerate face does not matter
	for (int kind = 0; kind < 2; ++kind) {
		Mesh m;
		m.vertices = { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 1, 1, 0 }, { 0, 0, 1 }, { 1, 0, 1 }, { 0.5f, 1e-3f, 1 }, { 2, 2, 2 } };
		m.faces = { 0, 1, 2, 1, 3, 2 };
		if (kind == 0)
			m.faces.insert(m.faces.end(), { 4, 5, 6 });  // a needle: height 1e-3 over a base of 1 -- it can be hit, and Cramer's rule makes its accepted region fuzzy
		else
			m.faces.insert(m.faces.end(), { 7, 7, 7 });  // a point: never accepted (|n| = 0), it needs no guarantee
		const PackedScene p = pack(m);
		const WalkArray w = make_walk_array(p, 0.2f);
		if (kind == 0)
			expect(w.nodes.empty() || std::isinf(w.prune_margin) || w.unpruned_bytes == 2 * sizeof(NodeRec), "a needle that can be hit lies where nothing is pruned");
		else
			expect(!w.nodes.empty() && std::isfinite(w.prune_margin) && w.unpruned_bytes == 0, "a face that can never be hit does not matter");
	}
	if (!bad)
		std::printf("prune_check: ok\n");
	return bad ? 1 : 0;
}
