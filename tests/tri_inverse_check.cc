// tri_inverse_check.cc -- CPU check (no GPU) of what pack_scene puts into TriRec::inv_d: the correctly rounded 1 / D of
// every triangle whose D allows the short form of the any-hit triangle test (opencl_raytracer_amd/csrc/tri_predicate.h),
// a NaN for the others (zero-area triangles, D beyond 1e+-30) -- and of tri_inverse_d's boundaries.  Exit code 0 = fine.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "scene_pack.h"
#include "tri_predicate.h"

using namespace ocrt;

static uint32_t bits_of(float x) {
	uint32_t b;
	std::memcpy(&b, &x, 4);
	return b;
}

int main() {
	int bad = 0;
	auto expect = [&](bool ok, const char *what) {
		if (!ok) {
			std::printf("FAILED: %s\n", what);
			++bad;
		}
	};
	// boundaries of the function itself
	expect(std::isnan(tri_inverse_d(0.0f)) && std::isnan(tri_inverse_d(-0.0f)), "zero");
	expect(std::isnan(tri_inverse_d(1.0e-38f)) && std::isnan(tri_inverse_d(-9.9e-31f)), "below 1e-30");
	expect(std::isnan(tri_inverse_d(1.1e30f)) && std::isnan(tri_inverse_d(-INFINITY)) && std::isnan(tri_inverse_d(NAN)), "above 1e30 / inf / NaN");
	expect(bits_of(tri_inverse_d(1.0e-30f)) == bits_of(1.0f / 1.0e-30f) && bits_of(tri_inverse_d(-1.0e30f)) == bits_of(1.0f / -1.0e30f), "the limits themselves");
	expect(bits_of(tri_inverse_d(-3.0f)) == bits_of(1.0f / -3.0f), "an ordinary D");
	// through pack_scene: a quad of two ordinary triangles, a zero-area triangle, a huge one, a tiny one
	const float S = 1.0e8f, T = 1.0e-9f;  // D ~ -S^4 = -1e32 (refused), D ~ -T^4 = -1e-36 (refused)
	std::vector<Vec3f> vertices = { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 1, 1, 0.25f }, { 2, 2, 2 }, { 3, 3, 3 }, { 4, 4, 4 },
		                            { 0, 0, 5 }, { S, 0, 5 }, { 0, S, 5 }, { 0, 0, 6 }, { T, 0, 6 }, { 0, T, 6 } };
	std::vector<uint32_t> faces = { 0, 1, 2, 1, 3, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12 };
	const size_t tri_count = faces.size() / 3;
	// a flat tree: root + one leaf per triangle, boxes = the triangles' own
	std::vector<uint32_t> nodes;
	std::vector<Vec3f> aabbs;
	auto box_of = [&](size_t first, size_t last, Vec3f &lo, Vec3f &hi) {
		lo = Vec3f(INFINITY, INFINITY, INFINITY);
		hi = Vec3f(-INFINITY, -INFINITY, -INFINITY);
		for (size_t t = first; t < last; ++t)
			for (unsigned c = 0; c < 3; ++c)
				for (unsigned k = 0; k < 3; ++k) {
					const float value = vertices[faces[3 * t + c]][k];
					lo[k] = std::fmin(lo[k], value);
					hi[k] = std::fmax(hi[k], value);
				}
	};
	// full binary tree over the leaves in pre-order: node count of a subtree with n leaves = 2n - 1
	struct Build {
		std::vector<uint32_t> &nodes;
		std::vector<Vec3f> &aabbs;
		decltype(box_of) &box;
		void operator()(size_t first, size_t last) {
			Vec3f lo, hi;
			box(first, last, lo, hi);
			nodes.push_back((uint32_t) (2 * (last - first) - 1));
			aabbs.push_back(lo);
			aabbs.push_back(hi);
			if (last - first > 1) {
				const size_t mid = (first + last) / 2;
				(*this)(first, mid);
				(*this)(mid, last);
			}
		}
	} build{ nodes, aabbs, box_of };
	build(0, tri_count);
	std::vector<Vec3f> normals(vertices.size(), Vec3f(0, 0, 1));
	const PackedScene packed = pack_scene(faces, nodes, aabbs, vertices, normals);
	expect(packed.tris.size() == tri_count, "triangle count");
	for (size_t t = 0; t < packed.tris.size(); ++t) {
		const TriRec &r = packed.tris[t];
		const float magnitude = std::fabs(r.D);
		const bool usable = magnitude >= 1.0e-30f && magnitude <= 1.0e30f;
		std::printf("triangle %zu: D %.9g inv_d %.9g\n", t, r.D, r.inv_d);
		if (usable)
			expect(bits_of(r.inv_d) == bits_of(1.0f / r.D), "inv_d is RN(1 / D)");
		else
			expect(std::isnan(r.inv_d), "inv_d is a NaN where D is no use");
	}
	expect(!std::isnan(packed.tris[0].inv_d) && !std::isnan(packed.tris[1].inv_d), "the ordinary triangles take the short form");
	expect(packed.tris.size() == 5 && std::isnan(packed.tris[2].inv_d) && std::isnan(packed.tris[3].inv_d) && std::isnan(packed.tris[4].inv_d),
	       "zero-area, huge and tiny triangles take the divisions");
	std::printf("%s\n", bad ? "tri_inverse_check: FAILED" : "tri_inverse_check: ok");
	return bad ? 1 : 0;
}
