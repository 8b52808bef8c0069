#include "bvh.h"

#include <limits>
#include <stdexcept>
#include <utility>

// Per-triangle quantities every split needs, computed once.  The values are
// those of reference src/triangle.cc:4-23: centroid = ((a + b) + c) / 3.0f and
// a tight box from std::min(a, std::min(b, c)) per axis.
struct BVH::Prims {
	std::vector<Vec3f> centroid;
	std::vector<AABB> box;

	explicit Prims(const Mesh &mesh) {
		const size_t count = mesh.faces.size() / 3;
		centroid.resize(count);
		box.resize(count);
		for (size_t f = 0; f < count; ++f) {
			const Vec3f &a = mesh.vertices[mesh.faces[3 * f + 0]];
			const Vec3f &b = mesh.vertices[mesh.faces[3 * f + 1]];
			const Vec3f &c = mesh.vertices[mesh.faces[3 * f + 2]];
			centroid[f] = (a + b + c) / 3.0f;
			Vec3f lo, hi;
			for (unsigned k = 0; k < 3; ++k) {
				lo[k] = std::min(a[k], std::min(b[k], c[k]));
				hi[k] = std::max(a[k], std::max(b[k], c[k]));
			}
			box[f] = AABB(lo, hi);
		}
	}
};

namespace {

// reference src/bvh.cc:38-43: float products, doubled in double, returned as float.
float surface_area(const AABB &bb) {
	const float w = bb.max[0] - bb.min[0];
	const float h = bb.max[1] - bb.min[1];
	const float d = bb.max[2] - bb.min[2];
	return (float) (2.0 * (w * h + h * d + d * w));
}

}  // namespace

// Midpoint split of the centroid box along its longest axis
// (reference src/bvh.cc:59-94).  Relative order inside each half is kept.
void BVH::splitLongestAxis(const Prims &prims, std::vector<uint32_t> &ids, std::vector<uint32_t> &left,
                           std::vector<uint32_t> &right, AABB &bb) const {
	AABB centroid_box;
	for (uint32_t id : ids) {
		centroid_box.merge(prims.centroid[id]);
		bb.merge(prims.box[id]);
	}
	const int axis = centroid_box.getLongestAxis();
	centroid_box.max[axis] = (centroid_box.max[axis] + centroid_box.min[axis]) / 2;
	for (uint32_t id : ids)
		(centroid_box.inside(prims.centroid[id]) ? left : right).push_back(id);
	if (left.empty()) {
		left.push_back(right.back());
		right.pop_back();
	}
	if (right.empty()) {
		right.push_back(left.back());
		left.pop_back();
	}
}

// Surface-area-heuristic split (reference src/bvh.cc:178-237).  Per axis the ids
// are std::sort-ed by DESCENDING centroid coordinate; candidate i puts ids[0..i)
// left and ids[i..n) right, cost = 1 + SA_l/SA * i + SA_r/SA * (n - i) evaluated
// in double and rounded to float, first strict minimum wins.  The reference
// rebuilds the right box from scratch for every candidate; min/max are exact,
// so a suffix scan yields the same boxes and therefore the same tree.
void BVH::splitSAH(const Prims &prims, std::vector<uint32_t> &ids, std::vector<uint32_t> &left,
                   std::vector<uint32_t> &right, AABB &bb) const {
	for (uint32_t id : ids)
		bb.merge(prims.box[id]);
	const float traversal_cost = 1.0f, primitive_cost = 1.0f;
	const float area = surface_area(bb);
	const size_t n = ids.size();
	size_t best_axis = 0;
	size_t best_pos = 1;
	float best_cost = std::numeric_limits<float>::max();
	std::vector<float> right_area(n);
	for (size_t axis = 0; axis < 3; ++axis) {
		std::sort(ids.begin(), ids.end(), [&](size_t i, size_t j) {
			return prims.centroid[i][(unsigned) axis] > prims.centroid[j][(unsigned) axis];
		});
		if (n < 3)
			continue;
		AABB suffix;
		for (size_t i = n - 1; i >= 1; --i) {
			suffix.merge(prims.box[ids[i]]);
			right_area[i] = surface_area(suffix);
		}
		AABB prefix = prims.box[ids[0]];
		double count_left = 1, count_right = (double) (n - 1);
		for (size_t i = 1; i + 1 < n; ++i) {
			const float sa_left = surface_area(prefix);
			const float sa_right = right_area[i];
			const float cost = (float) (traversal_cost + (sa_left / area) * count_left * primitive_cost +
			                            (sa_right / area) * count_right * primitive_cost);
			if (cost < best_cost) {
				best_cost = cost;
				best_pos = i;
				best_axis = axis;
			}
			prefix.merge(prims.box[ids[i]]);
			++count_left;
			--count_right;
		}
	}
	// ids are sorted along z now; restore the winning order if it was x or y.
	if (best_axis < 2)
		std::sort(ids.begin(), ids.end(), [&](size_t i, size_t j) {
			return prims.centroid[i][(unsigned) best_axis] > prims.centroid[j][(unsigned) best_axis];
		});
	left.assign(ids.begin(), ids.begin() + best_pos);
	right.assign(ids.begin() + best_pos, ids.end());
}

// Pre-order emission with an explicit work stack (reference src/bvh.cc:98-162
// recurses; a degenerate mesh would take that T levels deep).
void BVH::buildBVH(const Mesh &mesh) {
	const size_t count = mesh.faces.size() / 3;
	if (count == 0)
		throw std::runtime_error("Cannot build a BVH for an empty mesh");
	const Prims prims(mesh);
	triangles.clear();
	triangles.reserve(count);
	nodes.assign(count * 2 - 1, 0u);
	aabbs.assign((count * 2 - 1) * 2, Vec3f());

	struct Frame {
		std::vector<uint32_t> ids;
		std::vector<uint32_t> pending_right;
		size_t node;
		int stage;
	};
	std::vector<Frame> stack;
	size_t next_node = 0;
	{
		Frame root;
		root.ids.resize(count);
		for (size_t i = 0; i < count; ++i)
			root.ids[i] = (uint32_t) i;
		root.node = next_node++;
		root.stage = 0;
		stack.push_back(std::move(root));
	}
	while (!stack.empty()) {
		Frame &f = stack.back();
		if (f.stage == 0) {
			AABB bb;
			if (f.ids.size() <= 1) {
				for (uint32_t id : f.ids) {
					bb.merge(prims.box[id]);
					triangles.push_back(id);
				}
				aabbs[f.node * 2] = bb.min;
				aabbs[f.node * 2 + 1] = bb.max;
				nodes[f.node] = 1;
				stack.pop_back();
				continue;
			}
			std::vector<uint32_t> left, right;
			left.reserve(f.ids.size());
			right.reserve(f.ids.size());
			if (method == Method::CUT_LONGEST_AXIS)
				splitLongestAxis(prims, f.ids, left, right, bb);
			else
				splitSAH(prims, f.ids, left, right, bb);
			aabbs[f.node * 2] = bb.min;
			aabbs[f.node * 2 + 1] = bb.max;
			std::vector<uint32_t>().swap(f.ids);
			f.pending_right = std::move(right);
			f.stage = 1;
			Frame child;
			child.ids = std::move(left);
			child.node = next_node++;
			child.stage = 0;
			stack.push_back(std::move(child));  // invalidates f
		} else if (f.stage == 1) {
			f.stage = 2;
			Frame child;
			child.ids = std::move(f.pending_right);
			child.node = next_node++;
			child.stage = 0;
			stack.push_back(std::move(child));
		} else {
			nodes[f.node] = (uint32_t) (next_node - f.node);
			stack.pop_back();
		}
	}
	nodes.resize(nodes[0]);
	aabbs.resize((size_t) nodes[0] * 2);
}

std::vector<uint32_t> sort_faces_by_leaf_order(const Mesh &mesh, const BVH &bvh) {
	std::vector<uint32_t> sorted;
	sorted.reserve(bvh.triangles.size() * 3);
	for (uint32_t face : bvh.triangles)
		for (unsigned k = 0; k < 3; ++k)
			sorted.push_back(mesh.faces[(size_t) face * 3 + k]);
	return sorted;
}
