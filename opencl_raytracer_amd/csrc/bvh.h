// bvh.h -- CPU bounding-volume-hierarchy builder.
//
// Same public surface as reference include/bvh.h:6-26: construct with a Method,
// call buildBVH(mesh), read the three flat arrays.  The array encoding is the
// one the reference kernel walks (reference src/intersect_kernel.cl:184-213):
//   nodes[i]     = size of the subtree rooted at i, nodes in PRE-ORDER, leaf = 1
//   aabbs[2i]    = box min of node i, aabbs[2i+1] = box max (16-byte elements)
//   triangles[k] = face id stored in the k-th leaf met in pre-order
// Every leaf holds exactly one triangle, so #leaves(i) = (nodes[i] + 1) / 2.
//
// Both split strategies reproduce the reference's trees bit for bit
// (tests/test_bvh_golden.py); the SAH split gets there with an O(n) suffix
// sweep per axis instead of the reference's O(n^2) rescan.
#pragma once
#include <cstdint>
#include <vector>

#include "aabb.h"
#include "mesh.h"

class BVH {
	public:
		enum class Method { CUT_LONGEST_AXIS, SURFACE_AREA_HEURISTIC };

		BVH() : method(Method::CUT_LONGEST_AXIS) {}
		explicit BVH(Method m) : method(m) {}

		// Builds the hierarchy for `mesh` (throws std::runtime_error on an
		// empty mesh; the reference would die in an allocation there).
		void buildBVH(const Mesh &mesh);

		std::vector<uint32_t> triangles;
		std::vector<uint32_t> nodes;
		std::vector<Vec3f> aabbs;

	private:
		struct Prims;
		void splitLongestAxis(const Prims &prims, std::vector<uint32_t> &ids, std::vector<uint32_t> &left,
		                      std::vector<uint32_t> &right, AABB &bb) const;
		void splitSAH(const Prims &prims, std::vector<uint32_t> &ids, std::vector<uint32_t> &left,
		              std::vector<uint32_t> &right, AABB &bb) const;
		Method method;
};

// Leaf-order face table the kernel consumes: out[3k..3k+2] = the three vertex
// ids of triangles[k] (reference src/render.cc:88-95).
std::vector<uint32_t> sort_faces_by_leaf_order(const Mesh &mesh, const BVH &bvh);
