// Host-side scene build under AddressSanitizer + UBSan (the reference's Debug
// build enables the same pair, CMakeLists.txt:34-40).  Exercises the OFF loader,
// vertex normals, both BVH strategies, the leaf-order face sort, the device scene
// packer and the host resize on every mesh given on the command line.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "bvh.h"
#include "mesh.h"
#include "ray_tracer.h"
#include "scene_pack.h"
#include "walk_tree.h"

int main(int argc, char **argv) {
	for (int a = 1; a < argc; ++a) {
		Mesh mesh;
		load_off_mesh(argv[a], &mesh);
		compute_vertex_normals(&mesh);
		for (BVH::Method method : { BVH::Method::CUT_LONGEST_AXIS, BVH::Method::SURFACE_AREA_HEURISTIC }) {
			if (method == BVH::Method::SURFACE_AREA_HEURISTIC && mesh.faces.size() / 3 > 20000)
				continue;  // keep the sanitizer run short
			BVH bvh(method);
			bvh.buildBVH(mesh);
			const std::vector<uint32_t> sorted = sort_faces_by_leaf_order(mesh, bvh);
			const ocrt::PackedScene packed = ocrt::pack_scene(sorted, bvh.nodes, bvh.aabbs, mesh.vertices, mesh.vnormals);
			// (the walked tree may have fewer inner nodes than the uploaded one, never fewer leaves)
			size_t leaves = 0;
			for (const ocrt::NodeRec &n : packed.nodes)
				leaves += n.skip == 1;
			if (packed.nodes.size() > bvh.nodes.size() || leaves != mesh.faces.size() / 3 ||
			    packed.nodes[0].skip != packed.nodes.size() || packed.tris.size() != mesh.faces.size() / 3)
				throw std::logic_error("packed sizes");
			// the walked tree: a pre-order skip list whose subtrees nest, every reference leaf exactly once with the
			// uploaded leaf's box, every box inside its parent's, and not more expected box tests than the upload
			{
				const std::vector<ocrt::NodeRec> &t = packed.nodes;
				std::vector<char> seen(mesh.faces.size() / 3, 0);
				std::vector<size_t> open_end, open_at;
				for (size_t i = 0; i < t.size(); ++i) {
					while (!open_end.empty() && open_end.back() <= i) {
						open_end.pop_back();
						open_at.pop_back();
					}
					if (t[i].skip == 0 || i + t[i].skip > t.size() || (!open_end.empty() && i + t[i].skip > open_end.back()))
						throw std::logic_error("walk tree: subtree ranges do not nest");
					if (!open_at.empty())
						for (int k = 0; k < 3; ++k)
							if (!(t[open_at.back()].lo[k] <= t[i].lo[k]) || !(t[i].hi[k] <= t[open_at.back()].hi[k]))
								throw std::logic_error("walk tree: child box outside its parent's");
					if (t[i].skip == 1) {
						if (t[i].leaf >= seen.size() || seen[t[i].leaf]++)
							throw std::logic_error("walk tree: leaf missing or repeated");
					} else {
						if (t[i].leaf != 0xFFFFFFFFu)
							throw std::logic_error("walk tree: inner node carries a leaf");
						open_end.push_back(i + t[i].skip);
						open_at.push_back(i);
					}
				}
				// leaf boxes are the uploaded ones (leaf L of the upload is the L-th node with subtree size 1)
				size_t leaf_number = 0;
				std::vector<size_t> upload_node_of_leaf(seen.size());
				for (size_t i = 0; i < bvh.nodes.size(); ++i)
					if (bvh.nodes[i] == 1)
						upload_node_of_leaf[leaf_number++] = i;
				for (const ocrt::NodeRec &n : t)
					if (n.skip == 1)
						for (int k = 0; k < 3; ++k)
							if (n.lo[k] != bvh.aabbs[2 * upload_node_of_leaf[n.leaf]][k] || n.hi[k] != bvh.aabbs[2 * upload_node_of_leaf[n.leaf] + 1][k])
								throw std::logic_error("walk tree: leaf box differs from the uploaded one");
				setenv("OCRT_KEEP_TREE", "1", 1);
				const ocrt::PackedScene kept = ocrt::pack_scene(sorted, bvh.nodes, bvh.aabbs, mesh.vertices, mesh.vnormals);
				unsetenv("OCRT_KEEP_TREE");
				if (kept.nodes.size() != bvh.nodes.size() || ocrt::tree_cost(t) > ocrt::tree_cost(kept.nodes))
					throw std::logic_error("walk tree: costlier than the uploaded tree");
			}
			// a malformed array must be rejected, not read out of bounds
			std::vector<uint32_t> bad = bvh.nodes;
			if (bad.size() > 2) {
				bad[1] = 0x7FFFFFFFu;
				try {
					ocrt::pack_scene(sorted, bad, bvh.aabbs, mesh.vertices, mesh.vnormals);
					throw std::logic_error("malformed nodes accepted");
				} catch (const std::invalid_argument &) {
				}
			}
		}
		RayTracer::Options o = RayTracer::defaults();
		o.width = 13;
		o.height = 7;
		o.nSuperSamples = 9;
		RayTracer rt(o);
		std::vector<float> tmp((size_t) rt.totalWidth * rt.totalHeight, 0.5f);
		std::vector<unsigned char> img((size_t) o.width * o.height);
		rt.resize(tmp.data(), img.data());
		const std::vector<float> table = ocrt::uniform_ao_table(15, 4, 90);
		std::printf("%s ok (%zu triangles, AO table %zu)\n", argv[a], mesh.faces.size() / 3, table.size() / 4);
	}
	return 0;
}
