// Host-side scene build under AddressSanitizer + UBSan (the reference's Debug
// build enables the same pair, CMakeLists.txt:34-40).  Exercises the OFF loader,
// vertex normals, both BVH strategies, the leaf-order face sort, the device scene
// packer and the host resize on every mesh given on the command line.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "bvh.h"
#include "mesh.h"
#include "ray_tracer.h"
#include "scene_pack.h"

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
