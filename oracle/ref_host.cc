/*
 * ref_host.cc -- C wrapper around the reference's OWN host objects.
 *
 * TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles this file together with
 * /root/reference/src/{mesh,bvh,aabb,triangle,ray_tracer}.cc (read in place,
 * include dir /root/reference/include) into oracle/_ref/libref_host.so.  It is
 * used in the build container to (a) generate the golden vectors under
 * tests/golden/ and (b) cross-check this repo's own mesh loader / BVH builder /
 * resize.  Nothing here ships: the product never links it.
 */
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "bvh.h"
#include "mesh.h"
#include "ray_tracer.h"

namespace {
struct RefScene {
	Mesh mesh;
	BVH bvh;
	std::vector<uint32_t> sorted_faces;
};
void copy_vec3(const std::vector<Vec3f> &src, float *dst) {
	/* Vec3f is 16 bytes; its 4th lane is indeterminate for most ctors
	 * (reference include/vec3.h:12-22), so emit an explicit 0. */
	for (size_t i = 0; i < src.size(); ++i) {
		dst[4 * i + 0] = src[i][0];
		dst[4 * i + 1] = src[i][1];
		dst[4 * i + 2] = src[i][2];
		dst[4 * i + 3] = 0.0f;
	}
}
}  // namespace

extern "C" {

void *ref_scene_load(const char *path) {
	RefScene *s = new RefScene;
	try {
		load_off_mesh(path, &s->mesh);
		compute_vertex_normals(&s->mesh);
	} catch (...) {
		delete s;
		return nullptr;
	}
	return s;
}

void ref_scene_free(void *h) { delete static_cast<RefScene *>(h); }

uint32_t ref_scene_num_vertices(void *h) { return (uint32_t) static_cast<RefScene *>(h)->mesh.vertices.size(); }
uint32_t ref_scene_num_faces(void *h) { return (uint32_t) (static_cast<RefScene *>(h)->mesh.faces.size() / 3); }

void ref_scene_get_mesh(void *h, float *vertices4, float *normals4, uint32_t *faces) {
	RefScene *s = static_cast<RefScene *>(h);
	copy_vec3(s->mesh.vertices, vertices4);
	copy_vec3(s->mesh.vnormals, normals4);
	std::memcpy(faces, s->mesh.faces.data(), s->mesh.faces.size() * sizeof(uint32_t));
}

/* method: 0 = CUT_LONGEST_AXIS, 1 = SURFACE_AREA_HEURISTIC.  Also performs the
 * leaf-order face sort of reference src/render.cc:88-95.  Returns node count. */
uint32_t ref_scene_build_bvh(void *h, int method) {
	RefScene *s = static_cast<RefScene *>(h);
	s->bvh = BVH(method == 0 ? BVH::Method::CUT_LONGEST_AXIS : BVH::Method::SURFACE_AREA_HEURISTIC);
	s->bvh.buildBVH(s->mesh);
	s->sorted_faces.clear();
	for (size_t i = 0; i < s->bvh.triangles.size(); ++i) {
		const uint32_t f = s->bvh.triangles[i] * 3;
		s->sorted_faces.push_back(s->mesh.faces[f]);
		s->sorted_faces.push_back(s->mesh.faces[f + 1]);
		s->sorted_faces.push_back(s->mesh.faces[f + 2]);
	}
	return (uint32_t) s->bvh.nodes.size();
}

void ref_scene_get_bvh(void *h, uint32_t *nodes, float *aabbs4, uint32_t *triangles, uint32_t *sorted_faces) {
	RefScene *s = static_cast<RefScene *>(h);
	std::memcpy(nodes, s->bvh.nodes.data(), s->bvh.nodes.size() * sizeof(uint32_t));
	copy_vec3(s->bvh.aabbs, aabbs4);
	std::memcpy(triangles, s->bvh.triangles.data(), s->bvh.triangles.size() * sizeof(uint32_t));
	std::memcpy(sorted_faces, s->sorted_faces.data(), s->sorted_faces.size() * sizeof(uint32_t));
}

/* RayTracer::resize of the reference, driven with its own Options/ctor. */
void ref_resize(const float *tmp, uint8_t *image, uint32_t width, uint32_t height, uint32_t n_super_samples) {
	RayTracer::Options o{};
	o.width = width;
	o.height = height;
	o.nSuperSamples = n_super_samples;
	RayTracer rt(o);
	rt.resize(const_cast<float *>(tmp), image);
}

void ref_total_dims(uint32_t width, uint32_t height, uint32_t n_super_samples, uint32_t *tw, uint32_t *th) {
	RayTracer::Options o{};
	o.width = width;
	o.height = height;
	o.nSuperSamples = n_super_samples;
	RayTracer rt(o);
	*tw = rt.totalWidth;
	*th = rt.totalHeight;
}

}  // extern "C"
