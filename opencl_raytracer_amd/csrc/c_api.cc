// c_api.cc -- the extern "C" boundary declared in include/rt_hip.h (the seam), rt_hip_ring.h (streams of frames, several GPUs) and rt_hip_debug.h.
#include "../../include/rt_hip.h"
#include "../../include/rt_hip_debug.h"

#include <algorithm>
#include <hip/hip_runtime.h>

#include <cstring>
#include <memory>
#include <new>
#include <string>

#include "bvh.h"
#include "band_gather.h"
#include "device_renderer.h"
#include "frame_ring.h"
#include "hip_host.h"
#include "mesh.h"
#include "ray_tracer.h"
#include "scene_pack.h"

struct rt_scene {
	Mesh mesh;
	BVH bvh;
	std::vector<uint32_t> sorted_faces;
	bool built = false;
};

struct rt_host {
	std::unique_ptr<ocrt::DeviceRenderer> owned;  // empty for a view of a ring's renderer (rt_ring_host)
	ocrt::DeviceRenderer *dev = nullptr;
};

struct rt_ring {
	std::unique_ptr<ocrt::FrameRing> ring;
	std::vector<rt_host> views;  // rt_ring_host: the ring's renderers as borrowed rt_host handles
};

namespace {

thread_local std::string g_error;
thread_local int g_error_code = RT_OK;

int fail(int code, const std::string &message) {
	g_error = message;
	g_error_code = code;
	return code;
}

// Maps the exception in flight to an RT_E_* code.
int fail_from_exception() {
	try {
		throw;
	} catch (const ocrt::DeviceError &e) {
		return fail(RT_E_DEVICE, e.what());
	} catch (const std::invalid_argument &e) {
		return fail(RT_E_INVALID, e.what());
	} catch (const std::logic_error &e) {
		return fail(RT_E_STATE, e.what());
	} catch (const std::runtime_error &e) {
		const bool no_device = std::strcmp(e.what(), "No device found") == 0;
		return fail(no_device ? RT_E_NO_DEVICE : RT_E_IO, e.what());
	} catch (const std::exception &e) {
		return fail(RT_E_INVALID, e.what());
	} catch (...) {
		return fail(RT_E_INVALID, "unknown error");
	}
}

RayTracer::Options to_options(const rt_options &o) {
	RayTracer::Options r;
	r.width = o.width;
	r.height = o.height;
	r.focalLength = o.focal_length;
	r.nSuperSamples = o.n_super_samples;
	r.enableShading = o.enable_shading != 0;
	r.enableAO = o.enable_ao != 0;
	r.aoMaxDistance = o.ao_max_distance;
	r.aoNumSamples = o.ao_num_samples;
	r.aoMethod = o.ao_method == 0 ? RayTracer::AmbientOcclusionMethod::UNIFORM : RayTracer::AmbientOcclusionMethod::RANDOM;
	r.aoAlphaMin = o.ao_alpha_min;
	r.aoAlphaMax = o.ao_alpha_max;
	r.bvhMethod = o.bvh_method == 0 ? BVH::Method::CUT_LONGEST_AXIS : BVH::Method::SURFACE_AREA_HEURISTIC;
	return r;
}

template <class F> int guarded(F &&body) {
	try {
		body();
		return RT_OK;
	} catch (...) {
		return fail_from_exception();
	}
}

}  // namespace

extern "C" {

const char *rt_last_error(void) { return g_error.c_str(); }
int rt_last_error_code(void) { return g_error_code; }

void rt_options_default(rt_options *out) {
	const RayTracer::Options d = RayTracer::defaults();
	out->width = d.width;
	out->height = d.height;
	out->focal_length = d.focalLength;
	out->n_super_samples = d.nSuperSamples;
	out->enable_shading = d.enableShading;
	out->enable_ao = d.enableAO;
	out->ao_max_distance = d.aoMaxDistance;
	out->ao_num_samples = d.aoNumSamples;
	out->ao_method = 0;
	out->ao_alpha_min = d.aoAlphaMin;
	out->ao_alpha_max = d.aoAlphaMax;
	out->bvh_method = 0;
}

uint32_t rt_total_width(const rt_options *o) { return RayTracer(to_options(*o)).totalWidth; }
uint32_t rt_total_height(const rt_options *o) { return RayTracer(to_options(*o)).totalHeight; }

int rt_resize_cpu(const rt_options *o, const float *tmp, uint8_t *image) {
	if (!o || !tmp || !image)
		return fail(RT_E_INVALID, "null argument");
	if (RayTracer::gridSize(o->n_super_samples) == 0)
		return fail(RT_E_INVALID, "supersample count must be positive");
	RayTracer(to_options(*o)).resize(tmp, image);
	return RT_OK;
}

rt_scene *rt_scene_load_off(const char *path) {
	std::unique_ptr<rt_scene> s(new (std::nothrow) rt_scene);
	if (!s || !path) {
		fail(RT_E_INVALID, "null argument");
		return nullptr;
	}
	const int rc = guarded([&] {
		load_off_mesh(path, &s->mesh);
		compute_vertex_normals(&s->mesh);
	});
	return rc == RT_OK ? s.release() : nullptr;
}

rt_scene *rt_scene_from_arrays(const float *vertices4, uint32_t num_vertices, const uint32_t *faces, uint32_t num_faces) {
	if ((!vertices4 && num_vertices) || (!faces && num_faces)) {
		fail(RT_E_INVALID, "null argument");
		return nullptr;
	}
	std::unique_ptr<rt_scene> s(new (std::nothrow) rt_scene);
	if (!s)
		return nullptr;
	for (uint32_t i = 0; i < 3 * num_faces; ++i)
		if (faces[i] >= num_vertices) {
			fail(RT_E_INVALID, "face references a vertex out of range");
			return nullptr;
		}
	s->mesh.vertices.resize(num_vertices);
	for (uint32_t i = 0; i < num_vertices; ++i)
		s->mesh.vertices[i] = Vec3f(vertices4[4 * i], vertices4[4 * i + 1], vertices4[4 * i + 2]);
	s->mesh.faces.assign(faces, faces + 3 * (size_t) num_faces);
	compute_vertex_normals(&s->mesh);
	return s.release();
}

void rt_scene_free(rt_scene *s) { delete s; }
uint32_t rt_scene_num_vertices(const rt_scene *s) { return (uint32_t) s->mesh.vertices.size(); }
uint32_t rt_scene_num_faces(const rt_scene *s) { return (uint32_t) (s->mesh.faces.size() / 3); }

int rt_scene_build_bvh(rt_scene *s, int method) {
	if (!s)
		return fail(RT_E_INVALID, "null scene");
	return guarded([&] {
		s->built = false;
		s->bvh = BVH(method == 0 ? BVH::Method::CUT_LONGEST_AXIS : BVH::Method::SURFACE_AREA_HEURISTIC);
		s->bvh.buildBVH(s->mesh);
		s->sorted_faces = sort_faces_by_leaf_order(s->mesh, s->bvh);
		s->built = true;
	});
}

uint32_t rt_scene_num_nodes(const rt_scene *s) { return s->built ? (uint32_t) s->bvh.nodes.size() : 0; }
const float *rt_scene_vertices(const rt_scene *s) { return &s->mesh.vertices.data()->x; }
const float *rt_scene_vnormals(const rt_scene *s) { return &s->mesh.vnormals.data()->x; }
const uint32_t *rt_scene_faces(const rt_scene *s) { return s->mesh.faces.data(); }
const uint32_t *rt_scene_nodes(const rt_scene *s) { return s->bvh.nodes.data(); }
const float *rt_scene_aabbs(const rt_scene *s) { return &s->bvh.aabbs.data()->x; }
const uint32_t *rt_scene_triangles(const rt_scene *s) { return s->bvh.triangles.data(); }
const uint32_t *rt_scene_sorted_faces(const rt_scene *s) { return s->sorted_faces.data(); }

rt_host *rt_create_on(const rt_options *o, int device, uint32_t rank, uint32_t nranks) {
	if (!o) {
		fail(RT_E_INVALID, "null options");
		return nullptr;
	}
	std::unique_ptr<rt_host> h(new (std::nothrow) rt_host);
	if (!h)
		return nullptr;
	const int rc = guarded([&] {
		h->owned.reset(new ocrt::DeviceRenderer(to_options(*o), device, rank, nranks));
		h->dev = h->owned.get();
	});
	return rc == RT_OK ? h.release() : nullptr;
}

rt_host *rt_create(const rt_options *o) { return rt_create_on(o, -1, 0, 1); }

void rt_destroy(rt_host *h) {
	if (h && h->owned)  // (a view handed out by rt_ring_host belongs to its ring)
		delete h;
}

int rt_upload(rt_host *h, const uint32_t *faces, uint32_t num_faces, const uint32_t *nodes, uint32_t num_nodes,
              const float *aabbs, const float *vertices, uint32_t num_vertices, const float *vnormals) {
	if (!h || !faces || !nodes || !aabbs || !vertices || !vnormals)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] {
		auto as_vec3 = [](const float *p, size_t n) {
			std::vector<Vec3f> v(n);
			for (size_t i = 0; i < n; ++i)
				v[i] = Vec3f(p[4 * i], p[4 * i + 1], p[4 * i + 2]);
			return v;
		};
		const std::vector<uint32_t> f(faces, faces + 3 * (size_t) num_faces);
		const std::vector<uint32_t> n(nodes, nodes + num_nodes);
		h->dev->upload(ocrt::pack_scene(f, n, as_vec3(aabbs, 2 * (size_t) num_nodes), as_vec3(vertices, num_vertices),
		                                as_vec3(vnormals, num_vertices)));
	});
}

int rt_upload_scene(rt_host *h, const rt_scene *s) {
	if (!h || !s)
		return fail(RT_E_INVALID, "null argument");
	if (!s->built)
		return fail(RT_E_STATE, "scene has no BVH yet (call rt_scene_build_bvh)");
	return guarded([&] {
		h->dev->upload(ocrt::pack_scene(s->sorted_faces, s->bvh.nodes, s->bvh.aabbs, s->mesh.vertices, s->mesh.vnormals));
	});
}

int rt_render(rt_host *h) {
	if (!h)
		return fail(RT_E_INVALID, "null host");
	return guarded([&] {
		h->dev->enqueueRender();
		h->dev->synchronize();
	});
}

int rt_render_async(rt_host *h) {
	if (!h)
		return fail(RT_E_INVALID, "null host");
	return guarded([&] { h->dev->enqueueRender(); });
}

int rt_sync(rt_host *h) {
	if (!h)
		return fail(RT_E_INVALID, "null host");
	return guarded([&] { h->dev->synchronize(); });
}

int rt_download(rt_host *h, float *image) {
	if (!h || !image)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { h->dev->downloadFloat(image); });
}

int rt_download_u8(rt_host *h, uint8_t *image) {
	if (!h || !image)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { h->dev->downloadResizedFull(image); });
}

uint32_t rt_local_rows(const rt_host *h) { return h ? h->dev->localRows() : 0; }

int rt_download_u8_local(rt_host *h, uint8_t *rows) {
	if (!h || !rows)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { h->dev->downloadResizedLocal(rows); });
}

uint32_t rt_partition_local_rows(const rt_options *o, uint32_t rank, uint32_t nranks) {
	const uint32_t n = RayTracer::gridSize(o->n_super_samples);
	if (n == 0 || nranks == 0 || rank >= nranks)
		return 0;
	const ocrt::Partition part{ rank, nranks, ocrt::band_tile_rows_for(n) };
	return ocrt::local_tile_rows_for(o->height * n, part) * ocrt::TILE_H / n;
}

uint32_t rt_partition_global_row(const rt_options *o, uint32_t rank, uint32_t nranks, uint32_t local_row) {
	const uint32_t n = RayTracer::gridSize(o->n_super_samples);
	if (n == 0 || nranks == 0)
		return 0xFFFFFFFFu;
	const uint32_t rows_per_band = ocrt::band_tile_rows_for(n) * ocrt::TILE_H / n;
	const uint32_t band_local = local_row / rows_per_band;
	return (band_local * nranks + rank) * rows_per_band + local_row % rows_per_band;
}

uint32_t rt_local_to_global_row(const rt_host *h, uint32_t local_row) {
	if (!h)
		return 0xFFFFFFFFu;
	const ocrt::KernelParams &p = h->dev->params();
	const RayTracer::Options &ro = h->dev->rayTracer().options;
	rt_options o{};
	o.n_super_samples = ro.nSuperSamples;
	return rt_partition_global_row(&o, p.part.rank, p.part.nranks, local_row);
}

int rt_resize_into_device(rt_host *h, void *device_u8) {
	if (!h || !device_u8)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { h->dev->enqueueResizeInto(device_u8); });
}

int rt_set_stream(rt_host *h, void *hip_stream) {
	if (!h)
		return fail(RT_E_INVALID, "null host");
	return guarded([&] { h->dev->setStream(hip_stream); });
}

int rt_get_stream(rt_host *h, void **hip_stream) {
	if (!h || !hip_stream)
		return fail(RT_E_INVALID, "null argument");
	*hip_stream = h->dev->streamHandle();
	return RT_OK;
}

int rt_set_device_share(rt_host *h, unsigned int hosts) {
	if (!h)
		return fail(RT_E_INVALID, "null host");
	h->dev->setDeviceShare(hosts);
	return RT_OK;
}

int rt_expect_frames(rt_host *h, uint64_t frames) {
	if (!h)
		return fail(RT_E_INVALID, "null host");
	return guarded([&] { h->dev->expectFrames(frames); });
}

int rt_use_private_stream(rt_host *h) {
	if (!h)
		return fail(RT_E_INVALID, "null host");
	return guarded([&] { h->dev->usePrivateStream(); });
}

int rt_get_stats(rt_host *h, rt_stats *out) {
	if (!h || !out)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] {
		const ocrt::RenderStats s = h->dev->stats();
		out->primary_rays = s.primary_rays;
		out->primary_hits = s.primary_hits;
		out->ao_rays = s.ao_rays;
		out->ao_occluded = s.ao_occluded;
	});
}

float rt_last_kernel_ms(const rt_host *h) { return h ? h->dev->lastKernelMs() : 0.0f; }
double rt_total_kernel_ms(const rt_host *h) { return h ? h->dev->totalKernelMs() : 0.0; }
float rt_last_ao_ms(const rt_host *h) { return h ? h->dev->lastAoMs() : 0.0f; }
double rt_total_ao_ms(const rt_host *h) { return h ? h->dev->totalAoMs() : 0.0; }
uint64_t rt_kernel_launches(const rt_host *h) { return h ? h->dev->kernelLaunches() : 0; }
void rt_reset_timers(rt_host *h) {
	if (h)
		h->dev->resetTimers();
}

/* ---- frame ring ------------------------------------------------------------------------------------------------ */
rt_ring *rt_ring_create(const rt_options *o, int device, uint32_t rank, uint32_t nranks, uint32_t hosts) {
	if (!o) {
		fail(RT_E_INVALID, "null options");
		return nullptr;
	}
	std::unique_ptr<rt_ring> r(new (std::nothrow) rt_ring);
	if (!r)
		return nullptr;
	const int rc = guarded([&] {
		r->ring.reset(new ocrt::FrameRing(to_options(*o), device, rank, nranks, hosts));
		r->views.resize(r->ring->size());
		for (uint32_t k = 0; k < r->ring->size(); ++k)
			r->views[k].dev = &r->ring->host(k);
	});
	return rc == RT_OK ? r.release() : nullptr;
}

void rt_ring_destroy(rt_ring *r) { delete r; }

int rt_ring_upload(rt_ring *r, const uint32_t *faces, uint32_t num_faces, const uint32_t *nodes, uint32_t num_nodes,
                   const float *aabbs, const float *vertices, uint32_t num_vertices, const float *vnormals) {
	if (!r || !faces || !nodes || !aabbs || !vertices || !vnormals)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] {
		auto as_vec3 = [](const float *p, size_t n) {
			std::vector<Vec3f> v(n);
			for (size_t i = 0; i < n; ++i)
				v[i] = Vec3f(p[4 * i], p[4 * i + 1], p[4 * i + 2]);
			return v;
		};
		const std::vector<uint32_t> f(faces, faces + 3 * (size_t) num_faces);
		const std::vector<uint32_t> n(nodes, nodes + num_nodes);
		r->ring->upload(ocrt::pack_scene(f, n, as_vec3(aabbs, 2 * (size_t) num_nodes), as_vec3(vertices, num_vertices),
		                                 as_vec3(vnormals, num_vertices)));
	});
}

int rt_ring_upload_scene(rt_ring *r, const rt_scene *s) {
	if (!r || !s)
		return fail(RT_E_INVALID, "null argument");
	if (!s->built)
		return fail(RT_E_STATE, "scene has no BVH yet (call rt_scene_build_bvh)");
	return guarded([&] {
		r->ring->upload(ocrt::pack_scene(s->sorted_faces, s->bvh.nodes, s->bvh.aabbs, s->mesh.vertices, s->mesh.vnormals));
	});
}

int rt_ring_device_bytes(const rt_ring *r, uint64_t *scene_bytes, uint32_t *scene_copies, uint64_t *total_bytes) {
	if (!r)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] {
		std::vector<const ocrt::DeviceScene *> seen;
		for (uint32_t k = 0; k < r->ring->size(); ++k) {
			const ocrt::DeviceScene *s = r->ring->host(k).deviceScene().get();
			if (s && std::find(seen.begin(), seen.end(), s) == seen.end())
				seen.push_back(s);
		}
		if (scene_bytes)
			*scene_bytes = seen.empty() ? 0 : seen.front()->bytes();
		if (scene_copies)
			*scene_copies = (uint32_t) seen.size();
		if (total_bytes)
			*total_bytes = r->ring->uploadedBytes();
	});
}

int rt_ring_set_calibration(rt_ring *r, int on) {
	if (!r)
		return fail(RT_E_INVALID, "null argument");
	r->ring->setCalibration(on != 0);
	return RT_OK;
}

int rt_ring_calibration(const rt_ring *r, float *ms_without, float *ms_with, int *prefetch_in_use) {
	if (!r)
		return fail(RT_E_INVALID, "null argument");
	if (ms_without)
		*ms_without = r->ring->calibrationMs()[0];
	if (ms_with)
		*ms_with = r->ring->calibrationMs()[1];
	if (prefetch_in_use)
		*prefetch_in_use = r->ring->aoPrefetch() ? 1 : 0;
	return RT_OK;
}

int rt_walk_entries(rt_host *h, uint32_t *tiles_hit, uint32_t *tiles_narrowed, double *mean_share, double *mean_packet_share) {
	if (!h)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] {
		const ocrt::DeviceRenderer::WalkEntries e = h->dev->walkEntries();
		if (tiles_hit)
			*tiles_hit = e.tiles_hit;
		if (tiles_narrowed)
			*tiles_narrowed = e.tiles_narrowed;
		if (mean_share)
			*mean_share = e.mean_share;
		if (mean_packet_share)
			*mean_packet_share = e.mean_packet_share;
	});
}

int rt_set_ao_prefetch(rt_host *h, int on) {
	if (!h)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { h->dev->setAoPrefetch(on != 0); });
}

// ---- include/rt_hip_debug.h ----
int rt_debug_measure_tile_costs(rt_host *h, uint32_t frames, int reorder) {
	if (!h)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] {
		if (reorder) {
			h->dev->measureTileCosts(frames);
			return;
		}
		std::vector<uint32_t> order, constants, words;
		std::vector<float> cost;
		h->dev->tileOrder(order, constants, words, cost);
		h->dev->measureTileCosts(frames);      // (measures and reorders ...)
		h->dev->setTileOrder(order, constants);  // (... and the list that was in place goes back)
	});
}

int rt_debug_set_primary_split(rt_host *h, uint32_t above) {
	if (!h)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { h->dev->setPrimarySplit(above); });
}

int rt_debug_set_order_policy(rt_host *h, float heavy, float runway, float split_above) {
	if (!h)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { h->dev->setOrderPolicy(heavy, runway, split_above); });
}

uint32_t rt_debug_tile_order_slots(rt_host *h) {
	if (!h)
		return 0;
	std::vector<uint32_t> order, constants, words;
	std::vector<float> cost;
	h->dev->tileOrder(order, constants, words, cost);
	return (uint32_t) order.size();
}

uint32_t rt_debug_tiles(rt_host *h) {
	if (!h)
		return 0;
	const ocrt::KernelParams &kp = h->dev->params();
	return kp.tiles_x * kp.local_tile_rows;
}

int rt_debug_tile_order(rt_host *h, uint32_t *order_out, uint32_t *constants24, uint32_t *tile_words, float *tile_costs) {
	if (!h)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] {
		std::vector<uint32_t> order, constants, words;
		std::vector<float> cost;
		h->dev->tileOrder(order, constants, words, cost);
		if (order_out)
			std::copy(order.begin(), order.end(), order_out);
		if (constants24)
			std::copy(constants.begin(), constants.end(), constants24);
		const uint32_t tiles = rt_debug_tiles(h);
		if (tile_words)
			for (uint32_t t = 0; t < tiles; ++t)
				tile_words[t] = t < words.size() ? words[t] : 0u;
		if (tile_costs)
			for (uint32_t t = 0; t < tiles; ++t)
				tile_costs[t] = t < cost.size() ? cost[t] : 0.0f;
	});
}

int rt_debug_set_tile_order(rt_host *h, const uint32_t *order, uint32_t slots, const uint32_t *constants24) {
	if (!h || !order || !constants24)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] {
		h->dev->setTileOrder(std::vector<uint32_t>(order, order + slots), std::vector<uint32_t>(constants24, constants24 + 24));
	});
}

int rt_debug_set_frame_form(rt_host *h, int form) {
	if (!h)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { h->dev->setFrameForm(form); });
}

int rt_debug_frame_is_fused(rt_host *h) { return h && h->dev->frameIsFused() ? 1 : 0; }

int rt_debug_poison_hit_list(rt_host *h) {
	if (!h)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { h->dev->poisonHitList(); });
}

uint32_t rt_ring_size(const rt_ring *r) { return r ? r->ring->size() : 0; }
uint32_t rt_ring_slots(const rt_ring *r) { return r ? r->ring->slots() : 0; }
uint32_t rt_ring_local_rows(const rt_ring *r) { return r ? r->ring->localRows() : 0; }
uint32_t rt_ring_in_flight(const rt_ring *r) { return r ? r->ring->inFlight() : 0; }
rt_host *rt_ring_host(rt_ring *r, uint32_t slot) { return (r && slot < r->views.size()) ? &r->views[slot] : nullptr; }

int rt_ring_set_graph_mode(rt_ring *r, int on) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] { r->ring->setGraphMode(on != 0); });
}

int rt_ring_set_pacing(rt_ring *r, float beta) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	r->ring->setPacing(beta);
	return RT_OK;
}

int rt_ring_bind_output(rt_ring *r, uint32_t slot, void *device_u8) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] { r->ring->bindOutput(slot, device_u8); });
}

int rt_ring_submit(rt_ring *r, uint64_t *frame) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] {
		const uint64_t f = r->ring->submit();
		if (frame)
			*frame = f;
	});
}

int rt_ring_collect(rt_ring *r, uint64_t *frame, uint32_t *slot, const void **device_bands) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] {
		const ocrt::FrameRing::Collected c = r->ring->collect();
		if (frame)
			*frame = c.frame;
		if (slot)
			*slot = c.slot;
		if (device_bands)
			*device_bands = c.device_bands;
	});
}

int rt_ring_collect_into_device(rt_ring *r, void *device_u8) {
	if (!r || !device_u8)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] {
		const ocrt::FrameRing::Collected c = r->ring->collect();
		ocrt::DeviceRenderer &h = r->ring->host((uint32_t) (c.frame % r->ring->size()));
		if (hipSetDevice(h.deviceIndex()) != hipSuccess ||
		    hipMemcpyAsync(device_u8, c.device_bands, (size_t) h.localRows() * h.width(), hipMemcpyDeviceToDevice,
		                   (hipStream_t) h.streamHandle()) != hipSuccess ||
		    hipStreamSynchronize((hipStream_t) h.streamHandle()) != hipSuccess)
			throw ocrt::DeviceError("copying a collected frame's bands failed");
	});
}

int rt_ring_step(rt_ring *r) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] { r->ring->step(); });
}

int rt_ring_run(rt_ring *r, uint32_t frames) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] {
		for (uint32_t k = 0; k < frames; ++k)
			r->ring->step();
	});
}

int rt_ring_drain(rt_ring *r) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] { r->ring->drain(); });
}

int rt_ring_last_image_device(rt_ring *r, const void **device_u8) {
	if (!r || !device_u8)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { *device_u8 = r->ring->lastImageDevice(); });
}

int rt_ring_download_last(rt_ring *r, uint8_t *image) {
	if (!r || !image)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { r->ring->downloadLast(image); });
}

int rt_ring_reset_clock(rt_ring *r) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] { r->ring->resetClock(); });
}

int rt_ring_keep_frame_times(rt_ring *r, int on) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] { r->ring->keepFrameTimes(on != 0); });
}

int rt_ring_frame_times(const rt_ring *r, uint64_t frame, float ms[4]) {
	if (!r || !ms)
		return fail(RT_E_INVALID, "null argument");
	return r->ring->frameTimes(frame, ms) ? RT_OK : fail(RT_E_STATE, "no time stamps kept for that frame");
}

int rt_ring_timers(rt_ring *r, double *kernel_ms, uint64_t *frames, double *ao_ms, uint64_t *ao_frames) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	double k = 0, a = 0;
	uint64_t nk = 0, na = 0;
	for (uint32_t s = 0; s < r->ring->size(); ++s) {
		const ocrt::DeviceRenderer &h = r->ring->host(s);
		k += h.totalKernelMs();
		a += h.totalAoMs();
		nk += h.kernelLaunches();
		na += h.aoLaunches();
	}
	if (kernel_ms)
		*kernel_ms = k;
	if (frames)
		*frames = nk;
	if (ao_ms)
		*ao_ms = a;
	if (ao_frames)
		*ao_frames = na;
	return RT_OK;
}

int rt_ring_cpu_times(const rt_ring *r, double *submit_s, double *wait_s, double *collect_s, uint64_t *frames) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	const ocrt::FrameRing::CpuTimes &t = r->ring->cpuTimes();
	if (submit_s)
		*submit_s = t.submit_s;
	if (wait_s)
		*wait_s = t.wait_s;
	if (collect_s)
		*collect_s = t.collect_s;
	if (frames)
		*frames = t.frames;
	return RT_OK;
}

void rt_ring_reset_timers(rt_ring *r) {
	if (r)
		for (uint32_t s = 0; s < r->ring->size(); ++s)
			r->ring->host(s).resetTimers();
}

int rt_rccl_available(void) { return ocrt::rccl_available() ? 1 : 0; }

int rt_rccl_unique_id(void *out128) {
	if (!out128)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] { ocrt::rccl_unique_id(out128); });
}

int rt_ring_attach_rccl(rt_ring *r, const void *unique_id128) {
	if (!r || !unique_id128)
		return fail(RT_E_INVALID, "null argument");
	return guarded([&] {
		ocrt::DeviceRenderer &h = r->ring->host(0);
		const ocrt::Partition &part = h.params().part;
		r->ring->attachGather(std::unique_ptr<ocrt::BandGather>(new ocrt::BandGather(
		    h.rayTracer().options, part.rank, part.nranks, h.deviceIndex(), unique_id128, r->ring->slots())));
	});
}

int rt_ring_set_gather_timeout(rt_ring *r, double seconds) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] {
		if (!r->ring->hasGather())
			throw std::logic_error("no communicator attached (rt_ring_attach_rccl)");
		r->ring->gatherOrNull()->setTimeout(seconds);
	});
}

int rt_ring_rccl_info(rt_ring *r, int *comm_ranks, int *rccl_version) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] {
		if (!r->ring->hasGather())
			throw std::logic_error("no communicator attached (rt_ring_attach_rccl)");
		r->ring->gatherOrNull()->describe(comm_ranks, rccl_version);
	});
}

int rt_ring_rccl_self_test(rt_ring *r) {
	if (!r)
		return fail(RT_E_INVALID, "null ring");
	return guarded([&] {
		if (!r->ring->hasGather())
			throw std::logic_error("no communicator attached (rt_ring_attach_rccl)");
		r->ring->gatherSelfTest();
	});
}

void rt_print_info(void) { HipHost::printInfo(); }
int rt_device_count(void) { return ocrt::visible_device_count(); }

}  // extern "C"
