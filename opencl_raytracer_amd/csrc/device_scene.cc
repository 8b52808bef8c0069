// device_scene.cc -- ocrt::DeviceScene (device_renderer.h): one uploaded scene on one GPU, shared by every render host of
// that GPU.
#include <mutex>

#include "device_internal.h"

namespace ocrt {

namespace {
// the allocation made ahead of time by DeviceScene::reserve, waiting for the scene it was made for
std::mutex reserved_mutex;
void *reserved_arena = nullptr;
size_t reserved_bytes = 0;
int reserved_device = -1;

size_t round_up(size_t bytes) { return (bytes + 255) & ~(size_t) 255; }
}  // namespace

std::shared_ptr<const DeviceScene> DeviceScene::create(int device, const PackedScene &scene, const RayTracer::Options &opts, bool for_a_stream) {
	UploadClock clock;
	OCRT_HIP(hipSetDevice(device));
	std::shared_ptr<DeviceScene> out(new DeviceScene());
	out->device_index = device;
	std::vector<float> table;
	out->ao_on = opts.enableAO && opts.aoNumSamples > 0;
	out->ao_method = (int) opts.aoMethod;
	out->ao_samples = opts.aoNumSamples;
	out->ao_alpha_min = opts.aoAlphaMin;
	out->ao_alpha_max = opts.aoAlphaMax;
	if (out->ao_on) {
		if (opts.aoMethod == RayTracer::AmbientOcclusionMethod::UNIFORM) {
			// The reference's order, ring by ring.  (The occlusion count of a hit is a sum over the directions, so the order is
			// free: while a workgroup's four waves took FIXED quarters of a tile's directions the table was dealt round-robin
			// to the quarters so that they cost about the same; with the claim's cursor -- kernels.hip, ao_kernel -- the waves
			// balance themselves, and neighbouring directions cast at the same time are worth 0.5-2 %.)
			table = uniform_ao_table(opts.aoNumSamples, opts.aoAlphaMin, opts.aoAlphaMax);
			out->ao_dirs = (uint32_t) (table.size() / 4);
		} else {
			// RANDOM casts the normal ray plus AO_NUM_SAMPLES + 1 random ones (reference :260-275)
			out->ao_dirs = opts.aoNumSamples + 2;
		}
	}
	out->walk_distance = out->ao_on ? kernel_float(opts.aoMaxDistance) : 0.0f;
	// (a prepared array is taken if it was made for this distance and for at least as much as is wanted now)
	const std::shared_ptr<const WalkArray> made = (scene.walk && scene.walk_max_distance == out->walk_distance && (scene.walk_for_a_stream || !for_a_stream))
	                                                  ? scene.walk
	                                                  : std::make_shared<const WalkArray>(make_walk_array(scene, out->walk_distance, for_a_stream));
	const WalkArray &walk = *made;
	clock.mark("direction table + walk array (made here unless prepared)");
	out->scene_facts_ = scene_facts(scene, walk);
	out->node_count = (uint32_t) scene.nodes.size();
	out->tri_count = (uint32_t) scene.tris.size();
	const size_t nodes_bytes = scene.nodes.size() * sizeof(NodeRec);
	const size_t tris_bytes = scene.tris.size() * sizeof(TriRec);
	const size_t shade_bytes = scene.shade.size() * sizeof(ShadeRec);
	const size_t ao_bytes = table.size() * sizeof(float);
	// One allocation for the five arrays (each starts on a 256-byte boundary) -- the one made ahead of time if there is
	// one and it is large enough.  One node of zero padding behind the exact nodes: the shared walk fetches a node
	// together with its successor.
	const size_t walk_bytes = walk.nodes.size() * sizeof(NodeRec);
	const size_t need = round_up(nodes_bytes + sizeof(NodeRec)) + round_up(walk_bytes) + round_up(tris_bytes) + round_up(shade_bytes) +
	                    round_up(ao_bytes) + 256;
	{
		std::lock_guard<std::mutex> lock(reserved_mutex);
		if (reserved_arena && reserved_device == device && reserved_bytes >= need) {
			out->arena = reserved_arena;
		} else if (reserved_arena && reserved_device == device) {
			(void) hipFree(reserved_arena);  // (too small after all)
		}
		if (reserved_device == device) {
			reserved_arena = nullptr;
			reserved_bytes = 0;
			reserved_device = -1;
		}
	}
	if (!out->arena)
		out->arena = device_alloc(need);
	char *at = (char *) out->arena;
	auto take = [&](size_t bytes) {
		void *p = at;
		at += round_up(bytes);
		return p;
	};
	out->d_nodes = take(nodes_bytes + sizeof(NodeRec));
	out->d_walk = walk_bytes ? take(walk_bytes) : nullptr;
	out->d_tris = take(tris_bytes);
	out->d_shade = take(shade_bytes);
	out->d_ao = take(ao_bytes ? ao_bytes : 1);
	clock.mark("allocation");
	const NodeRec zero{};
	OCRT_HIP(hipMemcpy((char *) out->d_nodes + nodes_bytes, &zero, sizeof zero, hipMemcpyHostToDevice));
	if (walk_bytes)
		OCRT_HIP(hipMemcpy(out->d_walk, walk.nodes.data(), walk_bytes, hipMemcpyHostToDevice));
	OCRT_HIP(hipMemcpy(out->d_nodes, scene.nodes.data(), nodes_bytes, hipMemcpyHostToDevice));
	OCRT_HIP(hipMemcpy(out->d_tris, scene.tris.data(), tris_bytes, hipMemcpyHostToDevice));
	OCRT_HIP(hipMemcpy(out->d_shade, scene.shade.data(), shade_bytes, hipMemcpyHostToDevice));
	if (ao_bytes)
		OCRT_HIP(hipMemcpy(out->d_ao, table.data(), ao_bytes, hipMemcpyHostToDevice));
	OCRT_HIP(hipDeviceSynchronize());
#ifdef OCRT_OCML_BUILTINS
	// (test-only build: the table as the reference kernel's own float trigonometry makes it on this device, kernels.hip)
	if (ao_bytes && opts.aoMethod == RayTracer::AmbientOcclusionMethod::UNIFORM &&
	    ocml_ao_table(out->d_ao, opts.aoNumSamples, opts.aoAlphaMin, opts.aoAlphaMax, out->ao_dirs) != out->ao_dirs)
		throw DeviceError("the device's trigonometry counts another number of ambient-occlusion directions than the host's");
#endif
	clock.mark("copies of walk array, nodes, leaf records, normals, table");
	out->device_bytes = nodes_bytes + walk_bytes + tris_bytes + shade_bytes + ao_bytes;
	return out;
}

DeviceScene::~DeviceScene() {
	if (hipSetDevice(device_index) != hipSuccess)
		return;
	device_free(arena);
}

void DeviceScene::reserve(int device, size_t bytes) {
	if (bytes == 0 || hipSetDevice(device) != hipSuccess)
		return;
	void *p = nullptr;
	if (hipMalloc(&p, bytes) != hipSuccess) {
		(void) hipGetLastError();
		return;  // (create() allocates for itself)
	}
	std::lock_guard<std::mutex> lock(reserved_mutex);
	if (reserved_arena) {
		(void) hipSetDevice(reserved_device);
		(void) hipFree(reserved_arena);
		(void) hipSetDevice(device);
	}
	reserved_arena = p;
	reserved_bytes = bytes;
	reserved_device = device;
}

size_t DeviceScene::bytesFor(size_t triangles, size_t ao_directions) {
	const size_t nodes = triangles ? 2 * triangles - 1 : 0;
	// exact nodes + padding, two copies of the walk records with their END records and slack, leaf records, normals, table
	return round_up((nodes + 1) * sizeof(NodeRec)) + round_up((2 * (nodes + 2) + 2) * sizeof(NodeRec)) + round_up(triangles * sizeof(TriRec)) +
	       round_up(triangles * sizeof(ShadeRec)) + round_up(ao_directions * 4 * sizeof(float) + 1) + 256;
}

bool DeviceScene::servesOptions(const RayTracer::Options &opts) const {
	const bool on = opts.enableAO && opts.aoNumSamples > 0;
	if (on != ao_on)
		return false;
	if (!on)
		return true;
	return (int) opts.aoMethod == ao_method && opts.aoNumSamples == ao_samples && opts.aoAlphaMin == ao_alpha_min &&
	       opts.aoAlphaMax == ao_alpha_max && kernel_float(opts.aoMaxDistance) == walk_distance;
}

}  // namespace ocrt
