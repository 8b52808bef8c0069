#include "device_internal.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sstream>

namespace ocrt {


int visible_device_count() {
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess)
		return 0;
	return count;
}

void warm_up_device(int device) {
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
		return;
	if (device < 0) {
		const char *env = std::getenv("OCRT_DEVICE");
		device = env ? std::atoi(env) : 0;
	}
	if (device >= count || hipSetDevice(device) != hipSuccess)
		return;
	(void) hipFree(nullptr);  // (forces the context)
	preload_kernels();
	void *p = nullptr;        // the allocator's first call is slow too, and so are the first copies in either direction
	if (hipMalloc(&p, 1 << 20) == hipSuccess) {  // (the runtime sets up its staging buffers then: 5-10 ms that an upload would pay)
		std::vector<unsigned char> staging(1 << 16, 0);
		(void) hipMemcpy(p, staging.data(), staging.size(), hipMemcpyHostToDevice);
		(void) hipMemcpy(staging.data(), p, staging.size(), hipMemcpyDeviceToHost);
		(void) hipFree(p);
	}
	(void) hipGetLastError();
}

DeviceRenderer::DeviceRenderer(const RayTracer::Options &options, int device_, unsigned int rank, unsigned int nranks,
                               int ring_slot, bool spare_top_class)
	: opts(options)
	, rt(options)
	, device(device_)
	, grid(RayTracer::gridSize(options.nSuperSamples))
	, local_out_rows(0)
	, own_stream(nullptr)
	, stream(nullptr)
	, d_image(nullptr)
	, d_u8(nullptr)
	, d_hits(nullptr)
	, d_occluded(nullptr)
	, d_tile_hits(nullptr)
	, d_tile_base(nullptr)
	, d_tile_entry(nullptr)
	, d_order(nullptr)
	, d_counters(nullptr)
	, hit_slots(0)
	, image_bytes(0)
	, tile_count(0)
	, compute_units(0)
	, device_share(1)
	, scene_ready(false)
	, frame_ready(false)
	, graph_mode(true)
	, ao_prefetch(true)
	, scene_version(0)
	, ao_blocks_override(0)
	, epoch_event(nullptr)
	, keep_stamps(false)
	, last_times{ 0, 0, 0, 0 }
	, last_ms(0)
	, last_ao_ms(0)
	, total_ms(0)
	, total_ao_ms(0)
	, launches(0)
	, ao_launches(0) {
	if (nranks == 0 || rank >= nranks)
		throw std::invalid_argument("rank must be < nranks");
	if (opts.width == 0 || opts.height == 0 || grid == 0)
		throw std::invalid_argument("image width, height and supersample count must be positive");
	if ((unsigned long long) rt.totalWidth * rt.totalHeight >= (1ull << 32))
		throw std::invalid_argument("supersampled image too large for 32-bit pixel indices");
	const int count = visible_device_count();
	if (count <= 0)
		throw std::runtime_error("No device found");
	if (device < 0) {
		const char *env = std::getenv("OCRT_DEVICE");
		device = env ? std::atoi(env) : 0;
	}
	if (device >= count)
		throw std::invalid_argument("HIP device index out of range");
	part.rank = rank;
	part.nranks = nranks;
	part.band_tile_rows = band_tile_rows_for(grid);
	kp = make_kernel_params(rt, 0, 0, 0, part, nullptr);
	local_out_rows = kp.local_tile_rows * TILE_H / grid;
	tile_count = (size_t) kp.tiles_x * kp.local_tile_rows;

	useDevice();
	// Renderers that take frames in turn on one GPU (FrameRing) get streams of DIFFERENT priority.  HIP maps streams
	// onto a few hardware queues per priority class, and a hardware queue runs its packets in order: two renderers of
	// one scene whose frames are meant to share the device -- the last workgroups of one frame's ambient-occlusion
	// pass, the next frame's primary pass -- must not land in the same queue, which is what happened with equal
	// priorities once torch.distributed had created its streams (rocprofv3 --kernel-trace: every kernel of both
	// renderers in one queue, one after the other).  Different classes never share a queue.  The device offers
	// `least - greatest + 1` classes (three on this GPU); slot k of a ring takes class k modulo that, so CONSECUTIVE
	// renderers always differ -- which is what matters, a frame overlaps with its neighbours in the ring --, while
	// slots a whole cycle apart share a class and may share a queue.  A renderer on its own (slot -1) takes the lowest
	// class: whatever else the process runs on the device (a collective's kernels) is not held up by it.
	int least = 0, greatest = 0;
	OCRT_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
	// With an exchange step on the device (a multi-GPU job's ring) the HIGHEST class is the gather stream's alone
	// (band_gather.cc): an RCCL receive waiting for a slower peer must not sit in a hardware queue ahead of a renderer's
	// graph launches, nor the gather behind a whole frame; the renderers alternate over the classes below it.
	const int classes = least - greatest + 1 - (spare_top_class && least - greatest + 1 > 2 ? 1 : 0);
	const int priority = ring_slot < 0 || classes < 2 ? least : least - ring_slot % classes;
	hipStream_t s;
	OCRT_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
	own_stream = stream = s;
	// Float image and uint8 buffer both hold this rank's bands only, back to back
	// (whole tile rows, so the last band may run past the image's height).
	image_bytes = (size_t) kp.local_tile_rows * TILE_H * rt.totalWidth * sizeof(float);
	d_image = device_alloc(image_bytes);
	d_u8 = device_alloc((size_t) local_out_rows * opts.width);
	// hit list: 64 slots per tile; ordered tile lists: one segment per XCD group
	const size_t order_slots = (size_t) ((kp.tiles_x + MAX_STRIP_TILES - 1) / MAX_STRIP_TILES) * MAX_STRIP_TILES * kp.local_tile_rows;  // (whole strips, whatever their width)
	// (the hit list itself is sized once a scene is adopted: by what that scene's primary rays hit, sizeHitList)
	d_tile_hits = device_alloc(tile_count * sizeof(uint32_t));
	d_tile_base = device_alloc(tile_count * sizeof(uint32_t));
	d_order = device_alloc(order_slots * sizeof(uint32_t));
	d_order_need = device_alloc((order_slots ? order_slots : 1) * sizeof(uint32_t));
	{  // (2 x 2 tile blocks: whole strips x pairs of rows)
		const size_t padded_x = (size_t) ((kp.tiles_x + MAX_STRIP_TILES - 1) / MAX_STRIP_TILES) * MAX_STRIP_TILES;
		d_primary_order = device_alloc((padded_x / 2 * ((kp.local_tile_rows + 1) / 2) + 1) * sizeof(uint32_t));
		blocks_by_cost_capacity = padded_x / 2 * ((kp.local_tile_rows + 1) / 2) + 8;
		d_blocks_by_cost = device_alloc(blocks_by_cost_capacity * sizeof(uint32_t));
	}
	d_tile_ready = device_alloc((tile_count ? tile_count : 1) * sizeof(uint32_t));
	OCRT_HIP(hipMemsetAsync(d_tile_ready, 0, (tile_count ? tile_count : 1) * sizeof(uint32_t), (hipStream_t) stream));
	d_counters = device_alloc(sizeof(FrameCounters));
	OCRT_HIP(hipMemsetAsync(d_counters, 0, sizeof(FrameCounters), (hipStream_t) stream));  // (once: the kernels keep it clean, device_types.h)
	{
		hipDeviceProp_t prop;
		OCRT_HIP(hipGetDeviceProperties(&prop, device));
		compute_units = (uint32_t) prop.multiProcessorCount;
	}
	OCRT_HIP(hipMemsetAsync(d_image, 0, image_bytes, (hipStream_t) stream));
	OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
#ifdef OCRT_DEBUG_KNOBS
	if (const char *env = std::getenv("OCRT_AO_BLOCKS"))  // workgroups of the persistent ambient-occlusion pass
		ao_blocks_override = (uint32_t) std::atoi(env);
	if (const char *env = std::getenv("OCRT_PRIMARY_BY_COST"))  // 0: the primary pass's workgroups take their blocks in spatial order
		primary_by_cost = std::atoi(env) != 0;
#endif
}

DeviceRenderer::~DeviceRenderer() {
	if (hipSetDevice(device) != hipSuccess)
		return;
	if (stream)
		(void) hipStreamSynchronize((hipStream_t) stream);
	dropFrameGraphs();
	for (auto &ev : pending_events)
		free_events.push_back(ev);
	for (auto &ev : free_events) {
		(void) hipEventDestroy((hipEvent_t) ev.start);
		(void) hipEventDestroy((hipEvent_t) ev.ao_start);
		(void) hipEventDestroy((hipEvent_t) ev.ao_stop);
		(void) hipEventDestroy((hipEvent_t) ev.stop);
	}
	freeScene();
	device_free(d_image);
	device_free(d_u8);
	device_free(d_hits);
	device_free(d_occluded);
	device_free(d_tile_hits);
	device_free(d_tile_base);
	device_free(d_order);
	device_free(d_primary_order);
	device_free(d_order_need);
	device_free(d_blocks_by_cost);
	device_free(d_tile_ready);
	device_free(d_counters);
	if (own_stream)
		(void) hipStreamDestroy((hipStream_t) own_stream);
}

void DeviceRenderer::useDevice() const { OCRT_HIP(hipSetDevice(device)); }

void DeviceRenderer::freeScene() {
	scene_on_device.reset();  // (the arrays go when the last renderer lets go of them)
	scene_ready = false;
}

std::string DeviceRenderer::deviceName() const {
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess)
		return "unknown";
	return prop.name;
}

size_t DeviceRenderer::upload(const PackedScene &scene) {
	useDevice();
	synchronize();
	freeScene();
	std::shared_ptr<const DeviceScene> made = DeviceScene::create(device, scene, opts, expected_frames >= FRAMES_WORTH_INTERVALS);
	const size_t scene_bytes = made->bytes();
	return scene_bytes + adopt(std::move(made));
}

size_t DeviceRenderer::adopt(std::shared_ptr<const DeviceScene> scene, const DeviceRenderer *layout_from) {
	if (!scene || scene->device() != device)
		throw std::invalid_argument("the scene lives on another device than the renderer");
	if (!scene->servesOptions(opts))
		throw std::invalid_argument("the scene on the device was made for other ambient-occlusion options than the renderer's");
	useDevice();
	synchronize();
	freeScene();
	kp = make_kernel_params(rt, scene->nodeCount(), scene->triCount(), scene->aoDirs(), part, &scene->facts());
	kp.shared_device = device_share > 1u ? 1 : 0;
	// How wide the strips are that the image is dealt to the eight XCD groups in (kernels.hip, "Tile <-> workgroup
	// mapping").  Two tiles while the scene fits the caches -- the finest deal balances best.  A scene several times the
	// 32 MB of L2 is fetched from HBM by every XCD whose rays reach it: an ambient-occlusion ray reaches AO_MAX_DISTANCE
	// around its tile, twenty tiles of a 1080p frame on either side, so with 16-pixel strips all eight XCDs read the same
	// nodes.  Strips of sixteen tiles keep most of an XCD's geometry its own (terrain, 2 M and 20 M triangles, 1080p:
	// profiles/r04_notes.md) -- as long as the image is wide enough for every group to keep two strips.
	// The walk intervals (kernels.hip, entry_kernel): one per tile and table direction -- unless that table would be out of
	// proportion (many samples per pixel AND many directions: 64 x 301 at 1080p would be 5 GB), then the tiles' own only.
	// ... and for a caller that has not announced enough frames for the table to pay (expectFrames): one whole-array
	// interval per tile, nothing to compute.
	if (kp.ao_mode != AO_UNIFORM || (size_t) tile_count * (1 + (size_t) kp.ao_dirs) * 2 * sizeof(uint32_t) > MAX_ENTRY_TABLE_BYTES ||
	    expected_frames < FRAMES_WORTH_INTERVALS)
		kp.entry_stride = 1u;
#ifdef OCRT_DEBUG_KNOBS
	if (std::getenv("OCRT_ENTRY_PER_TILE"))  // the tiles' own intervals only (what a frame with too large a table gets)
		kp.entry_stride = 1u;
#endif
	kp.strip_tiles = 2u;
	scene_beyond_caches = scene->bytes() > BIG_SCENE_BYTES;
	if (scene_beyond_caches)
		while (kp.strip_tiles < MAX_STRIP_TILES && kp.tiles_x >= 2u * XCD_GROUPS * (kp.strip_tiles * 2u))
			kp.strip_tiles *= 2u;
#ifdef OCRT_DEBUG_KNOBS
	if (const char *env = std::getenv("OCRT_STRIP_TILES")) {  // 2, 4, 8, 16 or 32
		const uint32_t want = (uint32_t) std::atoi(env);
		if (want >= 2u && want <= MAX_STRIP_TILES && (want & (want - 1u)) == 0u)
			kp.strip_tiles = want;
	}
#endif
	scene_on_device = std::move(scene);
	scene_ready = true;
	++scene_version;  // (a captured frame bakes the scene's pointers and launch constants in)
	if (layout_from && (layout_from == this || layout_from->scene_on_device != scene_on_device || layout_from->device != device ||
	                    layout_from->tile_count != tile_count || layout_from->part.rank != part.rank ||
	                    layout_from->part.nranks != part.nranks || layout_from->rt.totalWidth != rt.totalWidth ||
	                    layout_from->rt.totalHeight != rt.totalHeight || layout_from->kp.entry_stride != kp.entry_stride ||
	                    (layout_from->expected_frames >= FRAMES_WORTH_INTERVALS) != (expected_frames >= FRAMES_WORTH_INTERVALS)))
		layout_from = nullptr;  // (not the same frame after all: count)
	sizeHitList(layout_from);
	return image_bytes + (size_t) local_out_rows * opts.width + hit_slots * (sizeof(HitRec) + sizeof(uint32_t)) +
	       tile_count * 3 * sizeof(uint32_t) + (tile_entry_owner.use_count() == 1 ? entryBytes() : 0) + sizeof(FrameCounters);
}

void DeviceRenderer::allocEntries(size_t bytes) {
	const int on_device = device;
	tile_entry_owner = std::shared_ptr<void>(device_alloc(bytes), [on_device](void *p) {
		if (p && hipSetDevice(on_device) == hipSuccess)
			(void) hipFree(p);
	});
	d_tile_entry = tile_entry_owner.get();
}

// The hit list is sized by what is hit.  Camera, scene and options are fixed for this renderer, so the number of hit
// sub-pixels per tile is the same in every frame: ONE pass of the primary kernel without a hit list (it then only counts)
// gives the tiles' hit counts, their exclusive prefix sum is where each tile's hits start (tile_base), and the list gets
// exactly that many slots -- 36 bytes per hit sub-pixel instead of 36 x 64 per tile: 42 MB instead of 75 for the 1080p
// bunny frame, 2.7 GB instead of 4.8 at 64 samples per pixel.  No allocation atomics between the passes: a tile's slots
// are still at a fixed address.  Costs one primary pass (0.15 ms at 1080p) and two small copies per upload.
void DeviceRenderer::sizeHitList(const DeviceRenderer *layout_from) {
	UploadClock clock;
	device_free(d_hits);
	device_free(d_occluded);
	tile_entry_owner.reset();  // (the table goes when the last renderer that walks by it lets go)
	d_hits = d_occluded = d_tile_entry = nullptr;
	hit_slots = 0;
	const bool has_ao = kp.ao_mode != AO_NONE && kp.ao_dirs > 0;
	if (tile_count == 0 || !has_ao) {
		blocks_by_cost_host.clear();  // (no pass has counted anything: the primary pass keeps its spatial mapping)
		order_host.clear();
		primary_order_host.clear();
		d_hits = device_alloc(sizeof(HitRec));  // (never read: no sub-pixel is left pending)
		d_occluded = device_alloc(sizeof(uint32_t));
		allocEntries(sizeof(uint32_t) * 2);
		OCRT_HIP(hipMemsetAsync(d_tile_base, 0, tile_count * sizeof(uint32_t), (hipStream_t) stream));
		OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
		// A stream of frames without ambient occlusion: one pass of the primary kernel tells what its tiles cost (the leaves
		// their packets stop at), and the frames' workgroups take the costly 2 x 2 blocks first (orderPrimaryBlocks).
		if (tile_count != 0 && expected_frames >= FRAMES_WORTH_INTERVALS && PRIMARY_BY_COST_OK(kp) && primary_by_cost && scene_on_device) {
			launch_primary(scene_on_device->buffers(), (float *) d_image, nullptr, nullptr, d_tile_hits, d_tile_base, d_counters, kp, stream);
			OCRT_HIP(hipGetLastError());
			tile_words.assign(tile_count, 0u);
			OCRT_HIP(hipMemcpyAsync(tile_words.data(), d_tile_hits, tile_count * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t) stream));
			OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
			queue_static = {};
			orderBlocksByCost();
			uploadBlocksByCost();
		}
		return;
	}
	if (layout_from) {  // (the same frame of the same scene: the same hits)
		hit_slots = layout_from->hit_slots;
		// the walk intervals are read-only between uploads: the ring's hosts share ONE table (like the scene's arrays)
		tile_entry_owner = layout_from->tile_entry_owner;
		d_tile_entry = tile_entry_owner.get();
		intervals_in_use = layout_from->intervals_in_use;
		// (on this renderer's stream, and waited for: the stream is a non-blocking one, which work on the null stream is not
		// ordered against, and a copy or fill between device buffers need not have finished when its call returns)
		OCRT_HIP(hipMemcpyAsync(d_tile_base, layout_from->d_tile_base, tile_count * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t) stream));
		// (the tile words too: hit count | cost class per tile, what walkEntries() and the statistics read before this
		// renderer's own first frame has written them)
		OCRT_HIP(hipMemcpyAsync(d_tile_hits, layout_from->d_tile_hits, tile_count * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t) stream));
		OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
		// ... and the order its tiles are claimed in by the ambient-occlusion pass
		tile_words = layout_from->tile_words;
		tile_cost = layout_from->tile_cost;
		installOrder(layout_from->order_host, layout_from->queue_static, layout_from->split_tiles);
		d_hits = device_alloc((hit_slots ? hit_slots : 1) * sizeof(HitRec));
		d_occluded = device_alloc((hit_slots ? hit_slots : 1) * sizeof(uint32_t));
		clock.mark("hit list laid out like the ring's first host");
		return;
	}
	launch_primary(scene_on_device->buffers(), (float *) d_image, nullptr, nullptr, d_tile_hits, d_tile_base, d_counters, kp, stream);
	OCRT_HIP(hipGetLastError());
	std::vector<uint32_t> words(tile_count);
	OCRT_HIP(hipMemcpyAsync(words.data(), d_tile_hits, tile_count * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t) stream));
	OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
	clock.mark("counting pass + tile words to the host");
	// the order the ambient-occlusion pass claims the tiles in: by the cost class the counting pass found (the leaves each
	// tile's primary packet stopped at) until somebody has measured what the tiles' any-hit rays really cost
	tile_words = words;
	tile_cost.clear();
	orderTiles();
	clock.mark("tiles ordered for the ambient-occlusion pass");
	unsigned long long running = 0;
	for (uint32_t &w : words) {
		const uint32_t count = w & 0xFFu;
		w = (uint32_t) running;
		running += count;
	}
	if (running >= (1ull << 32))
		throw std::invalid_argument("more than 2^32 hit sub-pixels in one rank's bands");
	hit_slots = (size_t) running;
	OCRT_HIP(hipMemcpyAsync(d_tile_base, words.data(), tile_count * sizeof(uint32_t), hipMemcpyHostToDevice, (hipStream_t) stream));
	OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));  // (`words` is pageable memory of this function)
	d_hits = device_alloc((hit_slots ? hit_slots : 1) * sizeof(HitRec));
	d_occluded = device_alloc((hit_slots ? hit_slots : 1) * sizeof(uint32_t));
	allocEntries(entryBytes());
	clock.mark("tile bases to the device, image cleared, hit list allocated");
	// ... and, now that there is a hit list, where each tile's any-hit rays enter the walk tree: a second pass of the
	// primary kernel fills the list, entry_kernel reads the tiles' hit points (kernels.hip; the whole array where the
	// fast walk is not used).  Like the bases: once per upload.
	// (the intervals are ranges of the centre / half-extent copy of the walk records -- the any-hit packets' --, which
	// exists where the scaled node test may be used: walk_scale > 0)
	intervals_in_use = kp.fast_walk && kp.walk_scale > 0.0f && scene_on_device->buffers().walk && expected_frames >= FRAMES_WORTH_INTERVALS;
	if (intervals_in_use) {
		launch_primary(scene_on_device->buffers(), (float *) d_image, d_hits, d_occluded, d_tile_hits, d_tile_base, d_counters, kp, stream);
		launch_entries(scene_on_device->buffers(), d_hits, d_tile_hits, d_tile_base, d_tile_entry, kp, stream);
		OCRT_HIP(hipGetLastError());
	} else {
		std::vector<uint32_t> all(entryBytes() / sizeof(uint32_t));
		for (size_t t = 0; t < all.size() / 2; ++t) {
			all[2 * t] = 0u;
			all[2 * t + 1] = 0xFFFFFFFFu;
		}
		OCRT_HIP(hipMemcpyAsync(d_tile_entry, all.data(), all.size() * sizeof(uint32_t), hipMemcpyHostToDevice, (hipStream_t) stream));
		OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
	}
	// (the passes above left their pending tags in the image.  On the renderer's own stream: it is a non-blocking one, a
	// fill on the null stream would not be ordered against the frames that follow -- and was not, until round 4: a 530 MB
	// image at 64 samples per pixel takes long enough to clear for the first frame's primary pass to be overtaken by it.)
	OCRT_HIP(hipMemsetAsync(d_image, 0, image_bytes, (hipStream_t) stream));
	OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
	clock.mark("entries of the tiles' any-hit walks");
}

bool DeviceRenderer::fusedFrame() const {
	const bool possible = kp.ao_mode == AO_UNIFORM && kp.ao_dirs > 0 && kp.shared_walk && tile_count > 0 && kp.tiles_x < 65536u &&
	                      kp.local_tile_rows < 65536u && !primary_order_host.empty();
	// Measured and NOT the rule (profiles/r05_notes.md): the fused frame renders the same bits, but a frame on its own takes
	// 2-9 % LONGER with it than as two kernels -- headline 1.09 against 1.07 ms, interior 1080p 1.08 against 1.02, 4K 3.74
	// against 3.43 -- whether the primary work is taken at the head of a group's turn or a little ahead of the any-hit work
	// all through the frame: primary waves that are tied to a persistent workgroup's barriers hold their wave slots for the
	// block's slowest tile, and what the launch boundary cost (the chip draining and filling once) is less than that.  The
	// kernel stays reachable for experiments (rt_debug_set_frame_form) and under test (tests/test_fused_frame.py).
	return possible && frame_form == FrameForm::FUSED;
}

void DeviceRenderer::setFrameForm(int form) {
	const FrameForm want = form == 1 ? FrameForm::FUSED : form == 2 ? FrameForm::SEPARATE : FrameForm::AUTO;
	if (want == frame_form)
		return;
	frame_form = want;
	++scene_version;  // (a captured frame bakes its kernels in)
}

void DeviceRenderer::poisonHitList() {
	useDevice();
	synchronize();
	if (d_hits && hit_slots) {
		// (on the renderer's own stream and waited for: a fill on the null stream is not ordered against a non-blocking
		// stream's work and need not have finished when its call returns)
		OCRT_HIP(hipMemsetAsync(d_hits, 0xFF, hit_slots * sizeof(HitRec), (hipStream_t) stream));
		OCRT_HIP(hipMemsetAsync(d_occluded, 0xFF, hit_slots * sizeof(uint32_t), (hipStream_t) stream));
		OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
	}
}

void DeviceRenderer::checkFrameHealth() {
	uint32_t stalled = 0;
	OCRT_HIP(hipMemcpy(&stalled, (const char *) d_counters + offsetof(FrameCounters, stalled), sizeof stalled, hipMemcpyDeviceToHost));
	if (stalled) {
		OCRT_HIP(hipMemsetAsync((char *) d_counters + offsetof(FrameCounters, stalled), 0, sizeof stalled, (hipStream_t) stream));
		OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
		throw DeviceError("the fused frame kernel waited in vain for a tile's hit records (" + std::to_string(stalled) + " waves gave up): the frame is not valid");
	}
}

DeviceRenderer::FrameEvents DeviceRenderer::takeEvents() {
	FrameEvents ev;
	if (!free_events.empty()) {
		ev = free_events.back();
		free_events.pop_back();
	} else {
		hipEvent_t a, b, c, d;
		OCRT_HIP(hipEventCreate(&a));
		OCRT_HIP(hipEventCreate(&b));
		OCRT_HIP(hipEventCreate(&c));
		OCRT_HIP(hipEventCreate(&d));
		ev = { a, b, c, d };
	}
	ev.ao_timed = false;
	return ev;
}

// The launches of one frame on the stream: primary pass (with the ordering step in its tail), ambient-occlusion pass,
// finishing kernel (AO factor into the float image and -- with a destination -- the device resize in the same sweep).
// `ao_start` / `ao_stop` bracket the ao_kernel launch.
void DeviceRenderer::launchFrame(void *device_u8, void *ao_start, void *ao_stop, void *tile_cost_out) {
#ifdef OCRT_STAMPS
	hipStream_t s = (hipStream_t) stream;
#endif
	const SceneBuffers scene = scene_on_device->buffers();
	const uint32_t workgroups = ao_blocks_override ? ao_blocks_override : aoWorkgroups();
	if (fusedFrame() && !tile_cost_out) {
		// the two ray passes as one persistent launch (kernels/frame.hip.h); the events bracket it
		launch_frame(scene, (float *) d_image, d_hits, d_occluded, d_tile_hits, d_order, d_primary_order, d_order_need, d_tile_ready, d_tile_base,
		             d_tile_entry, d_counters, kp, workgroups, ao_prefetch, stream, ao_start, ao_stop);
		OCRT_HIP(hipGetLastError());
	} else {
		launch_primary(scene, (float *) d_image, d_hits, d_occluded, d_tile_hits, d_tile_base, d_counters, kp, stream,
		               primary_by_cost && PRIMARY_BY_COST_OK(kp) && kp.primary_list_stride ? d_blocks_by_cost : nullptr);
		OCRT_HIP(hipGetLastError());
#ifdef OCRT_STAMPS  // (instrumented build: the AO pass takes the minimum of its waves' start times into this slot)
		OCRT_HIP(hipMemsetAsync((char *) d_counters + offsetof(FrameCounters, stamp) + 7 * sizeof(unsigned long long), 0xFF, sizeof(unsigned long long), s));
#endif
		launch_ao(scene, d_hits, d_occluded, d_order, d_tile_base, d_tile_entry, d_counters, kp, workgroups, ao_prefetch, stream, ao_start, ao_stop,
		          tile_cost_out);
		OCRT_HIP(hipGetLastError());
	}
	launch_finish((float *) d_image, d_hits, d_occluded, d_tile_base, (unsigned char *) device_u8, kp, opts.width, grid, local_out_rows, stream, d_counters);
	OCRT_HIP(hipGetLastError());
}

DeviceRenderer::WalkEntries DeviceRenderer::walkEntries() const {
	if (!scene_ready)
		throw std::logic_error("walkEntries: no scene on the device");
	useDevice();
	WalkEntries out{ 0, 0, 1.0, 1.0 };
	const bool has_ao = kp.ao_mode != AO_NONE && kp.ao_dirs > 0;
	if (tile_count == 0 || !has_ao || kp.node_count == 0)
		return out;
	const size_t stride = kp.entry_stride;  // (intervals per tile: entry_kernel)
	std::vector<uint32_t> words(tile_count);
	OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
	OCRT_HIP(hipMemcpyAsync(words.data(), d_tile_hits, tile_count * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t) stream));
	OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
	const double whole = (double) kp.node_count * sizeof(NodeRec);
	double sum = 0.0, sum_packets = 0.0;
	unsigned long long packets = 0;
	// the table in pieces of at most 32 MB (it is 0.5 GB at 64 samples per pixel and may be 2 GB: a statistic is not worth
	// a host copy of that size)
	const size_t tiles_per_piece = std::max<size_t>(1, ((size_t) 32 << 20) / (stride * 2 * sizeof(uint32_t)));
	std::vector<uint32_t> ranges;
	for (size_t first = 0; first < tile_count; first += tiles_per_piece) {
		const size_t count = std::min(tiles_per_piece, tile_count - first);
		ranges.resize(count * stride * 2);
		OCRT_HIP(hipMemcpyAsync(ranges.data(), (const uint32_t *) d_tile_entry + first * stride * 2, ranges.size() * sizeof(uint32_t), hipMemcpyDeviceToHost,
		                        (hipStream_t) stream));
		OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
		for (size_t t = first; t < first + count; ++t) {
			if ((words[t] & 0xFFu) == 0u)
				continue;
			++out.tiles_hit;
			const uint32_t *const of_tile = &ranges[2 * stride * (t - first)];
			const double bytes = std::min(whole, (double) of_tile[1]) - (double) of_tile[0];
			out.tiles_narrowed += of_tile[0] != 0u || bytes < whole;
			sum += bytes / whole;
			// what the tile's packets walk: a full tile's, one table direction each, have intervals of their own
			const bool per_direction = (words[t] & 0xFFu) == 64u && kp.ao_mode == AO_UNIFORM && stride > 1;
			for (size_t k = 1; k <= kp.ao_dirs; ++k) {
				const uint32_t *const r = per_direction ? of_tile + 2 * k : of_tile;
				sum_packets += (std::min(whole, (double) r[1]) - (double) r[0]) / whole;
				++packets;
			}
		}
	}
	out.mean_share = out.tiles_hit ? sum / out.tiles_hit : 1.0;
	out.mean_packet_share = packets ? sum_packets / (double) packets : 1.0;
	return out;
}

void DeviceRenderer::expectFrames(uint64_t frames) {
	const bool before = expected_frames >= FRAMES_WORTH_INTERVALS, after = frames >= FRAMES_WORTH_INTERVALS;
	expected_frames = frames;
	if (before == after || !scene_ready)
		return;
	// the scene is on the device already: lay the hit list out again, with or without the table (adopt() decides)
	std::shared_ptr<const DeviceScene> scene = scene_on_device;
	adopt(std::move(scene));
}

void DeviceRenderer::setAoPrefetch(bool on) {
	if (on == ao_prefetch)
		return;
	ao_prefetch = on;
	++scene_version;  // (a captured frame bakes the kernel in: capture again)
}

bool DeviceRenderer::calibrateAoPrefetch(float *ms_without, float *ms_with) {
	if (!scene_ready)
		throw std::logic_error("calibration before upload");
	const bool has_ao = kp.ao_mode == AO_UNIFORM && kp.ao_dirs > 0 && tile_count > 0 && kp.fast_walk && kp.walk_scale > 0.0f;
	float median[2] = { 0.0f, 0.0f };
	if (has_ao) {
		const bool before = ao_prefetch;
		std::vector<float> samples[2];
		// interleaved, the first frame of each form not counted (code object pages, caches)
		for (int round = 0; round < 6; ++round)
			for (int form = 0; form < 2; ++form) {
				ao_prefetch = form != 0;
				enqueueRender();
				synchronize();
				if (round > 0)
					samples[form].push_back(last_ao_ms);
			}
		for (int form = 0; form < 2; ++form) {
			std::sort(samples[form].begin(), samples[form].end());
			median[form] = samples[form][samples[form].size() / 2];
		}
		ao_prefetch = before;
		setAoPrefetch(median[1] <= median[0]);
		resetTimers();
		frame_ready = false;
	}
	if (ms_without)
		*ms_without = median[0];
	if (ms_with)
		*ms_with = median[1];
	return ao_prefetch;
}

void DeviceRenderer::enqueueRender() {
	if (!scene_ready)
		throw std::logic_error("render called before upload");
	useDevice();
	FrameEvents ev = takeEvents();
	hipStream_t s = (hipStream_t) stream;
	OCRT_HIP(hipEventRecord((hipEvent_t) ev.start, s));
	ev.ao_timed = kp.ao_mode != AO_NONE && kp.ao_dirs > 0 && tile_count > 0;
	launchFrame(nullptr, ev.ao_start, ev.ao_stop);
	OCRT_HIP(hipEventRecord((hipEvent_t) ev.stop, s));
	pending_events.push_back(ev);
	frame_ready = true;
}

void DeviceRenderer::dropFrameGraphs() {
	for (FrameGraph &g : frame_graphs) {
		if (g.exec)
			(void) hipGraphExecDestroy((hipGraphExec_t) g.exec);
		if (g.graph)
			(void) hipGraphDestroy((hipGraph_t) g.graph);
	}
	frame_graphs.clear();
}

// The captured frame for `dst` (captured now if there is none yet).
const DeviceRenderer::FrameGraph *DeviceRenderer::frameGraphFor(void *dst) {
	const bool has_ao = kp.ao_mode != AO_NONE && kp.ao_dirs > 0 && tile_count > 0;
	hipStream_t s = (hipStream_t) stream;
	// The captured frame for this destination; everything else a graph bakes in (scene, stream, share of the device)
	// invalidates all of them.
	if (!frame_graphs.empty()) {
		const FrameGraph &any = frame_graphs.front();
		if (any.stream != stream || any.scene_version != scene_version || any.device_share != device_share) {
			OCRT_HIP(hipStreamSynchronize(s));
			dropFrameGraphs();
		}
	}
	FrameGraph *g = nullptr;
	for (FrameGraph &have : frame_graphs)
		if (have.dst == dst)
			g = &have;
	if (!g) {
		// Capture.  The graph holds the frame's launches with their arguments by value: the scene's pointers and
		// launch constants, the destination of the resize, the size of the persistent grid.  No events inside: an
		// event-record node cannot be timed (hipEventElapsedTime: invalid handle); the kernels stamp the device clock
		// into the frame's counters instead (FrameCounters::tick_*).
		if (frame_graphs.size() >= 4) {  // (a caller that keeps changing the destination: start over)
			OCRT_HIP(hipStreamSynchronize(s));
			dropFrameGraphs();
		}
		FrameGraph fresh;
		OCRT_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
		hipGraph_t graph = nullptr;
		try {
			launchFrame(dst, nullptr, nullptr);
		} catch (...) {
			(void) hipStreamEndCapture(s, &graph);
			if (graph)
				(void) hipGraphDestroy(graph);
			throw;
		}
		OCRT_HIP(hipStreamEndCapture(s, &graph));
		fresh.graph = graph;
		hipGraphExec_t exec = nullptr;
		const hipError_t made = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
		if (made != hipSuccess) {
			(void) hipGraphDestroy(graph);
			OCRT_HIP(made);
		}
		fresh.exec = exec;
		fresh.dst = dst;
		fresh.stream = stream;
		fresh.scene_version = scene_version;
		fresh.device_share = device_share;
		fresh.ao_events = has_ao;
		fresh.valid = true;
		frame_graphs.push_back(fresh);
		g = &frame_graphs.back();
	}
	return g;
}

void DeviceRenderer::prepareFrame(void *device_u8) {
	if (!scene_ready)
		throw std::logic_error("prepare called before upload");
	useDevice();
	if (graph_mode)
		(void) frameGraphFor(device_u8 ? device_u8 : d_u8);
}

void DeviceRenderer::enqueueFrame(void *device_u8) {
	if (!scene_ready)
		throw std::logic_error("render called before upload");
	useDevice();
	void *dst = device_u8 ? device_u8 : d_u8;
	const bool has_ao = kp.ao_mode != AO_NONE && kp.ao_dirs > 0 && tile_count > 0;
	hipStream_t s = (hipStream_t) stream;
	if (!graph_mode) {
		FrameEvents ev = takeEvents();
		OCRT_HIP(hipEventRecord((hipEvent_t) ev.start, s));
		ev.ao_timed = has_ao;
		launchFrame(dst, ev.ao_start, ev.ao_stop);
		OCRT_HIP(hipEventRecord((hipEvent_t) ev.stop, s));
		pending_events.push_back(ev);
		frame_ready = true;
		return;
	}
	const FrameGraph *g = frameGraphFor(dst);
	FrameEvents ev = takeEvents();
	OCRT_HIP(hipEventRecord((hipEvent_t) ev.start, s));
	OCRT_HIP(hipGraphLaunch((hipGraphExec_t) g->exec, s));
	OCRT_HIP(hipEventRecord((hipEvent_t) ev.stop, s));
	ev.ao_timed = false;
	ev.graph_ao = g->ao_events;
	pending_events.push_back(ev);
	frame_ready = true;
}

void DeviceRenderer::enqueueResizeInto(void *device_u8) {
	if (!frame_ready)
		throw std::logic_error("resize called before render");
	useDevice();
	launch_resize((const float *) d_image, (unsigned char *) device_u8, kp, opts.width, grid, local_out_rows, stream);
	OCRT_HIP(hipGetLastError());
}

void DeviceRenderer::enqueueResize() { enqueueResizeInto(d_u8); }

void DeviceRenderer::waitForStream() {
	useDevice();
	OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
}

void DeviceRenderer::synchronize() {
	useDevice();
	OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
	for (size_t k = 0; k < pending_events.size(); ++k) {
		FrameEvents &ev = pending_events[k];
		const bool newest = k + 1 == pending_events.size();
		float ms = 0, ao_ms = 0;
		OCRT_HIP(hipEventElapsedTime(&ms, (hipEvent_t) ev.start, (hipEvent_t) ev.stop));
		// When the ao_kernel launch ran: by the frame's own HIP events (plain launches), or -- a replayed graph, whose
		// event-record nodes hipEventElapsedTime refuses -- by the device clock the kernels stamped into the frame's
		// counters (newest frame only, and only when asked for: it costs a small blocking copy).
		float ao_begin_after_start = 0.0f;
		bool ao_timed = ev.ao_timed;
		if (ao_timed) {
			OCRT_HIP(hipEventElapsedTime(&ao_ms, (hipEvent_t) ev.ao_start, (hipEvent_t) ev.ao_stop));
			OCRT_HIP(hipEventElapsedTime(&ao_begin_after_start, (hipEvent_t) ev.start, (hipEvent_t) ev.ao_start));
		} else if (ev.graph_ao && newest && keep_stamps) {
			FrameCounters c{};
			OCRT_HIP(hipMemcpy(&c, d_counters, sizeof c, hipMemcpyDeviceToHost));
			const unsigned long long tick[3] = { c.tick_begin, c.tick_ao_begin, c.tick_ao_end };
			if (tick[2] >= tick[1] && tick[1] >= tick[0] && tick[0] != 0) {
				ao_ms = (float) ((double) (tick[2] - tick[1]) * 1e-5);  // 100 MHz clock -> ms
				ao_begin_after_start = (float) ((double) (tick[1] - tick[0]) * 1e-5);
				ao_timed = true;
			}
		}
		if (ao_timed) {
			total_ao_ms += ao_ms;
			++ao_launches;
		}
		last_ms = ms;
		last_ao_ms = ao_ms;
		total_ms += ms;
		++launches;
		if (epoch_event && newest) {
			hipEvent_t epoch = (hipEvent_t) epoch_event;
			OCRT_HIP(hipEventElapsedTime(&last_times[0], epoch, (hipEvent_t) ev.start));
			OCRT_HIP(hipEventElapsedTime(&last_times[3], epoch, (hipEvent_t) ev.stop));
			last_times[1] = ao_timed ? last_times[0] + ao_begin_after_start : 0.0f;
			last_times[2] = ao_timed ? last_times[1] + ao_ms : 0.0f;
		}
		free_events.push_back(ev);
	}
	pending_events.clear();
}

void DeviceRenderer::downloadFloat(float *host_image) {
	synchronize();
	checkFrameHealth();
	const size_t row_bytes = (size_t) rt.totalWidth * sizeof(float);
	if (part.nranks == 1) {  // local rows are the image's rows (plus, possibly, padding below it)
		OCRT_HIP(hipMemcpy(host_image, d_image, row_bytes * rt.totalHeight, hipMemcpyDeviceToHost));
		return;
	}
	// a partitioned host: its bands go to their place in the caller's full-size image, the other ranks' rows are 0
	std::memset(host_image, 0, row_bytes * rt.totalHeight);
	const uint32_t band_rows = part.band_tile_rows * TILE_H;
	for (uint32_t local = 0; local < kp.local_tile_rows * TILE_H; local += band_rows) {
		const uint32_t global = ((local / band_rows) * part.nranks + part.rank) * band_rows;
		if (global >= rt.totalHeight)
			break;
		const uint32_t rows = rt.totalHeight - global < band_rows ? rt.totalHeight - global : band_rows;
		OCRT_HIP(hipMemcpy((char *) host_image + row_bytes * global, (const char *) d_image + row_bytes * local,
		                   row_bytes * rows, hipMemcpyDeviceToHost));
	}
}

void DeviceRenderer::downloadResizedLocal(unsigned char *host) {
	enqueueResize();
	synchronize();
	OCRT_HIP(hipMemcpy(host, d_u8, (size_t) local_out_rows * opts.width, hipMemcpyDeviceToHost));
}

void DeviceRenderer::downloadResizedFull(unsigned char *host) {
	if (part.nranks != 1)
		throw std::logic_error("full-image download needs an unpartitioned host");
	enqueueResize();
	synchronize();
	// local_out_rows may include padding rows below the image; copy only the image.
	OCRT_HIP(hipMemcpy(host, d_u8, (size_t) opts.height * opts.width, hipMemcpyDeviceToHost));
}

uint32_t DeviceRenderer::globalRowOf(uint32_t local_row) const {
	const uint32_t rows_per_band = part.band_tile_rows * TILE_H / grid;
	const uint32_t band_local = local_row / rows_per_band;
	return (band_local * part.nranks + part.rank) * rows_per_band + local_row % rows_per_band;
}

void DeviceRenderer::setStream(void *hip_stream) {
	synchronize();
	stream = hip_stream;
}

void DeviceRenderer::usePrivateStream() {
	synchronize();
	stream = own_stream;
}

RenderStats DeviceRenderer::stats() {
	RenderStats out{};
	if (!frame_ready)
		return out;
	synchronize();
	const bool has_ao = kp.ao_mode != AO_NONE && kp.ao_dirs > 0;
	if (has_ao) {
		// The frame's occlusion total is summed when it is asked for (kernels.hip, occluded_sum_kernel): the counts are in
		// the hit list until this host's next frame clears them.
		launch_occluded_sum(d_occluded, hit_slots, d_counters, stream);
		OCRT_HIP(hipGetLastError());
		OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
	}
	FrameCounters c{};
	OCRT_HIP(hipMemcpy(&c, d_counters, sizeof c, hipMemcpyDeviceToHost));
	if (c.stalled)
		checkFrameHealth();
	{  // hit sub-pixels: the low byte of the words the frame's primary pass left per tile
		std::vector<uint32_t> words(tile_count);
		OCRT_HIP(hipMemcpy(words.data(), d_tile_hits, tile_count * sizeof(uint32_t), hipMemcpyDeviceToHost));
		for (uint32_t w : words)
			out.primary_hits += w & 0xFFu;
	}
	out.ao_occluded = has_ao ? c.occluded : 0;
#ifdef OCRT_TAIL
	std::fprintf(stderr, "AO pass: last wave ended %.3f ms after the pass began; waves by the time they ended at (0.05 ms buckets):", (c.stamp[8] - c.tick_ao_begin) * 1e-5);
	for (int k = 0; k < 32; ++k)
		if (c.stamp[10 + k])
			std::fprintf(stderr, " %.2f:%llu", k * 0.05, c.stamp[10 + k]);
	std::fprintf(stderr, "\n");
#endif
#ifdef OCRT_STAMPS
	std::fprintf(stderr, "AO wave-time: claim %.3f ms, frames %.3f ms, walks %.3f ms, flush %.3f ms over %llu jobs / %llu packets\n",
	             c.stamp[0] * 1e-5, c.stamp[1] * 1e-5, c.stamp[2] * 1e-5, c.stamp[3] * 1e-5, c.stamp[4], c.stamp[5]);
	std::fprintf(stderr, "   wave lifetimes sum %.3f ms, kernel span %.3f ms, %llu waves worked\n", c.stamp[6] * 1e-5,
	             (c.stamp[8] - c.stamp[7]) * 1e-5, c.stamp[9]);
	std::fprintf(stderr, "   waves by the time they ended at, from the first wave's start (0.1 ms buckets):");
	for (int k = 0; k < 32; ++k)
		if (c.stamp[10 + k])
			std::fprintf(stderr, " %.1f:%llu", k * 0.1, c.stamp[10 + k]);
	std::fprintf(stderr, "\n   of the walks: node loop %.3f ms in %llu entries, batches %.3f ms in %llu batches, %llu leaf stops\n",
	             c.stamp[42] * 1e-5, c.stamp[44], c.stamp[43] * 1e-5, c.stamp[45], c.stamp[46]);
	std::fprintf(stderr, "   batched pairs %llu, of which the leaf's own box passes %llu\n", c.stamp[47], c.stamp[48]);
	std::fprintf(stderr, "   jobs by duration (from 2^k us on):");
	for (int k = 0; k < 14; ++k)
		if (c.stamp[49 + k])
			std::fprintf(stderr, " %d:%llu", 1 << k, c.stamp[49 + k]);
	std::fprintf(stderr, "\n   packets in the exact form %llu, wave-time of the jobs holding them %.3f ms\n", c.stamp[63], c.stamp[64] * 1e-5);
#endif
#ifdef OCRT_DEBUG_KNOBS
	if (std::getenv("OCRT_PRINT_COST")) {  // debug knob: the tiles' AO cost classes (leaves the primary packet stopped at)
		std::vector<uint32_t> th(tile_count);
		OCRT_HIP(hipMemcpy(th.data(), d_tile_hits, tile_count * sizeof(uint32_t), hipMemcpyDeviceToHost));
		unsigned long long tiles = 0, cost = 0, hits = 0;
		for (uint32_t v : th)
			if (v) {
				++tiles;
				cost += v >> 8;
				hits += v & 0xFFu;
			}
		std::fprintf(stderr, "hit tiles %llu, mean cost class %.2f, mean hits per tile %.1f\n", tiles, tiles ? (double) cost / tiles : 0.0,
		             tiles ? (double) hits / tiles : 0.0);
	}
#endif
	// Primary rays = sub-pixels of this rank's bands that lie inside the image.
	const uint32_t tile_rows = (kp.height + TILE_H - 1) / TILE_H;
	unsigned long long rows = 0;
	for (uint32_t j = 0; j < kp.local_tile_rows; ++j) {
		const uint32_t band_local = j / part.band_tile_rows;
		const uint32_t row = (band_local * part.nranks + part.rank) * part.band_tile_rows + j % part.band_tile_rows;
		if (row >= tile_rows)
			continue;
		const uint32_t y0 = row * TILE_H;
		rows += (kp.height - y0 < TILE_H) ? kp.height - y0 : TILE_H;
	}
	out.primary_rays = rows * kp.width;
	out.ao_rays = kp.ao_mode != AO_NONE ? out.primary_hits * kp.ao_dirs : 0;
	return out;
}

}  // namespace ocrt
