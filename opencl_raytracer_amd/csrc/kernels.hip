// kernels.hip -- gfx950 ray-casting kernels (the replacement for reference
// src/intersect_kernel.cl).
//
// Arithmetic contract: IEEE binary32 + - * / sqrt in the
// reference's operation order, NO fma contraction (this file is compiled with
// -ffp-contract=off), IEEE maxNum/minNum for max/min, double-literal
// comparisons folded to their exact float thresholds.  Anything else (data
// layout, traversal order, where a value is computed) is free and is chosen
// for the CDNA4 wave64 machine.
//
// Work decomposition: one 64-lane wavefront per 8x8 tile of sub-pixels, in two ray passes and a finishing sweep --
// three kernels per frame (round 3: six), captured once per host as a hipGraph and replayed.
//   primary_kernel  every lane casts its primary ray (closest hit) and computes
//                   the smooth normal and head-light term.  Sub-pixels that need
//                   no ambient occlusion are final; the tile's other hits leave a TAG in the image and
//                   are ballot-compacted into the tile's slots of the hit list (tile_base: sized by what is hit).
//                   Its TAIL is the ordering step: the last workgroup of each XCD group to finish sorts the
//                   group's non-empty tiles by AO cost class (counting sort, the costly blocks first).
//   ao_kernel       persistent workgroups claim runs of (tile, table direction) units in
//                   that order -- a tile at a time, whose directions the four waves take
//                   from a cursor in LDS.  A wave rebuilds the tile's tangent frames in its
//                   LDS slice and casts one packet of 64 any-hit rays per
//                   direction -- one table direction from the tile's neighbouring
//                   surface points -- that stop at the first accepted triangle and walk only
//                   the interval of the node array their segments can reach (entry_kernel, once per upload);
//                   occlusion counts are LDS atomics, flushed to a per-hit counter
//                   when the claim is done.
//   finish_kernel / finish_wide_kernel
//                   value * (1 - occluded / n) into the tagged sub-pixels and the supersample box filter +
//                   quantisation of the same sweep (n = 1: a thread per pixel; n >= 2: along the sub-pixel rows).
//   (on demand)     entry_kernel: the walk intervals, once per upload; occluded_sum_kernel: the frame's occlusion
//                   total when the statistics are asked for; resize_kernel: a box filter on its own.
// Why not one fused launch (it was, see profiles/r01_notes.md): cost per tile
// varies 30x (background vs model, 29 rays per hit sub-pixel), so the frame used
// to end on a long tail of half-empty CUs.  With the tiles' costs known after the
// primary pass, claiming the costly blocks first packs them almost perfectly.
//
// How rays walk the tree: the 64 rays of a wave share ONE node index ("shared
// walk", see walk_collect below) -- nodes and triangles arrive by
// scalar loads, boxes are tested out of SGPRs, nothing diverges and nothing is
// gathered.  The first generation, in which every lane walked on its own under a
// wave scheduler, is only compiled into the A/B build (-DOCRT_DEBUG_KNOBS, where
// OCRT_NO_SHARED_WALK=1 selects it); the product library does not contain it.
//
// What bounds it: the scene (19 MB) is cache-resident, HBM traffic is negligible;
// the walk is bound by vector-instruction issue (11 to 17 per node and primary packet, 12 per any-hit packet) and the
// latency of the one scalar load per pair of nodes (scenes beyond the caches: by that latency alone) -- see DESIGN.md section 5 and
// profiles/r0*_notes.md for the counters and the microbenchmarks.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_types.h"
#include "exact_reciprocal.h"
#include "tri_predicate.h"

#include "kernels/common.hip.h"
#ifdef OCRT_DEBUG_KNOBS
#include "kernels/first_generation.hip.h"
#endif
#include "kernels/walk.hip.h"
#include "kernels/primary.hip.h"
#include "kernels/ao.hip.h"
#include "kernels/frame.hip.h"
#include "kernels/entry.hip.h"
#include "kernels/finish.hip.h"


namespace ocrt {

// ---- host-callable launchers (keeps the launch syntax inside this TU) ----
// Makes the runtime load this library's code object for the current device now (it is otherwise loaded at the first
// launch): called from a warm-up thread while the CPU still builds the scene.
void preload_kernels() {
	hipFuncAttributes attr;
	(void) hipFuncGetAttributes(&attr, (const void *) primary_kernel<true>);
	(void) hipFuncGetAttributes(&attr, (const void *) ao_kernel<AO_UNIFORM, true, true>);
	(void) hipGetLastError();
}

#if defined(OCRT_STAMPS) || defined(OCRT_TAIL)
// (instrumented builds only: the debug stamps are sums and need zeroing; the frame's own counters do not -- device_types.h)
__global__ __launch_bounds__(256) void clear_stamps_kernel(FrameCounters *counters) {
	for (uint32_t i = threadIdx.x; i < sizeof(counters->stamp) / sizeof(counters->stamp[0]); i += blockDim.x)
		counters->stamp[i] = 0ull;
}
#endif

#ifdef OCRT_OCML_BUILTINS
// (test-only build: the UNIFORM direction table as the reference kernel computes it on the device -- per pixel, with the
// library's own float sin / cos / cospi / sinpi, src/intersect_kernel.cl:237-246 -- instead of the host's libm: the same
// expressions in the same order, one thread)
__global__ void ocml_ao_table_kernel(float4 *table, uint32_t *count, uint32_t rings, int alpha_min, int alpha_max, uint32_t capacity) {
	if (threadIdx.x != 0 || blockIdx.x != 0)
		return;
	uint32_t n = 0;
	const float degrees = (float) (M_PI / 180);
	const float amin = (float) alpha_min * degrees;
	const float amax = (float) alpha_max * degrees;
	for (uint32_t ring = 0; ring < rings; ++ring) {
		const float step = amax / rings;
		const float angle = (step * ring) + amin;
		const uint32_t ray_count = (uint32_t) ((2.0f * M_PI * ocl_cos(angle)) / step);
		const float theta = (float) (M_PI_2 - angle);
		for (uint32_t k = 0; k <= ray_count; ++k) {
			const float phi = (float) ((2.0f * M_PI * k) / ray_count);
			const float xs = ocl_sin(theta) * ocl_cospi(phi);
			const float ys = ocl_cos(theta);
			const float zs = ocl_sin(theta) * ocl_sinpi(phi);
			if (n < capacity)
				table[n] = make_float4(xs, ys, zs, 0.0f);
			++n;
		}
	}
	*count = n;
}
// Overwrites the `capacity` entries at `table` (device) with the device-made ones; returns how many the device counted.
uint32_t ocml_ao_table(void *table, uint32_t rings, int alpha_min, int alpha_max, uint32_t capacity) {
	uint32_t *d_count = nullptr, count = 0;
	if (hipMalloc(&d_count, sizeof(uint32_t)) != hipSuccess)
		return 0;
	hipLaunchKernelGGL(ocml_ao_table_kernel, dim3(1), dim3(64), 0, nullptr, (float4 *) table, d_count, rings, alpha_min, alpha_max, capacity);
	(void) hipDeviceSynchronize();
	(void) hipMemcpy(&count, d_count, sizeof count, hipMemcpyDeviceToHost);
	(void) hipFree(d_count);
	return count;
}
#endif

// The frame is three kernels (two without ambient occlusion): primary pass (+ ordering step in its tail), the
// ambient-occlusion pass, the finishing kernel (AO factor + box filter + quantisation).
#ifdef OCRT_PRIMARY_TICKS
extern void *primary_ticks_probe;
#endif
void launch_primary(const SceneBuffers &scene, float *image, void *hits, void *occluded_of, void *tile_hits,
                    const void *tile_base, void *counters, const KernelParams &P, void *stream, const void *blocks_by_cost) {
	hipStream_t s = (hipStream_t) stream;
#if defined(OCRT_STAMPS) || defined(OCRT_TAIL)
	hipLaunchKernelGGL(clear_stamps_kernel, dim3(1), dim3(256), 0, s, (FrameCounters *) counters);
#endif
	if (P.tiles_x * P.local_tile_rows == 0)
		return;
	const uint32_t strips = (P.tiles_x + P.strip_tiles - 1u) / P.strip_tiles, row_blocks = (P.local_tile_rows + PRIMARY_ROWS - 1u) / PRIMARY_ROWS;
	// (with a list -- DeviceRenderer::orderPrimaryBlocks -- every group's workgroups take its entries one by one)
	const uint32_t blocks = blocks_by_cost && P.primary_list_stride && PRIMARY_WAVES == 4u ? XCD_GROUPS * P.primary_list_stride
	                                                                                        : XCD_GROUPS * ((strips + XCD_GROUPS - 1u) >> 3) * row_blocks * (P.strip_tiles >> 1);
	auto launch = [&](auto kernel) {
		FrameArgs args{};
		args.walk_ptr = (const float4 *) scene.walk;
		args.tris_ptr = (const float4 *) scene.tris;
		args.nodes_ptr = (const float4 *) scene.nodes;
		args.shade = (const float4 *) scene.shade;
		args.image = image;
		args.hits = (HitRec *) hits;
		args.occluded_of = (uint32_t *) occluded_of;
		args.tile_hits = (uint32_t *) tile_hits;
		args.tile_base = (const uint32_t *) tile_base;
		args.counters = (FrameCounters *) counters;
		args.primary_order = P.primary_list_stride && PRIMARY_WAVES == 4u ? (const uint32_t *) blocks_by_cost : nullptr;  // (null: the spatial mapping)
#ifdef OCRT_PRIMARY_TICKS
		args.tile_cost = (uint32_t *) primary_ticks_probe;
#endif
		args.P = P;
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * PRIMARY_WAVES), 0, s, args);
	};
#ifdef OCRT_DEBUG_KNOBS
	if (!P.shared_walk)
		return launch(primary_kernel<false>);
#endif
	launch(primary_kernel<true>);
}

#ifdef OCRT_PRIMARY_TICKS
void *primary_ticks_probe = nullptr;  // (probe build: set by DeviceRenderer::measureTileCosts around its frames)
#endif

// Fills `tile_entry` (1 + ao_dirs intervals of two words per tile) for the frame whose hit list is in `hits`: once per
// upload (entry_kernel).
void launch_entries(const SceneBuffers &scene, const void *hits, const void *tile_hits, const void *tile_base, void *tile_entry,
                    const KernelParams &P, void *stream) {
	const uint32_t tiles = P.tiles_x * P.local_tile_rows;
	if (tiles == 0)
		return;
	hipLaunchKernelGGL(entry_kernel, dim3((tiles + ENTRY_WAVES - 1u) / ENTRY_WAVES), dim3(64 * ENTRY_WAVES), 0, (hipStream_t) stream,
	                   (const NodeRec *) ((const char *) scene.walk + P.walk_ce_bytes), (const HitRec *) hits, (const uint32_t *) tile_hits, (const uint32_t *) tile_base,
	                   (const float4 *) scene.ao_table, (uint2 *) tile_entry, tiles, P.entry_stride, P.ao_mode == AO_UNIFORM && scene.ao_table ? 1 : 0,
	                   P.ao_max_distance);
}

void launch_ao(const SceneBuffers &scene, void *hits, void *occluded_of, void *order, const void *tile_base, const void *tile_entry,
               void *counters, const KernelParams &params, uint32_t workgroups, bool prefetch, void *stream, void *event_before_ao,
               void *event_after_ao, void *tile_cost) {
	if (params.tiles_x * params.local_tile_rows == 0 || params.ao_mode == AO_NONE || params.ao_dirs == 0)
		return;
	hipStream_t s = (hipStream_t) stream;
	// persistent grid: what the chip holds, or one wave per (tile, direction) when the image is small
	const uint32_t tiles = params.tiles_x * params.local_tile_rows;
	const uint64_t units = (uint64_t) tiles * params.ao_dirs;
	uint32_t ao_blocks = workgroups;  // (DeviceRenderer::aoWorkgroups: 8 per CU for a host alone on its GPU)
	if ((units + AO_WAVES - 1) / AO_WAVES < ao_blocks)
		ao_blocks = (uint32_t) ((units + AO_WAVES - 1) / AO_WAVES);
	KernelParams P = params;
	const uint32_t waves_per_group = (ao_blocks * AO_WAVES + XCD_GROUPS - 1u) / XCD_GROUPS;
	P.ao_guide = P.ao_guide * (waves_per_group ? waves_per_group : 1u);
	P.ao_claim_div = waves_per_group ? waves_per_group : 1u;  // (the waves of a group, for ao_kernel's claim rule)
	auto launch = [&](auto kernel) {
		// (the events bracket the ao_kernel launch alone: its duration is the one the roofline is quoted for)
		if (event_before_ao)
			(void) hipEventRecord((hipEvent_t) event_before_ao, s);
		FrameArgs args{};
		args.walk_ptr = (const float4 *) scene.walk;
		args.tris_ptr = (const float4 *) scene.tris;
		args.nodes_ptr = (const float4 *) scene.nodes;
		args.ao_table = (const float4 *) scene.ao_table;
		args.hits = (HitRec *) hits;
		args.occluded_of = (uint32_t *) occluded_of;
		args.order = (const uint32_t *) order;
		args.tile_base = (const uint32_t *) tile_base;
		args.tile_entry = (const uint2 *) tile_entry;
		args.counters = (FrameCounters *) counters;
		args.tile_cost = (uint32_t *) tile_cost;
		args.P = P;
		hipLaunchKernelGGL(kernel, dim3(ao_blocks), dim3(64 * AO_WAVES), 0, s, args);
		if (event_after_ao)
			(void) hipEventRecord((hipEvent_t) event_after_ao, s);
	};
#ifdef OCRT_DEBUG_KNOBS
	if (!P.shared_walk) {
		if (P.ao_mode == AO_UNIFORM)
			launch(ao_kernel<AO_UNIFORM, false>);
		else
			launch(ao_kernel<AO_RANDOM, false>);
	} else
#endif
	if (P.ao_mode == AO_UNIFORM) {
		if (prefetch)
			launch(ao_kernel<AO_UNIFORM, true, true>);
		else
			launch(ao_kernel<AO_UNIFORM, true, false>);
	} else
		launch(ao_kernel<AO_RANDOM, true>);
}

// The two ray passes as ONE persistent launch (kernels/frame.hip.h): UNIFORM ambient occlusion, the shared walk.
// `primary_order`, `tile_ready`: DeviceRenderer::orderTiles / its flag array.
void launch_frame(const SceneBuffers &scene, float *image, void *hits, void *occluded_of, void *tile_hits, void *order,
                  const void *primary_order, const void *order_need, void *tile_ready, const void *tile_base, const void *tile_entry, void *counters,
                  const KernelParams &params, uint32_t workgroups, bool prefetch, void *stream, void *event_before, void *event_after) {
	if (params.tiles_x * params.local_tile_rows == 0)
		return;
	hipStream_t s = (hipStream_t) stream;
	KernelParams P = params;
	// persistent grid: what the chip holds, or -- a small image -- what there is to do: a workgroup per 2 x 2 tile block
	// of the primary work or per (tile, four directions) of the ambient-occlusion work, whichever is more
	{
		const uint64_t blocks = (uint64_t) ((P.tiles_x + 1u) / 2u) * ((P.local_tile_rows + 1u) / 2u);
		const uint64_t units = ((uint64_t) P.tiles_x * P.local_tile_rows * P.ao_dirs + AO_WAVES - 1) / AO_WAVES;
		const uint64_t most = blocks > units ? blocks : units;
		if (most < workgroups)
			workgroups = (uint32_t) (most ? most : 1u);
	}
	const uint32_t waves_per_group = (workgroups * AO_WAVES + XCD_GROUPS - 1u) / XCD_GROUPS;
	P.ao_guide = P.ao_guide * (waves_per_group ? waves_per_group : 1u);
	P.ao_claim_div = waves_per_group ? waves_per_group : 1u;
	FrameArgs args{};
	args.walk_ptr = (const float4 *) scene.walk;
	args.tris_ptr = (const float4 *) scene.tris;
	args.nodes_ptr = (const float4 *) scene.nodes;
	args.shade = (const float4 *) scene.shade;
	args.ao_table = (const float4 *) scene.ao_table;
	args.image = image;
	args.hits = (HitRec *) hits;
	args.occluded_of = (uint32_t *) occluded_of;
	args.tile_hits = (uint32_t *) tile_hits;
	args.order = (const uint32_t *) order;
	args.tile_base = (const uint32_t *) tile_base;
	args.tile_entry = (const uint2 *) tile_entry;
	args.counters = (FrameCounters *) counters;
	args.primary_order = (const uint32_t *) primary_order;
	args.order_need = (const uint32_t *) order_need;
	args.tile_ready = (uint32_t *) tile_ready;
	if (P.primary_ahead == 0u)  // the rule: a block per workgroup of the group
		P.primary_ahead = (workgroups + XCD_GROUPS - 1u) / XCD_GROUPS;
	args.P = P;
	if (event_before)
		(void) hipEventRecord((hipEvent_t) event_before, s);
	if (prefetch)
		hipLaunchKernelGGL((frame_kernel<AO_UNIFORM, true>), dim3(workgroups), dim3(64 * AO_WAVES), 0, s, args);
	else
		hipLaunchKernelGGL((frame_kernel<AO_UNIFORM, false>), dim3(workgroups), dim3(64 * AO_WAVES), 0, s, args);
	if (event_after)
		(void) hipEventRecord((hipEvent_t) event_after, s);
}

// `out`: this rank's 8-bit bands (local_out_rows x out_width), or null for a frame without the device resize.
void launch_finish(float *image, const void *hits, const void *occluded_of, const void *tile_base, unsigned char *out,
                   const KernelParams &P, uint32_t out_width, uint32_t n, uint32_t local_out_rows, void *stream, void *counters) {
	if (local_out_rows == 0 || out_width == 0 || n == 0)
		return;
	const bool has_ao = P.ao_mode != AO_NONE && P.ao_dirs > 0;
	if (!has_ao && !out)
		return;  // (nothing pending, nothing to filter)
	const uint32_t rows_per_band = P.part.band_tile_rows * TILE_H / n;
	const uint32_t cells_per_pixel = (n * n) | 1u;
	if (n >= 2u && cells_per_pixel <= FINISH_CELL_FLOATS) {
		// supersampled: a workgroup per run of output pixels, swept along the sub-pixel rows (finish_wide_kernel)
		uint32_t pixels_per_block = FINISH_CELL_FLOATS / cells_per_pixel;
		pixels_per_block = pixels_per_block > 256u ? 256u : pixels_per_block > 16u ? pixels_per_block & ~15u : pixels_per_block;
		hipLaunchKernelGGL(finish_wide_kernel<true>, dim3((out_width + pixels_per_block - 1u) / pixels_per_block, local_out_rows), dim3(256), 0,
		                   (hipStream_t) stream, image, (const HitRec *) hits, (const uint32_t *) occluded_of, (const uint32_t *) tile_base,
		                   out, out_width, P.height / n, P.width, n, P.tiles_x, P.part, rows_per_band,
		                   P.ao_divisor ? P.ao_divisor : 1u, pixels_per_block, (FrameCounters *) counters);
		return;
	}
	hipLaunchKernelGGL(finish_kernel, dim3((out_width + 255) / 256, local_out_rows), dim3(256), 0, (hipStream_t) stream, image,
	                   (const HitRec *) hits, (const uint32_t *) occluded_of, (const uint32_t *) tile_base, out,
	                   out_width, P.height / n,
	                   P.width, n, P.tiles_x, P.part, rows_per_band, P.ao_divisor ? P.ao_divisor : 1u, (FrameCounters *) counters);
}

// counters->occluded = the sum of the hit list's `slots` occlusion counts (stream-ordered: after the frames enqueued so far).
void launch_occluded_sum(const void *occluded_of, size_t slots, void *counters, void *stream) {
	FrameCounters *const c = (FrameCounters *) counters;
	(void) hipMemsetAsync(&c->occluded, 0, sizeof c->occluded, (hipStream_t) stream);
	if (slots == 0)
		return;
	const size_t blocks = (slots + 4095) / 4096;  // (16 counts per thread at least)
	hipLaunchKernelGGL(occluded_sum_kernel, dim3((unsigned) (blocks < 256 ? blocks : 256)), dim3(256), 0, (hipStream_t) stream,
	                   (const uint32_t *) occluded_of, slots, c);
}

void launch_resize(const float *tmp, unsigned char *out, const KernelParams &P, uint32_t out_width, uint32_t n,
                   uint32_t local_out_rows, void *stream) {
	if (local_out_rows == 0 || out_width == 0 || n == 0)
		return;
	const uint32_t rows_per_band = P.part.band_tile_rows * TILE_H / n;
	const uint32_t cells_per_pixel = (n * n) | 1u;
	if (n >= 2u && cells_per_pixel <= FINISH_CELL_FLOATS) {
		// supersampled: along the sub-pixel rows, like the frame's own finishing sweep (no tags to resolve, nothing written back)
		uint32_t pixels_per_block = FINISH_CELL_FLOATS / cells_per_pixel;
		pixels_per_block = pixels_per_block > 256u ? 256u : pixels_per_block > 16u ? pixels_per_block & ~15u : pixels_per_block;
		hipLaunchKernelGGL(finish_wide_kernel<false>, dim3((out_width + pixels_per_block - 1u) / pixels_per_block, local_out_rows), dim3(256), 0,
		                   (hipStream_t) stream, const_cast<float *>(tmp), (const HitRec *) nullptr, (const uint32_t *) nullptr,
		                   (const uint32_t *) nullptr, out, out_width, P.height / n, P.width, n, P.tiles_x, P.part, rows_per_band, 1u,
		                   pixels_per_block, (FrameCounters *) nullptr);
		return;
	}
	hipLaunchKernelGGL(resize_kernel, dim3((out_width + 255) / 256, local_out_rows), dim3(256), 0,
	                   (hipStream_t) stream, tmp, out, out_width, P.height / n, P.width, n, P.part, rows_per_band);
}

}  // namespace ocrt
