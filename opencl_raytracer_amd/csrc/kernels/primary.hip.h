// kernels/primary.hip.h -- pass 1: primary rays (closest hit, smooth normal, head-light term) and the ordering step in its tail
// (part of the one translation unit kernels.hip; see its head for the passes and the arithmetic contract)
#pragma once
#include "walk.hip.h"
#include "handoff.hip.h"

namespace ocrt {

// Tile <-> workgroup mapping shared by the passes.  A workgroup of the primary
// pass covers 2x2 tiles (16x16 sub-pixels).  Workgroups b and b+8 share an XCD
// and its L2 (MI355X_MICROARCH.md, dispatch is round-robin over XCDs), so the
// image is cut into vertical strips KernelParams::strip_tiles wide (two by default),
// strips are dealt round-robin to the 8 XCD groups, and each group walks its strips
// top to bottom, row by row: neighbouring workgroups of a group touch the same BVH
// region, while every group still sees the whole image height.  Strips of two tiles
// balance best and are right while the scene lives in the caches; a scene far beyond
// the L2s gets wider ones, so that an XCD's rays mostly meet geometry that only this
// XCD needs (with 16-pixel strips all eight fetch the same nodes from HBM).

// Kernel arguments that are READ AGAIN from the kernel-argument segment where they are used -- one scalar load each
// (asm volatile: the compiler can neither hoist it out of a loop nor merge it with another) -- instead of being held in
// scalar registers, and spilled from them to VGPR lanes, across the walks (see AoArgs).
template <uint32_t OFFSET>
__device__ __forceinline__ uint32_t cold_u32() {
	uint32_t v;
	asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "i"(OFFSET));
	return v;
}
template <uint32_t OFFSET>
__device__ __forceinline__ unsigned long long cold_u64() {
	unsigned long long v;
	asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "i"(OFFSET));
	return v;
}

// The ray passes' arguments as ONE block, the same for the primary pass, the ambient-occlusion pass and the fused
// frame kernel (kernels/frame.hip.h).  The walks leave a kernel some 40 scalar registers for everything it holds across
// them (80 per wave at 8 waves per SIMD, 22 of them the node loop's own and 14 its operands), and what does not fit is
// spilled to VGPR lanes: v_writelane / v_readlane -- VECTOR instructions, the resource the AO pass is bound by (round 3:
// 33 per packet and 9 per leaf stop or batch, a tenth of the pass's vector instructions).  So only what every leaf stop
// needs is held in registers (the two pointers at the head, node_count, the `below` limits, batch_below); every other
// argument is READ AGAIN from the kernel-argument segment where it is used -- one scalar load (cold_u32 / cold_u64 above)
// that hits the scalar cache and costs no vector issue slot.
struct FrameArgs {
	const float4 *walk_ptr, *tris_ptr;
	const float4 *nodes_ptr, *shade, *ao_table;
	float *image;
	HitRec *hits;
	uint32_t *occluded_of, *tile_hits;
	const uint32_t *order;      // the AO pass's claim order (DeviceRenderer::orderTiles)
	const uint32_t *tile_base;  // first slot of each tile in the hit list (DeviceRenderer: a prefix sum of the tiles' hit counts)
	const uint2 *tile_entry;    // per tile, 1 + ao_dirs byte ranges of the walk records: what its any-hit rays have to walk (entry_kernel)
	FrameCounters *counters;
	uint32_t *tile_cost;        // null but in a measuring frame (DeviceRenderer::measureTileCosts): per tile, the device-clock ticks its claims kept their workgroups
	const uint32_t *primary_order;  // fused frame kernel: the order its workgroups take the 2 x 2 tile blocks of the primary pass in
	const uint32_t *order_need;     // ... and, beside every entry of `order`, how many of those blocks the entries up to it need
	uint32_t *tile_ready;           // ... and per tile the frame number whose hit records are in the list (kernels/frame.hip.h)
	KernelParams P;
};
using PrimaryArgs = FrameArgs;
using AoArgs = FrameArgs;
#define OCRT_PCOLD_U32(FIELD) cold_u32<(uint32_t) offsetof(FrameArgs, FIELD)>()
#define OCRT_PCOLD_PTR(TYPE, FIELD) ((TYPE) cold_u64<(uint32_t) offsetof(FrameArgs, FIELD)>())
#define OCRT_COLD_U32(FIELD) cold_u32<(uint32_t) offsetof(FrameArgs, FIELD)>()
#define OCRT_COLD_F32(FIELD) __uint_as_float(cold_u32<(uint32_t) offsetof(FrameArgs, FIELD)>())
#define OCRT_COLD_PTR(TYPE, FIELD) ((TYPE) cold_u64<(uint32_t) offsetof(FrameArgs, FIELD)>())

// A hit sub-pixel that still waits for its ambient-occlusion factor holds, in the float image, a TAG instead of a value:
// a negative quiet NaN whose low six bits are the sub-pixel's slot in its tile's part of the hit list (no value the
// path computes is a NaN: the head-light term is clamped to [0, 1]).  The finishing kernel puts the value there.
constexpr uint32_t PENDING_TAG = 0xFFC00000u;
__device__ __forceinline__ bool is_pending(uint32_t bits) { return (bits & 0xFFFFFFC0u) == PENDING_TAG; }

// SHARED: the shared walk.  (The A/B build also instantiates the first generation, SHARED = false; two instantiations,
// so that its per-lane state stays out of the default path's register budget.)
// One tile of the primary pass, one wave.
// HANDOFF: the tile's hit records are read by another workgroup of the SAME launch (the fused frame kernel): they are
// stored device-coherently (kernels/handoff.hip.h); the caller drains the stores and raises the tile's flag.
// `cb`: this wave's LDS slice for the leaves it tests 64 pairs at a time.
// `part`: WHOLE_TILE, or 0..3 -- one QUARTER of the tile (4 x 4 sub-pixels), the other three being cast by the other waves of
// this workgroup at the same time (primary_kernel: the tiles whose packets stop at the most leaves are the critical path of
// the pass -- one wave, up to ~120 us of dependent loads where the pass takes 120 -- and a quarter's packet stops at a
// fraction of them).  The quarters meet once, for the tile's hit mask (`quarter_hits`: four LDS words): a hit's slot in the
// list is its rank among ALL the tile's hits.  Which rays share a packet never changes what a ray finds.
constexpr uint32_t WHOLE_TILE = 4u;
// (a value read from LDS at a wave-uniform address, kept in a vector register: nothing makes the compiler look for a scalar one)
__device__ __forceinline__ float lane_value(float x) {
	asm volatile("" : "+v"(x));
	return x;
}
template <bool SHARED, bool HANDOFF = false>
__device__ __forceinline__ void primary_tile(const FrameArgs &A, ClosestBatch &cb, uint32_t tile_x, uint32_t local_row, uint32_t part = WHOLE_TILE,
                                             unsigned long long *quarter_hits = nullptr) {
	// What the tile needs of the launch constants is READ HERE, by loads the compiler can neither hoist nor merge (cold_u32):
	// in the fused frame kernel this function sits inside the claim loops of a persistent workgroup, and constants held
	// across the ambient-occlusion pass's walks would be spilled to VGPR lanes there (FrameArgs); a wave of primary_kernel
	// casts one tile and reads them once either way.  (Fields used BEFORE or IN the walk; the epilogue reads its own.)
	struct {
		uint32_t tiles_x, width, height, node_count, tri_count, fast_walk, batch_below;
		float a, half_w, half_h, origin_limit, primary_below;
		Partition part;
#ifdef OCRT_DEBUG_KNOBS
		uint32_t scene_regular, leaf_min;
#endif
	} P;
	P.tiles_x = OCRT_PCOLD_U32(P.tiles_x); P.width = OCRT_PCOLD_U32(P.width); P.height = OCRT_PCOLD_U32(P.height);
	P.node_count = OCRT_PCOLD_U32(P.node_count); P.tri_count = OCRT_PCOLD_U32(P.tri_count); P.fast_walk = OCRT_PCOLD_U32(P.fast_walk);
	P.batch_below = OCRT_PCOLD_U32(P.batch_below);
	P.a = OCRT_COLD_F32(P.a); P.half_w = OCRT_COLD_F32(P.half_w); P.half_h = OCRT_COLD_F32(P.half_h);
	P.origin_limit = OCRT_COLD_F32(P.origin_limit); P.primary_below = OCRT_COLD_F32(P.primary_below);
	P.part.rank = OCRT_PCOLD_U32(P.part.rank); P.part.nranks = OCRT_PCOLD_U32(P.part.nranks); P.part.band_tile_rows = OCRT_PCOLD_U32(P.part.band_tile_rows);
#ifdef OCRT_DEBUG_KNOBS
	P.scene_regular = OCRT_PCOLD_U32(P.scene_regular); P.leaf_min = OCRT_PCOLD_U32(P.leaf_min);
#endif
	const float4 *__restrict__ const walk_ptr = A.walk_ptr, *__restrict__ const tris_ptr = A.tris_ptr;
	const float4 *__restrict__ const nodes_ptr = OCRT_PCOLD_PTR(const float4 *, nodes_ptr);  // (exact form and first-generation walk only)
	const uint32_t lane = threadIdx.x & 63u;
	const SceneViews scene = make_views(nodes_ptr, tris_ptr, P.node_count, P.tri_count);
	const uint32_t tile = local_row * P.tiles_x + tile_x;
	const uint32_t tile_y = global_tile_row(P.part, local_row);
	const uint32_t x = tile_x * TILE_W + (lane & 7u);
	const uint32_t y = tile_y * TILE_H + (lane >> 3);
	const bool active = x < P.width && y < P.height && (part == WHOLE_TILE || (((lane >> 2) & 1u) | ((lane >> 4) & 2u)) == part);
	// the float image holds this rank's bands only, one after the other: row `local_y` of it is image row `y`
	const uint32_t local_y = local_row * TILE_H + (lane >> 3);
	const uint32_t count = P.node_count;

	// reference src/intersect_kernel.cl:279-295
	float dx = ((float) x + 0.5f) / P.a - P.half_w;
	float dy = -(((float) y + 0.5f) / P.a - P.half_h);
	float dz = -1.0f;
	normalize3(dx, dy, dz);
	const Ray ray = make_ray(0.0f, 0.0f, 2.0f, dx, dy, dz);
	Hit best;
	best.distance = __builtin_inff();
	best.leaf = 0;
	best.s = best.t = 0.0f;
	best.px = best.py = best.pz = 0.0f;
	bool hit = false;
	uint32_t leaf_stops = 0u;  // leaves the tile's shared walk stopped at: how dense the geometry is along these rays
	if (SHARED) {
		const bool exact = !P.fast_walk || wave_ballot(active && !ray_is_selectable(ray, P.origin_limit)) != 0ull;
		// closest hit = minimum of (distance, reference leaf), see nearer(); reference :106-112
		auto leaf_test = [&](uint32_t leaf, bool box) {
			const float4 *tri = tris_ptr + LEAF_F4 * leaf + LEAF_TRI_F4;
			const float4 q0 = tri[0], q1 = tri[1], q2 = tri[2], q3 = tri[3];
			if (box) {
				const TriResult tr = tri_eval<true>(q0, q1, q2, q3, ray);
				if (tr.accepted) {
					hit = true;
					if (nearer(tr.distance, leaf, best)) {
						best.distance = tr.distance;
						best.leaf = leaf;
						best.s = tr.s;
						best.t = tr.t;
						best.px = tr.px; best.py = tr.py; best.pz = tr.pz;
					}
				}
			}
		};
		if (!exact) {
			// Leaves hit by few lanes are collected and tested 64 pairs at a time (see ClosestBatch).
			cb.best_key[lane] = KEY_NONE;
			if (lane < 2u)
				cb.hit_bits[lane] = 0u;
			unsigned long long my_key = KEY_NONE;  // from the leaves tested on the spot
			uint32_t waiting = 0u;
			auto key_of = [](float distance, uint32_t leaf) {
				return ((unsigned long long) __float_as_uint(distance) << 32) | leaf;
			};
			auto run_batch = [&](uint32_t n) {
				wave_lds_sync();
				const uint32_t pair = cb.entry[lane < n ? lane : 0u];
				const int owner = (int) (pair >> 26);
				Ray theirs = ray;  // (all primary rays start at the eye)
				theirs.dx = __shfl(ray.dx, owner); theirs.dy = __shfl(ray.dy, owner); theirs.dz = __shfl(ray.dz, owner);
				theirs.ix = __shfl(ray.ix, owner); theirs.iy = __shfl(ray.iy, owner); theirs.iz = __shfl(ray.iz, owner);
				if (lane < n) {
					// a candidate of the padded walk: the leaf's own box decides whether the reference tests it (:189)
					const uint32_t leaf = pair & 0x03FFFFFFu;
					const float4 lo = load_f4(scene.tris, leaf * LEAF_BYTES), hi = load_f4(scene.tris, leaf * LEAF_BYTES + 16u);
					if (exact_leaf_gate(lo, hi, theirs, P.primary_below)) {
						const uint32_t at = leaf * LEAF_BYTES + LEAF_TRI_OFFSET;
						const Candidate tr = tri_candidate(load_f4(scene.tris, at), load_f4(scene.tris, at + 16u), load_f4(scene.tris, at + 32u),
						                                   load_f4(scene.tris, at + 48u), hi.w, theirs);
						if (tr.accepted) {
							atomicMin(&cb.best_key[owner], key_of(tr.distance, leaf));
							atomicOr(&cb.hit_bits[owner >> 5], 1u << (owner & 31));
						}
					}
				}
			};
			const unsigned long long alive_mask = wave_ballot(active);
			const SignMasks sign = sign_masks(ray);
			const uint32_t variant = walk_variant(sign, alive_mask);
			const uint32_t first = 0u;  // (the plane-form records)
			const WalkRay walk_ray = make_walk_ray(ray, 1.0f);
			const uint32_t list_lds_address = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (uintptr_t) &cb.entry[0]);  // (low half of the flat address; scalar)
			const uint32_t end = first + OCRT_PCOLD_U32(P.primary_walk_bytes);
			uint32_t at = first;  // byte offset
			// What the reference does not do and no result can show: a lane that has a hit does not enter boxes that begin
			// BEHIND it.  The reference walks every box its ray meets below 100000 and keeps the minimum of (distance, leaf);
			// a triangle whose leaf box begins more than `prune_margin` + 1e-5 of the distance behind the nearest hit so far
			// cannot bring a distance that is smaller or equal (the margin: how far outside its box the reference's slack of
			// 1e-5 on s and t lets a hit lie -- scene_pack.cc -- and the rounding of the two distances).  `far_limit` is the
			// node test's upper limit, per lane; it costs the loop nothing (the operand was a scalar register).
			float far_limit = P.primary_below;
			{
				const float margin = OCRT_COLD_F32(P.prune_margin);
				const uint32_t unpruned = OCRT_PCOLD_U32(P.unpruned_bytes);
				if (lane == 0u) {
					cb.prune_margin = margin;
					cb.unpruned_bytes = unpruned;
				}
			}
			while (alive_mask != 0ull && at < end) {
				uint32_t leaf = 0u;
				unsigned long long hit_mask = 0ull;
				const uint32_t status = walk_collect<false>(variant, walk_ptr, at, walk_ray, sign, far_limit, alive_mask, hit_mask,
				                                            leaf, waiting, leaf_stops, list_lds_address, lane << 26, P.batch_below);
				if (status == 0u)
					break;
				if (status == 1u) {
					const float4 *rec = tris_ptr + LEAF_F4 * leaf;
					const float4 lo = rec[0], hi = rec[1], q0 = rec[2], q1 = rec[3], q2 = rec[4], q3 = rec[5];
					if (((hit_mask >> lane) & 1ull) && exact_leaf_gate(lo, hi, ray, P.primary_below)) {
						const Candidate tr = tri_candidate(q0, q1, q2, q3, hi.w, ray);
						if (tr.accepted) {
							hit = true;
							const unsigned long long key = key_of(tr.distance, leaf);
							my_key = key < my_key ? key : my_key;
							// (not while the walk is among the faces no box promises anything about -- make_walk_array: they
							// lie at the head of the records --; afterwards the lane's nearest hit so far counts, theirs included)
							if (at >= (uint32_t) __float_as_uint(lane_value(__uint_as_float(cb.unpruned_bytes))))
								far_limit = fminf(far_limit, __uint_as_float((uint32_t) (my_key >> 32)) * 1.00001f + lane_value(cb.prune_margin));
						}
					}
				} else {
					run_batch(64u);
					waiting -= 64u;
					if (lane < waiting)  // the pairs beyond the batch move to the front
						cb.entry[lane] = cb.entry[64u + lane];
					wave_lds_sync();
					const uint32_t nearest = (uint32_t) (cb.best_key[lane] >> 32);  // (distance bits; KEY_NONE: all ones)
					if (nearest < INF_BITS && at >= (uint32_t) __float_as_uint(lane_value(__uint_as_float(cb.unpruned_bytes))))
						far_limit = fminf(far_limit, __uint_as_float(nearest) * 1.00001f + lane_value(cb.prune_margin));
				}
				at += 32u;
			}
			if (waiting != 0u)
				run_batch(waiting);
			wave_lds_sync();
			const unsigned long long batched = cb.best_key[lane];
			const unsigned long long key = batched < my_key ? batched : my_key;
			hit = hit || ((cb.hit_bits[lane >> 5] >> (lane & 31u)) & 1u);
			// the nearest hit's barycentrics and position: the same test once more, on the ray's own lane.  (A
			// distance of +inf or NaN never satisfies the reference's `best.distance > distance`: `best` stays as it is.)
			if (hit && (uint32_t) (key >> 32) < INF_BITS) {
				const uint32_t leaf = (uint32_t) key;
				const TriResult tr = tri_test<true>(scene.tris, leaf, ray);
				best.distance = tr.distance;
				best.leaf = leaf;
				best.s = tr.s;
				best.t = tr.t;
				best.px = tr.px; best.py = tr.py; best.pz = tr.pz;
			}
		} else {
			uint32_t mine = 0u;
			uint32_t at = 0u;
			while (at < count) {
				const u32x8 node = scalar_load_node(nodes_ptr, at);
				const float4 lo = make_float4(__uint_as_float(node[0]), __uint_as_float(node[1]), __uint_as_float(node[2]), 0.0f);
				const float4 hi = make_float4(__uint_as_float(node[4]), __uint_as_float(node[5]), __uint_as_float(node[6]), 0.0f);
				const uint32_t skip = node[3], leaf = node[7];
				const bool box = exact_box(lo, hi, ray, 100000.0f, active, at, skip, mine);
				const bool any = wave_ballot(box) != 0ull;
				if (any && leaf != NONE) {
					leaf_test(leaf, box);
					++leaf_stops;
				}
				at = (uint32_t) __builtin_amdgcn_readfirstlane((int) (at + (any ? 1u : skip)));
			}
		}
	}
#ifdef OCRT_DEBUG_KNOBS
	if (!SHARED) {
		const bool regular = P.scene_regular && ray_is_regular(ray);
		uint32_t i = active ? 0u : count;
		Pending pending = { NONE, NONE };
		for (;;) {
			const unsigned long long walking = wave_ballot(can_walk(pending, i, count));
			const unsigned long long leaves = wave_ballot(pending.first != NONE);
			if (leaves != 0ull && ((uint32_t) __popcll(leaves) >= P.leaf_min || walking == 0ull)) {
				if (pending.first != NONE) {
					const TriResult tr = tri_test<true>(scene.tris, pending.first, ray);
					// closest hit = minimum of (distance, reference leaf), see nearer(); reference :106-112
					if (tr.accepted) {
						hit = true;
						if (nearer(tr.distance, pending.first, best)) {
							best.distance = tr.distance;
							best.leaf = pending.first;
							best.s = tr.s;
							best.t = tr.t;
							best.px = tr.px; best.py = tr.py; best.pz = tr.pz;
						}
					}
					pending.first = pending.second;
					pending.second = NONE;
				}
				continue;
			}
			if (walking == 0ull)
				break;
			advance_walkers(scene, ray, regular, 100000.0f, P.primary_below, count, i, pending);
			if ((uint32_t) __popcll(wave_ballot(pending.first != NONE)) < P.leaf_min)
				advance_walkers(scene, ray, regular, 100000.0f, P.primary_below, count, i, pending);
		}
	}
#endif

	// smooth normal and head-light term, reference :296-304
	float value = 0.0f;
	float nx = 0.0f, ny = 0.0f, nz = 0.0f;
	if (hit) {
		const float4 *const shade = OCRT_PCOLD_PTR(const float4 *, shade);
		const float4 n0 = shade[3 * (size_t) best.leaf + 0];
		const float4 n1 = shade[3 * (size_t) best.leaf + 1];
		const float4 n2 = shade[3 * (size_t) best.leaf + 2];
		const float b0 = 1.0f - best.s - best.t, b1 = best.s, b2 = best.t;
		nx = (n0.x * b0 + n1.x * b1) + n2.x * b2;
		ny = (n0.y * b0 + n1.y * b1) + n2.y * b2;
		nz = (n0.z * b0 + n1.z * b1) + n2.z * b2;
		normalize3(nx, ny, nz);
		value = 1.0f;
		if (OCRT_PCOLD_U32(P.shading))
			value = fminf(fmaxf(-dot3(nx, ny, nz, dx, dy, dz), 0.0f), 1.0f);
	}
	const bool want_ao = OCRT_PCOLD_U32(P.ao_mode) != (uint32_t) AO_NONE && OCRT_PCOLD_U32(P.ao_dirs) > 0u;
	const uint32_t image_width = OCRT_PCOLD_U32(P.width);
	// the tile's hits go into the tile's own 64 slots of the hit list, compacted
	unsigned long long hit_mask = wave_ballot(hit);
	if (part != WHOLE_TILE) {
		if (lane == 0u)
			quarter_hits[part] = hit_mask;
		__syncthreads();
		hit_mask = (quarter_hits[0] | quarter_hits[1]) | (quarter_hits[2] | quarter_hits[3]);
	}
	const uint32_t hit_count = (uint32_t) __popcll(hit_mask);
	const uint32_t slot_in_tile = rank_in(hit_mask);
	if (active)  // final already, or the tag that says which slot will bring the ambient-occlusion factor
		OCRT_PCOLD_PTR(float *, image)[(size_t) local_y * image_width + x] = (hit && want_ao) ? __uint_as_float(PENDING_TAG | slot_in_tile) : value;
	if (lane == 0u && part == WHOLE_TILE) {  // (a tile cast in quarters keeps the word the upload's counting pass wrote: the same hits, its whole packet's leaves)
		// hit count, and above it the tile's AO cost class 1..64 for the ordering step: its 28 AO packets
		// walk about as far as the primary packet did (correlation 0.8-0.9, tools/analysis/packet_union.cc).
		// The hit count does not predict the cost at all: a sparse tile's packets mix several directions
		// and walk as many nodes as a full tile's.
		uint32_t cost = SHARED ? leaf_stops : hit_count;
		cost = cost < 1u ? 1u : cost;
		cost = cost > 64u ? 64u : cost;
		OCRT_PCOLD_PTR(uint32_t *, tile_hits)[tile] = hit_count | (hit_count ? cost << 8 : 0u);
	}
	// The hit list holds a tile's hits at tile_base[tile] ..., in the order of the lanes.  (tile_base is the exclusive
	// prefix sum of the tiles' hit counts -- a function of scene, options and the fixed camera, counted once per upload by
	// a pass of this kernel that has no hit list yet: `hits` is null then and nothing is recorded.)
	HitRec *const hit_list = OCRT_PCOLD_PTR(HitRec *, hits);
	if (hit && want_ao && hit_list) {
		HitRec rec;
		rec.ox = best.px; rec.oy = best.py; rec.oz = best.pz;
		rec.value = value;
		rec.nx = nx; rec.ny = ny; rec.nz = nz;
		rec.pixel = local_y * image_width + x;  // index into this rank's band image
		const size_t slot = (size_t) OCRT_PCOLD_PTR(const uint32_t *, tile_base)[tile] + slot_in_tile;
		if (HANDOFF) {
			store_f4_device_coherent(&hit_list[slot], make_float4(rec.ox, rec.oy, rec.oz, rec.value));
			store_f4_device_coherent((char *) &hit_list[slot] + 16, make_float4(rec.nx, rec.ny, rec.nz, __uint_as_float(rec.pixel)));
			store_u32_device_coherent(&OCRT_PCOLD_PTR(uint32_t *, occluded_of)[slot], 0u);
		} else {
			hit_list[slot] = rec;
			OCRT_PCOLD_PTR(uint32_t *, occluded_of)[slot] = 0u;
		}
	}
}

// ---------------------------------------------------------------------------
// Pass 1: primary rays.  Four waves per workgroup, one tile each; they never meet.
//
// Until round 4 this kernel had a tail: the last workgroup of each XCD group to finish sorted the group's tiles by
// their ambient-occlusion cost for the next pass.  Camera, scene and options are fixed per upload, so that order is
// the same in every frame: it is now made once per upload, on the host (DeviceRenderer::orderTiles), and the
// hand-over between workgroups inside the kernel (device-coherent tile words, a `done` counter per group) is gone
// with it.  What is left of the tile word is a statistic: hit count | cost class, read by the host on demand.
// ---------------------------------------------------------------------------
#ifndef OCRT_PRIMARY_WAVES
#define OCRT_PRIMARY_WAVES 4
#endif
constexpr uint32_t PRIMARY_WAVES = OCRT_PRIMARY_WAVES;  // 4, 8 or 16: a workgroup covers a block of tiles 2 wide and PRIMARY_WAVES / 2 high
constexpr uint32_t PRIMARY_ROWS = PRIMARY_WAVES / 2u;
constexpr uint32_t PRIMARY_NO_ENTRY = 0xFFFFFFFFu;  // (an empty place in primary_kernel's list)

template <bool SHARED>
__global__ __launch_bounds__(64 * PRIMARY_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void primary_kernel(PrimaryArgs A) {
	__shared__ ClosestBatch closest_batches[PRIMARY_WAVES];
	const uint32_t wave = threadIdx.x >> 6;
	if (blockIdx.x == 0u && threadIdx.x < XCD_GROUPS) {
		FrameCounters *const counters = A.counters;
		// what the LATER kernels of this frame add to or count up (nobody touches it before this kernel has ended)
		counters->queue[threadIdx.x].head = counters->queue[threadIdx.x].split_units;
		counters->queue[threadIdx.x].split_head = 0u;
		if (threadIdx.x == 0u) {
			counters->tick_begin = __builtin_amdgcn_s_memrealtime();
			counters->occluded = 0ull;
			counters->tick_ao_end = 0ull;
		}
	}
	const uint32_t group = blockIdx.x & (XCD_GROUPS - 1u), seq = blockIdx.x >> 3;
	__shared__ unsigned long long quarter_hits[4];
#ifdef OCRT_PRIMARY_TICKS  // (a probe build: how long each tile's wave lives, in 10 ns ticks, where a measuring frame's costs go)
	__shared__ uint32_t tick0[PRIMARY_WAVES][2];  // (in LDS: scalar registers held across the walk would be spilled there)
#endif
	uint32_t tile_x, local_row, part = WHOLE_TILE;
	bool there;
	if (PRIMARY_WAVES == 4u && A.primary_order != nullptr) {
		// With a list of the group's work by falling cost (DeviceRenderer::orderPrimaryBlocks, once per upload: the leaves each
		// tile's primary packet stops at are the same in every frame) workgroup `seq` of the group takes entry `seq`: the
		// model's tiles -- 50-120 us of dependent loads each -- start first, the background's fill in behind them.  An entry
		// is a 2 x 2 block of tiles, a wave each (bits 26-29: the waves with nothing to do), or ONE tile for the four waves,
		// a quarter each (bit 30).
		const uint32_t entry = A.primary_order[group * A.P.primary_list_stride + seq];
		there = entry != PRIMARY_NO_ENTRY;
		const bool quarters = (entry >> 30) & 1u;
		tile_x = (entry & 0x1FFFu) + (quarters ? 0u : wave & 1u);
		local_row = ((entry >> 13) & 0x1FFFu) + (quarters ? 0u : wave >> 1);
		there = there && (quarters || !((entry >> (26u + wave)) & 1u));
		part = there && quarters ? wave : WHOLE_TILE;
	} else {
		const uint32_t strip_tiles = A.P.strip_tiles, columns = strip_tiles >> 1;  // (a workgroup is two tiles wide)
		const uint32_t strips = (A.P.tiles_x + strip_tiles - 1u) / strip_tiles;
		const uint32_t row_blocks = (A.P.local_tile_rows + PRIMARY_ROWS - 1u) / PRIMARY_ROWS;
		const uint32_t strips_here = (strips + XCD_GROUPS - 1u - group) >> 3;
		const uint32_t per_strip = row_blocks * columns;
		const uint32_t strip_index = seq / per_strip;
		const uint32_t rest = seq - strip_index * per_strip;
		const uint32_t row_block = rest / columns;
		tile_x = strip_tiles * (group + XCD_GROUPS * strip_index) + 2u * (rest - row_block * columns) + (wave & 1u);
		local_row = PRIMARY_ROWS * row_block + (wave >> 1);
		there = seq < strips_here * per_strip;
	}
	there = there && tile_x < A.P.tiles_x && local_row < A.P.local_tile_rows;
#ifdef OCRT_PRIMARY_TICKS
	if ((threadIdx.x & 63u) == 0u) {
		tick0[wave][0] = (uint32_t) __builtin_amdgcn_s_memrealtime();
		tick0[wave][1] = there && (part == WHOLE_TILE || part == 0u) ? local_row * A.P.tiles_x + tile_x : 0xFFFFFFFFu;
	}
#endif
	if (there)
		primary_tile<SHARED>(A, closest_batches[wave], tile_x, local_row, part, quarter_hits);
#ifdef OCRT_PRIMARY_TICKS
	uint32_t *const ticks_out = *(uint32_t *const volatile *) ((const char *) __builtin_amdgcn_kernarg_segment_ptr() + offsetof(FrameArgs, tile_cost));
	if (ticks_out && (threadIdx.x & 63u) == 0u && tick0[threadIdx.x >> 6][1] != 0xFFFFFFFFu)
		ticks_out[tick0[threadIdx.x >> 6][1]] = (uint32_t) __builtin_amdgcn_s_memrealtime() - tick0[threadIdx.x >> 6][0];
#endif
}

// ---------------------------------------------------------------------------
// The primary pass INSIDE the fused frame kernel (kernels/frame.hip.h): the group's 2 x 2 tile blocks are taken one by
// one from FrameArgs::primary_order -- a cursor per XCD group, the blocks in the order the ambient-occlusion pass will
// want their tiles, the background's last -- by whichever persistent workgroup finds the cursor BELOW ITS TARGET when it
// has made an ambient-occlusion claim: target = the blocks the claim's tiles need + primary_ahead (the whole list once the
// group's ambient-occlusion queue is drained).  So the primary work stays a little ahead of the any-hit work all through
// the frame instead of before it: at any time a few of a CU's waves walk primary packets -- latency-bound, ~50 dependent
// loads and ~30 leaf stops per model tile -- beside the vector-issue-bound any-hit packets of the others.
// A workgroup that takes a block casts its primary rays (one wave per tile, as in primary_kernel) and hands the tiles'
// hit records over: every wave drains its stores, the workgroup meets, one store instruction raises the tiles' flags
// (kernels/handoff.hip.h).  Returning from here means: the cursor has passed the target, i.e. every block the claim needs
// is in the hands of a workgroup that is RUNNING -- which is what makes waiting for a flag safe.
// `cb`: this wave's LDS slice for the leaves tested in batches (nothing of the caller's lives there at this point);
// `wg_primary`: two LDS words -- [0] the block taken, [1] the target (written by the caller before its barrier).
// ---------------------------------------------------------------------------
constexpr uint32_t NO_BLOCK = 0xFFFFFFFFu;
__device__ __forceinline__ void primary_top_up(const FrameArgs &A, uint32_t group, ClosestBatch &cb, unsigned int *wg_primary) {
	const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) threadIdx.x) >> 6;
	for (;;) {
		if (wave == 0u && fresh_lane() == 0u) {
			FrameCounters *const counters = OCRT_PCOLD_PTR(FrameCounters *, counters);
			const uint32_t target = wg_primary[1];
			uint32_t got = NO_BLOCK;
			// (a load first: a cursor beyond the target is not worth an atomic -- the common case)
			if (target != 0u && __hip_atomic_load(&counters->queue[group].primary_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
				got = atomicAdd(&counters->queue[group].primary_head, 1u);
				if (got >= counters->queue[group].primary_blocks)
					got = NO_BLOCK;
			}
			wg_primary[0] = got;
		}
		__syncthreads();
		const uint32_t index = (uint32_t) __builtin_amdgcn_readfirstlane((int) wg_primary[0]);
		if (index == NO_BLOCK)
			break;  // (the same for all four waves; the word is not written again before the caller's next barrier)
		// where the group's blocks start in primary_order: the groups' segments follow each other, each as long as the
		// group's share of the strips x the block rows x the block columns of a strip
		uint32_t segment = 0u;
		{
			const uint32_t strip_tiles = OCRT_PCOLD_U32(P.strip_tiles), columns = strip_tiles >> 1;
			const uint32_t strips = (OCRT_PCOLD_U32(P.tiles_x) + strip_tiles - 1u) / strip_tiles;
			const uint32_t row_blocks = (OCRT_PCOLD_U32(P.local_tile_rows) + 1u) >> 1;
			for (uint32_t g = 0; g < group; ++g)
				segment += ((strips + XCD_GROUPS - 1u - g) >> 3) * row_blocks * columns;
		}
		const uint32_t block = (uint32_t) __builtin_amdgcn_readfirstlane((int) OCRT_PCOLD_PTR(const uint32_t *, primary_order)[segment + index]);
		const uint32_t tile_x = (block & 0xFFFFu) + (wave & 1u), local_row = (block >> 16) + (wave >> 1);
		if (tile_x < OCRT_PCOLD_U32(P.tiles_x) && local_row < OCRT_PCOLD_U32(P.local_tile_rows))
			primary_tile<true, true>(A, cb, tile_x, local_row);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's records have left)
		__syncthreads();
		if (threadIdx.x < 4u) {  // one lane per tile of the block, one store instruction: the tiles' flags
			const uint32_t x = (block & 0xFFFFu) + (threadIdx.x & 1u), row = (block >> 16) + (threadIdx.x >> 1);
			if (x < OCRT_PCOLD_U32(P.tiles_x) && row < OCRT_PCOLD_U32(P.local_tile_rows))
				store_u32_device_coherent(&OCRT_PCOLD_PTR(uint32_t *, tile_ready)[row * OCRT_PCOLD_U32(P.tiles_x) + x], frame_number(OCRT_PCOLD_PTR(FrameCounters *, counters)));
		}
	}
}

}  // namespace ocrt
