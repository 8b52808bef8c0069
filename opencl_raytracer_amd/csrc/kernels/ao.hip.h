// kernels/ao.hip.h -- pass 2: ambient occlusion (persistent workgroups, any-hit packets)
// (part of the one translation unit kernels.hip; see its head for the passes and the arithmetic contract)
#pragma once
#include "walk.hip.h"
#include "primary.hip.h"

namespace ocrt {

// xorshift128 of the RANDOM hemisphere sampler, reference src/intersect_kernel.cl:128-152.
struct Rng {
	uint32_t x, y, z, w;
};
__device__ __forceinline__ uint32_t rng_next(Rng &v) {
	const uint32_t t = v.x ^ (v.x << 11u);
	v.x = v.y;
	v.y = v.z;
	v.z = v.w;
	return v.w = v.w ^ (v.w >> 19u) ^ (t ^ (t >> 8u));
}
__device__ __forceinline__ Rng rng_seed(uint32_t seed) {
	Rng v;
	v.x = (123456789u ^ seed) * 88675123u;
	v.y = (362436069u ^ seed) * 123456789u;
	v.z = (521288629u ^ seed) * 362436069u;
	v.w = (88675123u ^ seed) * 521288629u;
	rng_next(v);
	return v;
}
__device__ __forceinline__ float rng_float(Rng &v) { return 2.32830643653869629E-10f * rng_next(v); }

// ---------------------------------------------------------------------------
// Pass 2: ambient occlusion.  Persistent, independent waves; one tile at a time.
// ---------------------------------------------------------------------------
constexpr uint32_t AO_WAVES = AO_WORKGROUP_WAVES;  // (workgroups per CU: DeviceRenderer::aoWorkgroups, 8 for a host alone)

// LDS slice of one wave: the tile's hit table, structure of arrays and lane-major
// so that consecutive hits sit in consecutive banks.
struct TileShared {
	float frame[12][64];  // origin xyz, basis_x xyz, basis_y xyz, basis_z xyz
	unsigned int occluded[64];
	unsigned int pixel[64];  // RANDOM mode: the sub-pixel's image index seeds its generator
	LeafBatch batch;
};


// MODE is AO_UNIFORM or AO_RANDOM: two instantiations, so that the RANDOM sampler's
// code and registers stay out of the default path.
#ifdef OCRT_STAMPS
#define OCRT_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define OCRT_STAMP_ADD(slot, value) stamp_acc[slot] += (unsigned long long) (value)
#else
#define OCRT_STAMP(var)
#define OCRT_STAMP_ADD(slot, value)
#endif

// PREFETCH: the node loop touches a pair's two successors ahead of time (OCRT_PF_SUCCESSORS): a launch-time choice.
// FUSED: the pass runs inside the fused frame kernel (kernels/frame.hip.h): with every claim the workgroup first takes
// the group's PRIMARY work as far as the claim's tiles need it (primary_top_up), and a tile's hit records are only read
// once the workgroup that cast its primary rays has said they are there (tile_is_ready).
// `wg_claim`: the workgroup's current claim in LDS -- first unit, units per wave, end (if dealt by cursor), cursor;
// [4..7]: the claim before it, for measureTileCosts.
template <int MODE, bool SHARED, bool PREFETCH, bool FUSED>
__device__ __forceinline__ void ao_pass(const FrameArgs &A, TileShared *shared_tiles, unsigned int *wg_claim, unsigned int *wg_primary) {
	(void) wg_primary;
	const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) threadIdx.x) >> 6;  // (scalar)
	TileShared &sh = shared_tiles[wave];
	const float4 *__restrict__ const walk_ptr = A.walk_ptr, *__restrict__ const tris_ptr = A.tris_ptr;
	const uint32_t count = A.P.node_count;
#ifdef OCRT_DEBUG_KNOBS  // (the first-generation walk of the A/B build uses the arguments freely: its register budget is nobody's concern)
	const KernelParams &P = A.P;
	const SceneViews scene = make_views(A.nodes_ptr, A.tris_ptr, A.P);
#endif

#ifndef OCRT_STAMPS
	unsigned long long *walk_prof = nullptr;
#endif
#ifdef OCRT_STAMPS
	unsigned long long stamp_acc[6] = { 0, 0, 0, 0, 0, 0 };
	const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
	unsigned long long t_last_claim = t_begin;
	unsigned long long walk_prof_store[7] = { 0, 0, 0, 0, 0, 0, 0 };
	unsigned long long *walk_prof = walk_prof_store;
#endif
	if (blockIdx.x == 0u && threadIdx.x == 0u) {  // (when the pass began, by the device's clock: frames replayed from a graph have no events inside)
		FrameCounters *const counters = OCRT_COLD_PTR(FrameCounters *, counters);
		const unsigned long long now = (unsigned long long) __builtin_amdgcn_s_memrealtime();
		__hip_atomic_store(&counters->tick_ao_begin, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (FUSED) {  // (the frame begins here; the sums only the finishing kernel's successors add to)
			counters->tick_begin = now;
			counters->occluded = 0ull;
			counters->tick_ao_end = 0ull;
		}
	}
	bool gave_up = false;  // (FUSED: this wave has waited in vain for a tile once -- tile_is_ready)
	(void) gave_up;
	if (threadIdx.x == 0u)
		wg_claim[4] = wg_claim[5] = 0u;  // (no claim before the first; read by the same thread only)
	// Workgroups b and b+8 share an XCD: start with that group's queue, then help the others.
	const uint32_t home = blockIdx.x & (XCD_GROUPS - 1u);
	for (uint32_t turn = 0; turn < XCD_GROUPS; ++turn) {
		const uint32_t group = (home + turn) & (XCD_GROUPS - 1u);
		uint32_t segment = 0u;
		{
			const uint32_t strip_tiles = OCRT_COLD_U32(P.strip_tiles);
			const uint32_t strips = (OCRT_COLD_U32(P.tiles_x) + strip_tiles - 1u) / strip_tiles, rows = OCRT_COLD_U32(P.local_tile_rows);
			for (uint32_t g = 0; g < group; ++g)
				segment += ((strips + XCD_GROUPS - 1u - g) >> 3) * strip_tiles * rows;
		}
		// the group's work in units of (tile, table direction), tile-major
		uint32_t units, claim_max, split_units;
		{
		FrameCounters *const counters = OCRT_COLD_PTR(FrameCounters *, counters);
		const uint32_t ao_dirs = OCRT_COLD_U32(P.ao_dirs);
		// (the heaviest tiles of the list, by measured cost: claimed half a tile at a time -- below)
		split_units = (uint32_t) __builtin_amdgcn_readfirstlane((int) counters->queue[group].split_units);
		// (loads through a re-read pointer are vector loads -- the compiler cannot know the memory to be constant --: what they
		// return is made scalar again by hand)
		const uint32_t queued_tiles = (uint32_t) __builtin_amdgcn_readfirstlane((int) counters->queue[group].work_tiles);
		units = queued_tiles * ao_dirs;
		// A wave's largest claim.  A quarter of a tile's directions, so that the workgroup's four waves take ONE tile
		// together (the best locality, and the finest balance); whole tiles per wave where packets are cheap and
		// plentiful -- a tile's mean cost class (the leaves its primary packet stopped at) below 8 and 512 or more
		// units per wave: 16+ samples per pixel -- because there the ~12 us of set-up per claim (hit records,
		// tangent frames) weigh more than the locality.  Swept per workload: profiles/r02_notes.md.
		claim_max = OCRT_COLD_U32(P.ao_claim_max);
		if (claim_max == 0u) {
			const uint32_t tiles = queued_tiles, cost = (uint32_t) __builtin_amdgcn_readfirstlane((int) counters->queue[group].cost_sum);
			const uint32_t claim_div = OCRT_COLD_U32(P.ao_claim_div);
			const bool cheap_and_plenty = cost < 8u * tiles && units >= 512u * claim_div;
			const uint32_t quarter = (ao_dirs + AO_WAVES - 1u) / AO_WAVES;
			// ... and less than a quarter where work is scarce (one GPU's share of a frame split eight ways holds 8
			// units per wave): half a quarter below 24 units per wave, a third below 12 -- the frame then ends when its
			// heaviest tile does, and more waves should share that one (tools/partition_probe.py: -13 % at 1/8).  Not
			// where other frames run beside this one: what a pass leaves idle at its end is theirs, and the smaller
			// claims only cost (an eighth of the headline frame, six frames in flight: 0.25 ms with them, 0.21 without).
			const uint32_t per_wave = OCRT_COLD_U32(P.shared_device) ? 24u : units / claim_div;
			claim_max = cheap_and_plenty ? ao_dirs : per_wave < 12u ? quarter / 3u : per_wave < 24u ? (quarter + 1u) / 2u : quarter;
			claim_max = claim_max < 1u ? 1u : claim_max;
		}
		}
		for (;;) {
			OCRT_STAMP(t_claim);
			// The WORKGROUP claims (thread 0: a plain load first -- most visits to a foreign group find its queue
			// drained, and a load does not queue up behind the other workgroups' atomics --, then one returning
			// atomic), and its four waves share the claim: they then work on the same tile, or on neighbouring
			// ones, at the same time, and share its nodes in the CU's scalar cache (63 % of the scalar loads of the
			// per-wave claims missed it, profiles/r02_notes.md) and its hit records in L2.  A claim is four times
			// claim_max units, to the end of the queue; how the waves divide it is decided below.  (Guided self-scheduling -- the share
			// shrinking to 1/ao_guide of what is left per wave of the group -- is kept behind OCRT_AO_GUIDE: the
			// single directions it hands out at the end cost a claim each, two barriers and a tile set-up, and
			// lengthened the pass by 3-5 %; the costly tiles are claimed first anyway: order_group.)
			uint32_t per_wave = 0u, first = units;  // (wave 0's, scalar; in registers until the siblings are done with the last claim)
			uint32_t claim_limit = units;           // (where this claim must end at the latest)
			if (wave == 0u) {
				FrameCounters *const counters = OCRT_COLD_PTR(FrameCounters *, counters);
				// The group's HEAVIEST tiles first, half a tile per claim (a cursor of their own: the two kinds of claims never
				// straddle each other's tiles).  A tile whose packets keep a workgroup for a quarter of the pass or more decides
				// when the pass ends -- at 600 x 600 -s 4 the costliest tile takes as long as the pass's ideal length, and so it
				// does in one GPU's share of a frame split eight ways --: two workgroups share such a tile.  Which tiles: by
				// measured cost, per upload (DeviceRenderer::orderByMeasuredCost).
				if (split_units != 0u) {
					uint32_t seen = split_units;
					if (fresh_lane() == 0u)
						seen = __hip_atomic_load(&counters->queue[group].split_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					seen = (uint32_t) __builtin_amdgcn_readfirstlane((int) seen);
					if (seen < split_units) {
						const uint32_t half = OCRT_COLD_U32(P.ao_dirs) >> 1;
						uint32_t got = split_units;
						if (fresh_lane() == 0u)
							got = atomicAdd(&counters->queue[group].split_head, half);
						got = (uint32_t) __builtin_amdgcn_readfirstlane((int) got);
						if (got < split_units) {
							first = got;
							per_wave = (half + AO_WAVES - 1u) / AO_WAVES;
							claim_limit = got + half;
						}
					}
				}
				uint32_t seen = units;
				if (first >= units && fresh_lane() == 0u)
					seen = __hip_atomic_load(&counters->queue[group].head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				seen = (uint32_t) __builtin_amdgcn_readfirstlane((int) seen);
				if (seen < units) {
					const uint32_t guide = OCRT_COLD_U32(P.ao_guide);
					per_wave = guide ? (units - seen) / guide : claim_max;
					per_wave = per_wave < 1u ? 1u : per_wave > claim_max ? claim_max : per_wave;
					uint32_t got = 0u;
					if (fresh_lane() == 0u)
						got = atomicAdd(&counters->queue[group].head, per_wave * AO_WAVES);
					first = (uint32_t) __builtin_amdgcn_readfirstlane((int) got);
				}
			}
			__syncthreads();  // (everybody is done with the previous claim: its words in LDS, the cursor among them, are free)
			if (wave == 0u && fresh_lane() == 0u) {
				// A claim that lies in ONE tile (the rule: four quarters of a tile's directions) is not dealt out in fixed
				// quarters: the four waves take its directions from a cursor in LDS, a packet's worth at a time, so that they
				// finish within a packet of each other -- with fixed quarters a wave spent 12-18 % of its life waiting for the
				// slowest sibling at the barrier (profiles/r02_notes.md; what the cursor buys and where it does not:
				// profiles/r03_notes.md).  Other claims (whole tiles per wave, the short ones of scarce work) keep fixed shares.
				// wg_claim[2] = the claim's end, 0 for fixed shares.
				const uint32_t end = first + per_wave * AO_WAVES < claim_limit ? first + per_wave * AO_WAVES : claim_limit;
				const uint32_t ao_dirs = OCRT_COLD_U32(P.ao_dirs);
				const bool one_tile = SHARED && first < units && first / ao_dirs == (end - 1u) / ao_dirs && ao_dirs < 0x8000u;
				wg_claim[0] = first;
				wg_claim[1] = per_wave;
				wg_claim[2] = one_tile ? end : 0u;
				// the cursor, in directions of the tile: end << 16 | next (both 0 for fixed shares: nothing to take)
				const uint32_t base = first - first / ao_dirs * ao_dirs;
				wg_claim[3] = one_tile ? (base + (end - first)) << 16 | base : 0u;
				if (FUSED) {
					// how far the group's PRIMARY work has to be taken before this claim's tiles may be waited for: the blocks the
					// list needs up to the claim's last tile, and primary_ahead more; all of it once this queue is drained
					uint32_t target = OCRT_COLD_PTR(FrameCounters *, counters)->queue[group].primary_blocks;
					if (first < units) {
						const uint32_t need = OCRT_COLD_PTR(const uint32_t *, order_need)[segment + (end - 1u) / ao_dirs] + OCRT_COLD_U32(P.primary_ahead);
						target = need < target ? need : target;
					}
					wg_primary[1] = target;
				}
				// A measuring frame (once per upload, DeviceRenderer::measureTileCosts): how long the claim BEFORE this one kept
				// the workgroup -- every wave has left it, that is what the barrier above says -- goes to its tiles, by their
				// share of its units; this claim's first unit, its end and the clock go where the next one finds them.
				uint32_t *const tile_cost = OCRT_COLD_PTR(uint32_t *, tile_cost);
#ifdef OCRT_PRIMARY_TICKS
				if (false) {
#else
				if (tile_cost) {
#endif
					const uint32_t now = (uint32_t) __builtin_amdgcn_s_memrealtime();
					const uint32_t before_first = wg_claim[4], before_end = wg_claim[5], before_clock = wg_claim[6], before_segment = wg_claim[7];
					if (before_end > before_first) {
						const uint32_t ticks = now - before_clock, all = before_end - before_first;
						const uint32_t *const order = OCRT_COLD_PTR(const uint32_t *, order);
						for (uint32_t u = before_first; u < before_end;) {
							const uint32_t t = u / ao_dirs, stop = (t + 1u) * ao_dirs < before_end ? (t + 1u) * ao_dirs : before_end;
							atomicAdd(&tile_cost[order[before_segment + t] & 0x03FFFFFFu], (uint32_t) ((unsigned long long) ticks * (stop - u) / all));
							u = stop;
						}
					}
					wg_claim[4] = first < units ? first : 0u;
					wg_claim[5] = first < units ? end : 0u;
					wg_claim[6] = now;
					wg_claim[7] = segment;
				}
			}
			__syncthreads();
			if (FUSED)  // (returns when the blocks this claim needs are in the hands of RUNNING workgroups, this one's included)
				primary_top_up(A, group, *(ClosestBatch *) &shared_tiles[wave], wg_primary);
			const uint32_t wg_claimed = (uint32_t) __builtin_amdgcn_readfirstlane((int) wg_claim[0]);
			const uint32_t want = (uint32_t) __builtin_amdgcn_readfirstlane((int) wg_claim[1]);
			if (wg_claimed >= units)
				break;  // (the same for all four waves)
			// fixed shares: this wave's quarter; a claim with a cursor: one job, the tile, for every wave
			const bool dealt_by_cursor = __builtin_amdgcn_readfirstlane((int) wg_claim[2]) != 0;
			const uint32_t claimed = dealt_by_cursor ? wg_claimed : wg_claimed + wave * want;
			if (claimed >= units)
				continue;  // nothing left for this wave; it meets the others again at the next claim
			const uint32_t claim_end = dealt_by_cursor ? claimed + 1u : claimed + want < units ? claimed + want : units;
			OCRT_STAMP(t_claimed);
			OCRT_STAMP_ADD(0, t_claimed - t_claim);
#ifdef OCRT_STAMPS
			t_last_claim = t_claimed;
#endif
			for (uint32_t unit = claimed; unit < claim_end;) {
				OCRT_STAMP(t_job);
				// job = (tile, direction range); a claim that runs over a tile's last direction goes on in
				// the next tile.  Neighbouring claims work on the same tile: its hit records are shared in L2.
				uint32_t tile_index, dir0, n_dirs;
				{
					const uint32_t ao_dirs = OCRT_COLD_U32(P.ao_dirs);
					tile_index = unit / ao_dirs;
					dir0 = unit - tile_index * ao_dirs;
					n_dirs = ao_dirs - dir0 < claim_end - unit ? ao_dirs - dir0 : claim_end - unit;
				}
				unit += n_dirs;
				const uint32_t entry = (uint32_t) __builtin_amdgcn_readfirstlane((int) OCRT_COLD_PTR(const uint32_t *, order)[segment + tile_index]);
				const uint32_t tile = entry & 0x03FFFFFFu;
				const uint32_t hit_count = (entry >> 26) + 1u;
				uint32_t total = hit_count * n_dirs;  // the rays of this piece of the job: directions dir0 ...
				// A claim dealt by the cursor: the next piece of the tile's directions -- enough to fill a packet, twice that
				// at most (64 >> floor(log2(hit_count)) directions) -- or nothing, if the siblings have taken them all.
				auto take_from_cursor = [&]() {
					const uint32_t chunk = 64u >> (31u - (uint32_t) __builtin_clz(hit_count));
					// one LDS atomic, issued by lane 0 alone: the word holds the claim's end above its cursor (directions of the tile)
					uint32_t word, scratch;
					unsigned long long saved;
					asm volatile("s_mov_b64 %[saved], exec\n\t"
					             "s_mov_b64 exec, 1\n\t"
					             "v_mov_b32 %[scratch], %[address]\n\t"
					             "v_mov_b32 %[word], %[chunk]\n\t"
					             "ds_add_rtn_u32 %[word], %[scratch], %[word]\n\t"
					             "s_waitcnt lgkmcnt(0)\n\t"
					             "s_mov_b64 exec, %[saved]"
					             : [word] "=&v"(word), [scratch] "=&v"(scratch), [saved] "=&s"(saved)
					             : [address] "s"((uint32_t) (uintptr_t) &wg_claim[3]), [chunk] "s"(chunk)
					             : "memory");
					word = (uint32_t) __builtin_amdgcn_readfirstlane((int) word);
					const uint32_t at = word & 0xFFFFu, end = word >> 16;
					const uint32_t left = at < end ? end - at : 0u;
					dir0 = at;
					total = hit_count * (left < chunk ? left : chunk);
				};
				if (SHARED && __builtin_amdgcn_readfirstlane((int) wg_claim[2]) != 0) {
					take_from_cursor();
					if (total == 0u)
						continue;  // (not even the tile's table is needed)
				}

				// ---- the tile's tangent frames -> this wave's LDS slice (reference :215-236) ----
				{
				const uint32_t lane = fresh_lane();  // (recomputed where it is needed: no register held across the walks)
				if (FUSED)  // (the records come from another workgroup of this very launch)
					gave_up = !tile_is_ready(OCRT_COLD_PTR(const uint32_t *, tile_ready), tile, OCRT_COLD_PTR(FrameCounters *, counters), gave_up) || gave_up;
				if (lane < hit_count) {
					const size_t slot = (size_t) OCRT_COLD_PTR(const uint32_t *, tile_base)[tile] + lane;
					const float4 *const hits = OCRT_COLD_PTR(const float4 *, hits);
					float4 q0, q1;
					if (FUSED) {
						load_2f4_device_coherent(hits + 2 * slot, q0, q1);
					} else {
						q0 = hits[2 * slot];
						q1 = hits[2 * slot + 1];
					}
					float nx = q1.x, ny = q1.y, nz = q1.z;
					// p = point + normal * (1.0f / 100000.0f)
					const float eps = 1.0f / 100000.0f;
					sh.frame[0][lane] = q0.x + nx * eps;
					sh.frame[1][lane] = q0.y + ny * eps;
					sh.frame[2][lane] = q0.z + nz * eps;
					sh.pixel[lane] = __float_as_uint(q1.w);
					if (MODE == AO_RANDOM)
						normalize3(nx, ny, nz);  // hemisphere_sampler normalises once more, reference :155
					// tangent frame: the smallest |component| of the normal is replaced by 1
					float hx = nx, hy = ny, hz = nz;
					const float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
					if (ax <= ay && ax <= az)
						hx = 1.0f;
					else if (ay <= ax && ay <= az)
						hy = 1.0f;
					else if (az <= ax && az <= ay)
						hz = 1.0f;
					// basis_x = normalize(cross(h, basis_y)), basis_z = normalize(cross(basis_x, basis_y))
					float bxx, bxy, bxz;
					cross3(hx, hy, hz, nx, ny, nz, bxx, bxy, bxz);
					normalize3(bxx, bxy, bxz);
					float bzx, bzy, bzz;
					cross3(bxx, bxy, bxz, nx, ny, nz, bzx, bzy, bzz);
					normalize3(bzx, bzy, bzz);
					sh.frame[3][lane] = bxx; sh.frame[4][lane] = bxy; sh.frame[5][lane] = bxz;
					sh.frame[6][lane] = nx;  sh.frame[7][lane] = ny;  sh.frame[8][lane] = nz;
					sh.frame[9][lane] = bzx; sh.frame[10][lane] = bzy; sh.frame[11][lane] = bzz;
				}
				sh.occluded[lane] = 0u;
				if (lane < 2u)
					sh.batch.occluded_bits[lane] = 0u;
				}
				wave_lds_sync();
				OCRT_STAMP(t_frames);
				OCRT_STAMP_ADD(1, t_frames - t_job);

				// ---- the tile's hit_count * ao_dirs any-hit rays (reference :237-255).  Queue
				// order is direction-major, so neighbouring lanes cast the same table direction
				// from neighbouring pixels. ----
				uint32_t h = 0;
				Ray ray;
				bool tame = false;  // (wave-uniform) every ray of the packet set up last is "tame": ray_is_tame
#ifdef OCRT_DEBUG_KNOBS
				uint32_t next = 0u;  // wave-uniform queue head
				uint32_t i = count;
				Pending pending = { NONE, NONE };
				bool regular = true;
#endif

				// ray number `item` of the job -> this lane
				// `whole` (wave-uniform): the tile is full and the 64 rays are one table direction, `shared_dir`
				auto setup_ray = [&](uint32_t item, bool whole, const float4 shared_dir) {
					uint32_t k;
					float xs = shared_dir.x, ys = shared_dir.y, zs = shared_dir.z;
					bool along_normal = false;
					if (whole) {
						k = item >> 6;
						h = item & 63u;
					} else {
						k = item / hit_count;
						h = item - k * hit_count;
						if (MODE == AO_UNIFORM) {
							const float4 dir = OCRT_COLD_PTR(const float4 *, ao_table)[dir0 + k];
							xs = dir.x; ys = dir.y; zs = dir.z;
						}
					}
					if (MODE != AO_UNIFORM) {
						// RANDOM (reference :153-183, :257-276): ray 0 goes along the normal, ray
						// j >= 1 uses draws 2j-2 and 2j-1 of the sub-pixel's generator.  Device libm
						// rounds differently from the host's: this mode is outside the bit-exact contract with the CPU
						// -- but not with the reference kernel on this GPU: in the test-only library-builtins build the
						// trigonometry below is ROCm's OpenCL library's own (OCRT_SIN ..., kernels/common.hip.h) and the
						// frame equals the reference's bit for bit (tests/test_ocml_pin.py).
						const uint32_t j = dir0 + k;
						along_normal = j == 0u;
						// the generator is seeded with the sub-pixel's index in the WHOLE image (reference :169, :279-281)
						const uint32_t local_y = sh.pixel[h] / A.P.width, x = sh.pixel[h] - local_y * A.P.width;
						const uint32_t y = global_tile_row(A.P.part, local_y / TILE_H) * TILE_H + (local_y & (TILE_H - 1u));
						Rng rng = rng_seed(536870923u * (y * A.P.width + x));
						for (uint32_t skip = 1; skip < j; ++skip) {
							rng_next(rng);
							rng_next(rng);
						}
						const float xi1 = rng_float(rng);
						const float xi2 = rng_float(rng);
						const float theta = OCRT_ACOS(sqrtf(1.0f - xi1));
						const float phi = (float) (2.0 * (double) xi2);
						xs = OCRT_SIN(theta) * OCRT_COSPI(phi);
						ys = OCRT_COS(theta);
						zs = OCRT_SIN(theta) * OCRT_SINPI(phi);
					}
					// ray_dir = basis_x * xs + basis_y * ys + basis_z * zs, lane by lane
					float rx = (sh.frame[3][h] * xs + sh.frame[6][h] * ys) + sh.frame[9][h] * zs;
					float ry = (sh.frame[4][h] * xs + sh.frame[7][h] * ys) + sh.frame[10][h] * zs;
					float rz = (sh.frame[5][h] * xs + sh.frame[8][h] * ys) + sh.frame[11][h] * zs;
					if (MODE == AO_RANDOM) {
						normalize3(rx, ry, rz);
						if (along_normal) {
							// the un-normalised shading normal itself (:263), kept in the hit record
							const float4 q1 = ((const float4 *) A.hits)[2 * ((size_t) A.tile_base[tile] + h) + 1];
							rx = q1.x; ry = q1.y; rz = q1.z;
						}
					}
					const float ox = sh.frame[0][h], oy = sh.frame[1][h], oz = sh.frame[2][h];
					// (only the lanes with a ray are here: the ballot is over the packet's live lanes)
					tame = MODE == AO_UNIFORM && wave_ballot(!ray_is_tame(ox, oy, oz, rx, ry, rz, OCRT_COLD_F32(P.origin_limit))) == 0ull;
					if (tame) {
						ray.ox = ox; ray.oy = oy; ray.oz = oz;
						ray.dx = rx; ray.dy = ry; ray.dz = rz;
						ray.ix = short_reciprocal(rx); ray.iy = short_reciprocal(ry); ray.iz = short_reciprocal(rz);
					} else {  // (rare: a zero or tiny direction component, a NaN from a zero-length normal)
						ray.ox = ox; ray.oy = oy; ray.oz = oz;
						ray.dx = rx; ray.dy = ry; ray.dz = rz;
						ray.ix = 1.0f / rx; ray.iy = 1.0f / ry; ray.iz = 1.0f / rz;
					}
#ifdef OCRT_DEBUG_KNOBS
					regular = P.scene_regular && P.ao_regular && ray_is_regular(ray);
#endif
				};

#ifdef OCRT_DEBUG_KNOBS
				// Every lane walks on its own; idle lanes are refilled from the job's rays while
				// next < total.
				auto walk_individually = [&]() {
					for (;;) {
						const bool idle_lane = pending.first == NONE && !(i < count);
						const unsigned long long walking = wave_ballot(can_walk(pending, i, count));
						const uint32_t n_leaves = (uint32_t) __popcll(wave_ballot(pending.first != NONE));
						const unsigned long long idle_mask = wave_ballot(idle_lane);
						const uint32_t idle = (uint32_t) __popcll(idle_mask);
						if (next < total && idle >= P.refill_min) {
							const uint32_t item = next + rank_in(idle_mask);
							if (idle_lane && item < total) {
								setup_ray(item, false, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
								i = 0u;
							}
							next += idle;
							continue;
						}
						if (n_leaves != 0u && (n_leaves >= P.leaf_min || walking == 0ull)) {
							if (pending.first != NONE) {
								const TriResult tr = tri_test<false>(scene.tris, pending.first, ray);
								pending.first = pending.second;
								pending.second = NONE;
								if (tr.accepted) {
									atomicAdd(&sh.occluded[h], 1u);
									i = count;  // any-hit: the reference walks on but only uses the boolean (:251)
									pending.first = NONE;
								}
							}
							continue;
						}
						if (walking == 0ull)
							break;
						advance_walkers(scene, ray, regular, P.ao_max_distance, P.ao_below, count, i, pending);
						// a second node straight away while few leaves are pending: halves the scheduling overhead
						if ((uint32_t) __popcll(wave_ballot(pending.first != NONE)) < P.leaf_min)
							advance_walkers(scene, ray, regular, P.ao_max_distance, P.ao_below, count, i, pending);
					}
				};
#endif

#ifdef OCRT_DEBUG_KNOBS
				if (!SHARED)
					walk_individually();
#endif
				if (SHARED) {
					// shared walks (see walk_collect) of 64 consecutive rays of the job at a time; a lane
					// leaves at its first accepted triangle
					const bool scene_fast = OCRT_COLD_U32(P.fast_walk) && OCRT_COLD_U32(P.ao_regular) && OCRT_COLD_F32(P.walk_scale) > 0.0f;
#ifdef OCRT_STAMPS
					uint32_t job_exact = 0u;
#endif
					do {  // (once per piece: fixed shares are one piece, the cursor hands out the others)
					OCRT_STAMP_ADD(5, (total + 63u) / 64u);
					for (uint32_t base = 0u; base < total; base += 64u) {
						const uint32_t lane = fresh_lane();
						bool alive = base + lane < total;
						// a full tile's packet is one table direction: the entry comes by a scalar load
						const bool whole = hit_count == 64u;
						float4 shared_dir = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
						if (whole && MODE == AO_UNIFORM)
							shared_dir = OCRT_COLD_PTR(const float4 *, ao_table)[dir0 + (base >> 6)];
						// the interval of the walk array this packet has to walk (entry_kernel): the one of its table direction,
						// or the tile's.  Two words, made scalar by hand like every load through a re-read pointer.
						uint32_t entry_begin, entry_end;
						{
							const uint32_t stride = OCRT_COLD_U32(P.entry_stride);  // (1: this frame keeps the tiles' own intervals only)
							const uint32_t which = (whole && MODE == AO_UNIFORM && stride > 1u) ? 1u + dir0 + (base >> 6) : 0u;
							const uint2 range = OCRT_COLD_PTR(const uint2 *, tile_entry)[(size_t) tile * stride + which];
							entry_begin = (uint32_t) __builtin_amdgcn_readfirstlane((int) range.x);
							entry_end = (uint32_t) __builtin_amdgcn_readfirstlane((int) range.y);
						}
						if (alive)
							setup_ray(base + lane, whole, shared_dir);
						const bool exact = !scene_fast || (!tame && wave_ballot(alive && !ray_is_selectable(ray, OCRT_COLD_F32(P.origin_limit))) != 0ull);
#ifdef OCRT_STAMPS
						job_exact += exact ? 1u : 0u;
#endif
						if (exact)
							shared_walk_any_hit<true>(OCRT_COLD_PTR(const float4 *, nodes_ptr), walk_ptr, tris_ptr, count, ray, sh.frame, h,
							                          OCRT_COLD_F32(P.ao_max_distance), A.P.ao_below, 0.0f, alive, false, &sh.occluded[h], sh.batch,
							                          A.P.batch_below, walk_prof);
						else
							shared_walk_any_hit<false, PREFETCH>(nullptr, walk_ptr, tris_ptr, count, ray, sh.frame, h,
							                           0.0f, A.P.ao_below, OCRT_COLD_F32(P.walk_scale), alive, tame, &sh.occluded[h], sh.batch,
							                           A.P.batch_below, walk_prof, entry_begin, entry_end, OCRT_COLD_U32(P.walk_ce_bytes));
					}
					take_from_cursor();  // (fixed shares: the cursor holds nothing)
					} while (total != 0u);
#ifdef OCRT_STAMPS
					if (fresh_lane() == 0u && job_exact) {
						atomicAdd(&A.counters->stamp[63], (unsigned long long) job_exact);  // packets that took the exact form
						atomicAdd(&A.counters->stamp[64], (__builtin_amdgcn_s_memrealtime() - t_frames));  // ... and the time of the jobs holding them
					}
#endif
				}
				wave_lds_sync();
				OCRT_STAMP(t_walked);
				OCRT_STAMP_ADD(2, t_walked - t_frames);
				OCRT_STAMP_ADD(4, 1);

				// ---- this job's share of the occlusion counts ----
				{
					const uint32_t lane = fresh_lane();
					if (lane < hit_count) {
						const uint32_t occluded = sh.occluded[lane];
						if (occluded)
							atomicAdd(&OCRT_COLD_PTR(uint32_t *, occluded_of)[(size_t) OCRT_COLD_PTR(const uint32_t *, tile_base)[tile] + lane], occluded);
					}
				}
				wave_lds_sync();
				OCRT_STAMP(t_flushed);
				OCRT_STAMP_ADD(3, t_flushed - t_walked);
#ifdef OCRT_STAMPS
				if (fresh_lane() == 0u) {  // jobs by duration: bucket k holds those of 2^k .. 2^(k+1) microseconds
					const unsigned long long us = (t_flushed - t_job) / 100ull;
					const int bucket = us == 0ull ? 0 : 63 - __builtin_clzll(us);
					atomicAdd(&A.counters->stamp[49 + (bucket > 15 ? 15 : bucket)], 1ull);
				}
#endif
			}
		}
	}
	if (wave == 0u && fresh_lane() == 0u)  // (nothing kept across the pass: one clock read and one atomic per workgroup)
		atomicMax(&OCRT_COLD_PTR(FrameCounters *, counters)->tick_ao_end, (unsigned long long) __builtin_amdgcn_s_memrealtime());
#ifdef OCRT_TAIL  // minimal: nothing is kept across the pass, one load and two atomics when the wave ends
	if (fresh_lane() == 0u) {
		const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
		const unsigned long long origin = __hip_atomic_load(&A.counters->tick_ao_begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const unsigned long long end_bucket = (t_end - origin) / 5000ull;  // 0.05 ms
		atomicAdd(&A.counters->stamp[10 + (end_bucket > 31 ? 31 : end_bucket)], 1ull);
		atomicMax(&A.counters->stamp[8], t_end);
	}
#endif
#ifdef OCRT_STAMPS
	if (fresh_lane() == 0u) {
		const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
		for (int k = 0; k < 6; ++k)
			atomicAdd(&A.counters->stamp[k], stamp_acc[k]);           // claim, frames, walks, flush (10 ns ticks); jobs, packets
		atomicAdd(&A.counters->stamp[6], t_end - t_begin);            // sum of wave lifetimes
		atomicMin(&A.counters->stamp[7], t_begin);                     // first start
		atomicMax(&A.counters->stamp[8], t_end);                       // last end
		if (stamp_acc[4])
			atomicAdd(&A.counters->stamp[9], 1ull);                    // waves that got any work
		for (int k = 0; k < 7; ++k)
			atomicAdd(&A.counters->stamp[42 + k], walk_prof_store[k]);  // time in the node loop, in batches; loop entries, batches, leaf stops
		// when this wave ended, counted from the first wave's start (settled long before any wave ends), 0.1 ms buckets:
		// how the occupancy decays towards the end of the launch
		(void) t_last_claim;
		const unsigned long long first = __hip_atomic_load(&A.counters->stamp[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const unsigned long long end_bucket = (t_end - (first < t_begin ? first : t_begin)) / 10000ull;
		atomicAdd(&A.counters->stamp[10 + (end_bucket > 31 ? 31 : end_bucket)], 1ull);
	}
#endif
}

template <int MODE, bool SHARED, bool PREFETCH = false>
__global__ __launch_bounds__(64 * AO_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void ao_kernel(FrameArgs A) {
	__shared__ TileShared shared_tiles[AO_WAVES];
	__shared__ unsigned int wg_claim[8];
	ao_pass<MODE, SHARED, PREFETCH, false>(A, shared_tiles, wg_claim, nullptr);
}

}  // namespace ocrt
