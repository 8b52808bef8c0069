// kernels/entry.hip.h -- once per upload: the walk intervals of the tiles' any-hit packets
// (part of the one translation unit kernels.hip; see its head for the passes and the arithmetic contract)
#pragma once
#include "common.hip.h"

namespace ocrt {

// ---------------------------------------------------------------------------
// Once per upload (camera, scene and options are fixed, so a tile's hit points are the same in every frame): for every
// tile, WHERE in the walk array its ambient-occlusion packets have to walk.  An any-hit ray starts at one of the tile's hit
// points (+ normal * 1e-5) and is at most AO_MAX_DISTANCE long: the reference's slab test (src/intersect_kernel.cl:21-61:
// t_near < max_distance, t_far > 0, t_near <= t_far) only passes for a box that holds a point o + t d with 0 <= t <=
// max_distance, a point of the SEGMENT the ray covers.  So a leaf whose box stays clear of the box around the segments of
// a set of rays (grown by a margin for the roundings: 1 % of the distance + 2^-20 of the coordinates) is tested by none of
// them, in any tree.  The walk array is the tree in pre-order with skip offsets, so a walk can start at ANY record and
// stop at any other: it visits what lies between in the usual way.  An interval [begin, end) for a region: from the root
// down, `begin` moves to the first child that meets the region whenever that child follows clear ones or is the only one
// that meets it (its parent's test and the clear subtrees are skipped), `end` moves to the end of the last child that
// meets it.  Where exactly one child meets the region at every level this is the deepest subtree that holds everything
// reachable; below that it trims both flanks.
// Per tile, 1 + ao_dirs intervals: [0] for the tile's hit points grown by the distance on every side -- any ray from the
// tile: packets of tiles that are not full (several table directions in one packet) and the RANDOM mode --, [1 + k] for
// the 64 rays of table direction k, a full tile's packet: the box around 64 segments, a third of the other's volume or
// less.  One wave per tile: lanes = hits while the origins and tangent frames (reference :225-236) go to LDS, then lanes
// = intervals, each going down the tree on its own; speed is nobody's concern here.
// Node tests per packet against walking the whole array (tools/analysis/ao_packets.cc, rows TRIM and PDIR): the bunny's
// plane 7.4 -> 5.8, its model tiles 104.8 -> 91.2 (AO_MAX_DISTANCE is a fifth of the model), the interior scene 38.2 -> 21.2.
// ---------------------------------------------------------------------------
constexpr uint32_t ENTRY_WAVES = 4;
__global__ __launch_bounds__(64 * ENTRY_WAVES) void entry_kernel(const NodeRec *__restrict__ walk, const HitRec *__restrict__ hits,
                                                                 const uint32_t *__restrict__ tile_hits, const uint32_t *__restrict__ tile_base,
                                                                 const float4 *__restrict__ ao_table, uint2 *__restrict__ tile_entry,
                                                                 uint32_t tiles, uint32_t stride, int32_t uniform_table,
                                                                 float max_distance) {
	__shared__ float origin[ENTRY_WAVES][3][64];
	__shared__ float frame[ENTRY_WAVES][9][64];  // basis_x, basis_y (the normal), basis_z
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const uint32_t tile = blockIdx.x * ENTRY_WAVES + wave;
	if (tile >= tiles)
		return;  // (a whole wave: the waves of a workgroup never meet at a barrier)
	const uint32_t hit_count = tile_hits[tile] & 0xFFu;
	uint2 *const out = tile_entry + (size_t) tile * stride;
	const uint32_t whole = walk[0].skip / (uint32_t) sizeof(NodeRec);
	if (lane < hit_count) {
		const HitRec rec = hits[(size_t) tile_base[tile] + lane];
		const float nx = rec.nx, ny = rec.ny, nz = rec.nz;
		const float eps = 1.0f / 100000.0f;
		origin[wave][0][lane] = rec.ox + nx * eps;
		origin[wave][1][lane] = rec.oy + ny * eps;
		origin[wave][2][lane] = rec.oz + nz * eps;
		float hx = nx, hy = ny, hz = nz;
		const float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
		if (ax <= ay && ax <= az)
			hx = 1.0f;
		else if (ay <= ax && ay <= az)
			hy = 1.0f;
		else if (az <= ax && az <= ay)
			hz = 1.0f;
		float bxx, bxy, bxz;
					cross3(hx, hy, hz, nx, ny, nz, bxx, bxy, bxz);
		normalize3(bxx, bxy, bxz);
		float bzx, bzy, bzz;
					cross3(bxx, bxy, bxz, nx, ny, nz, bzx, bzy, bzz);
		normalize3(bzx, bzy, bzz);
		frame[wave][0][lane] = bxx; frame[wave][1][lane] = bxy; frame[wave][2][lane] = bxz;
		frame[wave][3][lane] = nx;  frame[wave][4][lane] = ny;  frame[wave][5][lane] = nz;
		frame[wave][6][lane] = bzx; frame[wave][7][lane] = bzy; frame[wave][8][lane] = bzz;
	}
	wave_lds_sync();
	const float inf = __builtin_inff();
	for (uint32_t j = lane; j < stride; j += 64u) {
		// ---- the region of interval j ----
		float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
		bool odd = !(max_distance > 0.0f) || hit_count == 0u;  // nothing can be said: the whole array it is
		const bool one_direction = j != 0u && uniform_table != 0 && hit_count == 64u;
		float4 table = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (one_direction)
			table = ao_table[j - 1u];
		for (uint32_t h = 0u; h < hit_count; ++h) {
			const float o[3] = { origin[wave][0][h], origin[wave][1][h], origin[wave][2][h] };
			for (int k = 0; k < 3; ++k) {
				float a = o[k], b = o[k];
				float margin = fabsf(o[k]) * 0x1.0p-20f + 1.0e-30f;
				if (one_direction) {
					// ray_dir = basis_x * xs + basis_y * ys + basis_z * zs (reference :246), its length 1 up to roundings
					const float d = (frame[wave][k][h] * table.x + frame[wave][3 + k][h] * table.y) + frame[wave][6 + k][h] * table.z;
					b = o[k] + d * (max_distance * 1.01f);
					margin += max_distance * 0.01f;
				} else {
					a = o[k] - max_distance * 1.01f;
					b = o[k] + max_distance * 1.01f;
				}
				odd = odd || !(fabsf(a) < inf) || !(fabsf(b) < inf);
				lo[k] = fminf(lo[k], fminf(a, b) - margin);
				hi[k] = fmaxf(hi[k], fmaxf(a, b) + margin);
			}
		}
		// ---- its interval (node indices; the walk array keeps byte offsets) ----
		const auto skip_of = [&](uint32_t n) { return walk[n].skip / (uint32_t) sizeof(NodeRec); };
		// (`walk`: the records the any-hit packets walk -- the centre / half-extent copy, whose order of children need not be
		// the plane form's (scene_pack.cc, make_walk_array): lo holds c, hi holds e, and e is padded far beyond the rounding
		// of c -+ e)
		const auto meets = [&](uint32_t n) {
			const NodeRec box = walk[n];
			return !(box.lo[0] - box.hi[0] > hi[0] || box.lo[0] + box.hi[0] < lo[0] || box.lo[1] - box.hi[1] > hi[1] || box.lo[1] + box.hi[1] < lo[1] ||
			         box.lo[2] - box.hi[2] > hi[2] || box.lo[2] + box.hi[2] < lo[2]);
		};
		uint32_t begin = 0u, end = whole ? whole : 1u;
		if (!odd && (j == 0u || one_direction)) {
			uint32_t n = 0u;
			while (skip_of(n) > 1u) {  // left
				uint32_t first = 0u, index = 0u, others = 0u;
				for (uint32_t c = n + 1u; c < n + skip_of(n); c += skip_of(c)) {
					if (first)
						others += meets(c) ? 1u : 0u;
					else {
						++index;
						if (meets(c))
							first = c;
					}
				}
				if (first == 0u) {  // (no child meets it: nothing under this node can be reached)
					n += skip_of(n);
					break;
				}
				if (index == 1u && others != 0u)
					break;
				n = first;
			}
			begin = n;
			n = 0u;
			while (skip_of(n) > 1u) {  // right
				uint32_t last = 0u;
				for (uint32_t c = n + 1u; c < n + skip_of(n); c += skip_of(c))
					if (meets(c))
						last = c;
				if (last == 0u) {
					end = n;
					break;
				}
				end = last + skip_of(last);
				n = last;
			}
			if (begin > end)
				begin = end;
		}
		// (a tile that is not full never looks at its per-direction intervals: they are filled with interval 0's rule
		// all the same -- the whole array here, harmless -- so that every word of the table is defined)
		out[j] = make_uint2(begin * (uint32_t) sizeof(NodeRec), end * (uint32_t) sizeof(NodeRec));
	}
}

}  // namespace ocrt
