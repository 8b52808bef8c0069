// scene_pack.h -- turns the five reference-format scene arrays into the device
// records of device_types.h and derives the launch constants.  Host only.
#pragma once
#include <cstdint>
#include <limits>
#include <memory>
#include <vector>

#include "device_types.h"
#include "ray_tracer.h"
#include "vec3.h"

namespace ocrt {

// What the fast form of the shared walk reads: the same tree as PackedScene::nodes with every box pushed outward by
// the margin derived at padded_bound(), the subtree size as a BYTE offset and two END records behind the last node
// (kernels.hip, walk_collect).  The outward margin makes the walk's 6-FMA box test conservative, so a ray can only
// visit MORE boxes than the reference's own test would let it; which triangles count is decided by the exact test on
// the leaf's own box (TriRec::lo / hi).  `nodes` is empty when the scene does not qualify (irregular or non-nested
// boxes, extent beyond 1e6): the kernels then only use the exact form.
struct WalkArray {
	std::vector<NodeRec> nodes;
	float origin_limit = 0.0f;  // rays whose origin exceeds this magnitude on some axis take the exact form
	float prune_margin = std::numeric_limits<float>::infinity();  // KernelParams::prune_margin (make_walk_array; +inf: the closest-hit walk must not prune)
	uint32_t primary_bytes = 0;   // bytes of the plane-form records (the primary rays' tree; two END records follow, then the other copy at ce_offset)
	uint32_t unpruned_bytes = 0;  // ... and the part at the head of the primary rays' records inside which no limit is lowered (the root and the faces without a bound)
	float ao_scale = 0.0f;      // walk_scale_for(ao_max_distance) the margins were sized for (0: none)
	uint32_t ce_offset = 0;     // byte offset of the same records in centre / half-extent form (0: none; ao_scale > 0 only)
};
struct PackedScene {
	std::vector<NodeRec> nodes;
	std::vector<TriRec> tris;
	std::vector<ShadeRec> shade;
	// The flags describe `nodes` as the kernels will walk it -- the uploaded array, or (rebuilt = true) the
	// cheaper tree over the same leaves that pack_scene put in its place and re-validated.
	bool regular = false;  // all boxes finite, |coord| <= 1e37, lo <= hi (see kernels.hip slab_hit_regular)
	bool binary_tree = false;  // sibling subtrees tile their parent's index range (the UPLOADED array must be a full
	                           // binary tree, what the reference's triangle counter assumes; a rebuilt one may have
	                           // inner nodes with more children: the skip list and the shared walk do not care)
	bool nested = false;       // ... and every node's box contains its children's boxes (true for any tree built by
	                           // uniting child boxes; arbitrary uploaded arrays need not be)
	bool rebuilt = false;      // `nodes` is the rebuilt tree, not the uploaded one
	// Optional: the walk array made ahead of the upload (prepare_walk_array) for this AO_MAX_DISTANCE -- CPU work that a
	// caller can do before it has a device; DeviceRenderer::upload makes its own when this one is absent or was made for
	// another distance.
	std::shared_ptr<const WalkArray> walk;
	float walk_max_distance = -1.0f;
	bool walk_for_a_stream = false;  // (what `walk` was made for)
};

// Validates the arrays against each other (every index and skip count is
// range-checked, so the kernels can never read out of bounds) and packs them.
// Throws std::invalid_argument with a description on malformed input.
PackedScene pack_scene(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes,
                       const std::vector<Vec3f> &aabbs, const std::vector<Vec3f> &vertices,
                       const std::vector<Vec3f> &vnormals);

// `ao_max_distance`: the kernel's AO_MAX_DISTANCE (bounds how far from a box an ambient-occlusion ray that hits it
// can start; <= 0 or not finite: no ambient occlusion, or no usable bound).
// `for_a_stream`: also what only pays over many frames of the scene -- the primary rays' copy re-ordered and its boxes grown
// so that the closest-hit walk may prune (2-3 ms of CPU for the bunny, ~0.02 ms per frame): a one-shot host (the
// reference's use) gets the records in the builder's order and prune_margin = +inf.
WalkArray make_walk_array(const PackedScene &scene, float ao_max_distance, bool for_a_stream = true);
// make_walk_array into scene.walk (see PackedScene::walk).
void prepare_walk_array(PackedScene &scene, float ao_max_distance, bool for_a_stream = true);
// The margin itself: the padded value of a box's lower (upper = false) or upper bound `b` for ray origins of
// magnitude up to `origin_bound` on that axis; always < b resp. > b.
// `scaled_reach`: the max_distance (x 1.001) of the rays that use the SCALED node test on this array, 0 if none do.
float padded_bound(float b, float origin_bound, bool upper, float scaled_reach = 0.0f);
// One axis of a padded box as centre and half-extent for the select-free form of the scaled node test (kernels.hip,
// OCRT_TEST_CE_SCALED): the half-extent also covers the rounding of t at the centre (scene_pack.cc, ce_record).
void padded_centre_extent(float padded_lo, float padded_hi, float origin_bound, float *centre, float *half_extent);
// The factor the scaled node test multiplies the reciprocal directions with: the largest float r with
// r * max_distance * (1 + 2^-23) <= 1, or 0 where the scaled form must not be used (max_distance not a positive
// number within 2^-20 .. 2^20).
float walk_scale_for(float max_distance);
// ... and whether origins up to `origin_limit` keep o * (2^100 * scale) and o * (1e30 * scale) finite (make_walk_array
// drops the scaled form otherwise).
bool walk_scale_usable(float scale, float origin_limit);

// The value a float option has once it went through the reference's -D string:
// printed with 6 significant digits ("-DNAME=0.2f", reference
// include/compiler_options.h:13-19) and re-parsed as a float literal.
float kernel_float(float v);

// Direction table of the UNIFORM hemisphere, one (xs, ys, zs, 0) per ray in
// casting order (reference src/intersect_kernel.cl:219-246, which recomputes
// these pixel-independent values for every pixel).
std::vector<float> uniform_ao_table(unsigned int rings, int alpha_min, int alpha_max);

// Largest band height (in tile rows) granularity such that a band holds whole
// supersample blocks: lcm(TILE_H, n) / TILE_H.
uint32_t band_tile_rows_for(unsigned int grid);

// Number of tile rows rank `rank` owns for an image of `total_height` rows.
uint32_t local_tile_rows_for(uint32_t total_height, const Partition &part);

// What the launch constants need to know of an uploaded scene (the arrays themselves stay on the device).
struct SceneFacts {
	bool regular = false, nested = false, binary_tree = false;  // PackedScene's flags
	bool has_walk = false;                                       // the padded walk array exists
	float origin_limit = 0.0f, ao_scale = 0.0f;                  // WalkArray's
	float prune_margin = 0.0f;                                   // WalkArray's
	uint32_t unpruned_bytes = 0, primary_bytes = 0, ce_offset = 0;
};
SceneFacts scene_facts(const PackedScene &scene, const WalkArray &walk);
KernelParams make_kernel_params(const RayTracer &rt, uint32_t node_count, uint32_t tri_count, uint32_t ao_dirs,
                                const Partition &part, const SceneFacts *facts);

}  // namespace ocrt
