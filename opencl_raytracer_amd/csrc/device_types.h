// device_types.h -- records shared by the host-side scene packer and the HIP
// kernels.  Plain structs, no HIP headers, so host-only code can include it.
#pragma once
#include <cstdint>

namespace ocrt {

// One BVH node, 32 bytes = two float4 loads.  Merges the reference's `nodes`
// (subtree size) and `aabbs` (min,max) arrays (reference
// src/intersect_kernel.cl:187-192) and adds the leaf's triangle index, which the
// reference recovers with a running counter.
struct NodeRec {
	float lo[3];
	uint32_t skip;  // subtree size in nodes; 1 = leaf
	float hi[3];
	uint32_t leaf;  // leaf index (= triangle index in leaf order), 0xFFFFFFFF for inner nodes
};
static_assert(sizeof(NodeRec) == 32, "NodeRec is two float4");

// Walk node, 32 bytes: the node's box ENLARGED by a safety margin, laid out as
// (lo, hi) pairs per axis so that the three fused multiply-adds of the
// conservative slab test are packed instructions.  A superset box can only add
// node visits, never remove one, and every triangle is still gated by its exact
// box (kept in TriRec) before its test, so results do not change
// (SURVEY.md 8a-0.7); kernels.hip states the error bound behind the margin.
//
// The walk array is stored "top first": entries [0, top_count) are the nodes of
// the first levels of the tree in pre-order (they are copied into LDS by every
// workgroup), followed by the body of every subtree that was cut off, each body
// contiguous and in pre-order.  `link`'s top two bits give the kind:
//   WALK_INNER   span = number of array entries of this node's (truncated) subtree
//   WALK_LEAF    span = 1, payload = leaf index (= triangle index in leaf order)
//   WALK_PORTAL  a cut: payload = index of the first body entry, span = body length
struct WalkNodeRec {
	float lox, hix, loy, hiy;
	float loz, hiz;
	uint32_t span;
	uint32_t link;
};
static_assert(sizeof(WalkNodeRec) == 32, "WalkNodeRec is two float4");
constexpr uint32_t WALK_INNER = 0u, WALK_LEAF = 1u, WALK_PORTAL = 2u;
constexpr uint32_t WALK_KIND_SHIFT = 30u;
constexpr uint32_t WALK_PAYLOAD_MASK = (1u << WALK_KIND_SHIFT) - 1u;
constexpr uint32_t WALK_TOP_CAPACITY = 288u;  // entries of the top of the tree kept in LDS (9 KB)

// Per-triangle record, 96 bytes = six float4: the invariants of the reference's
// plane/parametric test (reference src/intersect_kernel.cl:67-90), precomputed
// on the host with the SAME float operations the kernel would execute (so the
// bits are identical), followed by the leaf's exact box.
struct TriRec {
	float ta[3];
	float u[3];   // tb - ta
	float v[3];   // tc - ta
	float n[3];   // cross(u, v)
	float uu, uv, vv, D;  // dot(u,u), dot(u,v), dot(v,v), uv*uv - uu*vv
	float lo[3], pad0;    // exact leaf box (reference aabbs[2i], aabbs[2i+1])
	float hi[3], pad1;
};
static_assert(sizeof(TriRec) == 96, "TriRec is six float4");

// The three vertex normals of a leaf's triangle (replaces the faces[] ->
// normals[] double indirection of reference src/intersect_kernel.cl:118-127).
struct ShadeRec {
	float n0[4], n1[4], n2[4];
};
static_assert(sizeof(ShadeRec) == 48, "ShadeRec is three float4");

constexpr uint32_t XCD_GROUPS = 8;

enum AoMode : int32_t { AO_NONE = 0, AO_UNIFORM = 1, AO_RANDOM = 2 };

// Image-band ownership for multi-GPU runs: the image is cut into bands of
// `band_tile_rows` tile rows; band b belongs to rank b % nranks.
struct Partition {
	uint32_t rank;
	uint32_t nranks;
	uint32_t band_tile_rows;
};

// Launch-constant parameters (the reference bakes these into the kernel as -D
// macros, reference src/opencl_host.cc:42-53).
struct KernelParams {
	uint32_t width;        // supersampled width  (WIDTH)
	uint32_t height;       // supersampled height (HEIGHT)
	float a;               // FOCAL_LENGTH * max(WIDTH, HEIGHT)
	float half_w;          // WIDTH  / (2.0f * a)
	float half_h;          // HEIGHT / (2.0f * a)
	uint32_t node_count;   // nodes[0]
	uint32_t tri_count;    // triangles = leaves
	int32_t shading;       // SHADING_ENABLE
	int32_t ao_mode;       // AoMode
	float ao_max_distance; // AO_MAX_DISTANCE
	uint32_t ao_dirs;      // UNIFORM: rays per hit sub-pixel; RANDOM: AO_NUM_SAMPLES
	int32_t variant;       // debug: 0 = default, 2 = never use the walk nodes (exact boxes only)
	int32_t scene_regular; // every box finite, |coord| <= 1e37, lo <= hi, children inside parents
	int32_t walk_ok;       // WalkNodeRec array usable (scene_regular and coordinates small enough)
	float origin_limit;    // rays whose |origin| exceeds this use the exact boxes (margin was sized for it)
	uint32_t top_count;    // walk-array entries [0, top_count) are the top of the tree (see WalkNodeRec)
	uint32_t top_lds;      // how many of them the AO pass keeps in LDS (debug knob OCRT_TOP_LDS, default all)
	uint32_t group_offset[XCD_GROUPS + 1];  // hit-list segment of group g = [group_offset[g], group_offset[g+1])
	uint32_t dirs_per_batch;   // an AO batch = up to 64 hits x this many directions
	uint32_t batches_per_hits; // ceil(ao_dirs / dirs_per_batch)
	int32_t ao_regular;    // AO_MAX_DISTANCE > 0 (needed by the folded form of the slab test)
	float primary_below;   // largest float below the primary rays' max_distance (100000.0f)
	float ao_below;        // largest float below AO_MAX_DISTANCE
	uint32_t tiles_x;      // tiles per image row
	uint32_t local_tile_rows;  // tile rows this rank owns
	Partition part;
};

constexpr uint32_t TILE_W = 8;
constexpr uint32_t TILE_H = 8;
// Hit record handed from the primary pass to the ambient-occlusion pass, 32 bytes:
// (origin.xyz, head-light value), (normal.xyz, image index as bits).
struct HitRec {
	float ox, oy, oz, value;
	float nx, ny, nz;
	uint32_t pixel;
};
static_assert(sizeof(HitRec) == 32, "HitRec is two float4");

// Device-side counters of one frame (zeroed before the primary pass).
// The image's strips are dealt to 8 groups (one per XCD, see kernels.hip); each
// group has its own segment of the hit list and its own AO batch queue, so that
// an XCD's L2 keeps seeing the same part of the scene in both passes.
struct FrameCounters {
	uint32_t hit_count[XCD_GROUPS];   // records in each group's hit-list segment
	uint32_t queue_head[XCD_GROUPS];  // next unclaimed AO batch of each group
	uint32_t primary_hits;            // all hit sub-pixels
	uint32_t pad;
	unsigned long long occluded;      // occluded AO rays
};

// Ray statistics of the last frame (summed on the host from per-tile counters).
struct RenderStats {
	unsigned long long primary_rays;
	unsigned long long primary_hits;
	unsigned long long ao_rays;
	unsigned long long ao_occluded;
};

}  // namespace ocrt
