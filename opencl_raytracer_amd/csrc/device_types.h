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
// The copy of the node array that the fast form of the shared walk reads uses the same record with `skip` as a
// byte offset and boxes pushed outward (scene_pack.cc, pad_walk_boxes); its end marker's leaf field:
constexpr uint32_t WALK_END = 0xFFFFFFFEu;

// What a leaf's test needs, 96 bytes = six float4.  First the leaf's OWN box as uploaded, the reference's gate of
// the triangle test (src/intersect_kernel.cl:189,195): the walk itself only tests padded boxes.  Then the
// per-triangle invariants of the reference's plane/parametric test (src/intersect_kernel.cl:67-90), precomputed on
// the host with the SAME float operations the kernel would execute, so the bits are identical.
struct TriRec {
	float lo[3], pad0;  // the leaf's box
	float hi[3], inv_d;  // ... and RN(1 / D), or NaN where that is no use: tri_predicate.h
	float ta[3];
	float u[3];   // tb - ta
	float v[3];   // tc - ta
	float n[3];   // cross(u, v)
	float uu, uv, vv, D;  // dot(u,u), dot(u,v), dot(v,v), uv*uv - uu*vv
};
static_assert(sizeof(TriRec) == 96, "TriRec is six float4");

// The three vertex normals of a leaf's triangle (replaces the faces[] ->
// normals[] double indirection of reference src/intersect_kernel.cl:118-127).
struct ShadeRec {
	float n0[4], n1[4], n2[4];
};
static_assert(sizeof(ShadeRec) == 48, "ShadeRec is three float4");

// The image's strips are dealt to 8 groups (one per XCD, see kernels.hip); each
// group orders and queues its own tiles, so that an XCD's L2 keeps seeing the
// same part of the scene in both passes.
constexpr uint32_t XCD_GROUPS = 8;

// Hit record handed from the primary pass to the ambient-occlusion pass, 32 bytes:
// (hit point, head-light value), (smooth normal, image index as bits).
struct HitRec {
	float ox, oy, oz, value;
	float nx, ny, nz;
	uint32_t pixel;
};
static_assert(sizeof(HitRec) == 32, "HitRec is two float4");

// Device-side counters of one frame.  Nothing here needs clearing between frames by a kernel of its own: the queues'
// heads and the running sums are put back by the first workgroup of the primary pass, before any of the later kernels
// that add to them can run.  (The buffer is zeroed once, when it is allocated.)
// Each group's queue sits in its own 128-byte line: the heads are hammered with returning atomics by every workgroup
// of the AO pass.  What a queue holds is the same in every frame of an uploaded scene (camera, scene and options are
// fixed, so the tiles that are hit and what their any-hit rays cost are too): the ordered tile lists and the three
// constants below are made ONCE per upload, on the host, from the pass that sizes the hit list (DeviceRenderer::
// orderTiles) -- until round 4 the last workgroup of every frame's primary pass sorted its group's tiles again.
struct alignas(128) GroupQueue {
	uint32_t head;        // next unclaimed (tile, table direction) unit of the group's ordered tile list (per frame)
	uint32_t work_tiles;  // non-empty tiles of the group (per upload)
	uint32_t cost_sum;    // sum of their AO cost classes (per upload)
	uint32_t hits;        // hit sub-pixels in the group's tiles (per upload)
	// the fused frame kernel's primary work of the group (kernels/frame.hip.h): 2 x 2 tile blocks, in the order of
	// FrameArgs::primary_order
	uint32_t primary_blocks;  // how many (per upload)
	// The first `split_units` units of the list -- its heaviest tiles, by measured cost -- are claimed HALF A TILE at a time
	// from a cursor of their own (kernels/ao.hip.h; per upload; 0: none).  `head` starts at split_units.
	uint32_t split_units;
	uint32_t pad0[26];
	// ... the claim cursors of the beginning of a frame in a line of their own (`head` is hammered at its end): the fused
	// kernel's block cursor, the cursor of the split tiles
	uint32_t primary_head;
	uint32_t split_head;
	uint32_t pad1[30];
};
static_assert(sizeof(GroupQueue) == 256, "two lines per queue");
struct FrameCounters {
	GroupQueue queue[XCD_GROUPS];
	unsigned long long occluded;  // occluded AO rays: summed from the hit list's counts when the statistic is asked for (occluded_sum_kernel)
	// The device's own 100 MHz clock (s_memrealtime) read by the kernels: when the primary pass began (its first
	// workgroup), when the first workgroup of the ambient-occlusion pass began and when its last one ended.  A frame
	// replayed from a captured hipGraph has no HIP events inside it that could be timed (hipEventElapsedTime refuses
	// event-record nodes); these say when its passes ran.
	unsigned long long tick_begin, tick_ao_begin, tick_ao_end;
	uint32_t frame_seq;  // frames finished on these counters (the finishing kernel counts): what the fused frame kernel's flags are compared with
	uint32_t stalled;    // fused frame kernel: waves that gave up waiting for a tile's hit records (0 in every healthy frame; DeviceRenderer reports it)
#if defined(OCRT_STAMPS) || defined(OCRT_TAIL)
	unsigned long long stamp[10 + 32 + 7 + 16];  // debug build: wave-time (10 ns ticks) per phase of the AO pass, jobs, packets
#endif
};

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
	uint32_t ao_dirs;      // AO rays cast per hit sub-pixel (UNIFORM: table size; RANDOM: AO_NUM_SAMPLES + 2)
	uint32_t ao_divisor;   // n of `1 - hits / n` (UNIFORM: ao_dirs; RANDOM: AO_NUM_SAMPLES + 1, reference :260-275)
	int32_t scene_regular; // every box finite, |coord| <= 1e37 and lo <= hi: min/max slab form allowed
	int32_t ao_regular;    // AO_MAX_DISTANCE > 0 (needed by the folded form of the slab test)
	int32_t scene_nested;  // every child box lies inside its parent's box: the shared walk's fast form is allowed
	int32_t shared_walk;   // sibling subtrees tile their parent's index range (any arity): one shared node index is safe
	int32_t fast_walk;     // the padded walk array exists (regular, nested scene): the 6-FMA box test may be used
	float origin_limit;    // ... for rays whose origin coordinates do not exceed this magnitude
	int32_t shared_device;  // other hosts' frames run beside this one's (DeviceRenderer::setDeviceShare)
	float walk_scale;      // ~ 1 / ao_max_distance: the ambient-occlusion rays' node test measures t in these units
	                       // (kernels.hip, OCRT_TEST_COHERENT_SCALED); 0 = unusable, those rays take the exact form
	float primary_below;   // largest float below the primary rays' max_distance (100000.0f)
	float ao_below;        // largest float below AO_MAX_DISTANCE
	int32_t debug_no_sort; // debug knob OCRT_NO_SORT: claim tiles in arbitrary order instead of heaviest first
	uint32_t refill_min;    // wave scheduler: refill once this many lanes are idle (debug knob OCRT_REFILL_MIN)
	uint32_t leaf_min;      // ... test triangles once this many leaves are pending (OCRT_LEAF_MIN)
	uint32_t batch_below;   // AO: a leaf hit by fewer lanes than this has its triangle tests deferred and batched
	uint32_t cost_shift;    // ordering key of a block of 64 tiles = 1 + (sum of its tiles' cost classes >> cost_shift)
	uint32_t ao_claim_max;  // most (tile, direction) units one wave's share of a claim holds; 0 = ao_kernel's rule (a quarter of a tile, or a whole tile)
	uint32_t ao_claim_div;  // (set by launch_ao: the waves per XCD group)
	uint32_t ao_guide;      // 0 (default): AO claims never shrink; n > 0 (debug knob OCRT_AO_GUIDE): a claim takes
	                        // 1/ao_guide of the (tile, direction) units left in its queue, launch_ao multiplies n by the waves per XCD group
	uint32_t strip_tiles;  // width, in tiles, of the vertical strips the image is dealt to the XCD groups in: a power of two,
	                       // 2 by default (the finest deal: best balance, and a cache-resident scene does not care), wider for
	                       // scenes far beyond the L2s, whose XCDs should not all fetch the same geometry (device_renderer.cc)
	uint32_t primary_ahead; // fused frame kernel: how many 2 x 2 blocks beyond what a claim needs the group's primary work is taken (kernels/primary.hip.h, primary_top_up)
	uint32_t primary_list_stride;  // primary_kernel with a list of its groups' blocks (DeviceRenderer::orderPrimaryBlocks): entries per group -- the grid is 8 x this; 0: no list
	float prune_margin;  // closest-hit walk: a lane with a hit at distance d does not enter boxes whose near distance exceeds d (1 + 1e-5) + this (scene_pack.cc, make_walk_array: the rounding of the distances; +inf: no pruning in this scene)
	uint32_t unpruned_bytes;  // ... and while the walk is below this byte offset of the plane-form records no lane's limit is lowered (the faces no box can promise anything about: make_walk_array)
	uint32_t primary_walk_bytes;  // bytes of the plane-form walk records (the primary rays' tree; END records behind them)
	uint32_t walk_ce_bytes;       // byte offset of the centre / half-extent records (the any-hit rays' tree) in the walk array; 0: none
	uint32_t entry_stride; // walk intervals per tile (kernels.hip, entry_kernel): 1 + ao_dirs, or 1 where that table would be too large
	uint32_t tiles_x;      // tiles per image row
	uint32_t local_tile_rows;  // tile rows this rank owns
	Partition part;
};

// Device pointers of one uploaded scene, as the launchers of kernels.hip take them.
struct SceneBuffers {  // device pointers of one uploaded scene
	const void *nodes;       // NodeRec[node_count + 1]: exact boxes (exact form of the walk, first-generation kernels)
	const void *walk;        // NodeRec[node_count + 2]: padded boxes, byte skips, END records (fast form); may be null
	const void *tris;        // TriRec[tri_count]: leaf box + triangle invariants by leaf index
	const void *shade;       // ShadeRec[tri_count]
	const void *ao_table;    // float4[ao_dirs] (UNIFORM)
};

// Waves per workgroup of the ambient-occlusion pass: they take consecutive parts of a claim (kernels.hip), which is
// why the host deals the UNIFORM direction table to that many groups (device_renderer.cc).
constexpr uint32_t AO_WORKGROUP_WAVES = 4;

constexpr uint32_t TILE_W = 8;
constexpr uint32_t TILE_H = 8;

// Ray statistics of the last frame (summed on the host from per-tile counters).
struct RenderStats {
	unsigned long long primary_rays;
	unsigned long long primary_hits;
	unsigned long long ao_rays;
	unsigned long long ao_occluded;
};

}  // namespace ocrt
