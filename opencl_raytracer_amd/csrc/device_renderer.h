// device_renderer.h -- owns the HIP device state of one render host: scene
// buffers, image buffers, stream, timing events.  All failures are reported as
// exceptions (ocrt::DeviceError / std::invalid_argument); the two front ends
// (HipHost with the reference's print-and-exit convention, the C ABI with
// return codes) decide what to do with them.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "device_types.h"
#include "ray_tracer.h"
#include "scene_pack.h"

namespace ocrt {

struct DeviceError : std::runtime_error {
	using std::runtime_error::runtime_error;
};

// Number of visible HIP devices (0 when the runtime reports none or fails).
int visible_device_count();
// Initialises the HIP runtime, the device's context and this library's code object on it (device < 0: $OCRT_DEVICE or
// 0); never fails -- whatever is wrong shows up again, properly reported, when a renderer is created.  These one-off
// costs (100-200 ms) are independent of the scene: a front end runs this on a second thread while it loads the mesh
// and builds the BVH (render.cc).
void warm_up_device(int device);

// One uploaded scene on one GPU: node arrays, leaf records, normals, direction table.  Immutable once made and
// reference-counted, so that every renderer of that GPU that renders the scene -- the hosts of a frame ring -- walks
// the SAME arrays: one upload, one copy in HBM, and the frames that overlap on the device find each other's nodes in
// the XCDs' L2 and the scalar caches instead of competing for them with identical copies (the reference uploads a scene
// once, src/opencl_host.cc:120-136).  Holds what the launch constants need to know of it as well.
class DeviceScene {
	public:
		// Blocking upload; `options`: the direction table (AO samples, angles, method) and the any-hit rays' reach
		// (AO_MAX_DISTANCE, which sizes the walk array's margins) are baked in -- servesOptions() says whether another
		// renderer's options agree.
		// (`for_a_stream`: make_walk_array's -- what only pays over many frames of the scene)
		static std::shared_ptr<const DeviceScene> create(int device, const PackedScene &scene, const RayTracer::Options &options, bool for_a_stream = true);
		// A scene's arrays live in ONE device allocation.  A front end that knows the size of what is coming before it has
		// the arrays (the triangle count of a mesh file's header) can have that allocation made ahead of time -- on the
		// thread that brings the device up -- and create() takes it over if it is large enough (else it is dropped).
		// bytesFor: an upper bound of the allocation for a mesh of that many triangles.
		static void reserve(int device, size_t bytes);
		static size_t bytesFor(size_t triangles, size_t ao_directions);
		~DeviceScene();
		DeviceScene(const DeviceScene &) = delete;
		DeviceScene &operator=(const DeviceScene &) = delete;
		bool servesOptions(const RayTracer::Options &options) const;

		SceneBuffers buffers() const { return SceneBuffers{ d_nodes, d_walk, d_tris, d_shade, d_ao }; }
		int device() const { return device_index; }
		size_t bytes() const { return device_bytes; }  // requested on the device
		uint32_t nodeCount() const { return node_count; }
		uint32_t triCount() const { return tri_count; }
		uint32_t aoDirs() const { return ao_dirs; }
		const SceneFacts &facts() const { return scene_facts_; }

	private:
		DeviceScene() = default;
		int device_index = 0;
		void *arena = nullptr;  // the one allocation the five arrays below point into
		void *d_nodes = nullptr, *d_walk = nullptr, *d_tris = nullptr, *d_shade = nullptr, *d_ao = nullptr;
		size_t device_bytes = 0;
		uint32_t node_count = 0, tri_count = 0, ao_dirs = 0;
		SceneFacts scene_facts_;
		// what of the options went into it
		bool ao_on = false;
		int ao_method = 0;
		unsigned int ao_samples = 0;
		int ao_alpha_min = 0, ao_alpha_max = 0;
		float walk_distance = 0.0f;
};

class DeviceRenderer {
	public:
		// `ring_slot`: this renderer's index among the renderers that take frames in turn on its GPU (FrameRing), -1 for
		// a renderer on its own.  It picks the priority class of the renderer's stream, see the constructor.
		// `spare_top_class`: leave the highest stream-priority class to somebody else (the ring's exchange step, whose
		// stream must never queue behind a frame).
		DeviceRenderer(const RayTracer::Options &options, int device, unsigned int rank, unsigned int nranks, int ring_slot = -1,
		               bool spare_top_class = false);
		~DeviceRenderer();
		DeviceRenderer(const DeviceRenderer &) = delete;
		DeviceRenderer &operator=(const DeviceRenderer &) = delete;

		// Blocking upload of a packed scene; returns the bytes requested on the device (scene + this renderer's buffers).
		size_t upload(const PackedScene &scene);
		// Renders a scene that is on the device already (DeviceScene::create on this renderer's device, for options that
		// agree with this renderer's: std::invalid_argument otherwise); returns the bytes of this renderer's own buffers.
		// `layout_from`: a renderer of the SAME scene, options and partition that has adopted it already (a ring's first
		// host): the hit list's layout is copied from it instead of being counted again.
		size_t adopt(std::shared_ptr<const DeviceScene> scene, const DeviceRenderer *layout_from = nullptr);
		const std::shared_ptr<const DeviceScene> &deviceScene() const { return scene_on_device; }

		// Enqueues the ray-casting kernel for this rank's bands on the stream.
		void enqueueRender();
		// Enqueues the device-side resize of this rank's bands into the compact
		// uint8 band buffer (localRows() x width bytes).
		void enqueueResize();
		// Same, but writes into caller-provided DEVICE memory (e.g. a torch tensor).
		void enqueueResizeInto(void *device_u8);
		// One whole frame -- counters, primary pass, ordering step, ambient-occlusion pass, resolve, device resize into
		// `device_u8` (nullptr: this renderer's own band buffer) -- enqueued as ONE replay of a captured hipGraph
		// (setGraphMode(false): as the same sequence of launches).  The graph is captured at the first call and again
		// whenever what it bakes in has changed (scene, destination, stream, share of the device).
		void enqueueFrame(void *device_u8 = nullptr);
		// Captures the frame for that destination now (graph mode; a few milliseconds), so that the first enqueueFrame for
		// it is a replay like every later one.
		void prepareFrame(void *device_u8 = nullptr);
		void setGraphMode(bool on) { graph_mode = on; }
		bool sceneReady() const { return scene_ready; }
		bool graphMode() const { return graph_mode; }
		// Waits for everything enqueued so far and folds pending event pairs into
		// the kernel-time statistics.
		void synchronize();
		void waitForStream();  // the wait alone (synchronize() then finds the stream idle)
		// Frame time stamps: with an epoch event set (recorded by the caller on any stream of this device), synchronize()
		// also keeps, for the last frame it folds, the milliseconds from the epoch to the frame's begin, to the start and
		// the end of its ambient-occlusion kernel (0 for a frame without one) and to its end -- what a test needs to see
		// that frames of different renderers really overlap on the device.
		void setEpochEvent(void *event) { epoch_event = event; }
		// A frame replayed from the captured graph holds no HIP events that could be timed; with this on, synchronize()
		// reads the device-clock stamps the kernels left in the frame's counters (a 24-byte blocking copy per frame)
		// for the ao_kernel times.  Plain launches always carry their own events.
		void setKeepStamps(bool on) { keep_stamps = on; }
		const float *lastFrameTimes() const { return last_times; }

		// The compact uint8 band buffer on this renderer's device (localRows() x width bytes), valid after
		// enqueueResize() has completed; and the stream everything is enqueued on.
		const void *deviceBands() const { return d_u8; }
		void *streamHandle() const { return stream; }
		// How many hosts take frames in turn on this GPU.  Alone, the persistent ambient-occlusion pass fills the chip
		// (8 workgroups per CU); in company it leaves room -- 4.5 per CU with up to five hosts, 3 from six on (the small
		// shares of a frame split over many GPUs) -- so that the other frames' passes run beside it all the time and not
		// only while it runs out (rounds 2-3 measured 5.5 per CU; re-swept on round 4's kernels, profiles/r04_notes.md
		// section 14: three hosts, whole frame, 5.5 -> 4.5 per CU: headline -1.5 %, interior -3.5 %, 600^2 +-0; an eighth
		// of the headline frame with twelve hosts 0.162 -> 0.153 ms, a quarter with six 0.289 -> 0.272).
		void setDeviceShare(unsigned hosts) {
			device_share = hosts < 1u ? 1u : hosts;
			kp.shared_device = device_share > 1u ? 1 : 0;
		}
		// Which form of the ambient-occlusion pass's node loop is launched: with (default) or without the look-ahead loads
		// of kernels.hip, OCRT_PF_SUCCESSORS -- same results, the faster one depends on the scene.  calibrateAoPrefetch()
		// renders a few frames one at a time with each form (plain launches, HIP events around the ao_kernel), keeps the
		// faster one and returns the two medians in ms (a frame ring does this once per uploaded scene).
		void setAoPrefetch(bool on);
		// Where the tiles' any-hit walks enter the tree (sizeHitList, entry_kernel): of the tiles with hits, how many walk
		// less than the whole tree, and the mean share of the node records a tile's rays are confined to.
		struct WalkEntries {
			uint32_t tiles_hit, tiles_narrowed;
			double mean_share;         // ... by the tile's own interval (any ray from the tile)
			double mean_packet_share;  // ... by the intervals its packets use (a full tile's: one per table direction)
		};
		WalkEntries walkEntries() const;
		bool aoPrefetch() const { return ao_prefetch; }
		// How many frames of the scene the caller is going to render (default 1: the reference's use, one frame per
		// process, src/render.cc:86-111).  What an upload prepares beyond the scene itself only pays over a stream of frames:
		// the walk intervals (a second pass of the primary kernel + entry_kernel: ~0.55 ms at 1080p, worth ~0.04 ms per
		// headline frame, ~0.15 ms per frame of the interior scene) are made from FRAMES_WORTH_INTERVALS announced frames on,
		// and a frame ring -- which announces a stream -- also measures the tiles' costs and the form of the node loop.
		// Callable before or after the upload (after: the hit list is laid out again if the answer changes).
		static constexpr uint64_t FRAMES_WORTH_INTERVALS = 16;
		void expectFrames(uint64_t frames);
		uint64_t expectedFrames() const { return expected_frames; }
		bool walkIntervalsInUse() const { return intervals_in_use; }
		// What the tiles' ambient-occlusion packets really cost, and the order the pass claims them in made from that
		// (orderByMeasuredCost): `frames` frames one at a time (plain launches) whose AO pass books every claim's duration
		// -- device clock, between the workgroup's barriers -- to the claim's tiles.  Once per upload, for callers that
		// announce a stream of frames (a frame ring); returns false where there is nothing to measure.
		bool measureTileCosts(unsigned frames = 2);
		// ... and the order another renderer of the same frame (scene, options, partition) has arrived at
		void takeOrderFrom(const DeviceRenderer &other);
		bool orderIsMeasured() const { return tile_cost.size() == tile_count && tile_count != 0; }
		void setOrderPolicy(float heavy, float runway, float split_above = -1.0f);  // (split_above < 0: unchanged)
		// primary pass: tiles of cost class `above` or more (the leaves their packet stops at, 1 ... 64) are cast in quarters by
		// the four waves of a workgroup (0: none); re-makes the pass's list
		void setPrimarySplit(uint32_t above);
		// 0: the library's rule (two kernels), 1: fused, 2: two kernels -- same results
		void setFrameForm(int form);
		bool frameIsFused() const { return fusedFrame(); }
		// (test aid, include/rt_hip_debug.h: fills the hit list with NaN patterns between frames -- a fused frame that read a
		// record its own primary work had not written yet would show it)
		void poisonHitList();
		// (diagnostics / experiments, include/rt_hip_debug.h: the list as it is, and a list made elsewhere put in its place --
		// any order of the same tiles renders the same image)
		void tileOrder(std::vector<uint32_t> &order, std::vector<uint32_t> &constants, std::vector<uint32_t> &words, std::vector<float> &cost) const;
		void setTileOrder(const std::vector<uint32_t> &order, const std::vector<uint32_t> &constants);
		bool calibrateAoPrefetch(float *ms_without = nullptr, float *ms_with = nullptr);
		// (A scene far beyond the caches is another matter: its pass waits for memory, and what it needs is loads in
		// flight -- every host keeps the full grid: 2 M-triangle field, three hosts, 11.93 ms per frame with 4.5 per CU,
		// 11.64 with 5.5, 11.32 with 8.)
		uint32_t aoWorkgroups() const {
			if (scene_beyond_caches)
				return compute_units * 8u;
			return device_share >= 6u ? compute_units * 3u : device_share > 1u ? compute_units * 9u / 2u : compute_units * 8u;
		}
		uint32_t globalRowOf(uint32_t local_row) const;  // output row of a local band row (may be >= height: padding)

		void downloadFloat(float *host_image);          // full totalWidth x totalHeight (rows of other ranks' bands: 0)
		void downloadResizedLocal(unsigned char *host); // localRows() x width, compact
		void downloadResizedFull(unsigned char *host);  // width x height (needs nranks == 1)

		// Run on an externally owned hipStream_t (nullptr = the HIP default stream) /
		// go back to the private non-blocking stream.
		void setStream(void *hip_stream);
		void usePrivateStream();

		RenderStats stats();   // ray counts of the last frame (device counters)
		float lastKernelMs() const { return last_ms; }      // all passes of the last frame
		float lastAoMs() const { return last_ao_ms; }       // the ao_kernel launch alone (0 when the frame has no AO pass)
		double totalKernelMs() const { return total_ms; }
		double totalAoMs() const { return total_ao_ms; }
		uint64_t kernelLaunches() const { return launches; }
		uint64_t aoLaunches() const { return ao_launches; }  // frames whose ao_kernel launch was timed (totalAoMs)
		void resetTimers() { total_ms = total_ao_ms = 0; launches = ao_launches = 0; last_ms = last_ao_ms = 0; }

		uint32_t localRows() const { return local_out_rows; }  // output rows this rank owns
		uint32_t width() const { return opts.width; }
		uint32_t height() const { return opts.height; }
		const RayTracer &rayTracer() const { return rt; }
		const KernelParams &params() const { return kp; }
		int deviceIndex() const { return device; }
		std::string deviceName() const;

	private:
		void freeScene();
		void useDevice() const;

		RayTracer::Options opts;
		RayTracer rt;
		int device;
		Partition part;
		KernelParams kp;
		uint32_t grid;            // supersample grid side
		uint32_t local_out_rows;
		void *own_stream, *stream;
		std::shared_ptr<const DeviceScene> scene_on_device;
		void *d_image, *d_u8, *d_hits, *d_occluded, *d_tile_hits, *d_tile_base, *d_tile_entry, *d_order, *d_counters;
		void *d_primary_order = nullptr, *d_order_need = nullptr, *d_tile_ready = nullptr;  // the fused frame kernel's block order, what every claim needs of it, and the flags (kernels/frame.hip.h)
		size_t hit_slots;     // slots of the hit list: the scene's hit sub-pixels in this rank's bands (sizeHitList)
		size_t entryBytes() const { return (size_t) tile_count * kp.entry_stride * 2 * sizeof(uint32_t); }  // the tiles' walk intervals (entry_kernel)
		std::shared_ptr<void> tile_entry_owner;  // d_tile_entry: one table for the hosts of a ring (sizeHitList)
		void allocEntries(size_t bytes);
		void sizeHitList(const DeviceRenderer *layout_from);  // counts the hits per tile with one pass of the primary kernel (or copies another renderer's count) and sizes the list by them
		// The order the ambient-occlusion pass claims the tiles in, made once per upload on the host (orderTiles): from the
		// tile words of the pass that sizes the hit list (hit count | cost class), or from measured costs per tile.
		std::vector<uint32_t> tile_words, order_host;
		std::vector<float> tile_cost;  // measured: device-clock ticks per tile (empty: not measured)
		std::array<std::array<uint32_t, 3>, XCD_GROUPS> queue_static{};  // per group: non-empty tiles, sum of cost classes, hit sub-pixels
		std::vector<uint32_t> primary_order_host, order_need_host;  // fused frame kernel: the groups' 2 x 2 tile blocks in the order the AO claims want them; per AO entry the blocks needed so far
		std::array<uint32_t, XCD_GROUPS> primary_blocks{};    // ... and how many each group has
		std::vector<uint32_t> blocks_by_cost_host;            // primary_kernel: the groups' 2 x 2 blocks by falling cost (0xFFFFFFFF: no block)
		void *d_blocks_by_cost = nullptr;
		size_t blocks_by_cost_capacity = 0;  // (entries)
		size_t primary_quartered = 0;        // tiles the primary pass casts in quarters (orderBlocksByCost)
		bool primary_by_cost = true;
		void orderPrimaryBlocks();
		void orderBlocksByCost();
		void uploadBlocksByCost();
		// Which form a frame with UNIFORM ambient occlusion takes: two kernels (the rule), or the two ray passes as one
		// persistent launch (kernels/frame.hip.h: built, bit-exact, measured slower -- fusedFrame() says by how much).
		enum class FrameForm { AUTO, FUSED, SEPARATE } frame_form = FrameForm::AUTO;
		bool fusedFrame() const;
		void checkFrameHealth();  // throws DeviceError if a wave of the fused frame kernel ever gave up waiting (FrameCounters::stalled)
		std::array<uint32_t, XCD_GROUPS> split_tiles{};  // per group: the tiles at the head of its list that are claimed half a tile at a time
		struct OrderPolicy {  // (orderByMeasuredCost; swept in profiles/r05_order_policies.txt)
			float heavy = 2.0f;   // tiles beyond this many reference costs (the upper quartile) are claimed first
			float runway = 2.0f;  // what is held back for the end, by falling cost: this many reference claims per workgroup
			uint32_t primary_split_waves = 4096;  // ... unless those tiles, four waves each, are more than this many waves (half the chip's slots: then the slots are full anyway)
			uint32_t primary_split_above = 64;  // primary pass: tiles of this cost class (leaves their packet stops at, 1 ... 64) or more are cast in quarters (0: none)
			float split_above = 0.25f;  // tiles that cost more than this share of the pass's ideal length: half a tile per claim (0: none)
		} order_policy;
		void orderTiles();
		std::vector<uint32_t> orderByMeasuredCost(const std::vector<float> &cost, uint32_t *split = nullptr) const;
		void installOrder(const std::vector<uint32_t> &order, const std::array<std::array<uint32_t, 3>, XCD_GROUPS> &constants,
		                  const std::array<uint32_t, XCD_GROUPS> &splits = {});
		size_t image_bytes;  // float image of this rank's bands
		size_t tile_count;
		uint32_t compute_units;
		uint32_t device_share;  // hosts that take frames in turn on this GPU (setDeviceShare), 1 = this one alone
		bool scene_beyond_caches = false;  // the uploaded scene is several times the L2s (upload: wider strips, full AO grids)
		bool scene_ready, frame_ready;
		struct FrameEvents {
			void *start, *ao_start, *ao_stop, *stop;  // frame begin, around the ao_kernel launch alone, frame end
			bool ao_timed = false;                    // the frame had an AO pass (ao_start / ao_stop were recorded)
			bool graph_ao = false;                    // ... as a replayed graph: the graph's own events hold the AO times
		};
		std::vector<FrameEvents> pending_events, free_events;
		FrameEvents takeEvents();
		void launchFrame(void *device_u8, void *ao_start, void *ao_stop, void *tile_cost = nullptr);  // the launches of one frame, resize included
		// the captured frame (enqueueFrame) and what it was captured for
		struct FrameGraph {
			void *graph = nullptr, *exec = nullptr;
			void *dst = nullptr, *stream = nullptr;
			uint64_t scene_version = 0;
			uint32_t device_share = 0;
			bool valid = false, ao_events = false;
		};
		std::vector<FrameGraph> frame_graphs;  // one per destination seen lately (a ring alternates between two per renderer)
		const FrameGraph *frameGraphFor(void *dst);
		void dropFrameGraphs();
		bool graph_mode;
		uint64_t expected_frames = 1;
		bool intervals_in_use = false;
		bool ao_prefetch;
		uint64_t scene_version;
		uint32_t ao_blocks_override;  // (debug-knob builds: OCRT_AO_BLOCKS)
		void *epoch_event;
		bool keep_stamps;
		float last_times[4];
		float last_ms, last_ao_ms;
		double total_ms, total_ao_ms;
		uint64_t launches, ao_launches;
};

// kernels.hip
void preload_kernels();
#ifdef OCRT_OCML_BUILTINS
uint32_t ocml_ao_table(void *table, uint32_t rings, int alpha_min, int alpha_max, uint32_t capacity);
#endif
void launch_primary(const SceneBuffers &scene, float *image, void *hits, void *occluded_of, void *tile_hits,
                    const void *tile_base, void *counters, const KernelParams &P, void *stream, const void *blocks_by_cost = nullptr);
void launch_entries(const SceneBuffers &scene, const void *hits, const void *tile_hits, const void *tile_base, void *tile_entry,
                    const KernelParams &P, void *stream);
void launch_ao(const SceneBuffers &scene, void *hits, void *occluded_of, void *order, const void *tile_base, const void *tile_entry,
               void *counters, const KernelParams &P, uint32_t workgroups, bool prefetch, void *stream, void *event_before_ao,
               void *event_after_ao, void *tile_cost = nullptr);
void launch_frame(const SceneBuffers &scene, float *image, void *hits, void *occluded_of, void *tile_hits, void *order,
                  const void *primary_order, const void *order_need, void *tile_ready, const void *tile_base, const void *tile_entry,
                  void *counters, const KernelParams &P, uint32_t workgroups, bool prefetch, void *stream, void *event_before, void *event_after);
void launch_finish(float *image, const void *hits, const void *occluded_of, const void *tile_base,
                   unsigned char *out, const KernelParams &P, uint32_t out_width, uint32_t n, uint32_t local_out_rows, void *stream,
                   void *counters);
void launch_occluded_sum(const void *occluded_of, size_t slots, void *counters, void *stream);
void launch_resize(const float *tmp, unsigned char *out, const KernelParams &P, uint32_t out_width, uint32_t n,
                   uint32_t local_out_rows, void *stream);

}  // namespace ocrt
