/*
 * rt_hip.h -- C ABI of the MI355X ray-casting path (libocrt_hip.so).
 *
 * The reference has no FFI layer: its seam is the C++ class OpenCLHost used by
 * main() (reference src/render.cc:84,98,110,115; include/opencl_host.h:6-144).
 * The C++ drop-in for that class is HipHost (opencl_raytracer_amd/csrc/hip_host.h).
 * This header is the same boundary flattened to plain C -- opaque handles, plain
 * pointers and sizes, no C++ or torch types -- so that any host language can bind
 * it (INTEGRATION.md shows the bindings).  Each entry point cites the reference
 * interface it stands for.
 *
 * Conventions: functions returning int return 0 on success and a negative
 * RT_E_* code on failure; rt_last_error() then holds a message (thread-local).
 * Nothing here exits the process or throws across the boundary.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_E_INVALID (-1)   /* bad argument / malformed scene arrays */
#define RT_E_NO_DEVICE (-2) /* no HIP device visible (reference: "No device found") */
#define RT_E_DEVICE (-3)    /* a HIP call failed (reference: OpenCLHost::check) */
#define RT_E_STATE (-4)     /* call order violated (render before upload, ...) */
#define RT_E_IO (-5)        /* mesh file could not be read / parsed */

/* RayTracer::Options, field for field (reference include/ray_tracer.h:17-30). */
typedef struct rt_options {
	uint32_t width;
	uint32_t height;
	float focal_length;
	uint32_t n_super_samples;
	int32_t enable_shading;
	int32_t enable_ao;
	float ao_max_distance;
	uint32_t ao_num_samples;
	int32_t ao_method;   /* 0 = UNIFORM, 1 = RANDOM */
	int32_t ao_alpha_min;
	int32_t ao_alpha_max;
	int32_t bvh_method;  /* 0 = CUT_LONGEST_AXIS, 1 = SURFACE_AREA_HEURISTIC */
} rt_options;

typedef struct rt_stats {
	uint64_t primary_rays;
	uint64_t primary_hits;
	uint64_t ao_rays;
	uint64_t ao_occluded;
} rt_stats;

typedef struct rt_host rt_host;   /* one render host = one OpenCLHost */
typedef struct rt_ring rt_ring;   /* several render hosts of one scene on one GPU taking frames in turn */
typedef struct rt_scene rt_scene; /* CPU-side mesh + BVH (the input producer) */

const char *rt_last_error(void);

/* The CLI defaults, reference src/render.cc:17. */
void rt_options_default(rt_options *out);
/* RayTracer::totalWidth / totalHeight, reference include/ray_tracer.h:33-34. */
uint32_t rt_total_width(const rt_options *o);
uint32_t rt_total_height(const rt_options *o);
/* RayTracer::resize on the CPU, reference src/ray_tracer.cc:3-16. */
int rt_resize_cpu(const rt_options *o, const float *tmp, uint8_t *image);

/* ---- scene build on the CPU (feeds rt_upload) ------------------------------ */
/* load_off_mesh + compute_vertex_normals, reference src/mesh.cc:7-67,95-139. */
rt_scene *rt_scene_load_off(const char *path);
/* Same from arrays: vertices4 = float4[num_vertices], faces = uint32[3*num_faces]. */
rt_scene *rt_scene_from_arrays(const float *vertices4, uint32_t num_vertices, const uint32_t *faces, uint32_t num_faces);
void rt_scene_free(rt_scene *s);
uint32_t rt_scene_num_vertices(const rt_scene *s);
uint32_t rt_scene_num_faces(const rt_scene *s);
/* BVH::buildBVH + the leaf-order face sort, reference src/bvh.cc:98-111 and
 * src/render.cc:88-95.  method as rt_options.bvh_method. */
int rt_scene_build_bvh(rt_scene *s, int method);
uint32_t rt_scene_num_nodes(const rt_scene *s);
/* Read-only views, valid until the scene is rebuilt or freed.  float4 arrays
 * are 16-byte elements (x, y, z, 0). */
const float *rt_scene_vertices(const rt_scene *s);    /* float4[num_vertices] */
const float *rt_scene_vnormals(const rt_scene *s);    /* float4[num_vertices] */
const uint32_t *rt_scene_faces(const rt_scene *s);    /* uint32[3*num_faces], file order */
const uint32_t *rt_scene_nodes(const rt_scene *s);    /* uint32[num_nodes] subtree sizes */
const float *rt_scene_aabbs(const rt_scene *s);       /* float4[2*num_nodes] */
const uint32_t *rt_scene_triangles(const rt_scene *s);    /* uint32[num_faces] face id per leaf */
const uint32_t *rt_scene_sorted_faces(const rt_scene *s); /* uint32[3*num_faces] leaf order */

/* ---- render host: OpenCLHost --------------------------------------------- */
/* OpenCLHost::OpenCLHost(const RayTracer&), reference src/opencl_host.cc:15-75.
 * Device: $OCRT_DEVICE or 0.  NULL on failure (RT_E_NO_DEVICE when no GPU). */
rt_host *rt_create(const rt_options *o);
/* Same on an explicit device and for one rank of a band-partitioned image
 * (rank 0 of 1 = whole image); new, the reference is single-device. */
rt_host *rt_create_on(const rt_options *o, int device, uint32_t rank, uint32_t nranks);
/* Error code of the last failed rt_create* / rt_scene_* on this thread. */
int rt_last_error_code(void);
void rt_destroy(rt_host *h);

/* OpenCLHost::upload, reference src/opencl_host.cc:120-136.  Synchronous copy;
 * the caller may free its arrays on return.  faces: 3*num_faces leaf-ordered
 * vertex ids; nodes: num_nodes subtree sizes; aabbs: float4[2*num_nodes];
 * vertices / vnormals: float4[num_vertices]. */
int rt_upload(rt_host *h, const uint32_t *faces, uint32_t num_faces, const uint32_t *nodes, uint32_t num_nodes,
              const float *aabbs, const float *vertices, uint32_t num_vertices, const float *vnormals);
/* Convenience: rt_upload straight from a built rt_scene. */
int rt_upload_scene(rt_host *h, const rt_scene *s);

/* OpenCLHost::operator()(), reference src/opencl_host.cc:137-149: render the
 * frame (this rank's bands), block until done. */
int rt_render(rt_host *h);
/* Split form of the above for pipelining: enqueue only / wait. */
int rt_render_async(rt_host *h);
int rt_sync(rt_host *h);

/* OpenCLHost::download, reference src/opencl_host.cc:150-153: the
 * total_width*total_height float image into caller memory. */
int rt_download(rt_host *h, float *image);
/* New: RayTracer::resize fused onto the device; width*height bytes, needs an
 * unpartitioned host. */
int rt_download_u8(rt_host *h, uint8_t *image);

/* Band-partitioned form: rows this rank owns in the compact band buffer, the
 * download of that buffer (rt_local_rows * width bytes), and the global output
 * row of a local row (>= height: padding). */
uint32_t rt_local_rows(const rt_host *h);
int rt_download_u8_local(rt_host *h, uint8_t *rows);
uint32_t rt_local_to_global_row(const rt_host *h, uint32_t local_row);
/* The same partition arithmetic without a device (for gather/assembly code):
 * output rows rank `rank` of `nranks` owns, and the global row of a local one. */
uint32_t rt_partition_local_rows(const rt_options *o, uint32_t rank, uint32_t nranks);
uint32_t rt_partition_global_row(const rt_options *o, uint32_t rank, uint32_t nranks, uint32_t local_row);
/* Enqueue the device resize into caller-owned DEVICE memory of
 * rt_local_rows*width bytes (e.g. the send buffer of an RCCL gather). */
int rt_resize_into_device(rt_host *h, void *device_u8);
/* Run this host's work on a caller-owned hipStream_t (NULL = the HIP default
 * stream) / return to the host's private non-blocking stream. */
int rt_set_stream(rt_host *h, void *hip_stream);
int rt_use_private_stream(rt_host *h);
/* The hipStream_t this host's work is enqueued on right now (its private stream unless rt_set_stream replaced it),
 * for callers that order their own streams against it with events -- e.g. two hosts of the same scene on one GPU
 * taking frames alternately, so that one frame's last workgroups and the next frame's first share the device
 * (bench.py).  No counterpart in the reference: its queue is private to OpenCLHost (include/opencl_host.h:129). */
int rt_get_stream(rt_host *h, void **hip_stream);
/* Tell a host that `hosts` of them (it included) take frames in turn on its GPU: its persistent ambient-occlusion pass
 * then leaves part of the chip to the other frames' passes instead of filling it (1, the default: alone). */
int rt_set_device_share(rt_host *h, unsigned int hosts);
/* New: announce how many frames of the uploaded scene this host is going to render (default 1 -- the reference's use:
 * one frame per process, src/render.cc:86-111).  From 16 frames on an upload also prepares the walk intervals of the
 * tiles' any-hit packets (~0.5 ms once, 3 ... 16 % per frame after); before or after rt_upload.  Results never depend on
 * it.  A frame ring announces a stream by itself. */
int rt_expect_frames(rt_host *h, uint64_t frames);

/* Ray counts of the last frame and HIP-event timing of the ray-casting passes
 * on the launch stream (last frame, running total in ms and number of frames
 * since the reset): *_kernel_ms covers every pass of a frame, *_ao_ms the
 * launch of the ambient-occlusion kernel alone (0 for frames without AO).
 * Waits for the host's stream; the count of occluded rays is summed on the device when this is called (the frames do not
 * sum it), so it describes the LAST frame the host has been given -- ask before submitting the next one. */
int rt_get_stats(rt_host *h, rt_stats *out);
float rt_last_kernel_ms(const rt_host *h);
double rt_total_kernel_ms(const rt_host *h);
float rt_last_ao_ms(const rt_host *h);
double rt_total_ao_ms(const rt_host *h);
uint64_t rt_kernel_launches(const rt_host *h);
void rt_reset_timers(rt_host *h);

/* ---- frame ring: a steady stream of frames behind the same seam ------------------------------------------------
 * The reference renders one blocking frame per OpenCLHost::operator()() (src/opencl_host.cc:137-149, called once by
 * src/render.cc:109-111).  A ring is `hosts` render hosts of ONE scene on ONE GPU -- each with its own stream
 * (consecutive hosts in different priority classes, hence different hardware queues) and its own captured hipGraph,
 * so that a frame costs the CPU one graph launch -- that take frames in turn: the next frames' passes fill the wave
 * slots a finishing ambient-occlusion pass frees.  A ring of ONE host is the reference's blocking frame.  With a
 * communicator attached (rt_ring_attach_rccl) the ring also runs the one exchange step of a multi-GPU frame, the
 * gather of the ranks' 8-bit bands on rank 0 and the assembly of the image there, behind the next frames.
 * Every frame is complete: all ray passes + the device resize (RayTracer::resize, src/ray_tracer.cc:3-16). */
rt_ring *rt_ring_create(const rt_options *o, int device, uint32_t rank, uint32_t nranks, uint32_t hosts);
void rt_ring_destroy(rt_ring *r);
/* OpenCLHost::upload (src/opencl_host.cc:120-136) for every host of the ring; same arguments as rt_upload. */
int rt_ring_upload(rt_ring *r, const uint32_t *faces, uint32_t num_faces, const uint32_t *nodes, uint32_t num_nodes,
                   const float *aabbs, const float *vertices, uint32_t num_vertices, const float *vnormals);
int rt_ring_upload_scene(rt_ring *r, const rt_scene *s);
/* What the last upload put on the device: the bytes of the scene's arrays, how many copies of them exist among the
 * ring's hosts (ONE: the hosts share the arrays, like the single upload of src/opencl_host.cc:120-136) and the bytes
 * of everything requested, per-host frame buffers included.  Out pointers may be NULL. */
int rt_ring_device_bytes(const rt_ring *r, uint64_t *scene_bytes, uint32_t *scene_copies, uint64_t *total_bytes);
/* Run-time calibration (new; results never depend on it): the ambient-occlusion pass exists with and without
 * look-ahead loads in its node loop, and which is faster depends on the scene.  rt_ring_upload* measure both on the
 * uploaded scene (a few frames, ~15 ms) and keep the faster form for every host of the ring; rt_ring_set_calibration(r, 0)
 * before the upload switches that off (the hosts keep the default form, or what rt_set_ao_prefetch set).
 * rt_ring_calibration: ms per ao_kernel launch without / with the look-ahead as measured (0 = not measured) and the form
 * in use (1 = with).  Out pointers may be NULL. */
int rt_ring_set_calibration(rt_ring *r, int on);
int rt_ring_calibration(const rt_ring *r, float *ms_without, float *ms_with, int *prefetch_in_use);
int rt_set_ao_prefetch(rt_host *h, int on);
/* Walk intervals (new; results never depend on them): an upload finds, for every 8x8 tile of the host's band, the part of
 * the tree's node records -- an interval of the pre-order array -- outside which no leaf lies that an ambient-occlusion ray
 * of the tile can reach (AO_MAX_DISTANCE, src/intersect_kernel.cl:217, and the slab test of :21-61), once for any ray
 * from the tile and once for each table direction of a full tile (the 64 rays of one packet); the any-hit packets walk
 * their interval alone.  This reports, over the tiles with hits: their number, how many have a tile interval short of the
 * whole array, the mean share of the records inside the tile intervals, and the mean share inside the intervals the
 * packets actually use (1.0 = no narrowing: AO_MAX_DISTANCE of the scene's size).  Out pointers may be NULL.
 * RT_E_STATE without a scene. */
int rt_walk_entries(rt_host *h, uint32_t *tiles_hit, uint32_t *tiles_narrowed, double *mean_share, double *mean_packet_share);
uint32_t rt_ring_size(const rt_ring *r);        /* hosts */
/* Band buffers: frame f is rendered by host f % size into buffer f % slots, slots = 2 * size, so that a frame's bands
 * (and, with a communicator, its assembled image) stay untouched while the next `size` frames are submitted. */
uint32_t rt_ring_slots(const rt_ring *r);
uint32_t rt_ring_local_rows(const rt_ring *r);  /* output rows this rank owns (rt_local_rows) */
uint32_t rt_ring_in_flight(const rt_ring *r);   /* frames submitted and not yet collected */
/* Host `slot` of the ring as a BORROWED rt_host (statistics, timers, rt_download of its last frame): owned by the
 * ring, rt_destroy on it is a no-op. */
rt_host *rt_ring_host(rt_ring *r, uint32_t slot);
/* 1 (default): a frame is one replay of the host's captured hipGraph; 0: the same launches one by one. */
int rt_ring_set_graph_mode(rt_ring *r, int on);
/* Pacing of the submissions: a frame is enqueued no sooner than beta x (the running mean of the time per finished frame)
 * after the previous one, so that frames which finished together do not start their successors together and keep the
 * ring in lockstep.  Default 0.5; 0 switches it off.  (The reference submits one frame and waits for it:
 * src/opencl_host.cc:137-149; nothing to pace there.) */
int rt_ring_set_pacing(rt_ring *r, float beta);
/* Band buffer `slot` (< rt_ring_slots) is caller-owned DEVICE memory from now on (rt_ring_local_rows * width bytes;
 * NULL: the ring's own again), e.g. the send buffer of a caller-side collective. */
int rt_ring_bind_output(rt_ring *r, uint32_t slot, void *device_u8);
/* The non-blocking half of OpenCLHost::operator()(): enqueue the next frame on the next host; *frame (may be NULL)
 * receives its number.  RT_E_STATE when every host already has a frame in flight. */
int rt_ring_submit(rt_ring *r, uint64_t *frame);
/* The blocking half: wait for the OLDEST frame in flight.  Out (each may be NULL): its number, its band buffer's slot,
 * the device address of its bands (valid until frame + slots is submitted).  With a communicator attached its gather
 * is enqueued. */
int rt_ring_collect(rt_ring *r, uint64_t *frame, uint32_t *slot, const void **device_bands);
/* rt_ring_collect + a device-to-device copy of the bands into caller memory (complete on return). */
int rt_ring_collect_into_device(rt_ring *r, void *device_u8);
/* One frame of a steady stream: submit, then collect until at most hosts - 1 frames are in flight (one host: none).
 * rt_ring_run does `frames` such steps in one call; rt_ring_drain collects what is left and waits for the gathers. */
int rt_ring_step(rt_ring *r);
int rt_ring_run(rt_ring *r, uint32_t frames);
int rt_ring_drain(rt_ring *r);
/* The last collected frame on the device: its bands (= the image for nranks 1) without a communicator; with one, the
 * assembled width x height image on rank 0 (waits for its gather) and NULL on the other ranks.  rt_ring_download_last
 * copies width*height bytes to the host (OpenCLHost::download + RayTracer::resize, src/render.cc:114-123). */
int rt_ring_last_image_device(rt_ring *r, const void **device_u8);
int rt_ring_download_last(rt_ring *r, uint8_t *image);
/* Time stamps of collected frames: ms from the last rt_ring_reset_clock to the frame's begin, the start and the end of
 * its ambient-occlusion kernel, its end.  Begin and end are HIP events on the host's stream; the kernel's times are HIP
 * events too for plain launches, and for graph replays -- whose event nodes cannot be timed -- the device's 100 MHz
 * clock as the kernels stamped it into the frame's counters, read only after rt_ring_keep_frame_times(r, 1) (a small
 * blocking copy per frame; 0 without it).  Kept for the last 256 frames (RT_E_STATE for older ones). */
int rt_ring_reset_clock(rt_ring *r);
int rt_ring_keep_frame_times(rt_ring *r, int on);
int rt_ring_frame_times(const rt_ring *r, uint64_t frame, float ms[4]);
/* Sums over the ring's hosts since rt_ring_reset_timers: kernel time of whole frames / of the ao_kernel launches alone
 * (HIP events) and how many frames each sum covers.  Any out pointer may be NULL. */
int rt_ring_timers(rt_ring *r, double *kernel_ms, uint64_t *frames, double *ao_ms, uint64_t *ao_frames);
void rt_ring_reset_timers(rt_ring *r);
/* What the frames collected since the last rt_ring_reset_clock cost the CPU: seconds inside submit (the launches),
 * inside collect waiting for the device, inside collect otherwise; and their number.  Out pointers may be NULL. */
int rt_ring_cpu_times(const rt_ring *r, double *submit_s, double *wait_s, double *collect_s, uint64_t *frames);
/* The exchange step over RCCL, one process per GPU.  New: the reference is single-device (src/opencl_host.cc:16-32).
 * rt_rccl_unique_id (rank 0) fills 128 bytes that the caller hands to every rank (e.g. torch.distributed broadcast);
 * rt_ring_attach_rccl is collective (ncclCommInitRank with the ring's rank / nranks).  rt_rccl_available: 1 when
 * librccl.so.1 can be opened (it is opened at run time; the library loads without it). */
/* A ring's exchange step waits at most `seconds` (default 30) for a frame's gather: a peer rank that died or posted a
 * different number of frames makes rt_ring_run / rt_ring_drain / rt_ring_submit fail with RT_E_DEVICE instead of
 * hanging.  rt_ring_rccl_info: what the attached communicator says about itself -- ranks (ncclCommCount), RCCL's
 * version code (ncclGetVersion), -1 where unknown -- so that a run can state how many ranks RCCL really saw. */
int rt_ring_set_gather_timeout(rt_ring *r, double seconds);
int rt_ring_rccl_info(rt_ring *r, int *comm_ranks, int *rccl_version);
int rt_rccl_available(void);
int rt_rccl_unique_id(void *out128);
int rt_ring_attach_rccl(rt_ring *r, const void *unique_id128);
int rt_ring_rccl_self_test(rt_ring *r);  /* grouped self send/recv on the ring's communicator, checked */

/* OpenCLHost::printInfo, reference src/opencl_host.cc:76-119. */
void rt_print_info(void);
/* Number of visible HIP devices (0 without a GPU; never fails). */
int rt_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
