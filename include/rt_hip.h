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
 * This header is the boundary itself: options, mesh + BVH on the CPU, the render host (create / upload / render /
 * download / printInfo), errors.  What goes beyond the reference's seam lives beside it: rt_hip_ring.h (a steady stream
 * of frames, several GPUs: the frame ring, the band partition, the RCCL exchange step), rt_hip_debug.h (timers,
 * statistics and experiment switches used by tests, bench.py and the analysis tools).
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

/* New: announce how many frames of the uploaded scene this host is going to render (default 1 -- the reference's use:
 * one frame per process, src/render.cc:86-111).  From 16 frames on an upload also prepares the walk intervals of the
 * tiles' any-hit packets (~0.5 ms once, 3 ... 16 % per frame after) and the re-ordered, grown copy of the walk records that
 * lets the primary rays' closest-hit walk prune (2-3 ms of CPU once; announce BEFORE rt_upload for that one); before or
 * after rt_upload.  Results never depend on it.  A frame ring announces a stream by itself. */
int rt_expect_frames(rt_host *h, uint64_t frames);

/* Ray counts of the last frame, and the HIP-event time of its ray-casting passes on the launch stream in ms (the region
 * the reference calls "Rendering image", src/render.cc:109-111).  rt_get_stats waits for the host's stream; the count of
 * occluded rays is summed on the device when it is called (the frames do not sum it), so it describes the LAST frame the
 * host has been given -- ask before submitting the next one.  (Running totals and per-pass times: rt_hip_debug.h.) */
int rt_get_stats(rt_host *h, rt_stats *out);
float rt_last_kernel_ms(const rt_host *h);

/* OpenCLHost::printInfo, reference src/opencl_host.cc:76-119. */
void rt_print_info(void);

/* Number of visible HIP devices (0 without a GPU; never fails). */
int rt_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
