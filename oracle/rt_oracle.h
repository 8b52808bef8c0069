/*
 * rt_oracle.h -- CPU oracle for the ray-casting hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * OpenCL kernel (reference: src/intersect_kernel.cl) and of the host-side
 * supersample box filter (reference: src/ray_tracer.cc:3-16).  It exists so
 * that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
 * check / time the HIP path against it.  Nothing in the product path
 * (opencl_raytracer_amd/, include/) may include, link or call it.
 *
 * Parity pin: tests/golden/ holds PGM digests and float dumps produced by the
 * reference's own kernel source compiled for x86-64 (oracle/Makefile ->
 * oracle/_ref/), and tests/test_oracle_golden.py checks this restatement
 * against them.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors the -D macro set the reference bakes into the kernel at JIT time
 * (reference: src/opencl_host.cc:42-53). width/height are the SUPERSAMPLED
 * ("total") dimensions, exactly like WIDTH/HEIGHT there. */
typedef struct orc_params {
	uint32_t width;
	uint32_t height;
	float focal_length;
	int32_t shading_enable;
	int32_t ao_enable;
	float ao_max_distance;
	uint32_t ao_num_samples;
	int32_t ao_method; /* 0 = UNIFORM, 1 = RANDOM */
	int32_t ao_alpha_min;
	int32_t ao_alpha_max;
} orc_params;

/* The five read-only buffers of the reference kernel, in its layouts
 * (reference: src/intersect_kernel.cl:278): float4 arrays are 16-byte elements. */
typedef struct orc_scene {
	const uint32_t *faces;   /* 3T leaf-ordered vertex ids */
	const uint32_t *nodes;   /* pre-order subtree sizes, nodes[0] = count */
	const float *aabbs;      /* float4[2*count]: [2i]=min, [2i+1]=max */
	const float *vertices;   /* float4[V] */
	const float *normals;    /* float4[V] */
} orc_scene;

/* Work counters of the reference traversal (node visits = aabb_intersect
 * calls, tri tests = triangle_intersect calls). */
typedef struct orc_counters {
	uint64_t primary_rays;
	uint64_t primary_hits;
	uint64_t primary_node_visits;
	uint64_t primary_tri_tests;
	uint64_t ao_rays;
	uint64_t ao_occluded;
	uint64_t ao_node_visits;
	uint64_t ao_tri_tests;
} orc_counters;

/* Direction table of the UNIFORM hemisphere (reference:
 * src/intersect_kernel.cl:237-246).  Writes up to cap triples (xs,ys,zs) and
 * returns the number of directions the reference would cast per hit pixel. */
uint32_t orc_ao_table(const orc_params *p, float *xyz, uint32_t cap);

/* Renders rows [y0,y1) of the float image (row-major, width*height floats,
 * only the selected rows are written).  counters may be NULL.  nthreads<=0
 * means "all OpenMP threads". Returns the number of threads used. */
int orc_render_rows(const orc_params *p, const orc_scene *s, float *image,
                    uint32_t y0, uint32_t y1, orc_counters *counters, int nthreads);

/* Whole image. */
int orc_render(const orc_params *p, const orc_scene *s, float *image,
               orc_counters *counters, int nthreads);

/* Supersample box filter + 8-bit quantisation (reference: src/ray_tracer.cc:3-16).
 * n = (unsigned)sqrt(n_super_samples); tmp is (width*n) x (height*n). */
void orc_resize(const float *tmp, uint8_t *image, uint32_t width, uint32_t height,
                uint32_t n_super_samples);

/* (unsigned)sqrt(n) exactly as RayTracer's ctor does it (reference:
 * include/ray_tracer.h:33-34). */
uint32_t orc_ss_factor(uint32_t n_super_samples);

#ifdef __cplusplus
}
#endif
#endif
