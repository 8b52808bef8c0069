/*
 * ref_shim.cc -- lets the reference's OWN kernel source run on the CPU.
 *
 * TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles the unmodified
 * /root/reference/src/intersect_kernel.cl with ROCm's clang as OpenCL C for
 * x86-64 (one object per -D configuration, exactly the macro set
 * reference src/opencl_host.cc:42-53 emits).  That object calls the OpenCL
 * builtins below through their C++-mangled names; this file supplies them with
 * the IEEE definitions of SURVEY.md 8a-0.3 and a driver that walks the
 * NDRange.  Outputs go to oracle/_ref/ only (git-ignored); no reference source
 * is copied into this repository.
 */
#include <cmath>
#include <cstddef>
#include <cstdint>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef float float4 __attribute__((vector_size(16)));

static thread_local size_t g_gid[2];

/* ---- OpenCL builtins the kernel object imports (nm: _Z13get_global_idj, _Z3dotDv4_fS_, ...) ---- */
size_t get_global_id(unsigned int dim) { return dim < 2 ? g_gid[dim] : 0; }

float dot(float4 a, float4 b) { return ((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]; }
float4 cross(float4 a, float4 b) {
	float4 r = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0], 0.0f };
	return r;
}
float length(float4 a) { return sqrtf(dot(a, a)); }
float4 normalize(float4 a) {
	const float l = length(a);
	float4 r = { a[0] / l, a[1] / l, a[2] / l, a[3] / l };
	return r;
}
float max(float a, float b) { return fmaxf(a, b); }
float min(float a, float b) { return fminf(a, b); }
int max(int a, int b) { return a > b ? a : b; }
float fabs(float a) { return fabsf(a); }
float clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
float sin(float a) { return sinf(a); }
float cos(float a) { return cosf(a); }
float sqrt(float a) { return sqrtf(a); }
float acos(float a) { return acosf(a); }
float cospi(float a) { return (float) ::cos(M_PI * (double) a); }
float sinpi(float a) { return (float) ::sin(M_PI * (double) a); }

/* The kernel entry point of the per-configuration object. */
extern "C" void intersect(const uint32_t *faces, const uint32_t *nodes, const float4 *aabbs,
                          const float4 *vertices, const float4 *normals, float *image);

/* Walk rows [y0,y1) of the NDRange (global size = width x height). */
extern "C" int ref_render_rows(const uint32_t *faces, const uint32_t *nodes, const float *aabbs,
                               const float *vertices, const float *normals, float *image,
                               uint32_t width, uint32_t y0, uint32_t y1, int nthreads) {
	int used = 1;
#ifdef _OPENMP
	if (nthreads <= 0)
		nthreads = omp_get_max_threads();
	used = nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
	for (int64_t y = (int64_t) y0; y < (int64_t) y1; ++y) {
		for (uint32_t x = 0; x < width; ++x) {
			g_gid[0] = x;
			g_gid[1] = (size_t) y;
			intersect(faces, nodes, (const float4 *) aabbs, (const float4 *) vertices,
			          (const float4 *) normals, image);
		}
	}
	return used;
}
