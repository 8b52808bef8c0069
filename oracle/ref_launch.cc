/*
 * ref_launch.cc -- runs the reference's OWN kernel on the MI355X.
 *
 * TEST INFRASTRUCTURE ONLY (never linked, loaded or called by the product).
 *
 * oracle/Makefile (target ref-kernel-gfx950) compiles the unmodified
 * /root/reference/src/intersect_kernel.cl as OpenCL C for amdgcn gfx950 with
 * ROCm's clang and ROCm's own OpenCL builtin library (opencl.bc / ocml.bc /
 * ockl.bc, linked by the driver itself -- no stand-in for any builtin) into a
 * code object under oracle/_ref/.  This helper loads such a code object with
 * the HIP module API, uploads the five scene arrays in the reference's layouts
 * (reference src/opencl_host.cc:120-136), launches `intersect` over the NDRange
 * (width x height) the way reference src/opencl_host.cc:145 does and reads the
 * float image back (reference :150-153).  The launch is timed with HIP events.
 *
 * The reference launches work-groups of 16 x 16 without a bounds guard
 * (SURVEY.md fact 0.8), so width and height must be multiples of the block
 * size passed in; callers use 16 x 16 where the reference's launch is valid and
 * say so where it is not.
 */
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

thread_local char g_error[512];

bool fail(const char *what, hipError_t err) {
	std::snprintf(g_error, sizeof g_error, "%s: %s (%s)", what, hipGetErrorName(err), hipGetErrorString(err));
	return false;
}

#define REF_HIP(call)                           \
	do {                                        \
		const hipError_t err_ = (call);         \
		if (err_ != hipSuccess) {               \
			fail(#call, err_);                  \
			goto done;                          \
		}                                       \
	} while (0)

}  // namespace

extern "C" const char *refgpu_last_error(void) { return g_error; }

/* Returns 0 on success.  kernel_ms receives the average launch time over
 * `repeats` launches (after one untimed launch). */
extern "C" int refgpu_run(const char *code_object, const uint32_t *faces, size_t n_face_words, const uint32_t *nodes,
                          size_t n_nodes, const float *aabbs, const float *vertices, size_t n_vertices,
                          const float *normals, float *image, uint32_t width, uint32_t height, uint32_t block_x,
                          uint32_t block_y, int repeats, float *kernel_ms) {
	g_error[0] = 0;
	int rc = -1;
	hipModule_t module = nullptr;
	hipFunction_t kernel = nullptr;
	hipEvent_t start = nullptr, stop = nullptr;
	void *d_faces = nullptr, *d_nodes = nullptr, *d_aabbs = nullptr, *d_vertices = nullptr, *d_normals = nullptr,
	     *d_image = nullptr;
	const size_t image_bytes = (size_t) width * height * sizeof(float);
	if (block_x == 0 || block_y == 0 || width % block_x || height % block_y) {
		std::snprintf(g_error, sizeof g_error, "NDRange %ux%u is not a multiple of the work-group %ux%u (the kernel has no bounds guard)",
		              width, height, block_x, block_y);
		return -2;
	}
	REF_HIP(hipSetDevice(0));
	REF_HIP(hipModuleLoad(&module, code_object));
	REF_HIP(hipModuleGetFunction(&kernel, module, "intersect"));
	REF_HIP(hipMalloc(&d_faces, n_face_words * sizeof(uint32_t)));
	REF_HIP(hipMalloc(&d_nodes, n_nodes * sizeof(uint32_t)));
	REF_HIP(hipMalloc(&d_aabbs, n_nodes * 2 * 16));
	REF_HIP(hipMalloc(&d_vertices, n_vertices * 16));
	REF_HIP(hipMalloc(&d_normals, n_vertices * 16));
	REF_HIP(hipMalloc(&d_image, image_bytes));
	REF_HIP(hipMemcpy(d_faces, faces, n_face_words * sizeof(uint32_t), hipMemcpyHostToDevice));
	REF_HIP(hipMemcpy(d_nodes, nodes, n_nodes * sizeof(uint32_t), hipMemcpyHostToDevice));
	REF_HIP(hipMemcpy(d_aabbs, aabbs, n_nodes * 2 * 16, hipMemcpyHostToDevice));
	REF_HIP(hipMemcpy(d_vertices, vertices, n_vertices * 16, hipMemcpyHostToDevice));
	REF_HIP(hipMemcpy(d_normals, normals, n_vertices * 16, hipMemcpyHostToDevice));
	REF_HIP(hipMemset(d_image, 0xFF, image_bytes));
	REF_HIP(hipEventCreate(&start));
	REF_HIP(hipEventCreate(&stop));
	{
		/* argument order of reference src/opencl_host.cc:139-144 */
		void *args[6] = { &d_faces, &d_nodes, &d_aabbs, &d_vertices, &d_normals, &d_image };
		REF_HIP(hipModuleLaunchKernel(kernel, width / block_x, height / block_y, 1, block_x, block_y, 1, 0, nullptr, args,
		                              nullptr));
		REF_HIP(hipDeviceSynchronize());
		if (repeats > 0) {
			REF_HIP(hipEventRecord(start, nullptr));
			for (int k = 0; k < repeats; ++k)
				REF_HIP(hipModuleLaunchKernel(kernel, width / block_x, height / block_y, 1, block_x, block_y, 1, 0, nullptr,
				                              args, nullptr));
			REF_HIP(hipEventRecord(stop, nullptr));
			REF_HIP(hipEventSynchronize(stop));
			float ms = 0;
			REF_HIP(hipEventElapsedTime(&ms, start, stop));
			if (kernel_ms)
				*kernel_ms = ms / (float) repeats;
		}
	}
	REF_HIP(hipMemcpy(image, d_image, image_bytes, hipMemcpyDeviceToHost));
	rc = 0;
done:
	if (start)
		(void) hipEventDestroy(start);
	if (stop)
		(void) hipEventDestroy(stop);
	(void) hipFree(d_faces);
	(void) hipFree(d_nodes);
	(void) hipFree(d_aabbs);
	(void) hipFree(d_vertices);
	(void) hipFree(d_normals);
	(void) hipFree(d_image);
	if (module)
		(void) hipModuleUnload(module);
	return rc;
}
