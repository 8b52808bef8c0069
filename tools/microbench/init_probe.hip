// Probe (GPU box): what the HIP runtime's start-up costs, step by step, in a fresh process -- the floor under a one-shot
// `render` (one-off costs).  hipcc --offload-arch=gfx950 -O2 -o init_probe tools/microbench/init_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(int *p) { if (p) *p = 1; }
int main() {
	using clock = std::chrono::steady_clock;
	auto t = clock::now();
	auto lap = [&](const char *what) {
		const auto now = clock::now();
		std::printf("%-48s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
		t = now;
	};
	(void) hipInit(0); lap("hipInit");
	int n = 0; (void) hipGetDeviceCount(&n); lap("hipGetDeviceCount");
	(void) hipSetDevice(0); lap("hipSetDevice");
	(void) hipFree(nullptr); lap("hipFree(nullptr): the context");
	hipFuncAttributes a; (void) hipFuncGetAttributes(&a, (const void *) tiny); lap("hipFuncGetAttributes: the code object");
	void *p = nullptr; (void) hipMalloc(&p, 1 << 20); lap("first hipMalloc (1 MB)");
	std::vector<unsigned char> host(1 << 16);
	(void) hipMemcpy(p, host.data(), host.size(), hipMemcpyHostToDevice); lap("first copy to the device (64 KB)");
	(void) hipMemcpy(host.data(), p, host.size(), hipMemcpyDeviceToHost); lap("first copy from the device");
	hipStream_t s; (void) hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0); lap("hipStreamCreateWithPriority");
	hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, (int *) p); (void) hipStreamSynchronize(s); lap("first kernel launch + synchronize");
	void *big = nullptr; (void) hipMalloc(&big, 64 << 20); lap("hipMalloc 64 MB");
	hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, (int *) p); (void) hipStreamSynchronize(s); lap("second launch + synchronize");
	return 0;
}
