// Probe (GPU box): the start-up steps behind hipInit on separate threads -- do they overlap?
// hipcc --offload-arch=gfx950 -O2 -pthread -o init_probe_parallel tools/microbench/init_probe_parallel.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(int *p) { if (p) *p = 1; }
int main() {
	using clock = std::chrono::steady_clock;
	const auto t0 = clock::now();
	auto since = [&] { return std::chrono::duration<double, std::milli>(clock::now() - t0).count(); };
	(void) hipInit(0);
	(void) hipSetDevice(0);
	const double after_init = since();
	double done[3] = { 0, 0, 0 };
	std::thread a([&] {
		(void) hipSetDevice(0);
		void *p = nullptr; (void) hipMalloc(&p, 1 << 20);
		std::vector<unsigned char> host(1 << 16);
		(void) hipMemcpy(p, host.data(), host.size(), hipMemcpyHostToDevice);
		(void) hipMemcpy(host.data(), p, host.size(), hipMemcpyDeviceToHost);
		done[0] = since();
	});
	std::thread b([&] {
		(void) hipSetDevice(0);
		hipStream_t s; (void) hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0);
		done[1] = since();
	});
	std::thread c([&] {
		(void) hipSetDevice(0);
		hipFuncAttributes attr; (void) hipFuncGetAttributes(&attr, (const void *) tiny);
		done[2] = since();
	});
	a.join(); b.join(); c.join();
	std::printf("hipInit %.2f ms; then in parallel: first copies done at +%.2f, stream at +%.2f, code object at +%.2f; all at %.2f ms\n", after_init,
	            done[0] - after_init, done[1] - after_init, done[2] - after_init, since());
	return 0;
}
