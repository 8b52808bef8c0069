#include "hip_host.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <iostream>

#include "cli_support.h"
#include "device_renderer.h"
#include "scene_pack.h"

using ocrt::cli::Color;
using ocrt::cli::Info;

namespace {

// The reference's check(): report and leave (reference include/opencl_host.h:21-26).
[[noreturn]] void die(const std::string &message) {
	std::cerr << "HIP error: " << message << std::endl;
	std::exit(EXIT_FAILURE);
}

}  // namespace

HipHost::HipHost(const RayTracer &rt_, int device) : HipHost(rt_, device, 0, 1) {}

HipHost::HipHost(const RayTracer &rt_, int device, unsigned int rank, unsigned int nranks) : rt(rt_) {
	// std::runtime_error("No device found") propagates, as in the reference.
	impl.reset(new ocrt::DeviceRenderer(rt.options, device, rank, nranks));
	std::cout << Color::WHITE << "Using Device \"" << impl->deviceName() << "\"." << Color::RESET << std::endl
	          << std::endl;
}

HipHost::~HipHost() {}

void HipHost::upload(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes,
                     const std::vector<Vec3f> &aabbs, const std::vector<Vec3f> &vertices,
                     const std::vector<Vec3f> &vnormals) {
	try {
		const ocrt::PackedScene packed = ocrt::pack_scene(faces, nodes, aabbs, vertices, vnormals);
		const size_t bytes = impl->upload(packed);
		std::cout << "Requested " << bytes / 1024 << " kB of memory." << std::endl;
	} catch (const std::exception &e) {
		die(e.what());
	}
}

bool HipHost::operator()() {
	try {
		impl->enqueueRender();
		impl->synchronize();
	} catch (const std::exception &e) {
		die(e.what());
	}
	return true;
}

void HipHost::download(float *image) {
	try {
		impl->downloadFloat(image);
	} catch (const std::exception &e) {
		die(e.what());
	}
}

void HipHost::downloadResized(unsigned char *image) {
	try {
		impl->downloadResizedFull(image);
	} catch (const std::exception &e) {
		die(e.what());
	}
}

float HipHost::lastKernelMs() const { return impl->lastKernelMs(); }

ocrt::RenderStats HipHost::lastStats() {
	try {
		return impl->stats();
	} catch (const std::exception &e) {
		die(e.what());
	}
}

void HipHost::printInfo() {
	Info info;
	info.setTitle("Hardware information");
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess)
		count = 0;
	int runtime = 0, driver = 0;
	(void) hipRuntimeGetVersion(&runtime);
	(void) hipDriverGetVersion(&driver);
	Info platform;
	platform.setTitle("Platform #0");
	platform.add("Name", std::string("AMD HIP"));
	platform.add("Runtime version", runtime);
	platform.add("Driver version", driver);
	platform.add("Devices", count);
	for (int d = 0; d < count; ++d) {
		hipDeviceProp_t p;
		if (hipGetDeviceProperties(&p, d) != hipSuccess)
			continue;
		Info dev;
		dev.setTitle("Device #" + std::to_string(d));
		dev.add("Name", std::string(p.name));
		dev.add("Architecture", std::string(p.gcnArchName));
		dev.add("Type", std::string("GPU"));
		dev.add("Max compute units", p.multiProcessorCount);
		dev.add("Wavefront size", p.warpSize);
		dev.add("Max clock (MHz)", p.clockRate / 1000);
		dev.add("Local memory size (B)", p.sharedMemPerBlock);
		dev.add("L2 cache size (B)", p.l2CacheSize);
		dev.add("Global memory (MB)", p.totalGlobalMem / (1024 * 1024));
		dev.add("Max work item sizes", std::to_string(p.maxThreadsDim[0]) + ", " + std::to_string(p.maxThreadsDim[1]) +
		                                   ", " + std::to_string(p.maxThreadsDim[2]));
		platform.add(dev);
	}
	info.add(platform);
	std::cout << std::endl;
	std::cout << info.str();
}
