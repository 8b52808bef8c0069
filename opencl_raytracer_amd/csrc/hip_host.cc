#include "hip_host.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>

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

// ---------------------------------------------------------------------------------------------------------------
HipHostGroup::HipHostGroup(const RayTracer &rt_, unsigned int devices, int first) : rt(rt_), staging(nullptr), staging_bytes(0) {
	if (devices == 0)
		throw std::invalid_argument("--gpus needs at least one device");
	const int visible = ocrt::visible_device_count();
	if (visible <= 0)
		throw std::runtime_error("No device found");
	if (first < 0) {
		const char *env = std::getenv("OCRT_DEVICE");
		first = env ? std::atoi(env) : 0;
	}
	const bool share = std::getenv("OCRT_SHARE_DEVICES") != nullptr;
	if (!share && first + (int) devices > visible)
		throw std::invalid_argument("more ranks than visible HIP devices (one rank per GPU)");
	for (unsigned int r = 0; r < devices; ++r) {
		const int dev = share ? (first + (int) r) % visible : first + (int) r;
		hosts.emplace_back(new ocrt::DeviceRenderer(rt.options, dev, r, devices));
		std::cout << Color::WHITE << "Rank " << r << " of " << devices << ": device " << dev << " \"" << hosts.back()->deviceName()
		          << "\", " << hosts.back()->localRows() << " rows." << Color::RESET << std::endl;
	}
	std::cout << std::endl;
	try {
		for (auto &h : hosts)
			staging_bytes += (size_t) h->localRows() * rt.options.width;
		if (hipSetDevice(hosts[0]->deviceIndex()) != hipSuccess || hipMalloc(&staging, staging_bytes ? staging_bytes : 1) != hipSuccess)
			throw ocrt::DeviceError("cannot allocate the band staging buffer");
		// let the first device read its peers' memory directly where the topology allows it (else the copies are staged)
		for (size_t r = 1; r < hosts.size(); ++r)
			if (hosts[r]->deviceIndex() != hosts[0]->deviceIndex()) {
				int can = 0;
				if (hipDeviceCanAccessPeer(&can, hosts[0]->deviceIndex(), hosts[r]->deviceIndex()) == hipSuccess && can)
					(void) hipDeviceEnablePeerAccess(hosts[r]->deviceIndex(), 0);
			}
		(void) hipGetLastError();  // (peer access already enabled is not an error)
	} catch (const std::exception &e) {
		die(e.what());
	}
}

HipHostGroup::~HipHostGroup() {
	if (staging && !hosts.empty() && hipSetDevice(hosts[0]->deviceIndex()) == hipSuccess)
		(void) hipFree(staging);
}

void HipHostGroup::upload(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes,
                          const std::vector<Vec3f> &aabbs, const std::vector<Vec3f> &vertices,
                          const std::vector<Vec3f> &vnormals) {
	try {
		const ocrt::PackedScene packed = ocrt::pack_scene(faces, nodes, aabbs, vertices, vnormals);  // once; replicated
		size_t bytes = 0;
		for (auto &h : hosts)
			bytes += h->upload(packed);
		std::cout << "Requested " << bytes / 1024 << " kB of memory on " << hosts.size() << " devices." << std::endl;
	} catch (const std::exception &e) {
		die(e.what());
	}
}

bool HipHostGroup::operator()() {
	try {
		for (auto &h : hosts)  // (launches are asynchronous: the devices work at the same time)
			h->enqueueRender();
		for (auto &h : hosts)
			h->synchronize();
	} catch (const std::exception &e) {
		die(e.what());
	}
	return true;
}

void HipHostGroup::downloadResized(unsigned char *image) {
	try {
		const size_t width = rt.options.width;
		for (auto &h : hosts)
			h->enqueueResize();
		// every device's bands -> the staging buffer on the first device, on the source device's own stream
		size_t offset = 0;
		for (auto &h : hosts) {
			const size_t bytes = (size_t) h->localRows() * width;
			if (bytes && hipSetDevice(h->deviceIndex()) == hipSuccess &&
			    hipMemcpyPeerAsync((char *) staging + offset, hosts[0]->deviceIndex(), h->deviceBands(), h->deviceIndex(),
			                                bytes, (hipStream_t) h->streamHandle()) != hipSuccess)
				throw ocrt::DeviceError("hipMemcpyPeerAsync of a device's image bands failed");
			offset += bytes;
		}
		for (auto &h : hosts)
			h->synchronize();
		std::vector<unsigned char> stacked(staging_bytes);
		if (hipSetDevice(hosts[0]->deviceIndex()) != hipSuccess ||
		    hipMemcpy(stacked.data(), staging, staging_bytes, hipMemcpyDeviceToHost) != hipSuccess)
			throw ocrt::DeviceError("reading the gathered bands failed");
		// rows to their place (a device's last band may run past the image: padding rows are dropped)
		offset = 0;
		for (auto &h : hosts) {
			for (uint32_t j = 0; j < h->localRows(); ++j) {
				const uint32_t y = h->globalRowOf(j);
				if (y < rt.options.height)
					std::memcpy(image + (size_t) y * width, stacked.data() + offset + (size_t) j * width, width);
			}
			offset += (size_t) h->localRows() * width;
		}
	} catch (const std::exception &e) {
		die(e.what());
	}
}

float HipHostGroup::lastKernelMs() const {
	float slowest = 0.0f;
	for (auto &h : hosts)
		slowest = std::max(slowest, h->lastKernelMs());
	return slowest;
}

ocrt::RenderStats HipHostGroup::lastStats() {
	ocrt::RenderStats total{};
	try {
		for (auto &h : hosts) {
			const ocrt::RenderStats s = h->stats();
			total.primary_rays += s.primary_rays;
			total.primary_hits += s.primary_hits;
			total.ao_rays += s.ao_rays;
			total.ao_occluded += s.ao_occluded;
		}
	} catch (const std::exception &e) {
		die(e.what());
	}
	return total;
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
