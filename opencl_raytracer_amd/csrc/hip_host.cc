#include "hip_host.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <stdexcept>
#include <string>

#include "band_gather.h"
#include "cli_support.h"
#include "device_renderer.h"
#include "frame_ring.h"
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

namespace {

// The render host that HipHost::warmUp(rt, device) created ahead of time, waiting to be adopted.
// (A plain pointer on purpose: a host that is never adopted -- the run ended early -- is left to the process's exit
// rather than destroyed by a static destructor after the HIP runtime may have shut down.)
std::mutex prepared_mutex;
ocrt::DeviceRenderer *prepared = nullptr;
int prepared_device = -1;

bool same_options(const RayTracer::Options &a, const RayTracer::Options &b) {
	return a.width == b.width && a.height == b.height && a.focalLength == b.focalLength && a.nSuperSamples == b.nSuperSamples &&
	       a.enableShading == b.enableShading && a.enableAO == b.enableAO && a.aoMaxDistance == b.aoMaxDistance &&
	       a.aoNumSamples == b.aoNumSamples && a.aoMethod == b.aoMethod && a.aoAlphaMin == b.aoAlphaMin &&
	       a.aoAlphaMax == b.aoAlphaMax && a.bvhMethod == b.bvhMethod;
}

}  // namespace

HipHost::HipHost(const RayTracer &rt_, int device) : HipHost(rt_, device, 0, 1) {}

HipHost::HipHost(const RayTracer &rt_, int device, unsigned int rank, unsigned int nranks) : rt(rt_) {
	if (rank == 0 && nranks == 1) {
		std::lock_guard<std::mutex> lock(prepared_mutex);
		if (prepared && prepared_device == device && same_options(prepared->rayTracer().options, rt.options))
			impl.reset(prepared);
		else
			delete prepared;
		prepared = nullptr;
	}
	// std::runtime_error("No device found") propagates, as in the reference.
	if (!impl)
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

void HipHost::upload(const ocrt::PackedScene &packed) {
	try {
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
HipHostRing::HipHostRing(const RayTracer &rt_, unsigned int hosts, int device) : rt(rt_), last_host(0) {
	// std::runtime_error("No device found") propagates, as in the reference; bad arguments end like a device error
	try {
		ring.reset(new ocrt::FrameRing(rt.options, device, 0, 1, hosts));
	} catch (const std::invalid_argument &e) {
		die(e.what());
	} catch (const ocrt::DeviceError &e) {
		die(e.what());
	}
	std::cout << Color::WHITE << "Using Device \"" << ring->host(0).deviceName() << "\", " << ring->size()
	          << (ring->size() == 1 ? " render host." : " render hosts taking frames in turn.") << Color::RESET << std::endl
	          << std::endl;
}

HipHostRing::~HipHostRing() {}

unsigned int HipHostRing::size() const { return ring->size(); }

void HipHostRing::upload(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes,
                         const std::vector<Vec3f> &aabbs, const std::vector<Vec3f> &vertices,
                         const std::vector<Vec3f> &vnormals) {
	try {
		const ocrt::PackedScene packed = ocrt::pack_scene(faces, nodes, aabbs, vertices, vnormals);  // once; every host gets it
		const size_t bytes = ring->upload(packed);
		std::cout << "Requested " << bytes / 1024 << " kB of memory." << std::endl;
	} catch (const std::exception &e) {
		die(e.what());
	}
}

void HipHostRing::upload(const ocrt::PackedScene &packed) {
	try {
		const size_t bytes = ring->upload(packed);
		std::cout << "Requested " << bytes / 1024 << " kB of memory." << std::endl;
	} catch (const std::exception &e) {
		die(e.what());
	}
}

bool HipHostRing::operator()() {
	try {
		ring->drain();
		ring->submit();
		last_host = (unsigned int) (ring->collect().frame % ring->size());
	} catch (const std::exception &e) {
		die(e.what());
	}
	return true;
}

bool HipHostRing::frames(unsigned int count) {
	try {
		for (unsigned int k = 0; k < count; ++k)
			ring->step();
		ring->drain();
		if (ring->submitted())
			last_host = (unsigned int) ((ring->submitted() - 1) % ring->size());
	} catch (const std::exception &e) {
		die(e.what());
	}
	return true;
}

void HipHostRing::download(float *image) {
	try {
		ring->host(last_host).downloadFloat(image);
	} catch (const std::exception &e) {
		die(e.what());
	}
}

void HipHostRing::downloadResized(unsigned char *image) {
	try {
		ring->downloadLast(image);
	} catch (const std::exception &e) {
		die(e.what());
	}
}

float HipHostRing::lastKernelMs() const { return ring->host(last_host).lastKernelMs(); }

ocrt::RenderStats HipHostRing::lastStats() {
	try {
		return ring->host(last_host).stats();
	} catch (const std::exception &e) {
		die(e.what());
	}
}

// ---------------------------------------------------------------------------------------------------------------
HipHostGroup::HipHostGroup(const RayTracer &rt_, unsigned int devices, int first, const char *gather_choice)
	: rt(rt_), staging(nullptr), assembled(nullptr), staging_bytes(0) {
	try {
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
		std::vector<int> device_of;
		for (unsigned int r = 0; r < devices; ++r) {
			const int dev = share ? (first + (int) r) % visible : first + (int) r;
			device_of.push_back(dev);
			hosts.emplace_back(new ocrt::DeviceRenderer(rt.options, dev, r, devices));
			std::cout << Color::WHITE << "Rank " << r << " of " << devices << ": device " << dev << " \"" << hosts.back()->deviceName()
			          << "\", " << hosts.back()->localRows() << " rows." << Color::RESET << std::endl;
		}
		// The exchange step: RCCL where it can be had, else peer copies.
		const std::string want = gather_choice ? gather_choice : "auto";
		if (want != "auto" && want != "rccl" && want != "peer")
			throw std::invalid_argument("the gather is one of rccl, peer, auto");
		std::string why_not;
		if (want == "peer") {
			why_not = "asked for";
		} else {
			try {
				gather.reset(new ocrt::GroupGather(rt.options, device_of));
			} catch (const ocrt::DeviceError &e) {
				why_not = e.what();
			}
		}
		if (gather) {
			std::cout << Color::WHITE << "Bands are gathered on device " << device_of[0] << " over RCCL." << Color::RESET << std::endl;
		} else {
			if (want == "rccl")
				throw ocrt::DeviceError(why_not);
			std::cout << Color::WHITE << "Bands are gathered on device " << device_of[0] << " by peer copies (" << why_not << ")."
			          << Color::RESET << std::endl;
			const ocrt::BandPlan plan(rt.options, devices);
			staging_bytes = plan.stride() * devices;
			if (hipSetDevice(device_of[0]) != hipSuccess || hipMalloc(&staging, staging_bytes ? staging_bytes : 1) != hipSuccess ||
			    hipMalloc(&assembled, (size_t) plan.width * plan.height) != hipSuccess)
				throw ocrt::DeviceError("cannot allocate the band staging buffers");
			// The copies are enqueued on the SOURCE device's stream and write into the first device's staging buffer:
			// the source needs access to the first device (and the first device to its peers, for symmetry).
			for (size_t r = 1; r < hosts.size(); ++r) {
				if (device_of[r] == device_of[0])
					continue;
				int can = 0;
				if (hipSetDevice(device_of[r]) != hipSuccess)
					throw ocrt::DeviceError("hipSetDevice failed while enabling peer access");
				if (hipDeviceCanAccessPeer(&can, device_of[r], device_of[0]) == hipSuccess && can)
					(void) hipDeviceEnablePeerAccess(device_of[0], 0);
				(void) hipGetLastError();  // (peer access already enabled is not an error)
				if (hipSetDevice(device_of[0]) != hipSuccess)
					throw ocrt::DeviceError("hipSetDevice failed while enabling peer access");
				if (hipDeviceCanAccessPeer(&can, device_of[0], device_of[r]) == hipSuccess && can)
					(void) hipDeviceEnablePeerAccess(device_of[r], 0);
				(void) hipGetLastError();
			}
		}
		std::cout << std::endl;
	} catch (const std::runtime_error &e) {
		if (std::string(e.what()) == "No device found")
			throw;  // as in the reference (src/opencl_host.cc:30-31)
		die(e.what());
	} catch (const std::exception &e) {
		die(e.what());
	}
}

HipHostGroup::~HipHostGroup() {
	gather.reset();
	if (!hosts.empty() && hipSetDevice(hosts[0]->deviceIndex()) == hipSuccess) {
		if (staging)
			(void) hipFree(staging);
		if (assembled)
			(void) hipFree(assembled);
	}
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

void HipHostGroup::upload(const ocrt::PackedScene &packed) {
	try {
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
		const size_t bytes = (size_t) rt.options.width * rt.options.height;
		for (auto &h : hosts)
			h->enqueueResize();
		const void *result = nullptr;
		if (gather) {
			// one grouped send / receive over RCCL, then the row scatter, all on the devices' own streams
			std::vector<const void *> bands;
			std::vector<void *> streams;
			for (auto &h : hosts) {
				bands.push_back(h->deviceBands());
				streams.push_back(h->streamHandle());
			}
			gather->enqueue(bands, streams);
			result = gather->image();
		} else {
			// every other device's bands -> its stride of the staging buffer on the first device, on the source's stream
			const ocrt::BandPlan plan(rt.options, (unsigned int) hosts.size());
			for (size_t r = 1; r < hosts.size(); ++r) {
				auto &h = hosts[r];
				if (!plan.bytesOf((unsigned int) r))
					continue;
				if (hipSetDevice(h->deviceIndex()) != hipSuccess)
					throw ocrt::DeviceError("hipSetDevice failed before a band copy");
				if (hipMemcpyPeerAsync((char *) staging + r * plan.stride(), hosts[0]->deviceIndex(), h->deviceBands(), h->deviceIndex(),
				                       plan.bytesOf((unsigned int) r), (hipStream_t) h->streamHandle()) != hipSuccess)
					throw ocrt::DeviceError("hipMemcpyPeerAsync of a device's image bands failed");
			}
			for (auto &h : hosts)
				h->synchronize();
			if (hipSetDevice(hosts[0]->deviceIndex()) != hipSuccess)
				throw ocrt::DeviceError("hipSetDevice failed before the row assembly");
			ocrt::launch_assemble_rows(plan, hosts[0]->deviceBands(), staging, assembled, hosts[0]->streamHandle());
			result = assembled;
		}
		for (auto &h : hosts)
			h->synchronize();
		if (hipSetDevice(hosts[0]->deviceIndex()) != hipSuccess ||
		    hipMemcpy(image, result, bytes, hipMemcpyDeviceToHost) != hipSuccess)
			throw ocrt::DeviceError("reading the assembled image failed");
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

void HipHost::warmUp(int device) { ocrt::warm_up_device(device); }

void HipHost::warmUp(const RayTracer &rt, int device) {
	ocrt::warm_up_device(device);
	try {  // (whatever goes wrong here goes wrong again, and is reported, when the host proper is constructed)
		ocrt::DeviceRenderer *host = new ocrt::DeviceRenderer(rt.options, device, 0, 1);
		std::lock_guard<std::mutex> lock(prepared_mutex);
		delete prepared;
		prepared = host;
		prepared_device = device;
	} catch (const std::exception &) {
	}
}

void HipHost::reserveScene(const RayTracer &rt, int device, size_t triangles) {
	if (device < 0) {
		const char *env = std::getenv("OCRT_DEVICE");
		device = env ? std::atoi(env) : 0;
	}
	size_t directions = 0;
	if (rt.options.enableAO && rt.options.aoNumSamples > 0)
		directions = rt.options.aoMethod == RayTracer::AmbientOcclusionMethod::UNIFORM
		                 ? ocrt::uniform_ao_table(rt.options.aoNumSamples, rt.options.aoAlphaMin, rt.options.aoAlphaMax).size() / 4
		                 : rt.options.aoNumSamples + 2;
	ocrt::DeviceScene::reserve(device, ocrt::DeviceScene::bytesFor(triangles, directions));
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
