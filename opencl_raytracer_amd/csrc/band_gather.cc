#include "band_gather.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the entry points are looked up at run time, the library is not linked

#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <sstream>
#include <string>

#include "device_renderer.h"
#include "scene_pack.h"

namespace ocrt {

namespace {

void hip_check(hipError_t err, const char *what) {
	if (err != hipSuccess) {
		std::ostringstream ss;
		ss << "HIP error: " << hipGetErrorName(err) << " (" << hipGetErrorString(err) << ") in " << what;
		throw DeviceError(ss.str());
	}
}
#define OCRT_HIP(call) hip_check((call), #call)

// The RCCL entry points this file uses, resolved once.
struct Rccl {
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;             // optional
	ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;  // optional
	ncclResult_t (*GetVersion)(int *) = nullptr;                  // optional
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	std::string why;  // non-empty: not usable
};

const Rccl &rccl() {
	static Rccl api;
	static std::once_flag once;
	std::call_once(once, [] {
		// RTLD_NOLOAD first: the copy the process already holds (torch ships its own librccl.so with the same soname)
		void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
		if (!lib)
			lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
		if (!lib)
			lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
		if (!lib) {
			api.why = std::string("cannot open librccl.so.1: ") + dlerror();
			return;
		}
		auto sym = [&](const char *name) {
			void *p = dlsym(lib, name);
			if (!p && api.why.empty())
				api.why = std::string("librccl.so.1 lacks ") + name;
			return p;
		};
		api.GetUniqueId = (decltype(api.GetUniqueId)) sym("ncclGetUniqueId");
		api.CommInitRank = (decltype(api.CommInitRank)) sym("ncclCommInitRank");
		api.CommInitAll = (decltype(api.CommInitAll)) sym("ncclCommInitAll");
		api.CommDestroy = (decltype(api.CommDestroy)) sym("ncclCommDestroy");
		api.GroupStart = (decltype(api.GroupStart)) sym("ncclGroupStart");
		api.GroupEnd = (decltype(api.GroupEnd)) sym("ncclGroupEnd");
		api.Send = (decltype(api.Send)) sym("ncclSend");
		api.Recv = (decltype(api.Recv)) sym("ncclRecv");
		api.GetErrorString = (decltype(api.GetErrorString)) sym("ncclGetErrorString");
		// (not every build has these: looked up without complaint)
		api.CommAbort = (decltype(api.CommAbort)) dlsym(lib, "ncclCommAbort");
		api.CommCount = (decltype(api.CommCount)) dlsym(lib, "ncclCommCount");
		api.GetVersion = (decltype(api.GetVersion)) dlsym(lib, "ncclGetVersion");
	});
	return api;
}

const Rccl &rccl_or_throw() {
	const Rccl &api = rccl();
	if (!api.why.empty())
		throw DeviceError("RCCL is not usable: " + api.why);
	return api;
}

void nccl_check(ncclResult_t r, const char *what) {
	if (r != ncclSuccess) {
		const Rccl &api = rccl();
		std::ostringstream ss;
		ss << "RCCL error: " << (api.GetErrorString ? api.GetErrorString(r) : "?") << " in " << what;
		throw DeviceError(ss.str());
	}
}
#define OCRT_NCCL(call) nccl_check((call), #call)

// Calls made between ncclGroupStart and ncclGroupEnd: the first failure is remembered, the group is ALWAYS closed (an
// open group would swallow every later RCCL call of the process, torch.distributed's included), then it is thrown.
struct NcclGroup {
	const Rccl &api;
	ncclResult_t first = ncclSuccess;
	const char *where = nullptr;
	explicit NcclGroup(const Rccl &a) : api(a) { nccl_check(api.GroupStart(), "ncclGroupStart"); }
	void note(ncclResult_t r, const char *what) {
		if (r != ncclSuccess && first == ncclSuccess) {
			first = r;
			where = what;
		}
	}
	void end() {
		const ncclResult_t closed = api.GroupEnd();
		if (first != ncclSuccess)
			nccl_check(first, where);
		nccl_check(closed, "ncclGroupEnd");
	}
};
#define OCRT_IN_GROUP(group, expression) (group).note((expression), #expression)

double seconds_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// One workgroup per image row: the row's owner and its place in the owner's band buffer follow from the partition
// arithmetic (band b = y / rows_per_band belongs to rank b % nranks; ocrt::Partition).
__global__ __launch_bounds__(256) void assemble_rows_kernel(const unsigned char *__restrict__ own,
                                                            const unsigned char *__restrict__ stacked,
                                                            unsigned char *__restrict__ image, uint32_t width,
                                                            uint32_t nranks, uint32_t rows_per_band, size_t stride) {
	const uint32_t y = blockIdx.x;
	const uint32_t band = y / rows_per_band;
	const uint32_t owner = band % nranks;
	const uint32_t local = (band / nranks) * rows_per_band + (y - band * rows_per_band);
	const unsigned char *src = (owner == 0u ? own : stacked + (size_t) owner * stride) + (size_t) local * width;
	unsigned char *dst = image + (size_t) y * width;
	if ((((uintptr_t) src | (uintptr_t) dst | width) & 15u) == 0u) {  // (wave-uniform) 16 bytes per lane
		const uint4 *s4 = (const uint4 *) src;
		uint4 *d4 = (uint4 *) dst;
		for (uint32_t x = threadIdx.x; x < width / 16u; x += blockDim.x)
			d4[x] = s4[x];
		return;
	}
	for (uint32_t x = threadIdx.x; x < width; x += blockDim.x)
		dst[x] = src[x];
}

}  // namespace

BandPlan::BandPlan(const RayTracer::Options &options, unsigned int nranks_)
	: width(options.width), height(options.height), nranks(nranks_), rows_per_band(0), max_rows(0) {
	const uint32_t n = RayTracer::gridSize(options.nSuperSamples);
	if (n == 0 || nranks == 0 || width == 0 || height == 0)
		throw std::invalid_argument("band plan: image size, supersample count and rank count must be positive");
	const uint32_t band_tile_rows = band_tile_rows_for(n);
	rows_per_band = band_tile_rows * TILE_H / n;
	for (unsigned int r = 0; r < nranks; ++r) {
		const Partition part{ r, nranks, band_tile_rows };
		local_rows.push_back(local_tile_rows_for(height * n, part) * TILE_H / n);
		max_rows = local_rows.back() > max_rows ? local_rows.back() : max_rows;
	}
}

void launch_assemble_rows(const BandPlan &plan, const void *own, const void *stacked, void *image, void *stream) {
	hipLaunchKernelGGL(assemble_rows_kernel, dim3(plan.height), dim3(256), 0, (hipStream_t) stream,
	                   (const unsigned char *) own, (const unsigned char *) stacked, (unsigned char *) image, plan.width,
	                   plan.nranks, plan.rows_per_band, plan.stride());
	OCRT_HIP(hipGetLastError());
}

bool rccl_available() { return rccl().why.empty(); }

void rccl_unique_id(void *out128) {
	const Rccl &api = rccl_or_throw();
	ncclUniqueId id;
	OCRT_NCCL(api.GetUniqueId(&id));
	static_assert(sizeof id == RCCL_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
	std::memcpy(out128, &id, sizeof id);
}

// ---------------------------------------------------------------------------------------------------------------
BandGather::BandGather(const RayTracer::Options &options, unsigned int rank_, unsigned int nranks_, int device_,
                       const void *unique_id, unsigned int slots)
	: layout(options, nranks_), rank(rank_), nranks(nranks_), device(device_), comm(nullptr), stream(nullptr) {
	if (rank >= nranks || slots == 0 || !unique_id)
		throw std::invalid_argument("band gather: rank must be < nranks, slots positive, the unique id given");
	const Rccl &api = rccl_or_throw();
	OCRT_HIP(hipSetDevice(device));
	ncclUniqueId id;
	std::memcpy(&id, unique_id, sizeof id);
	ncclComm_t c = nullptr;
	OCRT_NCCL(api.CommInitRank(&c, (int) nranks, id, (int) rank));
	comm = c;
	try {
		allocate(slots);
	} catch (...) {
		release();
		throw;
	}
}

void BandGather::allocate(unsigned int slots) {
	// Its own stream, HIGHEST priority class: the transfer and the row scatter are a few microseconds of work that a
	// frame's consumer waits for, while the ring's persistent ambient-occlusion passes would keep a low-priority
	// kernel waiting for wave slots until one of them ends (measured: +4 % per frame with the lowest class).
	int least = 0, greatest = 0;
	OCRT_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
	hipStream_t s;
	OCRT_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
	stream = s;
	for (unsigned int k = 0; k < slots; ++k) {
		hipEvent_t e;
		OCRT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		done.push_back(e);
		void *recv = nullptr, *img = nullptr;
		if (rank == 0) {
			OCRT_HIP(hipMalloc(&recv, layout.stride() * nranks));
			OCRT_HIP(hipMalloc(&img, (size_t) layout.width * layout.height));
		}
		stacked.push_back(recv);
		final_image.push_back(img);
	}
}

BandGather::~BandGather() { release(); }

void BandGather::release() {
	if (hipSetDevice(device) != hipSuccess)
		return;
	if (stream && !broken)
		(void) hipStreamSynchronize((hipStream_t) stream);
	if (comm) {
		if (broken && rccl().CommAbort)
			(void) rccl().CommAbort((ncclComm_t) comm);  // (pending operations can never complete: do not wait for them)
		else if (!broken && rccl().CommDestroy)
			(void) rccl().CommDestroy((ncclComm_t) comm);
	}
	comm = nullptr;
	for (void *e : done)
		(void) hipEventDestroy((hipEvent_t) e);
	done.clear();
	for (void *p : stacked)
		if (p)
			(void) hipFree(p);
	stacked.clear();
	for (void *p : final_image)
		if (p)
			(void) hipFree(p);
	final_image.clear();
	if (stream)
		(void) hipStreamDestroy((hipStream_t) stream);
	stream = nullptr;
}

void BandGather::enqueue(unsigned int slot, const void *device_bands) {
	if (slot >= done.size() || !device_bands)
		throw std::invalid_argument("band gather: bad slot or null band buffer");
	const Rccl &api = rccl_or_throw();
	OCRT_HIP(hipSetDevice(device));
	hipStream_t s = (hipStream_t) stream;
	if (nranks > 1) {
		NcclGroup group(api);
		if (rank == 0) {
			for (unsigned int r = 1; r < nranks; ++r)
				if (layout.bytesOf(r))
					OCRT_IN_GROUP(group, api.Recv((char *) stacked[slot] + (size_t) r * layout.stride(), layout.bytesOf(r), ncclUint8,
					                              (int) r, (ncclComm_t) comm, s));
		} else if (layout.bytesOf(rank)) {
			OCRT_IN_GROUP(group, api.Send(device_bands, layout.bytesOf(rank), ncclUint8, 0, (ncclComm_t) comm, s));
		}
		group.end();
	}
	if (rank == 0)
		launch_assemble_rows(layout, device_bands, stacked[slot], final_image[slot], stream);
	OCRT_HIP(hipEventRecord((hipEvent_t) done[slot], s));
}

void BandGather::wait(unsigned int slot) {
	if (slot >= done.size())
		throw std::invalid_argument("band gather: bad slot");
	OCRT_HIP(hipSetDevice(device));
	// (mostly long done -- the slot comes round again a whole turn of the ring later --: ask before sleeping on it)
	// A BOUNDED wait: the gather completes only if every rank posts its half of it.  A rank that died, or that submitted
	// a different number of frames, would leave the others asleep in here for ever -- inside rt_ring_run, where nobody can
	// see why.  So the event is polled against a deadline, and missing it is an error (DeviceError -> RT_E_DEVICE: the rank
	// ends with a message and a non-zero status instead of hanging the job); the communicator is then marked broken and
	// aborted rather than destroyed (ncclCommDestroy would wait for the very operation that cannot finish).
	if (broken)
		throw DeviceError("band gather: the exchange step is broken (an earlier wait ran into its deadline)");
	const double begin = seconds_now();
	for (unsigned int polls = 0;; ++polls) {
		const hipError_t state = hipEventQuery((hipEvent_t) done[slot]);
		if (state == hipSuccess)
			return;
		if (state != hipErrorNotReady)
			OCRT_HIP(state);
		const double waited = seconds_now() - begin;
		if (waited > timeout_s) {
			broken = true;
			std::ostringstream ss;
			ss << "band gather: rank " << rank << " of " << nranks << " waited " << timeout_s << " s for the exchange step of slot " << slot
			   << " -- a peer rank is gone or posted a different number of frames";
			throw DeviceError(ss.str());
		}
		if (polls < 2000)
			std::this_thread::yield();
		else
			std::this_thread::sleep_for(std::chrono::microseconds(waited < 0.01 ? 20 : 500));
	}
}

void BandGather::describe(int *comm_ranks, int *rccl_version) const {
	const Rccl &api = rccl();
	int count = -1, version = -1;
	if (comm && api.CommCount)
		(void) api.CommCount((ncclComm_t) comm, &count);
	if (api.GetVersion)
		(void) api.GetVersion(&version);
	if (comm_ranks)
		*comm_ranks = count;
	if (rccl_version)
		*rccl_version = version;
}

const void *BandGather::image(unsigned int slot) const { return slot < final_image.size() ? final_image[slot] : nullptr; }

void BandGather::selfTest() {
	const Rccl &api = rccl_or_throw();
	OCRT_HIP(hipSetDevice(device));
	hipStream_t s = (hipStream_t) stream;
	unsigned char pattern[64], back[64];
	for (int i = 0; i < 64; ++i)
		pattern[i] = (unsigned char) (i * 7 + (int) rank);
	void *a = nullptr, *b = nullptr;
	OCRT_HIP(hipMalloc(&a, 64));
	OCRT_HIP(hipMalloc(&b, 64));
	try {
		OCRT_HIP(hipMemcpy(a, pattern, 64, hipMemcpyHostToDevice));
		OCRT_HIP(hipMemset(b, 0, 64));
		NcclGroup group(api);
		OCRT_IN_GROUP(group, api.Send(a, 64, ncclUint8, (int) rank, (ncclComm_t) comm, s));
		OCRT_IN_GROUP(group, api.Recv(b, 64, ncclUint8, (int) rank, (ncclComm_t) comm, s));
		group.end();
		OCRT_HIP(hipStreamSynchronize(s));
		OCRT_HIP(hipMemcpy(back, b, 64, hipMemcpyDeviceToHost));
	} catch (...) {
		(void) hipFree(a);
		(void) hipFree(b);
		throw;
	}
	(void) hipFree(a);
	(void) hipFree(b);
	if (std::memcmp(pattern, back, 64) != 0)
		throw DeviceError("RCCL self send/recv returned other bytes than were sent");
}

// ---------------------------------------------------------------------------------------------------------------
GroupGather::GroupGather(const RayTracer::Options &options, const std::vector<int> &devices_)
	: layout(options, (unsigned int) devices_.size()), devices(devices_), stacked(nullptr), final_image(nullptr) {
	const Rccl &api = rccl_or_throw();
	// (RCCL itself refuses a device that is listed twice -- "duplicate GPU" --: the caller then falls back to peer copies)
	std::vector<ncclComm_t> c(devices.size(), nullptr);
	OCRT_NCCL(api.CommInitAll(c.data(), (int) devices.size(), devices.data()));
	for (ncclComm_t x : c)
		comms.push_back(x);
	try {
		OCRT_HIP(hipSetDevice(devices[0]));
		OCRT_HIP(hipMalloc(&stacked, layout.stride() * layout.nranks));
		OCRT_HIP(hipMalloc(&final_image, (size_t) layout.width * layout.height));
	} catch (...) {  // (the destructor does not run for a constructor that throws)
		for (size_t r = 0; r < comms.size(); ++r)
			if (comms[r] && api.CommDestroy && hipSetDevice(devices[r]) == hipSuccess)
				(void) api.CommDestroy((ncclComm_t) comms[r]);
		if (stacked)
			(void) hipFree(stacked);
		throw;
	}
}

GroupGather::~GroupGather() {
	for (size_t r = 0; r < comms.size(); ++r)
		if (comms[r] && rccl().CommDestroy && hipSetDevice(devices[r]) == hipSuccess)
			(void) rccl().CommDestroy((ncclComm_t) comms[r]);
	if (!devices.empty() && hipSetDevice(devices[0]) == hipSuccess) {
		if (stacked)
			(void) hipFree(stacked);
		if (final_image)
			(void) hipFree(final_image);
	}
}

void GroupGather::enqueue(const std::vector<const void *> &bands, const std::vector<void *> &streams) {
	if (bands.size() != devices.size() || streams.size() != devices.size())
		throw std::invalid_argument("group gather: one band buffer and one stream per device");
	const Rccl &api = rccl_or_throw();
	if (devices.size() > 1) {
		NcclGroup group(api);
		for (unsigned int r = 1; r < devices.size(); ++r) {
			if (!layout.bytesOf(r))
				continue;
			OCRT_IN_GROUP(group, api.Send(bands[r], layout.bytesOf(r), ncclUint8, 0, (ncclComm_t) comms[r], (hipStream_t) streams[r]));
			OCRT_IN_GROUP(group, api.Recv((char *) stacked + (size_t) r * layout.stride(), layout.bytesOf(r), ncclUint8, (int) r,
			                              (ncclComm_t) comms[0], (hipStream_t) streams[0]));
		}
		group.end();
	}
	OCRT_HIP(hipSetDevice(devices[0]));
	launch_assemble_rows(layout, bands[0], stacked, final_image, streams[0]);
}

}  // namespace ocrt
