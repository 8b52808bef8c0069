#include "frame_ring.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <sstream>
#include <thread>

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

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

FrameRing::FrameRing(const RayTracer::Options &options, int device, unsigned int rank, unsigned int nranks, unsigned int count)
	: next_frame(0), last{ 0, 0, nullptr, 1 }, have_last(false), epoch(nullptr), pacing(0.5f), period_s(0.0), last_submit_s(0.0) {
	if (count == 0 || count > 16)
		throw std::invalid_argument("a frame ring holds 1 to 16 renderers");
	for (unsigned int k = 0; k < count; ++k) {
		// (a ring of a multi-GPU job will get a gather: its stream takes the highest priority class for itself, the
		// hosts are dealt over the others -- DeviceRenderer's constructor)
		hosts.emplace_back(new DeviceRenderer(options, device, rank, nranks, count > 1 ? (int) k : -1, nranks > 1));
		hosts.back()->setDeviceShare(count);
		hosts.back()->expectFrames(~0ull);  // (a ring is a stream of frames: what an upload can prepare for them pays)
	}
	bound.assign(2 * count, nullptr);
	gather_pending.assign(2 * count, false);
	extra.assign(count, nullptr);
	try {
		OCRT_HIP(hipSetDevice(hosts.front()->deviceIndex()));
		for (unsigned int k = 0; k < count; ++k) {
			const size_t bytes = (size_t) hosts[k]->localRows() * hosts[k]->width();
			OCRT_HIP(hipMalloc(&extra[k], bytes ? bytes : 1));
		}
		hipEvent_t e;
		OCRT_HIP(hipEventCreate(&e));
		epoch = e;
		resetClock();
	} catch (...) {  // (no destructor runs for a constructor that throws: give back what was taken)
		for (void *p : extra)
			if (p)
				(void) hipFree(p);
		if (epoch)
			(void) hipEventDestroy((hipEvent_t) epoch);
		throw;
	}
}

FrameRing::~FrameRing() {
	// The gather first: it waits for its own stream (whose pending steps read the band buffers that go next) -- or, if a
	// wait ran into its deadline, aborts the communicator, so that the operation that can never finish does not keep every
	// later hipFree of this process waiting.  Then the renderers (their destructors wait for their streams), the buffers,
	// the epoch event.
	try {
		drain();
	} catch (...) {
	}
	gather.reset();
	hosts.clear();
	for (void *p : extra)
		if (p)
			(void) hipFree(p);
	if (epoch)
		(void) hipEventDestroy((hipEvent_t) epoch);
}

size_t FrameRing::upload(const PackedScene &scene) {
	drain();
	// ONE copy of the scene on the GPU, whatever the number of hosts: they all walk the same arrays
	for (auto &h : hosts)
		h->synchronize();
	std::shared_ptr<const DeviceScene> on_device = DeviceScene::create(hosts.front()->deviceIndex(), scene, hosts.front()->rayTracer().options);
	size_t bytes = on_device->bytes();
	for (auto &h : hosts)
		bytes += h->adopt(on_device, h == hosts.front() ? nullptr : hosts.front().get());  // (the hit list's layout is counted once)
	// A stream of frames of this scene is coming: which form of the ambient-occlusion pass's node loop it takes is worth a
	// few frames of measuring (DeviceRenderer::calibrateAoPrefetch: ~15 ms once per upload).  The first host measures on
	// the idle device, all hosts follow.
	if (calibrate_at_upload) {
		// ... and so is the order its tiles are claimed in: three frames whose pass books every claim's duration to its tiles
		// (DeviceRenderer::measureTileCosts, ~4 ms; headline pass 1.02 -> 0.93 ms, profiles/r05_notes.md)
		if (hosts.front()->measureTileCosts(3))
			for (auto &h : hosts)
				h->takeOrderFrom(*hosts.front());
		const bool prefetch = hosts.front()->calibrateAoPrefetch(&calibration_ms[0], &calibration_ms[1]);
		for (auto &h : hosts)
			h->setAoPrefetch(prefetch);
	}
	// every host's two frames are captured now, not in the middle of the stream that follows
	for (unsigned int slot = 0; slot < bound.size(); ++slot)
		hosts[slot % hosts.size()]->prepareFrame(bufferOf(slot));
	uploaded_bytes = bytes;
	return bytes;
}

void FrameRing::setGraphMode(bool on) {
	for (auto &h : hosts)
		h->setGraphMode(on);
}

void FrameRing::bindOutput(unsigned int slot, void *device_u8) {
	if (slot >= bound.size())
		throw std::invalid_argument("frame ring: no such slot");
	for (const Collected &c : open)
		if (c.slot == slot)
			throw std::logic_error("frame ring: the slot has a frame in flight");
	waitSlotFree(slot);  // (a gather of the slot's last frame may still be reading the buffer that is about to be replaced)
	bound[slot] = device_u8;
	DeviceRenderer &h = *hosts[slot % hosts.size()];
	if (h.sceneReady())
		h.prepareFrame(bufferOf(slot));
}

void FrameRing::attachGather(std::unique_ptr<BandGather> g) {
	drain();
	if (g && g->slots() < bound.size())
		throw std::invalid_argument("frame ring: the gather needs one slot per band buffer");
	gather = std::move(g);
}

void FrameRing::waitSlotFree(unsigned int slot) {
	if (gather && gather_pending[slot]) {
		gather->wait(slot);  // (the slot's band buffer and final image are about to be written again)
		gather_pending[slot] = false;
	}
}

uint64_t FrameRing::submit() {
	if (open.size() >= hosts.size())
		throw std::logic_error("frame ring: every renderer has a frame in flight, collect one first");
	const unsigned int slot = (unsigned int) (next_frame % bound.size());
	if (pacing > 0.0f && hosts.size() > 1 && period_s > 0.0 && open.size() + 1 >= hosts.size()) {
		// (see setPacing: keep the frames of the ring out of step)
		const double t_pace = now_s(), due = last_submit_s + (double) pacing * period_s;
		if (due > t_pace && due - t_pace < 0.25) {
			if (due - t_pace > 200e-6)
				std::this_thread::sleep_for(std::chrono::duration<double>(due - t_pace - 100e-6));
			while (now_s() < due)
				std::this_thread::yield();
			cpu.wait_s += now_s() - t_pace;
		}
	}
	const double t0 = now_s();
	last_submit_s = t0;
	waitSlotFree(slot);  // (its gather was enqueued size() frames ago: long done)
	hosts[next_frame % hosts.size()]->enqueueFrame(bufferOf(slot));
	open.push_back(Collected{ next_frame, slot, nullptr, (unsigned int) open.size() + 1u });
	cpu.submit_s += now_s() - t0;
	return next_frame++;
}

FrameRing::Collected FrameRing::collect() {
	if (open.empty())
		throw std::logic_error("frame ring: no frame in flight");
	Collected c = open.front();
	open.pop_front();
	DeviceRenderer &h = *hosts[c.frame % hosts.size()];
	const double t0 = now_s();
	h.waitForStream();
	const double t1 = now_s();
	h.synchronize();
	c.device_bands = bufferOf(c.slot);
	const float *t = h.lastFrameTimes();
	times.push_back(Times{ c.frame, { t[0], t[1], t[2], t[3] } });
	if (times.size() > kept_times)
		times.pop_front();
	if (gather) {
		gather->enqueue(c.slot, c.device_bands);
		gather_pending[c.slot] = true;
	}
	last = c;
	have_last = true;
	// The time per finished frame, from the DEVICE's clock: the frame took lastKernelMs() from its first to its last kernel
	// (HIP events on its host's stream) while sharing the device with the frames still in flight -- so one frame is
	// finished every (that time / frames in flight) in the steady state.  (The CPU's clock between two collects would
	// also hold graph captures and whatever the caller did in between.)
	{
		const size_t sharing = std::max<size_t>(c.in_flight_at_submit, open.size() + 1);  // (a ring that is draining: as it began)
		const double sample = (double) h.lastKernelMs() * 1e-3 / (double) sharing;
		if (sample > 0.0)
			period_s = period_s > 0.0 ? 0.75 * period_s + 0.25 * sample : sample;
	}
	cpu.wait_s += t1 - t0;
	cpu.collect_s += now_s() - t1;
	++cpu.frames;
	return c;
}

void FrameRing::step() {
	submit();
	const size_t keep = hosts.size() > 1 ? hosts.size() - 1 : 0;
	while (open.size() > keep)
		collect();
}

void FrameRing::drain() {
	while (!open.empty())
		collect();
	for (unsigned int slot = 0; slot < bound.size(); ++slot)
		waitSlotFree(slot);
}

void *FrameRing::bufferOf(unsigned int slot) const {
	if (bound[slot])
		return bound[slot];
	const unsigned int n = (unsigned int) hosts.size();
	return slot < n ? const_cast<void *>(hosts[slot]->deviceBands()) : extra[slot - n];
}

const void *FrameRing::lastImageDevice() {
	if (!have_last)
		throw std::logic_error("frame ring: no frame collected yet");
	if (!gather)
		return last.device_bands;
	waitSlotFree(last.slot);
	return gather->image(last.slot);
}

void FrameRing::downloadLast(unsigned char *host_image) {
	const void *src = lastImageDevice();
	if (!src)
		throw std::logic_error("frame ring: the assembled image lives on rank 0");
	if (!gather && hosts.front()->params().part.nranks != 1)
		throw std::logic_error("frame ring: a partitioned ring without a gather holds bands, not the image");
	OCRT_HIP(hipSetDevice(hosts.front()->deviceIndex()));
	OCRT_HIP(hipMemcpy(host_image, src, (size_t) width() * height(), hipMemcpyDeviceToHost));
}

void FrameRing::resetClock() {
	drain();
	OCRT_HIP(hipSetDevice(hosts.front()->deviceIndex()));
	OCRT_HIP(hipEventRecord((hipEvent_t) epoch, (hipStream_t) hosts.front()->streamHandle()));
	OCRT_HIP(hipEventSynchronize((hipEvent_t) epoch));
	for (auto &h : hosts)
		h->setEpochEvent(epoch);
	times.clear();
	cpu = CpuTimes{};
}

void FrameRing::keepFrameTimes(bool on) {
	for (auto &h : hosts)
		h->setKeepStamps(on);
}

bool FrameRing::frameTimes(uint64_t frame, float out[4]) const {
	for (const Times &t : times)
		if (t.frame == frame) {
			for (int k = 0; k < 4; ++k)
				out[k] = t.t[k];
			return true;
		}
	return false;
}

}  // namespace ocrt
