// frame_ring.h -- several renderers of ONE scene on ONE GPU that take frames in turn.
//
// A render host is one stream: one kernel after the other.  Alone, the last quarter of a frame's ambient-occlusion
// pass runs at falling occupancy (its queues are drained, the persistent workgroups end one by one) and the primary
// pass is latency-bound on its own; with the NEXT frames already enqueued on other renderers' streams those wave slots
// are taken at once.  The ring owns the renderers (one captured hipGraph per renderer: a frame is one graph launch),
// their streams (consecutive renderers in different priority classes, so that they cannot share a hardware queue),
// their 8-bit band buffers and the frame bookkeeping; with a communicator attached it also runs the one exchange step
// of a multi-GPU frame -- the gather of the ranks' bands on rank 0 -- behind the next frames.  Callers see
// submit() / collect(), or step() for a steady stream of frames; nothing on the per-frame path is left to the caller.
//
// The reference renders one blocking frame per OpenCLHost::operator()() (src/opencl_host.cc:137-149,
// src/render.cc:109-111); a ring of ONE renderer is exactly that.  Failures are exceptions (DeviceError,
// std::logic_error, std::invalid_argument), as in DeviceRenderer.
#pragma once
#include <cstdint>
#include <deque>
#include <memory>
#include <vector>

#include "band_gather.h"
#include "device_renderer.h"

namespace ocrt {

class FrameRing {
	public:
		FrameRing(const RayTracer::Options &options, int device, unsigned int rank, unsigned int nranks, unsigned int hosts);
		~FrameRing();
		FrameRing(const FrameRing &) = delete;
		FrameRing &operator=(const FrameRing &) = delete;

		size_t upload(const PackedScene &scene);  // every renderer of the ring; returns the bytes requested on the device
		size_t uploadedBytes() const { return uploaded_bytes; }  // what the last upload() returned
		// upload() measures which form of the ambient-occlusion pass suits the scene (on by default); what it found:
		// ms per ao_kernel without / with the look-ahead loads (0: not measured) and the form the hosts now launch.
		void setCalibration(bool on) { calibrate_at_upload = on; }
		const float *calibrationMs() const { return calibration_ms; }
		bool aoPrefetch() const { return hosts.front()->aoPrefetch(); }
		unsigned int size() const { return (unsigned int) hosts.size(); }
		DeviceRenderer &host(unsigned int slot) { return *hosts.at(slot); }
		void setGraphMode(bool on);
		// Pacing: with the ring about to be full again, a frame is submitted no sooner than `beta` x the time per finished
		// frame (a frame's time on the device over the frames it shared it with, smoothed) after the previous submission.  Frames that finish together would otherwise start their successors together, and the
		// ring falls into lockstep -- every host in its primary pass at once, then every host at the falling end of its
		// ambient-occlusion pass at once --, which is exactly what several hosts are there to avoid (interior scene,
		// three hosts: 1.52 -> 1.33 ms per frame; profiles/r03_notes.md).  0 switches it off; default 0.5 (0.3 until the
		// AO pass dealt its claims by cursor; on those kernels 0.4-0.5 is 2-3 % better on the bunny frames, the interior's alike).
		void setPacing(float beta) { pacing = beta < 0.0f ? 0.0f : beta > 1.0f ? 1.0f : beta; }

		// Frames write their 8-bit bands (the device resize) into slots() = 2 x size() band buffers in turn: frame f
		// is rendered by renderer f % size() into buffer f % slots(), so a frame's bands -- and, with a gather, its
		// assembled image -- stay untouched while the next size() frames are submitted, and nobody waits for a buffer.
		unsigned int slots() const { return (unsigned int) bound.size(); }
		// Buffer `slot` is caller-owned DEVICE memory of localRows() x width bytes from now on (nullptr: the ring's own
		// again) -- e.g. a tensor that a caller-side collective sends.
		void bindOutput(unsigned int slot, void *device_u8);

		// Attaches the exchange step: from now on collect() also enqueues the gather of the frame's bands to rank 0
		// and, there, the assembly of the final image (BandGather), and a slot is not reused before its gather is done.
		void attachGather(std::unique_ptr<BandGather> gather);
		bool hasGather() const { return gather != nullptr; }
		void gatherSelfTest() { gather->selfTest(); }
		BandGather *gatherOrNull() { return gather.get(); }

		// Enqueues the next frame (all passes + device resize) on the next renderer and returns its number (0, 1, ...).
		// Throws std::logic_error when every renderer already has a frame in flight.
		uint64_t submit();
		struct Collected {
			uint64_t frame;
			unsigned int slot;         // the band buffer it was rendered into
			const void *device_bands;  // localRows() x width bytes; valid until frame + slots() is submitted
			unsigned int in_flight_at_submit = 1;  // frames on the device when this one joined them (itself included)
		};
		// Waits (on the CPU: a stream-level wait for a frame that has just begun would sit in a hardware queue as a
		// barrier packet ahead of whatever else shares that queue) for the OLDEST frame in flight.
		Collected collect();
		unsigned int inFlight() const { return (unsigned int) open.size(); }
		uint64_t submitted() const { return next_frame; }
		// A steady stream of frames: submit one, then collect the oldest ones until at most size() - 1 (at least one
		// renderer: none) are in flight.  drain() collects what is left and waits for the gathers.
		void step();
		void drain();

		// The last collected frame.  Without a gather: its bands (the whole image for nranks == 1).  With one: on rank 0
		// the assembled width x height image (waits for its gather), nullptr elsewhere.
		const void *lastImageDevice();
		// ... copied to the host: width x height bytes (unpartitioned ring, or rank 0 of a gathering one).
		void downloadLast(unsigned char *host_image);

		uint32_t localRows() const { return hosts.front()->localRows(); }
		uint32_t width() const { return hosts.front()->width(); }
		uint32_t height() const { return hosts.front()->height(); }

		// Time stamps of collected frames since resetClock(): milliseconds from the reset to the frame's begin, to the
		// start and end of its ao_kernel launch, to its end (HIP events on the renderers' streams).  Kept for the last
		// `kept_times` frames.
		void resetClock();
		void keepFrameTimes(bool on);  // (graph replay: costs a small blocking copy per collected frame; default off)
		bool frameTimes(uint64_t frame, float out[4]) const;
		// What the frames cost the CPU since resetClock(): seconds spent inside submit() (the launches), inside collect()
		// waiting for the device, and inside collect() otherwise (event arithmetic, enqueueing the gather).
		struct CpuTimes {
			double submit_s = 0, wait_s = 0, collect_s = 0;
			uint64_t frames = 0;
		};
		const CpuTimes &cpuTimes() const { return cpu; }

	private:
		void waitSlotFree(unsigned int slot);
		std::vector<std::unique_ptr<DeviceRenderer>> hosts;
		void *bufferOf(unsigned int slot) const;
		std::vector<void *> bound;        // per slot: caller-owned destination, or nullptr
		std::vector<void *> extra;        // the second band buffer of each renderer (slot size() + k), owned by the ring
		std::deque<Collected> open;       // frames in flight, oldest first (device_bands filled at collect)
		std::unique_ptr<BandGather> gather;
		std::vector<bool> gather_pending; // per slot: a gather of this slot's bands was enqueued and not yet waited for
		uint64_t next_frame;
		size_t uploaded_bytes = 0;
		bool calibrate_at_upload = true;
		float calibration_ms[2] = { 0.0f, 0.0f };
		Collected last;
		bool have_last;
		void *epoch;  // hipEvent_t
		struct Times {
			uint64_t frame;
			float t[4];
		};
		static constexpr size_t kept_times = 256;
		std::deque<Times> times;
		CpuTimes cpu;
		float pacing;
		double period_s;        // time per finished frame: a frame's time on the device / the frames it shared it with, smoothed (0: unknown)
		double last_submit_s;
};

}  // namespace ocrt
