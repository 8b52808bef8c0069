// band_gather.h -- the one exchange step of a multi-GPU frame: every rank's compact 8-bit band buffer goes to rank 0
// over RCCL (ncclSend / ncclRecv in one group: xGMI point-to-point, no ring), where a small kernel moves the rows to
// their place in the final image.  The reference is single-device (src/opencl_host.cc:16-32): this is new.
//
// Two forms of the same step:
//   BandGather   one rank of a job with ONE PROCESS PER GPU (bench.py under torch.distributed.run: the unique id is
//                made on rank 0 and handed round by torch.distributed, the communicator is this library's own);
//   GroupGather  all ranks in ONE process (`render --gpus N`, HipHostGroup): ncclCommInitAll.
// RCCL is opened at run time (dlopen of librccl.so.1, the one the process already holds if torch brought it): the
// library itself loads on machines without RCCL, and asking for a gather there fails loudly (DeviceError).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "ray_tracer.h"

namespace ocrt {

// Where the ranks' band rows live in rank 0's receive buffer and in the image.
struct BandPlan {
	uint32_t width, height, nranks;
	uint32_t rows_per_band;           // output rows per band (lcm(8, n) / n)
	uint32_t max_rows;                // largest localRows over the ranks: the stride of the receive buffer, in rows
	std::vector<uint32_t> local_rows; // per rank
	BandPlan(const RayTracer::Options &options, unsigned int nranks);
	size_t bytesOf(unsigned int rank) const { return (size_t) local_rows[rank] * width; }
	size_t stride() const { return (size_t) max_rows * width; }
};

// Enqueues the row scatter on `stream` (of the current device): image row y <- the band row holding it, read from
// `own` for rank 0's rows (its band buffer, never copied) and from `stacked + r * stride` for rank r's.
void launch_assemble_rows(const BandPlan &plan, const void *own, const void *stacked, void *image, void *stream);

bool rccl_available();                 // librccl.so.1 can be opened and has the entry points used here
constexpr size_t RCCL_UNIQUE_ID_BYTES = 128;
void rccl_unique_id(void *out128);     // ncclGetUniqueId; throws DeviceError

class BandGather {
	public:
		// Collective over the job: every rank constructs it with the same id (ncclCommInitRank).  `slots`: how many
		// frames may have their gather in flight at once (one receive buffer + one final image each, on rank 0).
		BandGather(const RayTracer::Options &options, unsigned int rank, unsigned int nranks, int device, const void *unique_id,
		           unsigned int slots);
		~BandGather();
		BandGather(const BandGather &) = delete;
		BandGather &operator=(const BandGather &) = delete;

		// `device_bands` (localRows x width bytes, complete on the device) -> rank 0, asynchronously on the gather's
		// own stream; on rank 0 followed by the assembly of slot `slot`'s final image.
		void enqueue(unsigned int slot, const void *device_bands);
		// CPU wait for that slot's last gather (and assembly) -- bounded: DeviceError after timeout() seconds without it
		// (a peer that died or posted fewer frames must not hang this rank), default 30 s.
		void wait(unsigned int slot);
		void setTimeout(double seconds) { timeout_s = seconds > 0.0 ? seconds : 30.0; }
		double timeout() const { return timeout_s; }
		// What the communicator says about itself: its number of ranks (ncclCommCount) and RCCL's version code
		// (ncclGetVersion); -1 where the library lacks the entry point.
		void describe(int *comm_ranks, int *rccl_version) const;
		const void *image(unsigned int slot) const;    // rank 0: width x height bytes on the device; else nullptr
		unsigned int slots() const { return (unsigned int) done.size(); }
		const BandPlan &plan() const { return layout; }
		// A grouped ncclSend / ncclRecv of a few bytes from this rank to itself, checked: proves the entry points on a
		// box where the job is a single rank and the gather proper moves nothing.
		void selfTest();

	private:
		void allocate(unsigned int slots);  // stream, events, rank 0's buffers
		void release();
		BandPlan layout;
		unsigned int rank, nranks;
		int device;
		double timeout_s = 30.0;
		bool broken = false;        // a wait timed out: the communicator is aborted, not destroyed
		void *comm;                 // ncclComm_t
		void *stream;               // hipStream_t: the gather's own
		std::vector<void *> done;   // hipEvent_t per slot
		std::vector<void *> stacked, final_image;  // rank 0, per slot
};

class GroupGather {
	public:
		// One communicator per device of `devices` (ncclCommInitAll; RCCL refuses a device listed twice -> DeviceError).
		GroupGather(const RayTracer::Options &options, const std::vector<int> &devices);
		~GroupGather();
		GroupGather(const GroupGather &) = delete;
		GroupGather &operator=(const GroupGather &) = delete;
		// bands[r] on device r, complete on streams[r]'s device queue order: the sends are enqueued on streams[r], the
		// receives and the assembly on streams[0]; returns after enqueueing.  image(): on devices[0].
		void enqueue(const std::vector<const void *> &bands, const std::vector<void *> &streams);
		const void *image() const { return final_image; }
		const BandPlan &plan() const { return layout; }

	private:
		BandPlan layout;
		std::vector<int> devices;
		std::vector<void *> comms;
		void *stacked, *final_image;
};

}  // namespace ocrt
