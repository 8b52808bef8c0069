// fake_rccl.cc -- TEST INFRASTRUCTURE: a stand-in for librccl.so.1 that lets the product's RCCL exchange step
// (ocrt::GroupGather / ocrt::BandGather, csrc/band_gather.cc) run with SEVERAL RANKS ON ONE GPU, which the real RCCL
// refuses ("duplicate GPU").  It implements the entry points the product resolves at run time, for communicators
// whose ranks all live in ONE process (ncclCommInitAll, or ncclCommInitRank called once per rank with the same id): an
// ncclSend / ncclRecv pair becomes a stream-ordered device-to-device copy -- the receive stream waits for an event
// recorded on the send stream, then copies.  What it checks is the PRODUCT's side of the exchange: which buffers,
// offsets, byte counts, peers and streams it hands to RCCL and in what grouping.  A receive must find its send already
// posted (the drivers of the tests post the senders first: a stream cannot be made to wait for a call that has not
// happened yet), with the same byte count; within an ncclCommInitAll communicator a send must also find its receive in
// the same group; what is still unmatched when a communicator is destroyed is reported.
// Built by tests/test_fake_rccl.py into a private directory that the test puts in front of LD_LIBRARY_PATH; never part of
// the product, never on the path of a normal run.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace {

struct FakeComm {
	int rank, nranks, device, group;
	bool strict;  // ncclCommInitAll: every rank's calls of a group come from one caller, so a send must meet its receive at once
};

struct Op {
	bool send;
	const void *src;
	void *dst;
	size_t bytes;
	int peer;
	FakeComm *comm;
	hipStream_t stream;
	hipEvent_t ready;  // sends: recorded on the send stream when the send was posted
};

// FAKE_RCCL_HANG_ON_MISSING_SEND=1: a receive that finds no posted send does not fail -- it BLOCKS its stream, the way a
// real receive waits for a peer that died or posted fewer frames (tests/test_fake_rccl.py: the product must notice and
// give up).  The block is a one-thread kernel that sleeps until ncclCommAbort / ncclCommDestroy sets a flag in mapped
// host memory, or a minute of the device clock has passed (every wave reaches its end whatever happens).
__global__ void hang_kernel(const volatile int *released) {
	const unsigned long long begin = __builtin_amdgcn_s_memrealtime();  // 100 MHz
	while (*released == 0 && __builtin_amdgcn_s_memrealtime() - begin < 6000000000ull)
		__builtin_amdgcn_s_sleep(100);
}
int *release_flag = nullptr;  // hipHostMalloc'ed, mapped

std::mutex mutex;
int depth = 0, next_group = 1;
std::vector<Op> batch;  // the calls of the group that is open
std::vector<Op> sends;  // posted sends that have not met their receive yet
std::vector<std::pair<std::string, int>> groups_by_id;
unsigned long long pairs_done = 0, bytes_done = 0;

ncclResult_t run_batch() {
	std::vector<Op> now;
	now.swap(batch);
	for (Op &op : now)  // the sends first: their data is ready where their stream stands now
		if (op.send) {
			if (hipSetDevice(op.comm->device) != hipSuccess || hipEventCreateWithFlags(&op.ready, hipEventDisableTiming) != hipSuccess ||
			    hipEventRecord(op.ready, op.stream) != hipSuccess)
				return ncclUnhandledCudaError;
			sends.push_back(op);
		}
	for (const Op &r : now) {
		if (r.send)
			continue;
		size_t match = sends.size();
		for (size_t j = 0; j < sends.size(); ++j)  // FIFO per (group, sender, receiver)
			if (sends[j].comm->group == r.comm->group && sends[j].comm->rank == r.peer && sends[j].peer == r.comm->rank) {
				match = j;
				break;
			}
		if (match == sends.size() || sends[match].bytes != r.bytes) {
			const char *hang = std::getenv("FAKE_RCCL_HANG_ON_MISSING_SEND");
			if (hang && hang[0] == '1' && match == sends.size()) {
				std::fprintf(stderr, "fake rccl: receive of %zu bytes on rank %d from %d finds no posted send: blocking its stream\n",
				             r.bytes, r.comm->rank, r.peer);
				if (!release_flag) {
					if (hipHostMalloc((void **) &release_flag, sizeof(int), hipHostMallocMapped) != hipSuccess)
						return ncclUnhandledCudaError;
					*release_flag = 0;
				}
				if (hipSetDevice(r.comm->device) != hipSuccess)
					return ncclUnhandledCudaError;
				hipLaunchKernelGGL(hang_kernel, dim3(1), dim3(1), 0, r.stream, (const volatile int *) release_flag);
				continue;
			}
			std::fprintf(stderr, "fake rccl: receive of %zu bytes on rank %d from %d finds no posted send of that size\n", r.bytes,
			             r.comm->rank, r.peer);
			return ncclInvalidUsage;
		}
		const Op s = sends[match];
		sends.erase(sends.begin() + (long) match);
		if (hipSetDevice(r.comm->device) != hipSuccess || hipStreamWaitEvent(r.stream, s.ready, 0) != hipSuccess ||
		    hipMemcpyAsync(r.dst, s.src, s.bytes, hipMemcpyDeviceToDevice, r.stream) != hipSuccess)
			return ncclUnhandledCudaError;
		(void) hipEventDestroy(s.ready);  // (released once the work that refers to it has completed)
		++pairs_done;
		bytes_done += s.bytes;
	}
	for (const Op &s : sends)
		if (s.comm->strict) {
			std::fprintf(stderr, "fake rccl: send of %zu bytes from rank %d to %d has no receive in its group\n", s.bytes, s.comm->rank, s.peer);
			return ncclInvalidUsage;
		}
	return ncclSuccess;
}

ncclResult_t add(const Op &op) {
	std::lock_guard<std::mutex> lock(mutex);
	if (!op.comm || op.peer < 0 || op.peer >= op.comm->nranks)
		return ncclInvalidArgument;
	batch.push_back(op);
	return depth == 0 ? run_batch() : ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
	static int serial = 0;
	std::memset(id, 0, sizeof *id);
	std::snprintf(id->internal, sizeof id->internal, "fake-rccl-%d", ++serial);
	return ncclSuccess;
}

// (every rank of the communicator is created in THIS process, one call per rank, same id)
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
	if (nranks < 1 || rank < 0 || rank >= nranks)
		return ncclInvalidArgument;
	int device = 0;
	(void) hipGetDevice(&device);
	std::lock_guard<std::mutex> lock(mutex);
	const std::string key(id.internal, sizeof id.internal);
	int group = 0;
	for (const auto &g : groups_by_id)
		if (g.first == key)
			group = g.second;
	if (!group) {
		group = next_group++;
		groups_by_id.emplace_back(key, group);
	}
	*comm = (ncclComm_t) new FakeComm{ rank, nranks, device, group, false };
	if (rank == 0)
		std::fprintf(stderr, "fake rccl: communicator of %d ranks, one call per rank\n", nranks);
	return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
	std::lock_guard<std::mutex> lock(mutex);
	const int group = next_group++;
	for (int r = 0; r < ndev; ++r)
		comms[r] = (ncclComm_t) new FakeComm{ r, ndev, devlist ? devlist[r] : r, group, true };
	std::fprintf(stderr, "fake rccl: %d ranks in one process\n", ndev);
	return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
	std::lock_guard<std::mutex> lock(mutex);
	for (const Op &op : sends)
		if (op.comm == (FakeComm *) comm)
			std::fprintf(stderr, "fake rccl: a send of %zu bytes from rank %d to %d was never received\n", op.bytes, op.comm->rank, op.peer);
	delete (FakeComm *) comm;
	if (pairs_done) {
		std::fprintf(stderr, "fake rccl: %llu send/receive pairs, %llu bytes so far\n", pairs_done, bytes_done);
		pairs_done = 0;
	}
	return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {
	if (release_flag)
		*release_flag = 1;  // (whatever blocks a stream on this communicator's behalf lets go)
	std::fprintf(stderr, "fake rccl: communicator of rank %d aborted\n", ((FakeComm *) comm)->rank);
	return ncclCommDestroy(comm);
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
	*count = ((const FakeComm *) comm)->nranks;
	return ncclSuccess;
}

ncclResult_t ncclGetVersion(int *version) {
	*version = 1;  // (no real RCCL reports this)
	return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
	std::lock_guard<std::mutex> lock(mutex);
	++depth;
	return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
	std::lock_guard<std::mutex> lock(mutex);
	if (depth <= 0)
		return ncclInvalidUsage;
	return --depth == 0 ? run_batch() : ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
	if (type != ncclUint8 && type != ncclChar)
		return ncclInvalidArgument;
	return add(Op{ true, buf, nullptr, count, peer, (FakeComm *) comm, stream, nullptr });
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
	if (type != ncclUint8 && type != ncclChar)
		return ncclInvalidArgument;
	return add(Op{ false, nullptr, buf, count, peer, (FakeComm *) comm, stream, nullptr });
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "invalid usage (fake rccl)" : "error (fake rccl)"; }

}  // extern "C"
