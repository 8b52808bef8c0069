// device_internal.h -- what the translation units behind device_renderer.h share: the HIP error check, the allocation
// helpers, a few limits.  Not installed, not included by any header.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>

#include "device_renderer.h"

namespace ocrt {

inline void hip_check(hipError_t err, const char *what) {
	if (err != hipSuccess) {
		std::ostringstream ss;
		ss << "HIP error: " << hipGetErrorName(err) << " (" << hipGetErrorString(err) << ") in " << what;
		throw DeviceError(ss.str());
	}
}
#define OCRT_HIP(call) ::ocrt::hip_check((call), #call)

constexpr size_t MAX_ENTRY_TABLE_BYTES = (size_t) 2 << 30;
// (the blocks' coordinates are packed into 16 bits each)
#define PRIMARY_BY_COST_OK(kp) ((kp).tiles_x < 8192u && (kp).local_tile_rows < 8192u && (kp).shared_walk)  // (13 bits each in an entry of the list)
constexpr uint32_t MAX_STRIP_TILES = 32u;
constexpr size_t BIG_SCENE_BYTES = (size_t) 96 << 20;  // three times the L2s

inline void *device_alloc(size_t bytes) {
	void *p = nullptr;
	OCRT_HIP(hipMalloc(&p, bytes ? bytes : 1));
	return p;
}

inline void device_free(void *&p) {
	if (p)
		(void) hipFree(p);
	p = nullptr;
}

// (A/B build: OCRT_UPLOAD_TIMINGS=1 prints where an upload's time goes)
struct UploadClock {
#ifdef OCRT_DEBUG_KNOBS
	std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
	const bool on = std::getenv("OCRT_UPLOAD_TIMINGS") != nullptr;
	void mark(const char *what) {
		if (!on)
			return;
		const auto now = std::chrono::steady_clock::now();
		std::fprintf(stderr, "upload: %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
		last = now;
	}
#else
	void mark(const char *) {}
#endif
};

}  // namespace ocrt
