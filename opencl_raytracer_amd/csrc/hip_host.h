// hip_host.h -- the render host: device selection, scene upload, frame launch,
// image download.  Drop-in for the reference's OpenCLHost (reference
// include/opencl_host.h:6-144, src/opencl_host.cc): same five members, same
// argument meaning, same error behaviour, so reference src/render.cc:84-116
// compiles against it through the alias at the bottom of this file.
#pragma once
// The reference's opencl_host.h pulls in <iostream> itself and, through CL/cl.hpp,
// <utility> <limits> <iterator> <exception> <cstring> <cstdlib> ...; callers written against it
// (reference src/render.cc:136 uses std::ostream_iterator without including <iterator>)
// rely on that, so the drop-in provides the same set.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <iostream>
#include <iterator>
#include <limits>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "device_types.h"
#include "ray_tracer.h"
#include "vec3.h"

namespace ocrt {
class DeviceRenderer;  // hip_host.cc
}

class HipHost {
	public:
		// Selects the HIP device (device < 0: $OCRT_DEVICE or 0) and prepares the
		// launch constants from rt.options / rt.totalWidth / rt.totalHeight.  Keeps
		// a reference to rt, which must outlive the host (as in the reference).
		// Throws std::runtime_error("No device found") when no GPU is visible
		// (reference src/opencl_host.cc:30-31).
		explicit HipHost(const RayTracer &rt, int device = -1);
		// Multi-GPU form: this host renders only the image bands of `rank` out of
		// `nranks` (see ocrt::Partition); rank 0 of 1 is the whole image.
		HipHost(const RayTracer &rt, int device, unsigned int rank, unsigned int nranks);
		~HipHost();
		HipHost(const HipHost &) = delete;
		HipHost &operator=(const HipHost &) = delete;

		// Synchronous host->device copy of the scene; the caller may clear its
		// vectors as soon as this returns (reference src/render.cc:96-103).
		//   faces    3T vertex ids in leaf order     nodes  pre-order subtree sizes
		//   aabbs    (min,max) per node, 16-B items  vertices / vnormals  16-B items
		// Prints "Requested N kB of memory." like reference src/opencl_host.cc:128.
		// Malformed arrays / device errors: message on stderr + exit(EXIT_FAILURE)
		// (the reference's check(), include/opencl_host.h:21-26).
		void upload(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes,
		            const std::vector<Vec3f> &aabbs, const std::vector<Vec3f> &vertices,
		            const std::vector<Vec3f> &vnormals);

		// Renders the whole frame (this rank's bands) and blocks until it is done;
		// false means the caller should exit (reference src/opencl_host.cc:137-149).
		bool operator()();

		// Copies the totalWidth*totalHeight float image into caller memory
		// (reference src/opencl_host.cc:150-153).
		void download(float *image);

		// New: box filter + 8-bit quantisation on the device (bit-identical to
		// RayTracer::resize), then copies width*height bytes into caller memory.
		void downloadResized(unsigned char *image);

		// Device table on stdout (reference src/opencl_host.cc:76-119).
		static void printInfo();

		// Milliseconds the ray-casting kernel of the last operator()() took,
		// measured with HIP events on the launch stream.
		float lastKernelMs() const;
		ocrt::RenderStats lastStats();

	private:
		const RayTracer &rt;
		std::unique_ptr<ocrt::DeviceRenderer> impl;
};

// One frame on several GPUs of a node, in ONE process (`render --gpus N`): the scene is replicated, the image is cut
// into bands of whole supersample blocks dealt round-robin to the devices (ocrt::Partition), every device renders and
// box-filters its own bands, and the 8-bit bands are copied device-to-device over xGMI (hipMemcpyPeerAsync) into a
// staging buffer on the first device, from where the assembled image is read.  Same member names as HipHost, so the
// CLI drives either.  New: the reference is single-device (src/opencl_host.cc:16-32).
class HipHostGroup {
	public:
		// `devices` GPUs starting at device `first` (< 0: $OCRT_DEVICE or 0).  More ranks than visible GPUs is an error
		// unless $OCRT_SHARE_DEVICES is set (rehearsal on a smaller box: ranks are mapped round-robin).
		HipHostGroup(const RayTracer &rt, unsigned int devices, int first = -1);
		~HipHostGroup();
		HipHostGroup(const HipHostGroup &) = delete;
		HipHostGroup &operator=(const HipHostGroup &) = delete;

		void upload(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes,
		            const std::vector<Vec3f> &aabbs, const std::vector<Vec3f> &vertices,
		            const std::vector<Vec3f> &vnormals);
		bool operator()();                            // all devices render their bands; blocks until every one is done
		void downloadResized(unsigned char *image);   // resize on every device, gather, assemble width x height bytes
		float lastKernelMs() const;                   // slowest device's kernel time of the last frame
		ocrt::RenderStats lastStats();                // summed over the devices
		unsigned int size() const { return (unsigned int) hosts.size(); }

	private:
		const RayTracer &rt;
		std::vector<std::unique_ptr<ocrt::DeviceRenderer>> hosts;
		void *staging;        // on hosts[0]'s device: the devices' band buffers back to back
		size_t staging_bytes;
};

// Source compatibility with callers written against the reference.
using OpenCLHost = HipHost;
