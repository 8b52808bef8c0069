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
struct PackedScene;    // scene_pack.h
class DeviceRenderer;  // device_renderer.h
class FrameRing;       // frame_ring.h
class GroupGather;     // band_gather.h
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

		// New: the same with the arrays already validated and packed (ocrt::pack_scene, scene_pack.h) -- the CPU half of
		// an upload, which a caller can run while the device is still coming up (HipHost::warmUp).
		void upload(const ocrt::PackedScene &packed);

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
		// New: brings up the HIP runtime, the device context and the kernels' code object without creating a host.  A
		// caller that still has CPU work to do before it needs the host (mesh loading, BVH build: src/render.cc:52-80)
		// runs this on a second thread; constructing the host afterwards costs milliseconds instead of 100-200 ms.
		static void warmUp(int device = -1);
		// ... and also creates the render host's device state for `rt` (streams, image and hit-list buffers): the next
		// HipHost(rt', device) whose options equal rt's adopts it instead of allocating its own.
		static void warmUp(const RayTracer &rt, int device);
		// ... and reserves the device allocation of a scene of `triangles` triangles (a mesh file's header says how many
		// before the mesh is read): the upload that follows finds its memory waiting (DeviceScene::reserve).
		static void reserveScene(const RayTracer &rt, int device, size_t triangles);

		// Milliseconds the ray-casting kernel of the last operator()() took,
		// measured with HIP events on the launch stream.
		float lastKernelMs() const;
		ocrt::RenderStats lastStats();

	private:
		const RayTracer &rt;
		std::unique_ptr<ocrt::DeviceRenderer> impl;
};

// A steady stream of frames behind the same seam: `hosts` render hosts of one scene on one GPU that take frames in
// turn (ocrt::FrameRing: own streams in different priority classes, one captured hipGraph per host), so that the next
// frames' passes fill the wave slots a finishing ambient-occlusion pass frees.  Same member names as HipHost --
// operator()() is one blocking frame, exactly the reference's (src/opencl_host.cc:137-149) --, plus frames(k).
class HipHostRing {
	public:
		HipHostRing(const RayTracer &rt, unsigned int hosts = 3, int device = -1);
		~HipHostRing();
		HipHostRing(const HipHostRing &) = delete;
		HipHostRing &operator=(const HipHostRing &) = delete;

		void upload(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes,
		            const std::vector<Vec3f> &aabbs, const std::vector<Vec3f> &vertices,
		            const std::vector<Vec3f> &vnormals);
		void upload(const ocrt::PackedScene &packed);
		bool operator()();                            // one frame, blocking
		bool frames(unsigned int count);              // `count` frames, up to size() - 1 in flight; blocks until all are done
		void download(float *image);                  // the float image of the last frame
		void downloadResized(unsigned char *image);   // its 8-bit image (resized on the device as part of the frame)
		float lastKernelMs() const;                   // kernel time of the last frame (HIP events on its host's stream)
		ocrt::RenderStats lastStats();
		unsigned int size() const;
		static void printInfo() { HipHost::printInfo(); }

	private:
		const RayTracer &rt;
		std::unique_ptr<ocrt::FrameRing> ring;
		unsigned int last_host;
};

// One frame on several GPUs of a node, in ONE process (`render --gpus N`): the scene is replicated, the image is cut
// into bands of whole supersample blocks dealt round-robin to the devices (ocrt::Partition), every device renders and
// box-filters its own bands, and ONE exchange step brings the 8-bit bands to the first device: an RCCL gather
// (ncclSend / ncclRecv in one group over xGMI, ocrt::GroupGather) followed by a kernel that moves the rows to their
// place; where RCCL cannot be used -- the library is absent, or two ranks share a device ($OCRT_SHARE_DEVICES
// rehearsal) -- device-to-device copies (hipMemcpyPeerAsync) into a staging buffer and the same kernel.  Same member
// names as HipHost, so the CLI drives either.  New: the reference is single-device (src/opencl_host.cc:16-32).
class HipHostGroup {
	public:
		// `devices` GPUs starting at device `first` (< 0: $OCRT_DEVICE or 0).  More ranks than visible GPUs is an error
		// unless $OCRT_SHARE_DEVICES is set (rehearsal on a smaller box: ranks are mapped round-robin).
		// `gather`: "rccl", "peer", or "auto" / nullptr (RCCL where it can be set up, else peer copies).
		HipHostGroup(const RayTracer &rt, unsigned int devices, int first = -1, const char *gather = nullptr);
		~HipHostGroup();
		HipHostGroup(const HipHostGroup &) = delete;
		HipHostGroup &operator=(const HipHostGroup &) = delete;

		void upload(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes,
		            const std::vector<Vec3f> &aabbs, const std::vector<Vec3f> &vertices,
		            const std::vector<Vec3f> &vnormals);
		void upload(const ocrt::PackedScene &packed);
		bool operator()();                            // all devices render their bands; blocks until every one is done
		void downloadResized(unsigned char *image);   // resize on every device, gather, assemble width x height bytes
		float lastKernelMs() const;                   // slowest device's kernel time of the last frame
		ocrt::RenderStats lastStats();                // summed over the devices
		unsigned int size() const { return (unsigned int) hosts.size(); }
		const char *gatherName() const { return gather ? "RCCL gather" : "peer copies"; }

	private:
		const RayTracer &rt;
		std::vector<std::unique_ptr<ocrt::DeviceRenderer>> hosts;
		std::unique_ptr<ocrt::GroupGather> gather;  // the exchange step over RCCL, when it could be set up
		void *staging;        // peer-copy form, on hosts[0]'s device: the ranks' band buffers, one stride apart
		void *assembled;      // ... and the assembled image
		size_t staging_bytes;
};

// Source compatibility with callers written against the reference.  (-DOCRT_DROPIN_RING: the same callers on a frame
// ring instead -- their operator()() is one blocking frame either way; oracle/Makefile builds the reference's main()
// both ways.)
#ifdef OCRT_DROPIN_RING
using OpenCLHost = HipHostRing;
#else
using OpenCLHost = HipHost;
#endif
