// tests/hip_host_stub.cc -- test infrastructure, NOT part of the product: the members of csrc/hip_host.h without a device,
// so that the `render` CLI (csrc/render.cc) links and runs on the CPU under AddressSanitizer + UBSan
// (tests/test_host_sanitizers.py; the reference's own Debug build is an ASan/UBSan build of the whole CLI,
// CMakeLists.txt:34-40, and GPU sanitizers are not available on the pool).
// What is real here: option parsing, the OFF loader, vertex normals, both BVH strategies, the face sort, pack_scene's
// validation and records, the walk-tree rebuild, RayTracer::resize and the PGM writer -- every line of host code the CLI
// runs around the device.  What stands in: the frame itself, a deterministic pattern instead of ray casting.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "hip_host.h"
#include "scene_pack.h"

namespace ocrt {
// (the stub's own state behind the pointers hip_host.h forward-declares)
class DeviceRenderer {
	public:
		std::vector<float> image;
		size_t triangles = 0;
		unsigned int frames = 0;
};
class FrameRing {
	public:
		DeviceRenderer host;
		unsigned int hosts = 1;
};
class GroupGather {};
}

namespace {
void no_device_if_asked() {
	if (std::getenv("OCRT_STUB_NO_DEVICE"))
		throw std::runtime_error("No device found");  // (reference src/opencl_host.cc:30-31)
}
void fill(ocrt::DeviceRenderer &d, const RayTracer &rt) {
	d.image.resize((size_t) rt.totalWidth * rt.totalHeight);
	for (uint32_t y = 0; y < rt.totalHeight; ++y)
		for (uint32_t x = 0; x < rt.totalWidth; ++x)
			d.image[(size_t) y * rt.totalWidth + x] = (float) ((x * 31u + y * 17u + d.triangles) % 256u) / 255.0f;
	++d.frames;
}
void take(ocrt::DeviceRenderer &d, const ocrt::PackedScene &packed) { d.triangles = packed.tris.size(); }
ocrt::RenderStats stats_of(const ocrt::DeviceRenderer &d, const RayTracer &rt) {
	ocrt::RenderStats s{};
	s.primary_rays = d.frames ? (unsigned long long) rt.totalWidth * rt.totalHeight : 0;
	return s;
}
}  // namespace

HipHost::HipHost(const RayTracer &rt_, int) : rt(rt_) {
	no_device_if_asked();
	impl.reset(new ocrt::DeviceRenderer());
}
HipHost::HipHost(const RayTracer &rt_, int, unsigned int rank, unsigned int nranks) : rt(rt_) {
	if (nranks == 0 || rank >= nranks)
		throw std::invalid_argument("rank must be < nranks");
	no_device_if_asked();
	impl.reset(new ocrt::DeviceRenderer());
}
HipHost::~HipHost() = default;
void HipHost::upload(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes, const std::vector<Vec3f> &aabbs,
                     const std::vector<Vec3f> &vertices, const std::vector<Vec3f> &vnormals) {
	try {
		take(*impl, ocrt::pack_scene(faces, nodes, aabbs, vertices, vnormals));
	} catch (const std::exception &e) {  // (the reference's check(): message + exit, include/opencl_host.h:21-26)
		std::fprintf(stderr, "%s\n", e.what());
		std::exit(EXIT_FAILURE);
	}
}
void HipHost::upload(const ocrt::PackedScene &packed) { take(*impl, packed); }
bool HipHost::operator()() {
	fill(*impl, rt);
	return true;
}
void HipHost::download(float *image) { std::copy(impl->image.begin(), impl->image.end(), image); }
void HipHost::downloadResized(unsigned char *image) { rt.resize(impl->image.data(), image); }
void HipHost::printInfo() { std::printf("(no device: tests/hip_host_stub.cc)\n"); }
void HipHost::warmUp(int) {}
void HipHost::warmUp(const RayTracer &, int) {}
void HipHost::reserveScene(const RayTracer &, int, size_t) {}
float HipHost::lastKernelMs() const { return 0.0f; }
ocrt::RenderStats HipHost::lastStats() { return stats_of(*impl, rt); }

HipHostRing::HipHostRing(const RayTracer &rt_, unsigned int hosts, int) : rt(rt_), last_host(0) {
	if (hosts == 0 || hosts > 16)
		throw std::invalid_argument("a frame ring holds 1 to 16 renderers");
	no_device_if_asked();
	ring.reset(new ocrt::FrameRing());
	ring->hosts = hosts;
}
HipHostRing::~HipHostRing() = default;
void HipHostRing::upload(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes, const std::vector<Vec3f> &aabbs,
                         const std::vector<Vec3f> &vertices, const std::vector<Vec3f> &vnormals) {
	take(ring->host, ocrt::pack_scene(faces, nodes, aabbs, vertices, vnormals));
}
void HipHostRing::upload(const ocrt::PackedScene &packed) { take(ring->host, packed); }
bool HipHostRing::operator()() { return frames(1); }
bool HipHostRing::frames(unsigned int count) {
	for (unsigned int k = 0; k < count; ++k)
		fill(ring->host, rt);
	last_host = (last_host + count) % ring->hosts;
	return true;
}
void HipHostRing::download(float *image) { std::copy(ring->host.image.begin(), ring->host.image.end(), image); }
void HipHostRing::downloadResized(unsigned char *image) { rt.resize(ring->host.image.data(), image); }
float HipHostRing::lastKernelMs() const { return 0.0f; }
ocrt::RenderStats HipHostRing::lastStats() { return stats_of(ring->host, rt); }
unsigned int HipHostRing::size() const { return ring->hosts; }

HipHostGroup::HipHostGroup(const RayTracer &rt_, unsigned int devices, int, const char *) : rt(rt_), staging(nullptr), assembled(nullptr), staging_bytes(0) {
	if (devices == 0)
		throw std::invalid_argument("at least one device");
	no_device_if_asked();
	for (unsigned int k = 0; k < devices; ++k)
		hosts.emplace_back(new ocrt::DeviceRenderer());
}
HipHostGroup::~HipHostGroup() = default;
void HipHostGroup::upload(const std::vector<uint32_t> &faces, const std::vector<uint32_t> &nodes, const std::vector<Vec3f> &aabbs,
                          const std::vector<Vec3f> &vertices, const std::vector<Vec3f> &vnormals) {
	const ocrt::PackedScene packed = ocrt::pack_scene(faces, nodes, aabbs, vertices, vnormals);
	for (auto &h : hosts)
		take(*h, packed);
}
void HipHostGroup::upload(const ocrt::PackedScene &packed) {
	for (auto &h : hosts)
		take(*h, packed);
}
bool HipHostGroup::operator()() {
	fill(*hosts.front(), rt);
	return true;
}
void HipHostGroup::downloadResized(unsigned char *image) { rt.resize(hosts.front()->image.data(), image); }
float HipHostGroup::lastKernelMs() const { return 0.0f; }
ocrt::RenderStats HipHostGroup::lastStats() { return stats_of(*hosts.front(), rt); }
