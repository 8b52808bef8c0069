// render.cc -- `render [OPTIONS] INPUT_MESH OUTPUT_IMAGE`: the reference's CLI
// (reference src/render.cc:16-139, flag table :23-30) on the HIP render host.
// Same flags, defaults, phase lines and PGM format; `-h` is HEIGHT, help is
// `--help` only.  New: `--device N`, `--gpus N` (one frame on N GPUs of the node, bands gathered over RCCL / xGMI;
// `--gather rccl|peer|auto`), `--frames K` / `--in-flight H` (a steady stream of K frames through a ring of H render
// hosts on the one GPU: throughput instead of one blocking frame), `--host-resize` (the reference's own download +
// RayTracer::resize instead of the fused device resize), and a Mrays/s summary line.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "bvh.h"
#include "cli_support.h"
#include "hip_host.h"
#include "mesh.h"
#include "ray_tracer.h"
#include "scene_pack.h"

using ocrt::cli::Color;
using ocrt::cli::Info;

namespace {

struct OptionSpec {
	char short_name;
	const char *long_name;
	const char *help;
};

const OptionSpec OPTIONS[] = {
	{ 0, "help", "Print this dialogue." },
	{ 'w', "width", "Specifies the width to use for the output image." },
	{ 'h', "height", "Specifies the height to use for the output image." },
	{ 'a', "ambient-occlusion-samples",
	  "Specifies the number of samples used for ambient occlusion. If the value `0` is specified, ambient occlusion "
	  "will be disabled." },
	{ 'd', "ambient-occlusion-max-distance",
	  "Specifies the maximum distance that should be allowed for ambient occlusion rays." },
	{ 'm', "ambient-occlusion-method", "Specifies the method of ambient occlusion [uniform|random]." },
	{ 'f', "focal-length", "Specifies the focal length that the camera should use." },
	{ 's', "supersamples", "Specifies the number of supersamples to use." },
	{ 'r', "bvh-strategy", "Specifies the strategy of BVH construction (longest|sah)." },
	{ 0, "device", "Specifies the HIP device index to render on (default: $OCRT_DEVICE or 0)." },
	{ 0, "gpus", "Renders the frame on this many GPUs of the node (image bands dealt round-robin, gathered over xGMI)." },
	{ 0, "gather", "How --gpus N brings the image bands together [rccl|peer|auto]." },
	{ 0, "frames", "Renders the frame this many times in a row (the image written is the last one)." },
	{ 0, "in-flight", "Render hosts taking those frames in turn on the GPU (default 3 when --frames > 1, else 1)." },
	{ 0, "host-resize", "Downloads the float image and resizes it on the host, as the reference does [0|1]." },
	{ 0, "timings", "Prints one more line with the phases' wall-clock times in milliseconds [0|1]." },
	{ 0, "warm-up", "Brings up the HIP device on a second thread while the mesh is loaded and the BVH built [0|1] (default 1)." },
};

void usage(const char *argv0) {
	std::cout << "A HIP raytracer that renders triangle meshes in OFF format." << std::endl << std::endl;
	std::cout << "Usage: " << argv0 << " [OPTIONS] INPUT_MESH OUTPUT_IMAGE" << std::endl;
	size_t widest = 0;
	for (const OptionSpec &o : OPTIONS)
		widest = std::max(widest, std::strlen(o.long_name));
	for (const OptionSpec &o : OPTIONS) {
		std::cout << "  ";
		if (o.short_name)
			std::cout << '-' << o.short_name << ", ";
		else
			std::cout << "    ";
		std::string name = o.long_name;
		name.resize(widest + 2, ' ');
		std::cout << "--" << name << o.help << std::endl;
	}
}

[[noreturn]] void usage_error(const char *argv0, const std::string &message) {
	usage(argv0);
	std::cout << std::endl;
	std::cerr << "Error: " << message << std::endl;
	std::exit(EXIT_FAILURE);
}

struct CliOptions : RayTracer::Options {
	std::string in, out;
	int device = -1;
	unsigned int gpus = 1;
	unsigned int frames = 1, in_flight = 0;
	std::string gather = "auto";
	bool host_resize = false;
	bool timings = false, warm_up = true;

	CliOptions(int argc, const char **argv) : RayTracer::Options(RayTracer::defaults()) {
		std::vector<std::string> positional;
		for (int i = 1; i < argc; ++i) {
			const char *arg = argv[i];
			const OptionSpec *spec = nullptr;
			const char *value = nullptr;
			if (arg[0] == '-' && arg[1] == '-' && arg[2]) {
				// --name, --name=value, --name value
				const char *eq = std::strchr(arg + 2, '=');
				const std::string name = eq ? std::string(arg + 2, eq) : std::string(arg + 2);
				for (const OptionSpec &o : OPTIONS)
					if (name == o.long_name)
						spec = &o;
				if (!spec)
					usage_error(argv[0], std::string("Invalid option ") + (arg + 2));
				if (eq)
					value = eq + 1;
			} else if (arg[0] == '-' && arg[1] && arg[1] != '-') {
				// -x value, -xvalue
				for (const OptionSpec &o : OPTIONS)
					if (o.short_name && o.short_name == arg[1])
						spec = &o;
				if (!spec)
					usage_error(argv[0], std::string("Invalid option ") + arg[1]);
				if (arg[2])
					value = arg + 2;
			} else {
				positional.push_back(arg);
				if (positional.size() > 2)
					usage_error(argv[0], "Too much non-optional arguments");
				continue;
			}
			if (std::strcmp(spec->long_name, "help") == 0) {
				usage(argv[0]);
				std::exit(EXIT_SUCCESS);
			}
			if (!value) {
				if (i + 1 >= argc)
					usage_error(argv[0], "Too few non-optional arguments");
				value = argv[++i];
			}
			apply(argv[0], *spec, value);
			enableAO = aoNumSamples != 0;
		}
		if (positional.size() < 2)
			usage_error(argv[0], "Too few non-optional arguments");
		in = positional[0];
		out = positional[1];
	}

	private:
	void apply(const char *argv0, const OptionSpec &spec, const char *value) {
		const std::string name = spec.long_name;
		// Integers via atoi, floats via atof, as the reference's parser does
		// (reference include/args.h:47-54).
		if (name == "width")
			width = (unsigned int) std::atoi(value);
		else if (name == "height")
			height = (unsigned int) std::atoi(value);
		else if (name == "ambient-occlusion-samples")
			aoNumSamples = (unsigned int) std::atoi(value);
		else if (name == "ambient-occlusion-max-distance")
			aoMaxDistance = (float) std::atof(value);
		else if (name == "focal-length")
			focalLength = (float) std::atof(value);
		else if (name == "supersamples")
			nSuperSamples = (unsigned int) std::atoi(value);
		else if (name == "device")
			device = std::atoi(value);
		else if (name == "gpus")
			gpus = (unsigned int) std::atoi(value);
		else if (name == "frames")
			frames = (unsigned int) std::atoi(value);
		else if (name == "in-flight")
			in_flight = (unsigned int) std::atoi(value);
		else if (name == "gather")
			gather = value;
		else if (name == "host-resize")
			host_resize = std::atoi(value) != 0;
		else if (name == "timings")
			timings = std::atoi(value) != 0;
		else if (name == "warm-up")
			warm_up = std::atoi(value) != 0;
		else if (name == "ambient-occlusion-method") {
			if (std::strcmp(value, "uniform") == 0)
				aoMethod = RayTracer::AmbientOcclusionMethod::UNIFORM;
			else if (std::strcmp(value, "random") == 0)
				aoMethod = RayTracer::AmbientOcclusionMethod::RANDOM;
			else
				usage_error(argv0, "Invalid enum value");
		} else if (name == "bvh-strategy") {
			if (std::strcmp(value, "longest") == 0)
				bvhMethod = BVH::Method::CUT_LONGEST_AXIS;
			else if (std::strcmp(value, "sah") == 0)
				bvhMethod = BVH::Method::SURFACE_AREA_HEURISTIC;
			else
				usage_error(argv0, "Invalid enum value");
		}
	}
};

// Wall-clock phases for --timings (the reference's own phase lines count whole milliseconds).
struct PhaseClock {
	using clock = std::chrono::steady_clock;
	clock::time_point origin = clock::now(), last = origin;
	std::vector<std::pair<std::string, double>> phases;
	void mark(const std::string &name) {
		const clock::time_point now = clock::now();
		phases.emplace_back(name, std::chrono::duration<double, std::milli>(now - last).count());
		last = now;
	}
	void print() const {
		std::cout << "Timings (ms):";
		for (const auto &p : phases)
			std::printf(" %s %.2f,", p.first.c_str(), p.second);
		std::fflush(stdout);
		std::cout << " wall " << std::chrono::duration<double, std::milli>(clock::now() - origin).count() << std::endl;
	}
};
PhaseClock phase_clock;

void download_floats(HipHost &host, float *image);
void download_floats(HipHostRing &host, float *image);
void download_floats(HipHostGroup &host, float *image);
// `count` frames: one blocking operator()() after the other, or -- a ring -- a steady stream with frames in flight
template <class Host> bool run_frames(Host &host, unsigned int count) {
	bool ok = true;
	for (unsigned int k = 0; k < count && ok; ++k)
		ok = host();
	return ok;
}
bool run_frames(HipHostRing &host, unsigned int count) { return count == 1 ? host() : host.frames(count); }

// The frame itself, for one device (HipHost) or several (HipHostGroup): the phases of reference src/render.cc:84-128
// under their own names -- scripts that read the reference's output keep working: "Loading OpenCL kernel" is the
// face sort + scene upload there as well (:86-105), whatever its name says.
template <class Host>
void render_frame(Host &host, const CliOptions &options, const RayTracer &rt, const ocrt::PackedScene &packed,
                  std::vector<unsigned char> &image) {
	phase_clock.mark("host");
	std::size_t total_time = 0;
	total_time += Info::measure("Loading OpenCL kernel", [&] {
		host.upload(packed);  // (the face sort and the packing of the scene ran while the device came up, main())
		return true;
	}, true);
	phase_clock.mark("upload");
	std::cout << std::endl
	          << Color::BLUE << "<- " << Info::Palette::SECTION << "Rendering section" << Color::BLUE << " ->" << std::endl;
	const std::size_t render_time = Info::measure("Rendering image", [&] { return run_frames(host, options.frames); });
	total_time += render_time;
	phase_clock.mark("render");
	std::cout << std::endl;
	if (options.host_resize) {
		// the reference's flow: float image to the host (:114-117), box filter there (:120-123)
		std::vector<float> tmp((size_t) rt.totalWidth * rt.totalHeight);
		Info::measure("Loading memory", [&] {
			download_floats(host, tmp.data());
			return true;
		});
		total_time += Info::measure("Resizing image on host", [&] {
			rt.resize(tmp.data(), image.data());
			return true;
		});
	} else {
		total_time += Info::measure("Resizing image on device", [&] {
			host.downloadResized(image.data());  // (includes "Loading memory": width x height bytes)
			return true;
		});
	}
	phase_clock.mark("resize+download");
	const ocrt::RenderStats stats = host.lastStats();
	const double rays = (double) stats.primary_rays + (double) stats.ao_rays;
	std::cout << Info::Palette::NORMAL << "Rays: " << Info::Palette::HIGHLIGHT << stats.primary_rays
	          << Info::Palette::NORMAL << " primary + " << Info::Palette::HIGHLIGHT << stats.ao_rays
	          << Info::Palette::NORMAL << " ambient occlusion; kernel " << Info::Palette::HIGHLIGHT
	          << host.lastKernelMs() << " ms" << Info::Palette::NORMAL << " = " << Info::Palette::HIGHLIGHT
	          << (host.lastKernelMs() > 0 ? rays / (host.lastKernelMs() * 1e3) : 0.0) << " Mrays/s" << Color::RESET
	          << std::endl;
	if (options.frames > 1)  // (Info::measure counts milliseconds)
		std::cout << Info::Palette::NORMAL << "Frames: " << Info::Palette::HIGHLIGHT << options.frames << Info::Palette::NORMAL
		          << " in " << Info::Palette::HIGHLIGHT << render_time << " ms" << Info::Palette::NORMAL << " = "
		          << Info::Palette::HIGHLIGHT << (render_time ? rays * options.frames / (render_time * 1e3) : 0.0)
		          << " Mrays/s" << Info::Palette::NORMAL << " over the whole stream" << Color::RESET << std::endl;
	std::cout << Info::Palette::NORMAL << "Total time (without loading memory and building the BVH): "
	          << Info::formatTime(total_time) << std::endl;
}

void download_floats(HipHost &host, float *image) { host.download(image); }
void download_floats(HipHostRing &host, float *image) { host.download(image); }
void download_floats(HipHostGroup &, float *) {
	std::cerr << Info::Palette::WARNING << "--host-resize needs a single device" << Color::RESET << std::endl;
	std::exit(EXIT_FAILURE);
}

}  // namespace

int main(int argc, const char **argv) {
	CliOptions options(argc, argv);
	// (--gpus N: RCCL's peer-to-peer set-up shares device memory between its devices by dmabuf; the hosts' driver refuses
	// the legacy IPC mode with `hipIpcGetMemHandle: invalid argument`.  Set before the HIP runtime starts, not overriding a caller's choice.)
	if (options.gpus > 1)
		setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
	// The HIP runtime, the device context and the kernels' code object cost 100-200 ms and do not depend on the scene:
	// they come up on a second thread while this one reads the mesh and builds the BVH (reference order of the output
	// kept: nothing is printed from that thread).
	if (options.gpus == 0 || options.frames == 0 || options.in_flight > 16)  // (before anything is started)
		usage_error(argv[0], "--gpus and --frames must be positive, --in-flight at most 16");
	const RayTracer rt(options);
	const bool plain_host = options.gpus == 1 && options.gather == "auto" && options.frames == 1 && options.in_flight <= 1;
	// How many triangles are coming is in the file's header: the warm-up thread can then also make the scene's device
	// allocation (a peek of a few bytes; whatever is wrong with the file is reported by the loader proper, below).
	size_t triangles_announced = 0;
	if (std::FILE *peek = std::fopen(options.in.c_str(), "rb")) {
		char magic[8] = { 0 };
		unsigned long v = 0, f = 0;
		if (std::fscanf(peek, "%7s %lu %lu", magic, &v, &f) == 3 && std::string(magic) == "OFF" && f < (1ul << 25))
			triangles_announced = f;
		std::fclose(peek);
	}
	std::thread warm_up;
	if (options.warm_up)
		warm_up = std::thread([&options, &rt, plain_host, triangles_announced] {
			if (plain_host)
				HipHost::warmUp(rt, options.device);  // (... and the host's own buffers: main() adopts them below)
			else
				HipHost::warmUp(options.device);
			if (triangles_announced && options.gpus == 1)
				HipHost::reserveScene(rt, options.device, triangles_announced);
		});
	phase_clock.mark("options");
	std::cout << Color::BLUE << "<- " << Info::Palette::SECTION << "BVH section" << Color::BLUE << " ->" << std::endl;
	std::cout << Info::Palette::NORMAL << "Reading input mesh\xE2\x80\xA6" << std::endl;
	Mesh mesh;
	try {
		load_off_mesh(options.in, &mesh);
		phase_clock.mark("load");
		compute_vertex_normals(&mesh);
		phase_clock.mark("normals");
	} catch (...) {
		// The loader throws like the reference's (src/mesh.cc:9,14,22,62: nobody catches, the process aborts with the
		// message).  Same here -- but only once the warm-up thread is through: aborting while it is inside the HIP
		// runtime's initialisation, with a joinable std::thread alive, would be a second failure on top of the first.
		if (warm_up.joinable())
			warm_up.join();
		throw;
	}
	std::cout << Color::BLUE << "- " << Info::Palette::NORMAL << "Vertices: " << Info::Palette::HIGHLIGHT
	          << mesh.vertices.size() << std::endl
	          << Color::BLUE << "- " << Info::Palette::NORMAL << "Triangles: " << Info::Palette::HIGHLIGHT
	          << (mesh.faces.size() / 3) << Color::RESET << std::endl;
	if (options.enableAO && options.aoMethod == RayTracer::AmbientOcclusionMethod::UNIFORM) {
		// the reference's notice and its own ray estimate (:63-74; it prints 25 for the default three rings, the
		// kernel casts 28: the inner loop there runs to ray_count inclusive, src/intersect_kernel.cl:242)
		unsigned int rays = 0;
		const float degrees = (float) (M_PI / 180);
		for (unsigned int ring = 0; ring < options.aoNumSamples; ++ring) {
			const float step = (options.aoAlphaMax * degrees) / options.aoNumSamples;
			const float elevation = (step * ring) + (options.aoAlphaMin * degrees);
			rays += (unsigned int) (2.0f * M_PI * std::cos(elevation) / step);
		}
		std::cout << Info::Palette::WARNING
		          << "IMPORTANT INFO: You've enabled 'Uniform AO hemispheres'. You have entered a circle count of "
		          << options.aoNumSamples << ". This will result in " << rays
		          << " rays. Note that the Uniform AO Hemisphere will generate much better pictures without noise with less "
		             "rays and time than you would need using randomized hemispheres."
		          << Color::RESET << std::endl;
	}
	BVH bvh(options.bvhMethod);
	try {
		Info::measure("Building BVH", [&] {
			bvh.buildBVH(mesh);
			return true;
		});
	} catch (...) {
		if (warm_up.joinable())
			warm_up.join();
		throw;
	}
	phase_clock.mark("bvh");
	// The CPU half of the upload -- faces into leaf order (reference src/render.cc:88-95), validation, device records,
	// the walk tree -- needs no device: it runs here, beside the warm-up thread.  Malformed arrays end the run as they
	// would inside upload(): message + exit(EXIT_FAILURE) (the reference's check(), include/opencl_host.h:21-26).
	ocrt::PackedScene packed;
	try {
		std::vector<uint32_t> sorted_faces = sort_faces_by_leaf_order(mesh, bvh);
		mesh.faces.clear();
		bvh.triangles.clear();
		packed = ocrt::pack_scene(sorted_faces, bvh.nodes, bvh.aabbs, mesh.vertices, mesh.vnormals);
		// (one frame, the reference's use: nothing that only pays over a stream of frames -- scene_pack.h, make_walk_array)
		ocrt::prepare_walk_array(packed, options.enableAO && options.aoNumSamples > 0 ? ocrt::kernel_float(options.aoMaxDistance) : 0.0f,
		                         options.frames >= 16);
	} catch (const std::exception &e) {
		if (warm_up.joinable())
			warm_up.join();
		std::cerr << "HIP error: " << e.what() << std::endl;
		std::exit(EXIT_FAILURE);
	}
	phase_clock.mark("pack");
	if (warm_up.joinable())
		warm_up.join();
	phase_clock.mark("wait for the device");
	std::cout << std::endl
	          << Color::BLUE << "<- " << Info::Palette::SECTION << "Device section" << Color::BLUE << " ->" << std::endl;
	HipHost::printInfo();
	phase_clock.mark("device table");
	std::vector<unsigned char> image((size_t) options.width * options.height);
	const unsigned int in_flight = options.in_flight ? options.in_flight : options.frames > 1 ? 3u : 1u;
	if (options.gpus > 1 || options.gather != "auto") {
		HipHostGroup host(rt, options.gpus, options.device, options.gather.c_str());
		render_frame(host, options, rt, packed, image);
	} else if (in_flight > 1 || options.frames > 1) {
		HipHostRing host(rt, in_flight, options.device);
		render_frame(host, options, rt, packed, image);
	} else {
		HipHost host(rt, options.device);
		render_frame(host, options, rt, packed, image);
	}
	std::FILE *out = std::fopen(options.out.c_str(), "wb");
	if (!out) {
		std::cerr << Info::Palette::WARNING << "Error opening output file!" << Color::RESET << std::endl;
		std::exit(EXIT_FAILURE);
	}
	std::fprintf(out, "P5 %u %u 255\n", options.width, options.height);
	std::fwrite(image.data(), 1, image.size(), out);
	std::fclose(out);
	phase_clock.mark("write");
	if (options.timings)
		phase_clock.print();
	return 0;
}
