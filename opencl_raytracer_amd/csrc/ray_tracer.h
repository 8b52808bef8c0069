// ray_tracer.h -- render options, supersampled image size and the host-side
// box filter.  Field-for-field the interface of reference
// include/ray_tracer.h:3-39 (Options, totalWidth/totalHeight, resize), so that
// a caller written against the reference compiles unchanged.
#pragma once
#include <cmath>
#include <cstdint>

#include "bvh.h"

class RayTracer {
	public:
		enum class AmbientOcclusionMethod { UNIFORM, RANDOM };
		struct Options {
			unsigned int width;          // output image width in pixels
			unsigned int height;         // output image height in pixels
			float focalLength;           // virtual camera focal length
			unsigned int nSuperSamples;  // samples per pixel; a floor(sqrt) x floor(sqrt) grid is cast
			bool enableShading;          // head-light Lambert term on/off
			bool enableAO;               // ambient occlusion on/off
			float aoMaxDistance;         // length limit of the occlusion rays
			unsigned int aoNumSamples;   // UNIFORM: number of rings; RANDOM: number of rays
			AmbientOcclusionMethod aoMethod;
			int aoAlphaMin;              // UNIFORM: lowest ring elevation in degrees
			int aoAlphaMax;              // UNIFORM: elevation span in degrees
			BVH::Method bvhMethod;
		};

		// The reference CLI's defaults (reference src/render.cc:17).
		static Options defaults() {
			return Options{ 600, 600, 1.f, 4, true, true, .2f, 3, AmbientOcclusionMethod::UNIFORM, 4, 90,
				        BVH::Method::CUT_LONGEST_AXIS };
		}
		// Side length of the supersample grid: (unsigned) sqrt(n), so 5 -> 2
		// (reference include/ray_tracer.h:33-34).
		static unsigned int gridSize(unsigned int nSuperSamples) {
			return (unsigned int) std::sqrt((double) nSuperSamples);
		}

		explicit RayTracer(Options opts)
			: options(opts)
			, totalWidth(opts.width * gridSize(opts.nSuperSamples))
			, totalHeight(opts.height * gridSize(opts.nSuperSamples)) {}

		// Box-averages the totalWidth x totalHeight float image `tmp` down to
		// width x height and quantises with (mean * 255) truncated to uint8
		// (reference src/ray_tracer.cc:3-16; row-major ssY, ssX summation order).
		void resize(const float *tmp, unsigned char *image) const;

		const Options options;
		const unsigned int totalWidth;
		const unsigned int totalHeight;
};
