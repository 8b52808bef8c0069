#include "ray_tracer.h"

#include <cstddef>

void RayTracer::resize(const float *tmp, unsigned char *image) const {
	const unsigned int n = gridSize(options.nSuperSamples);
	const float samples = (float) (n * n);
	for (unsigned int y = 0; y < options.height; ++y) {
		const float *block_row = tmp + (size_t) y * n * totalWidth;
		for (unsigned int x = 0; x < options.width; ++x) {
			float total = 0;
			for (unsigned int sy = 0; sy < n; ++sy) {
				const float *row = block_row + (size_t) sy * totalWidth + (size_t) x * n;
				for (unsigned int sx = 0; sx < n; ++sx)
					total += row[sx];
			}
			image[(size_t) y * options.width + x] = (unsigned char) ((total / samples) * 255);
		}
	}
}
