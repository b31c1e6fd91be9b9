// Post-processing of the finished frame on the device: Postprocessor.hpp under HDR_BLOOM
// (global.hpp:32) -- emissive extraction, separable Gaussian (KERNELSIZE 10, STDDEV 30), + original, exposure tone map
// -- and PPMGenerator::writePixel's gamma quantisation.  One thread per pixel, every kernel is a plain streaming
// pass over width*height*12 B; arithmetic order as in the reference, so everything but exp / pow is bit-exact.
//
// Every read of the reference goes through Texture::getRGBat(clamp(0, 0.999, x / w), clamp(0, 0.999, y / h))
// (Postprocessor.hpp:88-90, 140-141, 190-191): u = 0 takes getRGBat's "not positive" branch and becomes 1, so column 0
// reads the texel one row down and row 0 reads past the image, which the index clamp turns into the last texel [sic].
#pragma once
#include "device_math.h"

namespace tutu {

#define TUTU_POST_KERNEL 10  // KERNELSIZE, Postprocessor.hpp:12

// global.hpp:50-53
TUTU_DEV float post_clamp(float lo, float hi, float v) { return std_max(lo, std_min(hi, v)); }

// Texture::getRGBat (Texture.hpp:18-39) on a packed float3 image
TUTU_DEV V3 frame_rgb(const float* img, int width, int height, float u, float v) {
	if (width == 0 && height == 0) return mk1(0.f);
	if (u > 0) u = u - (float)(int)u;
	else u = 1 - (fabsf(u) - (float)(int)fabsf(u));
	if (v > 0) v = v - (float)(int)v;
	else v = 1 - (fabsf(v) - (float)(int)fabsf(v));
	const int x = (int)(u * (float)width);
	const int y = (int)(v * (float)height);
	int index = y * width + x;
	if (index < 0) index = 0;
	if (index >= width * height) index = width * height - 1;
	return mk(img[3 * (size_t)index], img[3 * (size_t)index + 1], img[3 * (size_t)index + 2]);
}

TUTU_DEV void post_store(float* img, size_t i, V3 c) {
	img[3 * i] = c.x;
	img[3 * i + 1] = c.y;
	img[3 * i + 2] = c.z;
}

// getEmmisiveTexture, Postprocessor.hpp:128-155: pixels brighter than |c| > 3 rescaled to a maximum channel of STRENGTH = 2
__global__ void __launch_bounds__(256) k_post_emissive(const float* src, float* dst, int w, int h) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= w * h) return;
	const int x = i % w, y = i / w;
	const float U = (float)x / w, V = (float)y / h;
	const V3 col = frame_rgb(src, w, h, post_clamp(0.f, 0.999f, U), post_clamp(0.f, 0.999f, V));
	V3 color = mk1(0.f);
	if (norm(col) > 3.f) {
		float mx = col.x > col.y ? col.x : col.y;
		mx = mx > col.z ? mx : col.z;
		// rescale(input, originMax = mx, originMin = 0, targetMax = 2, targetMin = 0), global.hpp:66-68
		color.x = 0.f + ((2.f - 0.f) * (col.x - 0.f) / (mx - 0.f));
		color.y = 0.f + ((2.f - 0.f) * (col.y - 0.f) / (mx - 0.f));
		color.z = 0.f + ((2.f - 0.f) * (col.z - 0.f) / (mx - 0.f));
	}
	post_store(dst, (size_t)i, color);
}

struct PostWeights {  // the ten weights of the `gaussian` lambda (Postprocessor.hpp:73-75), evaluated on the host with the same powf
	float g[TUTU_POST_KERNEL];
	float sum;  // kernelSum, accumulated in loop order
};

// one direction of getGaussianBlurTexture, Postprocessor.hpp:78-122
template <bool VERTICAL>
__global__ void __launch_bounds__(256) k_post_blur(const float* src, float* dst, int w, int h, PostWeights pw) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= w * h) return;
	const int x = i % w, y = i / w;
	const int start = -(TUTU_POST_KERNEL / 2);  // int startY = -kernelSize * 0.5
	V3 col = mk1(0.f);
#pragma unroll
	for (int k = 0; k < TUTU_POST_KERNEL; k++) {
		const float U = VERTICAL ? (float)x / w : (float)(x + k + start) / w;
		const float V = VERTICAL ? (float)(y + k + start) / h : (float)y / h;
		col = col + frame_rgb(src, w, h, post_clamp(0.f, 0.999f, U), post_clamp(0.f, 0.999f, V)) * pw.g[k];
	}
	post_store(dst, (size_t)i, col / pw.sum);
}

// add, Postprocessor.hpp:157-172
__global__ void __launch_bounds__(256) k_post_add(const float* a, const float* b, float* dst, int n3) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n3) dst[i] = a[i] + b[i];
}

// getHDRtexture, Postprocessor.hpp:178-201: 1 - exp(-c * EXPOSURE).  exp in double, rounded once: equal to the host's expf
// in all but a few 1e-4 of the cases (both are then the correctly rounded value)
__global__ void __launch_bounds__(256) k_post_hdr(const float* src, float* dst, int w, int h) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= w * h) return;
	const int x = i % w, y = i / w;
	const float U = (float)x / w, V = (float)y / h;
	const V3 c = frame_rgb(src, w, h, post_clamp(0.f, 0.999f, U), post_clamp(0.f, 0.999f, V));
	V3 m;
	m.x = 1 - (float)exp((double)(-c.x * 1.5f));
	m.y = 1 - (float)exp((double)(-c.y * 1.5f));
	m.z = 1 - (float)exp((double)(-c.z * 1.5f));
	post_store(dst, (size_t)i, m);
}

// PPMGenerator::writePixel, PPMGenerator.hpp:825-843 with GAMMA_COORECTION: (int)(255 * pow(clamp(0, 1, c), 0.78f))
__global__ void __launch_bounds__(256) k_post_quantise(const float* c, int32_t* out, uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float cl = post_clamp(0.f, 1.f, c[i]);
	const float p = (float)pow((double)cl, (double)0.78f);
	out[i] = (int32_t)(255 * p);
}

}  // namespace tutu
