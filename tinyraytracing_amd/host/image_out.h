// image_out.h — tonemap + PNG output, the role of imshow() (main.cpp:19-42) and
// svpng (svpng.inc:77-108): byte = (uchar) clamp(pow(x, 1/2.2f) * 255, 0, 255),
// RGB, rows top to bottom, written as an uncompressed (stored-deflate) PNG.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace trt {

// Linear radiance (double, as the reference accumulates it, main.cpp:74) -> 8-bit sRGB-ish bytes.
void tonemap(const double* linear_rgb, int width, int height, std::vector<uint8_t>& out);
void tonemap(const float* linear_rgb, int width, int height, std::vector<uint8_t>& out);

// Writes an 8-bit RGB PNG with stored (uncompressed) deflate blocks.  Returns false on I/O error.
bool writePNG(const std::string& path, int width, int height, const uint8_t* rgb);

// imshow(SRC, index, w, h) of the reference: writes <basedir>/image<index>.png.
bool imshow(const double* src, const std::string& basedir, const std::string& index, int img_width, int img_height);

}  // namespace trt
