// png_writer.h -- minimal PNG encoder (RGBA8, zlib "stored" blocks) standing in
// for the reference's vendored lodepng::encode(filename, image, w, h)
// (OptiXRenderer/utils.cpp:7).  Output is a valid PNG that any decoder reads
// back to the exact input pixels; it is not size-optimised.
#pragma once
#include <cstdint>
#include <string>

bool png_write_rgba8(const std::string& filename, const unsigned char* rgba, uint32_t width, uint32_t height, std::string* error);
