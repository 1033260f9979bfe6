// bmp_io.h -- BMP texture input and frame output for the host side.
// Replaces the two things the reference uses EasyBMP 1.06 for: BMP::ReadFromFile for texture / bump maps
// (texturemesh.cpp:4, bump.cpp:4) and -- new in this build, the reference only draws to a GL window --
// writing the rendered frame as a 24-bit BMP in the layout BMP::WriteToFile (EasyBMP.cpp:424) produces.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rt {
struct Image8 {
	int width = 1, height = 1;
	std::vector<uint8_t> rgb;   // 3 bytes per texel, row-major, row 0 = TOP row (EasyBMP's Pixels[i][j], j = 0 on top)
};
// Reads an uncompressed 8/24/32-bit BMP.  On any failure returns false and leaves a 1x1 white image, which is
// what EasyBMP's error path leaves behind (SetSize(1,1): EasyBMP.cpp:685-695,414-417).
bool read_bmp(const std::string& path, Image8& out);
// rgb8: W*H*3 bytes, row 0 = top row.
bool write_bmp(const std::string& path, const uint8_t* rgb8, int W, int H);
}
