#include "bmp_io.h"
#include <cstdio>
#include <cstring>

namespace rt {

static uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

bool read_bmp(const std::string& path, Image8& out) {
	out.width = out.height = 1;
	out.rgb.assign(3, 255);
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) return false;
	uint8_t hdr[54];
	bool ok = fread(hdr, 1, 54, f) == 54 && hdr[0] == 'B' && hdr[1] == 'M';
	if (!ok) { fclose(f); return false; }
	uint32_t offBits = rd32(hdr + 10);
	int32_t w = (int32_t)rd32(hdr + 18), h = (int32_t)rd32(hdr + 22);
	uint16_t bpp = rd16(hdr + 28);
	uint32_t compression = rd32(hdr + 30);
	if (w <= 0 || h <= 0 || compression != 0 || (bpp != 8 && bpp != 24 && bpp != 32)) { fclose(f); return false; }
	std::vector<uint8_t> palette;
	if (bpp == 8) {
		int ncol = ((int)offBits - 54) / 4;
		if (ncol > 256) ncol = 256;
		palette.assign(256 * 4, 255);
		if (ncol > 0 && fread(palette.data(), 4, ncol, f) != (size_t)ncol) { fclose(f); return false; }
	}
	if (fseek(f, (long)offBits, SEEK_SET) != 0) { fclose(f); return false; }
	size_t rowBytes = (((size_t)w * bpp + 31) / 32) * 4;
	std::vector<uint8_t> row(rowBytes);
	std::vector<uint8_t> px((size_t)w * h * 3, 255);
	for (int j = h - 1; j >= 0; j--) {   // file rows run bottom-up; the first row read is j = Height-1 (EasyBMP.cpp:936-971)
		if (fread(row.data(), 1, rowBytes, f) != rowBytes) break;   // EasyBMP keeps what it has read so far
		uint8_t* dst = px.data() + (size_t)j * w * 3;
		for (int i = 0; i < w; i++) {
			const uint8_t* s;
			if (bpp == 8) s = palette.data() + 4 * row[i];
			else s = row.data() + (size_t)i * (bpp / 8);
			dst[i * 3 + 0] = s[2]; dst[i * 3 + 1] = s[1]; dst[i * 3 + 2] = s[0];   // file order is B,G,R
		}
	}
	fclose(f);
	out.width = w; out.height = h; out.rgb.swap(px);
	return true;
}

bool write_bmp(const std::string& path, const uint8_t* rgb8, int W, int H) {
	if (W <= 0 || H <= 0) return false;
	FILE* f = fopen(path.c_str(), "wb");
	if (!f) return false;
	size_t rowBytes = (((size_t)W * 24 + 31) / 32) * 4;
	uint32_t dataSize = (uint32_t)(rowBytes * H);
	uint8_t hdr[54];
	memset(hdr, 0, sizeof hdr);
	auto wr32 = [&](int o, uint32_t v) { hdr[o] = v & 255; hdr[o + 1] = (v >> 8) & 255; hdr[o + 2] = (v >> 16) & 255; hdr[o + 3] = (v >> 24) & 255; };
	hdr[0] = 'B'; hdr[1] = 'M';
	wr32(2, 54 + dataSize); wr32(10, 54); wr32(14, 40); wr32(18, (uint32_t)W); wr32(22, (uint32_t)H);
	hdr[26] = 1; hdr[28] = 24;
	wr32(34, dataSize); wr32(38, 3780); wr32(42, 3780);   // EasyBMP's default 96 dpi
	bool ok = fwrite(hdr, 1, 54, f) == 54;
	std::vector<uint8_t> row(rowBytes, 0);
	for (int j = H - 1; j >= 0 && ok; j--) {
		const uint8_t* s = rgb8 + (size_t)j * W * 3;
		for (int i = 0; i < W; i++) { row[i * 3 + 0] = s[i * 3 + 2]; row[i * 3 + 1] = s[i * 3 + 1]; row[i * 3 + 2] = s[i * 3 + 0]; }
		ok = fwrite(row.data(), 1, rowBytes, f) == rowBytes;
	}
	fclose(f);
	return ok;
}

}  // namespace rt
