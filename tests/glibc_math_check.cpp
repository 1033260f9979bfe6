// tests/glibc_math_check.cpp -- rt_atan2f / rt_asinf (raytracer_amd/csrc/glibc_math.h, what the device evaluates bump-map coordinates
// with) against the host's libm (glibc: the library the compiled reference calls, bump.cpp:14-15), bit for bit.  Test infrastructure.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "glibc_math.h"
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(int argc, char** argv) {
	const long total = argc > 1 ? atol(argv[1]) : 20000000;
	unsigned long long s = 88172645463325252ull;
	auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
	long bad1 = 0, bad2 = 0, bad3 = 0;
	for (long i = 0; i < total; i++) {
		float a = (float)((double)(rnd() >> 11) / 9007199254740992.0 * 2 - 1), b = (float)((double)(rnd() >> 11) / 9007199254740992.0 * 2 - 1);
		if (i % 7 == 0) a *= 1e-3f;
		if (i % 11 == 0) b *= 1e-4f;
		if (i % 100003 == 0) a = 0;
		if (i % 100019 == 0) b = 0;
		if (i % 100043 == 0) b = 1;
		if (bits(atan2f(a, b)) != bits(rt_atan2f(a, b))) { if (bad1 < 5) printf("atan2f(%a, %a): %a vs %a\n", a, b, atan2f(a, b), rt_atan2f(a, b)); bad1++; }
		if (bits(asinf(a)) != bits(rt_asinf(a))) bad2++;
	}
	for (uint32_t u = 0; u <= 0x3f800000u; u++) {   // every float in [-1, 1]
		float x; memcpy(&x, &u, 4);
		if (bits(asinf(x)) != bits(rt_asinf(x)) || bits(asinf(-x)) != bits(rt_asinf(-x))) bad3++;
	}
	printf("n=%ld atan2f_mismatches=%ld asinf_mismatches=%ld asinf_exhaustive_mismatches=%ld\n", total, bad1, bad2, bad3);
	return (bad1 || bad2 || bad3) ? 1 : 0;
}
