// tests/csg_sort_check.cpp -- csgSortStd (raytracer_amd/csrc/csg_sort.h, what the device sorts Intersection lists with) against
// libstdc++'s std::sort, the function the compiled reference calls (csg.cpp:199): the same comparator, lists with many equal keys,
// every length up to 64.  The elements carry a tag the comparator does not see, so any difference in how ties are left shows.
// Built and run by tests/test_csg_sort.py with g++; test infrastructure.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
static long g_heapsorts = 0;
#define RT_CSG_SORT_TEST_HOOK g_heapsorts++
#include "csg_sort.h"

struct Iv { float t; int tag; };
static bool lessIv(const Iv& x, const Iv& y) { return x.t < y.t; }

static unsigned long long rngState = 88172645463325252ull;
static unsigned rnd() { rngState ^= rngState << 13; rngState ^= rngState >> 7; rngState ^= rngState << 17; return (unsigned)(rngState >> 11); }

static bool check(std::vector<Iv> v) {
	std::vector<Iv> a = v, b = v;
	std::sort(a.begin(), a.end(), lessIv);
	csgSortStd<Iv>([&](int i) -> Iv& { return b[(size_t)i]; }, (int)b.size(), lessIv);
	for (size_t i = 0; i < v.size(); i++)
		if (a[i].tag != b[i].tag || a[i].t != b[i].t) return false;
	return true;
}

int main() {
	long cases = 0, ties17 = 0;
	for (int n = 0; n <= 64; n++) {
		for (int rep = 0; rep < 4000; rep++) {
			const int keys = 1 + (int)(rnd() % (unsigned)(rep % 7 == 0 ? 2 : rep % 3 == 0 ? 4 : n + 1));   // few distinct keys: many ties
			std::vector<Iv> v((size_t)n);
			for (int i = 0; i < n; i++) v[(size_t)i] = { (float)(rnd() % (unsigned)keys) * 0.25f, i };
			const int shape = rep % 5;
			if (shape == 1) std::sort(v.begin(), v.end(), lessIv);                                   // already sorted
			if (shape == 2) { std::sort(v.begin(), v.end(), lessIv); std::reverse(v.begin(), v.end()); }
			if (shape == 3) for (int i = 0; i < n; i++) v[(size_t)i].t = (float)((i % 2) ? i : n - i);   // organ pipes
			if (!check(v)) { printf("MISMATCH n=%d rep=%d\n", n, rep); return 1; }
			cases++;
			if (n >= 17 && keys < n) ties17++;
		}
	}
	// median-of-three killers (Musser): drive the quicksort into its depth limit, i.e. into the heapsort branch
	for (int n = 18; n <= 64; n += 2) {
		std::vector<Iv> v((size_t)n);
		const int k = n / 2;
		for (int i = 1; i <= k; i++) {
			if (i % 2) { v[(size_t)i - 1] = { (float)i, i }; v[(size_t)i] = { (float)(k + i), k + i }; }
			v[(size_t)(k + i - 1)] = { (float)(2 * i), 1000 + i };
		}
		if (!check(v)) { printf("MISMATCH killer n=%d\n", n); return 1; }
		cases++;
	}
	// McIlroy's adversary ("A killer adversary for quicksort", 1999) run against std::sort itself: it fixes the keys while the sort
	// compares them so that every pivot turns out to be among the smallest, which takes the quicksort to its depth limit
	for (int n = 17; n <= 64; n++) {
		std::vector<int> val((size_t)n, n /* gas */), idx((size_t)n);
		int solid = 0, candidate = 0;
		for (int i = 0; i < n; i++) idx[(size_t)i] = i;
		std::sort(idx.begin(), idx.end(), [&](int x, int y) {
			if (val[(size_t)x] == n && val[(size_t)y] == n) { if (x == candidate) val[(size_t)x] = solid++; else val[(size_t)y] = solid++; }
			if (val[(size_t)x] == n) candidate = x; else if (val[(size_t)y] == n) candidate = y;
			return val[(size_t)x] < val[(size_t)y];
		});
		std::vector<Iv> v((size_t)n);
		for (int i = 0; i < n; i++) v[(size_t)i] = { (float)(val[(size_t)i] / 2), i };   // halved: ties on top
		const long before = g_heapsorts;
		if (!check(v)) { printf("MISMATCH adversary n=%d\n", n); return 1; }
		for (int i = 0; i < n; i++) v[(size_t)i].t = (float)val[(size_t)i];
		if (!check(v)) { printf("MISMATCH adversary n=%d\n", n); return 1; }
		if (n >= 40 && g_heapsorts == before) { printf("the adversary did not reach the heapsort branch at n=%d\n", n); return 1; }
		cases += 2;
	}
	printf("ok cases=%ld with_ties_17plus=%ld heapsorts=%ld\n", cases, ties17, g_heapsorts);
	return 0;
}
