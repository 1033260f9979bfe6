// Compares rt::stdSortParallel (raytracer_amd/csrc/host/std_sort_par.h) with libstdc++'s std::sort itself, element for element,
// on keys with many ties (the order among equal keys is the point).  Built and run by tests/test_std_sort_par.py.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
static std::atomic<long> g_heapsorts{0};
#define RT_STD_SORT_PAR_TEST_HOOK g_heapsorts++
#include "host/std_sort_par.h"

struct KeyIdx { float key; int idx; };
static bool check(const std::vector<KeyIdx>& v, long grain) {
	std::vector<KeyIdx> a = v, b = v;
	auto less = [](const KeyIdx& x, const KeyIdx& y) { return x.key < y.key; };
	std::sort(a.begin(), a.end(), less);
	rt::stdSortParallel(b.data(), (long)b.size(), less, grain);
	for (size_t i = 0; i < a.size(); i++) if (a[i].idx != b[i].idx || a[i].key != b[i].key) return false;
	return true;
}
int main() {
	std::mt19937 rng(12345);
	long cases = 0, elements = 0;
	const int sizes[] = { 0, 1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 64, 100, 257, 1000, 4099, 20000, 100003, 1500000 };
	for (int n : sizes) {
		for (int pattern = 0; pattern < 8; pattern++) {
			const int reps = n <= 1000 ? 200 : n <= 20000 ? 10 : 1;
			for (int r = 0; r < reps; r++) {
				std::vector<KeyIdx> v((size_t)n);
				for (int i = 0; i < n; i++) {
					float k;
					switch (pattern) {
					case 0: k = (float)(rng() % 1000003u); break;                       // mostly distinct
					case 1: k = (float)(rng() % 7u); break;                             // few distinct keys
					case 2: k = 1.0f; break;                                            // all equal
					case 3: k = (float)(i / 3); break;                                  // sorted, runs of ties
					case 4: k = (float)((n - i) / 5); break;                            // reversed, runs of ties
					case 5: k = (float)(i % 1581) * 0.25f; break;                       // a grid mesh's centroids along one axis
					case 6: k = (float)((i * 7919) % (n / 4 + 1)); break;               // scattered, every key about four times
					default: k = (float)std::min(i, n - 1 - i); break;                  // organ pipe
					}
					v[(size_t)i] = { k, i };
				}
				const long grain = n <= 1000 ? 32 : n <= 100003 ? 512 : 1 << 14;
				if (!check(v, grain)) { printf("MISMATCH n=%d pattern=%d rep=%d\n", n, pattern, r); return 1; }
				cases++; elements += n;
			}
		}
	}
	// McIlroy's adversary ("A killer adversary for quicksort", 1999) run against std::sort itself: it fixes the keys while the sort
	// compares them so that every pivot turns out to be among the smallest, which takes the quicksort to its depth limit
	for (int n : { 40, 100, 1000, 30000 }) {
		std::vector<int> val((size_t)n, n /* gas */), idx((size_t)n);
		int solid = 0, candidate = 0;
		for (int i = 0; i < n; i++) idx[(size_t)i] = i;
		std::sort(idx.begin(), idx.end(), [&](int x, int y) {
			if (val[(size_t)x] == n && val[(size_t)y] == n) { if (x == candidate) val[(size_t)x] = solid++; else val[(size_t)y] = solid++; }
			if (val[(size_t)x] == n) candidate = x; else if (val[(size_t)y] == n) candidate = y;
			return val[(size_t)x] < val[(size_t)y];
		});
		std::vector<KeyIdx> v((size_t)n);
		const long before = g_heapsorts;
		for (int i = 0; i < n; i++) v[(size_t)i] = { (float)val[(size_t)i], i };
		if (!check(v, 32)) { printf("MISMATCH adversary n=%d\n", n); return 1; }
		for (int i = 0; i < n; i++) v[(size_t)i].key = (float)(val[(size_t)i] / 2);   // halved: ties on top
		if (!check(v, 32)) { printf("MISMATCH adversary with ties n=%d\n", n); return 1; }
		if (g_heapsorts == before) { printf("the adversary did not reach the heapsort branch at n=%d\n", n); return 1; }
		cases += 2;
	}
	printf("ok cases=%ld elements=%ld heapsorts=%ld\n", cases, elements, g_heapsorts.load());
	return 0;
}
