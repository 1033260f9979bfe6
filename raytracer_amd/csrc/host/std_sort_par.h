// std_sort_par.h -- the permutation libstdc++'s std::sort produces, computed on several threads.
//
// The reference sorts a BVH node's objects by centroid with std::sort (bvh.cpp:188-213).  std::sort is not stable: where
// centroids are equal -- whole rows of a grid mesh -- the order it leaves them in, and with it which object lands in which
// leaf and the DFS order that decides ties between equal-t hits, is whatever the library's algorithm does.  The parity
// definition is the g++ / libstdc++ build of the reference, whose std::sort (bits/stl_algo.h) is
//   __introsort_loop:        while a range holds more than 16 elements: median of (first + 1, middle, last - 1) to the front,
//                            Hoare partition of the rest around it, recurse into the right part, go on with the left one;
//                            after 2 * floor(log2(n)) levels: heapsort of the range (std::partial_sort(first, last, last))
//   __final_insertion_sort:  one insertion sort over everything
// The two parts of a partition are disjoint and what happens in one does not depend on the other, so they can be sorted
// at the same time; and the final insertion sort never carries an element across a partition boundary (left <= pivot <=
// right), so it is the same as an insertion sort of every range the loop left alone, done where the loop leaves it.
// Same comparisons, same swaps, same result as std::sort -- tests/test_std_sort_par.py compares the two on random keys,
// keys with long runs of ties, and an input that drives std::sort into its heapsort branch.
//
// The comparison must be a strict weak ordering (as std::sort demands: keys without NaNs).
#pragma once
#include <algorithm>
#include <future>
#include <utility>
#include <vector>

namespace rt {

template <class T, class Less>
class StdSortPar {
public:
	// ranges of more than `grain` elements are handed to a thread of their own
	StdSortPar(T* data, Less less, long grain) : a(data), less(less), grain(grain < 32 ? 32 : grain) {}
	void sort(long n) {
		if (n <= 1) return;
		int lg = 0;
		for (long m = n; m > 1; m >>= 1) lg++;   // std::__lg
		loop(0, n, 2 * lg);
	}

private:
	T* a;
	Less less;
	long grain;

	void insertion(long first, long last) {   // std::__insertion_sort
		for (long i = first + 1; i < last; i++) {
			T v = std::move(a[i]);
			if (less(v, a[first])) {
				std::move_backward(a + first, a + i, a + i + 1);
				a[first] = std::move(v);
			} else {   // std::__unguarded_linear_insert
				long j = i;
				while (less(v, a[j - 1])) { a[j] = std::move(a[j - 1]); j--; }
				a[j] = std::move(v);
			}
		}
	}
	void loop(long first, long last, int depth) {   // std::__introsort_loop, then this range's share of the final insertion sort
		std::vector<std::future<void>> others;
		bool heaped = false;
		while (last - first > 16) {
			if (depth == 0) {
#ifdef RT_STD_SORT_PAR_TEST_HOOK
				RT_STD_SORT_PAR_TEST_HOOK;
#endif
				std::partial_sort(a + first, a + last, a + last, less);   // (the very call std::sort makes: __heap_select + __sort_heap)
				heaped = true;
				break;
			}
			depth--;
			// std::__unguarded_partition_pivot: std::__move_median_to_first(first, first + 1, mid, last - 1), then std::__unguarded_partition(first + 1, last, first)
			const long mid = first + (last - first) / 2;
			const long x = first + 1, y = mid, z = last - 1;
			long med;
			if (less(a[x], a[y])) med = less(a[y], a[z]) ? y : less(a[x], a[z]) ? z : x;
			else med = less(a[x], a[z]) ? x : less(a[y], a[z]) ? z : y;
			std::swap(a[first], a[med]);
			long lo = first + 1, hi = last;
			for (;;) {
				while (less(a[lo], a[first])) lo++;
				hi--;
				while (less(a[first], a[hi])) hi--;
				if (!(lo < hi)) break;
				std::swap(a[lo], a[hi]);
				lo++;
			}
			if (last - lo > grain && lo - first > grain) others.push_back(std::async(std::launch::async, [this, lo, last, depth] { loop(lo, last, depth); }));
			else loop(lo, last, depth);
			last = lo;
		}
		if (!heaped) insertion(first, last);   // (a heapsorted range is in order: the insertion sort would leave it as it is)
		for (auto& o : others) o.get();
	}
};

// sorts data[0, n) exactly as std::sort(data, data + n, less) would
template <class T, class Less>
void stdSortParallel(T* data, long n, Less less, long grain = 1 << 16) {
	StdSortPar<T, Less>(data, less, grain).sort(n);
}

}  // namespace rt
