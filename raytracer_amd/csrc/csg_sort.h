// csg_sort.h -- the order std::sort leaves an interval list in, restated so that the device sorts exactly as the reference does.
//
// csg_node::setIntervals sorts the list of an Intersection by entry t with std::sort (csg.cpp:199).  std::sort is not stable,
// so when two intervals have the same entry t their final order -- and with it which boundary's normal csgObject::rayhit returns --
// is whatever the library's algorithm happens to do.  The parity definition is the g++ / libstdc++ build of the reference
// (SURVEY.md section 8c), whose std::sort (bits/stl_algo.h) is: introsort -- median-of-three quicksort while a range holds more than
// 16 elements, heapsort once 2*floor(log2(n)) partitions did not get there -- followed by one insertion sort over everything.  For
// up to 16 elements that is a plain insertion sort, which is what every bundled and generated scene uses (a program of at most
// RT_MAX_CSG_LEAVES = 16 sphere / box leaves cannot build a longer list: DESIGN.md section 4); the rest of the algorithm is restated
// here for lists of 17 .. RTD_CSG_CAP elements so that this corner has no unpinned behaviour either.
//
// No HIP or libstdc++ dependency: tests/test_csg_sort.py compiles this header with g++ and compares it with std::sort itself.
#pragma once

#if defined(__HIPCC__)
#define RT_CSG_HD __host__ __device__ __forceinline__
#else
#define RT_CSG_HD inline
#endif

// `at(i)` returns a reference to element i; `less(x, y)` is the comparison (entry t).  T is the element type.
template <class T, class At, class Less>
RT_CSG_HD void csgAdjustHeap(At at, int first, int hole, int len, T value, Less less) {   // std::__adjust_heap + std::__push_heap
	const int top = hole;
	int second = hole;
	while (second < (len - 1) / 2) {
		second = 2 * (second + 1);
		if (less(at(first + second), at(first + second - 1))) second--;
		at(first + hole) = at(first + second);
		hole = second;
	}
	if ((len & 1) == 0 && second == (len - 2) / 2) {
		second = 2 * (second + 1);
		at(first + hole) = at(first + second - 1);
		hole = second - 1;
	}
	int parent = (hole - 1) / 2;
	while (hole > top && less(at(first + parent), value)) {
		at(first + hole) = at(first + parent);
		hole = parent;
		parent = (hole - 1) / 2;
	}
	at(first + hole) = value;
}

template <class T, class At, class Less>
RT_CSG_HD void csgSortStd(At at, int n, Less less) {
	if (n <= 1) return;
	// ---- std::__introsort_loop(first, last, 2 * __lg(n)) ----
	if (n > 16) {
		int lg = 0;
		for (int m = n; m > 1; m >>= 1) lg++;
		// the loop recurses into [cut, last) and goes on with [first, cut); the two ranges are disjoint, so the order in which
		// they are worked through does not matter and a work list stands in for the recursion
		int wf[40], wl[40], wd[40], sp = 0;
		wf[0] = 0; wl[0] = n; wd[0] = 2 * lg; sp = 1;
		while (sp > 0) {
			sp--;
			const int first = wf[sp];
			int last = wl[sp], depth = wd[sp];
			while (last - first > 16) {
				if (depth == 0) {
#ifdef RT_CSG_SORT_TEST_HOOK
				RT_CSG_SORT_TEST_HOOK;
#endif
				// std::__partial_sort(first, last, last): make_heap + sort_heap
					const int len = last - first;
					for (int parent = (len - 2) / 2;; parent--) {
						const T value = at(first + parent);
						csgAdjustHeap<T>(at, first, parent, len, value, less);
						if (parent == 0) break;
					}
					for (int end = last; end - first > 1;) {
						end--;
						const T value = at(end);
						at(end) = at(first);
						csgAdjustHeap<T>(at, first, 0, end - first, value, less);
					}
					break;
				}
				depth--;
				// std::__unguarded_partition_pivot: median of (first + 1, mid, last - 1) moved to first, then Hoare partition of [first + 1, last)
				const int mid = first + (last - first) / 2;
				const int a = first + 1, b = mid, c = last - 1;
				int med;
				if (less(at(a), at(b))) med = less(at(b), at(c)) ? b : less(at(a), at(c)) ? c : a;
				else med = less(at(a), at(c)) ? a : less(at(b), at(c)) ? c : b;
				{ const T tmp = at(first); at(first) = at(med); at(med) = tmp; }
				int lo = first + 1, hi = last;
				for (;;) {
					while (less(at(lo), at(first))) lo++;
					hi--;
					while (less(at(first), at(hi))) hi--;
					if (!(lo < hi)) break;
					const T tmp = at(lo); at(lo) = at(hi); at(hi) = tmp;
					lo++;
				}
				if (sp < 40) { wf[sp] = lo; wl[sp] = last; wd[sp] = depth; sp++; }
				last = lo;
			}
		}
	}
	// ---- std::__final_insertion_sort ----
	const int guarded = n > 16 ? 16 : n;
	for (int i = 1; i < guarded; i++) {   // std::__insertion_sort(first, first + 16)
		const T v = at(i);
		if (less(v, at(0))) {
			for (int j = i; j > 0; j--) at(j) = at(j - 1);
			at(0) = v;
		} else {
			int j = i;
			while (less(v, at(j - 1))) { at(j) = at(j - 1); j--; }
			if (j != i) at(j) = v;
		}
	}
	for (int i = guarded; i < n; i++) {   // std::__unguarded_insertion_sort(first + 16, last)
		const T v = at(i);
		int j = i;
		while (j > 0 && less(v, at(j - 1))) { at(j) = at(j - 1); j--; }   // (j > 0 never decides: the smallest element is among the first 16 by then)
		if (j != i) at(j) = v;
	}
}
