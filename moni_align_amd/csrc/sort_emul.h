// std::sort of libstdc++ (bits/stl_algo.h, bits/stl_heap.h), reproduced move for move so that device code orders
// elements exactly as the reference's host code does.  The reference sorts with comparators that leave ties (anchors by
// reference end: chain.hpp:246; chains by score: chain.hpp:402), std::sort is not stable, and the SAM output depends on
// which of the tied elements comes first — so "a sort" is not enough, it has to be this one:
//   introsort (depth limit 2*floor(log2 n), median-of-3 to first, unguarded Hoare partition, recurse right / loop left,
//   heap-sort fallback) down to 16-element pieces, then one insertion sort (guarded for the first 16, unguarded after).
// Usable from host and device (no recursion on the device: explicit stack; depth <= 2*log2 n).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SORT_HD __host__ __device__ __forceinline__
#define SORT_HD_BIG __host__ __device__ __attribute__((noinline))
#else
#define SORT_HD inline
#define SORT_HD_BIG inline
#endif

namespace lsort {

template <class T, class Less>
SORT_HD void move_median_to_first(T* a, long result, long x, long y, long z, Less less) {
    auto swp = [&](long i, long j) { T t = a[i]; a[i] = a[j]; a[j] = t; };
    if (less(a[x], a[y])) {
        if (less(a[y], a[z])) swp(result, y);
        else if (less(a[x], a[z])) swp(result, z);
        else swp(result, x);
    } else if (less(a[x], a[z])) swp(result, x);
    else if (less(a[y], a[z])) swp(result, z);
    else swp(result, y);
}

template <class T, class Less>
SORT_HD long unguarded_partition(T* a, long first, long last, long pivot, Less less) {
    while (true) {
        while (less(a[first], a[pivot])) ++first;
        --last;
        while (less(a[pivot], a[last])) --last;
        if (!(first < last)) return first;
        T t = a[first]; a[first] = a[last]; a[last] = t;
        ++first;
    }
}

template <class T, class Less>
SORT_HD void push_heap_(T* a, long first, long hole, long top, T value, Less less) {
    long parent = (hole - 1) / 2;
    while (hole > top && less(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}

template <class T, class Less>
SORT_HD void adjust_heap(T* a, long first, long hole, long len, T value, Less less) {
    const long top = hole;
    long second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (less(a[first + second], a[first + (second - 1)])) second--;
        a[first + hole] = a[first + second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        a[first + hole] = a[first + (second - 1)];
        hole = second - 1;
    }
    push_heap_(a, first, hole, top, value, less);
}

// std::__partial_sort(first, last, last): __heap_select with middle == last is just make_heap, then __sort_heap
template <class T, class Less>
SORT_HD_BIG void heap_sort(T* a, long first, long last, Less less) {
    const long len = last - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        while (true) {
            T value = a[first + parent];
            adjust_heap(a, first, parent, len, value, less);
            if (parent == 0) break;
            parent--;
        }
    }
    long l = last;
    while (l - first > 1) {
        --l;
        T value = a[l];
        a[l] = a[first];
        adjust_heap(a, first, 0, l - first, value, less);
    }
}

template <class T, class Less>
SORT_HD void unguarded_linear_insert(T* a, long last, Less less) {
    T val = a[last];
    long next = last - 1;
    while (less(val, a[next])) {
        a[last] = a[next];
        last = next;
        --next;
    }
    a[last] = val;
}

template <class T, class Less>
SORT_HD void insertion_sort(T* a, long first, long last, Less less) {
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (less(a[i], a[first])) {
            T val = a[i];
            for (long k = i; k > first; --k) a[k] = a[k - 1];      // std::move_backward(first, i, i + 1)
            a[first] = val;
        } else unguarded_linear_insert(a, i, less);
    }
}

struct frame { int first, last, depth; };      // a pending range of the introsort loop
enum { STACK_FRAMES = 72 };                      // at most 2*floor(log2 n) + 1 ranges are ever pending (n < 2^31)

// st: STACK_FRAMES frames of working memory (device callers keep it out of the kernel's private segment)
template <class T, class Less>
SORT_HD_BIG void sort(T* a, long n, Less less, frame* st) {
    if (n <= 0) return;
    // std::__introsort_loop(first, last, std::__lg(n) * 2, comp) with the right-hand recursion on an explicit stack
    long lg = 0;
    for (long v = n; v > 1; v >>= 1) ++lg;
    int sp = 0;
    st[sp++] = frame{0, (int)n, (int)(lg * 2)};
    while (sp > 0) {
        frame f = st[--sp];
        long first = f.first, last = f.last, depth = f.depth;
        while (last - first > 16) {
            if (depth == 0) { heap_sort(a, first, last, less); break; }
            --depth;
            const long mid = first + (last - first) / 2;
            move_median_to_first(a, first, first + 1, mid, last - 1, less);
            const long cut = unguarded_partition(a, first + 1, last, first, less);
            // the reference recurses into [cut, last) first and then loops on [first, cut): the two ranges are disjoint, so
            // finishing the left loop before the deferred right range gives the same result
            st[sp++] = frame{(int)cut, (int)last, (int)depth};
            last = cut;
        }
    }
    // std::__final_insertion_sort
    if (n > 16) {
        insertion_sort(a, 0, 16, less);
        for (long i = 16; i != n; ++i) unguarded_linear_insert(a, i, less);
    } else insertion_sort(a, 0, n, less);
}

#if !defined(__HIP_DEVICE_COMPILE__)
template <class T, class Less>
inline void sort(T* a, long n, Less less) { frame st[STACK_FRAMES]; sort(a, n, less, st); }
#endif

}  // namespace lsort
