// std::sort's permutation, faster.
//
// The reference orders keypoints by response with an unstable std::sort (src/extract/extract_features.cpp:55-56), and
// almost every image holds a few keypoints with exactly equal responses (a birthday effect among 20 k floats), whose
// final order is whatever libstdc++'s introsort makes of the order it was given.  Reproducing the reference therefore
// means reproducing that algorithm's sequence of moves, not just "a" sorted order - and a plain call of std::sort was
// the largest single item of the host tail (1.1 ms per image, a third of it), most of it branch mispredictions in the
// partition's two scanning loops.
//
// sort_like_std() is libstdc++'s std::sort (bits/stl_algo.h of GCC 11: __introsort_loop with the median-of-three pivot
// moved to the front, __unguarded_partition, depth limit 2 * lg n with the heap sort fallback, threshold 16,
// __final_insertion_sort) with ONE change that cannot alter the outcome: the partition finds the elements its two scans
// would stop at for a block of 32 positions at a time, without branches (BlockQuicksort's idea), and then swaps them
// pairwise in exactly the order the scanning loops would have - the k-th stop from the left with the k-th stop from the
// right.  Blocks are only taken from the part of the range neither scan has reached, so a left stop is always left of a
// right stop while blocks are in use; when fewer than two blocks of unscanned elements remain the original loop takes
// over from the state it would be in after the last swap (first = one past the last left stop, last = the last right
// stop), so the crossing of the scans, the returned cut and everything after it are the original code's.
// tests/test_host_extract_tail.py compares it with std::sort itself on arrays with ties, runs, and the patterns that
// drive introsort into its heap sort.
#pragma once

#include <algorithm>
#include <cstdint>

namespace opencalibration_amd
{
namespace sort_like_std_detail
{
constexpr long THRESHOLD = 16; // _S_threshold
constexpr int BLOCK = 32; // 64 measures the same; below 32 the block bookkeeping outweighs the saved mispredictions

template <class T, class Comp> inline void move_median_to_first(T *result, T *a, T *b, T *c, Comp comp)
{
    if (comp(*a, *b))
    {
        if (comp(*b, *c))
            std::iter_swap(result, b);
        else if (comp(*a, *c))
            std::iter_swap(result, c);
        else
            std::iter_swap(result, a);
    }
    else if (comp(*a, *c))
        std::iter_swap(result, a);
    else if (comp(*b, *c))
        std::iter_swap(result, c);
    else
        std::iter_swap(result, b);
}

// __unguarded_partition(first, last, pivot): same swaps in the same order, same return value
template <class T, class Comp> inline T *unguarded_partition(T *first, T *last, const T *pivot_at, Comp comp)
{
    const T pivot = *pivot_at; // the pivot's slot lies outside [first, last) and is not written during the partition
    uint8_t off_l[BLOCK], off_r[BLOCK];
    int n_l = 0, s_l = 0, n_r = 0, s_r = 0;
    T *scan_l = first; // everything in [first, scan_l) has been examined by the left scan, [scan_r, last) by the right
    T *scan_r = last;
    T *block_l = first, *block_r = last;
    for (;;)
    {
        if (n_l == 0)
        {
            if (scan_r - scan_l < 2 * BLOCK)
                break;
            block_l = scan_l;
            s_l = 0;
            for (int i = 0; i < BLOCK; i++)
            {
                off_l[n_l] = (uint8_t)i;
                n_l += !comp(block_l[i], pivot); // where "while (comp(*first, pivot)) ++first" stops
            }
            scan_l += BLOCK;
        }
        if (n_r == 0)
        {
            if (scan_r - scan_l < 2 * BLOCK)
                break;
            block_r = scan_r;
            s_r = 0;
            for (int i = 0; i < BLOCK; i++)
            {
                off_r[n_r] = (uint8_t)i;
                n_r += !comp(pivot, block_r[-1 - i]); // where "while (comp(pivot, *last)) --last" stops
            }
            scan_r -= BLOCK;
        }
        const int m = n_l < n_r ? n_l : n_r;
        for (int k = 0; k < m; k++)
            std::iter_swap(block_l + off_l[s_l + k], block_r - 1 - off_r[s_r + k]);
        if (m)
        {
            // the original loop's state after its latest swap
            first = block_l + off_l[s_l + m - 1] + 1;
            last = block_r - 1 - off_r[s_r + m - 1];
        }
        n_l -= m;
        s_l += m;
        n_r -= m;
        s_r += m;
    }
    for (;;)
    {
        while (comp(*first, pivot))
            ++first;
        --last;
        while (comp(pivot, *last))
            --last;
        if (!(first < last))
            return first;
        std::iter_swap(first, last);
        ++first;
    }
}

template <class T, class Comp> void introsort_loop(T *first, T *last, long depth_limit, Comp comp)
{
    while (last - first > THRESHOLD)
    {
        if (depth_limit == 0)
        {
            std::partial_sort(first, last, last, comp); // __heap_select + __sort_heap, as __introsort_loop does
            return;
        }
        --depth_limit;
        T *mid = first + (last - first) / 2;
        move_median_to_first(first, first + 1, mid, last - 1, comp);
        T *cut = unguarded_partition(first + 1, last, first, comp);
        introsort_loop(cut, last, depth_limit, comp);
        last = cut;
    }
}

template <class T, class Comp> inline void unguarded_linear_insert(T *last, Comp comp)
{
    T val = *last;
    T *next = last - 1;
    while (comp(val, *next))
    {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

template <class T, class Comp> inline void insertion_sort(T *first, T *last, Comp comp)
{
    if (first == last)
        return;
    for (T *i = first + 1; i != last; ++i)
    {
        if (comp(*i, *first))
        {
            T val = *i;
            std::move_backward(first, i, i + 1);
            *first = val;
        }
        else
            unguarded_linear_insert(i, comp);
    }
}
} // namespace sort_like_std_detail

// The permutation std::sort(first, last, comp) produces (libstdc++), for trivially copyable T.
template <class T, class Comp> void sort_like_std(T *first, T *last, Comp comp)
{
    using namespace sort_like_std_detail;
    if (first == last)
        return;
    long lg = 0;
    for (unsigned long n = (unsigned long)(last - first); n > 1; n >>= 1)
        lg++;
    introsort_loop(first, last, 2 * lg, comp);
    if (last - first > THRESHOLD)
    {
        insertion_sort(first, first + THRESHOLD, comp);
        for (T *i = first + THRESHOLD; i != last; ++i)
            unguarded_linear_insert(i, comp);
    }
    else
        insertion_sort(first, last, comp);
}

} // namespace opencalibration_amd
