"""Level-synchronous restatement of libstdc++'s std::sort whose partition is computed from the ORIGINAL content of the range
(prefix ranks of the two scans' stop predicates) - the formulation the device kernel uses - checked against std::sort."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from opencalibration_amd import host

L = host.load()


def std_order(resp):
    resp = np.ascontiguousarray(resp, np.float32)
    import ctypes as C
    out = np.zeros(max(len(resp), 1), np.uint32)
    L.och_sort_by_response.restype = None
    L.och_sort_by_response.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
    L.och_sort_by_response(resp.ctypes.data, len(resp), out.ctypes.data, 1)
    return out[:len(resp)]


def par_sort(keys):
    """keys: float array; comparator comp(a, b) = key[a] > key[b] (descending), records (key, index)."""
    n = len(keys)
    A = [(float(k), i) for i, k in enumerate(keys)]
    comp = lambda a, b: a[0] > b[0]
    if n == 0:
        return [], False
    lg = 0
    m = n
    while m > 1:
        m >>= 1
        lg += 1
    level = [(0, n, 2 * lg)]
    done = []
    fallback = False
    while level:
        nxt = []
        for first, last, depth in level:
            if last - first <= 16:
                done.append((first, last))
                continue
            if depth == 0:
                fallback = True
                done.append((first, last))
                continue
            depth -= 1
            mid = first + (last - first) // 2
            a, b, c = first + 1, mid, last - 1
            # move_median_to_first(first, a, b, c)
            if comp(A[a], A[b]):
                if comp(A[b], A[c]):
                    pick = b
                elif comp(A[a], A[c]):
                    pick = c
                else:
                    pick = a
            elif comp(A[a], A[c]):
                pick = a
            elif comp(A[b], A[c]):
                pick = c
            else:
                pick = b
            A[first], A[pick] = A[pick], A[first]
            p = A[first]
            lo, hi = first + 1, last
            listL = [i for i in range(lo, hi) if not comp(A[i], p)]
            listR = [i for i in range(lo, hi) if not comp(p, A[i])]
            nL, nR = len(listL), len(listR)
            K = 0
            while K < min(nL, nR) and listL[K] < listR[nR - 1 - K]:
                K += 1
            for k in range(K):
                i, j = listL[k], listR[nR - 1 - k]
                A[i], A[j] = A[j], A[i]
            if K == 0:
                cut = listL[0]
            else:
                cut = listR[nR - K]
                if K < nL:
                    cut = min(cut, listL[K])
            nxt.append((cut, last, depth))
            nxt.append((first, cut, depth))
        level = nxt
    # final insertion: every block on its own, stable insertion (moves left while strictly before)
    for first, last in done:
        for i in range(first + 1, last):
            v = A[i]
            j = i
            while j > first and comp(v, A[j - 1]):
                A[j] = A[j - 1]
                j -= 1
            A[j] = v
    return [x[1] for x in A], fallback


def check(keys, name):
    keys = np.asarray(keys, np.float32)
    got, fb = par_sort(keys)
    exp = std_order(keys)
    ok = fb or list(exp) == got
    print(name, len(keys), "fallback" if fb else "", "OK" if ok else "MISMATCH")
    return ok


def killer(n):
    k = n // 2
    a = np.zeros(n)
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return a


def main():
    rng = np.random.default_rng(1)
    ok = True
    for n in (0, 1, 2, 15, 16, 17, 18, 33, 100, 1000, 5000, 20477):
        ok &= check(rng.uniform(0, 1, n), "uniform")
        ok &= check(rng.integers(0, 8, n), "heavy ties")
        ok &= check(rng.integers(0, max(n // 4, 1), n), "some ties")
        ok &= check(np.arange(n), "ascending")
        ok &= check(np.arange(n)[::-1], "descending")
        ok &= check(np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]), "organ pipe")
        ok &= check(np.zeros(n), "all equal")
    # median-of-three killer
    def killer(n):
        k = n // 2
        a = np.zeros(n)
        for i in range(1, k + 1):
            if i % 2 == 1:
                a[i - 1] = i
                a[i] = k + i
            a[k + i - 1] = 2 * i
        return a
    for n in (200, 2000, 20000):
        ok &= check(killer(n), "killer")
        ok &= check(-killer(n), "killer reversed")
    print("ALL OK" if ok else "FAILURES")


if __name__ == "__main__":
    main()
