"""CPU tier: the identity the post-filter's sort on the wave rests on (yaha_amd/csrc/device/oqc_stage.h waveSort), as an executable statement.

The reference's quicksort (GraphPath.cpp:427-453: the pivot swapped to the right end, a Lomuto scan in which a tie against the pivot is decided by the read's next
random bit, the closing swap) must be followed comparison by comparison -- but ONE partition is a pure function of its inputs.  With E[k] the scanned elements and
less[k] = E[k] < pivot, or the k-th tie's bit:
  * a less element ends at rank(k) = number of less elements before it;
  * position k holds, after step k, E[k] if it was not less, else what position rank(k) held: src(k) = rank(k) if less[k] else k is a forest whose roots are the
    not-less elements (the device chases it by pointer doubling);
  * final[j] = E[root(j)] for nL < j < m, the pivot at nL = rank(m), and what position nL held at the right end.
The device code itself is checked on the GPU against the one-thread routine (ygpu_selftest_primitives, the synthetic clump lists); this test keeps the formulation
honest on every box."""
import random


def lomuto(arr, bits):
    a = list(arr); right = len(a) - 1; pivot = right // 2
    a[pivot], a[right] = a[right], a[pivot]
    store, pk, used = 0, a[right][0], 0
    for i in range(right):
        x = a[i]
        less = x[0] < pk
        if x[0] == pk:
            less = bool(bits[used]); used += 1
        if less:
            if i != store:
                a[i], a[store] = a[store], a[i]
            store += 1
    a[store], a[right] = a[right], a[store]
    return a, store, used


def on_the_wave(arr, bits):
    n = len(arr); m = n - 1; pivot = m // 2
    P = arr[pivot]
    E = [arr[m] if k == pivot else arr[k] for k in range(m)]
    less, used = [], 0
    for k in range(m):
        if E[k][0] == P[0]:
            less.append(bool(bits[used])); used += 1
        else:
            less.append(E[k][0] < P[0])
    rank, c = [], 0
    for k in range(m):
        rank.append(c); c += less[k]
    nL = c
    src = [rank[k] if less[k] else k for k in range(m)]
    rounds = 0
    while True:                                          # pointer doubling, as the lanes do it with __shfl
        nxt = [src[src[k]] for k in range(m)]; rounds += 1
        if nxt == src:
            break
        src = nxt
    assert rounds <= max(2, m.bit_length() + 1)
    out = [None] * n
    for k in range(m):
        if less[k]:
            out[rank[k]] = E[k]
    for j in range(nL, m + 1):
        out[j] = P if j == nL else E[src[nL]] if j == m else E[src[j]]
    return out, nL, used


def test_one_partition_is_a_function_of_ranks_and_a_forest():
    rng = random.Random(20261004)
    for _ in range(20000):
        n = rng.choice([2, 3, 4, 5, 7, 16, 33, 63, 64, 65, 100, 200])
        span = rng.choice([1, 2, 3, 10, 1000])
        arr = [(rng.randrange(span + 1), i) for i in range(n)]
        if rng.random() < 0.1:
            arr.sort(reverse=rng.random() < 0.5)
        bits = [rng.randrange(2) for _ in range(n)]
        assert lomuto(arr, bits) == on_the_wave(arr, bits)


def test_the_whole_sort_follows():
    """Both partitions driven by the same recursion (left part completely before the right part: the random bits are consumed in that order)."""
    def sort(arr, part, bits):
        a = list(arr); stack = [(0, len(a) - 1)]; used = 0
        while stack:
            lo, hi = stack.pop()
            if lo >= hi:
                continue
            piece, store, u = part(a[lo:hi + 1], bits[used:])
            a[lo:hi + 1] = piece; used += u
            stack.append((lo + store + 1, hi)); stack.append((lo, lo + store - 1))
        return a, used
    rng = random.Random(7)
    for _ in range(300):
        n = rng.randrange(2, 300); span = rng.choice([1, 3, 20, 10**6])
        arr = [(rng.randrange(span + 1), i) for i in range(n)]
        bits = [rng.randrange(2) for _ in range(40 * n + 64)]
        a, ua = sort(arr, lomuto, bits); b, ub = sort(arr, on_the_wave, bits)
        assert a == b and ua == ub and [x[0] for x in a] == sorted(x[0] for x in arr)
