"""A2's workgroup sort (yaha_amd/csrc/device/wgsort.h) as an executable statement, CPU tier.

The device code is checked on the GPU against std::stable_sort (ygpu_selftest_primitives) and through every golden; this file keeps the FORMULATION honest on any
machine: least-significant-digit passes of 8 bits over the hits of a segment in the wave-striped arrangement (a row of 64 lanes = 64 consecutive positions), ranks from
per-wave digit counters that the rows of a wave move on one after the other, a digit-major / wave-minor scan, one exchange a pass -- and the one thing the kernel takes from
the hardware rather than from a manual: with the ranking by LDS atomics the lanes of one instruction that meet on a counter must be served in ascending lane order.  The
model serves them in ANY order on request, and shows what the kernel's order check (every sorted hit strictly above the hit before it in (diagonal, query offset)) is for:
it is silent exactly when the result is the stable sort's.

The reference has no sort here -- it merges per-k-mer lists that are ascending (QueryMatch.c:52-121, QueryHeap.inl:70-134); a stable sort of the hits, written in
ascending query offset, on the diagonal alone reproduces that order."""
import random

import pytest


def wg_sort(hits, threads, ipt, lane_order=None):
    """hits: list of (diag32, qo15) in input order (ascending qo inside one diagonal).  Returns the list after four 8-bit passes.
    lane_order(row_lanes, rng) -> the order in which the lanes of one row's atomic instruction are served (None: ascending = what the hardware does)."""
    nw, n = threads // 64, len(hits)
    assert n <= threads * ipt
    cur = list(hits)
    for shift in (0, 8, 16, 24):
        # rank inside the wave: wave w holds positions [w * 64 * ipt, (w + 1) * 64 * ipt), row k of it the 64 positions from k * 64
        cnt = [[0] * 256 for _ in range(nw)]
        rank = [0] * n
        for w in range(nw):
            for k in range(ipt):
                row = [p for p in range(w * 64 * ipt + k * 64, w * 64 * ipt + k * 64 + 64) if p < n]
                served = row if lane_order is None else lane_order(row)
                for p in served:                                    # one ds_add_rtn a lane: the counter's old value is the lane's rank in its wave
                    d = (cur[p][0] >> shift) & 0xFF
                    rank[p] = cnt[w][d]; cnt[w][d] += 1
        # exclusive sums, digit-major, wave-minor
        base, run = [[0] * 256 for _ in range(nw)], 0
        for d in range(256):
            for w in range(nw):
                base[w][d] = run; run += cnt[w][d]
        out = [None] * n
        for p in range(n):
            w = p // (64 * ipt); d = (cur[p][0] >> shift) & 0xFF
            out[base[w][d] + rank[p]] = cur[p]
        cur = out
    return cur


def order_check_silent(sorted_hits):
    """the kernel's check: every hit strictly above the hit before it in (diagonal, query offset)"""
    return all(a < b for a, b in zip(sorted_hits, sorted_hits[1:]))


def make_segment(rng, n, pattern):
    base, alt = rng.getrandbits(32), rng.getrandbits(32)
    diags = []
    for i in range(n):
        if pattern == "random": d = rng.getrandbits(32)
        elif pattern == "one": d = base
        elif pattern == "two": d = base if i & 1 else alt
        elif pattern == "rows": d = (base + ((i // 64) % 3) * 0x01010101) & 0xFFFFFFFF
        elif pattern == "crowd": d = (base + rng.randrange(5)) & 0xFFFFFFFF if rng.randrange(4) else rng.getrandbits(32)
        else: d = base ^ (rng.randrange(3) << (8 * rng.randrange(4)))
        diags.append(d)
    return [(d, i) for i, d in enumerate(diags)]               # query offsets ascending in input order (k_expand_hits), unique per hit


@pytest.mark.parametrize("threads,ipt", [(128, 8), (256, 4), (512, 3)])
@pytest.mark.parametrize("pattern", ["random", "one", "two", "rows", "crowd", "ties"])
def test_passes_with_lanes_served_in_order_are_the_stable_sort(threads, ipt, pattern):
    rng = random.Random(threads * 131 + ipt * 17 + sum(map(ord, pattern)))
    for n in (1, 63, 64, 65, threads * ipt - 63, threads * ipt):
        hits = make_segment(rng, n, pattern)
        want = sorted(hits, key=lambda h: h[0])                  # Python's sort is stable
        got = wg_sort(hits, threads, ipt)
        assert got == want
        assert order_check_silent(got)


@pytest.mark.parametrize("pattern", ["one", "two", "rows", "crowd", "ties", "random"])
def test_the_order_check_is_silent_exactly_when_the_result_is_the_stable_sort(pattern):
    """Lanes served in a random order (what no manual rules out): the result is still a permutation sorted by the digits processed, but ties between equal diagonals
    may come out in another order -- and whenever they do, two neighbours are not strictly ascending in (diagonal, query offset)."""
    rng = random.Random(7 + len(pattern))
    wrong = 0
    for trial in range(12):
        hits = make_segment(rng, rng.randrange(200, 128 * 8 + 1), pattern)
        want = sorted(hits, key=lambda h: h[0])

        def shuffled(row):
            row = list(row); rng.shuffle(row); return row
        got = wg_sort(hits, 128, 8, lane_order=shuffled)
        assert sorted(got) == sorted(hits)                        # atomics still hand out every rank once: a permutation
        assert order_check_silent(got) == (got == want)
        wrong += got != want
    if pattern != "random":
        assert wrong > 0                                          # (ties are what the lane order decides: these patterns have them)
