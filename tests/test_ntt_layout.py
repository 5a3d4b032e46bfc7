"""CPU test of the per-stage twiddle layout of ntt.hip.h (k_ntt_twiddles): entry (s, j) = omega^(j << s), table s holds
(n >> (s + 1)) + 1 entries at offset n - (n >> s) + s.  Round 5 found the kernel computing the ONE padding entry of the buffer
(n + log n entries allocated, n + log n - 1 used) like a real one: exponent n, a read twice the length of the coarse table
away (profiles/r05_anomalies.md (c)).  The restatement below is the kernel's index arithmetic; it pins that every entry the
kernel LOADS for stays inside the two small tables and that exactly one entry is padding."""
import pytest


def entries(logn):
    """(g, s, j, exponent) for every slot of the buffer, as k_ntt_twiddles walks it."""
    n = 1 << logn
    total = n + logn
    out = []
    for g in range(total):
        s = 0
        while s + 1 <= logn - 1 and g >= n - (n >> (s + 1)) + (s + 1):
            s += 1
        j = g - (n - (n >> s) + s)
        out.append((g, s, j, j << s))
    return out


@pytest.mark.parametrize("logn", list(range(1, 15)))
def test_twiddle_slots(logn):
    n = 1 << logn
    l0 = min(12, logn - 1)
    nlo, nhi = 1 << l0, (n >> 1) >> l0
    padding = 0
    seen = set()
    for g, s, j, e in entries(logn):
        assert 0 <= s <= max(0, logn - 1) and j >= 0
        if e < n // 2:                      # a product of the two small tables: both loads in range
            assert (e >> l0) < max(nhi, 1) and (e & (nlo - 1)) < nlo
            assert j <= (n >> (s + 1))
        elif e == n // 2:                   # -1, no load
            assert j == (n >> (s + 1))
        else:                               # the padding slot: stored as zero, no load
            padding += 1
            assert g == n + logn - 1
        seen.add((s, j))
    assert padding == 1
    assert len(seen) == n + logn           # no slot is written twice
    # every (s, j) a transform asks for exists: stage shift s, j <= n >> (s + 1)
    for s in range(logn):
        for j in (0, 1, (n >> (s + 1)) - 1, n >> (s + 1)):
            if 0 <= j <= (n >> (s + 1)):
                assert (s, j) in seen
