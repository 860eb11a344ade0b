"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def make_problem(seed, M, Ks, Ds, N, scale="unit", dup=0):
    """Seeded codebooks / codes / queries.  scale: 'unit' = U[0,1), 'sift' = integer-valued 0..255."""
    rng = np.random.default_rng(seed)
    if scale == "sift":
        cw = np.round(rng.random((M, Ks, Ds)) * 255).astype(np.float32)
        qs = np.round(rng.random((16, M * Ds)) * 255).astype(np.float32)
    else:
        cw = rng.random((M, Ks, Ds)).astype(np.float32)
        qs = rng.random((16, M * Ds)).astype(np.float32)
    codes = rng.integers(0, Ks, size=(N, M), dtype=np.uint8) if Ks <= 256 else None
    if dup:   # force exactly tied distances: overwrite random rows with copies of other rows
        src = rng.integers(0, N, size=dup)
        dst = rng.integers(0, N, size=dup)
        codes[dst] = codes[src]
    return cw, codes, qs


def assert_same_result(got, want, what=""):
    gi, gd = got
    wi, wd = want
    assert len(gi) == len(wi), "%s: length %d vs %d" % (what, len(gi), len(wi))
    gd = np.asarray(gd, np.float32)
    wd = np.asarray(wd, np.float32)
    assert np.array_equal(gd.view(np.uint32), wd.view(np.uint32)), "%s: distances differ (bitwise)" % what
    assert list(gi) == list(wi), "%s: ids differ" % what


def ref_with_state(ref, cw, centers, codes, lists):
    """Build a reference engine with chosen coarse centres through its pickle hook (src/main.cpp:35-53)."""
    e = ref.RiiCpp.__new__(ref.RiiCpp)
    e.__setstate__((cw.tolist(), False, centers.tolist(), codes.reshape(-1).tolist(), lists))
    return e


def near_tie_assignment_problem(Ds, seed=None):
    """Codebooks whose entries 16.. are an anchor (entries 0..15) plus a signed *permutation* of one delta
    vector: all anchor<->entry distances agree up to the summation order, so the argmin of the coarse
    assignment is decided by the exact arithmetic of pqkmeans.cpp:164-173 as compiled."""
    rng = np.random.default_rng(Ds if seed is None else seed)
    M, Ks = 2, 256
    cw = np.zeros((M, Ks, Ds), np.float32)
    for m in range(M):
        anchors = rng.random((16, Ds)).astype(np.float32)
        delta = (rng.random(Ds) * 0.5).astype(np.float32)
        cw[m, :16] = anchors
        for k in range(16, Ks):
            cw[m, k] = anchors[k % 16] + rng.permutation(delta) * rng.choice([-1, 1], Ds)
    centers = np.stack([np.arange(16, Ks), np.arange(16, Ks)], 1).astype(np.uint8)
    rngc = np.random.default_rng(1)
    centers[:, 1] = 16 + ((centers[:, 1] - 16 + 16 * rngc.integers(0, 15, len(centers))) % 240)
    newc = np.stack(np.meshgrid(np.arange(16), np.arange(16)), -1).reshape(-1, 2).astype(np.uint8)
    return cw, centers, newc
