"""Round 6 GPU parity tests (through the C ABI): ivf_rot_kernel -- the conflict-free table gather of the one-query inverted-index
block (table [ks][64 columns], lanes skewed in time over rotated 64-row tiles) -- against ivf_fused_kernel and the CPU oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import make_problem, assert_same_result

pytestmark = pytest.mark.gpu
E = np.array([], np.int64)


@pytest.mark.parametrize("M,Ds,scale,nlist", [(64, 2, "sift", 1000), (32, 4, "sift", 1024), (64, 2, "unit", 100), (32, 4, "unit", 64), (64, 4, "sift", 37),
                                              (32, 2, "unit", 1), (64, 2, "sift", 129)])
def test_ivf_rot_kernel_equals_the_direct_gather_and_the_oracle(M, Ds, scale, nlist):
    """ivf_rot_kernel (option ivf_rot = 2: wherever it applies) against ivf_fused_kernel (ivf_rot = 0) and the oracle, bit for bit:
    ragged lists (lengths that are not multiples of the 64-row tile, empty lists when nlist is large), L from a handful (one partial
    tile) over list-boundary cuts to N (every list: the stop rule's "all w lists walked" arm and the tail walk -> flagged), integer-valued
    data with duplicated rows (exactly tied distances: first minimum in traversal order; tied coarse distances -> flagged queries, replayed
    by the block after it rebuilt the plain table, and -- ivf_inline_exact = 0 -- by the flag-gated exact kernels), every query flagged,
    one lane of work (nlist = 1), 1 .. 16 tiles of centres."""
    from rii_amd import RiiGpu
    N = 30011
    cw, codes, qs = make_problem(1200 + M + nlist + Ds, M, 256, Ds, N, scale, dup=2000 if scale == "sift" else 0)
    rng = np.random.default_rng(6)
    Q = np.concatenate([qs, rng.permutation(qs.reshape(-1)).reshape(qs.shape), qs * 0.5, qs[:5] + 1.0]).astype(np.float32)      # 53 queries
    if scale == "sift":
        Q = np.round(Q)
    o = O.OracleRii(cw, False, simd_arch="avx512")
    o.add_codes(codes, False)
    o.reconfigure(nlist, 2)
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.add_codes(codes, False)
    g.set_coarse_centers(np.array(o.coarse_centers, np.uint8))
    assert g.posting_lists == o.posting_lists
    g.set_option("ivf_quad", 0)
    n0 = g.get_option("ivf_rot_launches")
    L0 = max(1, N // nlist)
    for B in (53, 7):
        for L in (L0, max(1, L0 // 3), min(N, 7 * L0 + 13), min(N, 20 * L0), 5, N):
            want = [o.query_ivf(Q[b], 1, E, L) for b in range(B)]
            for rot, inline, force in ((2, 1, 0), (0, 1, 0), (2, 0, 0), (2, 1, 1), (2, 0, 1)):
                g.set_option("ivf_rot", rot)
                g.set_option("ivf_inline_exact", inline)
                g.set_option("ivf_force_exact", force)
                gi, gd, gc = g.query_ivf_batch(Q[:B], 1, E, L)
                for b in range(B):
                    n = int(gc[b])
                    assert_same_result((gi[b, :n], gd[b, :n]), want[b], "ivf rot=%d inline=%d force=%d M=%d Ds=%d nlist=%d B=%d L=%d b=%d"
                                       % (rot, inline, force, M, Ds, nlist, B, L, b))
    # every ivf_rot = 2 call above whose w = min(nlist, round(L nlist / N) + 3) is within the kernel's 32 picks ran ivf_rot_kernel
    ws = [min(nlist, int(np.round(L * nlist / N)) + 3) for L in (L0, max(1, L0 // 3), min(N, 7 * L0 + 13), min(N, 20 * L0), 5, N)]
    # (the seven-query calls may take the host-flag route of small host-pointer calls, which keeps the one-query kernel)
    assert g.get_option("ivf_rot_launches") - n0 >= 4 * sum(1 for w in ws if w <= 32)
    for k, v in (("ivf_rot", 1), ("ivf_inline_exact", 1), ("ivf_force_exact", 0), ("ivf_quad", 1)):
        g.set_option(k, v)


def test_ivf_rot_copies_follow_the_lists():
    """The rotated tile copies are rebuilt when the lists change: add_codes with update_flag, a second reconfigure, set_posting_lists --
    queries in between keep equal to the oracle's (and to the direct gather's)."""
    from rii_amd import RiiGpu
    import bench
    M, Ds, N = 64, 2, 20000
    cw, codes, qs = make_problem(77, M, 256, Ds, N + 3000, "unit")
    qs = np.concatenate([qs, qs * 0.5, qs[::-1] * 0.25 + 0.1]).astype(np.float32)        # 48 queries: past the small-call route, which keeps the one-query kernel
    o = O.OracleRii(cw, False, simd_arch="avx512")
    g = RiiGpu(cw, False, simd_arch="avx512")
    g.set_option("ivf_rot", 2)
    g.set_option("ivf_quad", 0)
    o.add_codes(codes[:N], False)
    g.add_codes(codes[:N], False)

    def check(L, what):
        gi, gd, gc = g.query_ivf_batch(qs, 1, E, L)
        for b in range(len(qs)):
            n = int(gc[b])
            assert_same_result((gi[b, :n], gd[b, :n]), o.query_ivf(qs[b], 1, E, L), what + " b=%d" % b)

    o.reconfigure(50, 2)
    g.set_coarse_centers(np.array(o.coarse_centers, np.uint8))
    check(2500, "after the first configuration")
    o.add_codes(codes[N:], True)
    g.add_codes(codes[N:], True)
    assert g.posting_lists == o.posting_lists
    check(2500, "after add_codes(update_flag)")
    o.reconfigure(200, 2)
    g.set_coarse_centers(np.array(o.coarse_centers, np.uint8))
    check(3000, "after the second configuration")
    cen = np.random.default_rng(4).integers(0, 256, size=(141, M), dtype=np.uint8)
    off, ids = bench.modulo_lists(N + 3000, 141)
    g.set_posting_lists(cen, off, ids)
    o.set_csr(cen, off, ids)
    check(2100, "after set_posting_lists")
    assert g.get_option("ivf_rot_launches") == 4
