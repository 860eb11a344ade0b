"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref, built by `make -C oracle ref`).

Run in the container that has /root/reference:   python tests/gen_golden.py
A fixture is data only: seeded inputs (codewords, codes, queries, target ids) and the reference's outputs
for a scripted list of calls against its pybind11 surface (src/main.cpp:12-54), once per compiled SIMD
flavour ("avx512" = -march=native on this host, "avx" = -march=x86-64-v3).  No reference source is stored.
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import make_problem, ref_with_state, near_tie_assignment_problem  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
E = np.array([], np.int64)

CASES = [  # name, M, Ks, Ds, N, scale, dup, nlists
    ("readme_m32", 32, 256, 4, 2000, "sift", 0, (20, 45)),
    ("deep_m16", 16, 256, 6, 2000, "unit", 0, (20, 45)),
    ("test_m4_ks20", 4, 20, 10, 1000, "unit", 0, (5, 100)),
    ("test_m20", 20, 256, 2, 1000, "unit", 0, (20,)),
    ("wide_ds16", 8, 256, 16, 1500, "unit", 0, (30,)),
    ("dup_codes", 16, 64, 4, 1500, "unit", 500, (30,)),
]


def script(N, nlists, rng):
    """The list of calls replayed against every engine (reference here; oracle and HIP engine in tests)."""
    sub = np.sort(rng.choice(N, N // 10, replace=False)).astype(np.int64)
    tiny = np.array([2, 24, 43, 55, 102, 139, 221, 342, 467, 473, 474, 499], np.int64)
    tsets = {"none": E, "sub": sub, "tiny": tiny, "full": np.arange(N, dtype=np.int64)}
    calls = []
    for qi in range(4):
        for topk in (1, 10, 100):
            calls.append(dict(op="linear", q=qi, topk=topk, tids="none"))
        for t in ("sub", "tiny", "full"):
            calls.append(dict(op="linear", q=qi, topk=min(10, len(tsets[t])), tids=t))
    calls.append(dict(op="linear", q=0, topk=N, tids="none"))
    for nlist in nlists:
        calls.append(dict(op="reconfigure", nlist=nlist, iter=5))
        L0 = int(np.round(N / nlist))
        for qi in range(4):
            for L in (L0, 4 * L0, N):
                for topk in (1, 10):
                    for t in ("none", "sub", "tiny", "full"):
                        if topk > L or (t != "none" and topk > len(tsets[t])):
                            continue
                        calls.append(dict(op="ivf", q=qi, topk=topk, tids=t, L=L))
    return tsets, calls


WORKER = r"""
import sys, json, numpy as np
sys.path.insert(0, %(root)r)
import os
os.environ["RII_REF_FLAVOUR"] = %(flavour)r
from oracle import oracle as O
from tests.gen_golden import CASES, script, run_case
ref, arch, flav = O.load_reference(%(flavour)r)
assert flav == %(flavour)r, (flav, %(flavour)r)
for case in CASES:
    run_case(ref, arch, case)
from tests.gen_golden import run_neartie, run_stale, run_state
run_neartie(ref, arch)
run_stale(ref, arch)
run_state(ref, arch)
"""


def run_case(ref, arch, case):
    name, M, Ks, Ds, N, scale, dup, nlists = case
    cw, codes, qs = make_problem(hash_name(name), M, Ks, Ds, N, scale, dup=dup)
    rng = np.random.default_rng(hash_name(name) + 1)
    tsets, calls = script(N, nlists, rng)
    n_hold = N // 5                                   # appended later with update_flag=True
    e = ref.RiiCpp(cw, False)
    e.add_codes(codes[:N - n_hold], False)
    e.add_codes(codes[N - n_hold:], False)
    out = {}
    for i, c in enumerate(calls):
        if c["op"] == "linear":
            ids, d = e.query_linear(qs[c["q"]], c["topk"], tsets[c["tids"]])
        elif c["op"] == "ivf":
            ids, d = e.query_ivf(qs[c["q"]], c["topk"], tsets[c["tids"]], c["L"])
        else:
            e.reconfigure(c["nlist"], c["iter"])
            pl = e.posting_lists
            out["c%d_centers" % i] = np.array(e.coarse_centers, np.uint8)
            out["c%d_pl_off" % i] = np.concatenate([[0], np.cumsum([len(l) for l in pl])]).astype(np.int64)
            out["c%d_pl_ids" % i] = np.concatenate([np.array(l, np.int32) for l in pl]).astype(np.int32)
            continue
        out["c%d_ids" % i] = np.array(ids, np.int64)
        out["c%d_d" % i] = np.array(d, np.float32)
    # add_codes(update_flag=True) on top of the last configuration (src/rii.h:188-192)
    extra = make_problem(hash_name(name) + 2, M, Ks, Ds, 300, scale)[1]
    e.add_codes(extra, True)
    pl = e.posting_lists
    out["extra_codes"] = extra
    out["final_pl_off"] = np.concatenate([[0], np.cumsum([len(l) for l in pl])]).astype(np.int64)
    out["final_pl_ids"] = np.concatenate([np.array(l, np.int32) for l in pl]).astype(np.int32)
    np.savez_compressed(os.path.join(GOLD, "%s.%s.out.npz" % (name, arch)), **out)
    if arch == "avx512" or not os.path.exists(os.path.join(GOLD, "%s.in.npz" % name)):
        np.savez_compressed(os.path.join(GOLD, "%s.in.npz" % name), codewords=cw, codes=codes, queries=qs[:4],
                            calls=np.array(json.dumps(calls)), n_hold=np.int64(n_hold),
                            **{"tids_" + k: v for k, v in tsets.items()})


NEARTIE_DS = (4, 6, 16)


def run_neartie(ref, arch):
    """Coarse assignment (src/rii.h:335-359) on a problem whose argmin is decided by the summation order the
    compiler gave pqkmeans.cpp:164-173 -- pins the per-flavour arithmetic of the symmetric tables."""
    for Ds in NEARTIE_DS:
        cw, centers, newc = near_tie_assignment_problem(Ds)
        nl = len(centers)
        r = ref_with_state(ref, cw, centers, np.zeros((0, 2), np.uint8), [[] for _ in range(nl)])
        r.add_codes(newc, True)
        pl = r.posting_lists
        off = np.concatenate([[0], np.cumsum([len(l) for l in pl])]).astype(np.int64)
        ids = np.concatenate([np.array(l, np.int32) for l in pl]).astype(np.int32)
        np.savez_compressed(os.path.join(GOLD, "neartie_ds%d.%s.out.npz" % (Ds, arch)), pl_off=off, pl_ids=ids)
        np.savez_compressed(os.path.join(GOLD, "neartie_ds%d.in.npz" % Ds), codewords=cw, centers=centers,
                            new_codes=newc)


STALE_CALLS = ((20, 100), (20, 40), (3, 30), (12, 50), (1, 51), (6, 49))


def run_stale(ref, arch):
    """the `vectors not found` return (src/rii.h:324-325) and the walk over the unsorted tail of the
    coarse order: codes appended with update_flag=False after a reconfigure, so the lists cover only 50 of 5050 ids."""
    cw, codes, qs = make_problem(13, 8, 64, 4, 5050, "unit")
    e = ref.RiiCpp(cw, False)
    e.add_codes(codes[:50], False)
    e.reconfigure(10, 3)
    e.add_codes(codes[50:], False)
    out = {}
    n_empty = 0
    for ci, (topk, L) in enumerate(STALE_CALLS):
        for b in range(8):
            ids, d = e.query_ivf(qs[b], topk, E, L)
            out["c%d_q%d_ids" % (ci, b)] = np.array(ids, np.int64)
            out["c%d_q%d_d" % (ci, b)] = np.array(d, np.float32)
            n_empty += (len(ids) == 0)
    assert n_empty > 0
    np.savez_compressed(os.path.join(GOLD, "stale_lists.%s.out.npz" % arch), **out)
    np.savez_compressed(os.path.join(GOLD, "stale_lists.in.npz"), codewords=cw, codes=codes, queries=qs[:8])


STATE_CALLS = (("linear", 1, 0), ("linear", 7, 0), ("ivf", 1, 50), ("ivf", 5, 120), ("ivf", 3, 600))


def run_state(ref, arch):
    """f2 (src/main.cpp:35-53): the 5-tuple the reference's py::pickle get-state returns for a configured index, stored
    verbatim (a pickle of plain Python lists / floats / bools: data, no reference objects), plus what a reference engine
    re-created from that very state through set-state answers."""
    import pickle
    cw, codes, qs = make_problem(29, 8, 32, 4, 600, "unit", dup=60)
    e = ref.RiiCpp(cw, False)
    e.add_codes(codes[:500], False)
    e.reconfigure(12, 4)
    e.add_codes(codes[500:], True)
    state = e.__getstate__()
    assert isinstance(state, tuple) and len(state) == 5
    with open(os.path.join(GOLD, "state_m8.%s.pkl" % arch), "wb") as f:
        pickle.dump(state, f, protocol=2)
    e2 = ref.RiiCpp.__new__(ref.RiiCpp)
    e2.__setstate__(state)
    out = {}
    for ci, (op, topk, L) in enumerate(STATE_CALLS):
        for b in range(6):
            ids, d = e2.query_linear(qs[b], topk, E) if op == "linear" else e2.query_ivf(qs[b], topk, E, L)
            assert (ids, d) == (e.query_linear(qs[b], topk, E) if op == "linear" else e.query_ivf(qs[b], topk, E, L))
            out["c%d_q%d_ids" % (ci, b)] = np.array(ids, np.int64)
            out["c%d_q%d_d" % (ci, b)] = np.array(d, np.float32)
    np.savez_compressed(os.path.join(GOLD, "state_m8.%s.out.npz" % arch), queries=qs[:6], **out)


def hash_name(name):
    return sum((i + 1) * ord(ch) for i, ch in enumerate(name)) % 100003


def main():
    os.makedirs(GOLD, exist_ok=True)
    for flavour in ("native", "v3"):
        code = WORKER % dict(root=ROOT, flavour=flavour)
        subprocess.check_call([sys.executable, "-c", code])       # one process per flavour (module name `main`)
    tot = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
    print("golden fixtures written to", GOLD, "(%.0f KB)" % (tot / 1024))


if __name__ == "__main__":
    main()
