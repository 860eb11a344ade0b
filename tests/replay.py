"""Replays the scripted calls of a golden fixture (tests/gen_golden.py) against any engine exposing the
RiiCpp surface (src/main.cpp:12-54) and compares with the reference's recorded outputs."""
import json
import os

import numpy as np

from tests.util import assert_same_result

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASE_NAMES = ["readme_m32", "deep_m16", "test_m4_ks20", "test_m20", "wide_ds16", "dup_codes"]


def load_case(name, arch):
    inp = np.load(os.path.join(GOLD, "%s.in.npz" % name))
    out = np.load(os.path.join(GOLD, "%s.%s.out.npz" % (name, arch)))
    return inp, out


def csr_to_lists(off, ids):
    return [ids[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]


def replay_case(make_engine, name, arch):
    """Every call must reproduce the reference's recorded ids and distances bit for bit, exactly tied distances
    included (std::partial_sort's heap order: src/rii.h:234-235,279-280,312-313)."""
    inp, out = load_case(name, arch)
    cw, codes, qs = inp["codewords"], inp["codes"], inp["queries"]
    calls = json.loads(str(inp["calls"]))
    n_hold = int(inp["n_hold"])
    N = codes.shape[0]
    tsets = {k[5:]: inp[k] for k in inp.files if k.startswith("tids_")}
    e = make_engine(cw)
    e.add_codes(codes[:N - n_hold], False)
    e.add_codes(codes[N - n_hold:], False)
    assert e.N == N
    n_checked = 0
    for i, c in enumerate(calls):
        what = "%s call %d %s" % (name, i, c)
        if c["op"] == "reconfigure":
            e.reconfigure(c["nlist"], c["iter"])
            assert np.array_equal(np.array(e.coarse_centers, np.uint8), out["c%d_centers" % i]), what
            want = csr_to_lists(out["c%d_pl_off" % i], out["c%d_pl_ids" % i])
            assert [list(l) for l in e.posting_lists] == want, what
            continue
        if c["op"] == "linear":
            got = e.query_linear(qs[c["q"]], c["topk"], tsets[c["tids"]])
        else:
            got = e.query_ivf(qs[c["q"]], c["topk"], tsets[c["tids"]], c["L"])
        want = (out["c%d_ids" % i], out["c%d_d" % i])
        assert_same_result(got, want, what)
        n_checked += 1
    e.add_codes(out["extra_codes"], True)
    want = csr_to_lists(out["final_pl_off"], out["final_pl_ids"])
    assert [list(l) for l in e.posting_lists] == want, "%s: add_codes(update_flag=True)" % name
    return n_checked


NEARTIE_DS = (4, 6, 16)


def replay_neartie(make_engine, Ds, arch):
    """Engine must offer set_coarse_centers(centers) (import of centres; the reference reaches the same state
    through its pickle hook, src/main.cpp:35-53) and add_codes(codes, True)."""
    inp = np.load(os.path.join(GOLD, "neartie_ds%d.in.npz" % Ds))
    out = np.load(os.path.join(GOLD, "neartie_ds%d.%s.out.npz" % (Ds, arch)))
    e = make_engine(inp["codewords"])
    e.set_coarse_centers(inp["centers"])
    e.add_codes(inp["new_codes"], True)
    assert [list(l) for l in e.posting_lists] == csr_to_lists(out["pl_off"], out["pl_ids"])


STALE_CALLS = ((20, 100), (20, 40), (3, 30), (12, 50), (1, 51), (6, 49))


def replay_stale(make_engine, arch):
    """Empty-return / unsorted-tail branches of QueryIvf (src/rii.h:283-326) with stale posting lists."""
    inp = np.load(os.path.join(GOLD, "stale_lists.in.npz"))
    out = np.load(os.path.join(GOLD, "stale_lists.%s.out.npz" % arch))
    cw, codes, qs = inp["codewords"], inp["codes"], inp["queries"]
    e = make_engine(cw)
    e.add_codes(codes[:50], False)
    e.reconfigure(10, 3)
    e.add_codes(codes[50:], False)
    empty = np.array([], np.int64)
    n_empty = 0
    for ci, (topk, L) in enumerate(STALE_CALLS):
        for b in range(8):
            got = e.query_ivf(qs[b], topk, empty, L)
            want = (out["c%d_q%d_ids" % (ci, b)], out["c%d_q%d_d" % (ci, b)])
            assert_same_result(got, want, "stale k=%d L=%d q=%d" % (topk, L, b))
            n_empty += (len(want[0]) == 0)
    assert n_empty > 0


STATE_CALLS = (("linear", 1, 0), ("linear", 7, 0), ("ivf", 1, 50), ("ivf", 5, 120), ("ivf", 3, 600))


def load_state_fixture(arch):
    """The reference's pickle state (src/main.cpp:35-53) recorded by tests/gen_golden.py: a plain 5-tuple of lists."""
    import pickle
    with open(os.path.join(GOLD, "state_m8.%s.pkl" % arch), "rb") as f:
        state = pickle.load(f)
    return state, np.load(os.path.join(GOLD, "state_m8.%s.out.npz" % arch))


def replay_state(engine_from_state, arch):
    """engine_from_state(5-tuple) -> engine; its answers must equal those of the reference re-created from the state."""
    state, out = load_state_fixture(arch)
    assert isinstance(state, tuple) and len(state) == 5
    e = engine_from_state(state)
    assert e.N == len(state[3]) // len(state[0]) and e.nlist == len(state[2])
    assert [list(l) for l in e.posting_lists] == [list(l) for l in state[4]]
    empty = np.array([], np.int64)
    qs = out["queries"]
    for ci, (op, topk, L) in enumerate(STATE_CALLS):
        for b in range(6):
            got = e.query_linear(qs[b], topk, empty) if op == "linear" else e.query_ivf(qs[b], topk, empty, L)
            assert_same_result(got, (out["c%d_q%d_ids" % (ci, b)], out["c%d_q%d_d" % (ci, b)]),
                               "state %s k=%d L=%d q=%d" % (op, topk, L, b))
    return e
