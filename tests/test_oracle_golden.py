"""CPU: the oracle (oracle/rii_oracle.c) against the committed golden vectors recorded from the real
reference (tests/golden/, generator tests/gen_golden.py).  Runs anywhere, no reference needed."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.replay import CASE_NAMES, NEARTIE_DS, replay_case, replay_neartie, replay_stale, replay_state, load_case, GOLD


@pytest.mark.parametrize("arch", ["avx512", "avx"])
@pytest.mark.parametrize("name", CASE_NAMES)
def test_oracle_replays_golden(name, arch):
    n = replay_case(lambda cw: O.OracleRii(cw, False, simd_arch=arch), name, arch)
    assert n > 50


def test_fixture_flavours_really_differ():
    """Ds=16 LUTs are summed 16-wide by the AVX512 build and 8-wide by the AVX build (distance.h:117-217):
    the two recorded flavours must not be byte-identical, otherwise the fixture would not pin the order."""
    _, a = load_case("wide_ds16", "avx512")
    _, b = load_case("wide_ds16", "avx")
    differs = any(not np.array_equal(a[k], b[k]) for k in a.files if k.endswith("_d"))
    assert differs


@pytest.mark.parametrize("arch", ["avx512", "avx"])
@pytest.mark.parametrize("Ds", NEARTIE_DS)
def test_oracle_assignment_neartie_golden(Ds, arch):
    replay_neartie(lambda cw: O.OracleRii(cw, False, simd_arch=arch), Ds, arch)


def test_neartie_flavours_differ():
    import os
    a = np.load(os.path.join(GOLD, "neartie_ds4.avx512.out.npz"))
    b = np.load(os.path.join(GOLD, "neartie_ds4.avx.out.npz"))
    assert not (np.array_equal(a["pl_off"], b["pl_off"]) and np.array_equal(a["pl_ids"], b["pl_ids"]))


@pytest.mark.parametrize("arch", ["avx512", "avx"])
def test_oracle_stale_lists_golden(arch):
    replay_stale(lambda cw: O.OracleRii(cw, False, simd_arch=arch), arch)


def _oracle_from_state(arch):
    def make(state):
        o = O.OracleRii.__new__(O.OracleRii)
        o.arch = arch
        o.__setstate__(state)
        return o
    return make


@pytest.mark.parametrize("arch", ["avx512", "avx"])
def test_oracle_loads_reference_pickle_state(arch):
    """f2: the reference's own 5-tuple state (src/main.cpp:35-53) loads, and the answers are the reference's."""
    o = replay_state(_oracle_from_state(arch), arch)
    state = o.__getstate__()
    from tests.replay import load_state_fixture
    want, _ = load_state_fixture(arch)
    assert state[1:] == want[1:]                                          # get-state emits the reference layout again
    assert np.array_equal(np.asarray(state[0], np.float32), np.asarray(want[0], np.float32))
