"""The filter-scan kernels keep table rows and lookups in flight from inline asm, invisible to the compiler (DESIGN 3.1 iv).
tools/check_isa_inflight.py walks the generated ISA and reports any instruction that touches a register whose load has not
been waited for; here it runs on the real build (must be clean) and on a copy with one such access injected (must be caught)."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_isa_inflight", os.path.join(ROOT, "tools", "check_isa_inflight.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def asm_path():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    return _tool().compile_asm()


def test_no_instruction_touches_a_register_with_a_load_in_flight(asm_path):
    t = _tool()
    funcs = t.split_functions(asm_path)
    names = [n for n in funcs if "fscan_mx_kernel" in n or "fscan_mx_dual_kernel" in n or "fscan_kernel" in n]
    assert len(names) >= 30                      # every (shape, mode) instance of both kernels
    for n in names:
        probs, ninst = t.check_function(n, funcs[n])
        assert ninst > 100
        assert not probs, (n, probs[:3])


@pytest.mark.parametrize("kernel", ["fscan_mx_kernelILi8ELi0ELi16E", "fscan_mx_kernelILi16ELi0ELi8E", "fscan_mx_kernelILi4ELi2ELi16E"])
def test_the_checker_catches_an_injected_early_read(asm_path, kernel):
    t = _tool()
    funcs = t.split_functions(asm_path)
    name = [n for n in funcs if kernel in n][0]
    lines = list(funcs[name])
    # the LAST row read of the kernel's hot loop: copy its destination out right behind it, before any wait
    idx = [i for i, ln in enumerate(lines) if re.match(r"\s*ds_read_b(64|128) v\[\d+:\d+\], v\d+\s*$", ln)]
    assert idx
    i = idx[len(idx) // 2]
    reg = int(re.search(r"v\[(\d+):", lines[i]).group(1))
    lines.insert(i + 1, "\tv_mov_b32_e32 v0, v%d" % reg)
    probs, _ = t.check_function(name, lines)
    assert probs and any(reg in p[2] for p in probs)

