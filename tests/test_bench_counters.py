"""bench.py's live counter passes (round 4): the parsing and the fall-backs, without a GPU -- a stand-in `rocprofv3` on PATH writes
the CSV a --pmc pass leaves behind."""
import importlib.util
import os
import stat
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_counters_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_rocprofv3_means_no_live_counters(monkeypatch):
    b = _bench()
    import shutil
    monkeypatch.setattr(shutil, "which", lambda name: None)
    assert b.live_counters("fscan_mx_kernel", []) == {}


def test_counter_csv_is_averaged_over_dispatches_and_summed_over_instances(tmp_path, monkeypatch):
    fake = tmp_path / "rocprofv3"
    fake.write_text("""#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
d = a[a.index("-d") + 1]
ctrs = a[a.index("--pmc") + 1:a.index("--kernel-include-regex")]
os.makedirs(os.path.join(d, "host", "123"), exist_ok=True)
rows = ["Kernel_Name,Dispatch_Id,Counter_Name,Counter_Value"]
for disp, scale in ((1, 1.0), (2, 3.0)):
    for c in ctrs:
        for inst in range(8):                                   # one row per XCD instance
            rows.append("void riiamd::fscan_mx_kernel<8>(FsArgs),%d,%s,%f" % (disp, c, scale * 10.0))
        rows.append("void riiamd::other_kernel(),%d,%s,%f" % (disp, c, 999.0))
open(os.path.join(d, "host", "123", "pmc_counter_collection.csv"), "w").write("\\n".join(rows) + "\\n")
""")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ.get("PATH", ""))
    b = _bench()
    lc = b.live_counters("fscan_mx_kernel", ["--batch", "8"], budget_s=60.0)
    # per launch: instances summed (8 x 10 x scale), dispatches averaged ((80 + 240) / 2)
    assert lc["FETCH_SIZE"] == 160.0 and lc["WRITE_SIZE"] == 160.0 and lc["SQ_INSTS_LDS"] == 160.0 and lc["_dispatches"] == 2
    assert "GRBM_GUI_ACTIVE" in lc


def test_a_failing_profiler_pass_is_not_fatal(tmp_path, monkeypatch):
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\nexit 3\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ.get("PATH", ""))
    assert _bench().live_counters("fscan_mx_kernel", [], budget_s=30.0) == {}
