#!/usr/bin/env python3
"""Folds a rocprofv3 --pmc summary (tools/profile_bench.sh -> gpurun_out/prof_summary/<tag>_pmc.json) into the two small
tables bench.py reads for its `roofline` object:
   profiles/traffic.json[key].hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024   (counters are in KB; FETCH_SIZE is
                                                      doubled per the MI355X guide's gfx950 note)
   profiles/pmc.json[key] = {lds_conflict_frac, lds_busy, valu_busy, ...}                   (SQ counters of the same kernel)
usage: tools/make_profile_tables.py <pmc.json> <kernel substring> <workload key> <source label> [launch ms]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, ksub, key, label = sys.argv[1:5]
    tab = json.load(open(src))
    # "a+b": the step is two kernels (round 6: the sharded inverted index = coarse pre-pass + walk) -- their counters are added
    names = []
    for sub in ksub.split("+"):
        hit = [k for k in tab if sub in k]
        assert len(hit) == 1, (sub, hit)
        names.append(hit[0])
    c = dict(tab[names[0]])
    for extra in names[1:]:
        for k, v in tab[extra].items():
            if isinstance(v, (int, float)) and not k.startswith("_"):
                c[k] = c.get(k, 0.0) + v
    names = [" + ".join(names)]
    n_cu, n_simd, n_xcd = 256, 1024, 8
    cycles = c["GRBM_GUI_ACTIVE"] / n_xcd                       # shader cycles of one launch
    pmc = {"kernel": names[0], "source": label,
           "lds_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
           "lds_busy": c["SQ_LDS_IDX_ACTIVE"] / n_cu / cycles,
           "valu_busy": c["SQ_INSTS_VALU"] * 4.0 / n_simd / cycles,
           "lds_cycles_per_read": c["SQ_LDS_IDX_ACTIVE"] / c["SQ_INSTS_LDS"],
           "valu_insts": c["SQ_INSTS_VALU"], "lds_insts": c["SQ_INSTS_LDS"], "gpu_cycles": cycles}
    if c.get("SQ_WAVE_CYCLES"):                                # wave-state split (all in the SQ's 4-cycle units)
        pmc["wave_wait_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]            # parked in s_waitcnt
        pmc["wave_wait_inst_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]  # waiting to issue (any pipe)
        pmc["wave_wait_inst_lds_frac"] = c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"]
        pmc["wave_active_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        pmc["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n_simd / cycles
        pmc["mfma_insts"] = c.get("SQ_INSTS_MFMA")
    if len(sys.argv) > 5:                                      # the launch's duration in ms (kernel trace) -> shader clock
        pmc["shader_clock_ghz"] = cycles / (float(sys.argv[5]) * 1e-3) / 1e9
    traffic = {"kernel": names[0], "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
               "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), "source": label}
    for name, entry in (("pmc.json", pmc), ("traffic.json", traffic)):
        path = os.path.join(ROOT, "profiles", name)
        t = json.load(open(path)) if os.path.exists(path) else {}
        t[key] = entry
        json.dump(t, open(path, "w"), indent=1)
    print(json.dumps({"pmc": pmc, "traffic": traffic}, indent=1))


if __name__ == "__main__":
    main()
