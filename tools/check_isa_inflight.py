#!/usr/bin/env python3
"""Static check of the filter-scan kernels' ISA: no instruction touches a VGPR that an earlier LDS / global load is still
going to write.

fscan_mx_kernel and fscan_kernel issue their table-row reads (ds_read_b64 / ds_read_b128) and lookup loads (global_load_*)
from inline asm and wait for them with hand-placed s_waitcnt: the loads are invisible to the compiler, which is what lets
several groups of rows stay in flight across the rare candidate branch -- and also what would let the register allocator
copy, spill or reuse a destination register while its load is still outstanding (seen once: rows pinned to fixed registers
were shuffled around the branch before they had landed).  This script compiles fastscan.hip to assembly and walks every
kernel named on the command line (default: all fscan kernels):

  * LGKM queue: every ds_* and s_load* / s_buffer_load* instruction, in order; ds_read* entries carry their destination
    registers.  `s_waitcnt lgkmcnt(N)` retires the oldest entries until N remain (LDS returns in order).
  * VM queue: global_load* / buffer_load* / flat_load* with their destinations; `vmcnt(N)` retires the oldest until N remain
    (stores are ignored: they only make the hardware's count larger, i.e. the real wait stricter).
  * any instruction naming a VGPR that is still pending is reported -- except the loads themselves and instructions carrying
    the marker `rii:inflight-ok` (the once-per-trip threshold refresh is read while in flight on purpose: any mix of old and
    new threshold words is valid).
  * only loads inside inline asm (;;#ASMSTART .. ;;#ASMEND in the assembly) carry destination registers: a load the compiler issued is waited
    for by the compiler (it still takes its slot in the queue).  (Round 4: block placement can put a cold block behind the code that
    follows it in program order, reached and left by branches; walking such text linearly made compiler loads look pending.)
  * control flow: at a forward conditional branch the state is remembered for the target label and merged there (the state
    with more pending registers wins: the skipped block can only have waited for more); loops (backward branches) are walked
    twice so that loads issued at the end of an iteration are seen by the start of the next.

usage: tools/check_isa_inflight.py [--asm file.s | --src kernels.hip] [kernel-substring ...]      exit status 1 if anything is reported."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rii_amd", "csrc")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^\s*s_(cbranch_\w+|branch)\s+(\.LBB\d+_\d+)")
LOAD_VM = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load)")
# gfx9 has no separate store counter: stores and atomics sit in the same vmcnt queue as the loads (the compiler counts its spill
# stores when it picks a vmcnt(N)); they carry no destination register unless they return a value
STORE_VM = re.compile(r"^(global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic|flat_atomic)")
LGKM = re.compile(r"^(ds_|s_load|s_buffer_load)")
OK_MARK = "rii:inflight-ok"


def compile_asm(src="fastscan.hip"):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only",
           os.path.join(CSRC, src), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def regs_of(text):
    regs = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def dest_regs(ops):
    first = ops.split(",")[0]
    return regs_of(first)


def split_functions(path):
    funcs, name, body = {}, None, []
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append(ln.rstrip("\n"))
        if "s_endpgm" in ln:
            funcs[name] = body
            name = None
    return funcs


class State:
    def __init__(self, lgkm=None, vm=None):
        self.lgkm = list(lgkm or [])          # entries: frozenset of pending dest regs (may be empty)
        self.vm = list(vm or [])

    def copy(self):
        return State(self.lgkm, self.vm)

    def pending(self):
        p = set()
        for e in self.lgkm:
            p |= e
        for e in self.vm:
            p |= e
        return p

    def weight(self):
        return len(self.pending())


def check_function(name, lines):
    # instruction list with labels
    insts = []
    in_app = False                                   # between ;;#ASMSTART and ;;#ASMEND: text of an inline asm statement
    for ln in lines:
        if ln.strip().startswith(";;#ASMSTART"):
            in_app = True
            continue
        if ln.strip().startswith(";;#ASMEND"):
            in_app = False
            continue
        code = ln.split(";;")[0]
        m = LABEL.match(code.strip())
        if m:
            insts.append(("label", m.group(1), ln, False))
            continue
        body, _, comment = code.partition(";")
        body = body.strip()
        if not body or body.startswith(".") or body.startswith(";"):
            continue
        insts.append(("inst", body, comment, in_app))
    label_pos = {x[1]: i for i, x in enumerate(insts) if x[0] == "label"}
    problems = []
    saved = {}                                       # label -> State from forward branches
    loop_heads = {}                                  # label index -> index of the last backward branch to it
    for i, x in enumerate(insts):
        if x[0] == "inst":
            m = BRANCH.match(x[1])
            if m and m.group(2) in label_pos and label_pos[m.group(2)] <= i:
                loop_heads[label_pos[m.group(2)]] = max(loop_heads.get(label_pos[m.group(2)], 0), i)

    def step(st, i, report):
        kind, body, comment, hand = insts[i]
        if kind == "label":
            if body in saved and saved[body].weight() > st.weight():
                st = saved[body].copy()
            return st
        mnem = body.split()[0]
        ops = body[len(mnem):]
        if mnem == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ops)
            if m:
                n = int(m.group(1))
                while len(st.lgkm) > n:
                    st.lgkm.pop(0)
            m = re.search(r"vmcnt\((\d+)\)", ops)
            if m:
                n = int(m.group(1))
                while len(st.vm) > n:
                    st.vm.pop(0)
            if not re.search(r"cnt\(", ops):         # numeric immediate form: treat as a full wait
                st.lgkm, st.vm = [], []
            return st
        m = BRANCH.match(body)
        if m:
            tgt = m.group(2)
            if tgt in label_pos and label_pos[tgt] > i:
                if tgt not in saved or saved[tgt].weight() < st.weight():
                    saved[tgt] = st.copy()
            return st
        touched = regs_of(ops)
        ok = OK_MARK in comment or OK_MARK in body
        if LGKM.match(mnem):
            dest = dest_regs(ops) if mnem.startswith("ds_read") else set()
            if not hand:                             # a load the compiler issued: it places the wait itself (and keeps its slot in the queue)
                st.lgkm.append(frozenset())
                return st if not (touched & st.pending()) or not report or ok else (problems.append((i, body, sorted(touched & st.pending()))) or st)
            src = touched - dest
            bad = src & st.pending()
            if bad and report and not ok:
                problems.append((i, body, sorted(bad)))
            # a load re-targeting a pending register is a WAW on in-flight data as well
            if dest & st.pending() and report and not ok:
                problems.append((i, body, sorted(dest & st.pending())))
            st.lgkm.append(frozenset() if ok else frozenset(dest))
            return st
        if LOAD_VM.match(mnem):
            dest = dest_regs(ops)
            if not hand:                             # compiler-issued: only its use of registers still awaited by HAND-PLACED loads matters
                badc = touched & st.pending()
                if badc and report and not ok:
                    problems.append((i, body, sorted(badc)))
                st.vm.append(frozenset())
                return st
            bad = (touched - dest) & st.pending()
            if bad and report and not ok:
                problems.append((i, body, sorted(bad)))
            # re-targeting a register whose pending load sits in the SAME (in-order) queue is harmless -- the compiler does it in
            # loops whose waits live on a backward edge this linear walk does not follow (round 4: the fused re-rank's tail);
            # one pending in the LDS queue is a real hazard (the two queues return independently)
            lg = set()
            for e in st.lgkm:
                lg |= e
            if dest & lg and report and not ok:
                problems.append((i, body, sorted(dest & lg)))
            st.vm.append(frozenset(dest))
            return st
        bad = touched & st.pending()
        if bad and report and not ok:
            problems.append((i, body, sorted(bad)))
        if STORE_VM.match(mnem):
            returns = "atomic" in mnem and (" sc0" in body or " glc" in body)
            st.vm.append(frozenset(dest_regs(ops)) if returns else frozenset())
        return st

    st = State()
    i = 0
    n = len(insts)
    while i < n:
        if i in loop_heads:                          # walk the loop body twice: the second pass starts from the first's end state
            end = loop_heads[i]
            s1 = st.copy()
            for j in range(i, end + 1):
                s1 = step(s1, j, False)
            s2 = s1.copy()
            if s1.weight() < st.weight():
                s2 = st.copy()
            for j in range(i, end + 1):
                s2 = step(s2, j, True)
            st = s2
            i = end + 1
            continue
        st = step(st, i, True)
        i += 1
    # dedupe
    seen, out = set(), []
    for p in problems:
        key = (p[0], tuple(p[2]))
        if key not in seen:
            seen.add(key)
            out.append(p)
    return out, len([x for x in insts if x[0] == "inst"])


def main():
    args = sys.argv[1:]
    asm = None
    if args[:1] == ["--asm"]:
        asm, args = args[1], args[2:]
    src = "fastscan.hip"
    if args[:1] == ["--src"]:                        # another source file of rii_amd/csrc (kernels.hip: scan_kernel's hand-placed row loads)
        src, args = args[1], args[2:]
    if asm is None:
        asm = compile_asm(src)
    subs = args or (["fscan_mx_kernel", "fscan_mx_dual_kernel", "fscan_kernel"] if src == "fastscan.hip" else ["scan_kernel"])
    funcs = split_functions(asm)
    total, bad = 0, 0
    for name, lines in sorted(funcs.items()):
        if not any(s in name for s in subs):
            continue
        probs, ninst = check_function(name, lines)
        total += 1
        status = "ok" if not probs else "%d PROBLEM(S)" % len(probs)
        print("%-72s %6d instructions  %s" % (name[:72], ninst, status))
        for i, body, regs in probs[:12]:
            print("    inst %d: %s   <- pending v%s" % (i, body, regs))
        bad += len(probs)
    print("%d kernels checked, %d problems" % (total, bad))
    return 1 if bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
