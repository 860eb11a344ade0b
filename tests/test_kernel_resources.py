"""CPU: the register / scratch budget of the hot kernels, read from the device code inside the built librii_amd.so.

A kernel's blocks-per-CU is part of its design: `ivf_shard_any_kernel` and `ivf_fused_kernel` are sized for FOUR
256-thread blocks per CU (one wave of each block per SIMD), which needs <= 128 VGPRs; the 1024-thread scan kernels need the same for
ONE block.  Round 5 lost a block per CU once (an inlined cold path took `ivf_shard_any_kernel` to 153 VGPRs: 39 -> 54 us per 1024
queries) and only a GPU run showed it; this test shows it at build time.  The numbers come from the AMDGPU metadata note of every
code object bundled in the library (`llvm-objcopy`, `clang-offload-bundler`, `llvm-readelf` of the ROCm LLVM)."""
import os
import re
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
TOOLS = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def kernel_resources(so, tmp):
    """{mangled kernel name: metadata dict} over every gfx950 code object in the library's .hip_fatbin section."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([TOOLS[0], "--dump-section", ".hip_fatbin=" + fat, so])
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    assert starts, "no offload bundle in " + so
    out = {}
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):
        piece, co = os.path.join(tmp, "b%d.bin" % n), os.path.join(tmp, "d%d.co" % n)
        open(piece, "wb").write(data[a:b])
        subprocess.check_call([TOOLS[1], "--unbundle", "--type=o", "--input=" + piece, "--targets=" + TARGET, "--output=" + co])
        notes = subprocess.run([TOOLS[2], "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:              # one block per kernel; .agpr_count is its first key
            d = dict(re.findall(r"\.(\w+):\s+(\S+)", "  .agpr_count:" + blk))
            out[d["name"]] = d
    return out


def waves_per_simd(d):
    """gfx950: 512 unified registers per lane-slot of a SIMD, VGPRs allocated in blocks of 8, AGPRs on top."""
    regs = (int(d["vgpr_count"]) + 7) // 8 * 8 + int(d.get("agpr_count", 0))
    return min(8, 512 // max(regs, 8))


@pytest.fixture(scope="module")
def resources(tmp_path_factory):
    if not all(os.path.exists(t) for t in TOOLS):
        pytest.skip("ROCm LLVM binary tools not present")
    from rii_amd import core
    return kernel_resources(core.build_library(), str(tmp_path_factory.mktemp("co")))


def _named(resources, stem):
    got = {k: v for k, v in resources.items() if re.match(r"_ZN6riiamd\d+%s(I|E)" % stem, k)}
    assert got, "no kernel named %s in the library" % stem
    return got


def test_every_hot_kernel_is_in_the_library(resources):
    for stem in ("fscan_mx_kernel", "fscan_mx_dual_kernel", "ivf_fused_kernel", "ivf_quad_kernel", "ivf_shard_any_kernel",
                 "ivf_shard_kernel", "lut_build_kernel", "lut_build_mfma_kernel", "rerank_top1_kernel", "merge_topk_kernel"):
        _named(resources, stem)


@pytest.mark.parametrize("stem,min_waves", [
    ("ivf_shard_any_kernel", 4),          # 256 threads, four blocks per CU (table + order in <= 40 KiB of LDS each)
    ("ivf_fused_kernel", 4),              # 256 threads, four blocks per CU at M = 32
    ("ivf_shard_kernel", 4),
    ("ivf_quad_kernel", 4),               # 1024 threads = four waves per SIMD: one block per CU
    ("shard_coarse_quad_kernel", 4),      # likewise (round 6: the coarse pre-pass of the database-sharded inverted index)
    ("fscan_mx_kernel", 4),
    ("fscan_mx_dual_kernel", 4),
    ("rerank_top1_kernel", 8),
    ("lut_build_kernel", 4),
])
def test_blocks_per_cu_budget(resources, stem, min_waves):
    for name, d in _named(resources, stem).items():
        assert waves_per_simd(d) >= min_waves, "%s: %s VGPRs + %s AGPRs -> %d waves per SIMD, designed for %d" % (
            name, d["vgpr_count"], d.get("agpr_count", 0), waves_per_simd(d), min_waves)


@pytest.mark.parametrize("name", [
    "_ZN6riiamd15fscan_mx_kernelILi8ELi0ELi16ELb0ELb0EEEvNS_6FsArgsE",           # the headline scan (M = 32, all codes)
    "_ZN6riiamd15fscan_mx_kernelILi4ELi0ELi16ELb0ELb0EEEvNS_6FsArgsE",           # M = 16
    "_ZN6riiamd20fscan_mx_dual_kernelILi0ELb0EEEvNS_6FsArgsE",                   # M = 16, two tiles per block (config 5's scan): 24 B/lane until round 6
    "_ZN6riiamd20fscan_mx_dual_kernelILi0ELb1EEEvNS_6FsArgsE",                   # ... with the fused re-rank tail (28 B/lane until round 6)
    "_ZN6riiamd20fscan_mx_dual_kernelILi1ELb0EEEvNS_6FsArgsE",
    "_ZN6riiamd20fscan_mx_dual_kernelILi2ELb0EEEvNS_6FsArgsE",
])
def test_headline_scan_has_no_scratch(resources, name):
    assert name in resources, "instantiation renamed? " + name
    d = resources[name]
    assert int(d["private_segment_fixed_size"]) == 0 and int(d["vgpr_spill_count"]) == 0, d


@pytest.mark.parametrize("stem", ["ivf_shard_any_kernel", "ivf_fused_kernel", "ivf_shard_kernel", "rerank_top1_kernel", "lut_build_kernel",
                                  "shard_coarse_quad_kernel"])
def test_no_scratch_in_the_one_block_per_query_kernels(resources, stem):
    for name, d in _named(resources, stem).items():
        assert int(d["private_segment_fixed_size"]) == 0, "%s uses %s bytes of scratch per lane" % (name, d["private_segment_fixed_size"])
