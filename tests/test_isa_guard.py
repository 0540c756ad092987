"""Tripwire on the code the compiler generates for the four headline kernels (runs on the CPU box: hipcc -S cross-compiles).

What five rounds of measurement settled for `fwd_diag_kernel<{0,1},8,2,4,true>` (the N = 8 forwards: 2 lanes per problem, 4
coordinates per lane, non-diagonal problems solved inside the kernel) and `bwd_diag_kernel<{0,1},8,4,false>`:
  * at most 128 VGPRs (4 waves per SIMD; `amdgpu_waves_per_eu(4, 8)` in fwd_diag.hip -- with (8, 8) the diagonal loops spill);
  * the spills the forward does have (the in-kernel general solve needs more than 128 registers) lie OUTSIDE the diagonal
    ADMM region -- the region every tile executes; NOTES.md records the time they slipped into it: headline step 56 -> 69 us;
  * the steady-state ADMM loop is 79 (QP) / 130 (QCQP) instructions for 4 coordinates per lane (round 5: 91 / 139; an
    instruction of that loop costs the kernel ~0.085 us whether it is scalar or vector, tools/ab_salu_probe_diag.sh).
A compiler or ROCm bump, or an edit that lengthens a live range, then fails HERE instead of silently costing 20 %.

The structure the parser relies on (checked, so that a change of structure fails loudly too): the kernel's largest loop is
the persistent tile loop; the first cluster of overlapping loops inside it that holds FP64 arithmetic is the diagonal ADMM
region (the re-spread stages 2 -> 4 -> 8 lanes per problem); the first innermost loop of that region is the steady state."""
import hashlib
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# kernel (mangled-name fragment) -> (VGPR limit, steady-state loop instructions today, None = no loop check)
FWD = {"fwd_diag_kernelILi0ELi8ELi2ELi4ELb1E": (128, 79), "fwd_diag_kernelILi1ELi8ELi2ELi4ELb1E": (128, 130)}
BWD = {"bwd_diag_kernelILi0ELi8ELi4ELb0E": (128, None), "bwd_diag_kernelILi1ELi8ELi4ELb0E": (128, None)}
SLACK = 1.05


def listing(unit):
    """hipcc -S of one translation unit with the flags of diffqcqp_amd/build.py (device code only); cached in /tmp by the
    hash of the sources and flags."""
    from diffqcqp_amd import build
    tag = hashlib.sha256((build.source_sha16() + unit).encode()).hexdigest()[:16]
    path = "/tmp/dqq_isa_%s_%s.s" % (unit.replace(".hip", ""), tag)
    if not os.path.exists(path):
        cmd = [build._hipcc()] + build.COMMON + build.UNITS[unit] + ["-I", build.INCLUDE, "-S", "--cuda-device-only",
                                                                     os.path.join(build.CSRC, unit), "-o", path + ".tmp"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        os.replace(path + ".tmp", path)
    return open(path).read()


def kernel(text, frag):
    """-> (instruction lines of the kernel, {NumVgprs, ScratchSize, Occupancy})."""
    m = re.search(r"\n(_ZN3dqq\d+" + frag + r"\w+):", text)
    assert m, "kernel %s is not in the listing (renamed? template arguments changed?)" % frag
    i = text.index("\n" + m.group(1) + ":")
    j = text.index(".Lfunc_end", i)
    k = text.index("; Occupancy", j)
    meta = {a: int(b) for a, b in re.findall(r"; (NumVgprs|ScratchSize|Occupancy): (\d+)", text[j:k + 40])}
    return text[i:j].split("\n"), meta


def is_inst(l):
    return l.startswith("\t") and not l.strip().startswith((".", ";"))


def loops(lines):
    """Back-edge regions [(first line, last line)] of a kernel."""
    lab = {}
    for n, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = n
    out = []
    for n, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and lab.get(m.group(1), n) < n:
            out.append((lab[m.group(1)], n))
    return out


def count(lines, a, b, pred):
    return sum(1 for l in lines[a:b + 1] if is_inst(l) and pred(l.strip()))


def diagonal_region(lines):
    """(region, steady-state loop) as described in the module docstring."""
    lp = loops(lines)
    outer = max(lp, key=lambda r: r[1] - r[0])
    inner = sorted(r for r in lp if outer[0] <= r[0] and r[1] <= outer[1] and r != outer and (r[1] - r[0]) < 0.9 * (outer[1] - outer[0]))
    clusters = []
    for a, b in inner:          # union of overlapping / nested regions
        if clusters and a <= clusters[-1][1]:
            clusters[-1][1] = max(clusters[-1][1], b)
        else:
            clusters.append([a, b])
    fp = [c for c in clusters if count(lines, c[0], c[1], lambda s: s.startswith("v_") and "_f64" in s) >= 100]
    assert fp, "no FP64 loop cluster inside the tile loop: the kernel's structure changed, re-derive this test"
    reg = fp[0]
    inside = [r for r in lp if reg[0] <= r[0] and r[1] <= reg[1]]
    innermost = sorted(r for r in inside if not any(o != r and r[0] <= o[0] and o[1] <= r[1] for o in inside))
    steady = next(r for r in innermost if count(lines, r[0], r[1], lambda s: "_f64" in s) >= 20)
    return reg, steady


@pytest.fixture(scope="module")
def fwd_listing():
    return listing("fwd_diag.hip")


@pytest.mark.parametrize("frag", sorted(FWD))
def test_forward_diagonal_loops_have_no_spills_and_stay_lean(fwd_listing, frag):
    lines, meta = kernel(fwd_listing, frag)
    limit, steady_today = FWD[frag]
    assert meta["NumVgprs"] <= limit, meta
    assert meta["Occupancy"] >= 4, meta
    reg, steady = diagonal_region(lines)
    n_reg = count(lines, reg[0], reg[1], lambda s: True)
    assert 800 < n_reg < 2500, ("the diagonal ADMM region is not what it was", reg, n_reg)
    assert count(lines, reg[0], reg[1], lambda s: s.startswith(("global_", "buffer_", "flat_"))) <= 16
    spills = [l.strip() for l in lines[reg[0]:reg[1] + 1] if is_inst(l) and l.strip().startswith("scratch_")]
    assert not spills, "spills inside the diagonal ADMM loops (every tile pays them): %s" % spills[:4]
    n = count(lines, steady[0], steady[1], lambda s: True)
    assert n <= SLACK * steady_today, "steady-state ADMM loop grew: %d instructions (was %d)" % (n, steady_today)
    assert n >= 0.6 * steady_today, "the loop found (%d instructions) is not the steady-state loop: re-derive this test" % n
    f64 = count(lines, steady[0], steady[1], lambda s: s.startswith("v_") and "_f64" in s)
    assert f64 >= 0.5 * n, "steady-state loop is not mostly FP64 arithmetic (%d of %d)" % (f64, n)
    # whatever the kernel spills (the in-kernel general solve) stays small and behind the diagonal region
    all_spills = [i for i, l in enumerate(lines) if is_inst(l) and l.strip().startswith("scratch_")]
    assert meta["ScratchSize"] <= 256 and all(i > reg[1] or i < reg[0] for i in all_spills)


@pytest.mark.parametrize("frag", sorted(BWD))
def test_backward_streaming_kernels_do_not_spill(frag):
    lines, meta = kernel(listing("bwd_diag.hip"), frag)
    assert meta["NumVgprs"] <= BWD[frag][0] and meta["ScratchSize"] == 0 and meta["Occupancy"] >= 4, meta
    assert not any(is_inst(l) and l.strip().startswith("scratch_") for l in lines)
