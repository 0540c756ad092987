#!/usr/bin/env python3
"""Condenses rocprofv3 output databases (rocpd sqlite) into the small summaries kept under profiles/.

    python tools/summarize_prof.py <kernel_trace.db> <sq_pmc.db> <fetch_pmc.db> <write_pmc.db> <outdir> <tag> [suffix]

Writes  <outdir>/<tag>_kernel_stats.csv   per-kernel calls / mean / min / max duration (ns) and share
        <outdir>/<tag>_pmc.json           per-kernel mean counter values per launch + derived HBM bytes
        <outdir>/pmc_latest<suffix>.json  the same HBM traffic keyed by bench.py's launch names (a launch = every
                                          kernel of that pass: DQQ_P_AUTO on dense P is the verifying fast-path
                                          kernel + the general kernel behind it), "_tag" = <tag>
HBM traffic per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and,
on gfx950 with this rocprofv3, FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE matched the known store
volume of these kernels without correction.
"""
import csv
import json
import os
import re
import sqlite3
import sys


N_SIMD, N_SE = 1024, 32    # MI355X: 256 CUs x 4 SIMDs; 8 XCDs x 4 shader engines


def short(name):
    m = re.search(r"dqq::(\w+)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    name = name.split("(")[0]
    if name.startswith("dqq::"):
        name = name[5:]
    return name[:80]


def bench_key(s):
    """bench.py launch name of a kernel: <family>_<pass>; the family is the KIND template argument (0 qp, 1 qcqp)."""
    if not s.startswith(("fwd_", "bwd_")):
        return None
    m = re.search(r"<(\d)", s)
    fam = {"0": "qp", "1": "qcqp"}.get(m.group(1) if m else "0")
    if "_qp_kernel" in s:
        fam = "qp"
    return None if fam is None else "%s_%s" % (fam, s[:3])


def main():
    kt, sq, fe, wr, outdir, tag = sys.argv[1:7]
    suffix = sys.argv[7] if len(sys.argv) > 7 else ""
    sq2 = sys.argv[8] if len(sys.argv) > 8 else ""
    os.makedirs(outdir, exist_ok=True)
    c = sqlite3.connect(kt)
    rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                     "from kernels group by name order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    with open(os.path.join(outdir, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "mean_ns", "min_ns", "max_ns", "total_ns", "percent"])
        for r in rows:
            w.writerow([short(r[0]), r[1], "%.0f" % r[2], r[3], r[4], r[5], "%.2f" % (100.0 * r[5] / tot)])
    pmc = {}
    for db in (sq, fe, wr, sq2):
        if not db or not os.path.exists(db):
            continue
        c = sqlite3.connect(db)
        for name, counter, val, n in c.execute("select kernel_name, counter_name, avg(value), count(*) "
                                               "from counters_collection group by kernel_name, counter_name"):
            pmc.setdefault(short(name), {})[counter] = val
            pmc[short(name)]["_dispatches"] = max(pmc[short(name)].get("_dispatches", 0), n)
    latest = {}
    for k, v in pmc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes_per_launch"] = 2.0 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024
            v["hbm_read_bytes_per_launch"] = 2.0 * v["FETCH_SIZE"] * 1024
            v["hbm_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024
        if "SQ_INSTS_VALU" in v and v.get("SQ_WAVES"):
            v["valu_insts_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
        # VALU lane utilisation (the gfx9 "VALUUtilization" formula): thread-cycles of VALU work / (cycles a VALU
        # instruction was active x 64 lanes)
        if v.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in v:
            v["valu_lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0)
        # share of the kernel's cycles in which a SIMD's VALU was executing an instruction -- from counters of ONE pass:
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1024 SIMDs, SQ_BUSY_CYCLES cycles summed over its 32
        # shader engines (8 XCDs x 4; checked against the kernels' durations: 2.0 M / 32 = 62.5 k cycles = 26 us at 2.4 GHz for
        # the 28 us QCQP forward).  A measured share: it cannot exceed 1, whatever the mix of FP64 and other instructions
        if v.get("SQ_ACTIVE_INST_VALU") and v.get("SQ_BUSY_CYCLES"):
            v["valu_busy_frac"] = (4.0 * v["SQ_ACTIVE_INST_VALU"] / N_SIMD) / (v["SQ_BUSY_CYCLES"] / N_SE)
        if v.get("SQ_WAVE_CYCLES"):
            for cn in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if cn in v:
                    v[cn + "_frac"] = v[cn] / v["SQ_WAVE_CYCLES"]
        key = bench_key(k)
        if key and "hbm_bytes_per_launch" in v:
            # a launch (one C-ABI call) = the kernels it enqueues; several kernels can serve one launch name over a run (the
            # first calls of a dense batch take other routes than the steady state): bytes per launch = the dispatch-weighted
            # sum over the kernels / the dispatches of the most frequent one (round 5: the plain sum of the per-kernel means
            # counted a once-only route as if every call had taken it)
            e = latest.setdefault(key, {"kernels": [], "_weighted": 0.0, "_calls": 0})
            e["kernels"].append(k)
            e["_weighted"] += v["hbm_bytes_per_launch"] * v.get("_dispatches", 1)
            e["_calls"] = max(e["_calls"], v.get("_dispatches", 1))
            e["hbm_bytes_per_launch"] = e["_weighted"] / e["_calls"]
            # instruction counters: those of the kernel that does the work (the largest), not of the empty
            # work-list launch next to it
            if v.get("SQ_INSTS_VALU", 0.0) >= e.get("SQ_INSTS_VALU", -1.0):
                e["main_kernel"] = k
                for extra in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES",
                              "SQ_INSTS_SALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES",
                              "valu_lane_utilisation", "valu_busy_frac", "SQ_WAIT_INST_ANY_frac", "SQ_WAIT_ANY_frac", "SQ_ACTIVE_INST_ANY_frac",
                              "SQ_ACTIVE_INST_VALU_frac"):
                    if extra in v:
                        e[extra] = v[extra]
    for e in latest.values():
        e.pop("_weighted", None)
        e.pop("_calls", None)
        if e.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
    latest["_tag"] = tag
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:   # which build these counters are of (bench.py quotes counter-derived figures only from a summary of its own build)
        from diffqcqp_amd import build as _b
        latest["_csrc_sha16"] = _b.source_sha16()
    except Exception as e:  # noqa: BLE001
        latest["_csrc_sha16"] = "unknown (%s)" % type(e).__name__
    json.dump(pmc, open(os.path.join(outdir, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    json.dump(latest, open(os.path.join(outdir, "pmc_latest%s.json" % suffix), "w"), indent=1, sort_keys=True)
    for r in rows[:10]:
        print("%-46s calls %5d mean %9.2f us  %5.1f%%" % (short(r[0]), r[1], r[2] / 1e3, 100.0 * r[5] / tot))


if __name__ == "__main__":
    main()
