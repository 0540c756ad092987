#!/usr/bin/env python3
"""The measured numbers of DESIGN.md section 5 and of README.md's first screen, generated from ONE full bench record
(profiles/<tag>_bench_details.json, written by `python bench.py --gpus 1 --steps 20 --warmup 5`) so that no figure is
transcribed by hand:    python tools/doc_numbers.py profiles/<tag>_bench_details.json [--write]
--write replaces the text between the markers <!-- measured:begin --> / <!-- measured:end --> and
<!-- result:begin --> / <!-- result:end --> (README.md)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sci(v):
    e = int(("%e" % v).split("e")[1])
    return "%.2f × 10%s" % (v / 10 ** e, str(e).translate(str.maketrans("0123456789-", "⁰¹²³⁴⁵⁶⁷⁸⁹⁻")))


def main():
    path = sys.argv[1]
    tag = os.path.basename(path).split("_")[0]
    d = json.load(open(path))
    rl, pc, d8, ex = d["roofline"], d["per_config"], d["dense_p_n8"], d["survey_8d_extras"]
    qp, qpl = d["qp_pair"], d["qp_pair_large"]
    k = d["kernels"]
    fig = ex["reference_figure_workload"]
    cb = d["cpu_baseline"]
    flip = d["parity_max_abs_err_vs_oracle_sample"]["qcqp"]["refinement_exit_flip_rate"]
    rows = [
        ("headline: B = 65536, N = 8, QP fwd+bwd ‖ QCQP fwd+bwd, two streams; **rotating buffers (`value`)**",
         "%.4f" % d["ms_per_step"], sci(d["value"]),
         "step %.2f; dominant launch (QCQP fwd, %.1f µs cold) `roofline.frac` %.3f on SURVEY's 704 B, %.3f on the 769 B it moves"
         % (rl["step_moved_frac"], rl["kernel_us_qcqp_fwd"], rl["frac"], rl["moved_frac"]),
         "launches %.1f / %.1f / %.1f / %.1f µs (QP fwd, bwd, QCQP fwd, bwd); traffic %.1f MB = %.2f × algorithmic; `valu_busy_frac` %s"
         % (k["qp_fwd"]["mean_us"], k["qp_bwd"]["mean_us"], k["qcqp_fwd"]["mean_us"], k["qcqp_bwd"]["mean_us"],
            (rl.get("traffic") or 0) / 1e6, rl.get("traffic_over_algorithmic") or 0, "%.2f" % rl["valu_busy_frac"] if "valu_busy_frac" in rl else "n/a")),
        ("the same step on ONE set of buffers (cache-resident, `hot_*`)", "%.4f" % d["hot"]["ms_per_step"], sci(d["hot"]["value"]), "", "one stream: %.4f ms" % d["single_stream"]["ms_per_step"]),
        ("`qp_pair` (north-star sentence): B = 65536, N = 8 QP fwd+bwd, one stream, rotating buffers", "%.4f" % qp["ms_per_step"], sci(qp["value"]),
         "%.2f (%.2f on SURVEY's 1 920 B)" % (qp["roofline"]["step_moved_frac"], qp["roofline"]["step_algorithmic_frac"]),
         "hot: %.4f ms" % qp["hot"]["ms_per_step"] if "hot" in qp else ""),
        ("`qp_pair_large`: B = 1 048 576", "%.3f" % qpl["ms_per_step"], sci(qpl["value"]),
         "**%.2f** (%.2f algorithmic): the ≥ 40 %% target is met where the chip is filled" % (qpl["roofline"]["step_moved_frac"], qpl["roofline"]["step_algorithmic_frac"]), ""),
        ("configs[1] QP fwd", "%.4f" % pc["config_2"]["ms_per_step"], sci(pc["config_2"]["value"]), "%.2f" % pc["config_2"]["roofline"]["step_moved_frac"], ""),
        ("configs[2] QCQP fwd+bwd", "%.4f" % pc["config_3"]["ms_per_step"], sci(pc["config_3"]["value"]), "%.2f" % pc["config_3"]["roofline"]["step_moved_frac"], ""),
        ("configs[3] B = 262144, N = 32 fwd+bwd (one GPU)", "%.3f" % pc["config_4"]["ms_per_step"], sci(pc["config_4"]["value"]),
         "%.2f" % pc["config_4"]["roofline"]["step_moved_frac"],
         "a rank's eighth (B = 32768): %.4f ms = %.2f of linear" % (pc["config_4_shard_1_of_8"]["ms_per_step"],
                                                                    pc["config_4"]["ms_per_step"] / 8 / pc["config_4_shard_1_of_8"]["ms_per_step"])),
        ("configs[4] B = 65536, N = 64 dense fwd+bwd", "%.2f (%.2f + %.2f)" % (pc["config_5"]["ms_per_step"], pc["config_5"]["kernels_us"]["qp_fwd"] / 1e3,
                                                                             pc["config_5"]["kernels_us"]["qp_bwd"] / 1e3), sci(pc["config_5"]["value"]),
         "%.2f; FP64 %.2f" % (pc["config_5"]["roofline"]["step_moved_frac"], pc["config_5"]["roofline"]["fp64_frac"]),
         "compute-bound; forward traffic = %.2f × algorithmic" % pc["config_5"]["roofline"].get("traffic_over_algorithmic", 0)),
        ("dense 8 × 8 QCQP fwd+bwd: `DENSE` / `AUTO` with the report word / `AUTO` + `DQQ_F_EXPECT_DENSE` given by the caller / `AUTO` hint-free",
         "%.3f / %.3f / %.3f / %.3f" % (d8["dense_ms_per_fwd_bwd"], d8["auto_ms_per_fwd_bwd"], d8.get("auto_explicit_flag_ms_per_fwd_bwd", 0),
                                        d8.get("auto_no_hint_ms_per_fwd_bwd", 0)), "", "0.08",
         "hint-free = %.2f × declared dense (closed, §4.6)" % (d8.get("auto_no_hint_ms_per_fwd_bwd", 0) / d8["dense_ms_per_fwd_bwd"])),
        ("stress `p ~ U(0,1)` QP fwd; **the reference's figure workload** (`test_script.py:91-123`, eps 1e-10) QP / QCQP fwd",
         "%.3f; %.2f / %.2f" % (ex["stress_p_u(0,1)_qp_fwd"]["ms_per_call"], fig["qp_fwd_ms"], fig["qcqp_fwd_ms"]), "", "",
         "bound by the slowest problem's serial iterations (QP: one problem at %d)" % fig["qp_iterations"]["max"]),
        ("CPU port, %d cores / 1 thread / Python loop (the reference's execution model)" % cb["cores"], "",
         "%s / %s / %s" % (sci(cb["value"]), sci(cb["single_thread_value"]), sci(cb["python_loop_value"])), "", "published reference: ≈ 1.1 × 10⁴ fwd/s/core"),
    ]
    table = "| workload | ms / step | solves/s | fraction of 8 TB/s on bytes that move | notes |\n|---|---|---|---|---|\n"
    table += "\n".join("| " + " | ".join(r) + " |" for r in rows)
    ptag = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get("_tag", tag)
    head = ("One run of the driver's command on the final tree (`profiles/%s_bench_line.json` = the 2.3 KB contract line, `%s_bench_details.json` = "
            "the full record; rocprofv3 summaries `profiles/%s_*kernel_stats.csv`, `%s_pmc.json`, `pmc_latest*.json`; generated by "
            "`tools/doc_numbers.py`). Boxes of the pool differ by a few per cent.\n\n" % (tag, tag, ptag, ptag))
    result = "\n".join([
        "| | |", "|---|---|",
        "| **N = 8 QP forward+backward, B = 65536** (`qp_pair`, one stream) | %s solves/s on fresh buffers every step (%.4f ms; %.4f ms cache-resident); "
        "%.2f of the 8 TB/s HBM peak on the 1 538 B per problem that move, %.2f on SURVEY's 1 920 B |"
        % (sci(qp["value"]), qp["ms_per_step"], qp.get("hot", {}).get("ms_per_step", float("nan")), qp["roofline"]["step_moved_frac"], qp["roofline"]["step_algorithmic_frac"]),
        "| **the same at B = 1 048 576** (the chip filled) | %s solves/s (%.3f ms); **%.2f** on bytes that move, %.2f on SURVEY's |"
        % (sci(qpl["value"]), qpl["ms_per_step"], qpl["roofline"]["step_moved_frac"], qpl["roofline"]["step_algorithmic_frac"]),
        "| **QCQP gradients** | within 1e-6 of the CPU oracle on the %.0f %% of problems whose refinement exit (1 or 3 bodies, decided by rounding noise: "
        "`Solver.cpp:32-41`) matches the oracle's; on the other %.1f %% they are the reference's formula at the other exit (~3e-4 apart) |"
        % (100 * (1 - flip), 100 * flip),
        "| **the reference's own figure workload** (`P = diag(exp(U(−10,10)))`, eps 1e-10) | %.1f ms per 65 536 QP forwards, bound by ONE problem that "
        "takes %d iterations (a lone wave: ≈ %.2f µs per iteration; 4.2–5.2 ms from run to run) |"
        % (fig["qp_fwd_ms"], fig["qp_iterations"]["max"], fig["qp_fwd_ms"] * 1e3 / fig["qp_iterations"]["max"]),
    ])
    print(head + table)
    print()
    print(result)
    if "--write" in sys.argv:
        for fn, a, b, text in (("DESIGN.md", "<!-- measured:begin -->", "<!-- measured:end -->", head + table),
                               ("README.md", "<!-- result:begin -->", "<!-- result:end -->", result)):
            p = os.path.join(ROOT, fn)
            s = open(p).read()
            assert a in s and b in s, (fn, "markers missing")
            s = s[:s.index(a) + len(a)] + "\n" + text + "\n" + s[s.index(b):]
            open(p, "w").write(s)
            print("wrote", fn)


if __name__ == "__main__":
    main()
