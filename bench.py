#!/usr/bin/env python3
"""bench.py -- benchmark of the batched QP/QCQP hot path on MI355X.

    python bench.py [--config {2,3,4,5}] [--gpus N] [--steps K] [--warmup W] [--repeats R] [--streams {1,2}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--config 4]

Metric (BASELINE.json): QP+QCQP solves/sec (fwd+bwd) per GPU; achieved HBM GB/s vs roofline.  A STEP is one
pass of the hot path over one batch of synthetic input; inputs are resident in HBM when the timed region starts,
outputs go to preallocated device buffers, every launch goes through the C ABI (include/diffqcqp_hip.h).

Workloads (`--config k` = k-th entry of BASELINE.json `configs`, 1-based as in BASELINE.md section 4):
  default   the headline: per GPU and step, QP forward+backward [configs[1] + its backward] AND QCQP
            forward+backward [configs[2]] on B=65536, N=8, diagonal P in the (B,8,8) layout = 2*B solves.  The two
            families are independent problems on DISTINCT inputs; their launch chains go to two HIP streams
            (--streams 1: one stream) and swap streams every step (steps work on distinct buffer sets, so both streams
            carry the same work; --alternate-streams 0: fixed assignment).  Weak scaling.
  2         B=65536 N=8 diagonal-P QP, forward only                                   (one launch per step)
  3         B=65536 N=8 QCQP forward+backward
  4         B=262144 N=32 QP forward+backward, the batch SPLIT over the ranks (strong scaling); `value` includes the
            RCCL all-gather of x in every step (behind the forward, beside the backward), the rates without it and
            with it serialised behind the backward are reported alongside
  5         B=65536 N=64 dense-P QP forward+backward (P = S S^T/64 + 0.1 I), through DQQ_P_AUTO as QPFn2 calls it
  8         `qp_pair`: B=65536 N=8 QP forward+backward, ONE stream -- north_star's target sentence ("N = 8 QP fwd+bwd
            solves/s on 1 x MI355X with its HBM fraction") as one timed step;   9  `qp_pair_large`: the same at
            B=1048576, the batch size that fills the chip
Timing: W warm-up steps, then R regions of EXACTLY K steps, each bracketed by barrier + torch.cuda.synchronize()
on both sides and reduced with MAX over the ranks; `ms_per_step` / `value` are the MEDIAN region, so a short K is not a
single sub-millisecond sample.  Step k works on input/output set k mod nsets (> 768 MiB of sets in total): a training
loop presents new data every step, so `value` is the step that finds nothing in the 256 MiB Infinity Cache ("cold");
the same step on one set, step after step, is reported beside it (`hot_*`).  --hot-only: one set.

Output.  Rank 0 prints ONE line on stdout: the contract keys, `config` (workload + <= 20 scalars), `roofline` (<= 24
scalars), `cpu_baseline` -- at most 6000 bytes, strict JSON, nothing nested (contract_line()).  The FULL record -- per-launch
brackets, the sub-records below, the environment -- goes to bench_details.json next to this file (--details PATH) and,
prefixed with "[bench details] ", to stderr.

Default run (no --config): on ONE GPU the headline step, followed by short runs of the two qp_pair workloads, BASELINE
configs 2, 3, 4, 5 (`per_config`: 3 regions each, dominant kernel, roofline fractions, CPU baseline) and the dense-P check
(`dense_p_n8`); their step times and fractions are scalars of the line.  Fractions: every key WITHOUT `algorithmic` in
its name is computed from the bytes the launches really move (the backward takes the forward's 8N-byte verified diagonal
instead of re-reading the 8N^2-byte P) and is physically bounded by 6.29 / 8.0 = 0.79; SURVEY.md 8(d)'s algorithmic bytes
(which count that read) give the `*_algorithmic*` keys, which can exceed 1.  Under torch.distributed.run (WORLD_SIZE set; any
number of ranks) the line is the SAME headline step on every rank (weak scaling, no data-path collective: value(N) / (N
value(1)) is the scaling efficiency); `with_gather` = the same step with the path's optional exchange step -- the RCCL
all-gather of both families' x, behind their forwards, beside their backwards --, and configs[3], the batch SPLIT over the
ranks (strong scaling, with / without / serialised gather), is the sub-record `strong_config4`.

  roofline      the launch with the largest mean duration: algorithmic bytes / its duration from
                HIP events on the launch stream vs 8 TB/s; `moved_frac` = on the bytes that launch really reads + writes;
                `traffic` = HBM bytes of that launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), quoted from the
                committed profiles/pmc_latest*.json (`traffic_source`; `pmc_matches_build` in the details says whether
                that summary is of this build) or measured now with --live-pmc; `valu_busy_frac` = the share of the
                kernel's cycles in which a SIMD's VALU was executing (counters of one run, only when they are of this
                build); for the compute-bound config 5 also the FP64 figure (`fp64`)
  cpu_baseline  the oracle (C port of the reference algorithm) on this box's host cores, bounded sample; also the
                reference's execution model -- a Python loop with one FFI call per problem (`python_loop_value`)
  details       kernels (per-launch breakdown), hot, single_stream, per_config, dense_p_n8, survey_8d_extras, environment
                (clocks / power cap: boxes of the pool differ by up to ~9 %)
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Once RCCL has been
# initialised it holds some of them, and the two streams of the headline step end up on ONE queue: their kernels
# serialise (measured, one rank: 80.6 us per step instead of 59.5).  Eight queues keep the chains concurrent.
# Must be in the environment before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Kernel arguments in device memory (the runtime's default on this image: unset and "1" measure alike, 56.7 us per
# headline step; "0" -- arguments fetched from host memory at every launch -- 64.2 us: six launches per step).  Pinned
# here so that a box whose runtime defaults differently measures the same thing; an explicit setting is respected.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EPS, MAX_ITER, MU_PROX = 1e-7, 1000, 1e-7
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md chip table (6290 GB/s measured copy)
FP64_PEAK_TFLOPS = 78.6   # MI355X datasheet, vector = matrix FP64 (not in the local guide; the measured issue
                          # floor, profiles/fp64_issue_ubench.txt, 2.08 ns per wave64 FMA per SIMD, is 63 TFLOP/s)
F64 = torch.float64


def algo_bytes(kind, N, pas):
    """Algorithmic bytes per problem, SURVEY.md 8(d): float64, dense (B,N,N) P layout, warm_start excluded."""
    w, nc = 8, N // 2
    if pas == "fwd":
        return N * N * w + 2 * N * w + (2 * nc * w if kind == "qcqp" else 0)
    b = 2 * N * N * w + 4 * N * w
    return b + (4 * nc * w if kind == "qcqp" else 0)


def moved_bytes(kind, N, pas, diag_handoff):
    """Bytes the launch really reads + writes.  With the diagonal hand-off (DQQ_P_AUTO on diagonal tiles) the
    forward additionally leaves pdiag (8N) + a flag byte, and the backward reads those instead of P."""
    b = algo_bytes(kind, N, pas)
    if not diag_handoff:
        return b
    return b + 8 * N + 1 if pas == "fwd" else b - 8 * N * N + 8 * N + 1


def csrc_sha16():
    """The build's source hash (diffqcqp_amd/build.py: source_sha16) -- what a counter summary under profiles/ records."""
    from diffqcqp_amd import build
    return build.source_sha16()


class Chain:
    """One problem family on one batch: forward [+ backward] launches with every pointer resolved once."""

    def __init__(self, kind, B, N, structure, backward, dev, seed, nsets=1, layout=0):
        from diffqcqp_amd import _capi, ops
        self.lib, self.kind, self.B, self.N, self.backward, self.layout = _capi.lib(), kind, B, N, backward, layout
        self.structure, self.dev = structure, dev
        self.seed = seed
        self.sets = [self._make_set(seed + 104729 * s) for s in range(nsets)]
        self.names = [kind + "_fwd"] + ([kind + "_bwd"] if backward else [])
        self.ws = {}
        self.calls = {}
        self.ops, self.capi = ops, _capi
        self.handoff = structure == "diag" and (layout & 0xff) == 0
        self.kidx = 0 if kind == "qp" else 1
        self.hinted = layout == 0 and N <= 8

    def _make_set(self, seed):
        """inputs are generated ON THE CPU with torch.Generator().manual_seed(seed) -- seed = 1000 + the
        BASELINE config number for set 0 of rank 0 -- in the order P-values, q, [l_n, mu], grad_l, then copied to the device,
        so that the numbers are reproducible from the survey's recipe (and the CPU arm sees identical bits)."""
        B, N, dev = self.B, self.N, self.dev
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.rand(*s, generator=g, dtype=F64)
        e = lambda *s: torch.empty(*s, dtype=F64, device=dev)
        if self.structure == "diag":      # p ~ U(0.1, 1.1) -> diag_embed
            P = torch.diag_embed((r(B, N) + 0.1).to(dev)).contiguous()
        else:                             # cfg 5: S ~ U(0,1)^(N x N), P = S S^T / N + 0.1 I
            S = r(B, N, N).to(dev)
            P = torch.bmm(S, S.transpose(1, 2)) / N
            del S
            P.diagonal(dim1=1, dim2=2).add_(0.1)
        t = {"P": P, "q": (2 * r(B, N, 1) - 1).to(dev), "x": e(B, N, 1)}
        if self.kind == "qcqp":
            t["l_n"], t["mu"] = r(B, N // 2, 1).to(dev), r(B, N // 2, 1).to(dev)
        if self.backward:
            t["g"] = torch.randn(B, N, 1, generator=g, dtype=F64).to(dev)
            t["gP"], t["gq"] = e(B, N, N), e(B, N, 1)
            if self.kind == "qcqp":
                t["gl"], t["gm"] = e(B, N // 2, 1), e(B, N // 2, 1)
        t["pdiag"], t["flags"] = e(B, N), torch.empty(B, dtype=torch.uint8, device=dev)
        return t

    def bytes_per_set(self):
        return sum(v.numel() * v.element_size() for v in self.sets[0].values())

    def add_sets(self, n):
        """n more input / output sets (distinct seeds): a step that rotates over them never finds its data in the 256 MiB
        Infinity Cache."""
        self.sets += [self._make_set(self.seed + 104729 * s) for s in range(len(self.sets), len(self.sets) + n)]

    def workspace(self, stream):
        if stream not in self.ws:
            self.ws[stream] = self.ops._workspace(self.dev, self.B, stream)
        return self.ws[stream]

    def _call(self, which, stream, s, flags=0, report=None):
        """(C-ABI function, full argument tuple) of one launch -- every pointer resolved ONCE (the host
        side of a 56 us step must not re-resolve 16 data_ptr() per call).  flags / report: the caller's side of the hint
        protocol of include/diffqcqp_hip.h (what diffqcqp_amd.ops does for its callers)."""
        t, L, p = self.sets[s], self.lib, (lambda a: a.data_ptr())
        ws = self.workspace(stream)
        wsb, B, N = ws.numel() * 4, self.B, self.N
        lay = self.layout | flags
        # the verified-diagonal hand-off exists for DQQ_P_AUTO only (diffqcqp_amd/qcqp.py: _cache_for): a batch declared
        # dense passes no pdiag / flags (with them the C ABI would clear the flags with a memset launch per forward)
        pd, fl = (p(t["pdiag"]), p(t["flags"])) if (self.layout & 0xff) == 0 else (None, None)
        if which == 0 and self.kind == "qp":
            return L.dqq_qp_fwd_f64, (p(t["P"]), p(t["q"]), p(t["x"]), B, N, EPS, MU_PROX, MAX_ITER, 1, lay, None,
                                      pd, fl, p(ws), wsb, stream)
        if which == 0:
            return L.dqq_qcqp_fwd_f64, (p(t["P"]), p(t["q"]), p(t["l_n"]), p(t["mu"]), p(t["x"]), B, N, EPS, MU_PROX,
                                        MAX_ITER, 1, lay, None, pd, fl, p(ws), wsb, stream)
        if self.kind == "qp":
            return L.dqq_qp_bwd_f64, (p(t["P"]), p(t["q"]), p(t["x"]), p(t["g"]), p(t["gP"]), p(t["gq"]), B, N, 1e-10,
                                      lay, None, pd, fl, report, p(ws), wsb, stream)
        return L.dqq_qcqp_bwd_f64, (p(t["P"]), p(t["q"]), p(t["l_n"]), p(t["mu"]), p(t["x"]), p(t["g"]), p(t["gP"]),
                                    p(t["gq"]), p(t["gl"]), p(t["gm"]), None, None, B, N, 1e-10, lay, None,
                                    pd, fl, report, p(ws), wsb, stream)

    def launch(self, which, stream, s=0):
        flags, report = 0, None
        if self.hinted:   # DQQ_P_AUTO, N <= 8: read this (device, kind, N)'s report word, derive the flags (pure function)
            flags, report = self.capi.hint(self.kidx, which, self.N, self.B, self.dev.index)
        key = (which, stream, s, flags)
        c = self.calls.get(key)
        if c is None:
            c = self.calls[key] = self._call(which, stream, s, flags, report)
        rc = c[0](*c[1])
        if rc != 0:
            raise RuntimeError("launch %s failed with %d" % (self.names[which], rc))

    def run(self, stream, s=0):
        for w in range(len(self.names)):
            self.launch(w, stream, s)

    # ---- checks and the CPU arm (rank 0 only; the oracle is the checker / the baseline, never the product)
    def host_sample(self, n, s=0):
        t = self.sets[s]
        keys = [k for k in ("P", "q", "l_n", "mu", "g") if k in t]
        return {k: t[k][:n].cpu().numpy() for k in keys}

    def check(self, n=1024):
        from oracle import oracle as O
        n = min(n, self.B)
        h, t, nt = self.host_sample(n), self.sets[0], O.max_threads()
        err = {}
        if self.kind == "qp":
            xo, _ = O.qp_fwd_batch(h["P"], h["q"], EPS, MAX_ITER, MU_PROX, nthreads=nt)
        else:
            xo, _ = O.qcqp_fwd_batch(h["P"], h["q"], h["l_n"], h["mu"], EPS, MAX_ITER, MU_PROX, nthreads=nt)
        err["x"] = float(np.abs(t["x"][:n].cpu().numpy() - xo).max())
        if self.backward and self.kind == "qp":
            gq = O.qp_bwd_batch(h["P"], h["q"], xo, h["g"], nthreads=nt)[1]
            err["grad_q"] = float(np.abs(t["gq"][:n].cpu().numpy() - gq).max())
        elif self.backward:
            # the reference's refinement exit (1 or 3 bodies) is decided by rounding noise (Solver.cpp:32-41): end to
            # end the gradients are compared where the exits agree; tests/ show the flipped ones are the reference
            # formula at the other exit, and that the backward is bit-exact on identical x
            ref = O.qcqp_bwd_batch(h["P"], h["q"], h["l_n"], h["mu"], xo, h["g"], nthreads=nt)
            st = self.ops.qcqp_backward(t["P"][:n], t["q"][:n], t["l_n"][:n], t["mu"][:n], t["x"][:n], t["g"][:n],
                                        need=(False, True, False, False), return_steps=True)[-1].cpu().numpy()
            same = st == ref[-1]
            d = np.abs(t["gq"][:n].cpu().numpy() - ref[1]).max(axis=(1, 2)) / np.maximum(1.0, np.abs(ref[1]).max(axis=(1, 2)))
            err["grad_q_rel_where_refinement_exit_agrees"] = float(d[same].max())
            err["refinement_exit_flip_rate"] = float(1.0 - same.mean())
        return err

    def cpu_solves_per_s(self, n, nthreads):
        from oracle import oracle as O
        h = self.host_sample(n)
        t0 = time.perf_counter()
        if self.kind == "qp":
            x, _ = O.qp_fwd_batch(h["P"], h["q"], EPS, MAX_ITER, MU_PROX, nthreads=nthreads)
            if self.backward:
                O.qp_bwd_batch(h["P"], h["q"], x, h["g"], nthreads=nthreads)
        else:
            x, _ = O.qcqp_fwd_batch(h["P"], h["q"], h["l_n"], h["mu"], EPS, MAX_ITER, MU_PROX, nthreads=nthreads)
            if self.backward:
                O.qcqp_bwd_batch(h["P"], h["q"], h["l_n"], h["mu"], x, h["g"], nthreads=nthreads)
        return n / (time.perf_counter() - t0)


WORKLOADS = {
    # key: (description, [(kind, N, structure, backward)], B_total, scaling, cpu sample per chain)
    0: ("per GPU and step: B=65536 N=8 diagonal-P (dense (B,8,8) layout) QP forward+backward [BASELINE configs[1] + "
        "backward] and B=65536 N=8 QCQP forward+backward [configs[2]] on distinct inputs; value counts one "
        "forward+backward as one solve", [("qp", 8, "diag", True), ("qcqp", 8, "diag", True)], 65536, "weak", 65536),
    2: ("BASELINE configs[1]: B=65536 N=8 diagonal-P QP, forward only", [("qp", 8, "diag", False)], 65536, "weak", 65536),
    3: ("BASELINE configs[2]: B=65536 N=8 QCQP (friction cones), forward+backward", [("qcqp", 8, "diag", True)], 65536,
        "weak", 65536),
    4: ("BASELINE configs[3]: B=262144 N=32 diagonal-P QP forward+backward, batch split over the ranks, RCCL "
        "all-gather of x after every step", [("qp", 32, "diag", True)], 262144, "strong", 16384),
    5: ("BASELINE configs[4]: B=65536 N=64 dense-P QP (P = S S^T/64 + 0.1 I) forward+backward",
        [("qp", 64, "dense", True)], 65536, "weak", 2048),
    # not a BASELINE config: what a real contact problem presents -- a dense 8x8 Delassus matrix (P = S S^T/8 + 0.1 I)
    6: ("dense-P B=65536 N=8 QCQP forward+backward through DQQ_P_AUTO (what QCQPFn2 passes)",
        [("qcqp", 8, "dense", True, 0)], 65536, "weak", 16384),
    7: ("dense-P B=65536 N=8 QCQP forward+backward, P declared dense (DQQ_P_DENSE)",
        [("qcqp", 8, "dense", True, 1)], 65536, "weak", 16384),
    11: ("dense-P B=65536 N=8 QCQP forward+backward through DQQ_P_AUTO | DQQ_F_EXPECT_DENSE given EXPLICITLY by the caller (no "
         "report word): what a first call or a captured graph runs when the caller says what it expects",
         [("qcqp", 8, "dense", True, 0x200)], 65536, "weak", 16384),
    # north_star's target sentence as ONE timed step: N = 8 QP forward+backward on one stream
    8: ("qp_pair: B=65536 N=8 diagonal-P (dense (B,8,8) layout) QP forward+backward on one stream [BASELINE configs[1] + its "
        "backward]", [("qp", 8, "diag", True)], 65536, "weak", 16384),
    9: ("qp_pair_large: B=1048576 N=8 diagonal-P (dense (B,8,8) layout) QP forward+backward on one stream (the batch that "
        "fills the chip)", [("qp", 8, "diag", True)], 1048576, "weak", 16384),
    # one rank's share of BASELINE configs[3] on an 8-GPU node (32768 of the 262144 problems), on this one GPU: what the
    # kernels leave of linear strong scaling BEFORE the exchange step -- measured, not a multi-GPU figure
    10: ("one eighth of BASELINE configs[3] (B=32768 N=32 diagonal-P QP forward+backward): the shard a rank of an 8-GPU run "
         "solves", [("qp", 32, "diag", True)], 32768, "weak", 4096),
}


def workload_seed(cfg, family):
    """torch.Generator().manual_seed(1000 + cfg), cfg = the BASELINE config (1-based) a family's inputs belong
    to: the headline's QP family is configs[1]'s (1002), its QCQP family configs[2]'s (1003); qp_pair* are configs[1]'s, the
    one-eighth shard configs[3]'s; the dense 8 x 8 workloads (not BASELINE configs) take 1006."""
    base = {0: (2, 3), 2: (2,), 3: (3,), 4: (4,), 5: (5,), 6: (6,), 7: (6,), 11: (6,), 8: (2,), 9: (2,), 10: (4,)}[cfg]
    return 1000 + base[family]


def gpu_environment():
    """Clocks, power cap and queue setting of this box (boxes of the pool differ by up to ~9 %).
    Best effort: rocm-smi may be missing or slow; nothing here is required for the measurement."""
    import subprocess
    env = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
           "HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG"), "device": torch.cuda.get_device_name(0)}
    try:
        from diffqcqp_amd import _capi
        env["dqq_hints"] = bool(_capi._hints_on)   # the caller-side route hints of _capi.py (DQQ_FEEDBACK=0 turns them off)
    except Exception:
        pass
    try:
        p = torch.cuda.get_device_properties(0)
        env.update({"compute_units": p.multi_processor_count, "clock_rate_khz_reported": getattr(p, "clock_rate", None)})
    except Exception:
        pass
    try:
        txt = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showmaxpower", "--showpower", "--showperflevel",
                              "--json"], capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(txt).values()))
        keep = ("sclk", "mclk", "fclk", "Max Graphics Package Power", "Average Graphics Package Power",
                "Current Socket Graphics Package Power", "Performance Level")
        env["rocm_smi"] = {k: v for k, v in card.items() if any(t.lower() in k.lower() for t in keep)}
    except Exception as e:  # noqa: BLE001
        env["rocm_smi"] = "unavailable (%s)" % type(e).__name__
    return env


def _pmc_bench_key(name):
    """bench.py launch name of a kernel from its demangled name: <family>_<pass>; the family is the KIND template argument
    (0 qp, 1 qcqp) -- the same mapping tools/summarize_prof.py uses for the committed summaries."""
    import re
    m = re.search(r"dqq::(\w+)<([^>]*)>", name)
    if not m:
        return None, name[:80]
    s = "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    if not s.startswith(("fwd_", "bwd_")):
        return None, s
    k = re.search(r"<(\d)", s)
    fam = {"0": "qp", "1": "qcqp"}.get(k.group(1) if k else "0")
    if "_qp_kernel" in s:
        fam = "qp"
    return (None if fam is None else "%s_%s" % (fam, s[:3])), s


def live_pmc(cfg, timeout=240):
    """HBM traffic of this workload's launches, measured now: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: they do not fit
    one pass -- MI355X_MICROARCH.md counter table -- and are collected with --kernel-trace only, never with other trace
    domains) over a CHILD process that issues each C-ABI call of the workload a few times (`--pmc-child`).  Returns
    {launch name: {"hbm_bytes_per_launch", "kernels", ...}}, {"_error": ...} or None (rocprofv3 not there)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="dqq_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DQQ_BENCH_FORCE_DIST"):
        env.pop(k, None)
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "p", "--", sys.executable,
                                os.path.abspath(__file__), "--pmc-child", str(cfg)], capture_output=True, text=True,
                               timeout=timeout, cwd="/tmp", env=env)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith("_results.db")]
            if r.returncode != 0 or not dbs:
                return {"_error": "rocprofv3 --pmc %s: rc %d, %s" % (counter, r.returncode, (r.stderr or "")[-300:])}
            c = sqlite3.connect(dbs[0])
            for name, val, n in c.execute("select kernel_name, avg(value), count(*) from counters_collection "
                                          "where counter_name = ? group by kernel_name", (counter,)):
                per.setdefault(name, {})[counter] = (val, n)
            c.close()
    except Exception as e:  # noqa: BLE001
        return {"_error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for name, v in per.items():
        key, short = _pmc_bench_key(name)
        if key is None or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        # bytes per launch (one C-ABI call = the kernels it enqueues): dispatch-weighted sum over the kernels that served
        # this launch name / the dispatches of the most frequent one (a route taken once does not count as if always taken)
        e = res.setdefault(key, {"rd": 0.0, "wr": 0.0, "kernels": [], "dispatches_sampled": 0})
        n = int(v["FETCH_SIZE"][1])
        e["rd"] += 2.0 * v["FETCH_SIZE"][0] * 1024 * n
        e["wr"] += v["WRITE_SIZE"][0] * 1024 * int(v["WRITE_SIZE"][1])
        e["kernels"].append(short)
        e["dispatches_sampled"] = max(e["dispatches_sampled"], n)
    for e in res.values():
        rd, wr = e.pop("rd") / e["dispatches_sampled"], e.pop("wr") / e["dispatches_sampled"]
        e.update({"hbm_bytes_per_launch": rd + wr, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr})
    return res


def pmc_child(cfg):
    """`--pmc-child CFG`: issue every C-ABI call of workload CFG five times on one stream and exit -- the process the
    counter passes of live_pmc() profile.  Prints nothing."""
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from diffqcqp_amd import _capi, ops
    _capi.lib()
    _, families, B_total, _, _ = WORKLOADS[cfg]
    chains = [Chain(f[0], B_total, f[1], f[2], f[3], dev, workload_seed(cfg, i),
                    layout=(f[4] if len(f) > 4 else 0)) for i, f in enumerate(families)]
    sh = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        for c in chains:
            c.run(sh)
    torch.cuda.synchronize()


def measure(cfg, args, ctx, light=False):
    """One workload -> its full record (rank 0; None on the other ranks).  light: a sub-record of the default run -- 3
    regions, no hot / single-stream context."""
    rank, world, dev, use_dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["use_dist"]
    dist, parallel, _capi = ctx["dist"], ctx["parallel"], ctx["capi"]
    desc, families, B_total, scaling, cpu_n = WORKLOADS[cfg]
    if scaling == "strong":
        lo, hi = parallel.shard_bounds(B_total, rank, world)
        B_rank = hi - lo
    else:
        B_rank = B_total
    # the path's one exchange step: the all-gather of x.  configs[3] (the batch SPLIT over the ranks) carries
    # it in `value`; the headline (weak scaling: every rank its own B problems per family -- the solve itself needs no
    # collective) is also timed with it when the run is distributed: `with_gather`
    gather = cfg == 4
    gather_extra = cfg == 0 and use_dist
    B_gather = B_total if scaling == "strong" else B_rank * world      # rows of a gathered x
    steps, repeats, warmup = max(args.steps, 1), max(args.repeats, 1), max(args.warmup, 1)
    if cfg == 5 and args.steps == 100 and args.repeats == 10:
        steps, repeats = 5, 5          # a step is ~6 ms of a 4.3 GB working set: bound the default run
    if cfg == 4 and args.steps == 100:
        steps = 20
    if light:
        steps, repeats, warmup = {2: 50, 3: 50, 4: 10, 5: 3, 0: 50, 6: 10, 7: 10, 8: 50, 9: 10, 10: 30, 11: 10}[cfg], 3, 3

    chains = [Chain(f[0], B_rank, f[1], f[2], f[3], dev, workload_seed(cfg, i) + 7919 * rank,
                    layout=(f[4] if len(f) > 4 else 0)) for i, f in enumerate(families)]
    # ---- rotating buffers: a training loop presents NEW data every step, so the step that `value` times never finds its
    # inputs or outputs in the 256 MiB Infinity Cache: every chain gets enough distinct input / output sets for > 768 MiB
    # in total and step k works on set k mod nsets.  (A workload whose single set is already that large needs no second
    # one.)  --hot-only: one set, the same buffers every step (the cache-resident figure, reported as `hot_*` otherwise)
    # Enough sets that even ONE launch repeated back to back (the per-launch runs below) wraps around more than 768 MiB of its
    # own buffers: the smallest launch of a step touches about P + three vectors of one family (38 MB at B = 65536, N = 8:
    # 21 sets; rotating over 5 sets, 168 MB of P, the forward runs found their matrices in the cache: 24.2 against 26.5 us).
    per_set = sum(c.bytes_per_set() for c in chains)
    per_launch = min(c.B * (c.N * c.N + 3 * c.N) * 8 for c in chains)
    nsets = 1 if (args.hot_only or per_launch >= 768 * 2**20) else min(32, max(3, int(np.ceil(768 * 2**20 / per_launch))))
    if nsets > 1 and nsets % 2:
        nsets += 1          # (even: with the families swapping streams every step a set keeps its stream)
    for c in chains:
        c.add_sets(nsets - 1)
    main_stream = torch.cuda.current_stream()
    sh = main_stream.cuda_stream
    side = ctx["side"] if (len(chains) == 2 and args.streams == 2) else None
    streams = [sh, side.cuda_stream if side is not None else sh]
    tstreams = [main_stream, side if side is not None else main_stream]   # the torch stream each chain is launched on
    x_all, gather_scratch = [None] * len(chains), [None] * len(chains)
    if (gather or gather_extra) and use_dist:   # the exchange step's buffers are the caller's: nothing is allocated inside a timed step
        rows = parallel.gather_scratch_rows(B_gather, world)
        for i, c in enumerate(chains):
            x_all[i] = torch.empty((B_gather, c.N, 1), dtype=F64, device=dev)
            gather_scratch[i] = torch.empty((rows, c.N, 1), dtype=F64, device=dev) if rows else None

    alternate = bool(args.alternate_streams)
    ctr = [0]          # steps issued so far: step k works on set k mod nsets
    last_set = [0]

    def next_set():
        s = ctr[0] % nsets
        ctr[0] += 1
        last_set[0] = s
        return s

    def step_on(s):
        """One pass of the hot path over this rank's batch (all chains; set s of each)."""
        if side is not None:  # interleave so that both streams are fed; the longer chain (the QCQP) first
            # The chains are unequal (QCQP ~46 us of kernels, QP ~37): with a fixed family -> stream assignment the QP stream
            # runs ahead and the region ends with the QCQP chain alone on the chip.  Steps on DISTINCT buffer sets are
            # independent, so the families swap streams every step and both streams carry the same work: 60.3 -> 58.0 us per step
            # (three alternations on one box).  A set's parity fixes its stream (nsets even), so the re-use of a set's buffers is
            # stream-ordered.  --alternate-streams 0, or one set of buffers (--hot-only, the hot_* context): fixed assignment
            a, b = (1, 0) if (alternate and nsets > 1 and (s & 1)) else (0, 1)
            chains[1].launch(0, streams[b], s)
            chains[0].launch(0, streams[a], s)
            chains[1].launch(1, streams[b], s)
            chains[0].launch(1, streams[a], s)
        else:
            for c in chains:
                c.run(sh, s)

    def step():
        step_on(next_set())

    def hot_step():
        step_on(0)

    def drain():
        if side is not None:
            side.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    enq, enq_log = [0.0], []

    def region(fn, k):
        """Exactly k calls of fn between two barrier+synchronize brackets; MAX over the ranks, seconds.  The clock
        stops when this rank's work has completed (synchronize); the closing barrier follows and the MAX over the
        ranks is what it would have measured -- without adding RCCL's own barrier latency (~0.1 ms) to a region
        that may only be a millisecond long."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        enq[0] = time.perf_counter() - t0   # the host is done enqueueing here (no synchronise yet): host-bound iff ~ el
        # spin on the streams' completion before the synchronize() that closes the region: a blocked synchronize() wakes
        # up ~30 us after the GPU is done, which is the host's scheduler, not the path being measured; the region still
        # ends with synchronize() on both streams
        while not (main_stream.query() and (side is None or side.query())):
            pass
        drain()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        barrier()
        if use_dist:
            tm = torch.tensor([el], dtype=F64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el = float(tm.item())
        enq_log.append((el, enq[0], k))
        return el

    def gather_x(i, async_op, s):
        """All-gather of chain i's x (set s), ordered behind the work already enqueued on that chain's stream (RCCL
        synchronises with torch's CURRENT stream: the side chain's collective is issued with its stream current)."""
        with torch.cuda.stream(tstreams[i]):
            return parallel.gather_batch(chains[i].sets[s]["x"], B_gather, async_op=async_op, out=x_all[i],
                                         scratch=gather_scratch[i])

    def step_and_gather_serial():
        s = next_set()
        step_on(s)
        if use_dist:
            for i in range(len(chains)):
                gather_x(i, False, s)

    def step_and_gather():
        """The path's one exchange step where it belongs: x is complete after the FORWARD, so its all-gather (RCCL's own
        stream, ordered behind the forward) travels over xGMI while the backward of the same problems runs -- the
        backward needs this rank's x only.  The step ends when both are done (the launch streams wait for the
        collectives)."""
        if not use_dist:
            return step()
        s = next_set()
        order = list(range(len(chains)))[::-1]        # the longer chain (the QCQP of the headline) first
        for i in order:
            chains[i].launch(0, streams[i], s)
        works = [(i, gather_x(i, True, s)[1]) for i in order]
        for i in order:
            for w in range(1, len(chains[i].names)):
                chains[i].launch(w, streams[i], s)
        for i, work in works:
            with torch.cuda.stream(tstreams[i]):
                work.wait()

    timed = step_and_gather if gather else step

    # ---- per-launch pass FIRST: HIP events around every C-ABI call of the step, then runs of each call back to back --
    # about 10 ms of GPU work.  After an idle spell the chip takes 10-20 ms to come back to its clocks, and a driver run
    # with --warmup 5 --steps 20 has only 0.3 ms of warm-up before 12 ms of timed regions: run behind the diagnostics, the
    # timed regions measure the kernels, not the ramp.
    # A bracket holds one C-ABI CALL: one kernel for the N = 8 forwards (non-diagonal tiles are solved inside the kernel);
    # the kernel plus the launch that drains its work-list for the backwards and for N >= 16 (rocprofv3's per-kernel
    # durations, profiles/, are the kernel-alone figures).  Launch r works on set r mod nsets, as the steps do.
    nrep = 5 if cfg == 5 else (20 if cfg in (4, 9) else (30 if light else 100))
    launches = [(c, w) for c in chains for w in range(len(c.names))]
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in launches]
          for _ in range(nrep)]
    for s in range(nsets):      # (first launches: module load, function attributes, workspace; every set gets its x,
        for c, w in launches:   # pdiag and flags from a forward before any backward is timed alone)
            c.launch(w, sh, s)
    torch.cuda.synchronize()
    for r in range(nrep):
        for j, (c, w) in enumerate(launches):
            ev[r][j][0].record(main_stream)
            c.launch(w, sh, r % nsets)
            ev[r][j][1].record(main_stream)
    torch.cuda.synchronize()
    # the same launches in RUNS: nrep launches of one kernel back to back between two events.  A bracket around a single
    # launch also times the two event packets (~3 us on a 10-30 us kernel); a run's mean is the kernel plus its launch
    # gap and agrees with rocprofv3 to ~2 % (three runs per kernel, the median counts: one run in a few hundred catches a
    # multi-millisecond stall of the box)
    runs = []
    for c, w in launches:
        trio = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_stream)
            for r in range(nrep):
                c.launch(w, sh, r % nsets)
            e1.record(main_stream)
            trio.append((e0, e1))
        runs.append(trio)
    torch.cuda.synchronize()
    kernels = {}
    for j, (c, w) in enumerate(launches):
        ts = sorted(ev[r][j][0].elapsed_time(ev[r][j][1]) for r in range(nrep))
        mean_ms, pas = sorted(a.elapsed_time(b) for a, b in runs[j])[1] / nrep, ("fwd" if w == 0 else "bwd")
        ab, mb = algo_bytes(c.kind, c.N, pas) * c.B, moved_bytes(c.kind, c.N, pas, c.handoff) * c.B
        kernels[c.names[w]] = {"mean_us": mean_ms * 1e3, "bracketed_mean_us": sum(ts) / len(ts) * 1e3,
                               "bracketed_median_us": ts[len(ts) // 2] * 1e3,
                               "bracketed_min_us": ts[0] * 1e3, "bracketed_max_us": ts[-1] * 1e3,
                               "moved_bytes_per_launch": mb, "moved_GBps": mb / (mean_ms * 1e-3) / 1e9,
                               "algorithmic_bytes_per_launch": ab, "algorithmic_GBps": ab / (mean_ms * 1e-3) / 1e9}
    # ---- warm-up (also warms RCCL's all-gather)
    for _ in range(warmup):
        timed()
    drain()
    # ---- timed regions: R times EXACTLY K steps
    times = sorted(region(timed, steps) for _ in range(repeats))
    elapsed = times[len(times) // 2]
    host_enqueue_us = sorted(e / k for _, e, k in enq_log)[len(enq_log) // 2] * 1e6
    # How much of a K-step region is fill / drain / synchronise latency rather than steady state?  The same step in
    # regions of 5K steps: T(K) = F + s K from the two medians (the driver times --steps 20; the fixed ~90 us of a region
    # are 8 % of that)
    region_fit = None
    if not light and cfg != 4:
        kl = 5 * steps
        tl = sorted(region(timed, kl) for _ in range(3))[1]
        s_fit = (tl - elapsed) / (kl - steps)
        region_fit = {"long_region_steps": kl, "ms_per_step_long_region": tl / kl * 1e3,
                      "us_per_step_steady_state": s_fit * 1e6, "region_fixed_us": (elapsed - steps * s_fit) * 1e6}
    extra = {}
    units_per_step_all = B_total if scaling == "strong" else sum(c.B for c in chains) * world
    if gather_extra:   # the same regions WITH the exchange step: all-gather of every family's x behind its forward
        for _ in range(3):
            step_and_gather()
        tw = sorted(region(step_and_gather, steps) for _ in range(max(repeats // 2, 1)))
        extra["with_gather"] = {"ms_per_step": tw[len(tw) // 2] / steps * 1e3, "value": units_per_step_all * steps / tw[len(tw) // 2],
                                "allgather_bytes_per_rank": sum(c.B * c.N * 8 for c in chains),
                                "rccl_world": dist.get_world_size(),
                                "note": "x of every rank all-gathered in every step (one collective per family, behind its forward, "
                                        "beside its backward); two torch.distributed calls per step: host-side cost included"}
    if (gather or gather_extra) and use_dist:
        torch.cuda.synchronize()
        lo_r, hi_r = parallel.shard_bounds(B_gather, rank, world)
        for i, c in enumerate(chains):
            assert x_all[i].shape[0] == B_gather and torch.equal(x_all[i][lo_r:hi_r], c.sets[last_set[0]]["x"]), "all-gather of x"
    if gather and use_dist:   # the same regions with the gather behind the whole step instead of beside the backward
        tg = sorted(region(step_and_gather_serial, steps) for _ in range(max(repeats // 2, 1)))
        extra["gather_after_backward"] = {"ms_per_step": tg[len(tg) // 2] / steps * 1e3,
                                          "value": units_per_step_all * steps / tg[len(tg) // 2],
                                          "note": "all-gather of x serialised behind the backward"}
    if gather:   # the same regions without the exchange step
        tn = sorted(region(step, steps) for _ in range(max(repeats // 2, 1)))
        extra["without_gather"] = {"ms_per_step": tn[len(tn) // 2] / steps * 1e3,
                                   "value": units_per_step_all * steps / tn[len(tn) // 2],
                                   "allgather_bytes_per_rank": sum(c.B * c.N * 8 for c in chains),
                                   "rccl_world": dist.get_world_size() if use_dist else 1}
    if side is not None and not light and not args.no_hot:   # context: the same steps strictly on one stream
        save, side = side, None
        t1 = sorted(region(step, steps) for _ in range(3))
        side = save
        extra["single_stream"] = {"ms_per_step": t1[1] / steps * 1e3,
                                  "value_this_rank": sum(c.B for c in chains) * steps / t1[1],
                                  "note": "all launches of a step on one stream"}
    # ---- hot variant: the same step on ONE set, step after step (everything a step touches fits the Infinity Cache when the
    # set is < 256 MiB) -- context, never `value`
    if nsets > 1 and (not light or cfg == 8) and not args.no_hot:   # (qp_pair: north_star's sentence, cold AND hot in the line)
        for _ in range(3):
            hot_step()
        drain()
        th = sorted(region(hot_step, steps) for _ in range(max(repeats // 2, 3)))
        extra["hot"] = {"ms_per_step": th[len(th) // 2] / steps * 1e3,
                        "value": units_per_step_all * steps / th[len(th) // 2],
                        "note": "the same buffers every step (%.0f MB: resident in the 256 MiB Infinity Cache)" % (per_set / 1e6)}

    dom = max(kernels, key=lambda k: kernels[k]["mean_us"])
    step_algo = sum(k["algorithmic_bytes_per_launch"] for k in kernels.values())
    step_moved = sum(k["moved_bytes_per_launch"] for k in kernels.values())
    step_s = elapsed / steps
    # Naming rule: a fraction WITHOUT `algorithmic` in its name is computed from bytes that move and cannot exceed
    # 6.29 / 8.0 = 0.79; `frac` / `achieved` (the contract's keys: SURVEY 8(d) algorithmic bytes of the dominant launch) obey
    # it too because the dominant launch is a forward, whose algorithmic bytes are fewer than the bytes it moves.
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": kernels[dom]["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": kernels[dom]["algorithmic_GBps"] / HBM_PEAK_GBS, "traffic": None,
        "moved_frac": kernels[dom]["moved_GBps"] / HBM_PEAK_GBS,
        # the whole step with the bytes that really move (the backward takes the forward's verified diagonal instead of P)
        "step_moved_frac": step_moved / step_s / 1e9 / HBM_PEAK_GBS,
        "step_moved_GBps": step_moved / step_s / 1e9,
        "moved_GBps": kernels[dom]["moved_GBps"],
        "step_algorithmic_frac": step_algo / step_s / 1e9 / HBM_PEAK_GBS,
        "step_algorithmic_GBps": step_algo / step_s / 1e9,
        "step_algorithmic_GBps_kernels_alone": step_algo / (sum(k["mean_us"] for k in kernels.values()) * 1e-6) / 1e9,
        "timing": "HIP events on the launch stream around a run of %d back-to-back C-ABI calls of the launch (mean)" % nrep,
        "host_enqueue_us_per_step": host_enqueue_us,
        "kernels_sum_us": sum(k["mean_us"] for k in kernels.values()),
        "buffer_sets": nsets,
    }
    for k, v in kernels.items():
        roofline["kernel_us_" + k] = v["mean_us"]
    if region_fit:
        roofline.update(region_fit)
    # HBM traffic and the VALU counters: quoted from the committed counter summary of the same workload (tools/profile.sh ->
    # profiles/pmc_latest*.json), with its provenance; `pmc_matches_build` says whether that summary was taken on THIS
    # build (the sha of csrc/ it records); the counter-derived VALU figures are dropped when it was not.  --live-pmc: two
    # rocprofv3 passes spawned by this run over its own launches measure the traffic now
    tag = {0: "", 2: "_cfg2", 3: "_cfg3", 4: "_cfg4", 5: "_cfg5", 6: "_cfg6", 7: "_cfg7", 8: "_cfg2", 9: None, 10: None, 11: None}[cfg]
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest%s.json" % tag) if tag is not None else None
    if pmc_path and os.path.exists(pmc_path):
        try:
            allp = json.load(open(pmc_path))
            pmc = allp.get(dom, {})
            fresh = allp.get("_csrc_sha16") == csrc_sha16()
            roofline["traffic"] = pmc.get("hbm_bytes_per_launch")
            roofline["traffic_source"] = "profiles/%s tag %s" % (os.path.basename(pmc_path), allp.get("_tag", "?"))
            roofline["pmc_matches_build"] = fresh
            if roofline["traffic"]:
                roofline["traffic_over_algorithmic"] = roofline["traffic"] / kernels[dom]["algorithmic_bytes_per_launch"]
            if fresh:
                # what binds the N = 8 forwards is VALU issue, not HBM: the share of the kernel's cycles in
                # which a SIMD's VALU was executing, from the counters of one run (SQ_ACTIVE_INST_VALU: quad-cycles summed
                # over the SIMDs; SQ_BUSY_CYCLES: cycles summed over the shader engines) -- a measured share, <= 1
                for k in ("valu_busy_frac", "valu_lane_utilisation", "SQ_WAIT_INST_ANY_frac", "SQ_WAIT_ANY_frac"):
                    if k in pmc:
                        roofline[k if k.startswith("valu") else "pmc_" + k] = pmc[k]
        except Exception:
            pass
    if ctx.get("live_pmc") and rank == 0 and world == 1 and not light:
        live = live_pmc(cfg)
        if live and dom in live:
            roofline["traffic"] = live[dom]["hbm_bytes_per_launch"]
            roofline["traffic_source"] = "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run"
            roofline["traffic_kernels"] = live[dom]["kernels"]
            roofline["traffic_over_algorithmic"] = live[dom]["hbm_bytes_per_launch"] / kernels[dom]["algorithmic_bytes_per_launch"]
            roofline["live_pmc"] = live
        elif live is not None:
            roofline["live_pmc_error"] = live.get("_error", "dominant launch not found in the counter pass")
    if cfg == 5:
        # compute-bound: FP64 flops with the reference's cost profile: per problem 2N^2 per mat-vec
        # (iterations + 11 power-iteration products), 2.33 N^3 per factorisation + explicit inverse (3.5 per solve
        # on this family), backward A^T A + LLT + inverse 4.33 N^3
        c = chains[0]
        it = c.ops.qp_forward(c.sets[0]["P"], c.sets[0]["q"], EPS, MAX_ITER, return_iters=True)[1].double().mean().item()
        n = c.N
        f_fwd = (2 * n * n * (it + 11) + 3.5 * 2.33 * n ** 3) * c.B
        f_bwd = (4.33 * n ** 3 + 2 * n * n * 5) * c.B
        fl = {"qp_fwd": f_fwd, "qp_bwd": f_bwd}
        roofline["fp64"] = {
            "bound": "mfma", "unit": "TFLOP/s", "peak": FP64_PEAK_TFLOPS, "mean_iterations": it,
            "algo_flops_per_launch": fl,
            "achieved": {k: fl[k] / (kernels[k]["mean_us"] * 1e-6) / 1e12 for k in fl},
            "frac": fl[dom] / (kernels[dom]["mean_us"] * 1e-6) / 1e12 / FP64_PEAK_TFLOPS,
            "note": "this config is FP64-compute-bound (SURVEY.md 8(d)): the HBM fraction above is capped at ~30-55 %",
        }
        roofline["binding"] = "fp64"
        roofline["fp64_frac"] = roofline["fp64"]["frac"]
        roofline["fp64_TFLOPs"] = roofline["fp64"]["achieved"][dom]
    if "single_stream" in extra:
        roofline["single_stream_ms_per_step"] = extra["single_stream"]["ms_per_step"]
    # `value` is the rotating-buffer ("cold") step whenever the workload's buffers would otherwise fit the cache
    roofline["cold_ms_per_step"], roofline["cold_value"] = (elapsed / steps * 1e3, units_per_step_all * steps / elapsed) \
        if (nsets > 1 or per_launch >= 768 * 2**20) else (None, None)
    if "hot" in extra:
        roofline["hot_ms_per_step"], roofline["hot_value"] = extra["hot"]["ms_per_step"], extra["hot"]["value"]
    elif nsets == 1 and per_launch < 768 * 2**20:
        roofline["hot_ms_per_step"], roofline["hot_value"] = elapsed / steps * 1e3, units_per_step_all * steps / elapsed

    if rank != 0:
        del chains
        torch.cuda.empty_cache()
        return None

    units_per_step = units_per_step_all
    out = {
        "metric": "QP+QCQP solves/sec (fwd+bwd)" if cfg != 2 else "QP forward solves/sec",
        "value": units_per_step * steps / elapsed, "unit": "solves/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": desc + "; eps=1e-7 max_iter=1000 mu_prox=1e-7",
            "baseline_config": cfg if cfg else "2'+3 (headline)",
            "B_total": units_per_step,
            "B_this_rank": [c.B for c in chains], "N": [c.N for c in chains],
            "buffers": ("step k works on input/output set k mod %d (%.0f MB per set): no step finds its data in the 256 MiB "
                        "Infinity Cache" % (nsets, per_set / 1e6)) if nsets > 1 else
                       ("one set of %.0f MB" % (per_set / 1e6) + (" (larger than the cache)" if per_launch >= 768 * 2**20 else
                                                                   ", the same buffers every step (cache-resident)")),
            "p_layout": ("dense (declared)" if chains[0].layout == 1 else
                         "auto + DQQ_F_EXPECT_DENSE given by the caller (verified in-kernel all the same)" if chains[0].layout == 0x200 else
                         "auto (off-diagonals verified in-kernel; non-diagonal tiles go to the general kernel)"),
            "launch": "eager, one C-ABI call per pass" + ((", the two families on two streams" + (
                ", swapping streams every step" if (alternate and nsets > 1) else "")) if side is not None else ""),
            "sharding": (("batch split over the ranks" if scaling == "strong" else "every rank its own batch (weak scaling)") +
                         ", no data-path collective; RCCL all-gather of x per step, issued after the forward and overlapped with the backward"
                         if (gather and use_dist) else ("every rank its own batch (weak scaling), no data-path collective; the optional "
                                                         "all-gather of x is timed separately: with_gather" if use_dist else "single GPU")),
            "rccl_world": dist.get_world_size() if use_dist else 1,
        },
        "repeats": {"R": len(times), "ms_per_step_median": elapsed / steps * 1e3, "ms_per_step_min": times[0] / steps * 1e3,
                    "ms_per_step_max": times[-1] / steps * 1e3,
                    "note": "R regions of exactly `steps` steps; value / ms_per_step are the median region; a region ends with "
                            "stream.query() spin + synchronize()"},
        "roofline": roofline,
        "kernels": kernels,
        "per_gpu_value": units_per_step * steps / elapsed / world,
    }
    out.update(extra)
    if not args.no_check:
        out["parity_max_abs_err_vs_oracle_sample"] = {c.names[0][:-4]: c.check(256 if c.N >= 32 else 2048) for c in chains}
    if world == 1 and not args.no_cpu_baseline and cfg not in (9, 10):   # (cfg 9 / 10: cfg 8's / cfg 4's family at another batch size)
        from oracle import oracle as O
        cores = O.max_threads()
        n = min(cpu_n, chains[0].B)
        if light:
            n = max(n // 4, 256)
        for c in chains:
            c.cpu_solves_per_s(min(n, 512), cores)   # spin up the OpenMP team
        passes = 2 if light else 5   # (the box's host cores are shared: single passes of ~25 ms vary several-fold)
        rates = [max(c.cpu_solves_per_s(n, cores) for _ in range(passes)) for c in chains]
        tot = sum(n for _ in chains) / sum(n / r for r in rates)
        n1 = max(n // 64, 64)
        one = sum(n1 for _ in chains) / sum(n1 / c.cpu_solves_per_s(n1, 1) for c in chains)
        out["cpu_baseline"] = {
            "value": tot, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": "oracle/diffqcqp_oracle.c, OpenMP over the batch, best of %d passes over the first %d problems of each family; "
                      "a dense C port of the reference algorithm" % (passes, n),
            "single_thread_value": one, "single_thread_sample": "same, 1 thread, first %d problems" % n1}
        if not light:
            out["cpu_baseline"].update(python_loop_baseline(chains))
        out["gpu_over_cpu_all_cores"] = out["value"] / tot
    del chains
    torch.cuda.empty_cache()
    return out


def python_loop_baseline(chains, n=2048):
    """The reference's EXECUTION MODEL: a Python `for i in range(B)` with one FFI call per problem --
    qcqp.py:29-31 (solveQP per item) and :45-47 (solveDerivativesQP per item) --, here over the oracle's single-problem
    functions (ctypes instead of pybind11), one thread, the first n problems of each family.  What the C port's
    single-thread figure leaves out is exactly this interpreter + marshalling overhead per problem."""
    from oracle import oracle as O
    t_tot, n_tot = 0.0, 0
    for c in chains:
        h = c.host_sample(min(n, c.B))
        m = h["q"].shape[0]
        ws = np.zeros(c.N)
        t0 = time.perf_counter()
        if c.kind == "qp":
            for i in range(m):
                x = O.solveQP(h["P"][i], h["q"][i], ws, EPS, MU_PROX, MAX_ITER)
                if c.backward:
                    O.solveDerivativesQP(h["P"][i], h["q"][i], x, h["g"][i])
        else:
            for i in range(m):
                x = O.solveQCQP(h["P"][i], h["q"][i], h["l_n"][i], h["mu"][i], ws, EPS, MU_PROX, MAX_ITER)
                if c.backward:
                    O.solveDerivativesQCQP(h["P"][i], h["q"][i], h["l_n"][i], h["mu"][i], x, h["g"][i])
        t_tot += time.perf_counter() - t0
        n_tot += m
    return {"python_loop_value": n_tot / t_tot,
            "python_loop_sample": "Python for-loop, one ctypes call per problem into the oracle's solveQP/solveQCQP + "
                                  "solveDerivatives* (reference qcqp.py:29-31, 45-47, 149-151, 167-172), 1 thread, first %d "
                                  "problems of each family" % min(n, chains[0].B),
            "reference_published": "qcqp_runtime.png (README): QCQP N=8, B=1, ~9e-5 s forward / ~2.7e-4 s backward per "
                                   "problem on the authors' CPU => ~1.1e4 forward/s, ~2.7e3 forward+backward solves/s per "
                                   "core through the reference's Python path (other hardware; context only)"}


def condensed(rec):
    """A per_config sub-record: what VERDICT r2 #4 / r3 #7 ask to be driver-measured for every BASELINE config."""
    rl = rec["roofline"]
    out = {"workload": rec["config"]["workload"][:90], "ms_per_step": rec["ms_per_step"], "value": rec["value"],
           "steps": rec["steps"], "regions": rec["repeats"]["R"],
           "ms_per_step_min_max": [rec["repeats"]["ms_per_step_min"], rec["repeats"]["ms_per_step_max"]],
           "dominant_kernel": rl["kernel"],
           "kernels_us": {k: round(v["mean_us"], 2) for k, v in rec["kernels"].items()},
           # (GB/s, not fractions of the HBM peak: a 10 us backward whose 55 MB sit in the 256 MiB Infinity Cache from the
           # forward before it reads above what HBM itself can deliver)
           "kernels_moved_GBps": {k: round(v["moved_GBps"], 1) for k, v in rec["kernels"].items()},
           "host_enqueue_us_per_step": rl["host_enqueue_us_per_step"],
           "roofline": {"bound": rl.get("binding", "hbm"), "frac": rl["frac"], "achieved_GBps": rl["achieved"],
                        "moved_frac": rl["moved_frac"], "moved_GBps": rl["moved_GBps"],
                        "step_moved_frac": rl["step_moved_frac"], "step_moved_GBps": rl["step_moved_GBps"],
                        "step_algorithmic_frac": rl["step_algorithmic_frac"],
                        "traffic": rl.get("traffic"), "traffic_source": rl.get("traffic_source")}}
    for k in ("fp64_frac", "fp64_TFLOPs", "valu_busy_frac", "valu_lane_utilisation", "traffic_over_algorithmic",
              "pmc_matches_build", "buffer_sets"):
        if k in rl:
            out["roofline"][k] = rl[k]
    for k in ("without_gather", "gather_after_backward", "parity_max_abs_err_vs_oracle_sample", "gpu_over_cpu_all_cores"):
        if k in rec:
            out[k] = rec[k]
    if "cpu_baseline" in rec:
        out["cpu_baseline"] = {k: rec["cpu_baseline"][k] for k in ("value", "cores", "kind", "single_thread_value")}
    return out


def flat_summary(prefix, rec):
    """Scalar entries for `config` / `roofline` (the driver's record keeps the first ~24 scalars of each): the step time
    and the STEP's fraction of the HBM peak on the bytes that move (cfg 4: 0.70 -- the algorithmic 0.99 counts an 8 KiB
    read of P per problem that the backward does not perform)."""
    rl = rec["roofline"]
    out = {prefix + "_ms_per_step": rec["ms_per_step"], prefix + "_moved_frac": rl["step_moved_frac"]}
    if "fp64_frac" in rl:
        out[prefix + "_fp64_frac"] = rl["fp64_frac"]
    return out


def flat_details(prefix, rec):
    """The rest of a sub-record's scalars (behind the kept ones)."""
    rl = rec["roofline"]
    out = {prefix + "_dominant_launch_frac": rl["frac"], prefix + "_dominant_launch_moved_frac": rl["moved_frac"],
           prefix + "_step_algorithmic_frac": rl["step_algorithmic_frac"]}
    for k, v in rec["kernels_us"].items():
        out[prefix + "_us_" + k] = v
    if "cpu_baseline" in rec:
        out[prefix + "_cpu_solves_per_s"] = rec["cpu_baseline"]["value"]
    return out


def dense_p_record(args, ctx):
    """Dense 8x8 P (what a real contact problem presents) at the bench's batch size, QCQP forward + backward through
    DQQ_P_AUTO (what QCQPFn2 passes) and DQQ_P_DENSE: full sub-records (per-launch durations, roofline, CPU baseline;
    VERDICT r3 #4), and their ratio (the cliff must not come back)."""
    rec = {"auto": condensed(measure(6, args, ctx, light=True)), "dense": condensed(measure(7, args, ctx, light=True))}
    rec["auto_ms_per_fwd_bwd"], rec["dense_ms_per_fwd_bwd"] = rec["auto"]["ms_per_step"], rec["dense"]["ms_per_step"]
    rec["auto_over_dense"] = rec["auto_ms_per_fwd_bwd"] / rec["dense_ms_per_fwd_bwd"]
    # "auto" above is the steady state of a caller that presents this kind of batch step after step: from the third step on
    # the caller-side hints (include/diffqcqp_hip.h: dqq_hint_flags) have moved the forward to one lane per
    # problem and the backward to one launch of the lane-per-problem kernel.  The same without hints (a first call, a
    # captured graph, DQQ_FEEDBACK=0):
    capi = ctx["capi"]
    if capi._hints_on:
        capi.enable_feedback(False)
        try:
            rec["auto_no_hint_ms_per_fwd_bwd"] = measure(6, args, ctx, light=True)["ms_per_step"]
            # ... and what the same caller gets, still without a report word (first call, captured graph), by SAYING what it
            # expects: DQQ_P_AUTO | DQQ_F_EXPECT_DENSE as an argument (diffqcqp_amd.qcqp.set_default_layout("auto_expect_dense"))
            rec["auto_explicit_flag_ms_per_fwd_bwd"] = measure(11, args, ctx, light=True)["ms_per_step"]
        finally:
            capi.enable_feedback(True)   # (the words as they were)
    return rec


def survey_extras_record(args, ctx):
    """The side lines SURVEY.md 8(d) asks for next to the headline: the iteration-count distribution of each headline
    family (a wave runs as long as its slowest problem), the long-tailed stress variant p ~ U(0,1) of config 2, and the
    compact diagonal layout (DQQ_P_DIAG, P as (B,N): 192 algorithmic bytes per QP forward at N=8) -- extension lines,
    never the headline."""
    from diffqcqp_amd import ops
    dev = ctx["dev"]
    B, N = 65536, 8
    g = torch.Generator(device=dev).manual_seed(1002)
    r = lambda *s: torch.rand(*s, generator=g, dtype=F64, device=dev)

    def stats(it):
        it = it.double()
        q = torch.quantile(it, torch.tensor([0.5, 0.99], dtype=F64, device=dev))
        tile = it.view(-1, 32).max(dim=1).values          # a wave tile of the bench shape holds 32 problems
        return {"mean": float(it.mean()), "p50": float(q[0]), "p99": float(q[1]), "max": float(it.max()),
                "mean_of_tile_max_32": float(tile.mean())}

    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / reps)
        return sorted(ts)[1]

    p, q = r(B, N) + 0.1, 2 * r(B, N, 1) - 1
    l_n, mu = r(B, N // 2, 1), r(B, N // 2, 1)
    P = torch.diag_embed(p).contiguous()
    rec = {"workload": "B=65536 N=8, eps=1e-7, max_iter=1000; one stream, median of 3 x 30 forward calls through the "
                       "Python ops layer into a preallocated x"}
    _, it_qp = ops.qp_forward(P, q, EPS, MAX_ITER, return_iters=True)
    _, it_qc = ops.qcqp_forward(P, q, l_n, mu, EPS, MAX_ITER, return_iters=True)
    rec["iterations"] = {"qp_p_u(0.1,1.1)": stats(it_qp), "qcqp_p_u(0.1,1.1)": stats(it_qc)}
    # stress variant of config 2: p ~ U(0,1): nearly singular coordinates, long-tailed iteration counts
    ps = r(B, N)
    Ps = torch.diag_embed(ps).contiguous()
    _, it_s = ops.qp_forward(Ps, q, EPS, MAX_ITER, return_iters=True)
    xb = torch.empty(B, N, 1, dtype=F64, device=dev)
    t = timed(lambda: ops.qp_forward(Ps, q, EPS, MAX_ITER, out=xb))
    rec["stress_p_u(0,1)_qp_fwd"] = {"ms_per_call": t * 1e3, "solves_per_s": B / t, "iterations": stats(it_s)}
    t0 = timed(lambda: ops.qp_forward(P, q, EPS, MAX_ITER, out=xb))
    rec["same_call_p_u(0.1,1.1)_qp_fwd"] = {"ms_per_call": t0 * 1e3, "solves_per_s": B / t0}
    # the workload of the reference's only published figure (test_script.py:91-123): P = diag(exp(U(-10,10))), q ~
    # U(-1,1), l_n, mu ~ U(0,1), eps 1e-10, max_iter 1e6 -- heavy-tailed iteration counts; here at B = 65536
    pf = torch.exp(r(B, N) * 20 - 10)
    Pf = torch.diag_embed(pf).contiguous()
    _, it_fq = ops.qp_forward(Pf, q, 1e-10, 1000000, return_iters=True)
    _, it_fc = ops.qcqp_forward(Pf, q, l_n, mu, 1e-10, 1000000, return_iters=True)
    tq = timed(lambda: ops.qp_forward(Pf, q, 1e-10, 1000000, out=xb), reps=5)
    tcq = timed(lambda: ops.qcqp_forward(Pf, q, l_n, mu, 1e-10, 1000000, out=xb), reps=5)
    # the same matrices at the bench's eps = 1e-7, max_iter = 1000 (what DESIGN.md / VERDICT r3 quote as 449 / 141 us)
    tq7 = timed(lambda: ops.qp_forward(Pf, q, EPS, MAX_ITER, out=xb), reps=10)
    tc7 = timed(lambda: ops.qcqp_forward(Pf, q, l_n, mu, EPS, MAX_ITER, out=xb), reps=10)
    rec["reference_figure_workload"] = {"qp_fwd_ms": tq * 1e3, "qcqp_fwd_ms": tcq * 1e3, "qp_iterations": stats(it_fq),
                                        "qcqp_iterations": stats(it_fc),
                                        "qp_fwd_ms_eps1e-7_maxiter1000": tq7 * 1e3, "qcqp_fwd_ms_eps1e-7_maxiter1000": tc7 * 1e3,
                                        "note": "P = diag(exp(U(-10,10))), eps 1e-10, max_iter 1e6 (reference test_script.py:91-123), B=65536"}
    del Pf
    # compact diagonal layout: P handed over as (B,N)
    tc = timed(lambda: ops.qp_forward(p, q, EPS, MAX_ITER, layout=2, out=xb))
    xc = ops.qp_forward(p, q, EPS, MAX_ITER, layout=2)
    xd = ops.qp_forward(P, q, EPS, MAX_ITER)
    rec["compact_diag_layout_qp_fwd"] = {"ms_per_call": tc * 1e3, "solves_per_s": B / tc,
                                         "algo_bytes_per_problem": 3 * N * 8,
                                         "algo_GBps": 3 * N * 8 * B / tc / 1e9,
                                         "bit_identical_to_dense_layout": bool(torch.equal(xc, xd))}
    return rec


# ---- the contract line -----------------------------------------------------------------------------------------------------
LINE_LIMIT = 6000      # bytes of the one stdout line (the driver's record of a longer line was unreadable: round 5)
CONFIG_KEYS = [        # `config`: workload + at most 20 scalars
    "baseline_config", "B_total", "rccl_world", "buffers",
    "cfg2_ms_per_step", "cfg2_moved_frac", "cfg3_ms_per_step", "cfg3_moved_frac", "cfg4_ms_per_step", "cfg4_moved_frac",
    "cfg5_ms_per_step", "cfg5_fp64_frac", "dense8_auto_ms_per_step", "dense8_auto_no_hint_ms_per_step",
    "dense8_auto_explicit_flag_ms_per_step", "dense8_dense_ms_per_step", "dense8_auto_no_hint_over_dense", "ref_figure_qp_fwd_ms",
    "ref_figure_qcqp_fwd_ms", "qcqp_grad_exit_flip_rate",
    # distributed runs (the sub-records above are measured by plain one-GPU runs only)
    "with_gather_ms_per_step", "with_gather_value", "strong_cfg4_ms_per_step", "strong_cfg4_value",
    "strong_cfg4_without_gather_ms_per_step", "strong_cfg4_gather_after_backward_ms_per_step"]
ROOFLINE_KEYS = [      # `roofline`: at most 24 scalars
    "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_over_algorithmic",
    "moved_frac", "valu_busy_frac", "cold_ms_per_step", "cold_value", "hot_ms_per_step", "hot_value",
    "qp_pair_ms_per_step", "qp_pair_solves_per_s", "qp_pair_moved_frac", "qp_pair_hot_ms_per_step",
    "qp_pair_large_ms_per_step", "qp_pair_large_solves_per_s", "qp_pair_large_moved_frac", "step_moved_frac", "kernel_us"]
CPU_KEYS = ["value", "unit", "cores", "kind", "sample", "single_thread_value", "python_loop_value"]
TOP_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data"]


def _clean(v, digits=6, maxlen=160):
    """A JSON-safe scalar: floats rounded to `digits` significant digits (NaN / inf -> None), strings cut to maxlen."""
    if isinstance(v, bool) or v is None or isinstance(v, int):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, v))
    if isinstance(v, str):
        return v if len(v) <= maxlen else v[:maxlen - 3] + "..."
    return _clean(float(v), digits, maxlen) if hasattr(v, "__float__") else str(v)[:maxlen]


def contract_line(full, details_path="bench_details.json"):
    """The ONE line bench.py prints: the contract keys, `config` (workload + <= 20 scalars), `roofline` (<= 24 scalars),
    `cpu_baseline` (<= 7), nothing nested, at most LINE_LIMIT bytes, strict JSON.  Everything else of the full record --
    per-kernel brackets, the sub-records of every BASELINE config, the environment -- is in bench_details.json."""
    cf, rl = full.get("config", {}), dict(full.get("roofline", {}))
    dom = rl.get("kernel")
    if dom is not None and "kernel_us_" + str(dom) in rl:
        rl["kernel_us"] = rl["kernel_us_" + dom]
    line = {k: (full.get(k) if k in ("value", "ms_per_step") else _clean(full.get(k))) for k in TOP_KEYS}
    for k in ("value", "ms_per_step"):    # (unrounded: value x ms_per_step = the units of a step, to the last digit)
        if isinstance(line[k], float) and line[k] != line[k]:
            line[k] = None
    conf = {"workload": _clean(cf.get("workload", ""), maxlen=330)}
    for k in CONFIG_KEYS:
        if k in cf and not isinstance(cf[k], (dict, list)) and len(conf) < 21:
            conf[k] = _clean(cf[k], maxlen=120)
    line["config"] = conf
    roof = {}
    for k in ROOFLINE_KEYS:
        if k in rl and not isinstance(rl[k], (dict, list)) and len(roof) < 24:
            roof[k] = _clean(rl[k], maxlen=80)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):   # the contract's keys are always there
        roof.setdefault(k, None)
    line["roofline"] = roof
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = {k: _clean(cb[k], maxlen=120) for k in CPU_KEYS if k in cb}
    line["details"] = details_path
    text = json.dumps(line, allow_nan=False)
    while len(text) > LINE_LIMIT and len(conf) > 1:      # (cannot happen with the limits above; never print an oversized line)
        conf.popitem()
        text = json.dumps(line, allow_nan=False)
    return text


def emit(full, real_stdout, details_path):
    """bench_details.json (the full record), a copy of it on stderr (prefixed, so that nothing on stderr looks like the
    contract line), and the contract line on the real stdout."""
    def safe(o):
        if isinstance(o, dict):
            return {str(k): safe(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [safe(v) for v in o]
        if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
            return None
        return o
    blob = json.dumps(safe(full), allow_nan=False)
    shown = details_path
    try:
        with open(details_path, "w") as f:
            f.write(blob + "\n")
        if os.path.dirname(os.path.abspath(details_path)) == ROOT:
            shown = os.path.basename(details_path)
    except OSError as e:
        shown = "stderr only (%s)" % type(e).__name__
    sys.stderr.write("[bench details] " + blob + "\n")
    sys.stderr.flush()
    os.write(real_stdout, (contract_line(full, shown) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=None, choices=(0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11),
                    help="BASELINE.json configs entry (1-based); 0 = the headline step (configs 2'+3).  Default: the "
                         "headline (distributed: on every rank, weak scaling; with_gather = with the RCCL all-gather of x) and "
                         "configs[3] (the batch split over the ranks) as the sub-record strong_config4")
    ap.add_argument("--repeats", type=int, default=10, help="timed regions of exactly --steps steps; the median is reported")
    ap.add_argument("--streams", type=int, default=2, choices=(1, 2),
                    help="headline only: 2 = the QP chain and the QCQP chain on two HIP streams")
    ap.add_argument("--alternate-streams", type=int, default=1, choices=(0, 1),
                    help="headline on two streams with rotating buffers: the two families swap streams every step, so that both "
                         "streams carry the same work (0: a fixed family -> stream assignment)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--hot-only", "--no-cold", dest="hot_only", action="store_true",
                    help="one input/output set, the same buffers every step (default: steps rotate over > 768 MiB of sets)")
    ap.add_argument("--no-hot", action="store_true", help="skip the hot / single-stream context regions")
    ap.add_argument("--no-per-config", action="store_true", help="default run only: skip the sub-records of bench_details.json")
    ap.add_argument("--live-pmc", action="store_true",
                    help="spawn two rocprofv3 counter passes that measure roofline.traffic now (default: quoted from profiles/)")
    ap.add_argument("--details", default=os.path.join(ROOT, "bench_details.json"), help="where the full record goes")
    ap.add_argument("--pmc-child", type=int, default=None, help=argparse.SUPPRESS)   # the process live_pmc() profiles
    args = ap.parse_args()
    if args.pmc_child is not None:
        args.hot_only = False
        return pmc_child(args.pmc_child)

    # Everything the native libraries print on stdout (RCCL prints its version banner there) goes to
    # stderr; the one JSON line is written to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs the GPU (the hot path is a HIP kernel; there is no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    # DQQ_BENCH_FORCE_DIST=1 exercises the RCCL path (init, barrier, all-gather) with a single rank
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # under torch.distributed.run (any world size)
    use_dist = world > 1 or os.environ.get("DQQ_BENCH_FORCE_DIST") == "1" or launched
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL

    from diffqcqp_amd import build, _capi, parallel
    if rank == 0:
        build.build()
    if use_dist:
        dist.barrier()
    _capi.lib()
    ctx = {"rank": rank, "world": world, "dev": dev, "use_dist": use_dist, "dist": dist, "parallel": parallel,
           "capi": _capi, "side": torch.cuda.Stream(),
           # the counter passes run only when asked for, for a plain one-GPU run of a whole workload
           "live_pmc": world == 1 and not use_dist and args.live_pmc}

    default_run = args.config is None
    # No --config: the headline step, on one GPU and on N (ONE workload across the driver's N = 1, 2, 4, 8 lines, so that
    # value(N) / (N value(1)) is a scaling efficiency; the solve needs no collective).  Distributed, the same step is also
    # timed with the path's optional exchange step (`with_gather`), and configs[3], the batch SPLIT over the ranks with the
    # all-gather of x (strong scaling), is the sub-record `strong_config4`.
    primary = args.config if args.config is not None else 0
    out = measure(primary, args, ctx)
    if rank == 0:
        for k in ("with_gather", "without_gather", "gather_after_backward"):
            if k in out:
                out["config"][k + "_ms_per_step"], out["config"][k + "_value"] = out[k]["ms_per_step"], out[k]["value"]
        par = out.get("parity_max_abs_err_vs_oracle_sample", {}).get("qcqp", {})
        if "refinement_exit_flip_rate" in par:
            out["config"]["qcqp_grad_exit_flip_rate"] = par["refinement_exit_flip_rate"]
    if default_run:
        if use_dist:
            sub = measure(4, args, ctx, light=True)
            if rank == 0:
                s4 = out["strong_config4"] = condensed(sub)
                s4["note"] = "BASELINE configs[3]: B=262144 N=32 split over the ranks, all-gather of x behind the forward " \
                             "(strong scaling); N=1 reference: per_config.config_4 of a plain run"
                out["config"].update(flat_summary("strong_cfg4", s4))
                out["config"]["strong_cfg4_value"] = s4["value"]
                for k in ("without_gather", "gather_after_backward"):
                    if k in s4:
                        out["config"]["strong_cfg4_%s_ms_per_step" % k] = s4[k]["ms_per_step"]
                out["config"].update(flat_details("strong_cfg4", s4))
        elif not args.no_per_config:
            cfgk, rlk = {}, {}     # scalars for `config` / `roofline` of the contract line
            # north_star's target sentence, measured: N = 8 QP forward+backward as ONE step on one stream, at the bench
            # batch and at the batch that fills the chip; fractions on the bytes that move (1538 B per pair)
            pairs = {}
            for name, cfg in (("qp_pair", 8), ("qp_pair_large", 9)):
                rec = measure(cfg, args, ctx, light=True)
                c = pairs[name] = condensed(rec)
                rlk[name + "_ms_per_step"] = c["ms_per_step"]
                rlk[name + "_solves_per_s"] = c["value"]
                rlk[name + "_moved_frac"] = c["roofline"]["step_moved_frac"]
                if name == "qp_pair":
                    rlk[name + "_algorithmic_frac"] = c["roofline"]["step_algorithmic_frac"]
                    if "hot" in rec:
                        rlk[name + "_hot_ms_per_step"] = rec["hot"]["ms_per_step"]
                        c["hot"] = rec["hot"]
                c["bytes_per_pair"] = {"moved": moved_bytes("qp", 8, "fwd", True) + moved_bytes("qp", 8, "bwd", True),
                                       "algorithmic": algo_bytes("qp", 8, "fwd") + algo_bytes("qp", 8, "bwd")}
            out["qp_pair"], out["qp_pair_large"] = pairs["qp_pair"], pairs["qp_pair_large"]
            per, details = {}, {}
            for cfg in (2, 3, 4, 5):
                per["config_%d" % cfg] = condensed(measure(cfg, args, ctx, light=True))
                cfgk.update(flat_summary("cfg%d" % cfg, per["config_%d" % cfg]))
                details.update(flat_details("cfg%d" % cfg, per["config_%d" % cfg]))
            out["per_config"] = per
            # one rank's share of configs[3] on an 8-GPU node, on this GPU: how much of linear strong scaling the kernels keep
            # at the smaller batch, before the all-gather of x (8.4 MB per rank; SURVEY 8(e): ~55 us over 7 xGMI links)
            sh = condensed(measure(10, args, ctx, light=True))
            per["config_4_shard_1_of_8"] = sh
            details["cfg4_shard8_ms_per_step"] = sh["ms_per_step"]
            details["cfg4_shard8_kernel_scaling_efficiency"] = per["config_4"]["ms_per_step"] / (8.0 * sh["ms_per_step"])
            out["per_config_note"] = "BASELINE.json configs 2-5 (1-based) measured by THIS run, 3 regions each; config_4 " \
                                     "is the whole B=262144 batch on this one GPU; cfgK_moved_frac = the STEP's bytes that " \
                                     "move / step time / 8 TB/s"
            out["dense_p_n8"] = dense_p_record(args, ctx)
            d8 = out["dense_p_n8"]
            cfgk["dense8_auto_ms_per_step"] = d8["auto"]["ms_per_step"]
            if "auto_no_hint_ms_per_fwd_bwd" in d8:
                cfgk["dense8_auto_no_hint_ms_per_step"] = d8["auto_no_hint_ms_per_fwd_bwd"]
                cfgk["dense8_auto_no_hint_over_dense"] = d8["auto_no_hint_ms_per_fwd_bwd"] / d8["dense"]["ms_per_step"]
            if "auto_explicit_flag_ms_per_fwd_bwd" in d8:
                cfgk["dense8_auto_explicit_flag_ms_per_step"] = d8["auto_explicit_flag_ms_per_fwd_bwd"]
            cfgk["dense8_dense_ms_per_step"] = d8["dense"]["ms_per_step"]
            cfgk["dense8_auto_over_dense"] = d8["auto_over_dense"]
            details.update(flat_details("dense8_auto", d8["auto"]))
            details.update(flat_details("dense8_dense", d8["dense"]))
            out["survey_8d_extras"] = survey_extras_record(args, ctx)
            ex = out["survey_8d_extras"]
            cfgk.update({"stress_p_u01_qp_fwd_ms": ex["stress_p_u(0,1)_qp_fwd"]["ms_per_call"],
                         "ref_figure_qp_fwd_ms": ex["reference_figure_workload"]["qp_fwd_ms"],
                         "ref_figure_qcqp_fwd_ms": ex["reference_figure_workload"]["qcqp_fwd_ms"]})
            details.update({"ref_figure_qp_fwd_ms_bench_eps": ex["reference_figure_workload"]["qp_fwd_ms_eps1e-7_maxiter1000"],
                            "ref_figure_qcqp_fwd_ms_bench_eps": ex["reference_figure_workload"]["qcqp_fwd_ms_eps1e-7_maxiter1000"]})
            out["config"].update(cfgk)
            out["config"].update(details)
            out["roofline"].update(rlk)
    if rank == 0:
        out["scaling_note"] = ("`value` = the headline step on every rank (weak scaling, no collective; with_gather = with the "
                               "all-gather of x); configs[3] split over the ranks (strong scaling) = strong_config4 (N>1) / "
                               "per_config.config_4 (plain N=1 run)")
        out["environment"] = gpu_environment()
        out["csrc_sha16"] = csrc_sha16()
        emit(out, real_stdout, args.details)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
