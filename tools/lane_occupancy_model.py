"""Lane occupancy of the N = 8 forward's main loop from the iteration counts alone (what share of the issued
VALU work runs with few lanes enabled?).  Iteration counts come from the oracle (identical to the kernels': every GPU test
asserts it); a wave tile = 32 consecutive problems on two lanes each; re-spread to four lanes per problem at <= 16 survivors and to
eight at <= 8 (admm_core.h).  Weights: instructions per trip of the E = 4 / 2 / 1 bodies (99 + rho block ... from the ISA: 141 / 99 / 60).
    python tools/lane_occupancy_model.py [B]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
O.build()
B, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 65536), 8
g = torch.Generator().manual_seed(1031)
r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
P = torch.diag_embed(r(B, N) + 0.1).numpy(); q = (2 * r(B, N, 1) - 1).numpy(); l_n = r(B, N // 2, 1).numpy(); mu = r(B, N // 2, 1).numpy()
for kind in ("qcqp", "qp"):
    it = (O.qcqp_fwd_batch(P, q, l_n, mu, 1e-7, 1000, nthreads=8)[1] if kind == "qcqp" else O.qp_fwd_batch(P, q, 1e-7, 1000, nthreads=8)[1])
    tiles = it.reshape(-1, 32)
    W = {4: 141.0, 2: 99.0, 1: 60.0}
    hist = np.zeros(5)           # weighted instructions by enabled-lane share: (0,.25], (.25,.5], (.5,.75], (.75,1)], total
    lane_cycles = total = 0.0
    for t in tiles:
        tmax = t.max()
        stage = 4
        for trip in range(tmax):
            live = int((t > trip).sum())
            lanes = live * (2 if stage == 4 else 4 if stage == 2 else 8)
            w = W[stage]
            share = lanes / 64.0
            hist[min(int(np.ceil(share * 4)) - 1, 3)] += w
            lane_cycles += w * share; total += w
            nxt = int((t > trip + 1).sum())
            if stage == 4 and nxt <= 16: stage = 2
            if stage == 2 and nxt <= 8: stage = 1
    print("%-5s mean %.1f tile max %.1f | modelled lane utilisation of the loop %.3f | share of loop instructions issued with <=25%% / 25-50%% / 50-75%% / >75%% of lanes: %s"
          % (kind, it.mean(), tiles.max(1).mean(), lane_cycles / total, " / ".join("%.3f" % (h / total) for h in hist[:4])))
