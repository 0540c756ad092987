"""The numeric inputs the reference hard-codes in its own smoke driver, as DATA (inputs only).

`Solver::test()` (`/root/reference/qcqplib/Solver.cpp:693-930`) prints results and holds no expected values, but
its matrices are the only problem data the reference itself ships: a singular 4x4 `P` with a 1e7-scale solution, a
12x12 Delassus-like product with entries from 1e-24 to 5e3, a block-diagonal 4x4 with a zero-radius contact, and a
rank-deficient 8x8 Delassus matrix of a real 4-contact problem (rows 0 = 4, 1 = 3, 2 = 6, 5 = 7).  SURVEY.md 8(c)(v)
lists them as pins for the inputs.  The values below are those literals; the expected outputs in `ref_*.npz` are the
ORACLE's (parity unpinned, see oracle/README.md), produced by `make_golden.py`.

Where the reference draws a value at random (`VectorXd::Random`, e.g. `l_ng`, `grad_l`), a seeded draw from the same
range stands in and the docstring of the case says so.
"""
import numpy as np


def m2_singular():
    """Solver.cpp:708-712: `m2.setZero(); m2(0,0) = .0005; m2(1,1) = 3.; q2 = (-8000, 0, 0, 0)`; solved as a QP with
    eps 1e-10, mu_prox 1e-7, max_iter 1000 (:712) and max_iter 1 (:729); `l_n = (1,1)*10000` (:702-704) is the
    radius handed to dualFromPrimalQCQP / solveDerivativesQCQP (:715, :726)."""
    P = np.zeros((4, 4))
    P[0, 0] = 0.0005
    P[1, 1] = 3.0
    q = np.array([-8000.0, 0.0, 0.0, 0.0])
    l_n = np.array([1.0, 1.0]) * 10000
    return P, q, l_n


def m2_first():
    """Solver.cpp:697-700, :705, :707: the values `m2`, `q2`, `warm_start2` hold before `:708` overwrites them (the
    commented-out QCQP call of :711 ran on them)."""
    P = np.array([[4.45434, 1.11359, -2.22717, 1.11359],
                  [1.11359, 4.45434, 1.11359, -2.22717],
                  [-2.22717, 1.11359, 4.45434, 1.11359],
                  [1.11359, -2.22717, 1.11359, 4.45434]])
    q = np.array([-0.0112815, -0.0083385, -0.0083385, -0.0112815])
    l_n = np.array([1.0, 1.0]) * 10000
    return P, q, l_n


def g2_product():
    """Solver.cpp:741-757: the 12x12 `G2` (entries 1e-24 ... 5e3), `G2(0,0) = 0; G2(1,1) = 4e1; G2 = G2*G2^T`, `g2`."""
    G2 = np.array([
        [6.6174e-24, 0, 0, 0, -4.8452e-04, 0, 0, 0, 0, 0, 0, 0],
        [0, -6.6174e-24, 0, 0, 0, 0, -3.9642e-04, 0, 0, 0, 0, 0],
        [0, 0, -6.6174e-24, 0, 0, 0, 0, 0, -3.9642e-04, -7.1925e-20, 0, 0],
        [0, 0, 0, 6.6174e-24, 0, 0, 0, 0, 0, 0, -4.8452e-04, 4.4048e-20],
        [-1.0544e+00, 0, 0, 0, 4.3570e+03, -1.6704e+00, 4.4543e+00, 1.6704e+00, 1.1136e+00, 1.6704e+00, 1.1136e+00,
         -1.6704e+00],
        [0, 0, 0, 0, -1.6704e+00, 4.3570e+03, -1.6704e+00, 1.1136e+00, 1.6704e+00, 1.1136e+00, 1.6704e+00,
         4.4543e+00],
        [0, -1.0544e+00, 0, 0, 4.4543e+00, -1.6704e+00, 5.3243e+03, 1.6704e+00, 1.1136e+00, 1.6704e+00, 1.1136e+00,
         -1.6704e+00],
        [0, 0, 0, 0, 1.6704e+00, 1.1136e+00, 1.6704e+00, 5.3243e+03, -1.6704e+00, 4.4543e+00, -1.6704e+00,
         1.1136e+00],
        [0, 0, -1.0544e+00, 0, 1.1136e+00, 1.6704e+00, 1.1136e+00, -1.6704e+00, 5.3243e+03, -1.6704e+00, 4.4543e+00,
         1.6704e+00],
        [0, 0, -1.9131e-16, 0, 1.6704e+00, 1.1136e+00, 1.6704e+00, 4.4543e+00, -1.6704e+00, 5.3243e+03, -1.6704e+00,
         1.1136e+00],
        [0, 0, 0, -1.0544e+00, 1.1136e+00, 1.6704e+00, 1.1136e+00, -1.6704e+00, 4.4543e+00, -1.6704e+00, 4.3570e+03,
         1.6704e+00],
        [0, 0, 0, 9.5861e-17, -1.6704e+00, 4.4543e+00, -1.6704e+00, 1.1136e+00, 1.6704e+00, 1.1136e+00, 1.6704e+00,
         4.3570e+03]])
    G2[0, 0] = 0.0
    G2[1, 1] = 4e1
    P = G2 @ G2.T
    q = np.array([0, 0, 0, 0, 7.2829e-04, 2.2609e-14, 7.2829e-04, 2.2609e-14, 7.2829e-04, 2.2609e-14, 7.2829e-04,
                  2.2609e-14])
    return P, q


def g_blockdiag():
    """Solver.cpp:784-791: the 4x4 block-diagonal `G` (with its -1.1102e-16 off-diagonal pair) and `g`; radii
    `l_ng2 = (0.00966, 0.)` (:870: one ZERO-radius contact) and `l_ng2[0] = 0.5893*0.7300` (:869)."""
    P = np.array([[1.1648e+00, -1.1102e-16, 0, 0],
                  [-1.1102e-16, 1.1648e+00, 0, 0],
                  [0, 0, 3.4989e+00, 0],
                  [0, 0, 0, 3.4989e+00]])
    q = np.array([0.5499, 0.5499, 0.0, 0.0])
    radii = [np.array([0.00966, 0.0]), np.array([0.5893 * 0.7300, 0.0])]
    return P, q, radii


def g4_delassus():
    """Solver.cpp:899-923: `G4` (8x8, rank 4: rows 0 = 4, 1 = 3, 2 = 6, 5 = 7), `g4`, `l_ng4 *= .15`."""
    P = np.array([[2.8750, -0.3750, 2.1250, -0.3750, 2.8750, 0.3750, 2.1250, 0.3750],
                  [-0.3750, 2.8750, 0.3750, 2.8750, -0.3750, 2.1250, 0.3750, 2.1250],
                  [2.1250, 0.3750, 2.8750, 0.3750, 2.1250, -0.3750, 2.8750, -0.3750],
                  [-0.3750, 2.8750, 0.3750, 2.8750, -0.3750, 2.1250, 0.3750, 2.1250],
                  [2.8750, -0.3750, 2.1250, -0.3750, 2.8750, 0.3750, 2.1250, 0.3750],
                  [0.3750, 2.1250, -0.3750, 2.1250, 0.3750, 2.8750, -0.3750, 2.8750],
                  [2.1250, 0.3750, 2.8750, 0.3750, 2.1250, -0.3750, 2.8750, -0.3750],
                  [0.3750, 2.1250, -0.3750, 2.1250, 0.3750, 2.8750, -0.3750, 2.8750]])
    q = np.array([3.9650e-01, 1.3222e-16, 3.9650e-01, 1.3222e-16, 3.9650e-01, 2.9742e-16, 3.9650e-01, 2.9742e-16])
    l_n = np.array([0.0159, 0.0159, 0.0086, 0.0086]) * 0.15
    return P, q, l_n


def rank_deficient(kind, B, N, seed, family="lowrank", q_in_range=True):
    """Seeded rank-deficient dense batches (not reference data; the regime `g4_delassus` stands for).
    family 'lowrank': P = S S^T / N with S of shape N x N/2 (rank N/2, PSD, singular);
           'duprows': a PSD `J M J^T` whose Jacobian J repeats rows, like G4 (row i = row i + N/2 for even i ...);
           'psd_eps': 'lowrank' + 1e-9 I (numerically singular, strictly PD).
    q_in_range: q = S w (as a contact problem's q = J v is), so the QP is bounded below; False: q ~ U(-1,1), for
    which the QP over x >= 0 is typically UNBOUNDED and the reference's loop runs into max_iter."""
    import torch
    g = torch.Generator().manual_seed(seed)
    r = N // 2
    S = torch.rand(B, N, r, generator=g, dtype=torch.float64) * 2 - 1
    if family == "duprows":
        idx = torch.arange(N) % r          # row k and row k + N/2 of the Jacobian coincide
        S = S[:, idx, :]
    P = torch.bmm(S, S.transpose(1, 2)) / N
    if family == "psd_eps":
        P = P + 1e-9 * torch.eye(N, dtype=torch.float64)
    q = 2 * torch.rand(B, N, 1, generator=g, dtype=torch.float64) - 1
    if q_in_range:
        w = 2 * torch.rand(B, r, 1, generator=g, dtype=torch.float64) - 1
        q = torch.bmm(S, w) / (r ** 0.5)
    out = {"P": P.contiguous(), "q": q, "grad_x": torch.randn(B, N, 1, generator=g, dtype=torch.float64)}
    if kind == "qcqp":
        out["l_n"] = torch.rand(B, N // 2, 1, generator=g, dtype=torch.float64)
        out["mu"] = torch.rand(B, N // 2, 1, generator=g, dtype=torch.float64)
    return out
