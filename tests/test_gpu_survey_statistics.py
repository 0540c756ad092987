"""The HIP forward at BASELINE's full batch size against the iteration statistics SURVEY.md Appendix C quotes from the survey
session's independent numpy restatement of the reference (-m gpu).  A size-independent property check in the sense of the
tier's rules: 65536 problems are far beyond what the oracle is asked to solve in a test, but their iteration-count
distribution must be the one an independent reading of Solver.cpp produced on 400."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_iteration_count_distributions_at_full_batch_size():
    from diffqcqp_amd import ops
    B = 65536
    g = torch.Generator(device="cuda").manual_seed(424242)
    U = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64, device="cuda")
    q = 2 * U(B, 8, 1) - 1
    p = U(B, 8) + 0.1
    # QP diag N = 8, p ~ U(.1, 1.1): survey mean 18.3 / p99 25 / max 31
    x, it = ops.qp_forward(torch.diag_embed(p).contiguous(), q, 1e-7, 1000, return_iters=True)
    it = it.double()
    assert abs(it.mean().item() - 18.3) < 0.4 and 24 <= torch.quantile(it, 0.99).item() <= 27 and it.max().item() <= 45
    err = (x[:, :, 0] - torch.clamp(-q[:, :, 0] / p, min=0)).abs().amax(dim=1)
    assert 0.8e-7 < err.median().item() < 4e-7 and err.max().item() < 1e-3          # survey: median 1.7e-7, max 8.6e-5
    # p ~ U(0, 1): survey mean 23.6 / median 21 / p99 64, heavy tail
    _, it = ops.qp_forward(torch.diag_embed(U(B, 8)).contiguous(), q, 1e-7, 1000, return_iters=True)
    it = it.double()
    assert abs(it.mean().item() - 23.6) < 0.8 and 19 <= it.median().item() <= 21 and 58 <= torch.quantile(it, 0.99).item() <= 70
    assert it.max().item() < 1000
    # QCQP diag N = 8: survey mean 17.6 / p99 27 / max 33
    _, it = ops.qcqp_forward(torch.diag_embed(p).contiguous(), q, U(B, 4, 1), U(B, 4, 1), 1e-7, 1000, return_iters=True)
    it = it.double()
    assert abs(it.mean().item() - 17.6) < 0.4 and 26 <= torch.quantile(it, 0.99).item() <= 29 and it.max().item() <= 50
    # QP diag N = 32, B = 32768 (a configs[3] shard): survey mean 21.6, max 25
    _, it = ops.qp_forward(torch.diag_embed(U(32768, 32) + 0.1).contiguous(), 2 * U(32768, 32, 1) - 1, 1e-7, 1000, return_iters=True)
    assert abs(it.double().mean().item() - 21.6) < 0.5 and it.max().item() <= 32
    # dense N = 64 (configs[4] family), 4096 problems: survey mean 91 (p99 131, max 136)
    S = U(4096, 64, 64)
    P = torch.bmm(S, S.transpose(1, 2)) / 64 + 0.1 * torch.eye(64, dtype=torch.float64, device="cuda")
    _, it = ops.qp_forward(P, 2 * U(4096, 64, 1) - 1, 1e-7, 1000, return_iters=True)
    it = it.double()
    assert abs(it.mean().item() - 91) < 3.0 and it.max().item() <= 200
