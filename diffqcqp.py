"""Top-level `diffqcqp` module: the importable name of the reference's pybind11 extension
(reference pybindings.cpp:74-83, CMakeLists.txt:26), so that its own import line

    from diffqcqp import solveQP, solveBoxQP, solveQCQP, solveDerivativesQP, solveDerivativesBoxQP, \\
        solveDerivativesQCQP, solveSignedBoxQP                                     # reference qcqp.py:17

resolves unchanged against this build.  Same seven functions, same argument order, keyword defaults and return
shapes; each call is one B = 1 launch of the HIP kernels through the C ABI (diffqcqp_amd/diffqcqp.py)."""
from diffqcqp_amd.diffqcqp import (solveBoxQP, solveDerivativesBoxQP, solveDerivativesQCQP, solveDerivativesQP,  # noqa: F401
                                   solveQCQP, solveQP, solveSignedBoxQP)

__all__ = ["solveQP", "solveBoxQP", "solveSignedBoxQP", "solveQCQP", "solveDerivativesQP", "solveDerivativesBoxQP",
           "solveDerivativesQCQP"]
