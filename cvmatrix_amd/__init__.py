"""cvmatrix_amd -- MI355X-native implementation of the cvmatrix per-fold training-matrix
hot path (CVMatrix.fit + training_XTX / training_XTY / training_XTX_XTY), behind the
reference's own Python API (sm00thix/cvmatrix, cvmatrix/__init__.py:1-4).

The arithmetic runs in hand-written HIP kernels (cvmatrix_amd/csrc/) reached
through the C ABI of include/cvmhip.h.  There is no CPU fallback: without a GPU and the
built libcvmhip.so, ``CVMatrix.fit`` raises."""

__version__ = "0.3.0"
__all__ = ["CVMatrix", "Partitioner", "FoldBatch"]

from .partitioner import Partitioner
from .cvmatrix import CVMatrix, FoldBatch
