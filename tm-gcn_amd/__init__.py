"""tmgcn_amd — MI355X-native TM-GCN propagation layer (tensor M-product GCN, fwd + bwd).

The hot path of IBM/TM-GCN (embedding_help_functions.EmbeddingGCN / EmbeddingGCN2 /
EmbeddingKWGCN) as hand-written gfx950 HIP kernels behind a C-ABI (include/tmgcn.h), with a
Python host side that keeps the reference's class contract:

    import tmgcn_amd.layers as ehf
    gcn = ehf.EmbeddingGCN2(Ct_train_2, X_train, edges_train, M, hidden_feat=[6,6,2],
                            condensed_W=True, use_Minv=False, nonlin2="selu")
"""
from .csr import BatchedCSR  # noqa: F401
from . import _lib, ops, layers  # noqa: F401
from .layers import EmbeddingGCN, EmbeddingGCN2, EmbeddingKWGCN  # noqa: F401
from .losses import WeightedCrossEntropy  # noqa: F401

# The hot path is the three classes above.  Two conveniences from outside that scope exist because a
# reference script can reach them through `ehf.` — layers.EmbeddingGCN_reg (ehf:359-423, the SEIR regression head) and
# data.compute_At (ehf:27 calls it unused) — and are exported by `tmgcn_amd.ehf` only: beyond §8, no further work.
__all__ = ["BatchedCSR", "EmbeddingGCN", "EmbeddingGCN2", "EmbeddingKWGCN", "WeightedCrossEntropy", "ops", "layers"]
