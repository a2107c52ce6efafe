"""The whole surface of the reference's ``embedding_help_functions`` in one module:

    import tmgcn_amd.ehf as ehf          # instead of: import embedding_help_functions as ehf

is the only edit a reference experiment script needs (experiment_*_our*.py:17).  Everything a
script reaches through ``ehf.`` is here under the same name with the same arguments and return
values — the four model classes (layers.py; the layer runs in the HIP kernels), the metrics
(metrics.py), and the data functions (data.py):

    EmbeddingGCN  EmbeddingGCN2  EmbeddingKWGCN  EmbeddingGCN_reg
    load_data  create_node_features  augment_edges  split_data  compute_At
    compute_f1  compute_MAP_MRR  get_MAP  get_MRR  get_row_MRR  print_f1

The scripts keep targets, class weights and the criterion on the host (``criterion(gcn(),
target_train)``, …_link_prediction.py:69,79), so the classes exported here deliver their logits
to the host: ``output_device = "cpu"``, an autograd-aware copy of the [E, C] result; the backward
copy of its gradient is the only other host↔device traffic of an epoch.  A script that moves
its targets and criterion to the device should use ``tmgcn_amd.layers`` (same classes, logits stay
on the device) or set ``ehf.EmbeddingGCN.output_device = None``.
"""
from . import layers as _layers
from .data import augment_edges, compute_At, create_node_features, load_data, print_f1, split_data  # noqa: F401
from .metrics import compute_f1, compute_MAP_MRR, get_MAP, get_MRR, get_row_MRR  # noqa: F401


class EmbeddingGCN(_layers.EmbeddingGCN):
    output_device = "cpu"


class EmbeddingGCN2(_layers.EmbeddingGCN2):
    output_device = "cpu"


class EmbeddingKWGCN(_layers.EmbeddingKWGCN):
    output_device = "cpu"


class EmbeddingGCN_reg(_layers.EmbeddingGCN_reg):
    output_device = "cpu"


for _c in (EmbeddingGCN, EmbeddingGCN2, EmbeddingKWGCN, EmbeddingGCN_reg):
    _c.__doc__ = getattr(_layers, _c.__name__).__doc__
del _c
