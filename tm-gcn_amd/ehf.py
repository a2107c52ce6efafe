"""The whole surface of the reference's ``embedding_help_functions`` in one module:

    import tmgcn_amd.ehf as ehf          # instead of: import embedding_help_functions as ehf

is the only edit a reference experiment script needs (experiment_*_our*.py:17).  Everything a
script reaches through ``ehf.`` is here under the same name with the same arguments and return
values — the four model classes (layers.py; the layer runs in the HIP kernels), the metrics
(metrics.py), and the data functions (data.py):

    EmbeddingGCN  EmbeddingGCN2  EmbeddingKWGCN                       (the hot path: SURVEY §8 a1-a8)
    load_data  create_node_features  augment_edges  split_data
    compute_f1  compute_MAP_MRR  get_MAP  get_MRR  get_row_MRR  print_f1
    EmbeddingGCN_reg  compute_At      (beyond §8 — the SEIR regression model and a function the reference itself calls
                                       unused, ehf:27 — kept so that every `ehf.` name resolves; no kernel work went into them)

The scripts keep targets, class weights and the criterion on the host (``criterion(gcn(),
target_train)``, …_link_prediction.py:69,79).  The classes exported here therefore return their
logits as ``hosted.DeviceResult``: a tensor that stays on the MI355X and pulls the host tensors it
is combined with over to the device, so the loss, its backward and the metrics run there too and
the script does not change.  Two alternatives, per class or per instance:
``output_device = "cpu"`` (plain host logits through an autograd-aware copy; the criterion then
runs on the CPU) and ``host_operands = False`` (plain device logits — ``tmgcn_amd.layers``'
behaviour, for scripts that move their targets themselves).
"""
from . import layers as _layers
from .data import augment_edges, compute_At, create_node_features, load_data, print_f1, split_data  # noqa: F401
from .metrics import compute_f1, compute_MAP_MRR, get_MAP, get_MRR, get_row_MRR  # noqa: F401


class EmbeddingGCN(_layers.EmbeddingGCN):
    host_operands = True


class EmbeddingGCN2(_layers.EmbeddingGCN2):
    host_operands = True


class EmbeddingKWGCN(_layers.EmbeddingKWGCN):
    host_operands = True


class EmbeddingGCN_reg(_layers.EmbeddingGCN_reg):
    host_operands = True


for _c in (EmbeddingGCN, EmbeddingGCN2, EmbeddingKWGCN, EmbeddingGCN_reg):
    _c.__doc__ = getattr(_layers, _c.__name__).__doc__
del _c
