"""blgnn -- host side of the MI355X bi-level message-passing engine.

Mirrors the ``nn.Module`` surface of codeKgu/BiLevel-Graph-Neural-Network's
``model/layers*.py`` for the forward hot path; all arithmetic is in libblgnn.so
(include/blgnn.h).  Importing this package loads the library and fails loudly if it
has not been built.
"""

from . import _lib

_lib.load()

from .config import FLAGS, make_flags, set_flags                       # noqa: E402
from . import ops                                                       # noqa: E402
from .graph import GraphData, MoleculeStore, InteractionGraph           # noqa: E402
from .dataset import BiGNNDataset, TrainData, make_synthetic            # noqa: E402
from .batch import BatchData                                            # noqa: E402
from .layers import NodeEmbedding, Loss                                 # noqa: E402
from .layers_gmn import GMNPropagator                                   # noqa: E402
from .layers_aggregation import NodeAggregation, NodeAggregationPairs   # noqa: E402
from .layers_load_interaction_graph import LoadInteractionGraph         # noqa: E402
from .layers_meta import MetaLayerWrapper                               # noqa: E402
from .layers_link_pred import LinkPred                                  # noqa: E402
from .layers_factory import create_layers, layer_ctors                  # noqa: E402
from .model import Model                                                # noqa: E402
from .engine import BiLevelEngine                                       # noqa: E402
from .train_step import TrainStep                                       # noqa: E402

__version__ = "0.1.0"
