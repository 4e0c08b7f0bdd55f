"""CPU oracle for the bi-level forward hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the *checker*.  Nothing under
``bilevel-graph-neural-network_amd/`` imports it: the product path is the HIP
library behind ``include/blgnn.h`` and it fails loudly when that library is
missing.

What it is: a pure-PyTorch (CPU, fp32 or fp64) restatement of the arithmetic the
reference performs on its forward hot path.  The reference delegates that
arithmetic to third-party wheels that are NOT vendored under /root/reference
(Dockerfile:31-35): torch-geometric==1.1.2, torch-scatter==1.1.2,
torch-sparse==0.2.4.  The restatement therefore follows (a) the reference's own
call sites (cited per function), (b) the plain-torch parts of the reference that
are fully visible (GMN match, MLP, LinkPred, merged-graph construction), and
(c) the published algorithm of the pinned PyG release for GINConv / GATConv /
GCNConv / scatter_* / to_undirected.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, and
it cannot be imported here (torch_geometric / torch_scatter / torch_sparse /
klepto absent, networkx==2.2 gate).  The oracle is pinned only by the
hand-computed known-answer tests in tests/test_oracle_kat.py and by the golden
fixtures under tests/golden/ which were produced BY this oracle from the
reference's own drugs_small molecules (script: tests/golden/make_golden.py).
The one semantic that cannot be settled from the visible call sites -- which
end of an edge GATConv 1.1.2 groups its softmax over -- is a parameter
(``softmax_group`` in {"src", "dst"}); both are implemented and tested.
"""

from .ops import *          # noqa: F401,F403
from .forward import *      # noqa: F401,F403
