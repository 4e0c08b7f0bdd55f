"""Configuration contract of the hot path.

The reference's ``src/config.py`` is an argparse *script* evaluated at import
(config.py:467) whose only products the model code reads are a handful of flags
and the ``--layer_i`` spec strings (config.py:297-433).  Here the same contract
is a plain namespace: ``FLAGS`` is a process-wide object the layer modules read
(as ``from config import FLAGS`` does in the reference) and ``make_flags``
assembles the identical spec strings for the five model families.
"""

__all__ = ["FLAGS", "Flags", "make_flags", "set_flags", "MODELS"]

MODELS = (
    "higher_level_gnn",               # DECAGON  (config.py:96)
    "lower_level_gnn",                # LL-GNN
    "lower_level_gmn",                # MHCADDI
    "lower_level_gnn_higher_level",   # BI-GNN
    "higher_level_init_to_pred",      # FP-PRED
)


class Flags(object):
    """Attribute bag; ``vars(FLAGS)['layer_3']`` works as in layers_factory.py:17."""

    def update(self, other):
        self.__dict__.clear()
        self.__dict__.update(vars(other) if not isinstance(other, dict) else other)
        return self


FLAGS = Flags()


def set_flags(flags):
    """Install ``flags`` as the process-wide FLAGS (in place, so modules that did
    ``from .config import FLAGS`` see it)."""
    return FLAGS.update(flags)


def _node_embedding(gnn, out_dim, act, bn, normalize, in_dim=None, higher=None):
    parts = ["type=%s" % gnn]
    if in_dim is not None:
        parts.append("input_dim=%d" % in_dim)
    parts += ["output_dim=%d" % out_dim, "act=%s" % act, "bn=%s" % bn]
    if higher is not None:
        parts.append("higher_level=%s" % higher)
    parts.append("normalize=%s" % normalize)
    return "NodeEmbedding:" + ",".join(parts)


def _aggregation(node_aggr, append, style, d):
    if node_aggr == "avg_pool":
        return "NodeAggregation%s:style=avg_pool" % append
    if node_aggr == "deepsets":
        return "NodeAggregation:style=deepsets,in_dim=%d,out_dim=%d,num_mlp_layers=2" % (d, d)
    if node_aggr == "gmn_aggr":
        return "NodeAggregation:style=gmn_aggr,in_dim=%d,out_dim=%d" % (d, d)
    if node_aggr == "multi_scale":
        return "NodeAggregation:style=%s,concat_multi_scale=True,in_dim=%d,out_dim=%d" % (style, d, d)
    raise NotImplementedError(node_aggr)


def make_flags(model="higher_level_gnn", dataset="drugbank", lower_level_gnn_type="gat",
               higher_level_gnn_type="gat", D_lower=64, D_higher=64,
               lower_level_num_layers=None, higher_level_num_layers=3, batch_size=None,
               device="cuda:0", node_aggr=None, bn=True, gnn_normalize=False,
               gat_softmax_group="src", d_init=64, feature_transform="f16x3", multi_edge_node_model="gat",
               fuse_readout=True, gmn_f_node="MLP", gmn_sp=True):
    """Reproduce the flag values and layer list of src/config.py for ``model``.
    Defaults are the reference's (config.py:96,126-131,180,237-238)."""
    if model not in MODELS:
        raise ValueError("unknown model {}".format(model))
    drugcombo = "drugcombo" in dataset
    lower = "lower_level" in model
    higher = "higher_level" in model
    gmn = "gmn" in model
    if lower_level_num_layers is None:
        lower_level_num_layers = 3 if gmn else 5
    different_edge_type_aggr = drugcombo          # config.py:77-83
    # config.py:120-123 hard-codes 'gat_multi_edge_aggr'; model/layers_meta.py:68 also builds 'gcn'
    node_model = ("%s_multi_edge_aggr" % multi_edge_node_model) if different_edge_type_aggr else None
    edge_model = "none" if different_edge_type_aggr else None
    batch_unique_graphs = model != "lower_level_gmn"
    if node_aggr is None:
        node_aggr = "multi_scale" if lower else "avg_pool"
    append = "Pairs" if (not batch_unique_graphs and gmn) else ""
    if lower and higher:
        init_embds = "model_init"
    elif lower:
        init_embds = "no_init"
    else:
        init_embds = "rand_init"
    multi = drugcombo

    layers = []
    last_lower = None
    d_low = D_lower
    if lower:
        layers.append(_node_embedding(lower_level_gnn_type, D_lower, "relu", bn, gnn_normalize))
        if gmn:
            gm = "GMNPropagator:input_dim=%d,output_dim=%d,act=relu,bn=%s,f_node=%s" % (D_lower, D_lower, bn, gmn_f_node)
            layers.append(gm)
        for i in range(lower_level_num_layers - 1):
            act = "relu" if i != lower_level_num_layers - 2 else "identity"
            layers.append(_node_embedding(lower_level_gnn_type, D_lower, act, bn, gnn_normalize,
                                          in_dim=D_lower))
            if gmn:
                layers.append(gm)
        layers.append(_aggregation(node_aggr, append, "avg_pool", D_lower))
        if node_aggr == "multi_scale":
            d_low = D_lower * lower_level_num_layers
        if "model_init" in init_embds:
            last_lower = len(layers) - 1          # config.py:372-373 (c.t() - 1)
    if higher:
        layers.append("LoadInteractionLayer")
        if model != "higher_level_init_to_pred":
            n_high = higher_level_num_layers

            def upper(in_dim, act):
                if different_edge_type_aggr:
                    s = "MetaLayer:"
                    if in_dim is not None:
                        s += "input_dim=%d," % in_dim
                    return s + "output_dim=%d,act=%s,higher_level=True,edge_model=%s,node_model=%s" % (
                        D_higher, act, edge_model, node_model)
                return _node_embedding(higher_level_gnn_type, D_higher, act, bn, gnn_normalize,
                                       in_dim=in_dim, higher=True)

            layers.append(upper(d_low if lower else None, "relu"))
            n_high -= 1
            for i in range(n_high):
                layers.append(upper(D_higher, "relu" if i != n_high - 1 else "identity"))
    d_pred = D_higher if higher else d_low
    layers.append("LinkPredictor:type=mlp_concat,multi_label_pred=%s,mlp_dim=%d,batch_unique_graphs=%s"
                  % (multi, d_pred, batch_unique_graphs))
    layers.append("Loss:type=%s" % ("CE" if multi else "BCE"))

    f = Flags()
    f.model, f.dataset, f.device = model, dataset, device
    f.lower_level_layers, f.higher_level_layers = lower, higher
    f.lower_level_gnn_type, f.higher_level_gnn_type = lower_level_gnn_type, higher_level_gnn_type
    f.different_edge_type_aggr = different_edge_type_aggr
    f.use_hyper_edge_attrs = False
    f.batch_unique_graphs = batch_unique_graphs
    f.init_embds = init_embds
    f.multi_label = f.multi_class_pred = multi
    f.batch_size = batch_size if batch_size is not None else (128 if gmn else 64)
    f.d_init = d_init
    f.negative_sample = True
    f.num_negative_samples = 1
    f.pair_interaction = False
    f.gat_softmax_group = gat_softmax_group     # SURVEY section 8c(5); not a reference flag
    # how the fused GIN layer runs its two dense transforms (not a reference flag):
    #   "f16x3":  fp16 matrix cores, operands range-normalised and split in two pieces (three MFMAs per
    #             product), fp32-equivalent accuracy; falls back to bf16x6 where the kernel has no f16 variant
    #   "bf16x6": bf16 matrix cores, operands split in three pieces (six MFMAs), fp32-equivalent accuracy
    #   "f32":    exact-f32 MFMA (v_mfma_f32_32x32x2_f32), 2.7x the matrix-core time of bf16x6
    f.feature_transform = feature_transform
    # node -> graph readout accumulated by the producing GIN layers' epilogues instead of a separate
    # segment-reduce launch (blgnn/layers.py ReadoutPlan; not a reference flag)
    f.fuse_readout = fuse_readout
    f.gmn_sp = gmn_sp                                  # pre-split GEMM operands in the GMN propagator (layers_gmn.py)
    f.gmn_match_sp = True                              # d = 256: the attention on x's sp planes, sp result (gmn_match_sp.hip)
    f.gmn_match_streams = 1                            # side streams of the attention's size-class kernels (1 | 2)
    f.gmn_shared_prefix = True                         # pair batches: the layers in front of the first cross-graph
    #                                                    attention run on the distinct molecules (layers.py)
    f.gin_sp = True                                    # GIN layers at d >= 512 on large batches: aggregation into the sp format +
    #                                                    two pre-split GEMMs (layers.py; ops.gin_sp_preferred)
    f.gmn_pooled_tail = True                           # last propagator + mean / sum readout: pool the hidden rows in the
    #                                                    GEMM epilogue, second Linear per graph (layers_gmn.py)
    if last_lower is not None:
        f.last_lower_lyr_num = last_lower
    for i, s in enumerate(layers):
        setattr(f, "layer_%d" % (i + 1), s)
    f.layer_num = len(layers)
    return f
