"""``GMNPropagator`` -- drop-in for the reference's model/layers_gmn.py.

What changes underneath (results equal to fp32 rounding):
  * message:  m_sum_i = sum_e f_message(cat(x_i, x_j)).  f_message is
    Linear(2D,2D) -> relu -> Linear(2D,2D) with NO activation after the last
    Linear (layers_util.py:48-49), so the sum over edges commutes with it:
        m_sum_i = W2 . ( sum_e relu(W1a x_i + W1b x_j + b1) ) + deg_i * b2
    The E x 2D message tensor and the two per-edge GEMMs disappear: one node-level
    GEMM (x -> [P|Q], N=4D), one edge kernel (sum_e relu(P_i + Q_j)) and one
    node-level GEMM with a rank-1 (deg_i * b2) term.
  * match: the python loop over pairs (layers_gmn.py:60-70, ~12 tiny kernels per
    pair, an n1 x n2 x D temporary, S recomputed for the mirror) is one kernel.
  * f_node input cat(x, m_sum, u_sum) is never concatenated: producers write
    straight into column slices of one [Nn, 4D] buffer.
  * m_sum itself is never formed: it only feeds f_node's first Linear (eval BN folded in),
        f_in @ W1 = x @ W1[:D] + (S @ W2 + deg b2) @ W1[D:3D] + u_sum @ W1[3D:]
                  = [x | S | u_sum] @ [W1[:D] ; W2 W1[D:3D] ; W1[3D:]] + deg (b2 W1[D:3D])
    with S = sum_e relu(P_i + Q_j).  W2 W1[D:3D] is a product of WEIGHTS (formed once in
    fp64, cached until a parameter changes), so the [Nn, 2D] x [2D, 2D] GEMM of every layer
    disappears and the edge kernel writes S straight into the f_node input.
"""

import os

import torch
import torch.nn as nn

from . import ops
from .config import FLAGS
from .layers_util import MLP, DerivedCache, act_name

__all__ = ["GMNPropagator"]


_MATCH_SP_ENV = os.environ.get("BLGNN_GMN_MATCH_SP", "1") != "0"      # A/B switch of the sp attention (bench / profiles)


class GMNPropagator(nn.Module):
    def __init__(self, input_dim, output_dim, distance_metric='cosine', f_node='MLP'):
        super().__init__()
        self.out_dim = output_dim
        self.input_dim = input_dim
        if distance_metric == 'cosine':
            self.distance_metric = nn.CosineSimilarity()
        elif distance_metric == 'euclidean':
            self.distance_metric = nn.PairwiseDistance()
        # (sic) the reference's attribute name, kept for state-dict compatibility
        self.f_messasge = MLP(2 * input_dim, 2 * input_dim, num_hidden_lyr=1,
                              hidden_channels=[2 * input_dim])
        self.f_node_name = f_node
        if f_node == 'MLP':
            self.f_node = MLP(4 * input_dim, output_dim, num_hidden_lyr=1, bn=True)
        elif f_node == 'GRU':
            self.f_node = nn.GRUCell(3 * input_dim, input_dim)
        else:
            raise ValueError("{} for f_node has not been implemented".format(f_node))
        self._cache = DerivedCache()

    _streams = {}

    @classmethod
    def _side_stream(cls, device, index=0):
        """side streams per launch stream (two forwards pipelined on two streams keep separate side streams)"""
        key = (str(device), torch.cuda.current_stream(device).cuda_stream, index)
        if key not in cls._streams:
            cls._streams[key] = torch.cuda.Stream(device=device)
        return cls._streams[key]

    def _message_weights(self):
        D = self.input_dim
        lin1, lin2 = self.f_messasge.layers[0], self.f_messasge.layers[1]

        def split(w, b):
            # [x_i | x_j] @ W1^T  ==  x_i @ W1[:, :D]^T + x_j @ W1[:, D:]^T ; bias rides on P
            wpq = torch.cat([w[:, :D].t(), w[:, D:].t()], dim=1).contiguous()        # [D, 4D]
            bpq = torch.cat([b, torch.zeros_like(b)]).contiguous()                   # [4D]
            return wpq, bpq

        wpq, bpq = self._cache.get("pq", split, lin1.weight, lin1.bias)
        w2 = self._cache.get("w2", lambda w: w.t().contiguous(), lin2.weight)
        return wpq, bpq, w2, lin2.bias

    def _node_weights(self):
        """f_node's first Linear (+ eval BN) with the message MLP's last Linear folded into its rows
        D..3D: -> (w [4D, 4D], bias [4D], rank-1 column factor [4D]), cached."""
        D = self.input_dim
        lin2 = self.f_messasge.layers[1]
        lin, bn = self.f_node.layers[0], self.f_node.bn[0]
        if bn.training:
            raise NotImplementedError("blgnn: train-mode BatchNorm is outside the forward-parity scope")

        def compose(w2, b2, w, b, g, beta, mean, var):
            scale = (g.double() / torch.sqrt(var.double() + bn.eps))
            w1 = w.double().t() * scale.view(1, -1)                                   # [4D in, 4D out]
            b1 = b.double() * scale + beta.double() - mean.double() * scale
            mid = w1[D:3 * D]                                                         # rows fed by m_sum
            wc = torch.cat([w1[:D], w2.double().t() @ mid, w1[3 * D:]], dim=0)
            return wc.float().contiguous(), b1.float().contiguous(), (b2.double() @ mid).float().contiguous()

        return self._cache.get("node1", compose, lin2.weight, lin2.bias, lin.weight, lin.bias, bn.weight,
                               bn.bias, bn.running_mean, bn.running_var)

    def _node_weights_split(self):
        """``_node_weights`` with the composed first Linear cut at row 3D: -> (w[:3D] for [x | S], w[3D:] for u, bias,
        rank-1 factor).  The two row slices are kept (views of the cached matrix) so that the packed-weight cache of
        ``ops.linear_sp`` recognises them from call to call."""
        w1, b1, rw = self._node_weights()
        D = self.input_dim
        sp = getattr(self, "_w1_split", None)
        if sp is None or sp[0] is not w1:
            sp = self._w1_split = (w1, w1[:3 * D], w1[3 * D:])
        return sp[1], sp[2], b1, rw

    def forward(self, ins, batch_data, model, edge_index=None):
        ops.require_device(ins, "ins")
        x = ins
        merge = batch_data.merge_data['merge']
        if edge_index is not None and edge_index is not merge.edge_index:
            csr = ops.csr_from_edge_index(edge_index, x.shape[0])
        else:
            csr = merge.csr()
        n, D = x.shape
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x, csr, merge, model)
        if self.f_node_name == 'GRU':
            return self._forward_gru(x, csr, merge, model)
        if self._use_sp(n, D):
            return self._forward_sp(x, csr, merge, model, batch_data)
        with torch.no_grad():
            wpq, bpq, _, _ = self._message_weights()
            w1, b1, rw = self._node_weights()
            f_in = getattr(x, "_blgnn_wide", None)               # the producing GIN layer already wrote x there
            if f_in is None or tuple(f_in.shape) != (n, 4 * D) or f_in.data_ptr() != x.data_ptr():
                f_in = torch.empty((n, 4 * D), dtype=torch.float32, device=x.device)
                f_in[:, :D].copy_(x)
            # The cross-graph attention is latency-bound (small per-pair matrix products, ~10 % of the HBM bandwidth,
            # two workgroups per CU) and depends on x only: it runs on a side stream next to the message GEMM and the
            # edge kernel and joins before f_node reads its columns (configs[2]: 28.65 -> 27.5 ms/step; launching it
            # after the GEMM, next to the edge kernel only, gained nothing).
            side = self._side_stream(x.device) if (x.is_cuda and getattr(FLAGS, "gmn_overlap_match", True)) else None
            seg = merge.segments()                               # (built on the launch stream, before the fork)
            h3 = ops.linear_writes_amax(x, wpq)                           # the f16x3 GEMMs take row magnitudes
            ax = getattr(x, "_blgnn_amax", None) if h3 else None          # ... of x: from the producing GIN layer
            if h3 and ax is None:
                ax = ops.row_absmax(x)
            act = act_name(self.f_node.activation)
            hid_amax = None
            if ops.linear_writes_amax(f_in, w1):
                hid_amax = torch.zeros(n, dtype=torch.float32, device=x.device)
            # ... of f_in = [x | S | u]: one pass over the [n, 4D] buffer -- or (FLAGS.gmn_fused_amax) merged by the
            # producers of S and u themselves, by atomic maximum into a copy of x's.  Exact either way; measured
            # 24.63 / 24.72 vs 24.54 / 24.62 ms per step at configs[2]: the attention kernels cannot share a CU with the
            # GEMM (LDS), so they sit on the critical path behind it and the row reductions added to them cost what the
            # removed pass saved.  Off by default.
            af = ax.clone() if (hid_amax is not None and ax is not None and getattr(FLAGS, "gmn_fused_amax", False)) else None
            if side is not None:
                cur = torch.cuda.current_stream(x.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    ops.gmn_match(seg, x, out=f_in[:, 3 * D:], row_amax=af, max_nodes=self._max_nodes(batch_data),
                                  order=self._pair_order(batch_data, x.device))
            pq = ops.linear(x, wpq, bpq, row_amax=ax)                     # [n, 4D] = [P | Q]
            ops.edge_relu_sum(csr, pq[:, :2 * D], pq[:, 2 * D:], out=f_in[:, D:3 * D], row_amax=af)
            if side is not None:
                cur.wait_stream(side)
            else:
                ops.gmn_match(seg, x, out=f_in[:, 3 * D:], row_amax=af, max_nodes=self._max_nodes(batch_data),
                                  order=self._pair_order(batch_data, x.device))
            if af is None and hid_amax is not None:
                af = ops.row_absmax(f_in)
            hid = ops.linear(f_in, w1, b1, act=act, rowvec=ops.row_degree(csr), rowvec_w=rw, amax_out=hid_amax, row_amax=af)
            out = ops.linear(hid, self.f_node.weight_kn(1), self.f_node.layers[1].bias, row_amax=hid_amax)
        model.store_layer_output(self, out)
        return out

    @staticmethod
    def _max_nodes(batch_data):
        """largest graph of the batch, from the host-side ind_list (src/merged_graph.py:56-57), cached on the merge
        object: the attention's size-class kernels that no pair needs are then not launched"""
        merge = batch_data.merge_data['merge']
        mx = getattr(merge, "_max_graph_nodes", None)
        if mx is None:
            ind = batch_data.merge_data.get('ind_list') if hasattr(batch_data.merge_data, "get") else None
            if ind is not None and hasattr(ind, "sizes"):
                mx = int(ind.sizes().max()) if len(ind) else 0
            else:
                mx = max((int(e) - int(s) for s, e in ind), default=0) if ind else 0
            merge._max_graph_nodes = mx
        return mx

    @staticmethod
    def _pair_order(batch_data, device):
        """the batch's pairs grouped by size class (ops.match_pair_order), cached on the merge object"""
        merge = batch_data.merge_data['merge']
        po = getattr(merge, "_match_pair_order", None)
        if po is None and hasattr(ops, "match_pair_order"):
            ind = batch_data.merge_data.get('ind_list') if hasattr(batch_data.merge_data, "get") else None
            po = ops.match_pair_order(ind, device) if ind is not None and len(ind) else None
            merge._match_pair_order = po
        return po

    @staticmethod
    def _use_sp(n, D):
        """The pre-split ("sp") pipeline: FLAGS.gmn_sp = True (default: when the GEMMs are large enough for the
        256 x 256 tiles), "force" (any size: tests) or False (the in-kernel split of blgnn_linear_h3)."""
        mode = getattr(FLAGS, "gmn_sp", True)
        if not hasattr(ops, "SpRows"):                 # (a stand-in backend without the sp kernels: host-logic tests)
            return False
        if not mode or getattr(FLAGS, "feature_transform", "f16x3") != "f16x3" or D % ops.SpRows.BLOCK != 0 or 4 * D > 2048:
            return False
        return mode == "force" or ops.linear_sp_supported(n, D, 4 * D)

    def _forward_sp(self, x, csr, merge, model, batch_data):
        """Eval forward with every GEMM operand in the sp format (include/blgnn.h: blgnn_linear_sp): each producer
        splits its result ONCE into two fp16 piece planes with a power-of-two scale per (row, 256 columns) -- the
        GIN layer x (its store loop), the edge kernel S, f_node's first Linear the hidden layer (GEMM epilogue); the
        attention's u by one streaming pass on its side stream -- and the three GEMMs stream both operands by LDS-DMA
        with all eight waves on the matrix cores.  Same arithmetic as ``forward``'s default path (three fp16 MFMA
        products per fp32 product); no row-magnitude passes, no [n, 4D] fp32 f_node input."""
        n, D = x.shape
        with torch.no_grad():
            wpq, bpq, _, _ = self._message_weights()
            w1, b1, rw = self._node_weights()
            # First propagator of a pair batch (the producing layer ran on the distinct molecules and replicated x:
            # layers.py ``_shared_molecules``): f_message and the message sum see no cross-graph edge either, and
            # f_node's first Linear is linear in its input columns -- [x | S | u] @ W = [x | S] @ W[:3D] + u @ W[3D:] --
            # so everything but the attention and the u part of that product stays on the distinct molecules; the
            # [x | S] part reaches every copy as a gathered addend in the epilogue of the u product.
            sh = getattr(x, "_blgnn_shared", None)
            f_sp = getattr(x, "_blgnn_sp", None)                 # the producing GIN layer wrote x's planes there
            if sh is not None:
                shared, x_u, f_sp_u = sh
                nu = x_u.shape[0]
                if f_sp_u is None or tuple(f_sp_u.shape) != (nu, 4 * D):
                    f_sp_u = ops.SpRows(nu, 4 * D, x.device)
                    ops.split_rows_sp(x_u, out=f_sp_u.cols(0, D))
                u_sp = ops.SpRows(n, D, x.device)
            else:
                if f_sp is None or tuple(f_sp.shape) != (n, 4 * D):
                    f_sp = ops.SpRows(n, 4 * D, x.device)
                    ops.split_rows_sp(x, out=f_sp.cols(0, D))
                u_sp = f_sp.cols(3 * D, 4 * D)
            side = self._side_stream(x.device) if (x.is_cuda and getattr(FLAGS, "gmn_overlap_match", True)) else None
            seg = merge.segments()                               # (built on the launch stream, before the fork)

            max_nodes = self._max_nodes(batch_data)
            order = self._pair_order(batch_data, x.device)

            # the attention on the sp planes of x with an sp result (csrc/gmn_match_sp.hip; d = 256): no fp32 u, no
            # split pass; in the shared-prefix case it reads the distinct molecules' planes through the row map
            match_sp = D == ops.SpRows.BLOCK and getattr(FLAGS, "gmn_match_sp", True) and _MATCH_SP_ENV
            x_planes = (f_sp_u if sh is not None else f_sp).cols(0, D)
            x_rows = shared.row_map if sh is not None else None
            u = None if match_sp else torch.empty((n, D), dtype=torch.float32, device=x.device)

            def match_all():
                if match_sp:
                    ops.gmn_match_sp(seg, x_planes, u_sp, order, row_map=x_rows, x=x, max_nodes=max_nodes)
                else:
                    ops.gmn_match(seg, x, out=u, max_nodes=max_nodes, order=order)
                    ops.split_rows_sp(u, out=u_sp)

            n_big = int(order[1][2]) + int(order[1][3])
            # (resolved HERE, on the launch stream: _side_stream keys on the current stream)
            side2 = self._side_stream(x.device, 1) if side is not None else None

            def fork_match(which="all"):
                # two side streams: the rarer size classes of the attention (pairs with a graph above 64 nodes: the fp32
                # kernels + the split of their rows -- few workgroups, long latency) run beside the common ones instead
                # of behind them.  FLAGS.gmn_match_streams = 2: both forked at the same point; fork "split" (below): the
                # rare classes beside the message GEMM, the common ones beside the edge kernel.
                cur = torch.cuda.current_stream(x.device)
                two = (getattr(FLAGS, "gmn_match_streams", 1) >= 2 or which != "all") and n_big > 0
                if which in ("all", "big") and two:
                    side2.wait_stream(cur)
                    with torch.cuda.stream(side2):
                        if match_sp:
                            ops.gmn_match_sp(seg, x_planes, u_sp, order, row_map=x_rows, x=x, max_nodes=max_nodes, part="big")
                        else:
                            ops.gmn_match(seg, x, out=u, max_nodes=max_nodes, classes=6, order=order)
                            u.record_stream(side2)
                if which == "big":
                    return
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    if two and match_sp:
                        ops.gmn_match_sp(seg, x_planes, u_sp, order, row_map=x_rows, x=x, max_nodes=max_nodes, part="small")
                        side.wait_stream(side2)
                    elif two:
                        ops.gmn_match(seg, x, out=u, max_nodes=max_nodes, classes=1, order=order)
                        side.wait_stream(side2)
                        ops.split_rows_sp(u, out=u_sp)
                    else:
                        match_all()
                    if u is not None:
                        u.record_stream(side)

            # Where the attention forks off (FLAGS.gmn_match_fork): "message" (default) = beside the message GEMM AND the
            # edge kernel, "edge" = beside the edge kernel only.  The attention's workgroups (66-133 KB of LDS) and the
            # GEMM's (141 KB) exclude each other on a CU, so beside the GEMM the two add up (the K = 256 GEMM stretches from
            # 1.0 to 1.7 ms in the step's kernel trace); the edge kernel is HBM-bound and uses no LDS, but its 0.6 ms do
            # not cover the attention's 1.3 ms chain.  Measured A/B in one process on the plain path (ms per step,
            # configs[2]): message 20.04 / 19.95, edge 20.82 / 20.74; with two attention streams 20.07 / 20.07 and
            # 20.60 / 20.40; the common size class beside the GEMM and the two rarer ones beside the edge kernel: 20.47 vs
            # 19.83.
            fork = getattr(FLAGS, "gmn_match_fork", "message") if side is not None else None
            if fork == "message":
                fork_match()
            elif fork == "split":
                fork_match("big")
            if sh is not None:
                csr_u = shared.graph.csr()
                pq, _ = ops.linear_sp(f_sp_u.cols(0, D), wpq, bpq)                        # [nu, 4D] = [P | Q], fp32
                if fork == "edge":
                    fork_match()
                elif fork == "split":
                    fork_match("small")
                ops.edge_relu_sum(csr_u, pq[:, :2 * D], pq[:, 2 * D:], out_sp=f_sp_u.cols(D, 3 * D))
                w1a, w1b, b1, rw = self._node_weights_split()
                part, _ = ops.linear_sp(f_sp_u.cols(0, 3 * D), w1a, b1, rowvec=ops.row_degree(csr_u), rowvec_w=rw)
            else:
                pq, _ = ops.linear_sp(f_sp.cols(0, D), wpq, bpq)                          # [n, 4D] = [P | Q], fp32
                if fork == "edge":
                    fork_match()
                elif fork == "split":
                    fork_match("small")
                ops.edge_relu_sum(csr, pq[:, :2 * D], pq[:, 2 * D:], out_sp=f_sp.cols(D, 3 * D))
            if side is not None:
                torch.cuda.current_stream(x.device).wait_stream(side)
            else:
                match_all()
            act = act_name(self.f_node.activation)
            kind, last = self._readout_of(model)
            tail = kind if (last and sh is None and act in ("relu", "identity")) else None
            if tail is not None:
                # LAST propagator in front of a mean / sum readout (model/layers_aggregation.py:16-19,32-41): the second
                # Linear of f_node has no activation behind it, so it commutes with the readout --
                # pool(hid @ W + b) = pool(hid) @ W + b -- and the hidden rows are pooled in the epilogue of the GEMM
                # that produces them: neither they nor the layer's node rows are ever written, the second Linear runs
                # on one row per graph.
                G = seg.numel() - 1
                pooled_hid = torch.zeros((G, 4 * D), dtype=torch.float32, device=x.device)
                ops.linear_sp(f_sp, w1, b1, act=act, rowvec=ops.row_degree(csr), rowvec_w=rw,
                              pool=(merge.row_graph(), seg, pooled_hid, tail), f32=False)
                scale = getattr(merge, "_bias_rows", {}).get(tail)           # per batch, like the segment table itself
                if scale is None:
                    sizes = (seg[1:] - seg[:-1]).to(torch.float32)          # (index arithmetic on the segment table)
                    scale = (sizes if tail == "sum" else (sizes > 0).to(torch.float32)).contiguous()
                    if not hasattr(merge, "_bias_rows"):
                        merge._bias_rows = {}
                    merge._bias_rows[tail] = scale
                out = ops.linear(pooled_hid, self.f_node.weight_kn(1), None, rowvec=scale,
                                 rowvec_w=self.f_node.layers[1].bias)
                out._blgnn_pooled = tail                                    # [G, D]: the readout takes it as is
                model.store_layer_output(self, out)
                return out
            hid = ops.SpRows(n, 4 * D, x.device)
            if sh is not None:
                ops.linear_sp(u_sp, w1b, None, act=act, addend=part, addend_rows=shared.row_map, out_sp=hid)
            else:
                ops.linear_sp(f_sp, w1, b1, act=act, rowvec=ops.row_degree(csr), rowvec_w=rw, out_sp=hid)
            if kind is not None:
                # an earlier propagator under a multi-scale mean / sum readout: its node rows feed the next layer AND the
                # readout -- the second Linear's epilogue pools them on the way out (the readout then copies [G, D])
                pooled = torch.zeros((seg.numel() - 1, D), dtype=torch.float32, device=x.device)
                out, _ = ops.linear_sp(hid, self.f_node.weight_kn(1), self.f_node.layers[1].bias,
                                       pool=(merge.row_graph(), seg, pooled, kind), f32=True)
                out._blgnn_pool_of = (kind, pooled)
            else:
                out, _ = ops.linear_sp(hid, self.f_node.weight_kn(1), self.f_node.layers[1].bias)
        model.store_layer_output(self, out)
        return out

    def _readout_of(self, model):
        """-> (kind, last): ``kind`` = "mean" | "sum" when the readout of ``model.layers`` is a plain mean / sum pooling of
        a lower-level-only model that reads THIS propagator's output (the layer right behind the last propagator; with
        ``concat_multi_scale`` every propagator's output), else None; ``last`` = this is the last propagator, i.e.
        nothing but that readout reads its node rows (the reference's ``get_layer_output`` has no caller, ``model.acts``
        is read by the readout alone).  FLAGS.gmn_pooled_tail = False: (None, False), node rows and a separate readout."""
        if not getattr(FLAGS, "gmn_pooled_tail", True) or getattr(FLAGS, "higher_level_layers", False):
            return None, False
        layers = list(getattr(model, "layers", []))
        at = [k for k, l in enumerate(layers) if l is self]
        if len(at) != 1 or model.acts is None or len(model.acts) != at[0] + 1:
            return None, False                              # (not driven layer by layer over model.layers)
        from .layers_aggregation import NodeAggregationPairs, _SegmentPool
        props = [k for k, l in enumerate(layers) if isinstance(l, GMNPropagator)]
        nxt = layers[props[-1] + 1] if props[-1] + 1 < len(layers) else None
        if not isinstance(nxt, NodeAggregationPairs) or not isinstance(nxt.agg_func, _SegmentPool):
            return None, False
        last = at[0] == props[-1]
        if not last and not nxt.concat_multi_scale:
            return None, False
        return nxt.agg_func.kind, last

    def _forward_gru(self, x, csr, merge, model):
        """f_node='GRU' (model/layers_gmn.py:27-30,48-50; non-default): out = GRUCell(cat(m_sum, u_sum), x).  The
        message sum and the attention as in the MLP variant, the two gate GEMMs through blgnn_linear, the gate
        arithmetic in one kernel (train mode: ``_forward_train``)."""
        n, D = x.shape
        cell = self.f_node
        with torch.no_grad():
            wpq, bpq, w2, b2 = self._message_weights()
            pq = ops.linear(x, wpq, bpq)
            s = ops.edge_relu_sum(csr, pq[:, :2 * D], pq[:, 2 * D:])
            gin = torch.empty((n, 3 * D), dtype=torch.float32, device=x.device)            # [m_sum | u_sum]
            ops.linear(s, w2, None, rowvec=ops.row_degree(csr), rowvec_w=b2, out=gin[:, :2 * D])
            ops.gmn_match(merge.segments(), x, out=gin[:, 2 * D:])
            w_ih = self._cache.get("w_ih", lambda w: w.t().contiguous(), cell.weight_ih)
            w_hh = self._cache.get("w_hh", lambda w: w.t().contiguous(), cell.weight_hh)
            gi = ops.linear(gin, w_ih, cell.bias_ih)
            gh = ops.linear(x, w_hh, cell.bias_hh)
            out = ops.gru_gates(gi, gh, x)
        model.store_layer_output(self, out)
        return out

    def _forward_train(self, x, csr, merge, model):
        """Train mode (src/train.py:177-182): the same restructured message, the cross-graph attention and the f_node
        MLP with train-mode BatchNorm as autograd nodes on libblgnn kernels (blgnn/autograd.py).  The weight-only
        products of the eval path (W2 folded into f_node) are not used: every Linear keeps its own gradient."""
        from . import autograd as ag
        D = self.input_dim
        lin1, lin2 = self.f_messasge.layers[0], self.f_messasge.layers[1]
        p = ag.linear(x, lin1.weight[:, :D], lin1.bias)                       # [x_i | x_j] @ W1^T split by halves
        q = ag.linear(x, lin1.weight[:, D:], None)
        s = ag.edge_relu_sum(p, q, csr)                                        # sum_e relu(P_i + Q_j)
        # m_sum = S @ W2^T + deg * b2 as ONE product on [S | deg] and [W2 | b2] (concatenation: data movement)
        deg = ops.row_degree(csr).view(-1, 1)
        m_sum = ag.linear(torch.cat([s, deg], dim=1), torch.cat([lin2.weight, lin2.bias.view(-1, 1)], dim=1), None)
        u = ag.gmn_match(x, merge.segments())
        if isinstance(self.f_node, nn.GRUCell):
            # out = GRUCell(cat(m_sum, u_sum), x): two gate products + the gate arithmetic, each with its backward kernel
            cell = self.f_node
            gi = ag.linear(torch.cat([m_sum, u], dim=1), cell.weight_ih, cell.bias_ih)
            gh = ag.linear(x, cell.weight_hh, cell.bias_hh)
            out = ag.gru_gates(gi, gh, x)
        else:
            out = self.f_node(torch.cat([x, m_sum, u], dim=1))
        model.store_layer_output(self, out)
        return out

    def f_match(self, x, batch_data):
        """layers_gmn.py:55-72 as a single kernel (kept as a method for callers)."""
        return ops.gmn_match(batch_data.merge_data['merge'].segments(), x)
