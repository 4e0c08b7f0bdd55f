"""Molecule loader for the reference's ``.gexf`` files (stdlib XML; no networkx 2.2,
no klepto).  Produces the same tensors the reference's pipeline would:
  * nodes must be 0..n-1 (utils/data/load_raw_data.py:205-212),
  * node features: for each attribute in ``natts`` order
    (utils/data/dataset.py:15-16), the sorted set of distinct values over the WHOLE
    collection -> index -> one-hot, concatenated
    (utils/data/representation_node_feat.py:10-33,86-97),
  * edges: undirected, symmetrised and sorted (model/layers_util.py:148-167).
"""

import glob
import os
import xml.etree.ElementTree as ET

import numpy as np

from .graph import to_undirected_host

__all__ = ["NATTS", "read_gexf", "load_gexf_dir", "encode_one_hot"]

NATTS = ['atom_type', 'aromatic', 'acceptor', 'donor', 'hybridization', 'num_h']
_NS = {"g": "http://www.gexf.net/1.2draft"}


def _convert(value, typ):
    # networkx's gexf reader: boolean -> bool, long/integer -> int, string -> str
    if typ == "boolean":
        return value == "true"
    if typ in ("long", "integer", "int"):
        return int(value)
    if typ in ("float", "double"):
        return float(value)
    return value


def read_gexf(path):
    """-> (num_nodes, attrs: {title: [value per node]}, edges int [E,2])."""
    root = ET.parse(path).getroot()
    graph = root.find("g:graph", _NS)
    titles = {}
    for a in graph.findall("g:attributes[@class='node']/g:attribute", _NS):
        titles[a.get("id")] = (a.get("title"), a.get("type"))
    nodes = graph.findall("g:nodes/g:node", _NS)
    ids = sorted(int(n.get("id")) for n in nodes)
    if ids != list(range(len(ids))):
        raise ValueError("{}: node ids must be 0..n-1".format(path))
    attrs = {t: [None] * len(ids) for t, _ in titles.values()}
    for n in nodes:
        i = int(n.get("id"))
        for av in n.findall("g:attvalues/g:attvalue", _NS):
            title, typ = titles[av.get("for")]
            attrs[title][i] = _convert(av.get("value"), typ)
    edges = [(int(e.get("source")), int(e.get("target")))
             for e in graph.findall("g:edges/g:edge", _NS)]
    return len(ids), attrs, np.asarray(edges, dtype=np.int64).reshape(-1, 2)


def encode_one_hot(all_attrs, natts=NATTS):
    """all_attrs: list (one per molecule) of {title: [values]} -> (list of float32
    [n_i, F] matrices, F).  Sorting follows python's ordering of the raw values, as
    ``sorted(inputs_set)`` does in representation_node_feat.py:20-21."""
    vocab = {}
    for t in natts:
        vals = set()
        for a in all_attrs:
            vals |= set(a[t])
        vocab[t] = {v: i for i, v in enumerate(sorted(vals))}
    F = sum(len(vocab[t]) for t in natts)
    xs = []
    for a in all_attrs:
        n = len(a[natts[0]])
        x = np.zeros((n, F), dtype=np.float32)
        off = 0
        for t in natts:
            idx = np.asarray([vocab[t][v] for v in a[t]], dtype=np.int64)
            x[np.arange(n), off + idx] = 1.0
            off += len(vocab[t])
        xs.append(x)
    return xs, F


def load_gexf_dir(path, natts=NATTS):
    """-> (gids, xs, edge_indices) for every ``*.gexf`` under ``path`` (sorted by file
    name; gid = numeric part of the DrugBank id)."""
    files = sorted(glob.glob(os.path.join(path, "*.gexf")))
    gids, all_attrs, eis = [], [], []
    for f in files:
        n, attrs, edges = read_gexf(f)
        stem = os.path.splitext(os.path.basename(f))[0]
        gids.append(int("".join(ch for ch in stem if ch.isdigit())))
        all_attrs.append(attrs)
        eis.append(to_undirected_host(edges, n))
    xs, _ = encode_one_hot(all_attrs, natts)
    return gids, xs, eis
