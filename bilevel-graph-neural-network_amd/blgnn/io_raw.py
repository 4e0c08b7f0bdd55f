"""Raw interaction-pair loaders and dataset assembly without networkx 2.2 / klepto: what the
reference's ``utils/data/load_raw_data.py`` does for this path, from the files a checkout holds.

  * ``load_ddi_pairs_tsv``       -- DrugBank ``ddi_snap.tsv`` (load_raw_data.py:20-38,89-117: one
                                    tab-separated pair of DrugBank ids per line, gid = int(id[2:]));
  * ``load_drugcombo_pairs_csv`` -- DrugCombo ``Syner&Antag_voting.csv`` (load_raw_data.py:40-57,
                                    120-139: header line, drug names in columns 1 and 2, label in the
                                    last column, majority vote per unordered pair); the name -> gid
                                    table lives in a klepto archive the checkout lacks, so the caller
                                    supplies it;
  * ``dataset_from_raw``         -- molecules from ``*.gexf`` (blgnn.io_gexf) + pairs -> BiGNNDataset with
                                    the reference's filtering (load_raw_data.py:142-166,172-196: only
                                    drugs that appear in the pair list AND have a structure with at least
                                    one bond; a pair needs both; graphs ordered by gid; one GraphPair per
                                    unordered pair).
  * ``codes_from_one_hot`` / ``MoleculeCodes`` -- the compact form of the node features: one int8 code per
                                    attribute (6 B per atom instead of 52 floats), expanded on the device.
"""

import collections

import numpy as np

from .dataset import BiGNNDataset, one_hot_from_codes
from .graph import MoleculeStore
from .io_gexf import NATTS, load_gexf_dir

__all__ = ["load_ddi_pairs_tsv", "load_drugcombo_pairs_csv", "dataset_from_raw", "codes_from_one_hot",
           "dataset_from_codes"]


def load_ddi_pairs_tsv(path, to_gid=lambda name: int(name[2:])):
    """-> int64 [P, 2] gid pairs in file order (duplicates and both orders kept, as the file has them)."""
    out = []
    with open(path, "r") as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            if len(parts) < 2 or not parts[0]:
                continue
            out.append((to_gid(parts[0]), to_gid(parts[1])))
    return np.asarray(out, dtype=np.int64).reshape(-1, 2)


def load_drugcombo_pairs_csv(path, name_to_gid):
    """-> (pairs int64 [P, 2], labels int64 [P]: 1 synergy, 2 antagonism), one row per csv line whose
    two drug names resolve.  The vote follows the reference to the letter, including its lookup of the
    antagonism count under the UNSORTED name pair (load_raw_data.py:131-134)."""
    with open(path, "r") as f:
        lines = f.readlines()[1:]
    counts = collections.defaultdict(lambda: collections.defaultdict(int))
    for line in lines:
        cols = line.split(",")
        counts[tuple(sorted([cols[1], cols[2]]))][cols[-1]] += 1
    pairs, labels = [], []
    for line in lines:
        cols = line.split(",")
        drugs = [cols[1], cols[2]]
        syn = counts[tuple(sorted(drugs))]["synergy\n"]
        ant = counts[tuple(drugs)]["antagonism\n"]
        try:
            a, b = name_to_gid[drugs[0].lower()], name_to_gid[drugs[1].lower()]
        except KeyError:
            continue
        pairs.append((a, b))
        labels.append(1 if syn >= ant else 2)
    return np.asarray(pairs, dtype=np.int64).reshape(-1, 2), np.asarray(labels, dtype=np.int64)


def _filter(gids, sizes_e, pairs, labels):
    """load_raw_data.py:142-196: drugs need a structure with >= 1 bond and must occur in the pair list;
    a pair needs both drugs; one entry per unordered pair (the first label seen wins).
    Deviation (parity unpinned: the klepto name -> gid table the reference's loader needs is not in the checkout):
    the reference's get_graph_pairs keeps a SET of edge types per unordered pair and lists the pair in every one
    of those edge-type graphs (load_raw_data.py:142-166); its DrugCombo antagonism vote is looked up under the
    unsorted name pair, so an 'A,B' line and a 'B,A' line can vote differently and such a pair sits in both
    graphs there, in one (the first line's) here.  BiGNNDataset models one label per pair."""
    have = {g for g, ne in zip(gids, sizes_e) if ne > 0}
    in_pairs = set(pairs.reshape(-1).tolist())
    have &= in_pairs
    seen, keep_p, keep_l = set(), [], []
    for (a, b), lab in zip(pairs.tolist(), labels.tolist()):
        if a not in have or b not in have:
            continue
        key = (a, b) if a <= b else (b, a)
        if key in seen:
            continue
        seen.add(key)
        keep_p.append(key)
        keep_l.append(lab)
    used = sorted({g for p in keep_p for g in p})
    return used, np.asarray(keep_p, dtype=np.int64).reshape(-1, 2), np.asarray(keep_l, dtype=np.int64)


def dataset_from_raw(gexf_dir, pairs, labels=None, name="drugbank", device="cuda:0", natts=NATTS):
    """Molecules of ``gexf_dir`` + interaction pairs -> BiGNNDataset (graphs ordered by gid)."""
    gids, xs, eis = load_gexf_dir(gexf_dir, natts)
    labels = np.ones(len(pairs), dtype=np.int64) if labels is None else np.asarray(labels)
    used, pairs, labels = _filter(gids, [e.shape[1] for e in eis], np.asarray(pairs), labels)
    pos = {g: i for i, g in enumerate(gids)}
    store = MoleculeStore.from_lists([xs[pos[g]] for g in used], [eis[pos[g]] for g in used], used, device)
    return BiGNNDataset(name, store, pairs, labels, multi_edge_types=len(np.unique(labels)) > 1)


def codes_from_one_hot(x, cards):
    """[A, sum(cards)] one-hot rows -> int8 [A, len(cards)] codes (exactly one 1 per attribute)."""
    x = np.asarray(x)
    out = np.empty((x.shape[0], len(cards)), dtype=np.int8)
    off = 0
    for a, c in enumerate(cards):
        blk = x[:, off:off + c]
        if not (blk.sum(axis=1) == 1).all():
            raise ValueError("attribute %d is not one-hot" % a)
        out[:, a] = blk.argmax(axis=1)
        off += c
    return out


def dataset_from_codes(gids, node_ptr, codes, cards, edge_ptr, edge_local, pairs, labels=None, name="drugbank",
                       device="cuda:0"):
    """The compact on-disk / on-wire form (6 bytes per atom) -> BiGNNDataset; the one-hot expansion
    (utils/data/representation_node_feat.py:86-97) happens on the device."""
    x = one_hot_from_codes(np.asarray(codes, dtype=np.int64), cards=tuple(int(c) for c in cards), device=device)
    store = MoleculeStore(x, np.asarray(edge_local, dtype=np.int64), node_ptr, edge_ptr, gids, device,
                          codes=np.asarray(codes), cards=[int(c) for c in cards])
    labels = np.ones(len(pairs), dtype=np.int64) if labels is None else np.asarray(labels)
    return BiGNNDataset(name, store, np.asarray(pairs, dtype=np.int64), labels,
                        multi_edge_types=len(np.unique(labels)) > 1)
