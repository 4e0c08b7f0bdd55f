#!/usr/bin/env python3
"""Builds tests/golden/drugbank_real.npz -- DATA ONLY -- from the reference checkout's DrugBank files
(/root/reference/data/DrugBank/ddi_data: 1 335 molecule .gexf files + ddi_snap.tsv), with this repo's
own stdlib loaders (blgnn.io_gexf, blgnn.io_raw).  The fixture holds the molecules in compact form
(int8 attribute codes, local int16 bond lists) and the interaction pairs after the reference's filtering,
so the GPU box can run BASELINE configs[0] / configs[1] at the REAL DrugBank shape without the reference.
Run in the build container only:  python tests/golden/make_drugbank.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bilevel-graph-neural-network_amd")]
SRC = "/root/reference/data/DrugBank/ddi_data"


def main():
    from blgnn import io_gexf, io_raw
    gids, xs, eis = io_gexf.load_gexf_dir(os.path.join(SRC, "drugs_snap"))
    raw = io_raw.load_ddi_pairs_tsv(os.path.join(SRC, "ddi_snap.tsv"))
    used, pairs, labels = io_raw._filter(gids, [e.shape[1] for e in eis], raw, np.ones(len(raw), dtype=np.int64))
    pos = {g: i for i, g in enumerate(gids)}
    # attribute vocabularies as encode_one_hot built them -> cardinalities
    all_attrs = []
    import glob
    for f in sorted(glob.glob(os.path.join(SRC, "drugs_snap", "*.gexf"))):
        all_attrs.append(io_gexf.read_gexf(f)[1])
    cards = [len(set(v for a in all_attrs for v in a[t])) for t in io_gexf.NATTS]
    xs_u = [xs[pos[g]] for g in used]
    eis_u = [eis[pos[g]] for g in used]
    codes = np.concatenate([io_raw.codes_from_one_hot(x, cards) for x in xs_u], axis=0)
    node_ptr = np.concatenate([[0], np.cumsum([x.shape[0] for x in xs_u])]).astype(np.int32)
    edge_ptr = np.concatenate([[0], np.cumsum([e.shape[1] for e in eis_u])]).astype(np.int32)
    edge_local = np.concatenate(eis_u, axis=1).astype(np.int16)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "drugbank_real.npz")
    np.savez_compressed(out, gids=np.asarray(used, dtype=np.int32), node_ptr=node_ptr, codes=codes,
                        cards=np.asarray(cards, dtype=np.int32), edge_ptr=edge_ptr, edge_local=edge_local,
                        pairs=pairs.astype(np.int32), raw_lines=np.int64(len(raw)))
    print("molecule files %d, with bonds and in the pair list %d; atoms %d, directed bonds %d, F_in %d; "
          "tsv lines %d -> unordered pairs %d; %s = %.0f KB" % (
              len(gids), len(used), codes.shape[0], edge_local.shape[1], sum(cards), len(raw), len(pairs), out,
              os.path.getsize(out) / 1024))


if __name__ == "__main__":
    main()
