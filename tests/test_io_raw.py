"""Host logic of the raw-data path (no GPU): pair-file parsers, the reference's filtering rules, the
compact code form of the node features, and the integrity of the committed real-DrugBank fixture."""

import os

import numpy as np
import pytest
import torch

import conftest  # noqa: F401
from blgnn import io_raw
from blgnn.dataset import one_hot_from_codes, FEATURE_CARDS

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "drugbank_real.npz")


def test_tsv_parser_and_filter(tmp_path):
    p = tmp_path / "ddi.tsv"
    p.write_text("DB00001\tDB00002\nDB00002\tDB00001\nDB00003\tDB00001\nDB00009\tDB00001\n\n")
    raw = io_raw.load_ddi_pairs_tsv(str(p))
    assert raw.tolist() == [[1, 2], [2, 1], [3, 1], [9, 1]]
    # drug 3 has a structure without bonds, drug 9 has no structure at all, drug 7 is in no pair
    used, pairs, labels = io_raw._filter([1, 2, 3, 7], [4, 2, 0, 6], raw, np.ones(4, dtype=np.int64))
    assert used == [1, 2] and pairs.tolist() == [[1, 2]] and labels.tolist() == [1]


def test_drugcombo_vote_follows_the_reference_including_its_unsorted_lookup(tmp_path):
    p = tmp_path / "vote.csv"
    rows = ["id,drug1,drug2,x,label",
            "0,B,A,0,antagonism", "1,B,A,0,antagonism", "2,A,B,0,synergy",     # sorted key (A,B): syn 1, ant 2
            "3,C,D,0,antagonism"]
    p.write_text("\n".join(rows) + "\n")
    pairs, labels = io_raw.load_drugcombo_pairs_csv(str(p), {"a": 10, "b": 20, "c": 30, "d": 40})
    # lines with drugs written (B, A): the antagonism count is looked up under the UNSORTED key (B, A) -> 0,
    # so synergy (1) >= 0 wins; the (A, B) line sees ant = 2 > syn = 1
    assert pairs.tolist() == [[20, 10], [20, 10], [10, 20], [30, 40]]
    assert labels.tolist() == [1, 1, 2, 2]
    pairs, _ = io_raw.load_drugcombo_pairs_csv(str(p), {"a": 10, "b": 20})       # unknown names are skipped
    assert len(pairs) == 3


def test_codes_round_trip():
    rng = np.random.default_rng(0)
    codes = np.stack([rng.integers(0, c, size=200) for c in FEATURE_CARDS], axis=1)
    x = one_hot_from_codes(codes, device="cpu")
    assert x.shape == (200, 52) and float(x.sum()) == 200 * 6
    back = io_raw.codes_from_one_hot(x.numpy(), FEATURE_CARDS)
    assert back.dtype == np.int8 and (back == codes).all()
    bad = x.numpy().copy()
    bad[0, 0:32] = 0
    with pytest.raises(ValueError):
        io_raw.codes_from_one_hot(bad, FEATURE_CARDS)


def test_real_drugbank_fixture_is_consistent():
    z = np.load(GOLD)
    gids, node_ptr, edge_ptr = z["gids"], z["node_ptr"], z["edge_ptr"]
    codes, cards, el, pairs = z["codes"], z["cards"], z["edge_local"].astype(np.int64), z["pairs"]
    G = len(gids)
    assert G == 1309 and (np.diff(gids) > 0).all()                       # graphs ordered by gid
    assert node_ptr[0] == 0 and node_ptr[-1] == codes.shape[0] == 36816 and len(node_ptr) == G + 1
    assert codes.shape[1] == 6 and int(cards.sum()) == 52
    assert (codes >= 0).all() and (codes < cards.reshape(1, -1)).all()
    assert edge_ptr[-1] == el.shape[1] == 78524 and (np.diff(edge_ptr) > 0).all()      # every molecule has a bond
    sizes = np.diff(node_ptr)
    mol = np.repeat(np.arange(G), np.diff(edge_ptr))
    assert (el >= 0).all() and (el[0] < sizes[mol]).all() and (el[1] < sizes[mol]).all()
    key = (mol * 512 + el[0]) * 512 + el[1]
    assert (np.diff(key) > 0).all()                                       # sorted by (molecule, row, col), no duplicates
    rev = np.sort((mol * 512 + el[1]) * 512 + el[0])
    assert (rev == key).all()                                             # symmetric
    assert pairs.shape == (41072, 2) and (pairs[:, 0] < pairs[:, 1]).all()
    assert np.isin(pairs, gids).all() and len(np.unique(pairs[:, 0].astype(np.int64) * 100000 + pairs[:, 1])) == len(pairs)
    assert int(z["raw_lines"]) == 48514
