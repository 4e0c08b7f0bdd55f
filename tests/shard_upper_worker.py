"""Worker of tests/test_gpu_model.py::test_engine_target_row_sharded_interaction_level_two_ranks: one of two ranks that
share the GPU over gloo (RCCL refuses two ranks on one device; the engine's sharding and collective code is the same) --
and of ::test_engine_collectives_over_rccl_one_rank: ONE rank whose all-gathers are real RCCL calls (BLGNN_WORKER_BACKEND=nccl).
Each rank runs the same pair batches through the engine with the interaction level replicated and target-row sharded
(``BiLevelEngine(shard_upper=...)``) and compares its scores; rank 0 also compares with a one-rank engine over all drugs."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import conftest  # noqa: F401  (puts the package on sys.path)
import blgnn
from blgnn.dist import ShardedLower
from blgnn.engine import BiLevelEngine
from blgnn.forward import all_molecule_pairs
from helpers import build


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("BLGNN_WORKER_BACKEND", "gloo")
    device = "cuda:0"
    torch.cuda.set_device(0)
    if backend == "nccl":
        # ONE rank over RCCL: the engine's collectives run as RCCL calls (``collective=True``) between its captured graphs
        assert world == 1
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(device))
            probe = torch.ones(4, device=device)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
        except Exception as e:                       # RCCL cannot come up on this box: the caller skips (exit code 77)
            print("RCCL_INIT_FAILED: %r" % (e,), flush=True)
            sys.exit(77)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = blgnn.make_synthetic(230, 2600, seed=31, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin", higher_level_gnn_type="gat",
                             D_lower=128, D_higher=128, device=device, fuse_readout=False)
    model, data = build(flags, ds, seed=13)
    g = ds.interaction_combo_nxgraph
    g.init_x = torch.zeros((ds.num_graphs(), 5 * 128), device=device)
    ds.init = True
    rng = np.random.default_rng(3)
    batches = [ds.pair_gids[rng.choice(len(ds.pair_gids), size=40, replace=False)] for _ in range(3)]
    uppers = [blgnn.BatchData(b, ds, is_train=False) for b in batches]

    sharded = ShardedLower(ds.molecules.gids, ds.molecules.sizes, rank, world)
    lower = blgnn.BatchData(all_molecule_pairs(sharded.local_gids), ds, is_train=False, ignore_pairs=True)
    outs = {}
    for shard in (False, True):
        for use_graph in (False, True):
            with BiLevelEngine(model, lower, uppers, g, sharded=sharded, world_size=world, use_graph=use_graph,
                               shard_upper=shard, exchange_transformed=True, collective=True) as eng:
                assert eng.shard_upper == shard and eng.exchange_transformed
                assert eng.n_upper_segments == (3 if shard else 1)
                eng.capture()
                res = []
                for i in range(5):
                    out = eng.step(i)
                    eng.synchronize()
                    res.append(out.clone())
                outs[(shard, use_graph)] = res
            dist.barrier()
    ref = outs[(False, False)]
    worst = 0.0
    for key, res in outs.items():
        for a, b in zip(res, ref):
            assert torch.isfinite(a).all()
            worst = max(worst, float((a - b).abs().max()))
    # one-rank engine over all drugs: the unsharded truth
    g.init_x = torch.zeros((ds.num_graphs(), 5 * 128), device=device)
    full = blgnn.BatchData(all_molecule_pairs(ds.molecules.gids), ds, is_train=False, ignore_pairs=True)
    plain = BiLevelEngine(model, full, uppers, g, use_graph=False)
    vs_plain = 0.0
    for i in range(3):
        vs_plain = max(vs_plain, float((plain.eager_step(i) - outs[(True, True)][i]).abs().max()))
    print("rank %d (%s, world %d): sharded vs replicated max |diff| %.3e, vs one-rank %.3e"
          % (rank, backend, world, worst, vs_plain), flush=True)
    assert worst <= 2e-6 and vs_plain <= 2e-6, (worst, vs_plain)
    if world == 1:
        assert worst == 0.0 and vs_plain == 0.0          # one rank: the same kernels on the same rows
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
