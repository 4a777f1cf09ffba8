"""CPU test of the N>1 path: world_size-2 `gloo` processes shard the histories of one projection with
`sharding.shard_range`, tally their share (the CPU oracle stands in for the kernel here -- test only),
and sum-reduce the int64 images onto rank 0 exactly as bench.py does with RCCL.  The reduced image must
equal the single-process image bit for bit."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, input_path, nbatch, hpt, out_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases
    import oracle_lib as ol
    import parity
    eng = cases.pkg.engine
    with eng.create(input_path, device=-1) as ctx:
        T = parity.tables_from_context(ctx)
        first, count = cases.pkg.sharding.shard_range(nbatch, rank, world)
        img, _ = T.track(0, 42, first, count, hpt, ol.MATH_PORTABLE)
        t = torch.from_numpy(img.view(np.int64).copy())
        R = cases.pkg.sharding.reduce_image
        results = {}
        for algo in ("scatter", "reduce"):
            # 64-bit words on the wire
            x = t.clone()
            sent = R(x, dst=0, narrow=False, algorithm=algo)
            assert sent >= t.numel() * 8
            results[(algo, "wide")] = x
            # narrowed payload (what bench.py sends when the sums fit 32 bits) ...
            x = t.clone()
            assert R(x, dst=0, narrow=True, algorithm=algo) <= t.numel() * 4 * (1 + 1 / world) + 64
            results[(algo, "narrow")] = x
            # ... and the agreed fallbacks: rank 1 alone holds a word >= 2^31 (sums no longer fit 32 bits) / >= 2^32 (not
            # even its own words do): both ranks must still take the same branch
            for big_word in (2 ** 30 + 5, 2 ** 31 + 5, 2 ** 33 + 7):
                x = t.clone()
                x[0] += big_word * rank
                R(x, dst=0, narrow=True, algorithm=algo)
                x[0] -= big_word
                results[(algo, big_word)] = x
        R(t, dst=0)
        if rank == 0:
            for key, x in results.items():
                assert torch.equal(x, t), key
        hist = torch.tensor([count * hpt], dtype=torch.int64)
        dist.reduce(hist, dst=0, op=dist.ReduceOp.SUM)  # exact global history count for the normalisation (MC-GPU_v1.3.cu:878)
        if rank == 0:
            np.savez(out_path, image=t.numpy().view(np.uint64), histories=hist.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    sys.path.insert(0, str(ROOT / "tests"))
    import cases
    sr = cases.pkg.sharding.shard_range
    for units in (0, 1, 7, 100, 666_752, 10**8 + 3):
        for world in (1, 2, 3, 4, 8):
            parts = [sr(units, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == units
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        sr(10, 2, 2)


def test_two_rank_history_sharding_reduces_to_single_process_image(case_dir, tmp_path):
    import cases
    import oracle_lib as ol
    import parity
    input_path = str(case_dir("catphan64"))
    nbatch, hpt = 61, 40
    out = tmp_path / "reduced.npz"
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, input_path, nbatch, hpt, str(out)), nprocs=2, join=True)
    got = np.load(out)
    with cases.pkg.engine.create(input_path, device=-1) as ctx:
        T = parity.tables_from_context(ctx)
        want, _ = T.track(0, 42, 0, nbatch, hpt, ol.MATH_PORTABLE)
    assert int(got["histories"][0]) == nbatch * hpt
    assert np.array_equal(got["image"], want)
