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


def test_shard_projections_partitions_exactly():
    """Projection sharding (SURVEY 8e's fallback, `BENCH_EXCHANGE=none` / `--shard projections`): the union of the ranks'
    projection sets is the trajectory, exactly once each, balanced to within one projection."""
    sys.path.insert(0, str(ROOT / "tests"))
    import cases
    sp = cases.pkg.sharding.shard_projections
    for n in (0, 1, 7, 894, 8940):
        for world in (1, 2, 3, 4, 8):
            parts = [list(sp(n, r, world)) for r in range(world)]
            assert sorted(p for part in parts for p in part) == list(range(n))
            assert max(len(part) for part in parts) - min(len(part) for part in parts) <= 1
    with pytest.raises(ValueError):
        sp(10, 2, 2)


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


class _FakeExchange:
    """Stands in for engine.Exchange where there is no GPU: records what the handshake hands it."""

    def __init__(self, rank, world, probe_fails=False):
        self.rank, self.world, self.connected, self.probed, self.probe_fails = rank, world, {}, 0, probe_fails

    def card(self):
        return bytes([self.rank]) * 64 * (3 + 2 * self.world)

    def connect(self, peer, card):
        assert peer != self.rank and peer not in self.connected
        self.connected[peer] = card

    def probe(self):
        assert len(self.connected) == self.world - 1  # only after every peer is mapped
        self.probed += 1
        if self.probe_fails:
            raise RuntimeError("the copy engine cannot reach a peer")


def _handshake_worker(rank, world, port, shm_path, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases
    eng = cases.pkg.engine
    # the host region of the exchange: rank 0 creates and zeroes it, the others map it after the barrier (bench.py's order)
    if rank == 0:
        m = eng.Exchange.open_shared(shm_path, world, create=True)
    dist.barrier()
    if rank != 0:
        m = eng.Exchange.open_shared(shm_path, world, create=False)
    assert len(m) == eng.Exchange.shared_bytes(world) and bytes(m[:]) == b"\0" * len(m)
    fx = _FakeExchange(rank, world)
    ok, err = cases.pkg.sharding.connect_exchange(fx, dist)
    assert ok and err is None and fx.probed == 1
    assert sorted(fx.connected) == [r for r in range(world) if r != rank]
    # the copy-engine probe fails on ONE rank: both ranks get the same verdict and take the same fallback
    ok, err = cases.pkg.sharding.connect_exchange(_FakeExchange(rank, world, probe_fails=(rank == 0)), dist)
    assert not ok and (err is not None) == (rank == 0)
    # a rank without an exchange end (its device refused IPC, say): EVERY rank learns it, nobody is left in a collective
    ok, err = cases.pkg.sharding.connect_exchange(None if rank == 1 else _FakeExchange(rank, world), dist)
    assert not ok and err is not None
    assert all(card == bytes([peer]) * 64 * (3 + 2 * world) for peer, card in fx.connected.items())
    # one mapping for everybody: a counter written by one rank is seen by the other
    m[64 + 8 * rank] = 7 + rank
    dist.barrier()
    assert [m[64 + 8 * r] for r in range(world)] == [7 + r for r in range(world)]
    (Path(out_dir) / f"ok_{rank}").write_text("1")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_handshake_over_gloo(tmp_path):
    """The host side of the N > 1 tally exchange that needs no GPU: the mapped counter region (one /dev/shm file, created by
    rank 0) and the all-gather of the address cards + connect-to-every-peer (sharding.connect_exchange).  The device side
    (copy-engine pushes, interprocess events, fused add) is covered on the GPU box by tests/test_exchange.py."""
    world = 2
    shm = f"/dev/shm/mcgpu_exchange_cputest_{os.getpid()}"
    port = 31500 + (os.getpid() % 2000)
    try:
        mp.spawn(_handshake_worker, args=(world, port, shm, str(tmp_path)), nprocs=world, join=True)
    finally:
        Path(shm).unlink(missing_ok=True)
    assert all((tmp_path / f"ok_{r}").exists() for r in range(world))


def test_bench_refuses_a_rank_count_it_cannot_honour():
    """`bench.py --gpus 2` starts its two rank processes itself (no launcher in the environment) and exits NON-ZERO with a
    clear message when they cannot run (this container has no GPU); under a launcher that started another number of ranks
    than --gpus says it refuses to print a mislabelled line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "rank 0 exited" in r.stderr and "rank 1 exited" in r.stderr
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4"], env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == "" and "refusing to run a mislabelled measurement" in r.stderr


def test_bench_adopts_the_launchers_world_size_when_gpus_is_not_given():
    """`torchrun ... bench.py` WITHOUT --gpus: the launcher's WORLD_SIZE is the GPU count; the run then stops where
    every GPU-less run stops -- at "needs a GPU" -- and not at the mislabelled-measurement refusal."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0"],
                       env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "mislabelled" not in r.stderr and "needs a GPU" in r.stderr, r.stderr[-500:]
