"""The tally exchange between the GPUs of one node (4d-cbct-mc_amd/csrc/exchange.cpp; the reference's MPI_Reduce of the
detector images, MC-GPU_v1.3.cu:1006-1024): copy-engine pushes into the owner's landing buffer, one fused add.  History ids
own their RNG streams and tallies are integers, so the summed sharded tally must equal one rank simulating the same ids alone
BIT FOR BIT -- between contexts of one process (what mcgpu_run_scan_multi runs) and between processes over IPC handles and
interprocess events (what bench.py's ranks run; here two processes on the one GPU of the box)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import cases

HERE = Path(__file__).resolve().parent


def test_exchange_sizes_and_argument_checks(engine):
    lib = engine.load_library()
    assert lib.mcgpu_exchange_card_bytes(8) == 64 * (1 + 2 + 16)
    assert engine.Exchange.shared_bytes(8) > engine.Exchange.shared_bytes(2) > 0
    with pytest.raises(engine.EngineError):  # no device in this process / bad world: an error return, not a crash
        engine.Exchange(0, 5, 2, 1024, bytearray(engine.Exchange.shared_bytes(2)))


@pytest.mark.gpu
@pytest.mark.parametrize("world,policy,hist", [(3, "rotate", 150_000), (2, "rank0", 150_000), (1, "rotate", 150_000), (3, "rotate", 1500), (2, "rank0", 1500),
                                               # BASELINE configs 3 and 5 say 8 GPUs: eight ranks on the one device of the box (seven pushes per
                                               # projection, the owner rotating over all eight; with 1500 histories the adds wait on their events)
                                               (8, "rotate", 150_000), (8, "rotate", 1500), (8, "rank0", 1500)])
def test_exchange_between_contexts_of_one_process(engine, case_dir, world, policy, hist):
    """hist = 1500: kernels of a few microseconds against pushes of 0.75 ms (full-size detector), so the owner's fused add is
    enqueued while the push it needs is still in flight: the events, not luck, must order them."""
    # the stress cases use the reference's full detector: 45 MB tallies, pushes of 0.75 ms against kernels of a few microseconds
    inp = case_dir("catphan64_ct") if hist > 10_000 else case_dir("catphan64_ct", n_detector_pixels=(1848, 768), detector_size=(717.024, 297.984))
    pol = (engine.EXCHANGE_ROTATE if policy == "rotate" else engine.EXCHANGE_ROOT0) | engine.EXCHANGE_LOCAL
    steps = 9 if hist > 10_000 else 24
    shared = bytearray(engine.Exchange.shared_bytes(world))
    ctxs = [engine.create(inp, device=0) for _ in range(world)]
    xs = []
    try:
        xs = [engine.Exchange(0, r, world, ctxs[0].image_words, shared, pol) for r in range(world)]
        for a in xs:
            for b in xs:
                if a is not b:
                    a.connect_local(b)
        for a in xs:
            a.probe()
        nproj, seed = ctxs[0].num_projections, ctxs[0].geti("seed")
        reduced = {}
        for k in range(steps + 1):
            if k < steps:
                for r in range(world):
                    tally = xs[r].begin(k)
                    ctxs[r].launch(k % nproj, tally, hist, mode="fast", seed=seed, first=r * hist)
                    xs[r].submit(k)
            if k > 0:
                owner = xs[0].owner(k - 1)
                assert owner == ((k - 1) % world if policy == "rotate" else 0)
                for r in range(world):
                    got = xs[r].collect(k - 1)
                    assert bool(got) == (r == owner)
                    if got:
                        reduced[k - 1] = ctxs[r].download_image(got)
        for k in range(steps):
            alone, _, _ = ctxs[0].run_projection(k % nproj, world * hist, mode="fast", seed=seed, first=0)
            assert np.array_equal(reduced[k], alone) and alone.sum() > 0, k
        with pytest.raises(engine.EngineError):  # a step is collected once
            xs[0].collect(steps - 1)
        with pytest.raises(engine.EngineError):  # the probe belongs before the first step
            xs[0].probe()
        if world > 1:
            st = [x.stats() for x in xs]
            assert sum(s["pushes"] for s in st) == steps * (world - 1) and sum(s["collects"] for s in st) == steps
    finally:
        for x in xs:
            x.close()
        for c in ctxs:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("policy,steps,hist", [(1, 7, 200_000), (0, 7, 200_000), (1, 30, 1000)])
def test_exchange_between_two_processes_on_one_gpu(engine, case_dir, tmp_path, policy, steps, hist):
    """IPC memory handles + interprocess events: two rank processes (fresh interpreters) share the box's GPU.  The last case
    runs 30 steps of tiny kernels on the full-size detector: the 45 MB pushes take much longer than the kernels, so every add
    really waits on its interprocess event."""
    inp = case_dir("catphan64_ct") if hist > 10_000 else case_dir("catphan64_ct", n_detector_pixels=(1848, 768), detector_size=(717.024, 297.984))
    world = 2
    shm = Path("/dev/shm") / f"mcgpu_exchange_test_{os.getpid()}_{policy}"
    engine.Exchange.open_shared(shm, world, create=True).close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    try:
        procs = [subprocess.Popen([sys.executable, str(HERE / "exchange_rank.py"), str(inp), str(r), str(world), str(policy), str(steps), str(hist), str(shm),
                                   str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
        outs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=300)
            except subprocess.TimeoutExpired:
                p.kill()
                out, _ = p.communicate()
            outs.append(out.decode(errors="replace"))
        assert all(p.returncode == 0 for p in procs), outs
    finally:
        shm.unlink(missing_ok=True)
    with engine.create(inp, device=0) as ctx:
        nproj, seed = ctx.num_projections, ctx.geti("seed")
        for k in range(steps):
            alone, _, _ = ctx.run_projection(k % nproj, world * hist, mode="fast", seed=seed, first=0)
            got = np.load(tmp_path / f"reduced_{k}.npy")
            assert np.array_equal(got, alone) and alone.sum() > 0, k


@pytest.mark.gpu
def test_exchange_gives_up_on_a_silent_peer_instead_of_hanging(engine, case_dir, tmp_path):
    """An owner whose peer never pushes gets an error return after MCGPU_EXCHANGE_TIMEOUT_S (default 120 s), and so does a
    rank whose peer has been destroyed -- the host side of the protocol never blocks forever, and nothing is left waiting on
    the device."""
    code = f"""
import sys, time
sys.path.insert(0, {str(HERE)!r})
import cases
eng = cases.pkg.engine
ctx = eng.create({str(case_dir("catphan64_ct"))!r}, device=0)
shared = bytearray(eng.Exchange.shared_bytes(2))
xs = [eng.Exchange(0, r, 2, ctx.image_words, shared, eng.EXCHANGE_ROOT0 | eng.EXCHANGE_LOCAL) for r in range(2)]
xs[0].connect_local(xs[1]); xs[1].connect_local(xs[0])
t = xs[0].begin(0); ctx.launch(0, t, 100000, mode="fast", seed=1); xs[0].submit(0)
t0 = time.time()
try:
    xs[0].collect(0)          # rank 1 never submitted step 0
    print("NO ERROR")
except eng.EngineError as e:
    print("ERROR after %.1f s: %s" % (time.time() - t0, e.message))
xs[1].close()                 # ... and a peer that has gone is noticed at once
t = xs[0].begin(1); ctx.launch(0, t, 100000, mode="fast", seed=1); xs[0].submit(1)
t0 = time.time()
try:
    xs[0].collect(1)
    print("NO ERROR")
except eng.EngineError as e:
    print("ERROR after %.1f s: %s" % (time.time() - t0, e.message))
xs[0].close(); ctx.close()
print("CLEAN EXIT")
"""
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MCGPU_EXCHANGE_TIMEOUT_S="2"), capture_output=True, text=True, timeout=120)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("ERROR", "NO ERROR", "CLEAN"))]
    assert r.returncode == 0 and len(lines) == 3 and lines[2] == "CLEAN EXIT", (r.stdout, r.stderr)
    assert lines[0].startswith("ERROR after 2.") and "gave up waiting for the push of rank 1" in lines[0]
    assert lines[1].startswith("ERROR after 0.") and "rank 1 left" in lines[1]
