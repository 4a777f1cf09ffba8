"""One rank of the cross-process tally-exchange test (tests/test_exchange.py starts two of these on one GPU).
usage: exchange_rank.py <input.in> <rank> <world> <policy> <steps> <histories> <shared file> <scratch dir>"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import cases  # noqa: E402

eng = cases.pkg.engine
inp, rank, world, policy, steps, hist, shm, scratch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7], Path(sys.argv[8])


def wait_for(path, seconds=120.0):
    t0 = time.time()
    while not path.exists():
        if time.time() - t0 > seconds:
            raise SystemExit(f"rank {rank}: {path.name} never appeared")
        time.sleep(0.01)


with eng.create(inp, device=0) as ctx:
    shared = eng.Exchange.open_shared(shm, world, create=False)  # the test created and zeroed the file
    x = eng.Exchange(0, rank, world, ctx.image_words, shared, policy)
    tmp = scratch / f"card_{rank}.tmp"
    tmp.write_bytes(x.card())
    tmp.rename(scratch / f"card_{rank}.bin")
    for peer in range(world):
        if peer != rank:
            wait_for(scratch / f"card_{peer}.bin")
            x.connect(peer, (scratch / f"card_{peer}.bin").read_bytes())
    (scratch / f"connected_{rank}").write_text("1")
    for peer in range(world):
        wait_for(scratch / f"connected_{peer}")
    x.probe()  # every peer has mapped every peer: the copy engine reaches their landing buffers
    nproj, seed = ctx.num_projections, ctx.geti("seed")
    for k in range(steps + 1):
        if k < steps:
            tally = x.begin(k)
            ctx.launch(k % nproj, tally, hist, mode="fast", seed=seed, first=rank * hist)
            x.submit(k)
        if k > 0:
            got = x.collect(k - 1)
            if got:
                np.save(scratch / f"reduced_{k - 1}.npy", ctx.download_image(got))
    st = x.stats()
    (scratch / f"done_{rank}").write_text(repr(st))
    for peer in range(world):  # nobody unmaps landing memory a peer may still address
        wait_for(scratch / f"done_{peer}")
    x.close()
print(f"rank {rank}: {st}")
