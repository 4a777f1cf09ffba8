"""The tally path of a step: one GPU's plain launch, and for N > 1 the sum of the per-rank tallies -- the reference's MPI_Reduce
(MC-GPU_v1.3.cu:1019) -- over one of three routes with an agreed fallback chain, its check and its report.

  "copy" (default): the engine's tally exchange -- every projection has an owner rank, the others push their tally into its
     landing buffer with a copy engine while the next projection is tracked, the owner adds them behind its next kernel
     (exchange.cpp; the path the drop-in executable runs between its devices)
  "rccl": sharding.reduce_image, one collective per G projections between two tracking kernels (exposed by design)
  "none": PROJECTION sharding (SURVEY 8e's fallback): rank r simulates all H histories of its own projections; no exchange, no
     collective on the data path.  Not north_star's split (a projection's histories stay on one GPU): the last line of defence
     on a node where neither the exchange nor RCCL works.
Fallback order: copy -> rccl -> none, agreed by all ranks over a gloo control group."""
from __future__ import annotations

import hashlib
import os
import sys
import time
from pathlib import Path

import numpy as np


class TallyRoute:
    def __init__(self, *, pkg, torch, dist, ctl, backend, rank, world, device, ctx, stream, args, H):
        self.pkg, self.eng, self.torch, self.dist, self.ctl, self.backend = pkg, pkg.engine, torch, dist, ctl, backend
        self.rank, self.world, self.device, self.ctx, self.stream, self.args, self.H = rank, world, device, ctx, stream, args, H
        self.nz, self.nx = ctx.detector_shape
        self.nproj, self.seed = ctx.num_projections, ctx.geti("seed")
        self.kind = os.environ.get("BENCH_EXCHANGE", "copy") if dist else None
        if self.kind not in (None, "copy", "rccl", "none"):
            raise SystemExit(f"bench.py: BENCH_EXCHANGE={self.kind}: expected copy, rccl or none")
        if dist and backend == "gloo" and self.kind == "rccl":
            raise SystemExit("bench.py: BENCH_EXCHANGE=rccl needs one GPU per rank")
        self.fallbacks = []  # routes tried and given up, with the reason (reported in config.parallelism_fallbacks)
        self.x = None
        self.policy = self.eng.EXCHANGE_ROTATE if os.environ.get("MCGPU_EXCHANGE_POLICY", "1") != "0" else self.eng.EXCHANGE_ROOT0
        self.kernel_events = []  # (start, stop) HIP events around every timed launch, on the stream it is launched on; read after the region
        self.reduce_bytes = 0
        self.last_reduced = None  # device pointer / tensor of the last complete tally this rank holds
        self.filled = 0
        first_step = self._setup_exchange() if self.kind == "copy" else 0
        self.G = max(1, int(os.environ.get("BENCH_REDUCE_GROUP", "8"))) if self.kind == "rccl" else 1
        self.images = None if self.x else torch.zeros((self.G, 4, self.nz, self.nx), dtype=torch.int64, device="cuda")
        self.narrow = self.kind == "rccl" and os.environ.get("BENCH_REDUCE_U32", "1") == "1"
        self.reduce_algo = os.environ.get("BENCH_REDUCE_ALGO", "scatter")
        self.n_step = first_step if self.x else 0     # exchange step counter (consecutive over the dry run, warm-up, timed region and the check)
        self.collected = first_step if self.x else 0  # exchange steps collected so far (each step is collected exactly once, in order)
        if self.kind == "rccl":
            self._warm_rccl()

    # ------------------------------------------------------------------ set-up
    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def agree(self, ok_here: bool) -> bool:
        """True iff `ok_here` is true on EVERY rank (all ranks get the same answer)."""
        flags = [None] * self.world
        self.dist.all_gather_object(flags, bool(ok_here), group=self.ctl)
        return all(flags)

    def _setup_exchange(self) -> int:
        eng, torch, rank, world = self.eng, self.torch, self.rank, self.world
        shm = Path("/dev/shm") / f"mcgpu_exchange_{os.environ['MASTER_PORT']}"
        shared_map = None
        if rank == 0:
            shared_map = eng.Exchange.open_shared(shm, world, create=True)
        self.barrier()
        if rank != 0:
            shared_map = eng.Exchange.open_shared(shm, world, create=False)
        try:
            x = eng.Exchange(self.device, rank, world, self.ctx.image_words, shared_map, self.policy)
            why = None
        except eng.EngineError as e:
            x, why = None, e
        ok, err = self.pkg.sharding.connect_exchange(x, self.dist, group=self.ctl)  # the same verdict on every rank
        if rank == 0:
            shm.unlink(missing_ok=True)  # every rank holds its mapping
        first_step = 0
        if ok:
            # dry run of the whole protocol on empty tallies, both buffer parities: copy-engine pushes into IPC memory of ANOTHER
            # device, stream waits on interprocess events, the fused add -- everything a real step does except the tracking kernel.
            # A node on which any of that fails between two devices falls back to RCCL with all ranks, here, not in the timed region.
            try:
                for k in (0, 1):
                    x.begin(k, self.stream)
                    x.submit(k, self.stream)
                for k in (0, 1):
                    x.collect(k, self.stream)
                torch.cuda.synchronize()
                dry = None
            except Exception as e:  # noqa: BLE001 -- reported, and agreed on below
                dry = e
            ok, err = self.agree(dry is None), (dry or err)
            first_step = 2
        if not ok:
            # no IPC between these ranks' devices (or the runtime refused an interprocess event): every rank falls back to the
            # RCCL reduction together -- slower (the collective is exposed between kernels), but a measurement instead of a failure
            nxt = "none" if self.backend == "gloo" else "rccl"  # ranks that share a GPU have no RCCL to fall back to
            print(f"bench.py: rank {rank}: the tally exchange is not available here ({why or err}); falling back to BENCH_EXCHANGE={nxt}", file=sys.stderr)
            self.fallbacks.append({"route": "copy", "reason": str(why or err)[:300]})
            if x:
                x.close()
            x = None
            self.kind = nxt
        self.x = x
        return first_step

    def _warm_rccl(self):
        # the reductions of the timed region (full groups of G projections and the remainder group) run once untimed: RCCL
        # sets up its channels and sharding.reduce_image its staging buffers on the first call with a payload shape.  A node
        # on which that fails makes all ranks take projection sharding together (agreed over the gloo control group).
        G, args = self.G, self.args
        try:
            for size in sorted({G if args.steps >= G else 0, args.steps % G, G if args.warmup >= G else 0} - {0}):
                self.pkg.sharding.reduce_image(self.images[:size], dst=0, narrow=self.narrow, algorithm=self.reduce_algo)
            self.torch.cuda.synchronize()
            trial = None
        except Exception as e:  # noqa: BLE001 -- reported, and agreed on below
            trial = e
        if not self.agree(trial is None):
            print(f"bench.py: rank {self.rank}: the RCCL reduction failed here ({trial}); falling back to projection sharding (BENCH_EXCHANGE=none)", file=sys.stderr)
            self.fallbacks.append({"route": "rccl", "reason": str(trial)[:300]})
            self.kind = "none"

    # ------------------------------------------------------------------ one step
    def _reduce_group(self):
        if self.kind == "rccl" and self.filled > 0:
            # on the current stream: the next tracking launch waits for it
            self.reduce_bytes += self.pkg.sharding.reduce_image(self.images[:self.filled], dst=0, narrow=self.narrow, algorithm=self.reduce_algo)
        self.filled = 0

    def _collect_up_to(self, k_excl):
        while self.x and self.collected < k_excl:
            got = self.x.collect(self.collected, self.stream)
            self.last_reduced = got or self.last_reduced
            self.collected += 1

    def step(self, i, timed, hist=None, projection=None):
        """One projection: clear / begin, the tracking launch, and this route's share of the tally sum."""
        torch, ctx, world, rank = self.torch, self.ctx, self.world, self.rank
        hist = self.H if hist is None else hist
        # spread the sampled projections over the arc; projection sharding: step i of rank r is projection number i * world + r
        # of that sequence (sharding.shard_projections), simulated whole by this rank
        k_seq = i * world + rank if self.kind == "none" else i
        p = (k_seq * 149) % self.nproj if projection is None else projection
        first_id = 0 if self.kind == "none" else rank * hist  # history sharding: disjoint history ids per rank
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if timed else None
        if self.x:
            k = self.n_step
            tally = self.x.begin(k, self.stream)
            if ev:
                ev[0].record()
            ctx.launch(p, tally, hist, mode="fast", seed=self.seed, first=first_id, stream=self.stream)
            if ev:
                ev[1].record()
            self.x.submit(k, self.stream)
            self._collect_up_to(k)  # step k - 1, behind this kernel: its pushes had the whole kernel to land
            self.n_step = k + 1
        else:
            image = self.images[self.filled]
            ctx.clear(image.data_ptr(), self.stream)
            if ev:
                ev[0].record()
            ctx.launch(p, image.data_ptr(), hist, mode="fast", seed=self.seed, first=first_id, stream=self.stream)
            if ev:
                ev[1].record()
            self.filled += 1
            self.last_reduced = image
        if ev:
            self.kernel_events.append(ev)  # no host wait inside the timed region: the stream never runs dry between two projections
        if not self.x and self.filled == self.G:
            self._reduce_group()

    def drain(self):
        self._collect_up_to(self.n_step)
        self._reduce_group()

    def kernel_ms(self):
        return [a.elapsed_time(b) for a, b in self.kernel_events]

    # ------------------------------------------------------------------ N > 1: correctness of the sharded sum
    def check(self, kernel_ms_mean):
        """The summed sharded tally against one rank simulating the same history ids alone, bit for bit (projection sharding: every
        rank's own projection against rank 0's run of it)."""
        torch, ctx, dist, rank, world, nproj, seed = self.torch, self.ctx, self.dist, self.rank, self.world, self.nproj, self.seed
        h = int(float(os.environ.get("BENCH_CHECK_HISTORIES", "1e7")))
        p_chk = 447 % nproj
        sharded = digests = None
        if self.x:
            k_chk = self.n_step
            self.step(0, False, hist=h, projection=p_chk)
            self.drain()
            owner = self.x.owner(k_chk)
            if rank == owner:
                sharded = ctx.download_image(self.last_reduced, self.stream)
        elif self.kind == "none":
            # projection sharding: every rank simulates a projection of its own, whole; rank 0 then repeats each of them alone
            # and compares the words (all ranks run the same code on the same inputs: this checks the plumbing, e.g. that no
            # rank's tally leaked into another's)
            owner = 0
            self.filled = 0
            self.step(0, False, hist=h, projection=(p_chk + rank) % nproj)
            torch.cuda.synchronize()
            mine = hashlib.sha256(self.images[0].cpu().numpy().tobytes()).hexdigest()
            digests = [None] * world
            dist.all_gather_object(digests, mine, group=self.ctl)
        else:
            owner = 0
            self.step(0, False, hist=h, projection=p_chk)
            self.drain()
            if rank == 0:
                torch.cuda.synchronize()
                sharded = self.images[0].cpu().numpy().view(np.uint64)
        self.barrier()
        verdict = None
        if self.kind == "none":
            if rank == 0:
                bad = 0
                for r_ in range(world):
                    alone, _, _ = ctx.run_projection((p_chk + r_) % nproj, h, mode="fast", seed=seed, first=0)
                    bad += int(hashlib.sha256(np.ascontiguousarray(alone).view(np.int64).tobytes()).hexdigest() != digests[r_])
                verdict = {"passed": bad == 0, "what": "every rank's own projection equals rank 0's run of that projection (SHA-256 of the tally)",
                           "projections": [(p_chk + r_) % nproj for r_ in range(world)], "histories": h, "ranks": world, "ranks_differing": bad}
        elif rank == owner:  # the others idle: one rank simulates ALL the history ids [0, world * h) of that projection alone
            alone, _, done = ctx.run_projection(p_chk, world * h, mode="fast", seed=seed, first=0)
            verdict = {"passed": bool(np.array_equal(alone, sharded)), "projection": p_chk, "histories_per_rank": h, "ranks": world,
                       "checked_on_rank": rank, "words_differing": int(np.count_nonzero(alone != sharded)),
                       "detected_energy_units": int(alone.sum())}
        verdicts = [None] * world
        dist.all_gather_object(verdicts, verdict, group=self.ctl)
        stats = [None] * world
        dist.all_gather_object(stats, (self.x.stats() if self.x else None, float(kernel_ms_mean)), group=self.ctl)
        return {"sharded_equals_single": verdicts[owner], "stats": stats}

    # ------------------------------------------------------------------ the first contact of two GPUs: both collectives, one invocation
    def collectives_comparison(self, steps=6):
        """After the timed region, whatever route it took: `steps` projections on each route that can run here, timed the same way
        (barrier + synchronize on both sides, max over ranks), so that ONE `--gpus N` command tells the exchange from north_star's
        literal collective -- a per-projection sum-reduction of the uint64 detector tally to rank 0 (`ncclReduce`, the reference's
        MPI_Reduce at MC-GPU_v1.3.cu:1019) -- plus who can reach whom (hipDeviceCanAccessPeer) and which device each rank drives.
        Ranks that share a GPU (development) have no RCCL between them: that leg is reported as unavailable, the topology is not."""
        torch, dist, ctx, world, rank = self.torch, self.dist, self.ctx, self.world, self.rank
        n_dev = torch.cuda.device_count()
        devices = [None] * world
        dist.all_gather_object(devices, {"rank": rank, "device": int(self.device), "name": torch.cuda.get_device_name(self.device)}, group=self.ctl)
        peer = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(n_dev)] for i in range(n_dev)]
        out = {"devices_of_the_ranks": devices, "peer_access_matrix": peer, "steps_per_leg": steps, "route_of_the_timed_region": self.kind, "ms_per_step": {}}

        def timed(step_fn, drain_fn):
            step_fn(0)  # untimed: the route's first call sets up channels / staging buffers
            drain_fn()
            self.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                step_fn(1 + i)
            drain_fn()
            self.barrier()
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item()) / steps * 1e3

        out["sharded_equals_single"] = {}
        h_chk, p_chk = int(float(os.environ.get("BENCH_CHECK_HISTORIES", "1e7"))), 447 % self.nproj

        def verify(owner, sharded):
            """On `owner`: the route's summed tally of h_chk histories per rank against ONE rank simulating all their ids alone."""
            verdict = None
            if rank == owner:
                alone, _, _ = ctx.run_projection(p_chk, world * h_chk, mode="fast", seed=self.seed, first=0)
                verdict = {"passed": bool(np.array_equal(alone, sharded)), "projection": p_chk, "histories_per_rank": h_chk, "ranks": world,
                           "checked_on_rank": rank, "words_differing": int(np.count_nonzero(alone != sharded))}
            verdicts = [None] * world
            dist.all_gather_object(verdicts, verdict, group=self.ctl)
            return verdicts[owner]

        if self.x:  # the exchange is up: time it like the main region (it may be the route that just ran)
            out["ms_per_step"]["copy"] = timed(lambda i: self.step(i, False), self.drain)
            k_chk = self.n_step
            self.step(0, False, hist=h_chk, projection=p_chk)
            self.drain()
            owner = self.x.owner(k_chk)
            out["sharded_equals_single"]["copy"] = verify(owner, ctx.download_image(self.last_reduced, self.stream) if rank == owner else None)
        else:
            out["ms_per_step"]["copy"] = None
            out["sharded_equals_single"]["copy"] = None
        if self.backend == "nccl":
            image = torch.zeros((4, self.nz, self.nx), dtype=torch.int64, device="cuda")
            first_id = rank * self.H

            def reduce_step(i):
                ctx.clear(image.data_ptr(), self.stream)
                ctx.launch((i * 149) % self.nproj, image.data_ptr(), self.H, mode="fast", seed=self.seed, first=first_id, stream=self.stream)
                dist.reduce(image, dst=0, op=dist.ReduceOp.SUM)  # ncclReduce(int64, sum, root 0) on the current stream, one per projection

            try:
                ms = timed(reduce_step, lambda: None)
                err = None
            except Exception as e:  # noqa: BLE001 -- reported, and agreed on below
                ms, err = None, e
            if self.agree(err is None):
                out["ms_per_step"]["rccl_reduce_per_projection"] = ms
                ctx.clear(image.data_ptr(), self.stream)
                ctx.launch(p_chk, image.data_ptr(), h_chk, mode="fast", seed=self.seed, first=rank * h_chk, stream=self.stream)
                dist.reduce(image, dst=0, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize()
                out["sharded_equals_single"]["rccl_reduce_per_projection"] = verify(0, image.cpu().numpy().view(np.uint64) if rank == 0 else None)
            else:
                out["ms_per_step"]["rccl_reduce_per_projection"] = None
                out["rccl_reduce_error"] = str(err)[:300] if err else "failed on another rank"
        else:
            out["ms_per_step"]["rccl_reduce_per_projection"] = None
            out["sharded_equals_single"]["rccl_reduce_per_projection"] = None
            out["rccl_reduce_unavailable"] = "the ranks share GPUs (gloo process group): RCCL refuses two ranks on one device"
        return out

    # ------------------------------------------------------------------ report
    def parallelism_text(self):
        world, eng = self.world, self.eng
        if self.kind == "none":
            return (f"PROJECTION-sharded x{world}: every rank simulates whole projections, no exchange and no collective (SURVEY 8e fallback mode; "
                    "NOT north_star's history split)")
        text = f"history-sharded x{world}"
        if not self.dist:
            return text
        if self.x:
            return (text + ", tally exchange: copy-engine pushes to the projection's owner (" +
                    ("owner = projection mod ranks" if self.policy == eng.EXCHANGE_ROTATE else "owner = rank 0") + "), one fused add per projection")
        return text + f", one RCCL sum-reduction ({self.reduce_algo}) of the detector tallies per {self.G} projections"

    def report(self, multi, elapsed, steps):
        """The `reduce` object of the JSON line."""
        world, eng = self.world, self.eng
        k_all = [s_[1] for s_ in multi["stats"]]
        red = {"kind": self.kind, "kernel_ms_avg_per_rank": k_all, "step_minus_slowest_kernel_ms": elapsed / steps * 1e3 - max(k_all)}
        if self.x:
            st = [s_[0] for s_ in multi["stats"]]
            push = [a["last_push_ms"] for a in st if a["pushes"] > 0]
            add = [a["last_add_ms"] for a in st if a["collects"] > 0 and a["last_add_ms"] > 0]
            bytes_push = st[0]["bytes_per_push"]
            gbps = bytes_push / (float(np.max(push)) * 1e-3) / 1e9 if push else None
            red.update({"bytes_per_push": bytes_push, "pushes_per_projection": world - 1,
                        # HIP events around the copy on the copy stream.  Without a profiler attached they bracket the
                        # SUBMISSION of a copy-engine transfer, not its duration, on some runs (a 45 MB push cannot take less than
                        # 0.7 ms at the engine's 60 GB/s): such a reading is flagged instead of being turned into a bandwidth;
                        # the profiler's figure is in profiles/r03i_exchange_overlap_rocprofv3_memory_copy_stats.txt
                        "push_ms_by_events": float(np.max(push)) if push else None,
                        "push_GBps": gbps if (gbps is not None and gbps < 100.0) else None,
                        "push_events_bracket_submission_only": bool(gbps is not None and gbps >= 100.0),
                        "fused_add_ms": float(np.max(add)) if add else None,
                        # what a tracking stream sees of the exchange per projection it OWNS: the fused add (+ a 45 MB memset per
                        # step on every rank, inside begin(), which the N = 1 step pays as well)
                        "exposed_ms_per_step_on_the_critical_rank": (float(np.max(add)) if add else 0.0) * (1.0 / world if self.policy == eng.EXCHANGE_ROTATE else 1.0),
                        "host_wait_s_per_rank": [a["host_wait_s"] for a in st],
                        "owner_policy": "rotate" if self.policy == eng.EXCHANGE_ROTATE else "rank0"})
        elif self.kind == "rccl":
            red.update({"bytes_per_rank_in_timed_region": self.reduce_bytes, "narrowed_to_u32_when_it_fits": bool(self.narrow), "algorithm": self.reduce_algo,
                        "projections_per_reduction": self.G})
        return red

    def close(self):
        if self.x:
            self.torch.cuda.synchronize()
            self.barrier()  # nobody unmaps landing memory a peer may still address
            self.x.close()
