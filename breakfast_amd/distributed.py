"""One-process-per-GPU driver for the row-sharded path.

Every rank holds the full CSR (<= 180 MB at 1M rows) and runs the prep kernels on it; the (k,f,g) CELLS of
the sorted order are dealt round-robin over the ranks (owner = index of the cell's first tile mod world —
cells, not tiles: the order of the rows inside a cell comes from atomics and differs from rank to rank, whole
cells are the same sets everywhere), so each rank evaluates ~1/world of the pair tiles and hooks the edges it
finds into a LOCAL union-find forest over all N rows.  (max_dist 1, the variant join: every rank builds the whole
hash table and the blocks of 8192 tokens — their table lookups — are dealt round-robin.)  The only
exchange step is the label merge, over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests):

  merge="allgather" (default): ONE all_gather of the int32[N] local labels (4 MB per rank at 1M rows —
      latency-bound on xGMI, not bandwidth-bound), then every rank unites the world*N pseudo-edges
      (i, L_g[i]) into its forest and re-flattens.  Exact after one round.
  merge="allreduce": the north-star form — all_reduce(MIN) of the labels, unite (i, L[i]), repeat until
      no rank changed (all_reduce(MAX) of a flag).  Same result, >= 2 rounds.

Labels are canonical (smallest row index of the component), so the result is bit-identical to the
1-GPU labels on every rank.

The compute engine is injected: `GpuEngine` drives libbfk through the resident-context C-ABI; the CPU
tests (tests/test_distributed_cpu.py) plug in an oracle-backed engine to exercise the sharding and the
merge protocol under gloo.  There is no CPU engine in the product.
"""

from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib


class GpuEngine:
    """libbfk resident context on one GPU; tensors are torch CUDA tensors.  The engine owns ONE torch stream: the kernels of
    libbfk and — through ShardedClusterer.step, which runs under it — the collectives between them are enqueued there, so the
    label exchange is ordered behind the shard's kernels and the merge behind the exchange by the stream itself.  (Handing
    libbfk the handle of torch's DEFAULT stream — 0 — means "the context's own stream" to bfk_ctx_set_stream: the kernels
    then ran on a stream the collectives did not wait for.  Found by the round-3 rehearsal's per-phase events.)"""

    def __init__(self, device_index: int, stream=None):
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.stream = stream if stream is not None else torch.cuda.Stream(self.device)
        self.ctx = _lib.Context(device_index)
        self.ctx.set_stream(self.stream.cuda_stream)
        self.n = 0
        self._keep = None

    def bind(self, indptr, indices):
        """indptr/indices: int32 numpy arrays or CUDA tensors; kept resident in HBM."""
        ip = torch.as_tensor(indptr, dtype=torch.int32).to(self.device).contiguous()
        ix = torch.as_tensor(indices, dtype=torch.int32).to(self.device).contiguous()
        if ix.numel() == 0:
            ix = torch.zeros(1, dtype=torch.int32, device=self.device)
        torch.cuda.synchronize(self.device)  # (the copies above ran on the caller's stream)
        self._keep = (ip, ix)
        self.n = ip.numel() - 1
        self.ctx.bind_csr_device(ip.data_ptr(), ix.data_ptr(), self.n)

    def bind_text(self, d_text: int, text_bytes: int, d_row_off: int, n_rows: int, sep: str):
        """the tokeniser on profile text that is resident in HBM (device pointers): first-appearance vocabulary + CSR built and
        bound on the engine's stream (bfk_ctx_build_csr_device; one wait for the device inside)"""
        self.ctx.build_csr_device(d_text, text_bytes, d_row_off, n_rows, sep)
        self.n = int(n_rows)
        self._keep = None

    def cluster_text(self, d_text: int, text_bytes: int, d_row_off: int, n_rows: int, sep: str, max_dist: int, labels_out):
        """tokeniser + clustering kernels as ONE enqueue (bfk_ctx_cluster_text_device): where the variant join serves the step
        the clustering kernels follow the tokeniser with no wait in between; the bind is completed by sync()"""
        self.n = int(n_rows)
        self._keep = None
        self.ctx.cluster_text_device(d_text, text_bytes, d_row_off, n_rows, sep, max_dist, labels_out.data_ptr())

    def run(self):
        """context manager: what is enqueued inside goes to the engine's stream"""
        return torch.cuda.stream(self.stream)

    def new_labels(self, parts: int = 1):
        return torch.empty((parts, max(self.n, 1)), dtype=torch.int32, device=self.device)

    def new_flag(self):
        return torch.zeros(1, dtype=torch.int32, device=self.device)

    def cluster_shard(self, max_dist, shard, n_shards, labels_out):
        self.ctx.cluster(max_dist, labels_out.data_ptr(), shard, n_shards)

    def merge(self, gathered, n_parts, labels_out, changed=None):
        self.ctx.merge_labels(gathered.data_ptr(), n_parts, labels_out.data_ptr(),
                              0 if changed is None else changed.data_ptr())

    def sync(self):
        return self.ctx.sync()


class ShardedClusterer:
    def __init__(self, engine, rank: int = 0, world: int = 1, merge: str = "allgather", group=None, force_exchange: bool = False):
        """force_exchange: also a world of ONE rank goes through the exchange + merge branch (the collective on the engine's
        stream, k_merge behind it) instead of the one-rank shortcut — how a one-GPU box executes the code an N-GPU node runs
        (tests/test_gpu_scale.py::test_world_1_over_rccl_runs_the_exchange_and_merge)"""
        if merge not in ("allgather", "allreduce"):
            raise ValueError("merge must be 'allgather' or 'allreduce'")
        self.e, self.rank, self.world, self.merge, self.group = engine, rank, world, merge, group
        self.force_exchange = force_exchange
        self.rounds = 0
        self._settled = set()  # (max_dist, kernel configuration) whose first step on the bound CSR has been synced (see step)

    def bind(self, indptr, indices):
        self.e.bind(indptr, indices)
        self.local = self.e.new_labels(1)
        self.labels = self.e.new_labels(1)
        self.gathered = self.e.new_labels(self.world) if (self.world > 1 or self.force_exchange) else None
        self.flag = self.e.new_flag()
        self._settled = set()

    def step_text(self, d_text: int, text_bytes: int, d_row_off: int, n_rows: int, sep: str, max_dist: int):
        """the whole hot path as ONE step: profile strings resident in HBM -> first-appearance vocabulary + CSR (every rank
        tokenises the whole text: the CSR is replicated) -> this rank's shard of the pair work -> label exchange
        -> global canonical labels in HBM on every rank.  A step binds a new CSR, so its shard is synced before the exchange
        (see _step) every time."""
        run = getattr(self.e, "run", None)
        with run():
            if getattr(self, "labels", None) is None or self.labels.shape[1] != max(int(n_rows), 1):
                self.e.n = int(n_rows)
                self.local = self.e.new_labels(1)
                self.labels = self.e.new_labels(1)
                self.gathered = self.e.new_labels(self.world) if (self.world > 1 or self.force_exchange) else None
                self.flag = self.e.new_flag()
            if self.world == 1 and not self.force_exchange:
                # one rank: tokeniser and clustering kernels in one enqueue, nothing waits in between where the library can
                # drive the clustering kernels from device-resident counts
                self.e.cluster_text(d_text, text_bytes, d_row_off, n_rows, sep, max_dist, self.labels)
                out = self.labels[0]
            else:
                self.e.bind_text(d_text, text_bytes, d_row_off, n_rows, sep)
                self._settled = set()
                out = self._step(max_dist)
        torch.cuda.current_stream(self.e.device).wait_stream(self.e.stream)
        return out

    def step(self, max_dist: int):
        """CSR (resident) -> global canonical labels on every rank.  Asynchronous on the GPU engine except
        for the fix-point test of merge='allreduce'.  Everything — kernels, collectives, merge — is enqueued on the engine's
        stream (engines without one, the CPU test engine, run in line)."""
        run = getattr(self.e, "run", None)
        if run is None:
            return self._step(max_dist)
        with run():
            out = self._step(max_dist)
        # the engine's stream is non-blocking: whatever the CALLER's stream does with the labels next (a .cpu(), a kernel of
        # its own) has to be ordered behind the step — one event wait on the device, no host sync
        torch.cuda.current_stream(self.e.device).wait_stream(self.e.stream)
        return out

    def _step(self, max_dist: int):
        if self.world == 1 and not self.force_exchange:
            self.e.cluster_shard(max_dist, 0, 1, self.labels)
            return self.labels[0]
        self.e.cluster_shard(max_dist, self.rank, self.world, self.local)
        # which kernels a step runs depends on the candidate generator, the exact-edges switch and edge capture as well: a
        # change of any of them through the context's setters starts over (Context.config_epoch)
        key = (max_dist, getattr(getattr(self.e, "ctx", None), "config_epoch", 0))
        if key not in self._settled:
            # The FIRST step on a CSR at this max_dist and kernel configuration is synced before its labels are exchanged: a candidate generator
            # that gives up on the input (variant join: probe chains; prefix groups: groups too big) or a candidate queue
            # that overflows is repaired inside sync() — the shard is redone on the band kernels / in slices, `local` is
            # rewritten — and what is merged below is the repaired shard.  The context remembers the give-up and the grown
            # queue for this CSR, so later steps run clean and stay asynchronous.  (Merging first and repairing later
            # would leave every rank with labels that miss this shard's edges: stats `n_retry_slices != 0` after an
            # UNSYNCED exchange means exactly that.)
            self.e.sync()
            self._settled.add(key)
        if self.merge == "allgather":
            dist.all_gather_into_tensor(self.gathered.view(-1), self.local.view(-1), group=self.group)
            self.e.merge(self.gathered, self.world, self.labels)
            self.rounds = 1
            return self.labels[0]
        cur = self.local
        self.rounds = 0
        while True:
            red = self.gathered[0:1]
            red.copy_(cur)
            dist.all_reduce(red, op=dist.ReduceOp.MIN, group=self.group)
            self.e.merge(red, 1, self.labels, self.flag)
            dist.all_reduce(self.flag, op=dist.ReduceOp.MAX, group=self.group)
            self.rounds += 1
            if int(self.flag.item()) == 0 or self.rounds >= 64:
                return self.labels[0]
            cur = self.labels


def concurrent_streams(device_index: int, depth: int, candidates: int = 12):
    """`depth` streams of which no two share a hardware queue, as far as that can be had.  The HIP runtime deals streams onto
    a few hardware queues (four by default, GPU_MAX_HW_QUEUES) and kernels of two streams on one queue run behind each other:
    a pipeline whose three streams sat on queues {2, 4, 4} never had three kernels in flight and took 0.166 ms per step, on
    {4, 1, 2} it had them 57 % of the time and took 0.134 (rocprofv3 kernel trace, `Queue_Id`: profiles/r05_d_pipeline_overlap.txt).
    The API does not say which queue a stream got, but it shows: a one-block spin kernel is put on a stream already chosen and a
    tiny kernel on the candidate — if the tiny one finishes while the spin kernel still runs, the two are on different queues.
    Candidates that fail are kept alive until the choice is made (so that the next stream created lands elsewhere)."""
    dev = torch.device("cuda", device_index)
    torch.cuda.set_device(dev)
    probe = torch.zeros(64, device=dev)
    torch.cuda.synchronize(dev)
    try:
        # the spin kernel's length in its own unit: aim at ~0.3 ms
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(200_000)
        e1.record()
        e1.synchronize()
        spin = int(200_000 * 0.3 / max(e0.elapsed_time(e1), 1e-3))
    except Exception:  # (no spin kernel in this torch build: the streams as they come)
        return [torch.cuda.Stream(dev) for _ in range(depth)]

    def beside(a, b):
        ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(a):
            torch.cuda._sleep(spin)
            ev_a.record(a)
        with torch.cuda.stream(b):
            probe.add_(1.0)
            ev_b.record(b)
        ev_b.synchronize()
        ran_beside = not ev_a.query()
        ev_a.synchronize()
        return ran_beside

    chosen, rejected = [], []
    for _ in range(max(candidates, depth)):
        if len(chosen) == depth:
            break
        s = torch.cuda.Stream(dev)
        (chosen if all(beside(c, s) for c in chosen) else rejected).append(s)
    while len(chosen) < depth:  # (fewer queues than contexts: the rest share)
        chosen.append(rejected.pop() if rejected else torch.cuda.Stream(dev))
    torch.cuda.synchronize(dev)
    return chosen


class TextPipeline:
    """Text steps (profile strings resident in HBM -> labels in HBM, bfk_ctx_cluster_text_device) on `depth` resident contexts of
    ONE GPU that take the steps in turn, each context with its own stream and its own buffers (0.15 GB at 100k rows, 2.5 GB at
    1M, of 288).  A step is a chain of twelve dependent launches of which the first ones of the vocabulary hash (34 us of ~190
    at 100k rows) and every launch's tail leave most of the chip idle; with the next batch's step running beside it on another
    stream the chip is shared by the two (measured, 100k rows, max-dist 1: 0.198 ms per step with one context, 0.146 with two,
    0.135 with three, 0.175 with four — the host then cannot enqueue fast enough).  Every step is the complete hot path on its
    batch; what changes is that steps of DIFFERENT batches overlap.

    step_text(..., labels_out) enqueues and returns an event that fires when the step's labels are in `labels_out` (a CUDA int32
    tensor of n_rows; the caller's, one per step in flight — like d_labels_out of the C-ABI).  The step's stream waits for the
    caller's current stream first (the text is there); the caller's stream is NOT made to wait for the step — that would order
    every later step behind it: wait for the event, or call sync().  sync() completes every context (a step whose input was
    outside what its device-driven launch assumed, or whose join gave up, is redone there) and returns the last step's stats,
    with `n_retry_slices` = the sum over the contexts."""

    def __init__(self, device_index: int, depth: int = 3, candidate_path: str | None = None):
        if depth < 1:
            raise ValueError("depth must be at least 1")
        # the streams the steps run on: on hardware queues of their own where that can be had (concurrent_streams: two of the
        # pipeline's streams on one queue run behind each other — 0.157-0.166 ms per step instead of 0.128-0.134 at 100k rows)
        torch.cuda.set_device(torch.device("cuda", device_index))
        streams = concurrent_streams(device_index, depth)
        self.engines = [GpuEngine(device_index, st) for st in streams]
        if candidate_path is not None:
            for e in self.engines:
                e.ctx.set_candidate_path(candidate_path)
        self.k = 0

    @property
    def depth(self) -> int:
        return len(self.engines)

    def step_text(self, d_text: int, text_bytes: int, d_row_off: int, n_rows: int, sep: str, max_dist: int, labels_out,
                  inputs_ready: bool = False, want_event: bool = True):
        """inputs_ready: the text and the offsets are complete in HBM already (nothing of the caller's stream to wait for);
        want_event = False: no event is recorded (the caller will sync()).  Both save host time per step — at 0.14 ms per step the
        host's twelve launches are half of it."""
        e = self.engines[self.k % len(self.engines)]
        self.k += 1
        if not inputs_ready:
            e.stream.wait_stream(torch.cuda.current_stream(e.device))
        e.n = int(n_rows)
        e._keep = None
        # (the context launches on the stream it was given at construction: no stream guard needed around the call)
        e.ctx.cluster_text_device(d_text, text_bytes, d_row_off, n_rows, sep, max_dist, labels_out.data_ptr())
        if not want_event:
            return None
        done = torch.cuda.Event()
        done.record(e.stream)
        return done

    def sync(self):
        last = (self.k - 1) % len(self.engines) if self.k else 0
        stats, retries = None, 0
        for i, e in enumerate(self.engines):
            st = e.sync()
            retries += int(st.get("n_retry_slices", 0))
            if i == last:
                stats = dict(st)
        stats["n_retry_slices"] = retries
        return stats

    def close(self):
        for e in self.engines:
            e.ctx.close()
