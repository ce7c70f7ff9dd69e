"""CPU, world_size 2, gloo: the one-process-per-GPU driver (breakfast_amd/distributed.py) — sharding of the
edge work, the label exchange (all_gather and the all_reduce(MIN) fix point) and the merge — with an
oracle-backed engine standing in for the HIP kernels (tests may use oracle/; the product has no CPU engine)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from conftest import load_stage

from breakfast_amd.distributed import ShardedClusterer


def _uf_labels(n, edges, init=None):
    parent = np.arange(n) if init is None else init.copy()

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b in edges:
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            if ra < rb:
                ra, rb = rb, ra
            parent[ra] = rb
    return np.array([find(i) for i in range(n)], dtype=np.int32), parent


class OracleEngine:
    """Same interface as GpuEngine; the 'kernels' are numpy union-find over the golden edge list, of which
    this rank only sees the edges dealt to its shard (like the work items dealt round-robin on the GPU)."""

    def __init__(self, edges):
        self.edges = edges
        self.parent = None

    def bind(self, indptr, indices):
        self.n = len(indptr) - 1

    def new_labels(self, parts=1):
        return torch.zeros((parts, max(self.n, 1)), dtype=torch.int32)

    def new_flag(self):
        return torch.zeros(1, dtype=torch.int32)

    def cluster_shard(self, max_dist, shard, n_shards, labels_out):
        mine = self.edges[shard::n_shards]
        lab, self.parent = _uf_labels(self.n, mine)
        labels_out[0, : self.n] = torch.from_numpy(lab)

    def merge(self, gathered, n_parts, labels_out, changed=None):
        g = gathered.numpy().reshape(n_parts, -1)[:, : self.n]
        pseudo = [(i, int(g[p, i])) for p in range(n_parts) for i in range(self.n) if g[p, i] != i]
        lab, self.parent = _uf_labels(self.n, pseudo, self.parent)
        labels_out[0, : self.n] = torch.from_numpy(lab)
        if changed is not None:
            changed[0] = int(np.any(lab != g[0]))

    def sync(self):
        return {}


def _worker(rank, world, port, name, merge, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_stage(name)
        eng = OracleEngine(g["edges"])
        sc = ShardedClusterer(eng, rank, world, merge)
        sc.bind(g["indptr"], g["indices"])
        labels = sc.step(g["max_dist"]).numpy()[: len(g["labels"])]
        np.save(os.path.join(out_dir, f"labels_{rank}.npy"), labels)
        np.save(os.path.join(out_dir, f"rounds_{rank}.npy"), np.array([sc.rounds]))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("merge", ["allgather", "allreduce"])
@pytest.mark.parametrize("name", ["syn2000_d1", "indel2000_d5", "multiset300_d2"])
def test_two_rank_label_merge_matches_reference(name, merge, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, merge, str(tmp_path)), nprocs=world, join=True)
    want = load_stage(name)["labels"]
    for r in range(world):
        got = np.load(tmp_path / f"labels_{r}.npy")
        assert np.array_equal(got, want), f"rank {r} labels differ from the 1-process reference labels"
    rounds = int(np.load(tmp_path / "rounds_0.npy")[0])
    assert rounds == 1 if merge == "allgather" else 1 <= rounds <= 8


def test_single_rank_is_passthrough():
    g = load_stage("syn200_d1")
    eng = OracleEngine(g["edges"])
    sc = ShardedClusterer(eng, 0, 1)
    sc.bind(g["indptr"], g["indices"])
    assert np.array_equal(sc.step(1).numpy()[: len(g["labels"])], g["labels"])
    with pytest.raises(ValueError):
        ShardedClusterer(eng, 0, 1, merge="ring")
