import json
import sys
from pathlib import Path

import numpy as np
import pytest
# torch bundles its own HIP / HSA runtime and loads it by path: if libbfk.so has already pulled the system runtime into
# the process, torch.cuda then finds "no HIP GPUs" (two runtimes, one device).  Loaded first, torch's copy serves both.
import torch  # noqa: F401

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "exact_edges: the test compares bfk_stats.n_edges with the number of edges of the graph: "
                            "every candidate is checked (BFK_EXACT_EDGES=1); unmarked tests run the library's default, which "
                            "drops candidates of already connected rows at max_dist >= 3 (tests/test_gpu_pruned.py)")


@pytest.fixture(autouse=True)
def _exact_edges_mode(request, monkeypatch):
    if request.node.get_closest_marker("exact_edges"):
        monkeypatch.setenv("BFK_EXACT_EDGES", "1")
    else:
        monkeypatch.delenv("BFK_EXACT_EDGES", raising=False)


def stage_names():
    return sorted(p.stem[len("stage_"):] for p in GOLD.glob("stage_*.npz"))


def load_stage(name):
    z = np.load(GOLD / f"stage_{name}.npz", allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["sep"] = str(d["sep"])
    d["max_dist"] = int(d["max_dist"])
    d["min_cluster_size"] = int(d["min_cluster_size"])
    d["features"] = [str(x) for x in d["features"]]
    d["ufeatures"] = [str(x) for x in d["ufeatures"]]
    d["clusters_tsv"] = d["clusters_tsv"].tobytes()
    return d


@pytest.fixture(scope="session")
def kats():
    return json.loads((GOLD / "kats.json").read_text())


@pytest.fixture(scope="session")
def cli_runs():
    return json.loads((GOLD / "cli_runs.json").read_text())


def set_generator(monkeypatch, generator):
    """force the candidate generator of the max-dist >= 2 kernels for a test: "band" = the band kernels; "prefix" = the prefix
    groups with round 2's WHOLE-GROUP walk (records row-major, BFK_PG_POS=0 — what token ids beyond 28 bits still get);
    "prefix_pos" = the prefix groups with the positional filter, the default at every size since round 3"""
    monkeypatch.setenv("BFK_PG", "0" if generator == "band" else "1")
    if generator == "prefix_pos":
        monkeypatch.setenv("BFK_PG_POS", "1")
    elif generator == "prefix":
        monkeypatch.setenv("BFK_PG_POS", "0")
    else:
        monkeypatch.delenv("BFK_PG_POS", raising=False)
