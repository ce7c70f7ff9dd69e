"""debug: one fuzz case under the exact and the pruning verify kernels (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch  # noqa
from test_gpu_parity import fuzz_case
from breakfast_amd import _lib
from oracle import ref_port as orc

seed, which = int(sys.argv[1]), [int(x) for x in sys.argv[2].split(",")]
rng = np.random.default_rng(5000 + seed)
for it in range(20):
    rows, indptr, indices, d, alphabet = fuzz_case(rng)
    if it not in which:
        continue
    want = orc.cluster_csr(indptr, indices, d, n_threads=4)["labels"]
    for env in ({"BFK_EXACT_EDGES": "1"}, {"BFK_SKIP_CONNECTED": "1"}, {"BFK_SKIP_CONNECTED": "1", "BFK_VERIFY_PHASES": "1"},
                {"BFK_SKIP_CONNECTED": "1", "BFK_VERIFY_GRID": "32"}):
        for k in ("BFK_EXACT_EDGES", "BFK_SKIP_CONNECTED", "BFK_VERIFY_PHASES", "BFK_VERIFY_GRID"):
            os.environ.pop(k, None)
        os.environ.update(env)
        os.environ["BFK_PG"] = "0"
        got, st = _lib.cluster_csr(indptr, indices, d)
        print(it, "n", len(rows), "d", d, env, "ok", np.array_equal(got, want), "cand", st["n_candidates"], "edges", st["n_edges"],
              "conn", st["n_connected"], "comps", len(np.unique(got)), flush=True)
