"""per-shard statistics of a sharded labels-only run on one device (debug helper)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from breakfast_amd import _lib
from breakfast_amd.synth import generate_profiles

rows = generate_profiles(30_000, p_del=0.05, p_ins=0.01)
uf = list(dict.fromkeys(rows))
indptr, indices, _ = _lib.build_csr(uf, " ")
n = len(uf)
for d in (3, 5):
    want, st1 = _lib.cluster_csr(indptr, indices, d)
    print("one shard", d, {k: st1[k] for k in ("n_candidates", "n_edges", "n_connected", "pairs_filtered")})
    ctx = _lib.Context(0)
    ctx.upload_csr(indptr, indices)
    d_gath = ctx.alloc(4 * n * 3)
    for s in range(3):
        ctx.cluster(d, d_gath + 4 * n * s, s, 3)
        st = ctx.sync()
        print(" shard", s, {k: st[k] for k in ("n_candidates", "n_edges", "n_connected", "pairs_filtered")})
    ctx.close()
