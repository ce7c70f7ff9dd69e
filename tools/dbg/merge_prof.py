"""the merge of an N-way split on one device, repeated (for rocprofv3 --kernel-trace --stats):
python tools/dbg/merge_prof.py <rows> <d> <indels 0/1> <n_shards>"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time
import numpy as np
from breakfast_amd import _lib
from breakfast_amd.synth import generate_profiles
rows, d, indels, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kw = dict(p_del=0.05, p_ins=0.01) if indels else {}
uf = list(dict.fromkeys(generate_profiles(rows, **kw)))
indptr, indices, _ = _lib.build_csr(uf, " ")
n = len(uf)
ctx = _lib.Context(0)
ctx.upload_csr(indptr, indices)
d_out = ctx.alloc(4 * n)
labels = []
for s in range(world):
    ctx.cluster(d, d_out, s, world)
    ctx.sync()
    labels.append(ctx.download_i32(d_out, n).copy())
d_g = ctx.alloc(4 * n * world)
ctx.upload_i32(np.concatenate(labels), d_g)
d_m = ctx.alloc(4 * n)
ts = []
for _ in range(6):
    ctx.cluster(d, d_out, 0, world)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.merge_labels(d_g, world, d_m)
    ctx.sync(want_stats=False)
    ts.append(round((time.perf_counter() - t0) * 1e3, 3))
print("merge ms (host-timed):", ts)
