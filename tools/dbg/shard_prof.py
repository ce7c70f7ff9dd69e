"""one shard of an N-way split, repeated (for rocprofv3 --kernel-trace --stats): python tools/dbg/shard_prof.py <rows> <d> <indels 0/1> <shard> <n_shards>"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from breakfast_amd import _lib
from breakfast_amd.synth import generate_profiles
rows, d, indels, shard, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
kw = dict(p_del=0.05, p_ins=0.01) if indels else {}
uf = list(dict.fromkeys(generate_profiles(rows, **kw)))
indptr, indices, _ = _lib.build_csr(uf, " ")
ctx = _lib.Context(0)
ctx.upload_csr(indptr, indices)
d_out = ctx.alloc(4 * len(uf))
for _ in range(12):
    ctx.cluster(d, d_out, shard, world)
st = ctx.sync()
print({k: st[k] for k in ("n_candidates", "n_edges", "n_connected", "path", "n_retry_slices")})
