"""The band kernels (k_sig .. k_prefilter .. k_verify: north_star's all-pairs design) forced on a resident CSR: ms per step, the
phases from HIP events, a CRC of the labels — and, with a library built by `make PF_DEBUG=1` and BFK_PF_DEBUG=4, the per-wave
stamps of k_prefilter for tools/pf_timeline.py.
Run on a GPU box:  python tools/allpairs_ab.py [rows] [max_dist] [steps]        (BFK_LIB=... selects another build)"""
import os
import sys
import time
import zlib
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from breakfast_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
fam = os.environ.get("AB_FAMILY", "default")   # default | long | star | aa (synth.generate_family)
rows = list(dict.fromkeys(synth.generate_profiles(n) if fam == "default" else synth.generate_family(fam, n)))
indptr, indices, _ = _lib.build_csr(rows, " ")
nu = len(indptr) - 1
ctx = _lib.Context(0)
ctx.set_candidate_path(os.environ.get("AB_PATH", "allpairs"))
ctx.upload_csr(indptr, indices)
d_out = ctx.alloc(4 * nu)
for _ in range(4):
    ctx.cluster(d, d_out)
    st = ctx.sync()
    if st["n_retry_slices"] == 0:
        break
if int(os.environ.get("BFK_PF_DEBUG", "0")) & 4:
    print("stamps written by the sync above; stats:", {k: st[k] for k in ("n_candidates", "n_edges", "n_work_items")})
    sys.exit(0)
best = []
for rep in range(5):
    for _ in range(50):
        ctx.cluster(d, d_out)
    ctx.sync(want_stats=False)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.cluster(d, d_out)
    ctx.sync(want_stats=False)
    best.append((time.perf_counter() - t0) / steps * 1e3)
ctx.set_profiling(True)
ph = []
for _ in range(16):
    ctx.cluster(d, d_out)
    ph.append(ctx.sync())
lab = ctx.download_i32(d_out, nu)
keys = ("ms_prep", "ms_prefilter", "ms_verify", "ms_flatten")
med = {k: round(sorted(p[k] for p in ph)[8], 4) for k in keys if k in ph[0]}
print(f"{fam} {os.environ.get('AB_PATH', 'allpairs')} {nu} rows k_mean {np.diff(indptr).mean():.1f} d={d}: ms/step min {min(best):.4f} median {sorted(best)[2]:.4f} | phases {med} | candidates {ph[-1]['n_candidates']} "
      f"edges {ph[-1]['n_edges']} tiles {ph[-1]['n_work_items']} | labels crc {zlib.crc32(lab.tobytes()):08x}")
