"""where does bfk_cluster_text spend its time on the host side (pieces through the ctx API)"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from breakfast_amd import _lib
from breakfast_amd.synth import generate_profiles
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rows = list(dict.fromkeys(generate_profiles(n)))
buf, off = _lib.pack_rows(rows)
ctx = _lib.Context(0)
d_out = ctx.alloc(4 * len(rows))
for rep in range(4):
    t0 = time.perf_counter(); ctx.build_csr(buf, off, " ")
    t1 = time.perf_counter(); ctx.cluster(1, d_out)
    t2 = time.perf_counter(); st = ctx.sync()
    t3 = time.perf_counter(); lab = ctx.download_i32(d_out, len(rows))
    t4 = time.perf_counter()
    print(f"build {1e3*(t1-t0):.3f}  enqueue {1e3*(t2-t1):.3f}  sync(+stats) {1e3*(t3-t2):.3f}  download {1e3*(t4-t3):.3f} ms  path {st['path']}")
for rep in range(3):
    t0 = time.perf_counter(); r = _lib.cluster_text(buf, off, " ", 1); t1 = time.perf_counter()
    print(f"cluster_text {1e3*(t1-t0):.3f} ms")
