"""max_dist 2: band kernels against prefix groups over the input size, per family — where PG_MIN_ROWS for d = 2 belongs.
Run on a GPU box:  python tools/d2_crossover.py [d]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from family_matrix import rows_of, time_path  # noqa: E402

from breakfast_amd import _lib  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for family in ("default", "star", "long", "aa"):
    for n in (50000, 100000, 150000, 200000, 300000):
        uf = list(dict.fromkeys(rows_of(family, n)))
        indptr, indices, _ = _lib.build_csr(uf, " ")
        ms = {p: time_path(indptr, indices, d, p, 20)[0] for p in ("allpairs", "prefix")}
        print(f"{family:8s} {n:7d} d={d}  band {ms['allpairs']:.3f}  prefix {ms['prefix']:.3f}  band/prefix {ms['allpairs'] / ms['prefix']:.2f}", flush=True)
