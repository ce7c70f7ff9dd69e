"""bfk_cluster_text from a pinned buffer (bfk_host_alloc) with the text sent in BFK_TOK_PIECES pieces: ms per call.
usage (GPU box): for n in 0 2 3 4; do BFK_TOK_PIECES=$n python tools/host_pieces.py; done"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from breakfast_amd import _lib  # noqa: E402
from breakfast_amd.synth import generate_profiles  # noqa: E402

rows = list(dict.fromkeys(generate_profiles(100000)))
buf, off = _lib.pack_rows(rows)
lab = np.empty(len(rows), dtype=np.int32)
pins = []
for _ in range(4):
    pb = _lib.PinnedBuffer(len(buf))
    pb.view[:] = np.frombuffer(buf, dtype=np.uint8)
    pins.append(pb)
ref = _lib.cluster_text(buf, off, " ", 1, want_stats=False, labels_out=lab)[0].copy()
ts = []
for i in range(40):
    t0 = time.perf_counter()
    out = _lib.cluster_text(pins[i % 4], off, " ", 1, want_stats=False, labels_out=lab)[0]
    ts.append((time.perf_counter() - t0) * 1e3)
assert np.array_equal(out, ref)
ts = sorted(ts[8:])
print(f"BFK_TOK_PIECES={os.environ.get('BFK_TOK_PIECES', '-')}: pinned call median {ts[len(ts) // 2]:.3f} ms, min {ts[0]:.3f}")
