"""Per-phase stamps of k_tok_hashb's blocks (BFK_TOK_STAMPS=1 prints them from the library) and the tokeniser's event times.
usage (on a GPU box): BFK_TOK_STAMPS=1 python tools/tok_stamps.py [rows]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from breakfast_amd import _lib  # noqa: E402
from breakfast_amd.synth import generate_profiles  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rows = list(dict.fromkeys(generate_profiles(n)))
buf, off = _lib.pack_rows(rows)
ctx = _lib.Context(0)
for _ in range(20):
    ctx.build_csr(buf, off, " ")
ctx.set_profiling(True)
ph = []
for _ in range(5):
    ctx.build_csr(buf, off, " ")
    ph.append(ctx.text_stats())
print({k: round(sorted(p[k] for p in ph)[2], 4) for k in ("ms_scan", "ms_hash", "ms_ids", "ms_total")})
