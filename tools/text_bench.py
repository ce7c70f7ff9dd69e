"""Timing of the device tokeniser (bfk_text.hip) against the host tokeniser, and of the text -> labels one-shot entry.
usage: python tools/text_bench.py [rows ...]   (on a GPU box; prints one JSON line per size)"""

import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from breakfast_amd import _lib  # noqa: E402
from breakfast_amd.synth import generate_profiles  # noqa: E402


def med(f, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2], r


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [100000]
    for n in sizes:
        rows = list(dict.fromkeys(generate_profiles(n)))
        buf, off = _lib.pack_rows(rows)
        t_host, (ip, ix, nv) = med(lambda: _lib.build_csr_bytes(buf, off, " "), 5)
        ctx = _lib.Context(0)
        ctx.build_csr(buf, off, " ")
        t_dev, _ = med(lambda: ctx.build_csr(buf, off, " "))
        ctx.set_profiling(True)
        phases = []
        for _ in range(5):
            ctx.build_csr(buf, off, " ")
            phases.append(ctx.text_stats())
        ph = {k: sorted(p[k] for p in phases)[2] for k in ("ms_h2d", "ms_scan", "ms_hash", "ms_ids", "ms_total")}
        ctx.set_profiling(False)
        d_ip, d_ix = ctx.download_csr()
        same = bool(np.array_equal(d_ip, ip) and np.array_equal(d_ix, ix))
        # a buffer the driver has never seen (first touch: pages are pinned on the fly)
        fresh = []
        for _ in range(3):
            b2 = bytes(bytearray(buf))
            t0 = time.perf_counter()
            ctx.build_csr(b2, off, " ")
            fresh.append((time.perf_counter() - t0) * 1e3)
        ctx.close()
        _lib.cluster_text(buf, off, " ", 1)
        t_text, _ = med(lambda: _lib.cluster_text(buf, off, " ", 1))
        t_csr, _ = med(lambda: _lib.cluster_csr(ip, ix, 1))
        print(json.dumps({"rows": n, "n_unique": len(rows), "text_bytes": len(buf), "nnz": int(ip[-1]), "n_vocab": nv,
                          "csr_equal": same, "host_build_csr_ms": round(t_host, 3), "device_build_csr_ms": round(t_dev, 3),
                          "device_build_fresh_buffer_ms": [round(x, 3) for x in fresh], "device_phases_ms": ph,
                          "cluster_text_ms": round(t_text, 3), "host_build_plus_cluster_csr_ms": round(t_host + t_csr, 3),
                          "text_GBps_tokeniser_kernels": round(len(buf) / (ph["ms_total"] - ph["ms_h2d"]) / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
