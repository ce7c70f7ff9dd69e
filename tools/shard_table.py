"""One-device REHEARSAL of the N-GPU split: what every rank of an N-rank run would execute, timed shard by
shard on ONE MI355X (each shard alone on the device, as it would be on a GPU of its own), plus the merge of the N label
arrays.  No xGMI, no RCCL: the label exchange is not measured here (its payload is stated); the driver's SCALE run on an
8-GPU node is the measurement.  Output: one JSON object per workload.

usage (on a GPU box): python tools/shard_table.py [c3] [c4] [--rows R]"""

import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from breakfast_amd import _lib  # noqa: E402
from breakfast_amd.synth import generate_profiles  # noqa: E402

PH = ("ms_prep", "ms_prefilter", "ms_verify", "ms_flatten", "ms_total")


def table(name, n_rows, d, indels, steps=8, path="auto"):
    kw = dict(p_del=0.05, p_ins=0.01) if indels else {}
    rows = list(dict.fromkeys(generate_profiles(n_rows, **kw)))
    indptr, indices, _ = _lib.build_csr(rows, " ")
    n = len(rows)
    ctx = _lib.Context(0)
    ctx.set_candidate_path(path)
    ctx.upload_csr(indptr, indices)
    d_out = ctx.alloc(4 * n)
    res = {"workload": name, "candidate_path": path, "rows": n_rows, "n_unique": n, "max_dist": d, "indels": indels,
           "what": "one-device rehearsal: every shard of an N-way split run alone on one MI355X (HIP events around the phases, "
                   f"mean of {steps} steps after 3 warm-up steps), then the merge of the N label arrays on that device; the "
                   "label exchange itself (all_gather of 4*N_u bytes per rank over xGMI) is NOT measured here",
           "splits": []}
    want = None
    for world in (1, 2, 4, 8):
        shards = []
        labels = []
        for s in range(world):
            for _ in range(3):
                ctx.cluster(d, d_out, s, world)
            ctx.sync()
            ctx.set_profiling(True)
            for _ in range(steps):
                ctx.cluster(d, d_out, s, world)
            st = ctx.sync()
            ctx.set_profiling(False)
            shards.append({"shard": s, **{k: round(st[k], 4) for k in PH}, "path": st["path"], "n_edges": st["n_edges"],
                           "n_candidates": st["n_candidates"], "retry": st["n_retry_slices"]})
            labels.append(ctx.download_i32(d_out, n).copy())
        entry = {"n_ranks": world, "per_shard": shards,
                 "slowest_shard_ms": {k: max(x[k] for x in shards) for k in PH}}
        if world == 1:
            want = labels[0]
            entry["labels"] = "reference"
        else:
            # merge on the device: rank 0's forest (its own shard: run it again so the forest is its) + the gathered parts
            d_g = ctx.alloc(4 * n * world)
            ctx.upload_i32(np.concatenate(labels), d_g)
            d_m = ctx.alloc(4 * n)
            import time
            ts = []
            for _ in range(5):
                ctx.cluster(d, d_out, 0, world)
                ctx.sync()
                t0 = time.perf_counter()
                ctx.merge_labels(d_g, world, d_m)
                ctx.sync(want_stats=False)  # (the statistics of a sync are host work: 0.7 ms at 1M rows on the group path)
                ts.append((time.perf_counter() - t0) * 1e3)
            got = ctx.download_i32(d_m, n)
            entry["merge_ms_host_timed"] = round(sorted(ts)[2], 4)
            entry["labels_equal_one_rank"] = bool(np.array_equal(got, want))
            entry["exchange_payload_bytes_per_rank"] = 4 * n
            entry["edges_found_by_exactly_one_shard"] = int(sum(x["n_edges"] for x in shards))
        res["splits"].append(entry)
    ctx.close()
    one = res["splits"][0]["slowest_shard_ms"]["ms_total"]
    for e in res["splits"]:
        e["kernels_vs_one_rank"] = round(one / e["slowest_shard_ms"]["ms_total"], 3)
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = 1000000
    if "--rows" in sys.argv:
        rows = int(sys.argv[sys.argv.index("--rows") + 1])
        args = [a for a in args if a != str(rows)]
    which = args or ["c3", "c4"]
    for w in which:
        if w == "c3":
            print(json.dumps(table("configs[3]: 1M profiles, max-dist 1", rows, 1, False)), flush=True)
        elif w == "c4":
            print(json.dumps(table("configs[4]: 1M profiles, max-dist 5, indels kept", rows, 5, True)), flush=True)
        elif w.startswith("d"):  # e.g. d3 / d3band / d2: indels kept, max-dist as given, band kernels forced with the suffix
            dd = int(w[1])
            pth = "allpairs" if w.endswith("band") else "auto"
            print(json.dumps(table(f"{rows} profiles, max-dist {dd}, indels kept, path {pth}", rows, dd, True, path=pth)), flush=True)


if __name__ == "__main__":
    main()
