"""Every candidate generator on workload shapes the dispatch thresholds were NOT fitted on (breakfast_amd/synth.py:
generate_family — long rows, a star phylogeny, amino-acid tokens) next to the default generator's family: ms per step of the
resident-CSR clustering step with the automatic choice and with each generator forced, the automatic choice against the best
forced one, queue retries, and that all of them give the same labels.
Run on a GPU box:  python tools/family_matrix.py [rows ...] > gpurun_out/family_matrix.txt"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from breakfast_amd import _lib, synth  # noqa: E402


def rows_of(family, n):
    if family == "default":
        return synth.generate_profiles(n)
    if family == "default_indels":
        return synth.generate_profiles(n, p_del=0.05, p_ins=0.01)
    return synth.generate_family(family, n)


def time_path(indptr, indices, d, path, steps):
    ctx = _lib.Context(0)
    ctx.set_candidate_path(path)
    ctx.upload_csr(indptr, indices)
    n = len(indptr) - 1
    d_out = ctx.alloc(4 * n)
    st = None
    for _ in range(4):  # (a queue that overflows is grown by the sync: until a step runs clean)
        ctx.cluster(d, d_out)
        st = ctx.sync()
        if st["n_retry_slices"] == 0:
            break
    first_retries = st["n_retry_slices"]
    ctx.cluster(d, d_out)
    ctx.sync(want_stats=False)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.cluster(d, d_out)
    ctx.sync(want_stats=False)
    ms = (time.perf_counter() - t0) / steps * 1e3
    st = ctx.sync()
    lab = ctx.download_i32(d_out, n).copy()
    ctx.close()
    return ms, st, lab, first_retries


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [20000, 100000, 1000000]
    print("family rows N_u k_mean k_max d | auto ms (generator) | allpairs ms | join / prefix ms | auto vs best | retries | labels equal")
    worst = 0.0
    for family in ("default", "default_indels", "long", "star", "aa"):
        for n in sizes:
            uf = list(dict.fromkeys(rows_of(family, n)))
            indptr, indices, _ = _lib.build_csr(uf, " ")
            k = np.diff(indptr)
            for d in (1, 2, 3, 5):
                steps = 20 if len(uf) <= 200000 else 8
                res = {}
                for path in ("auto", "allpairs", "join" if d == 1 else "prefix"):
                    try:
                        res[path] = time_path(indptr, indices, d, path, steps)
                    except _lib.BfkError as e:
                        res[path] = None
                        print(f"  ({family} {n} d={d} {path}: {e})")
                ok = {p: r for p, r in res.items() if r}
                best = min(r[0] for p, r in ok.items() if p != "auto")
                auto = ok["auto"][0]
                gen = {0: "band", 1: "join", 2: "prefix"}[ok["auto"][1]["path"]]
                same = all(np.array_equal(r[2], ok["auto"][2]) for r in ok.values())
                retries = {p: (r[3], r[1]["n_retry_slices"]) for p, r in ok.items()}
                third = "join" if d == 1 else "prefix"
                ratio = auto / best
                worst = max(worst, ratio)
                print(f"{family:14s} {n:8d} {len(uf):8d} {k.mean():6.1f} {int(k.max()):4d} d={d} | auto {auto:8.3f} ({gen:6s}) | "
                      f"allpairs {ok['allpairs'][0]:8.3f} | {third} {ok[third][0] if third in ok else float('nan'):8.3f} | "
                      f"auto/best {ratio:5.2f}{'  <-- > 1.3' if ratio > 1.3 else ''} | first-step / steady retries {retries} | "
                      f"{'same labels' if same else 'LABELS DIFFER'}", flush=True)
    print(f"worst auto / best forced: {worst:.2f}")


if __name__ == "__main__":
    main()
