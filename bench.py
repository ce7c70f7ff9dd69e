#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: genome-pair dists/sec + clusters.tsv wall-clock, 100k seqs, max-dist 1.

  python bench.py [--gpus N --steps K --warmup W] [--rows R --max-dist D --indels --merge allgather|allreduce]

`--gpus N` with N > 1 runs BASELINE configs[3] by default — 1M profiles, max-dist 1, the SAME input on every N (strong
scaling; --rows / --max-dist / --indels choose another, e.g. configs[4]) — and, with no WORLD_SIZE in the environment,
starts the N ranks itself (child processes of
`python -m torch.distributed.run`, before this process has touched the GPU), relays rank 0's JSON line and exits
with the children's status; under a launcher (WORLD_SIZE set) it is one rank.

What the ONE JSON line holds (every number is measured in this run unless it says otherwise):

  value / ms_per_step    a step = one pass of the WHOLE hot path over the batch with its input resident in HBM: the N_u profile
                         strings (one byte buffer + offsets, in device memory; the steps rotate over enough distinct copies
                         that the text does not come out of the Infinity Cache) -> separator scan, vocabulary table,
                         first-appearance ids, CSR (bfk_text.hip) -> clustering kernels -> canonical labels in HBM (+ RCCL label
                         merge for N > 1).  EXACTLY --steps steps timed between barrier + synchronize;
                         value = N_u(N_u-1)/2 pairs resolved per step / time.  (Rounds 1-3 timed the clustering kernels on a
                         resident CSR here; that figure is now value_resident_csr.)
  value_resident_csr     the second half of the step alone: CSR resident in HBM -> labels (asynchronous launches)
  t_cluster_host_ms      the PCIe-inclusive figures (never `value`): N_u strings in HOST memory -> labels in HOST memory through
                         ONE C-ABI call (bfk_cluster_text): first call; a fresh pageable buffer per call (median of 9);
                         the same pageable buffer again; a pinned buffer from bfk_host_alloc.
                         value_host_inclusive = pairs / fresh-pageable-buffer time.
  sustained              the same steps as `value` for >= 1 s (so that an outside sampler sees the GPU busy)
  clusters_tsv_wall_s    metric (2): a fresh subprocess of the CLI, input file -> clusters.tsv, sha256 of the output
                         checked against tests/golden/sha256.json (the digest of the reference's own output); with the CLI's
                         default fast exit and with an ordinary process exit (clusters_tsv_wall_ordinary_exit_s)
  all_pairs              the resident-CSR step with the all-pairs kernels forced (k_sig .. k_prefilter .. k_verify): the
                         design north_star describes; the default at max-dist 1 is the variant join
  roofline               the step's dominant kernel (longest measured duration — at 100k rows the tokeniser's k_tok_hash):
                         bytes the executed algorithm must move (compulsory reads + writes of that kernel) / its duration
                         (HIP events on the launch stream, inside libbfk) vs the 8 TB/s HBM peak; `second_kernel` = the
                         dominant kernel of the other half (k_join / k_prefilter / k_pgwalk16 / k_verify_connected).
                         `reference_equivalent` keeps SURVEY 8(d)'s untiled operand-stream figure (what the reference's CPU
                         kernel touches) for comparison only — it is never `achieved`.
  cpu_baseline           the scikit-learn kernel the reference calls, timed on this host's cores on a bounded sample
                         (oracle/sk_port.py); the C oracle leg beside it.  bench-only use of oracle/.
"""

from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9  # 256 CU x 4 SIMD32 x 2.4 GHz


def reference_equivalent_bytes(k: np.ndarray, d: int, nnz: int):
    """SURVEY.md 8(d) 'one figure': B_alg = sum over merged (in-band) unordered pairs 4(k_i+k_j)
    + 8 B per length-pruned pair + 4*nnz + 8*N_u — what an untiled all-pairs merge (the reference's CPU kernel)
    touches.  Exact from the length histogram.  Reported for comparison; NOT what these kernels move."""
    n = len(k)
    cnt = np.bincount(k).astype(np.float64)
    ks = np.arange(len(cnt), dtype=np.float64)
    merged_pairs = float(np.sum(cnt * (cnt - 1) / 2))
    merged_bytes = float(np.sum(cnt * (cnt - 1) / 2 * 8 * ks))
    for dd in range(1, d + 1):
        if dd >= len(cnt):
            break
        a, b = cnt[:-dd], cnt[dd:]
        merged_pairs += float(np.sum(a * b))
        merged_bytes += float(np.sum(a * b * 4 * (ks[:-dd] + ks[dd:])))
    resolved = n * (n - 1) / 2
    b_alg = merged_bytes + 8.0 * (resolved - merged_pairs) + 4.0 * nnz + 8.0 * n
    return b_alg, merged_pairs, resolved


def host_cores():
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(indptr, indices, d, n_u, target_s=10.0, full_limit_s=50.0):
    """CPU legs on the host cores, each on a bounded sample of S query rows (the reference's select_ind shape,
    breakfast.py:241-245), scaled to pairs/s as S*(N_u-1)/2 / t — the share of the job's unordered pairs those S
    query rows account for.  Top level: the scikit-learn kernel the reference itself calls
    (pairwise_distances_chunked / _sparse_manhattan on the length bands, oracle/sk_port.py; time inside the sklearn
    calls).  `c_oracle`: the C restatement (oracle/bfk_oracle.c), OpenMP over query rows.  bench-only use of
    oracle/ (never the measured product)."""
    from oracle import ref_port as orc

    cores = host_cores()
    rng = np.random.default_rng(0)
    probe = np.sort(rng.choice(n_u, size=min(n_u, 64 * cores), replace=False)).astype(np.int64)
    t0 = time.perf_counter()
    orc.cluster_csr(indptr, indices, d, select_ind=probe, n_threads=cores)
    t_probe = time.perf_counter() - t0
    s = int(min(n_u, max(len(probe), len(probe) * target_s / max(t_probe, 1e-3))))
    sel = np.sort(rng.choice(n_u, size=s, replace=False)).astype(np.int64)
    t0 = time.perf_counter()
    res = orc.cluster_csr(indptr, indices, d, select_ind=sel, n_threads=cores)
    t = time.perf_counter() - t0
    port = {
        "value": s * (n_u - 1) / 2 / t, "unit": "pairs/s", "cores": cores, "kind": "port", "sampled": s < n_u,
        "kernel": "oracle/bfk_oracle.c (C restatement of the reference path)",
        "sample": f"{s} of {n_u} query rows x all columns (select_ind shape), {res['n_merges']} row merges "
                  f"in {t:.1f} s, OpenMP over query rows like sklearn's prange; extrapolated to the full job",
        "seconds": round(t, 2),
    }
    try:
        from oracle import sk_port

        if sk_port.available():
            sel0 = np.sort(rng.choice(n_u, size=min(n_u, 200), replace=False)).astype(np.int64)
            _, t0s = sk_port.neighbours(indptr, indices, d, select_ind=sel0)
            # 10k / 100k are timed IN FULL where that fits a minute on this host (every row a query row) — decided
            # by a bounded sample first; otherwise that sample is the figure, and the line says `sampled`
            s2 = int(min(n_u, max(200, 200 * target_s / max(t0s, 1e-3))))
            sel2 = np.sort(rng.choice(n_u, size=s2, replace=False)).astype(np.int64)
            _, ts = sk_port.neighbours(indptr, indices, d, select_ind=sel2)
            if s2 < n_u and ts * n_u / s2 <= full_limit_s:  # (the bounded sample says the whole job fits: time the whole job)
                s2 = n_u
                sel2 = np.arange(n_u, dtype=np.int64)
                _, ts = sk_port.neighbours(indptr, indices, d, select_ind=sel2)
            return {"value": s2 * (n_u - 1) / 2 / ts, "unit": "pairs/s", "cores": cores, "kind": "port", "sampled": s2 < n_u,
                    "kernel": "scikit-learn pairwise_distances_chunked(metric='manhattan') -> _sparse_manhattan: the "
                              "third-party kernel the reference calls (breakfast.py:259-267), driven band by band by "
                              "oracle/sk_port.py; " + sk_port.versions(),
                    "sample": (f"all {n_u} rows x their length bands" if s2 == n_u else f"{s2} of {n_u} query rows x their length bands, "
                               "extrapolated to the full job") + f": {sk_port.merges(indptr, d, sel2)} row merges, {ts:.1f} s inside the sklearn calls",
                    "seconds": round(ts, 2), "c_oracle": port}
    except Exception as e:  # the baseline is a report, never a reason to lose the bench line
        port["sklearn_error"] = repr(e)
    return port


def predicted_speedup(one_gpu, exchange, world):
    """Amdahl reading of the N-rank step from the ONE-GPU phases measured in this run: every rank repeats the
    tokeniser, the table build of the candidate generator (`prep`) and the flatten; the pair work is dealt out; the label
    exchange + merge (measured on the N ranks: the slowest rank's median) comes on top.  What the driver's scaling record is to
    be read against — a prediction, never a measurement."""
    if not one_gpu or "phases_ms" not in one_gpu:
        return None
    ph = one_gpu["phases_ms"]
    repl = ph["tokeniser"] + ph["prep"] + ph["flatten"]
    shard = ph["pairs"]
    exch = max((x["all_gather"] + x["merge_flatten"] for x in exchange["per_rank_ms"]), default=0.0) if exchange else 0.0
    t1 = repl + shard
    return {"replicated_ms": repl, "sharded_ms": shard, "exchange_merge_ms": exch, "n": world,
            "amdahl_ceiling": t1 / (repl + shard / world), "with_exchange": t1 / (repl + shard / world + exch)}


def compact_line(out: dict) -> dict:
    """THE line: the contract's keys and every headline scalar first, nested detail (numbers only) after; what a number means is
    said once, in DESIGN.md 6 — the explanatory strings of the full record (--detail FILE) are not repeated in every run."""
    def pick(d, keys):
        return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}

    def r(x, n=6):
        return float(f"{x:.{n}g}") if isinstance(x, float) else x

    host, cli, roof = out.get("t_cluster_host_ms") or {}, out.get("clusters_tsv") or {}, out["roofline"]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    line["metric"] = "genome-pair dists/sec"
    line.update(pick(out, ("contexts", "contexts_trial_ms_per_step", "ms_per_step_one_context", "value_one_context", "value_host_inclusive_pinned",
                           "value_host_inclusive")))
    if host:
        line["t_cluster_host_pinned_ms"] = host.get("pinned_buffer")
        line["t_cluster_host_fresh_pageable_ms"] = host.get("fresh_pageable_buffer")
        line["h2d_ms"] = host.get("h2d_ms")
    if out.get("sustained"):
        line["sustained_ms_per_step"] = out["sustained"]["ms_per_step"]
    line.update(pick(out, ("clusters_tsv_wall_s", "clusters_tsv_wall_median_s", "clusters_tsv_wall_ordinary_exit_s")))
    if cli:
        line["clusters_tsv_sha256_matches_reference"] = cli.get("sha256_matches_reference")
    if out.get("labels_only_any_ids"):
        line["labels_only_any_ids"] = pick(out["labels_only_any_ids"], ("ms_per_step", "value", "labels_equal"))
    line["value_resident_csr"] = out.get("value_resident_csr")
    line["resident_csr_ms_per_step"] = (out.get("resident_csr") or {}).get("ms_per_step")
    if out.get("all_pairs"):
        ap_ = out["all_pairs"]
        line["all_pairs"] = {"ms_per_step": ap_["ms_per_step"], "labels_equal_default_path": ap_["labels_equal_default_path"],
                             "prefilter_ms": ap_["phases_ms"]["ms_prefilter"], "prep_ms": ap_["phases_ms"]["ms_prep"],
                             "frac_of_issue_ceiling": ap_["roofline"]["frac"]}
    cfg = out["config"]
    line["config"] = pick(cfg, ("workload", "workload_key", "kernel_source_digest", "n_unique", "nnz", "n_vocab", "max_dist", "text_bytes",
                                "untimed_steps_before_warmup", "candidate_path", "collective_backend", "world_size", "n_edges_per_rank",
                                "predicted_speedup"))
    line["config"]["step"] = ("profile text resident in HBM -> tokeniser, vocabulary, CSR -> clustering kernels -> labels in HBM (DESIGN 6)" +
                              (f"; {out['contexts']} resident contexts take the steps in turn: steps of different batches overlap"
                               if out.get("contexts", 1) > 1 else ""))
    if cfg.get("step_phases"):
        line["config"]["step_phases"] = pick(cfg["step_phases"], ("per_rank_ms", "tokeniser_ms_every_rank", "payload_bytes_per_rank"))
    if cfg.get("one_gpu_same_workload"):
        line["config"]["one_gpu_same_workload"] = pick(cfg["one_gpu_same_workload"], ("ms_per_step", "steps", "value", "labels_equal_n_gpu_run"))
    sk = roof.get("second_kernel") or {}
    line["roofline"] = {**pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "algorithmic_bytes_per_launch")),
                        "second_kernel": pick(sk, ("kernel", "achieved", "frac", "traffic", "kernel_ms", "algorithmic_bytes_per_launch")),
                        "whole_step": pick(roof.get("whole_step") or {}, ("bytes", "GBps", "frac", "frac_one_context")),
                        "reference_equivalent": pick(roof.get("reference_equivalent") or {}, ("bytes", "equivalent_GBps"))}
    if "traffic" not in line["roofline"]:
        line["roofline"]["traffic"] = None
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {**pick(cb, ("value", "unit", "cores", "kind", "sampled", "sample", "seconds")),
                                **({"c_oracle_value": cb["c_oracle"]["value"]} if "c_oracle" in cb else {})}
    line["phases_ms"] = {"tokeniser": out["phases_ms"]["tokeniser"], "clustering": out["phases_ms"]["clustering"]}
    line["counters"] = out["counters"]
    line["result"] = out["result"]

    def rnd(o):
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, list):
            return [rnd(v) for v in o]
        return r(o)

    return rnd(line)


def kernel_source_digest():
    """sha256 over the device + host sources a libbfk.so is built from: a PMC file is only quoted as `traffic` when
    it was taken on exactly these sources (tools/profile_gpu.sh stamps it)."""
    h = hashlib.sha256()
    for f in ("bfk_kernels.hip", "bfk_host.cpp", "bfk_device.h", "bfk_text.hip", "bfk_sort.hip", "bfk_prep.hip"):
        h.update((ROOT / "breakfast_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


def measured_traffic(kernel: str, workload_key: str, launches=None):
    """HBM-side bytes per launch of `kernel` from the newest committed rocprofv3 PMC passes of this very build and
    workload (profiles/*_pmc_per_launch.json, separate --pmc passes).  bench.py cannot collect PMC counters on
    itself; a file taken on other sources is not quoted (returns None).  `launches` = {instantiation: launches per step}:
    the kernel runs as several instantiations / launches per step and the bytes of a STEP are wanted (the file holds the mean
    per launch of every instantiation)."""
    import glob

    want = kernel_source_digest()
    for f in sorted(glob.glob(str(ROOT / "profiles" / "*_pmc_per_launch.json")), reverse=True):
        try:
            pm = json.load(open(f))
        except (OSError, ValueError):
            continue
        meta = pm.get("_meta", {})
        if meta.get("source_digest") != want or meta.get("workload") != workload_key:
            continue
        if launches:
            per = {inst: [v for kk, v in pm.items() if kernel in kk and inst in kk and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
                   for inst in launches}
            if not all(per.values()):
                continue
            f_kib = sum(launches[i] * per[i][0]["FETCH_SIZE"] for i in launches)
            w_kib = sum(launches[i] * per[i][0]["WRITE_SIZE"] for i in launches)
            hit = [{"FETCH_SIZE": f_kib, "WRITE_SIZE": w_kib}]
        else:
            hit = [v for kk, v in pm.items() if kk.startswith(kernel) and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
        if hit:
            # FETCH_SIZE / WRITE_SIZE are KiB.  The guide's x2 note on FETCH_SIZE holds for 16-B-per-lane streaming
            # reads; both readings are given, `traffic` uses the uncorrected counter (lower bound) and says so.
            return {"bytes": (hit[0]["FETCH_SIZE"] + hit[0]["WRITE_SIZE"]) * 1024.0,
                    "fetch_bytes": hit[0]["FETCH_SIZE"] * 1024.0, "write_bytes": hit[0]["WRITE_SIZE"] * 1024.0,
                    "fetch_bytes_x2": 2 * hit[0]["FETCH_SIZE"] * 1024.0,
                    "source": f"{os.path.basename(f)} (commit {meta.get('commit', '?')}, sources {want}): FETCH_SIZE + "
                              f"WRITE_SIZE per launch, separate --pmc passes; counts Infinity-Cache hits too" +
                              (f"; summed over the launches of a step {launches}" if launches else "")}
    return None


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int) -> int:
    """Parent of an N-rank run: never touches the GPU; starts the ranks, relays rank 0's line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cli_wall(n_rows: int, d: int, indels: bool):
    """metric (2): input file -> clusters.tsv through the CLI in a FRESH process (imports, library load, context,
    module load all inside), five times with each way of ending the process: the CLI's default (a successful run flushes,
    joins its preload thread and ends by os._exit: no interpreter / HIP-runtime teardown) and an ordinary exit
    (BFK_FAST_EXIT=0).  Both walls go into the line."""
    from breakfast_amd import synth

    tmp = Path(tempfile.mkdtemp(prefix="bfk_bench_"))
    inp = tmp / "in.tsv"
    kw = dict(p_del=0.05, p_ins=0.01) if indels else {}
    synth.generate_tsv(inp, n_rows, **kw)
    args = ["--input-file", str(inp), "--max-dist", str(d)] + (["--no-skip-del", "--no-skip-ins"] if indels else [])
    res = {"rows": n_rows}
    digest = None
    for mode, fast in (("fast_exit", "1"), ("ordinary_exit", "0")):
        runs = []
        for i in range(5):
            out = tmp / f"out_{mode}{i}"
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "breakfast_amd", *args, "--outdir", str(out)], cwd=str(ROOT),
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env={**os.environ, "BFK_FAST_EXIT": fast})
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": r.stderr.decode(errors="replace")[-400:]}
            runs.append(round(dt, 4))
            dg = hashlib.sha256((out / "clusters.tsv").read_bytes()).hexdigest()
            if digest is not None and dg != digest:
                return {"error": "clusters.tsv differs between runs"}
            digest = dg
        res[mode] = {"seconds": min(runs), "median_s": sorted(runs)[len(runs) // 2], "runs_s": runs}
    res.update(seconds=res["fast_exit"]["seconds"], median_s=res["fast_exit"]["median_s"], clusters_sha256=digest,
               what="python -m breakfast_amd --input-file <tsv> --outdir <dir>, fresh process each (interpreter start, imports, "
                    "libbfk + HIP runtime load, context, kernels, writer); min and median of 5 per exit mode (the HIP runtime's "
                    "device open varies by 0.1-0.2 s between runs on one box); fast_exit = the CLI's default (os._exit after "
                    "flushing), ordinary_exit = BFK_FAST_EXIT=0")
    gold = json.loads((ROOT / "tests" / "golden" / "sha256.json").read_text())
    key = f"syn{n_rows}_d{d}"
    if not indels and key in gold:
        res["sha256_matches_reference"] = digest == gold[key]["clusters_sha256"]
    shutil.rmtree(tmp, ignore_errors=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=0, help="input sequences (default: 100000 on one GPU, 1000000 on several)")
    ap.add_argument("--max-dist", type=int, default=1)
    ap.add_argument("--indels", action="store_true", help="config 5 generator: p_del=0.05 p_ins=0.01, indels kept")
    ap.add_argument("--merge", default="allgather", choices=["allgather", "allreduce"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="only the timed steps + roofline (no host / CLI / CPU legs)")
    ap.add_argument("--contexts", type=int, default=0, help="one GPU: resident contexts (a stream and buffers each) that take the text steps in "
                                                            "turn, distributed.TextPipeline; 1 = every step behind the one before; 0 = "
                                                            "2, 3 and 4 are tried on 150 untimed steps each and the fastest is taken")
    ap.add_argument("--detail", default="", help="also write the FULL record (every leg, with its explanatory strings) to this file; "
                                                 "stdout is always the ONE compact line (<= 4 KB, scalars first)")
    ap.add_argument("--path", default="auto", choices=["auto", "allpairs", "join", "prefix"], help="candidate generator of the main leg")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0 and a.gpus > 1:
        raise SystemExit(self_launch(a.gpus))  # nothing above has touched the GPU
    world = max(world, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world

    # N = 1: BASELINE's metric workload (configs[2]: 100k profiles, max-dist 1).  N > 1: configs[3] — 1M profiles, max-dist 1,
    # row-sharded across the ranks — the same input for every N (strong scaling)
    n_rows = a.rows or (100000 if world == 1 else 1000000)
    d = a.max_dist
    full = world == 1 and not a.quick

    # metric (2) first: a fresh CLI process, before this process initialises the GPU
    cli = cli_wall(n_rows, d, a.indels) if (full and rank == 0) else None

    import torch
    import torch.distributed as dist

    from breakfast_amd import _lib
    from breakfast_amd.distributed import GpuEngine, ShardedClusterer
    from breakfast_amd.synth import generate_profiles

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    # rehearsal of the N > 1 path on a one-GPU box: BFK_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and the
    # label exchange on gloo (RCCL refuses two ranks on one device); the driver's runs never set it
    one_device = os.environ.get("BFK_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "gloo" if one_device else "nccl"
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    kw = dict(p_del=0.05, p_ins=0.01) if a.indels else {}
    rows = list(dict.fromkeys(generate_profiles(n_rows, **kw)))  # collapse_duplicates: unique profiles
    n_u = len(rows)
    # the hot path's input as the C-ABI takes it: the N_u profile strings as ONE byte buffer + int64 offsets
    buf, off = _lib.pack_rows(rows)
    T = len(buf)

    # ---- metric (1), host-inclusive (PCIe inside; never `value`): strings in HOST memory -> labels in HOST memory through
    # ONE C-ABI call, bfk_cluster_text
    host = None
    if full:
        lab_out = np.empty(n_u, dtype=np.int32)

        def one_shot(b):
            t0 = time.perf_counter()
            lab, _, nnz_, nv_ = _lib.cluster_text(b, off, " ", d, want_stats=False, labels_out=lab_out)
            return (time.perf_counter() - t0) * 1e3, lab, nnz_, nv_

        t_first, lab_first, nnz_t, nv_t = one_shot(buf)
        lab_first = lab_first.copy()
        reps = sorted(one_shot(buf)[0] for _ in range(9))
        fresh = []
        for _ in range(9):  # a pageable buffer the driver has never seen, a new one per call: its pages are pinned on the way
            b2 = bytes(bytearray(buf))
            fresh.append(one_shot(b2)[0])
            del b2
        pins = []
        for _ in range(4):  # the caller owns pinned memory (bfk_host_alloc) and builds its text there: four buffers in turn
            pb = _lib.PinnedBuffer(T)
            pb.view[:] = np.frombuffer(buf, dtype=np.uint8)
            pins.append(pb)
        one_shot(pins[0])
        pinned = sorted(one_shot(pins[i % 4])[0] for i in range(12))
        for pb in pins:
            pb.free()

        def old_route():
            t0 = time.perf_counter()
            ip, ix, _ = _lib.build_csr_bytes(buf, off, " ")
            t1 = time.perf_counter()
            _lib.cluster_csr(ip, ix, d)
            return (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3

        old = [old_route() for _ in range(3)]
        # the H2D copy by itself (HIP events around it, a context of its own)
        tctx = _lib.Context(0)
        tctx.set_profiling(True)
        h2d = []
        for _ in range(5):
            tctx.build_csr(buf, off, " ")
            h2d.append(tctx.text_stats()["ms_h2d"])
        tctx.close()
        med = lambda xs: sorted(xs)[len(xs) // 2]
        host = {"first_call": round(t_first, 3), "steady_same_pageable_buffer": round(reps[len(reps) // 2], 3),
                "fresh_pageable_buffer": round(med(fresh), 3), "fresh_pageable_buffer_runs": [round(x, 3) for x in fresh],
                "pinned_buffer": round(pinned[len(pinned) // 2], 3), "pinned_buffer_min": round(pinned[0], 3),
                "h2d_ms": round(med(h2d), 3), "h2d_GBps": T / (med(h2d) * 1e-3) / 1e9 if med(h2d) > 0 else None,
                "what": "N_u profile strings as one byte buffer + offsets (the C-ABI's input) in HOST memory -> bfk_cluster_text: "
                        "text + offsets H2D, device tokeniser + first-appearance vocabulary + CSR (bfk_text.hip), clustering "
                        "kernels, labels D2H into the caller's array.  first_call includes context creation, code-object load "
                        "and allocations; fresh_pageable_buffer = median of 9 calls, EACH on a newly allocated pageable copy of "
                        "the text (the driver pins its pages on the way); steady_same_pageable_buffer = 9 further calls on one "
                        "buffer (the driver keeps its pages pinned); pinned_buffer = the text built in memory from "
                        "bfk_host_alloc (four buffers in turn, median of 12)",
                "text_bytes": T, "nnz": nnz_t, "n_vocab": nv_t,
                "host_tokeniser_path": {"build_csr_ms": round(sorted(x[0] for x in old)[1], 3),
                                        "cluster_csr_ms": round(sorted(x[1] for x in old)[1], 3),
                                        "what": "round 2's route: bfk_build_csr on the host cores, then bfk_cluster_csr (CSR H2D)"}}

    indptr, indices, n_vocab = _lib.build_csr(rows, " ")  # (host tokeniser: the CSR of the resident-CSR leg and of the CPU baseline)
    nnz = int(indptr[-1])
    k = np.diff(indptr)

    # ---- the profile text RESIDENT IN HBM: enough distinct device copies that a step's text does not come out of the 256 MiB
    # Infinity Cache (the steps take them in turn), each in a buffer of bfk_text_device_bytes (the library pads the tail)
    need = _lib.text_device_bytes(T)
    n_copies = int(max(4, min(16, math.ceil(320e6 / max(T, 1)))))
    h_text = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
    d_texts = []
    for _ in range(n_copies):
        t_ = torch.empty(need, dtype=torch.uint8, device="cuda")
        t_[:T].copy_(h_text)
        d_texts.append(t_)
    d_off = torch.from_numpy(off).cuda()
    torch.cuda.synchronize()

    eng = GpuEngine(local_rank)
    eng.ctx.set_candidate_path(a.path)
    sc = ShardedClusterer(eng, rank, world, a.merge)
    step_no = [0]

    def text_step():
        t_ = d_texts[step_no[0] % n_copies]
        step_no[0] += 1
        return sc.step_text(t_.data_ptr(), T, d_off.data_ptr(), n_u, " ", d)

    def csr_step():
        return sc.step(d)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def warm(step, count, sync=None):
        # a candidate-queue overflow (dense inputs) is repaired inside sync() and grows the queue, so repeat until a
        # step runs clean: the timed steps must be complete single-pass steps
        for _ in range(6):
            for _ in range(max(count, 1)):
                step()
            again = int((sync or eng.sync)()["n_retry_slices"] != 0)
            if world > 1:  # every rank must run the same number of steps (each step holds a collective)
                tt = torch.tensor([again], dtype=torch.int32, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                again = int(tt.item())
            if not again:
                return

    def timed(step, count, sync=None):
        barrier()
        t0 = time.perf_counter()
        for _ in range(count):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        st_ = (sync or eng.sync)()  # checks the device-side overflow / error flags of the last timed step(s)
        if st_["n_retry_slices"] != 0:
            raise SystemExit("bench invalid: a timed step overflowed the candidate queue")
        return elapsed, st_

    def profiled_text(count=16):
        # per-phase durations of whole steps: HIP events on the launch stream, recorded inside libbfk around the tokeniser's
        # phases and around each phase of the clustering kernels, over a separate pass of the same steps
        eng.ctx.set_profiling(True)
        sts, tks = [], []
        for _ in range(count):
            text_step()
            sts.append(eng.sync())
            tks.append(eng.ctx.text_stats())
        eng.ctx.set_profiling(False)
        med = lambda xs: sorted(xs)[len(xs) // 2]
        st_ = dict(sts[-1])
        for kk in ("ms_prep", "ms_prefilter", "ms_verify", "ms_flatten", "ms_total"):
            st_[kk] = med([x[kk] for x in sts])
        tk_ = {kk: med([x[kk] for x in tks]) for kk in ("ms_scan", "ms_hash", "ms_head", "ms_ids", "ms_total")}
        return st_, tk_

    def profiled_csr(count=64):
        eng.ctx.set_profiling(True)
        for _ in range(min(count, 64)):
            sc.step(d)
        st_ = eng.sync()
        eng.ctx.set_profiling(False)
        return st_

    # ---- the contract's timed region: W warm-up steps, EXACTLY K steps; a step = profile strings in HBM -> labels in HBM
    # (the timed region may be a few milliseconds — the driver runs --steps 20 --warmup 5 — right after tens of seconds of host work
    # with the GPU idle: PRE_ROLL further untimed steps in front of the W warm-up steps bring clocks, caches and the context's
    # buffers to the state a stream of steps runs in; K = 20 varied by 8 % from run to run without them.  Disclosed in
    # config.untimed_steps_before_warmup; with the default K = 400 they change nothing.)
    PRE_ROLL = 300
    warm(text_step, PRE_ROLL)
    warm(text_step, a.warmup)
    elapsed, st_timed = timed(text_step, a.steps)
    labels = sc.labels[0][:n_u].cpu().numpy()
    ms_step = elapsed / a.steps * 1e3
    # ---- one GPU: the same steps dealt to `--contexts` resident contexts in turn (distributed.TextPipeline: a stream and buffers
    # each), so that steps of different batches run beside each other — a step's twelve dependent launches leave the chip partly
    # idle (the vocabulary hash's first two launches: 34 us with a few hundred waves).  Every step is still the complete hot path on
    # its batch; `value` is this whole-job throughput, the one-context figure (every step behind the one before) stands beside it.
    one_ctx = {"ms_per_step": ms_step, "value": n_u * (n_u - 1) / 2 * a.steps / elapsed}
    n_ctx = a.contexts if world == 1 else 1
    run_step, run_sync = text_step, None
    calib = None
    if n_ctx != 1:
        from breakfast_amd.distributed import TextPipeline

        def make_pipe(depth):
            pp = TextPipeline(local_rank, depth, a.path)
            ll = [torch.empty(max(n_u, 1), dtype=torch.int32, device="cuda") for _ in range(depth)]

            def st_():
                i = step_no[0]
                step_no[0] += 1
                pp.step_text(d_texts[i % n_copies].data_ptr(), T, d_off.data_ptr(), n_u, " ", d, ll[i % depth], inputs_ready=True, want_event=False)

            return pp, ll, st_

        if n_ctx <= 0:
            # How the streams of a pipeline share the chip depends on the hardware queues the HIP runtime happened to deal them
            # (two of a pipeline's streams on one queue run behind each other: three contexts gave 0.128 or 0.157 - 0.18 ms per step at
            # 100k rows depending on what else had created streams before): a few INSTANCES are tried on 150 untimed steps each, the
            # fastest one — that very instance, streams and all — serves the timed steps, the others are closed.
            calib, best = {}, None
            for tag, depth in (("2", 2), ("3", 3), ("4", 4), ("3b", 3)):
                pp, ll, st_ = make_pipe(depth)
                warm(st_, 100, pp.sync)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(150):
                    st_()
                torch.cuda.synchronize()
                calib[tag] = (time.perf_counter() - t0) / 150 * 1e3
                pp.sync()
                if best is None or calib[tag] < best[0]:
                    if best is not None:
                        best[2].close()
                    best = (calib[tag], depth, pp, ll, st_)
                else:
                    pp.close()
            _, n_ctx, pipe, p_labels, pipe_step = best
        else:
            pipe, p_labels, pipe_step = make_pipe(n_ctx)
        run_step, run_sync = pipe_step, pipe.sync
        warm(pipe_step, PRE_ROLL, pipe.sync)
        warm(pipe_step, a.warmup, pipe.sync)
        elapsed, st_timed = timed(pipe_step, a.steps, pipe.sync)
        ms_step = elapsed / a.steps * 1e3
        for pl_ in p_labels:
            if not np.array_equal(pl_[:n_u].cpu().numpy(), labels):
                raise SystemExit("bench invalid: a pipelined step's labels differ from the one-context step's")

    # ---- the same steps for >= 1 s (long enough for an outside sampler of GPU utilisation to see the device busy)
    sustained = None
    if world == 1 and not a.quick:
        n_sus = int(min(100000, max(a.steps, math.ceil(1.0 / max(ms_step * 1e-3, 1e-6)))))
        e2, _ = timed(run_step, n_sus, run_sync)
        sustained = {"steps": n_sus, "seconds": round(e2, 4), "ms_per_step": e2 / n_sus * 1e3}
    # ---- beside it, not `value`: the same steps with bfk_ctx_set_token_ids(ctx, 1) — a labels-only step at max-dist 1 may keep the
    # vocabulary table's slot numbers as column ids (an injective renaming of the reference's first-appearance ids: same distances,
    # same labels; three kernels and the first-occurrence walk fewer).  `value` stays the step that builds the reference's CSR.
    any_ids = None
    if world == 1 and not a.quick and d == 1:
        ctxs = [e.ctx for e in pipe.engines] if n_ctx != 1 else [eng.ctx]
        for c_ in ctxs:
            c_.set_token_ids(True)
        warm(run_step, 100, run_sync)
        e3, _ = timed(run_step, a.steps, run_sync)
        lab_any = [x[:n_u].cpu().numpy() for x in (p_labels if n_ctx != 1 else [sc.labels[0]])]
        for c_ in ctxs:
            c_.set_token_ids(False)
        warm(run_step, 20, run_sync)
        any_ids = {"ms_per_step": e3 / a.steps * 1e3, "value": n_u * (n_u - 1) / 2 * a.steps / e3, "contexts": n_ctx,
                   "labels_equal": bool(all(np.array_equal(x, labels) for x in lab_any)),
                   "what": "the timed steps again with bfk_ctx_set_token_ids(ctx, 1): column ids = vocabulary table slots (no "
                           "k_voc_count / k_voc_ids / k_tok_ids, no first-occurrence walk); not the reference's CSR, the same labels"}
    st, tk = profiled_text()
    edges_per_rank = None
    if world > 1:
        ne = torch.tensor([st["n_edges"]], dtype=torch.int64, device="cuda")
        allne = [torch.zeros_like(ne) for _ in range(world)]
        dist.all_gather(allne, ne)
        edges_per_rank = [int(x.item()) for x in allne]

    # ---- the second half of the step by itself: the CSR RESIDENT in HBM -> labels in HBM (what rounds 1-3 printed as `value`)
    sc.bind(indptr, indices)
    warm(csr_step, a.warmup)
    n_res = a.steps if a.quick else int(max(a.steps, min(20000, math.ceil(0.2 / max(ms_step * 1e-3, 1e-6)))))
    e_res, _ = timed(csr_step, n_res)
    lab_res = sc.labels[0][:n_u].cpu().numpy()
    st_res = profiled_csr()
    resident = {"ms_per_step": e_res / n_res * 1e3, "steps": n_res, "value": n_u * (n_u - 1) / 2 * n_res / e_res,
                "labels_equal_text_steps": bool(np.array_equal(lab_res, labels)),
                "phases_ms": {kk: st_res[kk] for kk in ("ms_prep", "ms_prefilter", "ms_verify", "ms_flatten", "ms_total")},
                "what": "the clustering kernels alone on a CSR that stays bound (no tokeniser, no bind): asynchronous launches, "
                        "the same barriers around the timed steps"}

    # ---- N > 1: where a step's time goes on this rank (torch events on the launch stream, a separate pass of 16 steps:
    # the rank's shard of the kernels, the label exchange, the merge) and — rank 0 alone, the others idle — the SAME
    # workload on one GPU in the same run
    exchange = one_gpu = None
    if world > 1 and a.merge == "allgather":
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(16)]
        with eng.run():  # (the engine's stream: where a step's kernels and collectives go)
            for e4 in evs:
                e4[0].record()
                sc.e.cluster_shard(d, rank, world, sc.local)
                e4[1].record()
                dist.all_gather_into_tensor(sc.gathered.view(-1), sc.local.view(-1))
                e4[2].record()
                sc.e.merge(sc.gathered, world, sc.labels)
                e4[3].record()
        torch.cuda.synchronize()
        eng.sync()
        med = lambda xs: sorted(xs)[len(xs) // 2]
        mine = [med([e4[i].elapsed_time(e4[i + 1]) for e4 in evs]) for i in range(3)]
        tt = torch.tensor(mine, dtype=torch.float64, device="cuda")
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        exchange = {"per_rank_ms": [{"shard_kernels": round(float(x[0]), 4), "all_gather": round(float(x[1]), 4),
                                     "merge_flatten": round(float(x[2]), 4)} for x in allt],
                    "tokeniser_ms_every_rank": tk,
                    "payload_bytes_per_rank": 4 * n_u,
                    "what": "median of 16 steps on the bound CSR, events on the launch stream of each rank; all_gather includes "
                            "waiting for the slowest rank's shard; the tokeniser (replicated: every rank builds the whole CSR) "
                            "is listed apart"}
    if world > 1:
        dist.barrier()
        if rank == 0:
            sc1 = ShardedClusterer(eng, 0, 1, a.merge)

            def text_step1():
                t_ = d_texts[step_no[0] % n_copies]
                step_no[0] += 1
                return sc1.step_text(t_.data_ptr(), T, d_off.data_ptr(), n_u, " ", d)

            for _ in range(3):
                for _ in range(max(3, min(a.warmup, 10))):
                    text_step1()
                if eng.sync()["n_retry_slices"] == 0:
                    break
            k1 = max(10, min(a.steps, 100))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k1):
                text_step1()
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t0
            st1 = eng.sync()
            lab1 = sc1.labels[0][:n_u].cpu().numpy()
            # where the one-GPU step's time goes (HIP events inside libbfk, 8 profiled steps): what every rank of an N-rank step
            # repeats (tokeniser, the join's table build, flatten) and what the ranks share (the pair work)
            eng.ctx.set_profiling(True)
            p_st, p_tk = [], []
            for _ in range(8):
                text_step1()
                p_st.append(eng.sync())
                p_tk.append(eng.ctx.text_stats())
            eng.ctx.set_profiling(False)
            med1 = lambda xs: sorted(xs)[len(xs) // 2]
            ph1 = {"tokeniser": med1([x["ms_total"] - x["ms_h2d"] for x in p_tk]), "prep": med1([x["ms_prep"] for x in p_st]),
                   "pairs": med1([x["ms_prefilter"] + x["ms_verify"] for x in p_st]), "flatten": med1([x["ms_flatten"] for x in p_st])}
            one_gpu = {"ms_per_step": e1 / k1 * 1e3, "steps": k1, "value": n_u * (n_u - 1) / 2 * k1 / e1, "phases_ms": ph1,
                       "labels_equal_n_gpu_run": bool(np.array_equal(lab1, labels)), "path": st1["path"],
                       "what": "the same steps (text in HBM -> labels) by rank 0's GPU alone (world 1), timed after the N-rank "
                               "steps while the other ranks wait: ms_per_step of this line / this = the speed-up the N GPUs gave"}
            sc.bind(indptr, indices)
        dist.barrier()

    # ---- first step after a bind of a CSR, ONE step host-timed incl. its launch and the sync
    cold = None
    if world == 1:
        ts = []
        for _ in range(5):
            sc.bind(indptr, indices)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sc.step(d)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            eng.sync()
        cold = sorted(ts)[len(ts) // 2]
        warm(csr_step, 5)

    # ---- the all-pairs kernels on the same CSR (north_star's design), when the main leg ran the join
    join = st.get("path", 0) == 1 and n_u > 0
    prefix = st.get("path", 0) == 2
    allpairs = None
    if full and join:
        eng.ctx.set_candidate_path("allpairs")
        warm(csr_step, a.warmup)
        n_ap = int(min(50000, max(200, math.ceil(0.2 / max(resident["ms_per_step"] * 1.5e-3, 1e-6)))))
        e3, _ = timed(csr_step, n_ap)
        lab_ap = sc.labels[0][:n_u].cpu().numpy()
        st_ap = profiled_csr()
        w = st_ap["sig_words"]
        t_pf = st_ap["ms_prefilter"] * 1e-3
        cyc_per_slot = w * (2.6 + 4.3) + 4.3  # measured issue cost per pair slot (tools/ubench/valu_rate.hip)
        ceiling = 64 * 1024 * 2.4e9 / cyc_per_slot
        allpairs = {
            "ms_per_step": e3 / n_ap * 1e3, "steps": n_ap, "value": n_u * (n_u - 1) / 2 * n_ap / e3,
            "what": "CSR resident -> labels with the band kernels forced (compare with value_resident_csr)",
            "labels_equal_default_path": bool(np.array_equal(lab_ap, labels)),
            "phases_ms": {kk: st_ap[kk] for kk in ("ms_prep", "ms_prefilter", "ms_verify", "ms_flatten", "ms_total")},
            "counters": {kk: st_ap[kk] for kk in ("pairs_in_band", "pairs_filtered", "n_candidates", "n_edges", "n_work_items")},
            "roofline": {"bound": "valu", "kernel": f"k_prefilter<W={w}>", "pair_slots": st_ap["pairs_filtered"],
                         "pair_slots_per_s": st_ap["pairs_filtered"] / t_pf if t_pf > 0 else None,
                         "measured_issue_ceiling_slots_per_s": ceiling,
                         "frac": st_ap["pairs_filtered"] / t_pf / ceiling if t_pf > 0 else None,
                         "kernel_ms": st_ap["ms_prefilter"],
                         "hbm": {"compulsory_bytes": 8 * n_u * w + 16 * n_u,
                                 "note": "sorted signatures + row records once; the tile loop re-reads them from L2"}},
        }
        eng.ctx.set_candidate_path(a.path)

    if rank == 0:
        b_ref, merged_pairs, resolved = reference_equivalent_bytes(k, d, nnz)
        t_dom = st["ms_prefilter"] * 1e-3
        w = st["sig_words"]
        k_mean = nnz / max(n_u, 1)
        if join:
            dom = "k_join"
            # what k_join must move: every token once (4 nnz), row extents (4 N), row hashes (8 N); its table / bitmap
            # lookups and the rows of the few matches are algorithm-internal traffic, not compulsory
            comp = (4 * nnz + 12 * n_u) / world
            comp_what = "4*nnz tokens + 4*N_u extents + 8*N_u row hashes, read once"
            clus_bytes = (8 * nnz + 44 * n_u) / world
            # the exact certificate of a match compares the row with its PARTNER row, which another wave — on another XCD's
            # L2 seven times out of eight — streamed: those tokens cross the fabric a second time whatever the layout
            join_partner_bytes = st["n_candidates"] * (4 * k_mean + 8)
        elif prefix:
            # prefix groups.  What the two long kernels must move:
            #   k_pgwalk16: 16 B (row record) per group member visited, the row heads (recs * 8 B position/count +
            #   16 B length/signature/offset per row), 24 B written per queued pair (the offset of the second row is left to
            #   the verify, which reads it for the candidates it checks)
            #   verify: the 24-byte queue record + the two parent words of every candidate, both rows' tokens (4 B each) of
            #   the candidates that are checked (all of them, or those not dropped as connected: counters.n_connected)
            recs = d + 2
            R = recs * n_u / world
            checked = st["n_candidates"] - st.get("n_connected", 0)
            # bytes each kernel must move when everything is read / written ONCE (what `frac` prices), and — labelled
            # apart, never part of `frac` — the bytes of its repeated visits (a group member is visited by every row in
            # front of it in the group; a row is read by every candidate it is part of)
            once_pgjoin = 24 * R + 16 * n_u / world + 24 * st["n_candidates"]
            once_verify = 32 * st["n_candidates"] + 8 * n_u + min(checked * 8 * k_mean, 4 * nnz + 4 * n_u)
            visits_pgjoin = st["pairs_filtered"] * 16
            visits_verify = checked * 8 * k_mean
            if st["ms_verify"] >= st["ms_prefilter"]:
                dom = "k_verify_connected" if st.get("n_connected", 0) else "k_verify"
                t_dom = st["ms_verify"] * 1e-3
                comp = once_verify
                visits = visits_verify
                comp_what = ("read / written once: 24-byte queue record + 8 B of row extents per candidate, the forest (4 B read + "
                             "4 B written per row), the tokens of the rows that are checked (at most the CSR once)")
                visits_what = "both rows' tokens (8*k_mean B) per candidate checked exactly (n_candidates - n_connected)"
            else:
                dom = "k_pgwalk16"
                comp = once_pgjoin
                visits = visits_pgjoin
                comp_what = ("read / written once: the group order (16 B record + 8 B position/count per record, (d+2) records "
                             "per row), 16 B of length / signature / offset per row, 24 B per queued pair")
                visits_what = "16 B per group member visited (a member is visited once by every row in front of it in its group)"
            # clustering kernels: tokens twice (k_pgfreq sample + k_pgkeys), 8 B per record out, three radix passes (8 B in and
            # out each), k_pgplace (8 B in, 16 B gathered, 24 B out), then the two kernels above, flatten
            clus_bytes = (4 * nnz + 20 * n_u) / world + 8 * R + 3 * 16 * R + 48 * R + once_pgjoin + once_verify + 8 * n_u
        else:
            dom = f"k_prefilter<W={w}>"
            comp = (4 * w * n_u + 16 * n_u) / world
            comp_what = "sorted first-level signatures (4*W*N_u) + row records (16*N_u), read once"
            clus_bytes = (4 * nnz + (84 + 16 * w) * n_u) / world
        achieved = comp / t_dom / 1e9 if t_dom > 0 else None
        wl_key = f"{n_rows}_d{d}{'_indels' if a.indels else ''}_{'join' if join else 'prefix' if prefix else 'allpairs'}"
        tr = measured_traffic("bfk::k_join" if join else ("void bfk::" + dom) if (prefix and dom.startswith("k_verify")) else
                              "void bfk::k_pgwalk16" if prefix else "void bfk::k_prefilter", wl_key) if world == 1 else None
        roof_cluster = {
            "bound": "hbm", "kernel": dom,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS if achieved else None,
            "traffic": tr["bytes"] if tr else None,
            "traffic_detail": tr,
            "algorithmic_bytes_per_launch": comp, "algorithmic_bytes_what": comp_what,
            **({"algorithmic_visits": {"bytes": visits, "what": visits_what,
                                       "GBps": visits / t_dom / 1e9 if t_dom > 0 else None,
                                       "note": "re-reads served by L2 / Infinity Cache: work the kernel does, not bytes it must "
                                               "move once; never part of `frac`"}} if prefix else {}),
            "kernel_ms": t_dom * 1e3,
        }
        if not join and not prefix:
            t_pf = t_dom
            cyc_per_slot = w * (2.6 + 4.3) + 4.3
            ceiling = 64 * 1024 * 2.4e9 / cyc_per_slot
            roof_cluster["valu"] = {"pair_slots": st["pairs_filtered"], "pair_slots_per_s": st["pairs_filtered"] / t_pf if t_pf > 0 else None,
                                    "measured_issue_ceiling_slots_per_s": ceiling,
                                    "frac_of_measured_ceiling": st["pairs_filtered"] / t_pf / ceiling if t_pf > 0 else None,
                                    "note": "the pair kernel is VALU-issue-bound (xor + popcount + min per 32-bit signature "
                                            "word per pair slot), its working set lives in L2; this is the binding roofline"}
        elif join:
            roof_cluster["lookups"] = {"count": st["pairs_filtered"], "per_s": st["pairs_filtered"] / t_dom if t_dom > 0 else None}
            roof_cluster["traffic_breakdown"] = {
                "matches": st["n_candidates"], "partner_row_bytes": join_partner_bytes,
                "algorithmic_bytes_incl_partner_rows": comp + join_partner_bytes,
                "frac_incl_partner_rows": (comp + join_partner_bytes) / t_dom / 1e9 / HBM_PEAK_GBS if t_dom > 0 else None,
                "note": "`frac` prices the kernel against the bytes read ONCE (tokens, extents, hashes); the partner rows of the "
                        "matches are a second trip of bytes already counted there (DESIGN 6c)"}
        else:
            roof_cluster["groups"] = {"members_visited": st["pairs_filtered"], "candidates": st["n_candidates"], "edges_checked": st["n_edges"],
                                      "dropped_as_connected": st.get("n_connected", 0)}
        # the tokeniser's dominant kernel: every text byte once + one 4-byte slot per token written
        t_hash = max(tk["ms_hash"], 1e-6)  # (the three launches of k_tok_hash: the event bracket around them)
        tok_bytes = T + 4 * nnz
        # (the hash runs as THREE launches per step — the head, a wave per 1 KiB window: k_tok_hash<1>; a sample and the rest, a wave
        # per 4 KiB unit: k_tok_hash<4> twice)
        tok_tr = measured_traffic("bfk::k_tok_hash", wl_key, {"<1>": 1, "<4>": 2}) if world == 1 else None
        roof_tok = {"bound": "hbm", "kernel": "k_tok_hash (its three launches of a step together)",
                    "achieved": tok_bytes / (t_hash * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": tok_bytes / (t_hash * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": tok_tr["bytes"] if tok_tr else None, "traffic_detail": tok_tr,
                    "algorithmic_bytes_per_launch": tok_bytes,
                    "algorithmic_bytes_what": "every text byte once + one 4-byte slot per token written",
                    "kernel_ms": t_hash}
        tok_all_bytes = 2 * T + 12 * nnz + 4 * n_u  # text twice (scan, hash) + slot written, read, id written per token + indptr
        step_bytes = tok_all_bytes + clus_bytes
        # the step's dominant kernel is the one with the longest measured duration
        roof, other = (roof_tok, roof_cluster) if t_hash >= t_dom * 1e3 else (roof_cluster, roof_tok)
        roof = dict(roof)
        roof["kernel_ms_source"] = ("HIP events on the launch stream around the kernel inside libbfk, median of 16 whole steps "
                                    "(includes the launch gap, ~3 us more than rocprofv3's kernel time: profiles/)")
        roof["second_kernel"] = other
        roof["whole_step"] = {"bytes": step_bytes, "GBps": step_bytes / (ms_step * 1e-3) / 1e9,
                              "frac": step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "frac_one_context": step_bytes / (one_ctx["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "note": "compulsory bytes of every kernel of the step (tokeniser: text twice + 12 B per token + "
                                      "indptr; clustering kernels as listed in DESIGN 8) / ms_per_step"}
        roof["reference_equivalent"] = {
            "bytes": b_ref / world, "pairs_in_reference_band": merged_pairs,
            "equivalent_GBps": b_ref / world / (ms_step * 1e-3) / 1e9,
            "note": "SURVEY 8(d) untiled operand-stream bytes, 4(k_i+k_j) per pair of the reference's length band + "
                    "8 B per pruned pair: what an untiled all-pairs merge touches.  These kernels never stream those "
                    "operands, so this is an algorithmic-speedup figure, not a bandwidth claim"}
        out = {
            "metric": "genome-pair dists/sec (pairs resolved/s, N_u(N_u-1)/2 per step) + clusters.tsv wall-clock",
            "value": resolved * a.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak" if world == 1 else "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"{n_rows} synthetic SARS-CoV-2 profiles (SURVEY App. A, seed 20240601"
                            f"{', indels kept' if a.indels else ''}), N_u={n_u} unique, k_mean={nnz / n_u:.1f}, "
                            f"max-dist {d}",
                "workload_key": wl_key, "kernel_source_digest": kernel_source_digest(),
                "n_unique": n_u, "nnz": nnz, "n_vocab": n_vocab, "max_dist": d, "text_bytes": T,
                "step": "the whole hot path of SURVEY 8(d) minus PCIe: N_u profile strings RESIDENT IN HBM (one byte buffer + int64 "
                        "offsets) -> separator scan, vocabulary table, first-appearance ids, CSR (bfk_text.hip) -> " +
                        ("clustering kernels behind them on device-resident counts, no wait in between (max-dist 1 up to 800k "
                         "rows; one wait for the token count / longest row otherwise) -> canonical labels in HBM; through "
                         "bfk_ctx_cluster_text_device" if world == 1 else
                         "one wait for the token count / longest row -> this rank's shard of the clustering kernels -> label "
                         "exchange -> canonical labels in HBM; through bfk_ctx_build_csr_device + bfk_ctx_cluster"),
                "untimed_steps_before_warmup": PRE_ROLL,
                "input": f"text resident in HBM, {n_copies} distinct device copies taken in turn ({n_copies * T / 1e6:.0f} MB: more "
                         "than the 256 MiB Infinity Cache holds); PCIe-inclusive figures: t_cluster_host_ms / "
                         "value_host_inclusive; the clustering kernels alone on a resident CSR: value_resident_csr",
                "candidate_path": ("variant join (k_jhash + k_join, DESIGN 6b)" if join else
                                   "prefix groups (k_pgkeys .. radix sort .. k_pgplace .. k_pgwalk16 .. k_verify_connected, DESIGN 6d/6e)" if prefix else
                                   "all-pairs band kernels (k_sig .. k_prefilter .. k_verify)"),
                "sharding": ("every rank tokenises the whole text (CSR replicated); " if world > 1 else "") +
                            (f"blocks of 8192 tokens (their table lookups) round-robin over {world} rank(s)" if join else
                             f"blocks of 64 rows (the walks of their groups) round-robin over {world} rank(s)" if prefix else
                             f"(k,f,g) cells of the sorted order round-robin over {world} rank(s)") +
                            (f", label merge {a.merge} ({sc.rounds} round(s))" if world > 1 else ""),
                **({"collective_backend": backend, "world_size": world, "n_edges_per_rank": edges_per_rank,
                    "step_phases": exchange, "one_gpu_same_workload": one_gpu,
                    "predicted_speedup": predicted_speedup(one_gpu, exchange, world)} if world > 1 else {}),
            },
            "roofline": roof,
            "phases_ms": {"tokeniser": tk, "clustering": {kk: st[kk] for kk in ("ms_prep", "ms_prefilter", "ms_verify", "ms_flatten", "ms_total")},
                          "note": "HIP events inside libbfk, median of 16 PROFILED steps (profiled steps are completed one at a "
                                  "time: the host waits for the tokeniser's counters between the halves, which the timed "
                                  "steps — device-driven at max-dist 1 — do not); ms_hash = the three k_tok_hash launches"},
            "counters": {kk: st[kk] for kk in ("pairs_in_band", "pairs_filtered", "n_candidates", "n_edges", "n_connected",
                                               "n_retry_slices", "n_work_items", "max_row_len")},
            "result": {"components": int(len(np.unique(labels))), "labels_crc": int(np.bitwise_xor.reduce(
                (labels.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(13)))},
            "contexts": n_ctx, "ms_per_step_one_context": one_ctx["ms_per_step"], "value_one_context": one_ctx["value"],
            **({"contexts_trial_ms_per_step": calib} if calib else {}),
            "value_resident_csr": resident["value"],
            "resident_csr": resident,
        }
        if sustained:
            out["sustained"] = sustained
        if any_ids:
            out["labels_only_any_ids"] = any_ids
        if cold is not None:
            out["resident_csr"]["ms_per_step_cold"] = cold
        if host:
            host["labels_equal_resident_path"] = bool(np.array_equal(lab_first, labels))
            out["t_cluster_host_ms"] = host
            out["value_host_inclusive"] = resolved / (host["fresh_pageable_buffer"] * 1e-3)
            out["value_host_inclusive_pinned"] = resolved / (host["pinned_buffer"] * 1e-3)
        if cli:
            out["clusters_tsv_wall_s"] = cli.get("seconds")
            out["clusters_tsv_wall_median_s"] = cli.get("median_s")
            out["clusters_tsv_wall_ordinary_exit_s"] = (cli.get("ordinary_exit") or {}).get("seconds")
            out["clusters_tsv"] = cli
        if allpairs:
            out["all_pairs"] = allpairs
        if full and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(indptr, indices, d, n_u)
        if a.detail:
            Path(a.detail).write_text(json.dumps(out, indent=1) + "\n")
        print(json.dumps(compact_line(out)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
