"""CPU checks of bench.py's host logic: the N > 1 self-launch (ranks are children, the parent never touches the
GPU, the children's failure is the parent's exit code), the byte accounting, and that `roofline.traffic` is only ever
quoted from PMC passes taken on the very sources of the running build."""

import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def test_gpus_2_launches_two_ranks_and_relays_their_status():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--quick"],
                       capture_output=True, text=True, env=env, timeout=600)
    # no GPU here: every rank stops at "needs an MI355X" (no CPU fallback) and the parent exits non-zero
    assert r.returncode != 0
    # (the launcher tears the other rank down as soon as one has failed: at least one of them got to say it)
    assert r.stderr.count("bench.py needs an MI355X") >= 1 and "torch.distributed" in r.stderr


def test_reference_equivalent_bytes_against_brute_force():
    rng = np.random.default_rng(0)
    k = rng.integers(0, 9, size=60)
    nnz = int(k.sum())
    for d in (1, 3):
        b, merged, resolved = bench.reference_equivalent_bytes(k, d, nnz)
        mb = mp = 0
        for i in range(len(k)):
            for j in range(i + 1, len(k)):
                if abs(int(k[i]) - int(k[j])) <= d:
                    mp += 1
                    mb += 4 * (int(k[i]) + int(k[j]))
        assert merged == mp and resolved == len(k) * (len(k) - 1) / 2
        assert b == mb + 8 * (resolved - mp) + 4 * nnz + 8 * len(k)


def test_traffic_is_quoted_only_from_the_running_sources(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    (tmp_path / "breakfast_amd").symlink_to(ROOT / "breakfast_amd")
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    good = bench.kernel_source_digest()
    rec = {"k_join": {"FETCH_SIZE": 100.0, "WRITE_SIZE": 10.0}}
    (prof / "a_pmc_per_launch.json").write_text(json.dumps({**rec, "_meta": {"source_digest": "stale", "workload": "w"}}))
    assert bench.measured_traffic("k_join", "w") is None
    (prof / "b_pmc_per_launch.json").write_text(json.dumps({**rec, "_meta": {"source_digest": good, "workload": "other"}}))
    assert bench.measured_traffic("k_join", "w") is None
    (prof / "c_pmc_per_launch.json").write_text(json.dumps({**rec, "_meta": {"source_digest": good, "workload": "w", "commit": "x"}}))
    t = bench.measured_traffic("k_join", "w")
    assert t["bytes"] == 110 * 1024 and t["fetch_bytes_x2"] == 200 * 1024
