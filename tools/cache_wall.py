"""Wall clock of the CLI with the side-car cache against a no-cache run (a 1M-row --input-cache run within
1.5x of the no-cache run).  usage (GPU box): python tools/cache_wall.py [rows]"""
import json
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from breakfast_amd.synth import generate_profiles  # noqa: E402


def run(args):
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "breakfast_amd", *args], cwd=str(ROOT), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr.decode()[-400:]
    return dt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    rows = generate_profiles(n)
    tmp = Path(tempfile.mkdtemp(prefix="bfk_cache_"))

    def write(path, lo, hi):
        with open(path, "w") as f:
            f.write("accession\tdna_profile\n")
            for i in range(lo, hi):
                f.write(f"seq{i:07d}\t{rows[i]}\n")

    write(tmp / "a.tsv", 0, n * 9 // 10)
    write(tmp / "b.tsv", 0, n)
    res = {"rows": n}
    run(["--input-file", str(tmp / "b.tsv"), "--outdir", str(tmp / "warm")])  # page cache, code object cache
    res["no_cache_s"] = min(run(["--input-file", str(tmp / "b.tsv"), "--outdir", str(tmp / f"f{i}")]) for i in range(5))
    res["write_sidecar_90pct_s"] = min(run(["--input-file", str(tmp / "a.tsv"), "--outdir", str(tmp / f"oa{i}"), "--output-cache", str(tmp / "c.bfkc")])
                                       for i in range(5))
    res["sidecar_bytes"] = (tmp / "c.bfkc").stat().st_size
    res["input_cache_10pct_new_s"] = min(run(["--input-file", str(tmp / "b.tsv"), "--outdir", str(tmp / f"ob{i}"), "--input-cache",
                                              str(tmp / "c.bfkc")]) for i in range(5))
    res["input_and_output_cache_s"] = min(run(["--input-file", str(tmp / "b.tsv"), "--outdir", str(tmp / f"oc{i}"), "--input-cache",
                                               str(tmp / "c.bfkc"), "--output-cache", str(tmp / "d.bfkc")]) for i in range(5))
    res["timing"] = "fresh process each, min of 5"
    res["same_clusters_as_no_cache"] = (tmp / "ob0" / "clusters.tsv").read_bytes() == (tmp / "f0" / "clusters.tsv").read_bytes()
    res["ratio_input_cache_vs_no_cache"] = round(res["input_cache_10pct_new_s"] / res["no_cache_s"], 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
