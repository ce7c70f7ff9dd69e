"""Synthetic SARS-CoV-2 mutation-profile generator.

Measurement tooling, not part of the clustering path: bench.py and the parity tests use it
to make the BASELINE.json workloads ("N synthetic profiles, ~40 SNPs each") reproducibly.
A tree process: profile 0 carries ``root_k`` random mutations, profile i copies a random
earlier profile and adds 1 + Poisson(lam) new ones.  The RNG call order below is the
contract (seed 20240601 reproduces the input sha256 values listed in SURVEY.md App. A).
"""

from __future__ import annotations

import numpy as np

BASES = "ACGT"
DEFAULT_SEED = 20240601


def generate_profiles(
    n: int,
    seed: int = DEFAULT_SEED,
    root_k: int = 24,
    lam: float = 0.4,
    p_del: float = 0.0,
    p_ins: float = 0.0,
) -> list[str]:
    """Return ``n`` profile strings (tokens joined by " " in ascending position)."""
    rng = np.random.default_rng(seed)
    refg = rng.integers(0, 4, size=29904)

    def newmut():
        r = rng.random()
        pos = int(rng.integers(265, 29675))
        ref = int(refg[pos])
        if r < p_del:
            return pos, f"del:{pos}:{int(rng.integers(1, 30))}"
        if r < p_del + p_ins:
            nb = int(rng.integers(1, 4))
            ins = rng.integers(0, 4, size=nb)
            return pos, f"{BASES[ref]}{pos}{BASES[ref]}" + "".join(BASES[int(b)] for b in ins)
        alt = (ref + int(rng.integers(1, 4))) % 4
        return pos, f"{BASES[ref]}{pos}{BASES[alt]}"

    profiles: list[dict[int, str]] = []
    root: dict[int, str] = {}
    for _ in range(root_k):
        pos, tok = newmut()
        root[pos] = tok
    profiles.append(root)
    for i in range(1, n):
        prof = dict(profiles[int(rng.integers(0, i))])
        for _ in range(1 + int(rng.poisson(lam))):
            pos, tok = newmut()
            prof[pos] = tok
        profiles.append(prof)
    return [" ".join(p[k] for k in sorted(p)) for p in profiles]


def generate_tsv(path, n: int, **kw) -> None:
    """Write ``accession\\tdna_profile`` TSV (LF line ends, ids ``seq%07d``)."""
    rows = generate_profiles(n, **kw)
    with open(path, "w", newline="") as f:
        f.write("accession\tdna_profile\n")
        for i, r in enumerate(rows):
            f.write(f"seq{i:07d}\t{r}\n")
