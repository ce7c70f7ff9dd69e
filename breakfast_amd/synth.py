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


# ---- other workload shapes (round 4): the dispatch thresholds of libbfk were fitted on generate_profiles() alone ------------
# Each family keeps the tree process and changes what the thresholds could be sensitive to: the row length, the shape of the
# phylogeny, the tokens' byte shape.  tools/family_matrix.py times every candidate generator on them
# (profiles/r04_family_matrix.txt); tests/test_gpu_parity.py compares each with the oracle.
AA_GENES = ["S", "N", "M", "E", "ORF1a", "ORF1b", "ORF3a", "ORF6", "ORF7a", "ORF7b", "ORF8", "ORF9b"]
AA_LETTERS = "ACDEFGHIKLMNPQRSTVWY"


def generate_family(family: str, n: int, seed: int = DEFAULT_SEED) -> list[str]:
    """`long`: 75-mutation root, 1 + Poisson(1.5) new mutations per generation, indels kept — rows of 100+ tokens (beyond the
    128 tokens the variant join decides by itself).  `star`: ten hub profiles; 6 % of all rows are a hub plus ONE mutation
    (a hub row has thousands of neighbours at distance 1), the others grow a tree as usual.  `aa`: amino-acid tokens
    (S:N501Y, ORF1a:T3255I, S:H69-: 7-13 bytes, most of them over the 7 bytes a vocabulary slot holds inline), 30-token root."""
    if family == "long":
        return generate_profiles(n, seed=seed, root_k=75, lam=1.5, p_del=0.05, p_ins=0.01)
    rng = np.random.default_rng(seed)
    if family == "star":
        base = generate_profiles(max(n // 10, 50), seed=seed + 1)
        hubs = [base[i] for i in rng.choice(len(base), size=10, replace=False)]
        refg = rng.integers(0, 4, size=29904)

        def newmut():
            pos = int(rng.integers(265, 29675))
            ref = int(refg[pos])
            return pos, f"{BASES[ref]}{pos}{BASES[(ref + int(rng.integers(1, 4))) % 4]}"

        def parse(row):
            return {int(t[1:-1]): t for t in row.split(" ")} if row else {}

        profiles = [parse(h) for h in hubs]
        hubs_d = list(profiles)
        while len(profiles) < n:
            if rng.random() < 0.06:
                prof = dict(hubs_d[int(rng.integers(0, len(hubs_d)))])
                k = 1
            else:
                prof = dict(profiles[int(rng.integers(0, len(profiles)))])
                k = 1 + int(rng.poisson(0.4))
            for _ in range(k):
                pos, tok = newmut()
                prof[pos] = tok
            profiles.append(prof)
        return [" ".join(p[k] for k in sorted(p)) for p in profiles[:n]]
    if family == "aa":
        lengths = {g: int(rng.integers(60, 4400)) for g in AA_GENES}

        def newmut():
            g = AA_GENES[int(rng.integers(0, len(AA_GENES)))]
            pos = int(rng.integers(1, lengths[g]))
            a = AA_LETTERS[int(rng.integers(0, 20))]
            r = rng.random()
            b = "-" if r < 0.05 else ("*" if r < 0.06 else AA_LETTERS[int(rng.integers(0, 20))])
            return (AA_GENES.index(g), pos), f"{g}:{a}{pos}{b}"

        profiles = []
        root = {}
        for _ in range(30):
            key, tok = newmut()
            root[key] = tok
        profiles.append(root)
        for i in range(1, n):
            prof = dict(profiles[int(rng.integers(0, i))])
            for _ in range(1 + int(rng.poisson(0.4))):
                key, tok = newmut()
                prof[key] = tok
            profiles.append(prof)
        return [" ".join(p[k] for k in sorted(p)) for p in profiles]
    raise ValueError(f"unknown family {family!r}")


def generate_tsv(path, n: int, **kw) -> None:
    """Write ``accession\\tdna_profile`` TSV (LF line ends, ids ``seq%07d``)."""
    rows = generate_profiles(n, **kw)
    with open(path, "w", newline="") as f:
        f.write("accession\tdna_profile\n")
        for i, r in enumerate(rows):
            f.write(f"seq{i:07d}\t{r}\n")
