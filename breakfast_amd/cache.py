"""Incremental neighbour cache (reference src/breakfast/cache.py) — SURVEY.md 8(f2), a "next" row.

Not built yet in this round: the reference's gzip-pickle cache of neighbour lists needs the
GPU neighbour-list path (bfk_neighbours_csr) plus a lists->components kernel; until that lands the
CLI options fail loudly instead of silently recomputing.
"""


def cluster_with_cache(meta, indptr, indices, max_dist, input_cache, output_cache):
    raise NotImplementedError(
        "--input-cache/--output-cache are not supported by breakfast_amd yet (SURVEY.md 8 f2)")
