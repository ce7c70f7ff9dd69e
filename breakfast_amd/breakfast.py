"""Host-side mirror of the reference's function-level API (rki-mf1/breakfast src/breakfast/breakfast.py)
around the MI355X hot path.  Same names, argument meaning and error behaviour as the reference so a
caller (or a test written for the reference) can switch modules; the compute between the feature
strings and the component labels runs in libbfk.so (HIP, gfx950) through breakfast_amd._lib.

  read_input              breakfast.py:16-29     pandas reader + duplicate-id ValueError
  filter_features         breakfast.py:116-190   per-token classification, memoised per distinct token
  collapse_duplicates     breakfast.py:72-79
  sparse_feature_matrix   breakfast.py:193-215   -> bfk_build_csr_device (HIP tokeniser; bfk_build_csr, the host tokeniser of
                                                    the same contract, for what it declines and for GPU-less callers)
  get_neighbours_batch    breakfast.py:223-276   -> bfk_neighbours_csr (+ the band selection)
  cluster_features        breakfast.py:279-340   -> bfk_cluster_text (text -> labels in one call; with a cache or several GPUs:
                                                    the CSR from above + bfk_cluster_csr / the cache path)
  cluster_identical_features breakfast.py:343-364
  cluster                 breakfast.py:82-89
  write_output            breakfast.py:32-69

Differences that are unobservable in clusters.tsv (SURVEY.md 8a "contract"): intermediate cluster ids
are numbered by the smallest row of each component instead of networkx's traversal order.
"""

from __future__ import annotations

import re
import sys

import numpy as np
import pandas as pd

from . import _lib
from . import cache as ca


def read_input(input_file, sep, id_col, feature_col):
    meta = pd.read_table(
        input_file, sep=sep, usecols=[id_col, feature_col], dtype={id_col: str, feature_col: str}
    ).rename(columns={id_col: "id", feature_col: "feature"})
    dup = meta["id"][meta["id"].duplicated()].unique()
    if len(dup):
        raise ValueError("Duplicate sequence identifiers found: " + ", ".join(dup))
    meta["feature"] = meta["feature"].fillna("")
    print(f"Number of sequences: {meta.shape[0]}")
    return meta


# --- feature filtering ------------------------------------------------------------------------------
# (substitution, insertion, deletion) per --var-type; group 1 of a DNA substitution is its position.
_PATTERNS = {
    "covsonar_dna": (r"^[A-Z](\d+)[A-Z]$", r"^.*[A-Z][A-Z]$", r"^del:\d+:\d+$"),
    "covsonar_aa": (r"^[a-zA-Z0-9]+:[A-Z]\d+[A-Z]$", r"^[a-zA-Z0-9]+:[A-Z]\d+[A-Z][A-Z]+$",
                    r"^[a-zA-Z0-9]+:del:\d+:\d+$"),
    "nextclade_dna": (r"^[A-Z](\d+)[A-Z]$", r"^\d+:[A-Z]+$", r"^\d+(-\d+)?$"),
    "nextclade_aa": (r"^[a-zA-Z0-9]+:[A-Z]\d+[A-Z*]$", r"^$", r"^[a-zA-Z0-9]+:[A-Z]\d+-$"),
    "raw": None,
}
_KEEP, _DROP, _INVALID = 0, 1, 2


def filter_features(features, feature_sep, feature_type, skip_ins, skip_del, trim_start, trim_end,
                    reference_length):
    """Drop trimmed substitutions, skipped indels, unclassifiable and empty tokens (breakfast.py:116-190).
    Tokens repeat heavily between profiles, so each distinct token is classified once."""
    if not (skip_del or skip_ins or trim_start > 0 or trim_end > 0):
        return features  # nothing to filter: the input object is returned untouched (:128-129)
    if feature_type not in _PATTERNS:
        print(f"The feature type (--var-type) you chose is not supported: '{feature_type}'")
        sys.exit(1)
    pats = _PATTERNS[feature_type]
    if pats is not None:
        sub_re, ins_re, del_re = (re.compile(p) for p in pats)
    upper = reference_length - trim_end
    verdict: dict[str, int] = {}

    def classify(tok):
        v = _classify(tok)
        # an empty token that survives classification is dropped silently (:186-187)
        return _DROP if (v == _KEEP and not tok) else v

    def _classify(tok):
        if pats is None:
            return _KEEP
        m = sub_re.match(tok)
        if m:
            if m.lastindex:
                pos = int(m.group(1))
                if pos <= trim_start or pos >= upper:
                    return _DROP
            return _KEEP
        if ins_re.match(tok):
            return _DROP if skip_ins else _KEEP
        if del_re.match(tok):
            return _DROP if skip_del else _KEEP
        return _INVALID

    out = []
    for feature in features:
        kept = []
        for tok in feature.split(feature_sep):
            v = verdict.get(tok)
            if v is None:
                v = verdict[tok] = classify(tok)
            if v == _KEEP:
                kept.append(tok)
            elif v == _INVALID:
                print(f"Skipping invalid feature: '{tok}'")
        out.append(feature_sep.join(kept))
    return out


def collapse_duplicates(meta):
    print(f"Number of duplicates: {meta['feature'].duplicated().sum()}")
    meta_nodups = meta.groupby("feature", as_index=False, sort=False).agg(
        {"id": lambda x: tuple(x), "feature": "first"})
    print(f"Number of unique sequences: {meta_nodups.shape[0]}")
    return meta_nodups


def cluster(meta_nodups, sep2, max_dist, min_cluster_size, input_cache, output_cache, n_gpus=1):
    if max_dist == 0:
        return cluster_identical_features(meta_nodups, min_cluster_size)
    return cluster_features(meta_nodups, sep2, max_dist, min_cluster_size, input_cache, output_cache, n_gpus=n_gpus)


def _feature_csr(features, feature_sep):
    """sparse_feature_matrix's CSR (breakfast.py:193-215): tokenised on the GPU (bfk_build_csr_device) when there is one; the
    host tokeniser of the same library (bfk_build_csr — same contract, same CSR) takes what the device path declines
    (separators of over 16 bytes, 4 GiB of text) and the GPU-less callers of this function (the shell tests, the cache tools)."""
    features = list(features)
    if _lib.load().bfk_device_count() > 0 and len(feature_sep) > 0:
        try:
            return _lib.build_csr_device(features, feature_sep)
        except _lib.Unsupported:
            pass
    return _lib.build_csr(features, feature_sep)


def sparse_feature_matrix(features, feature_sep):
    """scipy CSR count matrix with first-appearance vocabulary, like the reference (incl. its ValueError
    when every row is empty: scipy cannot infer the shape)."""
    from scipy.sparse import csr_matrix

    indptr, indices, _ = _feature_csr(features, feature_sep)
    data = np.ones(len(indices), dtype=int)
    return csr_matrix((data, indices, indptr), dtype=int)


def _band(n_features_all, q, max_dist):
    return np.isclose(n_features_all, q, atol=max_dist)


def get_neighbours_batch(fmatrix, n_features_all, n_features_query, max_dist, select_ind=None):
    """List of index arrays, one per query row of the length band, each = rows of the band within
    max_dist (self included), in the reference's order.  fmatrix: scipy CSR (or (indptr, indices))."""
    if select_ind is not None and np.size(select_ind) == 0:
        return []
    indptr, indices = (fmatrix.indptr, fmatrix.indices) if hasattr(fmatrix, "indptr") else fmatrix
    nf = np.asarray(n_features_all).ravel()
    in_band = _band(nf, n_features_query, max_dist)
    rows = np.arange(len(nf)) if select_ind is None else np.asarray(select_ind, dtype=np.int64)
    q_rows = rows[in_band[rows]]
    if len(q_rows) == 0:
        return []
    nb_ptr, nb_idx = _lib.neighbours_csr(indptr, indices, max_dist, q_rows)
    out = []
    for s in range(len(q_rows)):
        l = nb_idx[nb_ptr[s]: nb_ptr[s + 1]].astype(np.int64)
        out.append(l[in_band[l]])
    return out


def _assign_cluster_ids(meta, labels, min_cluster_size):
    """breakfast.py:329-339 on canonical labels: a component counts the original sequences of its rows."""
    group = np.fromiter((len(t) for t in meta["id"]), dtype=np.int64, count=len(meta))
    uniq, inv = np.unique(labels, return_inverse=True)  # ascending smallest-row order
    size = np.bincount(inv, weights=group, minlength=len(uniq))
    keep = size >= min_cluster_size
    new_id = np.cumsum(keep) * keep
    cid = new_id[inv]
    col = pd.array(cid, dtype="Int64")
    col[cid == 0] = pd.NA
    meta["cluster_id"] = col.astype(object)
    n_clusters = int(keep.sum())
    print(f"Number of clusters found: {n_clusters}")
    return meta


def cluster_features(meta, feature_sep, max_dist, min_cluster_size, input_cache, output_cache, n_gpus=1):
    if input_cache is None and not output_cache and n_gpus == 1 and len(feature_sep) > 0:
        # no cache, one GPU: text -> labels in ONE call (bfk_cluster_text): the CSR is built on the device and stays there
        buf, off = _lib.pack_rows(list(meta["feature"]))
        indptr = np.zeros(len(off), dtype=np.int32)
        labels, _, nnz, _ = _lib.cluster_text(buf, off, feature_sep, max_dist, want_stats=False, indptr_out=indptr)
        if nnz == 0:
            raise ValueError("unable to infer matrix dimensions")  # (the reference dies here, :214)
        meta["n_features"] = np.diff(indptr).astype(np.int64)
        print("Imported cached results are not available. "
              "Distance matrix of complete dataset will be calculated.")
        print("Create graph and recover connected components")
        print("Save clusters")
        return _assign_cluster_ids(meta, labels, min_cluster_size)
    indptr, indices, _ = _feature_csr(meta["feature"], feature_sep)
    if len(indices) == 0:
        # the reference dies here: csr_matrix cannot infer the shape of an all-empty matrix (:214)
        raise ValueError("unable to infer matrix dimensions")
    meta["n_features"] = np.diff(indptr).astype(np.int64)

    if input_cache is not None or output_cache:
        from . import sidecar

        if sidecar.wants_sidecar(input_cache, output_cache):  # the cache's semantics on flat arrays (sidecar.py)
            if output_cache and not str(output_cache).endswith(sidecar.SUFFIX):
                raise ValueError(f"a side-car input cache can only be continued as a side-car: name the output cache *{sidecar.SUFFIX}")
            labels = sidecar.cluster_with_sidecar(_lib.hash_rows(list(meta["feature"])), indptr, indices, max_dist,
                                                  input_cache, output_cache)
        else:
            labels = ca.cluster_with_cache(meta, indptr, indices, max_dist, input_cache, output_cache)
    else:
        print("Imported cached results are not available. "
              "Distance matrix of complete dataset will be calculated.")
        labels, _ = _lib.cluster_csr(indptr, indices, max_dist, n_gpus)
    print("Create graph and recover connected components")
    print("Save clusters")
    return _assign_cluster_ids(meta, labels, min_cluster_size)


def cluster_identical_features(meta, min_cluster_size):
    print("Skip sparse matrix calculation since max-dist = 0")
    labels = np.arange(len(meta))
    return _assign_cluster_ids(meta, labels, min_cluster_size)


def write_output(meta_nodups, meta_original, outdir):
    """Expand id tuples, restore input order, renumber clusters by first appearance (breakfast.py:32-69)."""
    ids = meta_nodups["id"].tolist()
    cids = meta_nodups["cluster_id"].tolist()
    lens = np.fromiter((len(t) for t in ids), dtype=np.int64, count=len(ids))
    flat_ids = [s for t in ids for s in t]
    flat_cid = np.repeat(np.array([0 if pd.isna(c) else int(c) for c in cids], dtype=np.int64), lens)
    by_id = pd.Series(flat_cid, index=flat_ids)
    ordered = by_id.reindex(meta_original["id"])
    if ordered.shape[0] != meta_original.shape[0]:
        raise RuntimeError("Output row count differs from input row count")
    vals = ordered.to_numpy()
    if np.isnan(vals.astype(float)).any():
        raise RuntimeError("Output row count differs from input row count")
    vals = vals.astype(np.int64)
    nz = vals[vals != 0]
    first = pd.unique(nz)  # first-appearance order
    remap = np.zeros(int(vals.max()) + 1 if len(vals) else 1, dtype=np.int64)
    remap[first] = np.arange(1, len(first) + 1)
    new = remap[vals]
    col = pd.array(new, dtype="Int64")
    col[new == 0] = pd.NA
    out = pd.DataFrame({"id": meta_original["id"].to_numpy(), "cluster_id": col})
    outdir.mkdir(parents=True, exist_ok=True)
    out.to_csv(outdir / "clusters.tsv", sep="\t", index=False)
