"""Native end-to-end path of the CLI (SURVEY.md 8 f1 + f3): reader, filter + collapse + CSR and the
clusters.tsv writer run in libbfk (bfk_table_*), the clustering on the GPU; neither pandas nor numpy is imported, and the HIP library loads on a native thread
while the input is parsed (_front.preload).

Equivalent, byte for byte and print for print, to

    meta = read_input(...); meta["feature"] = filter_features(...); nodups = collapse_duplicates(meta)
    clustered = cluster(nodups, ...); write_output(clustered, meta, outdir)           (console.py:153-170)

for every input the native reader accepts; it declines (returns False, nothing printed or written) whenever
the result could depend on pandas' CSV dialect handling or a reference exception is due (duplicate ids,
missing columns ...).  Cache runs keep the reference's pickle format (pandas is imported for that alone) on the same
native host stages.
"""

from __future__ import annotations

import os

from . import _front


def cluster_ids(labels, weight, min_cluster_size):
    """breakfast.py:329-339 on canonical labels: a component counts the ORIGINAL sequences of its rows;
    -> (cluster number per unique row, 0 = none; number of clusters).  (numpy twin of what bfk_table_cluster_write does
    natively; used by the API path and the tests)"""
    import numpy as np

    # labels are row indices (the smallest row of the component): sizes by one bincount over them, no sort
    labels = np.asarray(labels, dtype=np.int64)
    size = np.bincount(labels, weights=weight, minlength=len(labels))
    keep = size >= min_cluster_size  # (only entries at component roots are populated: size 0 elsewhere)
    new_id = np.cumsum(keep) * keep
    return new_id[labels].astype(np.int32), int(keep.sum())


def run(input_file, sep, id_col, clust_col, var_type, sep2, skip_ins, skip_del, trim_start, trim_end,
        reference_length, max_dist, min_cluster_size, outdir, input_cache=None, output_cache=None, n_gpus=1) -> bool:
    if var_type not in _front.VAR_TYPES or len(sep2) == 0:
        return False
    if (input_cache is not None or output_cache) and max_dist != 0:
        if os.environ.get("BFK_DEVICE_PREP", "1") != "0" and os.environ.get("BFK_CACHE_REUSE", "0") != "1":
            done = _run_sidecar_on_device(input_file, sep, id_col, clust_col, var_type, sep2, skip_ins, skip_del, trim_start, trim_end,
                                          reference_length, max_dist, min_cluster_size, outdir, input_cache, output_cache)
            if done is not None:
                return done
        return _run_with_cache(input_file, sep, id_col, clust_col, var_type, sep2, skip_ins, skip_del, trim_start, trim_end,
                               reference_length, max_dist, min_cluster_size, outdir, input_cache, output_cache)
    if max_dist != 0:
        _front.preload(input_file)  # HIP runtime + context + code object on a native thread while the input is parsed
    try:
        table = _front.Table.open(input_file, sep, id_col, clust_col)
    except _front.Unsupported:
        return False
    if max_dist != 0 and os.environ.get("BFK_DEVICE_PREP", "1") != "0":
        # filter + collapse + CSR on the device, the unique rows clustered where they lie, the writer: ONE native call
        # (bfk_table_cluster_write_device).  It declines — nothing printed, nothing written — what only the host stage restates
        # (more than 65 536 tokens that match no pattern, a feature with non-ASCII bytes under a grammar, ...): the stages below take over.
        made = not outdir.exists()
        outdir.mkdir(parents=True, exist_ok=True)

        def drop_outdir():
            if made:
                try:
                    outdir.rmdir()
                except OSError:
                    pass

        try:
            info, n_clusters = table.pipeline_device(sep2, var_type, skip_ins, skip_del, trim_start, trim_end, reference_length,
                                                     max_dist, min_cluster_size, outdir / "clusters.tsv", n_gpus)
        except _front.Unsupported:
            info = None
        except _front.FrontError as e:
            # out of device memory for the prepare's extra buffers, a row table that overflowed: nothing has been printed or
            # written — the host stages below handled such inputs before there was a device prepare, and still do
            if e.code not in (_front.ENOMEM, _front.EHIP):
                drop_outdir()
                table.close()
                raise
            info = None
        if info is not None:
            n, nu = int(info.n_rows), int(info.n_unique)
            print(f"Number of sequences: {n}")
            listed = table.invalid_count()  # (0: every invalid token is an empty one)
            for i in range(int(info.n_invalid)):
                print(f"Skipping invalid feature: '{table.invalid(i) if listed else ''}'")
            print(f"Number of duplicates: {n - nu}")
            print(f"Number of unique sequences: {nu}")
            if info.nnz == 0:
                drop_outdir()
                table.close()
                # the reference dies here: csr_matrix cannot infer the shape of an all-empty matrix (:214)
                raise ValueError("unable to infer matrix dimensions")
            print("Imported cached results are not available. "
                  "Distance matrix of complete dataset will be calculated.")
            print("Create graph and recover connected components")
            print("Save clusters")
            print(f"Number of clusters found: {n_clusters}")
            table.close()
            return True
    try:
        try:
            info = table.prepare(sep2, var_type, skip_ins, skip_del, trim_start, trim_end, reference_length)
        except _front.Unsupported:
            return False
        n, nu = int(info.n_rows), int(info.n_unique)
        print(f"Number of sequences: {n}")
        for i in range(int(info.n_invalid)):
            print(f"Skipping invalid feature: '{table.invalid(i)}'")
        print(f"Number of duplicates: {n - nu}")
        print(f"Number of unique sequences: {nu}")
        outdir.mkdir(parents=True, exist_ok=True)
        if max_dist == 0:
            print("Skip sparse matrix calculation since max-dist = 0")
            n_clusters = table.cluster_write(0, min_cluster_size, outdir / "clusters.tsv")
        else:
            if info.nnz == 0:
                # the reference dies here: csr_matrix cannot infer the shape of an all-empty matrix (:214)
                raise ValueError("unable to infer matrix dimensions")
            print("Imported cached results are not available. "
                  "Distance matrix of complete dataset will be calculated.")
            n_clusters = table.cluster_write(max_dist, min_cluster_size, outdir / "clusters.tsv", n_gpus)
            print("Create graph and recover connected components")
            print("Save clusters")
        print(f"Number of clusters found: {n_clusters}")
        return True
    finally:
        table.close()


_SC_MAGICS = (b"BFKCACHE\x01\n", b"BFKCACHE\x02\n")  # (sidecar.MAGIC, MAGIC_EXACT / SUFFIX / _HEAD, without that module's numpy import)
_SC_SUFFIX = ".bfkc"


def _sidecar_head(path):
    """-> (max_dist, n_rows, n_lists, total, exact) of a side-car cache whose size is what its header says, else None (another
    format, or a damaged file: sidecar.load says what is wrong with it)"""
    import struct

    try:
        with open(path, "rb") as f:
            magic = f.read(10)
            if magic not in _SC_MAGICS:
                return None
            head = f.read(28)
            if len(head) != 28:
                return None
            d, n, n_lists, total = struct.unpack("<iqqq", head)
            if n < 0 or n_lists < 0 or total < 0 or os.fstat(f.fileno()).st_size != 10 + 28 + 16 * n + 8 * (n_lists + 1) + 4 * total:
                return None
            return d, n, n_lists, total, magic == _SC_MAGICS[1]
    except OSError:
        return None


def _run_sidecar_on_device(input_file, sep, id_col, clust_col, var_type, sep2, skip_ins, skip_del, trim_start, trim_end,
                           reference_length, max_dist, min_cluster_size, outdir, input_cache, output_cache):
    """Side-car cache runs on the device stages (round 5).  What a cache run computes — the components of (cached lists, re-indexed)
    + (lists of the new rows), cluster_features' cache branch, breakfast.py:294-326 — is what a run without a cache computes
    WHEN the cached lists are exact (sidecar.MAGIC_EXACT) and every cached row is still in the input (a row that is gone leaves
    its list behind, which still chains its neighbours: cache.py:51-71).  The cache exists to save the distance matrix, and
    here the whole clustering of a million rows is a millisecond, less than reading the cached lists back: so the native call
    checks just that (the cached rows' hashes against this input's, computed on the device) and the run is the no-cache run of
    `run` above; with an output cache every edge is recorded in that run and the side-car (feature hashes from the device,
    lists of ALL rows, exact) is written natively.
    -> True (done), False (the reader declined: pandas path), None (not a side-car run, a cache that is not exact or has lost
    a row or needs a closer look, or the device stages declined: `_run_with_cache` reuses the lists; nothing printed or
    written)."""
    out_sc = bool(output_cache) and str(output_cache).endswith(_SC_SUFFIX)
    if output_cache and not out_sc:
        return None
    head = None
    if input_cache is not None:
        head = _sidecar_head(input_cache)
        if head is None or (head[0] == max_dist and not head[4]):
            return None
    usable = head is not None and head[0] == max_dist  # (a cache of another max-dist is announced and not used: cache.py:35-48)
    if not out_sc and head is None:
        return None
    _front.preload(input_file)
    try:
        table = _front.Table.open(input_file, sep, id_col, clust_col)
    except _front.Unsupported:
        return False
    made = not outdir.exists()
    outdir.mkdir(parents=True, exist_ok=True)
    if out_sc:
        from pathlib import Path

        Path(output_cache).parent.mkdir(parents=True, exist_ok=True)

    def drop_outdir():
        if made:
            try:
                outdir.rmdir()
            except OSError:
                pass

    try:
        try:
            info, n_clusters = table.pipeline_device(sep2, var_type, skip_ins, skip_del, trim_start, trim_end, reference_length, max_dist,
                                                     min_cluster_size, outdir / "clusters.tsv", 1, output_cache if out_sc else None,
                                                     input_cache if usable else None)
        except _front.Unsupported:
            drop_outdir()
            return None
        except _front.FrontError as e:
            drop_outdir()
            if e.code in (_front.ENOMEM, _front.EHIP):
                return None
            raise
        n, nu = int(info.n_rows), int(info.n_unique)
        print(f"Number of sequences: {n}")
        listed = table.invalid_count()
        for i in range(int(info.n_invalid)):
            print(f"Skipping invalid feature: '{table.invalid(i) if listed else ''}'")
        print(f"Number of duplicates: {n - nu}")
        print(f"Number of unique sequences: {nu}")
        if info.nnz == 0:
            drop_outdir()
            raise ValueError("unable to infer matrix dimensions")  # (the reference dies here, :214)
        if head is not None:
            print("Import from side-car cache")
            if not usable:  # (cache.validate, cache.py:35-48)
                print("WARNING: Cached results were created using a differnt max-dist paramter")
                print(f"Current max-dist parameter: {max_dist}")
                print(f"Cached max-dist parameter: {head[0]}")
        if not usable:
            print("Imported cached results are not available. "
                  "Distance matrix of complete dataset will be calculated.")
        if out_sc:
            print("Export results as side-car cache")
        print("Create graph and recover connected components")
        print("Save clusters")
        print(f"Number of clusters found: {n_clusters}")
        return True
    finally:
        table.close()


def _run_with_cache(input_file, sep, id_col, clust_col, var_type, sep2, skip_ins, skip_del, trim_start, trim_end,
                    reference_length, max_dist, min_cluster_size, outdir, input_cache, output_cache) -> bool:
    """--input-cache / --output-cache on the native host stages: reader, filter, collapse, CSR and writer as above; the cache
    logic itself (cache.py: the reference's gzip-pickle of a DataFrame[id tuple, feature string] and a list of index arrays)
    needs the unique rows' strings and pandas for the pickle, nothing more.  Replaces the pandas reader, the per-token
    Python filter and the pandas writer of the mirror path for cache runs."""
    _front.preload(input_file)  # (cache runs too: the HIP start-up overlaps the parsing)
    import numpy as np

    from . import _lib, sidecar

    try:
        table = _lib.Table.open(input_file, sep, id_col, clust_col)
        info = table.prepare(sep2, var_type, skip_ins, skip_del, trim_start, trim_end, reference_length)
    except _lib.Unsupported:
        return False
    n, nu = int(info.n_rows), int(info.n_unique)
    print(f"Number of sequences: {n}")
    for i in range(int(info.n_invalid)):
        print(f"Skipping invalid feature: '{table.invalid(i)}'")
    print(f"Number of duplicates: {n - nu}")
    print(f"Number of unique sequences: {nu}")
    if info.nnz == 0:
        raise ValueError("unable to infer matrix dimensions")  # (the reference dies here, :214)
    indptr, indices = table.indptr, table.indices
    if sidecar.wants_sidecar(input_cache, output_cache):
        # side-car container (sidecar.py): the cache's semantics on flat arrays — rows matched by two 64-bit hashes of their
        # feature string, lists as one CSR; no pandas, no Python object per row
        if output_cache and not str(output_cache).endswith(sidecar.SUFFIX):
            raise ValueError(f"a side-car input cache can only be continued as a side-car: name the output cache *{sidecar.SUFFIX}")
        labels = sidecar.cluster_with_sidecar(table.feature_hashes(), indptr, indices, max_dist, input_cache, output_cache)
        print("Create graph and recover connected components")
        print("Save clusters")
        cid, n_clusters = cluster_ids(labels, table.weight, min_cluster_size)
        print(f"Number of clusters found: {n_clusters}")
        outdir.mkdir(parents=True, exist_ok=True)
        table.write(outdir / "clusters.tsv", cid)
        table.close()
        return True
    import pandas as pd

    from . import cache as ca

    # the frame cache.py works on: id = tuple of the accessions of a unique row (input order), feature = its filtered string
    ids = table.ids()
    order = np.argsort(table.group, kind="stable")
    bounds = np.concatenate([[0], np.cumsum(table.weight)]).tolist()
    order = order.tolist()
    meta = pd.DataFrame({"id": [tuple(ids[j] for j in order[bounds[u]: bounds[u + 1]]) for u in range(nu)],
                         "feature": table.features()})
    meta["n_features"] = np.diff(indptr).astype(np.int64)
    labels = ca.cluster_with_cache(meta, indptr, indices, max_dist, input_cache, output_cache)
    print("Create graph and recover connected components")
    print("Save clusters")
    cid, n_clusters = cluster_ids(labels, table.weight, min_cluster_size)
    print(f"Number of clusters found: {n_clusters}")
    outdir.mkdir(parents=True, exist_ok=True)
    table.write(outdir / "clusters.tsv", cid)
    table.close()
    return True
