/*
 * bfk.h — C-ABI of libbfk.so: the MI355X (gfx950) replacement for the clustering hot path of
 * rki-mf1/breakfast v0.4.6.  Plain pointers and sizes only; no torch / pandas / scipy types.
 *
 * The reference has no FFI: its boundary is a set of Python functions in src/breakfast/breakfast.py.
 * Each entry point below names the reference function (file:line) whose work it replaces; the
 * reference-side binding a maintainer would add is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - return 0 on success, <0 on error (BFK_E*); bfk_last_error() returns a thread-local message.
 *   - inputs are borrowed for the duration of the call only (device pointers bound with
 *     bfk_ctx_bind_csr_device must stay valid and unchanged until re-bound or the ctx is destroyed).
 *   - outputs are caller-allocated unless the name ends in _out with a ** type: those are
 *     library-allocated and released with bfk_free().
 *   - there is NO CPU fallback: every compute entry point needs a gfx950 device and fails with
 *     BFK_ENODEV otherwise.  bfk_build_csr (tokeniser + vocabulary) is host code by nature.
 *   - a bfk_ctx is not re-entrant: calls on one ctx must be serialised by the caller; different
 *     ctxs (one per GPU / per process rank) are independent.
 *   - labels are canonical: labels[i] = smallest row index in i's connected component.
 */
#ifndef BFK_H
#define BFK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFK_ABI_VERSION 3 /* 3: device-resident text entries, bfk_host_alloc / bfk_host_free; 2: bfk_stats grew (path, n_connected) */

#define BFK_OK 0
#define BFK_EARG -1      /* bad argument (NULL pointer, negative size, malformed indptr, ...) */
#define BFK_ENOMEM -2    /* host or device allocation failed */
#define BFK_ENODEV -3    /* no usable gfx950 device / HIP runtime error at init */
#define BFK_EHIP -4      /* HIP runtime error during the call (message has the hipError string) */
#define BFK_EOVERFLOW -5 /* an internal device buffer overflowed and could not be recovered */
#define BFK_ESTATE -6    /* call order violated (e.g. cluster before a CSR was bound) */
#define BFK_EUNSUPPORTED -7 /* bfk_table_*: the input needs the general (pandas) reader; nothing was done */
#define BFK_EIO -8       /* file could not be read / written */

typedef struct bfk_ctx bfk_ctx;

/* Counters and timings of the last bfk_ctx_cluster / bfk_ctx_merge_labels, filled by bfk_ctx_sync. */
typedef struct bfk_stats {
    int64_t n_rows;          /* N_u: rows of the bound CSR */
    int64_t nnz;             /* stored entries (multiset sizes summed) */
    int64_t pairs_resolved;  /* unordered pairs whose <=max_dist status this shard decided: N(N-1)/2 for 1 shard */
    int64_t pairs_in_band;   /* unordered pairs with |k_i-k_j| <= max_dist (need a set comparison), all shards */
    int64_t pairs_filtered;  /* pair slots the signature kernel evaluated in this shard (tile-padded); variant join: table lookups;
                              * prefix groups: group members visited */
    int64_t n_candidates;    /* pairs that passed both signature levels and were queued for the exact check (prefix groups,
                              * labels-only steps: a row with hundreds of neighbours may queue a pair more than once) */
    int64_t n_edges;         /* candidates checked exactly with distance <= max_dist: every edge of the graph when n_connected == 0 */
    int64_t n_retry_slices;  /* >0: the candidate queue overflowed and the run was redone in this many slices */
    int32_t max_row_len;     /* largest multiset size k */
    int32_t sig_words;       /* 32-bit words of the first-level signature used (1, 2 or 4) */
    int32_t n_work_items;    /* tiles (<= 64 sorted rows of one sort-key cell) of the pair kernel, all shards; 0 = variant join */
    int32_t profiled;        /* number of steps the ms fields below are averaged over (0 = profiling off) */
    float ms_prep;           /* row keys + signatures + cell histogram/ranks, scan, scatter; variant join: k_jhash */
    float ms_prefilter;      /* all-pairs signature kernel (the dominant kernel); variant join: k_join */
    float ms_verify;         /* exact check of the candidates + union-find hooks */
    float ms_flatten;        /* label flatten */
    float ms_total;          /* first launch to last launch completion */
    int32_t path;            /* candidate generator of the step: 0 band kernels, 1 variant join, 2 prefix groups (pairs_filtered = group members visited) */
    int32_t n_gpus_used;     /* devices the call ran on: bfk_cluster_csr(n_gpus > 1) declines inputs whose step is too short to win (1) */
    int64_t n_connected;     /* candidates dropped unchecked because their rows were in one component already (labels-only
                              * steps at max_dist >= 3; 0 with bfk_ctx_set_exact_edges(ctx, 1), BFK_EXACT_EDGES=1 or edge capture) */
} bfk_stats;

/* ---- library ------------------------------------------------------------------------------- */
int bfk_abi_version(void);
int bfk_device_count(void); /* number of visible gfx950 devices; 0 if none / no HIP runtime */
const char *bfk_last_error(void);
void bfk_free(void *p);

/* Pinned (page-locked) host memory for buffers a caller hands over — the profile text of bfk_cluster_text above all: a copy
 * from it runs at the PCIe rate from its first byte, while a pageable buffer the driver has not seen before is pinned page
 * by page on the way (100k profiles, 31 MB: ~0.6 ms against 1.5 ms; bench.py t_cluster_host_ms).  Needs a device.       */
int bfk_host_alloc(int64_t bytes, void **out);
int bfk_host_free(void *p);

/* Pay what a first call pays and that does not depend on the input — device context, stream, code-object load, and with
 * size hints (> 0) the workspace allocations — e.g. on a thread while the input is read (bfk_preload_start). */
int bfk_warmup(int device, int64_t rows_hint, int64_t nnz_hint);

/* ---- a1: vocabulary + CSR --------------------------------------------------------------------
 * Replaces sparse_feature_matrix(features, feature_sep)            src/breakfast/breakfast.py:193-215.
 * Row r is the byte range buf[row_off[r] .. row_off[r+1]).  Tokens are split on `sep` (non-overlapping,
 * left to right, like str.split), empty tokens are skipped (:208-209), ids are assigned by first
 * appearance (:210), repeated tokens are kept (the matrix is a count matrix).  indptr_out has
 * n_rows+1 entries and is caller-allocated; *indices_out is library-allocated (bfk_free).
 * sep_len == 0 -> BFK_EARG (Python raises ValueError("empty separator")).                          */
int bfk_build_csr(const char *buf, const int64_t *row_off, int64_t n_rows, const char *sep, int64_t sep_len,
                  int32_t *indptr_out, int32_t **indices_out, int64_t *nnz_out, int32_t *n_vocab_out);

/* The same contract computed ON THE DEVICE (bfk_text.hip: separator scan, vocabulary table keyed by hash + length + bytes
 * with atomicMin on the first byte offset, first-appearance ids by a prefix sum over the first occurrences), host outputs:
 * indptr / indices / n_vocab identical to bfk_build_csr's and to the reference's CSR.  A separator of several bytes (up to 16;
 * str.split takes any string, breakfast.py:204) is folded in the device copy of the text — every occurrence, leftmost and
 * never overlapping, never across a row boundary, becomes a run of a byte the text does not hold (k_sepfold) — and the kernels
 * run with that byte.  Text below 4 GiB, tokens below 64 KiB; anything else (a separator of over 16 bytes, a text that holds
 * every stand-in byte) -> BFK_EUNSUPPORTED and nothing done (bfk_build_csr takes those).  The entries that read text from the
 * CALLER's device memory (bfk_ctx_build_csr_device, bfk_ctx_cluster_text_device) do not write to it: one-byte separators there. */
int bfk_build_csr_device(const char *buf, const int64_t *row_off, int64_t n_rows, const char *sep, int64_t sep_len,
                         int32_t *indptr_out, int32_t **indices_out, int64_t *nnz_out, int32_t *n_vocab_out);

/* ---- a1..a8 one-shot: profile text in host memory -> labels in host memory -------------------------------------------
 * sparse_feature_matrix (:193-215) + the body of cluster_features (:287-326) in one call: the text is copied to the device
 * once, tokenised there (bfk_build_csr_device's kernels; the host tokeniser + an upload when they decline the input),
 * the CSR stays in HBM and is clustered like bfk_cluster_csr.  nnz_out / n_vocab_out / stats_out may be NULL; indptr_out
 * (int32[n_rows + 1], may be NULL) receives the CSR's row pointer — np.diff of it is the frame's n_features (:287).       */
int bfk_cluster_text(const char *buf, const int64_t *row_off, int64_t n_rows, const char *sep, int64_t sep_len,
                     int32_t max_dist, int32_t *labels_out, bfk_stats *stats_out, int64_t *nnz_out, int32_t *n_vocab_out,
                     int32_t *indptr_out);

/* ---- a2..a8 one-shot, host buffers ------------------------------------------------------------
 * Replaces the body of cluster_features between the CSR and the components:
 *   n_features / band loop (:287-319) -> get_neighbours_batch (:223-276) -> sklearn _sparse_manhattan
 *   -> _reduce_func (:226-228) -> _to_graph (:93-113) -> networkx connected_components (:325-326).
 * indices may be unsorted and may contain repeats (multiset rows).  labels_out: int32[n_rows].
 * n_gpus > 1: one context per device in THIS process, the work sharded over the devices (CSR replicated: uploaded to all
 * devices at once, a host thread each), label arrays copied to the first device (peer copies over xGMI) and merged there;
 * inputs whose single-GPU step is shorter than what the exchange costs run on one device (bfk_stats.n_gpus_used); the one-process-per-GPU form with RCCL collectives
 * goes through the ctx API below (breakfast_amd/distributed.py).  Same labels either way.
 * stats_out may be NULL.                                                                            */
int bfk_cluster_csr(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int32_t max_dist, int32_t n_gpus,
                    int32_t *labels_out, bfk_stats *stats_out);

/* ---- a4 canonicalised: neighbour lists ----------------------------------------------------------
 * Replaces what get_neighbours_batch (:223-276) computes, over all lengths at once: for every query row
 * (all rows, or select_ind[0..n_select) in that order) the ascending list of rows j with L1(i,j) <=
 * max_dist, self included.  CSR-style output: list s is nbr_indices[nbr_indptr[s] .. nbr_indptr[s+1]).
 * Both outputs are library-allocated (bfk_free).  select_ind == NULL means all rows.                 */
int bfk_neighbours_csr(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int32_t max_dist,
                       const int64_t *select_ind, int64_t n_select, int64_t **nbr_indptr_out,
                       int32_t **nbr_indices_out);

/* ---- cache path: components of neighbour lists ----------------------------------------------------
 * Replaces _to_graph/_to_edges + connected_components (breakfast.py:93-113, :325-326) when the neighbour
 * lists come (partly) from a cache (src/breakfast/cache.py): every list is united as a path
 * (a,b),(b,c),...; rows in no list stay singletons.  labels_out[i] = smallest row index of i's component. */
int bfk_labels_from_lists(int64_t n_rows, const int64_t *list_indptr, const int32_t *list_indices, int64_t n_lists,
                          int32_t *labels_out);

/* ---- resident context: device buffers, one ctx per GPU ---------------------------------------------
 * The same path with the CSR resident in HBM and launches enqueued on a caller-supplied HIP stream
 * (e.g. torch.cuda.current_stream().cuda_stream); used by bench.py and by the one-process-per-GPU
 * multi-GPU driver (breakfast_amd/distributed.py).                                                   */
int bfk_ctx_create(int device, bfk_ctx **ctx_out);
int bfk_ctx_destroy(bfk_ctx *ctx);
int bfk_ctx_set_stream(bfk_ctx *ctx, void *hip_stream);   /* NULL = the ctx's own stream */
int bfk_ctx_set_profiling(bfk_ctx *ctx, int32_t enable);  /* record HIP events between phases */

/* which candidate generator bfk_ctx_cluster uses: 0 = automatic (the default: variant join at max_dist 1 up to
 * 2M rows while join_pays() in bfk_host.cpp says so (rows of ~105 tokens turn to the band kernels from ~60k rows on, rows of ~43 never); prefix groups at max_dist 2..7 from the row counts of PG_MIN_ROWS() in bfk_host.cpp — the one place that
 * states them: 200k rows at max_dist 2 (more where the rows are longer than 50 tokens on average), 10k at 3, 4k at 4,
 * 2.5k at 5..7; the all-pairs band kernels otherwise and for
 * the shards of a multi-device max_dist 2 step), 1 = always the band kernels, 2 = the variant join wherever it
 * applies (max_dist 1), 3 = the prefix groups wherever they apply (max_dist 2..7).  Same labels either way; bench.py
 * times them against each other.                                                                            */
int bfk_ctx_set_candidate_path(bfk_ctx *ctx, int32_t mode);

/* copy a host CSR into ctx-owned device buffers (synchronous) */
int bfk_ctx_upload_csr(bfk_ctx *ctx, const int32_t *indptr, const int32_t *indices, int64_t n_rows);
/* borrow a CSR that already lives in device memory (int32 indptr[n_rows+1], int32 indices[nnz]);
 * sizes the workspace (synchronous: reads indptr[n_rows] and the longest row back) */
int bfk_ctx_bind_csr_device(bfk_ctx *ctx, const void *d_indptr, const void *d_indices, int64_t n_rows);

/* a1 on the device, CSR left resident and bound (bfk_build_csr_device without the download): buf / row_off are host
 * buffers, borrowed until the call returns (synchronous: the longest row and nnz come back for the workspace). */
int bfk_ctx_build_csr(bfk_ctx *ctx, const char *buf, const int64_t *row_off, int64_t n_rows, const char *sep, int64_t sep_len,
                      int64_t *nnz_out, int32_t *n_vocab_out);

/* ---- the same stages on text that is RESIDENT IN HBM (no PCIe inside the call) ------------------------------------------
 * d_text: device buffer of at least bfk_text_device_bytes(text_bytes) bytes whose first text_bytes hold the rows' bytes (rows
 * abut, as in bfk_build_csr); the library writes separator padding behind them.  d_row_off: device int64[n_rows + 1],
 * row_off[0] == 0, row_off[n_rows] == text_bytes, non-decreasing (checked by the kernels: BFK_EARG).
 *   bfk_ctx_build_csr_device     sparse_feature_matrix (:193-215) -> CSR resident and bound; waits once for the device (the
 *                                token count and the longest row size what follows)
 *   bfk_ctx_cluster_text_device  + the body of cluster_features (:287-326) -> canonical labels in d_labels_out (device
 *                                int32[n_rows]).  This is the step bench.py times: profile strings in HBM -> labels in HBM.
 *                                ASYNCHRONOUS: it returns with tokeniser and clustering kernels enqueued.  At max_dist 1 up to
 *                                800k rows (the device-driven form of the variant join; the join itself serves up to 2M short
 *                                rows, the host then sizes its launch after a wait) NOTHING waits in between — the clustering kernels read the token
 *                                count and the longest row from device memory — and up to four such steps may be open at once:
 *                                a caller that streams batches enqueues the next while the last one runs.  Every other entry
 *                                point — bfk_ctx_sync first of all — completes the open steps in order: a step whose input was
 *                                outside what the launch assumed (a row of more than 128 tokens, no token at all, a vocabulary
 *                                table that has to grow) is redone then, with every step enqueued behind it; an error of a
 *                                step (malformed offsets ...) is reported there and drops the steps behind it.  d_text,
 *                                d_row_off and d_labels_out therefore stay the library's until that bfk_ctx_sync returns.
 *                                Elsewhere (other max_dist, larger inputs) the call waits once between the halves.          */
int64_t bfk_text_device_bytes(int64_t text_bytes);
int bfk_ctx_build_csr_device(bfk_ctx *ctx, void *d_text, int64_t text_bytes, const void *d_row_off, int64_t n_rows,
                             const char *sep, int64_t sep_len, int64_t *nnz_out, int32_t *n_vocab_out);
int bfk_ctx_cluster_text_device(bfk_ctx *ctx, void *d_text, int64_t text_bytes, const void *d_row_off, int64_t n_rows,
                                const char *sep, int64_t sep_len, int32_t max_dist, void *d_labels_out);
/* copy the bound CSR to the host: indptr_out int32[n_rows + 1], indices_out int32[nnz] (synchronous) */
int bfk_ctx_download_csr(bfk_ctx *ctx, int32_t *indptr_out, int32_t *indices_out);
/* counters / phase times (with bfk_ctx_set_profiling) of the last bfk_ctx_build_csr / bfk_cluster_text */
typedef struct bfk_text_stats {
    int64_t text_bytes, n_rows, nnz;
    int64_t table_slots;     /* slots of the vocabulary table */
    int32_t n_vocab;
    int32_t table_growths;   /* times the table was enlarged 8x (kept with the context) */
    int32_t host_fallback;   /* 1: the device tokeniser declined the input (BFK_EUNSUPPORTED), the host tokeniser built the CSR */
    int32_t reserved_;
    float ms_h2d;            /* text + row offsets, host -> device */
    float ms_scan;           /* (k_tok_rowbits +) k_tok_scan */
    float ms_hash;           /* the three launches of k_tok_hash (first units, a sample spread over the text, the rest) */
    float ms_ids;            /* k_tok_rows + k_voc_count + k_voc_ids + k_tok_ids */
    float ms_total;          /* first copy to last kernel */
    float ms_head;           /* always 0 (round 3 had a k_tok_head kernel; kept for the layout of ABI 3) */
    float reserved2_;
    int64_t n_invalid;       /* filter mode: non-empty token occurrences that matched no pattern of the feature type (noted for the host) */
    int64_t n_empty;         /* filter mode: empty tokens inside the rows' spans */
} bfk_text_stats;
int bfk_ctx_text_stats(bfk_ctx *ctx, bfk_text_stats *out);

/* enqueue CSR -> labels for shard `shard` of `n_shards` (the cells of the sorted order are dealt round-robin; 0,1 = all).
 * d_labels_out: device int32[n_rows]; with n_shards > 1 these are the labels of the LOCAL forest.
 * Asynchronous: returns after the launches are enqueued.                                            */
int bfk_ctx_cluster(bfk_ctx *ctx, int32_t max_dist, int32_t shard, int32_t n_shards, void *d_labels_out);

/* enqueue the multi-GPU merge: d_gathered = int32[n_parts][n_rows] label arrays (all_gather output, or
 * 1 part holding an elementwise-min all-reduce); every (i, gathered[g][i]) is united into this ctx's
 * forest and d_labels_out is re-flattened.  With n_parts == n_shards of the last bfk_ctx_cluster, part g must be
 * the labels of shard g (all_gather order): the part of this ctx's own shard is skipped, it adds nothing.
 * d_changed (device int32, may be NULL) is set to 1 if any
 * label differs from d_gathered part 0 (fix-point test for the all-reduce(min) form).                */
int bfk_ctx_merge_labels(bfk_ctx *ctx, const void *d_gathered, int32_t n_parts, void *d_labels_out, void *d_changed);

/* wait for the stream, check device-side error flags, fill stats (may be NULL) */
int bfk_ctx_sync(bfk_ctx *ctx, bfk_stats *stats_out);

/* device <-> host helpers on the ctx stream (synchronous) */
int bfk_ctx_download(bfk_ctx *ctx, const void *d_src, void *h_dst, int64_t bytes);
int bfk_ctx_upload(bfk_ctx *ctx, const void *h_src, void *d_dst, int64_t bytes);
int bfk_ctx_device_alloc(bfk_ctx *ctx, int64_t bytes, void **d_out);
int bfk_ctx_device_free(bfk_ctx *ctx, void *d_ptr);

/* Steps that only deliver labels may drop a candidate pair whose rows are already in one component without computing its
 * distance (the reference computes every distance, breakfast.py:261-276, and then keeps only the components, :325-329;
 * the labels are the same).  enable = 1: every candidate is checked and bfk_stats.n_edges is the number of edges of the
 * graph (what the parity tests compare); 0: the default (pruning at max_dist >= 3). */
int bfk_ctx_set_exact_edges(bfk_ctx *ctx, int32_t enable);

/* Text steps that only deliver labels (bfk_ctx_cluster_text_device, bfk_cluster_text without n_vocab_out) at max_dist 1 do
 * not need the reference's first-appearance column numbers (sparse_feature_matrix hands out `len(vocabulary)` at a token's first
 * sight, breakfast.py:210; nothing the reference writes depends on the numbering): any injective renaming gives the same
 * distances and the same labels.  any_ids = 1: such steps stop at the vocabulary table's slot numbers — three kernels and the
 * first-occurrence walk fewer; bfk_ctx_download_csr then shows slots, bfk_text_stats.n_vocab is -1.  0: the default — every
 * bound CSR is the reference's CSR.  bfk_build_csr_device / bfk_ctx_build_csr / bfk_ctx_build_csr_device always number by
 * first appearance. */
int bfk_ctx_set_token_ids(bfk_ctx *ctx, int32_t any_ids);

/* edges of the last bfk_ctx_cluster run with edge capture enabled: (i<j) int32 pairs, library-allocated */
int bfk_ctx_set_edge_capture(bfk_ctx *ctx, int32_t enable);
int bfk_ctx_edges(bfk_ctx *ctx, int32_t **edges_out, int64_t *n_edges_out);


/* ---------------------------------------------------------------------------------------------------
 * Text front end and writer: the host stages either side of the GPU path, native.
 * A bfk_table holds the id and feature column of an input file (or of caller buffers) as byte ranges.
 *   bfk_table_open        replaces read_input            src/breakfast/breakfast.py:16-29
 *   bfk_table_prepare     replaces filter_features       :116-190
 *                                + collapse_duplicates   :72-79
 *                                + sparse_feature_matrix :193-215  (same CSR as bfk_build_csr on the unique rows)
 *   bfk_table_write       replaces write_output          :32-69
 * The reader restates read_table's default dialect as pandas' C tokeniser applies it — '"' opens a quoted
 * field only as the field's first byte, "" inside is one quote, separators and line breaks inside are
 * content, bytes behind the closing quote run on verbatim, NA strings are NaN quoted or not —, valid
 * UTF-8 with or without a byte-order mark, and its skipping of empty and blank-only lines.  It is strict
 * about the rest: what pandas would refuse or read in a way not restated here (a file that ends inside a
 * quoted field, a lone CR, NUL, invalid UTF-8, ragged rows, NA-valued or duplicate ids, duplicate /
 * missing column names, multi-byte separators, no data rows) returns BFK_EUNSUPPORTED and the caller uses the
 * reference's own pandas reader, which also raises the reference's exceptions.  For every input it accepts,
 * the result is byte-identical to that path (tests/test_frontend.py).
 * ------------------------------------------------------------------------------------------------- */
typedef struct bfk_table bfk_table;

#define BFK_VAR_COVSONAR_DNA 0
#define BFK_VAR_COVSONAR_AA 1
#define BFK_VAR_NEXTCLADE_DNA 2
#define BFK_VAR_NEXTCLADE_AA 3
#define BFK_VAR_RAW 4

typedef struct bfk_filter_opts {
    int32_t var_type;                                /* BFK_VAR_* (--var-type, breakfast.py:131-160) */
    int32_t skip_ins, skip_del;                      /* --skip-ins / --skip-del */
    int64_t trim_start, trim_end, reference_length;  /* --trim-start / --trim-end / --reference-length */
} bfk_filter_opts;

typedef struct bfk_prep_info {
    int64_t n_rows;     /* input sequences */
    int64_t n_unique;   /* distinct filtered feature strings (rows of the CSR), first-appearance order */
    int64_t nnz;        /* CSR entries */
    int64_t n_invalid;  /* token occurrences that matched no pattern ("Skipping invalid feature") */
    int32_t n_vocab;    /* distinct kept tokens */
    int32_t filtered;   /* 0: nothing to filter, features were taken verbatim (breakfast.py:128-129) */
} bfk_prep_info;

/* read `path`, locate the two columns by header name (read_input's usecols) */
int bfk_table_open(const char *path, const char *sep, int64_t sep_len, const char *id_col, const char *feature_col,
                   bfk_table **out);
/* same table from caller memory: ids / features are concatenated byte strings with N+1 offsets each
 * (copied).  No dialect checks apply except: no NUL, CR, LF or non-ASCII bytes.                       */
int bfk_table_from_buffers(const char *id_buf, const int64_t *id_off, const char *feat_buf, const int64_t *feat_off,
                           int64_t n_rows, bfk_table **out);
int64_t bfk_table_rows(const bfk_table *t);
void bfk_table_close(bfk_table *t);

/* filter + collapse + vocabulary/CSR in one pass over the feature bytes */
int bfk_table_prepare(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, bfk_prep_info *info_out);
/* views into the table, valid until the next prepare / close */
const int32_t *bfk_table_group(const bfk_table *t);    /* [n_rows]  unique-row index of every input row */
const int32_t *bfk_table_weight(const bfk_table *t);   /* [n_unique] input rows per unique row (len of the id tuple) */
const int32_t *bfk_table_indptr(const bfk_table *t);   /* [n_unique + 1] */
const int32_t *bfk_table_indices(const bfk_table *t);  /* [nnz] */
/* i-th invalid token occurrence, in the order the reference prints them (src/breakfast/breakfast.py:182-184) */
int bfk_table_invalid(const bfk_table *t, int64_t i, const char **tok_out, int64_t *len_out);
/* how many of them the table lists: info.n_invalid after the host prepare; after a device prepare info.n_invalid too, or 0 when
 * every invalid token is an empty one (info.n_invalid lines of ''); -1: not prepared */
int64_t bfk_table_invalid_count(const bfk_table *t);
/* filtered feature string of unique row u (sep2-joined kept tokens): library-allocated, bfk_free */
int bfk_table_feature(const bfk_table *t, int64_t u, char **str_out, int64_t *len_out);
/* id of input row r (view) */
int bfk_table_id(const bfk_table *t, int64_t r, const char **id_out, int64_t *len_out);
/* the same in bulk, for callers that need the strings themselves (the cache path: the reference's pickle stores the unique
 * rows' feature strings and id tuples, src/breakfast/cache.py:18-32): concatenated bytes + N+1 offsets, both library-allocated
 * (bfk_free).  bfk_table_features: the filtered feature strings of the n_unique rows; bfk_table_ids: the ids of the n_rows rows. */
int bfk_table_features(const bfk_table *t, char **buf_out, int64_t **off_out);
int bfk_table_ids(const bfk_table *t, char **buf_out, int64_t **off_out);

/* ---- side-car cache (breakfast_amd/sidecar.py): the reference's cache (src/breakfast/cache.py) matches rows of a new input
 * to cached rows by their feature STRING (:94-112); the side-car stores two 64-bit hashes of it instead.  bfk_hash_rows: rows
 * are byte ranges of one buffer, out[2 r .. 2 r + 1] the hashes of row r; bfk_table_feature_hashes: the same for the filtered
 * feature strings of a prepared table's unique rows (equal to bfk_hash_rows on bfk_table_features' output). */
int bfk_hash_rows(const char *buf, const int64_t *off, int64_t n_rows, uint64_t *out);
/* out[i] = the row j of b (hash pairs, rows distinct) equal to row i of a, or -1: cache.map_features (:94-112) on the hashes */
int bfk_match_hashes(const uint64_t *a, int64_t n_a, const uint64_t *b, int64_t n_b, int64_t *out);
int bfk_table_feature_hashes(const bfk_table *t, uint64_t *out);

/* ---- the CLI's whole tail in one call -----------------------------------------------------------------------------
 * bfk_preload_start: a native thread loads `libbfk_path` (libbfk.so: HIP runtime, code object) and runs bfk_warmup(device,
 * hints) while the caller parses its input with the bfk_table_* functions of libbfk_front.so (which has no HIP dependency).
 * bfk_table_cluster_write: labels of the prepared table's unique rows (max_dist 0: every unique row alone,
 * cluster_identical_features :343-364; else bfk_cluster_csr through the preloaded library), component sizes = summed
 * weights against min_cluster_size (:329-339), then bfk_table_write.  Replaces cluster + write_output (console.py:166-170). */
int bfk_preload_start(const char *libbfk_path, int device, int64_t rows_hint, int64_t nnz_hint);
int bfk_preload_wait(void);
/* joins the preload thread without reporting its result: every exit path that does not cluster (max-dist 0, a declined
 * input, an exception) calls it — the Python shell from an atexit hook — so the thread never outlives the HIP runtime */
void bfk_preload_join(void);
int bfk_table_cluster_write(const bfk_table *t, int32_t max_dist, int32_t min_cluster_size, int32_t n_gpus, const char *path,
                            int64_t *n_clusters_out);

/* ---- filter + collapse + CSR ON THE DEVICE (libbfk.so; bfk_text.hip's filter mode, bfk_prep.hip) --------------------------
 * bfk_table_prepare's contract with the work in HBM: the table's bytes cross PCIe once (from its first feature on), what lies
 * between the feature column's fields is blanked, every token occurrence is judged by the five feature grammars where it is
 * hashed (a dropped token never enters the vocabulary or the CSR), rows are collapsed by a hash of their identity (the kept-id
 * sequence, or the raw bytes when nothing is filtered: :128-129) with an exact comparison against the representative, unique
 * rows in first-appearance order (:72-79).  Tokens that match no pattern (the reference prints each: rows in input order, tokens
 * in row order, :182-184) are noted on the device as {offset, length}, put into that order on the host and handed out by
 * bfk_table_invalid (when every one of them is an EMPTY token the table lists nothing: info_out->n_invalid lines of '').
 * A token separator of several bytes (:164, str.split takes any string; up to 16) is folded on the device into a byte the table
 * does not hold (k_sepfold: leftmost, non-overlapping matches per row), the stages run with that byte.
 * BFK_EUNSUPPORTED — nothing done, the host stage takes the input — for: token separators of over 16 bytes or with a line break,
 * 4 GiB of text, more than 65 536 non-empty tokens that match no pattern, a FEATURE with non-ASCII bytes under a grammar (ids and
 * other columns may hold them).
 *   bfk_table_prepare_device        results installed in the table like bfk_table_prepare's (group, weight, CSR of the unique
 *                                   rows; the filtered feature STRINGS stay with the host stage: BFK_ESTATE from their accessors)
 *   bfk_table_cluster_write_device  the CLI's whole middle: the unique rows are clustered where the collapse left them (no CSR
 *                                   visits the host), component sizes against min_cluster_size (:329-339), bfk_table_write.
 *                                   max_dist > 0.  info_out->nnz == 0: nothing clustered or written (the reference cannot build
 *                                   its matrix from an all-empty input, :214: the caller raises).                            */
int bfk_table_prepare_device(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, bfk_prep_info *info_out);
int bfk_table_cluster_write_device(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                   int32_t min_cluster_size, const char *path, bfk_prep_info *info_out, int64_t *n_clusters_out);
/* the same through the library bfk_preload_start loaded (libbfk_front.so: no HIP dependency of its own) */
int bfk_table_pipeline_device(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                              int32_t min_cluster_size, const char *path, bfk_prep_info *info_out, int64_t *n_clusters_out);
/* the same with the clustering on n_gpus devices where several devices pay for the input (the rule of bfk_cluster_csr: max_dist 2
 * from 300k unique rows, max_dist >= 3 from 500k on >= 4 devices, max_dist 1 from 2M): filter + collapse + CSR on device 0, the
 * unique rows' CSR to every device, labels merged on device 0.  Replaces console.py:153-170 under `--gpus N` (round 5). */
int bfk_table_cluster_write_device_gpus(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                        int32_t min_cluster_size, int32_t n_gpus, const char *path, bfk_prep_info *info_out,
                                        int64_t *n_clusters_out);
int bfk_table_pipeline_device_gpus(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                   int32_t min_cluster_size, int32_t n_gpus, const char *path, bfk_prep_info *info_out,
                                   int64_t *n_clusters_out);
/* Side-car cache runs on the device stages (round 5; breakfast_amd/sidecar.py's container: the cache of src/breakfast/cache.py:18-32
 * on flat arrays).  Either path may be NULL (both NULL: bfk_table_cluster_write_device).  One device.
 *   cache_path (`--output-cache x.bfkc`): the same run with every edge recorded, and a side-car written from it — two 64-bit hashes
 *     of every unique row's feature string, computed on the device where the vocabulary is, and every row's neighbour list (itself
 *     + its neighbours, ascending: get_neighbours_batch's lists, breakfast.py:223-278) — marked EXACT (format 2).
 *   in_cache (`--input-cache x.bfkc`): a cache run yields the components of (cached lists, re-indexed) + (lists of the new rows)
 *     (breakfast.py:294-326), which is the no-cache run's result when the cached lists are exact and every cached row is still in
 *     the input.  That is checked (format 2, max_dist, the cached rows' hashes against this input's) and the run is the no-cache
 *     run — on this hardware cheaper than reading the lists back.  BFK_EUNSUPPORTED (nothing written) when a cached row is gone
 *     (its list still chains its neighbours, cache.py:51-71) or the cache is of format 1: bfk_neighbours_csr(select_ind) +
 *     bfk_labels_from_lists reuse the lists, as before.  A cache of another max_dist must not be passed (the reference does not
 *     use it either, cache.py:35-48).                                                                                         */
int bfk_table_cluster_write_device_cache(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                         int32_t min_cluster_size, const char *path, const char *in_cache, const char *cache_path,
                                         bfk_prep_info *info_out, int64_t *n_clusters_out);
int bfk_table_pipeline_device_cache(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                    int32_t min_cluster_size, const char *path, const char *in_cache, const char *cache_path,
                                    bfk_prep_info *info_out, int64_t *n_clusters_out);

/* write `path` = "id\tcluster_id" per input row in input order; cluster_of_unique[u] = any positive cluster
 * number or 0 for none; numbers are re-assigned 1.. by first appearance in input order (:51-60).       */
int bfk_table_write(const bfk_table *t, const char *path, const int32_t *cluster_of_unique, int64_t *n_clusters_out);

#ifdef __cplusplus
}
#endif
#endif /* BFK_H */
