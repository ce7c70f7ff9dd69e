// bfk_host.cpp — the C-ABI of libbfk.so (include/bfk.h): context / workspace management around the
// HIP pipeline in bfk_kernels.hip, plus the host-side tokeniser + vocabulary (bfk_build_csr).
// There is no CPU compute fallback in this library: without a gfx950 device every compute entry
// point returns BFK_ENODEV.
#include "../../include/bfk.h"
#include "bfk_device.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace bfk;

int bfk_fail(int code, const std::string &msg);  // bfk_base.cpp (libbfk_front.so): sets the thread-local message
static int fail(int code, const std::string &msg) { return bfk_fail(code, msg); }

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(BFK_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

// (bfk_abi_version, bfk_last_error, bfk_free: bfk_base.cpp)

extern "C" int bfk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strstr(p.gcnArchName, "gfx950")) ok++;
    }
    return ok;
}

// (a1, bfk_build_csr — tokeniser + first-appearance vocabulary — lives in bfk_frontend.cpp with the other text stages)

// ================================================================================================
// context
// ================================================================================================
struct bfk_ctx {
    int device = 0;
    int n_cus = 256;
    hipStream_t own_stream = nullptr, stream = nullptr;
    bool profiling = false, edge_capture = false;
    int exact_edges = -1;  // -1: library default (BFK_EXACT_EDGES, else off), 0 / 1: bfk_ctx_set_exact_edges
    // a run has been enqueued whose outcome — candidate queue overflow, a give-up of the join or of the prefix groups: repaired by
    // bfk_ctx_sync, which redoes the run — has not been looked at yet.  Whatever would put that out of reach (a new bind, a run
    // with another max_dist or into other labels) settles it first (ctx_settle)
    bool unsettled = false;
    int tok_any_ids = -1;  // -1: library default (BFK_TOK_ANY_IDS, else off), 0 / 1: bfk_ctx_set_token_ids
    bool tok_any_now = false;
    static constexpr int EV_SLOTS = 64;  // ring of per-step event sets: up to 64 steps are averaged per sync
    hipEvent_t ev[EV_SLOTS][5] = {};
    bool ev_ready = false;
    int n_prof_calls = 0;
    // CSR
    const int *d_indptr = nullptr;
    const uint32_t *d_indices = nullptr;
    int *own_indptr = nullptr;
    uint32_t *own_indices = nullptr;
    int64_t own_n_cap = 0, own_nnz_cap = 0;
    int *own_labels = nullptr, *own_gather = nullptr;  // one-shot entry points: labels; n_gpus > 1: the gathered label arrays
    int64_t own_labels_cap = 0, own_gather_cap = 0;
    int64_t n = -1, nnz = 0;
    int kcap = 0;
    // workspace
    char *d_head = nullptr;  // Counters | hist[bins]
    int64_t bins_cap = 0;
    int *d_start3 = nullptr, *d_tile_slots = nullptr, *d_blk_stats = nullptr;
    unsigned long long *d_chain = nullptr;
    int *d_start3c = nullptr;
    int hist_copies = 1;
    int64_t hist_ints_cap = 0;
    int fb = KEY_BUCKETS, gb = KEY_BUCKETS, hb = 1;
    int64_t bins3 = 0;
    uint32_t *d_gkey = nullptr;
    int *d_gcnt = nullptr;
    int64_t gslots = 0, gkey_cap = 0, gcnt_cap = 0;
    int *d_parent = nullptr;
    int4 *d_srec = nullptr;
    uint32_t *d_sig1 = nullptr, *d_sigu1 = nullptr, *d_sigu2 = nullptr;
    bool need_zero = true;  // head (counters + histogram) must be memset before the next run
    bool ctr_dirty = true;  // the Counters block alone must be zeroed before the next run (first use, after a failure or a change of
                            // path): a join step after a bind zeroes only that and leaves the histogram — stale by the bind's new
                            // row lengths, need_zero — to the next step that uses it
    int64_t rows_cap = 0;
    int4 *d_tiles = nullptr;
    int64_t tile_cap = 0, tile_slots_cap = 0;
    int64_t last_tiles = 0;  // tile count seen by the last sync on this CSR (sizes the pair kernel's grid)
    int rows_per_lane = 1;
    int *d_rowkey = nullptr, *d_rowrank = nullptr;
    int4 *d_cand = nullptr;
    int2 *d_candk = nullptr;
    int64_t candk_cap = 0;
    int64_t cand_cap_total = 0, cand_cap_shard = 0;
    int2 *d_edges = nullptr;
    int64_t edge_cap = 0;
    uint32_t *d_edge_sel = nullptr;  // edge capture restricted to edges with a selected end (bfk_neighbours_csr(select_ind)); NULL: all
    int64_t edge_sel_cap = 0;
    bool edge_sel_on = false;
    int *d_small = nullptr;  // 4 ints scratch (maxlen, err, ...)
    // variant join (max_dist == 1): [table 0 | table 1 | bitmap 0 | bitmap 1 | row hashes]
    char *d_join = nullptr;
    int64_t join_bytes = 0, join_slots = 0, join_bits = 0;
    int join_parity = 0;     // table set of the next step (the other one is cleared by that step)
    bool join_clear = true;  // both sets must be cleared before the next join step
    bool join_off = false;   // this CSR made the join give up once: all-pairs from now on
    bool pg_dense_d2 = false;  // a max-dist 2 band step on this CSR queued > 8 candidates per row: prefix groups from now on
    bool pg_off = false;     // this CSR made the prefix groups give up once (groups too big): band kernels from now on
    int path_mode = 0;       // bfk_ctx_set_candidate_path: 0 auto, 1 all-pairs kernels, 2 variant join where it applies, 3 prefix groups
    // prefix-group path (max_dist >= 2, large inputs): sampled token counts, records, sorted records, group-order signatures, tiles
    uint32_t *pg_cnt = nullptr;
    int64_t n_short = 0;  // rows of at most 2 * PG_MAX_DIST tokens (bind time): they all meet in one group
    uint32_t *pg_keys = nullptr, *pg_keys_s = nullptr, *pg_keys_pm = nullptr;
    int max_tok = -1;  // largest token id of the bound CSR (found when the prefix groups are first considered for it)
    int *pg_rows = nullptr, *pg_rows_s = nullptr;
    int2 *pg_recpos = nullptr;
    void *pg_temp = nullptr;
    int4 *pg_srec = nullptr, *pg_rowinfo = nullptr;
    int64_t pg_rec_cap = 0, pg_temp_cap = 0, pg_rowinfo_cap = 0;
    // (shard, n_shards) of a synced join step on this CSR that left the queue of k_verify empty: the queued set is a
    // function of the CSR and the sharding only, so later steps skip that launch (k_flatten re-checks)
    // device tokeniser (bfk_ctx_build_csr, bfk_text.hip): text, bit arrays, vocabulary table, per-token scratch
    uint8_t *tk_text = nullptr;
    long long *tk_rowoff = nullptr;
    char *tk_zero = nullptr;  // [TokCounters | rowbits | firstbits]: zeroed by ONE memset per build
    uint32_t *tk_bits = nullptr, *tk_winbase = nullptr;
    TokSlot *tk_table = nullptr;
    int *tk_tabid = nullptr;  // first-appearance id per slot
    int64_t tk_tabid_cap = 0;
    // device prepare of a table (filter + collapse, bfk_prep.hip): span lengths, row hashes, the hash table, representatives,
    // prefix sums, group index / first row of the unique rows, their CSR, {totals, failure flags}
    int *pr_spanlen = nullptr, *pr_rep = nullptr, *pr_group = nullptr, *pr_first = nullptr, *pr_uindptr = nullptr, *pr_small = nullptr;
    uint2 *pr_inv = nullptr;        // device prepare: {byte offset, length} of the token occurrences that match no pattern (PREP_INV_CAP)
    uint32_t *pr_empties = nullptr; // device prepare: empty tokens per row
    int64_t pr_empties_cap = 0;
    unsigned long long *pr_rowhash = nullptr;
    PrepSlot *pr_table = nullptr;
    int2 *pr_val = nullptr, *pr_blk = nullptr;
    uint32_t *pr_uindices = nullptr;
    int64_t pr_rows_cap = 0, pr_table_cap = 0, pr_uidx_cap = 0, pr_blk_cap = 0;
    uint32_t *tk_slots = nullptr;  // filter mode: the slots of ALL tokens in text order (the CSR holds the kept ones)
    int64_t tk_slots_cap = 0;
    int64_t tk_text_cap = 0, tk_rowoff_cap = 0, tk_zero_cap = 0, tk_bits_cap = 0, tk_winbase_cap = 0, tk_table_cap = 0;
    int tk_grow = 0;  // how often the table was enlarged 8x for this context's inputs (kept: the next input is likely alike)
    hipEvent_t tk_ev[7] = {};
    bool tk_ev_ready = false;
    // the text goes up in pieces on a copy stream of its own while the pieces already there are tokenised (ctx_build_text)
    static constexpr int TK_PIECES = 8;
    hipStream_t tk_copy_stream = nullptr;
    hipEvent_t tk_piece_ev[TK_PIECES] = {}, tk_start_ev = nullptr;
    bfk_text_stats tk_stats{};
    bool tok_pending = false;  // the tokeniser's counters (d_small[8..15]) are to come back with the next bind's copy
    int tok_host[8] = {0};
    int *h_small_dev = nullptr;  // the same memory as the device addresses it (hipHostGetDevicePointer)
    int *h_small = nullptr;    // pinned host copies of d_small (16 ints each): slot 0 for binds the host waits for, slots 1 .. SPEC_RING for open text steps
    // Text steps whose bind the host has not completed yet: bfk_ctx_cluster_text_device enqueued the clustering kernels behind
    // the tokeniser with the token count and the longest row read ON THE DEVICE (JoinArgs::dyn) and returned.  Up to SPEC_RING
    // such steps may be open at once — a caller that streams batches enqueues the next step while the last one runs; every
    // other entry (bfk_ctx_sync first of all) completes them in order through ctx_enter: the counters of each step have
    // landed in its own pinned slot, and what a step's assumptions did not cover is redone — that step and, in order, every
    // later one (they wrote their results behind it).
    struct TokPlan {
        uint8_t *d_text = nullptr;
        const long long *d_rowoff = nullptr;
        int64_t base = 0, T = 0, n_rows = 0;
        char sep = ' ';
        bool strict = false;
        int n_pieces = 1;
        unsigned piece_blk[9] = {0};
        TokFilter flt{};                // filter_features on the device (flt.on), judged token by token where it is hashed
        const int *d_span_len = nullptr;  // rows that do not abut (a table's feature column): the rows' lengths
        uint2 *d_inv = nullptr;           // filter mode: where the invalid token occurrences are noted (inv_cap of them)
        uint32_t inv_cap = 0;
        uint32_t *d_row_empties = nullptr;  // filter mode: empty tokens per row
        bool any_ids = false;               // the CSR's column ids may be any injective renaming (TokArgs::any_ids)
    };
    struct SpecStep {
        TokPlan tp;
        int d = 0;
        void *labels = nullptr;
        hipEvent_t ev = nullptr;  // recorded behind the copy of the step's counters
    };
    static constexpr int SPEC_RING = 4;
    SpecStep spec_ring[SPEC_RING];
    int spec_head = 0, spec_count = 0;  // oldest open step, number of open steps
    int64_t tk_rowbits_clean = 0;       // words of tk_zero's row-start bits that are zero when the stream gets to the next build
    int *spec_host = nullptr;           // ... and the pinned slot its counters go to
    bool spec_enqueueing = false;       // ctx_enqueue is being called for a device-driven step (JoinArgs::dyn is set)
    // last run
    bool ran = false;
    int last_d = 0, last_w1 = 0, last_shards = 1;
    Plan plan{};  // the last enqueued run (bfk_ctx_sync re-runs it in slices after a queue overflow)
    bfk_stats stats{};
};

static int ctx_spec_finish(bfk_ctx *c);

// every entry point starts here; one that finds a text step whose bind is still open (spec_pending) completes it first
static int ctx_enter(bfk_ctx *c, bool finish_spec = true) {
    if (!c) return fail(BFK_EARG, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (finish_spec && c->spec_count > 0) return ctx_spec_finish(c);
    return BFK_OK;
}

// An entry point that binds another CSR, or enqueues a run whose repair would not be the pending run's: the pending run is
// completed first, repaired where it has to be (the soak of round 5 found it: a max-dist 2 text step whose candidate queue
// overflowed, followed on the same context by another step before any bfk_ctx_sync, kept its incomplete labels).  An error
// of the pending run surfaces here, once.
static int ctx_settle(bfk_ctx *c) {
    if (!c->unsettled) return BFK_OK;
    c->unsettled = false;
    return bfk_ctx_sync(c, nullptr);
}

template <typename T>
static int dev_realloc(T **p, int64_t *cap, int64_t want, double slack = 1.0) {
    if (want <= *cap && *p) return BFK_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    int64_t ncap = (int64_t)((double)want * slack) + 16;
    hipError_t e = hipMalloc((void **)p, (size_t)ncap * sizeof(T));
    if (e != hipSuccess) {
        *cap = 0;
        return fail(BFK_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    *cap = ncap;
    return BFK_OK;
}

extern "C" int bfk_ctx_create(int device, bfk_ctx **ctx_out) {
    if (!ctx_out) return fail(BFK_EARG, "null ctx_out");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(BFK_ENODEV, "no HIP device visible (libbfk has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(BFK_EARG, "device index out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail(BFK_ENODEV, std::string("device is ") + prop.gcnArchName + ", libbfk is built for gfx950 only");
    HIP_TRY(hipSetDevice(device));
    bfk_ctx *c = new bfk_ctx();
    c->device = device;
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(BFK_EHIP, "hipStreamCreate failed");
    }
    c->stream = c->own_stream;
    if (hipMalloc((void **)&c->d_small, 64) != hipSuccess || hipHostMalloc((void **)&c->h_small, 64 * (bfk_ctx::SPEC_RING + 1), hipHostMallocMapped) != hipSuccess ||
        hipMalloc((void **)&c->d_blk_stats, (size_t)VERIFY_GRID_MAX * 4 * sizeof(int)) != hipSuccess) {
        delete c;
        return fail(BFK_ENOMEM, "hipMalloc failed");
    }
    (void)hipMemset(c->d_blk_stats, 0, (size_t)VERIFY_GRID_MAX * 4 * sizeof(int));
    if (hipHostGetDevicePointer((void **)&c->h_small_dev, c->h_small, 0) != hipSuccess) c->h_small_dev = c->h_small;
    *ctx_out = c;
    return BFK_OK;
}

extern "C" int bfk_ctx_destroy(bfk_ctx *c) {
    if (!c) return BFK_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->spec_count = 0;
    for (auto &st : c->spec_ring)
        if (st.ev) (void)hipEventDestroy(st.ev);
    void *ptrs[] = {c->own_indptr, c->own_indices, c->d_head,       c->d_start3, c->d_gkey,  c->d_srec,  c->d_sigu1,
                    c->d_parent,   c->d_gcnt,      c->d_sig1,       c->d_tiles,  c->d_rowkey, c->d_rowrank,
                    c->d_tile_slots, c->d_cand,    c->d_candk,      c->d_edges,  c->d_small, c->d_sigu2, c->d_chain,
                    c->d_blk_stats, c->d_start3c, c->d_join, c->own_labels, c->own_gather, c->pg_keys, c->pg_keys_s, c->pg_rows,
                    c->pg_rows_s, c->pg_recpos, c->pg_temp, c->pg_srec, c->pg_cnt, c->pg_rowinfo, c->pg_keys_pm, c->tk_text, c->tk_rowoff,
                    c->tk_zero, c->tk_bits, c->tk_winbase, c->tk_table, c->tk_tabid, c->tk_slots, c->d_edge_sel, c->pr_spanlen, c->pr_rep, c->pr_group, c->pr_first,
                    c->pr_uindptr, c->pr_small, c->pr_rowhash, c->pr_table, c->pr_val, c->pr_blk, c->pr_uindices, c->pr_inv, c->pr_empties};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &slot : c->ev)
        for (auto &e : slot)
            if (e) (void)hipEventDestroy(e);
    for (auto &e : c->tk_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->tk_piece_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->tk_start_ev) (void)hipEventDestroy(c->tk_start_ev);
    if (c->tk_copy_stream) (void)hipStreamDestroy(c->tk_copy_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return BFK_OK;
}

extern "C" int bfk_ctx_set_stream(bfk_ctx *c, void *hip_stream) {
    if (int rc = ctx_enter(c)) return rc;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return BFK_OK;
}

extern "C" int bfk_ctx_set_profiling(bfk_ctx *c, int32_t enable) {
    if (int rc = ctx_enter(c)) return rc;
    if (enable && !c->ev_ready) {
        for (auto &slot : c->ev)
            for (auto &e : slot)
                if (hipEventCreate(&e) != hipSuccess) return fail(BFK_EHIP, "hipEventCreate failed");
        c->ev_ready = true;
    }
    c->profiling = enable != 0;
    c->n_prof_calls = 0;
    return BFK_OK;
}

extern "C" int bfk_ctx_set_candidate_path(bfk_ctx *c, int32_t mode) {
    if (!c) return fail(BFK_EARG, "null ctx");
    if (mode < 0 || mode > 3) return fail(BFK_EARG, "candidate path: 0 auto, 1 all-pairs, 2 variant join, 3 prefix groups");
    if (mode != c->path_mode) {  // the other path's per-step invariants (clean histogram / cleared table sets) are void
        c->need_zero = true;
        c->ctr_dirty = true;
        c->join_clear = true;
        c->last_tiles = 0;
    }
    c->path_mode = mode;
    return BFK_OK;
}

// Text steps at max_dist 1 that hand out labels only: the column ids of the CSR they bind need not be the reference's
// first-appearance numbers — any injective renaming gives the same distances, hence the same labels — so the tokeniser may
// stop at the vocabulary table's slot numbers (TokArgs::any_ids: no first-occurrence walk, no k_voc_count / k_voc_ids /
// k_tok_ids).  Off by default: bfk_ctx_download_csr after such a step shows slots, not the reference's ids.
extern "C" int bfk_ctx_set_token_ids(bfk_ctx *c, int32_t any_ids) {
    if (int rc = ctx_enter(c)) return rc;
    c->tok_any_ids = any_ids != 0;
    return BFK_OK;
}
static bool tok_any_ids_wanted(const bfk_ctx *c, int32_t max_dist) {
    if (max_dist != 1) return false;  // (the prefix groups of max_dist >= 2 count tokens in tables indexed by id)
    if (c->tok_any_ids >= 0) return c->tok_any_ids != 0;
    const char *e = getenv("BFK_TOK_ANY_IDS");
    return e && atoi(e) != 0;
}

extern "C" int bfk_ctx_set_exact_edges(bfk_ctx *c, int32_t enable) {
    if (!c) return fail(BFK_EARG, "null ctx");
    c->exact_edges = enable != 0;
    return BFK_OK;
}

extern "C" int bfk_ctx_set_edge_capture(bfk_ctx *c, int32_t enable) {
    if (!c) return fail(BFK_EARG, "null ctx");
    c->edge_capture = enable != 0;
    return BFK_OK;
}

static int ctx_size_cand(bfk_ctx *c, int64_t total) {
    if (total > c->cand_cap_total || !c->d_cand || !c->d_candk) {
        if (int rc = dev_realloc(&c->d_cand, &c->cand_cap_total, total)) return rc;
        c->candk_cap = 0;
        if (int rc = dev_realloc(&c->d_candk, &c->candk_cap, c->cand_cap_total)) return rc;
    }
    c->cand_cap_shard = c->cand_cap_total / CAND_SHARDS;
    return BFK_OK;
}

static int ctx_size_workspace(bfk_ctx *c, int d_hint) {
    const int64_t n = c->n, nnz = c->nnz;
    // (k,f,g) sort key: KEY_BUCKETS f- and g-buckets per row length, halved (g first) for very long rows
    c->fb = c->gb = KEY_BUCKETS;
    // third key h: only while the (d+1)^3 candidate ranges of a tile fit the 64 lanes (d <= 3) — with wider
    // bands it would only cut the cells smaller — and only for large inputs: 16x the cells cost ~20 us more in
    // k_sig / k_cells and make the tiles small (measured: 0.113 vs 0.079 ms at 100k rows, 0.299 vs 0.237 ms at
    // 400k, but 0.533 vs 0.697 ms at 1M rows, where the pair kernel is throughput-bound)
    c->hb = (n >= 600000 && (int64_t)(d_hint + 1) * (d_hint + 1) * (d_hint + 1) <= 64) ? KEY_BUCKETS : 1;
    if (const char *e = getenv("BFK_KEY_H")) c->hb = atoi(e) > 1 ? KEY_BUCKETS : 1;
    while (((int64_t)c->kcap + 1) * c->fb * c->gb * c->hb > KEY_MAX_BINS3 && c->hb > 1) c->hb >>= 1;
    while (((int64_t)c->kcap + 1) * c->fb * c->gb > KEY_MAX_BINS3 && c->gb > 1) c->gb >>= 1;
    while (((int64_t)c->kcap + 1) * c->fb * c->gb > KEY_MAX_BINS3 && c->fb > 1) c->fb >>= 1;
    const int64_t bins3 = ((int64_t)c->kcap + 1) * c->fb * c->gb * c->hb + 2;
    if (bins3 > (int64_t)INT32_MAX / 2) return fail(BFK_EARG, "row too long for the sort-key index");
    c->bins3 = bins3;
    // the cell histogram is kept in several copies (k_sig block b counts into copy b % copies: the returning
    // atomics of the hub cells serialise per address): 8 while the cells are few and the hubs big, 2 with the
    // third key (16x the cells, hubs ~5x smaller), 1 beyond 2M cells
    const int want_copies = bins3 <= (1 << 16) ? 8 : (bins3 <= (1 << 21) ? 2 : 1);
    if (bins3 > c->bins_cap || (int64_t)want_copies * bins3 > c->hist_ints_cap || !c->d_head) {
        for (void *q : {(void *)c->d_head, (void *)c->d_start3, (void *)c->d_start3c, (void *)c->d_chain})
            if (q) (void)hipFree(q);
        c->d_head = nullptr;
        c->d_start3 = nullptr;
        c->d_start3c = nullptr;
        c->d_chain = nullptr;
        c->bins_cap = 0;
        c->hist_ints_cap = (int64_t)want_copies * bins3;
        size_t head = sizeof(Counters) + (size_t)c->hist_ints_cap * 4;
        if (hipMalloc((void **)&c->d_head, head) != hipSuccess || hipMalloc((void **)&c->d_start3, (size_t)bins3 * 4) != hipSuccess ||
            hipMalloc((void **)&c->d_start3c, (size_t)c->hist_ints_cap * 4) != hipSuccess ||
            hipMalloc((void **)&c->d_chain, (size_t)(bins3 / 1024 + 2) * 8) != hipSuccess)
            return fail(BFK_ENOMEM, "hipMalloc(histogram) failed");
        c->bins_cap = bins3;
        c->need_zero = true;
        c->ctr_dirty = true;
    }
    if (c->hist_copies != want_copies) c->need_zero = true;  // another stride: the copies must be clean
    c->hist_copies = want_copies;
    (void)nnz;
    {   // k_verify_long: pairs of very long rows that do not fit the 64 KiB LDS table use a global scratch table
        c->gslots = 0;
        if (2 * (int64_t)c->kcap * 4 > (int64_t)LONG_TABLE * 3) {
            int64_t tsz = 1024;
            while (tsz * 3 < 2 * (int64_t)c->kcap * 4) tsz <<= 1;
            c->gslots = tsz;
            if (int rc = dev_realloc(&c->d_gkey, &c->gkey_cap, tsz * LONG_BLOCKS, 1.0)) return rc;
            if (int rc = dev_realloc(&c->d_gcnt, &c->gcnt_cap, tsz * LONG_BLOCKS, 1.0)) return rc;
        }
    }
    if (n + SIG_PAD_ROWS > c->rows_cap) {
        int64_t cap = 0, want = n + SIG_PAD_ROWS;
        int rc = 0;
        cap = 0; rc |= dev_realloc(&c->d_srec, &cap, want);
        cap = 0; rc |= dev_realloc(&c->d_parent, &cap, want);
        cap = 0; rc |= dev_realloc(&c->d_rowkey, &cap, want);
        cap = 0; rc |= dev_realloc(&c->d_rowrank, &cap, want);
        cap = 0; rc |= dev_realloc(&c->d_sig1, &cap, want * 4);
        cap = 0; rc |= dev_realloc(&c->d_sigu1, &cap, want * 4);
        cap = 0; rc |= dev_realloc(&c->d_sigu2, &cap, want * SIG2_WORDS);
        if (rc) return BFK_ENOMEM;
        c->rows_cap = want;
        // padded signature rows are read (never trusted): give them defined contents once
        HIP_TRY(hipMemsetAsync(c->d_sig1, 0xFF, (size_t)(want * 4 + 16) * 4, c->stream));
    }
    {   // rows per lane of the prefilter tile: cells hold ~N/(lengths x f x g buckets) rows, a tile never crosses
        // a cell, so small inputs use 64-row tiles and only very large ones 256-row tiles
        c->rows_per_lane = n < (2 << 20) ? 1 : (n < (8 << 20) ? 2 : 4);
        if (const char *e = getenv("BFK_PF_ROWS")) c->rows_per_lane = atoi(e) >= 4 ? 4 : (atoi(e) >= 2 ? 2 : 1);
        // tiles: one per non-empty cell plus one per full tile of rows
        const int64_t want = std::min<int64_t>(bins3, n) + n / (64 * c->rows_per_lane) + 16;
        if (int rc = dev_realloc(&c->d_tiles, &c->tile_cap, want, 1.0)) return rc;
        if (int rc = dev_realloc(&c->d_tile_slots, &c->tile_slots_cap, c->tile_cap * PF_WAVES_MAX, 1.0)) return rc;
    }
    {
        // 64N candidate slots in all up to 1M rows (1.5 GB there), 32N beyond 2M (d=1 needs ~0.7N; a dense d=5 graph 18N at
        // 1M rows and 36N at 30k with the labels-only walk, which lets repeated pairs of hub rows through); an overflow is
        // repaired by bfk_ctx_sync (sliced re-run) and doubles the queue for the following runs
        const int64_t slots = std::max<int64_t>(32 * n, std::min<int64_t>(64 * n, 64ll << 20));
        int64_t want = std::max<int64_t>(1024, slots / CAND_SHARDS + 256);
        if (int rc = ctx_size_cand(c, want * CAND_SHARDS)) return rc;
    }
    return BFK_OK;
}

static int ctx_after_bind(bfk_ctx *c) {
    c->last_tiles = 0;  // another CSR: no tile count known yet
    // nnz, the longest row — and the tokeniser's counters, when it built this CSR (d_small[8..15]) — come back in ONE
    // small copy per bind (every small device-to-host copy costs ~20 us of driver time; set-up, not part of a timed step)
    int h[16] = {0};
    HIP_TRY(hipMemsetAsync(c->d_small, 0, 32, c->stream));
    if (c->n > 0) {
        if (int e = launch_maxlen(c->d_indptr, (int)c->n, c->d_small, c->stream))
            return fail(BFK_EHIP, std::string("k_maxlen launch: ") + hipGetErrorString((hipError_t)e));
    }
    if (c->n > 0 || c->tok_pending) {
        HIP_TRY(hipMemcpyAsync(h, c->d_small, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    memcpy(c->tok_host, h + 8, sizeof c->tok_host);
    c->tok_pending = false;
    if (c->n > 0 && (h[1] || h[4] != 0 || h[3] < 0)) return fail(BFK_EARG, "malformed CSR: indptr must start at 0 and be non-decreasing");
    c->nnz = c->n > 0 ? h[3] : 0;
    c->kcap = h[0];
    c->n_short = h[2];
    c->max_tok = -1;
    c->ran = false;
    c->need_zero = true;  // bins are laid out by kcap
    // (the variant join's table sets keep their state over a bind: the set the last step filled is cleared by the next step's
    // k_jhash, the other one is clean — whatever CSR comes next; ctx_size_join asks for a clearing when their size changes)
    c->join_off = false;
    c->pg_off = false;
    c->pg_dense_d2 = false;
    return ctx_size_workspace(c, 0);
}

extern "C" int bfk_ctx_bind_csr_device(bfk_ctx *c, const void *d_indptr, const void *d_indices, int64_t n_rows) {
    if (int rc = ctx_enter(c)) return rc;
    if (int rc = ctx_settle(c)) return rc;
    if (n_rows < 0 || n_rows > (int64_t)INT32_MAX - 2 * SIG_PAD_ROWS) return fail(BFK_EARG, "n_rows out of range");
    if (n_rows > 0 && !d_indptr) return fail(BFK_EARG, "null indptr");
    c->d_indptr = (const int *)d_indptr;
    c->d_indices = (const uint32_t *)d_indices;
    c->n = n_rows;
    return ctx_after_bind(c);
}

extern "C" int bfk_ctx_upload_csr(bfk_ctx *c, const int32_t *indptr, const int32_t *indices, int64_t n_rows) {
    if (int rc = ctx_enter(c)) return rc;
    if (int rc = ctx_settle(c)) return rc;
    if (n_rows < 0 || n_rows > (int64_t)INT32_MAX - 2 * SIG_PAD_ROWS) return fail(BFK_EARG, "n_rows out of range");
    if (!indptr) return fail(BFK_EARG, "null indptr");
    const int64_t nnz = indptr[n_rows];
    if (indptr[0] != 0 || nnz < 0 || (nnz > 0 && !indices)) return fail(BFK_EARG, "malformed CSR");
    if (int rc = dev_realloc(&c->own_indptr, &c->own_n_cap, n_rows + 1)) return rc;
    if (int rc = dev_realloc(&c->own_indices, &c->own_nnz_cap, nnz + 1)) return rc;
    HIP_TRY(hipMemcpyAsync(c->own_indptr, indptr, (size_t)(n_rows + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (nnz > 0) HIP_TRY(hipMemcpyAsync(c->own_indices, indices, (size_t)nnz * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->d_indptr = c->own_indptr;
    c->d_indices = c->own_indices;
    c->n = n_rows;
    return ctx_after_bind(c);
}

// ================================================================================================
// a1 on the device: text -> first-appearance vocabulary -> CSR (bfk_text.hip), left bound in the context
// ================================================================================================
static int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// bytes a DEVICE text buffer handed to bfk_ctx_cluster_text_device / bfk_ctx_build_csr_device must have: the text, separator
// padding up to a multiple of 64 KiB, and a slack for unaligned 8-byte reads of a token's tail (the library writes the padding)
extern "C" int64_t bfk_text_device_bytes(int64_t text_bytes) {
    if (text_bytes < 0) return -1;
    return round_up(text_bytes + 1, TOK_PAD_BYTES) + TOK_TEXT_SLACK;
}

// The tokeniser on text that is RESIDENT IN HBM: d_text holds T bytes of rows and has room for bfk_text_device_bytes(T);
// d_rowoff = int64[n_rows + 1] on the device, offsets relative to `base`.  `n_pieces` > 1: the text is still arriving on the copy
// stream, piece by piece (host wrapper below).  `strict`: the offsets come from the caller's device memory, the kernels also
// check row_off[0] == base and row_off[n_rows] == base + T (the host wrapper has checked its own).
//   ctx_tok_launch  allocations, clearing, the tokenising kernels (k_tok_rows leaves the row statistics of the new indptr), and
//                   ONE small copy of the counters (token count, longest row, failure flags) into pinned host memory — all
//                   enqueued, no wait
//   ctx_tok_finish  waits for the stream and completes the bind from those counters (workspace sized by the longest row);
//                   *retry: the vocabulary table was too small, it has been enlarged, launch again
static int ctx_tok_launch(bfk_ctx *c, const bfk_ctx::TokPlan &tp, int attempt, bool copy_counters = true) {
    const int64_t T = tp.T, n_rows = tp.n_rows;
    const int64_t T_pad = round_up(T + 1, TOK_PAD_BYTES);
    // every token but the last of a row is followed by a separator: at most T/2 + n_rows + 1 tokens
    const int64_t nnz_cap = T / 2 + n_rows + 1;
    if (nnz_cap > (int64_t)INT32_MAX - 4096) return fail(BFK_EUNSUPPORTED, "device tokeniser: more than 2^31 possible tokens");
    const int64_t nnz_alloc = round_up(nnz_cap, 1024) + 16;
    const int64_t n_win = T_pad / TOK_WIN;
    // zeroed region: rowbits | firstbits (the counters live in d_small[8..15] and travel with the bind's copy)
    const int64_t bit_words = T_pad / 32 + 16;
    const int64_t z_rowbits = 0, z_firstbits = z_rowbits + bit_words * 4, z_bytes = z_firstbits + bit_words * 4;
    const int64_t zero_cap_before = c->tk_zero_cap;
    if (int rc = dev_realloc(&c->tk_zero, &c->tk_zero_cap, z_bytes, 1.05)) return rc;
    if (c->tk_zero_cap != zero_cap_before) c->tk_rowbits_clean = 0;  // (new memory)
    if (int rc = dev_realloc(&c->tk_bits, &c->tk_bits_cap, 3 * bit_words, 1.05)) return rc;  // start | bound | kept
    const int64_t n_blk = round_up(T_pad / TOK_PAD_BYTES + 1, 4) + 4;  // (+ the total behind the last block; 16-byte pieces)
    if (int rc = dev_realloc(&c->tk_winbase, &c->tk_winbase_cap, 3 * (n_win + n_blk), 1.05)) return rc;
    if (tp.flt.on)
        if (int rc = dev_realloc(&c->tk_slots, &c->tk_slots_cap, nnz_alloc, 1.05)) return rc;
    if (int rc = dev_realloc(&c->own_indices, &c->own_nnz_cap, nnz_alloc, 1.05)) return rc;
    if (int rc = dev_realloc(&c->own_indptr, &c->own_n_cap, n_rows + 1, 1.05)) return rc;
    hipEvent_t *ev = c->profiling ? c->tk_ev : nullptr;
    // vocabulary table: 1/32 slot per possible token (real inputs: ~8 bytes per token — a quarter of the bound — and a
    // vocabulary of a few % of the tokens: load below 20%); an input with more distinct tokens overflows the probe
    // limit, the table grows 8x (twice at most: 2 slots per possible token) and the kernels run again on the
    // resident text
    int64_t slots = 1 << 16;
    while (slots < nnz_cap / 32) slots <<= 1;
    // (a big table is walked three times and gathered from once per token: from 8M slots on half of that rule — 1M rows: 4M slots
    // for 1.4M entries, hash + 5 us, the walks and the gather - 29 us; 100k rows, 512k slots: halving loses 1.4 us)
    if (slots >= ((int64_t)8 << 20) && c->tk_grow == 0) slots >>= 1;
    slots <<= 3 * c->tk_grow;
    if (const char *e = getenv("BFK_TOK_SLOTS_SHIFT")) slots = atoi(e) >= 0 ? slots << atoi(e) : std::max<int64_t>(1 << 12, slots >> -atoi(e));  // (experiments)
    if (slots > ((int64_t)1 << 31)) slots = (int64_t)1 << 31;
    if (int rc = dev_realloc(&c->tk_table, &c->tk_table_cap, slots)) return rc;
    if (int rc = dev_realloc(&c->tk_tabid, &c->tk_tabid_cap, slots)) return rc;
    TokArgs a{};
    a.text = tp.d_text;
    a.row_off = tp.d_rowoff;
    a.base = tp.base;
    a.T = (uint32_t)T;
    a.T_pad = (uint32_t)T_pad;
    a.n_rows = (int)n_rows;
    a.sep = (uint8_t)tp.sep;
    a.strict = tp.strict ? 1 : 0;
    a.tc = (TokCounters *)(c->d_small + 8);
    a.rowbits = (uint32_t *)(c->tk_zero + z_rowbits);
    a.firstbits = (uint32_t *)(c->tk_zero + z_firstbits);
    a.startbits = c->tk_bits;
    a.boundbits = c->tk_bits + bit_words;
    a.winbase = c->tk_winbase;
    a.vocwin = c->tk_winbase + n_win;
    a.blkbase = c->tk_winbase + 2 * n_win;
    a.vocblk = c->tk_winbase + 2 * n_win + n_blk;
    a.keptbits = c->tk_bits + 2 * bit_words;
    a.keptwin = c->tk_winbase + 2 * n_win + 2 * n_blk;
    a.keptblk = c->tk_winbase + 3 * n_win + 2 * n_blk;
    a.flt = tp.flt;
    a.any_ids = tp.any_ids && !tp.flt.on ? 1 : 0;
    a.inv_queue = tp.d_inv;
    a.inv_cap = tp.d_inv ? tp.inv_cap : 0u;
    a.row_empties = tp.d_row_empties;
    a.span_len = tp.d_span_len;
    a.table = c->tk_table;
    a.tabid = c->tk_tabid;
    a.tmask = (uint32_t)(slots - 1);
    a.tokslot = tp.flt.on ? c->tk_slots : c->own_indices;  // (in place without the filter: the slot array IS the indices array)
    a.indices = c->own_indices;
    a.indptr = c->own_indptr;
    a.nnz_cap = nnz_cap;
    a.dbg = getenv("BFK_TOK_DEBUG") ? atoi(getenv("BFK_TOK_DEBUG")) : 0;
    // The row-start bits: the build before this one left them zero (k_voc_ids clears them when the scan has used them) over
    // tk_rowbits_clean words — then k_tok_clear sets this build's bits itself and there is no k_tok_rowbits launch.  Otherwise
    // (first build, a longer text than any before, new memory, a build that was not enqueued to its end) they are zeroed with
    // the rest and set by their own launch.  BFK_TOK_ROWBITS=0: always the latter.
    const bool rows_fused = n_rows > 0 && bit_words <= c->tk_rowbits_clean && a.dbg == 0 && !(getenv("BFK_TOK_ROWBITS") && atoi(getenv("BFK_TOK_ROWBITS")) == 0);
    c->tk_rowbits_clean = 0;  // (until this build is enqueued to its end)
    a.rows_fused = rows_fused ? 1 : 0;
    a.rows_clear_after = T_pad <= TOK_FUSE_ROWBITS_BYTES && a.dbg == 0 ? 1 : 0;
    // bit arrays zero, table all ones, separator padding behind the text, the row statistics [0..7] and the tokeniser's counters
    // [8..15]: one launch
    {
        const TokRows rows{tp.d_rowoff, tp.base, (uint32_t)T, (int)n_rows, rows_fused ? a.rowbits : nullptr};
        char *zero = rows_fused ? c->tk_zero + z_firstbits : c->tk_zero;
        if (int e = launch_tok_clear(zero, (size_t)(rows_fused ? z_bytes - z_firstbits : z_bytes), c->tk_table, (size_t)slots * sizeof(TokSlot), tp.d_text + T,
                                     (uint32_t)(T_pad + TOK_TEXT_SLACK - T), (uint8_t)tp.sep, c->d_small, c->stream, rows))
            return fail(BFK_EHIP, std::string("k_tok_clear launch: ") + hipGetErrorString((hipError_t)e));
    }
    a.fine_head = getenv("BFK_TOK_FINE") ? atoi(getenv("BFK_TOK_FINE")) : 1;
    a.head_units = getenv("BFK_TOK_HEAD_UNITS") ? atoi(getenv("BFK_TOK_HEAD_UNITS")) : 16;
    // the sample launch should stay a few hundred to a thousand waves whatever the text size (its waves must not storm among
    // themselves): every 16th unit up to ~130 MB, then a stride that grows with the text — 1M rows (81k units): 64, measured
    // against 16 on three input shapes: tree 336 -> 322 us, forest 608 -> 496, sorted 406 -> 340 (tools/tok_split_ab.sh)
    int sample = 16;
    while (sample < 256 && (T_pad / (TOK_WPW * TOK_WIN)) / 1000 >= 2 * sample) sample *= 2;
    a.sample = getenv("BFK_TOK_SAMPLE") ? atoi(getenv("BFK_TOK_SAMPLE")) : sample;
    // (a second attempt — the table grew — finds the text resident: no pieces to wait for)
    const bool pieces = tp.n_pieces > 1 && attempt == 0;
    if (int e = launch_tokenize(a, c->stream, ev, pieces ? tp.n_pieces : 1, tp.piece_blk, pieces ? c->tk_piece_ev : nullptr))
        return fail(BFK_EHIP, std::string("tokeniser launch: ") + hipGetErrorString((hipError_t)e));
    // the bind's device half — longest row, token count: k_tok_rows left them in d_small[0..4] — is done; the counters go to
    // the host with a copy of their own (slot 0) when the host is about to wait for them anyway, and with the first clustering
    // kernel's own stores when the step is device-driven (JoinArgs::dyn_host)
    c->d_indptr = c->own_indptr;
    c->d_indices = c->own_indices;
    c->n = n_rows;
    c->tk_rowbits_clean = a.rows_clear_after ? bit_words : 0;  // (k_voc_ids has been enqueued: the row bits are zero for the next build)
    if (copy_counters) HIP_TRY(hipMemcpyAsync(c->h_small, c->d_small, 64, hipMemcpyDeviceToHost, c->stream));
    return BFK_OK;
}

static int ctx_tok_finish(bfk_ctx *c, const bfk_ctx::TokPlan &tp, bool *retry, int h_slot = 0, hipEvent_t done_ev = nullptr) {
    *retry = false;
    if (done_ev) HIP_TRY(hipEventSynchronize(done_ev));  // (recorded behind the step's last kernel)
    else HIP_TRY(hipStreamSynchronize(c->stream));
    int h[16];
    memcpy(h, c->h_small + 16 * h_slot, sizeof h);
    TokCounters tc{};
    static_assert(sizeof(TokCounters) == 8 * sizeof(int), "TokCounters = d_small[8..15]");
    memcpy(&tc, h + 8, sizeof tc);
    memcpy(c->tok_host, h + 8, sizeof c->tok_host);
    if (getenv("BFK_TOK_DEBUG") && (atoi(getenv("BFK_TOK_DEBUG")) & 32))
        fprintf(stderr, "[bfk] tokeniser: %d tokens deferred by %d waves, nnz %u, vocabulary %u\n", tc.pad_[0], tc.pad_[1], tc.nnz, tc.n_vocab);
    c->tk_stats = bfk_text_stats{};
    c->tk_stats.text_bytes = tp.T;
    c->tk_stats.n_rows = tp.n_rows;
    c->tk_stats.table_slots = (int64_t)c->tk_table_cap;
    if ((tc.fail & TOK_FAIL_TABLE) && !(tc.fail & (TOK_FAIL_ROWOFF | TOK_FAIL_LONG)) && c->tk_grow < 2 &&
        c->tk_table_cap < ((int64_t)1 << 31)) {
        c->tk_grow++;
        *retry = true;
        return BFK_OK;
    }
    c->tk_stats.table_growths = c->tk_grow;
    if (tc.fail) {
        c->n = -1;  // nothing usable is bound
        if (tc.fail & TOK_FAIL_ROWOFF) return fail(BFK_EARG, "bfk_ctx_build_csr: row_off not monotone");
        if (tc.fail & TOK_FAIL_LONG) return fail(BFK_EUNSUPPORTED, "device tokeniser: a token of 64 KiB or more (the host tokeniser takes it)");
        return fail(BFK_EUNSUPPORTED, "device tokeniser: vocabulary table overflow (the host tokeniser takes it)");
    }
    // the bind's host half (what ctx_after_bind does for a CSR that comes from elsewhere)
    c->last_tiles = 0;
    if (c->n > 0 && (h[1] || h[4] != 0 || h[3] < 0)) return fail(BFK_EARG, "malformed CSR: indptr must start at 0 and be non-decreasing");
    c->nnz = c->n > 0 ? h[3] : 0;
    c->kcap = h[0];
    c->n_short = h[2];
    c->ran = false;
    c->need_zero = true;  // bins are laid out by kcap
    c->join_off = false;
    c->pg_off = false;
    c->pg_dense_d2 = false;
    c->tok_pending = false;
    if (int rc = ctx_size_workspace(c, 0)) return rc;
    if ((int64_t)(tp.flt.on ? tc.nnz_kept : tc.nnz) != c->nnz) return fail(BFK_EHIP, "device tokeniser: token count and indptr disagree");
    c->tk_stats.n_invalid = tc.n_invalid;
    c->tk_stats.n_empty = tc.n_empty;
    const bool any_ids = tp.any_ids && !tp.flt.on;  // (slots as ids: the vocabulary was not counted, the largest id is not known)
    c->max_tok = any_ids ? -1 : (int)tc.n_vocab - 1;  // ids are 0 .. n_vocab - 1: the prefix-group path needs no k_maxtok pass
    c->tk_stats.nnz = c->nnz;
    c->tk_stats.n_vocab = any_ids ? -1 : (int32_t)tc.n_vocab;
    if (hipEvent_t *ev = c->profiling ? c->tk_ev : nullptr) {  // (profiled steps are completed one at a time: the events are this step's)
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) c->tk_stats.ms_scan = ms;
        if (hipEventElapsedTime(&ms, ev[1], ev[2]) == hipSuccess) c->tk_stats.ms_hash = ms;
        c->tk_stats.ms_head = 0.f;  // (round 3's k_tok_head is gone: the hash is three launches of k_tok_hash, all inside ms_hash)
        if (hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) c->tk_stats.ms_ids = ms;
        if (hipEventElapsedTime(&ms, ev[0], ev[3]) == hipSuccess) c->tk_stats.ms_total = ms;
    }
    return BFK_OK;
}

// launch + finish (and again while the vocabulary table has to grow): the CSR is bound when this returns
static int ctx_tokenize(bfk_ctx *c, const bfk_ctx::TokPlan &tp, int64_t *nnz_out, int32_t *n_vocab_out) {
    for (int attempt = 0;; attempt++) {
        if (int rc = ctx_tok_launch(c, tp, attempt)) return rc;
        bool retry = false;
        if (int rc = ctx_tok_finish(c, tp, &retry)) return rc;
        if (!retry) break;
    }
    if (nnz_out) *nnz_out = c->tk_stats.nnz;
    if (n_vocab_out) *n_vocab_out = c->tk_stats.n_vocab;
    return BFK_OK;
}

static int ctx_enqueue(bfk_ctx *c, int32_t max_dist, int32_t shard, int32_t n_shards, void *d_labels_out, bool allow_join);

// May the clustering kernels of a text step go out BEFORE the host knows the token count and the longest row?  Only the
// variant join (max_dist 1, the CLI's default, up to 800k rows) has a device-driven form: its kernels read both from the
// counters k_maxlen leaves in device memory (JoinArgs::dyn), their grids are sized by the most tokens the text can hold, and
// they assume what nearly every profile input satisfies — no row longer than JOIN_INLINE_ROW tokens (k_join then decides every
// match itself: no k_verify launch) and at least one token.  An input outside that flags it and is redone by ctx_spec_finish.
// max-dist 1: the variant join or the band kernels.  ms per step on a bound CSR, fitted on the default generator (20k .. 1M rows,
// ~43 tokens per row) and on rows of ~105 tokens (30k .. 400k rows: profiles/r05_long_crossover.txt) — the join looks every TOKEN
// up (and its table outgrows the L2s with the rows), the band kernels' pair work goes with the ROWS:
//     join 0.019 + 0.00525 nnz/1M + 0.176 n/1M        band 0.03 + 0.41 n/1M
// (long rows, join / band measured: 50k 0.057 / 0.060, 70k 0.072 / 0.068, 100k 0.097 / 0.081, 200k 0.175 / 0.127, 400k 0.321 / 0.202;
// default: 100k 0.053 / 0.067, 1M 0.42 / 0.45.)  The join keeps a 10 % margin: hubs do not hurt it, and they hurt the band
// kernels' tiles a lot (a star phylogeny at 1M rows: 0.45 / 1.45).
static bool join_pays(int64_t n, int64_t nnz) {
    const double nM = (double)n * 1e-6, zM = (double)nnz * 1e-6;
    return 0.019 + 0.00525 * zM + 0.176 * nM <= 1.1 * (0.03 + 0.41 * nM);
}

static bool spec_wanted(const bfk_ctx *c, int64_t n_rows, int64_t T, int32_t max_dist) {
    if (const char *e = getenv("BFK_SPEC")) if (atoi(e) == 0) return false;
    if (const char *e = getenv("BFK_JOIN")) if (atoi(e) == 0) return false;
    if (getenv("BFK_JOIN_INLINE") || getenv("BFK_JOIN_DEBUG")) return false;
    if (max_dist != 1 || n_rows < 1 || n_rows > 800000 || c->edge_capture) return false;
    if (c->path_mode != 0 && c->path_mode != 2) return false;
    // (long rows — the text says so before it is tokenised: a token is ~8 bytes with its separator — are the band kernels': the
    // step waits once for its counters and join_wanted decides on the real count)
    if (c->path_mode == 0 && !join_pays(n_rows, T / 8)) return false;
    return T / 2 + n_rows + 1 < ((int64_t)1 << 30);
}

static int ctx_spec_finish_one(bfk_ctx *c);


// tokeniser launched -> the join's kernels behind it on device-resident counts; nothing waits
static int ctx_spec_enqueue(bfk_ctx *c, const bfk_ctx::TokPlan &tp, int32_t max_dist, void *d_labels_out) {
    // with the ring full the oldest open step is completed first (it finished long ago); profiled steps one at a time
    while (c->spec_count >= (c->profiling ? 1 : bfk_ctx::SPEC_RING))
        if (int rc = ctx_spec_finish_one(c)) return rc;
    const int idx = (c->spec_head + c->spec_count) % bfk_ctx::SPEC_RING;
    bfk_ctx::SpecStep &st = c->spec_ring[idx];
    if (!st.ev) HIP_TRY(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    st.tp = tp;
    st.d = max_dist;
    st.labels = d_labels_out;
    if (int rc = ctx_tok_launch(c, tp, 0, false)) return rc;
    c->spec_host = c->h_small_dev + 16 * (1 + idx);  // (k_jhash stores the counters there)
    // what the host assumes until the counters are back: the most tokens the text can hold, no row over JOIN_INLINE_ROW
    c->nnz = tp.T / 2 + tp.n_rows + 1;
    c->kcap = JOIN_INLINE_ROW;
    c->n_short = 0;
    c->max_tok = -1;
    c->last_tiles = 0;
    c->join_off = false;
    c->pg_off = false;
    c->pg_dense_d2 = false;
    c->need_zero = true;
    c->spec_enqueueing = true;  // (ctx_enqueue marks the plan device-driven by this)
    const int rc = ctx_enqueue(c, max_dist, 0, 1, d_labels_out, true);
    c->spec_enqueueing = false;
    if (rc) {
        (void)hipStreamSynchronize(c->stream);
        c->n = -1;
        return rc;
    }
    HIP_TRY(hipEventRecord(st.ev, c->stream));  // (behind the step's last kernel: the counters AND the labels are there when it fires)
    c->spec_count++;
    return BFK_OK;
}

// completes the bind of the OLDEST open text step from the counters in its slot.  What its assumptions did not cover — a
// vocabulary table that has to grow, a row over JOIN_INLINE_ROW tokens, an all-empty input — is redone with the host in the
// loop, and so is, in order, every step enqueued behind it (their results were written behind this step's).
static int ctx_spec_finish_one(bfk_ctx *c) {
    if (c->spec_count <= 0) return BFK_OK;
    const int idx = c->spec_head;
    const bfk_ctx::SpecStep st = c->spec_ring[idx];
    c->spec_head = (c->spec_head + 1) % bfk_ctx::SPEC_RING;
    c->spec_count--;
    bool retry = false;
    int rc = ctx_tok_finish(c, st.tp, &retry, 1 + idx, st.ev);
    const bool outside = !rc && !retry && (c->kcap > JOIN_INLINE_ROW || c->nnz <= 0);
    // the step's own outcome of the variant join (k_flatten left it in the step's slot: [5] give-up, [6] queue overflow, [7] a
    // queue that should have stayed empty): the labels in the caller's buffer miss edges — this step is redone on the all-pairs
    // path, the ones behind it as they were
    const int *slot = c->h_small + 16 * (1 + idx);
    const bool join_gave_up = !rc && !retry && !outside && (slot[5] | slot[6] | slot[7]) != 0;
    if (!rc && !retry && !outside && !join_gave_up) {
        c->ran = true;  // (ctx_tok_finish cleared it: the step it belongs to has been enqueued)
        return BFK_OK;
    }
    // this step again — and the open ones behind it — one after the other, each completed before the next
    std::vector<bfk_ctx::SpecStep> redo{st};
    while (c->spec_count > 0) {
        redo.push_back(c->spec_ring[c->spec_head]);
        c->spec_head = (c->spec_head + 1) % bfk_ctx::SPEC_RING;
        c->spec_count--;
    }
    (void)hipStreamSynchronize(c->stream);
    c->need_zero = c->ctr_dirty = true;
    c->join_clear = true;
    if (rc) return rc;  // (malformed offsets, a token of 64 KiB: the caller's error; the later steps are dropped with it)
    bool first = true;
    for (const bfk_ctx::SpecStep &r : redo) {
        if (int r2 = ctx_tokenize(c, r.tp, nullptr, nullptr)) return r2;
        if (int r2 = ctx_enqueue(c, r.d, 0, 1, r.labels, !(first && join_gave_up))) return r2;
        // every redone step is completed before the next one goes out: its own give-up or queue overflow (bfk_ctx_sync would
        // only see the last step's) is repaired here
        if (int r2 = ctx_settle(c)) return r2;
        first = false;
    }
    return BFK_OK;
}

static int ctx_spec_finish(bfk_ctx *c) {
    while (c->spec_count > 0)
        if (int rc = ctx_spec_finish_one(c)) return rc;
    return BFK_OK;
}

static int ctx_text_events(bfk_ctx *c) {
    if (c->profiling && !c->tk_ev_ready) {
        for (auto &e : c->tk_ev)
            if (hipEventCreate(&e) != hipSuccess) return fail(BFK_EHIP, "hipEventCreate failed");
        c->tk_ev_ready = true;
    }
    return BFK_OK;
}

// A token separator of several bytes (breakfast.py:164, :204: str.split takes any string) is folded on the device into runs of ONE
// byte that the text does not hold (k_sepfold), and the kernels run with that byte.  -> the separator's pattern and the stand-in
// (control characters first: a table of profiles holds none of them), or BFK_EUNSUPPORTED.
static int sep_stand_in(const char *sep, int64_t sep_len, const char *text, int64_t T, SepPattern *pat, unsigned char *stand_in) {
    if (sep_len > SEP_MAX_BYTES) return fail(BFK_EUNSUPPORTED, "device stages: token separator of more than 16 bytes");
    for (int64_t i = 0; i < sep_len; i++) {
        const unsigned char b = (unsigned char)sep[i];
        if (b >= 0x80 || b == '\n' || b == '\r' || b == 0) return fail(BFK_EUNSUPPORTED, "device stages: token separator");
    }
    *pat = SepPattern{};
    pat->m = (int)sep_len;
    memcpy(pat->b, sep, (size_t)sep_len);
    static const unsigned char cand[] = {0x1F, 0x1E, 0x1D, 0x1C, 0x1B, 0x1A, 0x19, 0x18, 0x17, 0x16, 0x15, 0x14, 0x13, 0x12, 0x11, 0x10,
                                         0x0F, 0x0E, 0x0C, 0x0B, 0x08, 0x07, 0x06, 0x05, 0x04, 0x03, 0x02, 0x01, 0x7F};
    for (unsigned char b : cand)
        if (T <= 0 || !memchr(text, b, (size_t)T)) {
            *stand_in = b;
            return BFK_OK;
        }
    return fail(BFK_EUNSUPPORTED, "device stages: no free byte to stand for the token separator");
}

static int ctx_check_text_args(const int64_t n_rows, const char *sep, int64_t sep_len, int64_t T) {
    if (n_rows < 0) return fail(BFK_EARG, "bfk_ctx_build_csr: negative n_rows");
    if (!sep || sep_len <= 0) return fail(BFK_EARG, "empty separator");
    if (n_rows > (int64_t)INT32_MAX - 2 * SIG_PAD_ROWS) return fail(BFK_EARG, "n_rows out of range");
    if (sep_len != 1) return fail(BFK_EUNSUPPORTED, "device tokeniser: one-byte separators only (the host tokeniser takes the rest)");
    if (T < 0) return fail(BFK_EARG, "bfk_ctx_build_csr: row_off not monotone");
    if (T > (int64_t)0xFFF00000ll) return fail(BFK_EUNSUPPORTED, "device tokeniser: 32-bit byte offsets (text of 4 GiB or more)");
    return BFK_OK;
}

// host buffers: the text and the offsets cross PCIe once into context-owned device buffers, then ctx_tokenize
// spec_d >= 0: the caller clusters at that max_dist right away into spec_labels — where spec_wanted() allows it the clustering
// kernels are enqueued behind the tokeniser here (*spec_done) instead of after a wait for its counters
static int ctx_build_text(bfk_ctx *c, const char *buf, const int64_t *row_off, int64_t n_rows, const char *sep, int64_t sep_len,
                          int64_t *nnz_out, int32_t *n_vocab_out, int32_t spec_d = -1, void *spec_labels = nullptr,
                          bool *spec_done = nullptr) {
    if (!row_off) return fail(BFK_EARG, "bfk_ctx_build_csr: null argument");
    const int64_t base = n_rows >= 0 ? row_off[0] : 0, T = n_rows >= 0 ? row_off[n_rows] - base : -1;
    if (T > 0 && !buf) return fail(BFK_EARG, "bfk_ctx_build_csr: null text");
    SepPattern pat{};
    unsigned char sp = sep && sep_len > 0 ? (unsigned char)sep[0] : 0;
    if (sep && sep_len > 1 && n_rows >= 0 && T >= 0 && T <= (int64_t)0xFFF00000ll)  // (several bytes: folded below)
        if (int rc = sep_stand_in(sep, sep_len, buf + base, T, &pat, &sp)) return rc;
    {
        const char one = (char)sp;
        if (int rc = ctx_check_text_args(n_rows, sep_len > 1 ? &one : sep, sep_len > 1 ? 1 : sep_len, T)) return rc;
    }
    if (int rc = ctx_settle(c)) return rc;  // (the tokeniser is about to write this context's CSR)
    const int64_t T_pad = round_up(T + 1, TOK_PAD_BYTES);
    if (int rc = dev_realloc(&c->tk_text, &c->tk_text_cap, T_pad + TOK_TEXT_SLACK, 1.05)) return rc;
    if (int rc = dev_realloc(&c->tk_rowoff, &c->tk_rowoff_cap, n_rows + 1, 1.05)) return rc;
    if (int rc = ctx_text_events(c)) return rc;
    hipEvent_t *ev = c->profiling ? c->tk_ev : nullptr;
    // The text and the row offsets: the caller's buffers (pageable, or pinned: bfk_host_alloc), borrowed until the copies are
    // done.  ONE copy of the whole text, then the kernels.  BFK_TOK_PIECES=n (2..8, texts from 8 MB) sends the text in n pieces on a copy stream of
    // its own and tokenises piece k under the copy of piece k + 1 — measured and left off: every copy from pageable memory
    // costs ~60 us of driver work before it moves a byte, which is more than the ~30 us of kernels a piece hides
    // (bfk_cluster_text, 100k rows / 31 MB: 1.10 ms in one piece, 1.20-1.28 in two, 1.24 in four, 1.43 in eight; 1M rows /
    // 333 MB: 7.8 against 8.0-8.4).  (With profiling on everything runs on the one stream, the phases one after the other.)
    int n_pieces = 1;
    unsigned piece_blk[bfk_ctx::TK_PIECES + 1] = {0};
    const int64_t scan_blocks = T_pad / TOK_PAD_BYTES;
    if (!c->profiling && sep_len == 1 && T >= ((int64_t)8 << 20) && getenv("BFK_TOK_PIECES") && atoi(getenv("BFK_TOK_PIECES")) >= 2) {
        n_pieces = std::min((int)bfk_ctx::TK_PIECES, atoi(getenv("BFK_TOK_PIECES")));
        for (int k = 0; k <= n_pieces; k++)  // boundaries at multiples of 4 scan blocks (256 KiB: 16-byte pieces of blkbase)
            piece_blk[k] = k == n_pieces ? (unsigned)scan_blocks : (unsigned)(scan_blocks * k / n_pieces / 4 * 4);
        if (!c->tk_copy_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&c->tk_copy_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->tk_start_ev, hipEventDisableTiming));
            for (auto &e : c->tk_piece_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
    }
    // whatever way this function is left — also by an early return further down (an allocation that fails, a launch error)
    // — the caller's buffers are no longer being read when it returns: the text and the offsets were enqueued from the
    // caller's memory on the launch stream (one piece) or on the copy stream (pieces).  On the success path the launch
    // stream is idle by then (ctx_after_bind has synchronised it): the second wait costs nothing.
    struct CopyGuard {
        hipStream_t copy, launch;
        ~CopyGuard() {
            if (copy) (void)hipStreamSynchronize(copy);
            if (launch) (void)hipStreamSynchronize(launch);
        }
    } copy_guard{n_pieces > 1 ? c->tk_copy_stream : nullptr, c->stream};
    if (ev) HIP_TRY(hipEventRecord(ev[4], c->stream));
    HIP_TRY(hipMemcpyAsync(c->tk_rowoff, row_off, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (n_pieces == 1) {
        if (T > 0) HIP_TRY(hipMemcpyAsync(c->tk_text, buf + base, (size_t)T, hipMemcpyHostToDevice, c->stream));
    } else {
        // the copy stream may not write the text before the launch stream is done with what it holds (an earlier build)
        HIP_TRY(hipEventRecord(c->tk_start_ev, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->tk_copy_stream, c->tk_start_ev, 0));
        for (int k = 0; k < n_pieces; k++) {
            const int64_t o0 = (int64_t)piece_blk[k] * TOK_PAD_BYTES, o1 = std::min<int64_t>(T, (int64_t)piece_blk[k + 1] * TOK_PAD_BYTES);
            if (o1 > o0) HIP_TRY(hipMemcpyAsync(c->tk_text + o0, buf + base + o0, (size_t)(o1 - o0), hipMemcpyHostToDevice, c->tk_copy_stream));
            HIP_TRY(hipEventRecord(c->tk_piece_ev[k], c->tk_copy_stream));
        }
    }
    if (ev) HIP_TRY(hipEventRecord(ev[5], c->stream));
    if (sep_len > 1 && n_rows > 0)
        if (int e = launch_sepfold(c->tk_text, c->tk_rowoff, nullptr, (int)n_rows, base, pat, (uint8_t)sp, nullptr, nullptr, c->stream))
            return fail(BFK_EHIP, std::string("k_sepfold launch: ") + hipGetErrorString((hipError_t)e));
    bfk_ctx::TokPlan tp;
    tp.d_text = c->tk_text;
    tp.d_rowoff = c->tk_rowoff;
    tp.base = base;
    tp.T = T;
    tp.n_rows = n_rows;
    tp.sep = (char)sp;
    tp.strict = false;
    tp.n_pieces = n_pieces;
    memcpy(tp.piece_blk, piece_blk, sizeof tp.piece_blk);
    tp.any_ids = spec_d >= 0 && !n_vocab_out && tok_any_ids_wanted(c, spec_d);  // (bfk_cluster_text without a vocabulary count)
    int rc;
    if (spec_d >= 0 && spec_wanted(c, n_rows, T, spec_d)) {
        // bfk_cluster_text: the clustering kernels follow the tokeniser without the host in between; the caller's buffers are
        // released by the guard above (it waits for the stream), the bind is completed by the bfk_ctx_sync that follows
        rc = ctx_spec_enqueue(c, tp, spec_d, spec_labels);
        if (!rc) rc = ctx_spec_finish(c);  // (ONE wait: for the copies, the tokeniser and the clustering kernels together)
        if (!rc) *spec_done = true;
    } else {
        rc = ctx_tokenize(c, tp, nnz_out, n_vocab_out);
    }
    if (!rc && (nnz_out || n_vocab_out)) {
        if (nnz_out) *nnz_out = c->tk_stats.nnz;
        if (n_vocab_out) *n_vocab_out = c->tk_stats.n_vocab;
    }
    if (!rc && ev) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev[4], ev[5]) == hipSuccess) c->tk_stats.ms_h2d = ms;
        if (hipEventElapsedTime(&ms, ev[4], ev[3]) == hipSuccess) c->tk_stats.ms_total = ms;
    }
    return rc;
}

// a1 on text that already lives in device memory (include/bfk.h): no copy — the kernels read the caller's buffer
extern "C" int bfk_ctx_build_csr_device(bfk_ctx *c, void *d_text, int64_t text_bytes, const void *d_row_off, int64_t n_rows,
                                        const char *sep, int64_t sep_len, int64_t *nnz_out, int32_t *n_vocab_out) {
    if (int rc = ctx_enter(c)) return rc;
    if (int rc = ctx_check_text_args(n_rows, sep, sep_len, text_bytes)) return rc;
    if (!d_text || !d_row_off) return fail(BFK_EARG, "bfk_ctx_build_csr_device: null device pointer");
    if (n_rows == 0 && text_bytes != 0) return fail(BFK_EARG, "bfk_ctx_build_csr_device: text without rows");
    if (int rc = ctx_settle(c)) return rc;
    if (int rc = ctx_text_events(c)) return rc;
    bfk_ctx::TokPlan tp;
    tp.d_text = (uint8_t *)d_text;
    tp.d_rowoff = (const long long *)d_row_off;
    tp.T = text_bytes;
    tp.n_rows = n_rows;
    tp.sep = sep[0];
    tp.strict = true;
    tp.any_ids = c->tok_any_now;  // (set by bfk_ctx_cluster_text_device around its call: a caller of this entry gets the reference's ids)
    return ctx_tokenize(c, tp, nnz_out, n_vocab_out);
}

// a1 .. a8 with the text RESIDENT IN HBM and the labels left in HBM: what bench.py times as one step.  Where the variant join
// serves the step (spec_wanted) nothing waits between the tokeniser and the clustering kernels; elsewhere the host sizes the
// clustering kernels from the tokeniser's counters first (one wait).
extern "C" int bfk_ctx_cluster_text_device(bfk_ctx *c, void *d_text, int64_t text_bytes, const void *d_row_off, int64_t n_rows,
                                           const char *sep, int64_t sep_len, int32_t max_dist, void *d_labels_out) {
    if (max_dist < 0) return fail(BFK_EARG, "max_dist must be >= 0");
    if (n_rows > 0 && !d_labels_out) return fail(BFK_EARG, "null labels");
    if (int rc = ctx_enter(c)) return rc;
    if (int rc = ctx_check_text_args(n_rows, sep, sep_len, text_bytes)) return rc;
    if (!d_text || !d_row_off) return fail(BFK_EARG, "bfk_ctx_cluster_text_device: null device pointer");
    if (n_rows == 0 && text_bytes != 0) return fail(BFK_EARG, "bfk_ctx_cluster_text_device: text without rows");
    if (int rc = ctx_settle(c)) return rc;  // (a step that waited once and whose clustering kernels' outcome is still unread)
    if (int rc = ctx_text_events(c)) return rc;
    if (spec_wanted(c, n_rows, text_bytes, max_dist)) {
        bfk_ctx::TokPlan tp;
        tp.d_text = (uint8_t *)d_text;
        tp.d_rowoff = (const long long *)d_row_off;
        tp.T = text_bytes;
        tp.n_rows = n_rows;
        tp.sep = sep[0];
        tp.strict = true;
        tp.any_ids = tok_any_ids_wanted(c, max_dist);
        return ctx_spec_enqueue(c, tp, max_dist, d_labels_out);
    }
    c->tok_any_now = tok_any_ids_wanted(c, max_dist);
    const int rc = bfk_ctx_build_csr_device(c, d_text, text_bytes, d_row_off, n_rows, sep, sep_len, nullptr, nullptr);
    c->tok_any_now = false;
    if (rc) return rc;
    return bfk_ctx_cluster(c, max_dist, 0, 1, d_labels_out);
}

// pinned host memory for a caller's text (and anything else it hands over often): copies from it run at the PCIe rate from the
// first byte — a pageable buffer the driver has never seen is pinned page by page on the way (bench.py: t_cluster_host_ms)
extern "C" int bfk_host_alloc(int64_t bytes, void **out) {
    if (!out || bytes < 0) return fail(BFK_EARG, "bfk_host_alloc: bad argument");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, (size_t)std::max<int64_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? BFK_ENODEV : BFK_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    return BFK_OK;
}

extern "C" int bfk_host_free(void *p) {
    if (!p) return BFK_OK;
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) return fail(BFK_EARG, std::string("hipHostFree: ") + hipGetErrorString(e));
    return BFK_OK;
}

extern "C" int bfk_ctx_build_csr(bfk_ctx *c, const char *buf, const int64_t *row_off, int64_t n_rows, const char *sep,
                                 int64_t sep_len, int64_t *nnz_out, int32_t *n_vocab_out) {
    if (int rc = ctx_enter(c)) return rc;
    return ctx_build_text(c, buf, row_off, n_rows, sep, sep_len, nnz_out, n_vocab_out);
}

extern "C" int bfk_ctx_text_stats(bfk_ctx *c, bfk_text_stats *out) {
    if (!c || !out) return fail(BFK_EARG, "null argument");
    if (int rc = ctx_enter(c)) return rc;
    *out = c->tk_stats;
    return BFK_OK;
}

extern "C" int bfk_ctx_download_csr(bfk_ctx *c, int32_t *indptr_out, int32_t *indices_out) {
    if (int rc = ctx_enter(c)) return rc;
    if (c->n < 0) return fail(BFK_ESTATE, "no CSR bound");
    if (!indptr_out || (c->nnz > 0 && !indices_out)) return fail(BFK_EARG, "null output");
    HIP_TRY(hipMemcpyAsync(indptr_out, c->d_indptr, (size_t)(c->n + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    if (c->nnz > 0) HIP_TRY(hipMemcpyAsync(indices_out, c->d_indices, (size_t)c->nnz * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFK_OK;
}

static int sig_words_for(int d) {
    if (const char *e = getenv("BFK_SIG_WORDS")) {
        const int w = atoi(e);
        if (w == 1 || w == 2 || w == 4) return w;
    }
    return d <= 2 ? 1 : (d <= 5 ? 2 : 4);
}

// The variant join serves max_dist == 1 (BFK_JOIN=0 forces the all-pairs kernels, e.g. to measure them).
static bool join_wanted(const bfk_ctx *c, int max_dist) {
    if (max_dist != 1 || c->join_off || c->nnz <= 0 || c->n > ((int64_t)1 << 27) || c->kcap >= (1 << 25) ||
        c->nnz > (int64_t)INT32_MAX - 4 * JOIN_TPW)  // 32-bit token offsets, position packed above 6 bits
        return false;
    if (c->path_mode) return c->path_mode == 2;
    if (const char *e = getenv("BFK_JOIN")) return atoi(e) != 0;
    // measured (ms per step, join vs all-pairs): 100k rows 0.061 / 0.074, 300k 0.153 / 0.172, 600k 0.325 / 0.360,
    // 1M 0.537 / 0.505 — both grow about linearly there, the join's cost is the instruction stream per row.
    // Round 4, other workload shapes at 1M rows (profiles/r04_family_matrix.txt), join / all-pairs: default 0.480 / 0.475,
    // amino-acid tokens 0.511 / 0.486, rows of ~110 tokens 1.04 / 0.59 (the join's lookups are per token), a star phylogeny
    // 0.51 / 1.47 (the hubs' neighbours fill a few cells of the band kernels' sort key: tiles of thousands of rows).  Hence the
    // join beyond 800k rows too while rows are short — it never loses much there and is the one the hubs do not hurt.
    // (a device-driven text step has decided already — spec_wanted, on the text's size: what the context holds as nnz while the
    // step is being enqueued is the most tokens the text can hold, not a count)
    if (c->spec_enqueueing) return true;
    // Round 5: rows of ~105 tokens turn to the band kernels from ~60k rows on (0.097 / 0.081 at 100k, 0.321 / 0.202 at 400k): the
    // fixed "800k rows" became join_pays' two lines (above).
    // (with the filter bitmap capped at 2 MB the join stays ahead of the band kernels up to 2M short rows — 1.5M: 0.702 / 0.726 ms,
    // 2M: 0.957 / 0.997; beyond, 8 bits per row no longer fit the cap)
    return c->n <= 2000000 && join_pays(c->n, c->nnz);
}

static int ctx_size_join(bfk_ctx *c) {
    int64_t slots = 1024, bits = 4096;
    while (slots < 4 * c->n) slots <<= 1;  // load <= 1/4
    while (bits < 32 * c->n) bits <<= 1;   // ~3% of the bits set
    // The bitmap is looked up once per token: it has to stay in an XCD's 4 MB L2 next to everything else.  1M rows: 32 bits per
    // row = 4 MB -> 0.465 ms per step, 16 bits = 2 MB -> 0.411, 8 bits -> 0.416 (64 bits, 8 MB: 0.695); 100k rows (512 KB): no
    // difference.  Capped at 2 MB while that leaves 8 bits per row (a fuller filter costs a table probe per false hit:
    // 0.4 % of the lookups at 32 bits per row, 1.6 % at 16, 6 % at 8).
    {
        int64_t cap = (int64_t)16 << 20;
        while (cap < 8 * c->n) cap <<= 1;
        bits = std::min(bits, cap);
    }
    // (small inputs: a table of up to 8 MB at half the load — 100k rows: 0.0533 -> 0.0514 ms; at 1M rows a bigger table loses)
    if (slots * 8 <= ((int64_t)4 << 20)) slots <<= 1;
    if (const char *e = getenv("BFK_JOIN_BITS_SHIFT")) bits = atoi(e) >= 0 ? bits << atoi(e) : std::max<int64_t>(4096, bits >> -atoi(e));  // (experiments)
    if (const char *e = getenv("BFK_JOIN_SLOTS_SHIFT")) slots = atoi(e) >= 0 ? slots << atoi(e) : std::max<int64_t>(1024, slots >> -atoi(e));
    const int64_t blocks = c->nnz / (16 * JOIN_TPW) + 2;  // k_join blocks
    const int64_t batches = c->nnz / JOIN_TPW + 2;
    const int64_t bytes = 2 * slots * 8 + 2 * bits / 8 + (c->n + 16) * 8 + (4 * c->n + 65536) * 8 + blocks * 8 + batches * 4;
    if (bytes > c->join_bytes || !c->d_join) {
        if (c->d_join) (void)hipFree(c->d_join);
        c->d_join = nullptr;
        c->join_bytes = 0;
        if (hipMalloc((void **)&c->d_join, (size_t)bytes) != hipSuccess) return fail(BFK_ENOMEM, "hipMalloc(join tables) failed");
        c->join_bytes = bytes;
        c->join_clear = true;
    }
    if (slots != c->join_slots || bits != c->join_bits) c->join_clear = true;
    c->join_slots = slots;
    c->join_bits = bits;
    return BFK_OK;
}

// The prefix-group path serves max_dist 2..7 on large inputs (BFK_PG=0/1 or bfk_ctx_set_candidate_path(3) override): below the
// thresholds the sort and the passes over (d + 2) N records cost more than the band scan they save (measured: DESIGN 6d).
// Rows of at most 2 * max_dist tokens all meet in ONE group (prefix filtering says nothing about them): an input with
// many of those stays on the band path.
// measured, ms per step band / prefix groups (tools/pg_matrix.sh -> profiles/r02_pg_matrix.txt; labels-only steps, indels kept):
//   rows    d = 2          d = 3          d = 4          d = 5
//   30k                    0.30 / 0.27    0.54 / 0.38    0.99 / 0.40
//   70k                    0.39 / 0.42    0.74 / 0.53    1.46 / 0.50
//   100k    0.18 / 0.42    0.45 / 0.50    0.84 / 0.57    1.63 / 0.58
//   300k    0.48 / 0.58    1.28 / 0.67    2.07 / 0.72    3.22 / 0.92
//   1M      1.49 / 1.72    4.54 / 1.91    13.2 / 2.16    21.0 / 2.99
// the records, their sort and the group order cost 0.1 ms at 30k rows, 0.15 at 100k and 0.4 - 0.6 ms at 1M whatever max_dist
// is, the band scan they replace grows steeply with it
// round 3 (k_pgwalk16, positional filter at every size; profiles/r03_pg_matrix.txt), band / prefix groups:
//   rows    d = 2          d = 3          d = 4          d = 5
//   3k                     0.10 / 0.14    0.13 / 0.14    0.21 / 0.16
//   5k                     0.12 / 0.14    0.18 / 0.16    0.29 / 0.19
//   10k     0.10 / 0.14    0.17 / 0.16    0.28 / 0.17    0.52 / 0.22
//   30k     0.12 / 0.21    0.25 / 0.19    0.44 / 0.27    0.93 / 0.29
//   100k    0.18 / 0.37    0.40 / 0.32    0.77 / 0.34    1.52 / 0.42
//   200k    0.32 / 0.33
//   300k    0.47 / 0.41    1.21 / 0.49    1.97 / 0.52    3.11 / 0.60
//   1M      1.46 / 1.09    (4.5) / 1.23                  (21) / 1.79
//   3M      6.30 / <4.5
// round 4 (SHORT records left out of the sort where no row is short; tools/d2_crossover.py), max_dist 2, band / prefix groups:
//   rows    default        aa (45 tokens) long (105 tokens per row)   star
//   100k    0.18 / 0.23    0.20 / 0.24    0.17 / 0.27                 0.52 / 0.38
//   200k    0.31 / 0.32    0.36 / 0.33    0.27 / 0.37                 0.88 / 0.61
//   300k    0.46 / 0.40    0.55 / 0.41    0.37 / 0.44                 1.36 / 1.04
// the groups' prep reads every token, the band's pair kernel does not care how long the rows are: the crossover moves up
// with the mean row length (the star family — the band's queue takes ten times the candidates — would want the groups at
// every size: known only after a step, not used)
// round 5 (tools/allpairs_ab.py, profiles/r05_long_d3.txt), rows of ~105 tokens, band / prefix groups:
//   rows    d = 3            d = 4
//   20k     0.191 / 0.238
//   50k     0.223 / 0.275    0.316 / 0.289
//   100k    0.350 / 0.412    0.453 / 0.432
//   200k    0.595 / 0.580    0.796 / 0.594
//   400k    1.143 / 0.875
// at max-dist 3 the crossover sits at ~10k rows of 43 tokens and at ~180k rows of 105: it moves with about the cube of the mean
// row length (the groups' prep and walk read every token, the band's pair kernel does not care how long the rows are); from
// max-dist 4 on the band scan grows so steeply that the groups win at every size measured.
static int64_t PG_MIN_ROWS(int max_dist, int64_t n, int64_t nnz) {
    const int64_t mean_len = n > 0 ? nnz / n : 0;
    if (max_dist == 3) return (int64_t)(10000.0 * std::pow((double)std::max<int64_t>(44, mean_len) / 44.0, 3.3));
    if (max_dist >= 3) return max_dist >= 5 ? 2500 : 4000;
    return 200000 * std::max<int64_t>(50, mean_len) / 50;
}

static bool pg_wanted(const bfk_ctx *c, int max_dist, int n_shards) {
    if (max_dist < 2 || max_dist > PG_MAX_DIST || c->n < 2 || c->nnz <= 0 || c->pg_off) return false;
    if ((int64_t)c->n * (max_dist + 2) > (int64_t)INT32_MAX - 1024) return false;
    if (c->path_mode) return c->path_mode == 3;
    if (const char *e = getenv("BFK_PG")) return atoi(e) != 0;
    if (c->n_short > 4096) return false;
    // a shard of a max-dist 2 step: the band kernels — their pair kernel is the step and shards, the groups' records and sort
    // are replicated on every rank (one-device rehearsal, 1M rows, 8 ranks: 0.50 ms band / 1.05 groups)
    if (n_shards > 1 && max_dist == 2) return false;
    if (max_dist == 2 && c->pg_dense_d2) return true;
    return c->n >= PG_MIN_ROWS(max_dist, c->n, c->nnz);
}

// the 32-bit record keys hold token + 1 (0: SHORT record, all ones: none): the largest token id of the CSR — one pass over
// the indices, once per bind — gives the number of key bits the sort has to look at
static int pg_key_bits(bfk_ctx *c, int *tb_out) {
    if (c->max_tok < 0) {
        HIP_TRY(hipMemsetAsync(c->d_small, 0, 16, c->stream));
        if (int e = launch_maxtok(c->d_indices, (int)c->nnz, c->d_small, c->stream))
            return fail(BFK_EHIP, std::string("k_maxtok launch: ") + hipGetErrorString((hipError_t)e));
        int m = 0;
        HIP_TRY(hipMemcpyAsync(&m, c->d_small, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->max_tok = m;
    }
    int tb = 2;
    while (tb < 31 && ((int64_t)1 << tb) - 3 < (int64_t)c->max_tok) tb++;  // token + 1 <= 2^tb - 2
    *tb_out = tb;
    return BFK_OK;
}

static int ctx_size_pg(bfk_ctx *c, int recs, int key_bits, size_t *temp_bytes) {
    const int64_t total = c->n * recs;
    if (!c->pg_cnt && hipMalloc((void **)&c->pg_cnt, sizeof(uint32_t) << PG_CNT_BITS) != hipSuccess) return fail(BFK_ENOMEM, "hipMalloc(token counts) failed");
    if (int rc = dev_realloc(&c->pg_rowinfo, &c->pg_rowinfo_cap, c->n + SIG_PAD_ROWS)) return rc;
    if (total > c->pg_rec_cap) {
        int64_t cap;
        int rc = 0;
        cap = 0; rc |= dev_realloc(&c->pg_keys, &cap, total);
        cap = 0; rc |= dev_realloc(&c->pg_keys_s, &cap, total);
        cap = 0; rc |= dev_realloc(&c->pg_keys_pm, &cap, total);
        // (pg_rows — the unsorted values — is gone: the sort's first pass works a record's (row, slot) out from its position)
        cap = 0; rc |= dev_realloc(&c->pg_rows_s, &cap, total);
        cap = 0; rc |= dev_realloc(&c->pg_srec, &cap, total + SIG_PAD_ROWS);
        cap = 0; rc |= dev_realloc(&c->pg_recpos, &cap, total);
        if (rc) return BFK_ENOMEM;
        c->pg_rec_cap = total;
    }
    size_t tb = 0;
    if (int e = sort_records(nullptr, &tb, c->pg_keys, c->pg_keys_s, c->pg_rows, c->pg_rows_s, (size_t)total, key_bits, c->stream))  // (the bits that will be sorted: the size is asked for exactly that call)
        return fail(BFK_EHIP, std::string("radix sort set-up: ") + hipGetErrorString((hipError_t)e));
    if ((int64_t)tb > c->pg_temp_cap) {
        if (c->pg_temp) (void)hipFree(c->pg_temp);
        c->pg_temp = nullptr;
        c->pg_temp_cap = 0;
        if (hipMalloc(&c->pg_temp, tb + 256) != hipSuccess) return fail(BFK_ENOMEM, "hipMalloc(sort workspace) failed");
        c->pg_temp_cap = (int64_t)tb + 256;
    }
    *temp_bytes = tb;
    return BFK_OK;
}

static int ctx_enqueue(bfk_ctx *c, int32_t max_dist, int32_t shard, int32_t n_shards, void *d_labels_out, bool allow_join);

extern "C" int bfk_ctx_cluster(bfk_ctx *c, int32_t max_dist, int32_t shard, int32_t n_shards, void *d_labels_out) {
    if (int rc = ctx_enter(c)) return rc;
    if (c->n < 0) return fail(BFK_ESTATE, "bfk_ctx_cluster: no CSR bound");
    if (max_dist < 0) return fail(BFK_EARG, "max_dist must be >= 0");
    if (n_shards <= 0) n_shards = 1;
    if (shard < 0 || shard >= n_shards) return fail(BFK_EARG, "shard out of range");
    if (c->n > 0 && !d_labels_out) return fail(BFK_EARG, "null labels");
    // (the same run again — a caller that times steps — needs no settling in between: the last one is repaired by the sync)
    if (c->unsettled && (max_dist != c->last_d || d_labels_out != c->plan.labels || shard != c->plan.shard || n_shards != c->plan.n_shards))
        if (int rc = ctx_settle(c)) return rc;
    return ctx_enqueue(c, max_dist, shard, n_shards, d_labels_out, true);
}

static int ctx_enqueue(bfk_ctx *c, int32_t max_dist, int32_t shard, int32_t n_shards, void *d_labels_out, bool allow_join) {
    c->ran = true;
    if (!c->spec_enqueueing) c->unsettled = true;  // (a device-driven text step carries its outcome in its own slot: ctx_spec_finish_one)
    c->last_d = max_dist;
    c->last_shards = n_shards;
    c->last_w1 = sig_words_for(max_dist);
    if (c->n == 0) return BFK_OK;
    if (int rc = ctx_size_workspace(c, max_dist)) return rc;  // no-op unless max_dist needs a longer item list
    if (c->edge_capture) {
        if (int rc = dev_realloc(&c->d_edges, &c->edge_cap, c->cand_cap_shard * CAND_SHARDS)) return rc;
    }
    Plan pl{};
    pl.n = (int)c->n;
    pl.kcap = c->kcap;
    pl.nnz = (int)c->nnz;
    pl.d = std::min<int>(max_dist, 1 << 20);
    pl.w1 = c->last_w1;
    pl.rows_per_lane = c->rows_per_lane;
    pl.fb = c->fb;
    pl.gb = c->gb;
    pl.hb = c->hb;
    pl.shard = shard;
    pl.n_shards = n_shards;
    // x 16 groups, a multiple of CAND_SHARDS; measured best of 1024..4096: 6 blocks per CU with the per-wave table
    // (d <= 2, 100k rows), 2048 with per-group tables (d = 5, 1M rows)
    pl.wave_table_d = 1;  // d = 2: pairs with two separate insertions fail the certificate (0.17 vs 0.26 ms at 100k rows)
    if (const char *e = getenv("BFK_WAVE_TABLE_D")) pl.wave_table_d = atoi(e);
    pl.verify_grid = max_dist <= pl.wave_table_d ? 1536 : 2048;
    if (const char *e = getenv("BFK_VERIFY_GRID")) pl.verify_grid = std::min(VERIFY_GRID_MAX, std::max(32, atoi(e) / 32 * 32));
    pl.blk_stats = c->d_blk_stats;
    // two-phase verify (first 1/8 of every queue shard by splicing, compress the forest, the rest by find + hook): pays
    // where the hooks of a dense graph fight over few roots — measured (verify kernel, one / two phases): 100k rows
    // d = 3: 374 / 273 us, d = 4: 590 / 470 us; no gain at d = 5 (849 / 849: the exact counts dominate), a loss at
    // 1M rows (d = 3: 1.49 / 1.85 ms, d = 5: 5.1 / 5.7 ms) and nothing to win at d <= 2 (sparse forests, splicing)
    // Labels-only steps of dense graphs (max_dist >= 3) drop a candidate whose rows are in one tree already without
    // computing its distance (k_verify_connected); n_edges then counts the edges that were checked, n_connected the rest.
    // bfk_ctx_set_exact_edges(ctx, 1) / BFK_EXACT_EDGES=1 / edge capture: every candidate is checked.
    {
        int skip = max_dist >= 3;
        if (const char *e = getenv("BFK_SKIP_CONNECTED")) skip = atoi(e) != 0;
        int exact = 0;
        if (const char *e = getenv("BFK_EXACT_EDGES")) exact = atoi(e) != 0;
        if (c->exact_edges >= 0) exact = c->exact_edges;
        pl.skip_connected = skip && !exact && !c->edge_capture;
        // k_verify_connected holds 38 KiB of LDS per block: four blocks per CU are resident, and a grid of exactly those
        // was 0.93 ms against 1.05 (2048 blocks) / 1.09 (8192) at 1M rows, max_dist 5
        if (pl.skip_connected && !getenv("BFK_VERIFY_GRID")) pl.verify_grid = std::min(VERIFY_GRID_MAX, c->n_cus * 4);
    }
    {   // max_dist 2, labels only: sparse or dense is decided on the device from the candidate queue's fill (launch_verify)
        int exact = 0;
        if (const char *e = getenv("BFK_EXACT_EDGES")) exact = atoi(e) != 0;
        if (c->exact_edges >= 0) exact = c->exact_edges;
        pl.verify_adaptive = max_dist == 2 && !exact && !c->edge_capture && !getenv("BFK_SKIP_CONNECTED") && !getenv("BFK_VERIFY_PHASES");
        if (const char *e = getenv("BFK_VERIFY_ADAPTIVE")) pl.verify_adaptive = pl.verify_adaptive && atoi(e) != 0;
        pl.verify_grid2 = std::min(VERIFY_GRID_MAX, c->n_cus * 4);
        pl.verify_density_thr = 8 * (long long)c->n;
    }
    // (with the pruning kernel one phase is better: 100k rows d = 3: 0.217 / 0.241 ms, d = 4: 0.262 / 0.271, 300k d = 4: 0.431 / 0.479)
    pl.verify_phases = ((max_dist == 3 || max_dist == 4) && c->n < 400000 && !pl.skip_connected) ? 8 : 1;
    pl.verify_phase2_union = 0;
    if (const char *e = getenv("BFK_VERIFY_PHASES")) pl.verify_phases = std::max(1, std::min(64, atoi(e)));
    if (const char *e = getenv("BFK_VERIFY_PHASE2")) pl.verify_phase2_union = std::max(0, std::min(2, atoi(e)));
    pl.tile_cap = (int)std::min<int64_t>(c->tile_cap, INT32_MAX);
    pl.pf_blocks = c->n_cus * 256;  // upper bound of the pair kernel's grid (it strides over the tile entries)
    if (const char *e = getenv("BFK_PF_BLOCKS")) pl.pf_blocks = c->n_cus * std::max(1, atoi(e));
    pl.pf_waves = (c->rows_per_lane == 1 && c->n < 400000) ? 4 : 2;
    if (const char *e = getenv("BFK_PF_WAVES")) pl.pf_waves = atoi(e) == 4 && c->rows_per_lane == 1 ? 4 : 2;
    // grid of the pair kernel: the tile count of the previous step on this CSR (+12%), or a guess before the
    // first sync; the kernel strides, so a wrong hint costs balance, never results
    pl.tile_hint = (int)std::min<int64_t>(pl.tile_cap, c->last_tiles > 0 ? c->last_tiles + c->last_tiles / 8 + 64
                                                                          : std::max<int64_t>(4096, c->n / 8));
    pl.cand_cap_shard = (int)std::min<int64_t>(c->cand_cap_shard, INT32_MAX / CAND_SHARDS);
    if (const char *e = getenv("BFK_CAND_CAP_SHARD"))  // test knob: a small queue forces the overflow recovery path
        pl.cand_cap_shard = std::max(1, std::min(pl.cand_cap_shard, atoi(e)));
    pl.edge_cap = (int)std::min<int64_t>(c->edge_cap, INT32_MAX);
    pl.gkey = c->d_gkey;
    pl.gcnt = c->d_gcnt;
    pl.gslots = (unsigned)c->gslots;
    pl.dbg = getenv("BFK_PF_DEBUG") ? atoi(getenv("BFK_PF_DEBUG")) : 0;
#ifndef BFK_WITH_PF_DEBUG
    if (pl.dbg & 7) {
        static bool told = false;
        if (!told) fprintf(stderr, "[bfk] BFK_PF_DEBUG needs a library built with `make PF_DEBUG=1`; ignored\n");
        told = true;
        pl.dbg &= ~7;
    }
#endif
    pl.dbg_t = nullptr;
    if (pl.dbg & 4) {
        static unsigned long long *dbg_buf = nullptr;
        static int64_t dbg_cap = 0;
        if (dbg_cap < c->tile_cap) {
            if (dbg_buf) (void)hipFree(dbg_buf);
            (void)hipMalloc((void **)&dbg_buf, (size_t)c->tile_cap * PF_WAVES_MAX * 8 * 8);
            dbg_cap = c->tile_cap;
        }
        (void)hipMemsetAsync(dbg_buf, 0, (size_t)c->tile_cap * PF_WAVES_MAX * 8 * 8, c->stream);
        pl.dbg_t = dbg_buf;
    }
    pl.indptr = c->d_indptr;
    pl.indices = c->d_indices;
    pl.ctr = (Counters *)c->d_head;
    pl.hist3 = (int *)(c->d_head + sizeof(Counters));
    pl.start3 = c->d_start3;
    pl.rowkey = c->d_rowkey;
    pl.rowrank = c->d_rowrank;
    pl.tile_slots = c->d_tile_slots;
    pl.chain = c->d_chain;
    pl.start3c = c->d_start3c;
    pl.hist_copies = c->hist_copies;
    pl.srec = c->d_srec;
    pl.parent = c->d_parent;
    pl.sig1 = c->d_sig1;
    pl.sigu1 = c->d_sigu1;
    pl.sigu2 = c->d_sigu2;
    pl.tiles = c->d_tiles;
    pl.cand = c->d_cand;
    pl.candk = c->d_candk;
    pl.edges = c->edge_capture ? c->d_edges : nullptr;
    pl.edge_sel = c->edge_capture && c->edge_sel_on ? c->d_edge_sel : nullptr;
    pl.labels = (int *)d_labels_out;
    pl.join = 0;
    if (allow_join && join_wanted(c, max_dist)) {
        if (int rc = ctx_size_join(c)) return rc;
        char *tab0 = c->d_join, *tab1 = tab0 + c->join_slots * 8;
        char *bits0 = tab1 + c->join_slots * 8, *bits1 = bits0 + c->join_bits / 8;
        if (c->join_clear) {
            HIP_TRY(hipMemsetAsync(tab0, 0xFF, (size_t)c->join_slots * 16, c->stream));
            HIP_TRY(hipMemsetAsync(bits0, 0, (size_t)c->join_bits / 4, c->stream));
            c->join_clear = false;
        }
        const int cur = c->join_parity;
        c->join_parity ^= 1;
        pl.join = 1;
        pl.ja.tab = (unsigned long long *)(cur ? tab1 : tab0);
        pl.ja.tab_next = (unsigned long long *)(cur ? tab0 : tab1);
        pl.ja.bits = (uint32_t *)(cur ? bits1 : bits0);
        pl.ja.bits_next = (uint32_t *)(cur ? bits0 : bits1);
        pl.ja.rowhash = (uint2 *)(bits1 + c->join_bits / 8);
        // While no row of the bound CSR is longer than JOIN_INLINE_ROW tokens, k_join decides every match itself — the
        // positional certificate, and for what that cannot decide the exact count by the wave — and the step has no
        // k_verify launch: a property of the CSR, not of an earlier step (round 2 skipped the launch after a synced step
        // on the same CSR had left the queue empty; round 3 first launched it always: 4.7 us of an empty kernel).
        // k_flatten checks that the queue stayed empty; BFK_JOIN_INLINE=0: queue + k_verify as for longer rows
        pl.ja.inline_exact = c->kcap <= JOIN_INLINE_ROW ? 1 : 0;
        if (const char *e = getenv("BFK_JOIN_INLINE")) pl.ja.inline_exact = pl.ja.inline_exact && atoi(e) != 0;
        pl.join_skip_verify = pl.ja.inline_exact;
        // k_verify only sees what k_join could not certify itself (rows in no common order, rows over 64 tokens)
        if (!getenv("BFK_VERIFY_GRID")) pl.verify_grid = 256;
        pl.ja.dups = (int2 *)((char *)pl.ja.rowhash + (c->n + 16) * 8);
        pl.ja.dup_cap = (int)std::min<int64_t>(4 * c->n + 65536, INT32_MAX);
        pl.ja.stats = (int *)((char *)pl.ja.dups + (4 * c->n + 65536) * 8);
        pl.ja.batch_row = pl.ja.stats + 2 * (c->nnz / (16 * JOIN_TPW) + 2);
        pl.ja.mask = (uint32_t)(c->join_slots - 1);
        pl.ja.bmask = (uint32_t)(c->join_bits - 1);
        pl.ja.dbg = getenv("BFK_JOIN_DEBUG") ? atoi(getenv("BFK_JOIN_DEBUG")) : 0;
        // a text step whose bind is still open: token count, longest row and the tokeniser's failure flags are read on the device
        pl.ja.dyn = c->spec_enqueueing ? c->d_small : nullptr;
        pl.ja.dyn_host = c->spec_enqueueing ? c->spec_host : nullptr;
    }
    if (c->spec_enqueueing && !pl.join) return fail(BFK_ESTATE, "internal: a device-driven text step needs the variant join");
    pl.pg = 0;
    if (!pl.join && pg_wanted(c, max_dist, n_shards)) {
        if (int rc = pg_key_bits(c, &pl.pg_tb)) return rc;
        // positional filter (k_pgplace): the composite {token key : slot} it bisects on needs 3 bits above the token's and must
        // stay below PG_NONE; BFK_PG_POS=0 walks whole groups (round 2's walk)
        // (at every size since the walk of labels-only steps is k_pgwalk16: 10k rows, max-dist 5: 0.35 ms without, 0.23 with)
        pl.pg_pb = pl.pg_tb + 3 <= 31 ? 3 : 0;
        if (const char *e = getenv("BFK_PG_POS")) pl.pg_pb = atoi(e) && pl.pg_tb + 3 <= 31 ? 3 : 0;
        // SHORT records exist only for rows of <= 2 * max_dist tokens: a CSR without a row of <= 2 * PG_MAX_DIST (the bind counted
        // them) sorts max_dist + 1 records per row instead of max_dist + 2 (the slot is the tail of the position-major input)
        pl.pg_has_short = (c->n_short == 0 && pl.pg_pb && !(getenv("BFK_PG_SHORT") && atoi(getenv("BFK_PG_SHORT")) == 1)) ? 0 : 1;
        pl.pg_dense = c->max_tok >= 0 && c->max_tok < (1 << PG_CNT_BITS) ? 1 : 0;
        if (const char *e = getenv("BFK_PG_DENSE")) pl.pg_dense = pl.pg_dense && atoi(e) != 0;
        pl.pg_walk16 = 1;  // (BFK_PG_WALK16=0: a wave per row, k_pgjoin)
        if (const char *e = getenv("BFK_PG_WALK16")) pl.pg_walk16 = atoi(e) != 0;
        size_t tb = 0;
        if (int rc = ctx_size_pg(c, max_dist + 2, pl.pg_tb, &tb)) return rc;
        pl.pg = 1;
        pl.pg_recs = max_dist + 2;
        pl.pg_cnt = c->pg_cnt;
        pl.pg_keys = c->pg_keys;
        pl.pg_keys_s = c->pg_keys_s;
        pl.pg_rows = c->pg_rows;
        pl.pg_rows_s = c->pg_rows_s;
        pl.pg_temp = c->pg_temp;
        pl.pg_temp_bytes = tb;
        pl.pg_srec = c->pg_srec;
        pl.pg_keys_pm = c->pg_keys_pm;
        pl.pg_recpos = c->pg_recpos;
        pl.pg_rowinfo = c->pg_rowinfo;
    }
    c->plan = pl;
    if (pl.join) {  // (the join touches the counters only: a histogram made stale by a bind waits for the step that uses it)
        if (c->ctr_dirty) HIP_TRY(hipMemsetAsync(c->d_head, 0, sizeof(Counters), c->stream));
        c->ctr_dirty = false;
    } else if (c->need_zero || c->ctr_dirty) {  // steady state: k_cells leaves counters and histogram clean for the next step
        HIP_TRY(hipMemsetAsync(c->d_head, 0, sizeof(Counters) + (size_t)c->hist_ints_cap * 4, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_chain, 0, (size_t)(c->bins_cap / 1024 + 2) * 8, c->stream));
        c->need_zero = false;
        c->ctr_dirty = false;
    }
    hipEvent_t *evs = c->profiling ? c->ev[c->n_prof_calls++ % bfk_ctx::EV_SLOTS] : nullptr;
    if (int e = launch_pipeline(pl, c->stream, evs))
        return fail(BFK_EHIP, std::string("kernel launch: ") + hipGetErrorString((hipError_t)e));
    return BFK_OK;
}

extern "C" int bfk_ctx_merge_labels(bfk_ctx *c, const void *d_gathered, int32_t n_parts, void *d_labels_out,
                                    void *d_changed) {
    if (int rc = ctx_enter(c)) return rc;
    if (!c->ran) return fail(BFK_ESTATE, "bfk_ctx_merge_labels before bfk_ctx_cluster");
    if (c->n == 0) return BFK_OK;
    if (!d_gathered || !d_labels_out || n_parts <= 0) return fail(BFK_EARG, "bad merge arguments");
    if (d_changed) HIP_TRY(hipMemsetAsync(d_changed, 0, 4, c->stream));
    // the part gathered from this very forest adds nothing (all-gather form: part = shard index); sparse forests
    // (max_dist <= 2) are merged by splicing like the verify kernel hooks them
    const int skip = (n_parts == c->plan.n_shards && n_parts > 1 && !getenv("BFK_MERGE_ALL")) ? c->plan.shard : -1;
    int splice = c->last_d <= 2 ? 1 : 0;
    if (const char *e = getenv("BFK_UF_LINK")) splice = atoi(e) != 0;
    if (int e = launch_merge(c->d_parent, (int)c->n, (const int *)d_gathered, n_parts, (int *)d_labels_out,
                             (int *)d_changed, (Counters *)c->d_head, skip, splice, c->stream))
        return fail(BFK_EHIP, std::string("merge launch: ") + hipGetErrorString((hipError_t)e));
    return BFK_OK;
}

// k_verify leaves its per-block edge / candidate counts in plain stores (no same-word atomics): add them to
// the counters read back from the device (n_edges already holds the edges of the long-pair kernel)
static int ctx_pair_stats(bfk_ctx *c, Counters *h) {
    std::vector<int> v(c->plan.verify_adaptive && c->plan.verify_phases <= 1 && !c->plan.join
                           ? (size_t)2 * c->plan.verify_grid + (size_t)2 * c->plan.verify_grid2
                           : (size_t)2 * c->plan.verify_grid * (c->plan.verify_phases > 1 ? 2 : 1));
    if (c->plan.join && c->plan.join_skip_verify) v.clear();  // k_verify did not run: its counts are another step's
    if (!v.empty()) HIP_TRY(hipMemcpy(v.data(), c->d_blk_stats, v.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long e = 0, k = 0;
    for (size_t i = 0; i < v.size(); i += 2) {
        e += (unsigned)v[i];
        k += (unsigned)v[i + 1];
    }
    if (c->plan.join) {  // pairs k_join certified and hooked itself
        const int64_t blocks = std::max<int64_t>(1, (c->nnz + 16 * JOIN_TPW - 1) / (16 * JOIN_TPW));
        std::vector<int> j((size_t)2 * blocks);
        HIP_TRY(hipMemcpy(j.data(), c->plan.ja.stats, j.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < j.size(); i += 2) {
            e += (unsigned)j[i];
            k += (unsigned)j[i + 1];
        }
    }
    h->n_edges += e;
    h->n_cand_total = k;
    return BFK_OK;
}

// The candidate queue overflowed (very dense input: far more first-level hits than 32*N): the forest
// already holds every edge that was verified, the dropped ones are recovered by re-running prefilter +
// verify over slices of the work list small enough for the queue, halving a slice that still overflows.
// Unions are idempotent, so re-verifying pairs is harmless.  Synchronous (called from bfk_ctx_sync).
static int ctx_recover_overflow(bfk_ctx *c, Counters *h, int64_t *n_slices) {  // slices are unit ranges
    Plan &pl = c->plan;
    const int ulo = 0, uhi = (int)h->n_work;  // tile index range (every shard walks its own stride of it)
    std::vector<std::pair<int, int>> todo;
    const int step0 = std::max(1, (uhi - ulo) / 16);
    for (int b = uhi; b > ulo; b -= step0) todo.push_back({std::max(ulo, b - step0), b});
    h->n_edges = h->n_cand_total = h->n_edges_cap = h->n_connected = 0;
    unsigned long long edges_acc = 0, cand_acc = 0, conn_acc = 0;  // (a slice that overflows is redone: only finished slices count)
    while (!todo.empty()) {
        auto [b, e] = todo.back();
        todo.pop_back();
        for (auto &x : h->ncand) x = 0;
        h->overflow = 0;
        h->n_edges = h->n_cand_total = h->n_connected = 0;
        const unsigned long long cap_mark = h->n_edges_cap;
        HIP_TRY(hipMemcpyAsync(c->d_head, h, sizeof(Counters), hipMemcpyHostToDevice, c->stream));
        if (int er = launch_pairs(pl, b, e, c->stream, nullptr))
            return fail(BFK_EHIP, std::string("recovery launch: ") + hipGetErrorString((hipError_t)er));
        HIP_TRY(hipMemcpyAsync(h, c->d_head, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        (*n_slices)++;
        if (h->overflow) {
            h->n_edges_cap = cap_mark;  // drop this slice's captured edges, it is redone
            if (e - b > 1) {
                const int mid = b + (e - b) / 2;
                todo.push_back({mid, e});
                todo.push_back({b, mid});
            } else {  // one tile alone overflows the queue in use: use all of the allocation, grow it if that was all
                const int full = (int)std::min<int64_t>(c->cand_cap_shard, INT32_MAX / CAND_SHARDS);
                if (pl.cand_cap_shard >= full) {
                    if (int rc = ctx_size_cand(c, c->cand_cap_total * 2)) return rc;
                    pl.cand = c->d_cand;
                    pl.candk = c->d_candk;
                }
                pl.cand_cap_shard = (int)std::min<int64_t>(c->cand_cap_shard, INT32_MAX / CAND_SHARDS);
                todo.push_back({b, e});
            }
            continue;
        }
        if (int rc = ctx_pair_stats(c, h)) return rc;
        edges_acc += h->n_edges;
        cand_acc += h->n_cand_total;
        conn_acc += h->n_connected;
    }
    h->n_edges = edges_acc;
    h->n_cand_total = cand_acc;
    h->n_connected = conn_acc;
    h->overflow = 0;
    HIP_TRY(hipMemcpyAsync(c->d_head, h, sizeof(Counters), hipMemcpyHostToDevice, c->stream));
    if (int er = launch_flatten(pl, c->stream, nullptr))
        return fail(BFK_EHIP, std::string("recovery flatten: ") + hipGetErrorString((hipError_t)er));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFK_OK;
}

extern "C" int bfk_ctx_sync(bfk_ctx *c, bfk_stats *out) {
    if (int rc = ctx_enter(c)) return rc;  // (completes the bind of a device-driven text step, redoing it where it has to)
    HIP_TRY(hipStreamSynchronize(c->stream));
    bfk_stats s{};
    s.n_rows = c->n < 0 ? 0 : c->n;
    s.nnz = c->nnz;
    s.max_row_len = c->kcap;
    if (c->ran && c->n > 0) {
        Counters h;
        HIP_TRY(hipMemcpy(&h, c->d_head, sizeof(Counters), hipMemcpyDeviceToHost));
        if (h.err || h.err_rows) c->need_zero = c->ctr_dirty = true;
        c->last_tiles = (int64_t)h.n_work;
        if (h.err_rows) return fail(BFK_EARG, "CSR changed after bind: a row is longer than at bind time");
        if (h.err & ERR_WORKCAP) return fail(BFK_EOVERFLOW, "band work list overflow (input too large for 32-bit unit counts)");
        if (h.err & ERR_LABEL) return fail(BFK_EARG, "merge: label out of range");
        if (getenv("BFK_DEBUG"))
            fprintf(stderr, "[bfk] k_cells %.1f us, %u tiles; prefix groups: %d, estimate %llu sampled members, gave up %d\n",
                    (h.dbg[1] - h.dbg[0]) / 100.0, h.n_work, c->plan.pg, h.pg_est, h.pg_fail);
        if (c->plan.dbg_t) {
            std::vector<unsigned long long> t((size_t)h.n_work * c->plan.pf_waves * 8);
            (void)hipMemcpy(t.data(), c->plan.dbg_t, t.size() * 8, hipMemcpyDeviceToHost);
            if (FILE *f = fopen("gpurun_out/pf_waves.txt", "w")) {
                for (size_t i = 0; i < t.size(); i += 8)
                    fprintf(f, "%zu %llu %llu %llu %llu %llu %llu %llu %llu\n", i / 8, t[i], t[i + 1], t[i + 2], t[i + 3], t[i + 4],
                            t[i + 5], t[i + 6], t[i + 7]);
                fclose(f);
            }
        }
        int64_t retry_slices = 0;
        if (c->plan.join && (h.join_fail || h.overflow)) {
            // the variant join gave up (a probe chain beyond JOIN_MAX_PROBE: hundreds of rows that are one multiset)
            // or its candidates did not fit the queue: the step is redone on the all-pairs path, which has its own
            // recovery; a give-up also turns the join off for this CSR
            if (h.join_fail) c->join_off = true;
            c->need_zero = c->ctr_dirty = true;
            c->join_clear = true;
            if (int rc = ctx_enqueue(c, c->last_d, c->plan.shard, c->plan.n_shards, c->plan.labels, false)) return rc;
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpy(&h, c->d_head, sizeof(Counters), hipMemcpyDeviceToHost));
            c->last_tiles = (int64_t)h.n_work;
            retry_slices = 1;
        }
        if (c->plan.pg && h.pg_fail) {
            // the prefix groups are too big to pay (the walk did nothing): the step is redone on the band kernels
            c->pg_off = true;
            c->need_zero = c->ctr_dirty = true;
            if (int rc = ctx_enqueue(c, c->last_d, c->plan.shard, c->plan.n_shards, c->plan.labels, false)) return rc;
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpy(&h, c->d_head, sizeof(Counters), hipMemcpyDeviceToHost));
            c->last_tiles = (int64_t)h.n_work;
            retry_slices = 1;
        }
        if (!h.overflow)
            if (int rc = ctx_pair_stats(c, &h)) return rc;
        // A max-dist 2 step on the band kernels whose queue took many candidates per row (a star-like phylogeny: hub profiles with
        // thousands of neighbours; 20 per row against 2 on tree-like inputs): the prefix groups do such a CSR 1.2-1.45x faster at
        // every size (tools/d2_crossover.py) — a property of the CSR that only a step reveals; later steps on it take them.
        if (!c->plan.join && !c->plan.pg && c->last_d == 2 && c->last_shards == 1 && c->n >= 10000 && !h.overflow &&
            (int64_t)h.n_cand_total > 8 * c->n)
            c->pg_dense_d2 = true;
        if (h.overflow) {
            if (int rc = ctx_recover_overflow(c, &h, &retry_slices)) return rc;
            // the queue was too small for this input: double it so that the next run fits in one pass (not when
            // the run was held below the allocation by the test knob, and never beyond 256 slots per row)
            if (c->plan.cand_cap_shard >= (int)std::min<int64_t>(c->cand_cap_shard, INT32_MAX / CAND_SHARDS) &&
                c->cand_cap_total < 256 * std::max<int64_t>(c->n, 4096))
                if (int rc = ctx_size_cand(c, c->cand_cap_total * 2)) return rc;
        }
        const int64_t n = c->n;
        s.pairs_resolved = n * (n - 1) / 2 / c->last_shards;
        if (out) {  // pairs of the reference's length band (|k_i - k_j| <= d): row counts per length from start3
            const int planes = c->kcap + 2;
            const size_t pitch = (size_t)c->fb * c->gb * c->hb * sizeof(int);
            std::vector<int> sk((size_t)planes);
            if (c->plan.join || c->plan.pg) {  // no cell histogram on the join / prefix-group paths: count the row lengths from indptr
                std::vector<int> ip((size_t)n + 1);
                HIP_TRY(hipMemcpy(ip.data(), c->d_indptr, ip.size() * 4, hipMemcpyDeviceToHost));
                std::fill(sk.begin(), sk.end(), 0);
                for (int64_t i = 0; i < n; i++) {
                    const int k = std::min(std::max(ip[(size_t)i + 1] - ip[(size_t)i], 0), c->kcap);
                    sk[(size_t)k + 1]++;
                }
                for (int k = 1; k < planes; k++) sk[(size_t)k] += sk[(size_t)k - 1];
            } else
            HIP_TRY(hipMemcpy2D(sk.data(), sizeof(int), c->d_start3, pitch, sizeof(int), (size_t)planes, hipMemcpyDeviceToHost));
            const int d = c->last_d;
            long double acc = 0;
            for (int k = 0; k + 1 < planes; k++) {
                const long double ck = (long double)(sk[(size_t)k + 1] - sk[(size_t)k]);
                acc += ck * (ck - 1) / 2;
                for (int dl = 1; dl <= d && k + dl + 1 < planes; dl++) acc += ck * (long double)(sk[(size_t)k + dl + 1] - sk[(size_t)k + dl]);
            }
            s.pairs_in_band = (int64_t)acc;
        }
        if (out) {  // pair slots the prefilter evaluated: per-tile counts written by the waves, summed here
            std::vector<int> ts(c->plan.pg || c->plan.join ? (size_t)0 : (size_t)h.n_work * c->plan.pf_waves);
            if (!ts.empty()) HIP_TRY(hipMemcpy(ts.data(), c->d_tile_slots, ts.size() * 4, hipMemcpyDeviceToHost));
            int64_t acc = c->plan.pg ? (int64_t)h.pairs_filtered : 0;  // prefix groups: members of the rows' groups visited
            for (size_t t = 0; t < ts.size(); t++) acc += ts[t];  // tiles of other ranks' cells hold 0
            s.pairs_filtered = c->plan.join ? (c->nnz + n) / c->last_shards : acc;  // join: table lookups
        }
        s.n_candidates = (int64_t)h.n_cand_total;
        s.n_edges = (int64_t)h.n_edges;
        s.n_connected = (int64_t)h.n_connected;
        s.n_retry_slices = retry_slices;
        s.sig_words = c->last_w1;
        s.n_work_items = (int32_t)h.n_work;
        s.path = c->plan.join ? 1 : (c->plan.pg ? 2 : 0);
        s.n_gpus_used = 1;
        c->last_tiles = (int64_t)h.n_work;
        if (c->profiling && c->n_prof_calls > 0) {
            const int used = std::min(c->n_prof_calls, (int)bfk_ctx::EV_SLOTS);
            double acc[5] = {0, 0, 0, 0, 0};
            bool ok = true;
            for (int i = 0; i < used && ok; i++) {
                hipEvent_t *e = c->ev[i];
                float v[5];
                ok = hipEventElapsedTime(&v[0], e[0], e[1]) == hipSuccess && hipEventElapsedTime(&v[1], e[1], e[2]) == hipSuccess &&
                     hipEventElapsedTime(&v[2], e[2], e[3]) == hipSuccess && hipEventElapsedTime(&v[3], e[3], e[4]) == hipSuccess &&
                     hipEventElapsedTime(&v[4], e[0], e[4]) == hipSuccess;
                for (int k = 0; k < 5; k++) acc[k] += v[k];
            }
            if (ok) {
                s.profiled = used;
                s.ms_prep = (float)(acc[0] / used);
                s.ms_prefilter = (float)(acc[1] / used);
                s.ms_verify = (float)(acc[2] / used);
                s.ms_flatten = (float)(acc[3] / used);
                s.ms_total = (float)(acc[4] / used);
            }
            c->n_prof_calls = 0;
        }
    }
    c->stats = s;
    if (out) *out = s;
    c->unsettled = false;  // (the run — redone where it had to be — is complete)
    return BFK_OK;
}

extern "C" int bfk_ctx_upload(bfk_ctx *c, const void *h_src, void *d_dst, int64_t bytes) {
    if (int rc = ctx_enter(c)) return rc;
    if (bytes < 0 || (bytes > 0 && (!h_src || !d_dst))) return fail(BFK_EARG, "bad upload arguments");
    if (bytes == 0) return BFK_OK;
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFK_OK;
}

extern "C" int bfk_ctx_download(bfk_ctx *c, const void *d_src, void *h_dst, int64_t bytes) {
    if (int rc = ctx_enter(c)) return rc;
    if (bytes < 0 || (bytes > 0 && (!d_src || !h_dst))) return fail(BFK_EARG, "bad download arguments");
    if (bytes == 0) return BFK_OK;
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFK_OK;
}

extern "C" int bfk_ctx_device_alloc(bfk_ctx *c, int64_t bytes, void **d_out) {
    if (int rc = ctx_enter(c)) return rc;
    if (!d_out || bytes < 0) return fail(BFK_EARG, "bad alloc arguments");
    hipError_t e = hipMalloc(d_out, (size_t)std::max<int64_t>(bytes, 16));
    if (e != hipSuccess) return fail(BFK_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return BFK_OK;
}

extern "C" int bfk_ctx_device_free(bfk_ctx *c, void *d_ptr) {
    if (int rc = ctx_enter(c)) return rc;
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return BFK_OK;
}

extern "C" int bfk_ctx_edges(bfk_ctx *c, int32_t **edges_out, int64_t *n_edges_out) {
    if (int rc = ctx_enter(c)) return rc;
    if (!edges_out || !n_edges_out) return fail(BFK_EARG, "null output");
    if (!c->ran || !c->edge_capture) return fail(BFK_ESTATE, "edge capture was not enabled for the last run");
    HIP_TRY(hipStreamSynchronize(c->stream));
    int64_t ne = 0;
    if (c->n > 0) {
        Counters h;
        HIP_TRY(hipMemcpy(&h, c->d_head, sizeof(Counters), hipMemcpyDeviceToHost));
        ne = (int64_t)h.n_edges_cap;
        if (ne > c->edge_cap) return fail(BFK_EOVERFLOW, "edge buffer overflow");
    }
    int32_t *out = (int32_t *)malloc(std::max<size_t>(8, (size_t)ne * 8));
    if (!out) return fail(BFK_ENOMEM, "out of memory");
    if (ne > 0) HIP_TRY(hipMemcpy(out, c->d_edges, (size_t)ne * 8, hipMemcpyDeviceToHost));
    if (ne > 1) {  // a recovered (sliced) run may report an edge more than once
        int64_t *e64 = reinterpret_cast<int64_t *>(out);
        std::sort(e64, e64 + ne);
        ne = std::unique(e64, e64 + ne) - e64;
    }
    *edges_out = out;
    *n_edges_out = ne;
    return BFK_OK;
}

// ================================================================================================
// one-shot entry points on a lazily created default context (serialised by a mutex)
// ================================================================================================
static std::mutex g_mu;
static bfk_ctx *g_default = nullptr;

static int default_ctx(bfk_ctx **out) {
    if (!g_default) {
        int dev = 0;
        if (const char *e = getenv("BFK_DEVICE")) dev = atoi(e);
        if (int rc = bfk_ctx_create(dev, &g_default)) return rc;
    }
    *out = g_default;
    return BFK_OK;
}

// labels buffer of the one-shot entry points, kept with the context (no allocation per call)
static int ctx_own_labels(bfk_ctx *c, int64_t n_rows) { return dev_realloc(&c->own_labels, &c->own_labels_cap, std::max<int64_t>(n_rows, 1)); }

// Everything a first call pays that does not depend on the input: device context, stream, code-object load (first
// launch), and — with size hints — the workspace allocations.  bfk_preload_start (bfk_base.cpp) runs it on a thread
// while the input is parsed.
extern "C" int bfk_warmup(int device, int64_t rows_hint, int64_t nnz_hint) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_default) {
        if (const char *e = getenv("BFK_DEVICE")) device = atoi(e);
        if (int rc = bfk_ctx_create(device, &g_default)) return rc;
    }
    bfk_ctx *c = g_default;
    if (int rc = ctx_enter(c)) return rc;
    HIP_TRY(hipMemsetAsync(c->d_small, 0, 64, c->stream));
    if (int e = launch_maxlen(c->d_small, 1, c->d_small + 4, c->stream))  // any kernel: loads the code object
        return fail(BFK_EHIP, std::string("warm-up launch: ") + hipGetErrorString((hipError_t)e));
    if (rows_hint > 0 && nnz_hint > 0 && rows_hint < ((int64_t)1 << 27) && nnz_hint < ((int64_t)1 << 30)) {
        const int64_t n0 = c->n, nnz0 = c->nnz;
        const int k0 = c->kcap;
        int rc = dev_realloc(&c->own_indptr, &c->own_n_cap, rows_hint + 1);
        if (!rc) rc = dev_realloc(&c->own_indices, &c->own_nnz_cap, nnz_hint + 1);
        if (!rc) rc = ctx_own_labels(c, rows_hint);
        c->n = rows_hint;
        c->nnz = nnz_hint;
        c->kcap = 128;
        if (!rc) rc = ctx_size_workspace(c, 1);
        if (!rc) rc = ctx_size_join(c);
        c->n = n0;
        c->nnz = nnz0;
        c->kcap = k0;
        c->need_zero = c->ctr_dirty = true;
        c->join_clear = true;
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFK_OK;
}

// n_gpus > 1 from ONE process: a context per device, every device clusters its shard of the work (the CSR is replicated,
// SURVEY 8e), the label arrays are copied to the first device over xGMI (peer copies) and merged there like after an
// all_gather.  (The one-process-per-GPU form with RCCL collectives is breakfast_amd/distributed.py; this is the same
// split behind the one-shot C-ABI, for callers without a process launcher.)  BFK_MULTI_ONE_DEVICE=1 rehearses it with
// every context on device 0.
static std::vector<bfk_ctx *> g_multi;

static int cluster_multi(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int32_t max_dist, int32_t n_gpus,
                         int32_t *labels_out, bfk_stats *stats_out) {
    const bool one = getenv("BFK_MULTI_ONE_DEVICE") && atoi(getenv("BFK_MULTI_ONE_DEVICE")) != 0;
    const int ndev = bfk_device_count();
    if (!one && n_gpus > ndev) return fail(BFK_ENODEV, "bfk_cluster_csr: n_gpus = " + std::to_string(n_gpus) + " but " + std::to_string(ndev) + " gfx950 device(s) visible");
    while ((int)g_multi.size() < n_gpus) {
        bfk_ctx *c = nullptr;
        if (int rc = bfk_ctx_create(one ? 0 : (int)g_multi.size(), &c)) return rc;
        g_multi.push_back(c);
    }
    // one host thread per device: the CSR goes up to all devices at once (every device has its own PCIe link; one after the
    // other, each upload with its syncs, was n_gpus x 3 ms at 1M rows before the first kernel ran) and each device's shard is
    // enqueued as soon as ITS copy has landed
    {
        std::vector<int> rcs((size_t)n_gpus, BFK_OK);
        std::vector<std::string> msgs((size_t)n_gpus);
        auto work = [&](int g) {
            bfk_ctx *c = g_multi[(size_t)g];
            int rc = bfk_ctx_upload_csr(c, indptr, indices, n_rows);
            if (!rc) rc = ctx_own_labels(c, n_rows);
            if (!rc) rc = bfk_ctx_cluster(c, max_dist, g, n_gpus, c->own_labels);
            rcs[(size_t)g] = rc;
            if (rc) msgs[(size_t)g] = bfk_last_error();  // (thread-local: carried over to the caller's thread below)
        };
        std::vector<std::thread> th;
        for (int g = 1; g < n_gpus; g++) th.emplace_back(work, g);
        work(0);
        for (auto &t : th) t.join();
        for (int g = 0; g < n_gpus; g++)
            if (rcs[(size_t)g]) return fail(rcs[(size_t)g], "device " + std::to_string(g) + ": " + msgs[(size_t)g]);
    }
    bfk_stats total{};
    for (int g = 0; g < n_gpus; g++) {
        bfk_stats st{};
        if (int rc = bfk_ctx_sync(g_multi[(size_t)g], &st)) return rc;
        if (g == 0) total = st;
        else {
            total.pairs_filtered += st.pairs_filtered;
            total.n_candidates += st.n_candidates;
            total.n_edges += st.n_edges;
            total.n_retry_slices += st.n_retry_slices;
        }
    }
    total.pairs_resolved = n_rows * (n_rows - 1) / 2;
    total.n_gpus_used = n_gpus;
    bfk_ctx *c0 = g_multi[0];
    if (int rc = ctx_enter(c0)) return rc;
    if (n_rows > 0) {
        if (int rc = dev_realloc(&c0->own_gather, &c0->own_gather_cap, n_rows * n_gpus)) return rc;
        for (int g = 0; g < n_gpus; g++)
            HIP_TRY(hipMemcpyPeerAsync(c0->own_gather + (size_t)g * n_rows, c0->device, g_multi[(size_t)g]->own_labels,
                                       g_multi[(size_t)g]->device, (size_t)n_rows * 4, c0->stream));
        if (int rc = bfk_ctx_merge_labels(c0, c0->own_gather, n_gpus, c0->own_labels, nullptr)) return rc;
        if (int rc = bfk_ctx_sync(c0, nullptr)) return rc;
        if (int rc = bfk_ctx_download(c0, c0->own_labels, labels_out, n_rows * 4)) return rc;
    }
    if (stats_out) *stats_out = total;
    return BFK_OK;
}

// Is a one-shot call worth several devices?  Every device needs the whole CSR (an upload each, in parallel), the shards'
// labels have to meet on one device (peer copies) and be merged there (k_merge: 0.07 - 0.2 ms at 1M rows).  What the extra
// devices take off is the SHARDED part of the step (pair kernel / walk + verify; the prep is replicated).  One-device rehearsal
// of the split, each shard alone on the GPU (profiles/r03_rehearsal_shard_tables.json, 1M rows, kernels of the slowest shard at
// 1 / 8 ranks): max-dist 1: 0.48 / 0.25 ms (+ 0.11 merge: break-even); max-dist 2: 1.09 on the groups / 0.50 on the band
// kernels the shards run (+ 0.2: a win); max-dist 5 on the prefix groups: 1.81 / 1.26 at 4 ranks / 1.16 at 8 (+ 0.2 merge + the
// exchange: a win from 4 ranks, nothing at 2).  So: several devices at max-dist 2 from 300k rows, at max-dist >= 3 from 500k
// rows on at least 4 devices, at max-dist 1 from 2M; everything else runs on one device whatever n_gpus says
// (bfk_stats.n_gpus_used tells).  BFK_MULTI_FORCE=1 (and the one-device rehearsal mode of the tests) shards regardless.
static bool multi_worth(int64_t n_rows, int32_t max_dist, int32_t n_gpus) {
    if (getenv("BFK_MULTI_FORCE") && atoi(getenv("BFK_MULTI_FORCE")) != 0) return true;
    if (getenv("BFK_MULTI_ONE_DEVICE") && atoi(getenv("BFK_MULTI_ONE_DEVICE")) != 0) return true;
    if (max_dist <= 1) return n_rows >= 2000000;
    if (max_dist == 2) return n_rows >= 300000;
    // dense graphs (prefix groups): records and sort are replicated, walk and verify shard — 1M rows, max_dist 5, one-device
    // rehearsal: 1.81 ms on one rank, 1.46 / 1.27 / 1.18 ms + exchange + merge (~0.3 ms) on 2 / 4 / 8
    return n_rows >= 500000 && n_gpus >= 4;
}

extern "C" int bfk_cluster_csr(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int32_t max_dist,
                               int32_t n_gpus, int32_t *labels_out, bfk_stats *stats_out) {
    if (n_rows < 0 || !indptr || (n_rows > 0 && !labels_out)) return fail(BFK_EARG, "bad arguments");
    if (n_gpus < 1 || n_gpus > 64) return fail(BFK_EARG, "n_gpus must be 1..64");
    std::lock_guard<std::mutex> lk(g_mu);
    if (n_gpus > 1) {
        const bool one = getenv("BFK_MULTI_ONE_DEVICE") && atoi(getenv("BFK_MULTI_ONE_DEVICE")) != 0;
        const int ndev = bfk_device_count();
        if (!one && n_gpus > ndev) return fail(BFK_ENODEV, "bfk_cluster_csr: n_gpus = " + std::to_string(n_gpus) + " but " + std::to_string(ndev) + " gfx950 device(s) visible");
        if (multi_worth(n_rows, max_dist, n_gpus)) return cluster_multi(indptr, indices, n_rows, max_dist, n_gpus, labels_out, stats_out);
    }
    bfk_ctx *c;
    if (int rc = default_ctx(&c)) return rc;
    if (int rc = bfk_ctx_upload_csr(c, indptr, indices, n_rows)) return rc;
    if (int rc = ctx_own_labels(c, n_rows)) return rc;
    int rc = bfk_ctx_cluster(c, max_dist, 0, 1, c->own_labels);
    if (!rc) rc = bfk_ctx_sync(c, stats_out);
    if (!rc) rc = bfk_ctx_download(c, c->own_labels, labels_out, n_rows * 4);
    return rc;
}

// a1 with host outputs, computed on the device: bfk_build_csr's contract (include/bfk.h) — what the parity tests compare
// with the reference's CSR.  BFK_EUNSUPPORTED inputs (multi-byte separator, 4 GiB of text, 64 KiB tokens) are the
// host tokeniser's (bfk_build_csr).
extern "C" int bfk_build_csr_device(const char *buf, const int64_t *row_off, int64_t n_rows, const char *sep, int64_t sep_len,
                                    int32_t *indptr_out, int32_t **indices_out, int64_t *nnz_out, int32_t *n_vocab_out) {
    if (!row_off || !indptr_out || !indices_out || !nnz_out || !n_vocab_out || n_rows < 0) return fail(BFK_EARG, "bfk_build_csr_device: null argument");
    std::lock_guard<std::mutex> lk(g_mu);
    bfk_ctx *c;
    if (int rc = default_ctx(&c)) return rc;
    if (int rc = ctx_enter(c)) return rc;
    int64_t nnz = 0;
    int32_t nv = 0;
    if (int rc = ctx_build_text(c, buf, row_off, n_rows, sep, sep_len, &nnz, &nv)) return rc;
    int32_t *out = (int32_t *)malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(1, nnz));
    if (!out) return fail(BFK_ENOMEM, "bfk_build_csr_device: out of memory");
    if (int rc = bfk_ctx_download_csr(c, indptr_out, out)) {
        free(out);
        return rc;
    }
    *indices_out = out;
    *nnz_out = nnz;
    *n_vocab_out = nv;
    return BFK_OK;
}

// a1 .. a8 in one call: profile text in host memory -> canonical labels in host memory.  The text goes to the device once
// (56 GB/s from the caller's buffer), the CSR is built there (bfk_text.hip) and never visits the host.
extern "C" int bfk_cluster_text(const char *buf, const int64_t *row_off, int64_t n_rows, const char *sep, int64_t sep_len,
                                int32_t max_dist, int32_t *labels_out, bfk_stats *stats_out, int64_t *nnz_out, int32_t *n_vocab_out,
                                int32_t *indptr_out) {
    if (!row_off || n_rows < 0 || (n_rows > 0 && !labels_out)) return fail(BFK_EARG, "bfk_cluster_text: bad arguments");
    std::lock_guard<std::mutex> lk(g_mu);
    bfk_ctx *c;
    if (int rc = default_ctx(&c)) return rc;
    if (int rc = ctx_enter(c)) return rc;
    if (int r2 = ctx_own_labels(c, n_rows)) return r2;
    bool clustered = false;  // (the clustering kernels went out behind the tokeniser without a wait in between)
    int rc = ctx_build_text(c, buf, row_off, n_rows, sep, sep_len, nnz_out, n_vocab_out, max_dist, c->own_labels, &clustered);
    if (rc == BFK_EUNSUPPORTED) {  // the host tokeniser (same contract) and an upload of its CSR
        std::vector<int32_t> indptr((size_t)n_rows + 1);
        int32_t *indices = nullptr, nv = 0;
        int64_t nnz = 0;
        if (int r2 = bfk_build_csr(buf, row_off, n_rows, sep, sep_len, indptr.data(), &indices, &nnz, &nv)) return r2;
        rc = bfk_ctx_upload_csr(c, indptr.data(), indices, n_rows);
        free(indices);
        if (nnz_out) *nnz_out = nnz;
        if (n_vocab_out) *n_vocab_out = nv;
        c->tk_stats = bfk_text_stats{};
        c->tk_stats.host_fallback = 1;
    }
    if (rc) return rc;
    if (!clustered) rc = bfk_ctx_cluster(c, max_dist, 0, 1, c->own_labels);
    if (!rc && indptr_out)  // the row lengths (n_features of the reference's frame, :287) ride along with the kernels
        if (hipMemcpyAsync(indptr_out, c->d_indptr, (size_t)(n_rows + 1) * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
            rc = fail(BFK_EHIP, "bfk_cluster_text: indptr copy failed");
    if (!rc) rc = bfk_ctx_sync(c, stats_out);
    if (!rc) rc = bfk_ctx_download(c, c->own_labels, labels_out, n_rows * 4);
    return rc;
}

// ================================================================================================
// the CLI's stages between reader and writer on the device: filter_features + collapse_duplicates + sparse_feature_matrix of a
// bfk_table (breakfast.py:116-190, :72-79, :193-215) — the table's bytes cross PCIe once, everything else happens in HBM
// ================================================================================================
extern "C" int bfk_table_raw(const bfk_table *t, const char **bytes_out, int64_t *n_bytes_out, const void **feat_spans_out,
                             int64_t *span_stride_out, int64_t *n_rows_out);  // libbfk_front.so
extern "C" int bfk_table_any_high(const bfk_table *t);
extern "C" int bfk_table_feature_high(const bfk_table *t);
extern "C" int bfk_table_set_prepared(bfk_table *t, const int32_t *group, const int32_t *first_row, int64_t n_unique, const bfk_prep_info *info,
                                      const int32_t *indptr, const int32_t *indices, const char *sep2, int64_t sep2_len);
extern "C" int bfk_table_set_invalid(bfk_table *t, const int64_t *off, const int32_t *len, int64_t n);

// BFK_FRONT_TIMING=1: stage times of the device pipeline on stderr (like the host stages' StageTimer)
struct DevTimer {
    const bool on = getenv("BFK_FRONT_TIMING") && atoi(getenv("BFK_FRONT_TIMING")) != 0;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what, hipStream_t st = nullptr) {
        if (!on) return;
        if (st) (void)hipStreamSynchronize(st);
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[bfk_dev]   %-28s %8.2f ms   (at %.1f ms)\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t1.time_since_epoch()).count() - 1e3 * (double)(long long)(std::chrono::duration<double>(t1.time_since_epoch()).count() / 100) * 100);
        t0 = t1;
    }
};

constexpr uint32_t PREP_INV_CAP = 65536;  // invalid token occurrences the device prepare hands to the host (more: the host stage takes the input)

struct PrepResult {
    int64_t n_rows = 0, n_unique = 0, nnz = 0, n_invalid = 0;
    int32_t n_vocab = 0;
    bool filtering = false;
    int64_t base = 0;  // table byte of the device text's byte 0
    TokFilter flt{};   // the filter the tokeniser ran with
    unsigned char sep_byte = 0;  // the separator byte of the device text (a stand-in when --sep2 has several bytes)
    // the "Skipping invalid feature" lines in the reference's order (rows in input order, tokens in row order): spans into the
    // table's bytes, length 0 for an empty token.  Empty when every invalid token is an empty one (n_invalid lines of '').
    std::vector<int64_t> inv_off;
    std::vector<int32_t> inv_len;
};

// -> BFK_EUNSUPPORTED (nothing printed, nothing written: the host stages take the input) for what the device stages do not
// restate: token separators of over 16 bytes, 4 GiB of text, more than PREP_INV_CAP non-empty tokens that match no pattern, a feature
// with non-ASCII bytes under a grammar, two different rows with one 64-bit hash
static int ctx_prepare_table(bfk_ctx *c, const bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, PrepResult *res) {
    if (!t || !sep2 || !opts || !res) return fail(BFK_EARG, "device prepare: null argument");
    if (sep2_len <= 0) return fail(BFK_EARG, "empty separator");
    if (opts->var_type < BFK_VAR_COVSONAR_DNA || opts->var_type > BFK_VAR_RAW) return fail(BFK_EARG, "device prepare: unknown var_type");
    for (int64_t i = 0; i < sep2_len; i++) {
        const unsigned char b = (unsigned char)sep2[i];
        if (b >= 0x80 || b == '\n' || b == '\r' || b == 0) return fail(BFK_EUNSUPPORTED, "device prepare: token separator");
    }
    unsigned char sp = (unsigned char)sep2[0];  // (a separator of several bytes: the stand-in byte chosen below)
    const char *bytes = nullptr;
    const void *spans = nullptr;
    int64_t n_bytes = 0, stride = 0, n = 0;
    if (int rc = bfk_table_raw(t, &bytes, &n_bytes, &spans, &stride, &n)) return rc;
    if (n <= 0) return fail(BFK_EUNSUPPORTED, "device prepare: no rows");
    if (int rc = ctx_settle(c)) return rc;
    if (n > (int64_t)INT32_MAX - 2 * SIG_PAD_ROWS) return fail(BFK_EUNSUPPORTED, "device prepare: too many rows");
    const bool filtering = opts->skip_del || opts->skip_ins || opts->trim_start > 0 || opts->trim_end > 0;
    // (the reference's str patterns let \d match the digits of other scripts: a FEATURE with non-ASCII bytes under a grammar is
    // the pandas mirror's — the host stage declines it too; accents in ids or other columns do not matter)
    if (filtering && opts->var_type != BFK_VAR_RAW && bfk_table_feature_high(t))
        return fail(BFK_EUNSUPPORTED, "device prepare: non-ASCII bytes in a feature that is matched against the token patterns");
    struct SpanView {
        int64_t off;
        int32_t len;
    };
    auto span = [&](int64_t r) { return *(const SpanView *)((const char *)spans + (size_t)r * (size_t)stride); };
    const int64_t base = span(0).off, T = n_bytes - base;
    SepPattern pat{};
    if (sep2_len > 1)  // (several bytes: folded on the device, the stages below run with the stand-in byte)
        if (int rc = sep_stand_in(sep2, sep2_len, bytes + base, T, &pat, &sp)) return rc;
    {
        const char one = (char)sp;
        if (int rc = ctx_check_text_args(n, &one, 1, T)) return rc;
    }
    std::vector<int64_t> row_off((size_t)n + 1);
    std::vector<int32_t> row_len((size_t)n);
    {
        int64_t prev = base;
        for (int64_t r = 0; r < n; r++) {
            const SpanView sv = span(r);
            if (sv.off < prev || sv.len < 0 || sv.off + sv.len > n_bytes) return fail(BFK_EUNSUPPORTED, "device prepare: feature spans not in file order");
            row_off[(size_t)r] = sv.off;
            row_len[(size_t)r] = sv.len;
            prev = sv.off + sv.len;
        }
        row_off[(size_t)n] = n_bytes;
    }
    DevTimer tm;
    tm.lap("prepare: spans");
    const int64_t T_pad = round_up(T + 1, TOK_PAD_BYTES);
    if (int rc = dev_realloc(&c->tk_text, &c->tk_text_cap, T_pad + TOK_TEXT_SLACK, 1.05)) return rc;
    if (int rc = dev_realloc(&c->tk_rowoff, &c->tk_rowoff_cap, n + 1, 1.05)) return rc;
    if (n + 1 > c->pr_rows_cap) {
        int64_t cap;
        int rc = 0;
        const int64_t want = n + 1;
        cap = 0; rc |= dev_realloc(&c->pr_spanlen, &cap, want, 1.05);
        cap = 0; rc |= dev_realloc(&c->pr_rep, &cap, want, 1.05);
        cap = 0; rc |= dev_realloc(&c->pr_group, &cap, want, 1.05);
        cap = 0; rc |= dev_realloc(&c->pr_first, &cap, want, 1.05);
        cap = 0; rc |= dev_realloc(&c->pr_uindptr, &cap, want + 1, 1.05);
        cap = 0; rc |= dev_realloc(&c->pr_rowhash, &cap, want, 1.05);
        cap = 0; rc |= dev_realloc(&c->pr_val, &cap, want, 1.05);
        if (rc) return BFK_ENOMEM;
        c->pr_rows_cap = (int64_t)((double)want * 1.05);
    }
    if (int rc = dev_realloc(&c->pr_blk, &c->pr_blk_cap, n / 1024 + 2)) return rc;
    if (!c->pr_small && hipMalloc((void **)&c->pr_small, 64) != hipSuccess) return fail(BFK_ENOMEM, "hipMalloc failed");
    int64_t slots = 1024;
    while (slots < 2 * n) slots <<= 1;
    if (int rc = dev_realloc(&c->pr_table, &c->pr_table_cap, slots)) return rc;
    tm.lap("prepare: allocations");
    // the table's bytes from the first feature on, the rows' starts and lengths: one copy each; the caller's buffers are the
    // table's own (they outlive the call), the stream is waited for before the call returns
    HIP_TRY(hipMemcpyAsync(c->tk_rowoff, row_off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->pr_spanlen, row_len.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    // (ONE copy from the table's pageable buffer, which the driver has never seen: 31 MB in 8-10 ms, 330 MB in 18-28 ms — it pins
    // the pages on the way.  Slices copied by several host threads on streams of their own were 2-4x SLOWER: 44-84 ms at either
    // size, a stream costs more to create than the pinning it would share.)
    HIP_TRY(hipMemcpyAsync(c->tk_text, bytes + base, (size_t)T, hipMemcpyHostToDevice, c->stream));
    struct StreamGuard {
        hipStream_t s;
        ~StreamGuard() { (void)hipStreamSynchronize(s); }
    } guard{c->stream};
    tm.lap("prepare: H2D", c->stream);
    const int64_t sep_extra = sep2_len - 1;  // empty tokens every folded separator adds to the device's counts
    if (sep2_len > 1) {
        HIP_TRY(hipMemsetAsync(c->pr_small, 0, 64, c->stream));
        // (occurrences per row into pr_rep, which the collapse only fills further down; their sum into the words behind the
        // collapse's totals)
        if (int e = launch_sepfold(c->tk_text, c->tk_rowoff, c->pr_spanlen, (int)n, base, pat, (uint8_t)sp, c->pr_rep,
                                   (unsigned long long *)(c->pr_small + 8), c->stream))
            return fail(BFK_EHIP, std::string("k_sepfold launch: ") + hipGetErrorString((hipError_t)e));
    }
    if (int e = launch_blank(c->tk_text, c->tk_rowoff, c->pr_spanlen, (int)n, base, (uint32_t)T, (uint8_t)sp, c->stream))
        return fail(BFK_EHIP, std::string("k_blank launch: ") + hipGetErrorString((hipError_t)e));
    bfk_ctx::TokPlan tp;
    tp.d_text = c->tk_text;
    tp.d_rowoff = c->tk_rowoff;
    tp.base = base;
    tp.T = T;
    tp.n_rows = n;
    tp.sep = (char)sp;
    tp.d_span_len = c->pr_spanlen;
    tp.flt.on = filtering ? 1 : 0;
    tp.flt.var_type = opts->var_type;
    tp.flt.skip_ins = opts->skip_ins;
    tp.flt.skip_del = opts->skip_del;
    tp.flt.trim_start = opts->trim_start;
    tp.flt.upper = opts->reference_length - opts->trim_end;
    if (filtering) {
        if (!c->pr_inv && hipMalloc((void **)&c->pr_inv, (size_t)PREP_INV_CAP * sizeof(uint2)) != hipSuccess) return fail(BFK_ENOMEM, "hipMalloc failed");
        if (int rc = dev_realloc(&c->pr_empties, &c->pr_empties_cap, n, 1.05)) return rc;
        tp.d_inv = c->pr_inv;
        tp.inv_cap = PREP_INV_CAP;
        tp.d_row_empties = c->pr_empties;
    }
    if (int rc = ctx_text_events(c)) return rc;
    if (int rc = ctx_tokenize(c, tp, nullptr, nullptr)) return rc;  // (the CSR of ALL rows is bound now)
    tm.lap("prepare: tokenise + filter");
    std::vector<int32_t> row_seps;  // folded separators per row (several-byte separator, empty tokens counted)
    if (sep2_len > 1 && filtering) {
        unsigned long long total_seps = 0;
        HIP_TRY(hipMemcpy(&total_seps, c->pr_small + 8, 8, hipMemcpyDeviceToHost));
        const unsigned long long extra = total_seps * (unsigned long long)sep_extra;
        if (extra > (unsigned long long)c->tk_stats.n_empty) return fail(BFK_EHIP, "device prepare: folded separators and empty tokens do not add up");
        c->tk_stats.n_empty -= (int64_t)extra;
        if (total_seps) {
            row_seps.resize((size_t)n);
            HIP_TRY(hipMemcpy(row_seps.data(), c->pr_rep, (size_t)n * 4, hipMemcpyDeviceToHost));
        }
    }
    // empty tokens are "invalid" for every grammar whose patterns do not match the empty string (:182-184)
    const int64_t n_empty_inv = (filtering && opts->var_type != BFK_VAR_RAW && opts->var_type != BFK_VAR_NEXTCLADE_AA) ? c->tk_stats.n_empty : 0;
    const int64_t n_other_inv = c->tk_stats.n_invalid;
    if (n_other_inv > (int64_t)PREP_INV_CAP)
        return fail(BFK_EUNSUPPORTED, "device prepare: more than 65536 tokens that match no pattern of the feature type (the host stage lists them)");
    const int64_t n_invalid = n_empty_inv + n_other_inv;
    res->inv_off.clear();
    res->inv_len.clear();
    if (n_other_inv > 0) {
        // The reference prints every invalid token where it meets it: rows in input order, tokens in row order, empty ones as ''.
        // The device noted the non-empty ones in whatever order its waves met them: sorted by byte offset they are in the text's
        // order, and the empty ones (all alike) go between them by count — the empty tokens of the rows in front (k_tok_empties
        // left them per row) plus those of the token's own row in front of it (counted here, on the table's bytes).
        std::vector<uint2> q((size_t)n_other_inv);
        HIP_TRY(hipMemcpy(q.data(), c->pr_inv, q.size() * sizeof(uint2), hipMemcpyDeviceToHost));
        std::sort(q.begin(), q.end(), [](const uint2 &x, const uint2 &y) { return x.x < y.x; });
        std::vector<uint32_t> emp;
        if (n_empty_inv > 0) {
            emp.resize((size_t)n);
            HIP_TRY(hipMemcpy(emp.data(), c->pr_empties, (size_t)n * 4, hipMemcpyDeviceToHost));
            for (size_t r = 0; r < row_seps.size(); r++) emp[r] -= (uint32_t)(sep_extra * row_seps[r]);
        }
        res->inv_off.reserve((size_t)n_invalid);
        res->inv_len.reserve((size_t)n_invalid);
        int64_t emitted = 0, rows_done = 0, before_rows = 0;  // empties listed so far; rows whose empties are inside before_rows
        for (const uint2 &tk : q) {
            const int64_t off = base + (int64_t)tk.x;
            // the row that holds the token: the last one that starts at or before it
            const int64_t r = (std::upper_bound(row_off.begin(), row_off.begin() + n, off) - row_off.begin()) - 1;
            if (r < 0 || off + (int64_t)tk.y > row_off[(size_t)r] + row_len[(size_t)r]) return fail(BFK_EHIP, "device prepare: an invalid token outside its row");
            int64_t want = 0;
            if (n_empty_inv > 0) {
                for (; rows_done < r; rows_done++) before_rows += emp[(size_t)rows_done];
                // empty tokens of row r in front of the token: the row splits at every separator; an empty piece ends at a
                // separator that follows the row's start or another separator
                int64_t in_row = 0;
                const char *rb = bytes + row_off[(size_t)r];
                const int64_t lim = off - row_off[(size_t)r];
                if (sep2_len == 1) {
                    for (int64_t k = 0; k < lim; k++)
                        if ((unsigned char)rb[k] == sp && (k == 0 || (unsigned char)rb[k - 1] == sp)) in_row++;
                } else {  // str.split's matches, from the left: a piece that starts where it ends is empty
                    for (int64_t k = 0, piece = 0; k + sep2_len <= lim;) {
                        if (memcmp(rb + k, sep2, (size_t)sep2_len) == 0) {
                            if (k == piece) in_row++;
                            k += sep2_len;
                            piece = k;
                        } else {
                            k++;
                        }
                    }
                }
                want = before_rows + in_row;
            }
            for (; emitted < want; emitted++) {
                res->inv_off.push_back(off);
                res->inv_len.push_back(0);
            }
            res->inv_off.push_back(off);
            res->inv_len.push_back((int32_t)tk.y);
        }
        for (; emitted < n_empty_inv; emitted++) {
            res->inv_off.push_back(base);
            res->inv_len.push_back(0);
        }
        tm.lap("prepare: invalid tokens in order");
    }
    const int64_t nnz_all = c->nnz;
    if (int rc = dev_realloc(&c->pr_uindices, &c->pr_uidx_cap, nnz_all + 16, 1.05)) return rc;
    HIP_TRY(hipMemsetAsync(c->pr_table, 0xFF, (size_t)slots * sizeof(PrepSlot), c->stream));
    HIP_TRY(hipMemsetAsync(c->pr_small, 0, 64, c->stream));
    PrepArgs pa{};
    pa.n = (int)n;
    pa.by_bytes = filtering ? 0 : 1;
    pa.text = c->tk_text;
    pa.row_off = c->tk_rowoff;
    pa.base = base;
    pa.span_len = c->pr_spanlen;
    pa.indptr = c->own_indptr;
    pa.indices = c->own_indices;
    pa.rowhash = c->pr_rowhash;
    pa.table = c->pr_table;
    pa.mask = (uint32_t)(slots - 1);
    pa.rep = c->pr_rep;
    pa.val = c->pr_val;
    pa.blk = c->pr_blk;
    pa.totals = c->pr_small;
    pa.group = c->pr_group;
    pa.first_row = c->pr_first;
    pa.u_indptr = c->pr_uindptr;
    pa.u_indices = c->pr_uindices;
    pa.fail = c->pr_small + 2;
    if (int e = launch_collapse(pa, c->stream)) return fail(BFK_EHIP, std::string("collapse launch: ") + hipGetErrorString((hipError_t)e));
    int h[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h, c->pr_small, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    tm.lap("prepare: collapse");
    if (h[2] & PREP_FAIL_COLLISION) return fail(BFK_EUNSUPPORTED, "device prepare: two different rows with one hash (the host stage collapses)");
    if (h[2]) return fail(BFK_EHIP, "device prepare: row table overflow");
    res->n_rows = n;
    res->n_unique = h[0];
    res->nnz = h[1];
    res->n_invalid = n_invalid;
    res->n_vocab = c->tk_stats.n_vocab;
    res->filtering = filtering;
    res->base = base;
    res->flt = tp.flt;
    res->sep_byte = sp;
    return BFK_OK;
}

static void prep_info(const PrepResult &r, bfk_prep_info *info) {
    info->n_rows = r.n_rows;
    info->n_unique = r.n_unique;
    info->nnz = r.nnz;
    info->n_invalid = r.n_invalid;
    info->n_vocab = r.n_vocab;
    info->filtered = r.filtering ? 1 : 0;
}

// bfk_table_prepare's contract computed on the device, results installed in the table (group, weight, CSR of the unique rows, the
// invalid tokens in the reference's order): what the parity tests compare with the host stage field by field.  (When every
// invalid token is an empty one the table holds no list: info_out->n_invalid lines of ''.)
extern "C" int bfk_table_prepare_device(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, bfk_prep_info *info_out) {
    if (!info_out) return fail(BFK_EARG, "bfk_table_prepare_device: null argument");
    std::lock_guard<std::mutex> lk(g_mu);
    bfk_ctx *c;
    if (int rc = default_ctx(&c)) return rc;
    if (int rc = ctx_enter(c)) return rc;
    PrepResult r;
    if (int rc = ctx_prepare_table(c, t, sep2, sep2_len, opts, &r)) return rc;
    std::vector<int32_t> group((size_t)r.n_rows), first((size_t)std::max<int64_t>(r.n_unique, 1)), ip((size_t)r.n_unique + 1),
        ix((size_t)std::max<int64_t>(r.nnz, 1));
    HIP_TRY(hipMemcpyAsync(group.data(), c->pr_group, (size_t)r.n_rows * 4, hipMemcpyDeviceToHost, c->stream));
    if (r.n_unique) HIP_TRY(hipMemcpyAsync(first.data(), c->pr_first, (size_t)r.n_unique * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(ip.data(), c->pr_uindptr, (size_t)(r.n_unique + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    if (r.nnz) HIP_TRY(hipMemcpyAsync(ix.data(), c->pr_uindices, (size_t)r.nnz * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    prep_info(r, info_out);
    if (int rc = bfk_table_set_prepared(t, group.data(), first.data(), r.n_unique, info_out, ip.data(), ix.data(), sep2, sep2_len)) return rc;
    return bfk_table_set_invalid(t, r.inv_off.data(), r.inv_len.data(), (int64_t)r.inv_off.size());
}

// The CLI's whole middle in one call: filter + collapse + CSR on the device, the unique rows clustered where they lie (no CSR
// ever visits the host), component sizes against min_cluster_size (:329-339), clusters.tsv by the native writer.
// max_dist must be > 0 (max-dist 0 needs no device).  info_out->nnz == 0: nothing was clustered or written — the reference
// cannot build a matrix from an all-empty input (:214), the caller raises its error.
// n_gpus > 1 (round 5; `--gpus N` of the CLI): filter + collapse + CSR on device 0 as above, then — where several devices pay for
// this input (multi_worth: the rule of bfk_cluster_csr) — the unique rows' CSR comes back to the host ONCE and goes through the
// multi-device driver (cluster_multi: every device holds the CSR, takes its share of the pair work, labels merged on device 0);
// elsewhere the one-device path below, whatever n_gpus says.
// cache_path (round 5; `--output-cache x.bfkc` of the CLI): the same run with every edge recorded, and a side-car cache written from
// it — the two hashes of every unique row's feature string (k_row_hashes: the vocabulary never leaves the device) and the rows'
// neighbour lists (self + both directions, ascending: what bfk_neighbours_csr hands out for all rows) in breakfast_amd/sidecar.py's
// container, marked EXACT (format 2: every list is the neighbourhood of a row of the file, nothing else).
// in_cache (`--input-cache x.bfkc`): a cache run's result — the components of the cached lists, re-indexed, plus the new rows'
// lists (breakfast.py:294-326) — is the no-cache run's result when the cached lists are exact and every cached row is still in
// the input (a row that is gone leaves its list behind, which still chains its neighbours: cache.py:51-71 — then, or with a
// cache of format 1, BFK_EUNSUPPORTED and the caller's list path runs).  So an exact cache of the same max_dist is checked
// (its rows' hashes against this input's, computed on the device) and the run is the no-cache run: the whole clustering is a
// millisecond, less than reading the lists back.  A cache of another max_dist is not used by the reference either (:35-48).
// One device.
static int table_cluster_write_device(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                      int32_t min_cluster_size, int32_t n_gpus, const char *path, const char *in_cache, const char *cache_path,
                                      bfk_prep_info *info_out, int64_t *n_clusters_out);
static int lists_from_edges(int64_t n_rows, const int64_t *select_ind, int64_t nq, const int32_t *edges, int64_t ne, int64_t **nbr_indptr_out,
                            int32_t **nbr_indices_out);

extern "C" int bfk_table_cluster_write_device(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                              int32_t min_cluster_size, const char *path, bfk_prep_info *info_out, int64_t *n_clusters_out) {
    return table_cluster_write_device(t, sep2, sep2_len, opts, max_dist, min_cluster_size, 1, path, nullptr, nullptr, info_out, n_clusters_out);
}

extern "C" int bfk_table_cluster_write_device_cache(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                                    int32_t min_cluster_size, const char *path, const char *in_cache, const char *cache_path,
                                                    bfk_prep_info *info_out, int64_t *n_clusters_out) {
    return table_cluster_write_device(t, sep2, sep2_len, opts, max_dist, min_cluster_size, 1, path, in_cache, cache_path, info_out, n_clusters_out);
}

extern "C" int bfk_table_cluster_write_device_gpus(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                                   int32_t min_cluster_size, int32_t n_gpus, const char *path, bfk_prep_info *info_out,
                                                   int64_t *n_clusters_out) {
    if (n_gpus < 1 || n_gpus > 64) return fail(BFK_EARG, "n_gpus must be 1..64");
    return table_cluster_write_device(t, sep2, sep2_len, opts, max_dist, min_cluster_size, n_gpus, path, nullptr, nullptr, info_out, n_clusters_out);
}

// the side-car container of breakfast_amd/sidecar.py (little endian): MAGIC, int32 max_dist, int64 n_rows, n_lists, total; uint64[n_rows][2]
// hashes; int64[n_lists + 1] offsets; int32[total] members
static int write_sidecar(const char *path, int32_t max_dist, const uint64_t *hashes, int64_t n_rows, const int64_t *off, int64_t n_lists,
                         const int32_t *flat) {
    FILE *f = fopen(path, "wb");
    if (!f) return fail(BFK_EIO, std::string("cannot write ") + path);
    static const char magic[] = "BFKCACHE\x02\n";  // (2: exact lists)
    const int64_t head[3] = {n_rows, n_lists, off[n_lists]};
    bool ok = fwrite(magic, 1, 10, f) == 10 && fwrite(&max_dist, 4, 1, f) == 1 && fwrite(head, 8, 3, f) == 3;
    ok = ok && (n_rows == 0 || fwrite(hashes, 16, (size_t)n_rows, f) == (size_t)n_rows);
    ok = ok && fwrite(off, 8, (size_t)n_lists + 1, f) == (size_t)n_lists + 1;
    ok = ok && (off[n_lists] == 0 || fwrite(flat, 4, (size_t)off[n_lists], f) == (size_t)off[n_lists]);
    if (fclose(f) != 0 || !ok) return fail(BFK_EIO, std::string("short write on ") + path);
    return BFK_OK;
}

// the hashes of an EXACT side-car cache of this max_dist -> BFK_OK (*usable = 0: another max_dist, the cache plays no part);
// BFK_EUNSUPPORTED: format 1, or a file that is not what its header says (the list path reads it and says what is wrong)
// off / flat (may be NULL): the cached lists too — wanted when the run continues the cache into a new one — checked as sidecar.load
// checks them (offsets a non-decreasing run from 0 to the total, members rows of the cached input)
static int read_sidecar_hashes(const char *path, int32_t max_dist, std::vector<uint64_t> *hashes, int *usable, std::vector<int64_t> *off = nullptr,
                               std::vector<int32_t> *flat = nullptr) {
    *usable = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(BFK_EUNSUPPORTED, std::string("cannot read ") + path);
    char magic[10];
    int32_t d = 0;
    int64_t head[3] = {0, 0, 0};
    bool ok = fread(magic, 1, 10, f) == 10 && memcmp(magic, "BFKCACHE\x02\n", 10) == 0 && fread(&d, 4, 1, f) == 1 && fread(head, 8, 3, f) == 3 &&
              head[0] >= 0 && head[1] >= 0 && head[2] >= 0 && head[0] <= INT32_MAX && head[1] <= INT32_MAX;
    if (ok && d == max_dist) {
        hashes->resize((size_t)head[0] * 2);
        ok = head[0] == 0 || fread(hashes->data(), 16, (size_t)head[0], f) == (size_t)head[0];
        if (ok && off && flat) {
            off->resize((size_t)head[1] + 1);
            flat->resize((size_t)head[2]);
            ok = fread(off->data(), 8, off->size(), f) == off->size() && (head[2] == 0 || fread(flat->data(), 4, flat->size(), f) == flat->size());
            ok = ok && (*off)[0] == 0 && off->back() == head[2];
            for (size_t i = 1; ok && i < off->size(); i++) ok = (*off)[i] >= (*off)[i - 1];
            for (size_t i = 0; ok && i < flat->size(); i++) ok = (*flat)[i] >= 0 && (*flat)[i] < head[0];
        }
        *usable = 1;
    }
    fclose(f);
    if (!ok) return fail(BFK_EUNSUPPORTED, "side-car cache: not an exact cache (format 2), or not what its header says (the list path reads it and says what is wrong)");
    return BFK_OK;
}

static int table_cluster_write_device(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                      int32_t min_cluster_size, int32_t n_gpus, const char *path, const char *in_cache, const char *cache_path,
                                      bfk_prep_info *info_out, int64_t *n_clusters_out) {
    if (!info_out || !path || max_dist <= 0 || min_cluster_size < 0) return fail(BFK_EARG, "bfk_table_cluster_write_device: bad argument");
    if ((cache_path || in_cache) && n_gpus != 1) return fail(BFK_EARG, "bfk_table_cluster_write_device: side-car caches go with one device");
    std::vector<uint64_t> cached;
    std::vector<int64_t> c_off, where;  // the cached lists (when the run continues the cache); the cached rows' places in this input
    std::vector<int32_t> c_flat;
    int check_cached = 0;
    if (in_cache)
        if (int rc = read_sidecar_hashes(in_cache, max_dist, &cached, &check_cached, cache_path ? &c_off : nullptr, cache_path ? &c_flat : nullptr))
            return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    bfk_ctx *c;
    DevTimer tm;
    if (int rc = default_ctx(&c)) return rc;
    if (int rc = ctx_enter(c)) return rc;
    tm.lap("pipeline: context");
    PrepResult r;
    if (int rc = ctx_prepare_table(c, t, sep2, sep2_len, opts, &r)) return rc;
    tm.lap("pipeline: prepare (above)");
    prep_info(r, info_out);
    std::vector<int32_t> group((size_t)r.n_rows), first((size_t)std::max<int64_t>(r.n_unique, 1)), labels((size_t)std::max<int64_t>(r.n_unique, 1));
    HIP_TRY(hipMemcpyAsync(group.data(), c->pr_group, (size_t)r.n_rows * 4, hipMemcpyDeviceToHost, c->stream));
    if (r.n_unique) HIP_TRY(hipMemcpyAsync(first.data(), c->pr_first, (size_t)r.n_unique * 4, hipMemcpyDeviceToHost, c->stream));
    if (r.nnz > 0 && n_gpus > 1 && multi_worth(r.n_unique, max_dist, n_gpus)) {
        std::vector<int32_t> ip((size_t)r.n_unique + 1), ix((size_t)r.nnz);
        HIP_TRY(hipMemcpyAsync(ip.data(), c->pr_uindptr, ip.size() * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(ix.data(), c->pr_uindices, ix.size() * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (int rc = cluster_multi(ip.data(), ix.data(), r.n_unique, max_dist, n_gpus, labels.data(), nullptr)) return rc;
    } else if (r.nnz > 0) {
        std::vector<uint64_t> hashes;
        if (cache_path || check_cached) {  // (before the clustering kernels run: the text and the rows' spans are the prepare's)
            int64_t cap = 0;
            uint64_t *d_hash = nullptr;
            if (int rc = dev_realloc(&d_hash, &cap, 2 * r.n_unique)) return rc;
            RowHashArgs ha;
            ha.text = c->tk_text;
            ha.row_off = (const long long *)c->tk_rowoff;
            ha.base = r.base;
            ha.span_len = c->pr_spanlen;
            ha.first_row = c->pr_first;
            ha.n_unique = (int)r.n_unique;
            ha.sep = (uint8_t)r.sep_byte;
            ha.pat = SepPattern{};
            ha.pat.m = (int)sep2_len;
            memcpy(ha.pat.b, sep2, (size_t)sep2_len);
            ha.flt = r.flt;
            ha.out = (unsigned long long *)d_hash;
            hashes.resize((size_t)(2 * r.n_unique));
            int e = launch_row_hashes(ha, c->stream);
            hipError_t e2 = e ? hipSuccess : hipMemcpyAsync(hashes.data(), d_hash, hashes.size() * 8, hipMemcpyDeviceToHost, c->stream);
            if (!e && e2 == hipSuccess) e2 = hipStreamSynchronize(c->stream);
            (void)hipFree(d_hash);
            if (e || e2 != hipSuccess) return fail(BFK_EHIP, std::string("k_row_hashes: ") + hipGetErrorString(e ? (hipError_t)e : e2));
            tm.lap("pipeline: feature hashes");
            if (check_cached && !cached.empty()) {
                where.resize(cached.size() / 2);
                if (int rc = bfk_match_hashes(cached.data(), (int64_t)where.size(), hashes.data(), r.n_unique, where.data())) return rc;
                for (int64_t w : where)
                    if (w < 0) return fail(BFK_EUNSUPPORTED, "side-car cache: a cached row is gone from the input (its list still chains its neighbours: the list path)");
                tm.lap("pipeline: cached rows all present");
            }
        }
        // the unique rows' CSR is clustered where the collapse left it
        if (int rc = bfk_ctx_bind_csr_device(c, c->pr_uindptr, c->pr_uindices, r.n_unique)) return rc;
        if (int rc = ctx_own_labels(c, r.n_unique)) return rc;
        if (!cache_path) {
            if (int rc = bfk_ctx_cluster(c, max_dist, 0, 1, c->own_labels)) return rc;
            if (int rc = bfk_ctx_sync(c, nullptr)) return rc;
        } else {
            int32_t *edges = nullptr;
            int64_t ne = 0;
            int rc = BFK_OK;
            c->edge_sel_on = false;
            for (int attempt = 0; attempt < 8; attempt++) {  // grow the queues until the run fits (as bfk_neighbours_csr does)
                bfk_ctx_set_edge_capture(c, 1);
                rc = bfk_ctx_cluster(c, max_dist, 0, 1, c->own_labels);
                if (!rc) rc = bfk_ctx_sync(c, nullptr);
                if (!rc) rc = bfk_ctx_edges(c, &edges, &ne);
                if (rc != BFK_EOVERFLOW) break;
                if ((rc = ctx_size_cand(c, c->cand_cap_total * 4))) break;
            }
            bfk_ctx_set_edge_capture(c, 0);
            if (rc) return rc;
            tm.lap("pipeline: cluster with every edge recorded");
            int64_t *lp = nullptr;
            int32_t *li = nullptr;
            if (!check_cached) {  // no cache to continue: every row's list (cluster_features without a cache, breakfast.py:314-319)
                rc = lists_from_edges(r.n_unique, nullptr, r.n_unique, edges, ne, &lp, &li);
                free(edges);
                if (!rc) rc = write_sidecar(cache_path, max_dist, hashes.data(), r.n_unique, lp, r.n_unique, li);
            } else {
                // the cache continued, as the reference continues its own (:294-304, cache.py:51-71, sidecar.cluster_with_sidecar):
                // the cached lists re-indexed onto this input (no cached row is gone: nothing drops out) and, behind them, the
                // lists of the rows the cache did not know, in row order — what the NEXT run finds has to be what the list
                // path would have left, or a later run that loses rows chains other neighbours than the reference would
                std::vector<char> known((size_t)r.n_unique, 0);
                for (int64_t w : where) known[(size_t)w] = 1;
                std::vector<int64_t> fresh;
                for (int64_t u = 0; u < r.n_unique; u++)
                    if (!known[(size_t)u]) fresh.push_back(u);
                static const int64_t none = 0;
                rc = lists_from_edges(r.n_unique, fresh.empty() ? &none : fresh.data(), (int64_t)fresh.size(), edges, ne, &lp, &li);
                free(edges);
                if (!rc) {
                    const size_t n_old = c_off.empty() ? 0 : c_off.size() - 1, n_new = fresh.size();
                    std::vector<int64_t> o_off(n_old + n_new + 1, 0);
                    std::vector<int32_t> o_flat(c_flat.size() + (size_t)lp[n_new]);
                    for (size_t i = 0; i < n_old; i++) o_off[i + 1] = c_off[i + 1];
                    for (size_t i = 0; i < c_flat.size(); i++) o_flat[i] = (int32_t)where[(size_t)c_flat[i]];
                    const int64_t base_ = n_old ? c_off[n_old] : 0;
                    for (size_t i = 0; i < n_new; i++) o_off[n_old + i + 1] = base_ + lp[i + 1];
                    if (lp[n_new]) memcpy(o_flat.data() + c_flat.size(), li, sizeof(int32_t) * (size_t)lp[n_new]);
                    rc = write_sidecar(cache_path, max_dist, hashes.data(), r.n_unique, o_off.data(), (int64_t)(n_old + n_new), o_flat.data());
                }
            }
            free(lp);
            free(li);
            if (rc) return rc;
            tm.lap("pipeline: lists + side-car");
        }
        HIP_TRY(hipMemcpyAsync(labels.data(), c->own_labels, (size_t)r.n_unique * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    tm.lap("pipeline: cluster + D2H");
    if (int rc = bfk_table_set_prepared(t, group.data(), first.data(), r.n_unique, info_out, nullptr, nullptr, sep2, sep2_len)) return rc;
    if (int rc = bfk_table_set_invalid(t, r.inv_off.data(), r.inv_len.data(), (int64_t)r.inv_off.size())) return rc;
    if (r.nnz <= 0) return BFK_OK;
    // a component counts the ORIGINAL sequences of its rows (:329-339); labels are the component's smallest row
    const int32_t *weight = bfk_table_weight(t);
    const size_t nu = (size_t)r.n_unique;
    std::vector<int64_t> size(nu, 0);
    for (size_t u = 0; u < nu; u++) {
        if (labels[u] < 0 || (size_t)labels[u] >= nu) return fail(BFK_EHIP, "bfk_table_cluster_write_device: label out of range");
        size[(size_t)labels[u]] += weight[u];
    }
    std::vector<int32_t> cl(nu, 0);
    for (size_t u = 0; u < nu; u++) cl[u] = size[(size_t)labels[u]] >= min_cluster_size ? labels[u] + 1 : 0;
    tm.lap("pipeline: groups + sizes");
    const int rc = bfk_table_write(t, path, cl.data(), n_clusters_out);
    tm.lap("pipeline: writer");
    return rc;
}

extern "C" int bfk_neighbours_csr(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int32_t max_dist,
                                  const int64_t *select_ind, int64_t n_select, int64_t **nbr_indptr_out,
                                  int32_t **nbr_indices_out) {
    if (n_rows < 0 || !indptr || !nbr_indptr_out || !nbr_indices_out || (select_ind == nullptr && n_select > 0))
        return fail(BFK_EARG, "bad arguments");
    const int64_t nq = select_ind ? n_select : n_rows;
    for (int64_t s = 0; s < nq && select_ind; s++)
        if (select_ind[s] < 0 || select_ind[s] >= n_rows) return fail(BFK_EARG, "select_ind out of range");
    std::lock_guard<std::mutex> lk(g_mu);
    bfk_ctx *c;
    if (int rc = default_ctx(&c)) return rc;
    if (int rc = bfk_ctx_upload_csr(c, indptr, indices, n_rows)) return rc;
    if (int rc = ctx_own_labels(c, n_rows)) return rc;
    // select_ind (the rows that are new against a cache, breakfast.py:241-245, :300-304): the kernels record only the edges
    // with a selected end — a bit per row on the device — so what comes back, and what the lists below are built from, is
    // proportional to the selected rows' neighbourhoods, not to the whole graph
    std::vector<uint32_t> sel;
    c->edge_sel_on = false;
    if (select_ind && n_rows > 0) {
        sel.assign((size_t)(n_rows + 31) / 32 + 1, 0u);
        for (int64_t s = 0; s < nq; s++) sel[(size_t)(select_ind[s] >> 5)] |= 1u << (select_ind[s] & 31);
        if (int rc = dev_realloc(&c->d_edge_sel, &c->edge_sel_cap, (int64_t)sel.size())) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_edge_sel, sel.data(), sel.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->edge_sel_on = true;
    }
    void *d_labels = c->own_labels;
    int32_t *edges = nullptr;
    int64_t ne = 0;
    int rc = BFK_OK;
    for (int attempt = 0; attempt < 8; attempt++) {  // grow the queues until the run fits
        bfk_ctx_set_edge_capture(c, 1);
        rc = bfk_ctx_cluster(c, max_dist, 0, 1, d_labels);
        if (!rc) rc = bfk_ctx_sync(c, nullptr);
        if (!rc) rc = bfk_ctx_edges(c, &edges, &ne);
        if (rc != BFK_EOVERFLOW) break;
        if ((rc = ctx_size_cand(c, c->cand_cap_total * 4))) break;
    }
    bfk_ctx_set_edge_capture(c, 0);
    c->edge_sel_on = false;
    if (rc) return rc;
    rc = lists_from_edges(n_rows, select_ind, nq, edges, ne, nbr_indptr_out, nbr_indices_out);
    free(edges);
    return rc;
}

// lists of the query rows (all rows, or the selected ones in the caller's order) from the edges {a, b} of a run: neighbours in
// both directions + self, ascending (get_neighbours_batch's lists, breakfast.py:223-278) -> malloc'ed CSR
static int lists_from_edges(int64_t n_rows, const int64_t *select_ind, int64_t nq, const int32_t *edges, int64_t ne, int64_t **nbr_indptr_out,
                            int32_t **nbr_indices_out) {
    // slot[i] = first query position of row i (a row may be selected more than once: its list is copied)
    std::vector<int32_t> slot((size_t)n_rows, -1);
    std::vector<int64_t> qrow((size_t)nq);
    int64_t n_lists = 0;
    std::vector<int32_t> list_of((size_t)nq);  // query position -> list
    for (int64_t s = 0; s < nq; s++) {
        const int64_t i = select_ind ? select_ind[s] : s;
        if (slot[(size_t)i] < 0) {
            slot[(size_t)i] = (int32_t)n_lists;
            qrow[(size_t)n_lists++] = i;
        }
        list_of[(size_t)s] = slot[(size_t)i];
    }
    std::vector<int64_t> deg((size_t)n_lists + 1, 0);
    for (int64_t e = 0; e < ne; e++) {
        const int32_t a = edges[2 * e], b = edges[2 * e + 1];
        if (slot[(size_t)a] >= 0) deg[(size_t)slot[(size_t)a] + 1]++;
        if (slot[(size_t)b] >= 0) deg[(size_t)slot[(size_t)b] + 1]++;
    }
    for (int64_t l = 0; l < n_lists; l++) deg[(size_t)l + 1] += deg[(size_t)l] + 1;  // +1: self
    std::vector<int32_t> adj((size_t)deg[(size_t)n_lists]);
    std::vector<int64_t> fill(deg.begin(), deg.end() - 1);
    for (int64_t l = 0; l < n_lists; l++) adj[(size_t)fill[(size_t)l]++] = (int32_t)qrow[(size_t)l];
    for (int64_t e = 0; e < ne; e++) {
        const int32_t a = edges[2 * e], b = edges[2 * e + 1];
        if (slot[(size_t)a] >= 0) adj[(size_t)fill[(size_t)slot[(size_t)a]]++] = b;
        if (slot[(size_t)b] >= 0) adj[(size_t)fill[(size_t)slot[(size_t)b]]++] = a;
    }
    for (int64_t l = 0; l < n_lists; l++) std::sort(adj.begin() + deg[(size_t)l], adj.begin() + deg[(size_t)l + 1]);
    int64_t total = 0;
    for (int64_t s = 0; s < nq; s++) total += deg[(size_t)list_of[(size_t)s] + 1] - deg[(size_t)list_of[(size_t)s]];
    int64_t *op = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nq + 1));
    int32_t *oi = (int32_t *)malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(1, total));
    if (!op || !oi) {
        free(op);
        free(oi);
        return fail(BFK_ENOMEM, "out of memory");
    }
    op[0] = 0;
    for (int64_t s = 0; s < nq; s++) {
        const int64_t l = list_of[(size_t)s], len = deg[(size_t)l + 1] - deg[(size_t)l];
        memcpy(oi + op[s], adj.data() + deg[(size_t)l], sizeof(int32_t) * (size_t)len);
        op[s + 1] = op[s] + len;
    }
    *nbr_indptr_out = op;
    *nbr_indices_out = oi;
    return BFK_OK;
}

// ================================================================================================
// cache path: components of a set of neighbour lists (each list is united as a path)
// ================================================================================================
extern "C" int bfk_labels_from_lists(int64_t n_rows, const int64_t *list_indptr, const int32_t *list_indices,
                                     int64_t n_lists, int32_t *labels_out) {
    if (n_rows < 0 || n_lists < 0 || (n_lists > 0 && (!list_indptr || list_indptr[0] != 0)) || (n_rows > 0 && !labels_out))
        return fail(BFK_EARG, "bad arguments");
    if (n_rows > (int64_t)INT32_MAX - 2 * SIG_PAD_ROWS || n_lists > INT32_MAX) return fail(BFK_EARG, "too many rows/lists");
    const int64_t total = n_lists > 0 ? list_indptr[n_lists] : 0;
    if (total < 0 || (total > 0 && !list_indices)) return fail(BFK_EARG, "bad list arrays");
    if (n_rows == 0) return BFK_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    bfk_ctx *c;
    if (int rc = default_ctx(&c)) return rc;
    if (int rc = ctx_enter(c)) return rc;
    void *d_off = nullptr, *d_flat = nullptr, *d_par = nullptr, *d_lab = nullptr, *d_ctr = nullptr;
    int rc = BFK_OK;
    auto cleanup = [&]() {
        for (void *p : {d_off, d_flat, d_par, d_lab, d_ctr})
            if (p) (void)hipFree(p);
    };
    if (hipMalloc(&d_off, (size_t)(n_lists + 1) * 8) != hipSuccess || hipMalloc(&d_flat, (size_t)std::max<int64_t>(total, 1) * 4) != hipSuccess ||
        hipMalloc(&d_par, (size_t)n_rows * 4) != hipSuccess || hipMalloc(&d_lab, (size_t)n_rows * 4) != hipSuccess ||
        hipMalloc(&d_ctr, sizeof(Counters)) != hipSuccess) {
        cleanup();
        return fail(BFK_ENOMEM, "hipMalloc failed");
    }
    hipError_t e = hipMemsetAsync(d_ctr, 0, sizeof(Counters), c->stream);
    if (e == hipSuccess && n_lists > 0) e = hipMemcpyAsync(d_off, list_indptr, (size_t)(n_lists + 1) * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && total > 0) e = hipMemcpyAsync(d_flat, list_indices, (size_t)total * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        if (int le = launch_lists((int *)d_par, (int)n_rows, (const long long *)d_off, (const int *)d_flat, total, (int)n_lists,
                                  (int *)d_lab, (Counters *)d_ctr, c->stream))
            e = (hipError_t)le;
    }
    Counters h{};
    if (e == hipSuccess) e = hipMemcpyAsync(labels_out, d_lab, (size_t)n_rows * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(BFK_EHIP, std::string("labels_from_lists: ") + hipGetErrorString(e));
    else if (h.err & ERR_LABEL) rc = fail(BFK_EARG, "list index out of range");
    cleanup();
    return rc;
}
