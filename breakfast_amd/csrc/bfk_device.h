// bfk_device.h — shared between bfk_kernels.hip (device code + launchers) and bfk_host.cpp (C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfk {

constexpr int KEY_BUCKETS = 16;        // f / g buckets per row length in the (k,f,g) sort key (window around k/2)
constexpr int KEY_MAX_BINS3 = 1 << 24; // cap on (kmax+1)*fb*gb counters (64 MiB); buckets are halved beyond it
constexpr int SIG2_WORDS = 2;          // second-level signature: 64 bits (independent hash)
constexpr int CAND_SHARDS = 512;       // candidate queue shards: returning atomics on one word serialise (~90/us)
constexpr int PF_WAVES_MAX = 4;         // most waves per prefilter block (= waves sharing one tile: 2 or 4, Plan::pf_waves)
constexpr int VERIFY_GRID_MAX = 8192;   // blocks of k_verify (per-block statistics buffer)
constexpr int PF_LDS_QUEUE = 256;      // per-wave LDS coarse hit queue entries (2 KiB per wave)
constexpr int PF_PAIR_LIST = 256;      // per-wave LDS list of exact pairs inside flush_hits (2 KiB per wave)
constexpr int SIG_PAD_ROWS = 1024;     // signature arrays are padded so tile-rounded reads stay in bounds
// hash-table slots per 16-lane group of k_verify: 256 (2 KiB of LDS; 16 tables = 32 KiB per block) for rows of up to 96 tokens
// (the instantiations with up to 6 register steps), 512 for the instantiations that serve longer rows — a pair of 100-token
// rows, what present-day profiles look like, then still fits a group table instead of going to k_verify_long (a block per pair)
__host__ __device__ constexpr int verify_table(int steps) { return steps >= 8 ? 512 : 256; }
// pairs with more tokens (both rows) than 3/4 of the table: certificates here, the exact count in k_verify_long
__host__ __device__ constexpr int verify_max_tokens(int steps) { return verify_table(steps) * 3 / 4; }
constexpr int LONG_TABLE = 4096;        // hash-table slots in LDS per block of k_verify_long (32 KiB)
constexpr int LONG_BLOCKS = 64;         // blocks of k_verify_long when a global scratch table is in use (each owns a slice of it)
constexpr int LONG_BLOCKS_LDS = 1024;   // ... when every pair's table fits the block's LDS (4 blocks of 32 KiB per CU)
constexpr unsigned long long JOIN_EMPTY = ~0ull;  // free slot of the variant-join table {tag : row}
constexpr int JOIN_TPW = 512;           // tokens (entries of `indices`) per wave of k_join
constexpr int JOIN_INLINE_ROW = 128;    // longest row k_join's own exact check takes (two tokens per lane); beyond: k_verify's queue
constexpr int JOIN_MAX_PROBE = 256;     // longest probe chain of the variant join before it gives up (-> all-pairs path)

constexpr long long TOK_FUSE_ROWBITS_BYTES = 64ll << 20;  // texts up to this size: row-start bits set by k_tok_clear, cleared by k_voc_ids
constexpr int PG_MAX_DIST = 7;          // prefix-group path: max_dist + 1 prefix elements per row, at most 8
constexpr int PG_CNT_BITS = 20;         // hashed counters of the sampled token count
constexpr int PG_GIVE_UP = 4096;        // group members behind a row's records, per row (sampled), beyond which the band path is used
constexpr int PGK_ROWS = 1;             // rows per 16-lane group of k_pgkeys (4: all loads of four rows in flight, 82 VGPRs, 5 waves per SIMD — 26 us slower at 1M rows than 1)
constexpr int PG_EST_STRIDE = 1024;     // every so many positions of the sorted records measure their walk (power of two)

enum : int { ERR_ROWLEN = 1, ERR_WORKCAP = 2, ERR_LABEL = 4 };

struct Counters {
    unsigned int ncand[CAND_SHARDS];
    int err;
    int err_rows;  // set by k_sig (row longer than at bind time); cleared by the host only
    unsigned int n_work;   // tiles
    unsigned int ticket;   // arrival order of the k_cells blocks
    unsigned int n_dup;    // variant join: entries of the dup list (k_jhash fills, k_join reads, k_flatten resets)
    int join_fail;         // variant join gave up (probe chain too long): the host re-runs on the all-pairs path
    int pg_fail;           // prefix groups gave up (the groups are too big to pay: the walk did nothing): the host re-runs on the band path
    unsigned long long pg_est;  // members behind every PG_EST_STRIDE-th position of the sorted records in its group (k_pgplace)
    int overflow;
    unsigned long long pairs_in_band;
    unsigned long long pairs_filtered;
    unsigned long long n_cand_total;
    unsigned long long n_edges;
    unsigned long long n_edges_cap;
    unsigned long long n_connected;  // k_verify_connected: candidates dropped because their rows were in one tree already
    unsigned long long dbg[8];  // phase stamps of k_plan (s_memrealtime, 100 MHz), printed with BFK_DEBUG=1
};

// variant join (max_dist == 1): table / bitmap of this step, the set to clear for the next step, row hashes
struct JoinArgs {
    unsigned long long *tab, *tab_next;  // mask + 1 slots
    uint32_t *bits, *bits_next;          // bmask + 1 bits in 32-bit words (blocked Bloom filter, two bits per row)
    uint2 *rowhash;
    int *batch_row;  // per JOIN_TPW tokens: the row that holds the first of them
    int2 *dups;  // pairs of rows with one H, found while inserting
    int *stats;  // per block of k_join: {edges certified and hooked there, candidates}
    uint32_t mask, bmask;
    int dup_cap;
    int *dyn_host;     // with dyn: 16 ints of pinned host memory — block 0 of k_jhash copies dyn[0..15] there before anything else (the
                       // host reads the counters of the step when it completes the bind; a copy of their own is ~9 us of stream time, a
                       // store at the END of a kernel ~5: at the start of one it is under the kernel's own work)
    const int *dyn;    // NULL, or the device words of a bind the host has not completed yet (the row statistics k_tok_rows leaves + the tokeniser's
                       // counters: [0] longest row, [1] negative row length seen, [3] nnz, [10] tokeniser failure flags): the kernels
                       // take nnz and the longest row from there, and do nothing when the CSR is unusable or outside what the host
                       // assumed (a row over JOIN_INLINE_ROW tokens, no token at all: Counters::join_fail = 2, the host redoes the step)
    int inline_exact;  // no row of the bound CSR is longer than JOIN_INLINE_ROW: k_join decides every match itself, no k_verify launch
    int dbg;  // BFK_JOIN_DEBUG (timing experiments, results invalid): 1 no settle, 2 no table probe, 4 no queueing of
              // bitmap hits, 8 no table insert, 16 no clearing, 32 no unions, 128 no scattered bitmap loads, 256 no bitmap atomicOr
};

// Everything one enqueue of the pipeline needs (device pointers live in the ctx workspace).
struct Plan {
    int n, nnz, kcap, d, w1;
    int rows_per_lane, fb, gb, hb;
    int shard, n_shards;
    int verify_phases, verify_phase2_union;  // > 1: two-phase verify (first 1/phases of every shard, compress, the rest)
    int skip_connected;  // labels-only step: k_verify_connected drops candidates whose rows are connected already
    int verify_grid, wave_table_d;  // k_verify: per-wave hash table up to this max_dist, per-group tables beyond
    int verify_adaptive, verify_grid2;  // max_dist 2, labels only: k_verify AND k_verify_connected (grid2) are launched, the queue's
    long long verify_density_thr;       // fill against this many candidates decides on the device which of them works
    int tile_cap, tile_hint, pf_blocks, pf_waves, cand_cap_shard, edge_cap, dbg;
    unsigned gslots;  // slots per block of the global scratch table of k_verify_long (0 = none)
    const int *indptr;
    const uint32_t *indices;
    int *hist3, *start3, *start3c, *rowkey, *rowrank, *tile_slots, *blk_stats;
    int hist_copies;  // 8 or 1: copies of the cell histogram (k_sig block b counts into copy b % copies)
    unsigned long long *chain;
    int *parent;
    int4 *srec;  // per sorted position: {row, length, second-level signature lo, hi}
    uint32_t *gkey;
    int *gcnt;
    uint32_t *sig1, *sigu1, *sigu2;
    int4 *tiles;
    int4 *cand;
    int2 *candk;
    int2 *edges;  // NULL unless edge capture is on
    const uint32_t *edge_sel;  // edge capture: a bit per row, only edges with a selected end are recorded (NULL: every edge)
    int *labels;
    Counters *ctr;
    unsigned long long *dbg_t;
    // prefix-group path (max_dist >= 2, large inputs): the rows are grouped by the elements of their PREFIX (the
    // max_dist + 1 rarest tokens) — rows within max_dist share one — and the pair kernel scans groups instead of (k,f,g)
    // bands.  The sorted order has pg_recs * n positions (pg_recs = max_dist + 2 records per row).
    int pg, pg_recs;
    uint32_t *pg_cnt;                         // sampled token counts (2^PG_CNT_BITS hashed counters)
    uint32_t *pg_keys, *pg_keys_s;            // [n][recs] record keys, row-major; sorted
    int pg_tb;                                // key bits of a token: bits of (largest token id + 2)
    int pg_dense;                             // the token counters are indexed by the token id (largest id < 2^PG_CNT_BITS)
    int pg_walk16;                            // the groups are walked by k_pgwalk16 (16 lanes per row); 0: k_pgjoin, a wave per row
    int pg_has_short;                         // 0: no row has <= 2 * PG_MAX_DIST tokens and the records are position-major: the SHORT slot is left out of the sorted order
    int pg_pb;                                // position bits of the composite key k_pgplace bisects on (3; 0 = positional filter off)
    uint32_t *pg_keys_pm;                     // [recs][n] record keys position-major (the sort's input)
    int *pg_rows, *pg_rows_s;                 // the records' (row * recs + slot); sorted along
    void *pg_temp;
    size_t pg_temp_bytes;
    int4 *pg_srec;       // {row, length, second-level signature} in group order
    int4 *pg_rowinfo;    // [n]: {length, second-level signature, first token}
    int2 *pg_recpos;     // [n][recs]: {position of the record in the group order, members of its group behind it}
    int join_skip_verify;  // k_join decides every match itself (JoinArgs::inline_exact: no row over JOIN_INLINE_ROW tokens): no k_verify launch
    int join;  // 1: candidates come from the variant join (k_jhash + k_join) instead of k_sig .. k_prefilter
    JoinArgs ja;
};

// ---- device tokeniser (bfk_text.hip): profile text -> first-appearance vocabulary -> CSR --------------------------
constexpr int TOK_WIN = 1024;                          // bytes per wave step: 16 per lane
constexpr int TOK_WPW = 4;                             // windows per wave
constexpr int TOK_BLOCK_BYTES = 4 * TOK_WPW * TOK_WIN; // 16 KiB of text per block of 4 waves (k_tok_hash)
constexpr int TOK_SCAN_WINS = 64;                      // windows per block of k_tok_scan / k_voc_count (16 waves): 64 KiB of text
constexpr int TOK_PAD_BYTES = TOK_SCAN_WINS * TOK_WIN; // the text is padded to a multiple of this
constexpr int TOK_TEXT_SLACK = 64;                     // separator bytes behind T_pad (unaligned 8-byte reads of a token's tail)
constexpr uint32_t TOK_MAX_LEN = 65534;                // longest token the table word can describe (16-bit length)
constexpr unsigned TOK_HOLD_UNITS = 17;                // hash units (4 KiB) at the end of a text piece that wait for the next piece: > TOK_MAX_LEN + 64 bytes
constexpr int TOK_MAX_PROBE = 512;                     // probe chain at which the table counts as too full
constexpr unsigned long long TOK_EMPTY = ~0ull;        // free slot of the vocabulary table
constexpr int TOK_LIST_CAP = 1024;                     // token starts a wave lists at a time (a 4 KiB unit with more is taken window by window)
constexpr int TOK_LOOKUP_U = 10;                       // table lookups a lane keeps in flight (x 64 lanes: a typical unit of ~540 tokens is ONE round)
// slot of the vocabulary table (bfk_text.hip): INLINE key {len8 (1..7) : 56 bits of token bytes} or HASHED key
// {1 : tag15 : len16 : offset32}; `first` = smallest byte offset of an inline token; `id` = its first-appearance id
struct alignas(16) TokSlot {
    unsigned long long key;
    uint32_t first;
    int id;
};
enum : int { TOK_FAIL_ROWOFF = 1, TOK_FAIL_LONG = 2, TOK_FAIL_TABLE = 4 };

struct TokCounters {
    unsigned int nnz, n_vocab;
    int fail;  // TOK_FAIL_*
    unsigned int n_invalid;    // filter mode: non-empty token occurrences that match no pattern of the feature type
    unsigned int nnz_kept;     // filter mode: tokens that survive the filter (the CSR's entries)
    unsigned int n_empty;      // filter mode: empty tokens inside the rows' spans (k_tok_empties)
    int pad_[2];               // (debug counters)
};

// filter_features (breakfast.py:116-190) on the device: every token occurrence is judged where it is hashed
struct TokFilter {
    int on;          // 0: every token is kept (sparse_feature_matrix alone)
    int var_type;    // BFK_VAR_* of include/bfk.h (0 covsonar_dna, 1 covsonar_aa, 2 nextclade_dna, 3 nextclade_aa, 4 raw)
    int skip_ins, skip_del;
    long long trim_start, upper;  // a DNA substitution is dropped when pos <= trim_start or pos >= upper (= reference_length - trim_end)
};
constexpr uint32_t TOK_NONE = 0xFFFFFFFFu;  // slot of a token that the filter dropped

struct TokArgs {
    const uint8_t *text;       // T bytes + separator padding up to T_pad + TOK_TEXT_SLACK
    const long long *row_off;  // [n_rows + 1], as the caller passed them (base = row_off[0] is subtracted)
    long long base;
    uint32_t T, T_pad;         // text bytes; rounded up to TOK_BLOCK_BYTES with at least one byte of padding
    int n_rows;
    uint8_t sep;
    int strict;                // 1: also check row_off[0] == base and row_off[n_rows] == base + T (offsets from a caller's device memory)
    uint32_t *rowbits, *startbits, *boundbits;  // one bit per byte position: row start / token start / separator or row start
    int any_ids;               // 1: the CSR's column ids are the tokens' table SLOTS (an injective renaming of the first-appearance
                               // ids; what clustering needs) — no first-occurrence bits, no k_voc_count / k_voc_ids / k_tok_ids
    uint32_t *firstbits;       // ... / a token's first occurrence starts here
    uint32_t *keptbits;        // filter mode: ... / a token that survives the filter starts here (the CSR counts these instead of startbits)
    uint32_t *keptwin, *keptblk; // filter mode: kept tokens in front of every window inside its scan block / in front of every scan block
    const int *span_len;       // filter mode with rows that do not abut (bfk_table): length of row r's span (what lies behind it up to the
                               // next row's start has been blanked with separator bytes); NULL: a row ends where the next one starts
    TokFilter flt;
    uint2 *inv_queue;          // filter mode: {byte offset, length} of the first inv_cap token occurrences that match no pattern (any order)
    uint32_t inv_cap;
    uint32_t *row_empties;     // filter mode: empty tokens of every row (k_tok_empties); NULL: not wanted
    uint32_t *winbase, *vocwin; // [T_pad / TOK_WIN]: tokens / vocabulary entries in front of every window inside its scan block
    uint32_t *blkbase, *vocblk; // [T_pad / TOK_PAD_BYTES + 1]: ... in front of every scan block (k_scan_single)
    TokSlot *table;            // tmask + 1 slots of 16 bytes
    uint32_t tmask;
    int *tabid;                // first-appearance id per slot (4 bytes per slot: what k_tok_ids gathers from)
    uint32_t *tokslot;         // per token: its slot (aliases `indices`)
    uint32_t *indices;
    int *indptr;
    TokCounters *tc;
    long long nnz_cap;         // upper bound of the token count the buffers are sized for
    int rows_clear_after;      // k_voc_ids clears the row-start bits for the next build (texts up to TOK_FUSE_ROWBITS_BYTES)
    int rows_fused;            // the row-start bits were set by k_tok_clear (no k_tok_rowbits launch)
    int fine_head;             // bit 0: the head launch of the hash takes a wave per 1 KiB window instead of per 4 KiB unit; bit 1: the sample launch too
    int head_units;            // > 0: k_tok_hash runs the first so many units in a launch of their own before the rest
    int sample;                // > 1: then every sample-th unit, then the others (three launches in all)
    int dbg;                   // BFK_TOK_DEBUG (timing experiments, results invalid): 1 no atomicMin of a found token's first offset,
                               // 2 no table loads, 4 no lookups at all (staging + list only), 8 no head block
};
int launch_tokenize(const TokArgs &a, hipStream_t st, hipEvent_t *ev, int n_pieces, const unsigned *piece_blk,
                    hipEvent_t *piece_ev);  // bfk_text.hip

// ---- collapse of duplicate rows (bfk_prep.hip) ----------------------------------------------------------------------
constexpr unsigned long long PREP_EMPTY = ~0ull;
constexpr int PREP_MAX_PROBE = 4096;
enum : int { PREP_FAIL_TABLE = 1, PREP_FAIL_COLLISION = 2 };
struct alignas(16) PrepSlot {
    unsigned long long key;  // row hash
    uint32_t row;            // smallest row with it (0xFFFFFFFF: none yet)
    uint32_t pad_;
};
struct PrepArgs {
    int n;                       // input rows
    int by_bytes;                // identity of a row: 1 its raw bytes (nothing is filtered), 0 its sequence of CSR entries
    const uint8_t *text;         // by_bytes: the rows' bytes ...
    const long long *row_off;    // ... row r starts at row_off[r] - base
    long long base;
    const int *span_len;         // ... and is span_len[r] long (NULL: up to the next row's start)
    const int *indptr;           // CSR of ALL input rows (the tokeniser's)
    const uint32_t *indices;
    unsigned long long *rowhash; // [n]
    PrepSlot *table;             // mask + 1 slots
    uint32_t mask;
    int *rep;                    // [n] representative (smallest row with the same identity)
    int2 *val, *blk;             // [n] / [n / 1024 + 1]: {is representative, its entries} -> exclusive prefix sums
    int *totals;                 // [2]: unique rows, their CSR entries
    int *group;                  // [n]  out: unique index of every input row
    int *first_row;              // [n]  out: input row of every unique row
    int *u_indptr;               // [n + 1] out: CSR of the unique rows
    uint32_t *u_indices;
    int *fail;                   // PREP_FAIL_*
};
// side-car cache: the two hashes of every unique row's feature string (k_row_hashes, bfk_text.hip)
constexpr int SEP_MAX_BYTES = 16;  // longest token separator the device prepare folds (k_sepfold)
struct SepPattern {
    uint8_t b[SEP_MAX_BYTES];
    int m;
};
struct RowHashArgs {
    const uint8_t *text;       // the prepare's text (blanked between the features), byte 0 = table byte `base`
    const long long *row_off;  // [n_rows] absolute offsets of the rows' features
    long long base;
    const int *span_len;       // [n_rows]
    const int *first_row;      // [n_unique]
    int n_unique;
    uint8_t sep;               // the separator byte of the text (the stand-in, when the table's separator has several bytes)
    SepPattern pat;            // the table's separator as the feature strings hold it: pat.m bytes (1: sep itself)
    TokFilter flt;
    unsigned long long *out;   // [2 n_unique]
};
int launch_row_hashes(const RowHashArgs &a, hipStream_t st);
int launch_blank(uint8_t *text, const long long *row_off, const int *span_len, int n, long long base, uint32_t T, uint8_t sep, hipStream_t st);
int launch_sepfold(uint8_t *text, const long long *row_off, const int *span_len, int n, long long base, const SepPattern &pat,
                   uint8_t standin, int *seps, unsigned long long *total, hipStream_t st);
int launch_collapse(const PrepArgs &a, hipStream_t st);

int sort_records(void *temp, size_t *temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const int *rows_in,
                 int *rows_out, size_t n, int bits, hipStream_t st, int comp_recs = 0, int comp_pb = 0, int gen_n = 0, int gen_recs = 0);  // bfk_sort.hip
int launch_maxlen(const int *indptr, int n, int *out, hipStream_t st);
// rows of a build for k_tok_clear (rowbits == NULL: the bits are set by a k_tok_rowbits launch of their own)
struct TokRows {
    const long long *row_off;
    long long base;
    uint32_t T;
    int n_rows;
    uint32_t *rowbits;
};
int launch_tok_clear(void *zero, size_t zero_bytes, void *ones, size_t ones_bytes, uint8_t *pad, uint32_t pad_bytes, uint8_t pad_byte, int *small,
                     hipStream_t st, const TokRows &rows);  // bfk_text.hip
int launch_maxtok(const uint32_t *indices, int nnz, int *out, hipStream_t st);
int launch_pipeline(const Plan &pl, hipStream_t st, hipEvent_t *ev);
int launch_pairs(const Plan &pl, int t_begin, int t_end, hipStream_t st, hipEvent_t *ev);
int launch_flatten(const Plan &pl, hipStream_t st, hipEvent_t *ev);
int launch_lists(int *parent, int n, const long long *off, const int *flat, long long total, int n_lists, int *labels,
                 Counters *ctr, hipStream_t st);
int launch_merge(int *parent, int n, const int *gathered, int n_parts, int *labels, int *changed, Counters *ctr,
                 int skip, int splice, hipStream_t st);

}  // namespace bfk
