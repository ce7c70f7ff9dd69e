// bfk_device.h — shared between bfk_kernels.hip (device code + launchers) and bfk_host.cpp (C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfk {

constexpr int KEY_BUCKETS = 16;        // f / g buckets per row length in the (k,f,g) sort key (window around k/2)
constexpr int KEY_MAX_BINS3 = 1 << 24; // cap on (kmax+1)*fb*gb counters (64 MiB); buckets are halved beyond it
constexpr int SIG2_WORDS = 2;          // second-level signature: 64 bits (independent hash)
constexpr int CAND_SHARDS = 512;       // candidate queue shards: returning atomics on one word serialise (~90/us)
constexpr int PF_WAVES = 2;            // waves per prefilter block = waves sharing one tile
constexpr int PF_LDS_QUEUE = 256;      // per-wave LDS coarse hit queue entries (2 KiB per wave)
constexpr int PF_PAIR_LIST = 256;      // per-wave LDS list of exact pairs inside flush_hits (2 KiB per wave)
constexpr int VERIFY_LDS_ROW = 128;    // tokens of row B staged per 16-lane group in k_verify (512 B/group)
constexpr int SIG_PAD_ROWS = 1024;     // signature arrays are padded so tile-rounded reads stay in bounds
constexpr int LONG_LDS_CAP = 15360;    // tokens of a long row staged in LDS by k_canon_long (60 KiB)

enum : int { ERR_ROWLEN = 1, ERR_WORKCAP = 2, ERR_LABEL = 4 };

struct Counters {
    unsigned int ncand[CAND_SHARDS];
    int err;
    int err_rows;  // set by k_canon (row longer than at bind time); cleared by the host only
    unsigned int n_work;   // tiles
    unsigned int ticket;   // arrival order of the k_cells blocks
    unsigned int n_long;
    int overflow;
    unsigned long long pairs_in_band;
    unsigned long long pairs_filtered;
    unsigned long long n_cand_total;
    unsigned long long n_edges;
    unsigned long long n_edges_cap;
    unsigned long long dbg[8];  // phase stamps of k_plan (s_memrealtime, 100 MHz), printed with BFK_DEBUG=1
};

// Everything one enqueue of the pipeline needs (device pointers live in the ctx workspace).
struct Plan {
    int n, kcap, d, w1;
    int rows_per_lane, fb, gb;
    int shard, n_shards;
    int verify_grid, union_grid;
    int tile_cap, cand_cap_shard, edge_cap, long_lds_cap, dbg;
    const int *indptr;
    const uint32_t *indices;
    uint32_t *cols;
    int *hist3, *start3, *rowkey, *rowrank, *tile_slots;
    unsigned long long *chain;
    int *perm, *ksorted, *parent, *longrows;
    uint32_t *sig1, *sig2, *sigu1, *sigu2;
    int4 *tiles;
    int4 *cand;
    int2 *candk;
    int2 *edges;  // NULL unless edge capture is on
    int *labels;
    Counters *ctr;
    unsigned long long *dbg_t;
};

int launch_maxlen(const int *indptr, int n, int *out, hipStream_t st);
int launch_pipeline(const Plan &pl, hipStream_t st, hipEvent_t *ev);
int launch_pairs(const Plan &pl, int t_begin, int t_end, hipStream_t st, hipEvent_t *ev);
int launch_flatten(const Plan &pl, hipStream_t st, hipEvent_t *ev);
int launch_lists(int *parent, int n, const long long *off, const int *flat, long long total, int n_lists, int *labels,
                 Counters *ctr, hipStream_t st);
int launch_merge(int *parent, int n, const int *gathered, int n_parts, int *labels, int *changed, Counters *ctr,
                 hipStream_t st);

}  // namespace bfk
