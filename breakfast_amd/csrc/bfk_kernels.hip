// bfk_kernels.hip — HIP kernels for gfx950 (MI355X, CDNA4): the breakfast clustering hot path.
//
// Pipeline (one stream, no host round trip and no memset between kernels):
//   k_sig        one wave per row: sort key (k, f, g[, h]) and two XOR-parity signatures of the UNSORTED row
//                (a token repeated an even number of times cancels itself, so the bound holds for multisets
//                without sorting or ranking repeats); parent[i] = i; cell histogram (in copies) and the rank of
//                every row inside its cell by one returning atomic per row
//   k_cells      sum of the histogram copies, scan with decoupled look-back -> start3 / start3c, tile list
//                (tiles never cross cells); re-zeroes histogram and counters for the next step
//   k_place      counting-sort scatter into sort-key order: one 16-byte record {row, length, second-level
//                signature} and the first-level signature per row
//   k_prefilter  the pair kernel: one block per tile; the columns that can be within d of the tile's rows are
//                a few contiguous ranges (the (k,f,g) band); popcount(sig_row ^ sig_col) <= d is a necessary
//                condition for |A delta B| <= d; survivors pass a 64-bit second level and are queued
//   k_verify     one 16-lane group per candidate: matching certificate (prefix + shifted suffix) or, when it
//                fails, signed counting of both rows' tokens in an LDS hash table -> exact multiset distance;
//                lock-free union-find hooks of the verified edges (one edge per lane);
//                k_verify_long for pairs > 192 tokens
//   k_flatten    labels[i] = root(i) = smallest row index of the component
//   max_dist == 1 (up to 2M rows while join_pays says so — short rows always, rows of ~105 tokens up to ~60k: join_wanted in bfk_host.cpp) replaces
//                k_sig .. k_verify by the VARIANT JOIN: k_jhash (additive multiset hash
//                of every row -> hash table + bitmap), k_join (one lookup per token occurrence: H(B) - h(t); matches
//                certified by a 64-lane compare and hooked at once; what the compare cannot decide: counted exactly by
//                the wave while no row has more than 128 tokens — no k_verify launch then —, to k_verify's queue
//                otherwise) — O(nnz), no pairs formed
//   max_dist 2 .. 7 on inputs beyond PG_MIN_ROWS (bfk_host.cpp) replaces k_sig .. k_prefilter by the PREFIX GROUPS:
//                k_pgfreq (sampled token counts), k_pgkeys (the row's d + 1 rarest distinct tokens -> records), the radix
//                sort of bfk_sort.hip, k_pgplace (members of a group as one stream; where each record's walk ends:
//                positional filter), k_pgwalk16 (16 lanes per row walk the members behind the row's records; a 64-bit
//                signature level; queue), then k_verify (max_dist 2, exact steps) or k_verify_connected (labels-only
//                steps: candidates of rows that are connected already are dropped unchecked)
//   k_merge      multi-GPU: unite (own root, label in another rank's part) pseudo-edges;  k_union_lists: cache path
//
// What it replaces in the reference: the band loop + get_neighbours_batch + sklearn _sparse_manhattan +
// _reduce_func + networkx components (src/breakfast/breakfast.py:223-276, 287-326).
#include "bfk_device.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

namespace bfk {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// loads of the union-find forest: system scope (sc0 sc1), i.e. past this XCD's L2.  With agent scope a node
// hooked by another XCD kept looking like a root here, which cost a wasted hook attempt per such node (measured:
// -7% on the verify kernel at d = 2, 3; nothing at d = 1, where every node is hooked once)
__device__ __forceinline__ int ld_agent(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t ld_agent_u(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// path-compression store: a plain store (no sc bits, acked by L2) — it sits in the dependent chain of the
// next load (gfx9 vmcnt counts stores), so a write-through store would add its memory latency per hop
__device__ __forceinline__ void st_lazy(int *p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// Lock-free union-find, hooks always go from the larger index to a smaller one, so a root is the
// smallest index of its tree and parent[x] <= x.  A stale load is an older, still valid ancestor or "x is a
// root"; the deciding step is the memory-side atomic of the hook (see uf_union).

// find, continuing from the already loaded cur = parent[x]
__device__ __forceinline__ int uf_find_from(int *parent, int x, int cur) {
    if (cur != x) {
        int prev = x, next;
        while (cur > (next = ld_agent(parent + cur))) {
            st_lazy(parent + prev, next);  // path halving; only ever writes an ancestor to a non-root
            prev = cur;
            cur = next;
        }
    }
    return cur;
}

__device__ __forceinline__ int uf_find(int *parent, int x) {
    int cur = ld_agent(parent + x);
    if (cur != x) {
        int prev = x, next;
        while (cur > (next = ld_agent(parent + cur))) {
            st_lazy(parent + prev, next);  // path halving; only ever writes an ancestor to a non-root
            prev = cur;
            cur = next;
        }
    }
    return cur;
}

__device__ __forceinline__ bool uf_union(int *parent, int a, int b) {
    // both first hops in flight together; equal parents = same tree already (in a dense graph most edges are
    // redundant and, once the trees are compressed, end here without walking to the root)
    const int pa = ld_agent(parent + a), pb = ld_agent(parent + b);
    if (pa == pb) return false;
    int ra = uf_find_from(parent, a, pa), rb = uf_find_from(parent, b, pb);
    while (ra != rb) {
        if (ra < rb) {
            int t = ra;
            ra = rb;
            rb = t;
        }
        // Hook by atomicMin (ra > rb): it never fails.  If ra was a root (old == ra) it now hangs under rb.  If ra
        // had been hooked under `old` meanwhile, parent[ra] becomes min(old, rb) — either way a node of the set
        // that ra, old and rb end up in, because what is left to do is to unite old with rb, and this thread
        // goes on to do exactly that.  (Every value ever stored in parent[x] is <= x and belongs to x's final
        // component, also under concurrent path halving, so chains still end and labels stay canonical.)
        const int old = atomicMin(parent + ra, rb);
        if (old == ra) return true;
        ra = uf_find(parent, old);
    }
    return false;
}

// Union by splicing (Rem's algorithm), one returning atomic per step and no loads: parent[hi] = min(parent[hi], lo)
// keeps every invariant above (the value stored is <= hi and of hi's final component); if hi was a root, or already
// pointed at lo, the edge is in; otherwise hi's former parent `old` and lo still have to be united — both are below
// hi, so the walk ends.  Where nearly every edge meets two fresh or shallow nodes (max_dist <= 2) this is one
// memory round trip per edge instead of two loads, the finds and the hook; in dense graphs (max_dist >= 3), where
// most edges are redundant, uf_union's two loads end them sooner (see make_pair_args).
__device__ __forceinline__ void uf_link(int *parent, int a, int b) {
    while (a != b) {
        const int hi = max(a, b), lo = min(a, b);
        const int old = atomicMin(parent + hi, lo);
        if (old == hi || old == lo) break;
        a = old;
        b = lo;
    }
}

// after a compression pass (k_compress) nearly every node points at its root: equal first hops = same tree already,
// one round trip of loads for a redundant edge; otherwise splice from the parents
__device__ __forceinline__ void uf_link_checked(int *parent, int a, int b) {
    const int pa = ld_agent(parent + a), pb = ld_agent(parent + b);
    if (pa != pb) uf_link(parent, pa, pb);
}

// slot hash of the verify tables
__device__ __forceinline__ uint32_t hash3(uint32_t x) {
    uint32_t h = x * 0xC2B2AE35u;
    return h ^ (h >> 15);
}

// ------------------------------------------------------------------------------------------------
// k_maxlen: validate indptr, find the longest row (bind time only)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_maxlen(const int *__restrict__ indptr, int n, int *out /*[0]=max k, [1]=err, [2]=rows of <= 2*PG_MAX_DIST tokens, [3]=indptr[n], [4]=indptr[0]*/) {
    __shared__ int s_k[4], s_short[4];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[3] = indptr[n];
        out[4] = indptr[0];
    }
    int kmax = 0, n_short = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {  // (grid-stride: one set of atomics per block)
        int k = indptr[i + 1] - indptr[i];
        if (k < 0) {
            atomicOr(out + 1, 1);
            k = 0;
        }
        n_short += k <= 2 * PG_MAX_DIST ? 1 : 0;
        kmax = max(kmax, k);
    }
    for (int s = 32; s > 0; s >>= 1) {
        kmax = max(kmax, __shfl_xor(kmax, s));
        n_short += __shfl_xor(n_short, s);
    }
    if ((threadIdx.x & 63) == 0) {
        s_k[threadIdx.x >> 6] = kmax;
        s_short[threadIdx.x >> 6] = n_short;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        kmax = max(max(s_k[0], s_k[1]), max(s_k[2], s_k[3]));
        n_short = s_short[0] + s_short[1] + s_short[2] + s_short[3];
        if (kmax > 0) atomicMax(out, kmax);
        if (n_short > 0) atomicAdd(out + 2, n_short);
    }
}

// ------------------------------------------------------------------------------------------------
// Sort key.  Rows are ordered by (k, f, g): k = multiset size, f and g = number of tokens whose first /
// second hash bit is set.  All three are 1-Lipschitz in the distance: a pair with |A\B| = a, |B\A| = b,
// a + b <= d, k_B - k_A = delta >= 0 has a <= (d - delta)/2, b = a + delta, and f_B - f_A, g_B - g_A in
// [-a, b].  So for a run of consecutive sorted rows the rows that can be within d form, per column length
// k' and per column f', ONE contiguous range of the (k, f, g) order — the "band" the prefilter scans.
// (The reference bands by length only: np.isclose(n, q, atol=d), breakfast.py:250.)  f, g ~ Binomial(k, 1/2)
// (sigma ~ 3 at k = 40), so each of the two keys cuts the pair slots by ~2.5x at d = 1.
// f and g are stored as buckets: v - (k/2 - nb/2) clamped to [0, nb) (a window centred on the mean).
// For small d — while the (d+1)^3 candidate ranges of a tile still fit the 64 lanes — a third statistic h of
// the same kind is the least significant part of the key (k, f, g, h): another ~3x fewer pair slots (measured
// on the 100k / 400k benchmark inputs: 7.9e7 -> 2.5e7, 9.9e8 -> 2.5e8 tile-padded slots).
// ------------------------------------------------------------------------------------------------

struct KeyCfg {
    int fb, gb, hb;  // buckets per row length for f, g and h (powers of two; 1 = key unused)
    int fb_log, gb_log, hb_log;
};
__device__ __forceinline__ int key_center(int k, int nb) { return (k >> 1) - (nb >> 1); }
__device__ __forceinline__ int key_bucket(int v, int k, int nb) { return min(max(v - key_center(k, nb), 0), nb - 1); }
__device__ __forceinline__ int key_of(const KeyCfg &c, int k, int f, int g, int h) {
    return ((k * c.fb + key_bucket(f, k, c.fb)) * c.gb + key_bucket(g, k, c.gb)) * c.hb + key_bucket(h, k, c.hb);
}

// bucket range [lo, hi] a column of length kp can have in one key, given the rows' bucket range [rlo, rhi]
// at length k = kp - delta (delta >= 0): raw difference in [-amax, bmax], the window moves by s = c(kp) - c(k),
// and a clamped end bucket stands for an open interval.
__device__ __forceinline__ void key_band(int rlo, int rhi, int k, int kp, int amax, int bmax, int nb, int *lo, int *hi) {
    const int s = key_center(kp, nb) - key_center(k, nb);
    *lo = (rlo <= 0) ? 0 : max(rlo - amax - s, 0);
    *hi = (rhi >= nb - 1) ? nb - 1 : min(rhi + bmax - s, nb - 1);
}

struct CellArgs {
    int *hist3;     // copies x cells cell counters (copy-major), re-zeroed here
    int *start3;    // cells + 1: first sorted position of every (k,f,g) cell
    int *start3c;   // copies x cells: first sorted position of the rows counted in each histogram copy
    int4 *tiles;    // {first sorted row, rows, cell key, first tile of the cell}: tiles never cross a cell boundary
    unsigned long long *chain;  // one status word per block of k_cells (zeroed again by k_place)
    Counters *ctr;
    int n, cells, tr_shift, tile_cap;
    int shard0, nshards;  // multi-GPU: only the tiles of this rank's cells are listed (owner of a cell = hash of its key mod ranks)
};

// inclusive prefix sum over the 64 lanes (DPP row prefix, then row broadcasts)
__device__ __forceinline__ int wave_incl_scan_add(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true);  // row_bcast15 into rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true);  // row_bcast31 into rows 2,3
    return x;
}

// ------------------------------------------------------------------------------------------------
// k_cells<C>: exclusive scan of the (k,f,g) cell counters -> start3, and the tile list: every non-empty cell is
// cut into tiles of at most TR = 2^tr_shift rows (a tile never crosses a cell boundary, so its band is as
// tight as the sort key allows).  One cell per thread, 1024 cells per block.
// The histogram comes in C copies (k_sig block b counts into copy b % C: the returning atomics of the hub
// cells are what k_sig waits for, and they serialise per address); a cell's count is the sum of its copies,
// and the rows counted in copy q start at start3c[q][cell] = start3[cell] + the copies before q.
// Cross-block offsets by decoupled look-back: a block publishes the (tiles, rows) total of its own cells as
// soon as its local scan is done (status word: flag 1 + value, agent-scope release), looks back over its
// predecessors 64 at a time, adding aggregates until it meets a block that already published its INCLUSIVE
// prefix (flag 2), and then publishes its own inclusive prefix.  Grids larger than the guaranteed
// co-residency take their logical index from an arrival ticket (a block then only ever waits for blocks that
// have started).  The histogram and the per-step counters are re-zeroed here, the status words by k_place.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(1024) void k_cells(CellArgs a) {
    __shared__ int s_wr[16], s_wt[16];  // per-wave totals: rows, tiles
    __shared__ int s_base_r, s_base_t, s_bid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    if (gridDim.x > 256) {  // beyond 256 blocks co-residency is not certain: order by arrival
        if (threadIdx.x == 0) s_bid = (int)atomicAdd(&a.ctr->ticket, 1u);
        __syncthreads();
        bid = s_bid;
    }
#define PLAN_STAMP(i) if (threadIdx.x == 0) a.ctr->dbg[i] = wall_clock64();
    if (bid == 0) PLAN_STAMP(0)
    if (bid == 0) {
        if (threadIdx.x < CAND_SHARDS) a.ctr->ncand[threadIdx.x] = 0;
        if (threadIdx.x == 64) {
            a.ctr->err = 0;
            a.ctr->overflow = 0;
            a.ctr->n_edges = a.ctr->n_cand_total = a.ctr->n_edges_cap = a.ctr->n_connected = 0;
        }
    }
    const int c = bid * 1024 + threadIdx.x;
    int part[C];
    int rows = 0;
#pragma unroll
    for (int q = 0; q < C; q++) {
        part[q] = 0;
        if (c < a.cells) {
            int *h = a.hist3 + (size_t)q * a.cells + c;
            part[q] = *h;
            *h = 0;
        }
        rows += part[q];
    }
    const int trm = (1 << a.tr_shift) - 1;
    // Multi-GPU: a CELL belongs to one rank, never a tile (the order of the rows inside a cell comes from atomics and differs
    // from rank to rank; whole cells and whole column cells are the same sets everywhere), and a rank lists ONLY its own
    // cells' tiles: the pair kernel's grid is its share of the work.  (Listing every tile and letting the pair kernel skip
    // the foreign ones left 0.146 of 0.231 ms on every rank of an 8-way split at 1M rows: 7/8 of the blocks only to exit.)
    const bool owned = a.nshards <= 1 || (int)((((uint32_t)c * 0x9E3779B1u) >> 12) % (uint32_t)a.nshards) == a.shard0;
    const int tl = owned ? (rows + trm) >> a.tr_shift : 0;
    const int inc_r = wave_incl_scan_add(rows), inc_t = wave_incl_scan_add(tl);
    if (lane == 63) {
        s_wr[wave] = inc_r;
        s_wt[wave] = inc_t;
    }
    __syncthreads();
    if (wave == 0) {
        // totals of the 16 waves -> exclusive wave offsets; lane 15 holds the block totals
        const int wr = lane < 16 ? s_wr[lane] : 0, wt = lane < 16 ? s_wt[lane] : 0;
        const int ir = wave_incl_scan_add(wr), it = wave_incl_scan_add(wt);
        if (lane < 16) {
            s_wr[lane] = ir - wr;
            s_wt[lane] = it - wt;
        }
        const int tot_r = __builtin_amdgcn_readlane(ir, 15), tot_t = __builtin_amdgcn_readlane(it, 15);
        const unsigned long long FLAG_AGG = 1ull << 62, FLAG_INC = 2ull << 62, VAL = (1ull << 62) - 1ull;
        auto pack = [](int t, int r) { return (((unsigned long long)(unsigned)t) << 32) | (unsigned)r; };
        if (lane == 0 && bid > 0)
            __hip_atomic_store(&a.chain[bid], FLAG_AGG | pack(tot_t, tot_r), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int before_r = 0, before_t = 0;
        for (int w0 = bid - 1; w0 >= 0; w0 -= 64) {
            const int p = w0 - lane;
            unsigned long long v = 0ull;
            if (p >= 0) {
                int spins = 0;
                while (((v = __hip_atomic_load(&a.chain[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 62) == 0ull) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) {  // never expected: bound the spin
                        atomicOr(&a.ctr->err, ERR_WORKCAP);
                        v = FLAG_INC;
                        break;
                    }
                }
            }
            // nearest predecessor (lowest lane) that already knows its inclusive prefix: nothing beyond it counts
            const unsigned long long inc_mask = __builtin_amdgcn_ballot_w64(p >= 0 && (v >> 62) == 2ull);
            const int stop = inc_mask ? (int)__builtin_ctzll(inc_mask) : 64;
            const unsigned long long val = (p >= 0 && lane <= stop) ? (v & VAL) : 0ull;
            before_r += __builtin_amdgcn_readlane(wave_incl_scan_add((int)(unsigned)(val & 0xffffffffull)), 63);
            before_t += __builtin_amdgcn_readlane(wave_incl_scan_add((int)(val >> 32)), 63);
            if (inc_mask) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane == 0) {
            __hip_atomic_store(&a.chain[bid], FLAG_INC | pack(before_t + tot_t, before_r + tot_r), __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_AGENT);
            s_base_r = before_r;
            s_base_t = before_t;
            if (bid + 1 == (int)gridDim.x) {
                a.ctr->ticket = 0;  // every block has taken its ticket: clean for the next step
                a.start3[a.cells] = a.n;
                const unsigned nt = (unsigned)(before_t + tot_t);
                if (nt > (unsigned)a.tile_cap) atomicOr(&a.ctr->err, ERR_WORKCAP);
                a.ctr->n_work = min(nt, (unsigned)a.tile_cap);
            }
        }
    }
    __syncthreads();
    const int pos = s_base_r + s_wr[wave] + inc_r - rows;
    int t = s_base_t + s_wt[wave] + inc_t - tl;
    if (c < a.cells) {
        a.start3[c] = pos;
        if (C > 1) {
            int run = pos;
#pragma unroll
            for (int q = 0; q < C; q++) {
                a.start3c[(size_t)q * a.cells + c] = run;
                run += part[q];
            }
        }
        const int tr = 1 << a.tr_shift;
        const int t_first = t;  // every tile carries the index of its cell's first tile: the multi-GPU owner key
        for (int r0 = 0; owned && r0 < rows; r0 += tr, t++)
            if (t < a.tile_cap) a.tiles[t] = make_int4(pos + r0, min(tr, rows - r0), c, t_first);
    }
    if (bid + 1 == (int)gridDim.x) PLAN_STAMP(1)
}

// ------------------------------------------------------------------------------------------------
// k_sig: one wave per row, 64 tokens per step: the row's sort key (k, f, g) and its two XOR-parity
// signatures.  Nothing here needs the row sorted: a token that occurs an even number of times cancels itself
// in the signature, which keeps popcount(sig(A) ^ sig(B)) <= #{t : a_t + b_t odd} <= sum_t |a_t - b_t| = L1
// valid for multiset rows (the reference's count matrix) — so the rows are never sorted at all; the exact
// check (k_verify) works on the unsorted rows through a hash table.  parent[i] = i.
// ------------------------------------------------------------------------------------------------
// XOR of x over the 64 lanes, valid in lane 63 (DPP prefix within rows, then row broadcasts; a DPP op costs
// ~4 cycles per wave-instruction on gfx950, ds_bpermute ~26: tools/ubench/valu_rate.hip)
__device__ __forceinline__ uint32_t wave_xor_to_lane63(uint32_t x) {
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);  // row_shr:1
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);  // row_shr:8 -> lane 15 of each row
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true);  // row_bcast15 into rows 1,3
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true);  // row_bcast31 into rows 2,3
    return x;
}

// token hashes of k_sig: two 24x24-bit multiplies (v_mul_u32_u24 issues at full rate; v_mul_lo_u32 is a
// quarter-rate instruction).  Only the low 24 bits of a token id enter: ids that differ above bit 23 merely
// share signature bits, which weakens the filter and never the result.
__device__ __forceinline__ uint32_t sig_h1(uint32_t x) { return __umul24(x, 0x9E3779u); }
__device__ __forceinline__ uint32_t sig_h2(uint32_t h1) { return __umul24(h1 >> 8, 0x85EBCBu); }

// One block = 16 waves = up to 256 consecutive rows (rpw rows per wave).  Phase 1: one wave per row, 64 tokens
// per step; the extents of all the wave's rows come in one load and the first 64 tokens of ALL its rows are
// requested before the first row is reduced.  Phase 2, per wave (no block barrier): lane t finishes row t of the
// wave — key, the row-order stores, and one returning atomic on the block's copy of the cell histogram, which
// is the cell count and the row's rank inside its cell at once.  (Aggregating a block's rows per cell in an LDS
// hash table first saved 4% of the atomics — 208 consecutive input rows rarely share a cell — for three block
// barriers; the per-address serialisation of hub cells, ~90 returning atomics per us, is handled by the
// histogram copies instead.)
template <int W1>
__global__ __launch_bounds__(1024) void k_sig(const int *__restrict__ indptr, const uint32_t *__restrict__ indices, int n,
                                               int nnz, int kcap, int rpw, KeyCfg key, int *__restrict__ rowkey,
                                               int *__restrict__ parent, uint32_t *__restrict__ sigu1,
                                               uint32_t *__restrict__ sigu2, int *hist3, int *__restrict__ rowrank,
                                               Counters *ctr, int dbg, int cells, int copies) {
    constexpr int LOG1 = 5 + (W1 == 1 ? 0 : (W1 == 2 ? 1 : 2));  // bits of the first-level signature index
    constexpr int MAXRPW = 16, MAXR = 16 * MAXRPW;
    __shared__ int s_k[MAXR], s_f[MAXR], s_g[MAXR], s_h[MAXR];
    __shared__ uint32_t s_s1[MAXR * W1], s_s2[MAXR * SIG2_WORDS];
    static_assert(SIG2_WORDS == 2, "second-level signature = 64 bits");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rows_per_block = rpw * 16;
    const int base = blockIdx.x * rows_per_block;
    const int r0 = base + wave * rpw;
    const int nr = max(0, min(rpw, n - r0));
    if (nr > 0 && nnz > 0 && !(dbg & 32)) {
        // extents of all rows of this wave: lane l holds indptr[r0 + l]
        const int ext = indptr[min(r0 + min(lane, rpw), n)];
        // The first 64 tokens of ALL the wave's rows are requested up front (one register per row): with one row
        // fetched ahead, every row still waited for most of a ~2 us round trip (70% of the wave cycles were waits).
        // Unconditional loads from clamped addresses: a lane past its row reads a neighbour's token and
        // contributes nothing.
        uint32_t xs[MAXRPW];
#pragma unroll
        for (int t = 0; t < MAXRPW; t++) {
            const int bt = __builtin_amdgcn_readlane(ext, t);
            xs[t] = indices[min(max(bt, 0) + lane, nnz - 1)];
        }
#pragma unroll
        for (int t = 0; t < MAXRPW; t++) {
            if (t >= nr) continue;  // wave-uniform
            const int b = __builtin_amdgcn_readlane(ext, t), e = __builtin_amdgcn_readlane(ext, t + 1);
            const uint32_t xfirst = xs[t];
            int k = e - b;
            if (k < 0 || k > kcap) {
                if (lane == 0) atomicOr(&ctr->err_rows, ERR_ROWLEN);
                k = k < 0 ? 0 : kcap;
            }
            uint32_t s1[W1], s2all = 0, s2hi = 0;
#pragma unroll
            for (int w = 0; w < W1; w++) s1[w] = 0;
            int f = 0, g = 0, hc = 0;
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int rem = k - j0;  // wave-uniform
                const uint32_t x = j0 == 0 ? xfirst : ((lane < rem) ? indices[b + j0 + lane] : 0u);
                const unsigned long long vm = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
                const uint32_t one = lane < rem ? 1u : 0u;  // lanes past the row contribute no bit
                const uint32_t h1 = sig_h1(x), h2 = sig_h2(h1);
                f += __popcll(__builtin_amdgcn_ballot_w64((int)h2 < 0) & vm);
                g += __popcll(__builtin_amdgcn_ballot_w64((h2 & 0x40000000u) != 0u) & vm);
                hc += __popcll(__builtin_amdgcn_ballot_w64((h1 & 0x00100000u) != 0u) & vm);
                const uint32_t b1 = h1 >> (32 - LOG1);
                const uint32_t bit1 = one << (b1 & 31);
                if (W1 == 1) {
                    s1[0] ^= bit1;
                } else {
#pragma unroll
                    for (int w = 0; w < W1; w++) s1[w] ^= ((int)(b1 >> 5) == w) ? bit1 : 0u;
                }
                const uint32_t bit2 = one << ((h2 >> 24) & 31);               // index bits 24..28
                const uint32_t hi = (uint32_t)((int)(h2 << 2) >> 31) & bit2;  // bit 29 selects the word
                s2all ^= bit2;
                s2hi ^= hi;
            }
#pragma unroll
            for (int w = 0; w < W1; w++) s1[w] = wave_xor_to_lane63(s1[w]);
            s2all = wave_xor_to_lane63(s2all);
            s2hi = wave_xor_to_lane63(s2hi);
            const int lr = wave * rpw + t;
            if (lane == 63) {
#pragma unroll
                for (int w = 0; w < W1; w++) s_s1[lr * W1 + w] = s1[w];
                s_s2[lr * 2] = s2all ^ s2hi;
                s_s2[lr * 2 + 1] = s2hi;
            }
            if (lane == 0) {
                s_k[lr] = k;
                s_f[lr] = f;
                s_g[lr] = g;
                s_h[lr] = hc;
            }
        }
    } else if (nr > 0 && lane == 0) {  // no tokens at all: every row is empty
        for (int t = 0; t < nr; t++) {
            const int lr = wave * rpw + t;
            s_k[lr] = s_f[lr] = s_g[lr] = s_h[lr] = 0;
            for (int w = 0; w < W1; w++) s_s1[lr * W1 + w] = 0;
            s_s2[lr * 2] = s_s2[lr * 2 + 1] = 0;
        }
    }
    // phase 2, per wave (no block barrier): lane t finishes row t of this wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < nr) {
        const int tr = wave * rpw + lane;
        const int i = r0 + lane;
        const int key3 = key_of(key, s_k[tr], s_f[tr], s_g[tr], s_h[tr]);
        rowkey[i] = key3;
        parent[i] = i;
#pragma unroll
        for (int w = 0; w < W1; w++) sigu1[(size_t)i * W1 + w] = s_s1[tr * W1 + w];
        *reinterpret_cast<uint2 *>(sigu2 + (size_t)i * 2) = make_uint2(s_s2[tr * 2], s_s2[tr * 2 + 1]);
        // rank inside the cell: one returning atomic per row on the block's histogram copy
        int rk = 0;
        if (!(dbg & 16)) rk = atomicAdd(&hist3[(size_t)(blockIdx.x & (copies - 1)) * cells + key3], 1);
        rowrank[i] = rk;
    }
}

// k_place: counting-sort scatter.  Row i goes to sorted position start3[key] + rank; its length and its two
// signatures move with it (the prefilter reads signatures in sorted order, coalesced).
template <int W1>
__global__ __launch_bounds__(256) void k_place(const int *__restrict__ indptr, int n, int kcap,
                                                const int *__restrict__ start3, const int *__restrict__ rowkey,
                                                const int *__restrict__ rowrank, const uint32_t *__restrict__ sigu1,
                                                const uint32_t *__restrict__ sigu2, int4 *__restrict__ srec,
                                                uint32_t *__restrict__ sig1, unsigned long long *chain, int n_chain,
                                                int cells, int copies, int sig_rows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_chain) chain[i] = 0ull;  // k_cells' status words: clean for the next step
    if (i >= n) return;
    // start3 is start3c when copies > 1: the row was ranked inside histogram copy (its k_sig block) % copies
    const int p = start3[(size_t)((i / sig_rows) & (copies - 1)) * cells + rowkey[i]] + rowrank[i];
    const int k = indptr[i + 1] - indptr[i];
    // one 16-byte record per sorted position {row, length, 64-bit second-level signature}: a scattered store
    // costs a whole 64-byte sector whatever its width, so the three fields travel together
    static_assert(SIG2_WORDS == 2, "the record holds a 64-bit second-level signature");
    const uint2 s2 = *reinterpret_cast<const uint2 *>(sigu2 + (size_t)i * 2);
    srec[p] = make_int4(i, k < 0 ? 0 : (k > kcap ? kcap : k), (int)s2.x, (int)s2.y);
#pragma unroll
    for (int w = 0; w < W1; w++) sig1[(size_t)p * W1 + w] = sigu1[(size_t)i * W1 + w];
}

// ------------------------------------------------------------------------------------------------
// candidate queue: (row a, row b) pairs, 8 shards (block % 8) so no single counter word takes all atomics
// ------------------------------------------------------------------------------------------------
struct PairArgs {
    const int *indptr;
    const uint32_t *rows;  // the raw CSR indices (unsorted rows, repeats kept)
    const int4 *srec;  // per sorted position: {row, length, second-level signature lo, hi}
    int *parent;
    int4 *cand;   // {row a, row b, cols offset a, cols offset b}
    int2 *candk;  // {k_a, k_b}
    int cand_cap_shard;
    int d;
    int n, nnz;
    int union_batch;  // edges a 16-lane group of k_verify collects before it hooks them (power of two <= 16)
    int dbg;  // BFK_PF_DEBUG experiments: 1 = no flush, 2 = no rescan (results wrong; timing only)
    int use_link;  // k_verify hooks its edges with 0 uf_union (find + hook), 1 uf_link (splicing), 2 first hops, then splicing
    int part_lo, part_hi, part_den;  // k_verify works on entries [cnt * lo / den, cnt * hi / den) of every queue shard
    int stats_off;                   // ... and leaves its per-block counts at blk_stats + stats_off
    int skip_connected;              // labels-only step: candidates whose rows are in one tree already are dropped unchecked
    int density_mode;                // 0: the kernel runs.  max_dist 2, labels only: BOTH verify kernels are launched and the queue's fill
    long long density_thr;           // decides which of them works — 1: only when it holds at most density_thr candidates (k_verify: a
                                     // sparse forest, nearly every candidate joins two trees), 2: only when it holds more
                                     // (k_verify_connected: hubs with thousands of neighbours make the graph dense, most candidates are
                                     // between rows that are connected already)
    int max_tokens;                  // tokens (both rows) whose exact count fits a group table of the k_verify instantiation that runs
    const uint32_t *sel;             // edge capture: a bit per row — only edges with a selected end are recorded (NULL: all)
    Counters *ctr;
};

// flush `cnt` queued (p,q) hits: filter, translate to row ids, append to the shard's global queue with
// one global atomic per wave; a full global queue raises the overflow flag (host re-runs in slices).
// Every load that depends only on (p,q) is issued up front, so a flush costs ~3 dependent round trips.
__device__ __forceinline__ void flush_pairs(const PairArgs &a, const int2 *sbuf, int cnt, int shard) {
    const int lane = threadIdx.x & 63;
    for (int i0 = 0; i0 < cnt; i0 += 64) {  // called by ONE wave on its own queue
        const int i = i0 + lane;
        bool pass = false;
        int ra = 0, rb = 0;
        if (i < cnt) {
            const int2 pq = sbuf[i];
            const int p = pq.x, q = pq.y;
            if (q > p && q < a.n && p < a.n) {
                const int4 rp = a.srec[p], rq = a.srec[q];
                ra = rp.x;
                rb = rq.x;
                const int c = __popc((uint32_t)(rp.z ^ rq.z)) + __popc((uint32_t)(rp.w ^ rq.w));
                pass = (rq.y - rp.y <= a.d) && (c <= a.d);
            }
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
        if (mask == 0ull) continue;
        shard = (shard + 1) & (CAND_SHARDS - 1);  // a hub row's thousands of candidates spread over the shards
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(&a.ctr->ncand[shard], (unsigned)__popcll(mask));
        int ba = 0, ea = 0, bb = 0, eb = 0;
        if (pass) {  // row extents: in flight together with the queue-slot atomic
            ba = a.indptr[ra];
            ea = a.indptr[ra + 1];
            bb = a.indptr[rb];
            eb = a.indptr[rb + 1];
        }
        base = __shfl(base, 0);
        if (pass) {
            const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (idx < a.cand_cap_shard) {
                const size_t o = (size_t)shard * a.cand_cap_shard + idx;
                a.cand[o] = make_int4(ra, rb, ba, bb);
                a.candk[o] = make_int2(ea - ba, eb - bb);
            } else {
                a.ctr->overflow = 1;  // dropped: the host re-runs the unit range in smaller slices
            }
        }
    }
}

template <int W>
__device__ __forceinline__ uint32_t sigdist(const uint32_t (&a)[W], const uint32_t *b) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < W; w++) c += __popc(a[w] ^ b[w]);
    return c;
}

// Flush of the prefilter's COARSE hit queue.  An entry says "column q is within d (first-level signature) of
// at least one of the SB rows starting at sorted row pb".  Finding which rows is done here, one entry per
// lane (64 entries per instruction), instead of in the scan loop where a hit costs the whole wave a serial
// rescan: the 16 row dwords and the column signature are re-read from L2, the exact (row, column) pairs are
// compacted into `pairs` (LDS, PF_PAIR_LIST entries per round) with a wave prefix sum, and flush_pairs does the rest.
template <int W>
__device__ __forceinline__ void flush_hits(const PairArgs &a, const uint32_t *__restrict__ sig1, const int2 *coarse, int cnt,
                                           int row_end, int2 *pairs, int shard) {
    constexpr int SB = 16 / W;
    const int lane = threadIdx.x & 63;
    const uint32_t d = (uint32_t)a.d;
    for (int i0 = 0; i0 < cnt; i0 += 64) {
        const int i = i0 + lane;
        uint32_t mask = 0;
        int pb = 0, q = 0;
        if (i < cnt) {
            const int2 e = coarse[i];
            pb = e.x;
            q = e.y;
            uint32_t cq[W];
#pragma unroll
            for (int x = 0; x < W; x++) cq[x] = sig1[(size_t)q * W + x];
            uint32_t rw[16];
#pragma unroll
            for (int x = 0; x < 16; x++) rw[x] = sig1[(size_t)pb * W + x];  // padded array: in bounds past n
#pragma unroll
            for (int j = 0; j < SB; j++)
                if (pb + j < row_end && sigdist<W>(cq, &rw[j * W]) <= d) mask |= 1u << j;
        }
        const int mine = __popc(mask);
        const int incl = wave_incl_scan_add(mine);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        const int first = incl - mine;
        for (int base = 0; base < total; base += PF_PAIR_LIST) {  // usually one round: ~1 pair per entry
            int o = first;
            uint32_t mm = mask;
            while (mm) {
                const int j = __ffs((int)mm) - 1;
                mm &= mm - 1;
                if (o >= base && o < base + PF_PAIR_LIST) pairs[o - base] = make_int2(pb + j, q);
                o++;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            flush_pairs(a, pairs, min(PF_PAIR_LIST, total - base), shard);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

struct BandArgs {
    unsigned long long *dbg_t;  // BFK_PF_DEBUG & 4: per-wave stamps
    const int *start3;
    const int4 *tiles;
    int *tile_slots;  // per tile: pair slots evaluated (statistics, summed by the host)
    KeyCfg key;
    int kcap, d, inv_d1, inv_d2;
    int chunk_aligned;  // k_prefilter: 1 = column chunks on a grid aligned to 64 (BFK_PF_ALIGNED=1: the earlier form)
};

// ------------------------------------------------------------------------------------------------
// k_prefilter<W, R, PW, DBG>: the all-pairs kernel.  One BLOCK of PW waves per tile.  A tile is up to 64*R sorted
// rows of ONE (k,f,g) cell.  The columns that can be within d of those rows are, per column length
// k' = k + delta and per column f' bucket, one contiguous range of the sorted order (the g band): every
// lane turns one (delta, f') candidate into a column range with two start3 look-ups (all in flight at once).
// Orientation: the COLUMNS sit in the lanes (one signature per lane, 64 consecutive columns per chunk, the next
// chunk in flight while this one is compared) and the tile's ROWS are broadcast — cells are small (~25 rows at 100k rows), a
// column range is several cells wide, so this keeps the lanes full where rows-in-lanes would leave 60% idle.
// The tile's row signatures are parked once in the wave's LDS slice and re-read as wave-uniform
// ds_read_b128 broadcasts (v_xor with VGPR operands runs at full rate; an SGPR operand halves it on gfx950).
//   per pair slot: W x (v_xor + v_bcnt) + v_min; one v_cmp + ballot per sub-batch of 16 row-dwords;
//   a hit queues the COARSE fact (sub-batch, column) in the wave's LDS queue (slot from the ballot prefix, fill
//   level in a wave-uniform register); flush_hits works out the rows lane-parallel and applies the second level.
// The PW waves of a block take the tile's chunks round-robin and never synchronise.  PW = 4 (one wave per SIMD
// of the CU) for small inputs, where every block is resident at once and the kernel time is set by the SIMD
// that happened to draw the heaviest waves (measured with BFK_PF_DEBUG=4 / tools/pf_timeline.py at 100k rows:
// 54 chunks on the busiest SIMD against a mean of 30 with PW = 2); PW = 2 for large inputs, which are
// throughput-bound and only pay for the extra per-wave set-up.  Tried and dropped for the same imbalance:
// ticket counters for dynamic work distribution (returning atomics on one word serialise at ~90/us; spread
// over 32 counters they still sit in the in-order vmcnt queue in front of the next item's loads), capping
// residency so the hardware dispatcher balances (no gain: the heavy tiles are all dispatched in the first
// round), s_setprio for waves with long scans (no effect), issuing the tiles of big cells several times with
// the columns divided (helps at 100k, costs 40% at 1M rows).
// ------------------------------------------------------------------------------------------------
template <int W, int R, int PW, bool DBG>
__global__ __launch_bounds__(PW * 64) void k_prefilter(const uint32_t *__restrict__ sig1, BandArgs ba, int n,
                                                              int shard0, int nshards, int t_begin, int t_end,
                                                              PairArgs pa) {
    constexpr int CC = 64;              // columns per chunk: one per lane
    constexpr int SB = 16 / W;          // rows per sub-batch (16 dwords): the hit-detection granularity
    constexpr int QCAP = PF_LDS_QUEUE;  // per-wave coarse hit queue entries
    constexpr int TROWS = 64 * R;       // most rows a tile can have
    __shared__ int2 sbuf[PW][QCAP];
    __shared__ int2 spairs[PW][PF_PAIR_LIST];  // flush_hits: exact pairs of a batch of 64 coarse entries
    __shared__ __attribute__((aligned(16))) uint32_t srow[PW][TROWS * W];
    const int lane = threadIdx.x & 63;
    // readfirstlane: tell the compiler the wave index is wave-uniform, so that everything derived from it
    // (tile descriptor, candidate ranges) stays in SGPRs / scalar loads
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t d = (uint32_t)pa.d;
    const int n_tiles = min((int)pa.ctr->n_work, t_end);
    const int qshard = (blockIdx.x * PW + wave) & (CAND_SHARDS - 1);
    int2 *myq = sbuf[wave];
    uint32_t *myrow = srow[wave];
    // Work items = (tile, wave slot) pairs of this shard, taken by WAVES, grid-stride: the host launches one
    // block per expected tile (the count lives on the device; any grid is correct).
    const int n_items = max(n_tiles - t_begin, 0) * PW;
    int item = (int)blockIdx.x * PW + wave;
    while (item < n_items) {
    const int t = t_begin + item / PW;
    const int wslot = item & (PW - 1);
    const unsigned long long t_start = DBG ? wall_clock64() : 0ull;
    unsigned long long t_rng = 0, t_main = 0;
    int dbg_hits = 0, dbg_chunks = 0;
    int qn = 0;  // fill level of this wave's hit queue (wave-uniform)
    const int4 tile = ba.tiles[t];  // (multi-GPU: k_cells listed this rank's cells only)
    const int row0 = tile.x, nrows = tile.y;
    const int fb = ba.key.fb, gb = ba.key.gb, hb = ba.key.hb;
    // fb, gb, hb are powers of two
    const int h0 = tile.z & (hb - 1), g0 = (tile.z >> ba.key.hb_log) & (gb - 1);
    const int f0 = (tile.z >> (ba.key.hb_log + ba.key.gb_log)) & (fb - 1);
    const int k0 = tile.z >> (ba.key.hb_log + ba.key.gb_log + ba.key.fb_log);
    // the tile's row signatures -> LDS (rows past the tile are padding: they are compared but never pushed)
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int p = min(row0 + r * 64 + lane, n - 1);
#pragma unroll
        for (int x = 0; x < W; x++) myrow[(r * 64 + lane) * W + x] = sig1[(size_t)p * W + x];
    }
    const int nsb = (nrows + SB - 1) / SB;
    // Candidate column ranges, one per lane (all look-ups in flight together).  The finest split that fits the
    // 64 lanes is used: mode 3 = one range per (delta, f', g') with the h band contiguous inside (needs the
    // third key and (d+1)^3 <= 64), mode 2 = one per (delta, f') covering the g band and every h,
    // mode 1 = one per column length.
    const int D = (int)d, D1 = D + 1;
    const int mode = (hb > 1 && D1 * D1 * D1 <= 64) ? 3 : (D1 * D1 <= 64 ? 2 : 1);
    const int ncand = mode == 3 ? D1 * D1 * D1 : (mode == 2 ? D1 * D1 : D1);
    for (int cbase = 0; cbase < ncand; cbase += 64) {  // more than 64 candidates only when d >= 64
        const int c = cbase + lane;
        int my_cb = 0, my_ce = 0;
        if (c < ncand) {
            // c -> (delta, fi, gi) without divisions: 16-bit reciprocals of D1 and D1^2 (c < 64)
            int delta = c, fi = 0, gi = 0;
            if (mode == 3) {
                delta = (c * ba.inv_d2) >> 16;
                const int r = c - delta * D1 * D1;
                fi = (r * ba.inv_d1) >> 16;
                gi = r - fi * D1;
            } else if (mode == 2) {
                delta = (c * ba.inv_d1) >> 16;
                fi = c - delta * D1;
            }
            const int kp = k0 + delta;
            if (kp <= ba.kcap) {
                const int amax = (D - delta) >> 1, bmax = amax + delta;
                int fa, fz, ga, gz, ha, hz;
                key_band(f0, f0, k0, kp, amax, bmax, fb, &fa, &fz);
                key_band(g0, g0, k0, kp, amax, bmax, gb, &ga, &gz);
                key_band(h0, h0, k0, kp, amax, bmax, hb, &ha, &hz);
                const int fp = fa + fi, gp = ga + gi;
                if (mode == 3) {
                    if (fp <= fz && gp <= gz) {
                        my_cb = ba.start3[((kp * fb + fp) * gb + gp) * hb + ha];
                        my_ce = ba.start3[((kp * fb + fp) * gb + gp) * hb + hz + 1];
                    }
                } else if (mode == 2) {
                    if (fp <= fz) {
                        my_cb = ba.start3[((kp * fb + fp) * gb + ga) * hb];
                        my_ce = ba.start3[((kp * fb + fp) * gb + gz + 1) * hb];
                    }
                } else {
                    my_cb = ba.start3[(kp * fb + fa) * gb * hb];
                    my_ce = ba.start3[(kp * fb + fz + 1) * gb * hb];
                }
                my_cb = max(my_cb, row0);  // q > p >= row0
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (DBG) t_rng = wall_clock64() + (unsigned long long)(my_cb & 0);  // after the range look-ups landed
        long long slots = 0;
        const int nc = min(64, ncand - cbase);
        // Chunk iterator over all ranges of this round: this wave takes the chunks whose running number is
        // == wslot (mod PW).  Everything here is wave-uniform (SGPRs).
        int it_ci = -1, it_q0 = 0, it_ctrue = 0, it_cend = 0, chunk_no = 0;
        auto advance = [&]() {
            it_q0 += PW * CC;
            while (it_ci < nc && it_q0 >= it_cend) {
                it_ci++;
                if (it_ci >= nc) break;
                it_ctrue = __builtin_amdgcn_readlane(my_cb, it_ci);
                it_cend = __builtin_amdgcn_readlane(my_ce, it_ci);
                if (it_cend <= it_ctrue) {
                    it_cend = it_q0 = 0;  // empty range: keep looking
                    continue;
                }
                // chunks start AT the range: ranges are short (a cell or two, ~25 columns at 100k rows) and a chunk grid aligned
                // to 64 cut four in ten of them in two; the load is 64 consecutive dwords either way (round 5: k_prefilter
                // 27.4 -> 26.1 us at 100k rows, 210 -> 197 at 1M, max-dist 1)
                const int cal = ba.chunk_aligned ? (it_ctrue & ~(CC - 1)) : it_ctrue;
                const int nch = (it_cend - cal + CC - 1) / CC;
                it_q0 = cal + ((wslot - chunk_no) & (PW - 1)) * CC;
                chunk_no += nch;
            }
        };
        advance();
        // Column signatures: one coalesced load per chunk (64 columns, one per lane).  The NEXT chunk is requested
        // before this chunk is compared and first touched after it (the register hand-over at the bottom of the
        // loop), so its round trip overlaps the compare.  (An earlier version kept four chunks in flight but
        // rotated them through registers at the top of the loop: the rotation reads every pending register, so
        // the wave waited for the chunk it had requested a moment before — no overlap at all.)  The load is
        // unconditional from a clamped address, so nothing sits in a branch.
        int q0a = 0, cta = 0, cea = 0, q0b = 0, ctb = 0, ceb = 0;
        bool va = false, vb = false;
        uint32_t xa[W], xb[W];
#define PF_FETCH(q0x, ctx, cex, valid, xx)                                                \
    valid = it_ci < nc;                                                                   \
    q0x = it_q0;                                                                          \
    ctx = it_ctrue;                                                                       \
    cex = it_cend;                                                                        \
    _Pragma("unroll") for (int x = 0; x < W; x++) xx[x] = sig1[(size_t)min(max(it_q0, 0) + lane, n) * W + x]; \
    if (valid) advance();
        PF_FETCH(q0a, cta, cea, va, xa)
        while (va) {
            PF_FETCH(q0b, ctb, ceb, vb, xb)
            const int q = q0a + lane;                      // this lane's column
            const bool colok = q >= cta && q < cea;         // inside the range (chunks are aligned down/up)
            if (DBG) dbg_chunks++;
            slots += (long long)nsb * SB * (min(cea, q0a + CC) - max(cta, q0a));
#pragma unroll 1
            for (int sb = 0; sb < nsb; sb++) {
                uint32_t rsig[16];
#pragma unroll
                for (int x4 = 0; x4 < 4; x4++) {
                    const uint4 tt = *reinterpret_cast<const uint4 *>(&myrow[sb * 16 + x4 * 4]);
                    rsig[x4 * 4 + 0] = tt.x;
                    rsig[x4 * 4 + 1] = tt.y;
                    rsig[x4 * 4 + 2] = tt.z;
                    rsig[x4 * 4 + 3] = tt.w;
                }
                uint32_t m = 0xFFFFu;
#pragma unroll
                for (int j = 0; j < SB; j++) m = min(m, sigdist<W>(xa, &rsig[j * W]));
                // Hit: queue the COARSE fact (sub-batch, column); which of the SB rows it was is worked out in
                // flush_hits, lane-parallel.  The queue is private to the wave: the slot comes from the ballot
                // (v_mbcnt), the fill level is a wave-uniform register — no LDS atomic, no rescan.
                const bool h = m <= d && colok;
                const unsigned long long act = (DBG && (pa.dbg & 2)) ? 0ull : __builtin_amdgcn_ballot_w64(h);
                if (act != 0ull) {
                    if (qn > QCAP - 64) {  // make room
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        flush_hits<W>(pa, sig1, myq, qn, row0 + nrows, spairs[wave], qshard);
                        // drain: with an unknown number of stores in flight the compiler would wait for ALL loads
                        // (vmcnt(0)) at every sub-batch, i.e. for the chunk requested a moment ago; after a full
                        // drain it can count again
                        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                        qn = 0;
                    }
                    if (h) myq[qn + __popcll(act & ((1ull << lane) - 1ull))] = make_int2(row0 + sb * SB, q);
                    qn += __popcll(act);
                    if (DBG) dbg_hits += __popcll(act);
                }
            }
            // hand over the chunk that was in flight during the compare
            q0a = q0b; cta = ctb; cea = ceb; va = vb;
#pragma unroll
            for (int x = 0; x < W; x++) xa[x] = xb[x];
        }
#undef PF_FETCH
        if (lane == 0) {  // plain store (no statistics atomics); later rounds (d >= 64) accumulate
            int *ts = &ba.tile_slots[t * PW + wslot];
            *ts = (int)min((long long)(cbase ? *ts : 0) + slots, (long long)0x7fffffff);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (DBG) t_main = wall_clock64();
    if (qn > 0 && !(DBG && (pa.dbg & 1))) flush_hits<W>(pa, sig1, myq, qn, row0 + nrows, spairs[wave], qshard);  // one flush per wave
    if (DBG && ba.dbg_t && lane == 0) {
        unsigned long long *o = ba.dbg_t + (size_t)(t * PW + wslot) * 8;
        o[0] = t_start;
        o[1] = t_rng;
        o[2] = t_main;
        o[3] = wall_clock64();
        o[4] = (unsigned long long)dbg_chunks;
        o[5] = (unsigned long long)dbg_hits;
        o[6] = (unsigned long long)nrows;
        // where the wave ran: HW_ID (hwreg 4: simd [5:4], cu [11:8], sh [12], se [15:13]) and XCC_ID (hwreg 20)
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        o[7] = (unsigned long long)(unsigned)tile.z | ((unsigned long long)(hw & 0xFFFFu) << 32) |
               ((unsigned long long)(xcc & 0xFu) << 48);
    }
    item += (int)gridDim.x * PW;
    }  // items
}

// ------------------------------------------------------------------------------------------------
// k_verify: exact multiset distance of every queued candidate, one 16-lane group per pair (four per wave:
// the work is latency-bound, so the kernel is organised for pairs in flight, not lanes per pair).
// The rows are NOT sorted.  The group owns a 256-slot hash table in LDS ({token, signed count}); the tokens of
// row A are inserted with +1 and those of row B with -1, 16 at a time (CAS on the key claims a slot, equal
// tokens meet in the same slot, ds_add adds the sign); the distance is sum |count| over the table
// (= sum_t |a_t - b_t|, exact for multisets).  Pairs with more than VERIFY_MAX_TOKENS tokens are left to
// k_verify_long.  Edges are hooked into the union-find by the same kernel: lane i of a group keeps the group's
// i-th edge and all lanes hook theirs together at the end (the find / CAS chains of a wave's edges overlap).
// ------------------------------------------------------------------------------------------------
// the shard queues seen as one index space: prefix of min(ncand[s], cap), built once per block in LDS
// (first wave: 8 consecutive shards per lane, then a wave scan)
__device__ __forceinline__ void shard_prefix(const PairArgs &pa, int *pre /*CAND_SHARDS+1, LDS*/) {
    static_assert(CAND_SHARDS == 512, "8 shard counters per lane of the first wave");
    if (threadIdx.x < 64) {
        int c[8], sum = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c[j] = (int)min(pa.ctr->ncand[threadIdx.x * 8 + j], (unsigned)pa.cand_cap_shard);
            sum += c[j];
        }
        int run = wave_incl_scan_add(sum) - sum;
        if (threadIdx.x == 0) pre[0] = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            run += c[j];
            pre[threadIdx.x * 8 + j + 1] = run;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ size_t shard_slot(const int *pre, const PairArgs &pa, int c) {
    int lo = 0, hi = CAND_SHARDS - 1;  // largest s with pre[s] <= c
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pre[mid] <= c) lo = mid;
        else hi = mid - 1;
    }
    return (size_t)lo * pa.cand_cap_shard + (size_t)(c - pre[lo]);
}

// sum of x over the 16 lanes of a DPP row, valid in the row's lane 15
__device__ __forceinline__ int row16_sum_to_lane15(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);  // row_shr:8
    return x;
}

// insert token x with sign sgn into an open-addressing table of `mask`+1 slots (LDS or global)
__device__ __forceinline__ void table_add(uint32_t *tkey, int *tcnt, uint32_t mask, uint32_t x, int sgn) {
    uint32_t s = hash3(x) & mask;
    while (true) {
        const uint32_t old = atomicCAS(&tkey[s], 0xFFFFFFFFu, x);
        if (old == 0xFFFFFFFFu || old == x) break;
        s = (s + 1) & mask;
    }
    atomicAdd(&tcnt[s], sgn);
}

__device__ __forceinline__ void record_edge(const PairArgs &pa, int2 *edges, int edge_cap, int a, int b) {
    // (select_ind of get_neighbours_batch, breakfast.py:241-245: only the lists of the selected rows are wanted — an edge
    // between two rows that are not selected is in none of them)
    if (pa.sel && !(((pa.sel[a >> 5] >> (a & 31)) | (pa.sel[b >> 5] >> (b & 31))) & 1u)) return;
    const unsigned long long e = atomicAdd(&pa.ctr->n_edges_cap, 1ull);
    if (e < (unsigned long long)edge_cap) edges[e] = make_int2(min(a, b), max(a, b));
}

// all-reduce over the 16 lanes of a DPP row (row_ror:1,2,4,8): every lane gets the result
__device__ __forceinline__ int row16_allmin(int x) {
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x121, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x122, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x124, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false));
    return x;
}
__device__ __forceinline__ uint32_t row16_allmin_u(uint32_t x) {
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x121, 0xF, 0xF, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x122, 0xF, 0xF, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x124, 0xF, 0xF, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xF, 0xF, false));
    return x;
}
__device__ __forceinline__ uint32_t row16_allxor(uint32_t x) {
    x ^= (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x121, 0xF, 0xF, false);
    x ^= (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x122, 0xF, 0xF, false);
    x ^= (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x124, 0xF, 0xF, false);
    x ^= (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xF, 0xF, false);
    return x;
}
__device__ __forceinline__ int row16_allmax(int x) {
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x121, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x122, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x124, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false));
    return x;
}
__device__ __forceinline__ int row16_allsum(int x) {
    x += __builtin_amdgcn_update_dpp(x, x, 0x121, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(x, x, 0x122, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(x, x, 0x124, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false);
    return x;
}

// exact multiset distance of one pair by signed counting in the group's LDS table (the general path); the rows
// are already in registers: lane l16 holds positions 16 st + l16 of row A (a) and of row B (b0)
template <int STEPS, bool UNROLL>
__device__ __forceinline__ int table_distance(uint2 *mt, int l16, const uint32_t (&a)[STEPS], const uint32_t (&b0)[STEPS],
                                              int ka, int kb) {
    // table size: power of two >= 2 * tokens (load <= 1/2), capped at the group's table (load <= 3/4)
    const int kt = ka + kb;
    uint32_t tsz = 16;
    while ((int)tsz < 2 * kt && tsz < (uint32_t)verify_table(STEPS)) tsz <<= 1;
    const uint32_t mask = tsz - 1;
    for (uint32_t t = l16; t < tsz; t += 16) mt[t] = make_uint2(0xFFFFFFFFu, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto add = [&](uint32_t x, int sgn) {
        uint32_t p = hash3(x) & mask;
        while (true) {
            const uint32_t o = atomicCAS(&mt[p].x, 0xFFFFFFFFu, x);
            if (o == 0xFFFFFFFFu || o == x) break;
            p = (p + 1) & mask;  // occupied by another token: linear probing
        }
        atomicAdd(reinterpret_cast<int *>(&mt[p].y), sgn);
    };
    if (UNROLL) {  // the common path of this instantiation: straight-line code
#pragma unroll
        for (int st = 0; st < STEPS; st++) {
            const int j = st * 16 + l16;
            if (j < ka) add(a[st], 1);
            if (j < kb) add(b0[st], -1);
        }
    } else {  // the rare path: compact code
#pragma unroll 1
        for (int st = 0; st < STEPS; st++) {
            const int j = st * 16 + l16;
            uint32_t xa = a[0], xb = b0[0];  // register arrays are indexed by constants only
#pragma unroll
            for (int q = 1; q < STEPS; q++) {
                xa = st == q ? a[q] : xa;
                xb = st == q ? b0[q] : xb;
            }
            if (j < ka) add(xa, 1);
            if (j < kb) add(xb, -1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int dist = 0;
    for (uint32_t t = l16; t < tsz; t += 16) dist += abs((int)mt[t].y);
    dist = row16_allsum(dist);
    __builtin_amdgcn_wave_barrier();
    return dist;
}

// The exact test of one candidate by the 16 lanes of a group: is the multiset distance of rows rec.x and rec.y (tokens at
// rows + rec.z / rec.w, lengths kk) within pa.d?  Group-uniform result: 1 edge, 0 not an edge, -1 UNDECIDED — a pair with more
// than VERIFY_MAX_TOKENS tokens (its exact count does not fit a group table) that the certificates do not settle:
// k_verify_long takes it.  (Round 4: the certificates need no table, so pairs of long rows — 100+ tokens each, what a
// present-day SARS-CoV-2 profile against the Wuhan reference looks like — are tried here first as long as both rows fit the
// registers; before, every such pair went to k_verify_long, a block per pair: 2.5 ms for 68k candidates at 100k rows.)
// With WAVE_TABLE all four groups of the wave must call it together.
template <int STEPS, bool WAVE_TABLE>
__device__ __forceinline__ int verify_pair(const PairArgs &pa, uint2 *mt, int lane, int l16, const int4 &rec, const int2 &kk) {
    // lane l16 of the group holds, per step st, position j = 16 st + l16 of row A, of row B, and of row B
    // shifted by the length difference s = k_b - k_a
    auto tokens = [&](uint32_t(&a)[STEPS], uint32_t(&b0)[STEPS], uint32_t(&bs)[STEPS]) {
        const int sft = kk.y - kk.x;
        const uint32_t *A = pa.rows + rec.z + l16, *B = pa.rows + rec.w + l16;  // one address per row, then
        const uint32_t *Bs = B + sft;                                           // constant offsets 64 st
#pragma unroll
        for (int st = 0; st < STEPS; st++) {
            const int j = st * 16 + l16;
            a[st] = j < kk.x ? A[st * 16] : 0u;
            b0[st] = j < kk.y ? B[st * 16] : 0u;
            bs[st] = (uint32_t)(j + sft) < (uint32_t)kk.y ? Bs[st * 16] : 0u;
        }
    };
    uint32_t a[STEPS], b0[STEPS], bs[STEPS];
    tokens(a, b0, bs);
    const int ka = kk.x, kb = kk.y, kt = ka + kb;
    const bool small = kt <= verify_max_tokens(STEPS);  // the exact count fits a group table
    int verdict = -1;
    if (small || max(ka, kb) <= 16 * STEPS) {  // (both rows are in the registers in full)
        // Certificate first.  Profiles list their mutations in a fixed order (by genome position), so two
        // rows within d of each other are almost always the same sequence with a few tokens inserted: with
        // t = first position where the rows differ and s = k_b - k_a, the pairs (i, i) for i < t and
        // (i, i + s) for i >= max(t, t - s) with equal tokens are a matching of A into B (no position is
        // used twice), so distance <= k_a + k_b - 2 * matched holds for ANY two rows, whatever their
        // order.  If that bound is <= d the pair is an edge; otherwise it is counted exactly below.
        const int sft = kb - ka, kmin = min(ka, kb);
        uint32_t e0 = 0, es = 0;
#pragma unroll
        for (int st = 0; st < STEPS; st++) {
            const int j = st * 16 + l16;
            e0 |= (uint32_t)(j < kmin && a[st] == b0[st]) << st;
            es |= (uint32_t)(j < ka && (uint32_t)(j + sft) < (uint32_t)kb && a[st] == bs[st]) << st;
        }
        const int fst = __builtin_ctz(~e0);  // first step at which this lane's position differs (>= STEPS: none)
        const int t = min(row16_allmin(fst * 16 + l16), kmin);
        const int from = sft < 0 ? t - sft : t;              // first suffix position of row A
        const int st0 = max(0, (from - l16 + 15) >> 4);        // first step of this lane inside the suffix
        const int mine = st0 < STEPS ? __popc(es >> st0) : 0;
        const int matched = t + row16_allsum(mine);
        int dist = kt - 2 * matched;  // upper bound
        // not certified: count exactly
        if (WAVE_TABLE) {  // the wave's groups that need the table take turns
            unsigned long long want = __builtin_amdgcn_ballot_w64(dist > pa.d && small);
            while (want != 0ull) {
                const int g = (int)(__builtin_ctzll(want) >> 4);
                if ((lane >> 4) == g) dist = table_distance<STEPS, false>(mt, l16, a, b0, ka, kb);
                want &= ~(0xFFFFull << (g * 16));
            }
        } else if (dist > pa.d && pa.d > 3) {
            if (small) dist = table_distance<STEPS, true>(mt, l16, a, b0, ka, kb);  // exact
        } else if (dist > pa.d) {
            // Second certificate (d = 2, 3; with more edits allowed it rarely holds and only costs): pairs with two separate edits — e.g. one token inserted near
            // the front and one near the end — match under shift 0 before the first edit, under shift s
            // after the last, and under ONE other shift in between.  With r = last position of A that does
            // not match under shift s, the middle segment [t, r] is tried with shifts -1 and +1; its pairs
            // (i, i + sigma) are kept only while t <= i + sigma <= r + s, i.e. between the B positions
            // the prefix and the suffix use, so the three segments together are still a matching.
            int hi = -1;
#pragma unroll
            for (int st = 0; st < STEPS; st++) {
                const int j = st * 16 + l16;
                if (j >= from && j < ka && !((es >> st) & 1u)) hi = j;
            }
            const int r = max(row16_allmax(hi), from - 1);
            const uint32_t *Bn = pa.rows + rec.w + l16;
            int cm = 0, cp = 0;
#pragma unroll
            for (int st = 0; st < STEPS; st++) {
                const int j = st * 16 + l16;
                const bool mid = j >= t && j <= r && j < ka;
                const bool vm = mid && j - 1 >= t && j - 1 <= r + sft && j - 1 < kb;
                const bool vp = mid && j + 1 >= t && j + 1 <= r + sft && j + 1 < kb;
                const uint32_t xm = vm ? Bn[st * 16 - 1] : 0u, xp = vp ? Bn[st * 16 + 1] : 0u;
                cm += (int)(vm && a[st] == xm);
                cp += (int)(vp && a[st] == xp);
            }
            const int matched2 = t + max(row16_allsum(cm), row16_allsum(cp)) + (ka - 1 - r);
            dist = min(dist, kt - 2 * matched2);
            if (dist > pa.d && small) dist = table_distance<STEPS, true>(mt, l16, a, b0, ka, kb);  // exact
        }
        verdict = dist <= pa.d ? 1 : (small ? 0 : -1);  // (without the exact count a bound above d decides nothing)
    }
    return verdict;
}

// density-adaptive verify (PairArgs::density_mode): does this kernel sit the step out?  Block-uniform; one barrier.
__device__ __forceinline__ bool verify_sits_out(const PairArgs &pa) {
    __shared__ unsigned long long s_total;
    if (!pa.density_mode) return false;
    if (threadIdx.x < 64) {
        unsigned long long t = 0;
        for (int sh = threadIdx.x; sh < CAND_SHARDS; sh += 64) t += min(pa.ctr->ncand[sh], (unsigned)pa.cand_cap_shard);
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (threadIdx.x == 0) s_total = t;
    }
    __syncthreads();
    const bool dense = s_total > (unsigned long long)pa.density_thr;
    return (pa.density_mode == 1) == dense;
}

// STEPS x 16 >= longest row a pair of this kernel can have (pairs with more than VERIFY_MAX_TOKENS tokens in
// all are left to k_verify_long).  WAVE_TABLE: one hash table per wave, its groups take turns (d <= 1: nearly
// every candidate is certified without the table, and 8 KiB of LDS per block keeps 6+ blocks per CU resident);
// otherwise one table per group, all four groups of a wave count in parallel (d >= 2: pairs with several
// separate insertions fail the single-shift certificate).
template <int STEPS, bool WAVE_TABLE>
__global__ __launch_bounds__(256) void k_verify(PairArgs pa, int2 *edges, int edge_cap, int *blk_stats) {
    __shared__ uint2 tab[WAVE_TABLE ? 4 : 16][verify_table(STEPS)];  // {token, signed count}: exactly 32 KiB per block with group tables of
                                                                     // 256 slots, five blocks per CU (the block's two counters live in it at the end)
    const int lane = threadIdx.x & 63, l16 = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;  // 0..15 in the block
    uint2 *mt = tab[WAVE_TABLE ? (threadIdx.x >> 6) : grp];
    if (verify_sits_out(pa)) {  // (the other verify kernel of the step takes the queue)
        if (threadIdx.x == 0) blk_stats[pa.stats_off + 2 * blockIdx.x] = blk_stats[pa.stats_off + 2 * blockIdx.x + 1] = 0;
        return;
    }
    // Group u of U (a multiple of CAND_SHARDS) works on queue shard u % CAND_SHARDS, entries j0, j0 + jstep, ..:
    // the slot is computed, not searched, and the shards fill evenly (the prefilter rotates over them).
    const int u = blockIdx.x * 16 + grp, U = gridDim.x * 16;
    const int shard = u & (CAND_SHARDS - 1), j0 = u / CAND_SHARDS, jstep = U / CAND_SHARDS;
    const size_t base = (size_t)shard * pa.cand_cap_shard;
    // the shard's fill level and the group's first two queue records are requested together (the slots exist
    // whatever the fill level; records beyond it are never used)
    const int capm1 = pa.cand_cap_shard - 1;
    const unsigned cnt_raw = pa.ctr->ncand[shard];
    int4 rec = pa.cand[base + min(j0, capm1)];
    int2 kk = pa.candk[base + min(j0, capm1)];
    int4 rec_n = pa.cand[base + min(j0 + jstep, capm1)];
    int2 kk_n = pa.candk[base + min(j0 + jstep, capm1)];
    const int cnt = (int)min(cnt_raw, (unsigned)pa.cand_cap_shard);
    // lane i of the group keeps the group's i-th edge of the current batch; a batch is hooked when it is full and
    // at the end.  Batches of 8 at d = 3, where a group sees ~10 edges: the hooks then interleave with the
    // checks of other waves instead of arriving in one burst at the end of the kernel (100k rows, d = 3: 0.35 vs
    // 0.52 ms for the kernel with batches of 8 vs 16; 1M rows, d = 5: 16 is better; d <= 2 hooks by splicing: 16)
    int my_a = -1, my_b = -1;
    int nkept = 0;
    const int ubm = pa.union_batch - 1;
    // the part of the shard this launch works on (the whole shard unless the kernel runs in two phases)
    const int e_lo = (int)((long long)cnt * pa.part_lo / pa.part_den), e_hi = (int)((long long)cnt * pa.part_hi / pa.part_den);
    int e_first = j0;
    if (e_lo > j0) {
        e_first = j0 + (e_lo - j0 + jstep - 1) / jstep * jstep;
        rec = pa.cand[base + min(e_first, capm1)];
        kk = pa.candk[base + min(e_first, capm1)];
        rec_n = pa.cand[base + min(e_first + jstep, capm1)];
        kk_n = pa.candk[base + min(e_first + jstep, capm1)];
    }
    if (e_first < e_hi) {
        // A group has only ~2 candidates (32k groups are resident: 8 waves per SIMD), so latency is hidden by
        // occupancy, not by a deep pipeline per group: only the next queue record is fetched ahead.
        const int last = cnt - 1;
        for (int e = e_first; e < e_hi; e += jstep) {
            const int e2 = min(e + 2 * jstep, last);
            const int4 rec_nn = pa.cand[base + e2];
            const int2 kk_nn = pa.candk[base + e2];
            const bool is_edge = verify_pair<STEPS, WAVE_TABLE>(pa, mt, lane, l16, rec, kk) == 1;
            // a pair k_verify_long would take (too many tokens for a group table) that is settled here: out of its way
            if (is_edge && l16 == 0 && kk.x + kk.y > verify_max_tokens(STEPS)) pa.candk[base + e] = make_int2(0, 0);
            if (is_edge) {  // group-uniform
                if (l16 == (nkept & ubm)) {
                    my_a = rec.x;
                    my_b = rec.y;
                }
                nkept++;
                if ((nkept & ubm) == 0) {  // the batch is full: hook these edges now
                    if (my_a >= 0) {  // lanes beyond the batch size hold no edge
                        if (pa.use_link == 1) uf_link(pa.parent, my_a, my_b); else if (pa.use_link == 2) uf_link_checked(pa.parent, my_a, my_b); else uf_union(pa.parent, my_a, my_b);
                        if (edges) record_edge(pa, edges, edge_cap, my_a, my_b);
                    }
                    my_a = my_b = -1;
                }
            }
            rec = rec_n;
            kk = kk_n;
            rec_n = rec_nn;
            kk_n = kk_nn;
        }
    }
    // one edge per lane: the dependent find / CAS chains of all edges of the wave overlap
    if (my_a >= 0) {
        if (pa.use_link == 1) uf_link(pa.parent, my_a, my_b); else if (pa.use_link == 2) uf_link_checked(pa.parent, my_a, my_b); else uf_union(pa.parent, my_a, my_b);
        if (edges) record_edge(pa, edges, edge_cap, my_a, my_b);
    }
    __syncthreads();  // every group is done with its table
    unsigned *blk = reinterpret_cast<unsigned *>(&tab[0][0]);  // [0] edges, [1] candidates of the block
    if (threadIdx.x == 0) blk[0] = blk[1] = 0u;
    __syncthreads();
    if (l16 == 0 && nkept) atomicAdd(&blk[0], (unsigned)nkept);
    if (j0 == 0 && l16 == 0 && cnt && pa.part_lo == 0) atomicAdd(&blk[1], (unsigned)cnt);
    __syncthreads();
    if (threadIdx.x == 0) {  // plain stores, summed by the host (no same-word global atomics)
        blk_stats[pa.stats_off + 2 * blockIdx.x] = (int)blk[0];
        blk_stats[pa.stats_off + 2 * blockIdx.x + 1] = (int)blk[1];
    }
}

// k_verify_connected: k_verify for steps that only want the components (no edge list, no exact edge count).  A candidate
// whose rows are in one tree already cannot change the result, so its distance is not needed: each lane looks one queue
// record up and compares the roots of its two rows (64 finds of a wave in flight together); the survivors are packed in
// LDS and the wave's four groups take them four at a time through the same exact test as k_verify; the edges of the round
// are then hooked one per lane.  In the dense graphs of max_dist >= 3 (1M rows, max_dist 5: 2.1e7 candidates, 1.3e7 edges,
// 68 components) nearly all of the queue is dropped once the first unions have gone in.  A stale "not connected" only costs
// a check; "connected" is never stale (trees only merge), and the labels — the smallest row of each component — do not
// depend on which edges were used.
template <int STEPS, bool WAVE_TABLE>
__global__ __launch_bounds__(256) void k_verify_connected(PairArgs pa, int *blk_stats) {
    __shared__ uint2 tab[WAVE_TABLE ? 4 : 16][verify_table(STEPS)];  // {token, signed count}
    __shared__ int4 s_rec[4][64];                              // the round's survivors of each wave
    __shared__ int2 s_kk[4][64];
    __shared__ unsigned int blk_edges, blk_cands, blk_conn;
    const int lane = threadIdx.x & 63, l16 = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;  // 0..15 in the block
    const int wave = threadIdx.x >> 6, gw = (threadIdx.x >> 4) & 3;
    uint2 *mt = tab[WAVE_TABLE ? wave : grp];
    if (verify_sits_out(pa)) {  // (the other verify kernel of the step takes the queue)
        if (threadIdx.x == 0) blk_stats[pa.stats_off + 2 * blockIdx.x] = blk_stats[pa.stats_off + 2 * blockIdx.x + 1] = 0;
        return;
    }
    if (threadIdx.x == 0) blk_edges = blk_cands = blk_conn = 0;
    __syncthreads();
    // the same map of groups to queue entries as k_verify: group u works on shard u % CAND_SHARDS, entries j0, j0 + jstep, ..
    const int u = blockIdx.x * 16 + grp, U = gridDim.x * 16;
    const int shard = u & (CAND_SHARDS - 1), j0 = u / CAND_SHARDS, jstep = U / CAND_SHARDS;
    const size_t base = (size_t)shard * pa.cand_cap_shard;
    const int cnt = (int)min(pa.ctr->ncand[shard], (unsigned)pa.cand_cap_shard);
    if (j0 == 0 && l16 == 0 && cnt && pa.part_lo == 0) atomicAdd(&blk_cands, (unsigned)cnt);
    const int e_lo = (int)((long long)cnt * pa.part_lo / pa.part_den), e_hi = (int)((long long)cnt * pa.part_hi / pa.part_den);
    auto below = [](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
    unsigned n_edges = 0, n_conn = 0;  // wave-uniform
    // A group reads 16 CONSECUTIVE records (blocks of 16, jstep blocks apart), not 16 records jstep apart as k_verify does:
    // the queue is read once, in full lines (1M rows, max_dist 5: 0.95 -> 0.75 ms for this kernel; the records of a block are
    // mostly pairs of one row, which are then looked up together — 8% more exact tests).  The first block of a group is taken
    // in pieces of 1, 2, 4, 8 records: in the first rounds nothing is connected yet and everything a wave looks at has to be
    // checked, so they are kept short until the first unions have gone in (a queue of a few records per lane would otherwise
    // be one round).
    // A round is a chain of dependent loads (record -> parents -> roots -> rows -> unions) and the kernel runs at 4 waves per
    // SIMD (LDS): the records are asked for one round ahead.  (The parents too, one round ahead: slower — 0.63 -> 0.65 ms at 1M
    // rows, max_dist 5, 0.19 -> 0.30 at 30k — a parent that is a round old says "not connected" where "connected" is true by
    // now, and the exact tests that costs, +16% .. +90%, outweigh the round trip.)
    int blk = j0, width = 1, lo = 0;
    struct Slot {
        int4 rec;
        int2 kk;
        int p0, p1, e;
        bool have, more;
    };
    auto next_slot = [&]() {  // the next piece of the group's records; its queue record is requested
        Slot sl;
        const int e = blk * 16 + l16;
        sl.more = (long long)blk * 16 < e_hi;
        sl.have = l16 >= lo && l16 < lo + width && e >= e_lo && e < e_hi;
        lo += width;
        width = min(16, width * 2);
        if (lo >= 16) {
            lo = 0;
            width = 16;
            blk += jstep;
        }
        sl.rec = make_int4(0, 0, 0, 0);
        sl.kk = make_int2(0, 0);
        sl.p0 = sl.p1 = 0;
        sl.e = e;
        if (sl.have) {
            sl.rec = pa.cand[base + e];
            sl.kk = pa.candk[base + e];
        }
        return sl;
    };
    auto ask_parents = [&](Slot &sl) {
        // (pairs whose rows do not fit the registers are k_verify_long's, which looks their roots up itself; those with too many
        // tokens for a group table are looked up and tried by the certificates here — what that settles is taken out of its way)
        sl.have = sl.have && (sl.kk.x + sl.kk.y <= verify_max_tokens(STEPS) || max(sl.kk.x, sl.kk.y) <= 16 * STEPS);
        if (sl.have) {
            sl.p0 = ld_agent(pa.parent + sl.rec.x);
            sl.p1 = ld_agent(pa.parent + sl.rec.y);
        }
    };
    Slot sA = next_slot();
    for (;;) {
        Slot cur = sA;
        if (__builtin_amdgcn_ballot_w64(cur.more) == 0ull) break;
        sA = next_slot();
        ask_parents(cur);
        int4 rec = cur.rec;
        const int2 kk = cur.kk;
        bool todo = false, conn = false;
        if (cur.have) {
            todo = cur.p0 != cur.p1 && uf_find_from(pa.parent, rec.x, cur.p0) != uf_find_from(pa.parent, rec.y, cur.p1);
            conn = !todo;
        }
        const unsigned long long tm = __builtin_amdgcn_ballot_w64(todo);
        n_conn += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(conn));
        const int ns = __popcll(tm), rank = below(tm);
        if (todo) {
            if (rec.w < 0) rec.w = pa.indptr[rec.y];  // (k_pgwalk16 leaves B's offset to the few records that get this far)
            s_rec[wave][rank] = rec;
            s_kk[wave][rank] = kk;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned long long em = 0ull;  // survivors (by rank) that are edges
        for (int i = 0; i < ns; i += 4) {
            const bool valid = i + gw < ns;
            const int4 r = valid ? s_rec[wave][i + gw] : make_int4(0, 0, 0, 0);
            const int2 k = valid ? s_kk[wave][i + gw] : make_int2(0, 0);
            const bool is_edge = verify_pair<STEPS, WAVE_TABLE>(pa, mt, lane, l16, r, k) == 1 && valid;
            const unsigned long long b = __builtin_amdgcn_ballot_w64(is_edge);
            em |= ((b & 1ull) | ((b >> 15) & 2ull) | ((b >> 30) & 4ull) | ((b >> 45) & 8ull)) << i;
        }
        __builtin_amdgcn_wave_barrier();
        n_edges += (unsigned)__popcll(em);
        if (cur.have && kk.x + kk.y > verify_max_tokens(STEPS) && (conn || (todo && ((em >> rank) & 1ull))))
            pa.candk[base + cur.e] = make_int2(0, 0);  // settled here: nothing left for k_verify_long
        if (todo && ((em >> rank) & 1ull)) {  // one edge per lane: the chains of the wave's edges overlap
            if (pa.use_link == 1) uf_link(pa.parent, rec.x, rec.y); else if (pa.use_link == 2) uf_link_checked(pa.parent, rec.x, rec.y); else uf_union(pa.parent, rec.x, rec.y);
        }
    }
    if (lane == 0) {
        if (n_edges) atomicAdd(&blk_edges, n_edges);
        if (n_conn) atomicAdd(&blk_conn, n_conn);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // plain stores, summed by the host (no same-word global atomics)
        blk_stats[pa.stats_off + 2 * blockIdx.x] = (int)blk_edges;
        blk_stats[pa.stats_off + 2 * blockIdx.x + 1] = (int)blk_cands;
        if (blk_conn) atomicAdd(&pa.ctr->n_connected, (unsigned long long)blk_conn);
    }
}

// k_verify_long: pairs with more tokens than a group table holds: one block per pair, table in dynamic LDS
// (up to LONG_TABLE slots) or, beyond that, in the block's slice of a global scratch table.
__global__ __launch_bounds__(256) void k_verify_long(PairArgs pa, uint32_t *gkey, int *gcnt, unsigned gslots, int2 *edges,
                                                     int edge_cap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t ldyn[];
    __shared__ int spre[CAND_SHARDS + 1];
    __shared__ int sdist;
    shard_prefix(pa, spre);
    const int total = spre[CAND_SHARDS];
    for (int c = blockIdx.x; c < total; c += gridDim.x) {
        const size_t slot = shard_slot(spre, pa, c);
        const int2 kk = pa.candk[slot];
        const int ka = kk.x, kb = kk.y, kt = ka + kb;
        if (kt <= pa.max_tokens) continue;  // done by k_verify (its group tables hold that many)
        int4 rec = pa.cand[slot];
        if (rec.w < 0) rec.w = pa.indptr[rec.y];  // (k_pgwalk16)
        if (pa.skip_connected) {  // one thread decides for the block (trees merge while the block looks)
            if (threadIdx.x == 0) {
                sdist = uf_find(pa.parent, rec.x) == uf_find(pa.parent, rec.y) ? 1 : 0;
                if (sdist) atomicAdd(&pa.ctr->n_connected, 1ull);
            }
            __syncthreads();
            const bool conn = sdist != 0;
            __syncthreads();
            if (conn) continue;
        }
        uint32_t tsz = 1024;
        while ((unsigned long long)tsz * 3ull < (unsigned long long)kt * 4ull) tsz <<= 1;
        uint32_t *mk;
        int *mc;
        if (tsz <= LONG_TABLE) {
            mk = ldyn;
            mc = reinterpret_cast<int *>(ldyn + tsz);
        } else {
            if (tsz > gslots || !gkey) {  // cannot happen: the scratch table is sized from the longest row at bind time
                if (threadIdx.x == 0) atomicOr(&pa.ctr->err, ERR_WORKCAP);
                continue;
            }
            mk = gkey + (size_t)blockIdx.x * gslots;
            mc = gcnt + (size_t)blockIdx.x * gslots;
        }
        const uint32_t mask = tsz - 1;
        for (uint32_t t = threadIdx.x; t < tsz; t += 256) {
            mk[t] = 0xFFFFFFFFu;
            mc[t] = 0;
        }
        if (threadIdx.x == 0) sdist = 0;
        __threadfence_block();
        __syncthreads();
        const uint32_t *A = pa.rows + rec.z, *B = pa.rows + rec.w;
        for (int j = threadIdx.x; j < kt; j += 256) {
            const bool fromA = j < ka;
            table_add(mk, mc, mask, fromA ? A[j] : B[j - ka], fromA ? 1 : -1);
        }
        __threadfence_block();
        __syncthreads();
        int dist = 0;
        for (uint32_t t = threadIdx.x; t < tsz; t += 256) dist += abs(__hip_atomic_load(&mc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (dist) atomicAdd(&sdist, dist);
        __syncthreads();
        if (threadIdx.x == 0 && sdist <= pa.d) {
            uf_union(pa.parent, rec.x, rec.y);
            atomicAdd(&pa.ctr->n_edges, 1ull);
            if (edges) record_edge(pa, edges, edge_cap, rec.x, rec.y);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The variant join (max_dist == 1): sub-quadratic candidate generation.
//   Two rows are within distance 1 iff they are equal as multisets or one is the other plus ONE token occurrence.
//   With an additive multiset hash H(row) = (sum h1(t), sum h2(t)) mod 2^32 each, B = A + {t} implies
//   H(A) = H(B) - h(t) EXACTLY, so: k_jhash puts every row into a hash table keyed by H (and sets one bit of a
//   presence bitmap); k_join looks up H(B) - h(t) for every token occurrence t of every row B (first the bitmap —
//   one dependent load, 97% of the lookups end there — then the table), plus H(B) itself for equal multisets.
//   No pair within distance 1 can be missed; every table match goes through the same exact check as the
//   all-pairs path (k_verify), so hash collisions cost time, never results.  O(nnz) lookups instead of O(N^2)
//   signature compares.  A lookup chain longer than JOIN_MAX_PROBE (hundreds of rows that are the same multiset
//   in different orders) raises Counters::join_fail and the host re-runs the step on the all-pairs path.
//   The table and bitmap exist twice: the set of the NEXT step is cleared by this step's k_jhash.
// ------------------------------------------------------------------------------------------------
// token hashes of the row hash: NONLINEAR (with plain multiplicative hashes the additive row hash would be a
// function of the token sum and rows with equal sums would all look alike) and built from v_mul_u32_u24, which
// issues at full rate (v_mul_lo_u32 is quarter rate; with two 32-bit multiplies per hash the hashing alone was
// ~8 us of k_jhash at 100k rows).  Only the low 24 bits of a token id enter: ids that differ above bit 23 share
// their hashes, which costs false candidates and never a result.
__device__ __forceinline__ uint32_t jh_stage(uint32_t x) {
    uint32_t a = __umul24(x, 0xB5297Au);
    return a ^ (a >> 16);
}
__device__ __forceinline__ uint32_t jh1_of(uint32_t a) {
    a = __umul24(a, 0x68E31Du) + (a >> 9);
    return a ^ (a >> 15);
}
__device__ __forceinline__ uint32_t jh2_of(uint32_t a) {
    a = __umul24(a ^ 0x5BD1E9u, 0x1B873Bu) + (a >> 7);
    return a ^ (a >> 13);
}
__device__ __forceinline__ uint32_t jh1(uint32_t x) { return jh1_of(jh_stage(x)); }
__device__ __forceinline__ uint32_t jh2(uint32_t x) { return jh2_of(jh_stage(x)); }

// the presence bitmap is a blocked Bloom filter: a row sets TWO bits of ONE 32-bit word (word = bits 5.. of H.x, the
// bits = H.x bits 0..4 and 27..31), a lookup loads that one word — the same single scattered load as with one bit —
// and the false positives drop from 4.7% of the lookups to 0.4% (32 bits per row: 6% of the bits are set)
__device__ __forceinline__ uint32_t join_bloom_mask(uint32_t k1) { return (1u << (k1 & 31u)) | (1u << (k1 >> 27)); }

// sum over the 64 lanes, valid in lane 63 (same DPP pattern as wave_xor_to_lane63)
__device__ __forceinline__ uint32_t wave_add_to_lane63(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true);
    return x;
}

// device-driven text steps (JoinArgs::dyn): is the CSR unusable (the tokeniser failed, a negative row length) / outside what the
// host assumed when it sized the launch (a row over JOIN_INLINE_ROW tokens, no token at all)?  Wave-uniform scalar loads.
__device__ __forceinline__ bool join_dyn_unusable(const int *dyn) { return (dyn[1] | dyn[10]) != 0; }
__device__ __forceinline__ bool join_dyn_outside(const int *dyn) { return dyn[0] > JOIN_INLINE_ROW || dyn[3] <= 0; }

// k_jhash: same row-to-wave layout as k_sig.  H of every row -> rowhash[i], bitmap bit, table entry
// {H.y (tag) : row} by linear probing (load <= 1/8: 94% of the inserts take one CAS); parent[i] = i; counters
// reset; next step's table set cleared.  A failed CAS returns the entry in the way: if its tag is this row's, the
// two rows are (up to a 64-bit hash collision) the same multiset — the pair goes to the dup list, so k_join needs
// no lookup of H itself.  Entries with one H start probing at the same slot, so the later row passes every
// earlier one: each such pair is listed exactly once.
__global__ __launch_bounds__(1024) void k_jhash(const int *__restrict__ indptr, const uint32_t *__restrict__ indices, int n,
                                                 int nnz, int kcap, int rpw, JoinArgs ja, int *__restrict__ parent,
                                                 Counters *ctr) {
    constexpr int MAXRPW = 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (ja.dyn) {  // a text step whose bind the host has not completed: counts from the device (JoinArgs::dyn)
        // the step's counters to the host, first thing (also when the step is unusable: the host learns it from them) — ONE
        // 64-byte store across PCIe, acknowledged while the kernel works
        if (ja.dyn_host && blockIdx.x == 0 && threadIdx.x < 16) ja.dyn_host[threadIdx.x] = ja.dyn[threadIdx.x];
        if (join_dyn_unusable(ja.dyn)) return;
        nnz = ja.dyn[3];
        kcap = ja.dyn[0];
        if (join_dyn_outside(ja.dyn)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) ctr->join_fail = 2;
            return;
        }
    }
    if (!(ja.dbg & 16)) {  // clear the other table set for the next step (this step never touches it)
        const size_t tid = (size_t)blockIdx.x * 1024 + threadIdx.x, nth = (size_t)gridDim.x * 1024;
        ulonglong2 *t2 = reinterpret_cast<ulonglong2 *>(ja.tab_next);
        for (size_t i = tid; i < ((size_t)ja.mask + 1) / 2; i += nth) t2[i] = make_ulonglong2(JOIN_EMPTY, JOIN_EMPTY);
        uint4 *b4 = reinterpret_cast<uint4 *>(ja.bits_next);
        for (size_t i = tid; i < ((size_t)ja.bmask + 1) / 128; i += nth) b4[i] = make_uint4(0, 0, 0, 0);
    }
    if (blockIdx.x == 0) {  // what k_cells does on the all-pairs path
        if (threadIdx.x < CAND_SHARDS) ctr->ncand[threadIdx.x] = 0;
        if (threadIdx.x == 64) {
            ctr->err = 0;
            ctr->overflow = 0;
            ctr->n_work = 0;
            ctr->n_edges = ctr->n_cand_total = ctr->n_edges_cap = ctr->n_connected = 0;
        }
    }
    const int r0 = blockIdx.x * rpw * 16 + wave * rpw;
    const int nr = max(0, min(rpw, n - r0));
    if (nr <= 0) return;
    uint32_t my1 = 0, my2 = 0;  // lane t: H of row r0 + t
    unsigned long long first_old = JOIN_EMPTY;  // what the first CAS of this lane's row returned
    bool tried = false;
    const int ext = indptr[min(r0 + min(lane, rpw), n)];  // lane l: start of row r0 + l (the host sends nnz = 0 elsewhere)
    {
        uint32_t xs[MAXRPW];
#pragma unroll
        for (int t = 0; t < MAXRPW; t++) {
            const int bt = __builtin_amdgcn_readlane(ext, t);
            xs[t] = indices[min(max(bt, 0) + lane, nnz - 1)];
        }
#pragma unroll
        for (int t = 0; t < MAXRPW; t++) {
            if (t >= nr) continue;  // wave-uniform
            if (t == 8 && lane < 8 && !(ja.dbg & 8)) {
                // rows 0..7 are hashed: their first insert attempt goes out now, the hashing of rows 8.. hides its
                // round trip (all the inserts at the end were 3.6 us of exposed tail)
                first_old = atomicCAS(&ja.tab[my1 & ja.mask], JOIN_EMPTY,
                                      ((unsigned long long)my2 << 32) | (unsigned long long)(uint32_t)(r0 + lane));
                tried = true;
            }
            const int b = __builtin_amdgcn_readlane(ext, t), e = __builtin_amdgcn_readlane(ext, t + 1);
            int k = e - b;
            if (k < 0 || k > kcap) {
                if (lane == 0) atomicOr(&ctr->err_rows, ERR_ROWLEN);
                k = k < 0 ? 0 : kcap;
            }
            uint32_t a1 = 0, a2 = 0;
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int rem = k - j0;
                const uint32_t x = j0 == 0 ? xs[t] : ((lane < rem) ? indices[b + j0 + lane] : 0u);
                const uint32_t hs = jh_stage(x);
                a1 += lane < rem ? jh1_of(hs) : 0u;
                a2 += lane < rem ? jh2_of(hs) : 0u;
            }
            a1 = wave_add_to_lane63(a1);
            a2 = wave_add_to_lane63(a2);
            const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)a1, 63), t2 = (uint32_t)__builtin_amdgcn_readlane((int)a2, 63);
            if (lane == t) {
                my1 = t1;
                my2 = t2;
            }
        }
    }
    const int my_b = ext, my_e = min(__shfl_down(ext, 1), nnz);  // this lane's row
    if (lane < nr) {
        const int i = r0 + lane;
        ja.rowhash[i] = make_uint2(my1, my2);
        parent[i] = i;
        // k_join's waves own JOIN_TPW tokens each: the row that holds a wave's first token (extents from the
        // registers: a load here would sit in front of the insert below, one more round trip at the kernel's tail)
        for (int q = (my_b + JOIN_TPW - 1) / JOIN_TPW; q * JOIN_TPW < my_e; q++) ja.batch_row[q] = i;
        if (!(ja.dbg & 256)) atomicOr(&ja.bits[(my1 & ja.bmask) >> 5], join_bloom_mask(my1));
        const unsigned long long ent = ((unsigned long long)my2 << 32) | (unsigned long long)(uint32_t)i;
        uint32_t s = my1 & ja.mask;
        for (int probes = 0; !(ja.dbg & 8); probes++) {
            const unsigned long long old = (probes == 0 && tried) ? first_old : atomicCAS(&ja.tab[s], JOIN_EMPTY, ent);
            if (old == JOIN_EMPTY) break;
            if ((uint32_t)(old >> 32) == my2) {  // an earlier row with this H: (up to a hash collision) the same multiset
                const unsigned q = atomicAdd(&ctr->n_dup, 1u);
                if (q < (unsigned)ja.dup_cap) ja.dups[q] = make_int2((int)(uint32_t)old, i);
                else ctr->join_fail = 1;
            }
            if (probes >= JOIN_MAX_PROBE) {
                ctr->join_fail = 1;
                break;
            }
            s = (s + 1) & ja.mask;
        }
    }
}

// k_join: the lookups, the check of the matches and the unions.  TOKEN-parallel: a wave owns JOIN_TPW = 512
// consecutive entries of `indices` (8 windows of 64), whatever rows they belong to — all 64 lanes work, there is no
// per-row instruction stream and no special case for long rows (a wave per row kept 40 of 64 lanes busy and spent
// ~170 wave-instructions per row: 38 us at 100k rows).  Per wave:
//   (A) the 8 windows' tokens are requested at once, together with the extents and the hashes of the 64 rows from
//       the row that holds the wave's first token (k_jhash left that row in batch_row[]); a lane finds its row by
//       counting the row boundaries at or below its token (a short wave-uniform loop: ~1.6 boundaries per window),
//       takes the row's H from the hash register of that row's lane (ds_bpermute) and requests the bitmap word of
//       H - h(token).  A scattered load costs one request per distinct line: lanes with nothing to ask all ask
//       for word 0 through an UNCONDITIONAL load of a SELECTED address (a load under a lane mask made the compiler
//       wait for each load in turn).  Batches whose rows do not fit the 64 loaded extents (runs of tiny or empty
//       rows) find their rows by binary search (slow_window);
//   (B) bitmap hits -> the wave's LDS queue {key.x, key.y, row B, position p};
//   drain() probes the table for a whole queue at once, one lookup per lane, two slots per step;
//   settle() takes the matches (A, B, p) — B minus its token at position p hashes like A — and checks them right
//     here: profiles list their mutations in one order, so A is almost always B with position p deleted, i.e.
//     A[j] == B[j + (j >= p)] for all j, a 64-lane compare of two coalesced loads (four matches in flight); a
//     certified pair is hooked into the union-find at once, one edge per lane.  A token that occurs several times
//     in B is looked up at every occurrence: the pair counts at the first one only.  Whatever fails the test
//     (another token order, a hash collision) goes to the candidate queue and k_verify's exact count.
// Exact multiset distance of two rows of at most JOIN_INLINE_ROW tokens each, by a whole wave: every lane holds two tokens of A
// and two of B, the rows pass by once (64 rotations), and a token counts |cnt_A - cnt_B| at its first occurrence (a token of B
// that A does not have: at its first occurrence in B).  ~500 cross-lane reads per pair — for the matches the positional
// certificate cannot decide (another token order, equal multisets, hash collisions): none at all on ordered profiles.
// Returns (wave-uniform) whether the distance is <= d.  0xFFFFFFFF is not a token (k_verify's tables reserve it too).
__device__ __forceinline__ bool wave_rows_within(const uint32_t *__restrict__ indices, int ba, int ka, int bb, int kb, int d, int lane) {
    constexpr uint32_t NA = 0xFFFFFFFFu;
    uint32_t a[2], b[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        a[r] = r * 64 + lane < ka ? indices[ba + r * 64 + lane] : NA;
        b[r] = r * 64 + lane < kb ? indices[bb + r * 64 + lane] : NA;
    }
    int ca[2] = {0, 0}, cb[2] = {0, 0}, ca_b[2] = {0, 0}, cb_b[2] = {0, 0};
    bool first_a[2] = {true, true}, first_b[2] = {true, true};
    for (int s = 0; s < 64; s++) {
        const int src = (lane + s) & 63;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t xa = (uint32_t)__shfl((int)a[q], src), xb = (uint32_t)__shfl((int)b[q], src);
            const int pos = q * 64 + src;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int mine = r * 64 + lane;
                ca[r] += xa == a[r] ? 1 : 0;
                cb[r] += xb == a[r] ? 1 : 0;
                first_a[r] = first_a[r] && !(xa == a[r] && pos < mine);
                ca_b[r] += xa == b[r] ? 1 : 0;
                cb_b[r] += xb == b[r] ? 1 : 0;
                first_b[r] = first_b[r] && !(xb == b[r] && pos < mine);
            }
        }
    }
    int dist = 0;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (a[r] != NA && first_a[r]) dist += abs(ca[r] - cb[r]);
        if (b[r] != NA && first_b[r] && ca_b[r] == 0) dist += cb_b[r];
    }
    for (int o = 32; o > 0; o >>= 1) dist += __shfl_xor(dist, o);
    return dist <= d;
}

__global__ __launch_bounds__(1024, 8) void k_join(const int *__restrict__ indptr, const uint32_t *__restrict__ indices, int n,
                                                   int nnz, JoinArgs ja, PairArgs pa, int shard0, int nshards, int2 *edges,
                                                   int edge_cap) {
    constexpr int QCAP = 128, MCAP = 64;
    // one 16-byte record per entry (one ds_write_b128 / ds_read_b128 and one address instead of four)
    __shared__ int4 q_rec[16][QCAP];  // queued lookups: {key.x, key.y, row B, position of the looked-up token in B}
    __shared__ int4 m_rec[16][MCAP];  // matches: {row A, row B, position, -}
    __shared__ unsigned s_edges, s_cands;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int cshard = (blockIdx.x * 16 + wave) & (CAND_SHARDS - 1);
    if (ja.dyn) {  // (see k_jhash)
        if (join_dyn_unusable(ja.dyn) || join_dyn_outside(ja.dyn)) return;
        nnz = ja.dyn[3];
    }
    if (threadIdx.x == 0) s_edges = s_cands = 0;
    __syncthreads();

    // append (A, B) to the candidate queue of k_verify; `want` lanes hold one pair each (wave-uniform call)
    auto enqueue = [&](bool want, int A, int B, int ba, int bb, int ka, int kb) {
        const unsigned long long mm = __builtin_amdgcn_ballot_w64(want);
        if (mm == 0ull) return;
        cshard = (cshard + 1) & (CAND_SHARDS - 1);
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(&pa.ctr->ncand[cshard], (unsigned)__popcll(mm));
        base = __builtin_amdgcn_readfirstlane(base);
        if (want) {
            const int idx = base + __popcll(mm & lt);
            if (idx < pa.cand_cap_shard) {
                const size_t o = (size_t)cshard * pa.cand_cap_shard + idx;
                pa.cand[o] = make_int4(A, B, ba, bb);
                pa.candk[o] = make_int2(ka, kb);
            } else {
                pa.ctr->overflow = 1;
            }
        }
    };

    unsigned my_edges = 0, my_cands = 0;
    // what the certificate cannot decide: to k_verify's queue — or, while no row is longer than JOIN_INLINE_ROW tokens
    // (ja.inline_exact: a property of the bound CSR), decided right here by the exact count, pair after pair; the step
    // then has no k_verify launch at all (k_flatten checks that the queue stayed empty)
    auto resolve = [&](bool want, int A, int B, int ba, int bb, int ka, int kb) {
        if (!ja.inline_exact) {
            enqueue(want, A, B, ba, bb, ka, kb);
            return;
        }
        bool is_edge = false;
        for (unsigned long long m = __builtin_amdgcn_ballot_w64(want); m != 0ull; m &= m - 1ull) {
            const int src = (int)__builtin_ctzll(m);
            const bool e = wave_rows_within(indices, __shfl(ba, src), __shfl(ka, src), __shfl(bb, src), __shfl(kb, src), pa.d, lane);
            if (lane == src) is_edge = e;
        }
        if (is_edge) {
            uf_link(pa.parent, A, B);
            if (edges) record_edge(pa, edges, edge_cap, A, B);
        }
        my_edges += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(is_edge));
        my_cands += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(want));
    };

    {   // rows k_jhash found to share one H: the rank (later row) % nshards owns the pair (the same on every rank)
        const int ndup = (int)min(pa.ctr->n_dup, (unsigned)ja.dup_cap);
        const int per = (int)gridDim.x * 1024;
        for (int i0 = (int)blockIdx.x * 1024 + wave * 64; i0 < ndup; i0 += per) {  // wave-uniform trip count
            const int i = i0 + lane;
            bool mine = false;
            int A = 0, B = 0, ba = 0, bb = 0, ea = 0, eb = 0;
            if (i < ndup) {
                const int2 pr = ja.dups[i];
                A = min(pr.x, pr.y);
                B = max(pr.x, pr.y);
                mine = B % nshards == shard0;
                if (mine) {
                    ba = indptr[A];
                    ea = indptr[A + 1];
                    bb = indptr[B];
                    eb = indptr[B + 1];
                }
            }
            resolve(mine, A, B, ba, bb, ea - ba, eb - bb);
        }
    }
    // multi-GPU: the blocks (16 x 512 tokens, i.e. their lookups) are dealt round-robin
    const int gw = blockIdx.x * 16 + wave;
    const int T0 = gw * JOIN_TPW;  // (the host keeps nnz + JOIN_TPW below 2^31)
    if ((int)(blockIdx.x % (unsigned)nshards) == shard0 && T0 < nnz) {
        int4 *qr = q_rec[wave], *mr = m_rec[wave];
        int nq = 0, nm = 0;  // wave-uniform

        auto settle = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (ja.dbg & 1) nm = 0;
            const bool have = lane < nm;
            const int4 mrec = have ? mr[lane] : make_int4(0, 0, 0, 0);
            const int A = mrec.x, B = mrec.y, p = mrec.z;
            const int ba = have ? indptr[A] : 0, ea = have ? indptr[A + 1] : 0;
            const int bb = have ? indptr[B] : 0, eb = have ? indptr[B + 1] : 0;
            const int ka = ea - ba, kb = eb - bb;
            // B minus one token has k_b - 1 tokens: anything else is a hash collision
            const bool live = have && kb == ka + 1;
            const uint32_t tok = live ? indices[bb + p] : 0u;  // the token whose deletion was looked up
            bool cert = false, dup = false;
            for (int m0 = 0; m0 < nm; m0 += 4) {  // four matches in flight
                uint32_t av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int m = min(m0 + u, nm - 1);
                    const int ba_m = __builtin_amdgcn_readlane(ba, m), bb_m = __builtin_amdgcn_readlane(bb, m);
                    const int p_m = __builtin_amdgcn_readlane(p, m);
                    av[u] = indices[min(max(ba_m, 0) + lane, nnz - 1)];
                    bv[u] = indices[min(max(bb_m, 0) + lane + (lane >= p_m ? 1 : 0), nnz - 1)];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int m = min(m0 + u, nm - 1);
                    const int ka_m = __builtin_amdgcn_readlane(ka, m), p_m = __builtin_amdgcn_readlane(p, m);
                    const uint32_t tok_m = (uint32_t)__builtin_amdgcn_readlane((int)tok, m);
                    const bool ok = __builtin_amdgcn_ballot_w64(lane < ka_m && av[u] != bv[u]) == 0ull;
                    // the pair counts at the first occurrence of the token in B only (lanes below p hold B[lane])
                    const bool again = __builtin_amdgcn_ballot_w64(lane < p_m && bv[u] == tok_m) != 0ull;
                    if (lane == m) {
                        cert = ok;
                        dup = again;
                    }
                }
            }
            // rows over 64 tokens (rare): the same two tests, one match at a time, 64 positions per step
            for (unsigned long long lm = __builtin_amdgcn_ballot_w64(live && ka > 64); lm != 0ull; lm &= lm - 1ull) {
                const int m = (int)__builtin_ctzll(lm);
                const int ba_m = __builtin_amdgcn_readlane(ba, m), bb_m = __builtin_amdgcn_readlane(bb, m);
                const int ka_m = __builtin_amdgcn_readlane(ka, m), p_m = __builtin_amdgcn_readlane(p, m);
                const uint32_t tok_m = (uint32_t)__builtin_amdgcn_readlane((int)tok, m);
                bool bad = false, again = false;
                for (int j0 = 0; j0 < ka_m; j0 += 64) {
                    const int j = j0 + lane;
                    const bool in = j < ka_m;
                    const uint32_t xa = in ? indices[ba_m + j] : 0u, xb = in ? indices[bb_m + j + (j >= p_m ? 1 : 0)] : 0u;
                    bad |= in && xa != xb;
                    again |= in && j < p_m && xb == tok_m;
                }
                const bool ok = __builtin_amdgcn_ballot_w64(bad) == 0ull, ag = __builtin_amdgcn_ballot_w64(again) != 0ull;
                if (lane == m) {
                    cert = ok;
                    dup = ag;
                }
            }
            cert = cert && live && !dup;
            if (cert && !(ja.dbg & 32)) {
                uf_link(pa.parent, A, B);
                if (edges) record_edge(pa, edges, edge_cap, A, B);
            }
            const int nc = __popcll(__builtin_amdgcn_ballot_w64(cert));
            my_edges += (unsigned)nc;
            my_cands += (unsigned)nc;
            resolve(live && !cert && !dup, A, B, ba, bb, ka, kb);
            __builtin_amdgcn_wave_barrier();
            nm = 0;
        };

        auto drain = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (ja.dbg & 2) nq = 0;
            for (int q0 = 0; q0 < nq; q0 += 64) {
                const int i = q0 + lane;
                bool active = i < nq;
                const int4 qrec = active ? qr[i] : make_int4(0, 0, 0, 0);
                const uint32_t k2 = (uint32_t)qrec.y;
                const int qB = qrec.z, qP = qrec.w;
                uint32_t s = (uint32_t)qrec.x & ja.mask;
                int probes = 0;
                while (__builtin_amdgcn_ballot_w64(active) != 0ull) {
                    unsigned long long e0 = JOIN_EMPTY, e1 = JOIN_EMPTY;
                    if (active) {  // two slots per step: at load 1/4 nearly every chain ends inside them
                        e0 = ja.tab[s];
                        e1 = ja.tab[(s + 1) & ja.mask];
                    }
                    const bool h0 = active && e0 != JOIN_EMPTY && (uint32_t)(e0 >> 32) == k2;
                    const bool h1 = active && e0 != JOIN_EMPTY && e1 != JOIN_EMPTY && (uint32_t)(e1 >> 32) == k2;
                    if (active) {
                        if (e0 == JOIN_EMPTY || e1 == JOIN_EMPTY) active = false;
                        s = (s + 2) & ja.mask;
                        probes += 2;
                        if (probes > JOIN_MAX_PROBE) {
                            pa.ctr->join_fail = 1;
                            active = false;
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const bool mt = h ? h1 : h0;
                        const unsigned long long mm = __builtin_amdgcn_ballot_w64(mt);
                        if (mm == 0ull) continue;
                        if (nm + __popcll(mm) > MCAP) settle();
                        if (mt) {
                            const int pos = nm + __popcll(mm & lt);
                            mr[pos] = make_int4((int)(uint32_t)(h ? e1 : e0), qB, qP, 0);
                        }
                        nm += __popcll(mm);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            nq = 0;
        };

        // bitmap hits of one window -> queue
        auto push_hits = [&](bool hit, uint32_t k1, uint32_t k2, int B, int p) {
            unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
            if (ja.dbg & 4) hm = 0ull;
            if ((hm >> lane) & 1ull) {
                const int pos = nq + __popcll(hm & lt);
                qr[pos] = make_int4((int)k1, (int)k2, B, p);
            }
            nq += __popcll(hm);
        };

        const int r0 = ja.batch_row[gw];                      // the row that holds token T0
        const int extm = indptr[min(r0 + lane, n)];            // lane 0: start of row r0; lane l: end of row r0 + l - 1
        const uint2 rh = ja.rowhash[min(r0 + lane, n - 1)];    // lane l: H of row r0 + l
        const int Tend = min(T0 + JOIN_TPW, nnz);
        unsigned pend = 0;  // windows left to slow_window: the rows did not fit the 64 extents, or the queue was nearly full
        const bool covered = __builtin_amdgcn_readlane(extm, 63) >= Tend;
        int base = 0;  // boundaries (extm[1..]) already at or below the window start
#pragma unroll 1
        for (int ub = 0; ub < 8; ub += 4) {  // four windows in flight (eight do not fit 64 VGPRs = two blocks per CU)
            uint32_t xs[4], k1s[4], wv[4];  // xs: the token, then its first hash stage; wv: the bitmap word of the lookup
            int rp[4];                      // row offset from r0 (6 bits) | position in the row << 6
            if (T0 + 64 * ub >= Tend) break;
#pragma unroll
            for (int u = 0; u < 4; u++) xs[u] = indices[min(T0 + 64 * (ub + u) + lane, nnz - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j0 = T0 + 64 * (ub + u), j = j0 + lane;
                int ro = base, st = __builtin_amdgcn_readlane(extm, base);
                while (base < 63) {  // wave-uniform
                    const int e = __builtin_amdgcn_readlane(extm, base + 1);
                    if (e >= j0 + 64) break;
                    ro += j >= e ? 1 : 0;
                    st = j >= e ? e : st;
                    base++;
                }
                const uint32_t S1 = (uint32_t)__shfl((int)rh.x, ro);
                xs[u] = jh_stage(xs[u]);
                k1s[u] = S1 - jh1_of(xs[u]);
                rp[u] = ro | ((j - st) << 6);
                wv[u] = ja.bits[(covered && j < Tend && !(ja.dbg & 128)) ? ((k1s[u] & ja.bmask) >> 5) : 0u];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j0 = T0 + 64 * (ub + u), j = j0 + lane;
                if (j0 >= Tend) continue;  // wave-uniform
                if (!covered || nq > QCAP - 64) {
                    pend |= 1u << (ub + u);
                    continue;
                }
                const int ro = rp[u] & 63;
                const uint32_t S2 = (uint32_t)__shfl((int)rh.y, ro);
                push_hits(j < Tend && (wv[u] & join_bloom_mask(k1s[u])) == join_bloom_mask(k1s[u]), k1s[u], S2 - jh2_of(xs[u]), r0 + ro,
                          rp[u] >> 6);
            }
        }
        // the one drain site: after the fast windows, whenever a slow window could overfill the queue, at the end
        bool first = true, fin = false;
        while (true) {
            if (first || fin || nq > QCAP - 64) {
                drain();
                first = false;
                if (fin) break;
            }
            if (pend == 0u) {
                fin = true;
                continue;
            }
            // slow_window: every lane finds the row of its token by binary search over indptr
            const int u = (int)__builtin_ctz(pend);
            pend &= pend - 1u;
            const int j = T0 + 64 * u + lane;
            const bool valid = j < Tend;
            int lo = 0, hi = n;  // last row r with indptr[r] <= j (it holds token j: indptr[r + 1] > j)
            while (hi - lo > 1) {
                const int mid = lo + ((hi - lo) >> 1);
                if (indptr[mid] <= (valid ? j : 0)) lo = mid; else hi = mid;
            }
            const int B = lo;
            const uint2 hB = ja.rowhash[B];
            const uint32_t hs = jh_stage(indices[valid ? j : 0]);
            const uint32_t k1 = hB.x - jh1_of(hs);
            const uint32_t w = ja.bits[valid ? ((k1 & ja.bmask) >> 5) : 0u];
            push_hits(valid && (w & join_bloom_mask(k1)) == join_bloom_mask(k1), k1, hB.y - jh2_of(hs), B, j - indptr[B]);
        }
        settle();
    }
    if (lane == 0 && (my_edges | my_cands)) {
        atomicAdd(&s_edges, my_edges);
        atomicAdd(&s_cands, my_cands);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // plain stores, summed by the host
        ja.stats[2 * blockIdx.x] = (int)s_edges;
        ja.stats[2 * blockIdx.x + 1] = (int)s_cands;
    }
}

// ------------------------------------------------------------------------------------------------
// The prefix-group path (max_dist >= 2, large inputs): candidates from GROUPS instead of (k,f,g) bands.
//   Prefix filtering (the AllPairs / PPJoin idea of set-similarity joins, here for a symmetric-difference bound on
//   multisets): fix ANY total order on the elements — an element is (token, which occurrence of it in the row) — and
//   let the PREFIX of a row be its first max_dist + 1 elements in that order.  If two rows with at least
//   max_dist + 1 elements share no prefix element, then the max_dist + 1 prefix elements of the row whose prefix ends
//   first are all missing from the other row (an element of the other row that early would be in ITS prefix), so the
//   distance exceeds max_dist.  Rows within max_dist therefore share a prefix element — unless one of them has
//   max_dist elements or fewer; then both have at most 2*max_dist, and every such row also carries the SHORT record.
//   The order puts rare tokens first (a sampled token count, then the higher token id: vocabulary ids are handed out by
//   first appearance, so higher = newer = rarer), which makes the groups — rows that share a prefix element — small:
//   on the configs[4] generator 1.4e7 pairs inside groups at 100k rows, max_dist 5 (the length band has 3.1e9,
//   grouping by hash classes of the tokens 3.7e8), 144 per row, 178 at 300k.  The order affects only the work, never
//   the result.
//   k_pgfreq counts the tokens of a sample of the rows; k_pgkeys leaves max_dist + 2 records per row (prefix elements,
//   SHORT or unique sentinels : row, slot); the LSD radix sort of bfk_sort.hip (hand-written) orders them; k_pgplace lays {row, length,
//   64-bit signature} out in that order and notes where every record went; k_pgjoin walks, row by row, the groups of
//   the row's records and queues every member that passes the second level, once (see there).  Exact: sharing a prefix
//   element is a necessary condition, so is the signature level, the verify is exact, and every pair is queued once.
// ------------------------------------------------------------------------------------------------
// Record keys are 32 bits: token + 1 for a prefix element, 0 for the SHORT record, PG_NONE for "no such record".
// Elements are DISTINCT tokens: a repeated token counts once (capping multiplicities is a contraction of the distance —
// |min(a,1) - min(b,1)| <= |a - b| per token — so rows within max_dist stay within max_dist and the filter stays valid; a
// row's records then have distinct keys).  The radix sort is stable and the records are written row by row: a group lists
// its rows ascending, a row looks at the members BEHIND it, and every unordered pair is seen from its smaller row.
constexpr uint32_t PG_NONE = 0xFFFFFFFFu;

// counter of a token: the token id itself while the vocabulary fits the table (ids are handed out by first appearance: the
// common tokens are the low ids and share a few cache lines — hashed, the 70k tokens of the benchmark vocabulary were 70k
// different lines of a 4 MB table), a multiplicative hash beyond
__device__ __forceinline__ uint32_t pg_cnt_slot(uint32_t t, int dense) { return dense ? t : (t * 0x9E3779B1u) >> (32 - PG_CNT_BITS); }

// bind-time helper of the path: the largest token id (sets the number of key bits to sort)
__global__ __launch_bounds__(256) void k_maxtok(const uint32_t *__restrict__ indices, int nnz, int *out) {
    __shared__ int s_m[4];
    int m = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nnz; i += gridDim.x * 256) m = max(m, (int)(indices[i] & 0x7FFFFFFFu));
    for (int s = 32; s > 0; s >>= 1) m = max(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per block (one per wave of 4096 blocks was 190 us of same-address atomics, whatever the input size)
    if (threadIdx.x == 0) {
        m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m > 0) atomicMax(out, m);
    }
}

// Token counts over a sample of the rows (every stride-th row), 16 lanes per row, 64 rows per block.  A token that nearly every
// row carries would take thousands of adds to one address, one after the other (96 us of this kernel with one global add per
// occurrence): the block first counts in LDS — a direct-mapped table of counter slots, a slot that is taken by another counter
// sends the add to memory — and then adds what it holds, once per counter.  The totals are exact whatever the order.
__global__ __launch_bounds__(256) void k_pgfreq(const int *__restrict__ indptr, const uint32_t *__restrict__ indices, int n, int stride,
                                                uint32_t *__restrict__ cnt, Counters *ctr, int dense) {
    constexpr int LT = 2048, ROWS = 4;  // LDS slots; rows per 16-lane group
    __shared__ int l_tag[LT];
    __shared__ unsigned l_cnt[LT];
    if (blockIdx.x == 0) {  // the step's counters (the band kernels' k_cells does this on their path)
        for (int i = threadIdx.x; i < CAND_SHARDS; i += 256) ctr->ncand[i] = 0;
        if (threadIdx.x == 0) {
            ctr->pg_est = 0ull;
            ctr->pg_fail = 0;
            ctr->err = 0;
            ctr->overflow = 0;
            ctr->n_edges = ctr->n_cand_total = ctr->n_edges_cap = ctr->n_connected = 0;
            ctr->pairs_in_band = 0ull;
        }
    }
    for (int i = threadIdx.x; i < LT; i += 256) {
        l_tag[i] = -1;
        l_cnt[i] = 0u;
    }
    __syncthreads();
    const int l16 = threadIdx.x & 15;
    for (int q = 0; q < ROWS; q++) {
        const long long r = (long long)((blockIdx.x * ROWS + q) * 16 + (threadIdx.x >> 4)) * stride;
        if (r >= n) continue;
        for (int j = indptr[r] + l16, e = indptr[r + 1]; j < e; j += 16) {
            const int slot = (int)pg_cnt_slot(indices[j] & 0x7FFFFFFFu, dense);
            const int i = slot & (LT - 1);
            const int old = atomicCAS(&l_tag[i], -1, slot);
            if (old == -1 || old == slot) atomicAdd(&l_cnt[i], 1u);
            else atomicAdd(&cnt[slot], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LT; i += 256)
        if (l_tag[i] >= 0) atomicAdd(&cnt[l_tag[i]], l_cnt[i]);
}

// 16 lanes per row: the row's max_dist + 1 first DISTINCT tokens in the global order (sampled count, then the higher token
// id: vocabulary ids are handed out by first appearance, higher = newer = rarer), by repeated minimum over the group —
// the tokens are read once, coalesced, 64 at a time; lane i of the group ends up with the i-th element.  The order value
// of a token is 32 bits: (sampled count, saturated) above (2^tb - 2 - token), tb = bits of the largest token id + 2.
__global__ __launch_bounds__(256) void k_pgkeys(const int *__restrict__ indptr, const uint32_t *__restrict__ indices, int n, int recs,
                                                int max_dist, int tb, const uint32_t *__restrict__ cnt, uint32_t *__restrict__ keys,
                                                uint32_t *__restrict__ keys_pm, int *__restrict__ rows_pm, int pm, int kcap,
                                                int *__restrict__ parent, int4 *__restrict__ rowinfo, Counters *ctr, int dense, int has_short) {
    // A group takes PGK_ROWS consecutive rows and has the token loads of all of them in flight, then the count look-ups of all
    // of them: with one row per group a wave was three dependent loads and gone (190 us at 1M rows for 300 MB of traffic).
    constexpr int R = PGK_ROWS;
    const int r0 = (blockIdx.x * 16 + (threadIdx.x >> 4)) * R, l16 = threadIdx.x & 15;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctr->n_work = (unsigned)((n + 63) >> 6);  // work items of k_pgjoin: blocks of 64 rows (k_cells left its tile count here)
        ctr->pairs_filtered = 0ull;
    }
    // (no early return: the groups of a wave run the DPP reductions together)
    int b[R], e[R], c0[R];
    uint32_t sel[R], s2all[R], s2hi[R];  // lane i < pre: the i-th element so far; the row's second-level signature (XOR parity
                                         // of one of 64 bits per token, as k_sig's)
#pragma unroll
    for (int q = 0; q < R; q++) {
        const bool live = r0 + q < n;
        b[q] = live ? indptr[r0 + q] : 0;
        e[q] = live ? indptr[r0 + q + 1] : 0;
        c0[q] = b[q];
        sel[q] = 0xFFFFFFFFu;
        s2all[q] = s2hi[q] = 0u;
    }
    const uint32_t cmax = (1u << (32 - tb)) - 1u, tinv = (1u << tb) - 2u;
    const int pre = recs - 1;  // = max_dist + 1
    for (;;) {
        bool any = false;
#pragma unroll
        for (int q = 0; q < R; q++) any = any || c0[q] < e[q];
        if (__builtin_amdgcn_ballot_w64(any) == 0ull) break;  // wave-uniform: until every group of the wave is through its rows
        uint32_t x[R][4], v[R][4];
#pragma unroll
        for (int q = 0; q < R; q++)
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const int j = c0[q] + st * 16 + l16;
                x[q][st] = j < e[q] ? indices[j] : 0xFFFFFFFFu;
            }
#pragma unroll
        for (int q = 0; q < R; q++)
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const bool in = c0[q] + st * 16 + l16 < e[q];
                v[q][st] = in ? cnt[pg_cnt_slot(x[q][st] & 0x7FFFFFFFu, dense)] : 0u;
            }
#pragma unroll
        for (int q = 0; q < R; q++) {
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const bool in = c0[q] + st * 16 + l16 < e[q];
                const uint32_t t = x[q][st] & 0x7FFFFFFFu;
                v[q][st] = in ? (min(v[q][st], cmax) << tb) | (tinv - t) : 0xFFFFFFFFu;
                if (in) {
                    const uint32_t h2 = sig_h2(sig_h1(x[q][st]));
                    const uint32_t bit2 = 1u << ((h2 >> 24) & 31);
                    s2all[q] ^= bit2;
                    s2hi[q] ^= (uint32_t)((int)(h2 << 2) >> 31) & bit2;  // bit 29 selects the word
                }
            }
            if (__builtin_amdgcn_ballot_w64(c0[q] < e[q]) == 0ull) continue;  // (wave-uniform: nobody has a chunk of row q left)
            uint32_t carry = sel[q], nsel = 0xFFFFFFFFu;
            for (int i = 0; i < pre; i++) {
                const uint32_t m = row16_allmin_u(min(min(min(v[q][0], v[q][1]), min(v[q][2], v[q][3])), carry));
#pragma unroll
                for (int st = 0; st < 4; st++) v[q][st] = v[q][st] == m ? 0xFFFFFFFFu : v[q][st];  // every copy of the token leaves
                carry = carry == m ? 0xFFFFFFFFu : carry;
                if (l16 == i) nsel = m;
            }
            sel[q] = nsel;
            c0[q] += 64;
        }
    }
#pragma unroll
    for (int q = 0; q < R; q++) {
        const int r = r0 + q;
        // {length, second-level signature, offset} of the row in one 16-byte record: what k_pgplace gathers per position and
        // the walk per row
        const uint32_t sa = row16_allxor(s2all[q]), sh = row16_allxor(s2hi[q]);
        if (r >= n) continue;
        const int len = e[q] - b[q];
        if (l16 == 15) {
            rowinfo[r] = make_int4(len, (int)(sa ^ sh), (int)sh, b[q]);
            parent[r] = r;
            if (len < 0 || len > kcap) atomicOr(&ctr->err_rows, ERR_ROWLEN);  // the CSR changed after the bind
        }
        // The records go out twice: row-major (keys[row][slot]: what k_pgjoin reads row by row) and POSITION-major
        // (keys_pm[slot][row] with their (row, slot) values: the sort's input).  The sort is stable, so the group of a token comes
        // out as its records of slot 0 (rows ascending), then slot 1, ... — the order the positional filter of k_pgplace needs —
        // without a single key bit spent on the position.
        // (keys == NULL: the walk of a labels-only step — k_pgwalk16<false> — never looks at the row-major records: 28 MB of
        // 28-byte pieces at 1M rows, max_dist 5.  rows_pm == NULL: the value of a record is its position's (row, slot), which
        // the sort's first pass works out from the position itself instead of loading it.)
        if (l16 < recs) {
            uint32_t key;
            if (l16 < pre) key = sel[q] != 0xFFFFFFFFu ? tinv - (sel[q] & ((1u << tb) - 1u)) + 1u : PG_NONE;
            // a row of max_dist elements or fewer can be within max_dist of a row it shares nothing with; then both have at most
            // 2 * max_dist elements: all of those meet in the group of the SHORT record (key 0, first in the order)
            else key = len <= 2 * max_dist ? 0u : PG_NONE;
            if (keys) keys[(size_t)r * recs + l16] = key;
            // (pm = 0 — the positional filter is off: token ids that leave no room for the composite key — keeps the records in
            // row order, so that a group comes out with its rows ascending as the whole-group walk needs it)
            const size_t o = pm ? (size_t)l16 * n + r : (size_t)r * recs + l16;
            // (has_short = 0: the SHORT slot — the last n records of the position-major input — is not sorted at all)
            if (has_short || l16 < pre) {
                keys_pm[o] = key;
                if (rows_pm) rows_pm[o] = r * recs + l16;  // the sort carries (row, slot)
            }
        }
    }
}

// can the record in this slot of a row walk at all?  (slot recs - 1: the SHORT record; slot i < recs - 1: prefix position i, which
// with the positional filter walks sub-groups i .. d - i: none when d - i < i)
// has_short = 0: no row of the CSR is short enough for a SHORT record — those records (all "no such record") are then not part
// of the sorted order at all (a (d + 2)-th of the sort and of k_pgplace, a quarter at max_dist 2), and nobody walks from them
__device__ __forceinline__ bool pg_slot_walks(int slot, int recs, int pb, int max_dist, int has_short) {
    return slot == recs - 1 ? has_short != 0 : (!pb || max_dist - slot >= slot);
}

// one thread per position of the sorted records: for every record of a row where it went and how many positions behind it
// the row has to walk.
//
// POSITIONAL FILTER (round 3).  Let t* be the first element (in the global order) that rows A and B within max_dist = d share,
// at position i of A's prefix and j of B's (from 0).  The i elements of A in front of t* are not in B and the j elements of B
// in front of it are not in A, so i + j <= |A delta B| <= d.  Keys are {token : position}, so the group of a token lies in the
// sorted order as its sub-groups of position 0, 1, .. d one after the other, rows ascending inside each.  A pair is looked at
// from the row whose position is smaller (from the smaller row if they are equal): the record of row A at position i walks
// the rest of its own sub-group (the rows behind A) and then the sub-groups of positions i + 1 .. d - i WHOLE — ONE contiguous
// range of the sorted order, [p + 1, end of sub-group d - i) — and nothing at all when d - i < i.  The sub-groups that hold a
// token late in BOTH rows' prefixes — the big ones: a token that is late in a prefix is a common one — are never walked.
// Every pair within d is still met at t* by exactly one row; a pair can now also be met a second time from the other row
// at a later shared token, which a labels-only step lets through (the verify drops it as connected) and an exact-edges step
// decides by the records test for every candidate (k_pgjoin).
__global__ __launch_bounds__(256) void k_pgplace(const uint32_t *__restrict__ keys_s /*composite {key : slot}: the sort's last pass*/,
                                                 const int *__restrict__ vals_s, int total, int recs, const int4 *__restrict__ rowinfo,
                                                 int4 *__restrict__ srec, int2 *__restrict__ recpos, Counters *ctr, int pb, int max_dist, int has_short) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const uint32_t key = keys_s[p];
    const int v = vals_s[p];
    // the position's {row, length, second-level signature} in group order: the walk reads the members of a group as one
    // coalesced stream (the gather is in flight while the record looks for the end of its walk)
    const int row = v / recs;
    const int4 ri = rowinfo[row];
    int behind = 0;
    if (key != PG_NONE) {  // ("no such record": nothing behind it — the walk reads the counts without looking at the keys)
        // the last key this record walks to: its own (SHORT record, or no position bits), or {token : d - i}
        uint32_t target = key;
        bool walks = true;
        if (pb && key != 0u) {
            const int i = (int)(key & ((1u << pb) - 1u));
            walks = max_dist - i >= i;
            target = (key & ~((1u << pb) - 1u)) | (uint32_t)max(max_dist - i, 0);
        }
        if (walks) {  // first position whose key is above the target: gallop from p, then bisect (keys_s[p] <= target)
            int lo = p, step = 1;
            while (lo + step < total && keys_s[lo + step] <= target) {
                lo += step;
                step <<= 1;
            }
            int hi = min(total, lo + step);
            while (hi - lo > 1) {
                const int mid = lo + ((hi - lo) >> 1);
                if (keys_s[mid] <= target) lo = mid;
                else hi = mid;
            }
            behind = lo - p;
        }
    }
    srec[p] = make_int4(row, ri.x, ri.y, ri.z);
    // (7M scattered 8-byte stores were 88 us of this kernel at 1M rows, max_dist 5: the records of positions i > d - i never
    // walk — pg_slot_walks, the walk does not read their entries — and are not stored)
    const int slot = v - row * recs;
    if (pg_slot_walks(slot, recs, pb, max_dist, has_short)) recpos[v] = make_int2(p, behind);
    // every PG_EST_STRIDE-th position reports what k_pgjoin will walk from it, so that the walk can be called off when the
    // groups are too big
    if ((p & (PG_EST_STRIDE - 1)) == 0 && behind > 0) atomicAdd(&ctr->pg_est, (unsigned long long)behind);
}

// k_pgjoin: candidates of the prefix-group path, row by row.  One wave per row A: its records in the global order (the SHORT
// record first, then the prefix elements), and for each the members behind A's record in that record's group.  A member B
// passes the second level (length difference and 64-bit signature distance within max_dist) or is dropped; a pair must be
// queued once although its rows share several prefix elements: the FIRST element they share is the first of A's records in
// whose group B shows up, so "seen in an earlier group of A" is the whole test — a hash set in the wave's LDS, holding only
// members that passed (a row has ~16 of those at max_dist 5).  A set that fills up (a hub row with thousands of neighbours)
// stops taking entries; from then on a member that is not in it is checked the slow, equally exact way: is one of A's
// earlier elements among B's records.
// The members behind the row's records are a list of chunks of up to 64 positions that is known before the first load
// (k_pgplace counted them): two chunks are in flight while a third is worked on, and the head of the NEXT row (its records'
// positions and counts, its length and signature) is requested before the current row is walked.
// Work items are blocks of 64 rows (t_begin / t_end and the multi-GPU owner rule count in those).
__global__ __launch_bounds__(256) void k_pgjoin(const int4 *__restrict__ srec, const int2 *__restrict__ recpos,
                                                const uint32_t *__restrict__ keys, const int4 *__restrict__ rowinfo, int n, int recs,
                                                int total, int shard0, int nshards, int t_begin, int t_end, PairArgs pa, int pb, int has_short) {
    constexpr int SCAP = 1024, WAVES = 4;
    // A token that many rows carry can still be among a row's first d + 1 (small alphabets, random rows): the groups are then a
    // large part of all rows and walking them is quadratic.  k_pgplace reported the walk from every PG_EST_STRIDE-th position
    // of the sorted records: beyond PG_GIVE_UP members per row the kernel does nothing and the host redoes the step on the
    // band kernels (and keeps this CSR there).
    if (pa.ctr->pg_est * (unsigned long long)PG_EST_STRIDE > (unsigned long long)PG_GIVE_UP * (unsigned long long)n) {
        if (blockIdx.x == 0 && threadIdx.x == 0) pa.ctr->pg_fail = 1;
        return;
    }
    __shared__ int s_set[WAVES][SCAP];
    __shared__ int4 s_q[WAVES][64];  // fresh pairs {A, B, k_A, k_B} waiting for their queue slots
    __shared__ int2 s_pend[WAVES][128];  // members {row, length} that passed the second level, waiting to be settled
    __shared__ unsigned char s_pstep[WAVES][128];  // ... and the step of the group they were met in (25 KiB of LDS in all)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto below = [](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
    int *set = s_set[wave];
    int4 *sq = s_q[wave];
    int2 *pend = s_pend[wave];
    unsigned char *pstep = s_pstep[wave];
    for (int i = lane; i < SCAP; i += 64) set[i] = -1;
    __syncthreads();
    int cshard = (blockIdx.x * WAVES + wave) & (CAND_SHARDS - 1);
    unsigned long long visits = 0;
    const int d = pa.d;
    int nq = 0;  // pairs waiting in sq (wave-uniform)
    // The wave's pairs go to the queue 64 at a time: one returning atomic per 64 pairs instead of one per chunk that found
    // any (a round trip to the memory side each), and the queue records leave as full lines.
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        cshard = (cshard + 1) & (CAND_SHARDS - 1);
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(&pa.ctr->ncand[cshard], (unsigned)nq);
        int4 e = make_int4(0, 0, 0, 0);
        int ba = 0, bb = 0;
        if (lane < nq) {
            e = sq[lane];
            ba = pa.indptr[e.x];
            bb = pa.indptr[e.y];
        }
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < nq) {
            const int idx = base + lane;
            if (idx < pa.cand_cap_shard) {
                const size_t o = (size_t)cshard * pa.cand_cap_shard + idx;
                pa.cand[o] = make_int4(e.x, e.y, ba, bb);
                pa.candk[o] = make_int2(e.z, e.w);
            } else {
                pa.ctr->overflow = 1;  // dropped: the host re-runs the row range in smaller slices
            }
        }
        __builtin_amdgcn_wave_barrier();
        nq = 0;
    };
    const int r_begin = t_begin * 64, r_end = min(n, (int)min((long long)t_end * 64, (long long)n));
    const int stride = (int)gridDim.x * WAVES;
    auto owned_from = [&](int A) {  // the first row at or behind A (in steps of the wave's stride) that is this rank's
        while (nshards > 1 && A < r_end && ((A >> 6) % nshards) != shard0) A += stride;
        return A;
    };
    struct RowHead {
        uint32_t key;  // lane i < recs: record i of the row
        int2 pos;      // ... where it is in the group order, members behind it (stale for a record that does not exist)
        int len;
        uint32_t s0, s1;
    };
    auto load_head = [&](int A) {
        RowHead h;
        h.key = PG_NONE;
        h.pos = make_int2(0, 0);
        if (lane < recs) {
            h.key = keys[(size_t)A * recs + lane];
            h.pos = recpos[(size_t)A * recs + lane];
        }
        const int4 ri = rowinfo[A];
        h.len = ri.x;
        h.s0 = (uint32_t)ri.y;
        h.s1 = (uint32_t)ri.z;
        return h;
    };
    int A = owned_from(r_begin + blockIdx.x * WAVES + wave);
    RowHead cur{PG_NONE, make_int2(0, 0), 0, 0u, 0u};
    if (A < r_end) cur = load_head(A);
    for (; A < r_end;) {
        const int A_next = owned_from(A + stride);
        RowHead nxt{PG_NONE, make_int2(0, 0), 0, 0u, 0u};
        if (A_next < r_end) nxt = load_head(A_next);
        const int A_now = A;
        const RowHead hd = cur;
        cur = nxt;
        A = A_next;
        const int len_a = hd.len;
        const uint32_t sa0 = hd.s0, sa1 = hd.s1;
        // The members behind the row's records, group after group in the order of the STEPS (step 0 = the SHORT record in slot
        // recs - 1, step i = slot i - 1), are ONE list: lane l of chunk c takes entry 64 c + l of it, whichever group that
        // falls into.  Half of the records have 8 members or fewer behind them — a chunk per group was a third full — and
        // nothing below depends on the groups being worked off one after the other: of the lanes that find one row B in
        // several groups, in one chunk or not, exactly one gets its insert into the set through (and with a full set, exactly
        // the lane of the first group they share passes the test on the records).
        const int slot_j = lane == 0 ? recs - 1 : lane - 1;
        const uint32_t key_j = __shfl(hd.key, slot_j);
        const int pos_j = __shfl(hd.pos.x, slot_j);
        const int beh_all = __shfl(hd.pos.y, slot_j);  // (unconditional: a lane that sits this out cannot be read from)
        const int beh_j = (lane < recs && key_j != PG_NONE && pg_slot_walks(slot_j, recs, pb, d, has_short)) ? beh_all : 0;  // (other slots: no entry)
        const int incl_j = wave_incl_scan_add(beh_j);
        const int T = __builtin_amdgcn_readlane(incl_j, 63);
        if (T == 0) continue;
        const int base_j = pos_j + 1 - (incl_j - beh_j);  // entry f of the list, in group j, is position base_j + f
        int I[PG_MAX_DIST + 1];  // (scalars; lanes at and beyond recs repeat the total)
#pragma unroll
        for (int j = 0; j < PG_MAX_DIST + 1; j++) I[j] = __builtin_amdgcn_readlane(incl_j, j);
        auto entry = [&](int f, int &step) {  // position of entry f of the list, and the step of its group
            int st = 0;
#pragma unroll
            for (int j = 0; j < PG_MAX_DIST + 1; j++) st += f >= I[j] ? 1 : 0;
            st = min(st, recs - 1);
            step = st;
            return __shfl(base_j, st) + f;  // (ds_bpermute: the base of group st sits in lane st)
        };
        int n_in = 0;         // entries of the set (wave-uniform)
        bool full = false;    // the set stopped taking entries
        int np = 0;           // members that passed the second level and wait in `pend` (wave-uniform, < 64 between chunks)
        // The chunk loop only TESTS (length, signature) and parks what passes — ~7% of the members — in the wave's LDS; the
        // set, the slow test and the queue see them 64 at a time, one per lane (`settle`).  Doing that inside the loop meant
        // ~100 wave-instructions per chunk for the 4 lanes of 64 that had something to do: the walk is bound by its
        // instruction stream, not by its loads (1.44 -> see DESIGN 6d for the measured step).
        auto settle = [&](int cnt) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool have = lane < cnt;
            const int2 pe = have ? pend[lane] : make_int2(0, 0);
            const int B = pe.x, len_b = pe.y;
            const int step = have ? (int)pstep[lane] : 0;
            // seen in another group of A?
            bool fresh = false, unknown = false;
            if (pa.dbg & 256) {  // (BFK_PF_DEBUG=256: no set, every member that passes is queued; timing experiments)
                fresh = have;
            } else if (have) {
                // (the set by plain LDS reads and writes of {row : lane} slots — read, write if empty, read back: in order per
                // wave, the last writer stays — instead of the compare-and-swap: 2.81 ms against 2.33 for this kernel at 1M
                // rows, max_dist 5, 1.30 against 0.53 at 100k rows: three dependent LDS round trips per probe instead of one)
                uint32_t h = ((uint32_t)B * 0x9E3779B1u) >> 22;  // SCAP = 1024 slots
                for (;;) {
                    const int old = full ? set[h] : atomicCAS(&set[h], -1, B);
                    if (old == B) break;                     // seen: a duplicate
                    if (old == -1) {
                        // (just inserted; or the set takes no more entries: a labels-only step lets the pair through — the
                        // verify drops a second copy as connected — the others decide it the slow way)
                        // (with position bits a pair can also be met from its OTHER row, at a later shared token: an exact-edges
                        // step decides every candidate by the records)
                        fresh = (!full && !(pb && !pa.skip_connected)) || pa.skip_connected;
                        unknown = (full || pb) && !pa.skip_connected;
                        break;
                    }
                    h = (h + 1) & (SCAP - 1);
                }
            }
            if (__builtin_amdgcn_ballot_w64(unknown) != 0ull) {
                // the slow exact test: one of A's earlier elements among B's records
                bool dup = false;
                if (unknown) {
                    const uint32_t *kb = keys + (size_t)B * recs;
                    uint32_t gb[PG_MAX_DIST + 2];
#pragma unroll
                    for (int j = 0; j < PG_MAX_DIST + 2; j++) gb[j] = j < recs ? kb[j] : PG_NONE;
                    for (int e2 = 0; e2 < step; e2++) {
                        const uint32_t y = keys[(size_t)A_now * recs + (e2 == 0 ? recs - 1 : e2 - 1)];
                        if (y == PG_NONE) continue;
#pragma unroll
                        for (int j = 0; j < PG_MAX_DIST + 2; j++) dup = dup || gb[j] == y;
                    }
                }
                fresh = fresh || (unknown && !dup);
            }
            n_in += __popcll(__builtin_amdgcn_ballot_w64(fresh && !full));
            // (wave-uniform; entries made so far stay valid.)  Half full, not three quarters: the rows that get there are the
            // hubs — an early profile with hundreds of descendants within max_dist — and at 75% load the linear probe
            // chains of those rows were most of this kernel: 2.27 -> 1.44 ms at 1M rows, max_dist 5, 0.49 -> 0.23 at 100k
            // (a set of 2048 slots at 3/4: 1.92 / 0.24 — the LDS it takes costs more waves than the shorter chains give)
            if (n_in > SCAP / 2) full = true;
            // the fresh pairs wait in LDS for their queue slots
            const unsigned long long fm = __builtin_amdgcn_ballot_w64(fresh);
            if (fm != 0ull) {
                const int nf = __popcll(fm);
                if (nq + nf > 64) flush();
                if (fresh) sq[nq + below(fm)] = make_int4(A_now, B, len_a, len_b);
                nq += nf;
                if (pa.dbg & 128) nq = 0;  // (BFK_PF_DEBUG=128: nothing is queued; timing experiments, results invalid)
            }
        };
        const int nch = (T + 63) >> 6;
        int st0 = 0, st1 = 0, st2 = 0;
        int4 rec0 = make_int4(0, 0, 0, 0), rec1 = rec0;
        {
            const int q0 = entry(lane, st0), q1 = entry(64 + lane, st1);
            if (lane < T) rec0 = srec[q0];
            if (64 + lane < T) rec1 = srec[q1];
        }
        for (int c = 0; c < nch; c++) {
            int4 rec2 = make_int4(0, 0, 0, 0);
            {
                const int f2 = (c + 2) * 64 + lane;
                const int q2 = entry(f2, st2);
                if (f2 < T) rec2 = srec[q2];
            }
            {
                const int4 rec = rec0;
                const bool inb = c * 64 + lane < T;
                visits += (unsigned long long)min(64, T - c * 64);
                bool pass = inb && abs(rec.y - len_a) <= d && __popc((uint32_t)rec.z ^ sa0) + __popc((uint32_t)rec.w ^ sa1) <= d;
                if (pa.dbg & 64) pass = false;  // (BFK_PF_DEBUG=64: the walk alone; timing experiments, results invalid)
                const unsigned long long pm = __builtin_amdgcn_ballot_w64(pass);
                if (pm != 0ull) {
                    if (pass) {
                        const int i = np + below(pm);
                        pend[i] = make_int2(rec.x, rec.y);
                        pstep[i] = (unsigned char)st0;
                    }
                    np += __popcll(pm);
                    if (np >= 64) {
                        settle(64);
                        // what is left moves to the front
                        const int left = np - 64;
                        __builtin_amdgcn_wave_barrier();
                        const int2 mv = lane < left ? pend[64 + lane] : make_int2(0, 0);
                        const unsigned char ms = lane < left ? pstep[64 + lane] : (unsigned char)0;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (lane < left) {
                            pend[lane] = mv;
                            pstep[lane] = ms;
                        }
                        np = left;
                    }
                }
            }
            rec0 = rec1;
            st0 = st1;
            rec1 = rec2;
            st1 = st2;
        }
        if (np > 0) settle(np);  // (the set is this row's: nothing waits across rows)
        // clean the set for the next row
        if (n_in > 0 || full)
            for (int i = lane; i < SCAP; i += 64) set[i] = -1;
    }
    if (nq > 0) flush();
    __syncthreads();  // every wave is done with its set: the first one's holds the block's visit counts now
    if (lane == 0) {
        s_set[0][2 * wave] = (int)(unsigned)visits;
        s_set[0][2 * wave + 1] = (int)(unsigned)(visits >> 32);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0ull;
        for (int w = 0; w < WAVES; w++) sum += (unsigned long long)(unsigned)s_set[0][2 * w] | ((unsigned long long)(unsigned)s_set[0][2 * w + 1] << 32);
        if (sum) atomicAdd(&pa.ctr->pairs_filtered, sum);
    }
}

// k_pgwalk16: the walk of the prefix groups, 16 lanes per row (four rows per wave).  k_pgjoin gives a row a whole wave: with
// the positional filter a row has ~110 members behind its records — less than two chunks of 64 — and the set-up of the list,
// the clean-up of the de-duplication set and the launch of a wave are paid per row.  Two things bound that kernel at 1M rows
// (0.91 ms): 62 500 blocks — the grid that gives every wave one row, for balance — cost 0.5 ms just to be dispatched (the same
// walk without queueing: 0.80 ms on that grid, 0.31 on 8192 blocks), and the queue records took two scattered loads each (the
// rows' offsets in the CSR).  Here a 16-lane group owns a row: the set-up is four DPP steps inside the group, the steps' ends
// and bases sit in LDS, a chunk is 16 members with four chunks in flight, and a group that has finished its row takes its
// next one while the other three go on (no wave-wide row boundary), so a grid of 32 blocks per CU balances.  A's offset
// comes with the row's head; B's is left to the verify, which needs it for the ~6% of the records that are not dropped as
// connected (`w` < 0 in the queue record).  The de-duplication set is 64 slots per group and stops taking entries at 32: what
// it misses is queued twice and the verify drops the second copy as connected.  EXACT steps (max_dist 2, exact_edges, edge
// capture) decide every member that passes by the records test instead — is one of A's earlier elements among B's records —
// and write both offsets, so that every pair is queued exactly once and k_verify finds its records as from any other
// generator.  (k_pgjoin, a wave per row, stays behind BFK_PG_WALK16=0 for comparison.)
__device__ __forceinline__ int row16_incl_scan(int x) {  // inclusive prefix sum inside each 16-lane row (DPP row_shr)
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    return x;
}

template <bool EXACT>
__global__ __launch_bounds__(256) void k_pgwalk16(const int4 *__restrict__ srec, const int2 *__restrict__ recpos,
                                                  const int4 *__restrict__ rowinfo, const uint32_t *__restrict__ keys, int n, int recs,
                                                  int shard0, int nshards, int t_begin, int t_end, PairArgs pa, int pb, int has_short) {
    constexpr int WAVES = 4, GSET = 64, U = 4;
    if (pa.ctr->pg_est * (unsigned long long)PG_EST_STRIDE > (unsigned long long)PG_GIVE_UP * (unsigned long long)n) {
        if (blockIdx.x == 0 && threadIdx.x == 0) pa.ctr->pg_fail = 1;  // (groups too big to pay: the host redoes the step on the band kernels)
        return;
    }
    __shared__ int s_end[WAVES][4][16], s_base[WAVES][4][16];  // per group: inclusive end of every step in the row's list, base position
    __shared__ int s_set[WAVES][4][GSET];                       // per group: members of this row that were queued (labels-only steps)
    __shared__ uint32_t s_key[WAVES][4][16];                    // per group: the row's record keys, step by step (exact steps)
    __shared__ int4 s_q[WAVES][128];                            // pairs {A, B, offset of A, -1} waiting for their queue slots
    __shared__ int2 s_qk[WAVES][128];                           // ... and {k_A, k_B}
    __shared__ unsigned long long s_vis[WAVES];
    const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto below = [](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
    int *g_end = s_end[wave][grp], *g_base = s_base[wave][grp], *g_set = s_set[wave][grp];
    uint32_t *g_key = s_key[wave][grp];
    int4 *sq = s_q[wave];
    int2 *sqk = s_qk[wave];
#pragma unroll
    for (int j = 0; j < GSET / 16; j++) g_set[l16 + 16 * j] = -1;
    int cshard = (blockIdx.x * WAVES + wave) & (CAND_SHARDS - 1);
    const int d = pa.d;
    int nq = 0;  // pairs waiting in sq (wave-uniform)
    unsigned int my_visits = 0;  // members this lane looked at (summed over the wave at the end: no ballot per chunk)
    auto flush64 = [&]() {  // the first 64 waiting pairs go to the queue (full lines, one returning atomic), the rest moves to the front
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int cnt = min(nq, 64);
        cshard = (cshard + 1) & (CAND_SHARDS - 1);
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(&pa.ctr->ncand[cshard], (unsigned)cnt);
        int4 e = make_int4(0, 0, 0, 0), mv = e;
        int2 ek = make_int2(0, 0), mk = ek;
        if (lane < cnt) {
            e = sq[lane];
            ek = sqk[lane];
            if (EXACT) e.w = pa.indptr[e.y];  // (k_verify reads both offsets from the record)
        }
        if (64 + lane < nq) {
            mv = sq[64 + lane];
            mk = sqk[64 + lane];
        }
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < cnt) {
            const int idx = base + lane;
            if (idx < pa.cand_cap_shard) {
                const size_t o = (size_t)cshard * pa.cand_cap_shard + idx;
                pa.cand[o] = e;
                pa.candk[o] = ek;
            } else {
                pa.ctr->overflow = 1;  // dropped: the host re-runs the row range in smaller slices
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (64 + lane < nq) {
            sq[lane] = mv;
            sqk[lane] = mk;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        nq -= cnt;
    };
    // The rank's rows: the 64-row blocks b with b % nshards == shard0 inside [t_begin, t_end), numbered through — place v is
    // row 64 * (b0 + (v / 64) * nshards) + v % 64 — so that the rank's share is spread over all groups of the grid (a group
    // that skipped foreign rows in steps of the stride found either all of its rows or none: an 8th of the groups did the
    // work of the rank, each as long as before)
    const int blk_hi = (int)min((long long)t_end, ((long long)n + 63) >> 6);
    const int b0 = t_begin + ((shard0 - t_begin % nshards) % nshards + nshards) % nshards;
    const int v_end = b0 < blk_hi ? (blk_hi - b0 + nshards - 1) / nshards * 64 : 0;
    const int stride = (int)gridDim.x * WAVES * 4;
    auto row_of = [&](int v) { return ((b0 + (v >> 6) * nshards) << 6) | (v & 63); };
    auto place_from = [&](int v) {  // (only the input's last block has places without a row)
        while (v < v_end && row_of(v) >= n) v += stride;
        return v;
    };
    int at = place_from((blockIdx.x * WAVES + wave) * 4 + grp);  // (group-uniform)
    int A = at < v_end ? row_of(at) : 0;
    bool active = at < v_end, fresh_row = active;
    // the head of the group's next row {its records' positions and counts, length, signature, offset} is asked for one row ahead
    const int slot = l16 == 0 ? recs - 1 : l16 - 1;
    const bool walks = l16 < recs && pg_slot_walks(slot, recs, pb, pa.d, has_short);  // (k_pgplace stores only these records' entries)
    int2 h_rp = make_int2(0, 0);
    int4 h_ri = make_int4(0, 0, 0, 0);
    uint32_t h_key = PG_NONE;
    if (active) {
        if (walks) h_rp = recpos[(size_t)A * recs + slot];
        if (EXACT && l16 < recs) h_key = keys[(size_t)A * recs + slot];
        h_ri = rowinfo[A];
    }
    int T = 0, f = 0, st = 0, len_a = 0, off_a = 0, n_in = 0;
    uint32_t sa0 = 0, sa1 = 0;
    while (__builtin_amdgcn_ballot_w64(active) != 0ull) {
        if (fresh_row) {
            // the row's list: the members behind its records, step after step (step 0 = the SHORT record in slot recs - 1,
            // step i = slot i - 1); lane l of the group holds step l
            const int2 rp = h_rp;
            const int4 ri = h_ri;
            if (EXACT) g_key[l16] = h_key;
            const int nat = place_from(at + stride);
            h_rp = make_int2(0, 0);
            h_key = PG_NONE;
            if (nat < v_end) {
                const int nx = row_of(nat);
                if (walks) h_rp = recpos[(size_t)nx * recs + slot];
                if (EXACT && l16 < recs) h_key = keys[(size_t)nx * recs + slot];
                h_ri = rowinfo[nx];
            }
            const int incl = row16_incl_scan(rp.y);
            g_end[l16] = incl;
            g_base[l16] = rp.x + 1 - (incl - rp.y);  // entry e of the list, in step s, is position base[s] + e
            len_a = ri.x;
            sa0 = (uint32_t)ri.y;
            sa1 = (uint32_t)ri.z;
            off_a = ri.w;
            f = l16;
            st = 0;
            fresh_row = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (active) T = g_end[15];  // (steps at and beyond recs repeat the total)
        int4 rec[U];
        bool have[U];
        int step_u[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int fu = f + 16 * u;
            have[u] = active && fu < T;
            rec[u] = make_int4(0, 0, 0, 0);
            if (have[u]) {
                while (fu >= g_end[st]) st++;  // (fu < T = end[15]: stops at 15 at the latest)
                rec[u] = srec[g_base[st] + fu];
            }
            step_u[u] = st;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            bool pass = have[u] && abs(rec[u].y - len_a) <= d && __popc((uint32_t)rec[u].z ^ sa0) + __popc((uint32_t)rec[u].w ^ sa1) <= d;
            if (pa.dbg & 64) pass = have[u] && rec[u].x == -7;  // (BFK_PF_DEBUG=64: the walk alone; timing experiments, results invalid)
            my_visits += have[u] ? 1u : 0u;
            if (__builtin_amdgcn_ballot_w64(pass) == 0ull) continue;
            bool ins = false;
            if (EXACT) {
                // exact-edges steps (and edge capture) count a pair once: where the FIRST element its rows share is met — one of
                // A's earlier elements among B's records means this is a later one (seen from whichever row: both prefixes
                // are in the global order)
                if (pass) {
                    const uint32_t *kb = keys + (size_t)rec[u].x * recs;
                    uint32_t gb[PG_MAX_DIST + 2];
#pragma unroll
                    for (int j = 0; j < PG_MAX_DIST + 2; j++) gb[j] = j < recs ? kb[j] : PG_NONE;
                    bool dup = false;
                    for (int e2 = 0; e2 < step_u[u]; e2++) {
                        const uint32_t y = g_key[e2];
                        if (y == PG_NONE) continue;
#pragma unroll
                        for (int j = 0; j < PG_MAX_DIST + 2; j++) dup = dup || gb[j] == y;
                    }
                    pass = !dup;
                }
            } else if (pass && n_in < GSET / 2 && !(pa.dbg & 256)) {
                // met in an earlier step of this row?  (the set takes entries while it is less than half full; a member it
                // does not know is queued)
                const int B = rec[u].x;
                uint32_t h = ((uint32_t)B * 0x9E3779B1u) >> 26;  // GSET = 64 slots
                for (;;) {
                    const int old = atomicCAS(&g_set[h], -1, B);
                    if (old == B) {
                        pass = false;
                        break;
                    }
                    if (old == -1) {
                        ins = true;
                        break;
                    }
                    h = (h + 1) & (GSET - 1);
                }
            }
            n_in += __popc((unsigned)((__builtin_amdgcn_ballot_w64(ins) >> (grp * 16)) & 0xFFFFull));  // (group-uniform)
            const unsigned long long pm = __builtin_amdgcn_ballot_w64(pass);
            if (pm != 0ull) {
                if (pass) {
                    const int i = nq + below(pm);
                    sq[i] = make_int4(A, rec[u].x, off_a, -1);
                    sqk[i] = make_int2(len_a, rec[u].y);
                }
                nq += __popcll(pm);
                if (pa.dbg & 128) nq = 0;  // (BFK_PF_DEBUG=128: nothing is queued)
                if (nq >= 64) flush64();
            }
        }
        f += 16 * U;
        if (active && f - l16 >= T) {  // (group-uniform) this row is walked: the group's next one
            if (n_in > 0) {
#pragma unroll
                for (int j = 0; j < GSET / 16; j++) g_set[l16 + 16 * j] = -1;
                n_in = 0;
            }
            at = place_from(at + stride);
            active = at < v_end;
            if (active) A = row_of(at);
            fresh_row = active;
        }
    }
    while (nq > 0) flush64();
    // (a wave's members fit 32 bits by far: its rows x PG_GIVE_UP members per row on average at most)
    for (int o = 32; o > 0; o >>= 1) my_visits += (unsigned)__shfl_xor((int)my_visits, o);
    if (lane == 0) s_vis[wave] = my_visits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0ull;
        for (int w = 0; w < WAVES; w++) sum += s_vis[w];
        if (sum) atomicAdd(&pa.ctr->pairs_filtered, sum);
    }
}

// ------------------------------------------------------------------------------------------------
// k_flatten: labels[i] = root(i).  k_merge: unite (i, gathered[g][i]).  k_changed: fix-point flag.
// ------------------------------------------------------------------------------------------------
// k_compress: parent[i] = root(i) in place, between the two phases of a two-phase verify (only roots are written:
// a concurrent reader sees an ancestor either way)
__global__ void k_compress(int *parent, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cur = parent[i], next;
    while (cur > (next = parent[cur])) cur = next;
    parent[i] = cur;
}

// (no hooks run concurrently with this kernel, so plain cached loads and no compression stores)
// dyn_host (a device-driven text step): the step's OWN outcome of the variant join goes to its pinned slot — [5] a give-up of
// k_jhash / k_join (a probe chain over JOIN_MAX_PROBE, a dup list that overflowed), [6] the candidate queue overflowed, [7] the
// queue was expected empty and is not — and the device word is cleared: several such steps may be open at once, and the host
// looks at each step's slot when it completes that step (the device words alone would only ever describe the LAST step, and
// the next k_jhash resets `overflow`).
__global__ void k_flatten(const int *__restrict__ parent, int n, int *__restrict__ labels, Counters *ctr, int expect_empty_queue,
                          const int *dyn, int *dyn_host) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (dyn && (join_dyn_unusable(dyn) || join_dyn_outside(dyn))) return;  // (no forest was built: the host redoes the step)
    if (i == 0 && ctr) ctr->n_dup = 0;  // the dup list of the variant join: consumed, empty for the next step
    if (i == 0 && dyn_host) {  // (what the kernels in front of this one left: complete, this is a later launch)
        dyn_host[5] = ctr->join_fail;
        dyn_host[6] = ctr->overflow;
        ctr->join_fail = 0;
    }
    // k_verify was not launched because k_join decides every match of this CSR itself: a queue that is not empty all the
    // same means the step is redone on the all-pairs path
    if (expect_empty_queue && i < CAND_SHARDS && ctr->ncand[i] != 0u) {
        if (dyn_host) dyn_host[7] = 1;
        else ctr->join_fail = 1;
    }
    if (i >= n) return;
    int cur = parent[i], next;
    while (cur > (next = parent[cur])) cur = next;
    labels[i] = cur;
}

// (skip = the part that came from this forest itself, or -1; splice: uf_link instead of uf_union, see make_pair_args)
// The rank's own part IS its flat forest: own[i] = root of i.  A part that names own[i] adds nothing; otherwise (own[i], l)
// are united — and in a dense graph (max_dist >= 3: a few dozen components) nearly every row carries the SAME pair
// {the rank's giant root, the other rank's giant root}: a lane whose pair is its lower neighbour's leaves it to that lane.
// (One union per row instead: 0.75 - 0.98 ms at 1M rows, max_dist 5 — a million finds that all end in one word of the forest,
// served by one L2 channel per XCD.)
__global__ void k_merge(int *parent, int n, const int *__restrict__ gathered, int n_parts, Counters *ctr, int skip, int splice) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = i < n;
    int own = live && skip >= 0 ? gathered[(size_t)skip * n + i] : i;
    if (live && (own < 0 || own >= n)) {
        atomicOr(&ctr->err, ERR_LABEL);
        live = false;
        own = 0;
    }
    int last = own;  // (a part that says what the part before said adds nothing either)
    for (int g = 0; g < n_parts; g++) {  // (uniform trip count: the shuffles below are executed by whole waves)
        if (g == skip) continue;
        int l = live ? gathered[(size_t)g * n + i] : -1;
        bool act = live;
        if (live && (l < 0 || l >= n)) {
            atomicOr(&ctr->err, ERR_LABEL);
            act = false;
        }
        act = act && l != last && l != own;
        if (act) last = l;
        // the lower neighbour's pair (lane 0 of a wave has none)
        const int own_dn = __shfl_up(own, 1), l_dn = __shfl_up(l, 1);
        const bool act_dn = __shfl_up(act ? 1 : 0, 1) != 0;
        const bool same = (threadIdx.x & 63) != 0 && act_dn && own_dn == own && l_dn == l;
        if (act && !same) {
            // a first look by plain loads (served by the CU's cache: in a dense graph every chain ends in the one word of the
            // giant root, and a million coherent loads of one word queue up at one L2 channel): an ancestor is an ancestor
            // whatever its age, so equal ancestors mean connected; anything else is decided by the coherent union
            int ra = own, rb = l, nx;
            while (ra > (nx = parent[ra])) ra = nx;
            while (rb > (nx = parent[rb])) rb = nx;
            if (ra != rb) {
                if (splice) uf_link(parent, own, l); else uf_union(parent, own, l);
            }
        }
    }
}

// cache path: every neighbour list is a path (a,b),(b,c),... (reference _to_edges, breakfast.py:103-113)
__global__ void k_union_lists(int *parent, const long long *__restrict__ off, const int *__restrict__ flat,
                              long long total, int n_lists, int n, Counters *ctr) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    // list of element e: largest l with off[l] <= e
    int lo = 0, hi = n_lists - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= e) lo = mid;
        else hi = mid - 1;
    }
    if (e == off[lo]) return;  // first element of its list: no edge to a predecessor
    const int a = flat[e - 1], b = flat[e];
    if (a < 0 || a >= n || b < 0 || b >= n) {
        atomicOr(&ctr->err, ERR_LABEL);
        return;
    }
    if (a != b) uf_union(parent, a, b);
}

__global__ void k_init_parent(int *parent, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = i;
}

__global__ void k_changed(const int *__restrict__ labels, const int *__restrict__ ref, int n, int *changed) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool c = i < n && labels[i] != ref[i];
    if (__any(c) && (threadIdx.x & 63) == 0) atomicOr(changed, 1);
}

// ------------------------------------------------------------------------------------------------
// host-side launchers (called from bfk_host.cpp)
// ------------------------------------------------------------------------------------------------
#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

int launch_maxlen(const int *indptr, int n, int *out, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_maxlen, dim3(std::min(1024, (n + 255) / 256)), dim3(256), 0, st, indptr, n, out);
    LAUNCH_CHECK();
    return 0;
}

int launch_maxtok(const uint32_t *indices, int nnz, int *out, hipStream_t st) {
    if (nnz <= 0) return 0;
    hipLaunchKernelGGL(k_maxtok, dim3(std::min(1024, (nnz + 255) / 256)), dim3(256), 0, st, indices, nnz, out);
    LAUNCH_CHECK();
    return 0;
}

static PairArgs make_pair_args(const Plan &pl) {
    PairArgs pa;
    pa.indptr = pl.indptr;
    pa.rows = pl.indices;
    pa.srec = pl.srec;
    pa.n = pl.n;
    pa.nnz = pl.nnz;
    pa.union_batch = pl.d == 3 ? 8 : 16;  // (d = 2 hooks by splicing: 16 is better there, 0.169 vs 0.172 ms at 100k rows)
    if (const char *e = getenv("BFK_UNION_BATCH")) pa.union_batch = atoi(e) >= 16 ? 16 : (atoi(e) >= 8 ? 8 : (atoi(e) >= 4 ? 4 : 2));
    pa.dbg = pl.dbg;
    pa.part_lo = 0;
    pa.part_hi = pa.part_den = 1;
    pa.stats_off = 0;
    pa.skip_connected = pl.skip_connected && !pl.edges;
    // measured, verify kernel in us (splicing / find + hook): 100k rows d = 2: 77 / 115-128, 1M rows d = 1: 75 / 111 —
    // nearly every edge joins two trees and one atomic does it; d = 3: 470 / 361, d = 5: 2640 / 840 — most edges are
    // redundant there and find + hook ends them with two loads (equal parents), splicing walks up with atomics
    pa.use_link = pl.d <= 2 ? 1 : 0;
    if (const char *e = getenv("BFK_UF_LINK")) pa.use_link = atoi(e) != 0;
    pa.sel = pl.edges ? pl.edge_sel : nullptr;
    pa.density_mode = 0;
    pa.density_thr = 0;
    pa.parent = pl.parent;
    pa.cand = pl.cand;
    pa.candk = pl.candk;
    pa.cand_cap_shard = pl.cand_cap_shard;
    pa.d = pl.d;
    pa.ctr = pl.ctr;
    return pa;
}

// the STEPS instantiation that serves rows of `steps` 16-token steps
static int verify_steps(int steps) { return steps <= 3 ? 3 : steps <= 4 ? 4 : steps <= 5 ? 5 : steps <= 6 ? 6 : steps <= 8 ? 8 : steps <= 12 ? 12 : 16; }

// exact check + union of everything in the candidate queue
static int launch_verify(const Plan &pl, const PairArgs &pa_in0, hipStream_t st, hipEvent_t *ev) {
    // 16-token steps covering the longest row a pair of k_verify can have
    const int steps = (std::min(pl.kcap, 16 * 16) + 15) / 16;  // (rows of up to 256 tokens in the registers; longer ones: k_verify_long)
    PairArgs pa_in = pa_in0;
    pa_in.max_tokens = verify_max_tokens(verify_steps(steps));
    auto one = [&](const PairArgs &pa, int grid) {
#define VF_CASE(S)                                                                                                        \
    if (pa.skip_connected && pl.d <= pl.wave_table_d)                                                                     \
        hipLaunchKernelGGL((k_verify_connected<S, true>), dim3(grid), dim3(256), 0, st, pa, pl.blk_stats);                \
    else if (pa.skip_connected)                                                                                           \
        hipLaunchKernelGGL((k_verify_connected<S, false>), dim3(grid), dim3(256), 0, st, pa, pl.blk_stats);               \
    else if (pl.d <= pl.wave_table_d)                                                                                     \
        hipLaunchKernelGGL((k_verify<S, true>), dim3(grid), dim3(256), 0, st, pa, pl.edges, pl.edge_cap,                  \
                           pl.blk_stats);                                                                                 \
    else                                                                                                                  \
        hipLaunchKernelGGL((k_verify<S, false>), dim3(grid), dim3(256), 0, st, pa, pl.edges, pl.edge_cap,                 \
                           pl.blk_stats)
        if (steps <= 3) { VF_CASE(3); }
        else if (steps <= 4) { VF_CASE(4); }
        else if (steps <= 5) { VF_CASE(5); }
        else if (steps <= 6) { VF_CASE(6); }
        else if (steps <= 8) { VF_CASE(8); }
        else if (steps <= 12) { VF_CASE(12); }
        else { VF_CASE(16); }
#undef VF_CASE
    };
    const PairArgs &pa = pa_in;
    if (pl.verify_phases > 1) {
        // Two phases (the idea of Afforest): a first part of every queue shard is hooked by splicing, the forest is
        // compressed, and the rest — in a dense graph mostly edges inside components that exist by then — ends after
        // one round trip of loads (equal roots)
        PairArgs a = pa_in, b = pa_in;
        a.part_lo = 0;
        a.part_hi = 1;
        a.part_den = pl.verify_phases;
        a.use_link = 1;
        b.part_lo = 1;
        b.part_hi = b.part_den = pl.verify_phases;
        b.use_link = pl.verify_phase2_union;
        b.stats_off = 2 * pl.verify_grid;
        one(a, pl.verify_grid);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_compress, dim3((pl.n + 255) / 256), dim3(256), 0, st, pl.parent, pl.n);
        LAUNCH_CHECK();
        one(b, pl.verify_grid);
    } else if (pl.verify_adaptive) {
        // max_dist 2, labels only: whether the graph is a sparse forest (k_verify, splicing: the default generator's data, 2-3
        // candidates per row) or dense (a star phylogeny: a hub with thousands of neighbours at distance 1 makes all of THEM
        // neighbours at distance 2 — 180 candidates per row at 1M rows, 67-73 ms for k_verify) is known on the device when the
        // queue is full: both kernels are launched, the queue's fill decides which one works (the other costs ~5 us)
        PairArgs sp = pa_in, dn = pa_in;
        sp.density_mode = 1;
        dn.density_mode = 2;
        sp.density_thr = dn.density_thr = pl.verify_density_thr;
        dn.skip_connected = 1;
        dn.use_link = 0;
        dn.stats_off = 2 * pl.verify_grid;
        one(sp, pl.verify_grid);
        LAUNCH_CHECK();
        one(dn, pl.verify_grid2);
    } else {
        one(pa, pl.verify_grid);
    }
    LAUNCH_CHECK();
    if (2 * pl.kcap > verify_max_tokens(verify_steps(steps))) {  // some pair may exceed a group table
        // (a block per pair; with a global scratch table — rows of thousands of tokens — only LONG_BLOCKS slices of it exist)
        hipLaunchKernelGGL(k_verify_long, dim3(pl.gslots ? LONG_BLOCKS : LONG_BLOCKS_LDS), dim3(256), (size_t)LONG_TABLE * 8, st, pa, pl.gkey,
                           pl.gcnt, pl.gslots, pl.edges, pl.edge_cap);
        LAUNCH_CHECK();
    }
    if (ev) (void)hipEventRecord(ev[3], st);
    return 0;
}

// prefilter + verify + union over tiles [t_begin, t_end) (whole list for a normal run; slices in recovery)
int launch_pairs(const Plan &pl, int t_begin, int t_end, hipStream_t st, hipEvent_t *ev) {
    const int n = pl.n;
    PairArgs pa = make_pair_args(pl);
    BandArgs ba;
    ba.start3 = pl.start3;
    ba.tiles = pl.tiles;
    ba.tile_slots = pl.tile_slots;
    ba.dbg_t = (pl.dbg & 4) ? pl.dbg_t : nullptr;
    ba.key.fb = pl.fb;
    ba.key.gb = pl.gb;
    ba.key.hb = pl.hb;
    ba.key.fb_log = __builtin_ctz((unsigned)pl.fb);
    ba.key.gb_log = __builtin_ctz((unsigned)pl.gb);
    ba.key.hb_log = __builtin_ctz((unsigned)pl.hb);
    {
        static const int aligned_env = [] { const char *e = getenv("BFK_PF_ALIGNED"); return e ? atoi(e) : 0; }();
        ba.chunk_aligned = aligned_env;
    }
    ba.inv_d1 = (65536 + pl.d) / (pl.d + 1);  // ceil(65536 / (d + 1)); only used while (d+1)^2 <= 64
    {
        const long long d2 = (long long)(pl.d + 1) * (pl.d + 1);
        ba.inv_d2 = (int)((65536 + d2 - 1) / d2);  // only used while (d+1)^3 <= 64
    }
    ba.kcap = pl.kcap;
    ba.d = pl.d;
    // one block per tile of this shard; the tile count lives on the device, so the grid is sized from the
    // previous step's count (tile_hint) and the kernel strides over whatever the count turns out to be
    const long long span = (long long)std::min(std::min(t_end, pl.tile_cap), t_begin + pl.tile_hint) - t_begin;
    const int grid = std::max(1, (int)std::min<long long>(span, pl.pf_blocks));  // every rank walks all tiles, skips foreign cells

    if (pl.pg) {  // prefix-group path: one wave per row walks the row's groups (work items = blocks of 64 rows)
        const int items = std::max(0, std::min(t_end, (n + 63) / 64) - t_begin);
        // the waves stride over the rows; blocks per CU, k_pgjoin at 16 / 32 / 64 / 128 / 256 (round 3: one row per wave — 62 500
        // blocks at 1M rows — costs more to dispatch than its balance gives): 1M rows, max-dist 2: 0.45 / 0.44 / 0.45 / 0.58 / 0.86 ms;
        // 300k rows: . / 0.18 / 0.24 / . / 0.80; exact-edges step at max-dist 5, 1M rows: . / 0.95 / 0.89 / . / 1.08
        static const int per_cu_env = [] { const char *e = getenv("BFK_PG_BLOCKS"); return e ? std::max(1, atoi(e)) : 0; }();
        const int per_cu = per_cu_env ? per_cu_env : (pl.d <= 2 ? 32 : 64);
        const int blocks = std::max(1, std::min(std::min(pl.pf_blocks, pl.pf_blocks / 256 * per_cu), (int)std::min<long long>((long long)items * 16, 1 << 20)));
        if (pl.pg_walk16) {  // 16 lanes per row (BFK_PG_WALK16=0: a wave per row, k_pgjoin)
            // (a group takes rows in turn: 32 blocks per CU balance as well as one row per group did, and 62 500 blocks cost 0.5 ms
            // to dispatch; the grid is sized for the rank's share of the rows)
            const int wper = per_cu_env ? per_cu_env : 32;
            const int own_items = (items + pl.n_shards - 1) / std::max(1, pl.n_shards);
            const int wblocks = std::max(1, std::min(pl.pf_blocks / 256 * wper, (int)std::min<long long>((long long)own_items * 4, 1 << 20)));
            if (pa.skip_connected)
                hipLaunchKernelGGL(k_pgwalk16<false>, dim3(wblocks), dim3(256), 0, st, pl.pg_srec, pl.pg_recpos, pl.pg_rowinfo, pl.pg_keys, n,
                                   pl.pg_recs, pl.shard, pl.n_shards, t_begin, t_end, pa, pl.pg_pb, pl.pg_has_short);
            else
                hipLaunchKernelGGL(k_pgwalk16<true>, dim3(wblocks), dim3(256), 0, st, pl.pg_srec, pl.pg_recpos, pl.pg_rowinfo, pl.pg_keys, n,
                                   pl.pg_recs, pl.shard, pl.n_shards, t_begin, t_end, pa, pl.pg_pb, pl.pg_has_short);
        } else {
            hipLaunchKernelGGL(k_pgjoin, dim3(blocks), dim3(256), 0, st, pl.pg_srec, pl.pg_recpos, pl.pg_keys, pl.pg_rowinfo, n, pl.pg_recs,
                               n * pl.pg_recs, pl.shard, pl.n_shards, t_begin, t_end, pa, pl.pg_pb, pl.pg_has_short);
        }
        LAUNCH_CHECK();
        if (ev) (void)hipEventRecord(ev[2], st);
        return launch_verify(pl, pa, st, ev);
    }
// the instrumented instantiations (BFK_PF_DEBUG: per-wave stamps, phase switches) double the code of the pair
// kernel: they are compiled only with `make PF_DEBUG=1`
#ifdef BFK_WITH_PF_DEBUG
#define PF_CASE_PW(W, R, PW)                                                                                       \
    if (pl.dbg)                                                                                                    \
        hipLaunchKernelGGL((k_prefilter<W, R, PW, true>), dim3(grid), dim3(PW * 64), 0, st, pl.sig1, ba, n, pl.shard,      \
                           pl.n_shards, t_begin, t_end, pa);                                                       \
    else                                                                                                           \
        hipLaunchKernelGGL((k_prefilter<W, R, PW, false>), dim3(grid), dim3(PW * 64), 0, st, pl.sig1, ba, n, pl.shard,     \
                           pl.n_shards, t_begin, t_end, pa)
#else
#define PF_CASE_PW(W, R, PW)                                                                                       \
    hipLaunchKernelGGL((k_prefilter<W, R, PW, false>), dim3(grid), dim3(PW * 64), 0, st, pl.sig1, ba, n, pl.shard,         \
                       pl.n_shards, t_begin, t_end, pa)
#endif
#define PF_CASE(W, R) PF_CASE_PW(W, R, 2)
    // waves per tile: 2, or 4 for 64-row tiles of small inputs (there a tile is ~8 chunks and the kernel time is
    // set by the busiest SIMD: four waves on four SIMDs per tile even that out; large inputs are throughput-
    // bound and the extra per-wave set-up only costs)
    const bool pw4 = pl.pf_waves == 4;
    switch (pl.w1 * 10 + pl.rows_per_lane) {
        case 11: if (pw4) { PF_CASE_PW(1, 1, 4); } else { PF_CASE(1, 1); } break;
        case 12: PF_CASE(1, 2); break;
        case 14: PF_CASE(1, 4); break;
        case 21: if (pw4) { PF_CASE_PW(2, 1, 4); } else { PF_CASE(2, 1); } break;
        case 22: PF_CASE(2, 2); break;
        case 24: PF_CASE(2, 4); break;
        case 41: if (pw4) { PF_CASE_PW(4, 1, 4); } else { PF_CASE(4, 1); } break;
        case 42: PF_CASE(4, 2); break;
        default: PF_CASE(4, 4); break;
    }
#undef PF_CASE
#undef PF_CASE_PW
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[2], st);
    return launch_verify(pl, pa, st, ev);
}

int launch_flatten(const Plan &pl, hipStream_t st, hipEvent_t *ev) {
    hipLaunchKernelGGL(k_flatten, dim3((std::max(pl.n, CAND_SHARDS) + 255) / 256), dim3(256), 0, st, pl.parent, pl.n, pl.labels, pl.ctr,
                       pl.join && pl.join_skip_verify ? 1 : 0, pl.join ? pl.ja.dyn : nullptr, pl.join && pl.ja.dyn ? pl.ja.dyn_host : nullptr);
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[4], st);
    return 0;
}

int launch_pipeline(const Plan &pl, hipStream_t st, hipEvent_t *ev /*5 or NULL*/) {
    const int n = pl.n;
    if (ev) (void)hipEventRecord(ev[0], st);
    if (pl.join) {  // max_dist == 1: hash join instead of the all-pairs kernels
        const int rpw = max(1, min(16, (n + 8191) / 8192));
        const int blocks = (n + rpw * 16 - 1) / (rpw * 16);
        const PairArgs pa = make_pair_args(pl);
        hipLaunchKernelGGL(k_jhash, dim3(blocks), dim3(1024), 0, st, pl.indptr, pl.indices, n, pl.nnz, pl.kcap, rpw, pl.ja,
                           pl.parent, pl.ctr);
        LAUNCH_CHECK();
        if (ev) (void)hipEventRecord(ev[1], st);
        const int jblocks = std::max(1, (int)(((long long)pl.nnz + 16 * JOIN_TPW - 1) / (16 * JOIN_TPW)));
        hipLaunchKernelGGL(k_join, dim3(jblocks), dim3(1024), 0, st, pl.indptr, pl.indices, n, pl.nnz, pl.ja, pa, pl.shard,
                           pl.n_shards, pl.edges, pl.edge_cap);
        LAUNCH_CHECK();
        if (ev) (void)hipEventRecord(ev[2], st);
        if (pl.join_skip_verify) {  // k_join decided every match itself: nothing is queued (k_flatten checks that)
            if (ev) (void)hipEventRecord(ev[3], st);
        } else if (int e = launch_verify(pl, pa, st, ev)) {
            return e;
        }
        return launch_flatten(pl, st, ev);
    }
    if (pl.pg) {  // prefix-group path: sampled token counts, records (+ signatures, forest, counters), sort, group order — none of the band kernels' prep
        const int total = n * (pl.pg_has_short ? pl.pg_recs : pl.pg_recs - 1);  // records in the sorted order
        if (hipMemsetAsync(pl.pg_cnt, 0, sizeof(uint32_t) << PG_CNT_BITS, st) != hipSuccess) return (int)hipGetLastError();
        const int stride = std::max(1, n / 4096);  // ~4k sampled rows: the counts only have to tell common tokens from rare ones
        const int sampled = (n + stride - 1) / stride;
        hipLaunchKernelGGL(k_pgfreq, dim3((sampled + 63) / 64), dim3(256), 0, st, pl.indptr, pl.indices, n, stride, pl.pg_cnt, pl.ctr, pl.pg_dense);
        LAUNCH_CHECK();
        const bool walk_reads_keys = !(pl.pg_walk16 && pl.skip_connected && !pl.edges);  // (k_pgjoin and the exact-edges walk do)
        hipLaunchKernelGGL(k_pgkeys, dim3((n + 16 * PGK_ROWS - 1) / (16 * PGK_ROWS)), dim3(256), 0, st, pl.indptr, pl.indices, n, pl.pg_recs, pl.d, pl.pg_tb, pl.pg_cnt,
                           walk_reads_keys ? pl.pg_keys : (uint32_t *)nullptr, pl.pg_keys_pm, (int *)nullptr, pl.pg_pb ? 1 : 0, pl.kcap, pl.parent, pl.pg_rowinfo, pl.ctr, pl.pg_dense, pl.pg_has_short);
        LAUNCH_CHECK();
        size_t tb = pl.pg_temp_bytes;
        // (the sort's last pass leaves the composite {key : slot} the positional filter bisects on)
        // (values: position p of the position-major input holds (row p % n, slot p / n); of the row-major one, p itself)
        if (int e = sort_records(pl.pg_temp, &tb, pl.pg_keys_pm, pl.pg_keys_s, nullptr, pl.pg_rows_s, (size_t)total, pl.pg_tb, st,
                                 pl.pg_pb ? pl.pg_recs : 0, pl.pg_pb, pl.pg_pb ? n : 0, pl.pg_recs))
            return e;
        hipLaunchKernelGGL(k_pgplace, dim3((total + 255) / 256), dim3(256), 0, st, pl.pg_keys_s, pl.pg_rows_s, total, pl.pg_recs, pl.pg_rowinfo,
                           pl.pg_srec, pl.pg_recpos, pl.ctr, pl.pg_pb, pl.d, pl.pg_has_short);
        LAUNCH_CHECK();
        if (ev) (void)hipEventRecord(ev[1], st);
        if (int e = launch_pairs(pl, 0, 0x7FFFFFFF, st, ev)) return e;
        return launch_flatten(pl, st, ev);
    }
    KeyCfg key;
    key.fb = pl.fb;
    key.gb = pl.gb;
    key.hb = pl.hb;
    key.fb_log = __builtin_ctz((unsigned)pl.fb);
    key.gb_log = __builtin_ctz((unsigned)pl.gb);
    key.hb_log = __builtin_ctz((unsigned)pl.hb);
    // rows per wave of k_sig: at most 16, and for small inputs few enough that every CU holds two blocks
    // (512 blocks x 16 waves co-resident): block-granular imbalance would otherwise cost up to 30%
    const int rpw = max(1, min(16, (n + 8191) / 8192));
    const int sig_blocks = (n + rpw * 16 - 1) / (rpw * 16);
    CellArgs ca;
    ca.hist3 = pl.hist3;
    ca.start3 = pl.start3;
    ca.tiles = pl.tiles;
    ca.chain = pl.chain;
    ca.ctr = pl.ctr;
    ca.n = n;
    ca.cells = (pl.kcap + 1) * pl.fb * pl.gb * pl.hb;
    ca.tr_shift = 6 + (pl.rows_per_lane == 1 ? 0 : (pl.rows_per_lane == 2 ? 1 : 2));
    ca.start3c = pl.start3c;
    ca.shard0 = pl.shard;
    ca.nshards = pl.n_shards;
    const int cell_blocks = (ca.cells + 1023) / 1024;
    const int copies = pl.hist_copies;
    ca.tile_cap = pl.tile_cap;
    switch (pl.w1) {
#define PREP_CASE(W)                                                                                                      \
    case W:                                                                                                               \
        hipLaunchKernelGGL(k_sig<W>, dim3(sig_blocks), dim3(1024), 0, st, pl.indptr, pl.indices, n, pl.nnz, pl.kcap, rpw,  \
                           key,                                                                                           \
                           pl.rowkey, pl.parent, pl.sigu1, pl.sigu2, pl.hist3, pl.rowrank, pl.ctr, pl.dbg, ca.cells,       \
                           copies);                                                                                       \
        if (copies == 8)                                                                                                  \
            hipLaunchKernelGGL(k_cells<8>, dim3(cell_blocks), dim3(1024), 0, st, ca);                                     \
        else if (copies == 2)                                                                                             \
            hipLaunchKernelGGL(k_cells<2>, dim3(cell_blocks), dim3(1024), 0, st, ca);                                     \
        else                                                                                                              \
            hipLaunchKernelGGL(k_cells<1>, dim3(cell_blocks), dim3(1024), 0, st, ca);                                     \
        hipLaunchKernelGGL(k_place<W>, dim3((max(n, cell_blocks) + 255) / 256), dim3(256), 0, st, pl.indptr, n, pl.kcap,   \
                           copies > 1 ? pl.start3c : pl.start3, pl.rowkey, pl.rowrank, pl.sigu1, pl.sigu2, pl.srec,        \
                           pl.sig1, pl.chain, cell_blocks, ca.cells, copies, rpw * 16);                                   \
        break;
        PREP_CASE(1)
        PREP_CASE(2)
        PREP_CASE(4)
#undef PREP_CASE
        default:
            return -1;
    }
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[1], st);
    if (int e = launch_pairs(pl, 0, 0x7FFFFFFF, st, ev)) return e;
    return launch_flatten(pl, st, ev);
}

int launch_lists(int *parent, int n, const long long *off, const int *flat, long long total, int n_lists, int *labels,
                 Counters *ctr, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_init_parent, dim3((n + 255) / 256), dim3(256), 0, st, parent, n);
    LAUNCH_CHECK();
    if (total > 0 && n_lists > 0) {
        hipLaunchKernelGGL(k_union_lists, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, parent, off, flat, total,
                           n_lists, n, ctr);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_flatten, dim3((n + 255) / 256), dim3(256), 0, st, (const int *)parent, n, labels, (Counters *)nullptr, 0, (const int *)nullptr, (int *)nullptr);
    LAUNCH_CHECK();
    return 0;
}

int launch_merge(int *parent, int n, const int *gathered, int n_parts, int *labels, int *changed, Counters *ctr,
                 int skip, int splice, hipStream_t st) {
    if (n <= 0) return 0;
    dim3 g((n + 255) / 256), b(256);
    hipLaunchKernelGGL(k_merge, g, b, 0, st, parent, n, gathered, n_parts, ctr, skip, splice);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_flatten, g, b, 0, st, (const int *)parent, n, labels, (Counters *)nullptr, 0, (const int *)nullptr, (int *)nullptr);
    LAUNCH_CHECK();
    if (changed) {
        hipLaunchKernelGGL(k_changed, g, b, 0, st, (const int *)labels, gathered, n, changed);
        LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace bfk
