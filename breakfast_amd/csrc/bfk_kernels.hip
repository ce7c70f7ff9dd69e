// bfk_kernels.hip — HIP kernels for gfx950 (MI355X, CDNA4): the breakfast clustering hot path.
//
// Pipeline (one stream, no host round trip between kernels):
//   k_canon     one wave per row: sort key bin (k, f), bitonic sort of the token ids in registers, duplicate
//               ranks, two XOR-parity signatures; parent[i] = i
//   k_rowrank   (k,f,g) histogram + rank of every row inside its key
//   k_scan      start[] of the (k,f) bins, prefix of the g sub-bins; re-zeroes histogram and counters
//   k_tiles     per wave-tile the column ranges of its (k,f,g) band -> work items, cut into equal unit slices
//   k_place     counting-sort scatter of row ids / lengths / signatures into (k,f) order
//   k_canon_long block per row for k > 256 (rank sort, row staged in LDS)
//   k_prefilter THE dominant kernel: all in-band pairs, popcount(sig_p ^ sig_q) <= d  (necessary
//               condition for |A delta B| <= d); survivors pass a 128-bit second level and are queued
//   k_verify    one wave per candidate: row B staged in LDS, lanes binary-search A's elements,
//               ballot/popcount -> exact multiset distance; <= d -> lock-free union-find hook
//   k_flatten   labels[i] = root(i) = smallest row index of the component
//   k_merge     multi-GPU: unite (i, gathered[g][i]) pseudo-edges
//
// What it replaces in the reference: the band loop + get_neighbours_batch + sklearn _sparse_manhattan +
// _reduce_func + networkx components (src/breakfast/breakfast.py:223-276, 287-326).
#include "bfk_device.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfk {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ld_agent(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
// smallest index of its tree and parent[x] <= x.  Loads may be stale (other XCD's L2): a stale value
// is an older, still valid ancestor or "x is a root", and the deciding step is always the CAS.
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
    int ra = uf_find(parent, a), rb = uf_find(parent, b);
    while (ra != rb) {
        if (ra < rb) {
            int t = ra;
            ra = rb;
            rb = t;
        }
        int old = atomicCAS(parent + ra, ra, rb);  // ra > rb
        if (old == ra) return true;
        ra = uf_find(parent, old);  // ra was no longer a root: go to the current root (cheap loads, not CASes)
    }
    return false;
}

// signature hashes of the composite key (token id x, repeat rank r)
__device__ __forceinline__ uint32_t hash1(uint32_t x, uint32_t r) { return x * 0x9E3779B1u + r * 0x7FEB352Du; }
__device__ __forceinline__ uint32_t hash2(uint32_t x, uint32_t r) {
    uint32_t h = (x ^ (x >> 15)) * 0x85EBCA6Bu + r * 0xC2B2AE35u;
    return h ^ (h >> 13);
}

// ------------------------------------------------------------------------------------------------
// k_maxlen: validate indptr, find the longest row (bind time only)
// ------------------------------------------------------------------------------------------------
__global__ void k_maxlen(const int *__restrict__ indptr, int n, int *out /*[0]=max k, [1]=err*/) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int k = 0;
    if (i < n) {
        k = indptr[i + 1] - indptr[i];
        if (k < 0) {
            atomicOr(out + 1, 1);
            k = 0;
        }
    }
    for (int s = 32; s > 0; s >>= 1) k = max(k, __shfl_xor(k, s));
    if ((threadIdx.x & 63) == 0 && k > 0) atomicMax(out, k);
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
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fbit(uint32_t x) { return (x * 0xB5297A4Du) >> 31; }
__device__ __forceinline__ uint32_t gbit(uint32_t x) { return ((x ^ (x >> 11)) * 0x2C1B3C6Du) >> 31; }

struct KeyCfg {
    int fb, gb;  // buckets per row length for f and g (powers of two; 1 = key unused)
};
__device__ __forceinline__ int key_center(int k, int nb) { return (k >> 1) - (nb >> 1); }
__device__ __forceinline__ int key_bucket(int v, int k, int nb) { return min(max(v - key_center(k, nb), 0), nb - 1); }
__device__ __forceinline__ int key3_of(const KeyCfg &c, int k, int f, int g) {
    return (k * c.fb + key_bucket(f, k, c.fb)) * c.gb + key_bucket(g, k, c.gb);
}

// bucket range [lo, hi] a column of length kp can have in one key, given the rows' bucket range [rlo, rhi]
// at length k = kp - delta (delta >= 0): raw difference in [-amax, bmax], the window moves by s = c(kp) - c(k),
// and a clamped end bucket stands for an open interval.
__device__ __forceinline__ void key_band(int rlo, int rhi, int k, int kp, int amax, int bmax, int nb, int *lo, int *hi) {
    const int s = key_center(kp, nb) - key_center(k, nb);
    *lo = (rlo <= 0) ? 0 : max(rlo - amax - s, 0);
    *hi = (rhi >= nb - 1) ? nb - 1 : min(rhi + bmax - s, nb - 1);
}

// block-wide exclusive scan (1024 threads, one 64-bit value each); tmp: 40 x u64 of LDS
__device__ __forceinline__ unsigned long long block_excl_scan_1024(unsigned long long v, unsigned long long *tmp,
                                                                   unsigned long long *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
    for (int s = 1; s < 64; s <<= 1) {
        unsigned long long y = __shfl_up(inc, s);
        if (lane >= s) inc += y;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        unsigned long long w = lane < 16 ? tmp[lane] : 0ull;
        unsigned long long winc = w;
        for (int s = 1; s < 16; s <<= 1) {
            unsigned long long y = __shfl_up(winc, s);
            if (lane >= s) winc += y;
        }
        if (lane < 16) tmp[16 + lane] = winc - w;
        if (lane == 15) tmp[32] = winc;
    }
    __syncthreads();
    unsigned long long res = inc - v + tmp[16 + wave];
    *total = tmp[32];
    __syncthreads();
    return res;
}

struct PlanArgs {
    int *hist3;        // (kcap+1)*fb*gb counters, re-zeroed by k_scan
    int *sub3;         // exclusive prefix of hist3 inside every (k,f) bin
    int *start;        // (kcap+1)*fb + 1: first sorted position of every (k,f) bin
    const int *keysorted;  // key3 of the row at every sorted position (k_tiles)
    int4 *items;       // {row0, cbeg, cend, ustart}
    int *blk_item;     // first item of every worker
    Counters *ctr;
    KeyCfg key;
    int n, kcap, d, tr, cb, nvblocks, item_cap;
};

// ------------------------------------------------------------------------------------------------
// k_scan (one block): per (k,f) bin the exclusive prefix of its g sub-bins (sub3) and, across bins, the
// exclusive scan of the bin totals (start).  Re-zeroes the histogram and the per-step counters, so a step
// needs no memset.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan(PlanArgs a) {
    __shared__ unsigned long long tmp[40];
#define PLAN_STAMP(i) if (threadIdx.x == 0) a.ctr->dbg[i] = wall_clock64();
    PLAN_STAMP(0)
    const int bins = (a.kcap + 1) * a.key.fb, gb = a.key.gb;
    const int per = (bins + 1023) / 1024;
    if (threadIdx.x < CAND_SHARDS) a.ctr->ncand[threadIdx.x] = 0;
    if (threadIdx.x == 64) {
        a.ctr->err = 0;
        a.ctr->overflow = 0;
        a.ctr->n_long = 0;  // consumed by k_canon_long, which ran before this kernel
        a.ctr->n_edges = a.ctr->n_cand_total = a.ctr->n_edges_cap = 0;
    }
    const int b0 = threadIdx.x * per, b1 = min(bins, b0 + per);
    unsigned long long sum = 0;
    for (int b = b0; b < b1; b++) {
        int run = 0;
        for (int g = 0; g < gb; g++) {
            const int c = a.hist3[(size_t)b * gb + g];
            a.sub3[(size_t)b * gb + g] = run;
            a.hist3[(size_t)b * gb + g] = 0;
            run += c;
        }
        a.start[b] = run;  // bin total for now
        sum += (unsigned)run;
    }
    unsigned long long tot;
    const unsigned long long ex = block_excl_scan_1024(sum, tmp, &tot);
    int run = (int)ex;
    for (int b = b0; b < b1; b++) {
        const int c = a.start[b];
        a.start[b] = run;
        run += c;
    }
    if (threadIdx.x == 0) a.start[bins] = a.n;
    PLAN_STAMP(1)
}

// first sorted position of key (bin2, g); g == gb means the end of the bin
__device__ __forceinline__ int start3(const PlanArgs &a, int bin2, int g) {
    return g >= a.key.gb ? a.start[bin2 + 1] : a.start[bin2] + a.sub3[(size_t)bin2 * a.key.gb + g];
}

// ------------------------------------------------------------------------------------------------
// k_tiles (one block, after k_place): for every wave-tile of TR sorted rows the column ranges it must scan.
// A tile inside ONE (k,f) bin (the heavy case: ~1000 rows per bin at 100k rows) gets one range per column
// (k', f') with the g band of its own rows; a tile that spans bins gets one range per column length with
// the full f band of its rows.  Candidates are flattened (one thread each), touching ranges of a tile are
// merged, and the ranges become work items with a running count of "units" (one unit = one chunk of CB
// columns against the tile, plus a fixed price per item); the units are what the prefilter's waves divide
// among themselves, so every wave gets the same amount of work.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_tiles(PlanArgs a) {
    __shared__ unsigned long long tmp[40];
    __shared__ int4 s_tile[1024];   // {key3 of first row, key3 of last row, n candidates, single-bin flag}
    __shared__ int4 s_rng[1024];
    __shared__ int s_cpre[1025];
    PLAN_STAMP(2)
    const int fb = a.key.fb, gb = a.key.gb, d = a.d, d1 = a.d + 1;
    {   // statistic: unordered pairs inside the reference's length band  sum_k c_k(c_k-1)/2 + sum_{k<k'<=k+d} c_k c_k'
        unsigned long long acc = 0;
        for (int k = threadIdx.x; k <= a.kcap; k += 1024) {
            const unsigned long long c = (unsigned)(a.start[(k + 1) * fb] - a.start[k * fb]);
            if (!c) continue;
            const int k2 = min(k + d, a.kcap);
            const unsigned long long s = (unsigned)(a.start[(k2 + 1) * fb] - a.start[(k + 1) * fb]);
            acc += c * (c - 1) / 2 + c * s;
        }
        for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
        if (threadIdx.x == 0) a.ctr->pairs_in_band = 0;
        __syncthreads();
        if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&a.ctr->pairs_in_band, acc);
    }
    const int T = (a.n + a.tr - 1) / a.tr;
    unsigned long long carry_i = 0, carry_u = 0;
    for (int base = 0; base < T; base += 1024) {
        const int t = base + threadIdx.x;
        int nc = 0;
        if (t < T) {
            const int row0 = t * a.tr;
            const int plast = min(a.n, row0 + a.tr) - 1;
            const int key_lo = a.keysorted[row0], key_hi = a.keysorted[plast];
            const int bin_lo = key_lo / gb, bin_hi = key_hi / gb;
            const int k_lo = bin_lo / fb, k_hi = bin_hi / fb;
            const bool single = bin_lo == bin_hi && d1 * d1 <= 64;  // large d: one range per column length
            nc = single ? d1 * d1 : (min(k_hi + d, a.kcap) - k_lo + 1);
            s_tile[threadIdx.x] = make_int4(key_lo, key_hi, nc, single ? 1 : 0);
        }
        unsigned long long totc;
        const unsigned long long exc = block_excl_scan_1024((unsigned long long)nc, tmp, &totc);
        s_cpre[threadIdx.x] = (int)exc;
        if (threadIdx.x == 0) s_cpre[1024] = (int)totc;
        __syncthreads();
        const int ntile = min(1024, T - base), C = (int)totc;
        for (int cbase = 0; cbase < C; cbase += 1024) {
            const int c = cbase + threadIdx.x;
            int row0 = -1, cb = 0, ce = 0;
            bool has = false;
            if (c < C) {
                int lo = 0, hi = ntile - 1;  // tile of candidate c: largest tt with cpre[tt] <= c
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (s_cpre[mid] <= c) lo = mid;
                    else hi = mid - 1;
                }
                const int4 ti = s_tile[lo];
                const int ci = c - s_cpre[lo];
                row0 = (base + lo) * a.tr;
                const int bin_lo = ti.x / gb, bin_hi = ti.y / gb;
                const int k_lo = bin_lo / fb, f_lo = bin_lo % fb, k_hi = bin_hi / fb, f_hi = bin_hi % fb;
                if (ti.w) {  // tile inside one (k,f) bin: candidate = (column length, column f bucket)
                    const int kp = k_lo + ci / d1, delta = ci / d1;
                    if (kp <= a.kcap) {
                        const int amax = (d - delta) >> 1, bmax = amax + delta;
                        int fa, fz, ga, gz;
                        key_band(f_lo, f_lo, k_lo, kp, amax, bmax, fb, &fa, &fz);
                        key_band(ti.x % gb, ti.y % gb, k_lo, kp, amax, bmax, gb, &ga, &gz);
                        const int fp = fa + ci % d1;
                        if (fp <= fz && ga <= gz) {
                            cb = max(start3(a, kp * fb + fp, ga), row0);  // q > p >= row0
                            ce = start3(a, kp * fb + fp, gz + 1);
                            has = ce > cb;
                        }
                    }
                } else {  // tile spans bins: candidate = column length, full f band of the tile's rows, all g
                    const int kp = k_lo + ci;
                    int fa = 0x7fffffff, fz = -1;
                    for (int delta = 0; delta <= d; delta++) {
                        const int k = kp - delta;
                        if (k < k_lo || k > k_hi) continue;
                        const int rlo = (k == k_lo) ? f_lo : 0, rhi = (k == k_hi) ? f_hi : fb - 1;
                        const int amax = (d - delta) >> 1, bmax = amax + delta;
                        int x, y;
                        key_band(rlo, rhi, k, kp, amax, bmax, fb, &x, &y);
                        fa = min(fa, x);
                        fz = max(fz, y);
                    }
                    if (fa <= fz) {
                        cb = max(a.start[kp * fb + fa], row0);
                        ce = a.start[kp * fb + fz + 1];
                        has = ce > cb;
                    }
                }
            }
            // Ranges of one tile that touch are merged into ONE item: an item switch costs the prefilter a
            // dependent round trip.
            s_rng[threadIdx.x] = has ? make_int4(row0, cb, ce, 0) : make_int4(-1, 0, 0, 0);
            __syncthreads();
            bool head = has;
            if (has && threadIdx.x > 0) {
                const int4 pv = s_rng[threadIdx.x - 1];
                if (pv.x == row0 && pv.z == cb) head = false;  // continuation of the previous range
            }
            if (head) {
                int nx = threadIdx.x + 1;
                while (nx < 1024) {
                    const int4 nr = s_rng[nx];
                    if (nr.x != row0 || nr.y != ce) break;
                    ce = nr.z;
                    nx++;
                }
            }
            const unsigned nun = head ? ITEM_OVH_UNITS + (unsigned)((ce - (cb & ~(a.cb - 1)) + a.cb - 1) / a.cb) : 0u;
            unsigned long long tot_i, tot_u;
            const unsigned long long ex_i = block_excl_scan_1024(head ? 1ull : 0ull, tmp, &tot_i);
            const unsigned long long ex_u = block_excl_scan_1024((unsigned long long)nun, tmp, &tot_u);
            if (head) {
                const unsigned long long w = carry_i + ex_i;
                if (w < (unsigned long long)a.item_cap) a.items[w] = make_int4(row0, cb, ce, (int)(unsigned)(carry_u + ex_u));
            }
            carry_i += tot_i;  // the scan returns the same totals to all threads
            carry_u += tot_u;
        }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    PLAN_STAMP(3)
    const int n_items = (int)min(carry_i, (unsigned long long)0x7fffffff);
    const unsigned U = (unsigned)carry_u;
    if (threadIdx.x == 0) {
        if (n_items > a.item_cap || carry_u >= 0x7fffffffull) atomicOr(&a.ctr->err, ERR_WORKCAP);
        a.ctr->n_work = (unsigned)min(n_items, a.item_cap);
        a.ctr->n_units = U;
        a.ctr->pairs_filtered = (carry_u - (unsigned long long)ITEM_OVH_UNITS * carry_i) * (unsigned long long)a.tr * a.cb;
    }
    if (n_items > a.item_cap || U == 0 || carry_u >= 0x7fffffffull) return;
    // first item of every worker: worker vb starts at unit floor(vb * U / nvblocks).  The items' unit offsets are
    // staged in LDS (re-using s_rng) and every worker binary-searches them; beyond 4096 items walk the items.
    int *s_ust = reinterpret_cast<int *>(s_rng);
    if (n_items <= 4096) {
        for (int w = threadIdx.x; w < n_items; w += 1024) s_ust[w] = ld_agent(reinterpret_cast<const int *>(&a.items[w]) + 3);
        __syncthreads();
        for (int vb = threadIdx.x; vb < a.nvblocks; vb += 1024) {
            const unsigned u0 = (unsigned)(((unsigned long long)vb * U) / (unsigned)a.nvblocks);
            int lo = 0, hi = n_items - 1;  // largest w with ustart[w] <= u0
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if ((unsigned)s_ust[mid] <= u0) lo = mid;
                else hi = mid - 1;
            }
            a.blk_item[vb] = lo;
        }
    } else {
        for (int w = threadIdx.x; w < n_items; w += 1024) {
            const int *ip = reinterpret_cast<const int *>(&a.items[w]);  // written by this block: bypass L1
            const int4 it = make_int4(0, ld_agent(ip + 1), ld_agent(ip + 2), ld_agent(ip + 3));
            const unsigned ua = (unsigned)it.w;
            const unsigned ub = ua + ITEM_OVH_UNITS + (unsigned)((it.z - (it.y & ~(a.cb - 1)) + a.cb - 1) / a.cb);
            unsigned vb = (unsigned)(((unsigned long long)ua * a.nvblocks + U - 1) / U);
            while (vb < (unsigned)a.nvblocks && (unsigned)(((unsigned long long)vb * U) / a.nvblocks) < ub) {
                a.blk_item[vb] = w;
                vb++;
            }
        }
    }
    __syncthreads();
    PLAN_STAMP(4)
}

// ------------------------------------------------------------------------------------------------
// k_canon: one wave per row.  Bitonic sort of 64*E token ids held E per lane (index j = e*64 + lane:
// partners at distance < 64 are a cross-lane exchange, >= 64 an in-register one), duplicate ranks,
// signatures.  Writes the canonical row (sorted, repeats kept) to cols and the signatures at the row's
// length-sorted position.
// ------------------------------------------------------------------------------------------------
// cross-lane exchange x[lane ^ STRIDE] without the LDS crossbar: ds_bpermute costs ~26 cycles per
// wave-instruction per SIMD on gfx950 (tools/ubench/valu_rate.hip), a DPP move ~4, v_permlane*_swap ~8.
template <int STRIDE>
__device__ __forceinline__ uint32_t lane_xor(uint32_t x, int lane) {
    if constexpr (STRIDE == 1) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    } else if constexpr (STRIDE == 2) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    } else if constexpr (STRIDE == 4) {
        // banks 0,2 of each 16-lane row read lane+4 (row_shl:4), banks 1,3 read lane-4 (row_shr:4)
        int y = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xF, 0x5, false);
        return (uint32_t)__builtin_amdgcn_update_dpp(y, (int)x, 0x114, 0xF, 0xA, false);
    } else if constexpr (STRIDE == 8) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, false);  // row_ror:8
    } else if constexpr (STRIDE == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // {x0,x0,x2,x2}, {x1,x1,x3,x3}
        return (lane & 16) ? r[0] : r[1];
    } else {
        auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // {lo,lo}, {hi,hi}
        return (lane & 32) ? r[0] : r[1];
    }
}

// XOR of x over the 64 lanes, valid in lane 63 (DPP prefix within rows, then row broadcasts)
__device__ __forceinline__ uint32_t wave_xor_to_lane63(uint32_t x) {
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);  // row_shr:1
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);  // row_shr:8 -> lane 15 of each row
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true);  // row_bcast15 into rows 1,3
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true);  // row_bcast31 into rows 2,3
    return x;
}

template <int E, int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_step(uint32_t (&x)[E], int lane) {
    if constexpr (STRIDE >= 64) {
        constexpr int es = STRIDE >> 6;
#pragma unroll
        for (int e = 0; e < E; e++) {
            if ((e & es) == 0) {
                const int e2 = e | es;
                const bool asc = (((e * 64) & SIZE) == 0);
                uint32_t lo = min(x[e], x[e2]), hi = max(x[e], x[e2]);
                x[e] = asc ? lo : hi;
                x[e2] = asc ? hi : lo;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t y = lane_xor<STRIDE>(x[e], lane);
            const bool asc = ((((e * 64) + lane) & SIZE) == 0);
            const bool lower = ((lane & STRIDE) == 0);
            x[e] = (lower == asc) ? min(x[e], y) : max(x[e], y);
        }
    }
}

template <int E, int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_merge(uint32_t (&x)[E], int lane) {
    bitonic_step<E, SIZE, STRIDE>(x, lane);
    if constexpr (STRIDE > 1) bitonic_merge<E, SIZE, (STRIDE >> 1)>(x, lane);
}

template <int E, int SIZE>
__device__ __forceinline__ void bitonic_all(uint32_t (&x)[E], int lane) {
    if constexpr (SIZE > 2) bitonic_all<E, (SIZE >> 1)>(x, lane);
    bitonic_merge<E, SIZE, (SIZE >> 1)>(x, lane);
}

template <int E>
__device__ __forceinline__ void wave_bitonic(uint32_t (&x)[E], int lane) {
    bitonic_all<E, 64 * E>(x, lane);
}

template <int E>
__device__ __forceinline__ void wave_bitonic_bpermute(uint32_t (&x)[E], int lane) {
#pragma unroll
    for (int size = 2; size <= 64 * E; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64) {
                const int es = stride >> 6;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    if ((e & es) == 0) {
                        const int e2 = e | es;
                        const bool asc = (((e * 64) & size) == 0);
                        uint32_t lo = min(x[e], x[e2]), hi = max(x[e], x[e2]);
                        x[e] = asc ? lo : hi;
                        x[e2] = asc ? hi : lo;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    uint32_t y = __shfl_xor(x[e], stride);
                    const bool asc = ((((e * 64) + lane) & size) == 0);
                    const bool lower = ((lane & stride) == 0);
                    x[e] = (lower == asc) ? min(x[e], y) : max(x[e], y);
                }
            }
        }
    }
}

// the row's sort key (k, f, g) -> key3
struct RowKeyArgs {
    int *rowkey;
    KeyCfg key;
};

template <int E, int W1>
__device__ __forceinline__ void canon_row(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int k,
                                          int lane, uint32_t *lds_row, uint32_t *sig1_out, uint32_t *sig2_out,
                                          const RowKeyArgs &rk, int row) {
    uint32_t x[E];
    int f = 0, g = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        int j = e * 64 + lane;
        x[e] = j < k ? src[j] : 0xFFFFFFFFu;
        f += __popcll(__builtin_amdgcn_ballot_w64(j < k && fbit(x[e]) != 0u));
        g += __popcll(__builtin_amdgcn_ballot_w64(j < k && gbit(x[e]) != 0u));
    }
    const int key3 = key3_of(rk.key, k, f, g);
    wave_bitonic<E>(x, lane);
    // repeat rank r_j = number of equal predecessors (0 unless the multiset row repeats a token)
    uint32_t r[E];
    bool anydup = false;
#pragma unroll
    for (int e = 0; e < E; e++) {
        int j = e * 64 + lane;
        uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x[e], 0x138, 0xF, 0xF, false);  // wave_shr:1
        if (e > 0) {
            uint32_t tail = (uint32_t)__builtin_amdgcn_readlane((int)x[e - 1], 63);
            if (lane == 0) prev = tail;
        }
        bool eq = (j > 0) && (j < k) && (prev == x[e]);
        r[e] = eq ? 1u : 0u;
        anydup |= eq;
        if (j < k) dst[j] = x[e];
    }
    if (__any(anydup)) {  // rare: stage the sorted row in the wave's LDS slice and count runs
#pragma unroll
        for (int e = 0; e < E; e++) lds_row[e * 64 + lane] = x[e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int e = 0; e < E; e++) {
            int j = e * 64 + lane;
            uint32_t c = 0;
            if (j < k)
                while ((int)c < j && lds_row[j - 1 - (int)c] == x[e]) c++;
            r[e] = c;
        }
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t s1[W1], s2[SIG2_WORDS];
#pragma unroll
    for (int w = 0; w < W1; w++) s1[w] = 0;
#pragma unroll
    for (int w = 0; w < SIG2_WORDS; w++) s2[w] = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        int j = e * 64 + lane;
        if (j < k) {
            uint32_t b1 = hash1(x[e], r[e]) >> (32 - (5 + (W1 == 1 ? 0 : (W1 == 2 ? 1 : 2))));
            uint32_t b2 = hash2(x[e], r[e]) >> (32 - 6);
#pragma unroll
            for (int w = 0; w < W1; w++)
                if ((int)(b1 >> 5) == w) s1[w] ^= 1u << (b1 & 31);
#pragma unroll
            for (int w = 0; w < SIG2_WORDS; w++)
                if ((int)(b2 >> 5) == w) s2[w] ^= 1u << (b2 & 31);
        }
    }
#pragma unroll
    for (int w = 0; w < W1; w++) s1[w] = wave_xor_to_lane63(s1[w]);
#pragma unroll
    for (int w = 0; w < SIG2_WORDS; w++) s2[w] = wave_xor_to_lane63(s2[w]);
    if (lane == 63) {
#pragma unroll
        for (int w = 0; w < W1; w++) sig1_out[w] = s1[w];
#pragma unroll
        for (int w = 0; w < SIG2_WORDS; w++) sig2_out[w] = s2[w];
    }
    if (lane == 0) rk.rowkey[row] = key3;  // ranked by k_rowrank (a returning atomic here would stall the wave)
}

template <int W1>
__global__ __launch_bounds__(256) void k_canon(const int *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                int n, int kcap, RowKeyArgs rk, int *__restrict__ parent,
                                                uint32_t *__restrict__ cols, uint32_t *__restrict__ sigu1,
                                                uint32_t *__restrict__ sigu2, int *longrows, Counters *ctr) {
    __shared__ uint32_t lds_rows[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = gridDim.x * 4;
    int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    int b = indptr[i], e = indptr[i + 1];
    while (true) {
        // the next row's extent is fetched while this row is sorted (one dependent round trip fewer per row)
        const int inext = i + nwaves;
        int nb = 0, ne = 0;
        if (inext < n) {
            nb = indptr[inext];
            ne = indptr[inext + 1];
        }
        int k = e - b;
        if (k < 0 || k > kcap) {
            if (lane == 0) atomicOr(&ctr->err_rows, ERR_ROWLEN);
            k = k < 0 ? 0 : kcap;
        }
        if (lane == 0) parent[i] = i;
        const uint32_t *src = indices + b;
        uint32_t *dst = cols + b;
        uint32_t *o1 = sigu1 + (size_t)i * W1, *o2 = sigu2 + (size_t)i * SIG2_WORDS;  // row order; k_place sorts
        if (k <= 64) canon_row<1, W1>(src, dst, k, lane, lds_rows[wave], o1, o2, rk, i);
        else if (k <= 128) canon_row<2, W1>(src, dst, k, lane, lds_rows[wave], o1, o2, rk, i);
        else if (k <= 256) canon_row<4, W1>(src, dst, k, lane, lds_rows[wave], o1, o2, rk, i);
        else if (lane == 0) longrows[atomicAdd(&ctr->n_long, 1u)] = i;
        if (inext >= n) break;
        i = inext;
        b = nb;
        e = ne;
    }
}

// rows longer than 256 tokens: one block per row, rank sort (each element counts its predecessors);
// the row is staged in dynamic LDS when it fits (up to 32768 tokens = 128 KiB of the CU's 160 KiB).
template <int W1>
__global__ __launch_bounds__(256) void k_canon_long(const int *__restrict__ indptr,
                                                     const uint32_t *__restrict__ indices, int kcap, RowKeyArgs rk,
                                                     uint32_t *cols, uint32_t *sigu1, uint32_t *sigu2,
                                                     const int *__restrict__ longrows, const Counters *ctr,
                                                     int lds_cap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t row_lds[];
    __shared__ uint32_t s1[4], s2[SIG2_WORDS];
    __shared__ int sf, sg;
    const int nlong = (int)ctr->n_long;
    for (int li = blockIdx.x; li < nlong; li += gridDim.x) {
        int i = longrows[li];
        int b = indptr[i];
        int k = min(indptr[i + 1] - b, kcap);
        const uint32_t *src = indices + b;
        uint32_t *dst = cols + b;
        const bool staged = k <= lds_cap;
        if (threadIdx.x < 4) s1[threadIdx.x] = 0;
        if (threadIdx.x < SIG2_WORDS) s2[threadIdx.x] = 0;
        if (threadIdx.x == 0) sf = sg = 0;
        if (staged)
            for (int j = threadIdx.x; j < k; j += 256) row_lds[j] = src[j];
        __syncthreads();
        int fl = 0, gl = 0;
        for (int j = threadIdx.x; j < k; j += 256) {
            uint32_t x = staged ? row_lds[j] : src[j];
            fl += (int)fbit(x);
            gl += (int)gbit(x);
            int rank = 0;
            if (staged) {
                for (int m = 0; m < k; m++) {
                    uint32_t y = row_lds[m];
                    rank += (y < x) || (y == x && m < j);
                }
            } else {
                for (int m = 0; m < k; m++) {
                    uint32_t y = src[m];
                    rank += (y < x) || (y == x && m < j);
                }
            }
            __hip_atomic_store(dst + rank, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (fl) atomicAdd(&sf, fl);
        if (gl) atomicAdd(&sg, gl);
        __threadfence_block();
        __syncthreads();
        for (int j = threadIdx.x; j < k; j += 256) {
            uint32_t x = ld_agent_u(dst + j);
            uint32_t r = 0;
            while ((int)r < j && ld_agent_u(dst + j - 1 - (int)r) == x) r++;
            uint32_t b1 = hash1(x, r) >> (32 - (5 + (W1 == 1 ? 0 : (W1 == 2 ? 1 : 2))));
            uint32_t b2 = hash2(x, r) >> (32 - 6);
            atomicXor(&s1[b1 >> 5], 1u << (b1 & 31));
            atomicXor(&s2[b2 >> 5], 1u << (b2 & 31));
        }
        __syncthreads();
        if (threadIdx.x < W1) sigu1[(size_t)i * W1 + threadIdx.x] = s1[threadIdx.x];
        if (threadIdx.x < SIG2_WORDS) sigu2[(size_t)i * SIG2_WORDS + threadIdx.x] = s2[threadIdx.x];
        if (threadIdx.x == 0) rk.rowkey[i] = key3_of(rk.key, k, sf, sg);
        __syncthreads();
    }
}

// k_rowrank: histogram of the (k,f,g) keys and every row's rank inside its key: one returning atomic per row.
// (With the g sub-bins only ~13 rows share a counter at 100k rows, so the word-level serialisation that forced
// an LDS-aggregated version for (k,f) bins is gone.)
__global__ __launch_bounds__(256) void k_rowrank(const int *__restrict__ rowkey, int n, int *hist3,
                                                  int *__restrict__ rowrank) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) rowrank[i] = atomicAdd(&hist3[rowkey[i]], 1);
}

// k_place: counting-sort scatter.  Row i goes to sorted position start[bin] + sub3[key] + rank; its key, its
// length and its two signatures move with it (the prefilter reads signatures in sorted order, coalesced).
template <int W1>
__global__ __launch_bounds__(256) void k_place(const int *__restrict__ indptr, int n, int kcap, int gb,
                                                const int *__restrict__ start, const int *__restrict__ sub3,
                                                const int *__restrict__ rowkey, const int *__restrict__ rowrank,
                                                const uint32_t *__restrict__ sigu1, const uint32_t *__restrict__ sigu2,
                                                int *__restrict__ perm, int *__restrict__ ksorted,
                                                int *__restrict__ keysorted, uint32_t *__restrict__ sig1,
                                                uint32_t *__restrict__ sig2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int key = rowkey[i];
    const int p = start[key / gb] + sub3[key] + rowrank[i];
    const int k = indptr[i + 1] - indptr[i];
    perm[p] = i;
    ksorted[p] = k < 0 ? 0 : (k > kcap ? kcap : k);
    keysorted[p] = key;
#pragma unroll
    for (int w = 0; w < W1; w++) sig1[(size_t)p * W1 + w] = sigu1[(size_t)i * W1 + w];
#pragma unroll
    for (int w = 0; w < SIG2_WORDS; w++) sig2[(size_t)p * SIG2_WORDS + w] = sigu2[(size_t)i * SIG2_WORDS + w];
}

// ------------------------------------------------------------------------------------------------
// candidate queue: (row a, row b) pairs, 8 shards (block % 8) so no single counter word takes all atomics
// ------------------------------------------------------------------------------------------------
struct PairArgs {
    const int *indptr;
    const uint32_t *cols;
    const int *perm;
    const int *ksorted;
    const uint32_t *sig2;
    int *parent;
    int4 *cand;   // {row a, row b, cols offset a, cols offset b}
    int2 *candk;  // {k_a, k_b}
    int cand_cap_shard;
    int d;
    int n;
    unsigned long long *dbg_t;  // BFK_PF_DEBUG & 4: per-wave stamps {start, loop start, loop end, end, items}
    int dbg;  // BFK_PF_DEBUG experiments: 1 = no flush, 2 = no rescan (results wrong; timing only)
    Counters *ctr;
};

// queue record: everything k_verify needs in ONE round trip (row ids, row offsets, row lengths)
__device__ __forceinline__ void push_cand(const PairArgs &a, int shard, int idx, int ra, int rb) {
    if (idx >= a.cand_cap_shard) {
        a.ctr->overflow = 1;  // dropped: the host re-runs the item range in smaller slices
        return;
    }
    const int ba = a.indptr[ra], ea = a.indptr[ra + 1];
    const int bb = a.indptr[rb], eb = a.indptr[rb + 1];
    const size_t o = (size_t)shard * a.cand_cap_shard + idx;
    a.cand[o] = make_int4(ra, rb, ba, bb);
    a.candk[o] = make_int2(ea - ba, eb - bb);
}

// flush `cnt` queued (p,q) hits: filter, translate to row ids, append to the shard's global queue with
// one global atomic per wave; a full global queue raises the overflow flag (host re-runs in slices).
// Every load that depends only on (p,q) is issued up front, so a flush costs ~3 dependent round trips.
__device__ __forceinline__ void flush_hits(const PairArgs &a, const int2 *sbuf, int cnt, int shard) {
    const int lane = threadIdx.x & 63;
    for (int i0 = 0; i0 < cnt; i0 += 64) {  // called by ONE wave on its own queue
        const int i = i0 + lane;
        bool pass = false;
        int ra = 0, rb = 0;
        if (i < cnt) {
            const int2 pq = sbuf[i];
            const int p = pq.x, q = pq.y;
            if (q > p && q < a.n && p < a.n) {
                const int kp = a.ksorted[p], kq = a.ksorted[q];
                static_assert(SIG2_WORDS == 2, "second-level signature is read as one 64-bit word pair");
                const uint2 x = *reinterpret_cast<const uint2 *>(a.sig2 + (size_t)p * SIG2_WORDS);
                const uint2 y = *reinterpret_cast<const uint2 *>(a.sig2 + (size_t)q * SIG2_WORDS);
                ra = a.perm[p];
                rb = a.perm[q];
                const int c = __popc(x.x ^ y.x) + __popc(x.y ^ y.y);
                pass = (kq - kp <= a.d) && (c <= a.d);
            }
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
        if (mask == 0ull) continue;
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

// LDS queue full (very dense input): append the raw first-level hit to the global queue unfiltered;
// k_verify's exact merge decides.
__device__ __forceinline__ void push_raw(const PairArgs &a, int shard, int p, int q) {
    if (!(q > p && q < a.n && p < a.n)) return;
    const int idx = (int)atomicAdd(&a.ctr->ncand[shard], 1u);
    push_cand(a, shard, idx, a.perm[p], a.perm[q]);
}

template <int W>
__device__ __forceinline__ uint32_t sigdist(const uint32_t (&a)[W], const uint32_t *b) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < W; w++) c += __popc(a[w] ^ b[w]);
    return c;
}

// unit range of a (virtual) block: the U units are cut into nvb equal slices
__device__ __forceinline__ unsigned unit_cut(unsigned U, unsigned vb, unsigned nvb) {
    return (unsigned)(((unsigned long long)vb * U) / nvb);
}

template <int W, int R>
__global__ __launch_bounds__(256, 5) void k_prefilter(const uint32_t *__restrict__ sig1, const int4 *__restrict__ items,
                                                    const int *__restrict__ blk_item, int n, int vb0, int nvb,
                                                    int u_begin, int u_end, PairArgs pa) {
    constexpr int CC = 64 / W;          // columns per unit: one 256-byte chunk of signatures (64 dwords)
    constexpr int SB = 16 / W;          // columns per sub-batch (16 dwords): the hit-detection granularity
    constexpr int NG = 4, GC = SB / NG; // minima per group of GC columns
    __shared__ int2 sbuf[4][PF_LDS_QUEUE];
    __shared__ __attribute__((aligned(16))) uint32_t scol[4][64];
    const int lane = threadIdx.x & 63;
    // readfirstlane: tell the compiler the wave index is wave-uniform, so that everything derived from it
    // (unit slice, item descriptors) stays in SGPRs / scalar loads
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t d = (uint32_t)pa.d;
    const int qshard = (blockIdx.x * 4 + wave) & (CAND_SHARDS - 1);
    // Every WAVE is its own worker: a wave-tile is 64*R consecutive (k,f)-sorted rows (R per lane), a unit
    // is one chunk of CC columns against a wave-tile, and each wave takes an equal slice of the unit space.
    unsigned u0, u1;
    int w;
    const unsigned long long t_start = pa.dbg_t ? wall_clock64() : 0ull;
    if (blk_item) {  // normal run: slices of the whole unit space, first item precomputed by k_tiles
        const unsigned U = pa.ctr->n_units;
        const unsigned vb = (unsigned)(vb0 + blockIdx.x) * 4u + (unsigned)wave;
        u0 = unit_cut(U, vb, (unsigned)nvb);
        u1 = unit_cut(U, vb + 1u, (unsigned)nvb);
        if (u0 >= u1) return;
        w = blk_item[vb];
    } else {  // recovery run over [u_begin, u_end): locate the item by binary search
        const unsigned len = (unsigned)(u_end - u_begin);
        const unsigned vb = blockIdx.x * 4u + (unsigned)wave;
        u0 = (unsigned)u_begin + unit_cut(len, vb, gridDim.x * 4u);
        u1 = (unsigned)u_begin + unit_cut(len, vb + 1u, gridDim.x * 4u);
        if (u0 >= u1) return;
        int lo = 0, hi = (int)pa.ctr->n_work - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((unsigned)items[mid].w <= u0) lo = mid;
            else hi = mid - 1;
        }
        w = lo;
    }
    int2 *myq = sbuf[wave];
    uint32_t *mycol = scol[wave];
    int qn = 0;  // fill level of this wave's hit queue (wave-uniform)
    unsigned long long evaluated = 0;
    uint32_t rs[R][W];
    int cur_row0 = -1;
    const int n_items = (int)pa.ctr->n_work;
    int4 it = items[w];
    const unsigned long long t_loop = pa.dbg_t ? wall_clock64() : 0ull;
    int dbg_items = 0;
    while (true) {
        dbg_items++;
        const int4 nxt_it = items[min(w + 1, n_items - 1)];  // prefetch the next descriptor
        const int row0 = it.x, ctrue = it.y, cend = it.z;
        const int cal = ctrue & ~(CC - 1);
        const int nb = (cend - cal + CC - 1) / CC;
        // an item spans ITEM_OVH_UNITS + nb units: the fixed part prices the item switch for load balance
        const unsigned ub = (unsigned)it.w + ITEM_OVH_UNITS;
        const int first = u0 > ub ? (int)(u0 - ub) : 0;
        const int last = u1 > ub ? (int)min((unsigned)nb, u1 - ub) : 0;
        if (row0 != cur_row0) {
            cur_row0 = row0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int p = row0 + r * 64 + lane;
                const int pc = p < n ? p : n - 1;
#pragma unroll
                for (int x = 0; x < W; x++) rs[r][x] = sig1[(size_t)pc * W + x];
            }
        }
        const int qbeg = cal + first * CC, qend = cal + max(last, first) * CC;
        // Column signatures: one coalesced 256-byte load per chunk (the next chunk is in flight while this
        // one is compared), parked in the wave's LDS slice and re-read as wave-uniform ds_read_b128
        // broadcasts, so the v_xor operands are VGPRs (full rate; an SGPR operand halves it on gfx950).
        uint32_t v = sig1[(size_t)qbeg * W + lane];
        for (int q0 = qbeg; q0 < qend; q0 += CC) {
            const uint32_t vn = sig1[(size_t)(q0 + CC) * W + lane];  // array is padded past n
            mycol[lane] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int sb = 0; sb < CC / SB; sb++) {
                uint32_t cs[16];
#pragma unroll
                for (int x4 = 0; x4 < 4; x4++) {
                    const uint4 t = *reinterpret_cast<const uint4 *>(&mycol[sb * 16 + x4 * 4]);
                    cs[x4 * 4 + 0] = t.x;
                    cs[x4 * 4 + 1] = t.y;
                    cs[x4 * 4 + 2] = t.z;
                    cs[x4 * 4 + 3] = t.w;
                }
                uint32_t mg[NG];
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    mg[g] = 0xFFFFu;
#pragma unroll
                    for (int j = g * GC; j < (g + 1) * GC; j++) {
#pragma unroll
                        for (int r = 0; r < R; r++) mg[g] = min(mg[g], sigdist<W>(rs[r], &cs[j * W]));
                    }
                }
                const uint32_t m = min(min(mg[0], mg[1]), min(mg[2], mg[3]));
                if (!(pa.dbg & 2) && __builtin_amdgcn_ballot_w64(m <= d) != 0ull) {
                    // revisit only the column groups that hit; build a per-lane bit mask of the (column,row)
                    // hits (bit = j*R + r inside the sub-batch), then drain it in ONE place (small code).
                    // (Doing this on the scalar unit via v_readlane was tried: one SALU per CU makes it 5x slower.)
                    unsigned long long hm = 0ull;
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        if (__builtin_amdgcn_ballot_w64(mg[g] <= d) != 0ull) {
                            uint32_t bits = 0;
#pragma unroll
                            for (int j = (g + 1) * GC - 1; j >= g * GC; j--) {
#pragma unroll
                                for (int r = R - 1; r >= 0; r--)
                                    bits = (bits << 1) | (sigdist<W>(rs[r], &cs[j * W]) <= d ? 1u : 0u);
                            }
                            hm |= (unsigned long long)bits << (g * GC * R);
                        }
                    }
                    // the queue is private to the wave: slots come from a ballot prefix, the fill level is a
                    // wave-uniform register (no LDS atomic)
                    while (true) {
                        int b = -1;
                        if (hm) {
                            b = __ffsll((long long)hm) - 1;
                            hm &= hm - 1;
                            const int q = q0 + sb * SB + b / R;
                            if (q < ctrue || q >= cend) b = -1;  // alignment padding belongs to another item
                        }
                        const unsigned long long act = __builtin_amdgcn_ballot_w64(b >= 0);
                        if (act == 0ull && __builtin_amdgcn_ballot_w64(hm != 0ull) == 0ull) break;
                        const int na = __popcll(act);
                        if (qn + na > PF_LDS_QUEUE) {  // would overflow inside one chunk (very dense input): drain now
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            flush_hits(pa, myq, qn, qshard);
                            __builtin_amdgcn_wave_barrier();
                            qn = 0;
                        }
                        if (b >= 0)
                            myq[qn + __popcll(act & ((1ull << lane) - 1ull))] =
                                make_int2(row0 + (b % R) * 64 + lane, q0 + sb * SB + b / R);
                        qn += na;
                    }
                }
            }
            v = vn;
            // the hit queue is per wave, so it can be drained without a block barrier: keep it below half
            if (qn > PF_LDS_QUEUE / 2) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                flush_hits(pa, myq, qn, qshard);
                __builtin_amdgcn_wave_barrier();
                qn = 0;
            }
        }
        evaluated += (unsigned long long)max(last - first, 0);
        u0 = ub + (unsigned)nb;  // end of this item in unit space
        if (u0 >= u1 || w + 1 >= n_items) break;
        w++;
        it = nxt_it;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long t_main = pa.dbg_t ? wall_clock64() : 0ull;
    const int cnt = qn;
    if (cnt > 0 && !(pa.dbg & 1)) flush_hits(pa, myq, cnt, qshard);  // one flush per wave
    if (pa.dbg_t && lane == 0) {
        unsigned long long *o = pa.dbg_t + (size_t)(blockIdx.x * 4 + wave) * 8;
        o[0] = t_start;
        o[1] = t_loop;
        o[2] = t_main;
        o[3] = wall_clock64();
        o[4] = (unsigned long long)dbg_items;
        o[5] = (unsigned long long)cnt;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[6] = xcc;
        o[7] = evaluated;
    }
    // (no per-wave statistics atomics here: 5120 adds to one word serialise to ~50 us; k_tiles counts the units)
    (void)evaluated;
}

// ------------------------------------------------------------------------------------------------
// k_verify: exact multiset distance of every queued candidate, one 16-lane group per candidate (four
// per wave: the work is latency-bound — dependent memory round trips per pair — so the kernel is
// organised for pairs in flight, not lanes per pair).  The longer row (B) is staged in the group's LDS
// slice by a coalesced load; each lane takes elements of the shorter row (A, 16 per step), finds the
// lower bound in B by binary search and checks the (token, repeat-rank) match; the ballot's popcount
// over the group's 16 bits counts A elements without a partner.  |A delta B| = kA + kB - 2*matches.
// A pair farther apart than d is marked (row a = -1); k_union then hooks the surviving edges, one per lane.
// ------------------------------------------------------------------------------------------------
// the shard queues seen as one index space: prefix of min(ncand[s], cap), built once per block in LDS
__device__ __forceinline__ void shard_prefix(const PairArgs &pa, int *pre /*CAND_SHARDS+1, LDS*/) {
    if (threadIdx.x < 64) {
        static_assert(CAND_SHARDS == 64, "one shard counter per lane of the first wave");
        const int c = (int)min(pa.ctr->ncand[threadIdx.x], (unsigned)pa.cand_cap_shard);
        int inc = c;
        for (int s = 1; s < 64; s <<= 1) {
            const int y = __shfl_up(inc, s);
            if ((int)threadIdx.x >= s) inc += y;
        }
        pre[threadIdx.x + 1] = inc;
        if (threadIdx.x == 0) pre[0] = 0;
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

__global__ __launch_bounds__(256) void k_verify(PairArgs pa) {
    __shared__ uint32_t sB[16][VERIFY_LDS_ROW];
    const int lane = threadIdx.x & 63, l16 = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;           // 0..15 in the block
    const int gsh = (lane >> 4) * 16;           // bit offset of this group inside the wave ballot
    const int gg = blockIdx.x * 16 + grp, ng = gridDim.x * 16;
    uint32_t *myB = sB[grp];
    __shared__ int spre[CAND_SHARDS + 1];
    shard_prefix(pa, spre);
    const int total = spre[CAND_SHARDS];
    for (int c = gg; c < total; c += ng) {
        const size_t slot = shard_slot(spre, pa, c);
        const int4 rec = pa.cand[slot];
        const int2 kk = pa.candk[slot];
        int ba = rec.z, ka = kk.x, bb = rec.w, kb = kk.y;
        if (ka > kb) {  // A = shorter row
            int t = ba; ba = bb; bb = t;
            t = ka; ka = kb; kb = t;
        }
        const uint32_t *A = pa.cols + ba, *B = pa.cols + bb;
        const int allowed = (pa.d - (kb - ka)) >> 1;  // A elements allowed to stay unmatched
        bool ok = (kb - ka) <= pa.d;
        const bool staged = kb <= VERIFY_LDS_ROW;
        int miss = 0;
        for (int i0 = 0; i0 < ka; i0 += 64) {  // 64 A elements per step: 4 per lane, all loads in flight at once
            uint32_t x[4], xp[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = i0 + e * 16 + l16;
                x[e] = (ok && i < ka) ? A[i] : 0u;
                xp[e] = (ok && i < ka && i > 0) ? A[i - 1] : 0xFFFFFFFFu;
            }
            if (i0 == 0) {
                if (ok && staged)
                    for (int j = l16; j < kb; j += 16) myB[j] = B[j];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = i0 + e * 16 + l16;
                const bool valid = ok && i < ka;
                bool found = false;
                if (valid) {
                    int r = 0;
                    if (xp[e] == x[e]) {  // repeated token: rank inside its run
                        r = 1;
                        while (r < i && A[i - 1 - r] == x[e]) r++;
                    }
                    int lo = 0, hi = kb;  // lower bound of x in B
                    if (staged) {
                        while (lo < hi) {
                            int mid = (lo + hi) >> 1;
                            if (myB[mid] < x[e]) lo = mid + 1;
                            else hi = mid;
                        }
                        found = (lo + r < kb) && (myB[lo + r] == x[e]);
                    } else {
                        while (lo < hi) {
                            int mid = (lo + hi) >> 1;
                            if (B[mid] < x[e]) lo = mid + 1;
                            else hi = mid;
                        }
                        found = (lo + r < kb) && (B[lo + r] == x[e]);
                    }
                }
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(valid && !found);
                miss += __popc((unsigned)((bal >> gsh) & 0xFFFFull));
            }
            if (miss > allowed) ok = false;
        }
        __builtin_amdgcn_wave_barrier();
        if (!ok && l16 == 0) pa.cand[slot].x = -1;
    }
}

// k_union: one verified edge per lane -> lock-free hook (the dependent find/CAS chains of all edges overlap)
__global__ __launch_bounds__(256) void k_union(PairArgs pa, int2 *edges, int edge_cap) {
    const int gt = blockIdx.x * 256 + threadIdx.x, nt = gridDim.x * 256;
    unsigned int my_edges = 0;
    __shared__ int spre[CAND_SHARDS + 1];
    shard_prefix(pa, spre);
    const int total = spre[CAND_SHARDS];
    for (int c = gt; c < total; c += nt) {
        const int4 rec = pa.cand[shard_slot(spre, pa, c)];
        if (rec.x < 0) continue;
        uf_union(pa.parent, rec.x, rec.y);
        my_edges++;
        if (edges) {
            unsigned long long e = atomicAdd(&pa.ctr->n_edges_cap, 1ull);
            if (e < (unsigned long long)edge_cap) edges[e] = make_int2(min(rec.x, rec.y), max(rec.x, rec.y));
        }
    }
    for (int sft = 32; sft > 0; sft >>= 1) my_edges += __shfl_xor(my_edges, sft);
    __shared__ unsigned int blk_edges;
    if (threadIdx.x == 0) blk_edges = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && my_edges) atomicAdd(&blk_edges, my_edges);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blk_edges) atomicAdd(&pa.ctr->n_edges, (unsigned long long)blk_edges);
        if (blockIdx.x == 0 && total) atomicAdd(&pa.ctr->n_cand_total, (unsigned long long)total);
    }
}

// ------------------------------------------------------------------------------------------------
// k_flatten: labels[i] = root(i).  k_merge: unite (i, gathered[g][i]).  k_changed: fix-point flag.
// ------------------------------------------------------------------------------------------------
// (no hooks run concurrently with this kernel, so plain cached loads and no compression stores)
__global__ void k_flatten(const int *__restrict__ parent, int n, int *__restrict__ labels) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cur = parent[i], next;
    while (cur > (next = parent[cur])) cur = next;
    labels[i] = cur;
}

__global__ void k_merge(int *parent, int n, const int *__restrict__ gathered, int n_parts, Counters *ctr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int g = 0; g < n_parts; g++) {
        int l = gathered[(size_t)g * n + i];
        if (l < 0 || l >= n) {
            atomicOr(&ctr->err, ERR_LABEL);
            continue;
        }
        if (l != i) uf_union(parent, i, l);
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
    hipLaunchKernelGGL(k_maxlen, dim3((n + 255) / 256), dim3(256), 0, st, indptr, n, out);
    LAUNCH_CHECK();
    return 0;
}

static PairArgs make_pair_args(const Plan &pl) {
    PairArgs pa;
    pa.indptr = pl.indptr;
    pa.cols = pl.cols;
    pa.perm = pl.perm;
    pa.ksorted = pl.ksorted;
    pa.sig2 = pl.sig2;
    pa.n = pl.n;
    pa.dbg = pl.dbg;
    pa.dbg_t = pl.dbg_t;
    pa.parent = pl.parent;
    pa.cand = pl.cand;
    pa.candk = pl.candk;
    pa.cand_cap_shard = pl.cand_cap_shard;
    pa.d = pl.d;
    pa.ctr = pl.ctr;
    return pa;
}

// prefilter + verify + union.  Normal run (u_end < 0): every (virtual) block takes its precomputed slice of
// the unit space.  Recovery run: the unit range [u_begin, u_end) only.
int launch_pairs(const Plan &pl, int u_begin, int u_end, hipStream_t st, hipEvent_t *ev) {
    const int n = pl.n;
    PairArgs pa = make_pair_args(pl);
    const int *blk = u_end < 0 ? pl.blk_item : nullptr;
    const int vb0 = pl.shard * pl.pf_grid, nvb = pl.n_shards * pl.pf_grid * 4;  // workers are waves
    switch (pl.w1) {
        case 1:
            hipLaunchKernelGGL((k_prefilter<1, PF_ROWS_W1>), dim3(pl.pf_grid), dim3(256), 0, st, pl.sig1, pl.items, blk, n,
                               vb0, nvb, u_begin, u_end, pa);
            break;
        case 2:
            hipLaunchKernelGGL((k_prefilter<2, PF_ROWS_W2>), dim3(pl.pf_grid), dim3(256), 0, st, pl.sig1, pl.items, blk, n,
                               vb0, nvb, u_begin, u_end, pa);
            break;
        default:
            hipLaunchKernelGGL((k_prefilter<4, PF_ROWS_W4>), dim3(pl.pf_grid), dim3(256), 0, st, pl.sig1, pl.items, blk, n,
                               vb0, nvb, u_begin, u_end, pa);
            break;
    }
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[2], st);
    hipLaunchKernelGGL(k_verify, dim3(pl.verify_grid), dim3(256), 0, st, pa);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_union, dim3(pl.union_grid), dim3(256), 0, st, pa, pl.edges, pl.edge_cap);
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[3], st);
    return 0;
}

int launch_flatten(const Plan &pl, hipStream_t st, hipEvent_t *ev) {
    hipLaunchKernelGGL(k_flatten, dim3((pl.n + 255) / 256), dim3(256), 0, st, pl.parent, pl.n, pl.labels);
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[4], st);
    return 0;
}

int launch_pipeline(const Plan &pl, hipStream_t st, hipEvent_t *ev /*5 or NULL*/) {
    const int n = pl.n;
    if (ev) (void)hipEventRecord(ev[0], st);
    RowKeyArgs rk;
    rk.rowkey = pl.rowkey;
    rk.key.fb = pl.fb;
    rk.key.gb = pl.gb;
    const int canon_blocks = min((n + 3) / 4, 256 * 16);
    const int lds_cap = pl.long_lds_cap;
    PlanArgs pa;
    pa.hist3 = pl.hist3;
    pa.sub3 = pl.sub3;
    pa.start = pl.start;
    pa.keysorted = pl.keysorted;
    pa.items = pl.items;
    pa.blk_item = pl.blk_item;
    pa.ctr = pl.ctr;
    pa.key = rk.key;
    pa.n = n;
    pa.kcap = pl.kcap;
    pa.d = pl.d;
    pa.tr = pl.tr;
    pa.cb = pl.cb;
    pa.nvblocks = pl.n_shards * pl.pf_grid * 4;
    pa.item_cap = pl.item_cap;
    switch (pl.w1) {
#define PREP_CASE(W)                                                                                                      \
    case W:                                                                                                               \
        hipLaunchKernelGGL(k_canon<W>, dim3(canon_blocks), dim3(256), 0, st, pl.indptr, pl.indices, n, pl.kcap, rk,        \
                           pl.parent, pl.cols, pl.sigu1, pl.sigu2, pl.longrows, pl.ctr);                                   \
        if (pl.kcap > 256)                                                                                                \
            hipLaunchKernelGGL(k_canon_long<W>, dim3(min(n, 1024)), dim3(256), (size_t)lds_cap * 4, st, pl.indptr,         \
                               pl.indices, pl.kcap, rk, pl.cols, pl.sigu1, pl.sigu2, pl.longrows, pl.ctr, lds_cap);        \
        hipLaunchKernelGGL(k_rowrank, dim3((n + 255) / 256), dim3(256), 0, st, pl.rowkey, n, pl.hist3, pl.rowrank);        \
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, pa);                                                       \
        hipLaunchKernelGGL(k_place<W>, dim3((n + 255) / 256), dim3(256), 0, st, pl.indptr, n, pl.kcap, pl.gb, pl.start,    \
                           pl.sub3, pl.rowkey, pl.rowrank, pl.sigu1, pl.sigu2, pl.perm, pl.ksorted, pl.keysorted, pl.sig1, \
                           pl.sig2);                                                                                      \
        hipLaunchKernelGGL(k_tiles, dim3(1), dim3(1024), 0, st, pa);                                                      \
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
    if (int e = launch_pairs(pl, 0, -1, st, ev)) return e;
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
    hipLaunchKernelGGL(k_flatten, dim3((n + 255) / 256), dim3(256), 0, st, (const int *)parent, n, labels);
    LAUNCH_CHECK();
    return 0;
}

int launch_merge(int *parent, int n, const int *gathered, int n_parts, int *labels, int *changed, Counters *ctr,
                 hipStream_t st) {
    if (n <= 0) return 0;
    dim3 g((n + 255) / 256), b(256);
    hipLaunchKernelGGL(k_merge, g, b, 0, st, parent, n, gathered, n_parts, ctr);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_flatten, g, b, 0, st, parent, n, labels);
    LAUNCH_CHECK();
    if (changed) {
        hipLaunchKernelGGL(k_changed, g, b, 0, st, (const int *)labels, gathered, n, changed);
        LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace bfk
