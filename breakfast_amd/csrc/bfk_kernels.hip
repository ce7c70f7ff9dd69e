// bfk_kernels.hip — HIP kernels for gfx950 (MI355X, CDNA4): the breakfast clustering hot path.
//
// Pipeline (one stream, no host round trip between kernels):
//   k_hist      row lengths k_i -> histogram (LDS-aggregated), parent[i] = i
//   k_plan      scan -> start[k] (rows sorted by length), band work list of (row tile x column chunk) items
//   k_scatter   counting-sort scatter: perm / pos / ksorted
//   k_canon     one wave per row: bitonic sort of the token ids in registers, duplicate ranks,
//               two XOR-parity signatures (sum_duplicates + prefilter keys), written in length order
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
// k_hist: histogram of row lengths.  Global atomics on ~50 hot bins would serialise (one word takes
// ~90 atomics/us), so each 1024-row block aggregates in LDS and flushes one atomic per non-empty bin.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_hist(const int *__restrict__ indptr, int n, int kcap, int *hist, int *parent,
                                                Counters *ctr) {
    __shared__ int lh[HIST_LDS_BINS];
    for (int b = threadIdx.x; b < HIST_LDS_BINS; b += 1024) lh[b] = 0;
    __syncthreads();
    int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) {
        int k = indptr[i + 1] - indptr[i];
        parent[i] = i;
        if (k < 0 || k > kcap) {
            atomicOr(&ctr->err, ERR_ROWLEN);
            k = k < 0 ? 0 : kcap;
        }
        if (k < HIST_LDS_BINS) atomicAdd(&lh[k], 1);
        else atomicAdd(&hist[k], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < HIST_LDS_BINS && b <= kcap; b += 1024) {
        int c = lh[b];
        if (c) atomicAdd(&hist[b], c);
    }
}

// block-wide exclusive scan of one 1024-chunk held one value per thread; returns exclusive prefix,
// *total gets the chunk sum.  tmp: 32 ints of LDS.
__device__ __forceinline__ int block_excl_scan_1024(int v, int *tmp, int *total) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
    for (int s = 1; s < 64; s <<= 1) {
        int y = __shfl_up(inc, s);
        if (lane >= s) inc += y;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int w = lane < 16 ? tmp[lane] : 0;
        int winc = w;
        for (int s = 1; s < 16; s <<= 1) {
            int y = __shfl_up(winc, s);
            if (lane >= s) winc += y;
        }
        if (lane < 16) tmp[16 + lane] = winc - w;
        if (lane == 15) tmp[32] = winc;
    }
    __syncthreads();
    int res = inc - v + tmp[16 + wave];
    *total = tmp[32];
    __syncthreads();
    return res;
}

// ------------------------------------------------------------------------------------------------
// k_plan (one block): start[] = exclusive scan of hist (bins 0..kcap), padded with N up to kcap+1+d;
// cursor[] = copy for the scatter; then the band work list: for every row tile the column range
// [tile row0, first position whose length exceeds k_last + d) cut into chunks of TC columns.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_plan(const int *__restrict__ hist, int n, int kcap, int d, int *start,
                                                int *cursor, int tr, int tc, int4 *work, int work_cap, Counters *ctr) {
    __shared__ int tmp[40];
    __shared__ int carry_s;
    const int nb = kcap + 1;  // bins 0..kcap
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        int b = base + threadIdx.x;
        int v = b < nb ? hist[b] : 0;
        int tot;
        int ex = block_excl_scan_1024(v, tmp, &tot);
        int carry = carry_s;
        if (b < nb) {
            start[b] = carry + ex;
            cursor[b] = carry + ex;
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    for (int b = nb + threadIdx.x; b <= nb + d + 1; b += 1024) start[b] = n;
    // in-band unordered pairs: sum_k c_k(c_k-1)/2 + sum_{k<k'<=k+d} c_k c_k'
    {
        unsigned long long acc = 0;
        for (int k = threadIdx.x; k < nb; k += 1024) {
            unsigned long long c = (unsigned long long)hist[k];
            if (!c) continue;
            unsigned long long s = 0;
            for (int k2 = k + 1; k2 <= k + d && k2 < nb; k2++) s += (unsigned long long)hist[k2];
            acc += c * (c - 1) / 2 + c * s;
        }
        for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
        if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&ctr->pairs_in_band, acc);
    }
    __threadfence_block();
    __syncthreads();
    const int T = (n + tr - 1) / tr;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 1024) {
        int t = base + threadIdx.x;
        int nch = 0, cend = 0, row0 = 0;
        if (t < T) {
            row0 = t * tr;
            int plast = min(n, row0 + tr) - 1;
            // bin holding position plast: largest k with start[k] <= plast
            int lo = 0, hi = nb - 1;
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (start[mid] <= plast) lo = mid;
                else hi = mid - 1;
            }
            cend = start[min(lo + d, kcap) + 1];
            nch = (cend - row0 + tc - 1) / tc;
        }
        int tot;
        int ex = block_excl_scan_1024(nch, tmp, &tot);
        int off = carry_s + ex;
        if (t < T) {
            for (int c = 0; c < nch; c++) {
                int w = off + c;
                if (w < work_cap) work[w] = make_int4(row0, row0 + c * tc, min(cend, row0 + (c + 1) * tc), 0);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int nw = carry_s;
        if (nw > work_cap) {
            atomicOr(&ctr->err, ERR_WORKCAP);
            nw = work_cap;
        }
        ctr->n_work = nw;
    }
}

// ------------------------------------------------------------------------------------------------
// k_scatter: counting-sort scatter by length (order inside a length bin is arbitrary; labels are
// canonical in row-index space so the permutation is unobservable).  Ranks come from LDS atomics,
// one global atomic per (block, non-empty bin) reserves the range.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scatter(const int *__restrict__ indptr, int n, int kcap, int *cursor,
                                                   int *perm, int *pos, int *ksorted) {
    __shared__ int lh[HIST_LDS_BINS];
    for (int b = threadIdx.x; b < HIST_LDS_BINS; b += 1024) lh[b] = 0;
    __syncthreads();
    int i = blockIdx.x * 1024 + threadIdx.x;
    int k = 0, lr = 0;
    bool in_lds = false;
    if (i < n) {
        k = indptr[i + 1] - indptr[i];
        k = k < 0 ? 0 : (k > kcap ? kcap : k);
        in_lds = k < HIST_LDS_BINS;
        if (in_lds) lr = atomicAdd(&lh[k], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < HIST_LDS_BINS && b <= kcap; b += 1024) {
        int c = lh[b];
        if (c) lh[b] = atomicAdd(&cursor[b], c);
    }
    __syncthreads();
    if (i < n) {
        int p = in_lds ? lh[k] + lr : atomicAdd(&cursor[k], 1);
        perm[p] = i;
        pos[i] = p;
        ksorted[p] = k;
    }
}

// ------------------------------------------------------------------------------------------------
// k_canon: one wave per row.  Bitonic sort of 64*E token ids held E per lane (index j = e*64 + lane:
// partners at distance < 64 are a cross-lane exchange, >= 64 an in-register one), duplicate ranks,
// signatures.  Writes the canonical row (sorted, repeats kept) to cols and the signatures at the row's
// length-sorted position.
// ------------------------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void wave_bitonic(uint32_t (&x)[E], int lane) {
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

template <int E, int W1>
__device__ __forceinline__ void canon_row(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int k,
                                          int lane, uint32_t *lds_row, uint32_t *sig1_out, uint32_t *sig2_out) {
    uint32_t x[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        int j = e * 64 + lane;
        x[e] = j < k ? src[j] : 0xFFFFFFFFu;
    }
    wave_bitonic<E>(x, lane);
    // repeat rank r_j = number of equal predecessors (0 unless the multiset row repeats a token)
    uint32_t r[E];
    bool anydup = false;
#pragma unroll
    for (int e = 0; e < E; e++) {
        int j = e * 64 + lane;
        uint32_t prev = __shfl_up(x[e], 1);
        if (e > 0) {
            uint32_t tail = __shfl(x[e - 1], 63);
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
            uint32_t b2 = hash2(x[e], r[e]) >> (32 - 7);
#pragma unroll
            for (int w = 0; w < W1; w++)
                if ((int)(b1 >> 5) == w) s1[w] ^= 1u << (b1 & 31);
#pragma unroll
            for (int w = 0; w < SIG2_WORDS; w++)
                if ((int)(b2 >> 5) == w) s2[w] ^= 1u << (b2 & 31);
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
#pragma unroll
        for (int w = 0; w < W1; w++) s1[w] ^= __shfl_xor(s1[w], s);
#pragma unroll
        for (int w = 0; w < SIG2_WORDS; w++) s2[w] ^= __shfl_xor(s2[w], s);
    }
    if (lane == 0) {
#pragma unroll
        for (int w = 0; w < W1; w++) sig1_out[w] = s1[w];
#pragma unroll
        for (int w = 0; w < SIG2_WORDS; w++) sig2_out[w] = s2[w];
    }
}

template <int W1>
__global__ __launch_bounds__(256) void k_canon(const int *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                int n, const int *__restrict__ pos, uint32_t *__restrict__ cols,
                                                uint32_t *__restrict__ sig1, uint32_t *__restrict__ sig2,
                                                int *longrows, Counters *ctr) {
    __shared__ uint32_t lds_rows[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = gridDim.x * 4;
    for (int i = blockIdx.x * 4 + wave; i < n; i += nwaves) {
        int b = indptr[i];
        int k = indptr[i + 1] - b;
        if (k < 0) k = 0;
        int p = pos[i];
        const uint32_t *src = indices + b;
        uint32_t *dst = cols + b;
        uint32_t *o1 = sig1 + (size_t)p * W1, *o2 = sig2 + (size_t)p * SIG2_WORDS;
        if (k <= 64) canon_row<1, W1>(src, dst, k, lane, lds_rows[wave], o1, o2);
        else if (k <= 128) canon_row<2, W1>(src, dst, k, lane, lds_rows[wave], o1, o2);
        else if (k <= 256) canon_row<4, W1>(src, dst, k, lane, lds_rows[wave], o1, o2);
        else if (lane == 0) longrows[atomicAdd(&ctr->n_long, 1u)] = i;
    }
}

// rows longer than 256 tokens: one block per row, rank sort (each element counts its predecessors);
// the row is staged in dynamic LDS when it fits (up to 32768 tokens = 128 KiB of the CU's 160 KiB).
template <int W1>
__global__ __launch_bounds__(256) void k_canon_long(const int *__restrict__ indptr,
                                                     const uint32_t *__restrict__ indices, const int *__restrict__ pos,
                                                     uint32_t *cols, uint32_t *sig1, uint32_t *sig2,
                                                     const int *__restrict__ longrows, const Counters *ctr,
                                                     int lds_cap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t row_lds[];
    __shared__ uint32_t s1[4], s2[SIG2_WORDS];
    const int nlong = (int)ctr->n_long;
    for (int li = blockIdx.x; li < nlong; li += gridDim.x) {
        int i = longrows[li];
        int b = indptr[i];
        int k = indptr[i + 1] - b;
        const uint32_t *src = indices + b;
        uint32_t *dst = cols + b;
        const bool staged = k <= lds_cap;
        if (threadIdx.x < 4) s1[threadIdx.x] = 0;
        if (threadIdx.x < SIG2_WORDS) s2[threadIdx.x] = 0;
        if (staged)
            for (int j = threadIdx.x; j < k; j += 256) row_lds[j] = src[j];
        __syncthreads();
        for (int j = threadIdx.x; j < k; j += 256) {
            uint32_t x = staged ? row_lds[j] : src[j];
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
        __threadfence_block();
        __syncthreads();
        for (int j = threadIdx.x; j < k; j += 256) {
            uint32_t x = ld_agent_u(dst + j);
            uint32_t r = 0;
            while ((int)r < j && ld_agent_u(dst + j - 1 - (int)r) == x) r++;
            uint32_t b1 = hash1(x, r) >> (32 - (5 + (W1 == 1 ? 0 : (W1 == 2 ? 1 : 2))));
            uint32_t b2 = hash2(x, r) >> (32 - 7);
            atomicXor(&s1[b1 >> 5], 1u << (b1 & 31));
            atomicXor(&s2[b2 >> 5], 1u << (b2 & 31));
        }
        __syncthreads();
        int p = pos[i];
        if (threadIdx.x < W1) sig1[(size_t)p * W1 + threadIdx.x] = s1[threadIdx.x];
        if (threadIdx.x < SIG2_WORDS) sig2[(size_t)p * SIG2_WORDS + threadIdx.x] = s2[threadIdx.x];
        __syncthreads();
    }
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

// Second stage of the filter, one queued first-level hit per lane (run when a block flushes its LDS
// queue, so the dependent global loads of 64 hits overlap instead of stalling the scan loop):
// order / bounds / exact length band, then the 128-bit second-level signature.
__device__ __forceinline__ bool second_level(const PairArgs &a, int p, int q) {
    if (!(q > p && q < a.n && p < a.n)) return false;
    if (a.ksorted[q] - a.ksorted[p] > a.d) return false;
    const uint4 x = *reinterpret_cast<const uint4 *>(a.sig2 + (size_t)p * SIG2_WORDS);
    const uint4 y = *reinterpret_cast<const uint4 *>(a.sig2 + (size_t)q * SIG2_WORDS);
    int c = __popc(x.x ^ y.x) + __popc(x.y ^ y.y) + __popc(x.z ^ y.z) + __popc(x.w ^ y.w);
    return c <= a.d;
}

// flush `cnt` queued (p,q) hits: filter, translate to row ids, append to the shard's global queue with
// one global atomic per wave; a full global queue raises the overflow flag (host re-runs in slices).
__device__ __forceinline__ void flush_hits(const PairArgs &a, const int2 *sbuf, int cnt, int shard) {
    const int lane = threadIdx.x & 63;
    for (int i0 = (threadIdx.x >> 6) * 64; i0 < cnt; i0 += 256) {
        const int i = i0 + lane;
        int2 pq = make_int2(0, 0);
        bool pass = false;
        if (i < cnt) {
            pq = sbuf[i];
            pass = second_level(a, pq.x, pq.y);
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
        if (mask == 0ull) continue;
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(&a.ctr->ncand[shard], (unsigned)__popcll(mask));
        base = __shfl(base, 0);
        if (pass) {
            const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
            push_cand(a, shard, idx, a.perm[pq.x], a.perm[pq.y]);
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

template <int W, int R>
__global__ __launch_bounds__(256) void k_prefilter(const uint32_t *__restrict__ sig1,
                                                    const int4 *__restrict__ work, int n, int shard0, int nshards,
                                                    int w_begin, int w_end, PairArgs pa) {
    constexpr int CB = (W == 1) ? 16 : (W == 2 ? 8 : 4);  // columns per batch: 16 SGPRs of signature
    constexpr int NG = 4, GC = CB / NG;                    // minima per group of GC columns
    __shared__ int2 sbuf[PF_LDS_QUEUE];
    __shared__ int scount;
    const int tid = threadIdx.x;
    const uint32_t d = (uint32_t)pa.d;
    const int n_work = min((int)pa.ctr->n_work, w_end);
    const int qshard = blockIdx.x & (CAND_SHARDS - 1);
    if (tid == 0) scount = 0;
    __syncthreads();
    unsigned long long evaluated = 0;
    for (int w = w_begin + blockIdx.x * nshards + shard0; w < n_work; w += gridDim.x * nshards) {
        const int4 it = work[w];
        const int row0 = it.x, cbeg = it.y, cend = it.z;
        uint32_t rs[R][W];
        int prow[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            int p = row0 + r * 256 + tid;
            prow[r] = p;
            int pc = p < n ? p : n - 1;
#pragma unroll
            for (int x = 0; x < W; x++) rs[r][x] = sig1[(size_t)pc * W + x];
        }
        uint32_t cs[CB * W];
#pragma unroll
        for (int x = 0; x < CB * W; x++) cs[x] = sig1[(size_t)cbeg * W + x];  // wave-uniform -> s_load
        for (int q0 = cbeg; q0 < cend; q0 += CB) {
            uint32_t nx[CB * W];
#pragma unroll
            for (int x = 0; x < CB * W; x++) nx[x] = sig1[(size_t)(q0 + CB) * W + x];  // prefetch (array is padded)
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
            if (__builtin_amdgcn_ballot_w64(m <= d) != 0ull) {
                // revisit only the column groups that hit; build a per-lane bit mask of the (column,row)
                // hits (bit = j*R + r inside the batch), then drain it in ONE place (small code)
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
                while (hm) {
                    const int b = __ffsll((long long)hm) - 1;
                    hm &= hm - 1;
                    const int p = row0 + (b % R) * 256 + tid, q = q0 + b / R;
                    const int slot = atomicAdd(&scount, 1);
                    if (slot < PF_LDS_QUEUE) sbuf[slot] = make_int2(p, q);
                    else push_raw(pa, qshard, p, q);
                }
            }
#pragma unroll
            for (int x = 0; x < CB * W; x++) cs[x] = nx[x];
        }
        evaluated += (unsigned long long)(cend - cbeg);
        __syncthreads();
        const int cnt = min(scount, PF_LDS_QUEUE);
        if (cnt > 0) {  // block-uniform
            flush_hits(pa, sbuf, cnt, qshard);
            __syncthreads();
            if (tid == 0) scount = 0;
            __syncthreads();
        }
    }
    if (tid == 0 && evaluated) atomicAdd(&pa.ctr->pairs_filtered, evaluated * (unsigned long long)(256 * R));
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
// the 8 shard queues seen as one index space: prefix of min(ncand[s], cap)
struct ShardMap {
    int pre[CAND_SHARDS + 1];
};
__device__ __forceinline__ ShardMap shard_map(const PairArgs &pa) {
    const uint4 lo = *reinterpret_cast<const uint4 *>(&pa.ctr->ncand[0]);
    const uint4 hi = *reinterpret_cast<const uint4 *>(&pa.ctr->ncand[4]);
    const unsigned c[CAND_SHARDS] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    ShardMap m;
    m.pre[0] = 0;
#pragma unroll
    for (int s = 0; s < CAND_SHARDS; s++) m.pre[s + 1] = m.pre[s] + (int)min(c[s], (unsigned)pa.cand_cap_shard);
    return m;
}
__device__ __forceinline__ size_t shard_slot(const ShardMap &m, const PairArgs &pa, int c) {
    int s = 0;
#pragma unroll
    for (int t = 1; t < CAND_SHARDS; t++) s += (c >= m.pre[t]) ? 1 : 0;
    int base = m.pre[0];
#pragma unroll
    for (int t = 1; t < CAND_SHARDS; t++) base = (c >= m.pre[t]) ? m.pre[t] : base;
    return (size_t)s * pa.cand_cap_shard + (size_t)(c - base);
}

__global__ __launch_bounds__(256) void k_verify(PairArgs pa) {
    __shared__ uint32_t sB[16][VERIFY_LDS_ROW];
    const int lane = threadIdx.x & 63, l16 = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;           // 0..15 in the block
    const int gsh = (lane >> 4) * 16;           // bit offset of this group inside the wave ballot
    const int gg = blockIdx.x * 16 + grp, ng = gridDim.x * 16;
    uint32_t *myB = sB[grp];
    const ShardMap sm = shard_map(pa);
    const int total = sm.pre[CAND_SHARDS];
    for (int c = gg; c < total; c += ng) {
        const size_t slot = shard_slot(sm, pa, c);
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
    const ShardMap sm = shard_map(pa);
    const int total = sm.pre[CAND_SHARDS];
    for (int c = gt; c < total; c += nt) {
        const int4 rec = pa.cand[shard_slot(sm, pa, c)];
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
    pa.parent = pl.parent;
    pa.cand = pl.cand;
    pa.candk = pl.candk;
    pa.cand_cap_shard = pl.cand_cap_shard;
    pa.d = pl.d;
    pa.ctr = pl.ctr;
    return pa;
}

// prefilter + verify over work items [w_begin, w_end) of this shard
int launch_pairs(const Plan &pl, int w_begin, int w_end, hipStream_t st, hipEvent_t *ev) {
    const int n = pl.n;
    PairArgs pa = make_pair_args(pl);
    const int pf_grid = pl.pf_grid;
    switch (pl.w1) {
        case 1:
            hipLaunchKernelGGL((k_prefilter<1, PF_ROWS_W1>), dim3(pf_grid), dim3(256), 0, st, pl.sig1, pl.work, n, pl.shard,
                               pl.n_shards, w_begin, w_end, pa);
            break;
        case 2:
            hipLaunchKernelGGL((k_prefilter<2, PF_ROWS_W2>), dim3(pf_grid), dim3(256), 0, st, pl.sig1, pl.work, n, pl.shard,
                               pl.n_shards, w_begin, w_end, pa);
            break;
        default:
            hipLaunchKernelGGL((k_prefilter<4, PF_ROWS_W4>), dim3(pf_grid), dim3(256), 0, st, pl.sig1, pl.work, n, pl.shard,
                               pl.n_shards, w_begin, w_end, pa);
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
    const int nb1024 = (n + 1023) / 1024;
    if (ev) (void)hipEventRecord(ev[0], st);
    hipLaunchKernelGGL(k_hist, dim3(nb1024), dim3(1024), 0, st, pl.indptr, n, pl.kcap, pl.hist, pl.parent, pl.ctr);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, st, pl.hist, n, pl.kcap, pl.d, pl.start, pl.cursor, pl.tr, pl.tc,
                       pl.work, pl.work_cap, pl.ctr);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scatter, dim3(nb1024), dim3(1024), 0, st, pl.indptr, n, pl.kcap, pl.cursor, pl.perm, pl.pos,
                       pl.ksorted);
    LAUNCH_CHECK();
    const int canon_blocks = min((n + 3) / 4, 256 * 16);
    const int lds_cap = pl.long_lds_cap;
    switch (pl.w1) {
#define CANON_CASE(W)                                                                                                  \
    case W:                                                                                                            \
        hipLaunchKernelGGL(k_canon<W>, dim3(canon_blocks), dim3(256), 0, st, pl.indptr, pl.indices, n, pl.pos, pl.cols, \
                           pl.sig1, pl.sig2, pl.longrows, pl.ctr);                                                     \
        if (pl.kcap > 256)                                                                                             \
            hipLaunchKernelGGL(k_canon_long<W>, dim3(min(n, 1024)), dim3(256), (size_t)lds_cap * 4, st, pl.indptr,      \
                               pl.indices, pl.pos, pl.cols, pl.sig1, pl.sig2, pl.longrows, pl.ctr, lds_cap);            \
        break;
        CANON_CASE(1)
        CANON_CASE(2)
        CANON_CASE(4)
#undef CANON_CASE
        default:
            return -1;
    }
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[1], st);
    if (int e = launch_pairs(pl, 0, 0x7FFFFFFF, st, ev)) return e;
    return launch_flatten(pl, st, ev);
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
