// bfk_prep.hip — the stages between the reader and the clustering kernels on the device (gfx950):
//   filter_features      (src/breakfast/breakfast.py:116-190): inside the tokeniser (bfk_text.hip, TokFilter) — a dropped token
//                        never enters the vocabulary or the CSR
//   collapse_duplicates  (:72-79): rows with one filtered feature string are one unique row, unique rows in order of first
//                        appearance — HERE: row hash -> table keyed by the hash holding the smallest row -> exact comparison of
//                        every row with that representative -> prefix sums over the representatives -> group index of every
//                        input row, CSR of the unique rows (what sparse_feature_matrix builds from the collapsed frame, :193-215)
// The identity of a row is what the reference groups by: the filtered STRING — i.e. the sequence of kept tokens when the filter
// re-joins them (ids of equal tokens are equal: the id sequence), the raw bytes of the feature when nothing is filtered (:128-129
// returns the input untouched; "A  B" and "A B" then stay apart).  A hash only finds the candidates: every row is compared with
// its representative id for id / byte for byte, and a mismatch (two different rows with one 64-bit hash) is reported to the host,
// which then runs its own collapse.
#include "bfk_device.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

namespace bfk {

namespace {

__device__ __forceinline__ uint32_t ldu32p(const uint8_t *p) {
    uint32_t w;
    __builtin_memcpy(&w, p, 4);
    return w;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    return x ^ (x >> 16);
}

// sum over the 16 lanes of a row group (DPP row operations: every lane of the group ends with the total)
__device__ __forceinline__ uint32_t row16_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xF, 0xF, true);  // row_ror:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xF, 0xF, true);  // row_ror:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xF, 0xF, true);  // row_ror:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, true);  // row_ror:8
    return x;
}

__device__ __forceinline__ bool row16_all(bool ok) {
    const unsigned long long bal = __ballot(ok);
    const int g = (threadIdx.x & 63) >> 4;
    return ((bal >> (16 * g)) & 0xFFFFull) == 0xFFFFull;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// k_blank: what lies between the end of row r's span and the start of row r + 1's (the other columns of the table, the line
// end) becomes separator bytes: the tokeniser then sees nothing there, and the separator right behind a span ends its last token
__global__ __launch_bounds__(256) void k_blank(uint8_t *text, const long long *__restrict__ row_off, const int *__restrict__ span_len, int n,
                                               long long base, uint32_t T, uint8_t sep) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const uint32_t s = (uint32_t)(row_off[r] - base) + (uint32_t)span_len[r];
    const uint32_t e = r + 1 < n ? (uint32_t)(row_off[r + 1] - base) : T;
    for (uint32_t i = s; i < e && i < T; i++) text[i] = sep;
}

// ------------------------------------------------------------------------------------------------
// k_sepfold: a token separator of m > 1 bytes (breakfast.py:164 splits on any string).  Thread r walks row r's span from the left —
// str.split's matches: leftmost, never overlapping — and overwrites every occurrence with m STAND-IN bytes, a byte the table
// does not hold: the tokeniser then runs with that one byte as its separator.  Offsets stay what they are (the host prints a
// token from its own image of the table by offset and length), rows that differ still differ (no row holds the stand-in, so the
// runs of it say where the separators were), and an occurrence leaves m - 1 empty tokens that are not the reference's: seps[r]
// = occurrences in row r, *total = their sum — the host takes (m - 1) x these off the empty-token counts (both NULL where nobody
// counts empty tokens: sparse_feature_matrix skips them, :208-209).
__global__ __launch_bounds__(256) void k_sepfold(uint8_t *text, const long long *__restrict__ row_off, const int *__restrict__ span_len, int n,
                                                 long long base, SepPattern pat, uint8_t standin, int *__restrict__ seps,
                                                 unsigned long long *total) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    unsigned cnt = 0;
    if (r < n) {
        uint8_t *t = text + (size_t)(row_off[r] - base);
        const int len = span_len ? span_len[r] : (int)(row_off[r + 1] - row_off[r]), m = pat.m;  // (no spans: the rows abut)
        for (int k = 0; k + m <= len;) {
            bool eq = t[k] == pat.b[0];
            for (int j = 1; eq && j < m; j++) eq = t[k + j] == pat.b[j];
            if (eq) {
                for (int j = 0; j < m; j++) t[k + j] = standin;
                cnt++;
                k += m;
            } else {
                k++;
            }
        }
        if (seps) seps[r] = (int)cnt;
    }
    if (!total) return;  // (kernel argument: the whole grid leaves)
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(total, (unsigned long long)cnt);
}

// ------------------------------------------------------------------------------------------------
// k_row_hash: 16 lanes per row.  The 64-bit hash of the row's identity — order-dependent: position i enters every term — and the
// row's place in the table {hash -> smallest row with it}.  A plain load first: a hub (thousands of rows with one string) would
// otherwise send every one of its rows through a compare-and-swap and an atomicMin on ONE slot (same-address atomics serialise
// at ~11 ns); a slot that already shows the hash with a smaller row needs neither (the row only ever goes down).
__global__ __launch_bounds__(256) void k_row_hash(PrepArgs a) {
    const int lane = threadIdx.x & 15;
    const int r = (int)((blockIdx.x * 256u + threadIdx.x) >> 4);
    if (r >= a.n) return;  // (whole groups leave together)
    uint32_t a1 = 0, a2 = 0, len;
    if (a.by_bytes) {
        const uint32_t s = (uint32_t)(a.row_off[r] - a.base);
        len = a.span_len ? (uint32_t)a.span_len[r] : (uint32_t)(a.row_off[r + 1] - a.base) - s;
        const uint8_t *p = a.text + s;
        for (uint32_t i = 4 * lane; i < len; i += 64) {
            uint32_t w = ldu32p(p + i);  // (the text is padded: the tail read stays inside)
            if (len - i < 4) w &= (1u << (8 * (len - i))) - 1u;
            a1 += mix32(w + 0x9E3779B9u * (i + 1));
            a2 += mix32((w ^ 0x5BD1E995u) * 0x2545F491u + 0x85EBCA6Bu * (i + 1));
        }
    } else {
        const int b = a.indptr[r];
        len = (uint32_t)(a.indptr[r + 1] - b);
        for (uint32_t i = lane; i < len; i += 16) {
            const uint32_t x = a.indices[b + i];
            a1 += mix32(x + 0x9E3779B9u * (i + 1));
            a2 += mix32((x ^ 0x5BD1E995u) * 0x2545F491u + 0x85EBCA6Bu * (i + 1));
        }
    }
    a1 = row16_sum(a1);
    a2 = row16_sum(a2);
    unsigned long long h = ((unsigned long long)mix32(a2 ^ len) << 32) | mix32(a1 + 0x632BE5ABu * len);
    if (h == PREP_EMPTY) h--;
    if (lane != 0) return;
    a.rowhash[r] = h;
    uint32_t slot = (uint32_t)(h ^ (h >> 29)) & a.mask;
    for (int probes = 0;; probes++) {
        uint4 q = *reinterpret_cast<const uint4 *>(&a.table[slot]);
        unsigned long long cur = ((unsigned long long)q.y << 32) | q.x;
        if (cur == PREP_EMPTY) {
            cur = atomicCAS(&a.table[slot].key, PREP_EMPTY, h);
            if (cur == PREP_EMPTY) cur = h;
            q.z = 0xFFFFFFFFu;  // (what the plain load showed belongs to the free slot)
        }
        if (cur == h) {
            if (q.z > (uint32_t)r) atomicMin(&a.table[slot].row, (uint32_t)r);  // (a stale row is a larger one: at worst an atomic too many)
            return;
        }
        if (probes >= PREP_MAX_PROBE) {
            atomicOr(a.fail, PREP_FAIL_TABLE);
            return;
        }
        slot = (slot + 1) & a.mask;
    }
}

// k_row_rep: 16 lanes per row: the representative of row r = the smallest row with r's hash; r is compared with it exactly
// (id for id / byte for byte).  val[r] = {r is a representative, its number of CSR entries if so} for the prefix sums.
__global__ __launch_bounds__(256) void k_row_rep(PrepArgs a) {
    const int lane = threadIdx.x & 15;
    const int r = (int)((blockIdx.x * 256u + threadIdx.x) >> 4);
    if (r >= a.n) return;
    const unsigned long long h = a.rowhash[r];
    uint32_t slot = (uint32_t)(h ^ (h >> 29)) & a.mask;
    int rep = r;
    for (int probes = 0; probes <= PREP_MAX_PROBE; probes++) {
        const PrepSlot e = a.table[slot];
        if (e.key == h) {
            rep = (int)e.row;
            break;
        }
        if (e.key == PREP_EMPTY) break;  // (cannot happen: k_row_hash put it there)
        slot = (slot + 1) & a.mask;
    }
    bool same = true;
    if (rep != r) {
        if (a.by_bytes) {
            const uint32_t s = (uint32_t)(a.row_off[r] - a.base), s2 = (uint32_t)(a.row_off[rep] - a.base);
            const uint32_t len = a.span_len ? (uint32_t)a.span_len[r] : (uint32_t)(a.row_off[r + 1] - a.base) - s;
            const uint32_t len2 = a.span_len ? (uint32_t)a.span_len[rep] : (uint32_t)(a.row_off[rep + 1] - a.base) - s2;
            same = len == len2;
            for (uint32_t i = 4 * lane; same && i < len; i += 64) {
                uint32_t w = ldu32p(a.text + s + i), w2 = ldu32p(a.text + s2 + i);
                if (len - i < 4) {
                    const uint32_t m = (1u << (8 * (len - i))) - 1u;
                    w &= m;
                    w2 &= m;
                }
                same = w == w2;
            }
        } else {
            const int b = a.indptr[r], b2 = a.indptr[rep];
            const uint32_t len = (uint32_t)(a.indptr[r + 1] - b), len2 = (uint32_t)(a.indptr[rep + 1] - b2);
            same = len == len2;
            for (uint32_t i = lane; same && i < len; i += 16) same = a.indices[b + i] == a.indices[b2 + i];
        }
        same = row16_all(same);  // (all 16 lanes of a group are here together: rep is the group's)
        if (!same && lane == 0) atomicOr(a.fail, PREP_FAIL_COLLISION);  // two different rows with one hash: the host collapses
    }
    if (lane == 0) {
        a.rep[r] = rep;
        const bool is = rep == r;
        a.val[r] = make_int2(is ? 1 : 0, is ? a.indptr[r + 1] - a.indptr[r] : 0);
    }
}

// exclusive prefix sums of val[] (both components) inside blocks of 1024 rows; the blocks' totals go to blk[] for k_scan_single2
__global__ __launch_bounds__(1024) void k_prep_scan(int2 *val, int n, int2 *blk) {
    __shared__ int2 s_w[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int2 v = i < n ? val[i] : make_int2(0, 0);
    int x = v.x, y = v.y;
    for (int o = 1; o < 64; o <<= 1) {
        const int tx = __shfl_up(x, o), ty = __shfl_up(y, o);
        if (lane >= o) {
            x += tx;
            y += ty;
        }
    }
    if (lane == 63) s_w[wave] = make_int2(x, y);
    __syncthreads();
    int bx = 0, by = 0, totx = 0, toty = 0;
    for (int w = 0; w < 16; w++) {
        if (w < wave) {
            bx += s_w[w].x;
            by += s_w[w].y;
        }
        totx += s_w[w].x;
        toty += s_w[w].y;
    }
    if (i < n) val[i] = make_int2(bx + x - v.x, by + y - v.y);
    if (threadIdx.x == 0) blk[blockIdx.x] = make_int2(totx, toty);
}

// exclusive prefix of the block totals by one block (n_blk <= a few thousand); totals -> out[0..1]
__global__ __launch_bounds__(1024) void k_prep_scan_top(int2 *blk, int n_blk, int *out) {
    __shared__ int2 s_w[16];
    __shared__ int2 s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = make_int2(0, 0);
    __syncthreads();
    for (int base = 0; base < n_blk; base += 1024) {
        const int i = base + threadIdx.x;
        const int2 v = i < n_blk ? blk[i] : make_int2(0, 0);
        int x = v.x, y = v.y;
        for (int o = 1; o < 64; o <<= 1) {
            const int tx = __shfl_up(x, o), ty = __shfl_up(y, o);
            if (lane >= o) {
                x += tx;
                y += ty;
            }
        }
        if (lane == 63) s_w[wave] = make_int2(x, y);
        __syncthreads();
        int bx = s_carry.x, by = s_carry.y, totx = 0, toty = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) {
                bx += s_w[w].x;
                by += s_w[w].y;
            }
            totx += s_w[w].x;
            toty += s_w[w].y;
        }
        if (i < n_blk) blk[i] = make_int2(bx + x - v.x, by + y - v.y);
        __syncthreads();
        if (threadIdx.x == 0) s_carry = make_int2(s_carry.x + totx, s_carry.y + toty);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = s_carry.x;  // unique rows
        out[1] = s_carry.y;  // CSR entries of the unique rows
    }
}

// k_row_out: 16 lanes per row: group[r] = unique index of r's representative; a representative writes its row of the unique CSR
__global__ __launch_bounds__(256) void k_row_out(PrepArgs a) {
    const int lane = threadIdx.x & 15;
    const int r = (int)((blockIdx.x * 256u + threadIdx.x) >> 4);
    if (r >= a.n) return;
    const int rep = a.rep[r];
    const int2 pr = a.val[rep], pb = a.blk[rep >> 10];
    const int u = pr.x + pb.x;
    if (lane == 0) a.group[r] = u;
    if (rep == r) {
        const int o = pr.y + pb.y, b = a.indptr[r], k = a.indptr[r + 1] - b;
        if (lane == 0) {
            a.u_indptr[u] = o;
            a.first_row[u] = r;
        }
        for (int i = lane; i < k; i += 16) a.u_indices[o + i] = a.indices[b + i];
    }
    if (r == a.n - 1 && lane == 0) a.u_indptr[a.totals[0]] = a.totals[1];
}

#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

int launch_blank(uint8_t *text, const long long *row_off, const int *span_len, int n, long long base, uint32_t T, uint8_t sep, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_blank, dim3((n + 255) / 256), dim3(256), 0, st, text, row_off, span_len, n, base, T, sep);
    LAUNCH_CHECK();
    return 0;
}

int launch_sepfold(uint8_t *text, const long long *row_off, const int *span_len, int n, long long base, const SepPattern &pat,
                   uint8_t standin, int *seps, unsigned long long *total, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_sepfold, dim3((n + 255) / 256), dim3(256), 0, st, text, row_off, span_len, n, base, pat, standin, seps, total);
    LAUNCH_CHECK();
    return 0;
}

// enqueue the collapse: the caller has filled the table with PREP_EMPTY keys (0xFF bytes) and zeroed *fail; totals[0..1] =
// {unique rows, their CSR entries} when the stream is done
int launch_collapse(const PrepArgs &a, hipStream_t st) {
    if (a.n <= 0) return 0;
    const unsigned g16 = (unsigned)(((long long)a.n * 16 + 255) / 256);
    hipLaunchKernelGGL(k_row_hash, dim3(g16), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_row_rep, dim3(g16), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    const int n_blk = (a.n + 1023) / 1024;
    hipLaunchKernelGGL(k_prep_scan, dim3(n_blk), dim3(1024), 0, st, a.val, a.n, a.blk);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_prep_scan_top, dim3(1), dim3(1024), 0, st, a.blk, n_blk, a.totals);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_row_out, dim3(g16), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace bfk
