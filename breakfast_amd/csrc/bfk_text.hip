// bfk_text.hip — a1 on the device (gfx950): profile text -> first-appearance vocabulary -> CSR, resident in HBM.
//
// Replaces sparse_feature_matrix(features, feature_sep)  (src/breakfast/breakfast.py:193-215): split every row on the
// separator, skip empty tokens (:208-209), hand out ids by FIRST APPEARANCE (dict.setdefault, :210) in row order, keep
// repeats (the matrix is a count matrix).  The result — indptr / indices — equals the reference's CSR entry for entry
// (tests/test_gpu_text.py runs every golden stage vector, the KATs and the edge cases through it).
//
// Input: the rows' bytes as ONE buffer (row r = text[row_off[r] - row_off[0] .. row_off[r + 1] - row_off[0]), rows
// abut, there is no terminator between them) padded with separator bytes to a multiple of 16 KiB; one-byte separator.
// Byte-stream kernels: nothing below loops over a row, a wave never knows which rows its bytes belong to — rows enter
// as one bit per byte position (`rowbits`), so empty rows, rows of 100 kB and a million one-byte rows are the same code.
//
//   k_tok_rowbits  one thread per row: row_off validated (monotone, inside the text), bit `row start` set
//   k_tok_scan     16 bytes per lane, 1 KiB per wave step: separator bytes by SWAR compare -> 16-bit masks; a token
//                  STARTS at a non-separator byte whose predecessor is a separator or which starts a row; a token ENDS
//                  at the next separator or row start.  Start / bound masks are stored (T/8 bytes each), token starts
//                  counted per 1 KiB window
//   k_scan_single  exclusive prefix sum of the window counts by ONE block (n/1024 values: 31k at 100k rows, 325k at 1M).
//                  (A decoupled look-back inside k_tok_scan was tried first: blocks of equal, tiny work all start together
//                  and every one of them then walks back over ~2000 unfinished predecessors — 85 ns per block, serial:
//                  1.7 ms at 1M rows against 0.03 ms for this kernel)
//   k_tok_hash     a wave takes 4 KiB of text: the token starts go into a list in LDS (text order), the lanes take tokens
//                  from it, four per lane and round: end from the bound bits, 32-bit hash of the bytes,
//                  insert into an open-addressing table of 64-bit words {tag : length : byte offset of the token}:
//                  empty -> compare-and-swap; tag and length equal -> the BYTES are compared (the table is exact: hash
//                  collisions cost a probe, never an id) and the word is lowered to the smaller offset by atomicMin, so
//                  a slot ends up holding the offset of the token's FIRST occurrence.  Per token: its slot is stored.
//   k_tok_head     the same for the first 4 KiB of text alone, BEFORE k_tok_hash: the tokens every row carries are in the
//                  table before 8000 waves ask for them at once (same-address atomics serialise at ~11 ns each: 357 us for
//                  k_tok_hash at 100k rows without this launch, ~80 with it)
//   k_tok_rows     one thread per row: indptr[r] = number of token starts in front of row_off[r]; and one thread per
//                  table slot: bit `first occurrence` set at the byte offset the slot holds
//   k_voc_count    first-occurrence bits counted per window; k_scan_single again -> vocabulary entries in front of a window
//   k_voc_ids      one thread per slot: FIRST-APPEARANCE id = first occurrences in front of the slot's offset — ids by
//                  counting bits, no sort of the vocabulary and no pass over the tokens
//   k_tok_ids      indices[g] = id of token g's slot
//
// Every buffer is O(text): at 1M rows (333 MB of text) ~2 GB of the 288.
#include "bfk_device.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

namespace bfk {

namespace {

__device__ __forceinline__ uint32_t ldu32(const uint8_t *p) {  // unaligned (one global_load_dword)
    uint32_t w;
    __builtin_memcpy(&w, p, 4);
    return w;
}
__device__ __forceinline__ unsigned long long ldu64(const uint8_t *p) {
    unsigned long long w;
    __builtin_memcpy(&w, p, 8);
    return w;
}

__device__ __forceinline__ int tok_wave_incl_scan(int x) {  // inclusive prefix sum over the 64 lanes (DPP)
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true);  // row_bcast15 into rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true);  // row_bcast31 into rows 2,3
    return x;
}

// bytes of w equal to the byte replicated in sepx4 -> 4-bit mask (bit k = byte k).  Exact per byte: no borrow crosses
// a byte ((b & 0x7f) + 0x7f sets bit 7 iff b & 0x7f != 0); the multiply gathers bits 0, 8, 16, 24 into bits 24..27.
__device__ __forceinline__ uint32_t eq_bytes4(uint32_t w, uint32_t sepx4) {
    const uint32_t x = w ^ sepx4;
    const uint32_t nz = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
    const uint32_t z = (~nz & 0x80808080u) >> 7;
    return (z * 0x01020408u) >> 24;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tok_rowbits(TokArgs a) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_rows) return;
    const long long o = a.row_off[r] - a.base, e = a.row_off[r + 1] - a.base;
    if (o < 0 || e < o || e > (long long)a.T || (a.strict && ((r == 0 && o != 0) || (r == a.n_rows - 1 && e != (long long)a.T)))) {
        atomicOr(&a.tc->fail, TOK_FAIL_ROWOFF);
        return;
    }
    atomicOr(&a.rowbits[(uint32_t)o >> 5], 1u << ((uint32_t)o & 31u));
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_tok_scan(TokArgs a, uint32_t blk0) {
    __shared__ unsigned s_cnt[TOK_SCAN_WINS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t blk = blk0 + (uint32_t)blockIdx.x;
    const uint32_t win0 = blk * TOK_SCAN_WINS + wave * TOK_WPW;  // this wave's first window
    const uint32_t sepx4 = (uint32_t)a.sep * 0x01010101u;
    uint4 v[TOK_WPW];
    uint32_t rb[TOK_WPW];
#pragma unroll
    for (int i = 0; i < TOK_WPW; i++) {
        const uint32_t w0 = (win0 + i) * TOK_WIN;
        v[i] = *reinterpret_cast<const uint4 *>(a.text + w0 + 16 * lane);
        rb[i] = reinterpret_cast<const uint16_t *>(a.rowbits)[(w0 >> 4) + lane];
    }
    // is the byte in front of the wave's first window a separator (nothing in front of the text counts as one)
    uint32_t prev_sep = win0 == 0 ? 1u : (a.text[win0 * TOK_WIN - 1] == a.sep ? 1u : 0u);
    prev_sep = __builtin_amdgcn_readfirstlane(prev_sep);
#pragma unroll
    for (int i = 0; i < TOK_WPW; i++) {
        const uint32_t w0 = (win0 + i) * TOK_WIN;
        const uint32_t m = eq_bytes4(v[i].x, sepx4) | (eq_bytes4(v[i].y, sepx4) << 4) | (eq_bytes4(v[i].z, sepx4) << 8) |
                           (eq_bytes4(v[i].w, sepx4) << 12);
        uint32_t cin = ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x138, 0xF, 0xF, true) >> 15) & 1u;  // wave_shr:1
        if (lane == 0) cin = prev_sep;
        const uint32_t before_sep = ((m << 1) | cin) & 0xFFFFu;
        const uint32_t start = ~m & (before_sep | rb[i]) & 0xFFFFu;
        const uint32_t bound = (m | rb[i]) & 0xFFFFu;
        reinterpret_cast<uint16_t *>(a.startbits)[(w0 >> 4) + lane] = (uint16_t)start;
        reinterpret_cast<uint16_t *>(a.boundbits)[(w0 >> 4) + lane] = (uint16_t)bound;
        const int inc = tok_wave_incl_scan(__popc(start));
        if (lane == 63) s_cnt[wave * TOK_WPW + i] = (unsigned)inc;
        prev_sep = ((uint32_t)__builtin_amdgcn_readlane((int)m, 63) >> 15) & 1u;
    }
    __syncthreads();
    if (wave == 0) {  // tokens in front of every window INSIDE the block; the block's total goes to the scan of the blocks
        const int c = (int)s_cnt[lane];
        const int inc = tok_wave_incl_scan(c);
        a.winbase[blk * TOK_SCAN_WINS + lane] = (uint32_t)(inc - c);
        if (lane == 63) a.blkbase[blk] = (uint32_t)inc;
    }
}

// ------------------------------------------------------------------------------------------------
// k_scan_single: in-place exclusive prefix sum of data[0 .. n) by one block of 1024 threads, CONTINUING from *total (the sum
// of what earlier calls scanned: the text arrives in pieces and every piece is scanned when it is there): data[i] = *total +
// sum of data[0 .. i), then data[n] = *total = the new running sum.  Thread t owns a contiguous piece (a multiple of 4
// values: 16-byte loads and stores; `data` 16-byte aligned).
__global__ __launch_bounds__(1024) void k_scan_single(uint32_t *data, uint32_t n, unsigned *total_out) {
    __shared__ unsigned s_w[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned carry = *total_out;  // (read by every thread before the barrier below, written by thread 0 after it)
    const uint32_t per = ((n + 1023u) / 1024u + 3u) & ~3u;
    const uint32_t b = min(n, threadIdx.x * per), e = min(n, b + per);
    unsigned sum = 0;
    uint32_t i = b;
    for (; i + 4 <= e; i += 4) {
        const uint4 q = *reinterpret_cast<const uint4 *>(data + i);
        sum += q.x + q.y + q.z + q.w;
    }
    for (; i < e; i++) sum += data[i];
    const unsigned inc = (unsigned)tok_wave_incl_scan((int)sum);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned run = carry + inc - sum;
    unsigned total = carry;
    for (int w = 0; w < 16; w++) {
        if (w < wave) run += s_w[w];
        total += s_w[w];
    }
    i = b;
    for (; i + 4 <= e; i += 4) {
        uint4 q = *reinterpret_cast<const uint4 *>(data + i);
        const unsigned x0 = run, x1 = x0 + q.x, x2 = x1 + q.y, x3 = x2 + q.z;
        run = x3 + q.w;
        q.x = x0; q.y = x1; q.z = x2; q.w = x3;
        *reinterpret_cast<uint4 *>(data + i) = q;
    }
    for (; i < e; i++) {
        const unsigned x = data[i];
        data[i] = run;
        run += x;
    }
    if (threadIdx.x == 0) {
        data[n] = total;
        *total_out = total;
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mur_step(uint32_t h, uint32_t w) {
    w *= 0xCC9E2D51u;
    w = (w << 15) | (w >> 17);
    w *= 0x1B873593u;
    h ^= w;
    h = (h << 13) | (h >> 19);
    return h * 5u + 0xE6546B64u;
}
__device__ __forceinline__ uint32_t mur_final(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    return h ^ (h >> 16);
}

__device__ __forceinline__ bool same_bytes(const uint8_t *p, const uint8_t *q, uint32_t len) {
    uint32_t k = 0;
    for (; k + 8 <= len; k += 8)
        if (ldu64(p + k) != ldu64(q + k)) return false;
    const uint32_t rem = len - k;  // (the buffers are padded: the tail reads stay inside)
    if (rem == 0) return true;
    const unsigned long long mask = (1ull << (8 * rem)) - 1ull;
    return ((ldu64(p + k) ^ ldu64(q + k)) & mask) == 0ull;
}

// tokens of at most 16 bytes (all of a mutation profile's): bytes in two registers, no loop
__device__ __forceinline__ unsigned long long low_bytes(unsigned long long w, uint32_t n) {  // n in 0..8
    return n >= 8 ? w : (w & ((1ull << (8 * n)) - 1ull));
}

// the probe chain of one token from `slot` on (any token length): -> the token's slot
__device__ __forceinline__ uint32_t tok_probe(const TokArgs &a, uint32_t slot, uint32_t hi, uint32_t j, uint32_t len) {
    const uint8_t *p = a.text + j;
    const unsigned long long me = ((unsigned long long)hi << 32) | j;
    for (int probes = 0;; probes++) {
        // system scope: past this XCD's L2 (a slot filled by another XCD must not keep looking empty here)
        unsigned long long cur = __hip_atomic_load(&a.table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (cur == TOK_EMPTY) {
            cur = atomicCAS(&a.table[slot], TOK_EMPTY, me);
            if (cur == TOK_EMPTY) return slot;
        }
        if ((uint32_t)(cur >> 32) == hi && ((uint32_t)cur == j || same_bytes(a.text + (uint32_t)cur, p, len))) {
            if ((uint32_t)cur > j) atomicMin(&a.table[slot], me);
            return slot;
        }
        if (probes >= TOK_MAX_PROBE) {  // table too full: the host enlarges it and runs again
            atomicOr(&a.tc->fail, TOK_FAIL_TABLE);
            return 0;
        }
        slot = (slot + 1) & a.tmask;
    }
}

// k_tok_hash: a wave owns a UNIT of TOK_WPW consecutive windows (4 KiB of text).  The positions of the unit's token
// starts go into a list in the wave's LDS (prefix sums of the lanes' bit counts: the list is in text order, so token t
// of the list is token `first token of the unit + t` of the whole input), and the lanes then take tokens from the list,
// four per lane and round — every lane busy whatever the token lengths, the per-token stores coalesced, and four
// independent chains {bound bits + first 16 bytes -> table word -> bytes of the entry} in flight per lane.  (A lane
// working through the tokens of its own 16 bytes, one window after the other: half the lanes idle and one dependent load
// at a time.)  The first probe of a token is part of the pipelined round; the few it does not settle (a slot taken by
// another token, tokens over 16 bytes) go through tok_probe one at a time.
__device__ __forceinline__ void tok_hash_unit(const TokArgs &a, uint32_t unit, uint16_t *list /*LDS: TOK_WPW * TOK_WIN entries*/) {
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const uint32_t win0 = unit * TOK_WPW;
    uint32_t st[TOK_WPW];
#pragma unroll
    for (int u = 0; u < TOK_WPW; u++) st[u] = reinterpret_cast<const uint16_t *>(a.startbits)[(win0 + u) * (TOK_WIN / 16) + lane];
    const uint32_t g0 = a.blkbase[win0 / TOK_SCAN_WINS] + a.winbase[win0];  // (a unit lies in one scan block)
    uint32_t total = 0;
#pragma unroll
    for (int u = 0; u < TOK_WPW; u++) {
        const int c = __popc(st[u]);
        const int inc = tok_wave_incl_scan(c);
        uint32_t o = total + (uint32_t)(inc - c);
        for (uint32_t s = st[u]; s; s &= s - 1) list[o++] = (uint16_t)(u * TOK_WIN + 16 * lane + __builtin_ctz(s));
        total += (uint32_t)__builtin_amdgcn_readlane(inc, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t text0 = win0 * TOK_WIN;
    for (uint32_t r0 = 0; r0 < total; r0 += 64 * U) {
        bool act[U];
        uint32_t j[U], wi[U], bw0[U], bw1[U];
        unsigned long long t0[U], t1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {  // stage 1: four tokens of the list; bound bits + first 16 bytes requested
            const uint32_t t = r0 + u * 64 + lane;
            act[u] = t < total;
            j[u] = text0 + (act[u] ? (uint32_t)list[t] : 0u);
            wi[u] = (j[u] + 1) >> 5;
            bw0[u] = a.boundbits[wi[u]];
            bw1[u] = a.boundbits[wi[u] + 1];
            t0[u] = ldu64(a.text + j[u]);
            t1[u] = ldu64(a.text + j[u] + 8);
        }
        uint32_t len[U], hi[U], slot[U];
        unsigned long long m0[U], m1[U], cur[U];
        bool small[U];
#pragma unroll
        for (int u = 0; u < U; u++) {  // stage 2: token end -> length -> hash -> table word requested
            // end of the token: the first bound bit (separator or row start) behind j; the padding is all separators
            uint32_t bw = bw0[u] & (~0u << ((j[u] + 1) & 31u));
            uint32_t w = wi[u];
            if (!bw) {
                bw = bw1[u];
                w++;
                const uint32_t w_max = (j[u] + TOK_MAX_LEN + 64u) >> 5;  // (beyond: the token is too long whatever follows)
                while (!bw && w < w_max) bw = a.boundbits[++w];
                if (!bw) bw = 1u;
            }
            len[u] = w * 32 + (uint32_t)__builtin_ctz(bw) - j[u];
            small[u] = len[u] <= 16;
            m0[u] = low_bytes(t0[u], len[u]);
            m1[u] = low_bytes(t1[u], len[u] > 8 ? len[u] - 8 : 0u);
            uint32_t h = 0x9747B28Cu ^ len[u];
            if (small[u]) {  // the same words the loop below feeds
                h = mur_step(h, (uint32_t)m0[u]);
                if (len[u] > 4) h = mur_step(h, (uint32_t)(m0[u] >> 32));
                if (len[u] > 8) h = mur_step(h, (uint32_t)m1[u]);
                if (len[u] > 12) h = mur_step(h, (uint32_t)(m1[u] >> 32));
            } else if (act[u] && len[u] <= TOK_MAX_LEN) {
                const uint8_t *p = a.text + j[u];
                uint32_t k = 0;
                for (; k + 4 <= len[u]; k += 4) h = mur_step(h, ldu32(p + k));
                if (len[u] & 3u) h = mur_step(h, ldu32(p + k) & ((1u << (8 * (len[u] & 3u))) - 1u));
            }
            h = mur_final(h);
            hi[u] = (h & 0xFFFF0000u) | (len[u] & 0xFFFFu);  // tag : length
            slot[u] = h & a.tmask;
            // PLAIN load (this XCD's L2): entries only ever appear and only ever move to smaller offsets, so a stale word
            // is an older state — an entry seen here exists (its bytes are compared below), a stale offset is larger than
            // the true one and at worst costs an atomicMin that changes nothing; only what looks free or foreign is read
            // again past the L2.  The tokens every row carries are settled in L2 hits that way.
            cur[u] = a.table[act[u] ? slot[u] : 0u];
            if (a.dbg & 2) cur[u] = ((unsigned long long)hi[u] << 32) | j[u];
        }
        unsigned long long q0[U], q1[U];
        bool cmp[U], done[U];
#pragma unroll
        for (int u = 0; u < U; u++) {  // stage 3: free slot -> claim it; tag and length equal -> the entry's bytes requested
            done[u] = !act[u];
            if (!done[u] && !(a.dbg & 1) && (cur[u] == TOK_EMPTY || (uint32_t)(cur[u] >> 32) != hi[u]))
                cur[u] = __hip_atomic_load(&a.table[slot[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (act[u] && len[u] > TOK_MAX_LEN) {
                atomicOr(&a.tc->fail, TOK_FAIL_LONG);
                slot[u] = 0;
                done[u] = true;
            }
            if (!done[u] && cur[u] == TOK_EMPTY) {
                cur[u] = atomicCAS(&a.table[slot[u]], TOK_EMPTY, ((unsigned long long)hi[u] << 32) | j[u]);
                done[u] = cur[u] == TOK_EMPTY;
            }
            cmp[u] = !done[u] && small[u] && (uint32_t)(cur[u] >> 32) == hi[u] && !(a.dbg & 4);
            if (a.dbg & 4) done[u] = true;
            const uint8_t *q = a.text + (cmp[u] ? (uint32_t)cur[u] : j[u]);
            q0[u] = ldu64(q);
            q1[u] = ldu64(q + 8);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {  // stage 4: same bytes -> this is the token's slot; lower the entry to the first occurrence
            if (cmp[u] && low_bytes(q0[u], len[u]) == m0[u] && (len[u] <= 8 || low_bytes(q1[u], len[u] - 8) == m1[u])) {
                if ((uint32_t)cur[u] > j[u]) atomicMin(&a.table[slot[u]], ((unsigned long long)hi[u] << 32) | j[u]);
                done[u] = true;
            }
            if (!done[u])  // long tokens start here, short ones whose first slot holds another token go on behind it
                slot[u] = tok_probe(a, small[u] ? ((slot[u] + 1) & a.tmask) : slot[u], hi[u], j[u], len[u]);
            if (act[u]) a.tokslot[g0 + r0 + u * 64 + lane] = slot[u];
        }
    }
}

// the unit of the first 4 KiB, alone: the tokens every row carries are in the table before the whole text asks for them
__global__ __launch_bounds__(64) void k_tok_head(TokArgs a) {
    __shared__ uint16_t s_list[TOK_WPW * TOK_WIN];  // a token per byte at worst (rows of one byte)
    tok_hash_unit(a, 0u, s_list);
}

__global__ __launch_bounds__(256) void k_tok_hash(TokArgs a, uint32_t unit0, uint32_t n_units) {
    __shared__ uint16_t s_list[4][TOK_WPW * TOK_WIN];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t unit = unit0 + (uint32_t)blockIdx.x * 4 + wave;
    if (unit >= n_units) return;
    tok_hash_unit(a, unit, s_list[wave]);
}

// ------------------------------------------------------------------------------------------------
// k_tok_rows: thread r <= n_rows: indptr[r] = token starts in front of row_off[r] (the window's prefix + the bits of the
// window in front of the offset).  The same grid then walks the table: a slot in use sets the bit `first occurrence`
// at the byte offset it holds.
__global__ __launch_bounds__(256) void k_tok_rows(TokArgs a) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, nth = gridDim.x * 256u;
    for (uint32_t r = tid; r <= (uint32_t)a.n_rows; r += nth) {
        long long ol = a.row_off[r] - a.base;
        ol = ol < 0 ? 0 : (ol > (long long)a.T ? (long long)a.T : ol);  // (malformed offsets are reported by k_tok_rowbits)
        const uint32_t o = (uint32_t)ol;
        const uint32_t w = o / TOK_WIN;
        uint32_t cnt = a.blkbase[w / TOK_SCAN_WINS] + a.winbase[w];
        for (uint32_t q = w * (TOK_WIN / 32); q < (o >> 5); q++) cnt += (uint32_t)__popc(a.startbits[q]);
        cnt += (uint32_t)__popc(a.startbits[o >> 5] & ((1u << (o & 31u)) - 1u));
        a.indptr[r] = (int)cnt;
    }
    for (uint32_t s = tid; s <= a.tmask; s += nth) {
        const unsigned long long e = a.table[s];
        if (e != TOK_EMPTY) atomicOr(&a.firstbits[(uint32_t)e >> 5], 1u << ((uint32_t)e & 31u));
    }
}

// first-occurrence bits per window (32 words): 8 lanes per window, a uint4 each; a block = the 64 windows of a scan block
__global__ __launch_bounds__(512) void k_voc_count(TokArgs a) {
    __shared__ unsigned s_cnt[TOK_SCAN_WINS];
    const uint32_t t = blockIdx.x * 512u + threadIdx.x;  // word quad (T_pad is a multiple of the 64 KiB a block covers)
    const uint4 q = reinterpret_cast<const uint4 *>(a.firstbits)[t];
    uint32_t c = (uint32_t)(__popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w));
    c += __shfl_xor(c, 1);
    c += __shfl_xor(c, 2);
    c += __shfl_xor(c, 4);
    if ((threadIdx.x & 7u) == 0) s_cnt[threadIdx.x >> 3] = c;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int v = (int)s_cnt[threadIdx.x];
        const int inc = tok_wave_incl_scan(v);
        a.vocwin[blockIdx.x * TOK_SCAN_WINS + threadIdx.x] = (uint32_t)(inc - v);
        if (threadIdx.x == 63) a.vocblk[blockIdx.x] = (uint32_t)inc;
    }
}

// one thread per slot in use: id = first occurrences in front of its offset
__global__ __launch_bounds__(256) void k_voc_ids(TokArgs a) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, nth = gridDim.x * 256u;
    for (uint32_t s = tid; s <= a.tmask; s += nth) {
        const unsigned long long e = a.table[s];
        if (e == TOK_EMPTY) continue;
        const uint32_t o = (uint32_t)e;
        const uint32_t w = o / TOK_WIN;
        uint32_t cnt = a.vocblk[w / TOK_SCAN_WINS] + a.vocwin[w];
        for (uint32_t q = w * (TOK_WIN / 32); q < (o >> 5); q++) cnt += (uint32_t)__popc(a.firstbits[q]);
        cnt += (uint32_t)__popc(a.firstbits[o >> 5] & ((1u << (o & 31u)) - 1u));
        a.tabid[s] = (int)cnt;
    }
}

__global__ __launch_bounds__(256) void k_tok_ids(TokArgs a) {
    const uint32_t nnz = a.tc->nnz;
    const uint32_t g0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (g0 >= nnz) return;
    const uint4 s = *reinterpret_cast<const uint4 *>(a.tokslot + g0);
    uint4 o;
    o.x = (uint32_t)a.tabid[s.x];
    o.y = g0 + 1 < nnz ? (uint32_t)a.tabid[s.y] : 0u;
    o.z = g0 + 2 < nnz ? (uint32_t)a.tabid[s.z] : 0u;
    o.w = g0 + 3 < nnz ? (uint32_t)a.tabid[s.w] : 0u;
    *reinterpret_cast<uint4 *>(a.indices + g0) = o;
}

// ------------------------------------------------------------------------------------------------
#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// enqueue text -> CSR.  The caller has zeroed {counters, rowbits, firstbits}, filled the table with TOK_EMPTY and padded
// the text with separators up to T_pad + TOK_TEXT_SLACK.  The text may still be on its way: piece k (scan blocks
// [piece_blk[k], piece_blk[k + 1])) is scanned and hashed as soon as piece_ev[k] — recorded on the copy stream behind the
// piece's copy — has fired, so the kernels of one piece run under the copy of the next (n_pieces = 1, piece_ev = NULL: the
// text is there).
int launch_tokenize(const TokArgs &a, hipStream_t st, hipEvent_t *ev, int n_pieces, const unsigned *piece_blk, hipEvent_t *piece_ev) {
    const unsigned scan_blocks = a.T_pad / (TOK_SCAN_WINS * TOK_WIN);
    const unsigned table_blocks = (unsigned)std::min<unsigned long long>(((unsigned long long)a.tmask + 256) / 256, 8192);
    constexpr unsigned UNITS_PER_BLK = TOK_SCAN_WINS / TOK_WPW;  // hash units (4 KiB) per scan block (64 KiB)
    if (ev) (void)hipEventRecord(ev[0], st);
    if (a.n_rows > 0) {
        hipLaunchKernelGGL(k_tok_rowbits, dim3((a.n_rows + 255) / 256), dim3(256), 0, st, a);
        LAUNCH_CHECK();
    }
    unsigned hashed = 0;  // hash units done so far
    for (int k = 0; k < n_pieces; k++) {
        const unsigned b0 = n_pieces > 1 ? piece_blk[k] : 0u, b1 = n_pieces > 1 ? piece_blk[k + 1] : scan_blocks;
        if (piece_ev && hipStreamWaitEvent(st, piece_ev[k], 0) != hipSuccess) return (int)hipGetLastError();
        if (b1 > b0) {
            hipLaunchKernelGGL(k_tok_scan, dim3(b1 - b0), dim3(1024), 0, st, a, b0);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, a.blkbase + b0, b1 - b0, &a.tc->nnz);
            LAUNCH_CHECK();
        }
        if (ev && k == 0) (void)hipEventRecord(ev[1], st);
        // A token that starts near the end of the piece may reach into the next one (up to TOK_MAX_LEN bytes), whose text and
        // bound bits are not there yet: the last TOK_HOLD_UNITS units of a piece wait for the next piece's scan.
        unsigned u0 = hashed;
        const unsigned u_end = b1 * UNITS_PER_BLK;
        const unsigned u1 = k + 1 == n_pieces ? u_end : (u_end > hashed + TOK_HOLD_UNITS ? u_end - TOK_HOLD_UNITS : hashed);
        if (k == 0) {
            // the first 4 KiB alone (one wave), then the rest: the tokens every row carries are in the table before everyone
            // asks for them at once.  (A text so short that its first piece is held back whole hashes unit 0 with the rest.)
            if (u1 >= 1) {
                hipLaunchKernelGGL(k_tok_head, dim3(1), dim3(64), 0, st, a);
                LAUNCH_CHECK();
                u0 = 1;
            }
            if (ev) (void)hipEventRecord(ev[6], st);
        }
        if (u1 > u0) {
            hipLaunchKernelGGL(k_tok_hash, dim3((u1 - u0 + 3) / 4), dim3(256), 0, st, a, u0, u1);
            LAUNCH_CHECK();
        }
        hashed = std::max(hashed, u1);
    }
    if (ev) (void)hipEventRecord(ev[2], st);
    hipLaunchKernelGGL(k_tok_rows, dim3(std::max(table_blocks, (unsigned)std::min(8192, (a.n_rows + 256) / 256))), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_voc_count, dim3(scan_blocks), dim3(512), 0, st, a);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, a.vocblk, scan_blocks, &a.tc->n_vocab);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_voc_ids, dim3(table_blocks), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tok_ids, dim3(std::max(1u, (unsigned)((a.nnz_cap + 1023) / 1024))), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[3], st);
    return 0;
}

}  // namespace bfk
