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
//                  at the next separator or row start.  Start / bound masks are stored (T/8 bytes each), token starts are
//                  counted per window and prefix-summed over the whole text by a decoupled look-back (one pass)
//   k_tok_hash     a lane takes the tokens that start in its 16 bytes: end from the bound bits, 32-bit hash of the bytes,
//                  insert into an open-addressing table of 64-bit words {tag : length : byte offset of the token}:
//                  empty -> compare-and-swap; tag and length equal -> the BYTES are compared (the table is exact: hash
//                  collisions cost a probe, never an id) and the word is lowered to the smaller offset by atomicMin, so
//                  a slot ends up holding the offset of the token's FIRST occurrence.  Per token: slot + offset stored
//   k_tok_rows     one thread per row: indptr[r] = number of token starts in front of row_off[r]
//   k_tok_first    a token is the first occurrence of its vocabulary entry iff its offset is the one its slot holds;
//                  first-appearance id = number of first occurrences in front of it = a prefix sum over the token
//                  sequence (look-back again) — no sort of the vocabulary
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

// Decoupled look-back over a chain of status words (flag << 62 | value), one word per logical block; called by ONE full
// wave of the block with the block's total.  Returns the sum of the totals of the blocks in front.  A block's logical
// index comes from an arrival ticket, so it only ever waits for blocks that have started.
__device__ __forceinline__ unsigned lookback_exclusive(unsigned long long *chain, int bid, unsigned total, int lane, int *fail) {
    const unsigned long long FLAG_AGG = 1ull << 62, FLAG_INC = 2ull << 62, VAL = (1ull << 62) - 1ull;
    if (lane == 0 && bid > 0) __hip_atomic_store(&chain[bid], FLAG_AGG | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long before = 0;
    for (int w0 = bid - 1; w0 >= 0; w0 -= 64) {
        const int p = w0 - lane;
        unsigned long long v = 0ull;
        if (p >= 0) {
            int spins = 0;
            while (((v = __hip_atomic_load(&chain[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 62) == 0ull) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) {  // never expected: bound the spin
                    atomicOr(fail, TOK_FAIL_SPIN);
                    v = FLAG_INC;
                    break;
                }
            }
        }
        const unsigned long long inc_mask = __builtin_amdgcn_ballot_w64(p >= 0 && (v >> 62) == 2ull);
        const int stop = inc_mask ? (int)__builtin_ctzll(inc_mask) : 64;
        unsigned long long val = (p >= 0 && lane <= stop) ? (v & VAL) : 0ull;
        for (int s = 32; s > 0; s >>= 1) val += __shfl_xor(val, s);
        before += val;
        if (inc_mask) break;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (lane == 0)
        __hip_atomic_store(&chain[bid], FLAG_INC | (before + total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)before;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tok_rowbits(TokArgs a) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_rows) return;
    const long long o = a.row_off[r] - a.base, e = a.row_off[r + 1] - a.base;
    if (o < 0 || e < o || e > (long long)a.T) {
        atomicOr(&a.tc->fail, TOK_FAIL_ROWOFF);
        return;
    }
    atomicOr(&a.rowbits[(uint32_t)o >> 5], 1u << ((uint32_t)o & 31u));
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tok_scan(TokArgs a) {
    __shared__ unsigned s_cnt[4 * TOK_WPW];
    __shared__ int s_bid;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) s_bid = (int)atomicAdd(&a.tc->ticket_scan, 1u);
    __syncthreads();
    const int bid = s_bid;
    const uint32_t win0 = ((uint32_t)bid * 4 + wave) * TOK_WPW;  // this wave's first window
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
        const unsigned tot = (unsigned)__builtin_amdgcn_readlane(inc, 63);
        if (lane == 0) s_cnt[wave * TOK_WPW + i] = tot;
        prev_sep = ((uint32_t)__builtin_amdgcn_readlane((int)m, 63) >> 15) & 1u;
    }
    __syncthreads();
    if (wave == 0) {
        constexpr int NW = 4 * TOK_WPW;
        const int c = lane < NW ? (int)s_cnt[lane] : 0;
        const int inc = tok_wave_incl_scan(c);
        const unsigned total = (unsigned)__builtin_amdgcn_readlane(inc, 63);
        const unsigned before = lookback_exclusive(a.chain_scan, bid, total, lane, &a.tc->fail);
        if (lane < NW) a.winbase[(uint32_t)bid * NW + lane] = before + (unsigned)(inc - c);
        if (lane == 0 && bid + 1 == (int)gridDim.x) {
            a.tc->nnz = before + total;
            a.winbase[(uint32_t)gridDim.x * NW] = before + total;
        }
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

__global__ __launch_bounds__(256) void k_tok_hash(TokArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t win0 = ((uint32_t)blockIdx.x * 4 + wave) * TOK_WPW;
    uint32_t st[TOK_WPW], base[TOK_WPW];
#pragma unroll
    for (int i = 0; i < TOK_WPW; i++) {
        const uint32_t w0 = (win0 + i) * TOK_WIN;
        st[i] = reinterpret_cast<const uint16_t *>(a.startbits)[(w0 >> 4) + lane];
        base[i] = a.winbase[win0 + i];
    }
#pragma unroll
    for (int i = 0; i < TOK_WPW; i++) {
        const uint32_t w0 = (win0 + i) * TOK_WIN;
        uint32_t s = st[i];
        const int c = __popc(s);
        uint32_t g = base[i] + (uint32_t)(tok_wave_incl_scan(c) - c);
        while (s) {  // the tokens that start in this lane's 16 bytes (two on average)
            const uint32_t b = (uint32_t)__builtin_ctz(s);
            s &= s - 1;
            const uint32_t j = w0 + 16 * lane + b;
            // end of the token: the first bound bit (separator or row start) behind j; the padding is all separators
            uint32_t wi = (j + 1) >> 5;
            uint32_t bw = a.boundbits[wi] & (~0u << ((j + 1) & 31u));
            while (!bw) bw = a.boundbits[++wi];
            const uint32_t len = wi * 32 + (uint32_t)__builtin_ctz(bw) - j;
            uint32_t slot = ~0u;
            if (len > TOK_MAX_LEN) {
                atomicOr(&a.tc->fail, TOK_FAIL_LONG);
            } else {
                const uint8_t *p = a.text + j;
                uint32_t h = 0x9747B28Cu ^ len;
                uint32_t k = 0;
                for (; k + 4 <= len; k += 4) h = mur_step(h, ldu32(p + k));
                if (len & 3u) h = mur_step(h, ldu32(p + k) & ((1u << (8 * (len & 3u))) - 1u));
                h = mur_final(h);
                const uint32_t hi = (h & 0xFFFF0000u) | len;  // tag : length
                const unsigned long long me = ((unsigned long long)hi << 32) | j;
                slot = h & a.tmask;
                for (int probes = 0;; probes++) {
                    // system scope: past this XCD's L2 (a slot filled by another XCD must not keep looking empty here)
                    unsigned long long cur = __hip_atomic_load(&a.table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (cur == TOK_EMPTY) {
                        cur = atomicCAS(&a.table[slot], TOK_EMPTY, me);
                        if (cur == TOK_EMPTY) break;
                    }
                    if ((uint32_t)(cur >> 32) == hi && ((uint32_t)cur == j || same_bytes(a.text + (uint32_t)cur, p, len))) {
                        if ((uint32_t)cur > j) atomicMin(&a.table[slot], me);
                        break;
                    }
                    if (probes >= TOK_MAX_PROBE) {  // table too full: the host doubles it and runs again
                        atomicOr(&a.tc->fail, TOK_FAIL_TABLE);
                        slot = ~0u;
                        break;
                    }
                    slot = (slot + 1) & a.tmask;
                }
            }
            a.tokslot[g] = slot;
            a.tokoff[g] = j;
            g++;
        }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tok_rows(TokArgs a) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r > a.n_rows) return;
    long long ol = a.row_off[r] - a.base;
    ol = ol < 0 ? 0 : (ol > (long long)a.T ? (long long)a.T : ol);  // (malformed offsets are reported by k_tok_rowbits)
    const uint32_t o = (uint32_t)ol;
    const uint32_t w = o / TOK_WIN;
    uint32_t cnt = a.winbase[w];
    for (uint32_t q = w * (TOK_WIN / 32); q < (o >> 5); q++) cnt += (uint32_t)__popc(a.startbits[q]);
    cnt += (uint32_t)__popc(a.startbits[o >> 5] & ((1u << (o & 31u)) - 1u));
    a.indptr[r] = (int)cnt;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tok_first(TokArgs a) {
    __shared__ unsigned s_w[4];
    __shared__ unsigned s_base;
    __shared__ int s_bid;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) s_bid = (int)atomicAdd(&a.tc->ticket_first, 1u);
    __syncthreads();
    const int bid = s_bid;
    const uint32_t nnz = a.tc->nnz;
    if ((unsigned long long)bid * TOK_FIRST_PER_BLOCK >= nnz) return;  // (every block in front of a working block works too)
    const uint32_t g0 = (uint32_t)bid * TOK_FIRST_PER_BLOCK + threadIdx.x * 8;
    uint32_t sl[8], of[8];
    {
        const uint4 s0 = *reinterpret_cast<const uint4 *>(a.tokslot + g0), s1 = *reinterpret_cast<const uint4 *>(a.tokslot + g0 + 4);
        const uint4 o0 = *reinterpret_cast<const uint4 *>(a.tokoff + g0), o1 = *reinterpret_cast<const uint4 *>(a.tokoff + g0 + 4);
        sl[0] = s0.x; sl[1] = s0.y; sl[2] = s0.z; sl[3] = s0.w; sl[4] = s1.x; sl[5] = s1.y; sl[6] = s1.z; sl[7] = s1.w;
        of[0] = o0.x; of[1] = o0.y; of[2] = o0.z; of[3] = o0.w; of[4] = o1.x; of[5] = o1.y; of[6] = o1.z; of[7] = o1.w;
    }
    uint32_t first = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const bool ok = g0 + q < nnz && sl[q] != ~0u;
        const unsigned long long e = a.table[ok ? sl[q] : 0u];
        if (ok && (uint32_t)e == of[q]) first |= 1u << q;
    }
    const int c = __popc(first);
    const int inc = tok_wave_incl_scan(c);
    if (lane == 63) s_w[wave] = (unsigned)inc;
    __syncthreads();
    if (wave == 0) {
        const unsigned total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        const unsigned before = lookback_exclusive(a.chain_first, bid, total, lane, &a.tc->fail);
        if (lane == 0) {
            s_base = before;
            if ((unsigned long long)(bid + 1) * TOK_FIRST_PER_BLOCK >= nnz) a.tc->n_vocab = before + total;
        }
    }
    __syncthreads();
    unsigned id = s_base + (unsigned)(inc - c);
    for (int w = 0; w < wave; w++) id += s_w[w];
#pragma unroll
    for (int q = 0; q < 8; q++)
        if (first & (1u << q)) a.tabid[sl[q]] = (int)id++;
}

__global__ __launch_bounds__(256) void k_tok_ids(TokArgs a) {
    const uint32_t nnz = a.tc->nnz;
    const uint32_t g0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (g0 >= nnz) return;
    const uint4 s = *reinterpret_cast<const uint4 *>(a.tokslot + g0);
    uint4 o;
    o.x = (uint32_t)a.tabid[s.x != ~0u ? s.x : 0u];
    o.y = g0 + 1 < nnz ? (uint32_t)a.tabid[s.y != ~0u ? s.y : 0u] : 0u;
    o.z = g0 + 2 < nnz ? (uint32_t)a.tabid[s.z != ~0u ? s.z : 0u] : 0u;
    o.w = g0 + 3 < nnz ? (uint32_t)a.tabid[s.w != ~0u ? s.w : 0u] : 0u;
    *reinterpret_cast<uint4 *>(a.indices + g0) = o;
}

// ------------------------------------------------------------------------------------------------
#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// enqueue text -> CSR.  The caller has zeroed {counters, both chains, rowbits}, filled the table with TOK_EMPTY and
// padded the text with separators up to T_pad + TOK_TEXT_SLACK.
int launch_tokenize(const TokArgs &a, hipStream_t st, hipEvent_t *ev) {
    const unsigned scan_blocks = a.T_pad / TOK_BLOCK_BYTES;
    if (ev) (void)hipEventRecord(ev[0], st);
    if (a.n_rows > 0) {
        hipLaunchKernelGGL(k_tok_rowbits, dim3((a.n_rows + 255) / 256), dim3(256), 0, st, a);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_tok_scan, dim3(scan_blocks), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[1], st);
    hipLaunchKernelGGL(k_tok_hash, dim3(scan_blocks), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[2], st);
    hipLaunchKernelGGL(k_tok_rows, dim3((a.n_rows + 1 + 255) / 256), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    const unsigned first_blocks = (unsigned)((a.nnz_cap + TOK_FIRST_PER_BLOCK - 1) / TOK_FIRST_PER_BLOCK);
    hipLaunchKernelGGL(k_tok_first, dim3(std::max(1u, first_blocks)), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tok_ids, dim3(std::max(1u, (unsigned)((a.nnz_cap + 1023) / 1024))), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[3], st);
    return 0;
}

}  // namespace bfk
