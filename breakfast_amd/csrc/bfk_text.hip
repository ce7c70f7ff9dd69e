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
//                  THREE launches: the first 16 units (64 KiB), every 16th unit of the rest, the others — the tokens that many
//                  rows carry are in the table before 8000 waves ask for them at once (same-address compare-and-swaps
//                  serialise at ~11 ns each: 1.3 ms for one launch at 100k rows)
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
// separator / token-start / token-bound masks of one 1 KiB window step: 16 bytes per lane -> 16-bit masks per lane.
// `prev_sep`: is the byte in front of the window a separator (wave-uniform; nothing in front of the text counts as one).
__device__ __forceinline__ void tok_masks(const uint4 v, uint32_t rb, uint32_t sepx4, uint32_t prev_sep, int lane, uint32_t *start,
                                          uint32_t *bound, uint32_t *last_is_sep) {
    const uint32_t m = eq_bytes4(v.x, sepx4) | (eq_bytes4(v.y, sepx4) << 4) | (eq_bytes4(v.z, sepx4) << 8) | (eq_bytes4(v.w, sepx4) << 12);
    uint32_t cin = ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x138, 0xF, 0xF, true) >> 15) & 1u;  // wave_shr:1
    if (lane == 0) cin = prev_sep;
    const uint32_t before_sep = ((m << 1) | cin) & 0xFFFFu;
    *start = ~m & (before_sep | rb) & 0xFFFFu;
    *bound = (m | rb) & 0xFFFFu;
    *last_is_sep = ((uint32_t)__builtin_amdgcn_readlane((int)m, 63) >> 15) & 1u;
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

// ---- token classification (filter_features, breakfast.py:131-187): the five feature grammars as byte matchers ----------
// The same rules as Classifier in bfk_frontend.cpp (the host stage), which tests/test_frontend.py pins against the regex mirror;
// tests/test_gpu_prep.py runs both on the same tokens.  GET(i) = byte i of the token (0 <= i < n, n >= 1: empty tokens never
// get here).  -> 0 keep, 1 drop, 2 invalid ("Skipping invalid feature")
enum : int { TOKV_KEEP = 0, TOKV_DROP = 1, TOKV_INVALID = 2 };

template <typename GET>
__device__ __forceinline__ int tok_classify(const TokFilter &f, uint32_t n, GET get) {
    auto up = [](uint32_t c) { return c >= 'A' && c <= 'Z'; };
    auto dg = [](uint32_t c) { return c >= '0' && c <= '9'; };
    auto alnum = [&](uint32_t c) { return up(c) || dg(c) || (c >= 'a' && c <= 'z'); };
    auto all_digits = [&](uint32_t b, uint32_t e) {  // [b, e) non-empty and all digits
        if (e <= b) return false;
        for (uint32_t i = b; i < e; i++)
            if (!dg(get(i))) return false;
        return true;
    };
    auto find = [&](uint32_t b, uint32_t e, uint32_t ch) {  // first index of ch in [b, e), or e
        for (uint32_t i = b; i < e; i++)
            if (get(i) == ch) return i;
        return e;
    };
    auto num_colon_num = [&](uint32_t b, uint32_t e) {  // \d+:\d+ over [b, e)
        const uint32_t c = find(b, e, ':');
        return c < e && all_digits(b, c) && all_digits(c + 1, e);
    };
    auto is_del_prefix = [&](uint32_t b, uint32_t e) {  // del:\d+:\d+ over [b, e)
        return e > b + 4 && get(b) == 'd' && get(b + 1) == 'e' && get(b + 2) == 'l' && get(b + 3) == ':' && num_colon_num(b + 4, e);
    };
    const int ins = f.skip_ins ? TOKV_DROP : TOKV_KEEP, del = f.skip_del ? TOKV_DROP : TOKV_KEEP;
    // ^[A-Z](\d+)[A-Z]$ with the trims (:166-175): dropped when pos <= trim_start or pos >= upper
    auto dna_sub = [&](int *v) {
        if (n < 3 || !up(get(0)) || !up(get(n - 1)) || !all_digits(1, n - 1)) return false;
        uint32_t b = 1, e = n - 1;
        while (e - b > 1 && get(b) == '0') b++;
        bool trimmed = true;  // more than 18 digits: beyond any int64 bound, pos >= upper
        if (e - b <= 18) {
            long long pos = 0;
            for (uint32_t i = b; i < e; i++) pos = pos * 10 + (long long)(get(i) - '0');
            trimmed = pos <= f.trim_start || pos >= f.upper;
        }
        *v = trimmed ? TOKV_DROP : TOKV_KEEP;
        return true;
    };
    // [a-zA-Z0-9]+: -> index behind the ':' or 0
    auto gene_prefix = [&]() {
        uint32_t i = 0;
        while (i < n && alnum(get(i))) i++;
        return (i > 0 && i < n && get(i) == ':') ? i + 1 : 0u;
    };
    // [A-Z]\d+ from b -> index behind the digits or 0
    auto letter_digits = [&](uint32_t b) {
        if (n < b + 2 || !up(get(b)) || !dg(get(b + 1))) return 0u;
        uint32_t i = b + 1;
        while (i < n && dg(get(i))) i++;
        return i;
    };
    int v;
    switch (f.var_type) {
    case 0:  // covsonar_dna
        if (dna_sub(&v)) return v;
        if (n >= 2 && up(get(n - 1)) && up(get(n - 2))) return ins;  // ^.*[A-Z][A-Z]$
        if (is_del_prefix(0, n)) return del;                          // ^del:\d+:\d+$
        return TOKV_INVALID;
    case 2: {  // nextclade_dna
        if (dna_sub(&v)) return v;
        const uint32_t c = find(0, n, ':');
        if (c < n && all_digits(0, c)) {  // ^\d+:[A-Z]+$
            bool ok = n > c + 1;
            for (uint32_t i = c + 1; ok && i < n; i++) ok = up(get(i));
            if (ok) return ins;
        }
        const uint32_t m = find(0, n, '-');  // ^\d+(-\d+)?$
        if (m < n ? (all_digits(0, m) && all_digits(m + 1, n)) : all_digits(0, n)) return del;
        return TOKV_INVALID;
    }
    case 1: {  // covsonar_aa
        const uint32_t g = gene_prefix();
        if (!g) return TOKV_INVALID;
        const uint32_t e = letter_digits(g);
        if (e && e < n) {
            bool letters = true;
            for (uint32_t i = e; i < n; i++) letters = letters && up(get(i));
            if (letters) return n - e == 1 ? TOKV_KEEP : ins;  // [A-Z]\d+[A-Z] | [A-Z]\d+[A-Z][A-Z]+
        }
        if (is_del_prefix(g, n)) return del;
        return TOKV_INVALID;
    }
    case 3: {  // nextclade_aa (its insertion pattern is ^$: only an empty token matches it)
        const uint32_t g = gene_prefix();
        if (!g) return TOKV_INVALID;
        const uint32_t e = letter_digits(g);
        if (e && e + 1 == n) {
            const uint32_t c = get(e);
            if (up(c) || c == '*') return TOKV_KEEP;  // [A-Z]\d+[A-Z*]
            if (c == '-') return del;                 // [A-Z]\d+-
        }
        return TOKV_INVALID;
    }
    default:
        return TOKV_KEEP;  // raw: no patterns
    }
}

// ---- the vocabulary table -------------------------------------------------------------------------------------------
// 16-byte slots {key64, first32, id32}, open addressing, linear probing.  Two kinds of key:
//   INLINE  (top byte = length 1 .. 7, low 56 bits = the token's bytes): the token IS the key — a lookup is one 16-byte load
//           and one 64-bit compare, no bytes of the text are fetched (a mutation like A23403G is 7 bytes: every substitution
//           of a 29 903-base genome fits).  `first` = smallest byte offset at which the token was seen (atomicMin).
//   HASHED  (bit 63 set: {1 : tag15 : len16 : offset32}) for tokens of 8 bytes and more (insertions, del:11288:9): tag and
//           length must match, then the BYTES are compared with the entry's occurrence; the word is lowered to the smaller
//           offset by a 64-bit atomicMin.  The table is exact either way: a hash collision costs a probe, never an id.
// Keys only ever appear; an entry's kind, length and bytes never change — so a PLAIN (cached) load that shows a key shows the
// truth, and only a slot that looks free is read again past the caches before it is claimed by compare-and-swap.
__device__ __forceinline__ uint32_t tok_entry_offset(const TokSlot &e) {
    return (e.key >> 63) ? (uint32_t)e.key : e.first;
}

__device__ __forceinline__ uint32_t tok_hash_inline(unsigned long long key) {
    uint32_t h = (uint32_t)key * 0x9E3779B1u;
    h ^= h >> 15;
    h += (uint32_t)(key >> 32) * 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    return h ^ (h >> 16);
}

// A token occurrence that matches no pattern of the feature type (the reference prints every one of them, breakfast.py:182-184):
// counted, and — up to inv_cap of them — noted as {byte offset, length} for the host, which puts them into the text's order.
__device__ __forceinline__ void tok_note_invalid(const TokArgs &a, uint32_t j, uint32_t len) {
    const uint32_t q = atomicAdd(&a.tc->n_invalid, 1u);
    if (q < a.inv_cap) a.inv_queue[q] = make_uint2(j, len);
}

// a token of 8 bytes or more at byte offset j: length from the bound bits in global memory, the filter's verdict, hash over
// its bytes, HASHED entry.  -> its slot, or TOK_NONE when the filter drops it (*invalid: it matched no pattern)
__device__ __forceinline__ uint32_t tok_long(const TokArgs &a, uint32_t j, bool *invalid) {
    uint32_t w = (j + 1) >> 5;
    uint32_t bw = a.boundbits[w] & (~0u << ((j + 1) & 31u));
    const uint32_t w_max = (j + TOK_MAX_LEN + 64u) >> 5;  // (beyond: the token is too long whatever follows; the padding is all separators)
    while (!bw && w < w_max) bw = a.boundbits[++w];
    if (!bw) bw = 1u;
    const uint32_t len = w * 32 + (uint32_t)__builtin_ctz(bw) - j;
    if (len > TOK_MAX_LEN) {
        atomicOr(&a.tc->fail, TOK_FAIL_LONG);
        return 0;
    }
    const uint8_t *p = a.text + j;
    if (a.flt.on) {
        const int v = tok_classify(a.flt, len, [&](uint32_t i) { return (uint32_t)p[i]; });
        if (v != TOKV_KEEP) {
            *invalid = v == TOKV_INVALID;
            if (v == TOKV_INVALID) tok_note_invalid(a, j, len);
            return TOK_NONE;
        }
    }
    uint32_t h = 0x9747B28Cu ^ len, k = 0;
    for (; k + 4 <= len; k += 4) h = mur_step(h, ldu32(p + k));
    if (len & 3u) h = mur_step(h, ldu32(p + k) & ((1u << (8 * (len & 3u))) - 1u));
    h = mur_final(h);
    const uint32_t hi = 0x80000000u | (h & 0x7FFF0000u) | (len & 0xFFFFu);  // 1 : tag15 : len16
    const unsigned long long me = ((unsigned long long)hi << 32) | j;
    uint32_t slot = h & a.tmask;
    for (int probes = 0;; probes++) {
        unsigned long long cur = a.table[slot].key;
        if (cur == TOK_EMPTY || (uint32_t)(cur >> 32) == hi)  // free, or maybe this token: the state past the caches (the offset moves)
            cur = __hip_atomic_load(&a.table[slot].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (cur == TOK_EMPTY) {
            cur = atomicCAS(&a.table[slot].key, TOK_EMPTY, me);
            if (cur == TOK_EMPTY) return slot;
        }
        if ((uint32_t)(cur >> 32) == hi && ((uint32_t)cur == j || same_bytes(a.text + (uint32_t)cur, p, len))) {
            if ((uint32_t)cur > j) atomicMin(&a.table[slot].key, me);
            return slot;
        }
        if (probes >= TOK_MAX_PROBE) {
            atomicOr(&a.tc->fail, TOK_FAIL_TABLE);
            return 0;
        }
        slot = (slot + 1) & a.tmask;
    }
}

// LDS of one wave of the hashing kernels: its 4 KiB of text (+ 16 bytes behind it: a token may start on the unit's last byte),
// the bound bits of those bytes (+ 64 behind), the list of token starts (reused, from its front, as the list of the tokens whose
// first look at the table did not settle them) and those tokens' indices
struct alignas(16) TokUnitLds {
    uint32_t text[TOK_WPW * TOK_WIN / 4 + 4];
    uint32_t bound[TOK_WPW * TOK_WIN / 32 + 2];
    uint16_t list[TOK_LIST_CAP];
    uint16_t pend_t[TOK_LIST_CAP];
    uint32_t kept[TOK_WPW * TOK_WIN / 32];  // filter mode: a bit per byte of the unit — a token the filter keeps starts here
};

// the token that starts at byte `pos` of the staged unit: its inline key (length 1 .. 7 in the top byte, bytes below) and
// whether it is that short at all
__device__ __forceinline__ bool tok_key_at(const TokUnitLds &s, uint32_t pos, unsigned long long *key) {
    const uint32_t aw = pos >> 2, sh = pos & 3u;
    const uint32_t w0 = s.text[aw], w1 = s.text[aw + 1], w2 = s.text[aw + 2];
    const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh), hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
    const uint32_t bi = pos + 1;
    const uint32_t win = __builtin_amdgcn_alignbit(s.bound[(bi >> 5) + 1], s.bound[bi >> 5], bi & 31u);  // bound bits of bytes pos+1 ..
    const uint32_t len = (uint32_t)__builtin_ctz(win | 0x40u) + 1u;  // 1 .. 7 (the key is only used when the token is that short)
    const unsigned long long bytes = (((unsigned long long)hi << 32) | lo) & ((1ull << (8 * len)) - 1ull);
    *key = ((unsigned long long)len << 56) | bytes;
    return (win & 0x7Fu) != 0u;  // the token ends within 7 bytes
}

// unaligned 32 bits of the staged text at byte `pos`
__device__ __forceinline__ uint32_t tok_lds_u32(const TokUnitLds &s, uint32_t pos) {
    return __builtin_amdgcn_alignbyte(s.text[(pos >> 2) + 1], s.text[pos >> 2], pos & 3u);
}

// A MEDIUM token — 8 .. TOK_MED_LEN bytes, all of them inside the staged text (unit + 16 bytes): insertions, del:11288:9,
// amino-acid mutations.  tok_long for such a token without its first three dependent global round trips: the end from the
// staged bound bits, the filter's verdict and the hash (the same hash) from the staged bytes; what is left is the probe chain —
// slot, slot again past the caches, the entry's bytes — as long as a deferred short token's.  (Every token of 8+ bytes through
// tok_long made a text with 6 % such tokens 44 % slower to hash: 1M rows with indels 384 us against 266.)
// -> false: not a medium token (tok_long takes it).
constexpr uint32_t TOK_MED_LEN = 24;
__device__ __forceinline__ bool tok_medium(const TokArgs &a, const TokUnitLds &s, uint32_t pos, uint32_t staged, uint32_t j, uint32_t *slot_out,
                                           bool *invalid) {
    const uint32_t bi = pos + 1;
    const uint32_t win = __builtin_amdgcn_alignbit(s.bound[(bi >> 5) + 1], s.bound[bi >> 5], bi & 31u);  // bound bits of bytes pos+1 ..
    if (win == 0u) return false;  // longer than 32 bytes
    const uint32_t len = (uint32_t)__builtin_ctz(win) + 1u;
    if (len < 8u || len > TOK_MED_LEN || pos + len > staged) return false;
    if (a.flt.on) {
        const int v = tok_classify(a.flt, len, [&](uint32_t i) { return (tok_lds_u32(s, pos + (i & ~3u)) >> (8 * (i & 3u))) & 0xFFu; });
        if (v != TOKV_KEEP) {
            *invalid = v == TOKV_INVALID;
            if (v == TOKV_INVALID) tok_note_invalid(a, j, len);
            *slot_out = TOK_NONE;
            return true;
        }
    }
    uint32_t h = 0x9747B28Cu ^ len, k = 0;
    for (; k + 4 <= len; k += 4) h = mur_step(h, tok_lds_u32(s, pos + k));
    if (len & 3u) h = mur_step(h, tok_lds_u32(s, pos + k) & ((1u << (8 * (len & 3u))) - 1u));
    h = mur_final(h);
    const uint32_t hi = 0x80000000u | (h & 0x7FFF0000u) | (len & 0xFFFFu);  // 1 : tag15 : len16 (tok_long's word)
    const unsigned long long me = ((unsigned long long)hi << 32) | j;
    uint32_t slot = h & a.tmask;
    for (int probes = 0;; probes++) {
        unsigned long long cur = a.table[slot].key;
        if (cur == TOK_EMPTY || (uint32_t)(cur >> 32) == hi)  // free, or maybe this token: the state past the caches (the offset moves)
            cur = __hip_atomic_load(&a.table[slot].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (cur == TOK_EMPTY) {
            cur = atomicCAS(&a.table[slot].key, TOK_EMPTY, me);
            if (cur == TOK_EMPTY) break;
        }
        if ((uint32_t)(cur >> 32) == hi) {
            bool same = (uint32_t)cur == j;
            if (!same) {  // the entry's occurrence (global) against this one (staged)
                const uint8_t *e = a.text + (uint32_t)cur;
                uint32_t diff = 0;
                for (k = 0; k + 4 <= len; k += 4) diff |= ldu32(e + k) ^ tok_lds_u32(s, pos + k);
                if (len & 3u) diff |= (ldu32(e + k) ^ tok_lds_u32(s, pos + k)) & ((1u << (8 * (len & 3u))) - 1u);
                same = diff == 0u;
            }
            if (same) {
                if ((uint32_t)cur > j) atomicMin(&a.table[slot].key, me);
                break;
            }
        }
        if (probes >= TOK_MAX_PROBE) {
            atomicOr(&a.tc->fail, TOK_FAIL_TABLE);
            slot = 0;
            break;
        }
        slot = (slot + 1) & a.tmask;
    }
    *slot_out = slot;
    return true;
}

// Tokens [0, n_tok) of the list -> table; the slot of token t is stored at out[t].
// Phase 1, U tokens per lane and round: all their LDS reads, then all their table slots — ONE 16-byte plain load each — in
// flight together; a token whose slot shows its key is settled there (its first offset lowered when this occurrence is
// earlier).  Everything else — a slot that looks free (the token is new, or this XCD's L2 / this CU's L1 holds the line as it
// was before another XCD put the token in), a slot taken by another token, a token of 8 bytes or more — is DEFERRED: its list
// position goes to the front of the list, which the rounds have already consumed.
// Phase 2: the deferred tokens, 64 at a time, through the probe chain that reads past the caches.  (Settling them inside
// the rounds made every round as long as its slowest lane's chain — plain load, load past the caches, compare-and-swap,
// offset: 100 us at 100k rows against 30 for the rounds alone.)
__device__ __forceinline__ void tok_unit_lookup(const TokArgs &a, TokUnitLds &s, uint32_t text0, uint32_t n_tok, uint32_t *out, uint32_t staged) {
    constexpr int U = TOK_LOOKUP_U;
    const int lane = threadIdx.x & 63;
    uint32_t n_pend = 0;
    for (uint32_t r0 = 0; r0 < n_tok; r0 += 64 * U) {
        bool act[U], inl[U];
        uint32_t pos[U], slot[U];
        unsigned long long key[U];
        uint4 q[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t t = r0 + u * 64 + lane;
            act[u] = t < n_tok;
            pos[u] = act[u] ? (uint32_t)s.list[t] : 0u;
            inl[u] = tok_key_at(s, pos[u], &key[u]);
            if (a.flt.on && act[u] && inl[u]) {  // the filter's verdict on a short token, from the bytes in its key
                const unsigned long long kb = key[u];
                const int v = tok_classify(a.flt, (uint32_t)(kb >> 56), [&](uint32_t i) { return (uint32_t)(kb >> (8 * i)) & 0xFFu; });
                if (v != TOKV_KEEP) {  // dropped: no table entry, no CSR entry
                    act[u] = false;
                    out[t] = TOK_NONE;
                    if (v == TOKV_INVALID) tok_note_invalid(a, text0 + pos[u], (uint32_t)(kb >> 56));
                }
            }
            slot[u] = tok_hash_inline(key[u]) & a.tmask;
            q[u] = *reinterpret_cast<const uint4 *>(&a.table[(act[u] && inl[u] && !(a.dbg & 2)) ? slot[u] : 0u]);
            if (a.dbg & 2) {  // (timing experiment: every token "found" without a look at the table)
                q[u].x = (uint32_t)key[u];
                q[u].y = (uint32_t)(key[u] >> 32);
                q[u].z = 0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t t = r0 + u * 64 + lane;
            const bool found = act[u] && inl[u] && ((((unsigned long long)q[u].y << 32) | q[u].x) == key[u]);
            if (found) {
                const uint32_t j = text0 + pos[u];
                if (j < q[u].z && !(a.dbg & 1)) atomicMin(&a.table[slot[u]].first, j);
                out[t] = slot[u];
                if (a.flt.on) atomicOr(&s.kept[pos[u] >> 5], 1u << (pos[u] & 31u));
            }
            const bool pend = act[u] && !found && !(a.dbg & 16);
            const unsigned long long bal = __ballot(pend);
            if (pend) {
                const uint32_t idx = n_pend + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                s.list[idx] = (uint16_t)pos[u];  // (idx <= tokens consumed so far: nothing unread is overwritten)
                s.pend_t[idx] = (uint16_t)t;
            }
            n_pend += (uint32_t)__popcll(bal);
            if ((a.dbg & 16) && act[u] && !found) out[t] = slot[u];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if ((a.dbg & 32) && lane == 0) {  // (counting experiment: deferred tokens, waves)
        atomicAdd((unsigned *)&a.tc->pad_[0], n_pend);
        atomicAdd((unsigned *)&a.tc->pad_[1], 1u);
    }
    for (uint32_t p0 = 0; p0 < n_pend; p0 += 64) {
        const uint32_t p = p0 + lane;
        if (p >= n_pend) continue;
        const uint32_t pos = s.list[p], t = s.pend_t[p];
        unsigned long long key;
        const bool inl = tok_key_at(s, pos, &key);
        const uint32_t j = text0 + pos;
        uint32_t sl;
        if (!inl) {
            bool inval = false;
            if ((a.dbg & 8) || !tok_medium(a, s, pos, staged, j, &sl, &inval)) sl = tok_long(a, j, &inval);
        } else {
            uint32_t slot = tok_hash_inline(key) & a.tmask;
            // the slot's state past the caches, key and first offset requested together
            unsigned long long cur = __hip_atomic_load(&a.table[slot].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            uint32_t first = __hip_atomic_load(&a.table[slot].first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            sl = 0;
            for (int probes = 0;; probes++) {
                if (cur == TOK_EMPTY) {
                    cur = atomicCAS(&a.table[slot].key, TOK_EMPTY, key);
                    if (cur == TOK_EMPTY) {
                        cur = key;
                        first = 0xFFFFFFFFu;
                    } else if (cur == key)  // another wave put it there since the load: its offset may have landed by now
                        first = __hip_atomic_load(&a.table[slot].first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (cur == key) {
                    if (j < first) atomicMin(&a.table[slot].first, j);
                    sl = slot;
                    break;
                }
                if (probes >= TOK_MAX_PROBE) {  // table too full: the host enlarges it and runs again
                    atomicOr(&a.tc->fail, TOK_FAIL_TABLE);
                    break;
                }
                slot = (slot + 1) & a.tmask;
                cur = __hip_atomic_load(&a.table[slot].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                first = __hip_atomic_load(&a.table[slot].first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (a.flt.on && sl != TOK_NONE) atomicOr(&s.kept[pos >> 5], 1u << (pos & 31u));
        out[t] = sl;
    }
}

// A wave owns a UNIT of TOK_WPW consecutive windows (4 KiB of text).  The unit's text and bound bits are staged in the wave's
// LDS with coalesced loads (one round trip), the positions of its token starts go into a list in LDS (prefix sums of the
// lanes' bit counts: the list is in text order, so token t of the list is token `first token of the unit + t` of the whole
// input), and the lanes then take tokens from the list — every lane busy whatever the token lengths, the per-token stores
// coalesced, nothing but the table itself read from global memory per token.  (Round 3 read every token's bytes and bound
// bits with scattered global loads and compared them with the bytes of the entry's first occurrence, somewhere in the text:
// three dependent round trips per token instead of one; 101 us at 100k rows.)
// masks: start / bound masks of the unit's four windows per lane (from k_tok_scan's arrays, or computed by the caller).
// WPW = windows the wave takes: TOK_WPW (a whole unit), or 1 — the first launches of a build run with most of the chip idle and
// cost one wave's latency each (~19 us for a 4 KiB unit of new tokens: nine rounds of the deferred phase): in the HEAD launch a
// unit is split over four waves (19.5 -> 15 us).  (The sample launch too: 18.3 -> 13.9 us on the benchmark generator, but its
// 1 900 waves then storm among themselves on inputs whose common tokens appear late — forest 115 -> 142 us, sorted 101 -> 121:
// TokArgs::fine_head bit 1, off.)
template <int WPW>
__device__ __forceinline__ void tok_hash_unit(const TokArgs &a, uint32_t win0, TokUnitLds &s, const uint4 (&v)[WPW],
                                              const uint32_t (&st)[WPW], const uint32_t (&bd)[WPW], uint32_t g0) {
    const int lane = threadIdx.x & 63;
    const uint32_t text0 = win0 * TOK_WIN;
#pragma unroll
    for (int u = 0; u < WPW; u++) {
        *reinterpret_cast<uint4 *>(&s.text[u * (TOK_WIN / 4) + 4 * lane]) = v[u];
        reinterpret_cast<uint16_t *>(s.bound)[u * (TOK_WIN / 16) + lane] = (uint16_t)bd[u];
    }
    if (a.flt.on)
        for (int i = lane; i < WPW * TOK_WIN / 32; i += 64) s.kept[i] = 0u;
    if (lane < 4) s.text[WPW * TOK_WIN / 4 + lane] = ldu32(a.text + text0 + WPW * TOK_WIN + 4 * lane);
    if (lane < 2) s.bound[WPW * TOK_WIN / 32 + lane] = a.boundbits[(text0 + WPW * TOK_WIN) / 32 + lane];
    int cnt[WPW], inc[WPW];
    uint32_t total = 0;
#pragma unroll
    for (int u = 0; u < WPW; u++) {
        cnt[u] = __popc(st[u]);
        inc[u] = tok_wave_incl_scan(cnt[u]);
        total += (uint32_t)__builtin_amdgcn_readlane(inc[u], 63);
    }
    // the list holds TOK_LIST_CAP starts: a unit with more (rows of a byte or two) is taken one window at a time
    const int n_sub = total <= TOK_LIST_CAP ? 1 : WPW;
    uint32_t done = 0;
    for (int sub = 0; sub < n_sub; sub++) {
        uint32_t n_tok = 0;
#pragma unroll
        for (int u = 0; u < WPW; u++) {
            if (n_sub > 1 && u != sub) continue;
            uint32_t o = n_tok + (uint32_t)(inc[u] - cnt[u]);
            for (uint32_t b = st[u]; b; b &= b - 1) s.list[o++] = (uint16_t)(u * TOK_WIN + 16 * lane + __builtin_ctz(b));
            n_tok += (uint32_t)__builtin_amdgcn_readlane(inc[u], 63);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!(a.dbg & 4)) tok_unit_lookup(a, s, text0, n_tok, a.tokslot + g0 + done, (uint32_t)(WPW * TOK_WIN + 16));
        else  // (timing experiment: a defined slot for every token all the same — k_tok_ids follows them)
            for (uint32_t t = threadIdx.x & 63; t < n_tok; t += 64) a.tokslot[g0 + done + t] = 0u;
        done += n_tok;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the next sub-unit overwrites the list)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (a.flt.on)  // the kept-token bits of the wave's windows: 32 words per window, coalesced
        for (int i = lane; i < WPW * TOK_WIN / 32; i += 64) a.keptbits[text0 / 32 + i] = s.kept[i];
}

// ------------------------------------------------------------------------------------------------
// k_tok_scan: a block = 64 KiB of text — masks stored, token starts counted per window and block.
__global__ __launch_bounds__(1024) void k_tok_scan(TokArgs a, uint32_t blk0) {
    __shared__ unsigned s_cnt[TOK_SCAN_WINS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t sepx4 = (uint32_t)a.sep * 0x01010101u;
    const uint32_t blk = blk0 + (uint32_t)blockIdx.x;
    const uint32_t win0 = blk * TOK_SCAN_WINS + wave * TOK_WPW;  // this wave's first window
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
        uint32_t start, bound;
        tok_masks(v[i], rb[i], sepx4, prev_sep, lane, &start, &bound, &prev_sep);
        reinterpret_cast<uint16_t *>(a.startbits)[(w0 >> 4) + lane] = (uint16_t)start;
        reinterpret_cast<uint16_t *>(a.boundbits)[(w0 >> 4) + lane] = (uint16_t)bound;
        const int inc = tok_wave_incl_scan(__popc(start));
        if (lane == 63) s_cnt[wave * TOK_WPW + i] = (unsigned)inc;
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
template <int THREADS>
__device__ __forceinline__ void tok_scan_block(uint32_t *data, uint32_t n, unsigned *total_out) {
    __shared__ unsigned s_w[THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned carry = *total_out;  // (read by every thread before the barrier below, written by thread 0 after it)
    const uint32_t per = ((n + THREADS - 1u) / THREADS + 3u) & ~3u;
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
    for (int w = 0; w < THREADS / 64; w++) {
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

__global__ __launch_bounds__(1024) void k_scan_single(uint32_t *data, uint32_t n, unsigned *total_out) { tok_scan_block<1024>(data, n, total_out); }

// ------------------------------------------------------------------------------------------------
// Units [unit0, n_units).  sample = 0: all of them, in order.  sample = S > 1, part 0: every S-th unit (unit0, unit0 + S, ...);
// part 1: the others.  (The hash runs in THREE launches — the first units, a sample spread over the text, the rest: when all
// 8 000 waves start at once, every token that many rows carry is met by thousands of waves while its slot is still free, and
// each of them claims it by compare-and-swap: same-address atomics serialise at ~11 ns.  The waves of the earlier launches are
// few, and what they insert the later ones find by a plain load.)
// scan_n > 0 (the HEAD launch of a build: units of the first 64 KiB, which need no prefix over the blocks): the launch's LAST block
// is not a hash block — it turns the blocks' token totals into prefixes (what a k_scan_single launch of its own did between
// k_tok_scan and the hash: ~5 us of stream time; here it is under the head's 15).
template <int WPW>
__global__ __launch_bounds__(256) void k_tok_hash(TokArgs a, uint32_t unit0, uint32_t n_units, uint32_t sample, uint32_t part, uint32_t scan_n) {
    __shared__ TokUnitLds s_unit[4];
    if (scan_n && blockIdx.x == gridDim.x - 1) {  // (block-uniform)
        tok_scan_block<256>(a.blkbase, scan_n, &a.tc->nnz);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t SPLIT = TOK_WPW / WPW;  // waves per unit
    const uint32_t widx = (uint32_t)blockIdx.x * 4 + wave, idx = widx / SPLIT;
    uint32_t unit;
    if (sample <= 1) unit = unit0 + idx;
    else if (part == 0) unit = unit0 + idx * sample;
    else unit = unit0 + (idx / (sample - 1)) * sample + idx % (sample - 1) + 1;
    if (unit >= n_units) return;
    const uint32_t win0 = unit * TOK_WPW + (widx % SPLIT) * WPW;
    uint4 v[WPW];
    uint32_t st[WPW], bd[WPW];
#pragma unroll
    for (int u = 0; u < WPW; u++) {
        v[u] = *reinterpret_cast<const uint4 *>(a.text + (win0 + u) * TOK_WIN + 16 * lane);
        st[u] = reinterpret_cast<const uint16_t *>(a.startbits)[(win0 + u) * (TOK_WIN / 16) + lane];
        bd[u] = reinterpret_cast<const uint16_t *>(a.boundbits)[(win0 + u) * (TOK_WIN / 16) + lane];
    }
    // (a unit lies in one scan block; scan_n: the units of this launch lie in block 0, whose prefix — being computed next door — is 0)
    const uint32_t g0 = (scan_n ? 0u : a.blkbase[win0 / TOK_SCAN_WINS]) + a.winbase[win0];
    tok_hash_unit<WPW>(a, win0, s_unit[wave], v, st, bd, g0);
}

// ------------------------------------------------------------------------------------------------
// k_tok_rows: thread r <= n_rows: indptr[r] = token starts in front of row_off[r] (the window's prefix + the bits of the
// window in front of the offset).  The same grid then walks the table: a slot in use sets the bit `first occurrence`
// at the byte offset it holds.
// The bind's row statistics come out of the same pass (a k_maxlen launch over the new indptr was 11 us of a 220 us step):
// small[0] = longest row, [1] = a negative row length was seen, [2] = rows of <= 2 * PG_MAX_DIST tokens, [3] = indptr[n_rows],
// [4] = indptr[0] — the words in front of the tokeniser's counters (k_maxlen's layout; cleared by k_tok_clear).  A row's length
// is the next lane's count minus its own; the last lane of a wave counts its right neighbour's offset itself.
__global__ __launch_bounds__(256) void k_tok_rows(TokArgs a) {
    __shared__ int s_k[4], s_short[4], s_bad[4];
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, nth = gridDim.x * 256u;
    const int lane = threadIdx.x & 63;
    // (filter mode: the CSR holds the tokens the filter kept — their bits and prefixes instead of the token starts')
    const uint32_t *bits = a.flt.on ? a.keptbits : a.startbits;
    const uint32_t *win = a.flt.on ? a.keptwin : a.winbase, *blk = a.flt.on ? a.keptblk : a.blkbase;
    int *small = reinterpret_cast<int *>(a.tc) - 8;
    auto count_at = [&](long long ol) -> uint32_t {
        ol = ol < 0 ? 0 : (ol > (long long)a.T ? (long long)a.T : ol);  // (malformed offsets are reported below)
        const uint32_t o = (uint32_t)ol;
        const uint32_t w = o / TOK_WIN;
        uint32_t cnt = blk[w / TOK_SCAN_WINS] + win[w];
        for (uint32_t q = w * (TOK_WIN / 32); q < (o >> 5); q++) cnt += (uint32_t)__popc(bits[q]);
        return cnt + (uint32_t)__popc(bits[o >> 5] & ((1u << (o & 31u)) - 1u));
    };
    int kmax = 0, n_short = 0, bad = 0;
    const uint32_t n_rows = (uint32_t)a.n_rows;
    for (uint32_t rb = tid - (uint32_t)lane; rb <= n_rows; rb += nth) {  // (wave-uniform trips: the lanes exchange their counts)
        const uint32_t r = rb + (uint32_t)lane;
        const bool in = r <= n_rows;
        const long long off = in ? a.row_off[r] - a.base : 0ll;
        const uint32_t cnt = in ? count_at(off) : 0u;
        if (in) a.indptr[r] = (int)cnt;
        uint32_t nxt = (uint32_t)__shfl_down((int)cnt, 1);
        long long nxt_off = __shfl_down(off, 1);
        if (lane == 63 && r < n_rows) {
            nxt_off = a.row_off[r + 1] - a.base;
            nxt = count_at(nxt_off);
        }
        if (r < n_rows) {
            // the offsets are validated here as well (k_tok_rowbits does it where it runs; a build whose row bits were set by
            // k_tok_clear has nobody else to do it): monotone, inside the text, and — strict — covering all of it
            if (off < 0 || nxt_off < off || nxt_off > (long long)a.T ||
                (a.strict && ((r == 0 && off != 0) || (r == n_rows - 1 && nxt_off != (long long)a.T))))
                atomicOr(&a.tc->fail, TOK_FAIL_ROWOFF);
            int k = (int)(nxt - cnt);
            if (k < 0) {
                bad = 1;
                k = 0;
            }
            n_short += k <= 2 * PG_MAX_DIST ? 1 : 0;
            kmax = max(kmax, k);
        }
        if (r == n_rows) small[3] = (int)cnt;
        if (r == 0) small[4] = (int)cnt;
    }
    if (blockIdx.x * 256u <= n_rows) {  // (block-uniform: the block had rows)
        for (int s = 32; s > 0; s >>= 1) {
            kmax = max(kmax, __shfl_xor(kmax, s));
            n_short += __shfl_xor(n_short, s);
            bad |= __shfl_xor(bad, s);
        }
        if (lane == 0) {
            s_k[threadIdx.x >> 6] = kmax;
            s_short[threadIdx.x >> 6] = n_short;
            s_bad[threadIdx.x >> 6] = bad;
        }
        __syncthreads();
        if (threadIdx.x == 0) {  // one set of atomics per block, and only where they change something
            kmax = max(max(s_k[0], s_k[1]), max(s_k[2], s_k[3]));
            n_short = s_short[0] + s_short[1] + s_short[2] + s_short[3];
            if (kmax > __hip_atomic_load(small, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(small, kmax);
            if (n_short > 0) atomicAdd(small + 2, n_short);
            if (s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) atomicOr(small + 1, 1);
        }
    }
    if (a.any_ids) {  // slots serve as ids: no vocabulary walk; the row-start bits are cleared for the next build here (k_voc_ids' job)
        if (a.rows_clear_after) {
            uint4 *rb4 = reinterpret_cast<uint4 *>(a.rowbits);
            for (uint32_t i = tid, n16 = (a.T_pad / 32u + 16u) / 4u; i < n16; i += nth) rb4[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        return;
    }
    for (uint32_t s = tid; s <= a.tmask; s += nth) {
        const TokSlot e = a.table[s];
        if (e.key != TOK_EMPTY) {
            const uint32_t o = tok_entry_offset(e);
            atomicOr(&a.firstbits[o >> 5], 1u << (o & 31u));
        }
    }
}

// k_tok_empties (filter mode): the EMPTY tokens of every row — str.split hands them to the patterns too, and for every feature
// type but nextclade_aa (whose insertion pattern is ^$) and raw they are "invalid" and printed as such (breakfast.py:182-184).
// A row's span [s, e) splits into (separators in it) + 1 tokens, of which (token starts in it) are not empty: both counts
// from the bit arrays — bound bits are separators and row starts, so separators = bound bits - (a token starts on s).
__global__ __launch_bounds__(256) void k_tok_empties(TokArgs a) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    uint32_t cnt = 0;
    if (r < (uint32_t)a.n_rows) {
        const uint32_t s = (uint32_t)(a.row_off[r] - a.base);
        const uint32_t e = a.span_len ? s + (uint32_t)a.span_len[r] : (uint32_t)(a.row_off[r + 1] - a.base);
        uint32_t nb = 0, ns = 0;
        for (uint32_t q = s >> 5; q <= ((e - 1) >> 5) && e > s; q++) {
            uint32_t m = ~0u;
            if (q == (s >> 5)) m &= ~0u << (s & 31u);
            if (q == ((e - 1) >> 5)) m &= ~0u >> (31u - ((e - 1) & 31u));
            nb += (uint32_t)__popc(a.boundbits[q] & m);
            ns += (uint32_t)__popc(a.startbits[q] & m);
        }
        const uint32_t first_is_start = e > s ? (a.startbits[s >> 5] >> (s & 31u)) & 1u : 0u;
        cnt = (nb - first_is_start) + 1u - ns;
        if (a.row_empties) a.row_empties[r] = cnt;  // (the host interleaves the "invalid" lines of empty and other tokens by these)
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&a.tc->n_empty, cnt);
}

// first-occurrence bits per window (32 words): 8 lanes per window, a uint4 each; a block = the 64 windows of a scan block
__global__ __launch_bounds__(512) void k_voc_count(const uint32_t *__restrict__ bits, uint32_t *__restrict__ win_out, uint32_t *__restrict__ blk_out) {
    __shared__ unsigned s_cnt[TOK_SCAN_WINS];
    const uint32_t t = blockIdx.x * 512u + threadIdx.x;  // word quad (T_pad is a multiple of the 64 KiB a block covers)
    const uint4 q = reinterpret_cast<const uint4 *>(bits)[t];
    uint32_t c = (uint32_t)(__popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w));
    c += __shfl_xor(c, 1);
    c += __shfl_xor(c, 2);
    c += __shfl_xor(c, 4);
    if ((threadIdx.x & 7u) == 0) s_cnt[threadIdx.x >> 3] = c;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int v = (int)s_cnt[threadIdx.x];
        const int inc = tok_wave_incl_scan(v);
        win_out[blockIdx.x * TOK_SCAN_WINS + threadIdx.x] = (uint32_t)(inc - v);
        if (threadIdx.x == 63) blk_out[blockIdx.x] = (uint32_t)inc;
    }
}

// one thread per slot in use: id = first occurrences in front of its offset
// voc_scan_n > 0 (texts up to TOK_FUSE_ROWBITS_BYTES: at most 1024 scan blocks): `vocblk` holds the blocks' TOTALS as k_voc_count
// left them, and every block of this kernel turns them into prefixes for itself, in LDS (a few KB of L2 hits per block) — no
// one-block k_scan_single launch between the two (~5 us of stream time); block 0 leaves the grand total in tc->n_vocab.
__global__ __launch_bounds__(256) void k_voc_ids(TokArgs a, uint32_t voc_scan_n) {
    __shared__ uint32_t s_blk[1024];
    __shared__ unsigned s_w4[4];
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, nth = gridDim.x * 256u;
    if (voc_scan_n) {  // (block-uniform) thread t owns totals [4t, 4t + 4)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = 4u * threadIdx.x + j < voc_scan_n ? a.vocblk[4u * threadIdx.x + j] : 0u;
        const uint32_t sum = x[0] + x[1] + x[2] + x[3];
        const uint32_t inc = (uint32_t)tok_wave_incl_scan((int)sum);
        if (lane == 63) s_w4[wave] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (int w = 0; w < wave; w++) run += s_w4[w];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s_blk[4 * threadIdx.x + j] = run;
            run += x[j];
        }
        if (blockIdx.x == 0 && threadIdx.x == 255) a.tc->n_vocab = run;  // (the last thread's run is the sum of everything)
        __syncthreads();
    }
    for (uint32_t s = tid; s <= a.tmask; s += nth) {
        const TokSlot e = a.table[s];
        if (e.key == TOK_EMPTY) continue;
        const uint32_t o = tok_entry_offset(e);
        const uint32_t w = o / TOK_WIN;
        uint32_t cnt = (voc_scan_n ? s_blk[w / TOK_SCAN_WINS] : a.vocblk[w / TOK_SCAN_WINS]) + a.vocwin[w];
        for (uint32_t q = w * (TOK_WIN / 32); q < (o >> 5); q++) cnt += (uint32_t)__popc(a.firstbits[q]);
        cnt += (uint32_t)__popc(a.firstbits[o >> 5] & ((1u << (o & 31u)) - 1u));
        a.tabid[s] = (int)cnt;  // (an array of its own: k_tok_ids gathers from 4 bytes per slot, not from the 16-byte slots)
    }
    // the row-start bits have been used (k_tok_scan): cleared here, so that the NEXT build finds them zero and sets its own in
    // k_tok_clear — no k_tok_rowbits launch between the clearing and the scan (4.9 + 1.3 us of a 212 us step)
    // (texts up to TOK_FUSE_ROWBITS_BYTES: beyond — 1M rows, 333 MB — the million atomics inside k_tok_clear and 41 MB of zeros
    // here cost 7 us more than the launch they save)
    if (a.rows_clear_after) {
        uint4 *rb4 = reinterpret_cast<uint4 *>(a.rowbits);
        for (uint32_t i = tid, n16 = (a.T_pad / 32u + 16u) / 4u; i < n16; i += nth) rb4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}

template <int CACHE>
__global__ __launch_bounds__(256) void k_tok_ids(TokArgs a) {
    // (grid-stride over the REAL token count: a grid sized by the most tokens the text can hold — T/2 + rows — was three
    // quarters blocks that read the count and left)
    // The gather id = tabid[slot] is what the kernel costs (a scattered 4-byte load per token: 43.5M of them at 1M rows, from a
    // 16 MB array), and most tokens of a profile are the few hundred mutations nearly every row carries: a direct-mapped cache
    // {slot, id} in the block's LDS answers those (an entry is one 8-byte word, read and written whole by relaxed atomics: a stale
    // or lost entry is only a miss).  2048 entries (16 KB: 4096 are no better, 8192 cost occupancy — 213 us): 1M rows 182 -> 139 us, 100k rows
    // 13.3 -> 12.4.  CACHE = 0 gathers directly (BFK_TOK_IDCACHE=0).
    __shared__ unsigned long long s_cache[CACHE ? CACHE : 1];
    if (CACHE) {
        for (int i = threadIdx.x; i < CACHE; i += 256) s_cache[i] = ~0ull;
        __syncthreads();
    }
    const uint32_t nnz = a.tc->nnz;
    const uint32_t nth = gridDim.x * 256u * 4u;
    auto id_of = [&](uint32_t slot) -> uint32_t {
        if (!CACHE) return (uint32_t)a.tabid[slot];
        const uint32_t c = (slot ^ (slot >> 12)) & (uint32_t)(CACHE - 1);
        // (all waves of the block read and write the entries: relaxed 64-bit atomics — one ds_read_b64 / ds_write_b64 each — so that
        // an entry is never torn into the tag of one token and the id of another)
        const unsigned long long e = __hip_atomic_load(&s_cache[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)(e >> 32) == slot) return (uint32_t)e;
        const uint32_t id = (uint32_t)a.tabid[slot];
        __hip_atomic_store(&s_cache[c], ((unsigned long long)slot << 32) | id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return id;
    };
    for (uint32_t g0 = (blockIdx.x * 256u + threadIdx.x) * 4u; g0 < nnz; g0 += nth) {
        const uint4 s = *reinterpret_cast<const uint4 *>(a.tokslot + g0);
        uint4 o;
        o.x = id_of(s.x);
        o.y = g0 + 1 < nnz ? id_of(s.y) : 0u;
        o.z = g0 + 2 < nnz ? id_of(s.z) : 0u;
        o.w = g0 + 3 < nnz ? id_of(s.w) : 0u;
        *reinterpret_cast<uint4 *>(a.indices + g0) = o;
    }
}

// filter mode: the slots of ALL tokens lie in text order in `tokslot` (TOK_NONE where the filter dropped one); the CSR holds the
// kept ones.  A wave per unit of 4 KiB: the unit's tokens start at the all-token prefix of its first window, its kept tokens at
// the kept-token prefix; kept tokens are written densely, in order.  `tokslot` and `indices` are different arrays here.
__global__ __launch_bounds__(256) void k_tok_ids_kept(TokArgs a, uint32_t n_units) {
    const int lane = threadIdx.x & 63;
    const uint32_t unit = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (unit >= n_units) return;
    const uint32_t win0 = unit * TOK_WPW, sb = win0 / TOK_SCAN_WINS;
    const uint32_t g0 = a.blkbase[sb] + a.winbase[win0];
    // (the unit's end: the next unit's start, or — for the last unit of a scan block — the next block's base)
    const uint32_t win1 = win0 + TOK_WPW;
    const uint32_t g1 = (win1 % TOK_SCAN_WINS) ? a.blkbase[sb] + a.winbase[win1] : a.blkbase[sb + 1];
    uint32_t o = a.keptblk[sb] + a.keptwin[win0];
    for (uint32_t g = g0; g < g1; g += 64) {
        const uint32_t s = g + lane < g1 ? a.tokslot[g + lane] : TOK_NONE;
        const bool keep = s != TOK_NONE;
        const unsigned long long bal = __ballot(keep);
        if (keep) a.indices[o + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (uint32_t)a.tabid[s];
        o += (uint32_t)__popcll(bal);
    }
}

// ------------------------------------------------------------------------------------------------
#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// k_tok_clear: everything a build has to find cleared, in ONE launch: the bit arrays (zero), the vocabulary table (all ones:
// free slots), the separator padding behind the text, the 16 counter words.  (Four memsets of the runtime, ~5 us of stream
// time each whatever their size: 21-26 of a 230 us step at 100k rows.)
__global__ __launch_bounds__(256) void k_tok_clear(uint4 *zero, size_t n_zero16, uint4 *ones, size_t n_ones16, uint8_t *pad, uint32_t pad_bytes,
                                                   uint8_t pad_byte, int *small, TokRows rows) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    // rows.rowbits != NULL: the row-start bits are zero already (the build before this one cleared them when it was done with
    // them, k_voc_ids) and are NOT part of `zero`: this launch sets them — first, so that the atomics are under the streaming
    // stores.  Malformed offsets are skipped here and reported by k_tok_rows.
    if (rows.rowbits)
        for (size_t r = tid; r < (size_t)rows.n_rows; r += nth) {
            const long long o = rows.row_off[r] - rows.base, e = rows.row_off[r + 1] - rows.base;
            if (o >= 0 && e >= o && e <= (long long)rows.T) atomicOr(&rows.rowbits[(uint32_t)o >> 5], 1u << ((uint32_t)o & 31u));
        }
    for (size_t i = tid; i < n_zero16; i += nth) zero[i] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = tid; i < n_ones16; i += nth) ones[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (size_t i = tid; i < pad_bytes; i += nth) pad[i] = pad_byte;
    if (tid < 16) small[tid] = 0;
}

int launch_tok_clear(void *zero, size_t zero_bytes, void *ones, size_t ones_bytes, uint8_t *pad, uint32_t pad_bytes, uint8_t pad_byte, int *small,
                     hipStream_t st, const TokRows &rows) {
    const size_t n16 = zero_bytes / 16 + ones_bytes / 16;
    const unsigned blocks = (unsigned)std::min<size_t>(2048, std::max<size_t>(1, (n16 + 255) / 256 / 4));
    hipLaunchKernelGGL(k_tok_clear, dim3(blocks), dim3(256), 0, st, (uint4 *)zero, zero_bytes / 16, (uint4 *)ones, ones_bytes / 16, pad, pad_bytes,
                       pad_byte, small, rows);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

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
    if (a.n_rows > 0 && !a.rows_fused) {  // (rows_fused: k_tok_clear has set the bits)
        hipLaunchKernelGGL(k_tok_rowbits, dim3((a.n_rows + 255) / 256), dim3(256), 0, st, a);
        LAUNCH_CHECK();
    }
    unsigned hashed = 0;  // hash units done so far
    for (int k = 0; k < n_pieces; k++) {
        const unsigned b0 = n_pieces > 1 ? piece_blk[k] : 0u, b1 = n_pieces > 1 ? piece_blk[k + 1] : scan_blocks;
        if (piece_ev && hipStreamWaitEvent(st, piece_ev[k], 0) != hipSuccess) return (int)hipGetLastError();
        // the prefix over the blocks' token totals: a block of the head launch where there is one (the whole text in one piece, a
        // head inside the first scan block), a launch of its own otherwise
        const bool scan_in_head = n_pieces == 1 && a.head_units > 0 && (unsigned)a.head_units <= UNITS_PER_BLK &&
                                  scan_blocks * UNITS_PER_BLK > (unsigned)a.head_units && scan_blocks <= 16384 && !(a.dbg & 64);
        if (b1 > b0) {
            hipLaunchKernelGGL(k_tok_scan, dim3(b1 - b0), dim3(1024), 0, st, a, b0);
            LAUNCH_CHECK();
            if (!scan_in_head) {
                hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, a.blkbase + b0, b1 - b0, &a.tc->nnz);
                LAUNCH_CHECK();
            }
        }
        if (ev && k == 0) (void)hipEventRecord(ev[1], st);
        // A token that starts near the end of the piece may reach into the next one (up to TOK_MAX_LEN bytes), whose text and
        // bound bits are not there yet: the last TOK_HOLD_UNITS units of a piece wait for the next piece's scan.
        const unsigned u0 = hashed;
        const unsigned u_end = b1 * UNITS_PER_BLK;
        const unsigned u1 = k + 1 == n_pieces ? u_end : (u_end > hashed + TOK_HOLD_UNITS ? u_end - TOK_HOLD_UNITS : hashed);
        unsigned uh = u0;
        if (k == 0 && a.head_units > 0 && u1 > u0 + (unsigned)a.head_units) {  // the first units by themselves
            uh = u0 + (unsigned)a.head_units;
            const unsigned sn = scan_in_head ? scan_blocks : 0u, extra = scan_in_head ? 1u : 0u;
            if (a.fine_head & 1) hipLaunchKernelGGL(k_tok_hash<1>, dim3(uh - u0 + extra), dim3(256), 0, st, a, u0, uh, 0u, 0u, sn);  // (a wave per window)
            else hipLaunchKernelGGL(k_tok_hash<TOK_WPW>, dim3((uh - u0 + 3) / 4 + extra), dim3(256), 0, st, a, u0, uh, 0u, 0u, sn);
            LAUNCH_CHECK();
        }
        if (u1 > uh) {
            const unsigned cnt = u1 - uh, S = (unsigned)a.sample;
            if (S > 1 && cnt >= 4 * S) {  // a sample spread over the text, then the rest
                const unsigned n0 = (cnt + S - 1) / S, n1 = cnt - n0;
                if (a.fine_head & 2) hipLaunchKernelGGL(k_tok_hash<1>, dim3(n0), dim3(256), 0, st, a, uh, u1, S, 0u, 0u);
                else hipLaunchKernelGGL(k_tok_hash<TOK_WPW>, dim3((n0 + 3) / 4), dim3(256), 0, st, a, uh, u1, S, 0u, 0u);
                LAUNCH_CHECK();
                hipLaunchKernelGGL(k_tok_hash<TOK_WPW>, dim3((n1 + 3) / 4), dim3(256), 0, st, a, uh, u1, S, 1u, 0u);
                LAUNCH_CHECK();
            } else {
                hipLaunchKernelGGL(k_tok_hash<TOK_WPW>, dim3((cnt + 3) / 4), dim3(256), 0, st, a, uh, u1, 0u, 0u, 0u);
                LAUNCH_CHECK();
            }
        }
        hashed = std::max(hashed, u1);
    }
    if (ev) (void)hipEventRecord(ev[2], st);
    if (a.flt.on) {  // the tokens the filter kept: counted per window / block like the token starts, prefix over the blocks
        hipLaunchKernelGGL(k_voc_count, dim3(scan_blocks), dim3(512), 0, st, a.keptbits, a.keptwin, a.keptblk);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, a.keptblk, scan_blocks, &a.tc->nnz_kept);
        LAUNCH_CHECK();
        if (a.n_rows > 0) {
            hipLaunchKernelGGL(k_tok_empties, dim3((a.n_rows + 255) / 256), dim3(256), 0, st, a);
            LAUNCH_CHECK();
        }
    }
    const bool any_ids = a.any_ids && !a.flt.on;
    hipLaunchKernelGGL(k_tok_rows, dim3(std::max(any_ids ? 256u : table_blocks, (unsigned)std::min(8192, (a.n_rows + 256) / 256))), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    if (any_ids) {  // the slot array IS the indices array, and stays as it is
        if (ev) (void)hipEventRecord(ev[3], st);
        return 0;
    }
    hipLaunchKernelGGL(k_voc_count, dim3(scan_blocks), dim3(512), 0, st, a.firstbits, a.vocwin, a.vocblk);
    LAUNCH_CHECK();
    const bool voc_scan_in_ids = scan_blocks <= 1024 && !(a.dbg & 64);
    if (!voc_scan_in_ids) {
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, a.vocblk, scan_blocks, &a.tc->n_vocab);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_voc_ids, dim3(table_blocks), dim3(256), 0, st, a, voc_scan_in_ids ? scan_blocks : 0u);
    LAUNCH_CHECK();
    if (a.flt.on) {
        hipLaunchKernelGGL(k_tok_ids_kept, dim3((scan_blocks * UNITS_PER_BLK + 3) / 4), dim3(256), 0, st, a, scan_blocks * UNITS_PER_BLK);
    } else {
        const unsigned id_blocks = std::max(1u, (unsigned)std::min<long long>((a.nnz_cap + 1023) / 1024, 4096));
        if (getenv("BFK_TOK_IDCACHE") && atoi(getenv("BFK_TOK_IDCACHE")) == 0) hipLaunchKernelGGL(k_tok_ids<0>, dim3(id_blocks), dim3(256), 0, st, a);  // (A/B)
        else hipLaunchKernelGGL(k_tok_ids<2048>, dim3(id_blocks), dim3(256), 0, st, a);
    }
    LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[3], st);
    return 0;
}

// ---- side-car cache: the two 64-bit hashes of every unique row's feature string ---------------------------------------------------
// What bfk_table_feature_hashes computes on the host (bfk_frontend.cpp: bytes_hash / bytes_hash2 over the string
// bfk_table_feature hands out — the row's bytes as they are when nothing is filtered, the kept non-empty tokens joined by the
// separator otherwise), for a table whose vocabulary never left the device.  One lane per unique row, over its first row's
// bytes in the (blanked) text: both hashes start from the string's length, so a filtered row is walked twice.
// (a lane walks its row eight bytes at a time — ByteWin — and judges every token once: the verdicts of the first 128 tokens wait
// in two registers for the second walk; one byte load per step made the kernel 25 ms at 1M rows)
struct ByteWin {
    const uint8_t *p;
    unsigned long long w;
    uint32_t base;
    __device__ __forceinline__ uint32_t at(uint32_t i) {
        if (i - base >= 8u) {
            base = i & ~7u;
            w = ldu64(p + base);  // (the text is padded: the last window may reach past the row)
        }
        return (uint32_t)(w >> (8u * (i - base))) & 0xFFu;
    }
};

__global__ __launch_bounds__(256) void k_row_hashes(RowHashArgs a) {
    const int u = (int)(blockIdx.x * 256 + threadIdx.x);
    if (u >= a.n_unique) return;
    const int r = a.first_row[u];
    ByteWin bw{a.text + (a.row_off[r] - a.base), 0ull, 0xFFFFFFF0u};
    const uint32_t L = (uint32_t)a.span_len[r];
    unsigned long long n = L, keep0 = 0, keep1 = 0;
    if (a.flt.on) {
        n = 0;
        uint32_t kept = 0, k = 0;
        for (uint32_t pos = 0; pos < L; k++) {
            uint32_t e = pos;
            while (e < L && bw.at(e) != a.sep) e++;
            if (e > pos && tok_classify(a.flt, e - pos, [&](uint32_t i) { return bw.at(pos + i); }) == TOKV_KEEP) {
                n += (e - pos) + (kept++ ? (uint32_t)a.pat.m : 0u);
                if (k < 64) keep0 |= 1ull << k;
                else if (k < 128) keep1 |= 1ull << (k - 64);
            }
            pos = e + 1;
        }
    }
    unsigned long long h1 = 0x9E3779B97F4A7C15ull ^ (n * 0xFF51AFD7ED558CCDull), h2 = 0xD6E8FEB86659FD93ull ^ (n * 0x9FB21C651E98DF25ull);
    unsigned long long acc = 0;
    uint32_t fill = 0;
    auto put = [&](uint32_t c) {
        acc |= (unsigned long long)c << (8 * fill);
        if (++fill == 8) {
            h1 = (h1 ^ acc) * 0xC4CEB9FE1A85EC53ull;
            h1 ^= h1 >> 29;
            h2 = (h2 ^ acc) * 0xFF51AFD7ED558CCDull;
            h2 ^= h2 >> 31;
            acc = 0;
            fill = 0;
        }
    };
    if (!a.flt.on) {
        // (a separator of several bytes stands in the text as runs of the stand-in byte, k_sepfold: the string holds the separator)
        for (uint32_t i = 0, run = 0; i < L; i++) {
            const uint32_t c = bw.at(i);
            if (a.pat.m > 1 && c == a.sep) {
                put(a.pat.b[run]);
                run = run + 1 == (uint32_t)a.pat.m ? 0u : run + 1;
            } else {
                put(c);
                run = 0;
            }
        }
    } else {
        uint32_t kept = 0, k = 0;
        for (uint32_t pos = 0; pos < L; k++) {
            uint32_t e = pos;
            while (e < L && bw.at(e) != a.sep) e++;
            bool keep = false;
            if (e > pos) {
                if (k < 64) keep = (keep0 >> k) & 1ull;
                else if (k < 128) keep = (keep1 >> (k - 64)) & 1ull;
                else keep = tok_classify(a.flt, e - pos, [&](uint32_t i) { return bw.at(pos + i); }) == TOKV_KEEP;
            }
            if (keep) {
                if (kept++) {
                    if (a.pat.m > 1)
                        for (int j = 0; j < a.pat.m; j++) put(a.pat.b[j]);
                    else
                        put(a.sep);
                }
                for (uint32_t i = pos; i < e; i++) put(bw.at(i));
            }
            pos = e + 1;
        }
    }
    h1 = (h1 ^ acc) * 0xC4CEB9FE1A85EC53ull;
    h2 = (h2 ^ acc) * 0xFF51AFD7ED558CCDull;
    a.out[2 * (size_t)u] = h1 ^ (h1 >> 32);
    a.out[2 * (size_t)u + 1] = h2 ^ (h2 >> 29);
}

int launch_row_hashes(const RowHashArgs &a, hipStream_t st) {
    if (a.n_unique <= 0) return 0;
    hipLaunchKernelGGL(k_row_hashes, dim3((unsigned)(a.n_unique + 255) / 256), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

}  // namespace bfk
