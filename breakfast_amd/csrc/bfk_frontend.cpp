// bfk_frontend.cpp — the host stages either side of the GPU path, native:
//   bfk_table_open     TSV reader            (read_input,            src/breakfast/breakfast.py:16-29)
//   bfk_table_prepare  filter + collapse + vocabulary/CSR in one pass over the feature bytes
//                                            (filter_features :116-190, collapse_duplicates :72-79,
//                                             sparse_feature_matrix :193-215)
//   bfk_table_write    clusters.tsv writer   (write_output :32-69)
// Pure host code (no HIP); the per-var-type token patterns of breakfast.py:131-155 are restated as
// hand-written ASCII matchers.  The reader accepts only inputs whose meaning does not depend on pandas'
// CSV dialect handling and answers BFK_EUNSUPPORTED otherwise (the caller then uses the pandas reader).
#include "../../include/bfk.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

int bfk_fail(int code, const std::string &msg);  // bfk_base.cpp: sets the thread-local message
int bfk_front_cluster(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int32_t max_dist, int32_t n_gpus,
                      int32_t *labels_out);  // bfk_base.cpp: bfk_cluster_csr of the preloaded libbfk.so

namespace {

struct Span {
    int64_t off;
    int32_t len;
};

// vectors of trivially constructible elements whose resize() does not zero what is about to be overwritten (the file image:
// 330 MB at 1M rows; the CSR: 170 MB)
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = NoInitAlloc<U>;
    };
    NoInitAlloc() = default;
    template <class U>
    NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U, class... A>
    void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;  // default-init: nothing for char / int
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
using RawBytes = std::vector<char, NoInitAlloc<char>>;

inline uint64_t bytes_hash(const char *p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xFF51AFD7ED558CCDull);
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        h = (h ^ v) * 0xC4CEB9FE1A85EC53ull;
        h ^= h >> 29;
        p += 8;
        n -= 8;
    }
    uint64_t v = 0;
    memcpy(&v, p, n);
    h = (h ^ v) * 0xC4CEB9FE1A85EC53ull;
    return h ^ (h >> 32);
}

// ---- token classification (breakfast.py:131-190) -------------------------------------------------
enum Verdict : int8_t { KEEP = 0, DROP = 1, INVALID = 2 };

inline bool up(char c) { return c >= 'A' && c <= 'Z'; }
inline bool dg(char c) { return c >= '0' && c <= '9'; }
inline bool alnum(char c) { return up(c) || dg(c) || (c >= 'a' && c <= 'z'); }
inline bool all_digits(const char *p, int64_t n) {
    if (n <= 0) return false;
    for (int64_t i = 0; i < n; i++)
        if (!dg(p[i])) return false;
    return true;
}
// \d+:\d+ over the whole range
inline bool num_colon_num(const char *p, int64_t n) {
    const char *c = (const char *)memchr(p, ':', (size_t)std::max<int64_t>(n, 0));
    if (!c) return false;
    return all_digits(p, c - p) && all_digits(c + 1, n - (c - p) - 1);
}
// [a-zA-Z0-9]+: prefix; returns the length of the prefix incl. ':' or 0
inline int64_t gene_prefix(const char *p, int64_t n) {
    int64_t i = 0;
    while (i < n && alnum(p[i])) i++;
    return (i > 0 && i < n && p[i] == ':') ? i + 1 : 0;
}
// [A-Z]\d+ : returns the index after the digits or 0
inline int64_t letter_digits(const char *p, int64_t n) {
    if (n < 2 || !up(p[0]) || !dg(p[1])) return 0;
    int64_t i = 1;
    while (i < n && dg(p[i])) i++;
    return i;
}

struct Classifier {
    bfk_filter_opts o;
    int64_t upper;

    // position of a DNA substitution against the trims (:173-176): pos <= trim_start or pos >= upper -> dropped
    bool trimmed(const char *d, int64_t n) const {
        while (n > 1 && *d == '0') {
            d++;
            n--;
        }
        if (n > 18) return true;  // beyond any int64 bound: pos >= upper
        int64_t pos = 0;
        for (int64_t i = 0; i < n; i++) pos = pos * 10 + (d[i] - '0');
        return pos <= o.trim_start || pos >= upper;
    }
    bool dna_sub(const char *p, int64_t n, Verdict *v) const {  // ^[A-Z](\d+)[A-Z]$
        if (n < 3 || !up(p[0]) || !up(p[n - 1]) || !all_digits(p + 1, n - 2)) return false;
        *v = trimmed(p + 1, n - 2) ? DROP : KEEP;
        return true;
    }
    Verdict indel(bool is_ins) const { return (is_ins ? o.skip_ins : o.skip_del) ? DROP : KEEP; }

    Verdict match(const char *p, int64_t n) const {
        Verdict v;
        switch (o.var_type) {
        case BFK_VAR_COVSONAR_DNA:
            if (dna_sub(p, n, &v)) return v;
            if (n >= 2 && up(p[n - 1]) && up(p[n - 2])) return indel(true);                  // ^.*[A-Z][A-Z]$
            if (n > 4 && memcmp(p, "del:", 4) == 0 && num_colon_num(p + 4, n - 4)) return indel(false);  // ^del:\d+:\d+$
            return INVALID;
        case BFK_VAR_NEXTCLADE_DNA: {
            if (dna_sub(p, n, &v)) return v;
            const char *c = (const char *)memchr(p, ':', (size_t)n);
            if (c && all_digits(p, c - p)) {                                                  // ^\d+:[A-Z]+$
                const int64_t r = n - (c - p) - 1;
                bool ok = r > 0;
                for (int64_t i = 0; ok && i < r; i++) ok = up(c[1 + i]);
                if (ok) return indel(true);
            }
            const char *m = (const char *)memchr(p, '-', (size_t)n);                         // ^\d+(-\d+)?$
            if (m ? (all_digits(p, m - p) && all_digits(m + 1, n - (m - p) - 1)) : all_digits(p, n)) return indel(false);
            return INVALID;
        }
        case BFK_VAR_COVSONAR_AA: {
            const int64_t g = gene_prefix(p, n);
            if (!g) return INVALID;
            const char *r = p + g;
            const int64_t rn = n - g;
            const int64_t e = letter_digits(r, rn);
            if (e && e < rn) {
                bool letters = true;
                for (int64_t i = e; i < rn; i++) letters = letters && up(r[i]);
                if (letters) return rn - e == 1 ? KEEP : indel(true);  // [A-Z]\d+[A-Z] | [A-Z]\d+[A-Z][A-Z]+
            }
            if (rn > 4 && memcmp(r, "del:", 4) == 0 && num_colon_num(r + 4, rn - 4)) return indel(false);
            return INVALID;
        }
        case BFK_VAR_NEXTCLADE_AA: {
            if (n == 0) return indel(true);  // the insertion pattern of this type is ^$
            const int64_t g = gene_prefix(p, n);
            if (!g) return INVALID;
            const char *r = p + g;
            const int64_t rn = n - g;
            const int64_t e = letter_digits(r, rn);
            if (e && e + 1 == rn) {
                if (up(r[e]) || r[e] == '*') return KEEP;  // [A-Z]\d+[A-Z*]
                if (r[e] == '-') return indel(false);      // [A-Z]\d+-
            }
            return INVALID;
        }
        default:
            return KEEP;  // raw: no patterns
        }
    }
    // verdict of one token; an empty token that survives is dropped silently (:186-187)
    Verdict operator()(const char *p, int64_t n) const {
        const Verdict v = match(p, n);
        return (v == KEEP && n == 0) ? DROP : v;
    }
};

// 32 bytes: a probe touches one line, and a token of up to 8 bytes — nearly all of them — is compared by its inline copy
// (following `p` to the token's first occurrence, somewhere in 330 MB of text, was a second cache miss per token)
struct TokSlot {
    uint64_t h;
    uint64_t k8;    // the first min(len, 8) bytes, zero-padded
    const char *p;  // first occurrence (NULL: free slot)
    uint32_t len;
    int32_t id;     // >= 0: vocabulary id; TOK_FRESH: kept, not numbered yet; TOK_DROP / TOK_INVALID: the classifier's verdict
};
constexpr int32_t TOK_FRESH = -1, TOK_DROP = -2, TOK_INVALID = -3;

// ---- host threads ----------------------------------------------------------------------------------
// The text stages are row-parallel: contiguous row chunks, one std::thread each (BFK_THREADS overrides the count;
// default = the CPUs this process may use, at most 16).  Results that depend on the input ORDER (first-appearance
// vocabulary ids, the order of the "invalid feature" lines, first-appearance unique rows) are merged chunk by chunk
// in row order, so the output is identical for every thread count (tests/test_frontend.py runs 1, 3 and 8).
int host_threads() {
    if (const char *e = getenv("BFK_THREADS")) return std::max(1, std::min(64, atoi(e)));
    unsigned n = std::thread::hardware_concurrency();
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n ? n : 64, (unsigned)CPU_COUNT(&set));
#endif
    return (int)std::max(1u, std::min(16u, n));
}

void parallel_chunks(int n_chunks, const std::function<void(int)> &fn) {
    if (n_chunks <= 1) {
        if (n_chunks == 1) fn(0);
        return;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)n_chunks - 1);
    for (int c = 1; c < n_chunks; c++) th.emplace_back(fn, c);
    fn(0);
    for (auto &t : th) t.join();
}

// ---- phase A of the tokeniser: one contiguous row chunk ---------------------------------------------
// Rows are byte ranges of `base`; tokens are split on `sep` (non-overlapping, left to right, like str.split), empty
// tokens never enter a row (:208-209).  With a classifier every distinct token is judged once (memoised in the
// chunk's table); kept tokens get CHUNK-LOCAL ids in order of first kept appearance.
struct TokChunk {
    int64_t r0 = 0, r1 = 0;
    std::vector<int32_t> ids;       // local (later global) ids of the kept tokens of all rows of the chunk
    std::vector<int64_t> row_end;   // per row: end offset into ids
    std::vector<Span> vocab;        // local id -> token bytes
    std::vector<uint64_t> vhash;    // ... and their hashes (reused by the merge)
    std::vector<uint64_t> vk8;      // ... and their first 8 bytes, zero-padded (the merge compares short tokens without touching the text)
    std::vector<Span> invalid;      // token occurrences that matched no pattern, in order
    std::vector<int32_t> to_global;
};

void tokenize_chunk(const char *base, const std::function<Span(int64_t)> &row_span, const char *sep, int64_t sep_len,
                    const Classifier *cls, TokChunk &ck) {
    const char s0 = sep[0];
    size_t tcap = 1u << 12, tcount = 0;
    std::vector<TokSlot> tab(tcap, TokSlot{0, 0, nullptr, 0, TOK_FRESH});
    ck.row_end.reserve((size_t)(ck.r1 - ck.r0));
    if (ck.r1 > ck.r0) {  // (one token per ~7 bytes of profile text: no regrowth of the id array on the way)
        const Span a = row_span(ck.r0), z = row_span(ck.r1 - 1);
        const int64_t span_bytes = z.off + z.len - a.off;
        if (span_bytes > 0) ck.ids.reserve((size_t)(span_bytes / 5 + 64));
    }
    for (int64_t r = ck.r0; r < ck.r1; r++) {
        const Span sp = row_span(r);
        const char *s = base + sp.off;
        const int64_t len = sp.len;
        int64_t pos = 0;
        while (pos <= len) {
            int64_t nx = -1;  // next separator at or after pos
            if (sep_len == 1) {
                // (tokens are a handful of bytes: a plain loop beats the call into memchr)
                int64_t i = pos;
                while (i < len && s[i] != s0) i++;
                if (i < len) nx = i;
            } else {
                for (int64_t i = pos; i + sep_len <= len; i++)
                    if (s[i] == s0 && memcmp(s + i, sep, (size_t)sep_len) == 0) {
                        nx = i;
                        break;
                    }
            }
            const int64_t tl = (nx < 0 ? len : nx) - pos;
            const char *tk = s + pos;
            if (tl == 0) {
                // empty token: never in the CSR; with filtering its verdict still counts (it is "invalid" for most
                // types and printed as such)
                if (cls && (*cls)(tk, 0) == INVALID) ck.invalid.push_back(Span{tk - base, 0});
            } else {
                const uint64_t h = bytes_hash(tk, (size_t)tl);
                uint64_t k8 = 0;
                memcpy(&k8, tk, (size_t)std::min<int64_t>(tl, 8));
                size_t i = h & (tcap - 1);
                while (tab[i].p && !(tab[i].h == h && tab[i].len == (uint32_t)tl && tab[i].k8 == k8 &&
                                     (tl <= 8 || memcmp(tab[i].p + 8, tk + 8, (size_t)(tl - 8)) == 0)))
                    i = (i + 1) & (tcap - 1);
                if (!tab[i].p) {
                    const int v = cls ? (*cls)(tk, tl) : KEEP;
                    tab[i] = TokSlot{h, k8, tk, (uint32_t)tl, v == KEEP ? TOK_FRESH : (v == DROP ? TOK_DROP : TOK_INVALID)};
                    tcount++;
                }
                TokSlot &sl = tab[i];
                if (sl.id >= TOK_FRESH) {
                    if (sl.id < 0) {
                        sl.id = (int32_t)ck.vocab.size();
                        ck.vocab.push_back(Span{tk - base, (int32_t)tl});
                        ck.vhash.push_back(h);
                        ck.vk8.push_back(k8);
                    }
                    ck.ids.push_back(sl.id);
                } else if (sl.id == TOK_INVALID) {
                    ck.invalid.push_back(Span{tk - base, (int32_t)tl});
                }
                if (tcount * 2 > tcap) {
                    std::vector<TokSlot> nt(tcap * 2, TokSlot{0, 0, nullptr, 0, TOK_FRESH});
                    for (size_t o = 0; o < tcap; o++)
                        if (tab[o].p) {
                            size_t j = tab[o].h & (tcap * 2 - 1);
                            while (nt[j].p) j = (j + 1) & (tcap * 2 - 1);
                            nt[j] = tab[o];
                        }
                    tab.swap(nt);
                    tcap *= 2;
                }
            }
            if (nx < 0) break;
            pos = nx + sep_len;
        }
        ck.row_end.push_back((int64_t)ck.ids.size());
    }
}

// ---- phases B + C: the chunks' vocabularies merged in row order (first-appearance ids of the whole input), then every
// chunk's ids rewritten in place.  Returns the global vocabulary (token bytes per id).
void merge_vocab(const char *base, std::vector<TokChunk> &cks, std::vector<Span> &vocab_out) {
    size_t total = 0;
    for (const TokChunk &c : cks) total += c.vocab.size();
    size_t cap = 1u << 10;
    while (cap < total * 2 + 16) cap <<= 1;
    struct MSlot {  // one line per probe; tokens of up to 8 bytes are compared by their inline copy
        uint64_t h, k8;
        int32_t id, len;
    };
    std::vector<MSlot> slot(cap, MSlot{0, 0, -1, 0});
    vocab_out.clear();
    for (TokChunk &c : cks) {
        c.to_global.resize(c.vocab.size());
        for (size_t l = 0; l < c.vocab.size(); l++) {
            const Span v = c.vocab[l];
            const uint64_t h = c.vhash[l], k8 = c.vk8[l];
            size_t i = h & (cap - 1);
            while (slot[i].id >= 0) {
                const MSlot &m = slot[i];
                if (m.h == h && m.len == v.len && m.k8 == k8 &&
                    (v.len <= 8 || memcmp(base + vocab_out[(size_t)m.id].off + 8, base + v.off + 8, (size_t)(v.len - 8)) == 0))
                    break;
                i = (i + 1) & (cap - 1);
            }
            if (slot[i].id < 0) {
                slot[i] = MSlot{h, k8, (int32_t)vocab_out.size(), v.len};
                vocab_out.push_back(v);
            }
            c.to_global[l] = slot[i].id;
        }
    }
    parallel_chunks((int)cks.size(), [&](int q) {
        TokChunk &c = cks[(size_t)q];
        for (int32_t &x : c.ids) x = c.to_global[(size_t)x];
    });
}

// bytes of text a thread is worth starting for (BFK_CHUNK_BYTES: test knob — tiny inputs in many chunks)
int64_t chunk_bytes(int64_t dflt) {
    if (const char *e = getenv("BFK_CHUNK_BYTES")) return std::max<int64_t>(1, atoll(e));
    return dflt;
}

std::vector<TokChunk> make_chunks(int64_t n_rows, int64_t total_bytes) {
    int t = host_threads();
    t = (int)std::max<int64_t>(1, std::min<int64_t>(t, std::min<int64_t>(n_rows, total_bytes / chunk_bytes(256 << 10) + 1)));
    std::vector<TokChunk> cks((size_t)t);
    for (int q = 0; q < t; q++) {
        cks[(size_t)q].r0 = n_rows * q / t;
        cks[(size_t)q].r1 = n_rows * (q + 1) / t;
    }
    return cks;
}

}  // namespace

// a1: tokeniser + first-appearance vocabulary + CSR (replaces sparse_feature_matrix, breakfast.py:193-215); row-parallel
extern "C" int bfk_build_csr(const char *buf, const int64_t *row_off, int64_t n_rows, const char *sep, int64_t sep_len,
                             int32_t *indptr_out, int32_t **indices_out, int64_t *nnz_out, int32_t *n_vocab_out) {
    if (!row_off || !indptr_out || !indices_out || !nnz_out || !n_vocab_out || n_rows < 0 || (!buf && n_rows > 0 && row_off[n_rows] > 0))
        return bfk_fail(BFK_EARG, "bfk_build_csr: null argument");
    if (!sep || sep_len <= 0) return bfk_fail(BFK_EARG, "empty separator");
    for (int64_t r = 0; r < n_rows; r++)
        if (row_off[r + 1] < row_off[r] || row_off[r + 1] - row_off[r] > INT32_MAX) return bfk_fail(BFK_EARG, "bfk_build_csr: row_off not monotone");
    std::vector<TokChunk> cks = make_chunks(n_rows, n_rows > 0 ? row_off[n_rows] - row_off[0] : 0);
    const auto span = [&](int64_t r) { return Span{row_off[r], (int32_t)(row_off[r + 1] - row_off[r])}; };
    parallel_chunks((int)cks.size(), [&](int q) { tokenize_chunk(buf, span, sep, sep_len, nullptr, cks[(size_t)q]); });
    std::vector<Span> vocab;
    merge_vocab(buf, cks, vocab);
    int64_t nnz = 0;
    std::vector<int64_t> base(cks.size());
    for (size_t q = 0; q < cks.size(); q++) {
        base[q] = nnz;
        nnz += (int64_t)cks[q].ids.size();
    }
    if (nnz > (int64_t)INT32_MAX) return bfk_fail(BFK_EARG, "bfk_build_csr: more than 2^31-1 entries");
    int32_t *out = (int32_t *)malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(1, nnz));
    if (!out) return bfk_fail(BFK_ENOMEM, "bfk_build_csr: out of memory");
    indptr_out[0] = 0;
    parallel_chunks((int)cks.size(), [&](int q) {
        const TokChunk &c = cks[(size_t)q];
        if (!c.ids.empty()) memcpy(out + base[(size_t)q], c.ids.data(), c.ids.size() * sizeof(int32_t));
        for (int64_t r = c.r0; r < c.r1; r++) indptr_out[r + 1] = (int32_t)(base[(size_t)q] + c.row_end[(size_t)(r - c.r0)]);
    });
    *indices_out = out;
    *nnz_out = nnz;
    *n_vocab_out = (int32_t)vocab.size();
    return BFK_OK;
}

struct bfk_table {
    RawBytes bytes;  // owned copy of the file / buffers
    std::vector<Span> ids, feats;
    // prepare() results
    std::vector<int32_t> group, weight, first_row, indptr;
    std::vector<int32_t, NoInitAlloc<int32_t>> indices;  // (filled by a parallel copy: not zeroed first)
    std::vector<Span> invalid;  // into `bytes`
    std::string sep2;
    bool filtered = false, prepared = false;
    bool device_prepared = false;  // prepare() ran on the device (bfk_table_set_prepared): group / weight / first_row (/ CSR) are here, the
                                   // vocabulary's bytes are not — the feature-string accessors need the host prepare
    bool any_high = false;  // the bytes hold non-ASCII (valid UTF-8) somewhere: prepare() looks at the feature column when it filters
    int32_t n_vocab = 0;
    std::vector<Span> vocab;  // token bytes of every vocabulary id (for bfk_table_feature)
};

namespace {

const char *const NA_STRINGS[] = {"",        "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN",
                                  "<NA>",    "N/A",  "NA",       "NULL", "NaN",    "None",     "n/a",  "nan",  "null"};
inline bool is_na(const char *p, int64_t n) {
    if (n > 8) return false;
    for (const char *s : NA_STRINGS)
        if ((int64_t)strlen(s) == n && memcmp(s, p, (size_t)n) == 0) return true;
    return false;
}

// Bytes >= 0x80 are opaque to an unquoted, ASCII-separated table as long as they are VALID UTF-8: pandas decodes the file as
// UTF-8 (read_table's default) and raises UnicodeDecodeError otherwise, compares ids and features as code-point strings —
// which for valid UTF-8 is byte equality — and to_csv encodes them back unchanged.  The validator is Python's: no overlong
// forms, no surrogates (U+D800..DFFF), nothing beyond U+10FFFF.  Checks the sequences that START in [i, e); a sequence may
// end behind e (the buffer has a sentinel byte); continuation bytes at i that the last sequence of the slice in front covers
// are that slice's to check.
bool utf8_valid_from(const char *b, int64_t i, int64_t e, int64_t n_total, bool first_slice) {
    const unsigned char *u = (const unsigned char *)b;
    if (!first_slice && i < e && (u[i] & 0xC0) == 0x80) {
        // a slice that starts inside a sequence: exactly the continuation bytes that the sequence's lead byte (within the three
        // bytes in front of i) covers belong to the slice in front, which validates them; a continuation byte that no lead
        // covers is invalid whichever slice it falls into (the result must not depend on where the slices are cut)
        int64_t j = i - 1;
        while (j >= 0 && i - j <= 3 && (u[j] & 0xC0) == 0x80) j--;
        if (j < 0 || i - j > 3) return false;
        const unsigned char c = u[j];
        const int need = c >= 0xC2 && c <= 0xDF ? 1 : c >= 0xE0 && c <= 0xEF ? 2 : c >= 0xF0 && c <= 0xF4 ? 3 : 0;
        if (j + 1 + need <= i) return false;  // (no lead there, or its sequence ended in front of i)
        i = std::min<int64_t>(j + 1 + need, e);
    }
    while (i < e) {
        const unsigned char c = u[i];
        if (c < 0x80) {
            i++;
            continue;
        }
        int need;
        unsigned lo = 0x80, hi = 0xBF;
        if (c >= 0xC2 && c <= 0xDF) need = 1;
        else if (c >= 0xE0 && c <= 0xEF) {
            need = 2;
            if (c == 0xE0) lo = 0xA0;  // no overlong three-byte forms
            if (c == 0xED) hi = 0x9F;  // no surrogates
        } else if (c >= 0xF0 && c <= 0xF4) {
            need = 3;
            if (c == 0xF0) lo = 0x90;  // no overlong four-byte forms
            if (c == 0xF4) hi = 0x8F;  // nothing beyond U+10FFFF
        } else
            return false;  // a continuation byte where a sequence must start, 0xC0 / 0xC1, 0xF5..0xFF
        if (i + need >= n_total + 1) return false;
        if (u[i + 1] < lo || u[i + 1] > hi) return false;
        for (int k = 2; k <= need; k++)
            if ((u[i + k] & 0xC0) != 0x80) return false;
        i += need + 1;
    }
    return true;
}

bool has_high_byte(const char *p, int64_t n) {
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t v;
        memcpy(&v, p + i, 8);
        if (v & 0x8080808080808080ull) return true;
    }
    for (; i < n; i++)
        if ((unsigned char)p[i] >= 0x80) return true;
    return false;
}

// BFK_FRONT_TIMING=1: stage times of the host stages on stderr
struct StageTimer {
    const bool on = getenv("BFK_FRONT_TIMING") && atoi(getenv("BFK_FRONT_TIMING")) != 0;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[bfk_front] %-28s %8.2f ms   (at %.1f ms)\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t1.time_since_epoch()).count() - 1e3 * (double)(long long)(std::chrono::duration<double>(t1.time_since_epoch()).count() / 100) * 100);
        t0 = t1;
    }
};

int unsupported(const std::string &why) { return bfk_fail(BFK_EUNSUPPORTED, "bfk_table: input needs the general reader: " + why); }

// pandas' default CSV dialect (read_table: quotechar '"', doublequote, QUOTE_MINIMAL, no escapechar — breakfast.py:16-21 passes none
// of these) as its C tokeniser applies it: a quote OPENS a quoted field only as the field's first byte; inside, "" is one quote
// and any other byte — separators and line breaks too — is content; after the closing quote the field runs on, verbatim, up to
// the next separator or line end; a quote anywhere else is content.  One field of a record:
struct Fld {
    int64_t s0 = 0;        // first raw byte
    int64_t cs = 0, ce = 0;  // content, when it lies in the image as it is (!rewrite)
    int64_t e = 0;         // raw end: the separator or the line end behind it
    bool rewrite = false;  // "" inside, or bytes behind the closing quote: the content has to be put together (unquote_field)
};
// A record that starts at p, separator sp; the image has a '\n' at b[n] (sentinel) and every '\r' in it is followed by '\n'.
// -> the start of the next record (behind the record's LF — which need not be the first LF after p), or -1: the file ends inside
// a quoted field (pandas: "EOF inside string").  Fields id_i / ft_i are handed out, all of them when `all` is given.
int64_t quoted_record(const char *b, int64_t p, int64_t n, char sp, int id_i, int ft_i, int *ncol, Fld *id, Fld *ft, std::vector<Fld> *all) {
    int col = 0;
    int64_t i = p;
    for (;;) {
        Fld f;
        f.s0 = i;
        if (b[i] == '"') {
            int64_t j = i + 1;
            for (;;) {
                const char *q = (const char *)memchr(b + j, '"', (size_t)(n - j));
                if (!q) return -1;
                j = q - b;
                if (b[j + 1] != '"') break;
                f.rewrite = true;
                j += 2;
            }
            f.cs = i + 1;
            f.ce = j;
            i = j + 1;
            if (b[i] != sp && b[i] != '\n' && b[i] != '\r') {
                f.rewrite = true;
                while (b[i] != sp && b[i] != '\n' && b[i] != '\r') i++;
            }
        } else {
            while (b[i] != sp && b[i] != '\n' && b[i] != '\r') i++;
            f.cs = f.s0;
            f.ce = i;
        }
        f.e = i;
        if (col == id_i && id) *id = f;
        if (col == ft_i && ft) *ft = f;
        if (all) all->push_back(f);
        col++;
        if (b[i] == sp) {
            i++;
            continue;
        }
        *ncol = col;
        return (b[i] == '\r' ? i + 1 : i) + 1;
    }
}
// the content of a field with f.rewrite, written over its own raw bytes from f.s0 on (it is never longer) -> its length
int64_t unquote_field(char *b, const Fld &f) {
    int64_t w = f.s0, i = f.s0 + 1;
    for (;;) {
        const char c = b[i];
        if (c == '"') {
            if (b[i + 1] != '"') break;
            i++;
        }
        b[w++] = c;
        i++;
    }
    for (i++; i < f.e; i++) b[w++] = b[i];
    return w - f.s0;
}

// every id distinct?  (read_input raises on duplicates, :24-27: left to the pandas path)
// Row-parallel: an insert-only open-addressing table of row indices, slots taken by compare-and-swap; a row that meets
// an equal id on its probe path — whoever put it there — has found a duplicate (two rows with one id start probing at the
// same slot: the later to arrive walks over the earlier).  (Serial: 0.24 s of a 1M-row run.)
bool ids_distinct(const bfk_table &t) {
    size_t cap = 16;
    while (cap < t.ids.size() * 2) cap <<= 1;
    std::unique_ptr<std::atomic<int32_t>[]> tab(new std::atomic<int32_t>[cap]);
    const char *b = t.bytes.data();
    const int64_t n = (int64_t)t.ids.size();
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / 16384 + 1));
    parallel_chunks(parts, [&](int q) {
        for (size_t i = cap * (size_t)q / (size_t)parts, e = cap * ((size_t)q + 1) / (size_t)parts; i < e; i++) tab[i].store(-1, std::memory_order_relaxed);
    });
    std::atomic<int> dup{0};
    parallel_chunks(parts, [&](int q) {
        for (int64_t r = n * q / parts, e = n * (q + 1) / parts; r < e; r++) {
            if ((r & 1023) == 0 && dup.load(std::memory_order_relaxed)) return;
            const Span s = t.ids[(size_t)r];
            size_t i = bytes_hash(b + s.off, (size_t)s.len) & (cap - 1);
            for (;;) {
                int32_t cur = tab[i].load(std::memory_order_acquire);
                if (cur < 0 && tab[i].compare_exchange_strong(cur, (int32_t)r, std::memory_order_acq_rel)) break;
                // (cur now holds the row that sits here)
                const Span o = t.ids[(size_t)cur];
                if (o.len == s.len && memcmp(b + o.off, b + s.off, (size_t)s.len) == 0) {
                    dup.store(1);
                    return;
                }
                i = (i + 1) & (cap - 1);
            }
        }
    });
    return dup.load() == 0;
}

}  // namespace

extern "C" int bfk_table_open(const char *path, const char *sep, int64_t sep_len, const char *id_col, const char *feature_col,
                              bfk_table **out) {
    if (!path || !sep || !id_col || !feature_col || !out) return bfk_fail(BFK_EARG, "bfk_table_open: null argument");
    if (sep_len != 1) return unsupported("multi-byte column separator");
    const char sp = sep[0];
    if (sp == '\n' || sp == '\r' || sp == '"' || sp == 0 || (unsigned char)sp >= 0x80) return unsupported("separator");
    if (strcmp(id_col, feature_col) == 0) return unsupported("id and feature column are the same");
    StageTimer tm;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return bfk_fail(BFK_EIO, std::string("cannot open ") + path);
    struct stat st_;
    if (fstat(fd, &st_) != 0 || !S_ISREG(st_.st_mode)) {  // (pipes and the like: the general reader)
        close(fd);
        return unsupported("not a regular file");
    }
    bfk_table *t = new bfk_table();
    const int64_t sz = (int64_t)st_.st_size;
    t->bytes.resize((size_t)std::max<int64_t>(sz, 0) + 1);  // (not zeroed: every byte is read into)
    {   // the file image, read in parallel slices (one serial fread of 330 MB was 0.26 s of a 1M-row run)
        const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), sz / chunk_bytes(8 << 20) + 1));
        std::atomic<int> bad{0};
        char *dst = t->bytes.data();
        parallel_chunks(parts, [&](int q) {
            int64_t at = sz * q / parts;
            const int64_t end = sz * (q + 1) / parts;
            while (at < end) {
                const ssize_t got = pread(fd, dst + at, (size_t)(end - at), (off_t)at);
                if (got <= 0) {
                    bad.store(1);
                    return;
                }
                at += got;
            }
        });
        close(fd);
        if (bad.load()) {
            delete t;
            return bfk_fail(BFK_EIO, std::string("short read on ") + path);
        }
    }
    t->bytes[(size_t)sz] = '\n';  // sentinel: the last line always ends
    tm.lap("open: read");
    const char *b = t->bytes.data();
    const int64_t n = sz;
    bool any_high = false, any_quote = false;
    {   // byte checks, in parallel slices
        const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / chunk_bytes(1 << 20) + 1));
        std::atomic<int> bad{0}, high{0}, quotes{0};
        // (a byte-order mark in front of everything is dropped below: not "bytes >= 0x80" of the table)
        const int64_t bom = n >= 3 && (unsigned char)b[0] == 0xEF && (unsigned char)b[1] == 0xBB && (unsigned char)b[2] == 0xBF ? 3 : 0;
        parallel_chunks(parts, [&](int q) {
            int64_t i = std::max(n * q / parts, bom);
            const int64_t e = std::max(n * (q + 1) / parts, i);
            bool seen_high = false, seen_quote = false;
            auto bad_byte = [&](int64_t j) {
                const unsigned char c = (unsigned char)b[j];
                seen_high = seen_high || c >= 0x80;
                seen_quote = seen_quote || c == '"';
                // CR is accepted only in front of an LF (the reference's own fixtures are CRLF files)
                return c == 0 || (c == '\r' && b[j + 1] != '\n');
            };
            // eight bytes at a time: a word without a high bit, a zero byte, a quote or a CR needs no second look
            constexpr uint64_t L = 0x0101010101010101ull, H = 0x8080808080808080ull;
            auto has_zero = [](uint64_t v) { return ((v - L) & ~v & H) != 0; };
            for (; i + 8 <= e; i += 8) {
                uint64_t v;
                memcpy(&v, b + i, 8);
                if (((v & H) != 0) | has_zero(v) | has_zero(v ^ (L * (uint64_t)'"')) | has_zero(v ^ (L * (uint64_t)'\r'))) {
                    for (int64_t j = i; j < i + 8; j++)
                        if (bad_byte(j)) {
                            bad.store(1);
                            return;
                        }
                }
            }
            for (; i < e; i++)
                if (bad_byte(i)) {
                    bad.store(1);
                    return;
                }
            if (seen_high) high.store(1);
            if (seen_quote) quotes.store(1);
        });
        if (bad.load()) {
            delete t;
            return unsupported("lone CR or NUL byte");
        }
        any_high = high.load() != 0;
        any_quote = quotes.load() != 0;
        if (any_high) {
            // bytes >= 0x80 (accession names with accents, say) are opaque to a table with an ASCII separator — if
            // the file is valid UTF-8 (pandas raises UnicodeDecodeError otherwise)
            std::atomic<int> inval{0};
            parallel_chunks(parts, [&](int q) {
                if (!utf8_valid_from(b, n * q / parts, n * (q + 1) / parts, n, q == 0)) inval.store(1);
            });
            if (inval.load()) {
                delete t;
                return unsupported("bytes that are not valid UTF-8");
            }
        }
    }
    t->any_high = any_high;
    tm.lap("open: byte check");
    // end of the line starting at p: index of its LF, and the end of its content (CR of a CRLF stripped)
    auto line_end = [&](int64_t p, int64_t *content_end) {
        const int64_t lf = (const char *)memchr(b + p, '\n', (size_t)(n + 1 - p)) - b;
        *content_end = (lf > p && b[lf - 1] == '\r') ? lf - 1 : lf;
        return lf;
    };
    // a line that pandas skips (skip_blank_lines): empty, or nothing but blanks and tabs that are not the separator
    auto skipped_line = [&](int64_t p, int64_t le) {
        for (int64_t i = p; i < le; i++)
            if ((b[i] != ' ' && b[i] != '\t') || b[i] == sp) return false;
        return true;
    };
    // header = first line that is not skipped; a UTF-8 byte-order mark in front of everything is not data (pandas drops it)
    int64_t pos = 0, he = 0;
    if (n >= 3 && (unsigned char)b[0] == 0xEF && (unsigned char)b[1] == 0xBB && (unsigned char)b[2] == 0xBF) pos = 3;
    while (pos < n) {
        const int64_t lf = line_end(pos, &he);
        if (!skipped_line(pos, he)) break;
        pos = lf + 1;
    }
    if (pos >= n) {
        delete t;
        return unsupported("empty file");
    }
    const int64_t header_lf = line_end(pos, &he);
    int64_t data_start = header_lf + 1;
    int ncols = 0, id_i = -1, ft_i = -1;
    std::vector<std::string> names;
    if (any_quote && memchr(b + pos, '"', (size_t)(he - pos))) {  // quoted column names (the record may run over several lines)
        std::vector<Fld> hf;
        data_start = quoted_record(b, pos, n, sp, -1, -1, &ncols, nullptr, nullptr, &hf);
        if (data_start < 0) {
            delete t;
            return unsupported("the file ends inside a quoted field");
        }
        for (const Fld &f : hf) {
            if (!f.rewrite) {
                names.emplace_back(b + f.cs, (size_t)(f.ce - f.cs));
                continue;
            }
            std::string raw(b + f.s0, (size_t)(f.e - f.s0));
            raw += '\n';
            Fld g = f;
            g.s0 = 0;
            g.e = f.e - f.s0;
            raw.resize((size_t)unquote_field(&raw[0], g));
            names.push_back(raw);
        }
    } else {
        for (int64_t s = pos;;) {
            const char *c = (const char *)memchr(b + s, sp, (size_t)(he - s));
            const int64_t e = c ? c - b : he;
            names.emplace_back(b + s, (size_t)(e - s));
            ncols++;
            if (!c) break;
            s = e + 1;
        }
    }
    for (int i = 0; i < ncols; i++) {
        if (names[(size_t)i].empty()) {
            delete t;
            return unsupported("unnamed column");
        }
        for (int j = 0; j < i; j++)
            if (names[(size_t)i] == names[(size_t)j]) {
                delete t;
                return unsupported("duplicate column names");
            }
        if (names[(size_t)i] == id_col) id_i = i;
        if (names[(size_t)i] == feature_col) ft_i = i;
    }
    if (id_i < 0 || ft_i < 0) {
        delete t;
        return unsupported("column not found");
    }
    pos = std::min(data_start, n);
    // Data lines, in parallel slices that start at line starts; the slices' rows are concatenated in order.  With quotes in the
    // file a slice boundary (the first LF behind an even split) may lie INSIDE a quoted field: then the slice in front — whose
    // records are followed through their quoted line breaks — ends behind its boundary instead of on it.  Every slice ending
    // on its boundary proves all of them true record starts (the first slice starts at one; induction); otherwise the lines
    // are read again as one slice.  Nothing is written to the image before that is settled (fields that need unquoting are
    // listed and rewritten afterwards).
    struct Rewrite {
        uint32_t row;  // within the slice
        bool is_ft;
        Fld f;
    };
    for (int attempt = 0; attempt < 2; attempt++) {
        const int parts = attempt ? 1 : (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), (n - pos) / chunk_bytes(1 << 20) + 1));
        std::vector<int64_t> cut((size_t)parts + 1, n);
        cut[0] = pos;
        for (int q = 1; q < parts; q++) {
            const int64_t at = std::max(cut[(size_t)q - 1], pos + (n - pos) * q / parts);
            int64_t ce;
            cut[(size_t)q] = at >= n ? n : std::min<int64_t>(n, line_end(at, &ce) + 1);  // the line straddling the cut stays with the slice before
        }
        std::vector<std::vector<Span>> pid((size_t)parts), pft((size_t)parts);
        std::vector<std::vector<Rewrite>> prw((size_t)parts);
        std::vector<const char *> why((size_t)parts, nullptr);
        std::atomic<int> off_boundary{0};
        parallel_chunks(parts, [&](int q) {
            int64_t p = cut[(size_t)q];
            const int64_t pe = cut[(size_t)q + 1];
            auto &vi = pid[(size_t)q];
            auto &vf = pft[(size_t)q];
            vi.reserve((size_t)((pe - p) / 256 + 16));
            vf.reserve((size_t)((pe - p) / 256 + 16));
            while (p < pe) {
                int64_t le;
                const int64_t lf = line_end(p, &le);
                if (skipped_line(p, le)) {  // blank line
                    p = lf + 1;
                    continue;
                }
                if (any_quote && memchr(b + p, '"', (size_t)(le - p))) {
                    Fld fi, ff;
                    int col = 0;
                    const int64_t next = quoted_record(b, p, n, sp, id_i, ft_i, &col, &fi, &ff, nullptr);
                    if (next < 0) {
                        why[(size_t)q] = "the file ends inside a quoted field";
                        return;
                    }
                    if (col != ncols) {
                        why[(size_t)q] = "ragged row";
                        return;
                    }
                    if (fi.e - fi.s0 > INT32_MAX || ff.e - ff.s0 > INT32_MAX) {
                        why[(size_t)q] = "field longer than 2 GiB";
                        return;
                    }
                    Span id{fi.cs, (int32_t)(fi.ce - fi.cs)}, ft{ff.cs, (int32_t)(ff.ce - ff.cs)};
                    if (fi.rewrite) prw[(size_t)q].push_back(Rewrite{(uint32_t)vi.size(), false, fi});
                    else if (is_na(b + id.off, id.len)) {
                        why[(size_t)q] = "NA-valued id";
                        return;
                    }
                    if (ff.rewrite) prw[(size_t)q].push_back(Rewrite{(uint32_t)vi.size(), true, ff});
                    else if (is_na(b + ft.off, ft.len)) ft.len = 0;
                    vi.push_back(id);
                    vf.push_back(ft);
                    p = next;
                    continue;
                }
                int col = 0;
                Span id{0, -1}, ft{0, -1};
                for (int64_t s0 = p;;) {
                    const char *c = (const char *)memchr(b + s0, sp, (size_t)(le - s0));
                    const int64_t e = c ? c - b : le;
                    if (e - s0 > INT32_MAX) {
                        why[(size_t)q] = "field longer than 2 GiB";
                        return;
                    }
                    if (col == id_i) id = Span{s0, (int32_t)(e - s0)};
                    if (col == ft_i) ft = Span{s0, (int32_t)(e - s0)};
                    col++;
                    if (!c) break;
                    s0 = e + 1;
                }
                if (col != ncols) {
                    why[(size_t)q] = "ragged row";
                    return;
                }
                if (is_na(b + id.off, id.len)) {
                    why[(size_t)q] = "NA-valued id";
                    return;
                }
                if (is_na(b + ft.off, ft.len)) ft.len = 0;  // NaN -> "" (fillna, :28)
                vi.push_back(id);
                vf.push_back(ft);
                p = lf + 1;
            }
            if (pe < n && p != pe) off_boundary.store(1);  // (the last slice ends on the sentinel, or one behind it)
        });
        tm.lap("open: lines and columns");
        if (off_boundary.load() && attempt == 0) continue;  // (a line break inside a quoted field at a slice boundary: one slice)
        size_t total = 0;
        for (int q = 0; q < parts; q++) {
            if (why[(size_t)q]) {
                const std::string w = why[(size_t)q];
                delete t;
                return unsupported(w);
            }
            total += pid[(size_t)q].size();
        }
        // fields with doubled quotes, or with bytes behind their closing quote: the content, written over the raw field
        char *mb = t->bytes.data();
        parallel_chunks(parts, [&](int q) {
            for (const Rewrite &w : prw[(size_t)q]) {
                Span &sp_ = (w.is_ft ? pft : pid)[(size_t)q][w.row];
                sp_ = Span{w.f.s0, (int32_t)unquote_field(mb, w.f)};
                if (is_na(mb + sp_.off, sp_.len)) {
                    if (w.is_ft) sp_.len = 0;
                    else why[(size_t)q] = "NA-valued id";
                }
            }
        });
        for (int q = 0; q < parts; q++)
            if (why[(size_t)q]) {
                const std::string w = why[(size_t)q];
                delete t;
                return unsupported(w);
            }
        t->ids.reserve(total);
        t->feats.reserve(total);
        for (int q = 0; q < parts; q++) {
            t->ids.insert(t->ids.end(), pid[(size_t)q].begin(), pid[(size_t)q].end());
            t->feats.insert(t->feats.end(), pft[(size_t)q].begin(), pft[(size_t)q].end());
        }
        break;
    }
    tm.lap("open: concatenate");
    if (t->ids.empty()) {
        delete t;
        return unsupported("no data rows");
    }
    if (t->ids.size() > (size_t)INT32_MAX) {
        delete t;
        return unsupported("more than 2^31-1 rows");
    }
    if (!ids_distinct(*t)) {
        delete t;
        return unsupported("duplicate sequence identifiers");
    }
    tm.lap("open: ids distinct");
    *out = t;
    return BFK_OK;
}

extern "C" int bfk_table_from_buffers(const char *id_buf, const int64_t *id_off, const char *feat_buf, const int64_t *feat_off,
                                      int64_t n_rows, bfk_table **out) {
    if (!id_off || !feat_off || !out || n_rows < 0 || n_rows > INT32_MAX) return bfk_fail(BFK_EARG, "bfk_table_from_buffers: bad argument");
    const int64_t ni = id_off[n_rows], nf = feat_off[n_rows];
    if (ni < 0 || nf < 0 || (ni > 0 && !id_buf) || (nf > 0 && !feat_buf)) return bfk_fail(BFK_EARG, "bfk_table_from_buffers: bad offsets");
    bfk_table *t = new bfk_table();
    t->bytes.resize((size_t)(ni + nf + 1));
    if (ni) memcpy(t->bytes.data(), id_buf, (size_t)ni);
    if (nf) memcpy(t->bytes.data() + ni, feat_buf, (size_t)nf);
    for (int64_t i = 0; i < ni + nf; i++) {
        const unsigned char c = (unsigned char)t->bytes[(size_t)i];
        if (c == '\r' || c == '\n' || c == 0) {
            delete t;
            return unsupported("CR, LF or NUL byte");
        }
    }
    t->bytes[(size_t)(ni + nf)] = '\n';
    t->any_high = has_high_byte(t->bytes.data(), ni + nf);
    if (t->any_high) {  // every id and every feature has to be valid UTF-8 by itself (they were Python strings)
        bool ok = true;
        for (int64_t r = 0; ok && r < n_rows; r++) {
            ok = utf8_valid_from(t->bytes.data(), id_off[r], id_off[r + 1], id_off[r + 1] - 1, true) &&
                 utf8_valid_from(t->bytes.data(), ni + feat_off[r], ni + feat_off[r + 1], ni + feat_off[r + 1] - 1, true);
        }
        if (!ok) {
            delete t;
            return unsupported("bytes that are not valid UTF-8");
        }
    }
    t->ids.resize((size_t)n_rows);
    t->feats.resize((size_t)n_rows);
    for (int64_t r = 0; r < n_rows; r++) {
        const int64_t a = id_off[r + 1] - id_off[r], c = feat_off[r + 1] - feat_off[r];
        if (a < 0 || c < 0 || a > INT32_MAX || c > INT32_MAX) {
            delete t;
            return bfk_fail(BFK_EARG, "bfk_table_from_buffers: offsets not monotone");
        }
        t->ids[(size_t)r] = Span{id_off[r], (int32_t)a};
        t->feats[(size_t)r] = Span{ni + feat_off[r], (int32_t)c};
    }
    *out = t;
    return BFK_OK;
}

extern "C" int64_t bfk_table_rows(const bfk_table *t) { return t ? (int64_t)t->ids.size() : -1; }
extern "C" void bfk_table_close(bfk_table *t) { delete t; }

extern "C" int bfk_table_feature_high(const bfk_table *t);
extern "C" int bfk_table_prepare(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, bfk_prep_info *info) {
    if (!t || !sep2 || !opts || !info) return bfk_fail(BFK_EARG, "bfk_table_prepare: null argument");
    if (sep2_len <= 0) return bfk_fail(BFK_EARG, "empty separator");
    if (opts->var_type < BFK_VAR_COVSONAR_DNA || opts->var_type > BFK_VAR_RAW) return bfk_fail(BFK_EARG, "bfk_table_prepare: unknown var_type");
    for (int64_t i = 0; i < sep2_len; i++) {
        const unsigned char c = (unsigned char)sep2[i];
        if (c >= 0x80 || c == '\n' || c == '\r' || c == 0) return unsupported("token separator");
    }
    const int64_t n = (int64_t)t->ids.size();
    const char *b = t->bytes.data();
    Classifier cls{*opts, opts->reference_length - opts->trim_end};
    // nothing to filter: features are taken verbatim, identity = the raw string (:128-129)
    const bool filtering = opts->skip_del || opts->skip_ins || opts->trim_start > 0 || opts->trim_end > 0;
    if (filtering && t->any_high && opts->var_type != BFK_VAR_RAW) {
        // the reference's patterns (:135-155) are str patterns: \d also matches the decimal digits of other scripts, which the
        // ASCII matchers here do not restate — a FEATURE with non-ASCII bytes under a grammar is the general path's (ids and
        // other columns may hold what they like)
        if (bfk_table_feature_high(t)) return unsupported("non-ASCII bytes in a feature that is matched against the token patterns");
    }
    t->filtered = filtering;
    t->device_prepared = false;
    t->sep2.assign(sep2, (size_t)sep2_len);
    t->group.assign((size_t)n, 0);
    t->weight.clear();
    t->first_row.clear();
    t->invalid.clear();
    t->vocab.clear();

    StageTimer tm;
    // phases A-C (row-parallel): tokens, verdicts, first-appearance vocabulary ids
    std::vector<TokChunk> cks = make_chunks(n, (int64_t)t->bytes.size());
    const auto span = [&](int64_t r) { return t->feats[(size_t)r]; };
    parallel_chunks((int)cks.size(), [&](int q) { tokenize_chunk(b, span, sep2, sep2_len, filtering ? &cls : nullptr, cks[(size_t)q]); });
    tm.lap("prepare: tokenise chunks");
    merge_vocab(b, cks, t->vocab);
    for (const TokChunk &c : cks) t->invalid.insert(t->invalid.end(), c.invalid.begin(), c.invalid.end());
    tm.lap("prepare: merge vocabulary");

    // identity of the (filtered) feature string: the kept token sequence when the string is re-joined, the raw bytes when
    // it is passed through untouched.  Hashes row-parallel, the first-appearance grouping in row order.
    std::vector<uint64_t> rh((size_t)n);
    std::vector<const int32_t *> rp((size_t)n);  // the row's ids
    std::vector<int32_t> rl((size_t)n);          // ... and how many
    parallel_chunks((int)cks.size(), [&](int q) {
        const TokChunk &c = cks[(size_t)q];
        static const int32_t none = 0;
        for (int64_t r = c.r0; r < c.r1; r++) {
            const int64_t e0 = r == c.r0 ? 0 : c.row_end[(size_t)(r - c.r0 - 1)], e1 = c.row_end[(size_t)(r - c.r0)];
            rp[(size_t)r] = c.ids.data() + e0;
            rl[(size_t)r] = (int32_t)(e1 - e0);
            const Span f = t->feats[(size_t)r];
            rh[(size_t)r] = filtering ? bytes_hash((const char *)(e1 > e0 ? c.ids.data() + e0 : &none), (size_t)(e1 - e0) * sizeof(int32_t))
                                      : bytes_hash(b + f.off, (size_t)f.len);
        }
    });
    tm.lap("prepare: row hashes");
    size_t rcap = 1u << 12;
    while (rcap < (size_t)n * 2) rcap <<= 1;
    std::vector<int32_t> rtab(rcap, -1);  // row hash table -> unique index
    int64_t nnz = 0;
    for (int64_t r = 0; r < n; r++) {
        const uint64_t h = rh[(size_t)r];
        size_t i = h & (rcap - 1);
        int32_t u = -1;
        while (rtab[i] >= 0) {
            const int32_t c = rtab[i];
            const int64_t fr = t->first_row[(size_t)c];
            if (rh[(size_t)fr] == h) {
                bool same;
                if (filtering) {
                    same = rl[(size_t)fr] == rl[(size_t)r] &&
                           (rl[(size_t)r] == 0 || memcmp(rp[(size_t)fr], rp[(size_t)r], (size_t)rl[(size_t)r] * sizeof(int32_t)) == 0);
                } else {
                    const Span o = t->feats[(size_t)fr], f = t->feats[(size_t)r];
                    same = o.len == f.len && memcmp(b + o.off, b + f.off, (size_t)f.len) == 0;
                }
                if (same) {
                    u = c;
                    break;
                }
            }
            i = (i + 1) & (rcap - 1);
        }
        if (u < 0) {
            u = (int32_t)t->first_row.size();
            rtab[i] = u;
            t->first_row.push_back((int32_t)r);
            t->weight.push_back(0);
            nnz += rl[(size_t)r];
        }
        t->group[(size_t)r] = u;
        t->weight[(size_t)u]++;
    }
    tm.lap("prepare: collapse");
    if (nnz > (int64_t)INT32_MAX) return bfk_fail(BFK_EARG, "bfk_table_prepare: more than 2^31-1 entries");
    // CSR of the unique rows
    const int64_t nu = (int64_t)t->first_row.size();
    t->indptr.assign((size_t)nu + 1, 0);
    for (int64_t u = 0; u < nu; u++) t->indptr[(size_t)u + 1] = t->indptr[(size_t)u] + rl[(size_t)t->first_row[(size_t)u]];
    t->indices.reserve((size_t)std::max<int64_t>(nnz, 1));  // (data() must not be NULL for the views)
    t->indices.resize((size_t)nnz);
    {
        const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), nu / 8192 + 1));
        parallel_chunks(parts, [&](int q) {
            for (int64_t u = nu * q / parts; u < nu * (q + 1) / parts; u++) {
                const int64_t fr = t->first_row[(size_t)u];
                if (rl[(size_t)fr]) memcpy(t->indices.data() + t->indptr[(size_t)u], rp[(size_t)fr], (size_t)rl[(size_t)fr] * sizeof(int32_t));
            }
        });
    }
    tm.lap("prepare: CSR");
    t->n_vocab = (int32_t)t->vocab.size();
    t->prepared = true;
    info->n_rows = n;
    info->n_unique = nu;
    info->nnz = nnz;
    info->n_invalid = (int64_t)t->invalid.size();
    info->n_vocab = t->n_vocab;
    info->filtered = filtering ? 1 : 0;
    return BFK_OK;
}

// ---- the table's raw material for a prepare that runs elsewhere (libbfk.so: the device stages), and the way back ----------
extern "C" int bfk_table_raw(const bfk_table *t, const char **bytes_out, int64_t *n_bytes_out, const void **feat_spans_out,
                             int64_t *span_stride_out, int64_t *n_rows_out) {
    if (!t || !bytes_out || !n_bytes_out || !feat_spans_out || !span_stride_out || !n_rows_out) return bfk_fail(BFK_EARG, "bfk_table_raw: null argument");
    *bytes_out = t->bytes.data();
    *n_bytes_out = (int64_t)t->bytes.size();  // (includes the sentinel byte behind the last line)
    *feat_spans_out = t->feats.data();
    *span_stride_out = (int64_t)sizeof(Span);
    *n_rows_out = (int64_t)t->feats.size();
    return BFK_OK;
}

extern "C" int bfk_table_any_high(const bfk_table *t) { return t && t->any_high ? 1 : 0; }

// does any FEATURE hold a byte >= 0x80 (ids and other columns may): what decides whether the token patterns' ASCII matchers — the
// host stage's and the device stage's — may take a table whose bytes are not all ASCII
extern "C" int bfk_table_feature_high(const bfk_table *t) {
    if (!t || !t->any_high) return 0;
    const int64_t n = (int64_t)t->feats.size();
    const char *b = t->bytes.data();
    std::atomic<int> high{0};
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / 16384 + 1));
    parallel_chunks(parts, [&](int q) {
        for (int64_t r = n * q / parts, e = n * (q + 1) / parts; r < e && !high.load(std::memory_order_relaxed); r++)
            if (has_high_byte(b + t->feats[(size_t)r].off, t->feats[(size_t)r].len)) high.store(1);
    });
    return high.load();
}

extern "C" int bfk_table_set_prepared(bfk_table *t, const int32_t *group, const int32_t *first_row, int64_t n_unique, const bfk_prep_info *info,
                                      const int32_t *indptr, const int32_t *indices, const char *sep2, int64_t sep2_len) {
    if (!t || !group || !first_row || !info || n_unique < 0 || !sep2) return bfk_fail(BFK_EARG, "bfk_table_set_prepared: bad argument");
    const int64_t n = (int64_t)t->ids.size();
    t->group.assign(group, group + n);
    t->first_row.assign(first_row, first_row + n_unique);
    t->weight.assign((size_t)n_unique, 0);
    for (int64_t r = 0; r < n; r++) {
        if (group[r] < 0 || group[r] >= n_unique) return bfk_fail(BFK_EARG, "bfk_table_set_prepared: group index out of range");
        t->weight[(size_t)group[r]]++;
    }
    t->indptr.clear();
    t->indices.clear();
    if (indptr) {
        t->indptr.assign(indptr, indptr + n_unique + 1);
        t->indices.reserve((size_t)std::max<int64_t>(info->nnz, 1));
        t->indices.resize((size_t)info->nnz);
        if (info->nnz) memcpy(t->indices.data(), indices, (size_t)info->nnz * sizeof(int32_t));
    }
    t->invalid.clear();
    t->vocab.clear();  // (the vocabulary's bytes stay on the device: bfk_table_feature needs the host prepare)
    t->sep2.assign(sep2, (size_t)sep2_len);
    t->filtered = info->filtered != 0;
    t->n_vocab = info->n_vocab;
    t->prepared = true;
    t->device_prepared = true;
    return BFK_OK;
}

// the invalid token occurrences of a device prepare, in the reference's print order: spans into the table's bytes (length 0: an
// empty token).  n == 0 with info.n_invalid > 0: every one of them is an empty token.
extern "C" int bfk_table_set_invalid(bfk_table *t, const int64_t *off, const int32_t *len, int64_t n) {
    if (!t || !t->prepared || n < 0 || (n > 0 && (!off || !len))) return bfk_fail(BFK_EARG, "bfk_table_set_invalid: bad argument");
    t->invalid.clear();
    t->invalid.reserve((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if (off[i] < 0 || len[i] < 0 || off[i] + len[i] > (int64_t)t->bytes.size()) return bfk_fail(BFK_EARG, "bfk_table_set_invalid: span outside the table");
        t->invalid.push_back(Span{off[i], len[i]});
    }
    return BFK_OK;
}

extern "C" const int32_t *bfk_table_group(const bfk_table *t) { return t && t->prepared ? t->group.data() : nullptr; }
extern "C" const int32_t *bfk_table_weight(const bfk_table *t) { return t && t->prepared ? t->weight.data() : nullptr; }
extern "C" const int32_t *bfk_table_indptr(const bfk_table *t) { return t && t->prepared ? t->indptr.data() : nullptr; }
extern "C" const int32_t *bfk_table_indices(const bfk_table *t) { return t && t->prepared ? t->indices.data() : nullptr; }

extern "C" int64_t bfk_table_invalid_count(const bfk_table *t) { return t && t->prepared ? (int64_t)t->invalid.size() : -1; }

extern "C" int bfk_table_invalid(const bfk_table *t, int64_t i, const char **tok_out, int64_t *len_out) {
    if (!t || !t->prepared || !tok_out || !len_out || i < 0 || i >= (int64_t)t->invalid.size()) return bfk_fail(BFK_EARG, "bfk_table_invalid: bad argument");
    *tok_out = t->bytes.data() + t->invalid[(size_t)i].off;
    *len_out = t->invalid[(size_t)i].len;
    return BFK_OK;
}

extern "C" int bfk_table_id(const bfk_table *t, int64_t r, const char **id_out, int64_t *len_out) {
    if (!t || !id_out || !len_out || r < 0 || r >= (int64_t)t->ids.size()) return bfk_fail(BFK_EARG, "bfk_table_id: bad argument");
    *id_out = t->bytes.data() + t->ids[(size_t)r].off;
    *len_out = t->ids[(size_t)r].len;
    return BFK_OK;
}

extern "C" int bfk_table_feature(const bfk_table *t, int64_t u, char **str_out, int64_t *len_out) {
    if (t && t->device_prepared && t->filtered) return bfk_fail(BFK_ESTATE, "the filtered feature strings need bfk_table_prepare (the device prepare keeps the vocabulary in HBM)");
    if (!t || !t->prepared || !str_out || !len_out || u < 0 || u >= (int64_t)t->first_row.size()) return bfk_fail(BFK_EARG, "bfk_table_feature: bad argument");
    std::string s;
    if (!t->filtered) {
        const Span f = t->feats[(size_t)t->first_row[(size_t)u]];
        s.assign(t->bytes.data() + f.off, (size_t)f.len);
    } else {
        for (int32_t j = t->indptr[(size_t)u]; j < t->indptr[(size_t)u + 1]; j++) {
            if (j > t->indptr[(size_t)u]) s += t->sep2;
            const Span v = t->vocab[(size_t)t->indices[(size_t)j]];
            s.append(t->bytes.data() + v.off, (size_t)v.len);
        }
    }
    char *o = (char *)malloc(s.size() + 1);
    if (!o) return bfk_fail(BFK_ENOMEM, "bfk_table_feature: out of memory");
    memcpy(o, s.data(), s.size());
    o[s.size()] = 0;
    *str_out = o;
    *len_out = (int64_t)s.size();
    return BFK_OK;
}

extern "C" int bfk_table_write(const bfk_table *t, const char *path, const int32_t *cluster_of_unique, int64_t *n_clusters_out) {
    if (!t || !t->prepared || !path || !cluster_of_unique) return bfk_fail(BFK_EARG, "bfk_table_write: bad argument");
    const size_t nu = t->first_row.size();
    int32_t mx = 0;
    for (size_t u = 0; u < nu; u++) {
        if (cluster_of_unique[u] < 0) return bfk_fail(BFK_EARG, "bfk_table_write: negative cluster number");
        mx = std::max(mx, cluster_of_unique[u]);
    }
    std::vector<int32_t> remap((size_t)mx + 1, 0);  // first appearance in input order -> 1.. (:51-60)
    int32_t next = 0;
    const size_t n = t->ids.size();
    for (size_t r = 0; r < n; r++) {  // (the numbering is the one thing that depends on the order: a pass of its own, then the
                                      // lines are formatted in parallel slices — one serial loop with snprintf was 70 ms at 1M rows)
        const int32_t c = cluster_of_unique[(size_t)t->group[r]];
        if (c && !remap[(size_t)c]) remap[(size_t)c] = ++next;
    }
    const char *b = t->bytes.data();
    const int parts = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), n / 32768 + 1));
    std::vector<std::string> piece((size_t)parts);
    parallel_chunks(parts, [&](int q) {
        std::string &out = piece[(size_t)q];
        const size_t r0 = n * (size_t)q / (size_t)parts, r1 = n * ((size_t)q + 1) / (size_t)parts;
        out.reserve((r1 - r0) * 24 + 64);
        if (q == 0) out += "id\tcluster_id\n";
        char num[16];
        for (size_t r = r0; r < r1; r++) {
            const Span id = t->ids[r];
            // csv.QUOTE_MINIMAL (to_csv, :67-69): a field holding the output delimiter, a quote or a line break (an id that
            // was quoted in the input can) is quoted, its quotes doubled
            bool q2 = false, dq = false;
            for (int32_t k = 0; k < id.len; k++) {
                const char ch = b[id.off + k];
                q2 = q2 || ch == '\t' || ch == '"' || ch == '\n' || ch == '\r';
                dq = dq || ch == '"';
            }
            if (q2) out += '"';
            if (!dq) out.append(b + id.off, (size_t)id.len);
            else
                for (int32_t k = 0; k < id.len; k++) {
                    out += b[id.off + k];
                    if (b[id.off + k] == '"') out += '"';
                }
            if (q2) out += '"';
            out += '\t';
            const int32_t c = cluster_of_unique[(size_t)t->group[r]];
            if (c) {
                uint32_t v = (uint32_t)remap[(size_t)c];
                int m = 0;
                do {
                    num[sizeof num - 1 - (size_t)m++] = (char)('0' + v % 10);
                    v /= 10;
                } while (v);
                out.append(num + sizeof num - (size_t)m, (size_t)m);
            }
            out += '\n';
        }
    });
    FILE *f = fopen(path, "wb");
    if (!f) return bfk_fail(BFK_EIO, std::string("cannot write ") + path);
    bool ok = true;
    for (const std::string &out : piece) ok = ok && fwrite(out.data(), 1, out.size(), f) == out.size();
    if (fclose(f) != 0 || !ok) return bfk_fail(BFK_EIO, std::string("short write on ") + path);
    if (n_clusters_out) *n_clusters_out = next;
    return BFK_OK;
}

extern "C" int bfk_table_cluster_write(const bfk_table *t, int32_t max_dist, int32_t min_cluster_size, int32_t n_gpus, const char *path,
                                       int64_t *n_clusters_out) {
    if (!t || !t->prepared || !path || max_dist < 0 || min_cluster_size < 0) return bfk_fail(BFK_EARG, "bfk_table_cluster_write: bad argument");
    const size_t nu = t->first_row.size();
    std::vector<int32_t> labels(std::max<size_t>(nu, 1));
    if (max_dist == 0) {
        for (size_t u = 0; u < nu; u++) labels[u] = (int32_t)u;  // every unique string alone (:343-364)
    } else if (int rc = bfk_front_cluster(t->indptr.data(), t->indices.data(), (int64_t)nu, max_dist, n_gpus, labels.data())) {
        return rc;
    }
    // a component counts the ORIGINAL sequences of its rows (:329-339); labels are the component's smallest row
    std::vector<int64_t> size(std::max<size_t>(nu, 1), 0);
    for (size_t u = 0; u < nu; u++) {
        if (labels[u] < 0 || (size_t)labels[u] >= nu) return bfk_fail(BFK_EARG, "bfk_table_cluster_write: label out of range");
        size[(size_t)labels[u]] += t->weight[u];
    }
    std::vector<int32_t> cl(std::max<size_t>(nu, 1), 0);
    for (size_t u = 0; u < nu; u++) cl[u] = size[(size_t)labels[u]] >= min_cluster_size ? labels[u] + 1 : 0;
    return bfk_table_write(t, path, cl.data(), n_clusters_out);
}

// bulk accessors for callers that need the strings themselves (the cache path keeps the reference's pickle format, which
// stores the unique rows' feature strings and id tuples): one buffer + offsets instead of a call per row
extern "C" int bfk_table_features(const bfk_table *t, char **buf_out, int64_t **off_out) {
    if (t && t->device_prepared && t->filtered) return bfk_fail(BFK_ESTATE, "the filtered feature strings need bfk_table_prepare (the device prepare keeps the vocabulary in HBM)");
    if (!t || !t->prepared || !buf_out || !off_out) return bfk_fail(BFK_EARG, "bfk_table_features: bad argument");
    const size_t nu = t->first_row.size();
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (nu + 1));
    if (!off) return bfk_fail(BFK_ENOMEM, "bfk_table_features: out of memory");
    off[0] = 0;
    const size_t sl = t->sep2.size();
    for (size_t u = 0; u < nu; u++) {
        int64_t len = 0;
        if (!t->filtered) {
            len = t->feats[(size_t)t->first_row[u]].len;
        } else {
            const int32_t b = t->indptr[u], e = t->indptr[u + 1];
            for (int32_t j = b; j < e; j++) len += t->vocab[(size_t)t->indices[(size_t)j]].len;
            if (e > b) len += (int64_t)sl * (e - b - 1);
        }
        off[u + 1] = off[u] + len;
    }
    char *buf = (char *)malloc((size_t)std::max<int64_t>(off[nu], 1));
    if (!buf) {
        free(off);
        return bfk_fail(BFK_ENOMEM, "bfk_table_features: out of memory");
    }
    const char *base = t->bytes.data();
    const int parts = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), nu / 8192 + 1));
    parallel_chunks(parts, [&](int q) {
        for (size_t u = nu * (size_t)q / (size_t)parts; u < nu * (size_t)(q + 1) / (size_t)parts; u++) {
            char *o = buf + off[u];
            if (!t->filtered) {
                const Span f = t->feats[(size_t)t->first_row[u]];
                memcpy(o, base + f.off, (size_t)f.len);
            } else {
                for (int32_t j = t->indptr[u]; j < t->indptr[u + 1]; j++) {
                    if (j > t->indptr[u]) {
                        memcpy(o, t->sep2.data(), sl);
                        o += sl;
                    }
                    const Span v = t->vocab[(size_t)t->indices[(size_t)j]];
                    memcpy(o, base + v.off, (size_t)v.len);
                    o += v.len;
                }
            }
        }
    });
    *buf_out = buf;
    *off_out = off;
    return BFK_OK;
}

// ---- side-car cache (breakfast_amd/sidecar.py): 128-bit hashes of feature strings ----------------------------------------
// The reference's cache (src/breakfast/cache.py:94-112) matches the rows of a new input to the cached ones by their feature
// STRING.  The side-car keeps two 64-bit hashes of the string instead of the string (16 bytes per row instead of ~330).
namespace {
inline uint64_t bytes_hash2(const char *p, size_t n) {  // independent of bytes_hash: other seed, other multiplier
    uint64_t h = 0xD6E8FEB86659FD93ull ^ (n * 0x9FB21C651E98DF25ull);
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        h = (h ^ v) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 31;
        p += 8;
        n -= 8;
    }
    uint64_t v = 0;
    memcpy(&v, p, n);
    h = (h ^ v) * 0xFF51AFD7ED558CCDull;
    return h ^ (h >> 29);
}
}  // namespace

// out[2 r], out[2 r + 1] = the two hashes of row r = buf[off[r] .. off[r + 1])
extern "C" int bfk_hash_rows(const char *buf, const int64_t *off, int64_t n_rows, uint64_t *out) {
    if (!off || !out || n_rows < 0 || (!buf && n_rows > 0 && off[n_rows] > off[0])) return bfk_fail(BFK_EARG, "bfk_hash_rows: null argument");
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)host_threads(), n_rows / 8192 + 1));
    parallel_chunks(parts, [&](int q) {
        for (int64_t r = n_rows * q / parts; r < n_rows * (q + 1) / parts; r++) {
            const size_t len = (size_t)std::max<int64_t>(0, off[r + 1] - off[r]);
            out[2 * r] = bytes_hash(buf + off[r], len);
            out[2 * r + 1] = bytes_hash2(buf + off[r], len);
        }
    });
    return BFK_OK;
}

// a[2 i .. 2 i + 1] / b[2 j .. 2 j + 1]: hash pairs (the rows of b are distinct); out[i] = the j with b[j] == a[i], or -1
// (cache.map_features, src/breakfast/cache.py:94-112, on the side-car's hashes)
extern "C" int bfk_match_hashes(const uint64_t *a, int64_t n_a, const uint64_t *b, int64_t n_b, int64_t *out) {
    if ((n_a > 0 && (!a || !out)) || (n_b > 0 && !b) || n_a < 0 || n_b < 0) return bfk_fail(BFK_EARG, "bfk_match_hashes: bad argument");
    size_t cap = 16;
    while (cap < (size_t)n_b * 2 + 16) cap <<= 1;
    std::vector<int64_t> slot(cap, -1);
    for (int64_t j = 0; j < n_b; j++) {
        size_t i = (size_t)b[2 * j] & (cap - 1);
        while (slot[i] >= 0) i = (i + 1) & (cap - 1);
        slot[i] = j;
    }
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)host_threads(), n_a / 65536 + 1));
    parallel_chunks(parts, [&](int q) {
        for (int64_t r = n_a * q / parts; r < n_a * (q + 1) / parts; r++) {
            size_t i = (size_t)a[2 * r] & (cap - 1);
            int64_t hit = -1;
            while (slot[i] >= 0) {
                const int64_t j = slot[i];
                if (b[2 * j] == a[2 * r] && b[2 * j + 1] == a[2 * r + 1]) {
                    hit = j;
                    break;
                }
                i = (i + 1) & (cap - 1);
            }
            out[r] = hit;
        }
    });
    return BFK_OK;
}

// the same for the filtered feature strings of a prepared table's unique rows (what bfk_table_features would hand out),
// without building the strings' blob: out[2 u], out[2 u + 1]
extern "C" int bfk_table_feature_hashes(const bfk_table *t, uint64_t *out) {
    if (t && t->device_prepared && t->filtered) return bfk_fail(BFK_ESTATE, "the filtered feature strings need bfk_table_prepare (the device prepare keeps the vocabulary in HBM)");
    if (!t || !t->prepared || !out) return bfk_fail(BFK_EARG, "bfk_table_feature_hashes: bad argument");
    const size_t nu = t->first_row.size();
    const char *base = t->bytes.data();
    const size_t sl = t->sep2.size();
    const int parts = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), nu / 8192 + 1));
    parallel_chunks(parts, [&](int q) {
        std::string tmp;
        for (size_t u = nu * (size_t)q / (size_t)parts; u < nu * (size_t)(q + 1) / (size_t)parts; u++) {
            const char *p;
            size_t len;
            if (!t->filtered) {
                const Span f = t->feats[(size_t)t->first_row[u]];
                p = base + f.off;
                len = (size_t)f.len;
            } else {
                tmp.clear();
                for (int32_t j = t->indptr[u]; j < t->indptr[u + 1]; j++) {
                    if (j > t->indptr[u]) tmp.append(t->sep2.data(), sl);
                    const Span v = t->vocab[(size_t)t->indices[(size_t)j]];
                    tmp.append(base + v.off, (size_t)v.len);
                }
                p = tmp.data();
                len = tmp.size();
            }
            out[2 * u] = bytes_hash(p, len);
            out[2 * u + 1] = bytes_hash2(p, len);
        }
    });
    return BFK_OK;
}

extern "C" int bfk_table_ids(const bfk_table *t, char **buf_out, int64_t **off_out) {
    if (!t || !buf_out || !off_out) return bfk_fail(BFK_EARG, "bfk_table_ids: bad argument");
    const size_t n = t->ids.size();
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    if (!off) return bfk_fail(BFK_ENOMEM, "bfk_table_ids: out of memory");
    off[0] = 0;
    for (size_t r = 0; r < n; r++) off[r + 1] = off[r] + t->ids[r].len;
    char *buf = (char *)malloc((size_t)std::max<int64_t>(off[n], 1));
    if (!buf) {
        free(off);
        return bfk_fail(BFK_ENOMEM, "bfk_table_ids: out of memory");
    }
    const char *base = t->bytes.data();
    for (size_t r = 0; r < n; r++) memcpy(buf + off[r], base + t->ids[r].off, (size_t)t->ids[r].len);
    *buf_out = buf;
    *off_out = off;
    return BFK_OK;
}
