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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int bfk_fail(int code, const std::string &msg);  // bfk_host.cpp: sets the thread-local message

namespace {

struct Span {
    int64_t off;
    int32_t len;
};

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

struct TokSlot {
    const char *p;
    uint32_t len;
    int32_t id;  // vocabulary id, -1 until the token is first kept
    int8_t verdict;
    uint8_t used;
};

}  // namespace

struct bfk_table {
    std::vector<char> bytes;  // owned copy of the file / buffers
    std::vector<Span> ids, feats;
    // prepare() results
    std::vector<int32_t> group, weight, first_row, indptr, indices;
    std::vector<Span> invalid;  // into `bytes`
    std::string sep2;
    bool filtered = false, prepared = false;
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

int unsupported(const std::string &why) { return bfk_fail(BFK_EUNSUPPORTED, "bfk_table: input needs the general reader: " + why); }

// every id distinct?  (read_input raises on duplicates, :24-27: left to the pandas path)
bool ids_distinct(const bfk_table &t) {
    size_t cap = 16;
    while (cap < t.ids.size() * 2) cap <<= 1;
    std::vector<int32_t> tab(cap, -1);
    const char *b = t.bytes.data();
    for (size_t r = 0; r < t.ids.size(); r++) {
        const Span s = t.ids[r];
        size_t i = bytes_hash(b + s.off, (size_t)s.len) & (cap - 1);
        while (tab[i] >= 0) {
            const Span o = t.ids[(size_t)tab[i]];
            if (o.len == s.len && memcmp(b + o.off, b + s.off, (size_t)s.len) == 0) return false;
            i = (i + 1) & (cap - 1);
        }
        tab[i] = (int32_t)r;
    }
    return true;
}

}  // namespace

extern "C" int bfk_table_open(const char *path, const char *sep, int64_t sep_len, const char *id_col, const char *feature_col,
                              bfk_table **out) {
    if (!path || !sep || !id_col || !feature_col || !out) return bfk_fail(BFK_EARG, "bfk_table_open: null argument");
    if (sep_len != 1) return unsupported("multi-byte column separator");
    const char sp = sep[0];
    if (sp == '\n' || sp == '\r' || sp == '"' || sp == 0 || (unsigned char)sp >= 0x80) return unsupported("separator");
    if (strcmp(id_col, feature_col) == 0) return unsupported("id and feature column are the same");
    FILE *f = fopen(path, "rb");
    if (!f) return bfk_fail(BFK_EIO, std::string("cannot open ") + path);
    bfk_table *t = new bfk_table();
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    t->bytes.resize((size_t)std::max<long>(sz, 0) + 1);
    const size_t got = sz > 0 ? fread(t->bytes.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    if ((long)got != sz) {
        delete t;
        return bfk_fail(BFK_EIO, std::string("short read on ") + path);
    }
    t->bytes[(size_t)sz] = '\n';  // sentinel: the last line always ends
    const char *b = t->bytes.data();
    const int64_t n = sz;
    for (int64_t i = 0; i < n; i++) {
        const unsigned char c = (unsigned char)b[i];
        // CR is accepted only as part of a CRLF line end (the reference's own fixtures are CRLF files)
        if (c >= 0x80 || c == '"' || c == 0 || (c == '\r' && b[i + 1] != '\n')) {
            delete t;
            return unsupported("quote, lone CR, NUL or non-ASCII byte");
        }
    }
    // end of the line starting at p: index of its LF, and the end of its content (CR of a CRLF stripped)
    auto line_end = [&](int64_t p, int64_t *content_end) {
        const int64_t lf = (const char *)memchr(b + p, '\n', (size_t)(n + 1 - p)) - b;
        *content_end = (lf > p && b[lf - 1] == '\r') ? lf - 1 : lf;
        return lf;
    };
    // header = first non-empty line
    int64_t pos = 0, he = 0;
    while (pos < n) {
        const int64_t lf = line_end(pos, &he);
        if (he > pos) break;
        pos = lf + 1;
    }
    if (pos >= n) {
        delete t;
        return unsupported("empty file");
    }
    const int64_t header_lf = line_end(pos, &he);
    int ncols = 0, id_i = -1, ft_i = -1;
    std::vector<std::string> names;
    for (int64_t s = pos;;) {
        const char *c = (const char *)memchr(b + s, sp, (size_t)(he - s));
        const int64_t e = c ? c - b : he;
        names.emplace_back(b + s, (size_t)(e - s));
        ncols++;
        if (!c) break;
        s = e + 1;
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
    pos = header_lf + 1;
    t->ids.reserve((size_t)(n / 256 + 16));
    t->feats.reserve((size_t)(n / 256 + 16));
    while (pos < n) {
        int64_t le;
        const int64_t lf = line_end(pos, &le);
        if (le == pos) {  // blank line: skipped
            pos = lf + 1;
            continue;
        }
        int col = 0;
        Span id{0, -1}, ft{0, -1};
        for (int64_t s = pos;;) {
            const char *c = (const char *)memchr(b + s, sp, (size_t)(le - s));
            const int64_t e = c ? c - b : le;
            if (e - s > INT32_MAX) {
                delete t;
                return unsupported("field longer than 2 GiB");
            }
            if (col == id_i) id = Span{s, (int32_t)(e - s)};
            if (col == ft_i) ft = Span{s, (int32_t)(e - s)};
            col++;
            if (!c) break;
            s = e + 1;
        }
        if (col != ncols) {
            delete t;
            return unsupported("ragged row");
        }
        if (is_na(b + id.off, id.len)) {
            delete t;
            return unsupported("NA-valued id");
        }
        if (is_na(b + ft.off, ft.len)) ft.len = 0;  // NaN -> "" (fillna, :28)
        t->ids.push_back(id);
        t->feats.push_back(ft);
        pos = lf + 1;
    }
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
        if (c >= 0x80 || c == '\r' || c == '\n' || c == 0) {
            delete t;
            return unsupported("CR, LF, NUL or non-ASCII byte");
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
    t->filtered = filtering;
    t->sep2.assign(sep2, (size_t)sep2_len);
    t->group.assign((size_t)n, 0);
    t->weight.clear();
    t->first_row.clear();
    t->indptr.assign(1, 0);
    t->indices.clear();
    t->invalid.clear();
    t->vocab.clear();
    t->indices.reserve(t->bytes.size() / 6);

    size_t tcap = 1u << 16, tcount = 0;
    std::vector<TokSlot> tab(tcap, TokSlot{nullptr, 0, -1, 0, 0});
    size_t rcap = 1u << 12;
    while (rcap < (size_t)n * 2) rcap <<= 1;
    std::vector<int32_t> rtab(rcap, -1);  // row hash table -> unique index
    std::vector<uint64_t> rhash;          // hash of every unique row
    rhash.reserve((size_t)n);
    std::vector<int32_t> row;  // kept token ids of the current row
    const char s0 = sep2[0];

    for (int64_t r = 0; r < n; r++) {
        const char *s = b + t->feats[(size_t)r].off;
        const int64_t len = t->feats[(size_t)r].len;
        row.clear();
        int64_t pos = 0;
        while (pos <= len) {
            int64_t nx = -1;
            if (sep2_len == 1) {
                const void *f = pos < len ? memchr(s + pos, s0, (size_t)(len - pos)) : nullptr;
                if (f) nx = (const char *)f - s;
            } else {
                for (int64_t i = pos; i + sep2_len <= len; i++)
                    if (s[i] == s0 && memcmp(s + i, sep2, (size_t)sep2_len) == 0) {
                        nx = i;
                        break;
                    }
            }
            const int64_t tl = (nx < 0 ? len : nx) - pos;
            const char *tk = s + pos;
            if (tl == 0) {
                // empty token: never in the CSR (:208-209); with filtering its verdict still counts (it is
                // "invalid" for most types and printed as such)
                if (filtering) {
                    const Verdict v = cls(tk, 0);
                    if (v == INVALID) t->invalid.push_back(Span{tk - b, 0});
                }
            } else {
                size_t i = bytes_hash(tk, (size_t)tl) & (tcap - 1);
                while (tab[i].used && !(tab[i].len == (uint32_t)tl && memcmp(tab[i].p, tk, (size_t)tl) == 0)) i = (i + 1) & (tcap - 1);
                if (!tab[i].used) {
                    tab[i] = TokSlot{tk, (uint32_t)tl, -1, (int8_t)(filtering ? cls(tk, tl) : KEEP), 1};
                    tcount++;
                }
                TokSlot &sl = tab[i];
                if (sl.verdict == KEEP) {
                    if (sl.id < 0) {
                        sl.id = (int32_t)t->vocab.size();
                        t->vocab.push_back(Span{tk - b, (int32_t)tl});
                    }
                    row.push_back(sl.id);
                } else if (sl.verdict == INVALID) {
                    t->invalid.push_back(Span{tk - b, (int32_t)tl});
                }
                if (tcount * 2 > tcap) {
                    std::vector<TokSlot> nt(tcap * 2, TokSlot{nullptr, 0, -1, 0, 0});
                    for (const TokSlot &o : tab)
                        if (o.used) {
                            size_t j = bytes_hash(o.p, o.len) & (tcap * 2 - 1);
                            while (nt[j].used) j = (j + 1) & (tcap * 2 - 1);
                            nt[j] = o;
                        }
                    tab.swap(nt);
                    tcap *= 2;
                }
            }
            if (nx < 0) break;
            pos = nx + sep2_len;
        }
        // identity of the (filtered) feature string: the kept token sequence when the string is re-joined,
        // the raw bytes when it is passed through untouched
        static const int32_t none = 0;
        const uint64_t h = filtering ? bytes_hash((const char *)(row.empty() ? &none : row.data()), row.size() * sizeof(int32_t))
                                     : bytes_hash(s, (size_t)len);
        size_t i = h & (rcap - 1);
        int32_t u = -1;
        while (rtab[i] >= 0) {
            const int32_t c = rtab[i];
            if (rhash[(size_t)c] == h) {
                bool same;
                if (filtering) {
                    const int32_t cb = t->indptr[(size_t)c], ce = t->indptr[(size_t)c + 1];
                    same = (size_t)(ce - cb) == row.size() && (row.empty() || memcmp(&t->indices[(size_t)cb], row.data(), row.size() * sizeof(int32_t)) == 0);
                } else {
                    const Span o = t->feats[(size_t)t->first_row[(size_t)c]];
                    same = o.len == len && memcmp(b + o.off, s, (size_t)len) == 0;
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
            rhash.push_back(h);
            t->first_row.push_back((int32_t)r);
            t->weight.push_back(0);
            if (t->indices.size() + row.size() > (size_t)INT32_MAX) return bfk_fail(BFK_EARG, "bfk_table_prepare: more than 2^31-1 entries");
            t->indices.insert(t->indices.end(), row.begin(), row.end());
            t->indptr.push_back((int32_t)t->indices.size());
        }
        t->group[(size_t)r] = u;
        t->weight[(size_t)u]++;
    }
    t->n_vocab = (int32_t)t->vocab.size();
    t->prepared = true;
    info->n_rows = n;
    info->n_unique = (int64_t)t->first_row.size();
    info->nnz = (int64_t)t->indices.size();
    info->n_invalid = (int64_t)t->invalid.size();
    info->n_vocab = t->n_vocab;
    info->filtered = filtering ? 1 : 0;
    return BFK_OK;
}

extern "C" const int32_t *bfk_table_group(const bfk_table *t) { return t && t->prepared ? t->group.data() : nullptr; }
extern "C" const int32_t *bfk_table_weight(const bfk_table *t) { return t && t->prepared ? t->weight.data() : nullptr; }
extern "C" const int32_t *bfk_table_indptr(const bfk_table *t) { return t && t->prepared ? t->indptr.data() : nullptr; }
extern "C" const int32_t *bfk_table_indices(const bfk_table *t) { return t && t->prepared ? t->indices.data() : nullptr; }

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
    std::string out;
    out.reserve(t->ids.size() * 24 + 64);
    out += "id\tcluster_id\n";
    char num[16];
    const char *b = t->bytes.data();
    for (size_t r = 0; r < t->ids.size(); r++) {
        const Span id = t->ids[r];
        // csv.QUOTE_MINIMAL: only a field holding the output delimiter needs quotes here (quote, CR and LF
        // never reach a table)
        const bool q = memchr(b + id.off, '\t', (size_t)id.len) != nullptr;
        if (q) out += '"';
        out.append(b + id.off, (size_t)id.len);
        if (q) out += '"';
        out += '\t';
        const int32_t c = cluster_of_unique[(size_t)t->group[r]];
        if (c) {
            if (!remap[(size_t)c]) remap[(size_t)c] = ++next;
            const int m = snprintf(num, sizeof num, "%d", remap[(size_t)c]);
            out.append(num, (size_t)m);
        }
        out += '\n';
    }
    FILE *f = fopen(path, "wb");
    if (!f) return bfk_fail(BFK_EIO, std::string("cannot write ") + path);
    const size_t w = fwrite(out.data(), 1, out.size(), f);
    if (fclose(f) != 0 || w != out.size()) return bfk_fail(BFK_EIO, std::string("short write on ") + path);
    if (n_clusters_out) *n_clusters_out = next;
    return BFK_OK;
}
