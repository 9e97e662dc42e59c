// Snowball-English stemming of token batches (host code, no GPU): the per-token step of
// keywords_preprocess (aidial_rag/keywords_search.py:13-18: `stemmer.stem(t.lower())` with
// nltk.stem.snowball.SnowballStemmer("english")), which the reference runs in pure Python for every
// token of every chunk at index build (bm25_retriever.py:30-39,112) and for every query.
//
// Behaviour follows NLTK's EnglishStemmer (nltk 3.9.1 is the reference's pin, pyproject.toml; pinned here
// against nltk 3.6.5, the copy in this image - tests/golden/snowball_english.json), not the canonical
// Snowball program: NLTK carries the regions R1 / R2 as separate strings that are edited next to the word
// and can fall out of step with it (e.g. R2 becomes "e" after -ational -> -ate when it was shorter than the
// suffix), and later steps test those strings.  The same bookkeeping is kept here so the quirks come out
// the same.  Input tokens must already be lower-cased (Python's str.lower() is the reference's).
#include <cstdint>
#include <cstring>
#include <string>

#include "common.h"
#include "text_common.h"

namespace {

using U = std::u32string;

inline bool vowel(char32_t c) { return c == U'a' || c == U'e' || c == U'i' || c == U'o' || c == U'u' || c == U'y'; }

inline bool ends_n(const U &w, const char *suf, size_t n) {
    if (w.size() < n || w.back() != (char32_t)(unsigned char)suf[n - 1]) return false;  // most suffixes fail here
    for (size_t i = 0; i + 1 < n; ++i)
        if (w[w.size() - n + i] != (char32_t)(unsigned char)suf[i]) return false;
    return true;
}
inline bool ends(const U &w, const char *suf) { return ends_n(w, suf, std::strlen(suf)); }  // literals: strlen folds
struct Suf { const char *s; size_t n; };
#define SUF(x) Suf{x, sizeof(x) - 1}
inline bool starts(const U &w, const char *pre) {
    const size_t n = std::strlen(pre);
    if (w.size() < n) return false;
    for (size_t i = 0; i < n; ++i)
        if (w[i] != (char32_t)(unsigned char)pre[i]) return false;
    return true;
}
inline void chop(U &s, size_t n) { s.resize(s.size() > n ? s.size() - n : 0); }  // s[:-n], clamped like a slice
inline void put(U &s, const char *t) {
    for (; *t; ++t) s.push_back((char32_t)(unsigned char)*t);
}
// original[:-len(old)] + new when the region is at least as long as the old suffix, else `otherwise`
inline void region_replace(U &r, size_t old_len, const char *neu, const char *otherwise) {
    if (r.size() >= old_len) {
        chop(r, old_len);
        put(r, neu);
    } else {
        r.clear();
        put(r, otherwise);
    }
}
inline bool equals(const U &w, const char *s) { return w.size() == std::strlen(s) && ends(w, s); }

struct Special { const char *word, *stem; };
const Special kSpecial[] = {
    {"skis", "ski"}, {"skies", "sky"}, {"dying", "die"}, {"lying", "lie"}, {"tying", "tie"}, {"idly", "idl"},
    {"gently", "gentl"}, {"ugly", "ugli"}, {"early", "earli"}, {"only", "onli"}, {"singly", "singl"}, {"sky", "sky"},
    {"news", "news"}, {"howe", "howe"}, {"atlas", "atlas"}, {"cosmos", "cosmos"}, {"bias", "bias"}, {"andes", "andes"},
    {"inning", "inning"}, {"innings", "inning"}, {"outing", "outing"}, {"outings", "outing"}, {"canning", "canning"},
    {"cannings", "canning"}, {"herring", "herring"}, {"herrings", "herring"}, {"earring", "earring"},
    {"earrings", "earring"}, {"proceed", "proceed"}, {"proceeds", "proceed"}, {"proceeded", "proceed"},
    {"proceeding", "proceed"}, {"exceed", "exceed"}, {"exceeds", "exceed"}, {"exceeded", "exceed"},
    {"exceeding", "exceed"}, {"succeed", "succeed"}, {"succeeds", "succeed"}, {"succeeded", "succeed"},
    {"succeeding", "succeed"},
};

// suffix lists in NLTK's match order (first hit ends the step, whether or not it then applies)
const Suf kStep0[] = {SUF("'s'"), SUF("'s"), SUF("'")};
const Suf kStep1a[] = {SUF("sses"), SUF("ied"), SUF("ies"), SUF("us"), SUF("ss"), SUF("s")};
const Suf kStep1b[] = {SUF("eedly"), SUF("ingly"), SUF("edly"), SUF("eed"), SUF("ing"), SUF("ed")};
const Suf kStep2[] = {SUF("ization"), SUF("ational"), SUF("fulness"), SUF("ousness"), SUF("iveness"), SUF("tional"), SUF("biliti"), SUF("lessli"), SUF("entli"), SUF("ation"), SUF("alism"), SUF("aliti"), SUF("ousli"), SUF("iviti"), SUF("fulli"), SUF("enci"), SUF("anci"), SUF("abli"), SUF("izer"), SUF("ator"), SUF("alli"), SUF("bli"), SUF("ogi"), SUF("li")};
const Suf kStep3[] = {SUF("ational"), SUF("tional"), SUF("alize"), SUF("icate"), SUF("iciti"), SUF("ative"), SUF("ical"), SUF("ness"), SUF("ful")};
const Suf kStep4[] = {SUF("ement"), SUF("ance"), SUF("ence"), SUF("able"), SUF("ible"), SUF("ment"), SUF("ant"), SUF("ent"), SUF("ism"), SUF("ate"), SUF("iti"), SUF("ous"), SUF("ive"), SUF("ize"), SUF("ion"), SUF("al"), SUF("er"), SUF("ic")};
const Suf kDouble[] = {SUF("bb"), SUF("dd"), SUF("ff"), SUF("gg"), SUF("mm"), SUF("nn"), SUF("pp"), SUF("rr"), SUF("tt")};

inline bool is(const char *a, const char *b) { return std::strcmp(a, b) == 0; }

// r1, r2: scratch strings owned by the caller (their capacity is reused from token to token)
void stem(U &word, U &r1, U &r2) {
    if (word.size() <= 2) return;
    if (word.size() <= 10)  // the longest special word
        for (const Special &s : kSpecial)
            if (word[0] == (char32_t)(unsigned char)s.word[0] && equals(word, s.word)) {
                word.clear();
                put(word, s.stem);
                return;
            }
    for (char32_t &c : word)
        if (c == 0x2019 || c == 0x2018 || c == 0x201B) c = U'\'';
    if (!word.empty() && word[0] == U'\'') word.erase(0, 1);
    if (!word.empty() && word[0] == U'y') word[0] = U'Y';
    for (size_t i = 1; i < word.size(); ++i)
        if (vowel(word[i - 1]) && word[i] == U'y') word[i] = U'Y';

    auto region_after_vc = [](const U &s, U &out) {  // after the first non-vowel that follows a vowel
        out.clear();
        for (size_t i = 1; i < s.size(); ++i)
            if (!vowel(s[i]) && vowel(s[i - 1])) {
                out.assign(s, i + 1, U::npos);
                return;
            }
    };
    if (starts(word, "gener") || starts(word, "commun") || starts(word, "arsen")) {
        r1.assign(word, starts(word, "commun") ? 6 : 5, U::npos);
        region_after_vc(r1, r2);
    } else {
        region_after_vc(word, r1);
        region_after_vc(r1, r2);
    }

    // step 0
    for (const Suf &sf : kStep0)
        if (ends_n(word, sf.s, sf.n)) {
            const size_t n = sf.n;
            chop(word, n); chop(r1, n); chop(r2, n);
            break;
        }
    // step 1a
    for (const Suf &sf : kStep1a)
        if (ends_n(word, sf.s, sf.n)) {
            const char *suf = sf.s;
            if (is(suf, "sses")) {
                chop(word, 2); chop(r1, 2); chop(r2, 2);
            } else if (is(suf, "ied") || is(suf, "ies")) {
                const size_t n = word.size() - 3 > 1 ? 2 : 1;
                chop(word, n); chop(r1, n); chop(r2, n);
            } else if (is(suf, "s")) {
                bool found = false;
                for (size_t i = 0; i + 2 < word.size(); ++i) found = found || vowel(word[i]);
                if (found) { chop(word, 1); chop(r1, 1); chop(r2, 1); }
            }
            break;
        }
    // step 1b
    for (const Suf &sf : kStep1b)
        if (ends_n(word, sf.s, sf.n)) {
            const char *suf = sf.s;
            const size_t n = sf.n;
            if (is(suf, "eed") || is(suf, "eedly")) {
                if (ends_n(r1, sf.s, sf.n)) {
                    chop(word, n); put(word, "ee");
                    region_replace(r1, n, "ee", "");
                    region_replace(r2, n, "ee", "");
                }
            } else {
                bool found = false;
                for (size_t i = 0; i + n < word.size(); ++i) found = found || vowel(word[i]);
                if (found) {
                    chop(word, n); chop(r1, n); chop(r2, n);
                    bool dbl = false;
                    for (const Suf &d : kDouble) dbl = dbl || ends_n(word, d.s, d.n);
                    const size_t L = word.size();
                    if (ends(word, "at") || ends(word, "bl") || ends(word, "iz")) {
                        word.push_back(U'e');
                        r1.push_back(U'e');
                        if (word.size() > 5 || r1.size() >= 3) r2.push_back(U'e');
                    } else if (dbl) {
                        chop(word, 1); chop(r1, 1); chop(r2, 1);
                    } else if (r1.empty() &&
                               ((L >= 3 && !vowel(word[L - 1]) && word[L - 1] != U'w' && word[L - 1] != U'x' &&
                                 word[L - 1] != U'Y' && vowel(word[L - 2]) && !vowel(word[L - 3])) ||
                                (L == 2 && vowel(word[0]) && !vowel(word[1])))) {
                        word.push_back(U'e');  // r1 is empty here, and r2 (a piece of r1) with it
                        if (!r2.empty()) r2.push_back(U'e');
                    }
                }
            }
            break;
        }
    // step 1c
    if (word.size() > 2 && (word.back() == U'y' || word.back() == U'Y') && !vowel(word[word.size() - 2])) {
        word.back() = U'i';
        if (!r1.empty()) r1.back() = U'i';
        if (!r2.empty()) r2.back() = U'i';
    }
    // step 2
    for (const Suf &sf : kStep2)
        if (ends_n(word, sf.s, sf.n)) {
            const char *suf = sf.s;
            if (ends_n(r1, sf.s, sf.n)) {
                const size_t n = sf.n;
                auto all3 = [&](size_t k) { chop(word, k); chop(r1, k); chop(r2, k); };
                auto repl = [&](const char *neu, const char *r2_otherwise) {
                    chop(word, n); put(word, neu);
                    region_replace(r1, n, neu, "");
                    region_replace(r2, n, neu, r2_otherwise);
                };
                if (is(suf, "tional")) all3(2);
                else if (is(suf, "enci") || is(suf, "anci") || is(suf, "abli")) {
                    word.back() = U'e';
                    if (!r1.empty()) r1.back() = U'e';
                    if (!r2.empty()) r2.back() = U'e';
                } else if (is(suf, "entli")) all3(2);
                else if (is(suf, "izer") || is(suf, "ization")) repl("ize", "");
                else if (is(suf, "ational") || is(suf, "ation") || is(suf, "ator")) repl("ate", "e");
                else if (is(suf, "alism") || is(suf, "aliti") || is(suf, "alli")) repl("al", "");
                else if (is(suf, "fulness")) all3(4);
                else if (is(suf, "ousli") || is(suf, "ousness")) repl("ous", "");
                else if (is(suf, "iveness") || is(suf, "iviti")) repl("ive", "e");
                else if (is(suf, "biliti") || is(suf, "bli")) repl("ble", "");
                else if (is(suf, "ogi")) { if (word.size() >= 4 && word[word.size() - 4] == U'l') all3(1); }
                else if (is(suf, "fulli") || is(suf, "lessli")) all3(2);
                else if (is(suf, "li")) {
                    const char32_t c = word.size() >= 3 ? word[word.size() - 3] : 0;
                    if (c != 0 && c < 128 && std::strchr("cdeghkmnrt", (int)c) != nullptr) all3(2);
                }
            }
            break;
        }
    // step 3
    for (const Suf &sf : kStep3)
        if (ends_n(word, sf.s, sf.n)) {
            const char *suf = sf.s;
            if (ends_n(r1, sf.s, sf.n)) {
                const size_t n = sf.n;
                auto all3 = [&](size_t k) { chop(word, k); chop(r1, k); chop(r2, k); };
                auto repl = [&](const char *neu) {
                    chop(word, n); put(word, neu);
                    region_replace(r1, n, neu, "");
                    region_replace(r2, n, neu, "");
                };
                if (is(suf, "tional")) all3(2);
                else if (is(suf, "ational")) repl("ate");
                else if (is(suf, "alize")) all3(3);
                else if (is(suf, "icate") || is(suf, "iciti") || is(suf, "ical")) repl("ic");
                else if (is(suf, "ful") || is(suf, "ness")) all3(n);
                else if (is(suf, "ative") && ends_n(r2, sf.s, sf.n)) all3(5);
            }
            break;
        }
    // step 4
    for (const Suf &sf : kStep4)
        if (ends_n(word, sf.s, sf.n)) {
            const char *suf = sf.s;
            if (ends_n(r2, sf.s, sf.n)) {
                const size_t n = sf.n;
                if (is(suf, "ion")) {
                    const char32_t c = word.size() >= 4 ? word[word.size() - 4] : 0;  // (Python would raise on a 3-letter word)
                    if (c == U's' || c == U't') { chop(word, 3); chop(r1, 3); chop(r2, 3); }
                } else {
                    chop(word, n); chop(r1, n); chop(r2, n);
                }
            }
            break;
        }
    // step 5
    const size_t L = word.size();
    if (ends(r2, "l") && L >= 2 && word[L - 2] == U'l') {
        chop(word, 1);
    } else if (ends(r2, "e")) {
        chop(word, 1);
    } else if (ends(r1, "e")) {
        if (L >= 4 && (vowel(word[L - 2]) || word[L - 2] == U'w' || word[L - 2] == U'x' || word[L - 2] == U'Y' ||
                       !vowel(word[L - 3]) || vowel(word[L - 4])))
            chop(word, 1);
    }
    for (char32_t &c : word)
        if (c == U'Y') c = U'y';
}

// UTF-8 <-> code points; malformed bytes pass through as single code points >= 0x110000 and come back as they were
void decode(const unsigned char *p, size_t n, U &out) {
    out.clear();
    for (size_t i = 0; i < n;) {
        const unsigned char c = p[i];
        int len = c < 0x80 ? 1 : (c >> 5) == 0x6 ? 2 : (c >> 4) == 0xE ? 3 : (c >> 3) == 0x1E ? 4 : 0;
        bool ok = len > 0 && i + len <= n;
        for (int k = 1; ok && k < len; ++k) ok = (p[i + k] & 0xC0) == 0x80;
        if (!ok) {
            out.push_back(0x110000u + c);
            ++i;
            continue;
        }
        char32_t v = len == 1 ? c : len == 2 ? (c & 0x1F) : len == 3 ? (c & 0x0F) : (c & 0x07);
        for (int k = 1; k < len; ++k) v = (v << 6) | (p[i + k] & 0x3F);
        out.push_back(v);
        i += len;
    }
}
size_t encode_n(const char32_t *w, size_t n, unsigned char *dst) {
    size_t o = 0;
    for (size_t i = 0; i < n; ++i) {
        const char32_t v = w[i];
        if (v >= 0x110000u) dst[o++] = (unsigned char)(v - 0x110000u);
        else if (v < 0x80) dst[o++] = (unsigned char)v;
        else if (v < 0x800) { dst[o++] = 0xC0 | (v >> 6); dst[o++] = 0x80 | (v & 0x3F); }
        else if (v < 0x10000) { dst[o++] = 0xE0 | (v >> 12); dst[o++] = 0x80 | ((v >> 6) & 0x3F); dst[o++] = 0x80 | (v & 0x3F); }
        else { dst[o++] = 0xF0 | (v >> 18); dst[o++] = 0x80 | ((v >> 12) & 0x3F); dst[o++] = 0x80 | ((v >> 6) & 0x3F); dst[o++] = 0x80 | (v & 0x3F); }
    }
    return o;
}
size_t encode(const U &w, unsigned char *dst) { return encode_n(w.data(), w.size(), dst); }

}  // namespace

namespace mir {
namespace text {
void snowball_english(U &word, U &r1, U &r2) { stem(word, r1, r2); }
void utf8_decode(const unsigned char *p, size_t n, U &out) { decode(p, n, out); }
size_t utf8_encode(const char32_t *w, size_t n, unsigned char *dst) { return encode_n(w, n, dst); }
}  // namespace text
}  // namespace mir

extern "C" {

// tokens: n_bytes of UTF-8, tokens separated by `sep` (a byte that occurs in no token; no trailing separator).
// out: at least n_bytes bytes; receives the stems, same separator, same order.  A stem is never longer than
// its token.  *out_bytes = bytes written.
int32_t mir_stem_english(const char *tokens, int64_t n_bytes, char sep, char *out, int64_t *out_bytes) {
    MIR_REQUIRE(n_bytes >= 0 && out_bytes != nullptr, "bad argument");
    MIR_REQUIRE(n_bytes == 0 || (tokens != nullptr && out != nullptr), "NULL buffer");
    const unsigned char *p = reinterpret_cast<const unsigned char *>(tokens);
    unsigned char *dst = reinterpret_cast<unsigned char *>(out);
    int64_t o = 0;
    U w, r1, r2;
    w.reserve(64); r1.reserve(64); r2.reserve(64);
    for (int64_t i = 0; i <= n_bytes;) {
        int64_t j = i;
        while (j < n_bytes && tokens[j] != sep) ++j;
        bool ascii_short = j - i <= 2;
        if (ascii_short) {  // at most two bytes = at most two code points: returned as it is
            for (int64_t k = i; k < j; ++k) dst[o++] = p[k];
        } else {
            decode(p + i, (size_t)(j - i), w);
            stem(w, r1, r2);
            o += (int64_t)encode(w, dst + o);
        }
        if (j < n_bytes) dst[o++] = (unsigned char)sep;
        i = j + 1;
        if (j >= n_bytes) break;
    }
    *out_bytes = o;
    return MIR_OK;
}

}  // extern "C"
