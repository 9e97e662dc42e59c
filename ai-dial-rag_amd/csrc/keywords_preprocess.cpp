// keywords_preprocess for batches of chunk texts, on all host cores (host code, no GPU):
//   aidial_rag/keywords_search.py:13-18   word_tokenize -> drop stopwords (compared BEFORE lower-casing) -> Snowball stem
// which the reference runs per chunk in pure Python next to the encoder at index build
// (retrievers/bm25_retriever.py:30-39,106-114; documents.py:188-198) and per query.
//
// What is restated here, and its pin (ai-dial-rag_amd/keywords_search.py holds the same restatement in Python; the two are
// compared text by text in tests/test_keywords_native.py):
//   * NLTKWordTokenizer (nltk/tokenize/destructive.py, nltk 3.6.5): an ORDERED list of regular-expression substitutions
//     and a whitespace split.  Each substitution is one left-to-right pass below, written to match Python's `re`
//     semantics for that pattern (leftmost, non-overlapping, greedy; \w \d \s and IGNORECASE over Unicode through the
//     generated unicode_tables.h).  Pinned on tests/golden/treebank_tokenize.json.gz (24 744 sentences).
//   * the sentence splitter in front of it: the SAME rule-based approximation of Punkt as the Python mirror (Punkt is a
//     trained model whose parameters are absent here) - unpinned, DESIGN.md 7;
//   * the stopword list: the mirror's 179 words - unpinned;
//   * str.lower(): unicode_tables.h + the two context rules (U+0130, final sigma);
//   * the Snowball-English stemmer: stem_english.cpp, pinned on tests/golden/snowball_english.json.gz.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "common.h"
#include "text_common.h"
#include "unicode_tables.h"

namespace mir {
namespace text {
namespace {

inline bool in_ranges(const uint32_t (*t)[2], int n, uint32_t c) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (c < t[mid][0]) hi = mid - 1;
        else if (c > t[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}
inline bool is_space(char32_t c) {  // str.isspace == re \s
    if (c < 0x80) return c == U' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f);
    return in_ranges(kSpace, kSpace_N, c);
}
inline bool is_word(char32_t c) {  // re \w
    if (c < 0x80) return (c >= U'a' && c <= U'z') || (c >= U'A' && c <= U'Z') || (c >= U'0' && c <= U'9') || c == U'_';
    return in_ranges(kWord, kWord_N, c);
}
inline bool is_decimal(char32_t c) {  // re \d
    if (c < 0x80) return c >= U'0' && c <= U'9';
    return in_ranges(kDecimal, kDecimal_N, c);
}
inline bool is_alpha(char32_t c) {
    if (c < 0x80) return (c >= U'a' && c <= U'z') || (c >= U'A' && c <= U'Z');
    return in_ranges(kAlpha, kAlpha_N, c);
}
// the ASCII letter a code point matches under re.IGNORECASE, lower-cased (the code point itself when there is none)
inline char32_t fold(char32_t c) {
    if (c < 0x80) return (c >= U'A' && c <= U'Z') ? c + 32 : c;
    for (int i = 0; i < kFoldAscii_N; ++i)
        if (kFoldAscii[i][0] == c) return kFoldAscii[i][1];
    return c;
}
inline char32_t lower1(char32_t c) {  // context-free single-code-point part of str.lower()
    if (c < 0x80) return (c >= U'A' && c <= U'Z') ? c + 32 : c;
    int lo = 0, hi = kLower_N - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (c < kLower[mid][0]) hi = mid - 1;
        else if (c > kLower[mid][0]) lo = mid + 1;
        else return kLower[mid][1];
    }
    return c;
}
// str.lower(): per code point, U+0130 -> "i" + U+0307, and U+03A3 by the final-sigma rule (CPython unicodeobject.c, handle_capital_sigma)
void lower_into(const char32_t *s, size_t n, U &out) {
    out.clear();
    for (size_t i = 0; i < n; ++i) {
        const char32_t c = s[i];
        if (c < 0x80) { out.push_back((c >= U'A' && c <= U'Z') ? c + 32 : c); continue; }
        if (c == 0x130) { out.push_back(U'i'); out.push_back(0x307); continue; }
        if (c == 0x3A3) {
            ptrdiff_t j = (ptrdiff_t)i - 1;
            while (j >= 0 && in_ranges(kCaseIgnorable, kCaseIgnorable_N, s[j])) --j;
            bool fin = j >= 0 && in_ranges(kCasedNotIgnorable, kCasedNotIgnorable_N, s[j]);
            if (fin) {
                size_t k = i + 1;
                while (k < n && in_ranges(kCaseIgnorable, kCaseIgnorable_N, s[k])) ++k;
                fin = k == n || !in_ranges(kCasedNotIgnorable, kCasedNotIgnorable_N, s[k]);
            }
            out.push_back(fin ? 0x3C2 : 0x3C3);
            continue;
        }
        out.push_back(lower1(c));
    }
}

struct UHash {
    size_t operator()(const U &s) const {
        uint64_t h = 1469598103934665603ull;
        for (char32_t c : s) { h ^= c; h *= 1099511628211ull; }
        return (size_t)h;
    }
};
const char *const kStopwords =
    "i me my myself we our ours ourselves you you're you've you'll you'd your yours yourself yourselves he him his "
    "himself she she's her hers herself it it's its itself they them their theirs themselves what which who whom this that that'll these those am "
    "is are was were be been being have has had having do does did doing a an the and but if or because as until while of at by for with about "
    "against between into through during before after above below to from up down in out on off over under again further then once here there when "
    "where why how all any both each few more most other some such no nor not only own same so than too very s t can will just don don't should "
    "should've now d ll m o re ve y ain aren aren't couldn couldn't didn didn't doesn doesn't hadn hadn't hasn hasn't haven haven't isn isn't ma "
    "mightn mightn't mustn mustn't needn needn't shan shan't shouldn shouldn't wasn wasn't weren weren't won won't wouldn wouldn't";
const char *const kAbbreviations =
    "mr mrs ms dr prof sr jr st vs etc inc ltd co corp no fig figs eq vol pp ed eds e.g i.e cf al approx dept est "
    "jan feb mar apr jun jul aug sep sept oct nov dec mon tue wed thu fri sat sun u.s u.k a.m p.m ph.d gen col lt sgt capt rev hon";
std::unordered_set<U, UHash> word_set(const char *list) {
    std::unordered_set<U, UHash> s;
    U w;
    for (const char *p = list;; ++p) {
        if (*p == ' ' || *p == 0) {
            if (!w.empty()) s.insert(w);
            w.clear();
            if (*p == 0) break;
        } else w.push_back((char32_t)(unsigned char)*p);
    }
    return s;
}
const std::unordered_set<U, UHash> &stopwords() { static const auto s = word_set(kStopwords); return s; }
const std::unordered_set<U, UHash> &abbreviations() { static const auto s = word_set(kAbbreviations); return s; }

// ---------------------------------------------------------------------------------------------------------------
// NLTKWordTokenizer.tokenize: every function is `re.sub` of one rule, old string -> new string
// ---------------------------------------------------------------------------------------------------------------
inline void sp(U &o, char32_t c) { o.push_back(U' '); o.push_back(c); o.push_back(U' '); }

struct Tokenizer {
    U a, b;  // ping-pong
    bool has[128];
    bool non_ascii;
    void scan() {
        std::memset(has, 0, sizeof(has));
        non_ascii = false;
        for (char32_t c : a) {
            if (c < 0x80) has[c] = true;
            else non_ascii = true;
        }
    }
    void swap() { a.swap(b); }
    // Lazy copy: a rule's pass writes nothing until its first match (most rules match nowhere in most sentences, and a
    // sentence goes through ~35 passes): `hit(i)` starts the output with a[0, i); `keep(c)` appends once started.
    bool started;
    void hit(size_t i) {
        if (!started) { b.assign(a, 0, i); started = true; }
    }
    void keep(char32_t c) {
        if (started) b.push_back(c);
    }
    void done() {
        if (started) swap();
    }

    // STARTING_QUOTES
    void s1() {  // ([«“‘„]|[`]+) -> " \1 "
        if (!non_ascii && !has[(int)'`']) return;
        b.clear();
        const size_t n = a.size();
        for (size_t i = 0; i < n;) {
            const char32_t c = a[i];
            if (c == 0xAB || c == 0x201C || c == 0x2018 || c == 0x201E) { sp(b, c); ++i; }
            else if (c == U'`') {
                size_t j = i;
                while (j < n && a[j] == U'`') ++j;
                b.push_back(U' '); b.append(a, i, j - i); b.push_back(U' ');
                i = j;
            } else { b.push_back(c); ++i; }
        }
        swap();
    }
    void s2() {  // ^" -> ``
        if (a.empty() || a[0] != U'"') return;
        b.assign(U"``");
        b.append(a, 1, U::npos);
        has[(int)'`'] = true;
        swap();
    }
    void s3() {  // (``) -> " \1 "
        if (!has[(int)'`']) return;
        started = false;
        const size_t n = a.size();
        for (size_t i = 0; i < n;) {
            if (a[i] == U'`' && i + 1 < n && a[i + 1] == U'`') { hit(i); b.append(U" `` "); i += 2; }
            else keep(a[i++]);
        }
        done();
    }
    void s4() {  // ([ \(\[{<])(\"|'{2}) -> \1 ``<space>
        if (!has[(int)'"'] && !has[(int)'\'']) return;
        started = false;
        const size_t n = a.size();
        for (size_t i = 0; i < n;) {
            const char32_t c = a[i];
            if ((c == U' ' || c == U'(' || c == U'[' || c == U'{' || c == U'<') && i + 1 < n) {
                size_t len = 0;
                if (a[i + 1] == U'"') len = 2;
                else if (a[i + 1] == U'\'' && i + 2 < n && a[i + 2] == U'\'') len = 3;
                if (len) {
                    hit(i);
                    b.push_back(c); b.append(U" `` ");
                    has[(int)'`'] = true;
                    i += len;
                    continue;
                }
            }
            keep(c);
            ++i;
        }
        done();
    }
    void s5() {  // (?i)(')(?!re|ve|ll|m|t|s|d|n)(\w)\b -> "\1 \2"
        if (!has[(int)'\'']) return;
        started = false;
        const size_t n = a.size();
        for (size_t i = 0; i < n;) {
            if (a[i] == U'\'' && i + 1 < n && is_word(a[i + 1]) && (i + 2 == n || !is_word(a[i + 2]))) {
                // (the two-letter alternatives of the lookahead cannot apply: the next character is followed by a boundary)
                const char32_t f = fold(a[i + 1]);
                if (!(f == U'm' || f == U't' || f == U's' || f == U'd' || f == U'n')) {
                    hit(i);
                    b.push_back(U'\''); b.push_back(U' '); b.push_back(a[i + 1]);
                    i += 2;
                    continue;
                }
            }
            keep(a[i++]);
        }
        done();
    }
    // PUNCTUATION
    // ([^\.])(\.)(<set>*)\s*$ -> "\1 \2 \3 " (SPACED) or "\1 \2\3 ": at most one match (it is anchored at the end)
    template <bool UNI>
    static bool in_tail_set(char32_t c) {
        if (c == U']' || c == U')' || c == U'}' || c == U'>' || c == U'"' || c == U'\'') return true;
        return UNI && (c == 0xBB || c == 0x201D || c == 0x2019 || c == U' ');
    }
    template <bool UNI>
    void final_period() {
        if (!has[(int)'.']) return;
        const size_t n = a.size();
        for (size_t i = 0; i + 1 < n; ++i) {
            if (a[i] == U'.' || a[i + 1] != U'.') continue;
            size_t j = i + 2;
            while (j < n && in_tail_set<UNI>(a[j])) ++j;
            const size_t g3 = j;
            while (j < n && is_space(a[j])) ++j;
            if (j != n) continue;
            b.assign(a, 0, i + 1);
            b.push_back(U' '); b.push_back(U'.');
            if (UNI) b.push_back(U' ');
            b.append(a, i + 2, g3 - (i + 2));
            b.push_back(U' ');
            swap();
            return;
        }
    }
    void p2() {  // ([:,])([^\d]) -> " \1 \2"
        if (!has[(int)':'] && !has[(int)',']) return;
        started = false;
        const size_t n = a.size();
        for (size_t i = 0; i < n;) {
            const char32_t c = a[i];
            if ((c == U':' || c == U',') && i + 1 < n && !is_decimal(a[i + 1])) {
                hit(i);
                b.push_back(U' '); b.push_back(c); b.push_back(U' '); b.push_back(a[i + 1]);
                i += 2;
            } else { keep(c); ++i; }
        }
        done();
    }
    void p3() {  // ([:,])$ -> " \1 "   ($: at the end, or before a newline that ends the string)
        const size_t n = a.size();
        if (n == 0) return;
        size_t i = n;
        if (a[n - 1] == U':' || a[n - 1] == U',') i = n - 1;
        else if (n >= 2 && a[n - 1] == U'\n' && (a[n - 2] == U':' || a[n - 2] == U',')) i = n - 2;
        if (i == n) return;
        b.assign(a, 0, i);
        sp(b, a[i]);
        b.append(a, i + 1, U::npos);
        swap();
    }
    void p4() {  // \.{2,} -> " \g<0> "
        if (!has[(int)'.']) return;
        started = false;
        const size_t n = a.size();
        for (size_t i = 0; i < n;) {
            if (a[i] == U'.' && i + 1 < n && a[i + 1] == U'.') {
                size_t j = i;
                while (j < n && a[j] == U'.') ++j;
                hit(i);
                b.push_back(U' '); b.append(a, i, j - i); b.push_back(U' ');
                i = j;
            } else keep(a[i++]);
        }
        done();
    }
    void pad_chars(const char *set) {  // [set] -> " \g<0> "
        bool any = false;
        for (const char *p = set; *p; ++p) any |= has[(int)*p];
        if (!any) return;
        b.clear();
        for (char32_t c : a) {
            if (c < 0x80 && c != 0 && std::strchr(set, (int)c)) sp(b, c);
            else b.push_back(c);
        }
        swap();
    }
    void p8() {  // ([^'])' -> "\1 ' "   (pattern: a non-apostrophe, an apostrophe, a space)
        if (!has[(int)'\'']) return;
        started = false;
        const size_t n = a.size();
        for (size_t i = 0; i < n;) {
            if (a[i] != U'\'' && i + 2 < n && a[i + 1] == U'\'' && a[i + 2] == U' ') {
                hit(i);
                b.push_back(a[i]); b.append(U" ' ");
                i += 3;
            } else keep(a[i++]);
        }
        done();
    }
    void double_dashes() {  // -- -> " -- "
        if (!has[(int)'-']) return;
        started = false;
        const size_t n = a.size();
        for (size_t i = 0; i < n;) {
            if (a[i] == U'-' && i + 1 < n && a[i + 1] == U'-') { hit(i); b.append(U" -- "); i += 2; }
            else keep(a[i++]);
        }
        done();
    }
    // ENDING_QUOTES
    void e1() {  // ([»”’]) -> " \1 "
        if (!non_ascii) return;
        b.clear();
        for (char32_t c : a) {
            if (c == 0xBB || c == 0x201D || c == 0x2019) sp(b, c);
            else b.push_back(c);
        }
        swap();
    }
    void e2() {  // " -> " '' "
        if (!has[(int)'"']) return;
        b.clear();
        for (char32_t c : a) {
            if (c == U'"') b.append(U" '' ");
            else b.push_back(c);
        }
        has[(int)'\''] = true;
        swap();
    }
    void e3() {  // (\S)('') -> "\1 \2 "
        if (!has[(int)'\'']) return;
        started = false;
        const size_t n = a.size();
        for (size_t i = 0; i < n;) {
            if (i + 2 < n && a[i + 1] == U'\'' && a[i + 2] == U'\'' && !is_space(a[i])) {
                hit(i);
                b.push_back(a[i]); b.append(U" '' ");
                i += 3;
            } else keep(a[i++]);
        }
        done();
    }
    void e4() {  // ([^' ])('[sS]|'[mM]|'[dD]|') -> "\1 \2 "   (the pattern ends with a space)
        if (!has[(int)'\'']) return;
        started = false;
        const size_t n = a.size();
        for (size_t i = 0; i < n;) {
            if (i + 1 < n && a[i + 1] == U'\'' && a[i] != U'\'' && a[i] != U' ') {
                size_t g = 0;
                if (i + 3 < n && a[i + 3] == U' ') {
                    const char32_t c = a[i + 2];
                    if (c == U's' || c == U'S' || c == U'm' || c == U'M' || c == U'd' || c == U'D') g = 2;
                }
                if (!g && i + 2 < n && a[i + 2] == U' ') g = 1;
                if (g) {
                    hit(i);
                    b.push_back(a[i]); b.push_back(U' '); b.append(a, i + 1, g); b.push_back(U' ');
                    i += 1 + g + 1;
                    continue;
                }
            }
            keep(a[i++]);
        }
        done();
    }
    void e5() {  // ([^' ])('ll|'LL|'re|'RE|'ve|'VE|n't|N'T) -> "\1 \2 "   (the pattern ends with a space)
        if (!has[(int)'\'']) return;
        started = false;
        const size_t n = a.size();
        for (size_t i = 0; i < n;) {
            if (i + 4 < n && a[i + 4] == U' ' && (a[i + 1] == U'\'' || a[i + 2] == U'\'') && a[i] != U'\'' && a[i] != U' ') {
                const char32_t x = a[i + 1], y = a[i + 2], z = a[i + 3];
                const bool m = (x == U'\'' && ((y == U'l' && z == U'l') || (y == U'L' && z == U'L') || (y == U'r' && z == U'e') ||
                                               (y == U'R' && z == U'E') || (y == U'v' && z == U'e') || (y == U'V' && z == U'E'))) ||
                               (y == U'\'' && ((x == U'n' && z == U't') || (x == U'N' && z == U'T')));
                if (m) {
                    hit(i);
                    b.push_back(a[i]); b.push_back(U' '); b.append(a, i + 1, 3); b.push_back(U' ');
                    i += 5;
                    continue;
                }
            }
            keep(a[i++]);
        }
        done();
    }
    // CONTRACTIONS2 / 3: (?i) <lead> (g1)(g2) <trail> -> " \1 \2 "
    bool boundary(size_t pos) const {  // \b between a[pos - 1] and a[pos]
        const bool l = pos > 0 && is_word(a[pos - 1]), r = pos < a.size() && is_word(a[pos]);
        return l != r;
    }
    bool lit_ci(size_t pos, const char *lit, size_t len) const {
        if (pos + len > a.size()) return false;
        for (size_t k = 0; k < len; ++k)
            if (fold(a[pos + k]) != (char32_t)(unsigned char)lit[k]) return false;
        return true;
    }
    // lead: 'b' = \b, ' ' = a literal space (consumed);  trail: 'b' = \b, 's' = one \s character (consumed)
    void contraction(const char *lit, size_t n1, size_t n2, char lead, char trail) {
        const size_t n = a.size(), len = n1 + n2;
        if (n < len) return;
        // quick reject on the first letter (either case; the non-ASCII folds are letters i, k, s)
        const char32_t c0 = (char32_t)(unsigned char)lit[0];
        bool maybe = non_ascii;
        if (!maybe) maybe = c0 < 0x80 && (has[c0] || (c0 >= U'a' && c0 <= U'z' && has[c0 - 32]));
        if (!maybe) return;
        started = false;
        for (size_t i = 0; i < n;) {
            size_t st = i;
            bool ok = true;
            if (lead == ' ') { ok = a[i] == U' '; st = i + 1; }
            if (ok && lit_ci(st, lit, len)) {
                if (lead == 'b') ok = boundary(st);
                size_t end = st + len;
                if (ok && trail == 'b') ok = boundary(end);
                if (ok && trail == 's') { ok = end < n && is_space(a[end]); ++end; }
                if (ok) {
                    hit(i);
                    b.push_back(U' '); b.append(a, st, n1); b.push_back(U' '); b.append(a, st + n1, n2); b.push_back(U' ');
                    i = end;
                    continue;
                }
            }
            keep(a[i++]);
        }
        done();
    }

    // One pass in front of the ten contraction rules: can ANY of them match?  (A rule changes the text only where it matches,
    // so if none of the ten literals occurs now - boundaries aside - none occurs after any of the rules either.)
    bool any_contraction_literal() const {
        static const char *const lits[] = {"cannot", "d'ye", "gimme", "gonna", "gotta", "lemme", "more'n", "wanna", "'tis", "'twas"};
        const size_t n = a.size();
        for (size_t i = 0; i + 4 <= n; ++i) {
            const char32_t f = fold(a[i]);
            if (!(f == U'c' || f == U'd' || f == U'g' || f == U'l' || f == U'm' || f == U'w' || f == U'\'')) continue;
            for (const char *l : lits)
                if ((char32_t)(unsigned char)l[0] == f && lit_ci(i, l, std::strlen(l))) return true;
        }
        return false;
    }

    // NLTKWordTokenizer().tokenize(sentence), tokens appended to `out` as (offset, length) into `store`
    void tokenize(const char32_t *sent, size_t n, U &store, std::vector<std::pair<uint32_t, uint32_t>> &out) {
        a.assign(sent, n);
        scan();
        s1(); s2(); s3(); s4(); s5();
        final_period<true>();
        p2(); p3(); p4();
        pad_chars(";@#$%&");
        final_period<false>();
        pad_chars("?!");
        p8();
        pad_chars("*");
        pad_chars("][(){}<>");
        double_dashes();
        b.assign(1, U' '); b.append(a); b.push_back(U' ');
        swap();
        has[(int)' '] = true;
        e1(); e2(); e3(); e4(); e5();
        if (any_contraction_literal()) {
            contraction("cannot", 3, 3, 'b', 'b');
            contraction("d'ye", 1, 3, 'b', 'b');
            contraction("gimme", 3, 2, 'b', 'b');
            contraction("gonna", 3, 2, 'b', 'b');
            contraction("gotta", 3, 2, 'b', 'b');
            contraction("lemme", 3, 2, 'b', 'b');
            contraction("more'n", 4, 2, 'b', 'b');
            contraction("wanna", 3, 2, 'b', 's');
            contraction("'tis", 2, 2, ' ', 'b');
            contraction("'twas", 2, 3, ' ', 'b');
        }
        // str.split()
        const size_t m = a.size();
        for (size_t i = 0; i < m;) {
            while (i < m && is_space(a[i])) ++i;
            size_t j = i;
            while (j < m && !is_space(a[j])) ++j;
            if (j > i) {
                out.emplace_back((uint32_t)store.size(), (uint32_t)(j - i));
                store.append(a, i, j - i);
            }
            i = j;
        }
    }
};

// split_sentences of the Python mirror (an APPROXIMATION of Punkt): (begin, end) code-point ranges of `t`
//   _SENT_END = ([.?!]+)(["'”’)\]]*)(\s+)(?=["'“‘(\[]*[A-Z0-9])
void split_sentences(const U &t, std::vector<std::pair<size_t, size_t>> &out, U &scratch, U &scratch2) {
    out.clear();
    const size_t n = t.size();
    auto closer = [](char32_t c) { return c == U'"' || c == U'\'' || c == 0x201D || c == 0x2019 || c == U')' || c == U']'; };
    auto opener = [](char32_t c) { return c == U'"' || c == U'\'' || c == 0x201C || c == 0x2018 || c == U'(' || c == U'['; };
    auto blank = [&](size_t b0, size_t e0) {
        for (size_t i = b0; i < e0; ++i)
            if (!is_space(t[i])) return false;
        return true;
    };
    size_t start = 0;
    for (size_t i = 0; i < n;) {
        const char32_t c = t[i];
        if (!(c == U'.' || c == U'?' || c == U'!')) { ++i; continue; }
        size_t j = i;
        while (j < n && (t[j] == U'.' || t[j] == U'?' || t[j] == U'!')) ++j;
        size_t k = j;
        while (k < n && closer(t[k])) ++k;
        size_t l = k;
        while (l < n && is_space(t[l])) ++l;
        size_t m = l;
        while (m < n && opener(t[m])) ++m;
        const bool ok = l > k && m < n && ((t[m] >= U'A' && t[m] <= U'Z') || (t[m] >= U'0' && t[m] <= U'9'));
        if (!ok) { i = j; continue; }  // (a match attempt from inside the run meets the same continuation)
        // a match [i, l): group 1 = [i, j), group 2 = [j, k)
        bool skip = false;
        if (j - i == 1 && c == U'.') {
            // last = (text[start:i].rsplit(None, 1)[-1] or "").lower().strip("\"'([")
            size_t e = i;
            while (e > start && is_space(t[e - 1])) --e;
            size_t b0 = e;
            while (b0 > start && !is_space(t[b0 - 1])) --b0;
            lower_into(t.data() + b0, e - b0, scratch);
            size_t x = 0, y = scratch.size();
            auto strip = [](char32_t ch) { return ch == U'"' || ch == U'\'' || ch == U'(' || ch == U'['; };
            while (x < y && strip(scratch[x])) ++x;
            while (y > x && strip(scratch[y - 1])) --y;
            scratch2.assign(scratch, x, y - x);
            if (abbreviations().count(scratch2) || (scratch2.size() == 1 && is_alpha(scratch2[0]))) skip = true;
        }
        if (!skip) {
            if (!blank(start, k)) out.emplace_back(start, k);
            start = l;
        }
        i = l;
    }
    if (!blank(start, n)) out.emplace_back(start, n);
}

struct Worker {
    Tokenizer tk;
    U text, store, low, r1, r2, s1, s2;
    std::vector<std::pair<size_t, size_t>> sents;
    std::vector<std::pair<uint32_t, uint32_t>> toks;
    std::vector<unsigned char> enc;

    // keywords_preprocess(text): stems appended to `bytes` (each followed by a NUL); returns the token count or -1 (NUL in the text)
    int64_t run(const unsigned char *p, size_t n, std::string &bytes) {
        if (std::memchr(p, 0, n) != nullptr) return -1;
        utf8_decode(p, n, text);
        split_sentences(text, sents, s1, s2);
        store.clear();
        toks.clear();
        for (const auto &se : sents) tk.tokenize(text.data() + se.first, se.second - se.first, store, toks);
        int64_t count = 0;
        const auto &stop = stopwords();
        for (const auto &t : toks) {
            const char32_t *w = store.data() + t.first;
            bool ascii_lower_or_other = true;  // stopwords are lower-case ASCII words of <= 10 characters
            if (t.second <= 10) {
                for (uint32_t i = 0; i < t.second && ascii_lower_or_other; ++i)
                    ascii_lower_or_other = (w[i] >= U'a' && w[i] <= U'z') || w[i] == U'\'';
                if (ascii_lower_or_other) {
                    s1.assign(w, t.second);
                    if (stop.count(s1)) continue;
                }
            }
            lower_into(w, t.second, low);
            snowball_english(low, r1, r2);
            if (enc.size() < low.size() * 4 + 1) enc.resize(low.size() * 4 + 1);
            const size_t nb = utf8_encode(low.data(), low.size(), enc.data());
            bytes.append(reinterpret_cast<const char *>(enc.data()), nb);
            bytes.push_back('\0');
            ++count;
        }
        return count;
    }
};

}  // namespace
}  // namespace text
}  // namespace mir

struct mir_kwp_result {
    std::string bytes;             // every token's UTF-8 bytes followed by a NUL, text after text
    std::vector<int32_t> counts;   // tokens per text
    std::vector<int64_t> ends;     // bytes[0, ends[i]) = the tokens of texts 0..i
    int64_t n_tokens = 0;
    // mir_kwp_result_dedupe: the batch's distinct tokens in order of first appearance, and every token as an index into them
    std::string uniq;              // every distinct token followed by a NUL
    std::vector<int32_t> ids;      // [n_tokens]
    int32_t n_unique = 0;
    bool deduped = false;
};

extern "C" {

// keywords_preprocess (aidial_rag/keywords_search.py:13-18) of n_texts UTF-8 texts back to back (text i =
// [offsets[i], offsets[i + 1])) on n_threads host threads (<= 0: all cores).  mode: 0 = keywords_preprocess;
// 1 = word_tokenize only (sentence split + Treebank rules; no stopword filter, no stemming); 2 = the Treebank rules alone
// (the text is ONE sentence: NLTKWordTokenizer().tokenize) - for the tests that pin the tokenizer.  A text that contains a NUL byte is refused (MIR_ERR_INVALID), as stem_tokens refuses such tokens.
int32_t mir_keywords_preprocess(const char *texts, const int64_t *offsets, int32_t n_texts, int32_t n_threads, int32_t mode,
                                mir_kwp_result **out) {
    using namespace mir::text;
    MIR_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    MIR_REQUIRE(n_texts >= 0 && (n_texts == 0 || (texts != nullptr && offsets != nullptr)), "bad argument");
    MIR_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (keywords_preprocess), 1 (word_tokenize) or 2 (Treebank rules alone)");
    for (int32_t i = 0; i < n_texts; ++i) MIR_REQUIRE(offsets[i + 1] >= offsets[i], "offsets must not decrease (text %d)", i);
    mir_kwp_result *res = new (std::nothrow) mir_kwp_result();
    MIR_REQUIRE(res != nullptr, "out of host memory");
    res->counts.assign((size_t)n_texts, 0);
    res->ends.assign((size_t)n_texts, 0);
    (void)stopwords();
    (void)abbreviations();  // built before the threads start
    constexpr int32_t kGrain = 32;  // texts per work item
    const int32_t n_items = (n_texts + kGrain - 1) / kGrain;
    // n_threads <= 0: the host's cores, but no more than 16 (MIR_HOST_THREADS overrides): the reported core count of a shared
    // GPU host (256 on the bench box) says nothing about this process's share, and the index build runs this beside the
    // encoder's host threads (documents.py:188-198) - 256 threads per call cut the COMBINED rate of the two builders to a third
    int hw = (int)std::thread::hardware_concurrency();
    if (hw <= 0) hw = 1;
    static const int cap = [] {
        const char *e = getenv("MIR_HOST_THREADS");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 16;
    }();
    int nt = n_threads > 0 ? n_threads : std::min(hw, cap);
    nt = std::max(1, std::min(nt, std::max(1, (int)n_items)));
    std::vector<std::string> parts((size_t)n_items);
    std::atomic<int32_t> next{0};
    std::atomic<int32_t> bad{-1};
    auto work = [&]() {
        Worker wk;
        for (;;) {
            const int32_t it = next.fetch_add(1);
            if (it >= n_items) break;
            std::string &dst = parts[(size_t)it];
            const int32_t t0 = it * kGrain, t1 = std::min(n_texts, t0 + kGrain);
            dst.reserve((size_t)(offsets[t1] - offsets[t0]) + 64);
            for (int32_t t = t0; t < t1; ++t) {
                const unsigned char *p = reinterpret_cast<const unsigned char *>(texts) + offsets[t];
                const size_t n = (size_t)(offsets[t + 1] - offsets[t]);
                int64_t c;
                if (mode == 0) c = wk.run(p, n, dst);
                else {
                    if (std::memchr(p, 0, n) != nullptr) c = -1;
                    else {
                        utf8_decode(p, n, wk.text);
                        if (mode == 1) split_sentences(wk.text, wk.sents, wk.s1, wk.s2);
                        else wk.sents.assign(1, std::make_pair((size_t)0, wk.text.size()));
                        wk.store.clear();
                        wk.toks.clear();
                        for (const auto &se : wk.sents) wk.tk.tokenize(wk.text.data() + se.first, se.second - se.first, wk.store, wk.toks);
                        for (const auto &tk : wk.toks) {
                            if (wk.enc.size() < (size_t)tk.second * 4 + 1) wk.enc.resize((size_t)tk.second * 4 + 1);
                            const size_t nb = utf8_encode(wk.store.data() + tk.first, tk.second, wk.enc.data());
                            dst.append(reinterpret_cast<const char *>(wk.enc.data()), nb);
                            dst.push_back('\0');
                        }
                        c = (int64_t)wk.toks.size();
                    }
                }
                if (c < 0) {
                    int32_t expect = -1;
                    bad.compare_exchange_strong(expect, t);
                    c = 0;
                }
                res->counts[(size_t)t] = (int32_t)c;
                res->ends[(size_t)t] = (int64_t)dst.size();  // within the work item; made global below
            }
        }
    };
    try {
        if (nt == 1) work();
        else {
            std::vector<std::thread> th;
            th.reserve((size_t)nt);
            for (int i = 0; i < nt; ++i) th.emplace_back(work);
            for (auto &t : th) t.join();
        }
        size_t total = 0;
        for (const auto &p : parts) total += p.size();
        res->bytes.reserve(total);
        int64_t base = 0;
        for (int32_t it = 0; it < n_items; ++it) {
            const int32_t t0 = it * kGrain, t1 = std::min(n_texts, t0 + kGrain);
            for (int32_t t = t0; t < t1; ++t) res->ends[(size_t)t] += base;
            base += (int64_t)parts[(size_t)it].size();
            res->bytes.append(parts[(size_t)it]);
        }
    } catch (const std::exception &e) {
        delete res;
        mir::set_error("keywords_preprocess failed: %s", e.what());
        return MIR_ERR_INVALID;
    }
    if (bad.load() >= 0) {
        const int32_t t = bad.load();
        delete res;
        mir::set_error("text %d contains a NUL character", t);
        return MIR_ERR_INVALID;
    }
    for (int32_t c : res->counts) res->n_tokens += c;
    *out = res;
    return MIR_OK;
}

// the result's buffers (owned by the result, valid until mir_kwp_result_free): n_bytes of tokens, each followed by a NUL,
// text after text; counts[n_texts] tokens per text; byte_ends[n_texts]: text i's tokens are bytes [byte_ends[i - 1], byte_ends[i])
int32_t mir_kwp_result_data(const mir_kwp_result *r, const char **bytes, int64_t *n_bytes, const int32_t **counts,
                            const int64_t **byte_ends, int64_t *n_tokens) {
    MIR_REQUIRE(r != nullptr, "result is NULL");
    if (bytes) *bytes = r->bytes.data();
    if (n_bytes) *n_bytes = (int64_t)r->bytes.size();
    if (counts) *counts = r->counts.data();
    if (byte_ends) *byte_ends = r->ends.data();
    if (n_tokens) *n_tokens = r->n_tokens;
    return MIR_OK;
}

// The batch's DISTINCT tokens (order of first appearance) and every token as an index into them: a consumer that maps tokens
// to term ids (BM25Retriever's process-wide vocabulary, bm25_retriever.py:78) touches each distinct token once - ~20 000 per
// 4096 chunks instead of ~800 000 tokens - and a Python caller needs no str object per token.  Threads dedupe their own
// stretch of texts; the stretches' distinct tokens meet in one table, in order.
int32_t mir_kwp_result_dedupe(mir_kwp_result *r, int32_t n_threads) {
    MIR_REQUIRE(r != nullptr, "result is NULL");
    if (r->deduped) return MIR_OK;
    const int32_t n_texts = (int32_t)r->counts.size();
    int hw = (int)std::thread::hardware_concurrency();
    if (hw <= 0) hw = 1;
    int nt = n_threads > 0 ? n_threads : std::min(hw, 16);
    nt = std::max(1, std::min(nt, std::max(1, n_texts / 64)));
    struct Part {
        int32_t t0 = 0, t1 = 0;
        int64_t tok0 = 0;
        std::vector<std::string_view> uniq;
        std::vector<int32_t> table;
    };
    std::vector<Part> parts((size_t)nt);
    try {
        r->ids.assign((size_t)r->n_tokens, 0);
        {
            int64_t tok = 0;
            for (int p = 0; p < nt; ++p) {
                parts[p].t0 = (int32_t)((int64_t)n_texts * p / nt);
                parts[p].t1 = (int32_t)((int64_t)n_texts * (p + 1) / nt);
                parts[p].tok0 = tok;
                for (int32_t t = parts[p].t0; t < parts[p].t1; ++t) tok += r->counts[(size_t)t];
            }
        }
        const char *base = r->bytes.data();
        auto work = [&](int p) {
            Part &P = parts[p];
            std::unordered_map<std::string_view, int32_t> seen;
            seen.reserve(1 << 14);
            const int64_t b0 = P.t0 > 0 ? r->ends[(size_t)P.t0 - 1] : 0, b1 = P.t1 > 0 ? r->ends[(size_t)P.t1 - 1] : 0;
            int64_t tok = P.tok0;
            for (int64_t i = b0; i < b1;) {
                const size_t len = std::strlen(base + i);
                const std::string_view sv(base + i, len);
                auto it = seen.find(sv);
                int32_t id;
                if (it == seen.end()) {
                    id = (int32_t)P.uniq.size();
                    seen.emplace(sv, id);
                    P.uniq.push_back(sv);
                } else id = it->second;
                r->ids[(size_t)tok++] = id;
                i += (int64_t)len + 1;
            }
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int p = 0; p < nt; ++p) th.emplace_back(work, p);
            for (auto &t : th) t.join();
        }
        std::unordered_map<std::string_view, int32_t> all;
        all.reserve(1 << 16);
        std::vector<std::string_view> order;
        for (Part &P : parts) {
            P.table.resize(P.uniq.size());
            for (size_t u = 0; u < P.uniq.size(); ++u) {
                auto it = all.find(P.uniq[u]);
                if (it == all.end()) {
                    P.table[u] = (int32_t)order.size();
                    all.emplace(P.uniq[u], (int32_t)order.size());
                    order.push_back(P.uniq[u]);
                } else P.table[u] = it->second;
            }
        }
        auto remap = [&](int p) {
            const Part &P = parts[p];
            const int64_t e = p + 1 < nt ? parts[p + 1].tok0 : r->n_tokens;
            for (int64_t i = P.tok0; i < e; ++i) r->ids[(size_t)i] = P.table[(size_t)r->ids[(size_t)i]];
        };
        if (nt == 1) remap(0);
        else {
            std::vector<std::thread> th;
            for (int p = 0; p < nt; ++p) th.emplace_back(remap, p);
            for (auto &t : th) t.join();
        }
        size_t ub = 0;
        for (const auto &sv : order) ub += sv.size() + 1;
        r->uniq.clear();
        r->uniq.reserve(ub);
        for (const auto &sv : order) { r->uniq.append(sv.data(), sv.size()); r->uniq.push_back('\0'); }
        r->n_unique = (int32_t)order.size();
    } catch (const std::exception &e) {
        mir::set_error("keywords dedupe failed: %s", e.what());
        return MIR_ERR_INVALID;
    }
    r->deduped = true;
    return MIR_OK;
}

// after mir_kwp_result_dedupe: uniq_bytes = the distinct tokens, each followed by a NUL; ids[n_tokens] index them
int32_t mir_kwp_result_unique(const mir_kwp_result *r, const char **uniq_bytes, int64_t *n_uniq_bytes, int32_t *n_unique,
                              const int32_t **ids) {
    MIR_REQUIRE(r != nullptr && r->deduped, "result is NULL or not deduplicated (mir_kwp_result_dedupe)");
    if (uniq_bytes) *uniq_bytes = r->uniq.data();
    if (n_uniq_bytes) *n_uniq_bytes = (int64_t)r->uniq.size();
    if (n_unique) *n_unique = r->n_unique;
    if (ids) *ids = r->ids.data();
    return MIR_OK;
}

int32_t mir_kwp_result_free(mir_kwp_result *r) {
    delete r;
    return MIR_OK;
}

}  // extern "C"
