// BERT WordPiece tokenisation of text batches (host code, no GPU): the step in front of the encoder on both of
// its entry points - `aembed_documents` at index build and `aembed_query` per request
// (aidial_rag/embeddings/embeddings.py:79-96 -> langchain HuggingFaceBgeEmbeddings -> sentence-transformers ->
// the model's BertTokenizer).  With the forward pass on the GPU (52k chunks/s), the Hugging Face tokenizer called
// per outer batch of 128 texts was the index build's bottleneck (~150 us per chunk and thread).
//
// Behaviour follows transformers' BertTokenizer / tokenizers' BertNormalizer + BertPreTokenizer + WordPiece
// (do_lower_case = True, strip_accents = None -> follows lower-casing, tokenize_chinese_chars = True, max 100
// characters per word, "##" continuation, [UNK] for a word with an unmatched piece), pinned against it on a
// synthetic vocabulary (tests/test_wordpiece.py).  The per-code-point facts - class (whitespace / control / punctuation
// / CJK) and the normalised form (NFD, combining marks dropped, lower-cased) - come from the CALLER as tables over the
// Basic Multilingual Plane (the Python side fills them from `unicodedata`), so this file carries no Unicode data; a
// text with a code point beyond the BMP is reported back (`fallback`) and tokenised by the caller's reference
// tokenizer instead.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace {

enum : uint8_t { CLS_OTHER = 0, CLS_SPACE = 1, CLS_REMOVE = 2, CLS_PUNCT = 3, CLS_CJK = 4, CLS_FALLBACK = 5 };

struct StrHash {
    size_t operator()(const std::string &s) const noexcept {
        uint64_t h = 1469598103934665603ull;
        for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
        return (size_t)h;
    }
};

}  // namespace

struct mir_wordpiece {
    std::unordered_map<std::string, int32_t, StrHash> vocab;
    std::vector<uint8_t> cls;        // [65536] class of a RAW code point
    std::vector<uint8_t> ncls;       // [65536] class of a NORMALISED code point (for the punctuation split)
    std::vector<uint32_t> map;       // [65536][3] normalised form
    std::vector<uint8_t> maplen;     // [65536] 0..3
    int32_t unk = 0, cls_id = 0, sep_id = 0;
    int max_chars = 100;
};

namespace {

inline void put_utf8(std::string &s, uint32_t c) {
    if (c < 0x80) s.push_back((char)c);
    else if (c < 0x800) { s.push_back((char)(0xC0 | (c >> 6))); s.push_back((char)(0x80 | (c & 0x3F))); }
    else if (c < 0x10000) { s.push_back((char)(0xE0 | (c >> 12))); s.push_back((char)(0x80 | ((c >> 6) & 0x3F))); s.push_back((char)(0x80 | (c & 0x3F))); }
    else { s.push_back((char)(0xF0 | (c >> 18))); s.push_back((char)(0x80 | ((c >> 12) & 0x3F))); s.push_back((char)(0x80 | ((c >> 6) & 0x3F))); s.push_back((char)(0x80 | (c & 0x3F))); }
}

// one word (normalised code points) -> WordPiece ids appended to out
void wordpiece_word(const mir_wordpiece &t, const uint32_t *w, int n, std::vector<int32_t> &out, std::string &buf,
                    std::vector<int> &off) {
    if (n > t.max_chars) { out.push_back(t.unk); return; }
    // UTF-8 of the word once, with the byte offset of every character
    buf.assign("##");
    off.resize(n + 1);
    for (int i = 0; i < n; ++i) { off[i] = (int)buf.size(); put_utf8(buf, w[i]); }
    off[n] = (int)buf.size();
    const size_t mark = out.size();
    int start = 0;
    std::string piece;
    while (start < n) {
        int end = n, found = -1;
        while (end > start) {
            if (start == 0) piece.assign(buf, off[0], off[end] - off[0]);
            else { piece.assign("##"); piece.append(buf, off[start], off[end] - off[start]); }
            auto it = t.vocab.find(piece);
            if (it != t.vocab.end()) { found = it->second; break; }
            --end;
        }
        if (found < 0) { out.resize(mark); out.push_back(t.unk); return; }
        out.push_back(found);
        start = end;
    }
}

// one text -> ids (without specials); returns false when the text needs the caller's tokenizer (non-BMP code point or
// malformed UTF-8)
bool encode_text(const mir_wordpiece &t, const unsigned char *s, int64_t n, std::vector<int32_t> &out,
                 std::vector<uint32_t> &norm, std::vector<uint8_t> &ncls, std::string &buf, std::vector<int> &off) {
    norm.clear(); ncls.clear();
    // A literal special token in the text ("[SEP]", "[CLS]", "[MASK]", "[PAD]", "[UNK]"): BertTokenizerFast cuts those out of
    // the RAW text before normalisation and emits their ids; here '[', 'sep', ']' would come out.  Such a text (a page about
    // tokenizers, say) is handed back to the caller's tokenizer.
    for (int64_t i = 0; i + 5 <= n; ++i) {
        if (s[i] != '[') continue;
        static const char *const kSpecial[] = {"[SEP]", "[CLS]", "[MASK]", "[PAD]", "[UNK]"};
        for (const char *sp : kSpecial) {
            const size_t L = std::strlen(sp);
            if ((size_t)(n - i) >= L && std::memcmp(s + i, sp, L) == 0) return false;
        }
    }
    for (int64_t i = 0; i < n;) {
        uint32_t c = s[i];
        int len = 1;
        if (c >= 0x80) {
            if ((c & 0xE0) == 0xC0) { len = 2; c &= 0x1F; }
            else if ((c & 0xF0) == 0xE0) { len = 3; c &= 0x0F; }
            else return false;  // 4-byte sequences (beyond the BMP) and stray bytes
            if (i + len > n) return false;
            for (int k = 1; k < len; ++k) {
                if ((s[i + k] & 0xC0) != 0x80) return false;
                c = (c << 6) | (s[i + k] & 0x3F);
            }
        }
        i += len;
        const uint8_t k = t.cls[c];
        if (k == CLS_FALLBACK) return false;                  // a code point whose normal form the tables cannot hold
        if (k == CLS_REMOVE) continue;                       // clean_text: NUL, U+FFFD, control characters
        if (k == CLS_SPACE) { norm.push_back(' '); ncls.push_back(CLS_SPACE); continue; }
        if (k == CLS_CJK) { norm.push_back(' '); ncls.push_back(CLS_SPACE); }  // tokenize_chinese_chars: spaces around
        const int ml = t.maplen[c];
        for (int j = 0; j < ml; ++j) {
            const uint32_t m = t.map[(size_t)c * 3 + j];
            norm.push_back(m);
            ncls.push_back(m < 0x10000 ? t.ncls[m] : (uint8_t)CLS_OTHER);
        }
        if (k == CLS_CJK) { norm.push_back(' '); ncls.push_back(CLS_SPACE); }
    }
    // whitespace split, punctuation characters as words of their own
    const int total = (int)norm.size();
    int i = 0;
    while (i < total) {
        if (ncls[i] == CLS_SPACE) { ++i; continue; }
        if (ncls[i] == CLS_PUNCT) { wordpiece_word(t, &norm[i], 1, out, buf, off); ++i; continue; }
        int j = i;
        while (j < total && ncls[j] != CLS_SPACE && ncls[j] != CLS_PUNCT) ++j;
        wordpiece_word(t, &norm[i], j - i, out, buf, off);
        i = j;
    }
    return true;
}

}  // namespace

extern "C" {

// vocab: the lines of vocab.txt (UTF-8, '\n'-separated; id = line number).  cls / ncls: class per BMP code point of the
// raw text / of normalised text: 0 other, 1 whitespace, 2 removed, 3 punctuation, 4 CJK (raw only), 5 = hand the text back.  map / maplen: the
// normalised form of every BMP code point (up to 3 code points).
int32_t mir_wordpiece_create(const char *vocab, int64_t vocab_bytes, const uint8_t *cls, const uint8_t *ncls, const uint32_t *map,
                             const uint8_t *maplen, int32_t max_chars_per_word, mir_wordpiece **out) {
    MIR_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    MIR_REQUIRE(vocab && vocab_bytes > 0 && cls && ncls && map && maplen, "NULL argument");
    mir_wordpiece *t = new (std::nothrow) mir_wordpiece();
    MIR_REQUIRE(t != nullptr, "out of host memory");
    t->cls.assign(cls, cls + 65536);
    t->ncls.assign(ncls, ncls + 65536);
    t->map.assign(map, map + (size_t)65536 * 3);
    t->maplen.assign(maplen, maplen + 65536);
    t->max_chars = max_chars_per_word > 0 ? max_chars_per_word : 100;
    int32_t id = 0;
    for (int64_t i = 0; i < vocab_bytes;) {
        int64_t j = i;
        while (j < vocab_bytes && vocab[j] != '\n') ++j;
        int64_t e = j;
        if (e > i && vocab[e - 1] == '\r') --e;
        t->vocab[std::string(vocab + i, (size_t)(e - i))] = id++;  // a repeated token keeps its LAST id, as the dict the reference loads
        i = j + 1;
    }
    auto need = [&](const char *tok, int32_t *dst) {
        auto it = t->vocab.find(tok);
        if (it == t->vocab.end()) return false;
        *dst = it->second;
        return true;
    };
    if (!need("[UNK]", &t->unk) || !need("[CLS]", &t->cls_id) || !need("[SEP]", &t->sep_id)) {
        delete t;
        mir::set_error("vocabulary lacks [UNK] / [CLS] / [SEP]");
        return MIR_ERR_INVALID;
    }
    *out = t;
    return MIR_OK;
}

int32_t mir_wordpiece_destroy(mir_wordpiece *t) {
    delete t;
    return MIR_OK;
}

// texts: n UTF-8 strings back to back, text i = [text_ptr[i], text_ptr[i+1]).  Output: ids with [CLS] / [SEP] in
// place, truncated to max_len tokens in all (the tokenizer's `truncation=True, max_length=512`), sequence i at
// out_ids[i * max_len .. + out_len[i]).  fallback[i] = 1: text i holds a code point beyond the BMP (or invalid UTF-8);
// nothing was written for it.  Runs on `threads` host threads (<= 0: one per core, at most 32).
int32_t mir_wordpiece_encode(const mir_wordpiece *t, const char *texts, const int64_t *text_ptr, int32_t n, int32_t max_len,
                             int32_t threads, int32_t *out_ids, int32_t *out_len, uint8_t *fallback) {
    MIR_REQUIRE(t != nullptr && n >= 0 && max_len >= 2, "bad argument");
    if (n == 0) return MIR_OK;
    MIR_REQUIRE(texts && text_ptr && out_ids && out_len && fallback, "NULL buffer");
    int nthr = threads > 0 ? threads : (int)std::min<unsigned>(32, std::max<unsigned>(1, std::thread::hardware_concurrency()));
    // at least 64 texts (~0.5 ms) per thread: the index build calls this from up to 32 Python threads at once with outer
    // batches of 128 texts, and 8 threads per call were 256 on the cores of one GPU's share of the host
    nthr = std::max(1, std::min(nthr, (n + 63) / 64));
    auto work = [&](int lo, int hi) {
        std::vector<int32_t> ids;
        std::vector<uint32_t> norm;
        std::vector<uint8_t> ncls;
        std::vector<int> off;
        std::string buf;
        for (int i = lo; i < hi; ++i) {
            ids.clear();
            const bool ok = encode_text(*t, reinterpret_cast<const unsigned char *>(texts) + text_ptr[i], text_ptr[i + 1] - text_ptr[i],
                                        ids, norm, ncls, buf, off);
            fallback[i] = ok ? 0 : 1;
            if (!ok) { out_len[i] = 0; continue; }
            const int keep = (int)std::min<size_t>(ids.size(), (size_t)max_len - 2);
            int32_t *dst = out_ids + (size_t)i * max_len;
            dst[0] = t->cls_id;
            std::memcpy(dst + 1, ids.data(), sizeof(int32_t) * keep);
            dst[keep + 1] = t->sep_id;
            out_len[i] = keep + 2;
        }
    };
    if (nthr == 1) {
        work(0, n);
    } else {
        std::vector<std::thread> pool;
        for (int w = 0; w < nthr; ++w) pool.emplace_back(work, (int)((int64_t)n * w / nthr), (int)((int64_t)n * (w + 1) / nthr));
        for (auto &th : pool) th.join();
    }
    return MIR_OK;
}

}  // extern "C"
