// Device-side BM25 model build (bm25_build.hip), used by mir_bm25_create (bm25.hip).
#pragma once
#include <cstdint>
#include <vector>

namespace mir {

struct Bm25Built {
    // in HBM, owned by the caller once bm25_build_device returns (also when it fails half way)
    int32_t *p_doc = nullptr;   // [n_postings] documents, ascending within a term
    double *p_w = nullptr;      // [n_postings] tf (k1 + 1) / (tf + k1 (1 - b + b dl / avgdl))
    int32_t *p_tf = nullptr;    // [n_postings] term frequency (kept so the weights can be re-derived for another avgdl)
    int32_t *doc_len = nullptr; // [n_docs] tokens per document
    int64_t *t_ptr = nullptr;   // [vocab + 1]
    uint32_t *t_tile = nullptr; // [vocab][ntiles + 1]
    int64_t n_postings = 0;
    int ntiles = 0;
    int64_t hbm_bytes = 0;
    // on the host, for the idf
    std::vector<int64_t> t_ptr_host;             // [vocab + 1]: df(t) = t_ptr[t + 1] - t_ptr[t]
    std::vector<unsigned long long> first_pos;   // [vocab]: position of the term's first token (~0 = absent)
};

// indptr[n_docs + 1] / term_ids: host arrays as given to mir_bm25_create; the device must be current.
// p_w for a different average length (document-sharded corpus: the global avgdl arrives after the local build)
int32_t bm25_reweight_device(const int32_t *p_doc, const int32_t *p_tf, const int32_t *doc_len, int64_t n_postings, double k1,
                             double b, double avgdl, double *p_w, void *stream);

int32_t bm25_build_device(const int64_t *indptr, const int32_t *term_ids, int64_t n_docs, int32_t vocab, double k1,
                          double b, double avgdl, int tile, Bm25Built *out);

}  // namespace mir
