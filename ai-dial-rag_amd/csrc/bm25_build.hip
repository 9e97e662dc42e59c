// BM25 model build on the device: token-id documents -> postings by term (documents ascending), their
// float64 weights, per-term offsets and per-term tile offsets, all left in HBM for bm25.hip's kernels.
//
// The reference builds BM25Okapi(corpus) inside every request with Python dict loops over every token
// (bm25_retriever.py:64-79 -> rank_bm25.BM25Okapi.__init__); the host builder this replaces was one C++
// pass at ~10 ns per token (1.5 s for the 150M tokens of 1M chunks).  Here:
//   1. key[j] = term << 32 | doc for token j (doc by binary search in indptr), first position of every term
//      (atomicMin) for the host's idf average, whose summation order is first appearance (a Python dict);
//   2. stable radix sort on the term bits only: tokens arrive grouped by ascending document, so within a
//      term the documents stay ascending and a document's repeats stay adjacent;
//   3. run-length encode -> one posting per (term, doc) with tf = run length;
//   4. weights tf (k1 + 1) / (tf + k1 (1 - b + b dl / avgdl)) in float64, operation for operation as the
//      package (no contraction: this file is built with -ffp-contract=off), t_ptr and the tile table by
//      binary searches.
// The idf itself stays on the host (V logarithms through libm, as math.log in the package).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>
#include <vector>

#include "bm25_build.h"
#include "common.h"

namespace mir {
namespace {

__global__ __launch_bounds__(256) void make_keys_kernel(const int32_t *__restrict__ terms, const int64_t *__restrict__ indptr,
                                                        int64_t n_docs, int64_t total, int32_t vocab,
                                                        uint64_t *__restrict__ keys,
                                                        unsigned long long *__restrict__ first_pos,
                                                        int32_t *__restrict__ bad) {
    const int64_t base = indptr[0];
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < total; j += (int64_t)gridDim.x * 256) {
        const int32_t t = terms[j];
        if (t < 0 || t >= vocab) {
            atomicMax(bad, 1);
            keys[j] = ~0ull;
            continue;
        }
        // document of token j: the last i with indptr[i] - base <= j (empty documents share a boundary)
        int64_t lo = 0, hi = n_docs;  // answer in [lo, hi)
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (indptr[mid] - base <= j) lo = mid; else hi = mid;
        }
        keys[j] = ((uint64_t)(uint32_t)t << 32) | (uint64_t)(uint32_t)lo;
        if ((unsigned long long)j < first_pos[t]) atomicMin(&first_pos[t], (unsigned long long)j);
    }
}

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long *p, int64_t n, unsigned long long v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}

__global__ __launch_bounds__(256) void postings_kernel(const uint64_t *__restrict__ uniq, const int32_t *__restrict__ tf,
                                                       int64_t n_post, const int64_t *__restrict__ indptr, double k1,
                                                       double b, double avgdl, int32_t *__restrict__ p_doc,
                                                       double *__restrict__ p_w, int32_t *__restrict__ p_tf) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_post; i += (int64_t)gridDim.x * 256) {
        const uint32_t doc = (uint32_t)uniq[i];
        const double dl = (double)(indptr[doc + 1] - indptr[doc]);
        const double denom_len = k1 * ((1.0 - b) + (b * dl) / avgdl);
        const double f = (double)tf[i];
        p_doc[i] = (int32_t)doc;
        p_tf[i] = tf[i];
        p_w[i] = (f * (k1 + 1.0)) / (f + denom_len);
    }
}

__global__ __launch_bounds__(256) void doc_len_kernel(const int64_t *__restrict__ indptr, int64_t n_docs, int32_t *__restrict__ doc_len) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_docs; i += (int64_t)gridDim.x * 256)
        doc_len[i] = (int32_t)(indptr[i + 1] - indptr[i]);
}

// the same expression as postings_kernel, operation for operation, for another avgdl
__global__ __launch_bounds__(256) void reweight_kernel(const int32_t *__restrict__ p_doc, const int32_t *__restrict__ p_tf,
                                                       const int32_t *__restrict__ doc_len, int64_t n_post, double k1, double b,
                                                       double avgdl, double *__restrict__ p_w) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_post; i += (int64_t)gridDim.x * 256) {
        const double dl = (double)doc_len[p_doc[i]];
        const double denom_len = k1 * ((1.0 - b) + (b * dl) / avgdl);
        const double f = (double)p_tf[i];
        p_w[i] = (f * (k1 + 1.0)) / (f + denom_len);
    }
}

// t_ptr[t] = first posting whose term is >= t, for t in [0, vocab]
__global__ __launch_bounds__(256) void term_offsets_kernel(const uint64_t *__restrict__ uniq, int64_t n_post, int32_t vocab,
                                                           int64_t *__restrict__ t_ptr) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t > vocab) return;
    int64_t lo = 0, hi = n_post;  // first index with (uniq >> 32) >= t
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(uniq[mid] >> 32) < t) lo = mid + 1; else hi = mid;
    }
    t_ptr[t] = lo;
}

// t_tile[t][j] = postings of term t with doc < j * tile, j in [0, T]
__global__ __launch_bounds__(256) void tile_offsets_kernel(const int32_t *__restrict__ p_doc, const int64_t *__restrict__ t_ptr,
                                                           int32_t vocab, int T, int tile, uint32_t *__restrict__ t_tile) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)vocab * (T + 1)) return;
    const int64_t t = e / (T + 1);
    const int j = (int)(e - t * (T + 1));
    const int64_t bound = (int64_t)j * tile;
    int64_t lo = t_ptr[t], hi = t_ptr[t + 1];
    const int64_t first = lo;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)p_doc[mid] < bound) lo = mid + 1; else hi = mid;
    }
    t_tile[e] = (uint32_t)(lo - first);
}

struct Scratch {  // frees what the build allocated and no longer needs, on every exit path
    std::vector<void *> ptrs;
    ~Scratch() { for (void *p : ptrs) (void)hipFree(p); }
    hipError_t take(void **out, size_t bytes) {
        hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) ptrs.push_back(*out);
        return e;
    }
};

unsigned grid_for(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1 << 16)); }

}  // namespace

#define BUILD_TRY(call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_error("%s failed: %s", #call, hipGetErrorString(e_));                              \
            return MIR_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

int32_t bm25_build_device(const int64_t *indptr, const int32_t *term_ids, int64_t n_docs, int32_t vocab, double k1,
                          double b, double avgdl, int tile, Bm25Built *out) {
    const int64_t total = n_docs ? indptr[n_docs] - indptr[0] : 0;
    MIR_REQUIRE(total < ((int64_t)1 << 31), "%lld tokens: more than 2^31 - 1 in one model", (long long)total);
    const int T = (int)std::max<int64_t>(1, (n_docs + tile - 1) / tile);
    hipStream_t s = nullptr;
    Scratch tmp;
    int64_t *d_indptr = nullptr;
    int32_t *d_terms = nullptr, *d_bad = nullptr, *d_tf = nullptr;
    uint64_t *d_keys = nullptr, *d_sorted = nullptr, *d_uniq = nullptr;
    unsigned long long *d_first = nullptr;
    int64_t *d_nruns = nullptr;
    BUILD_TRY(tmp.take((void **)&d_indptr, (size_t)(n_docs + 1) * 8));
    BUILD_TRY(tmp.take((void **)&d_terms, (size_t)total * 4));
    BUILD_TRY(tmp.take((void **)&d_keys, (size_t)total * 8));
    BUILD_TRY(tmp.take((void **)&d_sorted, (size_t)total * 8));
    BUILD_TRY(tmp.take((void **)&d_first, (size_t)vocab * 8));
    BUILD_TRY(tmp.take((void **)&d_bad, 16));
    BUILD_TRY(tmp.take((void **)&d_nruns, 16));
    if (n_docs) BUILD_TRY(hipMemcpyAsync(d_indptr, indptr, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice, s));
    if (total) BUILD_TRY(hipMemcpyAsync(d_terms, term_ids + indptr[0], (size_t)total * 4, hipMemcpyHostToDevice, s));
    BUILD_TRY(hipMemsetAsync(d_bad, 0, 16, s));
    BUILD_TRY(hipMemsetAsync(d_nruns, 0, 16, s));
    fill_u64_kernel<<<dim3(grid_for(vocab)), dim3(256), 0, s>>>(d_first, vocab, ~0ull);
    int64_t n_post = 0;
    if (total) {
        make_keys_kernel<<<dim3(grid_for(total)), dim3(256), 0, s>>>(d_terms, d_indptr, n_docs, total, vocab, d_keys, d_first, d_bad);
        BUILD_TRY(hipGetLastError());
        int32_t bad = 0;
        BUILD_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s));
        BUILD_TRY(hipStreamSynchronize(s));
        MIR_REQUIRE(bad == 0, "a term id lies outside [0, %d)", vocab);
        int term_bits = 1;
        while (((int64_t)1 << term_bits) < (int64_t)vocab) ++term_bits;
        size_t sort_bytes = 0, rle_bytes = 0;
        BUILD_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, d_keys, d_sorted, (int)total, 32, 32 + term_bits, s));
        void *d_tmp = nullptr;
        BUILD_TRY(tmp.take(&d_tmp, sort_bytes));
        BUILD_TRY(hipcub::DeviceRadixSort::SortKeys(d_tmp, sort_bytes, d_keys, d_sorted, (int)total, 32, 32 + term_bits, s));
        // d_keys is free again: the unique (term, doc) keys go there
        d_uniq = d_keys;
        BUILD_TRY(tmp.take((void **)&d_tf, (size_t)total * 4));
        BUILD_TRY(hipcub::DeviceRunLengthEncode::Encode(nullptr, rle_bytes, d_sorted, d_uniq, d_tf, d_nruns, (int)total, s));
        void *d_tmp2 = nullptr;
        BUILD_TRY(tmp.take(&d_tmp2, rle_bytes));
        BUILD_TRY(hipcub::DeviceRunLengthEncode::Encode(d_tmp2, rle_bytes, d_sorted, d_uniq, d_tf, d_nruns, (int)total, s));
        BUILD_TRY(hipMemcpyAsync(&n_post, d_nruns, 8, hipMemcpyDeviceToHost, s));
        BUILD_TRY(hipStreamSynchronize(s));
    }
    MIR_REQUIRE((int64_t)vocab * (T + 1) < ((int64_t)1 << 33), "vocab x tiles table too large (%d x %d)", vocab, T + 1);

    // products: owned by the caller from here on (freed by it on failure too)
    BUILD_TRY(hipMalloc((void **)&out->p_doc, std::max<size_t>((size_t)n_post * 4, 16)));
    BUILD_TRY(hipMalloc((void **)&out->p_w, std::max<size_t>((size_t)n_post * 8, 16)));
    BUILD_TRY(hipMalloc((void **)&out->p_tf, std::max<size_t>((size_t)n_post * 4, 16)));
    BUILD_TRY(hipMalloc((void **)&out->doc_len, std::max<size_t>((size_t)n_docs * 4, 16)));
    BUILD_TRY(hipMalloc((void **)&out->t_ptr, (size_t)(vocab + 1) * 8));
    BUILD_TRY(hipMalloc((void **)&out->t_tile, (size_t)vocab * (T + 1) * 4));
    out->n_postings = n_post;
    out->ntiles = T;
    out->hbm_bytes = n_post * 16 + n_docs * 4 + (int64_t)(vocab + 1) * 8 + (int64_t)vocab * (T + 1) * 4;
    if (n_post) {
        postings_kernel<<<dim3(grid_for(n_post)), dim3(256), 0, s>>>(d_uniq, d_tf, n_post, d_indptr, k1, b, avgdl, out->p_doc, out->p_w,
                                                                    out->p_tf);
        BUILD_TRY(hipGetLastError());
    }
    if (n_docs) {
        doc_len_kernel<<<dim3(grid_for(n_docs)), dim3(256), 0, s>>>(d_indptr, n_docs, out->doc_len);
        BUILD_TRY(hipGetLastError());
    }
    term_offsets_kernel<<<dim3((unsigned)((vocab + 1 + 255) / 256)), dim3(256), 0, s>>>(d_uniq, n_post, vocab, out->t_ptr);
    BUILD_TRY(hipGetLastError());
    const int64_t cells = (int64_t)vocab * (T + 1);
    tile_offsets_kernel<<<dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s>>>(out->p_doc, out->t_ptr, vocab, T, tile, out->t_tile);
    BUILD_TRY(hipGetLastError());
    // what the host needs for the idf: document frequency and first position of every term
    out->t_ptr_host.resize((size_t)vocab + 1);
    out->first_pos.resize((size_t)vocab);
    BUILD_TRY(hipMemcpyAsync(out->t_ptr_host.data(), out->t_ptr, (size_t)(vocab + 1) * 8, hipMemcpyDeviceToHost, s));
    BUILD_TRY(hipMemcpyAsync(out->first_pos.data(), d_first, (size_t)vocab * 8, hipMemcpyDeviceToHost, s));
    BUILD_TRY(hipStreamSynchronize(s));
    return MIR_OK;
}

int32_t bm25_reweight_device(const int32_t *p_doc, const int32_t *p_tf, const int32_t *doc_len, int64_t n_postings, double k1,
                             double b, double avgdl, double *p_w, void *stream) {
    if (n_postings == 0) return MIR_OK;
    reweight_kernel<<<dim3(grid_for(n_postings)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(p_doc, p_tf, doc_len, n_postings, k1,
                                                                                                 b, avgdl, p_w);
    BUILD_TRY(hipGetLastError());
    return MIR_OK;
}

}  // namespace mir
