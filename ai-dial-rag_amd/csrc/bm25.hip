// BM25 keyword scoring + top-k for gfx950.  C ABI in include/miretr.h.
//
// Replaces the per-request rank_bm25.BM25Okapi model built and queried at
//   aidial_rag/retrievers/bm25_retriever.py:64-84
// (third-party rank-bm25 0.2.2: k1 = 1.5, b = 0.75, epsilon = 0.25; the
// arithmetic restated in oracle/bm25.py).  All scores are float64 in the
// package's operation order, so results are bit-identical - which is what makes
// the reference's tie-break (`argsort(stable)[::-1]`: equal scores, the zero
// scores included, go to the HIGHEST flat index) reproducible.
//
// Layout in HBM
//   postings grouped by term, documents ascending inside a term:
//     p_doc i32[P], p_w f64[P]     w = tf*(k1+1) / (tf + k1*(1 - b + b*dl/avgdl))
//   t_ptr  i64[V+1]                posting range of term t
//   t_tile u32[V][T+1]             offset (inside the term's range) of the first
//                                  posting whose document falls in tile >= j
//   idf    f64[V]
// Scoring never materialises the dense float64[N] score vector the reference
// builds per query.  A workgroup owns one tile of 8192 consecutive documents
// for one query: it accumulates the tile's scores in LDS, one query term after
// the other in query order (deterministic rounding, a document occurs at most
// once per term so no atomics), then selects the tile's top-k from LDS.  HBM
// traffic per query is the postings of its terms (12 B each), read once,
// coalesced.  A second small kernel merges the tiles' candidates.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "common.h"

namespace mir {

constexpr int kBm25Tile = 8192;   // documents per LDS tile (64 KiB of float64)
constexpr int kBm25MaxK = 64;

// (a before b): score descending, then index DESCENDING (bm25_retriever.py:84)
__device__ __forceinline__ bool bm25_before(double sa, int64_t ia, double sb, int64_t ib) {
    return sa > sb || (sa == sb && ia > ib);
}

// Block-wide selection of the best `kout` entries of `n` candidates read through
// `score(i)` / `index(i)`; `taken(i)` marks consumed entries.  Each thread
// caches the best of its strided subset; per round one block arg-max picks the
// winner and only the winner's owner rescans.  emit(rank, score, index).
template <typename ScoreF, typename IndexF, typename TakeF, typename EmitF>
__device__ inline void block_select(int n, int kout, int tid, ScoreF score, IndexF index, TakeF take, EmitF emit,
                                    double *red_s, int64_t *red_i, int *red_p) {
    const double NEG = -__builtin_inf();
    auto local_best = [&](double &bs, int64_t &bi, int &bp) {
        bs = NEG; bi = -1; bp = -1;
        for (int i = tid; i < n; i += 256) {
            const double s = score(i);
            if (s == NEG) continue;  // taken (or padding)
            const int64_t ix = index(i);
            if (bp < 0 || bm25_before(s, ix, bs, bi)) { bs = s; bi = ix; bp = i; }
        }
    };
    double bs; int64_t bi; int bp;
    local_best(bs, bi, bp);
    for (int r = 0; r < kout; ++r) {
        // wave arg-max
        double ws = bs; int64_t wi = bi; int wp = bp;
        for (int off = 32; off >= 1; off >>= 1) {
            const double os = __shfl_xor(ws, off, 64);
            const int64_t oi = __shfl_xor(wi, off, 64);
            const int op = __shfl_xor(wp, off, 64);
            if (op >= 0 && (wp < 0 || bm25_before(os, oi, ws, wi))) { ws = os; wi = oi; wp = op; }
        }
        if ((tid & 63) == 0) { red_s[tid >> 6] = ws; red_i[tid >> 6] = wi; red_p[tid >> 6] = wp; }
        __syncthreads();
        double gs = red_s[0]; int64_t gi = red_i[0]; int gp = red_p[0];
        for (int w = 1; w < 4; ++w)
            if (red_p[w] >= 0 && (gp < 0 || bm25_before(red_s[w], red_i[w], gs, gi))) { gs = red_s[w]; gi = red_i[w]; gp = red_p[w]; }
        __syncthreads();
        if (gp < 0) break;  // fewer candidates than kout (uniform)
        if (tid == 0) emit(r, gs, gi);
        if (gp == bp) {  // this thread owned the winner
            take(gp);
            local_best(bs, bi, bp);
        }
    }
}

struct Bm25Dev {
    const int32_t *p_doc;
    const double *p_w;
    const int64_t *t_ptr;
    const uint32_t *t_tile;
    const double *idf;
    int vocab;
    int64_t n_docs;
    int ntiles;
};

// grid = (ntiles, b), block = 256.  q_ptr[b+1] slices q_terms.
__global__ __launch_bounds__(256) void bm25_tile_kernel(Bm25Dev m, const int32_t *__restrict__ q_terms,
                                                        const int32_t *__restrict__ q_ptr, int k,
                                                        double *__restrict__ out_scores,
                                                        double *__restrict__ part_score,
                                                        int32_t *__restrict__ part_idx,
                                                        int32_t *__restrict__ part_cnt) {
    __shared__ double sc[kBm25Tile];
    __shared__ double red_s[4];
    __shared__ int64_t red_i[4];
    __shared__ int red_p[4];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, q = blockIdx.y;
    const int64_t base = (int64_t)tile * kBm25Tile;
    const int cnt = (int)((m.n_docs - base) < kBm25Tile ? (m.n_docs - base) : kBm25Tile);
    for (int i = tid; i < kBm25Tile; i += 256) sc[i] = 0.0;
    for (int j = q_ptr[q]; j < q_ptr[q + 1]; ++j) {
        const int t = q_terms[j];
        if (t < 0 || t >= m.vocab) continue;  // unknown term: `(doc.get(q) or 0)` everywhere
        const double w_idf = m.idf[t];
        if (w_idf == 0.0) continue;           // `(self.idf.get(q) or 0)`: adds +-0
        const int64_t t0 = m.t_ptr[t];
        const uint32_t *to = m.t_tile + (size_t)t * (m.ntiles + 1) + tile;
        const int64_t lo = t0 + to[0], hi = t0 + to[1];
        __syncthreads();  // previous term's adds are complete (t is block-uniform)
        for (int64_t p = lo + tid; p < hi; p += 256) {
            const int i = m.p_doc[p] - (int)base;
            sc[i] = sc[i] + w_idf * m.p_w[p];  // one rounding for the product, one for the sum
        }
    }
    __syncthreads();
    if (out_scores) {
        double *o = out_scores + (size_t)q * m.n_docs + base;
        for (int i = tid; i < cnt; i += 256) o[i] = sc[i];
    }
    if (part_score) {
        const double NEG = -__builtin_inf();
        for (int i = cnt + tid; i < kBm25Tile; i += 256) sc[i] = NEG;  // padding never selected
        __syncthreads();
        const int kout = k < cnt ? k : cnt;
        const size_t pb = ((size_t)q * m.ntiles + tile) * k;
        block_select(
            cnt, kout, tid, [&](int i) { return sc[i]; }, [&](int i) { return (int64_t)i; },
            [&](int i) { sc[i] = NEG; },
            [&](int r, double s, int64_t i) {
                part_score[pb + r] = s;
                part_idx[pb + r] = (int32_t)(base + i);
            },
            red_s, red_i, red_p);
        if (tid == 0) part_cnt[(size_t)q * m.ntiles + tile] = kout;
    }
}

// grid = b, block = 256: merge the tiles' candidates of one query.
__global__ __launch_bounds__(256) void bm25_merge_kernel(double *__restrict__ part_score,
                                                         const int32_t *__restrict__ part_idx,
                                                         const int32_t *__restrict__ part_cnt, int ntiles, int k,
                                                         int64_t doc_offset, int64_t *__restrict__ out_idx,
                                                         double *__restrict__ out_score,
                                                         int32_t *__restrict__ out_count) {
    __shared__ double red_s[4];
    __shared__ int64_t red_i[4];
    __shared__ int red_p[4];
    __shared__ int s_total;
    const int tid = threadIdx.x, q = blockIdx.x;
    double *ps = part_score + (size_t)q * ntiles * k;
    const int32_t *pi = part_idx + (size_t)q * ntiles * k;
    const int32_t *pc = part_cnt + (size_t)q * ntiles;
    const double NEG = -__builtin_inf();
    int mine = 0;
    for (int e = tid; e < ntiles * k; e += 256) {
        const int tile = e / k, p = e - tile * k;
        if (p >= pc[tile]) ps[e] = NEG; else ++mine;
    }
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if ((tid & 63) == 0) red_p[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) s_total = red_p[0] + red_p[1] + red_p[2] + red_p[3];
    __syncthreads();
    const int kout = k < s_total ? k : s_total;
    __syncthreads();
    block_select(
        ntiles * k, kout, tid, [&](int e) { return ps[e]; }, [&](int e) { return (int64_t)pi[e]; },
        [&](int e) { ps[e] = NEG; },
        [&](int r, double s, int64_t i) {
            out_score[(size_t)q * k + r] = s;
            out_idx[(size_t)q * k + r] = doc_offset + i;
        },
        red_s, red_i, red_p);
    if (tid == 0) out_count[q] = kout;
}

}  // namespace mir

using namespace mir;

struct mir_bm25 {
    int device = 0;
    int64_t n_docs = 0;
    int vocab = 0;
    int64_t n_postings = 0;
    int ntiles = 0;
    int64_t doc_offset = 0;
    double avgdl = 0.0, average_idf = 0.0;
    std::vector<double> h_idf;
    int32_t *p_doc = nullptr;
    double *p_w = nullptr;
    int64_t *t_ptr = nullptr;
    uint32_t *t_tile = nullptr;
    double *idf = nullptr;
    int64_t hbm_bytes = 0;
    std::mutex mu;  // serialises use of the scratch below (searches on one handle run one at a time)
    void *scratch = nullptr;
    size_t scratch_cap = 0;
    hipStream_t stream = nullptr;
};

namespace mir {

static void free_bm25(mir_bm25 *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->p_doc);
    (void)hipFree(h->p_w);
    (void)hipFree(h->t_ptr);
    (void)hipFree(h->t_tile);
    (void)hipFree(h->idf);
    (void)hipFree(h->scratch);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static Bm25Dev dev_view(const mir_bm25 *h) {
    Bm25Dev m;
    m.p_doc = h->p_doc; m.p_w = h->p_w; m.t_ptr = h->t_ptr; m.t_tile = h->t_tile; m.idf = h->idf;
    m.vocab = h->vocab; m.n_docs = h->n_docs; m.ntiles = h->ntiles;
    return m;
}

static int32_t ensure_scratch(mir_bm25 *h, size_t need) {
    if (h->scratch_cap >= need) return MIR_OK;
    if (h->scratch) (void)hipFree(h->scratch);
    h->scratch = nullptr;
    h->scratch_cap = 0;
    MIR_HIP(hipMalloc(&h->scratch, need));
    h->scratch_cap = need;
    return MIR_OK;
}

}  // namespace mir

extern "C" {

// Builds the model from token-id documents.  `indptr[i]:indptr[i+1]` slices the
// tokens of document i in text order.  If `idf_override` is non-NULL it supplies
// idf[vocab] and `avgdl_override` the average length (global statistics of a
// sharded corpus); otherwise both are derived from these documents exactly as
// BM25Okapi.__init__ does.
int32_t mir_bm25_create(const int64_t *indptr, const int32_t *term_ids, int64_t n_docs, int32_t vocab, double k1,
                        double b, double epsilon, const double *idf_override, double avgdl_override,
                        int32_t device, int64_t doc_offset, mir_bm25 **out) {
    MIR_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    MIR_REQUIRE(n_docs >= 0 && n_docs < ((int64_t)1 << 31), "n_docs=%lld out of range", (long long)n_docs);
    MIR_REQUIRE(vocab >= 1, "vocab=%d must be >= 1", vocab);
    MIR_REQUIRE(n_docs == 0 || indptr != nullptr, "indptr is NULL");
    const int64_t total = n_docs ? indptr[n_docs] - indptr[0] : 0;
    if (total == 0 && idf_override == nullptr) {
        set_error("Text index is empty.");  // bm25_retriever.py:75-76
        return MIR_ERR_EMPTY;
    }
    MIR_REQUIRE(total == 0 || term_ids != nullptr, "term_ids is NULL");
    const int64_t t_base = n_docs ? indptr[0] : 0;
    for (int64_t i = 0; i < n_docs; ++i)
        MIR_REQUIRE(indptr[i + 1] >= indptr[i], "indptr is not monotone at %lld", (long long)i);
    for (int64_t j = 0; j < total; ++j)
        MIR_REQUIRE(term_ids[t_base + j] >= 0 && term_ids[t_base + j] < vocab, "term id %d at %lld outside [0, %d)",
                    term_ids[t_base + j], (long long)j, vocab);
    int32_t rc = use_device(device, nullptr);
    if (rc != MIR_OK) return rc;

    // ---- BM25._initialize: per-document tf, df in first-appearance order ----
    std::vector<int64_t> df(vocab, 0);
    std::vector<int64_t> last_doc(vocab, -1);
    std::vector<int64_t> slot(vocab, 0);
    std::vector<int32_t> order;                 // terms in order of first appearance (dict insertion order)
    std::vector<int32_t> e_term; std::vector<int32_t> e_tf; std::vector<int64_t> e_ptr(n_docs + 1, 0);
    e_term.reserve((size_t)total); e_tf.reserve((size_t)total);
    for (int64_t i = 0; i < n_docs; ++i) {
        for (int64_t j = indptr[i]; j < indptr[i + 1]; ++j) {
            const int32_t t = term_ids[j];
            if (last_doc[t] != i) {
                last_doc[t] = i;
                slot[t] = (int64_t)e_term.size();
                e_term.push_back(t);
                e_tf.push_back(1);
                if (df[t] == 0) order.push_back(t);
                ++df[t];
            } else {
                ++e_tf[slot[t]];
            }
        }
        e_ptr[i + 1] = (int64_t)e_term.size();
    }
    const int64_t P = (int64_t)e_term.size();

    mir_bm25 *h = new (std::nothrow) mir_bm25();
    MIR_REQUIRE(h != nullptr, "out of host memory");
    h->device = device; h->n_docs = n_docs; h->vocab = vocab; h->n_postings = P; h->doc_offset = doc_offset;
    h->ntiles = (int)std::max<int64_t>(1, (n_docs + kBm25Tile - 1) / kBm25Tile);
    h->h_idf.assign(vocab, 0.0);
    if (idf_override) {
        std::memcpy(h->h_idf.data(), idf_override, sizeof(double) * vocab);
        h->avgdl = avgdl_override;
    } else {
        h->avgdl = (double)total / (double)n_docs;
        // BM25Okapi._calc_idf
        double idf_sum = 0.0;
        std::vector<int32_t> negative;
        for (int32_t t : order) {
            const double v = std::log((double)(n_docs - df[t]) + 0.5) - std::log((double)df[t] + 0.5);
            h->h_idf[t] = v;
            idf_sum += v;
            if (v < 0) negative.push_back(t);
        }
        h->average_idf = idf_sum / (double)order.size();
        const double eps = epsilon * h->average_idf;
        for (int32_t t : negative) h->h_idf[t] = eps;
    }

    // ---- postings by term (documents ascending), weights in the package's operation order ----
    std::vector<int64_t> t_ptr(vocab + 1, 0);
    for (int t = 0; t < vocab; ++t) t_ptr[t + 1] = t_ptr[t] + df[t];
    std::vector<int64_t> fill(t_ptr.begin(), t_ptr.end() - 1);
    std::vector<int32_t> p_doc((size_t)P);
    std::vector<double> p_w((size_t)P);
    for (int64_t i = 0; i < n_docs; ++i) {
        const double dl = (double)(indptr[i + 1] - indptr[i]);
        const double denom_len = k1 * ((1.0 - b) + (b * dl) / h->avgdl);
        for (int64_t e = e_ptr[i]; e < e_ptr[i + 1]; ++e) {
            const double tf = (double)e_tf[e];
            const int64_t pos = fill[e_term[e]]++;
            p_doc[pos] = (int32_t)i;
            p_w[pos] = (tf * (k1 + 1.0)) / (tf + denom_len);
        }
    }
    // ---- per-term tile offsets ----
    const int T = h->ntiles;
    MIR_REQUIRE((int64_t)vocab * (T + 1) < ((int64_t)1 << 33), "vocab x tiles table too large (%d x %d)", vocab, T + 1);
    std::vector<uint32_t> t_tile((size_t)vocab * (T + 1));
    for (int t = 0; t < vocab; ++t) {
        uint32_t *row = t_tile.data() + (size_t)t * (T + 1);
        int64_t p = t_ptr[t];
        for (int j = 0; j <= T; ++j) {
            const int64_t bound = (int64_t)j * kBm25Tile;
            while (p < t_ptr[t + 1] && p_doc[p] < bound) ++p;
            row[j] = (uint32_t)(p - t_ptr[t]);
        }
    }

    auto fail = [&](int32_t code) {
        free_bm25(h);
        return code;
    };
#define MIR_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_error("%s failed: %s", #call, hipGetErrorString(e_));                              \
            return fail(MIR_ERR_HIP);                                                              \
        }                                                                                          \
    } while (0)
    auto up = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        h->hbm_bytes += bytes;
        return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    MIR_TRY(up((void **)&h->p_doc, p_doc.data(), (size_t)P * 4));
    MIR_TRY(up((void **)&h->p_w, p_w.data(), (size_t)P * 8));
    MIR_TRY(up((void **)&h->t_ptr, t_ptr.data(), (size_t)(vocab + 1) * 8));
    MIR_TRY(up((void **)&h->t_tile, t_tile.data(), t_tile.size() * 4));
    MIR_TRY(up((void **)&h->idf, h->h_idf.data(), (size_t)vocab * 8));
    MIR_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
#undef MIR_TRY
    *out = h;
    return MIR_OK;
}

int32_t mir_bm25_destroy(mir_bm25 *h) {
    free_bm25(h);
    return MIR_OK;
}

int32_t mir_bm25_info(const mir_bm25 *h, int64_t *n_docs, int32_t *vocab, int64_t *n_postings, double *avgdl,
                      double *average_idf, int64_t *hbm_bytes) {
    MIR_REQUIRE(h != nullptr, "handle is NULL");
    if (n_docs) *n_docs = h->n_docs;
    if (vocab) *vocab = h->vocab;
    if (n_postings) *n_postings = h->n_postings;
    if (avgdl) *avgdl = h->avgdl;
    if (average_idf) *average_idf = h->average_idf;
    if (hbm_bytes) *hbm_bytes = h->hbm_bytes;
    return MIR_OK;
}

int32_t mir_bm25_idf(const mir_bm25 *h, double *out_idf_host) {
    MIR_REQUIRE(h != nullptr && out_idf_host != nullptr, "NULL argument");
    std::memcpy(out_idf_host, h->h_idf.data(), sizeof(double) * h->vocab);
    return MIR_OK;
}

// Shared implementation: queries already on the device (q_terms[nt], q_ptr[b+1]).
static int32_t bm25_run(mir_bm25 *h, const int32_t *d_terms, const int32_t *d_ptr, int b, int k, double *d_scores,
                        int64_t *d_out_idx, double *d_out_score, int32_t *d_out_count, void *part, hipStream_t s) {
    const int T = h->ntiles;
    double *part_score = nullptr;
    int32_t *part_idx = nullptr, *part_cnt = nullptr;
    if (k > 0) {
        char *p = static_cast<char *>(part);
        part_score = reinterpret_cast<double *>(p);
        part_idx = reinterpret_cast<int32_t *>(p + (size_t)b * T * k * 8);
        part_cnt = reinterpret_cast<int32_t *>(p + (size_t)b * T * k * 12);
    }
    bm25_tile_kernel<<<dim3(T, b), dim3(256), 0, s>>>(dev_view(h), d_terms, d_ptr, k, d_scores, part_score, part_idx,
                                                      part_cnt);
    MIR_HIP(hipGetLastError());
    if (k > 0) {
        bm25_merge_kernel<<<dim3(b), dim3(256), 0, s>>>(part_score, part_idx, part_cnt, T, k, h->doc_offset,
                                                        d_out_idx, d_out_score, d_out_count);
        MIR_HIP(hipGetLastError());
    }
    return MIR_OK;
}

static size_t part_bytes(int b, int T, int k) { return (size_t)b * T * k * 12 + (size_t)b * T * 4 + 64; }

// BM25Okapi.get_scores(query) -> float64[n_docs] (bm25_retriever.py:83)
int32_t mir_bm25_scores(mir_bm25 *h, const int32_t *q_terms_host, int32_t nq, double *out_scores_host) {
    MIR_REQUIRE(h != nullptr, "handle is NULL");
    MIR_REQUIRE(nq >= 0 && (nq == 0 || q_terms_host), "bad query");
    if (h->n_docs == 0) return MIR_OK;
    MIR_REQUIRE(out_scores_host != nullptr, "out_scores is NULL");
    int32_t rc = use_device(h->device, nullptr);
    if (rc != MIR_OK) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t off_ptr = ((size_t)nq * 4 + 255) & ~(size_t)255;
    const size_t off_sc = off_ptr + 256;
    rc = ensure_scratch(h, off_sc + (size_t)h->n_docs * 8);
    if (rc != MIR_OK) return rc;
    char *base = static_cast<char *>(h->scratch);
    const int32_t ptr2[2] = {0, nq};
    if (nq) MIR_HIP(hipMemcpyAsync(base, q_terms_host, (size_t)nq * 4, hipMemcpyHostToDevice, h->stream));
    MIR_HIP(hipMemcpyAsync(base + off_ptr, ptr2, 8, hipMemcpyHostToDevice, h->stream));
    rc = bm25_run(h, reinterpret_cast<int32_t *>(base), reinterpret_cast<int32_t *>(base + off_ptr), 1, 0,
                  reinterpret_cast<double *>(base + off_sc), nullptr, nullptr, nullptr, nullptr, h->stream);
    if (rc != MIR_OK) { (void)hipStreamSynchronize(h->stream); return rc; }
    MIR_HIP(hipMemcpyAsync(out_scores_host, base + off_sc, (size_t)h->n_docs * 8, hipMemcpyDeviceToHost, h->stream));
    MIR_HIP(hipStreamSynchronize(h->stream));
    return MIR_OK;
}

// _get_top_n_indexes for b queries (bm25_retriever.py:81-84): q_ptr[b+1] slices q_terms.
// out_idx[b][k] (doc_offset + local index), out_score[b][k], out_count[b] = min(k, n_docs).
int32_t mir_bm25_search(mir_bm25 *h, const int32_t *q_terms_host, const int32_t *q_ptr_host, int32_t b, int32_t k,
                        int64_t *out_idx, double *out_score, int32_t *out_count) {
    MIR_REQUIRE(h != nullptr, "handle is NULL");
    MIR_REQUIRE(b >= 0 && k >= 1, "bad shape b=%d k=%d", b, k);
    if (b == 0) return MIR_OK;
    MIR_REQUIRE(q_ptr_host && out_idx && out_score && out_count, "NULL buffer");
    if (k > kBm25MaxK) {
        set_error("k=%d exceeds this build's BM25 top-k limit %d", k, kBm25MaxK);
        return MIR_ERR_UNSUPPORTED;
    }
    const int nt = q_ptr_host[b];
    MIR_REQUIRE(q_ptr_host[0] == 0 && nt >= 0 && (nt == 0 || q_terms_host), "bad q_ptr");
    for (int i = 0; i < b; ++i) MIR_REQUIRE(q_ptr_host[i + 1] >= q_ptr_host[i], "q_ptr not monotone");
    int32_t rc = use_device(h->device, nullptr);
    if (rc != MIR_OK) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_terms = take((size_t)nt * 4 + 4), o_ptr = take((size_t)(b + 1) * 4);
    const size_t o_idx = take((size_t)b * k * 8), o_sc = take((size_t)b * k * 8), o_cnt = take((size_t)b * 4);
    const size_t o_part = take(part_bytes(b, h->ntiles, k));
    rc = ensure_scratch(h, off);
    if (rc != MIR_OK) return rc;
    char *base = static_cast<char *>(h->scratch);
    hipStream_t s = h->stream;
    if (nt) MIR_HIP(hipMemcpyAsync(base + o_terms, q_terms_host, (size_t)nt * 4, hipMemcpyHostToDevice, s));
    MIR_HIP(hipMemcpyAsync(base + o_ptr, q_ptr_host, (size_t)(b + 1) * 4, hipMemcpyHostToDevice, s));
    rc = bm25_run(h, reinterpret_cast<int32_t *>(base + o_terms), reinterpret_cast<int32_t *>(base + o_ptr), b, k,
                  nullptr, reinterpret_cast<int64_t *>(base + o_idx), reinterpret_cast<double *>(base + o_sc),
                  reinterpret_cast<int32_t *>(base + o_cnt), base + o_part, s);
    if (rc != MIR_OK) { (void)hipStreamSynchronize(s); return rc; }
    MIR_HIP(hipMemcpyAsync(out_idx, base + o_idx, (size_t)b * k * 8, hipMemcpyDeviceToHost, s));
    MIR_HIP(hipMemcpyAsync(out_score, base + o_sc, (size_t)b * k * 8, hipMemcpyDeviceToHost, s));
    MIR_HIP(hipMemcpyAsync(out_count, base + o_cnt, (size_t)b * 4, hipMemcpyDeviceToHost, s));
    MIR_HIP(hipStreamSynchronize(s));
    return MIR_OK;
}

// Same with every buffer in HBM, asynchronous on `stream`.  `workspace` must hold
// mir_bm25_workspace_bytes(h, b, k) bytes and stay untouched until the stream has passed.
int64_t mir_bm25_workspace_bytes(const mir_bm25 *h, int32_t b, int32_t k) {
    if (!h || b < 0 || k < 1) return -1;
    return (int64_t)part_bytes(b, h->ntiles, k);
}

int32_t mir_bm25_search_device(mir_bm25 *h, const int32_t *q_terms_device, const int32_t *q_ptr_device, int32_t b,
                               int32_t k, int64_t *out_idx, double *out_score, int32_t *out_count,
                               void *workspace, void *stream) {
    MIR_REQUIRE(h != nullptr, "handle is NULL");
    MIR_REQUIRE(b >= 0 && k >= 1, "bad shape b=%d k=%d", b, k);
    if (b == 0) return MIR_OK;
    MIR_REQUIRE(q_ptr_device && out_idx && out_score && out_count && workspace, "NULL buffer");
    if (k > kBm25MaxK) {
        set_error("k=%d exceeds this build's BM25 top-k limit %d", k, kBm25MaxK);
        return MIR_ERR_UNSUPPORTED;
    }
    int32_t rc = use_device(h->device, nullptr);
    if (rc != MIR_OK) return rc;
    return bm25_run(h, q_terms_device, q_ptr_device, b, k, nullptr, out_idx, out_score, out_count, workspace,
                    static_cast<hipStream_t>(stream));
}

}  // extern "C"
