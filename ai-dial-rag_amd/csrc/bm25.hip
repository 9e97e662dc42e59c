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
// The layout is built on the device from the token-id documents (bm25_build.hip); the idf on the host.
// Scoring never materialises the dense float64[N] score vector the reference
// builds per query.  A workgroup owns one tile of 8192 consecutive documents:
// it accumulates the tile's scores in LDS, one query term after the other in
// query order (deterministic rounding, a document occurs at most once per term
// so no atomics on scores), then selects the tile's top-k from LDS.  HBM traffic
// per query is the postings of its terms (12 B each), read once, coalesced.
// The kernels of a search, five dispatches (seven until late in round 3):
//   bm25_plan_kernel    routes every query: light (one wave per (tile, query) pair) or heavy (the tile kernel)
//   bm25_wave_kernel    the light queries' fast pass: every touched document with its score into the query's pool slice
//   bm25_sparse_kernel  the heavy queries' fast pass: several queries per workgroup, software-pipelined; ranks only the
//                       positive touched documents of a tile
//   bm25_finish_kernel  per query: the light query's selection over its pool slice / the heavy query's merge of its tiles;
//                       flags the queries short of k positive documents
//   bm25_tile_kernel    the exact dense pass (all 8192 documents of a tile, zeros and negatives included) for the flagged
//                       queries, whose last tile merges them; also get_scores
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "bm25_build.h"
#include <cstdint>
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
        // the winner's owner consumes it; every thread whose cached best is the winner (or another
        // posting of the same document) rescans once the consumption is visible
        const bool stale = bp >= 0 && gi == bi;
        if (gp == bp) take(gp);
        __syncthreads();
        if (stale) local_best(bs, bi, bp);
    }
}

// Block-wide exact top-k without rounds (a round of block_select is a chain of ~30 dependent
// shuffles; an exact float64+index comparison loop costs ~50 cycles per candidate pair).  Three
// cheap stages cut the candidates down before any exact comparison:
//   A  (n > 256) the k-th best float32 key of a 256-candidate sample bounds the answer from below;
//   B  candidates whose float32 key reaches that bound go to an LDS list (expected ~ k*n/256);
//   C  a list entry is a finalist if fewer than k list keys are strictly greater than its own -
//      float32 rounding is monotone, so every true top-k entry is a finalist;
//   D  finalists (about k of them) are ranked exactly: (score desc, index desc).
// emit(rank, score, index) runs on the owning thread.  List overflows (adversarial order, or a mass
// of equal keys) go to `fallback`.  Returns the number of results (uniform).
constexpr int kTopkList = 512;
constexpr int kTopkDirect = 256;
constexpr int kTopkFinal = 64;

struct __align__(16) TopkLds {
    float kf[kTopkList + 4];
    double cs[kTopkList];
    double fs[kTopkFinal];
    int ci[kTopkList];
    int fi[kTopkFinal];
    int cnt, fcnt;
    float thr;
};

// wave-aggregated append: one LDS atomic per wave instead of one per lane (all lanes must call)
__device__ __forceinline__ int wave_append(bool want, int *counter) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    int start = 0;
    if (lane == leader) start = atomicAdd(counter, __popcll(mask));
    start = __shfl(start, leader, 64);
    return want ? start + __popcll(mask & ((1ull << lane) - 1ull)) : -1;
}

// number of keys in kf[0, m4) strictly greater than (gt) / not less than (ge) `key`; m4 % 4 == 0
__device__ __forceinline__ void count_keys(const float *kf, int m4, float key, int &gt, int &ge) {
    gt = 0;
    ge = 0;
    for (int u = 0; u < m4; u += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(kf + u);
        gt += (v.x > key) + (v.y > key) + (v.z > key) + (v.w > key);
        ge += (v.x >= key) + (v.y >= key) + (v.z >= key) + (v.w >= key);
    }
}
__device__ __forceinline__ int count_greater(const float *kf, int m4, float key) {
    int gt = 0;
    for (int u = 0; u < m4; u += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(kf + u);
        gt += (v.x > key) + (v.y > key) + (v.z > key) + (v.w > key);
    }
    return gt;
}

template <typename ScoreF, typename IndexF, typename EmitF, typename FallbackF>
__device__ inline int block_topk(int n, int k, int tid, ScoreF score, IndexF index, EmitF emit, FallbackF fallback,
                                 TopkLds &L) {
    const double NEG = -__builtin_inf();
    const float NEGF = -__builtin_inff();
    int m, result;
    if (n <= kTopkDirect) {  // the candidates themselves are the list, one per thread
        double s = NEG;
        int ix = -1;
        if (tid < n) { s = score(tid); ix = (int)index(tid); }
        L.cs[tid] = s;
        L.ci[tid] = ix;
        L.kf[tid] = (float)s;
        if (tid == 0) L.fcnt = 0;
        const int nvalid = __syncthreads_count(s != NEG);
        m = n;
        result = k < nvalid ? k : nvalid;
    } else {
        {   // A: bound from the first 256 candidates
            const double s = score(tid);
            const float key = (float)s;
            L.kf[tid] = key;
            if (tid == 0) { L.thr = NEGF; L.cnt = 0; L.fcnt = 0; }
            __syncthreads();
            if (s != NEG) {
                int gt, ge;
                count_keys(L.kf, 256, key, gt, ge);
                if (gt < k && k <= ge) L.thr = key;  // the tie group holding rank k (all its members write the same value)
            }
            __syncthreads();
        }
        const float thr = L.thr;
        __syncthreads();  // everyone has read the sample keys and the bound before the list overwrites them
        for (int i0 = 0; i0 < n; i0 += 256) {  // B
            const int i = i0 + tid;
            double s = NEG;
            int ix = -1;
            if (i < n) { s = score(i); ix = (int)index(i); }
            const float key = (float)s;
            const bool want = s != NEG && key >= thr;
            const int slot = wave_append(want, &L.cnt);
            if (want && slot < kTopkList) { L.cs[slot] = s; L.ci[slot] = ix; L.kf[slot] = key; }
        }
        __syncthreads();
        m = L.cnt;
        __syncthreads();  // all threads hold m before a fallback may reuse the counters
        if (m > kTopkList) return fallback();
        if (tid < 4) L.kf[m + tid] = NEGF;  // pad to a multiple of 4
        __syncthreads();
        result = k < m ? k : m;
    }
    const int m4 = (m + 3) & ~3;
    for (int e0 = 0; e0 < m; e0 += 256) {  // C
        const int e = e0 + tid;
        bool fin = false;
        if (e < m && L.cs[e] != NEG) fin = count_greater(L.kf, m4, L.kf[e]) < k;
        const int slot = wave_append(fin, &L.fcnt);
        if (fin && slot < kTopkFinal) { L.fs[slot] = L.cs[e]; L.fi[slot] = L.ci[e]; }
    }
    __syncthreads();
    const int f = L.fcnt;
    __syncthreads();
    if (f > kTopkFinal) return fallback();
    if (tid < f) {  // D
        const double s = L.fs[tid];
        const int ix = L.fi[tid];
        int rank = 0;
        for (int u = 0; u < f; ++u) rank += bm25_before(L.fs[u], (int64_t)L.fi[u], s, (int64_t)ix) ? 1 : 0;
        if (rank < k) emit(rank, s, (int64_t)ix);
    }
    return result;
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
//
// The tile's work is a handful of dependent HBM hops, so the kernel is written to keep the chain
// short rather than to save instructions: the metadata of up to 64 query terms is fetched by 64
// lanes at once, then every thread fetches its share of ALL those terms' postings in one go
// (registers), and only the LDS adds are serialised term by term - the order rank-bm25 adds in.
//
// This is the DENSE pass: it ranks all documents of the tile, zeros and negatives included, exactly
// as `argsort(stable)[::-1]` does.  With `need_dense` it is the fallback behind bm25_sparse_kernel
// (workgroups of queries whose word is 0 return at once); with `out_scores` it is get_scores.
constexpr int kBm25Chunk = 64;     // query terms resolved per metadata round
constexpr int kBm25Regs = 4;       // postings a thread holds per accumulate round (256*4 per round)
constexpr int kBm25Cand = 1024;    // distinct touched documents listed per tile (sparse pass)

// the dense fallback's own merge (bm25_tile_kernel's last-arriving workgroup per query); arrive == nullptr: none
struct DenseMerge {
    uint32_t *arrive;     // [b], zero at launch
    int64_t doc_offset, n_docs;
    int32_t *need_dense;
    int64_t *out_idx;
    double *out_score;
    int32_t *out_count;
};
__device__ void bm25_merge_tail(double *part_score, const int32_t *part_idx, const int32_t *part_cnt, int ntiles, int k,
                                const DenseMerge &dm, int q);

__global__ __launch_bounds__(256) void bm25_tile_kernel(Bm25Dev m, const int32_t *__restrict__ q_terms,
                                                        const int32_t *__restrict__ q_ptr, int k,
                                                        const int32_t *__restrict__ need_dense,
                                                        double *__restrict__ out_scores,
                                                        double *__restrict__ part_score,
                                                        int32_t *__restrict__ part_idx,
                                                        int32_t *__restrict__ part_cnt, DenseMerge dm,
                                                        const int32_t *__restrict__ dense_list = nullptr,
                                                        const int32_t *__restrict__ dense_n = nullptr) {
    __shared__ double sc[kBm25Tile];
    __shared__ double red_s[4];
    __shared__ int64_t red_i[4];
    __shared__ int red_p[4];
    __shared__ int s_last;
    __shared__ int64_t m_lo[kBm25Chunk];
    __shared__ double m_idf[kBm25Chunk];
    __shared__ int m_off[kBm25Chunk + 1];
    __shared__ int m_cnt;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    // fallback launch: the listed queries only, blockIdx.y, + gridDim.y, ...; otherwise query blockIdx.y
    const int n_list = dense_list ? *dense_n : (int)gridDim.y;
    for (int li = blockIdx.y; li < n_list; li += gridDim.y) {
    const int q = dense_list ? dense_list[li] : li;
    if (li != (int)blockIdx.y) __syncthreads();  // (the previous query's use of the shared arrays is over)
    [&]() {
    if (need_dense && !need_dense[q]) return;
    const int base = tile * kBm25Tile;
    const int cnt = (int)((m.n_docs - base) < kBm25Tile ? (m.n_docs - base) : kBm25Tile);
    const int qb = q_ptr[q], qe = q_ptr[q + 1];
    for (int i = tid; i < kBm25Tile; i += 256) sc[i] = 0.0;
    for (int c0 = qb; c0 < qe; c0 += kBm25Chunk) {
        __syncthreads();  // the previous chunk's table is consumed (and sc / touched are zeroed)
        if (tid < 64) {   // wave 0, all lanes: one term each
            const int j = c0 + tid;
            bool valid = false;
            int64_t lo = 0; int n = 0; double w_idf = 0.0;
            if (j < qe) {
                const int t = q_terms[j];
                if (t >= 0 && t < m.vocab) {          // unknown term: `(doc.get(q) or 0)` everywhere
                    w_idf = m.idf[t];
                    const uint32_t *to = m.t_tile + (size_t)t * (m.ntiles + 1) + tile;
                    const uint32_t a = to[0], b = to[1];
                    lo = m.t_ptr[t] + a;
                    n = (int)(b - a);
                    valid = (w_idf != 0.0) && n > 0;  // `(self.idf.get(q) or 0)`: adds +-0
                }
            }
            const unsigned long long mask = __ballot(valid);
            const int pos = __popcll(mask & ((1ull << tid) - 1ull));
            const int nt = __popcll(mask);
            // exclusive prefix of n over the valid lanes, in lane (= query) order
            int incl = valid ? n : 0;
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (tid >= off) incl += o;
            }
            if (valid) { m_lo[pos] = lo; m_idf[pos] = w_idf; m_off[pos] = incl - n; }
            if (tid == 63) { m_off[nt] = incl; m_cnt = nt; }
        }
        __syncthreads();
        const int nt = m_cnt, total = m_off[nt];
        for (int r0 = 0; r0 < total; r0 += 256 * kBm25Regs) {
            int d[kBm25Regs], tj[kBm25Regs];
            double x[kBm25Regs];
            int jmin = nt, jmax = -1;
#pragma unroll
            for (int r = 0; r < kBm25Regs; ++r) {
                const int e = r0 + r * 256 + tid;
                tj[r] = -1; d[r] = 0; x[r] = 0.0;
                if (e < total) {
                    int j = 0;
                    while (m_off[j + 1] <= e) ++j;
                    const int64_t p = m_lo[j] + (e - m_off[j]);
                    d[r] = m.p_doc[p] - base;
                    x[r] = m_idf[j] * m.p_w[p];  // one rounding for the product ...
                    tj[r] = j;
                }
            }
            {   // terms present in this round (uniform): those overlapping [r0, r0 + 1024)
                int j = 0;
                while (m_off[j + 1] <= r0) ++j;
                jmin = j;
                const int last = (r0 + 256 * kBm25Regs < total ? r0 + 256 * kBm25Regs : total) - 1;
                while (m_off[j + 1] <= last) ++j;
                jmax = j;
            }
            for (int j = jmin; j <= jmax; ++j) {
#pragma unroll
                for (int r = 0; r < kBm25Regs; ++r) {
                    if (tj[r] == j) {
                        sc[d[r]] = sc[d[r]] + x[r];  // ... and one for the sum; a document occurs once per term
                    }
                }
                __syncthreads();  // term j's adds are complete before term j+1 touches the same documents
            }
        }
    }
    __syncthreads();
    if (out_scores) {
        double *o = out_scores + (size_t)q * m.n_docs + base;
        for (int i = tid; i < cnt; i += 256) o[i] = sc[i];
    }
    if (!part_score) return;
    const double NEG = -__builtin_inf();
    const size_t pb = ((size_t)q * m.ntiles + tile) * k;
    int32_t *my_cnt = part_cnt + (size_t)q * m.ntiles + tile;
    {
        for (int i = cnt + tid; i < kBm25Tile; i += 256) sc[i] = NEG;  // padding never selected
        __syncthreads();
        const int kout = k < cnt ? k : cnt;
        block_select(
            cnt, kout, tid, [&](int i) { return sc[i]; }, [&](int i) { return (int64_t)i; }, [&](int i) { sc[i] = NEG; },
            [&](int r, double v, int64_t i) {
                part_score[pb + r] = v;
                part_idx[pb + r] = (int32_t)(base + i);
            },
            red_s, red_i, red_p);
        if (tid == 0) *my_cnt = kout;
    }
    if (!dm.arrive) return;
    // the fallback launch merges its own tiles: the query's last workgroup to get here (its candidates and everybody else's
    // are visible behind the fences) does what a second dispatch of bm25_merge_kernel did
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&dm.arrive[q], 1u) == (uint32_t)(m.ntiles - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    bm25_merge_tail(part_score, part_idx, part_cnt, m.ntiles, k, dm, q);
    }();
    }
}

// ---- routing between the two fast passes (bm25_wave_kernel below, one wave per (tile, query); this tile kernel for the
// queries that do not fit it)
constexpr int kWvSlots = 768;              // distinct touched documents of a (tile, query) pair
constexpr int kWvHeavy = 512;              // a query averaging more postings per tile than this goes to the tile kernel
constexpr int kWvWaves = 4;                // waves per workgroup
constexpr int kWvCountStride = 32;         // a query's candidate counter has a 128-byte line of its own
struct __align__(16) WaveLds {
    double sc[kWvSlots];
    uint32_t bm[kBm25Tile / 32];
    uint16_t pre[kBm25Tile / 32];
};
// the light queries' candidates (every touched document with its score, all tiles): a query's postings bound them, so
// bm25_plan_kernel gives each light query exactly that many entries of one pool - and calls a query heavy once the pool is full
struct WavePool {
    double *score;     // [capacity]
    int32_t *doc;      // [capacity] local document index
    uint32_t *count;   // [b][kWvCountStride] candidates written so far (bit 31: a tile overflowed its slots)
    int32_t *light;    // [b] 1: the query is the wave kernel's
    uint32_t *off;     // [b] its first pool entry
    int32_t *hlist;    // [b] the heavy queries, for bm25_sparse_kernel; hlist[b] = their number
    uint32_t *arrive;  // [b] dense pass: tiles of the query that have written their candidates (the last one merges them)
    int32_t *dense_list;  // [b] the queries the dense pass has to run (bm25_finish_kernel appends), dense_n[0] of them
    int32_t *dense_n;
    long long capacity;
};
__host__ __device__ inline long long wave_pool_capacity(int b, int ntiles) {
    const long long one = (long long)kWvHeavy * ntiles;  // the largest light query
    const long long per = one < 16384 ? one : 16384;
    const long long c = (long long)b * per;
    return c > one ? c : one;
}

// a query's postings (the sum of its terms' document frequencies) if the wave kernel can take it - at most 64 terms and at
// most kWvHeavy postings per tile on average - else -1
__device__ __forceinline__ long long bm25_query_need(const Bm25Dev &m, const int32_t *__restrict__ q_terms, const int32_t *__restrict__ q_ptr, int q) {
    const int qb = q_ptr[q], len = q_ptr[q + 1] - qb;
    if (len > kBm25Chunk) return -1;
    long long sdf = 0;
    for (int j0 = 0; j0 < len; j0 += 8) {  // eight terms' two dependent loads in flight at once (one by one: 59 us at b = 4096)
        int t[8];
        int64_t a[8], e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = j0 + u < len ? q_terms[qb + j0 + u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = t[u] >= 0 && t[u] < m.vocab;
            a[u] = ok ? m.t_ptr[t[u]] : 0;
            e[u] = ok ? m.t_ptr[t[u] + 1] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sdf += e[u] - a[u];
    }
    return sdf <= (long long)kWvHeavy * m.ntiles ? sdf : -1;
}
// Large batches (b > 1024): the three dependent loads per query (its slice, its terms, their posting ranges) spread over the
// chip instead of waiting in ONE workgroup (59 us at b = 4096, 91 % of it waiting: profiles/r03_bm25_pmc.md); the need lands
// in pool.off (it fits 32 bits: <= kWvHeavy * tiles), "does not fit" in pool.light, and bm25_plan_kernel only scans.
__global__ __launch_bounds__(256) void bm25_plan_need_kernel(Bm25Dev m, const int32_t *__restrict__ q_terms, const int32_t *__restrict__ q_ptr,
                                                             int b, WavePool pool) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= b) return;
    const long long need = bm25_query_need(m, q_terms, q_ptr, q);
    pool.light[q] = need >= 0 ? 1 : 0;
    pool.off[q] = need >= 0 ? (uint32_t)need : 0u;
}

// One block of 1024 threads: which queries are light (at most 64 terms, at most kWvHeavy postings per tile on average, and
// room left in the candidate pool - in query order), where their candidates go, and their counters zeroed.
// PRE: the needs were computed by bm25_plan_need_kernel (pool.light / pool.off); each thread then takes b / 1024 CONSECUTIVE
// queries and the block scans once.
template <bool PRE>
__global__ __launch_bounds__(1024) void bm25_plan_kernel(Bm25Dev m, const int32_t *__restrict__ q_terms, const int32_t *__restrict__ q_ptr,
                                                         int b, WavePool pool) {
    __shared__ long long s_scan[1024];
    __shared__ long long s_base;
    __shared__ int s_nheavy;
    const int tid = threadIdx.x, nthreads = (int)blockDim.x;  // 1024, or 64 for b <= 64
    if (tid == 0) { s_base = 0; s_nheavy = 0; }
    __syncthreads();
    if (PRE) {
        // one round: thread t owns queries [t * per, (t + 1) * per): their needs summed, one block scan of the sums, then the
        // queries' own offsets in order (same `at`, same holes as the round-by-round form below)
        const int per = (b + 1023) / 1024, qa = tid * per, qe = min(b, qa + per);
        long long sum = 0;
        for (int q = qa; q < qe; ++q) sum += pool.light[q] ? (long long)pool.off[q] : 0;
        s_scan[tid] = sum;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const long long o = tid >= off ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += o;
            __syncthreads();
        }
        long long at = s_scan[tid] - sum;
        for (int q = qa; q < qe; ++q) {
            const bool fits = pool.light[q] != 0;
            const long long need = fits ? (long long)pool.off[q] : 0;
            const bool light = fits && at + need <= pool.capacity;
            pool.light[q] = light ? 1 : 0;
            pool.off[q] = light ? (uint32_t)at : 0u;
            pool.count[(size_t)q * kWvCountStride] = 0;
            pool.arrive[q] = 0;
            if (!light) pool.hlist[atomicAdd(&s_nheavy, 1)] = q;
            at += need;
        }
        __syncthreads();
        if (tid == 0) { pool.hlist[b] = s_nheavy; *pool.dense_n = 0; }
        return;
    }
    for (int q0 = 0; q0 < b; q0 += nthreads) {
        const int q = q0 + tid;
        long long need = 0;
        bool fits = false;
        if (q < b) {
            const long long nd = bm25_query_need(m, q_terms, q_ptr, q);
            if (nd >= 0) { need = nd; fits = true; }
        }
        // inclusive scan of `need` over the queries of this round: 1024 of them through LDS, or - the block is ONE wave for
        // batches of at most 64 queries - a wave scan (the LDS form's twenty block barriers were 5 us of a single query's 42)
        long long incl = need;
        if (nthreads == 64) {
            for (int off = 1; off < 64; off <<= 1) {
                const long long o = ((long long)__shfl_up((int)(incl >> 32), off, 64) << 32) | (unsigned int)__shfl_up((int)(unsigned int)incl, off, 64);
                if (tid >= off) incl += o;
            }
            s_scan[tid] = incl;
        } else {
            s_scan[tid] = need;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) {
                const long long o = tid >= off ? s_scan[tid - off] : 0;
                __syncthreads();
                s_scan[tid] += o;
                __syncthreads();
            }
            incl = s_scan[tid];
        }
        const long long at = s_base + incl - need;
        const bool light = fits && at + need <= pool.capacity;  // (a query that does not fit leaves a hole: harmless)
        if (q < b) {
            pool.light[q] = light ? 1 : 0;
            pool.off[q] = light ? (uint32_t)at : 0u;
            pool.count[(size_t)q * kWvCountStride] = 0;
            pool.arrive[q] = 0;
            if (!light) pool.hlist[atomicAdd(&s_nheavy, 1)] = q;  // (any order: a heavy query's tiles are merged by bm25_merge_kernel)
        }
        __syncthreads();
        if (tid == nthreads - 1) s_base += s_scan[nthreads - 1];
        __syncthreads();
    }
    if (tid == 0) { pool.hlist[b] = s_nheavy; *pool.dense_n = 0; }
}

// Fast pass.  grid = (ntiles, ceil(b / qc)), block = 256: a workgroup owns one tile and walks `qc`
// queries through it, software-pipelined - while query q is accumulated and ranked in LDS, the
// postings of q+1, the term metadata of q+2 and the term ids of q+3 are in flight, so the
// three dependent HBM hops of a query cost nothing after the first.  Between queries only the
// touched score slots are re-zeroed.  Emits, per (query, tile), the top-k among the POSITIVE
// touched documents and their count; bm25_merge_kernel flags the queries that need the dense pass.
constexpr int kBm25QcMax = 64;

__device__ __forceinline__ void bm25_sparse_body(const Bm25Dev &m, const int32_t *__restrict__ q_terms,
                                                 const int32_t *__restrict__ q_ptr, int b, int qc, int k,
                                                 const int32_t *__restrict__ hlist, double *__restrict__ part_score,
                                                 int32_t *__restrict__ part_idx, int32_t *__restrict__ part_cnt, int block_x,
                                                 int block_y) {
    __shared__ double sc[kBm25Tile];
    __shared__ double red_s[4];
    __shared__ int64_t red_i[4];
    __shared__ int red_p[4];
    __shared__ int64_t tb_lo[2][kBm25Chunk];
    __shared__ double tb_idf[2][kBm25Chunk];
    __shared__ int tb_off[2][kBm25Chunk + 1];
    __shared__ int tb_cnt[2];
    __shared__ int tb_mine[2];  // the query is this kernel's (heavy); the others are bm25_wave_kernel's
    __shared__ int s_qptr[kBm25QcMax + 1], s_qlen[kBm25QcMax], s_qid[kBm25QcMax];
    __shared__ uint32_t touched[kBm25Tile / 32];
    __shared__ uint16_t cand[kBm25Cand];
    __shared__ int ncand;
    __shared__ TopkLds L;
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const int tile = block_x;
    // this workgroup walks heavy queries hlist[q0 .. q0 + nq) (bm25_plan_kernel; the light ones are bm25_wave_kernel's)
    const int nheavy = hlist[b];
    const int q0 = block_y * qc;
    if (q0 >= nheavy) return;
    const int nq = (nheavy - q0) < qc ? (nheavy - q0) : qc;
    const int base = tile * kBm25Tile;
    const int cnt = (int)((m.n_docs - base) < kBm25Tile ? (m.n_docs - base) : kBm25Tile);
    const double NEG = -__builtin_inf();
    for (int i = tid; i < kBm25Tile; i += 256) sc[i] = 0.0;
    touched[tid] = 0;
    if (tid == 0) ncand = 0;
    if (tid < nq) {
        const int qid = hlist[q0 + tid];
        s_qid[tid] = qid;
        s_qptr[tid] = q_ptr[qid];
        s_qlen[tid] = q_ptr[qid + 1] - q_ptr[qid];
    }
    __syncthreads();

    // --- wave 0: one query term per lane -------------------------------------------------
    int t_reg = -1, t_len = 0;            // term id, query length (hop 1)
    bool m_ok = false;                    // metadata registers hold a real term (hop 2)
    int m_len = 0;
    double r_idf = 0.0; uint32_t r_a = 0, r_b = 0; int64_t r_tp = 0;
    auto load_term = [&](int qi) {
        t_reg = -1;
        t_len = 0;  // (1 = the query is this kernel's)
        if (tid < 64 && qi < nq) {
            const int qb = s_qptr[qi], len = s_qlen[qi];
            t_len = 1;
            if (len <= kBm25Chunk && tid < len) t_reg = q_terms[qb + tid];  // longer queries: nothing here, dense pass
        }
    };
    auto load_meta = [&]() {
        m_ok = false;
        m_len = t_len;
        if (tid < 64 && t_reg >= 0 && t_reg < m.vocab) {
            const uint32_t *to = m.t_tile + (size_t)t_reg * (m.ntiles + 1) + tile;
            r_idf = m.idf[t_reg];
            r_a = to[0];
            r_b = to[1];
            r_tp = m.t_ptr[t_reg];
            m_ok = true;
        }
    };
    auto build_table = [&](int buf) {
        if (tid < 64) {
            if (tid == 0) tb_mine[buf] = m_len;
            const int n = m_ok ? (int)(r_b - r_a) : 0;
            const bool valid = m_ok && r_idf != 0.0 && n > 0;
            const unsigned long long mask = __ballot(valid);
            const int pos = __popcll(mask & ((1ull << tid) - 1ull));
            const int nt = __popcll(mask);
            int incl = valid ? n : 0;
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (tid >= off) incl += o;
            }
            if (valid) { tb_lo[buf][pos] = r_tp + r_a; tb_idf[buf][pos] = r_idf; tb_off[buf][pos] = incl - n; }
            if (tid == 63) { tb_off[buf][nt] = incl; tb_cnt[buf] = nt; }
        }
    };
    // --- all threads: up to 1024 postings of the next query held raw in registers ---------
    int d_raw[kBm25Regs], tj[kBm25Regs];
    double w_raw[kBm25Regs];
    auto load_postings = [&](int buf) {
        const int total = tb_off[buf][tb_cnt[buf]];
#pragma unroll
        for (int r = 0; r < kBm25Regs; ++r) {
            const int e = r * 256 + tid;
            tj[r] = -1;
            if (e < total) {
                int j = 0;
                while (tb_off[buf][j + 1] <= e) ++j;
                const int64_t p = tb_lo[buf][j] + (e - tb_off[buf][j]);
                d_raw[r] = m.p_doc[p];
                w_raw[r] = m.p_w[p];
                tj[r] = j;
            }
        }
    };
    auto add = [&](int di, double x) {
        sc[di] = sc[di] + x;  // product rounded, then the sum; a document occurs once per term
        const uint32_t bit = 1u << (di & 31);
        const uint32_t old = atomicOr(&touched[di >> 5], bit);
        if (!(old & bit)) {
            const int slot = atomicAdd(&ncand, 1);
            if (slot < kBm25Cand) cand[slot] = (uint16_t)di;
        }
    };

    // prologue: fill the pipeline
    load_term(0);
    load_meta();
    build_table(0);
    load_term(1);
    __syncthreads();
    load_postings(0);
    load_meta();
    load_term(2);

    for (int qi = 0; qi < nq; ++qi) {
        const int cur = qi & 1;
        // ---- accumulate query qi ----
        const int nt = tb_cnt[cur], total = tb_off[cur][nt];
        if (total > 0) {
            const int last = (total < 256 * kBm25Regs ? total : 256 * kBm25Regs) - 1;
            int jmax = 0;
            while (tb_off[cur][jmax + 1] <= last) ++jmax;
            for (int j = 0; j <= jmax; ++j) {
#pragma unroll
                for (int r = 0; r < kBm25Regs; ++r)
                    if (tj[r] == j) add(d_raw[r] - base, tb_idf[cur][j] * w_raw[r]);
                __syncthreads();  // term j complete before term j+1 touches the same documents
            }
            for (int r0 = 256 * kBm25Regs; r0 < total; r0 += 256) {  // the rare long tail, unpipelined
                const int e = r0 + tid;
                int j = -1, di = 0;
                double x = 0.0;
                if (e < total) {
                    j = 0;
                    while (tb_off[cur][j + 1] <= e) ++j;
                    const int64_t p = tb_lo[cur][j] + (e - tb_off[cur][j]);
                    di = m.p_doc[p] - base;
                    x = tb_idf[cur][j] * m.p_w[p];
                }
                int ja = 0;
                while (tb_off[cur][ja + 1] <= r0) ++ja;
                const int lastb = (r0 + 256 < total ? r0 + 256 : total) - 1;
                int jb = ja;
                while (tb_off[cur][jb + 1] <= lastb) ++jb;
                for (int jj = ja; jj <= jb; ++jj) {
                    if (j == jj) add(di, x);
                    __syncthreads();
                }
            }
        }
        // ---- keep the pipeline full: table of qi+1, then its postings, metadata of qi+2, terms of qi+3 ----
        build_table(cur ^ 1);
        __syncthreads();
        load_postings(cur ^ 1);
        load_meta();
        load_term(qi + 3);
        // ---- top-k of query qi among the touched positives (a light query: the wave kernel's, nothing to do) ----
        if (!tb_mine[cur]) {
            __syncthreads();  // (everybody has read tb_mine[cur] before the next iteration's build_table rewrites it)
            continue;
        }
        const int q = s_qid[qi];
        const size_t pb = ((size_t)q * m.ntiles + tile) * k;
        const int nc = ncand;
        auto emit = [&](int r, double v, int64_t i) {
            part_score[pb + r] = v;
            part_idx[pb + r] = (int32_t)(base + i);
        };
        auto pos = [&](int i) { const double v = sc[i]; return v > 0.0 ? v : NEG; };
        int got;
        if (nc <= kBm25Cand) {
            got = block_topk(
                nc, k, tid, [&](int i) { return pos(cand[i]); }, [&](int i) { return (int64_t)cand[i]; }, emit,
                [&]() {
                    if (tid == 0) s_cnt = 0;
                    block_select(
                        nc, k < nc ? k : nc, tid, [&](int i) { return pos(cand[i]); },
                        [&](int i) { return (int64_t)cand[i]; }, [&](int i) { sc[cand[i]] = NEG; },
                        [&](int r, double v, int64_t i) { emit(r, v, i); s_cnt = r + 1; }, red_s, red_i, red_p);
                    __syncthreads();
                    return s_cnt;
                },
                L);
        } else {  // more distinct documents than the list holds: the positives of the whole tile
            got = block_topk(
                cnt, k, tid, pos, [&](int i) { return (int64_t)i; }, emit,
                [&]() {
                    if (tid == 0) s_cnt = 0;
                    block_select(
                        cnt, k < cnt ? k : cnt, tid, pos, [&](int i) { return (int64_t)i; }, [&](int i) { sc[i] = NEG; },
                        [&](int r, double v, int64_t i) { emit(r, v, i); s_cnt = r + 1; }, red_s, red_i, red_p);
                    __syncthreads();
                    return s_cnt;
                },
                L);
        }
        if (tid == 0) part_cnt[(size_t)q * m.ntiles + tile] = got;
        // ---- reset only what query qi touched ----
        __syncthreads();
        if (nc <= kBm25Cand) {
            for (int i = tid; i < nc; i += 256) {
                const int c = cand[i];
                sc[c] = 0.0;
                touched[c >> 5] = 0;
            }
        } else {
            for (int i = tid; i < kBm25Tile; i += 256) sc[i] = 0.0;
            touched[tid] = 0;
        }
        if (tid == 0) ncand = 0;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fast pass at the right grain (round 3).  A (tile, query) pair of the SURVEY 8(d) workload touches 53 postings at the
// median and 143 on average; a 256-thread workgroup with a barrier per query term was parked at s_barrier / s_waitcnt
// 70 % of its cycles (profiles/r02_bm25_pmc.md).  Here ONE WAVE owns a pair and nothing in its way is a barrier:
//   1. lanes = query terms: term id, then idf / posting range of the tile / document frequency, then a wave scan;
//   2. the pair's postings into registers (PL per lane: 2 for the small pairs, 12 otherwise), every load in flight at once;
//   3. the touched documents as an 8192-bit map in LDS (atomic OR), popcount prefix per word: a document's SLOT is its
//      rank among the touched documents - slots ascend with the document index, no hashing, no collisions;
//   4. float64 scores per slot, the query's terms applied ONE AFTER THE OTHER in query order (a document occurs once per
//      term, so a term's adds never meet; LDS operations of one wave execute in order): rank-bm25's own summation order,
//      bit for bit;
//   5. every touched document with its score goes to the QUERY's candidate array (one atomic per pair reserves the
//      range); bm25_select_kernel then takes the query's top k over all tiles at once.  (A top-k per pair - ten wave
//      arg-max rounds - was two thirds of a pair's instructions, repeated 123 times per query.)
// Queries that do not fit (more than 64 terms, or more postings than kWvHeavy per tile on average) are left to
// bm25_sparse_kernel + bm25_merge_kernel, which skip all others (bm25_plan_kernel decides).  A pair
// whose touched documents outnumber the slots marks its query for the exact dense pass (its count word gets bit 31).
template <int PL>
__device__ __forceinline__ void wave_pair(const Bm25Dev &m, WaveLds &L, int lane, int nt, int total, int base, double c_idf, int c_lo_hi,
                                          int c_lo_lo, int c_off, const WavePool &pool, int q) {
    // term of posting e_i = r0 + 64 i + lane: the last term whose offset is <= e_i.  Offsets come through v_readlane
    // (u is wave-uniform), one compare per posting and term.
    int dl[PL], tj[PL];
    double w[PL];
    auto load_round = [&](int r0, bool want_w) {
#pragma unroll
        for (int i = 0; i < PL; ++i) tj[i] = 0;
        for (int u = 1; u < nt; ++u) {
            const int off_u = __builtin_amdgcn_readlane(c_off, u);
#pragma unroll
            for (int i = 0; i < PL; ++i) tj[i] += (r0 + i * 64 + lane >= off_u) ? 1 : 0;
        }
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int e = r0 + i * 64 + lane;
            const int j = tj[i];
            const int64_t lo_j = ((int64_t)__shfl(c_lo_hi, j, 64) << 32) | (uint32_t)__shfl(c_lo_lo, j, 64);
            const int off_j = __shfl(c_off, j, 64);
            dl[i] = 0; w[i] = 0.0;
            if (e < total) {
                const int64_t pp = lo_j + (e - off_j);
                dl[i] = m.p_doc[pp] - base;
                if (want_w) w[i] = m.p_w[pp];
            } else {
                tj[i] = -1;
            }
        }
    };
    const bool one_round = total <= 64 * PL;  // (nearly always: the postings stay in registers for both phases)
    // ---- 2./3. the touched documents
    for (int r0 = 0; r0 < total; r0 += 64 * PL) {
        load_round(r0, one_round);
#pragma unroll
        for (int i = 0; i < PL; ++i)
            if (tj[i] >= 0) atomicOr(&L.bm[dl[i] >> 5], 1u << (dl[i] & 31));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its LDS operations execute in order; this pins the compiler's)
    // slots: popcount prefix over the 256 map words (4 per lane)
    uint32_t w4[4];
    int ndist, run0;
    {
        int c = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { w4[u] = L.bm[4 * lane + u]; c += __popc(w4[u]); }
        int inc = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        ndist = __shfl(inc, 63, 64);
        run0 = inc - c;
        int run = run0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { L.pre[4 * lane + u] = (uint16_t)run; run += __popc(w4[u]); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t *cnt = pool.count + (size_t)q * kWvCountStride;
    if (ndist > kWvSlots) {  // (skewed tile: more touched documents than slots) -> the exact dense pass takes the query
        if (lane == 0) atomicOr(cnt, 0x80000000u);
#pragma unroll
        for (int u = 0; u < 4; ++u) L.bm[4 * lane + u] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
    }
    // ---- 4. scores: terms in query order
    for (int r0 = 0; r0 < total; r0 += 64 * PL) {
        if (!one_round) load_round(r0, true);
        const int jmin = __builtin_amdgcn_readfirstlane(tj[0]);  // lane 0's first posting of the round: the round's first term
        int jmx = -1;
#pragma unroll
        for (int i = 0; i < PL; ++i) jmx = tj[i] > jmx ? tj[i] : jmx;
        for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(jmx, off, 64); jmx = o > jmx ? o : jmx; }
        for (int j = jmin; j <= jmx; ++j) {
            const double idf_j = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(c_idf), j), __builtin_amdgcn_readlane(__double2loint(c_idf), j));
#pragma unroll
            for (int i = 0; i < PL; ++i) {
                if (tj[i] == j) {
                    const int wd = dl[i] >> 5;
                    const int slot = (int)L.pre[wd] + __popc(L.bm[wd] & ((1u << (dl[i] & 31)) - 1u));
                    L.sc[slot] = L.sc[slot] + idf_j * w[i];  // product rounded, then the sum: rank-bm25's order
                }
            }
            asm volatile("" ::: "memory");  // term j's adds are issued before term j + 1 reads the same slots
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- 5. every touched document and its score -> the query's candidates (slot order = document order)
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(cnt, (uint32_t)ndist) & 0x7fffffffu;
    at = __builtin_amdgcn_readfirstlane(at);
    double *cs = pool.score + (size_t)pool.off[q] + at;
    int32_t *cd = pool.doc + (size_t)pool.off[q] + at;
    {
        int slot = run0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t bits = w4[u];
            while (bits) {
                const int bit = __builtin_ctz(bits);
                bits &= bits - 1;
                cs[slot] = L.sc[slot];
                cd[slot] = base + (4 * lane + u) * 32 + bit;
                L.sc[slot] = 0.0;  // (reset as we go)
                ++slot;
            }
            L.bm[4 * lane + u] = 0;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void bm25_wave_body(const Bm25Dev &m, const int32_t *__restrict__ q_terms, const int32_t *__restrict__ q_ptr,
                                               int b, const WavePool &pool, int block_x, int grid_x) {
    __shared__ WaveLds lds[kWvWaves];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    WaveLds &L = lds[wave];
    for (int i = lane; i < kBm25Tile / 32; i += 64) L.bm[i] = 0;
    for (int i = lane; i < kWvSlots; i += 64) L.sc[i] = 0.0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long npairs = (long long)b * m.ntiles;
    const long long stride = (long long)grid_x * kWvWaves;
    for (long long p = (long long)block_x * kWvWaves + wave; p < npairs; p += stride) {
        const int q = (int)(p / m.ntiles), tile = (int)(p - (long long)q * m.ntiles);
        const int base = tile * kBm25Tile;
        const int qb = q_ptr[q], len = q_ptr[q + 1] - qb;
        // ---- 1. lanes = terms
        int t = -1;
        if (lane < len && len <= kBm25Chunk) t = q_terms[qb + lane];
        const bool known = t >= 0 && t < m.vocab;
        double idf = 0.0;
        int64_t lo = 0;
        int n = 0;
        if (known) {
            const uint32_t *to = m.t_tile + (size_t)t * (m.ntiles + 1) + tile;
            const int64_t tp = m.t_ptr[t];
            idf = m.idf[t];
            const uint32_t a = to[0], e = to[1];
            lo = tp + a;
            n = (int)(e - a);
        }
        if (!pool.light[q]) continue;  // bm25_sparse_kernel's (bm25_plan_kernel decided)
        const bool valid = known && idf != 0.0 && n > 0;   // `(self.idf.get(q) or 0)`: adds +-0
        const unsigned long long vmask = __ballot(valid);
        const int nt = __popcll(vmask);
        if (nt == 0) continue;
        // compact the valid terms to lanes 0 .. nt-1, in query order (a forward permutation: valid lane -> its rank among the
        // valid ones, the others behind them), with the exclusive prefix of their posting counts
        const int pos = __popcll(vmask & ((1ull << lane) - 1ull));
        int incl = valid ? n : 0;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const int total = __shfl(incl, 63, 64);
        const int tgt = (valid ? pos : nt + (lane - pos)) * 4;
        auto push = [&](int x) { return __builtin_amdgcn_ds_permute(tgt, x); };
        const double c_idf = __hiloint2double(push(__double2hiint(idf)), push(__double2loint(idf)));
        const int c_lo_hi = push((int)(lo >> 32)), c_lo_lo = push((int)(uint32_t)lo);
        const int c_off = push(incl - (valid ? n : 0));  // lane j < nt: offset of term j's first posting among the pair's
        if (total <= 128) wave_pair<2>(m, L, lane, nt, total, base, c_idf, c_lo_hi, c_lo_lo, c_off, pool, q);
        else wave_pair<12>(m, L, lane, nt, total, base, c_idf, c_lo_hi, c_lo_lo, c_off, pool, q);
    }
}

__global__ __launch_bounds__(64 * kWvWaves) void bm25_wave_kernel(Bm25Dev m, const int32_t *__restrict__ q_terms,
                                                                  const int32_t *__restrict__ q_ptr, int b, WavePool pool) {
    bm25_wave_body(m, q_terms, q_ptr, b, pool, (int)blockIdx.x, (int)gridDim.x);
}
__global__ __launch_bounds__(256) void bm25_sparse_kernel(Bm25Dev m, const int32_t *__restrict__ q_terms,
                                                          const int32_t *__restrict__ q_ptr, int b, int qc, int k,
                                                          const int32_t *__restrict__ hlist, double *__restrict__ part_score,
                                                          int32_t *__restrict__ part_idx, int32_t *__restrict__ part_cnt) {
    bm25_sparse_body(m, q_terms, q_ptr, b, qc, k, hlist, part_score, part_idx, part_cnt, (int)blockIdx.x, (int)blockIdx.y);
}
// A handful of queries (b <= 8, the live path's one): both fast passes in ONE dispatch - workgroups [0, wgs_wave) take the light
// queries' (tile, query) pairs, the others the heavy queries' tiles.  A dispatch that finds nothing to do is ~4.5 us of a
// single query's ~44; the occupancy this kernel's combined LDS costs does not matter at this size.
__global__ __launch_bounds__(256) void bm25_small_kernel(Bm25Dev m, const int32_t *__restrict__ q_terms, const int32_t *__restrict__ q_ptr,
                                                         int b, int qc, int k, WavePool pool, int wgs_wave, double *__restrict__ part_score,
                                                         int32_t *__restrict__ part_idx, int32_t *__restrict__ part_cnt) {
    const int bx = (int)blockIdx.x;
    if (bx < wgs_wave) {
        bm25_wave_body(m, q_terms, q_ptr, b, pool, bx, wgs_wave);
    } else {
        const int r = bx - wgs_wave;
        bm25_sparse_body(m, q_terms, q_ptr, b, qc, k, pool.hlist, part_score, part_idx, part_cnt, r % m.ntiles, r / m.ntiles);
    }
}

constexpr int kSelList = 2048;
// grid = b, block = 256: a light query's top k over the candidates of all its tiles (bm25_wave_kernel); flags it for the
// dense pass when fewer than k documents are positive or a tile overflowed.  Heavy queries: bm25_merge_kernel's.
__device__ __forceinline__ void bm25_select_body(WavePool pool, int k, int64_t doc_offset, int64_t n_docs,
                                                 int32_t *__restrict__ need_dense, int64_t *__restrict__ out_idx,
                                                 double *__restrict__ out_score, int32_t *__restrict__ out_count, TopkLds &L) {
    __shared__ double red_s[4];
    __shared__ int64_t red_i[4];
    __shared__ int red_p[4];
    __shared__ int s_cnt, s_m;
    __shared__ float s_thr;
    __shared__ __attribute__((aligned(16))) float s_key[1024];
    __shared__ double l_s[kSelList];
    __shared__ int l_i[kSelList];
    const int tid = threadIdx.x, q = blockIdx.x;
    const uint32_t word = pool.count[(size_t)q * kWvCountStride];
    const bool overflow = (word >> 31) != 0;
    const int n = (int)(word & 0x7fffffffu);
    double *cs = pool.score + (size_t)pool.off[q];
    const int32_t *cd = pool.doc + (size_t)pool.off[q];
    const double NEG = -__builtin_inf();
    auto emit = [&](int r, double sc, int64_t i) {
        out_score[(size_t)q * k + r] = sc;
        out_idx[(size_t)q * k + r] = doc_offset + i;
    };
    auto pos = [&](int i) { const double v = cs[i]; return v > 0.0 ? v : NEG; };
    auto whole = [&]() {  // block_topk over all n candidates: one dependent load per thread and 256 candidates
        return block_topk(
            n, k, tid, pos, [&](int i) { return (int64_t)cd[i]; }, emit,
            [&]() {
                if (tid == 0) s_cnt = 0;
                block_select(
                    n, k < n ? k : n, tid, pos, [&](int i) { return (int64_t)cd[i]; }, [&](int i) { cs[i] = NEG; },
                    [&](int r, double v, int64_t i) { emit(r, v, i); s_cnt = r + 1; }, red_s, red_i, red_p);
                __syncthreads();
                return s_cnt;
            },
            L);
    };
    int got;
    if (n <= 1024) {
        got = whole();
    } else {
        // more than a thousand candidates: first cut them down with EIGHT loads in flight per thread.  The k-th best float key
        // of 1024 candidates sampled across the array bounds the answer from below (~k n / 1024 candidates reach it); those go
        // to an LDS list, and block_topk runs on the list
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (int)(((long long)(tid + 256 * u) * n) >> 10);
            s_key[tid + 256 * u] = (float)pos(i);
        }
        if (tid == 0) s_m = 0;
        // the sample sorted descending (bitonic, 55 steps of two compare-exchanges per thread; counting, for each of a thread's
        // four keys, how many of the 1024 are greater cost six times as many instructions)
        for (int size = 2; size <= 1024; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t2 = tid + 256 * u;
                    const int lo = 2 * t2 - (t2 & (stride - 1)), hi = lo + stride;
                    const bool desc = (lo & size) == 0;
                    const float x = s_key[lo], y = s_key[hi];
                    if ((x < y) == desc) { s_key[lo] = y; s_key[hi] = x; }
                }
            }
        __syncthreads();
        if (tid == 0) s_thr = s_key[k - 1];  // -inf when fewer than k sampled candidates are positive: everything passes
        __syncthreads();
        const float thr = s_thr;
        for (int i0 = 0; i0 < n; i0 += 256 * 8) {
            double v[8];
            int ix[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256 + tid;
                v[u] = NEG; ix[u] = 0;
                if (i < n) { v[u] = cs[i]; ix[u] = cd[i]; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool want = v[u] > 0.0 && (float)v[u] >= thr;
                const int slot = wave_append(want, &s_m);
                if (want && slot < kSelList) { l_s[slot] = v[u]; l_i[slot] = ix[u]; }
            }
        }
        __syncthreads();
        const int mlist = s_m;
        __syncthreads();
        if (mlist > kSelList) {
            got = whole();  // (adversarial order or a mass of equal scores: the slow exact way)
        } else {
            got = block_topk(
                mlist, k, tid, [&](int i) { return l_s[i]; }, [&](int i) { return (int64_t)l_i[i]; }, emit,
                [&]() {
                    if (tid == 0) s_cnt = 0;
                    block_select(
                        mlist, k < mlist ? k : mlist, tid, [&](int i) { return l_s[i]; }, [&](int i) { return (int64_t)l_i[i]; },
                        [&](int i) { l_s[i] = NEG; }, [&](int r, double v2, int64_t i) { emit(r, v2, i); s_cnt = r + 1; }, red_s, red_i, red_p);
                    __syncthreads();
                    return s_cnt;
                },
                L);
        }
    }
    if (tid == 0) {
        out_count[q] = got;
        need_dense[q] = ((got < k && (int64_t)got < n_docs) || overflow) ? 1 : 0;
    }
}

// grid = b, block = 256: merge the tiles' candidates of one query.
// mode 0: merge the sparse pass and flag the queries that came up short (need_dense[q] = 1);
// mode 1: merge the dense pass, for the flagged queries only.
__device__ __forceinline__ void bm25_merge_body(double *__restrict__ part_score, const int32_t *__restrict__ part_idx,
                                                const int32_t *__restrict__ part_cnt, int ntiles, int k, int64_t doc_offset,
                                                int64_t n_docs, int mode, int q, int32_t *__restrict__ need_dense,
                                                int64_t *__restrict__ out_idx, double *__restrict__ out_score,
                                                int32_t *__restrict__ out_count, TopkLds &L) {
    __shared__ double red_s[4];
    __shared__ int64_t red_i[4];
    __shared__ int red_p[4];
    __shared__ int s_total;
    const int tid = threadIdx.x;
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
    auto emit = [&](int r, double s, int64_t i) {
        out_score[(size_t)q * k + r] = s;
        out_idx[(size_t)q * k + r] = doc_offset + i;
    };
    block_topk(
        ntiles * k, k, tid, [&](int e) { return ps[e]; }, [&](int e) { return (int64_t)pi[e]; }, emit,
        [&]() {
            block_select(
                ntiles * k, kout, tid, [&](int e) { return ps[e]; }, [&](int e) { return (int64_t)pi[e]; },
                [&](int e) { ps[e] = NEG; }, emit, red_s, red_i, red_p);
            return kout;
        },
        L);
    if (tid == 0) {
        out_count[q] = kout;
        if (mode == 0) need_dense[q] = (kout < k && (int64_t)kout < n_docs) ? 1 : 0;
    }
}

__device__ void bm25_merge_tail(double *part_score, const int32_t *part_idx, const int32_t *part_cnt, int ntiles, int k,
                                const DenseMerge &dm, int q) {
    __shared__ TopkLds L;
    bm25_merge_body(part_score, part_idx, part_cnt, ntiles, k, dm.doc_offset, dm.n_docs, 1, q, dm.need_dense, dm.out_idx, dm.out_score,
                    dm.out_count, L);
}

// grid = b, block = 256: the end of the fast passes, one dispatch for both kinds of query - a light query's selection over
// its candidates (bm25_wave_kernel's), a heavy query's merge of its tiles (bm25_sparse_kernel's).  (Two kernels until late
// in round 3, each returning at once for the other kind: a dispatch is ~5 us of a single query's ~45.)
__global__ __launch_bounds__(256) void bm25_finish_kernel(WavePool pool, double *__restrict__ part_score,
                                                          const int32_t *__restrict__ part_idx,
                                                          const int32_t *__restrict__ part_cnt, int ntiles, int k,
                                                          int64_t doc_offset, int64_t n_docs, int32_t *__restrict__ need_dense,
                                                          int64_t *__restrict__ out_idx, double *__restrict__ out_score,
                                                          int32_t *__restrict__ out_count) {
    __shared__ TopkLds L;  // (one for both kinds: the block's LDS decides how many queries a CU finishes at a time)
    const int q = blockIdx.x;
    if (pool.light[q]) bm25_select_body(pool, k, doc_offset, n_docs, need_dense, out_idx, out_score, out_count, L);
    else bm25_merge_body(part_score, part_idx, part_cnt, ntiles, k, doc_offset, n_docs, 0, q, need_dense, out_idx, out_score, out_count, L);
    // the dense pass walks a LIST of the queries that came up short (thread 0 wrote the flag above): a grid of (tiles x b)
    // workgroups that each read a flag and return was 200 us of the 4096-query step
    if (threadIdx.x == 0 && need_dense[q]) pool.dense_list[atomicAdd(pool.dense_n, 1)] = q;
}

// ---- any n (bm25_retriever.py:81-84 takes any n): beyond the 64 results the selection kernels hold, the dense score
// vectors (bm25_tile_kernel's get_scores form) are ranked in rounds of 64: round r admits only documents strictly after
// round r - 1's last result in the reference's order (score descending, then index DESCENDING).  One block of 16 waves per
// query; a wave walks documents wave, wave + 16, ... and keeps its best 64 in registers, one per lane, sorted.
constexpr int kDkThreads = 1024, kDkWaves = 16, kDkRound = 64;
__device__ __forceinline__ void bm25_wave_insert(double ns, int64_t ni, int kk, int lane, double &my_s, int64_t &my_i, int &cnt) {
    const unsigned long long before = __ballot(lane < cnt && bm25_before(my_s, my_i, ns, ni));
    const int pos = __popcll(before);
    if (pos >= kk) return;
    const double up_s = __shfl_up(my_s, 1, 64);
    const int64_t up_i = ((int64_t)__shfl_up((int)(my_i >> 32), 1, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)my_i, 1, 64);
    if (lane > pos) { my_s = up_s; my_i = up_i; }
    else if (lane == pos) { my_s = ns; my_i = ni; }
    cnt = cnt < kk ? cnt + 1 : kk;
}
__global__ __launch_bounds__(kDkThreads) void bm25_dense_topk_kernel(const double *__restrict__ scores, int64_t n_docs, int k, int round,
                                                                     int64_t doc_offset, int q0, double *__restrict__ bound_s,
                                                                     int64_t *__restrict__ bound_i, int64_t *__restrict__ out_idx,
                                                                     double *__restrict__ out_score, int32_t *__restrict__ out_count) {
    __shared__ double s_s[kDkThreads];
    __shared__ int64_t s_i[kDkThreads];
    __shared__ int s_c[kDkWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = blockIdx.x, q = q0 + ql;
    const double *sc = scores + (size_t)ql * n_docs;
    const int64_t found = k < n_docs ? k : n_docs;
    const int kk = (int)((found - (int64_t)kDkRound * round) < kDkRound ? (found - (int64_t)kDkRound * round) : kDkRound);
    if (kk <= 0) return;
    const bool bounded = round > 0;
    const double b_s = bounded ? bound_s[ql] : 0.0;
    const int64_t b_i = bounded ? bound_i[ql] : 0;
    double my_s = 0.0;
    int64_t my_i = 0;
    int cnt = 0;
    // 64 documents per wave step, one per lane; a step costs one vote unless a document enters the list
    for (int64_t d0 = (int64_t)wave * 64; d0 < n_docs; d0 += (int64_t)kDkWaves * 64) {
        const int64_t d = d0 + lane;
        const double x = d < n_docs ? sc[d] : 0.0;
        const double worst_s = __shfl(my_s, kk - 1, 64);
        const int64_t worst_i = ((int64_t)__shfl((int)(my_i >> 32), kk - 1, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)my_i, kk - 1, 64);
        bool ok = d < n_docs && (cnt < kk || bm25_before(x, d, worst_s, worst_i));
        if (bounded) ok = ok && bm25_before(b_s, b_i, x, d);
        unsigned long long mm = __ballot(ok);
        while (mm) {
            const int l = __builtin_ctzll(mm);
            mm &= mm - 1;
            bm25_wave_insert(__shfl(x, l, 64), d0 + l, kk, lane, my_s, my_i, cnt);
        }
    }
    // the 16 wave lists -> ranks
    s_s[tid] = my_s;
    s_i[tid] = my_i;
    if (lane == 0) s_c[wave] = cnt;
    __syncthreads();
    int total = 0, rank = 0;
    const bool valid = lane < cnt;
    for (int w = 0; w < kDkWaves; ++w) {
        const int c = s_c[w];
        total += c;
        if (valid)
            for (int l = 0; l < c; ++l) rank += bm25_before(s_s[w * 64 + l], s_i[w * 64 + l], my_s, my_i) ? 1 : 0;
    }
    const int kout = total < kk ? total : kk;
    if (valid && rank < kout) {
        const size_t o = (size_t)q * k + (size_t)kDkRound * round + rank;
        out_score[o] = my_s;
        out_idx[o] = doc_offset + my_i;
        if (rank == kout - 1) { bound_s[ql] = my_s; bound_i[ql] = my_i; }
    }
    if (tid == 0 && round == 0) out_count[q] = (int)found;
}

}  // namespace mir

using namespace mir;

struct mir_bm25 {
    int device = 0;
    int num_cus = 256;
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
    int32_t *p_tf = nullptr;      // term frequency per posting and tokens per document: what the weights derive from
    int32_t *d_doclen = nullptr;  // (mir_bm25_set_global_stats re-derives them for a sharded corpus's global avgdl)
    double k1 = 1.5, b = 0.75, epsilon = 0.25;
    int64_t total_tokens = 0;
    std::vector<int64_t> h_df, h_first;  // per term: documents containing it, position of its first token (INT64_MAX = absent)
    int64_t hbm_bytes = 0;
    std::mutex mu;  // serialises use of the scratch below (searches on one handle run one at a time)
    void *scratch = nullptr;
    size_t scratch_cap = 0;
    char *pin = nullptr;       // pinned host staging of mir_bm25_search: queries in, results out, one copy each
    size_t pin_cap = 0;
    hipStream_t stream = nullptr;
    int qc_pin = 0;  // mir_bm25_tune: queries per workgroup of the fast pass (0 = chosen per call)
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
    (void)hipFree(h->p_tf);
    (void)hipFree(h->d_doclen);
    (void)hipFree(h->scratch);
    if (h->pin) (void)hipHostFree(h->pin);
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

// BM25Okapi._calc_idf (rank-bm25 0.2.2) from corpus statistics: idf = ln(N - n + 0.5) - ln(n + 0.5) through libm (the
// package's math.log), summed in order of first appearance - the insertion order of the package's dict, which fixes the
// float64 rounding of the average - and terms with a negative idf floored to epsilon * average.  Used by
// mir_bm25_create on its own documents and, with all-reduced statistics, by every rank of a document-sharded corpus.
int32_t mir_bm25_idf_from_stats(const int64_t *df, const int64_t *first_pos, int32_t vocab, int64_t n_docs, double epsilon,
                                double *out_idf, double *out_average_idf) {
    MIR_REQUIRE(df && first_pos && out_idf && vocab >= 1 && n_docs >= 0, "bad argument");
    std::vector<int32_t> order;
    for (int32_t t = 0; t < vocab; ++t) {
        out_idf[t] = 0.0;
        MIR_REQUIRE(df[t] >= 0 && df[t] <= n_docs, "df[%d]=%lld outside [0, %lld]", t, (long long)df[t], (long long)n_docs);
        if (df[t] > 0) order.push_back(t);
    }
    std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return first_pos[x] < first_pos[y] || (first_pos[x] == first_pos[y] && x < y); });
    double idf_sum = 0.0;
    std::vector<int32_t> negative;
    for (int32_t t : order) {
        const double v = std::log((double)(n_docs - df[t]) + 0.5) - std::log((double)df[t] + 0.5);
        out_idf[t] = v;
        idf_sum += v;
        if (v < 0) negative.push_back(t);
    }
    const double avg = order.empty() ? 0.0 : idf_sum / (double)order.size();
    const double eps = epsilon * avg;
    for (int32_t t : negative) out_idf[t] = eps;
    if (out_average_idf) *out_average_idf = avg;
    return MIR_OK;
}

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
    for (int64_t i = 0; i < n_docs; ++i)
        MIR_REQUIRE(indptr[i + 1] >= indptr[i], "indptr is not monotone at %lld", (long long)i);
    int cus = 0;
    int32_t rc = use_device(device, &cus);
    if (rc != MIR_OK) return rc;

    mir_bm25 *h = new (std::nothrow) mir_bm25();
    MIR_REQUIRE(h != nullptr, "out of host memory");
    h->num_cus = cus;
    h->device = device; h->n_docs = n_docs; h->vocab = vocab; h->doc_offset = doc_offset;
    h->avgdl = idf_override ? avgdl_override : (double)total / (double)n_docs;

    // ---- postings, weights, per-term and per-tile offsets: built on the device (bm25_build.hip) ----
    Bm25Built built;
    rc = bm25_build_device(indptr, term_ids, n_docs, vocab, k1, b, h->avgdl, kBm25Tile, &built);
    h->p_doc = built.p_doc; h->p_w = built.p_w; h->t_ptr = built.t_ptr; h->t_tile = built.t_tile;  // freed with h
    h->p_tf = built.p_tf; h->d_doclen = built.doc_len; h->k1 = k1; h->b = b; h->epsilon = epsilon; h->total_tokens = total;
    h->n_postings = built.n_postings; h->ntiles = std::max(1, built.ntiles); h->hbm_bytes += built.hbm_bytes;
    if (rc != MIR_OK) {
        free_bm25(h);
        return rc;
    }

    // ---- BM25Okapi._calc_idf on the host: V logarithms, summed in order of first appearance (dict order) ----
    h->h_df.assign(vocab, 0);
    h->h_first.assign(vocab, INT64_MAX);
    for (int32_t t = 0; t < vocab; ++t) {
        h->h_df[t] = built.t_ptr_host[t + 1] - built.t_ptr_host[t];
        if (h->h_df[t] > 0) h->h_first[t] = (int64_t)built.first_pos[t];
    }
    h->h_idf.assign(vocab, 0.0);
    if (idf_override) {
        std::memcpy(h->h_idf.data(), idf_override, sizeof(double) * vocab);
    } else {
        rc = mir_bm25_idf_from_stats(h->h_df.data(), h->h_first.data(), vocab, n_docs, epsilon, h->h_idf.data(), &h->average_idf);
        if (rc != MIR_OK) {
            free_bm25(h);
            return rc;
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
    MIR_TRY(up((void **)&h->idf, h->h_idf.data(), (size_t)vocab * 8));
    MIR_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
#undef MIR_TRY
    *out = h;
    return MIR_OK;
}

// Host utility for callers that keep term ids in a vocabulary LARGER than one corpus (a process-wide one):
// rewrites ids[n] to 0..n_used-1 in order of first appearance and fills remap[vocab] (old id -> new id, -1 =
// not in this corpus).  The per-term tables of a model are sized by its vocab, so a corpus is compacted first.
int32_t mir_compact_term_ids(const int32_t *ids, int64_t n, int32_t vocab, int32_t *out_ids, int32_t *remap,
                             int32_t *n_used) {
    MIR_REQUIRE(vocab >= 0 && n >= 0 && remap != nullptr && n_used != nullptr, "bad argument");
    MIR_REQUIRE(n == 0 || (ids != nullptr && out_ids != nullptr), "NULL id buffer");
    for (int32_t t = 0; t < vocab; ++t) remap[t] = -1;
    int32_t next = 0;
    for (int64_t j = 0; j < n; ++j) {
        const int32_t t = ids[j];
        MIR_REQUIRE(t >= 0 && t < vocab, "term id %d at %lld outside [0, %d)", t, (long long)j, vocab);
        int32_t c = remap[t];
        if (c < 0) c = remap[t] = next++;
        out_ids[j] = c;
    }
    *n_used = next;
    return MIR_OK;
}

int32_t mir_bm25_corpus_stats(const mir_bm25 *h, int64_t *out_df, int64_t *out_first_pos, int64_t *out_total_tokens,
                              int64_t *out_n_docs) {
    MIR_REQUIRE(h != nullptr, "handle is NULL");
    if (out_df) std::memcpy(out_df, h->h_df.data(), sizeof(int64_t) * h->vocab);
    if (out_first_pos) std::memcpy(out_first_pos, h->h_first.data(), sizeof(int64_t) * h->vocab);
    if (out_total_tokens) *out_total_tokens = h->total_tokens;
    if (out_n_docs) *out_n_docs = h->n_docs;
    return MIR_OK;
}

int32_t mir_bm25_set_global_stats(mir_bm25 *h, const double *idf_host, double avgdl, double average_idf) {
    MIR_REQUIRE(h != nullptr && idf_host != nullptr, "NULL argument");
    MIR_REQUIRE(avgdl > 0.0, "avgdl=%g must be positive", avgdl);
    int32_t rc = use_device(h->device, nullptr);
    if (rc != MIR_OK) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    rc = bm25_reweight_device(h->p_doc, h->p_tf, h->d_doclen, h->n_postings, h->k1, h->b, avgdl, h->p_w, h->stream);
    if (rc != MIR_OK) return rc;
    MIR_HIP(hipMemcpyAsync(h->idf, idf_host, sizeof(double) * h->vocab, hipMemcpyHostToDevice, h->stream));
    MIR_HIP(hipStreamSynchronize(h->stream));
    std::memcpy(h->h_idf.data(), idf_host, sizeof(double) * h->vocab);
    h->avgdl = avgdl;
    h->average_idf = average_idf;
    return MIR_OK;
}

int32_t mir_bm25_tune(mir_bm25 *h, int32_t queries_per_workgroup) {
    MIR_REQUIRE(h != nullptr, "handle is NULL");
    MIR_REQUIRE(queries_per_workgroup >= 0 && queries_per_workgroup <= kBm25QcMax, "queries_per_workgroup=%d outside [0, %d]",
                queries_per_workgroup, kBm25QcMax);
    std::lock_guard<std::mutex> lk(h->mu);
    h->qc_pin = queries_per_workgroup;
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

static int32_t bm25_run_large_k(mir_bm25 *h, const int32_t *d_terms, const int32_t *d_ptr, int b, int k, int64_t *d_out_idx,
                                double *d_out_score, int32_t *d_out_count, void *ws, hipStream_t s);

// Shared implementation: queries already on the device (q_terms[nt], q_ptr[b+1]).
// part = [part_score f64 | part_idx i32 | part_cnt i32 | need_dense i32]
static int32_t bm25_run(mir_bm25 *h, const int32_t *d_terms, const int32_t *d_ptr, int b, int k, double *d_scores,
                        int64_t *d_out_idx, double *d_out_score, int32_t *d_out_count, void *part, hipStream_t s) {
    const int T = h->ntiles;
    if (k <= 0) {  // get_scores: dense score vector only
        bm25_tile_kernel<<<dim3(T, b), dim3(256), 0, s>>>(dev_view(h), d_terms, d_ptr, 0, nullptr, d_scores, nullptr,
                                                                 nullptr, nullptr, DenseMerge{});
        MIR_HIP(hipGetLastError());
        return MIR_OK;
    }
    if (k > kBm25MaxK) return bm25_run_large_k(h, d_terms, d_ptr, b, k, d_out_idx, d_out_score, d_out_count, part, s);
    char *p = static_cast<char *>(part);
    double *part_score = reinterpret_cast<double *>(p);
    int32_t *part_idx = reinterpret_cast<int32_t *>(p + (size_t)b * T * k * 8);
    int32_t *part_cnt = reinterpret_cast<int32_t *>(p + (size_t)b * T * k * 12);
    int32_t *need = part_cnt + (size_t)b * T;
    WavePool pool;
    pool.capacity = wave_pool_capacity(b, T);
    pool.light = need + b;
    pool.off = reinterpret_cast<uint32_t *>(pool.light + b);
    pool.hlist = reinterpret_cast<int32_t *>(pool.off + b);
    pool.arrive = reinterpret_cast<uint32_t *>(pool.hlist + b + 1);
    pool.dense_list = reinterpret_cast<int32_t *>(pool.arrive + b);
    pool.dense_n = pool.dense_list + b;
    pool.count = reinterpret_cast<uint32_t *>(pool.dense_n + 1);
    {
        size_t o = (size_t)b * T * k * 12 + (size_t)b * T * 4 + (size_t)b * 24 + 8 + (size_t)b * kWvCountStride * 4;
        o = (o + 255) & ~(size_t)255;
        pool.score = reinterpret_cast<double *>(p + o);
        pool.doc = reinterpret_cast<int32_t *>(p + o + (size_t)pool.capacity * 8);
    }
    // 1. fast passes: positives among touched documents.  Light queries (bm25_plan_kernel) one wave per (tile, query) and one
    //    selection per query; the others on the tile kernel + merge
    if (b > 1024) {
        bm25_plan_need_kernel<<<dim3((b + 255) / 256), dim3(256), 0, s>>>(dev_view(h), d_terms, d_ptr, b, pool);
        bm25_plan_kernel<true><<<dim3(1), dim3(1024), 0, s>>>(dev_view(h), d_terms, d_ptr, b, pool);
    } else {
        bm25_plan_kernel<false><<<dim3(1), dim3(b <= 64 ? 64 : 1024), 0, s>>>(dev_view(h), d_terms, d_ptr, b, pool);
    }
    MIR_HIP(hipGetLastError());
    //    tile kernel: queries per workgroup: as many as still leave ~8 workgroups per CU of parallelism
    int qc = (int)((int64_t)b * T / 2048);
    qc = qc < 1 ? 1 : (qc > kBm25QcMax ? kBm25QcMax : qc);
    if (h->qc_pin > 0) qc = h->qc_pin;  // tests pin the pipeline depth (short pipelines x long query queues)
    const long long npairs = (long long)b * T;
    const int wgs = (int)std::max<long long>(1, std::min<long long>((npairs + kWvWaves - 1) / kWvWaves, (long long)h->num_cus * 5));
    if (b <= 8 && h->qc_pin <= 0) {
        bm25_small_kernel<<<dim3(wgs + T * ((b + qc - 1) / qc)), dim3(256), 0, s>>>(dev_view(h), d_terms, d_ptr, b, qc, k, pool, wgs, part_score,
                                                                                  part_idx, part_cnt);
        MIR_HIP(hipGetLastError());
    } else {
        bm25_wave_kernel<<<dim3(wgs), dim3(64 * kWvWaves), 0, s>>>(dev_view(h), d_terms, d_ptr, b, pool);
        MIR_HIP(hipGetLastError());
        bm25_sparse_kernel<<<dim3(T, (b + qc - 1) / qc), dim3(256), 0, s>>>(dev_view(h), d_terms, d_ptr, b, qc, k, pool.hlist, part_score,
                                                                            part_idx, part_cnt);
        MIR_HIP(hipGetLastError());
    }
    bm25_finish_kernel<<<dim3(b), dim3(256), 0, s>>>(pool, part_score, part_idx, part_cnt, T, k, h->doc_offset, h->n_docs, need, d_out_idx,
                                                     d_out_score, d_out_count);
    MIR_HIP(hipGetLastError());
    // 2. exact dense pass for the queries with fewer than k positive documents (workgroups of the other queries exit at
    //    once; usually that is all of them); a query's last tile merges them
    DenseMerge dm;
    dm.arrive = pool.arrive; dm.doc_offset = h->doc_offset; dm.n_docs = h->n_docs; dm.need_dense = need;
    dm.out_idx = d_out_idx; dm.out_score = d_out_score; dm.out_count = d_out_count;
    bm25_tile_kernel<<<dim3(T, std::min(b, 16)), dim3(256), 0, s>>>(dev_view(h), d_terms, d_ptr, k, need, nullptr, part_score, part_idx, part_cnt, dm,
                                                                      pool.dense_list, pool.dense_n);
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

// [part_score | part_idx | part_cnt | need, light, off (b each), hlist (b + 1), arrive (b), dense_list (b), dense_n (1) | count (b x 32) | pool scores | pool documents]
static size_t part_bytes(int b, int T, int k) {
    size_t o = (size_t)b * T * k * 12 + (size_t)b * T * 4 + (size_t)b * 24 + 8 + (size_t)b * kWvCountStride * 4;
    o = (o + 255) & ~(size_t)255;
    return o + (size_t)wave_pool_capacity(b, T) * 12 + 64;
}

// n > 64: dense scores for chunks of queries, then rounds of 64.  Workspace: [scores chunk x n_docs f64 | bound_s | bound_i]
static int large_k_chunk(const mir_bm25 *h, int b) {
    const int64_t per = std::max<int64_t>(h->n_docs, 1) * 8;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, ((int64_t)1 << 30) / per));
}
static size_t large_k_bytes(const mir_bm25 *h, int b) {
    const int c = large_k_chunk(h, b);
    return (size_t)c * (size_t)std::max<int64_t>(h->n_docs, 1) * 8 + (size_t)c * 16 + 256;
}
static int32_t bm25_run_large_k(mir_bm25 *h, const int32_t *d_terms, const int32_t *d_ptr, int b, int k, int64_t *d_out_idx,
                                double *d_out_score, int32_t *d_out_count, void *ws, hipStream_t s) {
    const int chunk = large_k_chunk(h, b);
    double *scores = static_cast<double *>(ws);
    double *bound_s = scores + (size_t)chunk * std::max<int64_t>(h->n_docs, 1);
    int64_t *bound_i = reinterpret_cast<int64_t *>(bound_s + chunk);
    const int64_t found = std::min<int64_t>(k, h->n_docs);
    const int rounds = (int)std::max<int64_t>(1, (found + kDkRound - 1) / kDkRound);
    for (int q0 = 0; q0 < b; q0 += chunk) {
        const int nq = std::min(chunk, b - q0);
        // (the tile kernel reads q_ptr[q], q_ptr[q + 1] of query q = blockIdx.y: offset the ptr array, scores land at [0, nq))
        bm25_tile_kernel<<<dim3(h->ntiles, nq), dim3(256), 0, s>>>(dev_view(h), d_terms, d_ptr + q0, 0, nullptr, scores, nullptr, nullptr, nullptr,
                                                                   DenseMerge{});
        MIR_HIP(hipGetLastError());
        for (int r = 0; r < rounds; ++r) {
            bm25_dense_topk_kernel<<<dim3(nq), dim3(kDkThreads), 0, s>>>(scores, h->n_docs, k, r, h->doc_offset, q0, bound_s, bound_i, d_out_idx,
                                                                        d_out_score, d_out_count);
            MIR_HIP(hipGetLastError());
        }
    }
    return MIR_OK;
}

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
    const size_t o_part = take(k > kBm25MaxK ? large_k_bytes(h, b) : part_bytes(b, h->ntiles, k));
    rc = ensure_scratch(h, off);
    if (rc != MIR_OK) return rc;
    char *base = static_cast<char *>(h->scratch);
    hipStream_t s = h->stream;
    // pinned staging: [terms | ptr] go in with one copy, [idx | score | count] come back with one
    // (pageable buffers make every hipMemcpyAsync a synchronous staged copy of its own)
    const size_t in_bytes = o_idx;                 // o_terms = 0 .. end of the q_ptr slot
    const size_t out_bytes = o_part - o_idx;       // idx, score, count slots
    if (h->pin_cap < in_bytes + out_bytes) {
        if (h->pin) (void)hipHostFree(h->pin);
        h->pin = nullptr;
        h->pin_cap = 0;
        MIR_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin), in_bytes + out_bytes, hipHostMallocDefault));
        h->pin_cap = in_bytes + out_bytes;
    }
    if (nt) std::memcpy(h->pin + o_terms, q_terms_host, (size_t)nt * 4);
    std::memcpy(h->pin + o_ptr, q_ptr_host, (size_t)(b + 1) * 4);
    MIR_HIP(hipMemcpyAsync(base, h->pin, in_bytes, hipMemcpyHostToDevice, s));
    rc = bm25_run(h, reinterpret_cast<int32_t *>(base + o_terms), reinterpret_cast<int32_t *>(base + o_ptr), b, k,
                  nullptr, reinterpret_cast<int64_t *>(base + o_idx), reinterpret_cast<double *>(base + o_sc),
                  reinterpret_cast<int32_t *>(base + o_cnt), base + o_part, s);
    if (rc != MIR_OK) { (void)hipStreamSynchronize(s); return rc; }
    char *res = h->pin + in_bytes;
    MIR_HIP(hipMemcpyAsync(res, base + o_idx, out_bytes, hipMemcpyDeviceToHost, s));
    MIR_HIP(hipStreamSynchronize(s));
    std::memcpy(out_idx, res, (size_t)b * k * 8);
    std::memcpy(out_score, res + (o_sc - o_idx), (size_t)b * k * 8);
    std::memcpy(out_count, res + (o_cnt - o_idx), (size_t)b * 4);
    return MIR_OK;
}

// Same with every buffer in HBM, asynchronous on `stream`.  `workspace` must hold
// mir_bm25_workspace_bytes(h, b, k) bytes and stay untouched until the stream has passed.
int64_t mir_bm25_workspace_bytes(const mir_bm25 *h, int32_t b, int32_t k) {
    if (!h || b < 0 || k < 1) return -1;
    return (int64_t)(k > kBm25MaxK ? large_k_bytes(h, b) : part_bytes(b, h->ntiles, k));
}

int32_t mir_bm25_search_device(mir_bm25 *h, const int32_t *q_terms_device, const int32_t *q_ptr_device, int32_t b,
                               int32_t k, int64_t *out_idx, double *out_score, int32_t *out_count,
                               void *workspace, void *stream) {
    MIR_REQUIRE(h != nullptr, "handle is NULL");
    MIR_REQUIRE(b >= 0 && k >= 1, "bad shape b=%d k=%d", b, k);
    if (b == 0) return MIR_OK;
    MIR_REQUIRE(q_ptr_device && out_idx && out_score && out_count && workspace, "NULL buffer");
    int32_t rc = use_device(h->device, nullptr);
    if (rc != MIR_OK) return rc;
    return bm25_run(h, q_terms_device, q_ptr_device, b, k, nullptr, out_idx, out_score, out_count, workspace,
                    static_cast<hipStream_t>(stream));
}

}  // extern "C"
