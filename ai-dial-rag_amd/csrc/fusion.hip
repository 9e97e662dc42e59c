// Weighted reciprocal-rank fusion of the retrievers' result lists (host code).
//
// Replaces langchain 0.3.21 EnsembleRetriever.weighted_reciprocal_rank as wired
// at aidial_rag/retrieval_chain.py:239-245 (weights all 1.0, c = 60): at most
// 4 lists x 7 items, so this is host logic, not a kernel.
// Restated in oracle/fusion.py: score[key] += w / (rank + c), rank from 1, over
// every list in order (an item repeated inside a list is credited again);
// unique keys in first-seen order over the chained lists; stable sort by score,
// descending.
#include <algorithm>
#include <cstdint>
#include <map>
#include <numeric>
#include <utility>
#include <vector>

#include "common.h"

extern "C" int32_t mir_rrf_fuse(const int64_t *keys, const int32_t *list_ptr, const double *weights, int32_t n_lists,
                                int32_t c, int64_t *out_keys, double *out_scores, int32_t *out_count) {
    MIR_REQUIRE(n_lists >= 0 && list_ptr && out_count, "bad arguments");
    MIR_REQUIRE(n_lists == 0 || weights, "weights is NULL");
    MIR_REQUIRE(list_ptr[0] == 0, "list_ptr[0] must be 0");
    for (int l = 0; l < n_lists; ++l) MIR_REQUIRE(list_ptr[l + 1] >= list_ptr[l], "list_ptr not monotone");
    const int total = list_ptr[n_lists];
    MIR_REQUIRE(total == 0 || (keys && out_keys && out_scores), "NULL buffer");
    typedef std::pair<int64_t, int64_t> Key;
    constexpr int kSmall = 64;  // the product's case is <= 4 lists x 7 items: no containers, a linear scan for the key
    if (total <= kSmall) {
        Key uniq[kSmall];
        double score[kSmall];
        int order[kSmall], nu = 0;
        for (int l = 0; l < n_lists; ++l) {
            for (int i = list_ptr[l]; i < list_ptr[l + 1]; ++i) {
                const Key k(keys[2 * i], keys[2 * i + 1]);
                int s = 0;
                while (s < nu && uniq[s] != k) ++s;
                if (s == nu) { uniq[nu] = k; score[nu] = 0.0; ++nu; }
                const int rank = i - list_ptr[l] + 1;
                score[s] += weights[l] / (double)(rank + c);
            }
        }
        for (int i = 0; i < nu; ++i) {  // stable insertion sort by score, descending
            int j = i;
            while (j > 0 && score[order[j - 1]] < score[i]) { order[j] = order[j - 1]; --j; }
            order[j] = i;
        }
        for (int r = 0; r < nu; ++r) {
            out_keys[2 * r] = uniq[order[r]].first;
            out_keys[2 * r + 1] = uniq[order[r]].second;
            out_scores[r] = score[order[r]];
        }
        *out_count = nu;
        return MIR_OK;
    }
    std::map<Key, int> slot;       // key -> position in first-seen order
    std::vector<Key> uniq;
    std::vector<double> score;
    for (int l = 0; l < n_lists; ++l) {
        for (int i = list_ptr[l]; i < list_ptr[l + 1]; ++i) {
            const Key k(keys[2 * i], keys[2 * i + 1]);
            auto it = slot.find(k);
            int s;
            if (it == slot.end()) {
                s = (int)uniq.size();
                slot.emplace(k, s);
                uniq.push_back(k);
                score.push_back(0.0);
            } else {
                s = it->second;
            }
            const int rank = i - list_ptr[l] + 1;
            score[s] += weights[l] / (double)(rank + c);
        }
    }
    std::vector<int> order(uniq.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return score[a] > score[b]; });
    for (size_t r = 0; r < order.size(); ++r) {
        out_keys[2 * r] = uniq[order[r]].first;
        out_keys[2 * r + 1] = uniq[order[r]].second;
        out_scores[r] = score[order[r]];
    }
    *out_count = (int32_t)order.size();
    return MIR_OK;
}

// b independent fusions in one call (a batch of queries, each with n_lists result lists): query q's lists are
// list_ptr[q * (n_lists + 1) .. +n_lists], offsets into `keys` relative to key_base[q]; outputs are [b][cap] with
// cap = the largest total of one query.  One ctypes crossing per batch instead of one per query (~5 us each).
extern "C" int32_t mir_rrf_fuse_batch(const int64_t *keys, const int64_t *key_base, const int32_t *list_ptr, const double *weights,
                                      int32_t n_lists, int32_t c, int32_t b, int32_t cap, int64_t *out_keys, double *out_scores,
                                      int32_t *out_count) {
    MIR_REQUIRE(b >= 0 && cap >= 0 && n_lists >= 0, "bad shape");
    MIR_REQUIRE(b == 0 || (key_base && list_ptr && out_count), "NULL buffer");
    for (int32_t q = 0; q < b; ++q) {
        const int32_t *lp = list_ptr + (size_t)q * (n_lists + 1);
        MIR_REQUIRE(lp[n_lists] <= cap, "query %d has %d items, cap is %d", q, lp[n_lists], cap);
        int32_t rc = mir_rrf_fuse(keys ? keys + 2 * key_base[q] : nullptr, lp, weights, n_lists, c,
                                  out_keys ? out_keys + (size_t)q * cap * 2 : nullptr,
                                  out_scores ? out_scores + (size_t)q * cap : nullptr, out_count + q);
        if (rc != MIR_OK) return rc;
    }
    return MIR_OK;
}
