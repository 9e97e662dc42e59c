/*
 * miretr.h - C ABI of libmiretr.so: the MI355X (gfx950) retrieval hot path of
 * epam/ai-dial-rag, built from scratch in HIP.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only: no torch, no
 * C++ types.  The reference is pure Python and dictates no FFI of its own
 * (SURVEY.md 8(b)); every entry point below names the reference interface it
 * replaces (paths relative to the upstream repo root), and INTEGRATION.md
 * shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns an int32 status (MIR_OK == 0) and never aborts;
 *     mir_last_error() returns a thread-local message for the last failure.
 *   - "host" pointers are ordinary process memory, borrowed for the call.
 *     "_device" entry points take HBM pointers valid on the index's device
 *     and enqueue on the given hipStream_t (passed as void*); they do not
 *     synchronise.
 *   - handles are opaque, thread-safe and re-entrant: concurrent searches on
 *     one handle are allowed (the reference calls `find` from several
 *     executor threads at once, semantic_retriever.py:54-56).
 *   - the library sets the device it needs per call (never relies on the
 *     ambient device).
 *   - there is NO CPU fallback: without a usable gfx950 device every compute
 *     entry point fails with MIR_ERR_NO_DEVICE.
 */
#ifndef MIRETR_H
#define MIRETR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIR_ABI_VERSION 4 /* 4: mir_keywords_preprocess / mir_kwp_result_*;
                            3: 2: results always exact (exact pass), any k; mir_bm25_tune / _corpus_stats / _idf_from_stats /
                             _set_global_stats, mir_rrf_fuse_batch, mir_wordpiece_* added
                             3: mir_index_scan_stats added; mir_bm25_search[_device] take any k (no MIR_ERR_UNSUPPORTED past 64);
                                mir_bm25_workspace_bytes grew (candidate pool of the wave-grain fast pass) */

/* status codes */
#define MIR_OK 0
#define MIR_ERR_INVALID 1     /* bad argument (maps to ValueError) */
#define MIR_ERR_HIP 2         /* HIP runtime failure */
#define MIR_ERR_NO_DEVICE 3   /* no usable GPU */
#define MIR_ERR_EMPTY 4       /* "Text index is empty." (bm25_retriever.py:75-76) */
#define MIR_ERR_UNSUPPORTED 5 /* valid request outside what this build handles */

/* aidial_rag/retrievers/embeddings_metrics.py:7-11 (Metric), same order as
 * the string values are listed there.  All metrics are "smaller is better". */
#define MIR_METRIC_COSINE_SIM 0
#define MIR_METRIC_EUCLIDEAN_DIST 1
#define MIR_METRIC_SQEUCLIDEAN_DIST 2
#define MIR_METRIC_INNER_PRODUCT 3

/* element type of the stored embedding matrix.  F16 rows are scored as the
 * float32 values they widen to, exactly (the reference up-casts fp16 storage the
 * same way).  With 256 < d <= 1024 (MultimodalRetriever's embeddings,
 * multimodal_retriever.py:96-153) the index stays 2 bytes per element in HBM;
 * other F16 shapes are widened to float32 on the device.  Any k is accepted on
 * every index (beyond the filter scan's candidate lists the exact pass answers). */
#define MIR_DTYPE_F32 0
#define MIR_DTYPE_F16 1

/* per-query result flags (out_flags).  Results are ALWAYS the reference's.
 * float32 shards of >= 32K rows at d <= 384 are searched by the sieve: a bf16
 * filter (an int8 one where the rows are finite and of one norm and k <= 16)
 * with a rigorous error margin lets through every row that can reach a
 * proven lower bound of the k-th best distance; the candidates whose filter
 * value, widened by the margin, can still reach the k-th best EXACT distance are
 * re-scored with the reference formula in float64 (the others are proven
 * outside the top k), and the reference order is taken over them - exact by
 * construction; only a full candidate buffer (thousands of rows inside the
 * margin) hands a query to the exact pass.  The other shapes use
 * filter scans with candidate lists whose completeness is proven a posteriori;
 * a query they cannot prove (more near-ties at the cut than the lists hold), and
 * any k beyond the filters (64 on the sieve; 52 / 48 / 56 / 28 on the list scans)
 * is recomputed by the exact pass - the reference formula in float64 over every
 * row, stable order on (distance, row) - before the call returns / in stream
 * order.  The flag only reports which route a query took. */
#define MIR_FLAG_UNCERTAIN 1  /* never returned since ABI 2 (kept for callers of ABI 1) */
#define MIR_FLAG_EXACT_PASS 2 /* answered by the exact pass */

int32_t mir_abi_version(void);
const char *mir_last_error(void);
int32_t mir_device_count(int32_t *out_count);

/* ------------------------------------------------------------------------
 * Vector index: replaces EmbeddingsIndex / DocIndex
 * (aidial_rag/retrievers/embeddings_index.py:14-89).
 *
 * One handle holds the flattened rows of all DocIndex objects of one
 * EmbeddingsIndex, in (doc_index, row) order - the order that defines the
 * reference's tie-break (stable argsort per document, then across documents,
 * embeddings_index.py:57-58,81).  `doc_ids[i]` / `chunk_ids[i]` are the
 * (doc_id, chunk_id) pair `to_metadata_doc` would receive for row i
 * (index_record.py:29-38).  NULL chunk_ids means 0..n-1, NULL doc_ids means 0.
 * `row_offset` is the global index of local row 0 when the handle is one
 * row-shard of a larger index (multi-GPU); returned row numbers include it.
 * ---------------------------------------------------------------------- */
typedef struct mir_index mir_index;

int32_t mir_index_create(const void *emb_host, int64_t n, int32_t d, int32_t dtype,
                         const int64_t *chunk_ids_host, const int32_t *doc_ids_host,
                         int32_t device, int64_t row_offset, mir_index **out);

/* Same, from a matrix already in HBM on `device` (e.g. the encoder's output).
 * The matrix is copied; the caller keeps ownership of its buffers. */
int32_t mir_index_create_from_device(const void *emb_device, int64_t n, int32_t d, int32_t dtype,
                                     const int64_t *chunk_ids_device, const int32_t *doc_ids_device,
                                     int32_t device, int64_t row_offset, void *stream, mir_index **out);

/* Row blocks: one document's rows resident in HBM, and an index composed from blocks.
 * The reference rebuilds its matrix inside every request from the DocIndex of each document the
 * request names (semantic_retriever.py:26-41 -> embeddings_index.py:121-136); requests over
 * different document sets share most documents.  A block is uploaded once per document
 * (chunk_ids NULL = 0..n-1); mir_index_create_from_rows concatenates blocks device-to-device in
 * the order given (= the reference's document order, which fixes the tie-break) and gives every
 * row of block p the doc id part_doc_ids[p] (NULL = p).  Blocks are copied: they may be destroyed
 * or reused for other indexes afterwards.  All blocks must share d, dtype and the device. */
typedef struct mir_rows mir_rows;
int32_t mir_rows_create(const void *emb_host, int64_t n, int32_t d, int32_t dtype,
                        const int64_t *chunk_ids_host, int32_t device, mir_rows **out);
int32_t mir_rows_info(const mir_rows *rows, int64_t *n, int32_t *d, int32_t *dtype, int32_t *device,
                      int64_t *hbm_bytes);
int32_t mir_rows_destroy(mir_rows *rows);
int32_t mir_index_create_from_rows(const mir_rows *const *parts, const int32_t *part_doc_ids,
                                   int32_t nparts, int32_t device, int64_t row_offset, mir_index **out);

int32_t mir_index_destroy(mir_index *idx);

/* n rows, dimension, dtype, device, bytes of HBM held */
int32_t mir_index_info(const mir_index *idx, int64_t *n, int32_t *d, int32_t *dtype, int32_t *device,
                       int64_t *hbm_bytes);

/* EmbeddingsIndex.find for a batch of queries (embeddings_index.py:62-89).
 * queries_host: double[b][d] - the live path hands a float64 query
 * (semantic_retriever.py:49,53); a float32 caller widens exactly.
 * Outputs, each [b][k], filled for the first out_count[q] entries of row q:
 *   out_doc / out_chunk : the (doc_id, chunk_id) pairs, best first
 *   out_row             : global flattened row (row_offset + local row)
 *   out_dist            : the metric value in float64 (reference promotion)
 * out_count[q] = min(k, n).  Any k >= 1 (the reference takes any `limit`,
 * embeddings_index.py:58,81).  out_flags may be NULL.
 * Any of out_doc / out_chunk / out_row / out_dist may be NULL. */
int32_t mir_index_search(mir_index *idx, const double *queries_host, int32_t b, int32_t k, int32_t metric,
                         int32_t *out_doc, int64_t *out_chunk, int64_t *out_row, double *out_dist,
                         int32_t *out_count, int32_t *out_flags);

/* The same with every buffer in HBM, asynchronous on `stream`. */
int32_t mir_index_search_device(mir_index *idx, const double *queries_device, int32_t b, int32_t k,
                                int32_t metric, int32_t *out_doc, int64_t *out_chunk, int64_t *out_row,
                                double *out_dist, int32_t *out_count, int32_t *out_flags, void *stream);

/* Benchmark instrumentation: when enabled, every launch of the scan kernel (the
 * dominant, HBM-streaming kernel) is bracketed by HIP events on the stream it
 * is launched on.  mir_index_profile_read waits for those launches, returns
 * their number and summed duration in milliseconds, and optionally resets. */
int32_t mir_index_profile(mir_index *idx, int32_t enable);
int32_t mir_index_profile_read(mir_index *idx, int32_t reset, int64_t *launches, double *total_ms);

/* Counters of the sieve - the search of large float32 shards: a filter on half the index image, the reference formula
 * for every (row, query) candidate it lets through, the reference order over those - since the last reset
 * (synchronises the device).  out8[0], out8[1]: candidates written by its first / second filter launch; out8[2]:
 * queries it answered; out8[3]: queries it handed to the exact pass (a full candidate buffer); out8[4]: candidates
 * listed when the second launch's thresholds were taken; out8[5]: rows evaluated with the reference's float64 formula
 * (the candidates that could be among the first k); out8[6]: 1 if the shard holds the int8 image (its searches for
 * up to 16 results run the int8 first stage, csrc/vec_kernels_i8.h), else 0; out8[7]: 0. */
int32_t mir_index_scan_stats(mir_index *idx, int32_t reset, int64_t *out8);

/* ENUM_TO_METRIC[metric](query, docs) -> float64[n]
 * (embeddings_metrics.py:53-58) over all rows of the index, on the GPU.
 * query_host: double[d]; out_host: double[n]. */
int32_t mir_index_metric_eval(mir_index *idx, const double *query_host, int32_t metric, double *out_host);

/* One-shot form of the same for a host matrix that is not an index
 * (the reference's metric functions are called on bare arrays in its tests). */
int32_t mir_metric_eval(const void *docs_host, int64_t n, int32_t d, int32_t dtype, const double *query_host,
                        int32_t metric, int32_t device, double *out_host);

/* ------------------------------------------------------------------------
 * Cross-shard merge: the step after the all-gather of per-shard partial
 * top-k (does not exist in the reference; it is the second stable argsort of
 * embeddings_index.py:81 applied across shards).  Inputs are [s][b][k]
 * (shard-major) with counts [s][b]; ordering is (dist ascending, NaN last,
 * row ascending).  `shard_stride_bytes` is the distance between consecutive
 * shards' blocks in each of the three arrays (0 = densely packed: b*k*8,
 * b*k*8 and b*4 bytes) - an all-gather of one per-rank blob holding
 * {dist[b][k], row[b][k], count[b]} is merged in place by passing the three
 * base pointers of shard 0 and the blob size.  For BM25 pass descending_scores = 1: ordering becomes
 * (score descending, row DESCENDING), the reversed stable argsort of
 * bm25_retriever.py:84.
 * ---------------------------------------------------------------------- */
int32_t mir_topk_merge_device(const double *dist, const int64_t *row, const int32_t *count, int32_t s,
                              int64_t shard_stride_bytes, int32_t b, int32_t k, int32_t descending_scores,
                              double *out_dist, int64_t *out_row, int32_t *out_count, int32_t device,
                              void *stream);
int32_t mir_topk_merge_host(const double *dist, const int64_t *row, const int32_t *count, int32_t s,
                            int64_t shard_stride_bytes, int32_t b, int32_t k, int32_t descending_scores,
                            double *out_dist, int64_t *out_row, int32_t *out_count);

/* ------------------------------------------------------------------------
 * BM25: replaces the rank_bm25.BM25Okapi model that BM25Retriever builds and
 * queries (aidial_rag/retrievers/bm25_retriever.py:64-84; third-party
 * rank-bm25 0.2.2, k1 = 1.5, b = 0.75, epsilon = 0.25 by default).
 *
 * Documents are token-id lists: indptr[i]..indptr[i+1] slices term_ids (text
 * order, repeats kept) for document i, in the flattened (doc_record, chunk)
 * order of from_doc_records (bm25_retriever.py:68-72).  The str -> id
 * vocabulary lives on the host side of the boundary.  Scores are float64 in
 * the package's operation order (bit-identical), ties - the all-zero scores
 * included - go to the HIGHEST index (`argsort(stable)[::-1]`, :84).
 * mir_bm25_create fails with MIR_ERR_EMPTY ("Text index is empty.", :75-76)
 * when there is no token at all.  For a document-sharded corpus pass the
 * GLOBAL idf[vocab] and average length as overrides and the shard's first
 * global document index as doc_offset; otherwise pass NULL / 0.
 * ---------------------------------------------------------------------- */
typedef struct mir_bm25 mir_bm25;

int32_t mir_bm25_create(const int64_t *indptr_host, const int32_t *term_ids_host, int64_t n_docs, int32_t vocab,
                        double k1, double b, double epsilon, const double *idf_override_host,
                        double avgdl_override, int32_t device, int64_t doc_offset, mir_bm25 **out);
/* Host utility: Snowball-English stems of a batch of lower-cased tokens - the per-token step of
 * keywords_preprocess (aidial_rag/keywords_search.py:13-18, `stemmer.stem(t.lower())`), pure Python in the
 * reference and the CPU hot spot of its index build (bm25_retriever.py:30-39,112).  Follows NLTK's
 * EnglishStemmer including its quirks (csrc/stem_english.cpp).  tokens: n_bytes of UTF-8 separated by `sep`
 * (no trailing separator); out: >= n_bytes bytes, receives the stems in the same form. */
int32_t mir_stem_english(const char *tokens, int64_t n_bytes, char sep, char *out, int64_t *out_bytes);

/* Host utility (ABI 4): keywords_preprocess (aidial_rag/keywords_search.py:13-18) for a BATCH of chunk texts on all
 * host cores - word_tokenize -> drop stopwords (compared before lower-casing) -> Snowball stem of the lower-cased
 * token - what BM25Retriever.build_index runs per chunk on a CPU pool next to the encoder
 * (retrievers/bm25_retriever.py:30-39,106-114; documents.py:188-198).  csrc/keywords_preprocess.cpp: NLTK's
 * Treebank word tokenizer restated rule by rule (pinned), the mirror's rule-based sentence splitter and stopword
 * list (both unpinned approximations of nltk_data), str.lower() from generated Unicode tables.
 * texts: n_texts UTF-8 texts back to back, text i = [offsets[i], offsets[i + 1]); n_threads <= 0 = all cores;
 * mode 0 = keywords_preprocess, 1 = word_tokenize only, 2 = the Treebank rules alone (the text is one sentence).
 * A text holding a NUL byte -> MIR_ERR_INVALID.
 * The result owns its buffers: `bytes` = every token followed by a NUL, text after text; counts[n_texts] tokens
 * per text; byte_ends[n_texts]: text i's tokens are bytes [byte_ends[i - 1], byte_ends[i]). */
typedef struct mir_kwp_result mir_kwp_result;
int32_t mir_keywords_preprocess(const char *texts, const int64_t *offsets, int32_t n_texts, int32_t n_threads,
                                int32_t mode, mir_kwp_result **out);
int32_t mir_kwp_result_data(const mir_kwp_result *r, const char **bytes, int64_t *n_bytes, const int32_t **counts,
                            const int64_t **byte_ends, int64_t *n_tokens);
/* The batch's DISTINCT tokens in order of first appearance (each followed by a NUL) and every token as an index into
 * them: what a vocabulary mapping (bm25_retriever.py:78, rank-bm25's dicts) needs - one lookup per distinct token. */
int32_t mir_kwp_result_dedupe(mir_kwp_result *r, int32_t n_threads);
int32_t mir_kwp_result_unique(const mir_kwp_result *r, const char **uniq_bytes, int64_t *n_uniq_bytes, int32_t *n_unique,
                              const int32_t **ids);
int32_t mir_kwp_result_free(mir_kwp_result *r);

/* Host utility: term ids written in a vocabulary larger than this corpus (e.g. a process-wide str -> id map
 * shared by every document the process has tokenised) -> ids 0..n_used-1 in order of first appearance, which is
 * the order rank-bm25's dicts would have (bm25_retriever.py:78).  remap[vocab]: old id -> new id or -1. */
int32_t mir_compact_term_ids(const int32_t *ids, int64_t n, int32_t vocab, int32_t *out_ids, int32_t *remap,
                             int32_t *n_used);

/* Document-sharded corpus (one model per GPU over a contiguous range of documents; idf, its average and the
 * average document length are GLOBAL statistics).  Either pass them to mir_bm25_create as overrides, or
 * build the shard first and install them afterwards:
 *   mir_bm25_corpus_stats      this model's per-term document frequency df[vocab], position of each term's first token
 *                              in this model's token stream (INT64_MAX = absent), token and document counts - the
 *                              quantities to all-reduce (SUM, MIN of shard-offset positions, SUM, SUM);
 *   mir_bm25_idf_from_stats    host: BM25Okapi._calc_idf (rank-bm25 0.2.2, behind bm25_retriever.py:78) from such
 *                              statistics - the routine mir_bm25_create itself uses, so a sharded model's idf is
 *                              bit-identical to the unsharded one's;
 *   mir_bm25_set_global_stats  re-derives the posting weights for `avgdl` on the device and installs idf[vocab]. */
int32_t mir_bm25_corpus_stats(const mir_bm25 *h, int64_t *out_df, int64_t *out_first_pos, int64_t *out_total_tokens,
                              int64_t *out_n_docs);
int32_t mir_bm25_idf_from_stats(const int64_t *df, const int64_t *first_pos, int32_t vocab, int64_t n_docs, double epsilon,
                                double *out_idf, double *out_average_idf);
int32_t mir_bm25_set_global_stats(mir_bm25 *h, const double *idf_host, double avgdl, double average_idf);

int32_t mir_bm25_destroy(mir_bm25 *h);
/* Tuning / test hook: how many queries one workgroup of the fast scoring pass walks through its document tile
 * (the depth of its software pipeline), 1..64; 0 = chosen per call from the batch size (the default).  Results do
 * not depend on it. */
int32_t mir_bm25_tune(mir_bm25 *h, int32_t queries_per_workgroup);
int32_t mir_bm25_info(const mir_bm25 *h, int64_t *n_docs, int32_t *vocab, int64_t *n_postings, double *avgdl,
                      double *average_idf, int64_t *hbm_bytes);
/* the model's idf table, float64[vocab] (0 for terms that never occur) */
int32_t mir_bm25_idf(const mir_bm25 *h, double *out_idf_host);

/* BM25Okapi.get_scores(query) -> float64[n_docs] (bm25_retriever.py:83).
 * Unknown ids (< 0 or >= vocab) contribute 0, repeated ids count again. */
int32_t mir_bm25_scores(mir_bm25 *h, const int32_t *q_terms_host, int32_t nq, double *out_scores_host);

/* _get_top_n_indexes (bm25_retriever.py:81-84) for b queries; q_ptr[b+1]
 * slices q_terms.  Outputs [b][k]: out_idx = doc_offset + local document
 * index, best first; out_count[q] = min(k, n_docs).  Any k >= 1 (the reference takes any n): up to 64 by the
 * selection kernels, beyond that the dense score vectors are ranked in rounds of 64. */
int32_t mir_bm25_search(mir_bm25 *h, const int32_t *q_terms_host, const int32_t *q_ptr_host, int32_t b,
                        int32_t k, int64_t *out_idx, double *out_score, int32_t *out_count);
/* Same with every buffer in HBM, asynchronous on `stream`; `workspace` holds
 * mir_bm25_workspace_bytes(h, b, k) bytes of HBM (no initial contents required; since ABI 3 it includes the candidate
 * pool of the wave-grain fast pass: up to b x 16 384 x 12 bytes, and for k > 64 a chunk of dense score vectors). */
int64_t mir_bm25_workspace_bytes(const mir_bm25 *h, int32_t b, int32_t k);
int32_t mir_bm25_search_device(mir_bm25 *h, const int32_t *q_terms_device, const int32_t *q_ptr_device,
                               int32_t b, int32_t k, int64_t *out_idx, double *out_score, int32_t *out_count,
                               void *workspace, void *stream);

/* ------------------------------------------------------------------------
 * Rank fusion: langchain EnsembleRetriever.weighted_reciprocal_rank as wired
 * at aidial_rag/retrieval_chain.py:239-245 (weights 1.0, c = 60).  Host code:
 * at most 4 lists x 7 items.  keys[total][2] are the (doc_id, chunk_id) pairs
 * of every list back to back (the reference keys items by page_content =
 * "{doc_id}_{chunk_id}", index_record.py:33-34); list_ptr[n_lists+1] slices
 * them.  Outputs hold the unique keys, best first, and their scores.
 * ---------------------------------------------------------------------- */
int32_t mir_rrf_fuse(const int64_t *keys, const int32_t *list_ptr, const double *weights, int32_t n_lists,
                     int32_t c, int64_t *out_keys, double *out_scores, int32_t *out_count);
/* The same for b queries in one call: query q's lists are list_ptr[q * (n_lists + 1) ..], offsets relative to
 * key_base[q] (in keys, i.e. pairs); outputs [b][cap][2], [b][cap], [b]; cap >= the items of any one query. */
int32_t mir_rrf_fuse_batch(const int64_t *keys, const int64_t *key_base, const int32_t *list_ptr, const double *weights,
                           int32_t n_lists, int32_t c, int32_t b, int32_t cap, int64_t *out_keys, double *out_scores,
                           int32_t *out_count);

/* ------------------------------------------------------------------------
 * WordPiece tokenisation (host code): the BertTokenizer step in front of the encoder
 * (aidial_rag/embeddings/embeddings.py:79-96 reaches it through sentence-transformers).  The caller supplies the
 * vocabulary (the lines of vocab.txt) and per-code-point tables over the Basic Multilingual Plane (class of raw and
 * of normalised code points: 0 other, 1 whitespace, 2 removed, 3 punctuation, 4 CJK, 5 hand the text back; the
 * normalised form = NFD, combining marks dropped, lower-cased, up to 3 code points): the library carries no Unicode
 * data.  mir_wordpiece_encode: n UTF-8 texts back to back (text i = [text_ptr[i], text_ptr[i+1])) -> ids with
 * [CLS] / [SEP], truncated to max_len in all, sequence i at out_ids[i * max_len .. + out_len[i]); fallback[i] = 1
 * when text i holds a code point beyond the BMP or invalid UTF-8 (nothing written: the caller tokenises it with
 * the reference tokenizer).  threads <= 0: one per core, at most 32.
 * ---------------------------------------------------------------------- */
typedef struct mir_wordpiece mir_wordpiece;
int32_t mir_wordpiece_create(const char *vocab, int64_t vocab_bytes, const uint8_t *cls, const uint8_t *ncls,
                             const uint32_t *map, const uint8_t *maplen, int32_t max_chars_per_word, mir_wordpiece **out);
int32_t mir_wordpiece_destroy(mir_wordpiece *t);
int32_t mir_wordpiece_encode(const mir_wordpiece *t, const char *texts, const int64_t *text_ptr, int32_t n, int32_t max_len,
                             int32_t threads, int32_t *out_ids, int32_t *out_len, uint8_t *fallback);

/* ------------------------------------------------------------------------
 * Text encoder: replaces the sentence-transformers forward behind
 * bge_embedding_impl / AsyncEmbeddings (aidial_rag/embeddings/embeddings.py:
 * 52-108): bge-small-en = BERT with hidden 384, 12 heads, FFN 1536, CLS
 * pooling, L2-normalised output (`normalize_embeddings=True`, :60-62).
 * Tokenisation (WordPiece, truncation at 512, the BGE query instruction) stays
 * on the host side of the boundary: the entry points take token ids.
 *
 * mir_encoder_create: float32 host tensors in the Hugging Face BertModel
 * layout.  `layer_tensors` holds layers*16 pointers, per layer in this order:
 *   attention.self.query.{weight,bias}, key.{weight,bias}, value.{weight,bias},
 *   attention.output.dense.{weight,bias}, attention.output.LayerNorm.{weight,bias},
 *   intermediate.dense.{weight,bias}, output.dense.{weight,bias},
 *   output.LayerNorm.{weight,bias}
 * (Linear weights are [out][in]).  type_emb is row 0 of token_type_embeddings.
 * This build is specialised for the bge-small-en shape and refuses others.
 *
 * mir_encoder_encode: token_ids = the sequences back to back (with [CLS] /
 * [SEP] already in place), seq_lens[n_seq] in 1..512; out float32
 * [n_seq][384].  normalize = 1 for the reference's behaviour.
 * ---------------------------------------------------------------------- */
typedef struct mir_encoder mir_encoder;

int32_t mir_encoder_create(int32_t hidden, int32_t layers, int32_t heads, int32_t intermediate, int32_t vocab,
                           int32_t max_pos, const float *word_emb, const float *pos_emb, const float *type_emb,
                           const float *emb_ln_gamma, const float *emb_ln_beta, const float *const *layer_tensors,
                           int32_t device, mir_encoder **out);
int32_t mir_encoder_destroy(mir_encoder *e);
int32_t mir_encoder_info(const mir_encoder *e, int32_t *layers, int32_t *hidden, int64_t *hbm_bytes);
int32_t mir_encoder_encode(mir_encoder *e, const int32_t *token_ids_host, const int32_t *seq_lens_host,
                           int32_t n_seq, int32_t normalize, float *out_host);
/* embeddings written straight to HBM (e.g. into the matrix an index is built from) */
int32_t mir_encoder_encode_to_device(mir_encoder *e, const int32_t *token_ids_host, const int32_t *seq_lens_host,
                                     int32_t n_seq, int32_t normalize, float *out_device, void *stream);
/* test hook: hidden states after `run_layers` layers (0 = embeddings), unpacked to
 * float32 [padded tokens][384] (each sequence padded to a multiple of 32 tokens) */
int32_t mir_encoder_debug_hidden(mir_encoder *e, const int32_t *token_ids_host, const int32_t *seq_lens_host,
                                 int32_t n_seq, int32_t run_layers, float *pooled_out_host,
                                 float *hidden_out_host, int64_t hidden_capacity_tokens);

#ifdef __cplusplus
}
#endif
#endif /* MIRETR_H */
