// Host side of the vector index: handle lifetime, workspaces, kernel dispatch.
// C ABI declared in include/miretr.h.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "common.h"
#include "vec_kernels.h"
#include "vec_kernels_f16.h"
#include "vec_kernels_q16.h"
#include "vec_kernels_h16.h"
#include "vec_kernels_sieve.h"
#include "vec_kernels_i8.h"
#include "vec_kernels_exact.h"

namespace mir {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int32_t use_device(int32_t device, int *num_cus) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no usable HIP device (hipGetDeviceCount: %s); libmiretr has no CPU fallback",
                  e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        (void)hipGetLastError();
        return MIR_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range (have %d)", device, count);
        return MIR_ERR_INVALID;
    }
    MIR_HIP(hipSetDevice(device));
    if (num_cus) {
        hipDeviceProp_t prop;
        MIR_HIP(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
            return MIR_ERR_NO_DEVICE;
        }
        *num_cus = prop.multiProcessorCount;
    }
    return MIR_OK;
}

// k-steps (16 columns each) the register-resident scan is instantiated for
static int pad_ksteps(int d) {
    const int ks = (d + 15) / 16;
    static const int kInst[] = {1, 2, 4, 8, 16, 24};
    for (int v : kInst)
        if (ks <= v) return v;
    if (ks <= 32) return 32;  // 384 < d <= 1024: the 64-query K-split scan over the bf16 hi/lo image (whole 16-k-step stages)
    if (ks <= 48) return 48;
    if (ks <= 64) return 64;
    return (ks + 7) / 8 * 8;  // generic kernel: multiple of the ring depth
}

struct Workspace {
    hipStream_t stream = nullptr;    // own stream for the host-buffer API
    hipEvent_t done = nullptr;       // recorded after the last enqueue that used the buffers
    hipStream_t last_stream = nullptr;
    bool pending = false;
    void *buf = nullptr;             // one slab, carved below
    size_t cap = 0;
    char *pin = nullptr;             // pinned host staging of the host-buffer API: queries in, results out
    size_t pin_cap = 0;
};

}  // namespace mir

using namespace mir;

struct mir_index {
    int device = 0;
    int num_cus = 0;
    int64_t n = 0;
    int d = 0;
    int dtype = MIR_DTYPE_F32;
    int64_t row_offset = 0;
    int ksteps = 0;
    uint32_t n_tiles = 0;
    float *d_orig = nullptr;     // float32 rows (re-scoring); null on a float16-native index
    _Float16 *d_f16 = nullptr;   // float16-native index: the rows as given (re-scoring)
    bool native16 = false;       // float16 storage scanned as 2-byte fragments (vec_kernels_f16.h)
    bool layout16 = false;       // float32, d padded to 128 / 256 / 384: the 16x16x32 image of vec_kernels_q16.h
    bool hi_only = false;        // layout16 shard that only the sieve ever scans (>= 32K rows): the image holds the bf16 hi blocks alone
    bool wide16 = false;         // float32, 384 < d <= 1024, >= 32K rows: `d_hi16` = the bf16 hi parts in the float16-native image's layout,
    int ks16 = 0;                //   ks16 blocks of 1 KiB per tile (d padded to 512 / 1024); the sieve's BF filter scans it, nothing reads d_split
    uint4 *d_hi16 = nullptr;
    uint4 *d_split = nullptr;    // bf16 hi/lo fragments, or the float16 fragments of a native16 index
    float *d_docsq = nullptr;    // padded to n_tiles*32
    float *d_invnorm = nullptr;  // padded to n_tiles*32
    float *d_dnorm = nullptr;    // max((float)|row|, 1e-8), exact_metric_wave's summation order: cosine_sim of the batched exact pass
    float *d_tilemax = nullptr;  // [n_tiles] a tile's largest row norm: the sieve's per-tile margin (layout16)
    bool norms_spread = false;   // the largest row norm exceeds the smallest tile maximum by more than 1/16: the sieve takes its margins per tile /
                                 // per row (unit-norm embeddings - the headline - keep round 3's one margin per query: the per-tile form cost that step ~3 %)
    float *d_maxnorm = nullptr;
    // the sieve's int8 first stage (vec_kernels_i8.h): built after the float image where the shard qualifies (build_i8)
    bool i8 = false;
    uint4 *d_i8 = nullptr;       // n_stages x 2 tiles x ks64 * 2 blocks of 1 KiB
    float *d_i8stats = nullptr;  // kI8StatWords floats
    float4 *d_i8tp = nullptr;    // [n_stages * 2] tile parameters (scale, residual bound, 1 / (2 scale))
    float *d_i8rec = nullptr;    // [2][n_stages][72] the filter's per-stage records: squared norms + parameters, inverse norms + parameters
    int ks64 = 0;
    uint32_t n_stages = 0;       // 64-row stages = ceil(n_tiles / 2)
    unsigned long long *d_stats = nullptr;  // 8 counters of the sieve (mir_index_scan_stats)
    int64_t *d_chunk = nullptr;
    int32_t *d_doc = nullptr;
    int64_t hbm_bytes = 0;
    std::mutex mu;
    std::vector<Workspace *> pool;
    // benchmark instrumentation (mir_index_profile)
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    std::vector<hipEvent_t> prof_free;  // recycled events: hipEventCreate inside a timed loop costs milliseconds now and then
    int64_t prof_launches = 0;
    double prof_ms = 0.0;
};

// One document's rows resident in HBM (mir_rows_create): the unit that outlives requests, so that an
// index over ANY set of documents is composed device-to-device (mir_index_create_from_rows).
struct mir_rows {
    int device = 0;
    int64_t n = 0;
    int d = 0;
    int dtype = MIR_DTYPE_F32;
    void *d_emb = nullptr;       // n x d, as given (float32 or float16)
    int64_t *d_chunk = nullptr;  // n
    int64_t hbm_bytes = 0;
};

namespace mir {

static void free_index(mir_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    for (Workspace *w : ix->pool) {
        if (w->pending) (void)hipEventSynchronize(w->done);
        if (w->buf) (void)hipFree(w->buf);
        if (w->pin) (void)hipHostFree(w->pin);
        if (w->done) (void)hipEventDestroy(w->done);
        if (w->stream) (void)hipStreamDestroy(w->stream);
        delete w;
    }
    for (auto &pr : ix->prof_events) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (hipEvent_t ev : ix->prof_free) (void)hipEventDestroy(ev);
    (void)hipFree(ix->d_orig);
    (void)hipFree(ix->d_f16);
    (void)hipFree(ix->d_split);
    (void)hipFree(ix->d_docsq);
    (void)hipFree(ix->d_invnorm);
    (void)hipFree(ix->d_dnorm);
    (void)hipFree(ix->d_tilemax);
    (void)hipFree(ix->d_i8);
    (void)hipFree(ix->d_i8stats);
    (void)hipFree(ix->d_i8tp);
    (void)hipFree(ix->d_i8rec);
    (void)hipFree(ix->d_hi16);
    (void)hipFree(ix->d_maxnorm);
    (void)hipFree(ix->d_stats);
    (void)hipFree(ix->d_chunk);
    (void)hipFree(ix->d_doc);
    delete ix;
}

// Threshold pre-pass of the wide scans: kSampleWgs workgroups x up to kSampleTilesPerWg tiles (32K rows)
constexpr int kSampleWgs = 256;
constexpr int kSampleTilesPerWg = 4;  // (8 / 16 / 32 tiles per workgroup measured within 0.5 % on 10M x 384, 1-4 % slower on 6.25M x 1024 float16)

// Build the derived device state from ix->d_orig (already filled) on `stream`.
// doc_sq / inv_norm / max norm: rows staged through LDS when at least 32 of them fit (d <= 511), else one thread
// per row (measured at d = 1024: 15 rows per workgroup, 37 ms per 6.25M rows against ~18 ms direct)
template <typename T>
static void launch_row_norms(const T *src, int64_t n, int d, mir_index *ix, hipStream_t stream) {
    unsigned int *mx = reinterpret_cast<unsigned int *>(ix->d_maxnorm);
    const int rows_per_wg = std::min<int64_t>(128, kNormsLdsBytes / ((int64_t)norms_row_stride(d) * 4));
    if (rows_per_wg >= 32) {
        const size_t lds = (size_t)rows_per_wg * norms_row_stride(d) * 4;
        row_norms_lds_kernel<T><<<dim3((unsigned)((n + rows_per_wg - 1) / rows_per_wg)), dim3(256), lds, stream>>>(
            src, n, d, rows_per_wg, ix->d_docsq, ix->d_invnorm, mx);
    } else {
        row_norms_kernel<T><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(src, n, d, ix->d_docsq, ix->d_invnorm, mx);
    }
}

static int32_t build_derived(mir_index *ix, hipStream_t stream) {
    const int64_t n = ix->n;
    const int d = ix->d;
    ix->ksteps = ix->native16 ? (d + 511) / 512 * 32 : pad_ksteps(d);  // native16: whole 32-k-step stages
    ix->layout16 = !ix->native16 && (ix->ksteps == 8 || ix->ksteps == 16 || ix->ksteps == 24);  // d padded to 128 / 256 / 384
    ix->n_tiles = (uint32_t)((n + kTileRows - 1) / kTileRows);
    // A layout16 shard of >= 32K rows is scanned by the sieve alone (plan(): k <= 64 -> sieve, beyond -> the exact pass), and the
    // sieve reads only the hi blocks: no lo blocks are stored for it - 7.68 GB at 10M x 384 - and a tile's
    // hi blocks follow the previous tile's directly, so the filter's stream is contiguous.  (MIR_NO_SIEVE - the A/B switch back to
    // round 2's scan, which does read lo blocks - is read at build time too.)
    ix->hi_only = ix->layout16 && ix->n_tiles >= 4u * kSampleWgs && getenv("MIR_NO_SIEVE") == nullptr;
    // wide float32 shards of >= 32K rows: the sieve's bf16 filter over a hi-only image of its own (round 4); the hi/lo split image
    // of the K-split list scan is then never read (k <= 64 -> sieve, beyond -> the exact pass) and is not built
    ix->wide16 = !ix->native16 && !ix->layout16 && (ix->ksteps == 32 || ix->ksteps == 48 || ix->ksteps == 64) &&
                 ix->n_tiles >= 4u * kSampleWgs && getenv("MIR_NO_SIEVE") == nullptr && getenv("MIR_NO_SIEVE_WIDE") == nullptr;
    ix->ks16 = ix->wide16 ? (d + 511) / 512 * 32 : 0;
    const size_t split_bytes = ix->wide16 ? 16 : (size_t)ix->n_tiles * ix->ksteps * ((ix->native16 || ix->hi_only) ? 1024 : 2048);
    const size_t aux_bytes = (size_t)(ix->n_tiles + 1) * kTileRows * sizeof(float);  // (one tile more: the int8 filter reads 64-row stages)
    MIR_HIP(hipMalloc(&ix->d_split, std::max<size_t>(split_bytes, 16)));
    MIR_HIP(hipMalloc(&ix->d_docsq, std::max<size_t>(aux_bytes, 16)));
    MIR_HIP(hipMalloc(&ix->d_invnorm, std::max<size_t>(aux_bytes, 16)));
    MIR_HIP(hipMalloc(&ix->d_maxnorm, 32));  // norm statistics: [0] largest norm, [1] non-finite flag, [2] largest bf16 residual, [3] largest residual / norm, [4] smallest tile maximum (bits), [5..7] -
    MIR_HIP(hipMalloc(&ix->d_dnorm, std::max<size_t>((size_t)n * 4, 16)));
    ix->hbm_bytes += split_bytes + 2 * aux_bytes + 32 + (size_t)n * 4;
    MIR_HIP(hipMemsetAsync(ix->d_docsq, 0, std::max<size_t>(aux_bytes, 16), stream));
    MIR_HIP(hipMemsetAsync(ix->d_invnorm, 0, std::max<size_t>(aux_bytes, 16), stream));
    MIR_HIP(hipMemsetAsync(ix->d_maxnorm, 0, 32, stream));
    MIR_HIP(hipMemsetAsync(reinterpret_cast<char *>(ix->d_maxnorm) + 16, 0xff, 4, stream));  // word 4: a minimum over unsigned float bits
    MIR_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_stats), 64));
    MIR_HIP(hipMemsetAsync(ix->d_stats, 0, 64, stream));
    if (n > 0) {
        const int64_t total_lanes = (int64_t)ix->n_tiles * ix->ksteps * 64;
        const int64_t blocks = (total_lanes + 255) / 256;
        MIR_REQUIRE(blocks < (int64_t)0x7fffffff, "index too large for one pack launch");
        if (ix->native16) {
            pack_f16_16_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(ix->d_f16, n, d, ix->ksteps / 2, total_lanes,
                                                                               ix->d_split);  // a tile: ksteps blocks of 1 KiB
            MIR_HIP(hipGetLastError());
            launch_row_norms(ix->d_f16, n, d, ix, stream);
        } else if (ix->layout16) {
            const int64_t lanes16 = total_lanes;  // tiles x (ks32 * 2) blocks x 64 lanes; one thread writes a hi and a lo block
            pack_split16_f32_kernel<<<dim3((unsigned)((lanes16 + 255) / 256)), dim3(256), 0, stream>>>(ix->d_orig, n, d, ix->ksteps / 2,
                                                                                                   lanes16, ix->d_split, ix->hi_only);
            MIR_HIP(hipGetLastError());
            launch_row_norms(ix->d_orig, n, d, ix, stream);
        } else if (ix->wide16) {
            const int64_t lanes = (int64_t)ix->n_tiles * ix->ks16 * 64;
            MIR_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_hi16), (size_t)ix->n_tiles * ix->ks16 * 1024));
            ix->hbm_bytes += (size_t)ix->n_tiles * ix->ks16 * 1024;
            pack_hi16_f32_kernel<<<dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream>>>(ix->d_orig, n, d, ix->ks16 / 2, lanes, ix->d_hi16);
            MIR_HIP(hipGetLastError());
            launch_row_norms(ix->d_orig, n, d, ix, stream);
        } else {
            pack_split_f32_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(ix->d_orig, n, d, ix->ksteps,
                                                                                  total_lanes, ix->d_split);
            MIR_HIP(hipGetLastError());
            launch_row_norms(ix->d_orig, n, d, ix, stream);
        }
        MIR_HIP(hipGetLastError());
        if (ix->native16) row_dnorm_kernel<_Float16><<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream>>>(ix->d_f16, n, d, ix->d_dnorm);
        else row_dnorm_kernel<float><<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream>>>(ix->d_orig, n, d, ix->d_dnorm);
        MIR_HIP(hipGetLastError());
        if (ix->layout16) {  // the sieve's per-tile margin (vec_kernels_q16.h, hihi_coeff)
            MIR_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_tilemax), (size_t)ix->n_tiles * 4));
            ix->hbm_bytes += (size_t)ix->n_tiles * 4;
            tile_maxnorm_kernel<<<dim3((ix->n_tiles + 255) / 256), dim3(256), 0, stream>>>(ix->d_dnorm, n, ix->n_tiles, ix->d_tilemax,
                                                                                            reinterpret_cast<unsigned int *>(ix->d_maxnorm) + 4);
            MIR_HIP(hipGetLastError());
        }
    }
    return MIR_OK;
}

// The int8 image of a shard the bf16 sieve serves, where its one-scale-per-index quantisation and one-margin-per-query bound
// apply: finite rows, norms not spread (vec_kernels_i8.h).  Runs after build_derived's statistics are known on the host.
static int32_t build_i8(mir_index *ix, hipStream_t stream) {
    const int64_t n = ix->n;
    const int d = ix->d;
    ix->ks64 = ix->ksteps / 4;  // ksteps counts 16 columns: 8 / 16 / 24 -> 2 / 4 / 6 k-steps of 64
    ix->n_stages = (ix->n_tiles + 1) / 2;
    const size_t image = (size_t)ix->n_stages * 2 * ix->ks64 * 2 * 1024;
    MIR_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_i8), image));
    MIR_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_i8stats), kI8StatWords * 4));
    MIR_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_i8tp), (size_t)ix->n_stages * 2 * 16));
    ix->hbm_bytes += image + kI8StatWords * 4 + (size_t)ix->n_stages * 2 * 16;
    MIR_HIP(hipMemsetAsync(ix->d_i8stats, 0, kI8StatWords * 4, stream));
    i8_tile_scale_kernel<<<dim3(ix->n_stages * 2), dim3(256), 0, stream>>>(ix->d_orig, n, d, ix->d_i8tp);
    i8_scale_kernel<<<dim3(1), dim3(1), 0, stream>>>(ix->d_i8stats, ix->d_maxnorm);
    const int64_t lanes = (int64_t)ix->n_stages * 2 * ix->ks64 * 2 * 64;
    pack_i8_kernel<<<dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream>>>(ix->d_orig, n, d, ix->ks64, lanes, ix->d_i8tp, ix->d_i8);
    i8_residual_kernel<<<dim3((unsigned)((n + 15) / 16)), dim3(256), 0, stream>>>(ix->d_orig, n, d, ix->d_docsq, ix->d_i8tp, ix->d_i8stats);
    MIR_HIP(hipGetLastError());
    float st[kI8StatWords] = {};
    MIR_HIP(hipMemcpyAsync(st, ix->d_i8stats, sizeof(st), hipMemcpyDeviceToHost, stream));
    MIR_HIP(hipStreamSynchronize(stream));
    // the integer test takes the smallest squared norm for every row's: rows of ONE norm only (normalised embeddings)
    ix->i8 = st[7] > 0.f && st[1] - st[7] <= 1e-3f * st[1];
    if (ix->i8) {  // the norms' range, for cosine
        // (doc_sq is a float32 sum of d squares: 1e-4 covers its rounding many times over and costs the bound nothing)
        const float nr[3] = {sqrtf(st[7]) * (1.0f - 1e-4f), sqrtf(st[1]) * (1.0f + 1e-4f), 1.0f / (sqrtf(st[7]) * (1.0f - 1e-4f))};
        MIR_HIP(hipMemcpy(ix->d_i8stats + 4, nr, sizeof(nr), hipMemcpyHostToDevice));
    }
    if (!ix->i8) {
        (void)hipFree(ix->d_i8);
        (void)hipFree(ix->d_i8tp);
        ix->d_i8 = nullptr;
        ix->d_i8tp = nullptr;
        ix->hbm_bytes -= image + (size_t)ix->n_stages * 2 * 16;
    } else {
        const size_t rec = (size_t)ix->n_stages * 72 * 4;
        MIR_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_i8rec), 2 * rec));
        ix->hbm_bytes += 2 * rec;
        const int64_t n_col = (int64_t)(ix->n_tiles + 1) * kTileRows;  // (the columns are padded by a tile: build_derived)
        i8_stage_record_kernel<<<dim3(ix->n_stages), dim3(128), 0, stream>>>(ix->d_docsq, n_col, ix->d_i8tp, ix->d_i8rec);
        i8_stage_record_kernel<<<dim3(ix->n_stages), dim3(128), 0, stream>>>(ix->d_invnorm, n_col, ix->d_i8tp, ix->d_i8rec + (size_t)ix->n_stages * 72);
        MIR_HIP(hipGetLastError());
        MIR_HIP(hipStreamSynchronize(stream));
    }
    return MIR_OK;
}

static int32_t check_create_args(int64_t n, int32_t d, int32_t dtype, mir_index **out) {
    MIR_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    MIR_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "n=%lld out of range [0, 2^31)", (long long)n);
    MIR_REQUIRE(d >= 1 && d <= 128 * 4096, "d=%d out of range", d);
    if (dtype != MIR_DTYPE_F32 && dtype != MIR_DTYPE_F16) {
        set_error("dtype %d not supported (float32 = 0, float16 = 1)", dtype);
        return MIR_ERR_UNSUPPORTED;
    }
    return MIR_OK;
}


// carve helper
struct Carver {
    char *base;
    size_t off = 0;
    template <typename T>
    T *take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

struct SearchBuffers {
    double *q;       // [b][d]      (host API only)
    uint4 *qsplit;   // [ngroups][ksteps][2][64]
    double *q_sq;    // [b]
    double *q_err;   // [b] |q - bf16(q)|: what the hi fragments lose on the query's side (layout16; the sieve's margin)
    double *q_norm;  // [b]
    float *q_amax;   // [b] the int8 filter: a query's largest |q_i| (-1: not finite); its scales go to `qscale`
    float *qscale;   // [b] 1 / (query scale) of the float16-native scan
    uint64_t *part;  // [ngroups][nwg][qpw][klist]
    uint64_t *gthr;  // [ngroups][128] shared per-query thresholds of the 128-query scan; the control words below
                     // follow it in ONE allocation, which the prep kernel zeroes (ctl_words)
    int32_t *nflag;  // [2] number of queries handed to the exact pass
    uint32_t *arrive;  // [b] arrival counters of the exact pass
    int ctl_words;   // 64-bit words from gthr to the end of arrive
    int32_t *flagged;  // [b]
    double *bound_dist;  // [b] exact pass, k > 64: last result of the previous round
    uint32_t *bound_row; // [b]
    uint64_t *part_exact;  // [b][exact grid][64][2]
    double *qt;            // [ceil(b / 32)][d padded to 256][32]: the flagged queries, transposed, for the batched exact pass
    uint64_t *part_sample;  // [kSampleWgs][128][klist], reused by every launch (stream-ordered)
    // the sieve (vec_kernels_sieve.h): candidate regions of its two launches, the queries' verified lists
    uint64_t *sv_cand;      // [2][nwg][kSieveRegion]
    float *sv_candv;        // [2][nwg][kSieveRegion]
    uint32_t *sv_ccount;    // [2][nwg]
    SieveLists sv;          // rv / row [b][kSieveQueryCap]; count / over [b] live in the zeroed control block
    int32_t *o_doc;  // host API staging of outputs, [b][k]
    int64_t *o_chunk;
    int64_t *o_row;
    double *o_dist;
    int32_t *o_count;  // [b]
    int32_t *o_flags;  // [b]
};

// How a search runs (plan()): `ngroups` launches of the filter scan, `qpw` queries each, on `nwg` workgroups
// with candidate lists of `klist`; exact_only = no filter scan at all (k beyond the lists): every query takes
// the exact pass.
struct SearchPlan {
    int ngroups = 0, nwg = 1, klist = 2, qpw = 32;
    int exact_grid = 1;       // workgroups of exact_topk_kernel
    bool exact_only = false;
    int nwg_first = 0;        // layout16 progressive scan: workgroups (= candidate lists) and tiles of the first launch;
    uint32_t tiles_first = 0; // nwg counts the lists of BOTH launches (0 = one launch)
    bool sieve = false;       // large layout16 shard: hi-only filter + exhaustive verification (vec_kernels_sieve.h); nwg = workgroups per launch
    uint32_t sample_tpw = 0;  // the sieve's threshold sample: tiles per sample workgroup
};

static size_t carve(SearchBuffers &sb, char *base, int b, int k, int d, int ksteps, const SearchPlan &pl, bool host_api, size_t i8_rows = 0) {
    const int ngroups = pl.ngroups, nwg = pl.nwg, klist = pl.klist, qpw = pl.qpw;
    Carver c{base};
    sb.q = host_api ? c.take<double>((size_t)b * d) : nullptr;
    sb.qsplit = c.take<uint4>((size_t)ngroups * (qpw / 32) * ksteps * 128);  // (ksteps: the caller passes a wide16 index's ks16 when that is larger)
    sb.q_sq = c.take<double>(b);
    sb.q_err = c.take<double>(b);
    sb.q_norm = c.take<double>(b);
    sb.q_amax = c.take<float>(i8_rows ? b : 0);
    sb.qscale = c.take<float>((size_t)ngroups * std::max(128, qpw));
    sb.part = c.take<uint64_t>(pl.sieve ? 0 : (size_t)ngroups * nwg * qpw * klist);  // per-workgroup lists of the list scans (the sieve has its own regions: 33.5 MB per launch group saved)
    // one zeroed control block: gthr | nflag | arrive[b] | sieve over[b] | sieve count[b][32]  (u32 arrays padded to u64)
    const size_t gthr_words = (size_t)ngroups * std::max(128, qpw), arrive_words = ((size_t)b + 1) / 2;
    const size_t count_words = pl.sieve ? (size_t)b * kSieveCountStride / 2 : 0;
    sb.ctl_words = (int)(gthr_words + 1 + 2 * arrive_words + count_words);
    sb.gthr = c.take<uint64_t>((size_t)sb.ctl_words);
    sb.nflag = base ? reinterpret_cast<int32_t *>(sb.gthr + gthr_words) : nullptr;
    sb.arrive = base ? reinterpret_cast<uint32_t *>(sb.gthr + gthr_words + 1) : nullptr;
    sb.sv.over = base ? reinterpret_cast<uint32_t *>(sb.gthr + gthr_words + 1 + arrive_words) : nullptr;
    sb.sv.count = base ? reinterpret_cast<uint32_t *>(sb.gthr + gthr_words + 1 + 2 * arrive_words) : nullptr;
    sb.flagged = c.take<int32_t>(b);
    sb.bound_dist = c.take<double>(b);
    sb.bound_row = c.take<uint32_t>(b);
    sb.part_exact = c.take<uint64_t>((size_t)b * pl.exact_grid * std::min(k, kExactRound) * 2);
    sb.qt = c.take<double>((size_t)((b + kXbQ - 1) / kXbQ) * xb_dpad(d) * kXbQ);
    sb.part_sample = c.take<uint64_t>(std::max((size_t)kSampleWgs * 128 * klist, (size_t)kSampleWgs * std::max(128, qpw)));  // (the sieve's sample: two floats per workgroup and query)
    const size_t sv_q = pl.sieve ? (size_t)b * kSieveQueryCap : 0;
    const size_t region = i8_rows ? (size_t)kI8Region : (size_t)kSieveRegion;  // (the int8 filter: eight wave-private parts per workgroup)
    sb.sv_cand = c.take<uint64_t>(pl.sieve ? (size_t)2 * nwg * region : 0);
    sb.sv_candv = c.take<float>(pl.sieve ? (size_t)2 * nwg * region : 0);
    sb.sv_ccount = c.take<uint32_t>(pl.sieve ? (size_t)2 * nwg * (i8_rows ? 8 : 1) : 0);
    sb.sv.rv = c.take<float>(sv_q);
    sb.sv.row = c.take<uint32_t>(sv_q);
    if (host_api) {
        sb.o_doc = c.take<int32_t>((size_t)b * k);
        sb.o_chunk = c.take<int64_t>((size_t)b * k);
        sb.o_row = c.take<int64_t>((size_t)b * k);
        sb.o_dist = c.take<double>((size_t)b * k);
        sb.o_count = c.take<int32_t>(b);
        sb.o_flags = c.take<int32_t>(b);
    } else {
        sb.o_doc = nullptr; sb.o_chunk = nullptr; sb.o_row = nullptr; sb.o_dist = nullptr;
        sb.o_count = nullptr; sb.o_flags = nullptr;
    }
    return c.off + 256;
}

// Take a workspace that is safe to enqueue on `stream` (null = use its own).
static int32_t acquire_ws(mir_index *ix, hipStream_t stream, size_t need, Workspace **out) {
    Workspace *w = nullptr;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        for (size_t i = 0; i < ix->pool.size(); ++i) {
            Workspace *c = ix->pool[i];
            bool ok = !c->pending || (stream && c->last_stream == stream);
            if (!ok && hipEventQuery(c->done) == hipSuccess) {
                c->pending = false;
                ok = true;
            }
            if (ok) {
                w = c;
                ix->pool.erase(ix->pool.begin() + i);
                break;
            }
        }
    }
    if (!w) {
        w = new (std::nothrow) Workspace();
        MIR_REQUIRE(w != nullptr, "out of host memory");
        hipError_t e1 = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
        hipError_t e2 = hipEventCreateWithFlags(&w->done, hipEventDisableTiming);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            set_error("workspace stream/event creation failed");
            delete w;
            return MIR_ERR_HIP;
        }
    }
    if (w->cap < need) {
        if (w->pending) {
            (void)hipEventSynchronize(w->done);
            w->pending = false;
        }
        if (w->buf) (void)hipFree(w->buf);
        w->buf = nullptr;
        w->cap = 0;
        hipError_t e = hipMalloc(&w->buf, need);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) for search workspace failed: %s", need, hipGetErrorString(e));
            std::lock_guard<std::mutex> lk(ix->mu);
            ix->pool.push_back(w);
            return MIR_ERR_HIP;
        }
        w->cap = need;
    }
    *out = w;
    return MIR_OK;
}

static void release_ws(mir_index *ix, Workspace *w, hipStream_t used, bool pending) {
    w->last_stream = used;
    w->pending = pending;
    if (pending) (void)hipEventRecord(w->done, used);
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->pool.push_back(w);
}

// float32 index with 384 < d <= 1024: scanned 64 queries per pass by the K-split kernel of vec_kernels_f16.h (SPLIT form)
static bool wide64_split(const mir_index *ix) { return !ix->native16 && (ix->ksteps == 32 || ix->ksteps == 48 || ix->ksteps == 64); }

template <int KIND>
static int32_t launch_scan(const mir_index *ix, const uint4 *qsplit_g, int nq, int klist, int nwg,
                           uint64_t *part_g, hipStream_t stream) {
    const float *aux = KIND == SCAN_L2 ? ix->d_docsq : KIND == SCAN_COS ? ix->d_invnorm : nullptr;
    const size_t lds = ((size_t)klist * 256 + 32 * (size_t)klist) * 8;
    const uint32_t n_rows = (uint32_t)ix->n;
#define MIR_SCAN_CASE(KS)                                                                                    \
    case KS: {                                                                                               \
        auto kern = scan_topk_kernel<KS, KIND>;                                                              \
        MIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
        kern<<<dim3(nwg), dim3(256), lds, stream>>>(ix->d_split, aux, qsplit_g, n_rows, ix->n_tiles, nq,     \
                                                    klist, part_g);                                          \
        break;                                                                                               \
    }
    switch (ix->ksteps) {
        MIR_SCAN_CASE(1)
        MIR_SCAN_CASE(2)
        MIR_SCAN_CASE(4)  // (8 / 16 / 24 k-steps = d padded to 128 / 256 / 384 are layout16 indexes: launch_scan_q16)
        default: {
            auto kern = scan_topk_generic_kernel<KIND>;
            MIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            kern<<<dim3(nwg), dim3(256), lds, stream>>>(ix->d_split, aux, qsplit_g, ix->ksteps, n_rows,
                                                        ix->n_tiles, nq, klist, part_g);
            break;
        }
    }
#undef MIR_SCAN_CASE
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

// float32 layout16 scan: 128 queries per launch, 16 per wave (vec_kernels_q16.h)
template <int KIND>
static int32_t launch_scan_q16(const mir_index *ix, const uint4 *qsplit_g, const double *q_norm_g, int nq, int klist, int nwg,
                               uint32_t tile0, uint32_t n_tiles, uint64_t *part_g, uint64_t *gthr_g, bool sample, hipStream_t stream) {
    const float *aux = KIND == SCAN_L2 ? ix->d_docsq : KIND == SCAN_COS ? ix->d_invnorm : nullptr;
    const int ks32 = ix->ksteps / 2;
    const size_t lds = q16_lds_bytes(ks32, klist);
    const uint32_t n_rows = (uint32_t)ix->n;
    const int ns = q16_ring_stages(klist);
#define MIR_Q16_LAUNCH(KS, NSV)                                                                                        \
    do {                                                                                                               \
        auto kern = sample ? scan_topk_q16_kernel<KS, KIND, true, NSV> : scan_topk_q16_kernel<KS, KIND, false, NSV>;   \
        MIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        kern<<<dim3(nwg), dim3(512), lds, stream>>>(ix->d_split, aux, qsplit_g, q_norm_g, ix->d_maxnorm, n_rows, tile0, n_tiles, nq, \
                                                    klist, part_g, gthr_g);                                            \
    } while (0)
#define MIR_Q16_CASE(KS)                                                                                               \
    case KS:                                                                                                           \
        if (ns == 5) MIR_Q16_LAUNCH(KS, 5);                                                                            \
        else MIR_Q16_LAUNCH(KS, 4);                                                                                    \
        break;
    switch (ks32) {
        MIR_Q16_CASE(4)
        MIR_Q16_CASE(8)
        MIR_Q16_CASE(12)
        default:
            set_error("internal: q16 scan has no instance for %d k-steps of 32", ks32);
            return MIR_ERR_UNSUPPORTED;
    }
#undef MIR_Q16_CASE
#undef MIR_Q16_LAUNCH
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

// the sieve's filter launch over tiles [tile0, tile0 + n_tiles) (vec_kernels_sieve.h); sample = the threshold pre-pass
template <int KIND>
static int32_t launch_sieve(const mir_index *ix, int qpw, const uint4 *qsplit_g, const double *q_norm_g, const double *q_sq_g,
                            const double *q_err_g, int nq, int nwg,
                            uint32_t tile0, uint32_t n_tiles, int nan_guard, const uint64_t *gthr_g, uint64_t *cand, float *candv,
                            uint32_t *ccount, float *part_sample, bool sample, unsigned long long *stat, hipStream_t stream) {
    const float *aux = KIND == SCAN_L2 ? ix->d_docsq : KIND == SCAN_COS ? ix->d_invnorm : nullptr;
    const int ks32 = ix->ksteps / 2;
    const size_t lds = sieve_lds_bytes(ks32);
    const uint32_t n_rows = (uint32_t)ix->n;
#define MIR_SIEVE_PICK(KS, P)                                                                                          \
    (qpw > 128 ? (sample ? sieve_q16_kernel<KS, KIND, true, 2, P> : sieve_q16_kernel<KS, KIND, false, 2, P>)           \
               : (sample ? sieve_q16_kernel<KS, KIND, true, 1, P> : sieve_q16_kernel<KS, KIND, false, 1, P>))
#define MIR_SIEVE_CASE(KS)                                                                                             \
    case KS: {                                                                                                         \
        auto kern = ix->norms_spread ? MIR_SIEVE_PICK(KS, true) : MIR_SIEVE_PICK(KS, false);                           \
        MIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        kern<<<dim3(nwg), dim3(512), lds, stream>>>(ix->d_split, aux, qsplit_g, q_norm_g, q_sq_g, q_err_g, ix->d_maxnorm, n_rows, tile0, n_tiles, \
                                                    nq, nan_guard, gthr_g, cand, candv, ccount, part_sample, stat,     \
                                                    (uint32_t)(ks32 * 2 * 64 * (ix->hi_only ? 1 : 2)), ix->d_tilemax); \
        break;                                                                                                         \
    }
    switch (ks32) {
        MIR_SIEVE_CASE(4)
        MIR_SIEVE_CASE(8)
        MIR_SIEVE_CASE(12)
        default:
            set_error("internal: the sieve has no instance for %d k-steps of 32", ks32);
            return MIR_ERR_UNSUPPORTED;
    }
#undef MIR_SIEVE_CASE
#undef MIR_SIEVE_PICK
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

// the int8 filter's launch over 64-row stages [stage0, stage0 + n_stages) (vec_kernels_i8.h)
template <int KIND>
static int32_t launch_sieve_i8(const mir_index *ix, int qpw, const float *q_scale_g, const uint4 *qfrag_g, const double *q_norm_g,
                               const double *q_sq_g, const double *q_err_g, int nq, int nwg, uint32_t stage0, uint32_t n_stages,
                               int nan_guard, const uint64_t *gthr_g, uint64_t *cand, float *candv, uint32_t *ccount, float *part_sample,
                               bool sample, unsigned long long *stat, hipStream_t stream) {
    // from 65 queries per launch: four waves per workgroup (two or four query tiles each) and two workgroups per CU - 0.83 against
    // 0.92 ms per 128-query step at 10M x 384, 1.13-1.16 against 1.27-1.30 per 256; up to 64 queries half of four waves' query
    // tiles would be empty: eight waves with one tile each.  MIR_SIEVE_I8_WAVES = 4 / 8 forces one geometry (A/B runs)
    static const int nw_env = getenv("MIR_SIEVE_I8_WAVES") ? atoi(getenv("MIR_SIEVE_I8_WAVES")) : 0;
    const bool four = nw_env ? nw_env == 4 : nq > 64;
    const size_t lds = sieve_i8_lds_bytes(ix->ks64, four ? 4 : 8);
    const uint32_t n_rows = (uint32_t)ix->n;
    const int grid = (four && !sample) ? 2 * nwg : nwg;  // (the candidate parts are per wave: nwg x 8 of them either way)
#define MIR_I8_PICK(KS)                                                                                                \
    (four ? (qpw > 128 ? (sample ? sieve_i8_kernel<KS, KIND, true, 4, 4> : sieve_i8_kernel<KS, KIND, false, 4, 4>)     \
                       : (sample ? sieve_i8_kernel<KS, KIND, true, 2, 4> : sieve_i8_kernel<KS, KIND, false, 2, 4>))    \
          : (qpw > 128 ? (sample ? sieve_i8_kernel<KS, KIND, true, 2, 8> : sieve_i8_kernel<KS, KIND, false, 2, 8>)     \
                       : (sample ? sieve_i8_kernel<KS, KIND, true, 1, 8> : sieve_i8_kernel<KS, KIND, false, 1, 8>)))
#define MIR_I8_CASE(KS)                                                                                                \
    case KS: {                                                                                                         \
        auto kern = MIR_I8_PICK(KS);                                                                                   \
        MIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        kern<<<dim3(grid), dim3(four ? 256 : 512), lds, stream>>>(ix->d_i8, ix->d_i8rec + (KIND == SCAN_COS ? (size_t)ix->n_stages * 72 : 0), qfrag_g, q_norm_g, q_sq_g, q_err_g, ix->d_i8stats, q_scale_g, n_rows, \
                                                    stage0, n_stages, nq, nan_guard, gthr_g, cand, candv, ccount, part_sample, stat); \
        break;                                                                                                         \
    }
    switch (ix->ks64) {
        MIR_I8_CASE(2)
        MIR_I8_CASE(4)
        MIR_I8_CASE(6)
        default:
            set_error("internal: the int8 filter has no instance for %d k-steps of 64", ix->ks64);
            return MIR_ERR_UNSUPPORTED;
    }
#undef MIR_I8_CASE
#undef MIR_I8_PICK
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

// the float16-native sieve's filter launch (vec_kernels_sieve.h, sieve_h16_kernel)
// BF: the bf16 hi image of a wide float32 shard (ix->d_hi16, ix->ks16), the query fragments of prep_queries16_kernel, `q_err_g`
template <int KIND, bool BF = false>
static int32_t launch_sieve16(const mir_index *ix, const uint4 *qfrag_g, const float *qscale_g, const double *q_norm_g, const double *q_sq_g,
                              int nq, int nwg, uint32_t tile0, uint32_t n_tiles, int nan_guard, const uint64_t *gthr_g, uint64_t *cand,
                              float *candv, uint32_t *ccount, float *part_sample, bool sample, unsigned long long *stat, hipStream_t stream,
                              const double *q_err_g = nullptr) {
    const float *aux = KIND == SCAN_L2 ? ix->d_docsq : KIND == SCAN_COS ? ix->d_invnorm : nullptr;
    const size_t lds = sieve16_lds_bytes();
    const uint32_t n_rows = (uint32_t)ix->n;
    const uint4 *image = BF ? ix->d_hi16 : ix->d_split;
    const int ks = BF ? ix->ks16 : ix->ksteps;
#define MIR_SIEVE16_CASE(KS)                                                                                           \
    do {                                                                                                               \
        auto kern = sample ? sieve_h16_kernel<KS, KIND, true, BF> : sieve_h16_kernel<KS, KIND, false, BF>;             \
        MIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        kern<<<dim3(nwg), dim3(512), lds, stream>>>(image, aux, qfrag_g, qscale_g, q_norm_g, q_sq_g, q_err_g, ix->d_maxnorm, n_rows, tile0, \
                                                    n_tiles, nq, nan_guard, gthr_g, cand, candv, ccount, part_sample, stat); \
    } while (0)
    if (ks == 64) MIR_SIEVE16_CASE(32);
    else if (ks == 32) MIR_SIEVE16_CASE(16);
    else {
        set_error("internal: the float16 sieve has no instance for %d k-steps", ks);
        return MIR_ERR_UNSUPPORTED;
    }
#undef MIR_SIEVE16_CASE
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

// float16-native scan: 128 queries per launch, 16 per wave, one float16 product per fragment (vec_kernels_h16.h)
template <int KIND>
static int32_t launch_scan_h16(const mir_index *ix, const uint4 *qfrag_g, const float *qscale_g, int nq, int klist, int nwg,
                               uint32_t tile0, uint32_t n_tiles, uint64_t *part_g, uint64_t *gthr_g, bool sample, hipStream_t stream) {
    const float *aux = KIND == SCAN_L2 ? ix->d_docsq : KIND == SCAN_COS ? ix->d_invnorm : nullptr;
    const size_t lds = h16_lds_bytes(klist);
    const uint32_t n_rows = (uint32_t)ix->n;
    const int ns = h16_ring_stages(klist);
#define MIR_H16_LAUNCH(KS, NSV)                                                                                        \
    do {                                                                                                               \
        auto kern = sample ? scan_topk_h16_kernel<KS, KIND, true, NSV, H16_QT> : scan_topk_h16_kernel<KS, KIND, false, NSV, H16_QT>;   \
        MIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        kern<<<dim3(nwg), dim3(512 / H16_QT), lds, stream>>>(ix->d_split, aux, qfrag_g, qscale_g, n_rows, tile0, n_tiles, nq, klist, part_g, \
                                                    gthr_g);                                                           \
    } while (0)
    // ksteps = 64: 512 < d <= 1024, 32 k-steps of 32 columns (128 VGPRs of query fragments per wave); 32: 256 < d <= 512
#define MIR_H16_CASES(KS)                                                                                              \
    switch (ns) {                                                                                                      \
        case 4: MIR_H16_LAUNCH(KS, 4); break;                                                                          \
        case 3: MIR_H16_LAUNCH(KS, 3); break;                                                                          \
        default: MIR_H16_LAUNCH(KS, 2); break;                                                                         \
    }
    if (ix->ksteps == 64) {
        MIR_H16_CASES(32)
    } else if (ix->ksteps == 32) {
        MIR_H16_CASES(16)
    } else {
        set_error("internal: float16 scan has no instance for %d k-steps", ix->ksteps);
        return MIR_ERR_UNSUPPORTED;
    }
#undef MIR_H16_CASES
#undef MIR_H16_LAUNCH
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

// float32 rows with 384 < d <= 1024: the 64-query K-split scan over the bf16 hi/lo image
template <int KIND>
static int32_t launch_scan_f16(const mir_index *ix, const uint4 *qfrag_g, const float *qscale_g, int nq, int klist, int nwg,
                               uint32_t n_tiles, uint64_t *part_g, uint64_t *gthr_g, bool sample, hipStream_t stream) {
    const float *aux = KIND == SCAN_L2 ? ix->d_docsq : KIND == SCAN_COS ? ix->d_invnorm : nullptr;
    const size_t lds = f16_lds_bytes(klist);
    const uint32_t n_rows = (uint32_t)ix->n;
#define MIR_SCAN_CASE_SPLIT(KS)                                                                              \
    case KS: {                                                                                               \
        auto kern = sample ? scan_topk_f16_kernel<KS, KIND, true, true> : scan_topk_f16_kernel<KS, KIND, false, true>; \
        MIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
        kern<<<dim3(nwg), dim3(512), lds, stream>>>(ix->d_split, aux, qfrag_g, qscale_g, n_rows, n_tiles,    \
                                                    nq, klist, part_g, gthr_g);                              \
        break;                                                                                               \
    }
    switch (ix->ksteps) {
        MIR_SCAN_CASE_SPLIT(32)
        MIR_SCAN_CASE_SPLIT(48)
        MIR_SCAN_CASE_SPLIT(64)
        default:
            set_error("internal: wide split scan has no instance for %d k-steps", ix->ksteps);
            return MIR_ERR_UNSUPPORTED;
    }
#undef MIR_SCAN_CASE_SPLIT
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

static int32_t check_search_args(const mir_index *ix, const void *queries, int32_t b, int32_t k, int32_t metric,
                                 const int32_t *out_count) {
    MIR_REQUIRE(ix != nullptr, "index is NULL");
    MIR_REQUIRE(b >= 0, "b=%d is negative", b);
    MIR_REQUIRE(b == 0 || queries != nullptr, "queries is NULL");
    MIR_REQUIRE(k >= 1, "k=%d must be >= 1", k);
    MIR_REQUIRE(metric >= 0 && metric <= 3, "unknown metric %d", metric);
    MIR_REQUIRE(b == 0 || out_count != nullptr, "out_count is NULL");
    return MIR_OK;
}

// Enqueue prep + scan(s) + finalize for device-resident queries/outputs.
static int32_t enqueue_search(mir_index *ix, const double *dq, int b, int k, int metric, const SearchBuffers &sb,
                              const SearchPlan &pl, int32_t *o_doc, int64_t *o_chunk, int64_t *o_row,
                              double *o_dist, int32_t *o_count, int32_t *o_flags, hipStream_t stream) {
    const int ngroups = pl.ngroups, nwg = pl.nwg, klist = pl.klist, qpw = pl.qpw;
    const int d = ix->d;
    const int ntiles32 = ngroups * (qpw / 32);  // 32-query fragment tiles, padded to whole launches
    // thresholds of the wide scans + the exact pass's control words: zeroed by the first blocks of the prep kernel
    unsigned long long *gz = reinterpret_cast<unsigned long long *>(sb.gthr);
    const int gwords = sb.ctl_words;
    // (blocks past the fragment and per-query blocks only zero their 64 words and return)
    const int prep_blocks = std::max(ntiles32 * ix->ksteps + b, (gwords + 63) / 64);
    ExactBatchArgs ea;
    ea.docs = ix->d_orig; ea.docs16 = ix->d_f16; ea.doc_sq = ix->d_docsq; ea.dnorm = ix->d_dnorm; ea.n_rows = (uint32_t)ix->n; ea.d = d;
    ea.metric = metric; ea.qt = sb.qt; ea.q_sq = sb.q_sq; ea.nflag = sb.nflag; ea.flagged = sb.flagged;
    ExactArgs es;  // the serial pass's view of the same buffers (one or two flagged queries)
    es.docs = ix->d_orig; es.docs16 = ix->d_f16; es.doc_sq = ix->d_docsq; es.n_rows = (uint32_t)ix->n; es.d = d; es.metric = metric;
    es.q = dq; es.q_sq = sb.q_sq; es.q_norm = sb.q_norm; es.nflag = sb.nflag; es.flagged = sb.flagged;
    auto exact_pass = [&]() {  // device-gated: exits at once when no query was flagged
        es.k = ea.k; es.round = ea.round; es.list_stride = ea.list_stride; es.part = ea.part; es.arrive = ea.arrive;
        es.bound_dist = ea.bound_dist; es.bound_row = ea.bound_row; es.chunk_ids = ea.chunk_ids; es.doc_ids = ea.doc_ids;
        es.row_offset = ea.row_offset; es.out_doc = ea.out_doc; es.out_chunk = ea.out_chunk; es.out_row = ea.out_row;
        es.out_dist = ea.out_dist; es.out_count = ea.out_count; es.out_flags = ea.out_flags;
        const bool cosine = metric == MIR_METRIC_COSINE_SIM;
        if (ix->native16 && cosine) exact_pass_kernel<_Float16, true><<<dim3(pl.exact_grid), dim3(kXbThreads), 0, stream>>>(ea, es);
        else if (ix->native16) exact_pass_kernel<_Float16, false><<<dim3(pl.exact_grid), dim3(kXbThreads), 0, stream>>>(ea, es);
        else if (cosine) exact_pass_kernel<float, true><<<dim3(pl.exact_grid), dim3(kXbThreads), 0, stream>>>(ea, es);
        else exact_pass_kernel<float, false><<<dim3(pl.exact_grid), dim3(kXbThreads), 0, stream>>>(ea, es);
    };
    ea.k = k; ea.round = 0; ea.list_stride = std::min(k, kExactRound); ea.part = sb.part_exact; ea.arrive = sb.arrive;
    ea.bound_dist = sb.bound_dist; ea.bound_row = sb.bound_row; ea.chunk_ids = ix->d_chunk; ea.doc_ids = ix->d_doc;
    ea.row_offset = ix->row_offset; ea.out_doc = o_doc; ea.out_chunk = o_chunk; ea.out_row = o_row; ea.out_dist = o_dist;
    ea.out_count = o_count; ea.out_flags = o_flags;
    if (pl.exact_only) {
        // k beyond the filter's candidate lists: no scan; every query is the reference's own computation.
        // ceil(min(k, n) / 64) rounds, each one pass over the rows per query.
        prep_queries_kernel<<<dim3(std::max(b, (gwords + 63) / 64)), dim3(64), 0, stream>>>(dq, b, d, ix->ksteps, 0, nullptr, sb.q_sq, sb.q_norm, gz, gwords);
        flag_all_kernel<<<dim3(b), dim3(256), 0, stream>>>(b, sb.nflag, sb.flagged, dq, d, metric, sb.q_norm, sb.qt);
        MIR_HIP(hipGetLastError());
        const int64_t found = std::min<int64_t>(k, ix->n);
        const int rounds = (int)std::max<int64_t>(1, (found + kExactRound - 1) / kExactRound);
        for (int r = 0; r < rounds; ++r) {
            ea.round = r;
            exact_pass();
            MIR_HIP(hipGetLastError());
        }
        return MIR_OK;
    }
    // the int8 first stage serves this call: the shard has the image, the metric ranks in the rows' own units
    // ... and k is small: the int8 margin is 4.3 x the bf16 filter's, its lists grow with k (10M x 384: 4.2k candidates per query at k = 10,
    // 7.5k at 20, 11k at 32 - where the first queries overflow their 16384-entry lists and take the exact pass; the bf16 filter
    // lists 3.8k at k = 64) - beyond kI8MaxK the same index's bf16 image serves the call
    const bool use_i8 = ix->i8 && pl.sieve && sb.q_amax != nullptr && k <= kI8MaxK;
    if (use_i8) {
        const int ntiles16 = ngroups * (qpw / 16);
        prep_queries_i8_stats_kernel<<<dim3(std::max(b, (gwords + 63) / 64)), dim3(64), 0, stream>>>(dq, b, d, sb.q_sq, sb.q_norm, sb.q_amax, gz, gwords);
        prep_queries_i8_quant_kernel<<<dim3(ntiles16 * ix->ks64 + b), dim3(64), 0, stream>>>(dq, b, d, ix->ks64, ntiles16, sb.q_amax, sb.qsplit,
                                                                                            sb.q_err, sb.qscale);
    } else if (ix->layout16 && qpw >= kQ16Queries) {
        const int ks32 = ix->ksteps / 2, ntiles16 = ngroups * (qpw / 16);
        prep_queries16_kernel<<<dim3(std::max(ntiles16 * ks32 + b, (gwords + 63) / 64)), dim3(64), 0, stream>>>(
            dq, b, d, ks32, ntiles16, sb.qsplit, sb.q_sq, sb.q_norm, gz, gwords, sb.q_err);
    } else if (ix->wide16 && pl.sieve) {  // bf16 hi (and lo, unread) fragments of 16 queries x 32 columns, |q - bf16(q)|
        const int ks32 = ix->ks16 / 2, ntiles16 = ngroups * (kQ16Queries / 16);
        prep_queries16_kernel<<<dim3(std::max(ntiles16 * ks32 + b, (gwords + 63) / 64)), dim3(64), 0, stream>>>(
            dq, b, d, ks32, ntiles16, sb.qsplit, sb.q_sq, sb.q_norm, gz, gwords, sb.q_err);
    } else if (ix->native16) {
        const int ks32 = ix->ksteps / 2, ntiles16 = ngroups * (kQ16Queries / 16);
        query_stats_h16_kernel<<<dim3(std::max(b, (gwords + 63) / 64)), dim3(64), 0, stream>>>(dq, b, d, sb.q_sq, sb.q_norm, sb.qscale, gz, gwords);
        prep_queries_h16_kernel<<<dim3(ntiles16 * ks32), dim3(64), 0, stream>>>(dq, b, d, ks32, sb.qscale, sb.qsplit);
    } else
        prep_queries_kernel<<<dim3(prep_blocks), dim3(64), 0, stream>>>(dq, b, d, ix->ksteps, ntiles32,
                                                                                    sb.qsplit, sb.q_sq, sb.q_norm, gz, gwords);
    MIR_HIP(hipGetLastError());
    for (int g = 0; g < ngroups; ++g) {
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        auto begin_profile = [&]() -> int32_t {  // brackets the dominant (full-shard) scan launch only
            if (!ix->profiling) return MIR_OK;
            {
                std::lock_guard<std::mutex> lk(ix->mu);
                if (ix->prof_free.size() >= 2) {
                    ev0 = ix->prof_free.back(); ix->prof_free.pop_back();
                    ev1 = ix->prof_free.back(); ix->prof_free.pop_back();
                }
            }
            if (!ev0) {
                MIR_HIP(hipEventCreate(&ev0));
                MIR_HIP(hipEventCreate(&ev1));
            }
            MIR_HIP(hipEventRecord(ev0, stream));
            return MIR_OK;
        };
        const uint4 *qs = sb.qsplit + (size_t)g * (qpw / 32) * ix->ksteps * 128;
        uint64_t *pg = sb.part + (size_t)g * nwg * qpw * klist;
        const int nq = std::min(qpw, b - qpw * g);
        int32_t rc;
        if (pl.sieve) {
            // filter (hi blocks only) -> scatter to the queries' lists -> select (reference formula for the rows that can be among
            // the first k): exact by construction
            uint64_t *gt = sb.gthr + (size_t)g * std::max(128, qpw);
            const int q0 = qpw * g;
            const double *qn = sb.q_norm + q0, *qsq = sb.q_sq + q0, *qerr = sb.q_err + q0;
            const int guard = metric == MIR_METRIC_EUCLIDEAN_DIST ? 1 : 0;
            const uint4 *qs16 = sb.qsplit + (size_t)g * (kQ16Queries / 16) * (ix->ksteps / 2) * 64;  // native16: hi fragments only
            const float *qsc = sb.qscale + (size_t)g * qpw;
            auto sieve = [&](uint32_t t0, uint32_t nt, int wgs, uint64_t *cand, float *cv, uint32_t *cc, bool smp) {
                float *ps = reinterpret_cast<float *>(sb.part_sample);
                unsigned long long *st = smp ? nullptr : ix->d_stats + (t0 ? 1 : 0);
                if (use_i8) {  // tiles -> 64-row stages: [t0 / 2, (t0 + nt + 1) / 2) (t0 is even or 0 wherever two launches meet: see below)
                    const uint32_t s0 = t0 / 2, s1 = std::min<uint32_t>((t0 + nt + 1) / 2, ix->n_stages);
                    const uint4 *qf = sb.qsplit + (size_t)g * (qpw / 16) * ix->ks64 * 64;
                    if (metric == MIR_METRIC_INNER_PRODUCT)
                        return launch_sieve_i8<SCAN_IP>(ix, qpw, sb.qscale + q0, qf, qn, qsq, qerr, nq, wgs, s0, s1 - s0, guard, gt, cand, cv, cc, ps, smp, st, stream);
                    if (metric == MIR_METRIC_COSINE_SIM)
                        return launch_sieve_i8<SCAN_COS>(ix, qpw, sb.qscale + q0, qf, qn, qsq, qerr, nq, wgs, s0, s1 - s0, guard, gt, cand, cv, cc, ps, smp, st, stream);
                    return launch_sieve_i8<SCAN_L2>(ix, qpw, sb.qscale + q0, qf, qn, qsq, qerr, nq, wgs, s0, s1 - s0, guard, gt, cand, cv, cc, ps, smp, st, stream);
                }
                if (ix->native16) {
                    if (metric == MIR_METRIC_INNER_PRODUCT) return launch_sieve16<SCAN_IP>(ix, qs16, qsc, qn, qsq, nq, wgs, t0, nt, guard, gt, cand, cv, cc, ps, smp, st, stream);
                    if (metric == MIR_METRIC_COSINE_SIM) return launch_sieve16<SCAN_COS>(ix, qs16, qsc, qn, qsq, nq, wgs, t0, nt, guard, gt, cand, cv, cc, ps, smp, st, stream);
                    return launch_sieve16<SCAN_L2>(ix, qs16, qsc, qn, qsq, nq, wgs, t0, nt, guard, gt, cand, cv, cc, ps, smp, st, stream);
                }
                if (ix->wide16) {  // (fragments of group g: 8 query tiles x ks16 / 2 k-steps x (hi, lo) blocks)
                    const uint4 *qw = sb.qsplit + (size_t)g * (kQ16Queries / 16) * (ix->ks16 / 2) * 128;
                    if (metric == MIR_METRIC_INNER_PRODUCT) return launch_sieve16<SCAN_IP, true>(ix, qw, nullptr, qn, qsq, nq, wgs, t0, nt, guard, gt, cand, cv, cc, ps, smp, st, stream, qerr);
                    if (metric == MIR_METRIC_COSINE_SIM) return launch_sieve16<SCAN_COS, true>(ix, qw, nullptr, qn, qsq, nq, wgs, t0, nt, guard, gt, cand, cv, cc, ps, smp, st, stream, qerr);
                    return launch_sieve16<SCAN_L2, true>(ix, qw, nullptr, qn, qsq, nq, wgs, t0, nt, guard, gt, cand, cv, cc, ps, smp, st, stream, qerr);
                }
                if (metric == MIR_METRIC_INNER_PRODUCT) return launch_sieve<SCAN_IP>(ix, qpw, qs, qn, qsq, qerr, nq, wgs, t0, nt, guard, gt, cand, cv, cc, ps, smp, st, stream);
                if (metric == MIR_METRIC_COSINE_SIM) return launch_sieve<SCAN_COS>(ix, qpw, qs, qn, qsq, qerr, nq, wgs, t0, nt, guard, gt, cand, cv, cc, ps, smp, st, stream);
                return launch_sieve<SCAN_L2>(ix, qpw, qs, qn, qsq, qerr, nq, wgs, t0, nt, guard, gt, cand, cv, cc, ps, smp, st, stream);
            };
            SieveScatterArgs ca;
            ca.q0 = q0; ca.nq = nq; ca.l = sb.sv;
            SieveSelectArgs sa;
            sa.l = sb.sv; sa.q0 = q0; sa.nq = nq; sa.k = k; sa.metric = metric; sa.d = d; sa.nan_guard = guard;
            sa.rel_err = ix->native16 ? (float)kH16RelErr : kHiHiRelErr;
            sa.extra_slop = ix->wide16 ? wide_accum_slop(ix->ks16 * 16) : 0.f;
            sa.docs = ix->d_orig; sa.docs16 = ix->d_f16; sa.doc_sq = ix->d_docsq;
            sa.dnorm = (ix->native16 || !ix->norms_spread) ? nullptr : ix->d_dnorm;
            sa.q = dq; sa.q_sq = sb.q_sq; sa.q_norm = sb.q_norm; sa.max_norm = ix->d_maxnorm;
            sa.q_err = ix->native16 ? nullptr : sb.q_err;
            sa.i8_qscale = nullptr;
            sa.i8_tparam = nullptr;
            if (use_i8) {  // the margin of the int8 filter's values: its statistics, its query residuals and scales (i8_margin)
                sa.max_norm = ix->d_i8stats;
                sa.i8_qscale = sb.qscale;
                sa.i8_tparam = ix->d_i8tp;
                sa.dnorm = nullptr;
            }
            sa.gthr = reinterpret_cast<unsigned long long *>(gt);
            sa.chunk_ids = ix->d_chunk; sa.doc_ids = ix->d_doc; sa.row_offset = ix->row_offset;
            sa.out_doc = o_doc; sa.out_chunk = o_chunk; sa.out_row = o_row; sa.out_dist = o_dist; sa.out_count = o_count;
            sa.out_flags = o_flags; sa.nflag = sb.nflag; sa.flagged = sb.flagged; sa.stats = ix->d_stats; sa.qt = sb.qt;
            rc = sieve(0, (uint32_t)kSampleWgs * pl.sample_tpw, kSampleWgs, nullptr, nullptr, nullptr, true);
            if (rc != MIR_OK) return rc;
            sample_threshold_kernel<<<dim3(nq), dim3(256), 0, stream>>>(reinterpret_cast<const float *>(sb.part_sample), kSampleWgs, qpw,
                                                                        k, nq, reinterpret_cast<unsigned long long *>(gt));
            MIR_HIP(hipGetLastError());
            rc = begin_profile();
            if (rc != MIR_OK) return rc;
            for (int phase = pl.tiles_first ? 0 : 1; phase < 2 && rc == MIR_OK; ++phase) {  // (one launch: the final phase alone)
                const size_t region = use_i8 ? (size_t)kI8Region : (size_t)kSieveRegion;
                uint64_t *cand = sb.sv_cand + (size_t)phase * nwg * region;
                float *cv = sb.sv_candv + (size_t)phase * nwg * region;
                uint32_t *cc = sb.sv_ccount + (size_t)phase * nwg * (use_i8 ? 8 : 1);
                rc = phase == 0 ? sieve(0, pl.tiles_first, nwg, cand, cv, cc, false)
                                : sieve(pl.tiles_first, ix->n_tiles - pl.tiles_first, nwg, cand, cv, cc, false);
                if (rc != MIR_OK) break;
                if (phase == 1 && ev0) {  // the bracket: both filter launches and what runs between them
                    MIR_HIP(hipEventRecord(ev1, stream));
                    std::lock_guard<std::mutex> lk(ix->mu);
                    ix->prof_events.emplace_back(ev0, ev1);
                    ev1 = nullptr;
                }
                ca.cand = cand; ca.candv = cv; ca.ccount = cc;
                if (use_i8) sieve_scatter_i8_kernel<<<dim3(nwg * 8), dim3(256), 0, stream>>>(ca);
                else sieve_scatter_kernel<<<dim3(nwg * kSieveScatterSplit), dim3(256), 0, stream>>>(ca);
                sa.mode = phase;
                MIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sieve_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)sieve_select_lds_bytes()));
                sieve_select_kernel<<<dim3(nq), dim3(kSieveSelectThreads), sieve_select_lds_bytes(), stream>>>(sa);
                MIR_HIP(hipGetLastError());
            }
            if (rc != MIR_OK) return rc;
            continue;
        }
        if (qpw != 32) {
            uint64_t *gt = sb.gthr + (size_t)g * 128;
            const float *qsc = sb.qscale + (size_t)g * qpw;
            auto run = [&](int wgs, uint32_t tiles, uint64_t *out, bool sample) {
                if (ix->layout16 || ix->native16) {
                    const double *qn = sb.q_norm + (size_t)g * qpw;
                    const uint4 *qs16 = sb.qsplit + (size_t)g * (kQ16Queries / 16) * (ix->ksteps / 2) * 64;  // native16: hi fragments only
                    auto one = [&](int w, uint32_t t0, uint32_t nt, uint64_t *o, bool smp) {
                        if (ix->native16) {
                            if (metric == MIR_METRIC_INNER_PRODUCT) return launch_scan_h16<SCAN_IP>(ix, qs16, qsc, nq, klist, w, t0, nt, o, gt, smp, stream);
                            if (metric == MIR_METRIC_COSINE_SIM) return launch_scan_h16<SCAN_COS>(ix, qs16, qsc, nq, klist, w, t0, nt, o, gt, smp, stream);
                            return launch_scan_h16<SCAN_L2>(ix, qs16, qsc, nq, klist, w, t0, nt, o, gt, smp, stream);
                        }
                        if (metric == MIR_METRIC_INNER_PRODUCT) return launch_scan_q16<SCAN_IP>(ix, qs, qn, nq, klist, w, t0, nt, o, gt, smp, stream);
                        if (metric == MIR_METRIC_COSINE_SIM) return launch_scan_q16<SCAN_COS>(ix, qs, qn, nq, klist, w, t0, nt, o, gt, smp, stream);
                        return launch_scan_q16<SCAN_L2>(ix, qs, qn, nq, klist, w, t0, nt, o, gt, smp, stream);
                    };
                    if (sample || pl.nwg_first == 0) return one(wgs, 0, tiles, out, sample);
                    // Progressive scan: the first 1/16 of the tiles with the sample's thresholds, then the klist-th best of
                    // what that found as the threshold of the rest (list_threshold_kernel): the rest runs its correction
                    // products for ~1 % of its blocks instead of ~25 %.  Lists of both launches go to finalize.
                    int32_t r1 = one(pl.nwg_first, 0, pl.tiles_first, out, false);
                    if (r1 != MIR_OK) return r1;
                    list_threshold_kernel<<<dim3(nq), dim3(256), 0, stream>>>(out, pl.nwg_first, qpw, klist, nq,
                                                                             reinterpret_cast<unsigned long long *>(gt));
                    return one(wgs - pl.nwg_first, pl.tiles_first, tiles - pl.tiles_first, out + (size_t)pl.nwg_first * qpw * klist, false);
                }
                if (wide64_split(ix)) {
                    if (metric == MIR_METRIC_INNER_PRODUCT) return launch_scan_f16<SCAN_IP>(ix, qs, qsc, nq, klist, wgs, tiles, out, gt, sample, stream);
                    if (metric == MIR_METRIC_COSINE_SIM) return launch_scan_f16<SCAN_COS>(ix, qs, qsc, nq, klist, wgs, tiles, out, gt, sample, stream);
                    return launch_scan_f16<SCAN_L2>(ix, qs, qsc, nq, klist, wgs, tiles, out, gt, sample, stream);
                }
                set_error("internal: no wide scan for this index");
                return (int32_t)MIR_ERR_UNSUPPORTED;
            };
            // Scan a sample of the shard first and seed every query's threshold with a lower bound of its
            // klist-th best value (sample_threshold_kernel), so the full pass starts with tight thresholds
            // instead of accepting almost everything for its first ~50 tiles per workgroup.
            // Measured: the pre-pass pays from ~32K rows up (500K rows: 0.48 -> 0.24 ms per step; 1M: 0.69 ->
            // 0.39; 10M: see DESIGN.md), 4 tiles per workgroup being the best size from 100K to 10M rows; it
            // must stay a small part of the shard, so smaller shards sample fewer tiles or skip it
            const uint32_t tpw = std::min<uint32_t>(kSampleTilesPerWg, ix->n_tiles / (4u * kSampleWgs));
            const uint32_t sample_tiles = (uint32_t)kSampleWgs * tpw;
            rc = MIR_OK;
            if (tpw >= 1) {
                rc = run(kSampleWgs, sample_tiles, sb.part_sample, true);
                if (rc == MIR_OK) {
                    sample_threshold_kernel<<<dim3(nq), dim3(256), 0, stream>>>(reinterpret_cast<const float *>(sb.part_sample),
                                                                                kSampleWgs, qpw, klist, nq,
                                                                                reinterpret_cast<unsigned long long *>(gt));
                    MIR_HIP(hipGetLastError());
                }
            }
            if (rc == MIR_OK) rc = begin_profile();
            if (rc == MIR_OK) rc = run(nwg, ix->n_tiles, pg, false);
        } else {
            rc = begin_profile();
            if (rc != MIR_OK) return rc;
            if (metric == MIR_METRIC_INNER_PRODUCT) rc = launch_scan<SCAN_IP>(ix, qs, nq, klist, nwg, pg, stream);
            else if (metric == MIR_METRIC_COSINE_SIM) rc = launch_scan<SCAN_COS>(ix, qs, nq, klist, nwg, pg, stream);
            else rc = launch_scan<SCAN_L2>(ix, qs, nq, klist, nwg, pg, stream);
        }
        if (ev1) {
            (void)hipEventRecord(ev1, stream);
            std::lock_guard<std::mutex> lk(ix->mu);
            ix->prof_events.emplace_back(ev0, ev1);
        }
        if (rc != MIR_OK) return rc;
    }
    if (pl.sieve) {  // the select kernel wrote the results; queries whose buffers overflowed take the exact pass
        exact_pass();
        MIR_HIP(hipGetLastError());
        return MIR_OK;
    }
    FinalizeArgs fa;
    fa.part = sb.part; fa.nwg = nwg; fa.qpw = qpw; fa.klist = klist; fa.k = k; fa.b = b; fa.d = d; fa.metric = metric;
    fa.docs = ix->d_orig; fa.docs16 = ix->d_f16; fa.doc_sq = ix->d_docsq; fa.max_norm = ix->d_maxnorm;
    fa.scan_rel_err = ix->native16 ? kH16RelErr : scan_rel_err(d);  // the bound of the scan whose values the lists hold
    fa.q = dq; fa.q_sq = sb.q_sq; fa.q_norm = sb.q_norm;
    fa.chunk_ids = ix->d_chunk; fa.doc_ids = ix->d_doc; fa.row_offset = ix->row_offset;
    fa.out_doc = o_doc; fa.out_chunk = o_chunk; fa.out_row = o_row; fa.out_dist = o_dist;
    fa.out_count = o_count; fa.out_flags = o_flags; fa.nflag = sb.nflag; fa.flagged = sb.flagged; fa.qt = sb.qt;
    finalize_kernel<<<dim3(b), dim3(256), 0, stream>>>(fa);
    MIR_HIP(hipGetLastError());
    // queries whose candidate set finalize could not prove complete: exact pass, gated on the device (it exits at
    // once when there are none)
    exact_pass();
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

static int32_t plan(const mir_index *ix, int b, int k, SearchPlan *pl) {
    // the exact pass: one workgroup of 16 waves per CU, one per 64-row block on small shards; its per-workgroup lists are
    // b x grid x min(k, 64) x 16 bytes of workspace - fewer workgroups for huge batches
    {
        const int64_t per_wg = (int64_t)b * std::min(k, kExactRound) * 16;
        const int64_t by_mem = std::max<int64_t>(64, ((int64_t)256 << 20) / std::max<int64_t>(per_wg, 1));
        pl->exact_grid = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ix->num_cus, by_mem), (ix->n + kXbRows - 1) / kXbRows));
    }
    pl->klist = std::min(k + (ix->native16 ? kH16ListMargin : kListMargin), kMaxList);
    // The reference takes any `limit` (embeddings_index.py:58,81).  Beyond what the filter's per-lane candidate
    // lists hold (LDS), an index that could actually return that many rows is searched by the exact pass alone.
    static const bool sieve_off = getenv("MIR_NO_SIEVE") != nullptr;  // (A/B measurements against the round-2 scan)
    static const bool sieve16_off = getenv("MIR_NO_SIEVE16") != nullptr;
    if (((ix->layout16 && !sieve_off) || (ix->native16 && !sieve16_off && !sieve_off) || ix->wide16) && k <= kSieveMaxK) {
        // large shards (the progressive scan's: >= 64 tiles per workgroup): filter on the hi blocks alone, verify every candidate
        const int wgs = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(ix->num_cus, 1024), (int64_t)ix->n_tiles));
        if (ix->n_tiles >= 4u * kSampleWgs) {  // from 32K rows: there is a sample to take the first threshold from
            pl->sieve = true;
            // 128 queries per filter launch, or 256 (two query tiles per wave) when the batch has more: the stream of a pass is
            // the same, the matrix work doubles (MIR_SIEVE_QPL=128 keeps 128)
            static const int qpl_max = getenv("MIR_SIEVE_QPL") ? atoi(getenv("MIR_SIEVE_QPL")) : 256;
            pl->qpw = (ix->layout16 && b > kQ16Queries && qpl_max >= 256) ? 2 * kQ16Queries : kQ16Queries;
            pl->ngroups = (b + pl->qpw - 1) / pl->qpw;
            pl->nwg = wgs;
            pl->klist = std::min(k, kMaxList);
            // shards of >= 64 tiles per workgroup in two filter launches (1/16 of the tiles, but not fewer than ~157K rows
            // nor more than 1/4, then the rest with the exact k-th best so far as the threshold); smaller ones in one
            // (experiments: MIR_SIEVE_TWO_PHASE_TILES = shards with at least this many tiles per workgroup take two launches;
            //  MIR_SIEVE_SAMPLE_TPW = tiles per sample workgroup on one-launch shards)
            static const int two_phase_tiles = getenv("MIR_SIEVE_TWO_PHASE_TILES") ? atoi(getenv("MIR_SIEVE_TWO_PHASE_TILES")) : 64;
            static const int single_tpw = getenv("MIR_SIEVE_SAMPLE_TPW") ? atoi(getenv("MIR_SIEVE_SAMPLE_TPW")) : kSampleTilesPerWg;
            const bool two = (int64_t)ix->n_tiles >= (int64_t)two_phase_tiles * wgs;
            static const uint32_t first_div = getenv("MIR_SIEVE_FIRST_DIV") ? (uint32_t)atoi(getenv("MIR_SIEVE_FIRST_DIV")) : 16u;  // (experiments)
            pl->tiles_first = two ? std::max<uint32_t>(ix->n_tiles / first_div, std::min<uint32_t>(ix->n_tiles / 4, 4896u)) : 0;
            pl->tiles_first &= ~1u;  // (the int8 filter walks 64-row stages: the two launches meet at a stage boundary)
            static const int two_tpw = getenv("MIR_SIEVE_SAMPLE_TPW2") ? atoi(getenv("MIR_SIEVE_SAMPLE_TPW2")) : kSampleTilesPerWg;  // (experiments)
            pl->sample_tpw = std::max<uint32_t>(1, std::min<uint32_t>(two ? two_tpw : single_tpw, ix->n_tiles / (4u * kSampleWgs)));
            return MIR_OK;
        }
    }
    const bool wide64 = wide64_split(ix);
    const bool lists_fit = k + (ix->native16 ? kH16ListMargin : kListMargin) <= kMaxList;
    if (!lists_fit && ix->n > (int64_t)kMaxList) {  // (n <= 64: every row fits the lists)
        pl->exact_only = true;
        pl->ngroups = 0;
        pl->qpw = 32;
        pl->nwg = 1;
        return MIR_OK;
    }
    if (wide64 && f16_lds_bytes(pl->klist) <= 160 * 1024) {  // 64 queries per pass; its LDS holds lists up to k = 28 (a float32
        pl->qpw = kF16Queries;                                  // index with a larger k falls through to the 32-query kernels)
        pl->ngroups = (b + pl->qpw - 1) / pl->qpw;
        const int64_t want16 = (int64_t)ix->n_tiles;
        pl->nwg = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(ix->num_cus, 1024), want16));
        return MIR_OK;
    }
    if ((ix->layout16 || ix->native16) && pl->klist > kQ16MaxList) {  // k > 52 (48 on a float16-native index) on a 16-queries-per-wave index: the exact pass alone (any n)
        pl->exact_only = true;
        pl->ngroups = 0;
        pl->qpw = 32;
        pl->nwg = 1;
        return MIR_OK;
    }
    if (ix->layout16 || ix->native16) {  // the images only the 16-queries-per-wave kernels read: they take every k their buffers hold (k <= 52; 48 on a float16-native index)
        pl->qpw = kQ16Queries;
        pl->ngroups = (b + pl->qpw - 1) / pl->qpw;
        pl->nwg = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(ix->num_cus, 1024), (int64_t)ix->n_tiles));
        // Shards of >= 64 tiles per workgroup are scanned in two launches (1/16 first, re-seed, the rest).  Measured per
        // 128-query step (1.25M / 2.5M / 5M rows): two launches 0.427 / 0.689 / 1.240 ms, one launch 0.441 / 0.773 / 1.496;
        // first-launch fractions 1/8, 1/32, 1/64 on 10M rows: 2.351 / 2.346 / 2.445 ms against 2.324 at 1/16.
        if ((int64_t)ix->n_tiles >= 64 * (int64_t)pl->nwg) {
            // 1/16 of the tiles, but not fewer than ~157K rows (their klist-th best is what makes the second launch's
            // thresholds tight: 1.25M rows measured 0.429 ms per step with 78K rows first, 0.411 with 156K) nor more than 1/4
            pl->tiles_first = std::max<uint32_t>(ix->n_tiles / 16, std::min<uint32_t>(ix->n_tiles / 4, 4896u));
            pl->nwg_first = pl->nwg;
            pl->nwg = 2 * pl->nwg;  // lists for finalize
        }
        return MIR_OK;
    }
    pl->qpw = 32;  // the register-ring kernels: d <= 64, d > 1024, or a float32 wide index with k > 28
    pl->ngroups = (b + pl->qpw - 1) / pl->qpw;
    // one workgroup per CU, or one per tile on shards smaller than that: a workgroup's fixed costs
    // (ring start-up, filling empty lists) grow with the tiles it walks, and on a 1k-5k-row index one
    // tile per workgroup takes a single search from 128 to 90 us
    const int64_t want = (int64_t)ix->n_tiles;
    // (finalize's tournament gives each of its 256 threads up to 4 workgroup lists)
    pl->nwg = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(ix->num_cus, 1024), want));
    return MIR_OK;
}

}  // namespace mir

// ============================================================ C ABI

// ---- composition of an index from per-document row blocks (all in HBM) ----
struct ConcatItem {
    const void *src;    // first unit of this piece
    int64_t dst_unit;   // where it goes, in units of the destination type
    int32_t n_units;    // <= kConcatPiece
    int32_t doc_id;     // for the doc-id fill (pieces are in row units there)
};
constexpr int kConcatPiece = 1 << 18;

template <typename U>
__global__ __launch_bounds__(256) void concat_kernel(const ConcatItem *__restrict__ items, U *__restrict__ dst) {
    const ConcatItem it = items[blockIdx.x];
    const U *src = static_cast<const U *>(it.src);
    U *out = dst + it.dst_unit;
    for (int i = threadIdx.x; i < it.n_units; i += 256) out[i] = src[i];
}
__global__ __launch_bounds__(256) void fill_doc_kernel(const ConcatItem *__restrict__ items, int32_t *__restrict__ dst) {
    const ConcatItem it = items[blockIdx.x];
    int32_t *out = dst + it.dst_unit;
    for (int i = threadIdx.x; i < it.n_units; i += 256) out[i] = it.doc_id;
}

// Where the rows of a new index come from: one matrix (host or device), or row blocks in HBM.
struct RowSource {
    const void *emb = nullptr;
    bool on_device = false;
    const int64_t *chunk_ids = nullptr;
    const int32_t *doc_ids = nullptr;
    const mir_rows *const *parts = nullptr;
    const int32_t *part_doc_ids = nullptr;
    int32_t nparts = 0;
};

// pieces of at most kConcatPiece units covering every part; unit_per_row units per row
static std::vector<ConcatItem> concat_items(const RowSource &src, int64_t units_per_row, size_t unit_bytes, int which) {
    std::vector<ConcatItem> items;
    int64_t row0 = 0;
    for (int p = 0; p < src.nparts; ++p) {
        const mir_rows *r = src.parts[p];
        const char *base = which == 0 ? static_cast<const char *>(r->d_emb) : reinterpret_cast<const char *>(r->d_chunk);
        const int64_t total = r->n * units_per_row;
        for (int64_t u = 0; u < total; u += kConcatPiece)
            items.push_back({base ? base + (size_t)u * unit_bytes : nullptr, row0 * units_per_row + u,
                             (int32_t)std::min<int64_t>(kConcatPiece, total - u), src.part_doc_ids ? src.part_doc_ids[p] : p});
        row0 += r->n;
    }
    return items;
}

template <typename Launch>
static hipError_t run_items(const std::vector<ConcatItem> &items, hipStream_t stream, Launch launch) {
    if (items.empty()) return hipSuccess;
    ConcatItem *d_items = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_items), items.size() * sizeof(ConcatItem));
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(ConcatItem), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        launch(d_items, (unsigned)items.size());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);  // `items` (pageable) and d_items die here
    (void)hipFree(d_items);
    return e;
}

// rows of `src` -> dst (n x d elements of `esize` bytes)
static hipError_t copy_rows(const RowSource &src, void *dst, int64_t n, int32_t d, size_t esize, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (!src.parts)
        return hipMemcpyAsync(dst, src.emb, (size_t)n * d * esize, src.on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream);
    const std::vector<ConcatItem> items = concat_items(src, d, esize, 0);
    return run_items(items, stream, [&](const ConcatItem *di, unsigned ni) {
        if (esize == 4) concat_kernel<uint32_t><<<dim3(ni), dim3(256), 0, stream>>>(di, static_cast<uint32_t *>(dst));
        else concat_kernel<uint16_t><<<dim3(ni), dim3(256), 0, stream>>>(di, static_cast<uint16_t *>(dst));
    });
}

extern "C" {

int32_t mir_abi_version(void) { return MIR_ABI_VERSION; }
const char *mir_last_error(void) { return mir::g_err; }

int32_t mir_device_count(int32_t *out_count) {
    MIR_REQUIRE(out_count != nullptr, "out_count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *out_count = c;
    return MIR_OK;
}

static int32_t create_common(const RowSource &src, int64_t n, int32_t d, int32_t dtype,
                             int32_t device, int64_t row_offset, hipStream_t stream, mir_index **out) {
    int32_t rc = check_create_args(n, d, dtype, out);
    if (rc != MIR_OK) return rc;
    MIR_REQUIRE(n == 0 || src.emb != nullptr || src.parts != nullptr, "embeddings pointer is NULL");
    const int64_t *chunk_ids = src.chunk_ids;
    const int32_t *doc_ids = src.doc_ids;
    int cus = 0;
    rc = use_device(device, &cus);
    if (rc != MIR_OK) return rc;
    mir_index *ix = new (std::nothrow) mir_index();
    MIR_REQUIRE(ix != nullptr, "out of host memory");
    ix->device = device; ix->num_cus = cus; ix->n = n; ix->d = d; ix->dtype = dtype; ix->row_offset = row_offset;
    const hipMemcpyKind kind = src.on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    auto fail = [&](int32_t code) {
        free_index(ix);
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
    const size_t orig_bytes = (size_t)n * d * sizeof(float);
    // float16 rows with 256 < d <= 1024: kept as they are and scanned as 2-byte fragments, columns
    // zero-padded to 512 or 1024 (vec_kernels_h16.h); everything else goes through the float32 / bf16-split
    // layout (at d <= 256 the padding to 512 columns would scan as many bytes as that does)
    ix->native16 = dtype == MIR_DTYPE_F16 && d > 256 && d <= 1024;
    if (ix->native16) {
        MIR_TRY(hipMalloc(reinterpret_cast<void **>(&ix->d_f16), std::max<size_t>(orig_bytes / 2, 16)));
        ix->hbm_bytes += orig_bytes / 2;
        MIR_TRY(copy_rows(src, ix->d_f16, n, d, 2, stream));
    } else {
    MIR_TRY(hipMalloc(&ix->d_orig, std::max<size_t>(orig_bytes, 16)));
    ix->hbm_bytes += orig_bytes;
    }
    if (ix->native16) {
    } else if (n > 0 && dtype == MIR_DTYPE_F32) {
        MIR_TRY(copy_rows(src, ix->d_orig, n, d, 4, stream));
    } else if (n > 0) {
        // float16 input: widened exactly to float32 on the device.  Every float16 is hi + lo in
        // bfloat16 exactly (11 significant bits <= 8 + 8), so the scan is EXACT on such an index.
        void *tmp = nullptr;
        MIR_TRY(hipMalloc(&tmp, orig_bytes / 2));
        hipError_t e1 = copy_rows(src, tmp, n, d, 2, stream);
        if (e1 == hipSuccess) {
            const int64_t total = n * (int64_t)d;
            widen_f16_kernel<<<dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20)), dim3(256), 0, stream>>>(
                static_cast<const _Float16 *>(tmp), total, ix->d_orig);
            e1 = hipGetLastError();
        }
        if (e1 == hipSuccess) e1 = hipStreamSynchronize(stream);
        (void)hipFree(tmp);
        MIR_TRY(e1);
    }
    if (src.parts && n > 0) {  // chunk ids of the blocks, doc id = the block's
        MIR_TRY(hipMalloc(&ix->d_chunk, (size_t)n * 8));
        MIR_TRY(hipMalloc(&ix->d_doc, (size_t)n * 4));
        ix->hbm_bytes += (size_t)n * 12;
        const std::vector<ConcatItem> items = concat_items(src, 1, 8, 1);
        MIR_TRY(run_items(items, stream, [&](const ConcatItem *di, unsigned ni) {
            concat_kernel<uint64_t><<<dim3(ni), dim3(256), 0, stream>>>(di, reinterpret_cast<uint64_t *>(ix->d_chunk));
            fill_doc_kernel<<<dim3(ni), dim3(256), 0, stream>>>(di, ix->d_doc);
        }));
    }
    if (chunk_ids && n > 0) {
        MIR_TRY(hipMalloc(&ix->d_chunk, (size_t)n * 8));
        MIR_TRY(hipMemcpyAsync(ix->d_chunk, chunk_ids, (size_t)n * 8, kind, stream));
        ix->hbm_bytes += (size_t)n * 8;
    }
    if (doc_ids && n > 0) {
        MIR_TRY(hipMalloc(&ix->d_doc, (size_t)n * 4));
        MIR_TRY(hipMemcpyAsync(ix->d_doc, doc_ids, (size_t)n * 4, kind, stream));
        ix->hbm_bytes += (size_t)n * 4;
    }
#undef MIR_TRY
    rc = build_derived(ix, stream);
    if (rc != MIR_OK) return fail(rc);
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        set_error("index build failed: %s", hipGetErrorString(e));
        return fail(MIR_ERR_HIP);
    }
    if (ix->d_tilemax) {  // (the build is complete: two floats of its norm statistics decide the sieve's margin form)
        float st[8] = {};
        e = hipMemcpy(st, ix->d_maxnorm, 32, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            set_error("index build failed: %s", hipGetErrorString(e));
            return fail(MIR_ERR_HIP);
        }
        ix->norms_spread = !(st[0] <= 1.0625f * st[4]) || getenv("MIR_SIEVE_PER_TILE") != nullptr;  // (NaN / infinite norms: spread)
        // the int8 first stage (vec_kernels_i8.h) serves the squared-L2 / euclidean / inner-product searches of a shard of finite rows
        // of one norm; MIR_SIEVE_I8=0 keeps the bf16 filter for everything (read per build: tests and A/B runs switch it)
        const bool i8_on = !(getenv("MIR_SIEVE_I8") != nullptr && atoi(getenv("MIR_SIEVE_I8")) == 0);
        unsigned int nonfinite = 0;
        memcpy(&nonfinite, &st[1], 4);
        if (ix->hi_only && !ix->norms_spread && nonfinite == 0u && st[0] > 0.f && i8_on) {
            rc = build_i8(ix, stream);
            if (rc != MIR_OK) return fail(rc);
        }
    }
    *out = ix;
    return MIR_OK;
}

int32_t mir_index_create(const void *emb_host, int64_t n, int32_t d, int32_t dtype, const int64_t *chunk_ids_host,
                         const int32_t *doc_ids_host, int32_t device, int64_t row_offset, mir_index **out) {
    RowSource src;
    src.emb = emb_host; src.chunk_ids = chunk_ids_host; src.doc_ids = doc_ids_host;
    return create_common(src, n, d, dtype, device, row_offset, nullptr, out);
}

int32_t mir_index_create_from_device(const void *emb_device, int64_t n, int32_t d, int32_t dtype,
                                     const int64_t *chunk_ids_device, const int32_t *doc_ids_device,
                                     int32_t device, int64_t row_offset, void *stream, mir_index **out) {
    RowSource src;
    src.emb = emb_device; src.on_device = true; src.chunk_ids = chunk_ids_device; src.doc_ids = doc_ids_device;
    return create_common(src, n, d, dtype, device, row_offset, static_cast<hipStream_t>(stream), out);
}

// ---- row blocks ----
int32_t mir_rows_create(const void *emb_host, int64_t n, int32_t d, int32_t dtype, const int64_t *chunk_ids_host,
                        int32_t device, mir_rows **out) {
    mir_index *dummy = nullptr;
    int32_t rc = check_create_args(n, d, dtype, &dummy);
    if (rc != MIR_OK) return rc;
    MIR_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    MIR_REQUIRE(n == 0 || emb_host != nullptr, "embeddings pointer is NULL");
    rc = use_device(device, nullptr);
    if (rc != MIR_OK) return rc;
    mir_rows *r = new (std::nothrow) mir_rows();
    MIR_REQUIRE(r != nullptr, "out of host memory");
    r->device = device; r->n = n; r->d = d; r->dtype = dtype;
    const size_t eb = (size_t)n * d * (dtype == MIR_DTYPE_F16 ? 2 : 4);
    hipError_t e = hipSuccess;
    if (n > 0) {
        e = hipMalloc(&r->d_emb, eb);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&r->d_chunk), (size_t)n * 8);
        if (e == hipSuccess) e = hipMemcpy(r->d_emb, emb_host, eb, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            if (chunk_ids_host) {
                e = hipMemcpy(r->d_chunk, chunk_ids_host, (size_t)n * 8, hipMemcpyHostToDevice);
            } else {  // NULL = 0..n-1, as mir_index_create
                std::vector<int64_t> ar((size_t)n);
                for (int64_t i = 0; i < n; ++i) ar[(size_t)i] = i;
                e = hipMemcpy(r->d_chunk, ar.data(), (size_t)n * 8, hipMemcpyHostToDevice);
            }
        }
        r->hbm_bytes = (int64_t)eb + n * 8;
    }
    if (e != hipSuccess) {
        set_error("mir_rows_create: %s", hipGetErrorString(e));
        (void)hipFree(r->d_emb); (void)hipFree(r->d_chunk);
        delete r;
        return MIR_ERR_HIP;
    }
    *out = r;
    return MIR_OK;
}

int32_t mir_rows_info(const mir_rows *rows, int64_t *n, int32_t *d, int32_t *dtype, int32_t *device, int64_t *hbm_bytes) {
    MIR_REQUIRE(rows != nullptr, "handle is NULL");
    if (n) *n = rows->n;
    if (d) *d = rows->d;
    if (dtype) *dtype = rows->dtype;
    if (device) *device = rows->device;
    if (hbm_bytes) *hbm_bytes = rows->hbm_bytes;
    return MIR_OK;
}

int32_t mir_rows_destroy(mir_rows *rows) {
    if (!rows) return MIR_OK;
    (void)hipSetDevice(rows->device);
    (void)hipFree(rows->d_emb);
    (void)hipFree(rows->d_chunk);
    delete rows;
    return MIR_OK;
}

int32_t mir_index_create_from_rows(const mir_rows *const *parts, const int32_t *part_doc_ids, int32_t nparts,
                                   int32_t device, int64_t row_offset, mir_index **out) {
    MIR_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    MIR_REQUIRE(nparts >= 1 && parts != nullptr, "no row blocks");
    int64_t n = 0;
    for (int p = 0; p < nparts; ++p) {
        MIR_REQUIRE(parts[p] != nullptr, "row block %d is NULL", p);
        MIR_REQUIRE(parts[p]->device == device, "row block %d lives on device %d, the index on %d", p, parts[p]->device, device);
        MIR_REQUIRE(parts[p]->d == parts[0]->d && parts[p]->dtype == parts[0]->dtype,
                    "row block %d is %d-dimensional dtype %d, block 0 is %d-dimensional dtype %d", p, parts[p]->d,
                    parts[p]->dtype, parts[0]->d, parts[0]->dtype);
        n += parts[p]->n;
    }
    RowSource src;
    src.on_device = true; src.parts = parts; src.part_doc_ids = part_doc_ids; src.nparts = nparts;
    return create_common(src, n, parts[0]->d, parts[0]->dtype, device, row_offset, nullptr, out);
}

int32_t mir_index_destroy(mir_index *idx) {
    free_index(idx);
    return MIR_OK;
}

int32_t mir_index_info(const mir_index *idx, int64_t *n, int32_t *d, int32_t *dtype, int32_t *device,
                       int64_t *hbm_bytes) {
    MIR_REQUIRE(idx != nullptr, "index is NULL");
    if (n) *n = idx->n;
    if (d) *d = idx->d;
    if (dtype) *dtype = idx->dtype;
    if (device) *device = idx->device;
    if (hbm_bytes) *hbm_bytes = idx->hbm_bytes;
    return MIR_OK;
}

int32_t mir_index_profile(mir_index *idx, int32_t enable) {
    MIR_REQUIRE(idx != nullptr, "index is NULL");
    if (enable) {
        int32_t rc = use_device(idx->device, nullptr);
        if (rc != MIR_OK) return rc;
        std::vector<hipEvent_t> fresh;
        for (int i = 0; i < 512; ++i) {  // enough for 256 launches between reads, created outside any timed region
            hipEvent_t ev = nullptr;
            MIR_HIP(hipEventCreate(&ev));
            fresh.push_back(ev);
        }
        std::lock_guard<std::mutex> lk(idx->mu);
        if (idx->prof_free.size() < 512) idx->prof_free.insert(idx->prof_free.end(), fresh.begin(), fresh.end());
        else for (hipEvent_t ev : fresh) (void)hipEventDestroy(ev);
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->profiling = enable != 0;
    return MIR_OK;
}

int32_t mir_index_profile_read(mir_index *idx, int32_t reset, int64_t *launches, double *total_ms) {
    MIR_REQUIRE(idx != nullptr, "index is NULL");
    int32_t rc = use_device(idx->device, nullptr);
    if (rc != MIR_OK) return rc;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    {
        std::lock_guard<std::mutex> lk(idx->mu);
        ev.swap(idx->prof_events);
    }
    double ms = 0.0;
    for (auto &pr : ev) {
        float t = 0.f;
        hipError_t e = hipEventSynchronize(pr.second);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, pr.first, pr.second);
        {
            std::lock_guard<std::mutex> lk(idx->mu);
            idx->prof_free.push_back(pr.first);
            idx->prof_free.push_back(pr.second);
        }
        if (e != hipSuccess) {
            set_error("profile read failed: %s", hipGetErrorString(e));
            return MIR_ERR_HIP;
        }
        ms += t;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->prof_launches += (int64_t)ev.size();
    idx->prof_ms += ms;
    if (launches) *launches = idx->prof_launches;
    if (total_ms) *total_ms = idx->prof_ms;
    if (reset) {
        idx->prof_launches = 0;
        idx->prof_ms = 0.0;
    }
    return MIR_OK;
}

// Counters of the sieve (large float32 shards, vec_kernels_sieve.h) since the last reset; synchronises the device.
// out[0], out[1]: (row, query) candidates its first / second filter launch wrote; out[2]: queries it answered;
// out[3]: queries it handed to the exact pass (a full buffer); out[4], out[5]: verified candidates that entered the
// ranking after the first launch / at the end.  out[6], out[7]: reserved (0).
int32_t mir_index_scan_stats(mir_index *idx, int32_t reset, int64_t *out8) {
    MIR_REQUIRE(idx != nullptr && out8 != nullptr, "NULL argument");
    int32_t rc = use_device(idx->device, nullptr);
    if (rc != MIR_OK) return rc;
    MIR_HIP(hipDeviceSynchronize());
    MIR_HIP(hipMemcpy(out8, idx->d_stats, 64, hipMemcpyDeviceToHost));
    if (reset) MIR_HIP(hipMemset(idx->d_stats, 0, 64));
    out8[6] = idx->i8 ? 1 : 0;  // the shard has the int8 image: its searches for up to kI8MaxK results use the int8 first stage
    return MIR_OK;
}

int32_t mir_index_search_device(mir_index *idx, const double *queries_device, int32_t b, int32_t k,
                                int32_t metric, int32_t *out_doc, int64_t *out_chunk, int64_t *out_row,
                                double *out_dist, int32_t *out_count, int32_t *out_flags, void *stream_) {
    int32_t rc = check_search_args(idx, queries_device, b, k, metric, out_count);
    if (rc != MIR_OK) return rc;
    if (b == 0) return MIR_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    rc = use_device(idx->device, nullptr);
    if (rc != MIR_OK) return rc;
    SearchPlan pl;
    rc = plan(idx, b, k, &pl);
    if (rc != MIR_OK) return rc;
    SearchBuffers sb;
    const size_t i8_rows = (idx->i8 && pl.sieve) ? (size_t)idx->n_stages * 64 : 0;
    const size_t need = carve(sb, nullptr, b, k, idx->d, std::max(idx->ksteps, idx->ks16), pl, false, i8_rows);
    Workspace *w = nullptr;
    rc = acquire_ws(idx, stream, need, &w);
    if (rc != MIR_OK) return rc;
    carve(sb, static_cast<char *>(w->buf), b, k, idx->d, std::max(idx->ksteps, idx->ks16), pl, false, i8_rows);
    rc = enqueue_search(idx, queries_device, b, k, metric, sb, pl, out_doc, out_chunk, out_row,
                        out_dist, out_count, out_flags, stream);
    release_ws(idx, w, stream, true);
    return rc;
}

int32_t mir_index_search(mir_index *idx, const double *queries_host, int32_t b, int32_t k, int32_t metric,
                         int32_t *out_doc, int64_t *out_chunk, int64_t *out_row, double *out_dist,
                         int32_t *out_count, int32_t *out_flags) {
    int32_t rc = check_search_args(idx, queries_host, b, k, metric, out_count);
    if (rc != MIR_OK) return rc;
    if (b == 0) return MIR_OK;
    rc = use_device(idx->device, nullptr);
    if (rc != MIR_OK) return rc;
    SearchPlan pl;
    rc = plan(idx, b, k, &pl);
    if (rc != MIR_OK) return rc;
    SearchBuffers sb;
    const size_t i8_rows = (idx->i8 && pl.sieve) ? (size_t)idx->n_stages * 64 : 0;
    const size_t need = carve(sb, nullptr, b, k, idx->d, std::max(idx->ksteps, idx->ks16), pl, true, i8_rows);
    Workspace *w = nullptr;
    rc = acquire_ws(idx, nullptr, need, &w);
    if (rc != MIR_OK) return rc;
    carve(sb, static_cast<char *>(w->buf), b, k, idx->d, std::max(idx->ksteps, idx->ks16), pl, true, i8_rows);
    hipStream_t s = w->stream;
    auto bail = [&](int32_t code) {
        (void)hipStreamSynchronize(s);
        release_ws(idx, w, s, false);
        return code;
    };
#define MIR_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_error("%s failed: %s", #call, hipGetErrorString(e_));                              \
            return bail(MIR_ERR_HIP);                                                              \
        }                                                                                          \
    } while (0)
    // One pinned staging buffer per workspace: the queries go in through it and ALL result arrays come
    // back in ONE copy of the contiguous [o_doc .. o_flags] span.  With pageable user buffers every
    // hipMemcpyAsync is a synchronous staged copy of its own (~10-15 us each, six of them on the way
    // out): on a 1k-row index they were most of a 195 us call.
    const size_t q_bytes = (size_t)b * idx->d * sizeof(double);
    const size_t q_pad = (q_bytes + 255) & ~(size_t)255;
    const char *span0 = reinterpret_cast<const char *>(sb.o_doc);
    const size_t span = (size_t)(reinterpret_cast<const char *>(sb.o_flags) + (size_t)b * 4 - span0);
    if (w->pin_cap < q_pad + span) {
        if (w->pin) (void)hipHostFree(w->pin);
        w->pin = nullptr;
        w->pin_cap = 0;
        MIR_TRY(hipHostMalloc(reinterpret_cast<void **>(&w->pin), q_pad + span, hipHostMallocDefault));
        w->pin_cap = q_pad + span;
    }
    std::memcpy(w->pin, queries_host, q_bytes);
    MIR_TRY(hipMemcpyAsync(sb.q, w->pin, q_bytes, hipMemcpyHostToDevice, s));
    rc = enqueue_search(idx, sb.q, b, k, metric, sb, pl, out_doc ? sb.o_doc : nullptr,
                        out_chunk ? sb.o_chunk : nullptr, out_row ? sb.o_row : nullptr,
                        out_dist ? sb.o_dist : nullptr, sb.o_count, sb.o_flags, s);
    if (rc != MIR_OK) return bail(rc);
    const size_t bk = (size_t)b * k;
    char *res = w->pin + q_pad;
    MIR_TRY(hipMemcpyAsync(res, span0, span, hipMemcpyDeviceToHost, s));
    MIR_TRY(hipStreamSynchronize(s));
    auto at = [&](const void *dev_ptr) { return res + (reinterpret_cast<const char *>(dev_ptr) - span0); };
    if (out_doc) std::memcpy(out_doc, at(sb.o_doc), bk * 4);
    if (out_chunk) std::memcpy(out_chunk, at(sb.o_chunk), bk * 8);
    if (out_row) std::memcpy(out_row, at(sb.o_row), bk * 8);
    if (out_dist) std::memcpy(out_dist, at(sb.o_dist), bk * 8);
    std::memcpy(out_count, at(sb.o_count), (size_t)b * 4);
    if (out_flags) std::memcpy(out_flags, at(sb.o_flags), (size_t)b * 4);
#undef MIR_TRY
    release_ws(idx, w, s, false);
    return MIR_OK;
}

int32_t mir_index_metric_eval(mir_index *idx, const double *query_host, int32_t metric, double *out_host) {
    MIR_REQUIRE(idx != nullptr, "index is NULL");
    MIR_REQUIRE(metric >= 0 && metric <= 3, "unknown metric %d", metric);
    if (idx->n == 0) return MIR_OK;
    MIR_REQUIRE(query_host != nullptr && out_host != nullptr, "NULL buffer");
    int32_t rc = use_device(idx->device, nullptr);
    if (rc != MIR_OK) return rc;
    const int d = idx->d;
    const int64_t n = idx->n;
    // slab: q[d] | q_sq | q_norm | out[n]
    Carver c{nullptr};
    c.take<double>(d); c.take<double>(1); c.take<double>(1); c.take<double>(n);
    const size_t need = c.off + 256;
    Workspace *w = nullptr;
    rc = acquire_ws(idx, nullptr, need, &w);
    if (rc != MIR_OK) return rc;
    Carver cc{static_cast<char *>(w->buf)};
    double *dq = cc.take<double>(d);
    double *dsq = cc.take<double>(1);
    double *dnm = cc.take<double>(1);
    double *dout = cc.take<double>(n);
    hipStream_t s = w->stream;
    hipError_t e = hipMemcpyAsync(dq, query_host, (size_t)d * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        // ngroups = 0: only the per-query norm block runs
        prep_queries_kernel<<<dim3(1), dim3(64), 0, s>>>(dq, 1, d, idx->ksteps, 0, nullptr, dsq, dnm, nullptr, 0);
        if (idx->native16)
            metric_eval_kernel<_Float16><<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s>>>(idx->d_f16, idx->d_docsq, n, d,
                                                                                            dq, dsq, dnm, metric, dout);
        else
            metric_eval_kernel<float><<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s>>>(idx->d_orig, idx->d_docsq, n, d, dq,
                                                                                         dsq, dnm, metric, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, dout, (size_t)n * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    else (void)hipStreamSynchronize(s);
    release_ws(idx, w, s, false);
    if (e != hipSuccess) {
        set_error("metric_eval failed: %s", hipGetErrorString(e));
        return MIR_ERR_HIP;
    }
    return MIR_OK;
}

int32_t mir_metric_eval(const void *docs_host, int64_t n, int32_t d, int32_t dtype, const double *query_host,
                        int32_t metric, int32_t device, double *out_host) {
    mir_index *ix = nullptr;
    int32_t rc = mir_index_create(docs_host, n, d, dtype, nullptr, nullptr, device, 0, &ix);
    if (rc != MIR_OK) return rc;
    rc = mir_index_metric_eval(ix, query_host, metric, out_host);
    mir_index_destroy(ix);
    return rc;
}

int32_t mir_topk_merge_device(const double *dist, const int64_t *row, const int32_t *count, int32_t s,
                              int64_t shard_stride_bytes, int32_t b, int32_t k, int32_t descending_scores,
                              double *out_dist, int64_t *out_row, int32_t *out_count, int32_t device,
                              void *stream) {
    MIR_REQUIRE(s >= 1 && b >= 0 && k >= 1, "bad merge shape s=%d b=%d k=%d", s, b, k);
    MIR_REQUIRE(shard_stride_bytes >= 0 && shard_stride_bytes % 8 == 0, "shard stride must be a multiple of 8");
    if (b == 0) return MIR_OK;
    MIR_REQUIRE(dist && row && count && out_dist && out_row && out_count, "NULL buffer");
    int32_t rc = use_device(device, nullptr);
    if (rc != MIR_OK) return rc;
    const int64_t sd = shard_stride_bytes ? shard_stride_bytes : (int64_t)b * k * 8;
    const int64_t sr = shard_stride_bytes ? shard_stride_bytes : (int64_t)b * k * 8;
    const int64_t sc = shard_stride_bytes ? shard_stride_bytes : (int64_t)b * 4;
    const int64_t nk = (int64_t)s * k;
    if (nk <= kMergeLds)
        merge_topk_kernel<<<dim3(b), dim3(nk <= 64 ? 64 : nk <= 128 ? 128 : 256), 0, static_cast<hipStream_t>(stream)>>>(
            reinterpret_cast<const char *>(dist), reinterpret_cast<const char *>(row),
            reinterpret_cast<const char *>(count), s, sd, sr, sc, b, k, descending_scores, out_dist, out_row, out_count);
    else
        merge_topk_global_kernel<<<dim3(b), dim3(64), 0, static_cast<hipStream_t>(stream)>>>(
            reinterpret_cast<const char *>(dist), reinterpret_cast<const char *>(row),
            reinterpret_cast<const char *>(count), s, sd, sr, sc, b, k, descending_scores, out_dist, out_row, out_count);
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

int32_t mir_topk_merge_host(const double *dist, const int64_t *row, const int32_t *count, int32_t s,
                            int64_t shard_stride_bytes, int32_t b, int32_t k, int32_t descending_scores,
                            double *out_dist, int64_t *out_row, int32_t *out_count) {
    MIR_REQUIRE(s >= 1 && b >= 0 && k >= 1, "bad merge shape s=%d b=%d k=%d", s, b, k);
    MIR_REQUIRE(shard_stride_bytes >= 0 && shard_stride_bytes % 8 == 0, "shard stride must be a multiple of 8");
    if (b == 0) return MIR_OK;
    MIR_REQUIRE(dist && row && count && out_dist && out_row && out_count, "NULL buffer");
    const int64_t sd = shard_stride_bytes ? shard_stride_bytes : (int64_t)b * k * 8;
    const int64_t sr = shard_stride_bytes ? shard_stride_bytes : (int64_t)b * k * 8;
    const int64_t sc = shard_stride_bytes ? shard_stride_bytes : (int64_t)b * 4;
    struct Item {
        double d;
        int64_t r;
    };
    std::vector<Item> items;
    for (int q = 0; q < b; ++q) {
        items.clear();
        for (int sh = 0; sh < s; ++sh) {
            const int c = reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(count) + sh * sc)[q];
            MIR_REQUIRE(c >= 0 && c <= k, "count[%d][%d]=%d out of range", sh, q, c);
            const double *dp = reinterpret_cast<const double *>(reinterpret_cast<const char *>(dist) + sh * sd);
            const int64_t *rp = reinterpret_cast<const int64_t *>(reinterpret_cast<const char *>(row) + sh * sr);
            for (int p = 0; p < c; ++p) items.push_back({dp[(size_t)q * k + p], rp[(size_t)q * k + p]});
        }
        if (!descending_scores) {
            std::sort(items.begin(), items.end(), [](const Item &a, const Item &c) {
                const bool na = a.d != a.d, nc = c.d != c.d;
                if (na || nc) return na == nc ? a.r < c.r : nc;
                return a.d < c.d || (a.d == c.d && a.r < c.r);
            });
        } else {
            std::sort(items.begin(), items.end(),
                      [](const Item &a, const Item &c) { return a.d > c.d || (a.d == c.d && a.r > c.r); });
        }
        const int kout = (int)std::min<size_t>(items.size(), (size_t)k);
        for (int p = 0; p < kout; ++p) {
            out_dist[(size_t)q * k + p] = items[p].d;
            out_row[(size_t)q * k + p] = items[p].r;
        }
        out_count[q] = kout;
    }
    return MIR_OK;
}

}  // extern "C"
