// The sieve: the float32 index's search on large shards (round 3).  Streams HALF the bytes of the 128-query scan of
// vec_kernels_q16.h and is exact by construction instead of by an a-posteriori check.
//
// Why.  The q16 scan streams a tile's bf16 hi blocks and its lo blocks, but uses the lo blocks for ~1 % of the
// tile halves (only where a row's hi*hi value comes within the hi*hi error bound of a threshold): half of the 15.4 GB
// it reads per 10M x 384 pass is never looked at.  Here the scan reads ONLY the hi blocks (N*d*2 bytes) and does no
// correction, no list and no compaction at all:
//
//   filter   sieve_q16_kernel: v = hi*hi value of (row, query) in the scan's ranking units.  With
//            |true value - v| <= mg = 8e-3*|x||q| (kHiHiRelErr, worst case), a row whose true value reaches a
//            threshold T has v >= T - mg.  Every (row, query) with !(v < T - mg) is written out as a CANDIDATE
//            (row, query, v: 12 bytes; ~4e-5 of the pairs with the thresholds below) - a superset of all rows at or above T.
//   scatter  sieve_scatter_kernel: the candidates, written per workgroup, are appended to their queries' lists.
//   select   sieve_select_kernel, per query: kv = the k-th largest v of the list; k rows have true values >= kv - mg, so
//            every row of the true first k has v >= kv - 2 mg: those few (2-4 k of the few hundred listed) get the
//            reference's own float64 formula from the float32 rows (exact_metric_wave) and the reference's order
//            (distance, NaN last, row); the first k are the result.  (Until late in round 3 EVERY candidate was
//            evaluated in float64, by a kernel of its own after each launch: ~25 % of a 1.25M-row shard's step.)
//
// Exactness.  T is a lower bound of the query's k-th best TRUE value (k rows at or above it are known), so all of
// the true top k are at or above T, hence among the candidates, hence - by the same argument with kv - among the rows
// evaluated and ordered exactly.  No candidate list is bounded by k: the only way to lose a row is a full buffer, which is
// counted; the query then takes the exact pass (exact_pass_kernel).  There is no "uncertain" outcome and no
// completeness check.
//
// Thresholds, progressively (as the q16 scan): a 32K-row sample gives T0 (the k-th largest of 512 per-lane lower
// bounds v - mg over distinct rows); launch 1 sieves the first 1/16 of the tiles with T0; the k-th largest v of its
// candidates gives T1 = kv - mg; launch 2 sieves the rest with T1.  The candidates of both launches are selected from
// together at the end.
//
// euclidean_dist quirk (embeddings_metrics.py:50): sqrt of a negative rounding residue is NaN and sorts LAST, so
// a row (nearly) identical to the query has the LARGEST ranking value and the WORST rank.  Rows that could be NaN
// (v + mg reaches |q|^2, `nan_guard`) are left out of kv and of T0's sample - k rows at or above T that are certainly not
// NaN keep the argument above intact - and are always evaluated.
#pragma once
#include "vec_kernels_q16.h"
#include "vec_kernels_h16.h"

#ifndef SIEVE_ABL
#define SIEVE_ABL 0  // measurement builds only (tools/run_vec_variants.sh; results are wrong, only the time means something): 1 = half the
                     // MFMAs, 3 = no DMA after the prologue, 4 = no filter, 5 = no barrier.  Combinable with 4 (nothing is emitted, so
                     // the chain around the filter stays short): -DSIEVE_ABL_NODMA (as 3), -DSIEVE_ABL_NOWAIT (DMA issued, never
                     // waited for), -DSIEVE_ABL_HOT (every workgroup streams the same 64 tiles: the DMA's instructions and L2 path
                     // without HBM traffic) - round 4, DESIGN.md 3.4
#endif
#ifndef SIEVE_MAX3
#define SIEVE_MAX3 1  // 0 = the eight-compare form of the filter's common path on every index (measurement builds)
#endif
#ifndef SIEVE_SELECT_ABL
#define SIEVE_SELECT_ABL 0  // timing only: the select kernel's final mode returns 1 = after the compaction, 2 = after the first evaluation
#endif
#ifndef SIEVE_PREFETCH
#define SIEVE_PREFETCH 2  // k-steps a fragment pair is requested from LDS ahead of its MFMAs
#endif

namespace mir {

constexpr int kSieveStages = 6;        // LDS-DMA ring: 6 x 24 KiB at d = 384, five stages in flight
#ifndef MIR_SIEVE_REGION
#define MIR_SIEVE_REGION 8192
#endif
constexpr int kSieveRegion = MIR_SIEVE_REGION;  // candidates a workgroup can write per launch (96 KiB of HBM each)
constexpr int kSieveQueryCap = 16384;  // candidates listed per query (~120 k of them arrive on a 10M-row shard; k = 64 lists ~6 600)
constexpr int kSieveSelectCap = 4096;  // of which at most this many may need the float64 formula (within 2 mg of the k-th largest v):
                                       // a few dozen as a rule; thousands of near-copies of one row cost a query's block ~1 ms, not the exact pass
constexpr int kSieveMaxK = 64;
constexpr int kSieveCountStride = 32;  // a query's append counter has a 128-byte line of its own (43K appends on 4 shared lines took 120 us)

constexpr int kSieveAuxStride = kTileRows + 8;  // floats per stage: the tile's 32 norms + its largest row norm (+ padding; the IP form keeps a copy per wave)
__host__ __device__ constexpr size_t sieve_lds_bytes(int ks32) { return (size_t)kSieveStages * (ks32 * 2 * 1024 + kSieveAuxStride * 4) + 64; }

// QT = query tiles (16 queries each) per wave: 1 -> 128 queries per launch; 2 -> 256: every 1-KiB document fragment read from
// LDS feeds BOTH tiles' MFMAs (the stream and the LDS traffic of a pass are the same, the matrix work doubles: at 128 queries
// the matrix pipe is half idle behind the HBM stream).
// PTMT: margins per tile (the host's choice per index: its norms are spread, mir_index::norms_spread) instead of one per query.
template <int KS32, int KIND, bool SAMPLE, int QT, bool PTMT>
__global__ __launch_bounds__(512, 2) void sieve_q16_kernel(const uint4 *__restrict__ docs, const float *__restrict__ aux,
                                                           const uint4 *__restrict__ qsplit, const double *__restrict__ q_norm,
                                                           const double *__restrict__ q_sq, const double *__restrict__ q_err,
                                                           const float *__restrict__ max_norm,
                                                           uint32_t n_rows, uint32_t tile0, uint32_t n_tiles, int nq, int nan_guard,
                                                           const uint64_t *__restrict__ gthr, uint64_t *__restrict__ cand,
                                                           float *__restrict__ candv, uint32_t *__restrict__ ccount,
                                                           float *__restrict__ part_sample, unsigned long long *__restrict__ stat,
                                                           uint32_t tile_u4,    // uint4s from one tile's hi blocks to the next tile's
                                                           const float *__restrict__ tile_max) {  // [tiles] largest row norm (IP, L2: per-tile margin)
    static_assert(QT == 1 || QT == 2, "query tiles per wave");
    constexpr int NS = kSieveStages;
    constexpr int SB = KS32 * 2;          // 1-KiB blocks per stage = a tile's hi blocks
    constexpr int STAGE_U4 = SB * 64;
    // (tile_u4: STAGE_U4 on a hi-only image, 2 * STAGE_U4 where a tile's lo blocks follow its hi blocks - never read here)
    constexpr int PPW = SB / 8;           // 1-KiB DMA pieces per wave per stage
    constexpr bool AUX = KIND != SCAN_IP; // the tile's 32 norms travel with it: 16 bytes per wave, one more DMA
    constexpr bool PTM = PTMT && KIND != SCAN_COS;  // per-tile margin: the tile's largest row norm travels too (cosine's margin is relative already)
    constexpr int PW = PPW + ((AUX || PTM) ? 1 : 0);  // vector-memory operations per wave per stage
    constexpr int AS = kSieveAuxStride;
    constexpr int D = NS - 1;             // stages in flight: stage g's slot is free again once stage g has been read
    constexpr int QPL = 128 * QT;         // queries per launch
    static_assert(SB % 8 == 0, "sieve: d padded to a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *ring = reinterpret_cast<uint4 *>(smem);                                  // [NS][STAGE_U4]
    float *aux_lds = reinterpret_cast<float *>(smem + (size_t)NS * STAGE_U4 * 16);   // [NS][32]
    uint32_t *s_count = reinterpret_cast<uint32_t *>(aux_lds + NS * AS);

    const int tid = threadIdx.x, lane = tid & 63, qc = lane & 15, jg = lane >> 4;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t G = gridDim.x;
    if (tid == 0) *s_count = 0;

    // per query tile u of the wave: this lane's query, the hi*hi value's margin in ranking units, and the lane's pass bound
    // T - mg (with the float32 rounding of v itself)
    int qloc[QT];
    bool lane_live[QT];
    unsigned long long live_mask[QT];
    float mg[QT], bound[QT], guard[QT], best[QT];
    float cq[QT], tb[QT];  // PTM: the margin's per-query coefficient (x a tile's largest norm = mg) and the threshold with its rounding slack
    bf16x8 qh[QT][KS32];
    const bool active = nq > wave8 * QT * 16;  // (the wave's first tile has queries)
    // (word 1 of the index's norm statistics: a row with a NaN or an infinity was seen at build time)
    const bool finite_rows = __builtin_amdgcn_readfirstlane(__float_as_uint(max_norm[1])) == 0u;
    constexpr bool ptm = PTM;
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int t16 = wave8 * QT + u;
        qloc[u] = t16 * 16 + qc;
        lane_live[u] = qloc[u] < nq;
        live_mask[u] = __builtin_amdgcn_ballot_w64(lane_live[u]);
        mg[u] = 0.f; bound[u] = -__builtin_inff(); guard[u] = __builtin_inff(); best[u] = -__builtin_inff();
        cq[u] = 0.f; tb[u] = -__builtin_inff();
        if (lane_live[u]) {
            const float qn = (float)q_norm[qloc[u]] * (1.0f + 1e-6f);
            const float eq = (float)q_err[qloc[u]] * (1.0f + 1e-6f);
            mg[u] = hihi_margin(KIND == SCAN_COS, KIND == SCAN_L2, qn, eq, max_norm);
            if (PTM) cq[u] = hihi_coeff(KIND == SCAN_L2, qn, eq, max_norm);
            if (!SAMPLE) {
                const uint64_t key = gthr[qloc[u]];
                if (key != 0) {
                    const float t = key_value(key);
                    bound[u] = t - mg[u] - 2e-6f * fabsf(t);
                    tb[u] = t - 2e-6f * fabsf(t);
                }
            } else if (nan_guard) {
                // sq = q_sq - v < 0 is NaN under euclidean_dist: rows whose v + mg could reach q_sq do not count as known rows
                const float qs = (float)q_sq[qloc[u]];
                guard[u] = qs - 1e-5f * fabsf(qs);
            }
        }
        const uint4 *qs = qsplit + (size_t)t16 * KS32 * 128 + lane;
#pragma unroll
        for (int s = 0; s < KS32; ++s) qh[u][s] = __builtin_bit_cast(bf16x8, qs[(s * 2 + 0) * 64]);
    }
    const uint32_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + G - 1) / G : 0;
    const uint32_t NG = my_tiles;

    auto issue = [&](uint32_t g) {
#ifdef SIEVE_ABL_HOT
        const uint32_t tile = (g & 63u);
#else
        const uint32_t tile = tile0 + blockIdx.x + g * G;
#endif
        const uint4 *src = docs + (size_t)tile * tile_u4 + (wave8 * PPW) * 64 + lane;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr_of(ring) + ((g % NS) * STAGE_U4 + (wave8 * PPW) * 64) * 16);
#pragma unroll
        for (int i = 0; i < PPW; ++i) glds16_b128(src + i * 64, dst + i * 1024);
        if (AUX) {  // lanes 0..3 of wave w bring norms 4w .. 4w + 3 of the tile; lane 4 of the last wave the tile's largest row norm (slot 32)
            const uint32_t adst = __builtin_amdgcn_readfirstlane(lds_addr_of(aux_lds) + ((g % NS) * AS + wave8 * 4) * 4);
            const bool tm_lane = PTM && wave8 == 7 && lane == 4;
            const float *src1 = tm_lane ? tile_max + tile : aux + (size_t)tile * kTileRows + wave8 * 4 + lane;
            if (lane < 4 || tm_lane) glds4_b32(src1, adst);
        } else if (PTM) {  // no norm column (inner product): every wave brings its own copy of the tile's largest row norm (slot 4w)
            const uint32_t adst = __builtin_amdgcn_readfirstlane(lds_addr_of(aux_lds) + ((g % NS) * AS + wave8 * 4) * 4);
            if (lane == 0) glds4_b32(tile_max + tile, adst);
        }
    };
    // ordinary loads are complete before the first DMA (the counted waits below count DMAs only)
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int s = 0; s < KS32; ++s) asm volatile("" : "+v"(qh[u][s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t g = 0; g < (uint32_t)D && g < NG; ++g) issue(g);

    auto wait_stage = [&](uint32_t g) {  // stage g has landed: all but the younger stages' operations are done
#if SIEVE_ABL == 3 || defined(SIEVE_ABL_NODMA) || defined(SIEVE_ABL_NOWAIT)
        return;
#endif
        const uint32_t younger = (NG - 1 - g) < (uint32_t)(D - 1) ? (NG - 1 - g) : (uint32_t)(D - 1);
        if (younger == (uint32_t)(D - 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 1) * PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // only in the last D - 1 stages of the launch
    };
    uint64_t *region = cand + (size_t)blockIdx.x * kSieveRegion;
    float *regionv = candv + (size_t)blockIdx.x * kSieveRegion;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // The filter of one tile and query tile: this lane's 8 values against its bound.  It runs one tile LATE, in the shadow of
    // the next tile's MFMAs (its ~30 vector instructions after a tile's last MFMA were exposed at every barrier).
    // SCAN_L2: a tile's accumulators START at -|x|^2 / 2 (row-wise, the MFMA's C operand), so that c = x.q - |x|^2 / 2 and the
    // ranking value 2 x.q - |x|^2 is 2 c: the filter compares c with half the bound and needs no arithmetic at all.
    auto filter = [&](int u, const f32x4 &c0, const f32x4 &c1, const float (&ax)[8], uint32_t t, float tmax) {
        float v[8];
        // PTM: this tile's bound T - (largest row norm of the tile) x (the query's coefficient); cosine: the launch's constant
        const float mgt = ptm ? tmax * cq[u] : mg[u];
        const float bound_t = ptm ? tb[u] - mgt : bound[u];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = KIND == SCAN_COS ? c0[i] * ax[i] : c0[i];
            v[4 + i] = KIND == SCAN_COS ? c1[i] * ax[4 + i] : c1[i];
        }
        if (SAMPLE) {
            // a LOWER bound of this lane's best true value over rows that are certainly not NaN (sample tiles are whole tiles)
            if (lane_live[u]) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float w = KIND == SCAN_L2 ? 2.0f * v[r] : v[r];
                    if (w + mgt < guard[u]) best[u] = fmaxf(best[u], w - mgt);
                }
            }
            return;
        }
        // one compare per value, their results OR-ed as lane masks (scalar unit): beside the MFMAs every vector instruction
        // of the common path costs matrix-pipe issue slots
        const float bnd = KIND == SCAN_L2 ? 0.5f * bound_t : bound_t;
        if (SIEVE_MAX3 && finite_rows) {
            // no row of the index holds a NaN or an infinity (its largest norm is finite): a value is NaN only if the QUERY is,
            // and then all eight are - the maximum of the eight (v_max3: a NaN operand is ignored) decides: four instructions and
            // one compare into a scalar mask.  (As C++ the maxima came with a canonicalising v_max x, x per input.)
            float m1, m2, m3, m;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m2) : "v"(v[3]), "v"(v[4]), "v"(v[5]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(v[6]), "v"(v[7]), "v"(m1));
            asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m2), "v"(m3));
            unsigned long long mk;
            asm("v_cmp_nlt_f32 %0, %1, %2" : "=s"(mk) : "v"(m), "v"(bnd));
            if ((mk & live_mask[u]) == 0ull) return;
        } else {
            bool pass = false;
#pragma unroll
            for (int r = 0; r < 8; ++r) pass |= !(v[r] < bnd);  // NaN passes; no bound yet: everything passes
            pass &= lane_live[u];
            if (__builtin_amdgcn_ballot_w64(pass) == 0ull) return;
        }
        asm volatile("" : "+s"(t));  // (everything below depends on t: hipcc must not compute the rare path's row masks ahead of the branch)
        if (KIND == SCAN_L2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] *= 2.0f;
        }
        uint32_t pm = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) pm |= (uint32_t)(!(v[r] < bound_t)) << r;
        if (!lane_live[u]) pm = 0;
        if (KIND == SCAN_L2 && ptm && __any(pm != 0)) {
            // squared L2: the ROW's own margin - its norm is in the tile's norm column - now that something in the tile passed
            // the tile's bound (the tile's largest norm): a long row widens the band of its 31 neighbours in the common path
            // only, where that costs nothing.  (The norms go through an asm that depends on the opaque tile index: left
            // alone, hipcc computes the eight square roots in front of the common path's branch - measured: +24 % per step.)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float a2 = ax[r];
                asm volatile("" : "+v"(a2) : "s"(t));
                const float br = tb[u] - __builtin_amdgcn_sqrtf(a2) * cq[u] * (1.0f + 3e-5f);  // (v_sqrt_f32: ~1 ulp, covered by the factor)
                if (v[r] < br) pm &= ~(1u << r);
            }
        }
        if (!__any(pm != 0)) return;
        // rare (a few per cent of the wave-tiles): write the passing (query, row) pairs and their values to this workgroup's region
        const uint32_t row0 = t * kTileRows + 4 * jg;  // this lane's rows: row0 + 16 rh + i
        if (t * kTileRows + kTileRows > n_rows) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (row0 + 16 * (r >> 2) + (r & 3) >= n_rows) pm &= ~(1u << r);
        }
        while (__any(pm != 0)) {
            const bool has = pm != 0;
            const int r = has ? __builtin_ctz(pm) : 0;
            const unsigned long long bal = __ballot(has);
            const int leader = __builtin_ctzll(bal);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(s_count, (uint32_t)__popcll(bal));
            base = __shfl(base, leader, 64);
            const uint32_t slot = base + (uint32_t)__popcll(bal & lt_mask);
            float vr = v[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) vr = r == i ? v[i] : vr;
            if (has && slot < (uint32_t)kSieveRegion) {
                region[slot] = ((uint64_t)(uint32_t)qloc[u] << 32) | (uint64_t)(row0 + 16 * (r >> 2) + (r & 3));
                regionv[slot] = vr;
            }
            pm &= pm - 1;
        }
    };

    f32x4 p0[QT], p1[QT];  // the previous tile: accumulators, norms, index
#pragma unroll
    for (int u = 0; u < QT; ++u) { p0[u] = f32x4{0.f, 0.f, 0.f, 0.f}; p1[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float pax[8] = {};
    float p_tmax = 0.f, c_tmax = 0.f;  // the previous / this tile's largest row norm
    uint32_t pt = 0;
    bool have_prev = false;
    for (uint32_t g = 0; g < my_tiles; ++g) {
        const uint32_t t = tile0 + blockIdx.x + g * G;
        wait_stage(g);
#if SIEVE_ABL != 5
        __builtin_amdgcn_s_barrier();   // every wave's pieces of stage g are in; everybody has left stage g - 1's slot
#endif
        if (!active) {
            if (g + D < NG) issue(g + D);
            continue;
        }
        const uint4 *st = ring + (size_t)(g % NS) * STAGE_U4 + lane;
        constexpr int PF = SIEVE_PREFETCH;  // fragment pairs requested ahead of the k-step that multiplies them
        uint4 f0[PF + 1], f1[PF + 1];
#pragma unroll
        for (int i = 0; i < PF; ++i) { f0[i] = st[(2 * i + 0) * 64]; f1[i] = st[(2 * i + 1) * 64]; }
        float cax[8] = {};
        if (AUX) {  // rows 16 rh + 4 jg + i of this tile
            const float4 a0 = *reinterpret_cast<const float4 *>(aux_lds + (g % NS) * AS + 4 * jg);
            const float4 a1 = *reinterpret_cast<const float4 *>(aux_lds + (g % NS) * AS + 16 + 4 * jg);
            cax[0] = a0.x; cax[1] = a0.y; cax[2] = a0.z; cax[3] = a0.w;
            cax[4] = a1.x; cax[5] = a1.y; cax[6] = a1.z; cax[7] = a1.w;
        }
        if (ptm) c_tmax = aux_lds[(g % NS) * AS + (AUX ? kTileRows : 4 * wave8)];
        f32x4 c0[QT], c1[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            if (KIND == SCAN_L2) {
                c0[u] = f32x4{-0.5f * cax[0], -0.5f * cax[1], -0.5f * cax[2], -0.5f * cax[3]};
                c1[u] = f32x4{-0.5f * cax[4], -0.5f * cax[5], -0.5f * cax[6], -0.5f * cax[7]};
            } else {
                c0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                c1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int s = 0; s < KS32; ++s) {
            if (s + PF < KS32) {
                f0[(s + PF) % (PF + 1)] = st[(2 * (s + PF) + 0) * 64];
                f1[(s + PF) % (PF + 1)] = st[(2 * (s + PF) + 1) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#if SIEVE_ABL == 1  // half of the MFMAs, all LDS reads
            if (!(s & 1)) {
#endif
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                c0[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f0[s % (PF + 1)]), qh[u][s], c0[u], 0, 0, 0);
                c1[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f1[s % (PF + 1)]), qh[u][s], c1[u], 0, 0, 0);
            }
#if SIEVE_ABL == 1
            }
#endif
            if (s == 1) {  // the next stage's DMA once the matrix pipe has work queued
                __builtin_amdgcn_sched_barrier(0);
#if SIEVE_ABL != 3 && !defined(SIEVE_ABL_NODMA)
                if (g + D < NG) issue(g + D);
#endif
            }
#pragma unroll
            for (int u = 0; u < QT; ++u)
                if (s == 2 + u && have_prev) {  // the previous tile's filter, one query tile per k-step
                    __builtin_amdgcn_sched_barrier(0);
#if SIEVE_ABL != 4
                    filter(u, p0[u], p1[u], pax, pt, p_tmax);
#else
                    if (g == 0xffffff) filter(u, p0[u], p1[u], pax, pt, p_tmax);
#endif
                }
        }
#pragma unroll
        for (int u = 0; u < QT; ++u) { p0[u] = c0[u]; p1[u] = c1[u]; }
        pt = t; have_prev = true; p_tmax = c_tmax;
#pragma unroll
        for (int i = 0; i < 8; ++i) pax[i] = cax[i];
    }
    if (have_prev) {
#pragma unroll
        for (int u = 0; u < QT; ++u) filter(u, p0[u], p1[u], pax, pt, p_tmax);
    }
    if (SAMPLE) {
        // four lanes hold a query's column: two values per query, each the maximum over distinct rows
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const float o = __shfl_xor(best[u], 16, 64);
            const float b2 = fmaxf(best[u], o);
            if (lane_live[u] && (jg == 0 || jg == 2)) part_sample[((size_t)blockIdx.x * QPL + qloc[u]) * 2 + (jg >> 1)] = b2;
        }
        return;
    }
    __syncthreads();
    if (tid == 0) {
        ccount[blockIdx.x] = *s_count;  // may exceed kSieveRegion: the verify kernel then hands every query to the exact pass
        if (stat && *s_count) atomicAdd(stat, (unsigned long long)*s_count);  // mir_index_scan_stats: candidates of this launch
    }
}

// ---------------------------------------------------------------- the filter on a float16-native index (config C5)
// The same sieve over the float16 image of vec_kernels_h16.h (rows exact in float16, ONE product per fragment on
// v_mfma_f32_16x16x32_f16, the query scaled by a power of two and rounded to float16): |true value - v| <= 6e-4 |x||q|
// (kH16RelErr), a seventh of the bf16 margin.  A tile is 64 KiB at d = 1024: the ring holds four stages of 16 k-steps (32 KiB,
// half a tile), three in flight; the accumulators run over a tile's stages, the filter over the finished tile runs in the
// shadow of the next tile's first MFMAs.  Every stage brings the tile's 32 norms along (16 bytes per wave): one more
// counted DMA per stage, the same for every stage.
constexpr int kSieve16Stages = 4;
__host__ __device__ constexpr size_t sieve16_lds_bytes() { return (size_t)kSieve16Stages * (kH16StageKs * 2048 + kTileRows * 4) + 64; }

// BF (round 4): the same kernel over the bf16 HI image of a wide FLOAT32 shard (384 < d <= 1024: the multimodal / description
// retrievers' page embeddings, embeddings_index.py:139-153) - bf16 products, the query's bf16 hi fragments as
// prep_queries16_kernel writes them (hi and lo blocks alternate: `qfrag` is read with a stride of two blocks), no query scale,
// and sieve_q16_kernel's margin (hihi_margin with the query's own residual `q_err`, plus d 2^-24 for the longer float32
// accumulation).  Until round 4 those shards streamed hi AND lo blocks through scan_topk_f16_kernel<SPLIT>: twice the bytes.
template <int KS32, int KIND, bool SAMPLE, bool BF = false>
__global__ __launch_bounds__(512, 2) void sieve_h16_kernel(const uint4 *__restrict__ docs, const float *__restrict__ aux,
                                                           const uint4 *__restrict__ qfrag, const float *__restrict__ qscale_inv,
                                                           const double *__restrict__ q_norm, const double *__restrict__ q_sq,
                                                           const double *__restrict__ q_err,
                                                           const float *__restrict__ max_norm, uint32_t n_rows, uint32_t tile0,
                                                           uint32_t n_tiles, int nq, int nan_guard, const uint64_t *__restrict__ gthr,
                                                           uint64_t *__restrict__ cand, float *__restrict__ candv,
                                                           uint32_t *__restrict__ ccount, float *__restrict__ part_sample,
                                                           unsigned long long *__restrict__ stat) {
    static_assert(KS32 % kH16StageKs == 0, "float16 sieve: d padded to a multiple of 512");
    constexpr int NS = kSieve16Stages;
    constexpr int SPT = KS32 / kH16StageKs;   // stages per tile
    constexpr int SB = kH16StageKs * 2;       // 1-KiB blocks per stage
    constexpr int STAGE_U4 = SB * 64;
    constexpr int TILE_U4 = SPT * STAGE_U4;
    constexpr int PPW = SB / 8;               // 1-KiB DMA pieces per wave per stage
    constexpr bool AUX = KIND != SCAN_IP;
    constexpr int PW = PPW + (AUX ? 1 : 0);   // vector-memory operations per wave per stage
    constexpr int D = NS - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *ring = reinterpret_cast<uint4 *>(smem);                                   // [NS][STAGE_U4]
    float *aux_lds = reinterpret_cast<float *>(smem + (size_t)NS * STAGE_U4 * 16);    // [NS][32]
    uint32_t *s_count = reinterpret_cast<uint32_t *>(aux_lds + NS * kTileRows);

    const int tid = threadIdx.x, lane = tid & 63, qc = lane & 15, jg = lane >> 4;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qloc = wave8 * 16 + qc;
    const bool lane_live = qloc < nq;
    const bool active = nq > wave8 * 16;
    const uint32_t G = gridDim.x;
    if (tid == 0) *s_count = 0;

    float mg = 0.f, bound = -__builtin_inff(), guard = __builtin_inff(), best = -__builtin_inff(), inv_s = 0.f;
    if (lane_live) {
        inv_s = BF ? 1.0f : qscale_inv[qloc];
        const float qn = (float)q_norm[qloc] * (1.0f + 1e-6f);
        if (BF) mg = hihi_margin(KIND == SCAN_COS, KIND == SCAN_L2, qn, (float)q_err[qloc] * (1.0f + 1e-6f), max_norm) +
                     wide_accum_slop(KS32 * 32) * qn * (KIND == SCAN_COS ? 1.0f : max_norm[0]) * (KIND == SCAN_L2 ? 2.0f : 1.0f);
        else mg = (float)kH16RelErr * qn * (KIND == SCAN_COS ? 1.0f : max_norm[0]) * (KIND == SCAN_L2 ? 2.0f : 1.0f);
        if (!SAMPLE) {
            const uint64_t key = gthr[qloc];
            if (key != 0) {
                const float t = key_value(key);
                bound = t - mg - 2e-6f * fabsf(t);
            }
        } else if (nan_guard) {
            const float qs = (float)q_sq[qloc];
            guard = qs - 1e-5f * fabsf(qs);
        }
    }
    uint4 qh[KS32];  // f16x8 or bf16x8 fragments, by BF
    {
        constexpr int QB = BF ? 2 : 1;  // blocks per k-step in `qfrag`
        const uint4 *qs = qfrag + (size_t)wave8 * KS32 * QB * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS32; ++s) qh[s] = qs[s * QB * 64];
    }
    const uint32_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + G - 1) / G : 0;
    const uint32_t NG = my_tiles * SPT;

    auto issue = [&](uint32_t g) {
        const uint32_t tile = tile0 + blockIdx.x + (g / SPT) * G;
        const uint4 *src = docs + (size_t)tile * TILE_U4 + (size_t)(g % SPT) * STAGE_U4 + (wave8 * PPW) * 64 + lane;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr_of(ring) + ((g % NS) * STAGE_U4 + (wave8 * PPW) * 64) * 16);
#pragma unroll
        for (int i = 0; i < PPW; ++i) glds16_b128(src + i * 64, dst + i * 1024);
        if (AUX) {
            const uint32_t adst = __builtin_amdgcn_readfirstlane(lds_addr_of(aux_lds) + ((g % NS) * kTileRows + wave8 * 4) * 4);
            if (lane < 4) glds4_b32(aux + (size_t)tile * kTileRows + wave8 * 4 + lane, adst);
        }
    };
#pragma unroll
    for (int s = 0; s < KS32; ++s) asm volatile("" : "+v"(qh[s].x), "+v"(qh[s].y), "+v"(qh[s].z), "+v"(qh[s].w));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t g = 0; g < (uint32_t)D && g < NG; ++g) issue(g);
    auto wait_stage = [&](uint32_t g) {
        const uint32_t younger = (NG - 1 - g) < (uint32_t)(D - 1) ? (NG - 1 - g) : (uint32_t)(D - 1);
        if (younger == (uint32_t)(D - 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 1) * PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    uint64_t *region = cand + (size_t)blockIdx.x * kSieveRegion;
    float *regionv = candv + (size_t)blockIdx.x * kSieveRegion;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    auto filter = [&](const f32x4 &c0, const f32x4 &c1, const float (&ax)[8], uint32_t t) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d0 = c0[i] * inv_s, d1 = c1[i] * inv_s;
            v[i] = KIND == SCAN_IP ? d0 : KIND == SCAN_L2 ? fmaf(2.0f, d0, -ax[i]) : d0 * ax[i];
            v[4 + i] = KIND == SCAN_IP ? d1 : KIND == SCAN_L2 ? fmaf(2.0f, d1, -ax[4 + i]) : d1 * ax[4 + i];
        }
        if (SAMPLE) {
            if (lane_live) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (v[r] + mg < guard) best = fmaxf(best, v[r] - mg);
            }
            return;
        }
        uint32_t pm = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) pm |= (uint32_t)(!(v[r] < bound)) << r;
        if (!lane_live) pm = 0;
        if (!__any(pm != 0)) return;
        const uint32_t row0 = t * kTileRows + 4 * jg;
        if (t * kTileRows + kTileRows > n_rows) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (row0 + 16 * (r >> 2) + (r & 3) >= n_rows) pm &= ~(1u << r);
        }
        while (__any(pm != 0)) {
            const bool has = pm != 0;
            const int r = has ? __builtin_ctz(pm) : 0;
            const unsigned long long bal = __ballot(has);
            const int leader = __builtin_ctzll(bal);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(s_count, (uint32_t)__popcll(bal));
            base = __shfl(base, leader, 64);
            const uint32_t slot = base + (uint32_t)__popcll(bal & lt_mask);
            float vr = v[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) vr = r == i ? v[i] : vr;
            if (has && slot < (uint32_t)kSieveRegion) {
                region[slot] = ((uint64_t)(uint32_t)qloc << 32) | (uint64_t)(row0 + 16 * (r >> 2) + (r & 3));
                regionv[slot] = vr;
            }
            pm &= pm - 1;
        }
    };

    f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
    float pax[8] = {};
    uint32_t pt = 0;
    bool have_prev = false;
    for (uint32_t ts = 0; ts < my_tiles; ++ts) {
        const uint32_t t = tile0 + blockIdx.x + ts * G;
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        float cax[8] = {};
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const uint32_t g = ts * SPT + j;
            wait_stage(g);
            __builtin_amdgcn_s_barrier();
            if (!active) {
                if (g + D < NG) issue(g + D);
                continue;
            }
            const uint4 *st = ring + (size_t)(g % NS) * STAGE_U4 + lane;
            uint4 f0[3], f1[3];
            f0[0] = st[0 * 64]; f1[0] = st[1 * 64];
            f0[1] = st[2 * 64]; f1[1] = st[3 * 64];
            if (AUX && j == SPT - 1) {  // rows 16 rh + 4 jg + i of this tile (every stage's slot holds them)
                const float4 a0 = *reinterpret_cast<const float4 *>(aux_lds + (g % NS) * kTileRows + 4 * jg);
                const float4 a1 = *reinterpret_cast<const float4 *>(aux_lds + (g % NS) * kTileRows + 16 + 4 * jg);
                cax[0] = a0.x; cax[1] = a0.y; cax[2] = a0.z; cax[3] = a0.w;
                cax[4] = a1.x; cax[5] = a1.y; cax[6] = a1.z; cax[7] = a1.w;
            }
#pragma unroll
            for (int s = 0; s < kH16StageKs; ++s) {
                if (s + 2 < kH16StageKs) {
                    f0[(s + 2) % 3] = st[(2 * (s + 2) + 0) * 64];
                    f1[(s + 2) % 3] = st[(2 * (s + 2) + 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (BF) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f0[s % 3]), __builtin_bit_cast(bf16x8, qh[j * kH16StageKs + s]), c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f1[s % 3]), __builtin_bit_cast(bf16x8, qh[j * kH16StageKs + s]), c1, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f0[s % 3]), __builtin_bit_cast(f16x8, qh[j * kH16StageKs + s]), c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f1[s % 3]), __builtin_bit_cast(f16x8, qh[j * kH16StageKs + s]), c1, 0, 0, 0);
                }
                if (s == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + D < NG) issue(g + D);
                }
                if (j == 0 && s == 2 && have_prev) {
                    __builtin_amdgcn_sched_barrier(0);
                    filter(p0, p1, pax, pt);
                }
            }
        }
        if (active) {
            p0 = c0; p1 = c1; pt = t; have_prev = true;
#pragma unroll
            for (int i = 0; i < 8; ++i) pax[i] = cax[i];
        }
    }
    if (have_prev) filter(p0, p1, pax, pt);
    if (SAMPLE) {
        const float o = __shfl_xor(best, 16, 64);
        const float b2 = fmaxf(best, o);
        if (lane_live && (jg == 0 || jg == 2)) part_sample[((size_t)blockIdx.x * kQ16Queries + qloc) * 2 + (jg >> 1)] = b2;
        return;
    }
    __syncthreads();
    if (tid == 0) {
        ccount[blockIdx.x] = *s_count;
        if (stat && *s_count) atomicAdd(stat, (unsigned long long)*s_count);
    }
}

// ---------------------------------------------------------------- scatter
// The candidates of a filter launch lie in per-workgroup regions; a thread per candidate appends (row, filter value) to its
// query's list.  No row is read here: what is worth the reference formula is decided per query, from the values alone.
struct SieveLists {
    float *rv;         // [b][kSieveQueryCap] the filter's value v of (row, query), in the scan's ranking units: |true value - v| <= mg
    uint32_t *row;     // [b][kSieveQueryCap]
    uint32_t *count;   // [b][kSieveCountStride] (word 0) appended so far (may exceed the capacity: the rest was dropped)
    uint32_t *over;    // [b] != 0: a candidate of this query was dropped somewhere -> exact pass
};

struct SieveScatterArgs {
    const uint64_t *cand;   // [regions][kSieveRegion] (query of the launch << 32 | row)
    const float *candv;     // [regions][kSieveRegion]
    const uint32_t *ccount; // [regions]
    int q0, nq;             // the launch's queries are q0 .. q0 + nq - 1
    SieveLists l;
};

constexpr int kSieveScatterSplit = 2;  // workgroups per region
__global__ __launch_bounds__(256) void sieve_scatter_kernel(SieveScatterArgs a) {
    const int tid = threadIdx.x;
    const int reg = blockIdx.x / kSieveScatterSplit, part = blockIdx.x % kSieveScatterSplit;
    uint32_t cnt = a.ccount[reg];
    if (cnt > (uint32_t)kSieveRegion) {  // the region overflowed: whose candidates were lost is unknown
        if (part == 0)
            for (int i = tid; i < a.nq; i += 256) a.l.over[a.q0 + i] = 1;
        cnt = kSieveRegion;
    }
    const uint64_t *region = a.cand + (size_t)reg * kSieveRegion;
    const float *regionv = a.candv + (size_t)reg * kSieveRegion;
    for (uint32_t e = part * 256 + tid; e < cnt; e += 256 * kSieveScatterSplit) {
        const uint64_t key = region[e];
        const float v = regionv[e];
        const int qi = a.q0 + (int)(key >> 32);
        const uint32_t slot = atomicAdd(&a.l.count[(size_t)qi * kSieveCountStride], 1u);
        if (slot < (uint32_t)kSieveQueryCap) {
            a.l.row[(size_t)qi * kSieveQueryCap + slot] = (uint32_t)key;
            a.l.rv[(size_t)qi * kSieveQueryCap + slot] = v;
        } else {
            a.l.over[qi] = 1;
        }
    }
}

// ---- wave-level selection helpers of the select kernel ----
// maximum over the wave of a 32-bit unsigned key, uniform: four DPP steps within each row of 16 lanes, then the four rows
__device__ __forceinline__ uint32_t sieve_wave_max_u32(uint32_t x) {
#define MIR_DPP_MAX(CTRL)                                                                              \
    {                                                                                                  \
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);    \
        x = x > o ? x : o;                                                                             \
    }
    MIR_DPP_MAX(0xB1)   // quad_perm [1,0,3,2]
    MIR_DPP_MAX(0x4E)   // quad_perm [2,3,0,1]
    MIR_DPP_MAX(0x141)  // row_half_mirror
    MIR_DPP_MAX(0x140)  // row_mirror
#undef MIR_DPP_MAX
    const uint32_t a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
    const uint32_t c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
// the wave's largest key over every lane's r[0..R), removed from the one register that holds it (one instance); 0 = none left
template <int R>
__device__ __forceinline__ uint32_t sieve_extract_max(uint32_t (&r)[R], int lane) {
    uint32_t m = r[0];
#pragma unroll
    for (int j = 1; j < R; ++j) m = r[j] > m ? r[j] : m;
    const uint32_t wm = sieve_wave_max_u32(m);
    const unsigned long long bal = __ballot(m == wm);
    if (wm != 0 && lane == __builtin_ctzll(bal)) {
        bool done = false;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const bool hit = !done && r[j] == wm;
            r[j] = hit ? 0u : r[j];
            done = done || hit;
        }
    }
    return wm;
}
// this wave's share of the n values (entries tid, tid + nt, ...; -inf = nothing) -> its k largest as orderable keys, out[0..k)
template <int R>
__device__ __forceinline__ void sieve_wave_topk(const float *vals, int n, int k, int tid, int nt, uint32_t *out) {
    const int lane = tid & 63;
    uint32_t r[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int e = j * nt + tid;
        const float v = e < n ? vals[e] : -__builtin_inff();
        r[j] = v > -__builtin_inff() ? orderable(v) : 0u;
    }
    if (__builtin_amdgcn_readfirstlane(tid - lane) >= n) {  // (a wave with no entry at all)
        if (lane < k) out[lane] = 0u;
        return;
    }
    for (int t = 0; t < k; ++t) {
        const uint32_t wm = sieve_extract_max<R>(r, lane);
        if (lane == 0) out[t] = wm;
    }
}

// exact_metric_wave's arithmetic for an aligned group of 16 lanes sharing a row, with 16-byte loads (d % 4 == 0): lane l takes
// elements 4l .. 4l + 3, 4l + 64 .., all of a row's loads in flight at once.  (With one dword per lane and load the texture
// addresser, not the memory, set the pace: 18 us for 47 rows.)
template <typename T>
__device__ __forceinline__ void sieve_load4(const T *p, float (&v)[4]);
template <>
__device__ __forceinline__ void sieve_load4<float>(const float *p, float (&v)[4]) {
    const float4 x = *reinterpret_cast<const float4 *>(p);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
}
template <>
__device__ __forceinline__ void sieve_load4<_Float16>(const _Float16 *p, float (&v)[4]) {
    typedef _Float16 __attribute__((ext_vector_type(4))) h4;
    const h4 x = *reinterpret_cast<const h4 *>(p);
    v[0] = (float)x[0]; v[1] = (float)x[1]; v[2] = (float)x[2]; v[3] = (float)x[3];
}
template <typename T>
__device__ __forceinline__ double sieve_metric_g16(const T *__restrict__ row, const double *__restrict__ q, int d, int metric,
                                                   float doc_sq32, double q_sq, double q_norm, int lg, double *rank_value) {
    constexpr int U = 6;  // 6 x 64 elements per pass: d = 384 in one
    double dn_inv_dummy = 0.0;
    (void)dn_inv_dummy;
    float dn = 1.0f;
    double qn = 1.0;
    const bool cosine = metric == MIR_METRIC_COSINE_SIM;
    if (cosine) {
        double s = 0.0;
        for (int j0 = 4 * lg; j0 < d; j0 += 64 * U) {
            float v[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.f;
                if (j0 + 64 * u < d) sieve_load4<T>(row + j0 + 64 * u, v[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double x = (double)v[u][e];
                    s += x * x;
                }
        }
        s = group_sum<16>(s);
        dn = fmaxf((float)sqrt(s), 1e-8f);
        qn = fmax(q_norm, 1e-8);
    }
    double dot = 0.0;
    for (int j0 = 4 * lg; j0 < d; j0 += 64 * U) {
        float v[U][4];
        double qv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.f;
            qv[u][0] = qv[u][1] = qv[u][2] = qv[u][3] = 0.0;
            if (j0 + 64 * u < d) {
                sieve_load4<T>(row + j0 + 64 * u, v[u]);
                const double2 a = *reinterpret_cast<const double2 *>(q + j0 + 64 * u), b = *reinterpret_cast<const double2 *>(q + j0 + 64 * u + 2);
                qv[u][0] = a.x; qv[u][1] = a.y; qv[u][2] = b.x; qv[u][3] = b.y;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (cosine) dot += (double)__fdiv_rn(v[u][e], dn) * (qv[u][e] / qn);
                else dot += (double)v[u][e] * qv[u][e];
            }
    }
    dot = group_sum<16>(dot);
    if (cosine) {
        *rank_value = dot * qn;
        return -dot;
    }
    if (metric == MIR_METRIC_INNER_PRODUCT) {
        *rank_value = dot;
        return -dot;
    }
    const double sq = ((double)doc_sq32 - 2.0 * dot) + q_sq;
    *rank_value = 2.0 * dot - (double)doc_sq32;
    return metric == MIR_METRIC_SQEUCLIDEAN_DIST ? sq : sqrt(sq);
}

// ---------------------------------------------------------------- select
// (the int8 filter's margin coefficients - vec_kernels_i8.h, i8_margin_ab - are needed here too)
__device__ __forceinline__ void i8_margin_ab_decl(float qn, float eq, const float *__restrict__ stats, float &A, float &B) {
    A = (qn + eq) * (1.0f + 1e-5f) * MIR_MARGIN_SCALE;
    B = (stats[0] * eq + 3.0e-5f * stats[0] * qn) * (1.0f + 1e-5f) * MIR_MARGIN_SCALE;
}
struct SieveSelectArgs {
    SieveLists l;
    int q0, nq, k, metric, mode;   // mode 0: thresholds for the next launch; 1: the result
    int d, nan_guard;
    float rel_err;                 // the float16 filter's bound (kH16RelErr); a float32 index: the margin of hihi_margin()
    float extra_slop;              // wide float32 shards (d > 384): the longer float32 accumulation's share of the margin, relative (wide_accum_slop)
    const double *q_err;           // [b] |q - bf16(q)| (float32 index; |q - s_q Q| behind the int8 filter), or null
    const float *i8_qscale;        // [b] the int8 filter's query scales (vec_kernels_i8.h: max_norm then holds ITS statistics), or null
    const float4 *i8_tparam;       // [tiles] its tile parameters (s_t, e_t, ..): a row's margin is its tile's
    const float *docs;             // f32 [n][d], or null with
    const _Float16 *docs16;        // f16 [n][d] (float16-native index)
    const float *doc_sq;
    const float *dnorm;            // [n] row norms (row_dnorm_kernel): the per-row margin of inner product / squared L2 on a float32 index, or null
    const double *q, *q_sq, *q_norm;
    const float *max_norm;
    unsigned long long *gthr;      // [nq] of this launch group
    const int64_t *chunk_ids;
    const int32_t *doc_ids;
    int64_t row_offset;
    int32_t *out_doc;
    int64_t *out_chunk;
    int64_t *out_row;
    double *out_dist;
    int32_t *out_count;
    int32_t *out_flags;
    int32_t *nflag;
    int32_t *flagged;
    unsigned long long *stats;     // mir_index_scan_stats counters (see there)
    double *qt;                    // the exact pass's transposed copy of the queries handed to it
};

// grid = nq (one block per query), block = 1024.  Everything up to the last step works on the filter's values alone.
//   With mg the filter's bound (round 4: PER ROW for inner product / squared L2 on a float32 index - the row's norm times the
//   query's coefficient, hihi_coeff - otherwise one value per query), a listed row's true value lies in [v - mg, v + mg].
//   kv = the k-th largest v over listed rows
//   that are certainly not NaN; then k rows have true values >= kv - mg =: T, so T is a lower bound of the k-th best true value:
//   mode 0: T is the next launch's threshold (the filter lets v >= T - mg through);
//   mode 1: a row of the true first k has true value >= T, hence v >= kv - 2 mg: only those (~2-4 k of the few hundred listed)
//           are worth the reference's float64 formula (exact_metric_wave, one wave per row) and are ranked in the
//           reference's order; the first k are the result.
// Rows that may be NaN under euclidean_dist (v + mg reaches |q|^2: sqrt of a negative residue, sorts LAST) count for nothing
// in kv and are always evaluated: a row identical to the query may as well be the best one.
__host__ __device__ constexpr size_t sieve_select_lds_bytes() {
    // s_v f32 [QueryCap + 4] | s_d f64 [SelectCap] | s_x f32 [SelectCap] | s_row u32 [SelectCap] | s_fin u16 [SelectCap] | part u32 [16][64]
    return (size_t)kSieveQueryCap * 4 + 16 + (size_t)kSieveSelectCap * 18 + 16 * 64 * 4;
}
constexpr int kSieveSelectThreads = 1024;
__global__ __launch_bounds__(kSieveSelectThreads) void sieve_select_kernel(SieveSelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sel_lds[];
    constexpr size_t kOffD = (size_t)kSieveQueryCap * 4 + 16;
    float *s_v = reinterpret_cast<float *>(sel_lds);                                             // [kSieveQueryCap + 4] -inf: may be NaN / padding
    double *s_d = reinterpret_cast<double *>(sel_lds + kOffD);                                    // [kSieveSelectCap] reference distances
    float *s_x = reinterpret_cast<float *>(sel_lds + kOffD + (size_t)kSieveSelectCap * 8);        // [kSieveSelectCap]
    uint32_t *s_row = reinterpret_cast<uint32_t *>(sel_lds + kOffD + (size_t)kSieveSelectCap * 12);  // [kSieveSelectCap] rows of s_fin's entries
    uint16_t *s_fin = reinterpret_cast<uint16_t *>(sel_lds + kOffD + (size_t)kSieveSelectCap * 16);  // [kSieveSelectCap] list entries worth evaluating
    uint32_t *part = reinterpret_cast<uint32_t *>(sel_lds + kOffD + (size_t)kSieveSelectCap * 18);   // [16][64]
    __shared__ int s_ns, s_f, s_f1, s_slot, s_have, s_nn;
    __shared__ float s_kv;
    constexpr int NT = kSieveSelectThreads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = a.q0 + blockIdx.x;
    const uint32_t total = a.l.count[(size_t)qi * kSieveCountStride];
    const int n = (int)(total < (uint32_t)kSieveQueryCap ? total : (uint32_t)kSieveQueryCap);
    const uint32_t *lr = a.l.row + (size_t)qi * kSieveQueryCap;
    const float *lv = a.l.rv + (size_t)qi * kSieveQueryCap;
    const bool over = a.l.over[qi] != 0;
    if (tid == 0) { s_ns = 0; s_f = 0; s_f1 = 0; s_have = 0; s_nn = 0; }
    __syncthreads();
    auto to_exact_pass = [&]() {  // (whole block)
        if (a.mode != 1) return;
        if (tid == 0) {
            atomicAdd(a.stats + 3, 1ull);
            s_slot = atomicAdd(a.nflag, 1);
            a.flagged[s_slot] = qi;
            if (a.out_flags) a.out_flags[qi] = MIR_FLAG_EXACT_PASS;
            if (a.out_count) a.out_count[qi] = 0;
        }
        __syncthreads();
        exact_publish_query(a.qt, s_slot, a.q + (size_t)qi * a.d, a.d, a.metric, a.q_norm[qi], tid, NT);
    };
    if (over) {  // a candidate of this query was dropped somewhere (mode 0: the sample's threshold stays)
        to_exact_pass();
        return;
    }
    // the k-th largest of vals[0..m) (entries of -inf do not count; the caller knows at least k count), for the whole block:
    // every wave extracts the k largest of its share (k rounds of a wave maximum that removes one instance), wave 0 the k
    // largest of those.  (Counting, for every entry, the entries above it was 23 us for 340 entries and 0.9 ms for 3900 at k = 64.)
    auto kth_largest = [&](const float *vals, int m) -> float {
        __syncthreads();
        if (m <= NT) sieve_wave_topk<1>(vals, m, a.k, tid, NT, part + wave * 64);
        else if (m <= 2 * NT) sieve_wave_topk<2>(vals, m, a.k, tid, NT, part + wave * 64);
        else if (m <= 4 * NT) sieve_wave_topk<4>(vals, m, a.k, tid, NT, part + wave * 64);
        else sieve_wave_topk<kSieveQueryCap / kSieveSelectThreads>(vals, m, a.k, tid, NT, part + wave * 64);
        __syncthreads();
        if (wave == 0) {
            const int nw = m <= NT ? (m + 63) / 64 : NT / 64;   // waves that held entries
            uint32_t r[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = j * 64 + lane;  // (wave i / k, its i % k-th largest)
                r[j] = i < nw * a.k ? part[(i / a.k) * 64 + i % a.k] : 0u;
            }
            uint32_t wm = 0;
            for (int t = 0; t < a.k; ++t) wm = sieve_extract_max<16>(r, lane);
            if (lane == 0) s_kv = unorderable(wm);
        }
        __syncthreads();
        return s_kv;
    };
    const float qn = (float)a.q_norm[qi] * (1.0f + 1e-6f);
    const bool l2 = !(a.metric == MIR_METRIC_INNER_PRODUCT || a.metric == MIR_METRIC_COSINE_SIM);
    const float mg = a.q_err ? hihi_margin(a.metric == MIR_METRIC_COSINE_SIM, l2, qn, (float)a.q_err[qi] * (1.0f + 1e-6f), a.max_norm) +
                                   a.extra_slop * qn * (a.metric == MIR_METRIC_COSINE_SIM ? 1.0f : a.max_norm[0]) * (l2 ? 2.0f : 1.0f)
                             : a.rel_err * qn * (a.metric == MIR_METRIC_COSINE_SIM ? 1.0f : a.max_norm[0]) * (l2 ? 2.0f : 1.0f);  // (as the filter's)
    // Per-row margins (float32 index, inner product / squared L2): a row's true value lies within m = |x| * cq of its v (hihi_coeff);
    // the filter used the row's TILE maximum, which is no smaller.  Everything below is written in LOWER bounds lb = v - m and
    // UPPER bounds lb + 2 m; with one margin for all rows (cosine, float16-native) that is the round-3 arithmetic in other words.
    const bool i8 = a.i8_qscale != nullptr;  // behind the int8 filter: a row's margin is its tile's (i8_margin_tile, the filter's own formula)
    const bool per_row = i8 || (a.q_err != nullptr && a.dnorm != nullptr && a.metric != MIR_METRIC_COSINE_SIM);  // (dnorm: only where the filter ran per tile)
    const float cq = (per_row && !i8) ? hihi_coeff(l2, qn, (float)a.q_err[qi] * (1.0f + 1e-6f), a.max_norm) : 0.f;
    float i8A = 0.f, i8B = 0.f;
    const float i8sq = i8 ? a.i8_qscale[qi] : 0.f;
    if (i8) {
        const float eq = (float)a.q_err[qi] * (1.0f + 1e-6f);
        if (qn < __builtin_inff() && eq < __builtin_inff()) i8_margin_ab_decl(qn, eq, a.max_norm, i8A, i8B);
        else i8A = i8B = __builtin_inff();
    }
    auto margin_of = [&](uint32_t row) {
        if (i8) {
            const float4 tp = a.i8_tparam[row / kTileRows];
            const float m = fmaf(tp.y, i8A, i8B) + 2.0f * tp.x * i8sq;
            return l2 ? 2.0f * m : a.metric == MIR_METRIC_COSINE_SIM ? m * a.max_norm[6] : m;  // (cosine: x the largest inverse norm)
        }
        return per_row ? a.dnorm[row] * cq : mg;
    };
    const float eps_m = per_row ? 0.f : 1e-6f * mg;  // (per row: the margins carry their own 1e-6; an index-wide mg may be infinite there)
    float guard = __builtin_inff();
    if (a.nan_guard) {
        const float qs = (float)a.q_sq[qi];
        guard = qs - 1e-5f * fabsf(qs);
    }
    // ---- 0. the list -> LDS; s_v = -inf for rows that may be NaN (and for a NaN / infinite v)
    for (int e0 = 0; e0 < n; e0 += NT) {
        const int e = e0 + tid;
        bool safe = false;
        if (e < n) {
            const float v = lv[e];
            const float m = margin_of(lr[e]);
            safe = v + m < guard && v > -__builtin_inff();
            s_v[e] = safe ? v - m : -__builtin_inff();  // the row's LOWER bound (a NaN margin - a NaN row norm - is not safe)
        }
        const unsigned long long bal = __ballot(safe);
        if (bal && lane == __builtin_ctzll(bal)) atomicAdd(&s_ns, __popcll(bal));
    }
    __syncthreads();
    const int ns = s_ns;
    // ---- 1. kv = the k-th largest of the safe rows' lower bounds: k rows have true values >= kv, so kv bounds the k-th best
    //         true value from below
    const bool have = ns >= a.k;
    const float kv = have ? kth_largest(s_v, n) : 0.f;
    if (a.mode == 0) {
        if (have && tid == 0) {
            const float thr = kv - 4e-6f * fabsf(kv) - eps_m;
            const unsigned long long key = (unsigned long long)orderable(thr) << 32;
            if (thr == thr && key > a.gthr[blockIdx.x]) a.gthr[blockIdx.x] = key;
        }
        if (tid == 0) atomicAdd(a.stats + 4, (unsigned long long)n);  // entries listed after launch 1
        return;
    }
    // ---- 2. the rows that can be among the first k.  Class 1: lower bound >= kv (at least k of them), the ones that may be
    //         NaN, and everything when there is no kv - always evaluated.  Class 2: upper bound >= kv > lower bound - evaluated
    //         only if class 1's exact values leave them a chance (step 3).  s_fin = class 1, then class 2.
    const float cut = have ? kv - 1e-5f * fabsf(kv) - eps_m : -__builtin_inff();
    auto compact = [&](int cls, int *counter) {
        for (int e0 = 0; e0 < n; e0 += NT) {
            const int e = e0 + tid;
            bool fin = false;
            if (e < n) {
                const float lb = s_v[e];
                const bool c1 = !have || !(lb > -__builtin_inff()) || !(lb < kv);
                if (cls == 1) fin = c1;
                else if (!c1) fin = !(lb + 2.0f * margin_of(lr[e]) * (1.0f + 1e-6f) < cut);
            }
            const unsigned long long bal = __ballot(fin);
            int base = 0;
            if (bal && lane == __builtin_ctzll(bal)) base = atomicAdd(counter, __popcll(bal));
            base = __shfl(base, bal ? __builtin_ctzll(bal) : 0, 64);
            const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (fin && slot < kSieveSelectCap) { s_fin[slot] = (uint16_t)e; s_row[slot] = lr[e]; }
        }
        __syncthreads();
    };
    compact(1, &s_f);
    const int f1 = s_f;
    if (tid == 0) s_f1 = f1;
    compact(2, &s_f);
    int f = s_f;
    if (f > kSieveSelectCap) {  // a mass of rows within the filter's resolution of the cut: the exact pass orders them
        to_exact_pass();
        return;
    }
    // the reference's float64 formula for s_fin[lo..hi): four rows per wave at a time, 16 lanes each (a row is a chain of
    // dependent fetches - list -> row index -> row: with one row per wave a clustered corpus's ~1800 rows per query were 112
    // rounds of that latency).  s_d = the distance, s_x = its ranking value rounded to float
    const double *qv = a.q + (size_t)qi * a.d;
    const double q_sq = a.q_sq[qi], q_norm = a.q_norm[qi];
    constexpr int GW = 16, GPW = 64 / GW;
    const int sub = lane / GW, lg = lane % GW;
    const bool vec4 = (a.d & 3) == 0;  // rows and queries 16-byte aligned (float16 rows: 8)
    auto evaluate = [&](int lo, int hi) {
        for (int i0 = lo + wave * GPW; i0 < hi; i0 += (NT / 64) * GPW) {
            const int i = i0 + sub;
            const bool live = i < hi;
            const uint32_t row = s_row[live ? i : i0];  // (a group without a row repeats the wave's first: no divergence)
            double rv, dist;
            if (vec4) {
                dist = a.docs16 ? sieve_metric_g16<_Float16>(a.docs16 + (size_t)row * a.d, qv, a.d, a.metric, a.doc_sq[row], q_sq, q_norm, lg, &rv)
                                : sieve_metric_g16<float>(a.docs + (size_t)row * a.d, qv, a.d, a.metric, a.doc_sq[row], q_sq, q_norm, lg, &rv);
            } else {
                dist = a.docs16 ? exact_metric_wave<_Float16, GW, 8>(a.docs16 + (size_t)row * a.d, qv, a.d, a.metric, a.doc_sq[row], q_sq, q_norm, lg, &rv)
                                : exact_metric_wave<float, GW, 8>(a.docs + (size_t)row * a.d, qv, a.d, a.metric, a.doc_sq[row], q_sq, q_norm, lg, &rv);
            }
            if (live && lg == 0) {
                s_d[i] = dist;
                s_x[i] = dist == dist ? (float)rv : -__builtin_inff();  // (a NaN distance ranks last whatever its ranking value)
            }
        }
        __syncthreads();
    };
#if SIEVE_SELECT_ABL == 1
    return;
#endif
    // (a handful of class-2 rows - the rule on isotropic data - are not worth the second round and its bookkeeping)
    const bool two_rounds = f - f1 > 128;
    evaluate(0, two_rounds ? f1 : f);
#if SIEVE_SELECT_ABL == 2
    return;
#endif
    // ---- 3. class 2 against class 1's exact values: with rk = the k-th largest exact ranking value of class 1 (numeric
    //         distances only), a row whose v + mg stays below rk is beaten by k rows for certain and is dropped unevaluated
    if (two_rounds) {
        for (int i0 = 0; i0 < f1; i0 += NT) {
            const int i = i0 + tid;
            const bool num = i < f1 && s_x[i] > -__builtin_inff();
            const unsigned long long bal = __ballot(num);
            if (bal && lane == __builtin_ctzll(bal)) atomicAdd(&s_nn, __popcll(bal));
        }
        __syncthreads();
        if (s_nn >= a.k) {
            const float rk = kth_largest(s_x, f1);
            const float need = rk - 4e-6f * fabsf(rk) - eps_m;  // (float)rv rounds to nearest: the slack covers it
            if (tid == 0) s_f = f1;
            __syncthreads();
            // survivors of class 2 move up behind class 1 (in place: a survivor's slot is never beyond its old one)
            for (int i0 = f1; i0 < f; i0 += NT) {
                const int i = i0 + tid;
                uint16_t e = 0;
                uint32_t er = 0;
                bool keep = false;
                if (i < f) {
                    e = s_fin[i];
                    er = s_row[i];
                    keep = !(s_v[e] + 2.0f * margin_of(er) * (1.0f + 1e-6f) < need);  // its UPPER bound reaches the k-th exact value
                }
                __syncthreads();
                const unsigned long long bal = __ballot(keep);
                int base = 0;
                if (bal && lane == __builtin_ctzll(bal)) base = atomicAdd(&s_f, __popcll(bal));
                base = __shfl(base, bal ? __builtin_ctzll(bal) : 0, 64);
                const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (keep) { s_fin[slot] = e; s_row[slot] = er; }
                __syncthreads();
            }
            f = s_f;
        }
        evaluate(f1, f);
    }
    // ---- 4. the reference's order among the evaluated rows: a float pre-filter first (the k-th smallest distance rounded to
    //         float; rounding is monotone, so every row of the true first k stays), then the exact order among what is left
    //         (ranking all pairs of ~1800 evaluated rows of a clustered corpus in float64 was half a millisecond)
    bool pre = false;
    if (f > 256) {  // (little to rank: all pairs directly)
        for (int i0 = 0; i0 < f; i0 += NT) {  // s_x = -(float)dist, -inf for NaN
            const int i = i0 + tid;
            if (i < f) {
                const double dd = s_d[i];
                s_x[i] = dd == dd ? -(float)dd : -__builtin_inff();
            }
        }
        if (tid == 0) s_nn = 0;
        __syncthreads();
        for (int i0 = 0; i0 < f; i0 += NT) {
            const int i = i0 + tid;
            const bool num = i < f && s_x[i] > -__builtin_inff();
            const unsigned long long bal = __ballot(num);
            if (bal && lane == __builtin_ctzll(bal)) atomicAdd(&s_nn, __popcll(bal));
        }
        __syncthreads();
        pre = s_nn >= a.k;  // (fewer numeric distances than k: the NaN ones are needed too, everything is ranked)
    }
    const float kx = pre ? kth_largest(s_x, f) : -__builtin_inff();
    const int kout = a.k < f ? a.k : f;
    for (int i = tid; i < f; i += NT) {
        if (pre && s_x[i] < kx) continue;
        const double dd = s_d[i];
        const uint32_t rr = s_row[i];
        int rank = 0;
        for (int c = 0; c < f; ++c) {
            if (pre && s_x[c] < kx) continue;
            rank += dist_before(s_d[c], s_row[c], dd, rr) ? 1 : 0;  // (an entry is not before itself)
        }
        if (rank < a.k) {
            const size_t o = (size_t)qi * a.k + rank;
            if (a.out_row) a.out_row[o] = a.row_offset + (int64_t)rr;
            if (a.out_dist) a.out_dist[o] = dd;
            if (a.out_doc) a.out_doc[o] = a.doc_ids ? a.doc_ids[rr] : 0;
            if (a.out_chunk) a.out_chunk[o] = a.chunk_ids ? a.chunk_ids[rr] : (int64_t)rr;
        }
    }
    if (tid == 0) {
        atomicAdd(a.stats + 5, (unsigned long long)f);  // rows evaluated in float64
        atomicAdd(a.stats + 2, 1ull);
        if (a.out_count) a.out_count[qi] = kout;
        if (a.out_flags) a.out_flags[qi] = 0;
    }
}

}  // namespace mir
