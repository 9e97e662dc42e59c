// Host side of the bge-small-en encoder: weight packing, batching, launches.
// C ABI in include/miretr.h; device code in encoder_kernels.h.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "common.h"
#include "encoder_kernels.h"

using namespace mir;
using namespace mir::enc;

namespace {

// Token tiles per pass (the workspace holds five activation buffers of 24 KiB per tile, twice: 3 GB of the 288).  3072
// tiles (98 304 tokens) fill the chip exactly once in qkv_kernel; larger passes amortise the per-pass work and the
// launch tails: 8192 chunks of ~220 tokens ran at 62.9k chunks/s with 3072, 63.9k with 6144, 64.8k with 9216 and
// 65.5k with 12288 tiles per pass (52.4k with 1536).
#ifndef ENC_PASS_TILES
#define ENC_PASS_TILES 12288
#endif
constexpr int kMaxTilesPerPass = ENC_PASS_TILES;
// the throughput path's attention kernel: 1 = attention_kernel (one wave per query tile, four waves per SIMD; also the latency
// path's kernel), 2 = attention2_kernel (one wave per TWO query tiles: half the K / V fetches, two waves per SIMD) - within 4 % of
// each other (what bounds them differs); a third, persistent form with an LDS-DMA ring was slower and is
// in the history only (commit "Encoder attention: fix a latent MFMA operand hazard ...")
#ifndef ENC_ATT
#define ENC_ATT 1
#endif
// fused_qkv_attention_kernel for the sequences of at most kFqaTiles token tiles (throughput path): OFF by default - correct
// (bit-identical to the other kernels, tests/test_gpu_encoder.py runs it) but 5 % SLOWER on its share than qkv_kernel +
// attention_kernel (427 against 405 us per pass: DESIGN.md 4b, round 4); MIR_ENC_FUSED_QKV_ATTENTION=1 at mir_encoder_create, or
// -DENC_FQA=1, selects it
#ifndef ENC_FQA
#define ENC_FQA 0
#endif

// Pack a Hugging Face Linear weight W[out][in] (y = x W^T + b) into MFMA A-fragment
// order for out^T = W x^T: block (nt, ks) = 64 lanes x 8 halfs, lane (row = l&31,
// h = l>>5) element j = W[32*nt + row][32*(ks>>1) + fi(8*(ks&1) + j, h)].
void pack_block(const float *W, int in_dim, int nt, int ks, _Float16 *dst /*512 halfs*/) {
    for (int l = 0; l < 64; ++l) {
        const int row = 32 * nt + (l & 31), h = l >> 5;
        for (int j = 0; j < 8; ++j) {
            const int r = 8 * (ks & 1) + j;
            const int kf = 32 * (ks >> 1) + ((r & 3) + 8 * (r >> 2) + 4 * h);
            dst[l * 8 + j] = (_Float16)W[(size_t)row * in_dim + kf];
        }
    }
}

struct Layer {
    uint4 *wqkv = nullptr;        // [36][24][64]
    float *bqkv = nullptr;        // [1152]
    uint4 *wo = nullptr;          // [12][24][64]
    float *attn_params = nullptr; // bo | gamma | beta  [3*384]
    unsigned char *wffn = nullptr;// [48][48 KiB]
    float *ffn_params = nullptr;  // b1 | b2 | gamma | beta | GELU table (encoder_common.h)
};

}  // namespace

struct mir_encoder {
    int device = 0;
    bool fused_qkv_attention = ENC_FQA != 0;
    int layers = 0;
    int vocab = 0, max_pos = 0;
    float *word = nullptr, *pos = nullptr, *type0 = nullptr, *emb_g = nullptr, *emb_b = nullptr;
    std::vector<Layer> L;
    int64_t hbm_bytes = 0;
    std::mutex mu;  // one encode at a time per handle (the reference runs its encoder on 1-thread pools, cpu_pools.py:25-34)
    hipStream_t stream = nullptr;
    // two workspace slots: the host stages pass p+1 while the GPU runs pass p
    void *ws[2] = {nullptr, nullptr};
    size_t ws_cap[2] = {0, 0};
    hipEvent_t ws_done[2] = {nullptr, nullptr};
    // pinned host staging per slot (pageable memory would make every hipMemcpyAsync block the
    // host until the stream reaches it, serialising host staging with GPU work)
    char *pin[2] = {nullptr, nullptr};
    size_t pin_cap[2] = {0, 0};
};

namespace {

void free_encoder(mir_encoder *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipFree(e->word); (void)hipFree(e->pos); (void)hipFree(e->type0); (void)hipFree(e->emb_g); (void)hipFree(e->emb_b);
    for (Layer &l : e->L) {
        (void)hipFree(l.wqkv); (void)hipFree(l.bqkv); (void)hipFree(l.wo); (void)hipFree(l.attn_params);
        (void)hipFree(l.wffn); (void)hipFree(l.ffn_params);
    }
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(e->ws[i]);
        if (e->pin[i]) (void)hipHostFree(e->pin[i]);
        if (e->ws_done[i]) (void)hipEventDestroy(e->ws_done[i]);
    }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

hipError_t upload(mir_encoder *e, void **dst, const void *src, size_t bytes) {
    hipError_t er = hipMalloc(dst, bytes);
    if (er != hipSuccess) return er;
    e->hbm_bytes += (int64_t)bytes;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
}

struct Batch {
    std::vector<int32_t> ids;        // [n_tiles*32], 0-padded
    std::vector<TileInfo> tiles;     // [n_tiles]
    std::vector<int32_t> seq_first;  // [n_seq]
    std::vector<int32_t> units;      // attention units, 4 ints each: first tile | (tiles - 1) << 24, key tiles, sequence length, the sequence's first tile
    std::vector<int32_t> bins;       // fused QKV + attention: kFqaTiles slots of 4 ints per bin (encoder_attention.hip)
    std::vector<int32_t> tile_pos;   // [n_tiles] a tile's position in INPUT order (the debug hidden states are returned in it)
    int n_long_tiles = 0;            // tiles [0, n_long_tiles): the sequences that stay on qkv_kernel + attention_kernel
};

// Sequences [s0, s1) -> padded tiles.  `fused`: the pass will run the fused QKV + attention kernel (throughput path): the
// sequences of more than kFqaTiles tiles come FIRST (their tiles are qkv_kernel's and attention_kernel's), the others after
// them, packed - whole sequences, longest first, first fit - into bins of at most kFqaTiles tiles.  Every per-sequence result
// is addressed through seq_first, so the order of the tiles inside a pass is free.
void build_batch(const int32_t *token_ids, const int64_t *offs, const int32_t *lens, int s0, int s1, bool fused, Batch &b) {
    b.ids.clear(); b.tiles.clear(); b.seq_first.clear(); b.units.clear(); b.bins.clear(); b.tile_pos.clear();
    b.n_long_tiles = 0;
    const int ns = s1 - s0;
    b.seq_first.assign((size_t)ns, 0);
    std::vector<int32_t> in_pos((size_t)ns);  // a sequence's first tile in input order
    {
        int p = 0;
        for (int s = s0; s < s1; ++s) { in_pos[s - s0] = p; p += (lens[s] + 31) / 32; }
    }
    auto place = [&](int s) {  // the sequence's tiles, token ids and attention units at the end of the pass
        const int len = lens[s];
        const int nt = (len + 31) / 32;
        const int first = (int)b.tiles.size();
        b.seq_first[s - s0] = first;
        for (int t = 0; t < nt; ++t) {
            b.tiles.push_back(TileInfo{first, nt, len, s - s0});
            b.tile_pos.push_back(in_pos[s - s0] + t);
        }
        const size_t base = b.ids.size();
        b.ids.resize(base + (size_t)nt * 32, 0);
        std::memcpy(b.ids.data() + base, token_ids + offs[s], sizeof(int32_t) * len);
        return first;
    };
    if (!fused) {
        for (int s = s0; s < s1; ++s) {
            const int nt = (lens[s] + 31) / 32, first = place(s);
            for (int t = 0; t < nt; t += 2) b.units.insert(b.units.end(), {(first + t) | ((t + 1 < nt ? 1 : 0) << 24), nt, lens[s], first});
        }
        b.n_long_tiles = (int)b.tiles.size();
        return;
    }
    for (int s = s0; s < s1; ++s) {
        const int nt = (lens[s] + 31) / 32;
        if (nt <= kFqaTiles) continue;
        const int first = place(s);
        for (int t = 0; t < nt; t += 2) b.units.insert(b.units.end(), {(first + t) | ((t + 1 < nt ? 1 : 0) << 24), nt, lens[s], first});
    }
    b.n_long_tiles = (int)b.tiles.size();
    // the short sequences by tiles, descending (counting sort, stable), then first fit: open[r] = bins with r free slots
    std::vector<int32_t> by_len[kFqaTiles + 1];
    for (int s = s0; s < s1; ++s) {
        const int nt = (lens[s] + 31) / 32;
        if (nt <= kFqaTiles) by_len[nt].push_back(s);
    }
    std::vector<int32_t> open[kFqaTiles + 1];
    for (int nt = kFqaTiles; nt >= 1; --nt)
        for (int32_t s : by_len[nt]) {
            int bin = -1, room = 0;
            for (int r = nt; r <= kFqaTiles && bin < 0; ++r)
                if (!open[r].empty()) { bin = open[r].back(); open[r].pop_back(); room = r; }
            if (bin < 0) {
                bin = (int)(b.bins.size() / (4 * kFqaTiles));
                b.bins.resize(b.bins.size() + 4 * kFqaTiles, 0);
                for (int i = 0; i < kFqaTiles; ++i) b.bins[(size_t)(bin * kFqaTiles + i) * 4] = -1;
                room = kFqaTiles;
            }
            const int slot0 = kFqaTiles - room, first = place(s);
            for (int t = 0; t < nt; ++t) {
                int32_t *sl = &b.bins[(size_t)(bin * kFqaTiles + slot0 + t) * 4];
                sl[0] = first + t; sl[1] = slot0 | (nt << 8); sl[2] = lens[s]; sl[3] = 0;
            }
            if (room - nt > 0) open[room - nt].push_back(bin);
        }
}

}  // namespace

extern "C" {

int32_t mir_encoder_create(int32_t hidden, int32_t layers, int32_t heads, int32_t intermediate, int32_t vocab,
                           int32_t max_pos, const float *word_emb, const float *pos_emb, const float *type_emb,
                           const float *emb_ln_gamma, const float *emb_ln_beta, const float *const *layer_tensors,
                           int32_t device, mir_encoder **out) {
    MIR_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    if (hidden != H || heads != NH || intermediate != FF) {
        set_error("encoder build is specialised for bge-small-en (hidden 384, 12 heads, FFN 1536); got %d/%d/%d",
                  hidden, heads, intermediate);
        return MIR_ERR_UNSUPPORTED;
    }
    MIR_REQUIRE(layers >= 1 && layers <= 64 && vocab >= 1 && max_pos >= 1, "bad encoder shape");
    MIR_REQUIRE(word_emb && pos_emb && type_emb && emb_ln_gamma && emb_ln_beta && layer_tensors, "NULL weight pointer");
    for (int i = 0; i < layers * 16; ++i) MIR_REQUIRE(layer_tensors[i] != nullptr, "layer tensor %d is NULL", i);
    int32_t rc = use_device(device, nullptr);
    if (rc != MIR_OK) return rc;
    mir_encoder *e = new (std::nothrow) mir_encoder();
    MIR_REQUIRE(e != nullptr, "out of host memory");
    e->device = device; e->layers = layers; e->vocab = vocab; e->max_pos = max_pos;
    if (const char *fq = getenv("MIR_ENC_FUSED_QKV_ATTENTION")) e->fused_qkv_attention = atoi(fq) != 0;
    auto fail = [&](int32_t code) { free_encoder(e); return code; };
#define MIR_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_error("%s failed: %s", #call, hipGetErrorString(e_));                              \
            return fail(MIR_ERR_HIP);                                                              \
        }                                                                                          \
    } while (0)
    MIR_TRY(upload(e, (void **)&e->word, word_emb, (size_t)vocab * H * 4));
    // position table padded to 512 rows so the kernel's clamp is always in range
    {
        std::vector<float> p((size_t)512 * H, 0.f);
        std::memcpy(p.data(), pos_emb, sizeof(float) * (size_t)std::min(max_pos, 512) * H);
        MIR_TRY(upload(e, (void **)&e->pos, p.data(), p.size() * 4));
    }
    MIR_TRY(upload(e, (void **)&e->type0, type_emb, (size_t)H * 4));  // token_type 0 only
    MIR_TRY(upload(e, (void **)&e->emb_g, emb_ln_gamma, (size_t)H * 4));
    MIR_TRY(upload(e, (void **)&e->emb_b, emb_ln_beta, (size_t)H * 4));

    e->L.resize(layers);
    std::vector<_Float16> buf;
    for (int li = 0; li < layers; ++li) {
        const float *const *t = layer_tensors + (size_t)li * 16;
        // order: q.w q.b k.w k.b v.w v.b ao.w ao.b aln.g aln.b i.w i.b o.w o.b oln.g oln.b
        Layer &l = e->L[li];
        // QKV: 36 tiles x 24 k-steps
        buf.assign((size_t)36 * KS_H * 512, (_Float16)0);
        for (int tile = 0; tile < 36; ++tile) {
            const float *W = t[(tile / 12) * 2];
            for (int ks = 0; ks < KS_H; ++ks) pack_block(W, H, tile % 12, ks, buf.data() + ((size_t)tile * KS_H + ks) * 512);
        }
        MIR_TRY(upload(e, (void **)&l.wqkv, buf.data(), buf.size() * 2));
        {
            std::vector<float> b(3 * H);
            for (int p = 0; p < 3; ++p) std::memcpy(b.data() + p * H, t[p * 2 + 1], sizeof(float) * H);
            MIR_TRY(upload(e, (void **)&l.bqkv, b.data(), b.size() * 4));
        }
        // attention output projection: [nt][ks]
        buf.assign((size_t)NFB * KS_H * 512, (_Float16)0);
        for (int nt = 0; nt < NFB; ++nt)
            for (int ks = 0; ks < KS_H; ++ks) pack_block(t[6], H, nt, ks, buf.data() + ((size_t)nt * KS_H + ks) * 512);
        MIR_TRY(upload(e, (void **)&l.wo, buf.data(), buf.size() * 2));
        {
            std::vector<float> p(3 * H);
            std::memcpy(p.data(), t[7], sizeof(float) * H);
            std::memcpy(p.data() + H, t[8], sizeof(float) * H);
            std::memcpy(p.data() + 2 * H, t[9], sizeof(float) * H);
            MIR_TRY(upload(e, (void **)&l.attn_params, p.data(), p.size() * 4));
        }
        // FFN stages: [ht][48 pieces]: 0..23 W1 (out tile ht, k-step ks), 24..47 W2 (out tile nt, k-step 2*ht + s2)
        buf.assign((size_t)NHT * 48 * 512, (_Float16)0);
        for (int ht = 0; ht < NHT; ++ht) {
            _Float16 *st = buf.data() + (size_t)ht * 48 * 512;
            for (int ks = 0; ks < KS_H; ++ks) pack_block(t[10], H, ht, ks, st + (size_t)ks * 512);
            for (int nt = 0; nt < NFB; ++nt)
                for (int s2 = 0; s2 < 2; ++s2) pack_block(t[12], FF, nt, 2 * ht + s2, st + (size_t)(24 + 2 * nt + s2) * 512);
        }
        MIR_TRY(upload(e, (void **)&l.wffn, buf.data(), buf.size() * 2));
        {
            std::vector<float> p(FFN_PARAM_FLOATS + GELU_LUT_FLOATS);  // b1 | b2 | gamma | beta | GELU table
            gelu_table(p.data() + FFN_PARAM_FLOATS);
            std::memcpy(p.data(), t[11], sizeof(float) * FF);
            std::memcpy(p.data() + FF, t[13], sizeof(float) * H);
            std::memcpy(p.data() + FF + H, t[14], sizeof(float) * H);
            std::memcpy(p.data() + FF + 2 * H, t[15], sizeof(float) * H);
            MIR_TRY(upload(e, (void **)&l.ffn_params, p.data(), p.size() * 4));
        }
    }
    MIR_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    MIR_TRY(hipEventCreateWithFlags(&e->ws_done[0], hipEventDisableTiming));
    MIR_TRY(hipEventCreateWithFlags(&e->ws_done[1], hipEventDisableTiming));
    if (ffn_prepare() != MIR_OK || fqa_prepare() != MIR_OK) return fail(MIR_ERR_HIP);
    {
        auto kern = qkv_kernel;
        MIR_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, QKV_LDS_BYTES));
    }
    {
        auto kern = oproj_ln_kernel;
        MIR_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, OPROJ_LDS_BYTES));
    }
#undef MIR_TRY
    *out = e;
    return MIR_OK;
}

int32_t mir_encoder_destroy(mir_encoder *e) {
    free_encoder(e);
    return MIR_OK;
}

int32_t mir_encoder_info(const mir_encoder *e, int32_t *layers, int32_t *hidden, int64_t *hbm_bytes) {
    MIR_REQUIRE(e != nullptr, "handle is NULL");
    if (layers) *layers = e->layers;
    if (hidden) *hidden = H;
    if (hbm_bytes) *hbm_bytes = e->hbm_bytes;
    return MIR_OK;
}

// Core: encode sequences; out (device or host) float32 [n_seq][384].
// run_layers < 0 = all.  hidden_out (host, optional): unpacked hidden states of the LAST pass'
// padded tokens after `run_layers` layers, for tests.
static int32_t encode_impl(mir_encoder *e, const int32_t *token_ids, const int32_t *seq_lens, int32_t n_seq,
                           int32_t normalize, float *out, bool out_on_device, hipStream_t user_stream,
                           int32_t run_layers, float *hidden_out, int64_t hidden_cap_tokens) {
    MIR_REQUIRE(e != nullptr, "handle is NULL");
    MIR_REQUIRE(n_seq >= 0, "n_seq is negative");
    if (n_seq == 0) return MIR_OK;
    MIR_REQUIRE(token_ids && seq_lens && out, "NULL buffer");
    std::vector<int64_t> offs(n_seq + 1, 0);
    for (int s = 0; s < n_seq; ++s) {
        MIR_REQUIRE(seq_lens[s] >= 1 && seq_lens[s] <= 512, "sequence %d has length %d (must be 1..512)", s, seq_lens[s]);
        offs[s + 1] = offs[s] + seq_lens[s];
    }
    for (int64_t i = 0; i < offs[n_seq]; ++i)
        MIR_REQUIRE(token_ids[i] >= 0 && token_ids[i] < e->vocab, "token id %d at %lld outside the vocabulary", token_ids[i], (long long)i);
    int32_t rc = use_device(e->device, nullptr);
    if (rc != MIR_OK) return rc;
    std::lock_guard<std::mutex> lk(e->mu);
    hipStream_t s = out_on_device ? user_stream : e->stream;
    const int nl = run_layers < 0 ? e->layers : std::min(run_layers, e->layers);

    Batch batches[2];
    bool slot_busy[2] = {false, false};
    struct Pending { float *dst; size_t off, bytes; float *hid_dst; size_t hid_off, hid_bytes; } pend[2] = {};
    auto retire = [&](int slot) -> int32_t {  // wait for the slot's pass, move its pinned results to the caller
        if (!slot_busy[slot]) return MIR_OK;
        MIR_HIP(hipEventSynchronize(e->ws_done[slot]));
        slot_busy[slot] = false;
        if (pend[slot].dst) std::memcpy(pend[slot].dst, e->pin[slot] + pend[slot].off, pend[slot].bytes);
        if (pend[slot].hid_dst) std::memcpy(pend[slot].hid_dst, e->pin[slot] + pend[slot].hid_off, pend[slot].hid_bytes);
        pend[slot] = Pending{};
        return MIR_OK;
    };
    int s0 = 0, pass = 0;
    while (s0 < n_seq) {
        const int slot = pass & 1;
        Batch &b = batches[slot];
        // greedy pass: as many sequences as fit in kMaxTilesPerPass tiles
        int s1 = s0, tiles = 0;
        while (s1 < n_seq && tiles + (seq_lens[s1] + 31) / 32 <= kMaxTilesPerPass) { tiles += (seq_lens[s1] + 31) / 32; ++s1; }
        // this slot's previous pass (two passes ago) must be done before its staging vectors and
        // device buffers are reused; the pass in between keeps the GPU busy meanwhile
        rc = retire(slot);
        if (rc != MIR_OK) return rc;
        build_batch(token_ids, offs.data(), seq_lens, s0, s1, e->fused_qkv_attention && tiles > kSmallTiles, b);
        const int nt = (int)b.tiles.size();
        const size_t act_b = (size_t)nt * NFB * 2 * 64 * 16;
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
        const size_t o_ids = take(b.ids.size() * 4), o_ti = take(b.tiles.size() * sizeof(TileInfo)), o_sf = take(b.seq_first.size() * 4);
        const size_t o_un = take(b.units.size() * 4);
        const size_t o_bn = take(b.bins.size() * 4), o_tp = take(b.tile_pos.size() * 4);
        const size_t o_a = take(act_b), o_b = take(act_b), o_q = take(act_b), o_k = take(act_b), o_v = take(act_b);
        // latency path (<= kSmallTiles tiles): Y float32 [tile][384][32] and the FFN's h fragments [tile][96][64] x 16 B
        const bool small = nt <= kSmallTiles;
        const size_t o_y = small ? take((size_t)nt * NFB * 16 * 64 * 4) : 0, o_hb = small ? take((size_t)nt * 2 * NHT * 64 * 16) : 0;
        const size_t o_out = take((size_t)(s1 - s0) * H * 4);
        const size_t o_hid = hidden_out ? take((size_t)nt * 32 * H * 4) : 0;
        if (e->ws_cap[slot] < off) {
            if (e->ws[slot]) (void)hipFree(e->ws[slot]);   // idle: its event was waited for above
            e->ws[slot] = nullptr; e->ws_cap[slot] = 0;
            MIR_HIP(hipMalloc(&e->ws[slot], off));
            e->ws_cap[slot] = off;
        }
        char *w = static_cast<char *>(e->ws[slot]);
        // pinned mirror of the small host<->device regions: [ids | tiles | seq_first | out | hidden]
        const size_t p_ids = 0, p_ti = p_ids + ((b.ids.size() * 4 + 255) & ~(size_t)255);
        const size_t p_sf = p_ti + ((b.tiles.size() * sizeof(TileInfo) + 255) & ~(size_t)255);
        const size_t p_un = p_sf + ((b.seq_first.size() * 4 + 255) & ~(size_t)255);
        const size_t p_bn = p_un + ((b.units.size() * 4 + 255) & ~(size_t)255);
        const size_t p_tp = p_bn + ((b.bins.size() * 4 + 255) & ~(size_t)255);
        const size_t p_out = p_tp + ((b.tile_pos.size() * 4 + 255) & ~(size_t)255);
        const size_t p_hid = p_out + (((size_t)(s1 - s0) * H * 4 + 255) & ~(size_t)255);
        const size_t p_need = p_hid + (hidden_out ? (size_t)nt * 32 * H * 4 : 0);
        if (e->pin_cap[slot] < p_need) {
            if (e->pin[slot]) (void)hipHostFree(e->pin[slot]);
            e->pin[slot] = nullptr; e->pin_cap[slot] = 0;
            MIR_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->pin[slot]), p_need, hipHostMallocDefault));
            e->pin_cap[slot] = p_need;
        }
        char *pn = e->pin[slot];
        std::memcpy(pn + p_ids, b.ids.data(), b.ids.size() * 4);
        std::memcpy(pn + p_ti, b.tiles.data(), b.tiles.size() * sizeof(TileInfo));
        std::memcpy(pn + p_sf, b.seq_first.data(), b.seq_first.size() * 4);
        std::memcpy(pn + p_un, b.units.data(), b.units.size() * 4);
        std::memcpy(pn + p_bn, b.bins.data(), b.bins.size() * 4);
        std::memcpy(pn + p_tp, b.tile_pos.data(), b.tile_pos.size() * 4);
        // [ids | tiles | seq_first | units | bins | tile_pos] have the same 256-aligned offsets on both sides: one copy
        MIR_REQUIRE(o_ids == p_ids && o_ti == p_ti && o_sf == p_sf && o_un == p_un && o_bn == p_bn && o_tp == p_tp, "staging layout mismatch");
        MIR_HIP(hipMemcpyAsync(w + o_ids, pn + p_ids, p_tp + b.tile_pos.size() * 4, hipMemcpyHostToDevice, s));
        const int32_t *d_ids = reinterpret_cast<int32_t *>(w + o_ids);
        const TileInfo *d_ti = reinterpret_cast<TileInfo *>(w + o_ti);
        uint4 *a0 = reinterpret_cast<uint4 *>(w + o_a), *a1 = reinterpret_cast<uint4 *>(w + o_b);
        uint4 *qf = reinterpret_cast<uint4 *>(w + o_q), *kf = reinterpret_cast<uint4 *>(w + o_k), *vf = reinterpret_cast<uint4 *>(w + o_v);
        const dim3 g4((nt + 3) / 4), blk(256);
        embed_ln_kernel<<<g4, blk, 0, s>>>(d_ids, d_ti, nt, e->word, e->pos, e->type0, e->emb_g, e->emb_b, a0);
        // latency path (encoder_kernels.h, section L): the same products spread over the weight dimension
        // Single-tile sequences (queries): the layer's closing LayerNorm moves into the NEXT layer's first dispatch, so a
        // layer is 4 dispatches instead of 7 (the context goes to the otherwise unused Q buffer: the fused kernel still
        // reads the residual from a1 while it writes).
        const bool single = small && nt == s1 - s0;
        for (int li = 0; li < nl && small; ++li) {
            const Layer &l = e->L[li];
            float *Y = reinterpret_cast<float *>(w + o_y);
            uint4 *hb = reinterpret_cast<uint4 *>(w + o_hb);
            uint4 *cx = single ? qf : a1;  // context
            if (single && li == 0) {
                const int32_t arc = launch_qkv_attention_single(a0, l.wqkv, l.bqkv, d_ti, nt, cx, s);
                if (arc != MIR_OK) return arc;
            } else if (single) {
                const Layer &p = e->L[li - 1];  // its FFN block's LayerNorm: Y + b2 + residual a1 -> a0
                const int32_t arc = launch_ln_qkv_attention_single(Y, a1, p.ffn_params + FF, p.ffn_params + FF + H, p.ffn_params + FF + 2 * H,
                                                                   a0, l.wqkv, l.bqkv, d_ti, nt, cx, s);
                if (arc != MIR_OK) return arc;
            } else {
                qkv_small_kernel<<<dim3(36, nt), dim3(64), 0, s>>>(a0, l.wqkv, l.bqkv, qf, kf, vf);
                const int32_t arc = launch_attention(qf, kf, vf, d_ti, nt, cx, s);
                if (arc != MIR_OK) return arc;
            }
            oproj_small_kernel<<<dim3(NFB, nt), dim3(64), 0, s>>>(cx, l.wo, Y);
            ln_ffn1_small_kernel<<<dim3(NHT / 4, nt), dim3(256), 0, s>>>(Y, a0, l.attn_params, l.attn_params + H, l.attn_params + 2 * H, a1,
                                                                       l.wffn, l.ffn_params, hb);  // LayerNorm -> a1, FFN1 -> hb
            ffn2_small_kernel<<<dim3(NFB, nt), dim3(64), 0, s>>>(hb, l.wffn, Y);
            if (!single || li == nl - 1)
                ln_small_kernel<<<dim3(nt), dim3(256), 0, s>>>(Y, a1, l.ffn_params + FF, l.ffn_params + FF + H,
                                                              l.ffn_params + FF + 2 * H, a0);
        }
        for (int li = 0; li < nl && !small; ++li) {
            const Layer &l = e->L[li];
            // sequences of more than kFqaTiles tiles (tiles [0, nl)): QKV projection, then attention from Q / K / V in HBM; the others:
            // both in one kernel, bin by bin, Q / K / V never leaving the CU (fused_qkv_attention_kernel).  a1 = context
            const int nl = b.n_long_tiles;
            if (nl > 0) {
                qkv_kernel<<<dim3((nl + QKV_WAVES * QKV_G - 1) / (QKV_WAVES * QKV_G)), dim3(64 * QKV_WAVES), QKV_LDS_BYTES, s>>>(a0, nl, l.wqkv, l.bqkv, qf, kf, vf);
#if ENC_ATT == 2
                const int32_t arc = launch_attention2(qf, kf, vf, reinterpret_cast<const int32_t *>(w + o_un), (int)b.units.size() / 4, a1, s);
#else
                const int32_t arc = launch_attention(qf, kf, vf, d_ti, nl, a1, s);
#endif
                if (arc != MIR_OK) return arc;
            }
            {
                const int32_t frc = launch_fused_qkv_attention(a0, l.wqkv, l.bqkv, reinterpret_cast<const int32_t *>(w + o_bn),
                                                               (int)(b.bins.size() / (4 * kFqaTiles)), a1, s);
                if (frc != MIR_OK) return frc;
            }
            oproj_ln_kernel<<<dim3(std::min((nt + 3) / 4, OPROJ_MAX_GRID)), dim3(512), OPROJ_LDS_BYTES, s>>>(a1, nt, l.wo, l.attn_params, l.attn_params + H, l.attn_params + 2 * H,
                                                                   a0, a1);  // in place: a tile's two waves read its context before the first barrier and write after the last
            {
                const int32_t frc = launch_ffn(a1, nt, l.wffn, l.ffn_params, a0, s);
                if (frc != MIR_OK) return frc;
            }
        }
        MIR_HIP(hipGetLastError());
        float *d_out = out_on_device ? out + (size_t)s0 * H : reinterpret_cast<float *>(w + o_out);
        pool_normalize_kernel<<<dim3(s1 - s0), dim3(64), 0, s>>>(a0, reinterpret_cast<int32_t *>(w + o_sf), s1 - s0, normalize, d_out);
        MIR_HIP(hipGetLastError());
        if (!out_on_device) {
            MIR_HIP(hipMemcpyAsync(pn + p_out, d_out, (size_t)(s1 - s0) * H * 4, hipMemcpyDeviceToHost, s));
            pend[slot].dst = out + (size_t)s0 * H; pend[slot].off = p_out; pend[slot].bytes = (size_t)(s1 - s0) * H * 4;
        }
        if (hidden_out) {
            MIR_REQUIRE((int64_t)nt * 32 <= hidden_cap_tokens, "hidden_out too small: need %d tokens", nt * 32);
            act_unpack_kernel<<<dim3(nt), dim3(64), 0, s>>>(a0, nt, reinterpret_cast<const int32_t *>(w + o_tp), reinterpret_cast<float *>(w + o_hid));
            MIR_HIP(hipMemcpyAsync(pn + p_hid, w + o_hid, (size_t)nt * 32 * H * 4, hipMemcpyDeviceToHost, s));
            pend[slot].hid_dst = hidden_out; pend[slot].hid_off = p_hid; pend[slot].hid_bytes = (size_t)nt * 32 * H * 4;
        }
        MIR_HIP(hipEventRecord(e->ws_done[slot], s));
        slot_busy[slot] = true;
        s0 = s1;
        ++pass;
    }
    // drain: results reach the caller's buffers, the workspaces become reusable
    rc = retire(0);
    if (rc == MIR_OK) rc = retire(1);
    if (rc != MIR_OK) return rc;
    MIR_HIP(hipStreamSynchronize(s));
    return MIR_OK;
}

int32_t mir_encoder_encode(mir_encoder *e, const int32_t *token_ids_host, const int32_t *seq_lens_host, int32_t n_seq,
                           int32_t normalize, float *out_host) {
    return encode_impl(e, token_ids_host, seq_lens_host, n_seq, normalize, out_host, false, nullptr, -1, nullptr, 0);
}

int32_t mir_encoder_encode_to_device(mir_encoder *e, const int32_t *token_ids_host, const int32_t *seq_lens_host,
                                     int32_t n_seq, int32_t normalize, float *out_device, void *stream) {
    return encode_impl(e, token_ids_host, seq_lens_host, n_seq, normalize, out_device, true,
                       static_cast<hipStream_t>(stream), -1, nullptr, 0);
}

int32_t mir_encoder_debug_hidden(mir_encoder *e, const int32_t *token_ids_host, const int32_t *seq_lens_host,
                                 int32_t n_seq, int32_t run_layers, float *pooled_out_host, float *hidden_out_host,
                                 int64_t hidden_capacity_tokens) {
    return encode_impl(e, token_ids_host, seq_lens_host, n_seq, 0, pooled_out_host, false, nullptr, run_layers,
                       hidden_out_host, hidden_capacity_tokens);
}

}  // extern "C"
