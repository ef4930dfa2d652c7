// Engine + C ABI (include/texocr.h) for the MI355X-native OCRModel.generate() path.
// Host side only orchestrates: all arithmetic is in the HIP kernels of this directory.
#include "../../include/texocr.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include "common.h"
#include "conv.h"
#include "dec_attn.h"
#include "dec_gemm.h"
#include "enc_attn.h"
#include "gemm_big.h"
#include "gemm_pp.h"
#include "gemm_split.h"
#include "lat_attn.h"
#include "persist.h"
#include "prefill.h"
#include "rows.h"
#include "step.h"

namespace txo {

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

// inside a bool lambda: records the failure (look_failed) and returns false
#define HIP_TRY_B(expr)                                                                                 \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) { (void)fail(TXO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); look_failed = true; return false; } \
    } while (0)
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(TXO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                  \
    } while (0)

static const bool g_dbg = getenv("TXO_DEBUG_SYNC") != nullptr;
static void dbg(hipStream_t s, const char* what, int l = -1) {
    if (!g_dbg) return;
    hipError_t e = hipStreamSynchronize(s);
    fprintf(stderr, "[txo] %s %d -> %s\n", what, l, hipGetErrorString(e));
    fflush(stderr);
}

struct HostTensor { std::vector<int64_t> shape; std::vector<float> data; };

// -------------------------------------------------------------------------------------------------
struct EngineBase {
    txo_config cfg{};
    std::map<std::string, HostTensor> host;     // reference-keyed weights until finalize
    bool ready = false;
    virtual ~EngineBase() {}
    virtual int finalize() = 0;
    virtual int encode(const float* img, int B, int C, int H, int W, float* enc_out, hipStream_t s) = 0;
    virtual int decode_begin(const float* enc, int B, int N, int eos, hipStream_t s) = 0;
    virtual int decode_step(const int64_t* tok_in, int t, float* logits_out, int64_t* tok_out, hipStream_t s) = 0;
    virtual int decode_prefill(const int64_t* tokens, int t, float* logits_out, hipStream_t s) = 0;
    virtual int decode_set_key_mask(const unsigned char* mask, int cols, hipStream_t s) = 0;
    virtual int generate(const float* img, const float* enc, int B, int C, int H, int W, int N, int max_len, int eos,
                         int64_t* tokens_out, int* n_steps, float* logits_out, hipStream_t s) = 0;
    virtual int generate_beam(const float* img, const float* enc, int B, int C, int H, int W, int N, int beams, int max_len,
                              int eos, int64_t* tokens_out, float* scores_out, int64_t* all_tokens_out, int* n_steps,
                              hipStream_t s) = 0;
    int sample_mode = 0, sample_topk = 0; float sample_temp = 1.f; unsigned long long sample_seed = 0;
    int stop_mode = 0;                          // txo_set_stop_mode: 0 = the reference's global eos break only, 1 = per-row stop (pad behind a row's first eos)
    virtual int query(int what, int64_t* out) = 0;
    virtual int profile_enable(int on) = 0;
    virtual int profile_read(int kind, double* avg_ms, int64_t* count) = 0;
};

// rows [2X][K] -> groups of 2G rows: G "value" rows (g*G..) followed by their G "gate" rows (X + g*G..).
// G = 16 for the encoder GEMM (value / gate in adjacent MFMA column tiles), 8 for the decode GEMMs (both halves inside one tile)
static std::vector<float> interleave(const std::vector<float>& w, int X, int K, int G) {
    std::vector<float> o((size_t)2 * X * K);
    for (int g = 0; g < X / G; ++g)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < G; ++r)
                memcpy(&o[((size_t)g * 2 * G + h * G + r) * K], &w[((size_t)h * X + g * G + r) * K], sizeof(float) * K);
    return o;
}

struct EventPool {
    std::vector<hipEvent_t> ev; size_t used = 0;
    hipEvent_t next() {
        if (used == ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; ev.push_back(e); }
        return ev[used++];
    }
    ~EventPool() { for (auto e : ev) (void)hipEventDestroy(e); }
};

template <typename T>
struct Engine : EngineBase {
    // ----- device weights -----
    std::vector<void*> allocs;
    struct AttnW { T* wqkv = nullptr; T* wq = nullptr; T* wo = nullptr; float* bo = nullptr;
                   T* wo16 = nullptr; float* bo16 = nullptr;    // decoder: second copy interleaved by 16 (prefill: the large-GEMM epilogues)
                   T* wkT = nullptr; T* wv = nullptr;   
                   T* wqp = nullptr; T* wo_f = nullptr; };      // ... folded where inner == 2 D: q' = z (0.125 Wk_h^T Wq_h)^T [heads*D][D]; Wo' = Wo blockdiag(Wv_h) [2D][heads*D], rows interleaved by 8        // decoder cross attention, latent form (lat_attn.h): Wk per head transposed [heads][D][64], Wv [inner][D]
    struct MlpW { T* w1 = nullptr; float* b1 = nullptr; T* w2 = nullptr; float* b2 = nullptr;
                  T* w1_16 = nullptr; float* b1_16 = nullptr; };   // decoder, wide rows: second copy interleaved by 16 (large-batch FFN-in)
    float *cls = nullptr, *pos = nullptr, *patch_b = nullptr; T* patch_w = nullptr;
    // hybrid ResNetV2 embedder: standardised conv weights as [oc][kh][kw][ic] of T, GroupNorm affine fp32
    struct GnW { float* g = nullptr; float* b = nullptr; };
    template <typename TB> struct BlockW { TB *c1 = nullptr, *c2 = nullptr, *c3 = nullptr, *ds = nullptr; GnW n1, n2, n3, nds; int cin, mid, cout, stride; bool has_ds; };
    // the backbone in storage type TB: weights, the four activation buffers, the 1x1 projection to the embedding width
    template <typename TB> struct Backbone { TB* stem_w = nullptr; GnW stem_gn; std::vector<BlockW<TB>> blocks; TB* act[4] = {nullptr, nullptr, nullptr, nullptr}; TB* proj_w = nullptr; };
    Backbone<T> bk; Backbone<float> bk32;
    float *gn_partial = nullptr, *gn_stats = nullptr;
    bool hybrid = false;
    // bf16 mode, hybrid embedder: the BACKBONE still stores and multiplies in fp32 (bk32; r06).  With the reference's 45 conv / GroupNorm
    // layers at random weights a perturbation of 2^-9 anywhere -- rounding the input pixels alone -- moves the backbone's output by 10-20 %
    // (tests/test_oracle_golden.py), so no choice of WHICH tensors are stored as bf16 gives a usable mode; fp32 here costs ~3 ms per 64
    // images and puts the bf16 engine's encoder output within 1-2 % of the fp32 reference.  TXO_BACKBONE_BF16=1: the bf16 backbone
    // (r02-r05; kernels checked against a CPU emulation of bf16 storage) for checkpoints known to tolerate it.
    bool bk_fp32 = sizeof(T) == 2 && getenv("TXO_BACKBONE_BF16") == nullptr;
    float *enc_g = nullptr, *enc_b = nullptr, *encn_g = nullptr, *encn_b = nullptr, *enc_gb = nullptr;   // enc_gb: gamma then beta (GEMM epilogues)
    std::vector<AttnW> enc_attn; std::vector<MlpW> enc_mlp;
    float *tok_emb = nullptr, *pos_emb = nullptr, *dec_g = nullptr, *dec_b = nullptr, *decn_g = nullptr, *decn_b = nullptr, *dec_gb = nullptr;
    T* wlog = nullptr; float* blog = nullptr; T* wckv = nullptr;
    std::vector<AttnW> dec_self, dec_cross; std::vector<MlpW> dec_mlp;
    // ----- workspace -----
    float *ex = nullptr, *ey = nullptr, *eqkv = nullptr, *eenc = nullptr, *estats = nullptr; T *ez = nullptr, *eao = nullptr, *ehid = nullptr;
    T* enc_t = nullptr;               // bf16 copy of the encoder output (A operand of the cross K/V GEMM)
    T *ckv = nullptr, *skv = nullptr;  // cross [Ld][2][B*h][N][64], self [Ld][2][B*h][Tmax][64]
    float *dx = nullptr, *dy = nullptr, *dq = nullptr, *dlogits = nullptr; T *dao = nullptr, *dhid = nullptr, *dz = nullptr;
    T* zc = nullptr;                  // self attention in latent form: history of normalised block inputs [Ld][B][Tmax][D] (lat_attn.h)
    T *dqt = nullptr, *dqp = nullptr, *dcl = nullptr;   // latent cross attention: q [B][inner], q' and c [B][heads*D] in the storage type
    int64_t* cur_tok = nullptr; int *eos_seen = nullptr, *done_flag = nullptr; StepState* st = nullptr;
    unsigned char* kmask = nullptr; bool kmask_on = false;   // padding mask over the decoded positions of a decode_step session (txo_decode_set_key_mask)
    // ----- decode session -----
    // A decode runs as 1..MAXL independent "lanes" (contiguous row ranges of the batch), each on its own HIP
    // stream with its own step state, so the latency chains of one lane's small kernels overlap the other's.
    // Every lane's step is a fixed launch sequence (the position lives on the device) -> captured once as a
    // hipGraph and replayed per step.
    static constexpr int MAXL = 4;
    struct Lane {
        int b0 = 0, nb = 0;
        hipStream_t stream = nullptr;      // lane 0 runs on the caller's stream
        hipStream_t own = nullptr;         // engine-owned stream for lanes > 0
        // captured steps, by what they were built for: {b0, nb, N, eos, sB, sImg, form flags} (the cache strides sB / sImg and the row count are
        // baked into a graph's launches).  Per-row stop replays a step per row count of the shrinking range (multiples of 16): a handful of
        // entries per range, built once and kept across generates.  exec = the entry the current decode replays.
        std::map<std::array<int, 7>, std::pair<hipGraph_t, hipGraphExec_t>> graphs;
        hipGraphExec_t exec = nullptr;
    };
    Lane lanes[MAXL];
    int n_lanes = 1, max_lanes = 2;
    // Run-time knobs of generate() (development / test switches).  Read ONCE per engine, at creation; txo_engine_query(TXO_Q_RELOAD_KNOBS)
    // reads them again (the Python binding does that when it sees the TXO_* environment change between two calls: tests flip
    // TXO_PERSIST / TXO_LANES on a live engine).  Everything else in this struct's neighbourhood is read once, in the member initialisers.
    struct RunKnobs {
        int persist = -1, graph = -1, lanes = 0;        // -1 / 0 = unset
        bool stamps = false, pstamps = false;
        std::string stamps_file, pstamps_file;
        int stagger_ticks = 0, inject_fail = 0;
        bool has_stagger = false, has_inject = false;
        void read() {
            *this = RunKnobs{};
            if (const char* e = getenv("TXO_PERSIST")) persist = atoi(e) != 0;
            if (const char* e = getenv("TXO_GRAPH")) graph = atoi(e) != 0;
            if (const char* e = getenv("TXO_LANES")) lanes = std::max(1, atoi(e));
            if (const char* e = getenv("TXO_STAMPS")) { stamps = true; stamps_file = e; }
            if (const char* e = getenv("TXO_PSTAMPS")) { pstamps = true; pstamps_file = e; }
            if (const char* e = getenv("TXO_PS_STAGGER_US")) { has_stagger = true; stagger_ticks = (int)(atof(e) * 100.0); }
            if (const char* e = getenv("TXO_PERSIST_INJECT_FAIL")) { has_inject = true; inject_fail = atoi(e); }
        }
    } knobs;
    bool self_plain = getenv("TXO_SELF_FUSED") == nullptr;
    // experiment knobs are read ONCE per engine (never on a launch path)
    bool dec_wide_off = getenv("TXO_DEC_WIDE_OFF") != nullptr;
    // narrow decoder, folded latent out-projection (K = heads * D): 32 x 32 blocks from wide_min_rows rows of a range on, 32 x 16 blocks from
    // wide_mid_rows on, 16 x 16 blocks below
    static constexpr int wide_min_rows = 100000;   // (257 until the tiled operands: at a beam search's 320 rows per range 32 x 16 is now ahead, 98.5 vs 100.5 ms)
    static constexpr int wide_mid_rows = 129;
    // encoder GEMM outputs with non-temporal stores (gemm_big.h: store8; the epilogues keep the parameter).  probes/pp_store_policy.hip
    // measured +22 % for a plain 256x256 store epilogue at K = 768 (1.85 GB of output per launch), but the encoder's own epilogues
    // (GeGLU halves the columns, the fp32 stream is read-modify-write) run the same with either policy: 45.84 vs 45.77 ms per ViT-Base
    // encode (probes/enc_nt.py) -- so the default stays the plain store.
    int enc_nt(size_t) const { return 0; }
    // cross attention in latent form (lat_attn.h): scores / values against the raw encoder rows instead of projected K/V panels.
    // latent_ok: the tile exists for this engine's width / storage type.  lat_mode (TXO_LATENT, read once): 1 = every decode runs with
    // launches in latent form, 0 = never, unset = where it measured faster (auto_latent).  use_latent: what the current session does.
    bool latent_ok = false, use_latent = false;
    int n_cus = 256;                  // compute units of this engine's device (init): one latent tile per CU is the grouping target
    // lat_self: this session's SELF attention also runs in latent form (the history is z, not k / v: a quarter of the bytes at config.yml
    // dims).  Only inside generate() / generate_beam() with launches in latent form: a session opened through txo_decode_begin may be
    // prefilled or masked, which work on the K/V history.  OPT-IN (TXO_LATENT_SELF=1, read once per engine): measured on MI355X it
    // ties the K/V history at batch 256 (76.5 vs 76.7 ms per generate: the core is 9.8 us against 14.8 for the K/V kernel, averaged over
    // the 256 positions, but the folded output projection's K = heads*D costs 9.1 us against 4.8) and loses in beam search (121 vs 115 ms
    // at 5 x 128); what it saves is history capacity (a quarter).  profiles/r05_latent_self.txt.
    bool lat_self = false;
    int lat_self_env = getenv("TXO_LATENT_SELF") ? atoi(getenv("TXO_LATENT_SELF")) : 0;
    bool lat_self_ok() const {
        // (the beam slot table of the tile covers 16 keys x its wave count x LA_PATH_TILES positions: the bf16 tile at width 256 runs on 4 waves)
        const int waves = (D <= 256 && !(lat_nw4 && D == 256)) ? 8 : 4;
        return lat_self_env != 0 && zc != nullptr && !dec_self.empty() && dec_self[0].wqp != nullptr && Tmax <= 16 * waves * LA_PATH_TILES;
    }
    int lat_mode = getenv("TXO_LATENT") ? atoi(getenv("TXO_LATENT")) : -1;
    int lat_g_env = getenv("TXO_LAT_G") ? atoi(getenv("TXO_LAT_G")) : 0;
    // bf16, width 256: the latent tile on FOUR waves (r05).  Same one tile per CU, half the waves to
    // merge and twice the keys per wave: the whole-batch launch at 256 rows 19.3 -> 17.9 us (rocprofv3), generate +2-3 % at 130-192 rows,
    // beam search 5 x 128 +5 %.  (Two such tiles per CU, 512 tiles of 4 heads at 256 rows, were slower: 22.2 us -- every row's encoder
    // rows cross the CU's L2 port twice.)  The wave count changes the order in which a row's key tiles are merged: low bits differ from the
    // 8-wave tile's, within the same bound against the reference.
    static constexpr bool lat_nw4 = sizeof(T) == 2;
    bool ckv_valid = false;           // the projected cross K/V panels of this session exist (the prefill needs them; the latent form does not)
    bool use_pp = getenv("TXO_GEMM_OLD") == nullptr;   // bf16: 256x256 LDS-DMA GEMM for the encoder-side projections
    static constexpr int enc_walk = 1;   // encoder kernels walk the rows alternately up and down (encode())
    int pp_tr = getenv("TXO_PP_TR") ? (atoi(getenv("TXO_PP_TR")) != 0) : -1;   // its epilogue form: 1 direct, 0 staged through LDS, unset = by epilogue (gemm_pp.h)
    int pp_ct = getenv("TXO_PP_CT") ? atoi(getenv("TXO_PP_CT")) : 0;     // experiment: column tiles per band of the multi-band GEMMs (0 = by size, gemm_pp.h)
    int pp_sb_mb = getenv("TXO_PP_SB_MB") ? atoi(getenv("TXO_PP_SB_MB")) : PP_SB_MB;   // ... its row super-blocks: MB of A per super-block, 0 = none (gemm_pp.h)
    // Two row ranges need two streams whose launches really run side by side.  Which HIP streams do depends on how the runtime mapped them onto
    // hardware queues -- on every stream the process created before (profiles/r06_b256_stream_pairs.txt: 66 / 72 / 80 / 110 ms per generate
    // at batch 256 for the same engine, by the number of streams created earlier) -- so the pair is CHOSEN by measurement, once per engine,
    // the first time two ranges are wanted: tune_lane_streams().  TXO_TUNE_LANES=0: off (range 0 on the caller's stream, as r02-r05).
    int tune_lanes = getenv("TXO_TUNE_LANES") ? atoi(getenv("TXO_TUNE_LANES")) : 1;
    bool lanes_tuned = false;
    double tune_best_ms = 0, tune_worst_ms = 0;
    int* flags_host = nullptr;        // pinned: done flags of the chunk being looked at (generate)
    hipEvent_t ev_flags[MAXL] = {};
    // per-row stop (step.h): batch row held by every slot of a row range, scratch of the compaction, {live rows, moves} per range;
    // live_host (pinned): the ranges' finished-row counts on their way to the host, ev_live behind them
    int *row_map = nullptr, *row_map2 = nullptr, *cmoves = nullptr, *cinfo = nullptr; int64_t* cur_tok2 = nullptr;
    int* live_host = nullptr; hipEvent_t ev_live[MAXL] = {};
    bool stop_graph_on = getenv("TXO_STOP_GRAPH") ? atoi(getenv("TXO_STOP_GRAPH")) != 0 : true;
    int stop_every = getenv("TXO_STOP_EVERY") ? std::max(1, atoi(getenv("TXO_STOP_EVERY"))) : 16;   // positions between two looks at the live-row counts
    int stop_gain = getenv("TXO_STOP_GAIN") ? std::max(1, atoi(getenv("TXO_STOP_GAIN"))) : 16;      // rows a compaction must free (one 16-row tile)
    int last_compactions = 0;         // compactions of the last generate (TXO_Q_LAST_COMPACTIONS)
    bool look_failed = false;
    int step_host_t = -1;             // position of the step being enqueued when the host knows it (see enqueue_step)
    // TXO_STAMPS=<file>: diagnostic -- every decode launch of ONE step records per-block entry / mid / exit times
    unsigned long long* stamp_buf = nullptr; int stamp_slot = -1; static constexpr int STAMP_BLOCKS = 2048, STAMP_KERNELS = 64;
    std::vector<std::string> stamp_names;
    unsigned long long* next_stamp(const char* name) {
        if (stamp_slot < 0 || stamp_slot >= STAMP_KERNELS) return nullptr;
        stamp_names.push_back(name);
        return stamp_buf + (size_t)(stamp_slot++) * STAMP_BLOCKS * 3;
    }
    hipStream_t cap_stream = nullptr;      // graphs are captured here, never on the caller's stream
    hipEvent_t ev_fork = nullptr, ev_join[MAXL] = {nullptr, nullptr, nullptr, nullptr};
    int64_t* tok_buf = nullptr;            // [Bmax][Tmax] generated ids (engine-owned so graphs do not bake user pointers)
    int sB = 0, sN = 0, sImg = 0; bool session = false;   // decode rows, encoder tokens, images behind the cross K/V cache
    // persistent decode launch (persist.h): control block (device + pinned host copy), per-stage stamps of one position
    PersistCtl* pctl = nullptr; PersistCtl* pctl_host = nullptr; unsigned long long* pstamps = nullptr;
    int persist_fallbacks = 0;                            // launches that gave up (placement / time-out) and were redone with launches
    int persist_strikes = 0, persist_cooldown = 0;        // two give-ups in a row switch the fast path off for 64 generates (then it is tried again)
    bool last_persist = false;                            // the last generate() ran as ONE persistent launch
    int last_ranges = 1;                                  // row ranges (streams) the last launch-path decode ran on
    // beam search state (rows = images * beams)
    float* bscore = nullptr; int* bfin = nullptr; short* bpath[2] = {nullptr, nullptr}; short* bparent = nullptr; int* btok = nullptr;
    struct BeamCtx { int k; const short* path_cur; short* path_nxt; };
    // ----- profiling -----
    bool prof = false;                // full: per-launch cross-attention events + markers around every encode and step
    bool prof_cross = false;          // light: only the cross-attention dispatches carry events (no extra packets)
    bool prof_persist = false;        // mode 3: HIP events bound to the persistent decode launch (texocr_amd/csrc/persist.h)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_persist;
    unsigned cross_seq = 0;
    EventPool pool;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_cross, ev_enc, ev_step;

    int D, Ie, Id, Fe, Fd, V, Tmax, Nmax, Bmax;

    ~Engine() override {
        for (auto& ln : lanes) {
            for (auto& g : ln.graphs) { if (g.second.second) (void)hipGraphExecDestroy(g.second.second); if (g.second.first) (void)hipGraphDestroy(g.second.first); }
            if (ln.own) (void)hipStreamDestroy(ln.own);
        }
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (auto e : ev_join) if (e) (void)hipEventDestroy(e);
        for (auto e : ev_flags) if (e) (void)hipEventDestroy(e);
        for (auto e : ev_live) if (e) (void)hipEventDestroy(e);
        if (flags_host) (void)hipHostFree(flags_host);
        if (live_host) (void)hipHostFree(live_host);
        if (pctl_host) (void)hipHostFree(pctl_host);
        for (void* p : allocs) (void)hipFree(p);
    }

    // All device memory of the engine comes from two arenas (workspace at create, weights at finalize), one hipMalloc
    // each: a decode step touches ~40 small buffers and every launch starts with cold caches, so sharing large pages keeps
    // address-translation misses off the operand round trip each dependent launch already pays.
    struct Arena { char* base = nullptr; size_t cap = 0, off = 0; bool measuring = false; };
    Arena arena;
    template <typename U> int dalloc(U** p, size_t n) {
        const size_t bytes = (n * sizeof(U) + 511) & ~(size_t)255;
        if (arena.measuring) { arena.off += bytes; *p = nullptr; return 0; }
        if (arena.base && arena.off + bytes <= arena.cap) {
            *p = reinterpret_cast<U*>(arena.base + arena.off);
            arena.off += bytes;
            return 0;
        }
        void* q = nullptr;
        HIP_TRY(hipMalloc(&q, bytes));
        allocs.push_back(q);
        *p = reinterpret_cast<U*>(q);
        return 0;
    }
    int arena_begin(size_t bytes) {
        void* q = nullptr;
        HIP_TRY(hipMalloc(&q, bytes + (2u << 20)));
        allocs.push_back(q);
        arena = Arena{};
        arena.base = reinterpret_cast<char*>(((uintptr_t)q + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));   // 2 MiB aligned
        arena.cap = bytes;
        return 0;
    }
    int upload_f32(float** dst, const std::vector<float>& v) {
        if (int r = dalloc(dst, v.size())) return r;
        HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        return 0;
    }
    int upload_T(T** dst, const std::vector<float>& v) { return upload_as<T>(dst, v); }
    template <typename TB> int upload_as(TB** dst, const std::vector<float>& v) {
        if (int r = dalloc(dst, v.size())) return r;
        if constexpr (sizeof(TB) == 4) {
            HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        } else {
            std::vector<uint16_t> h(v.size());
            for (size_t i = 0; i < v.size(); ++i) {      // round-to-nearest-even f32 -> bf16 (inputs are finite)
                uint32_t u; memcpy(&u, &v[i], 4);
                h[i] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
            }
            HIP_TRY(hipMemcpy(*dst, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        return 0;
    }

    const HostTensor* get(const std::string& key, std::initializer_list<int64_t> shape) {
        auto it = host.find(key);
        if (it == host.end()) { g_err = "missing weight: " + key; return nullptr; }
        if (it->second.shape != std::vector<int64_t>(shape)) { g_err = "wrong shape for " + key; return nullptr; }
        return &it->second;
    }

    // shared LayerNorm of a stack: every alias layers.{s}.0.* that was provided must be identical
    int shared_ln(const std::string& prefix, int n_sub, float** g, float** b) {
        const HostTensor* g0 = get(prefix + ".layers.0.0.weight", {D});
        const HostTensor* b0 = get(prefix + ".layers.0.0.bias", {D});
        if (!g0 || !b0) return TXO_E_STATE;
        for (int s = 1; s < n_sub; ++s)
            for (const char* leaf : {"weight", "bias"}) {
                auto it = host.find(prefix + ".layers." + std::to_string(s) + ".0." + leaf);
                if (it == host.end()) continue;
                const HostTensor& ref = (leaf[0] == 'w') ? *g0 : *b0;
                if (it->second.data != ref.data)
                    return fail(TXO_E_INVALID, prefix + ".layers." + std::to_string(s) + ".0." + leaf +
                                " differs from layers.0.0: the reference shares ONE LayerNorm per stack");
            }
        if (int r = upload_f32(g, g0->data)) return r;
        return upload_f32(b, b0->data);
    }

    // inner == 2 D (config.yml dims: 8 heads of 64 at width 256): the latent form's two per-head projections fold into their neighbours at
    // load time (products in double, rounded once to the storage type) and an attention sub-layer is THREE launches instead of five:
    //   q'[h*D + d'] = sum_d z[d] * M[h*D + d'][d],   M = 0.125 Wk_h^T Wq_h            (in the LN-prologue GEMM, N = heads*D)
    //   y[n]         = sum_{h,d} c[h*D + d] * Wo'[n][h*D + d],  Wo' = Wo[:, h] Wv_h     (in the gated output projection, K = heads*D)
    // (attention.py:124-127,148,166,172-173: k and v are linear images of the SAME D-element row -- the raw encoder row in the cross
    // attention, the normalised block input in the self attention, attention.py:114-116)
    int fold_latent(const HostTensor* q, const HostTensor* k, const HostTensor* v, const HostTensor* wo, int inner, AttnW* w, int G) {
        const int H = inner / DH, HD = H * D;
        std::vector<float> M((size_t)HD * D), WoF((size_t)2 * D * HD);
        std::vector<double> acc(std::max(D, HD));
        for (int h = 0; h < H; ++h)
            for (int d1 = 0; d1 < D; ++d1) {
                std::fill(acc.begin(), acc.begin() + D, 0.0);
                for (int j = 0; j < DH; ++j) {
                    const double kk = (double)ATTN_SCALE * k->data[((size_t)h * DH + j) * D + d1];
                    const float* qr = &q->data[((size_t)h * DH + j) * D];
                    for (int d = 0; d < D; ++d) acc[d] += kk * qr[d];
                }
                for (int d = 0; d < D; ++d) M[((size_t)h * D + d1) * D + d] = (float)acc[d];
            }
        for (int n = 0; n < 2 * D; ++n) {
            std::fill(acc.begin(), acc.begin() + HD, 0.0);
            for (int f = 0; f < inner; ++f) {
                const double wn = wo->data[(size_t)n * inner + f];
                const float* vr = &v->data[(size_t)f * D];
                double* dst = &acc[(size_t)(f / DH) * D];
                for (int d = 0; d < D; ++d) dst[d] += wn * vr[d];
            }
            for (int c2 = 0; c2 < HD; ++c2) WoF[(size_t)n * HD + c2] = (float)acc[c2];
        }
        if (int r = upload_T(&w->wqp, M)) return r;
        return upload_T(&w->wo_f, interleave(WoF, D, HD, G));
    }

    int load_attn(const std::string& p, int inner, bool cross, AttnW* w, std::vector<float>* kv_concat, int G) {
        const HostTensor *q = get(p + ".q.weight", {inner, D}), *k = get(p + ".k.weight", {inner, D}),
                         *v = get(p + ".v.weight", {inner, D}), *wo = get(p + ".fc_out.0.weight", {2 * D, inner}),
                         *bo = get(p + ".fc_out.0.bias", {2 * D});
        if (!q || !k || !v || !wo || !bo) return TXO_E_STATE;
        if (cross) {
            if (int r = upload_T(&w->wq, q->data)) return r;
            kv_concat->insert(kv_concat->end(), k->data.begin(), k->data.end());
            kv_concat->insert(kv_concat->end(), v->data.begin(), v->data.end());
            // latent form: q'_h = 0.125 Wk_h^T q_h is a per-head GEMM whose weight rows are COLUMNS of Wk_h -> stored transposed per head, [heads][D][64]
            std::vector<float> kT((size_t)inner * D);
            for (int h = 0; h < inner / DH; ++h)
                for (int d = 0; d < D; ++d)
                    for (int j = 0; j < DH; ++j) kT[((size_t)h * D + d) * DH + j] = ATTN_SCALE * k->data[((size_t)h * DH + j) * D + d];   // (0.125: exact)
            if (int r = upload_T(&w->wkT, kT)) return r;
            if (int r = upload_T(&w->wv, v->data)) return r;
            if (latent_fold()) { if (int r = fold_latent(q, k, v, wo, inner, w, G)) return r; }
        } else {
            std::vector<float> cat(q->data);
            cat.insert(cat.end(), k->data.begin(), k->data.end());
            cat.insert(cat.end(), v->data.begin(), v->data.end());
            if (int r = upload_T(&w->wqkv, cat)) return r;
            // decoder self attention in latent form (r05): the same two folds, against the history of normalised block inputs
            // (opt-in, TXO_LATENT_SELF=1: without it neither the fold nor the z history exists)
            if (G == 8 && latent_fold() && lat_self_env != 0) { if (int r = fold_latent(q, k, v, wo, inner, w, G)) return r; }
        }
        if (int r = upload_T(&w->wo, interleave(wo->data, D, inner, G))) return r;
        if (G != 16) {                                        // decoder: the prefill runs these projections on the encoder-side GEMMs
            if (int r = upload_T(&w->wo16, interleave(wo->data, D, inner, 16))) return r;
            if (int r = upload_f32(&w->bo16, interleave(bo->data, D, 1, 16))) return r;
        }
        return upload_f32(&w->bo, interleave(bo->data, D, 1, G));
    }
    int load_mlp(const std::string& p, int F, MlpW* w, int G) {
        const HostTensor *w1 = get(p + ".fc_in.fc.weight", {2 * F, D}), *b1 = get(p + ".fc_in.fc.bias", {2 * F}),
                         *w2 = get(p + ".fc_out.weight", {D, F}), *b2 = get(p + ".fc_out.bias", {D});
        if (!w1 || !b1 || !w2 || !b2) return TXO_E_STATE;
        if (int r = upload_T(&w->w1, interleave(w1->data, F, D, G))) return r;
        if (int r = upload_f32(&w->b1, interleave(b1->data, F, 1, G))) return r;
        if (int r = upload_T(&w->w2, w2->data)) return r;
        if (G == 8) {       // decoder: the prefill's FFN-in; also enqueue_step: FFN-in of wide bf16 decoders at >= 128 rows goes through the large-GEMM kernel
            if (int r = upload_T(&w->w1_16, interleave(w1->data, F, D, 16))) return r;
            if (int r = upload_f32(&w->b1_16, interleave(b1->data, F, 1, 16))) return r;
        }
        return upload_f32(&w->b2, b2->data);
    }

    // StdConv2d weight standardisation (resnet.py:58-61): per output channel over (ic, kh, kw), biased variance,
    // eps 1e-6 inside the sqrt (F.batch_norm, training=True).  Input-independent, so folded at load time; the result
    // is stored [oc][kh][kw][ic] (K order of the implicit-GEMM loader).  kpad > 0 pads K with zeros (stem: 49 -> 64).
    template <typename TB> int upload_conv(TB** dst, const HostTensor& w, int kpad) {
        const int oc = (int)w.shape[0], ic = (int)w.shape[1], kh = (int)w.shape[2], kw = (int)w.shape[3];
        const int k = ic * kh * kw, kk = kpad > 0 ? kpad : k;
        std::vector<float> o((size_t)oc * kk, 0.f);
        for (int n = 0; n < oc; ++n) {
            const float* src = &w.data[(size_t)n * k];
            double m = 0, v = 0;
            for (int i = 0; i < k; ++i) m += src[i];
            m /= k;
            for (int i = 0; i < k; ++i) v += (src[i] - m) * (src[i] - m);
            const double rs = 1.0 / std::sqrt(v / k + 1e-6);
            for (int c2 = 0; c2 < ic; ++c2)
                for (int y = 0; y < kh; ++y)
                    for (int x = 0; x < kw; ++x)
                        o[(size_t)n * kk + ((size_t)y * kw + x) * ic + c2] = (float)((src[((size_t)c2 * kh + y) * kw + x] - m) * rs);
        }
        return upload_as<TB>(dst, o);
    }
    int load_gn(const std::string& p, int ch, GnW* g) {
        const HostTensor *w = get(p + ".weight", {ch}), *b = get(p + ".bias", {ch});
        if (!w || !b) return TXO_E_STATE;
        if (int r = upload_f32(&g->g, w->data)) return r;
        return upload_f32(&g->b, b->data);
    }
    template <typename TB> int load_backbone(const std::string& p, Backbone<TB>& k) {
        const HostTensor* t;
        if (!(t = get(p + ".stem.0.weight", {64, 1, 7, 7}))) return TXO_E_STATE;
        if (int r = upload_conv(&k.stem_w, *t, 64)) return r;
        if (int r = load_gn(p + ".stem.1", 64, &k.stem_gn)) return r;
        static const int depths[3] = {2, 4, 6}, chans[3] = {256, 512, 1024};
        int prev = 64;
        for (int st = 0; st < 3; ++st)
            for (int i = 0; i < depths[st]; ++i) {
                BlockW<TB> b{};
                b.cin = prev; b.cout = chans[st]; b.mid = chans[st] / 4; b.stride = (i == 0 && st > 0) ? 2 : 1; b.has_ds = i == 0;
                const std::string q = p + ".stages." + std::to_string(st) + ".stage_blocks." + std::to_string(i);
                // every layer is registered twice (block_list.N and block.N, resnet.py:132-141); if both are given they must agree
                for (int j = 0; j < 6; ++j)
                    for (const char* leaf : {".weight", ".bias"}) {
                        auto a = host.find(q + ".block_list." + std::to_string(j) + leaf), c2 = host.find(q + ".block." + std::to_string(j) + leaf);
                        if (a != host.end() && c2 != host.end() && a->second.data != c2->second.data)
                            return fail(TXO_E_INVALID, q + ".block." + std::to_string(j) + leaf + " differs from its block_list alias");
                    }
                if (b.has_ds) {
                    if (!(t = get(q + ".downsample.conv.weight", {b.cout, b.cin, 1, 1}))) return TXO_E_STATE;
                    if (int r = upload_conv(&b.ds, *t, 0)) return r;
                    if (int r = load_gn(q + ".downsample.norm", b.cout, &b.nds)) return r;
                }
                if (!(t = get(q + ".block_list.0.weight", {b.mid, b.cin, 1, 1}))) return TXO_E_STATE;
                if (int r = upload_conv(&b.c1, *t, 0)) return r;
                if (int r = load_gn(q + ".block_list.1", b.mid, &b.n1)) return r;
                if (!(t = get(q + ".block_list.2.weight", {b.mid, b.mid, 3, 3}))) return TXO_E_STATE;
                if (int r = upload_conv(&b.c2, *t, 0)) return r;
                if (int r = load_gn(q + ".block_list.3", b.mid, &b.n2)) return r;
                if (!(t = get(q + ".block_list.4.weight", {b.cout, b.mid, 1, 1}))) return TXO_E_STATE;
                if (int r = upload_conv(&b.c3, *t, 0)) return r;
                if (int r = load_gn(q + ".block_list.5", b.cout, &b.n3)) return r;
                k.blocks.push_back(b);
                prev = b.cout;
            }
        return 0;
    }

    int finalize() override {
        if (ready) return fail(TXO_E_STATE, "weights already finalized");
        const txo_config& c = cfg;
        // non-finite weights are refused: the library is built with -fno-honor-nans (build.py), and a NaN / inf in a weight would reach
        // every row of every batch.  (Non-finite PIXELS reach only their own image's rows: txo_encode's contract.)
        for (auto& kv : host)
            for (float v : kv.second.data)
                if (!std::isfinite(v)) return fail(TXO_E_INVALID, "non-finite value in weight " + kv.first);
        {   // weights arena: fp32 size of everything handed in is an upper bound for both storage types
            size_t bytes = 0;
            for (auto& kv : host) bytes += kv.second.data.size() * sizeof(float) + 512;
            // + the decoder's second (16-interleaved) copies of the gated projections for the prefill
            bytes += (size_t)cfg.dec_layers * ((size_t)2 * 2 * D * Id + (size_t)2 * Fd * D + 4 * D + 2 * Fd + 4096) * sizeof(float);
            // + the latent form's copies of the cross-attention Wk (transposed per head) and Wv, and the folded pair where inner == 2 D
            bytes += (size_t)cfg.dec_layers * ((size_t)2 * D * Id + 1024) * sizeof(float);
            if (latent_fold()) bytes += (size_t)cfg.dec_layers * ((size_t)3 * cfg.dec_heads * D * D + 1024) * sizeof(float);
            if (int r = arena_begin(bytes + (1u << 20))) return r;
        }
        const int npos = 1 + (c.canvas_h / 16) * (c.canvas_w / 16);
        const HostTensor *t;
        if (!(t = get("encoder.cls_token", {1, 1, D}))) return TXO_E_STATE;
        if (int r = upload_f32(&cls, t->data)) return r;
        if (!(t = get("encoder.pos_embed", {1, npos, D}))) return TXO_E_STATE;
        if (int r = upload_f32(&pos, t->data)) return r;
        if (!hybrid) {
            if (!(t = get("encoder.patch_embed.proj.weight", {D, c.in_channels, 16, 16}))) return TXO_E_STATE;
            if (int r = upload_T(&patch_w, t->data)) return r;
        } else {
            if (!(t = get("encoder.patch_embed.proj.weight", {D, 1024, 1, 1}))) return TXO_E_STATE;
            if (bk_fp32) {
                if (int r = load_backbone("encoder.patch_embed.backbone_net", bk32)) return r;
                if (int r = upload_as<float>(&bk32.proj_w, t->data)) return r;
            } else {
                if (int r = load_backbone("encoder.patch_embed.backbone_net", bk)) return r;
                if (int r = upload_T(&patch_w, t->data)) return r;
            }
        }
        if (!(t = get("encoder.patch_embed.proj.bias", {D}))) return TXO_E_STATE;
        if (int r = upload_f32(&patch_b, t->data)) return r;
        if (int r = shared_ln("encoder.attn_layers", 2 * c.enc_layers, &enc_g, &enc_b)) return r;
        {
            std::vector<float> gb(get("encoder.attn_layers.layers.0.0.weight", {D})->data);
            const std::vector<float>& bb = get("encoder.attn_layers.layers.0.0.bias", {D})->data;
            gb.insert(gb.end(), bb.begin(), bb.end());
            if (int r = upload_f32(&enc_gb, gb)) return r;
        }
        enc_attn.resize(c.enc_layers); enc_mlp.resize(c.enc_layers);
        for (int l = 0; l < c.enc_layers; ++l) {
            const std::string p = "encoder.attn_layers.layers.";
            if (int r = load_attn(p + std::to_string(2 * l) + ".1", Ie, false, &enc_attn[l], nullptr, 16)) return r;
            if (int r = load_mlp(p + std::to_string(2 * l + 1) + ".1", Fe, &enc_mlp[l], 16)) return r;
        }
        if (!(t = get("encoder.norm.weight", {D}))) return TXO_E_STATE;
        if (int r = upload_f32(&encn_g, t->data)) return r;
        if (!(t = get("encoder.norm.bias", {D}))) return TXO_E_STATE;
        if (int r = upload_f32(&encn_b, t->data)) return r;

        if (!(t = get("decoder.net.token_embedding.weight", {V, D}))) return TXO_E_STATE;
        if (int r = upload_f32(&tok_emb, t->data)) return r;
        if (!(t = get("decoder.net.pos_embedding.embedding.weight", {Tmax, D}))) return TXO_E_STATE;
        if (int r = upload_f32(&pos_emb, t->data)) return r;
        if (int r = shared_ln("decoder.net.attn_layers", 3 * c.dec_layers, &dec_g, &dec_b)) return r;
        {
            std::vector<float> gb(get("decoder.net.attn_layers.layers.0.0.weight", {D})->data);
            const std::vector<float>& bb = get("decoder.net.attn_layers.layers.0.0.bias", {D})->data;
            gb.insert(gb.end(), bb.begin(), bb.end());
            if (int r = upload_f32(&dec_gb, gb)) return r;
        }
        dec_self.resize(c.dec_layers); dec_cross.resize(c.dec_layers); dec_mlp.resize(c.dec_layers);
        std::vector<float> kv_concat;
        for (int l = 0; l < c.dec_layers; ++l) {
            const std::string p = "decoder.net.attn_layers.layers.";
            if (int r = load_attn(p + std::to_string(3 * l) + ".1", Id, false, &dec_self[l], nullptr, 8)) return r;
            if (int r = load_attn(p + std::to_string(3 * l + 1) + ".1", Id, true, &dec_cross[l], &kv_concat, 8)) return r;
            if (int r = load_mlp(p + std::to_string(3 * l + 2) + ".1", Fd, &dec_mlp[l], 8)) return r;
        }
        if (int r = upload_T(&wckv, kv_concat)) return r;
        if (!(t = get("decoder.net.norm.weight", {D}))) return TXO_E_STATE;
        if (int r = upload_f32(&decn_g, t->data)) return r;
        if (!(t = get("decoder.net.norm.bias", {D}))) return TXO_E_STATE;
        if (int r = upload_f32(&decn_b, t->data)) return r;
        if (!(t = get("decoder.net.to_logits.weight", {V, D}))) return TXO_E_STATE;
        if (int r = upload_T(&wlog, t->data)) return r;
        if (!(t = get("decoder.net.to_logits.bias", {V}))) return TXO_E_STATE;
        if (int r = upload_f32(&blog, t->data)) return r;
        host.clear();
        if (w_tiled_on) {    // tiled copies of every weight the decode-step projections read (dec_gemm.h: w_tiled)
            HIP_TRY(hipDeviceSynchronize());
            const int H = c.dec_heads;
            for (int l = 0; l < c.dec_layers; ++l) {
                want_tiled(dec_self[l].wqkv, 3 * Id, D); want_tiled(dec_self[l].wo, 2 * D, Id);
                want_tiled(dec_self[l].wqp, H * D, D); want_tiled(dec_self[l].wo_f, 2 * D, H * D);
                want_tiled(dec_cross[l].wq, Id, D); want_tiled(dec_cross[l].wo, 2 * D, Id);
                want_tiled(dec_cross[l].wqp, H * D, D); want_tiled(dec_cross[l].wo_f, 2 * D, H * D);
                want_tiled(dec_mlp[l].w1, 2 * Fd, D); want_tiled(dec_mlp[l].w2, D, Fd);
            }
            want_tiled(wlog, V, D);
            if (int r = make_tiled_all()) return r;
        }
        ready = true;
        return 0;
    }

    int init() {
        knobs.read();
        arena = Arena{}; arena.measuring = true;
        if (int r = init_buffers()) return r;                 // pass 1: sizes only
        if (int r = arena_begin(arena.off)) return r;
        if (int r = init_buffers()) return r;                 // pass 2: carve
        HIP_TRY(hipMemset(st, 0, sizeof(StepState) * MAXL));
        // the self-attention cache starts as zeros: clamped loads may touch rows no step has written yet (fused self-attention
        // at t = 0 multiplies such a row by p = 0, which must not meet NaN/Inf bit patterns of recycled memory)
        HIP_TRY(hipMemset(skv, 0, sizeof(T) * (size_t)cfg.dec_layers * 2 * Bmax * Id * Tmax));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&flags_host), sizeof(int) * MAXL * Tmax, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pctl_host), sizeof(PersistCtl), hipHostMallocDefault));
        max_lanes = MAXL;
        for (int i = 1; i < max_lanes; ++i) HIP_TRY(hipStreamCreateWithFlags(&lanes[i].own, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        for (int i = 0; i < MAXL; ++i) HIP_TRY(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
        for (int i = 0; i < MAXL; ++i) HIP_TRY(hipEventCreateWithFlags(&ev_flags[i], hipEventDisableTiming));
        for (int i = 0; i < MAXL; ++i) HIP_TRY(hipEventCreateWithFlags(&ev_live[i], hipEventDisableTiming));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&live_host), sizeof(int) * MAXL, hipHostMallocDefault));
        return 0;
    }
    int init_buffers() {
        const txo_config& c = cfg;
        D = c.embed_dim; Ie = c.enc_heads * DH; Id = c.dec_heads * DH; Fe = c.enc_exp * D; Fd = c.dec_exp * D;
        V = c.vocab; Tmax = c.max_len; Bmax = c.max_batch;
        hybrid = c.embed == TXO_EMBED_HYBRID;
        {
            int dev = 0; hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n_cus = prop.multiProcessorCount;
        }
        {
            bool exists = (D == 64 && la_supported<T, 64>()) || (D == 256 && la_supported<T, 256>()) || (D == 768 && la_supported<T, 768>());
            // the tile needs up to 145 KB of dynamic LDS: the opt-in is asked for HERE, and a refusal switches the latent form off for this
            // engine (every decode then takes the K/V form) instead of surfacing as a failed launch in the middle of a generate
            auto opt_in = [&](const void* kern, size_t lds) {
                if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); exists = false; }
            };
            if (exists) {
                if constexpr (la_supported<T, 64>()) { if (D == 64) opt_in(reinterpret_cast<const void*>(&lat_core_kernel<T, 64>), la_lds_bytes<T, 64>()); }
                if constexpr (la_supported<T, 256>()) { if (D == 256) opt_in(reinterpret_cast<const void*>(&lat_core_kernel<T, 256>), la_lds_bytes<T, 256>()); }
                if constexpr (la_supported<T, 256>() && sizeof(T) == 2) { if (D == 256) opt_in(reinterpret_cast<const void*>(&lat_core_kernel<T, 256, 4>), la_lds_bytes<T, 256, 4>()); }
                if constexpr (la_supported<T, 768>()) { if (D == 768) opt_in(reinterpret_cast<const void*>(&lat_core_kernel<T, 768>), la_lds_bytes<T, 768>()); }
            }
            latent_ok = exists;
        }
        Nmax = c.max_tokens > 0 ? c.max_tokens : 1 + (c.canvas_h / 16) * (c.canvas_w / 16);
        const size_t M = (size_t)Bmax * Nmax;
        const int Imax = Ie > Id ? Ie : Id, Fmax = Fe > Fd ? Fe : Fd;
        if (hybrid) {
            // largest NHWC activation per image: stem output (H/2 x W/2 x 64) = stage-0 output (H/4 x W/4 x 256) = 4096 per token
            const size_t E = (size_t)Bmax * (Nmax - 1) * 4096;
            if (bk_fp32) {
                for (auto& a : bk32.act) if (int r = dalloc(&a, E)) return r;
                // GroupNorm partial sums per 128-row output tile of the split convolutions: the stem's output is the largest (H/2 x W/2 pixels)
                const size_t rows = (size_t)Bmax * ((c.canvas_h + 1) / 2) * ((c.canvas_w + 1) / 2);
                if (int r = dalloc(&gn_tiles, (rows / GB_BM + 2) * 2 * 2 * 32 * 2)) return r;
            }
            else { for (auto& a : bk.act) if (int r = dalloc(&a, E)) return r; }
            if (int r = dalloc(&gn_partial, (size_t)Bmax * 64 * 64)) return r;
            if (int r = dalloc(&gn_stats, (size_t)Bmax * 64)) return r;
        }
        if (int r = dalloc(&ex, M * D)) return r;
        if (int r = dalloc(&estats, M * 2)) return r;
        if (int r = dalloc(&ey, M * D)) return r;
        if (int r = dalloc(&ez, M * D)) return r;
        if (int r = dalloc(&eenc, M * D)) return r;
        if (int r = dalloc(&enc_t, M * D)) return r;
        // (the prefill of the decoder runs in the encoder's workspace: sized for the wider of the two stacks)
        if (int r = dalloc(&eqkv, 3 * M * Imax)) return r;
        if (int r = dalloc(&eao, M * Imax)) return r;
        if (int r = dalloc(&ehid, M * Fmax)) return r;
        if (int r = dalloc(&ckv, (size_t)c.dec_layers * 2 * M * Id)) return r;
        if (int r = dalloc(&skv, (size_t)c.dec_layers * 2 * Bmax * Id * Tmax)) return r;
        if (int r = dalloc(&dx, (size_t)Bmax * D)) return r;
        if (int r = dalloc(&dy, (size_t)Bmax * D)) return r;
        if (int r = dalloc(&dq, (size_t)Bmax * Imax)) return r;
        if (int r = dalloc(&dao, (size_t)Bmax * Imax)) return r;
        if (int r = dalloc(&dhid, (size_t)Bmax * Fmax)) return r;
        if (int r = dalloc(&dz, (size_t)Bmax * D)) return r;
        if (int r = dalloc(&dqt, (size_t)Bmax * Imax)) return r;
        if (lat_self_env != 0) { if (int r = dalloc(&zc, (size_t)c.dec_layers * Bmax * Tmax * D)) return r; }   // z history of the opt-in latent self attention
        if (int r = dalloc(&dqp, (size_t)Bmax * c.dec_heads * D)) return r;
        // (+ tile alignment of the row ranges' regions: range li starts at ceil16(b0) + 16 li and spans ceil16(nb) rows -- beam ranges are not
        // multiples of 16 -- so the last of MAXL ranges can end at Bmax + 30 + 16 (MAXL - 1))
        if (int r = dalloc(&dcl, ((size_t)Bmax + 16 * MAXL + 32) * c.dec_heads * D)) return r;
        if (int r = dalloc(&dlogits, (size_t)Bmax * V)) return r;
        if (int r = dalloc(&cur_tok, (size_t)Bmax)) return r;
        if (int r = dalloc(&eos_seen, (size_t)Bmax)) return r;
        if (int r = dalloc(&row_map, (size_t)Bmax)) return r;
        if (int r = dalloc(&row_map2, (size_t)Bmax)) return r;
        if (int r = dalloc(&cur_tok2, (size_t)Bmax)) return r;
        if (int r = dalloc(&cmoves, (size_t)2 * Bmax)) return r;
        if (int r = dalloc(&cinfo, (size_t)2 * MAXL)) return r;
        if (int r = dalloc(&kmask, (size_t)Bmax * Tmax)) return r;
        if (int r = dalloc(&done_flag, (size_t)Tmax * MAXL)) return r;
        if (int r = dalloc(&st, MAXL)) return r;
        if (int r = dalloc(&tok_buf, (size_t)Bmax * Tmax)) return r;
        if (int r = dalloc(&bscore, (size_t)Bmax)) return r;
        if (int r = dalloc(&bfin, (size_t)Bmax)) return r;
        if (int r = dalloc(&bpath[0], (size_t)Bmax * Tmax)) return r;
        if (int r = dalloc(&bpath[1], (size_t)Bmax * Tmax)) return r;
        if (int r = dalloc(&bparent, (size_t)Bmax * Tmax)) return r;
        if (int r = dalloc(&btok, (size_t)Bmax * Tmax)) return r;
        if (int r = dalloc(&pctl, 1)) return r;
        if (int r = dalloc(&pstamps, (size_t)PS_TEAMS * PS_STAMP_RANKS * PS_MAX_STAGES * PS_STAMP_WORDS)) return r;
        return 0;
    }

    // ------------------------------------------------------------------------------------------
    template <int MODE, typename TZ>
    void launch_ln(hipStream_t s, const float* in, float* x_out, TZ* z_out, const float* g, const float* b, int rows, int rev = 0) {
        const dim3 grid((rows + 3) / 4), blk(256);
        if (D == 256) hipLaunchKernelGGL((ln_rows_kernel<TZ, 1, MODE>), grid, blk, 0, s, in, x_out, z_out, g, b, rows, rev);
        else if (D == 512) hipLaunchKernelGGL((ln_rows_kernel<TZ, 2, MODE>), grid, blk, 0, s, in, x_out, z_out, g, b, rows, rev);
        else if (D == 768) hipLaunchKernelGGL((ln_rows_kernel<TZ, 3, MODE>), grid, blk, 0, s, in, x_out, z_out, g, b, rows, rev);
        else hipLaunchKernelGGL((ln_rows_generic_kernel<TZ, MODE>), grid, blk, 0, s, in, x_out, z_out, g, b, rows, D);
    }

    // C = A * W^T with A plain row-major [M][K]: the 256x256 LDS-DMA kernel (gemm_pp.h) for bf16 shapes it fits, else the
    // 128x128 register-staged kernel.  TXO_GEMM_OLD=1 forces the latter (tests compare the two bit for bit).
    template <class Epi>
    void gemm_plain(hipStream_t s, const T* A, const T* W, int M, int N, int K, Epi epi, int rev = 0) {
        if constexpr (sizeof(T) == 2) {
            if (use_pp && gemm_pp_fits(M, N, K)) { launch_gemm_pp(s, A, W, M, N, K, epi, pp_tr, rev, pp_sb_mb, pp_ct); return; }
        }
        launch_gemm_big<T>(s, LoadPlain<T>{A, K}, W, M, N, K, epi);
    }

    // ---- hybrid embedder: ResNetV2 [2,4,6] on NHWC activations (conv.h) -----------------------------------------
    static void same_pad(int in, int k, int stride, int* out, int* pad_lo) {   // utils.py:97-99,112-123 (TF "SAME")
        *out = (in + stride - 1) / stride;
        const int pad = std::max((*out - 1) * stride + (k - 1) + 1 - in, 0);
        *pad_lo = pad / 2;
    }
    template <typename TB>
    void conv(hipStream_t s, const TB* in, const TB* w, TB* out, int B, int H, int W, int C, int OC, int k, int stride) {
        int OH, OW, pt, pl;
        same_pad(H, k, stride, &OH, &pt); same_pad(W, k, stride, &OW, &pl);
        // a 1x1 stride-1 convolution on NHWC activations IS a plain row-major GEMM: [B*H*W][C] x [OC][C]^T -> (bf16 backbone) the 256x256
        // LDS-DMA kernel where the shape fits it (the bottlenecks' expanding convolutions, 256 / 512 / 1024 output channels)
        if (k == 1 && stride == 1) {
            if constexpr (std::is_same<TB, T>::value) gemm_plain(s, in, w, B * H * W, OC, C, EpiStore<T>{out, OC, nullptr});
            else bk_gemm(s, LoadPlain<TB>{in, C}, w, B * H * W, OC, C, EpiStore<TB>{out, OC, nullptr}, H * W);
            return;
        }
        LoadConv<TB> ld{in, H, W, C, stride, pt, pl, FastDiv(OH * OW), FastDiv(OW), FastDiv(C), FastDiv(k)};
        bk_gemm(s, ld, w, B * OH * OW, OC, k * k * C, EpiStore<TB>{out, OC, nullptr}, OH * OW);
    }
    // a backbone GEMM in storage type TB.  fp32 backbone INSIDE the bf16 engine: fp32 operands split onto the bf16 matrix pipe
    // (gemm_split.h: ~2^-16 per product, 5x less matrix time than exact-f32 MFMA); everything else -- the fp32 parity engine first of all --
    // the exact kernel.  TXO_BACKBONE_EXACT=1 keeps the exact-f32 kernel in the bf16 engine too (A/B, tests).
    bool bk_exact = getenv("TXO_BACKBONE_EXACT") != nullptr;
    // gn_hw > 0: the output is the input of a GroupNorm over images of gn_hw pixels -- the split kernel then leaves the norm's partial sums
    // behind (gemm_split.h: GnPart) and group_norm() skips its own pass over the tensor (gn_fused)
    bool gn_fused = false;
    float* gn_tiles = nullptr;        // [row tiles][2 halves][2 images][32 groups][2]
    template <typename TB, class ALoad, class Epi>
    void bk_gemm(hipStream_t s, ALoad ld, const TB* w, int M, int N, int K, Epi epi, int gn_hw = 0) {
        gn_fused = false;
        if constexpr (sizeof(TB) == 4 && sizeof(T) == 2) {
            if (!bk_exact && gemm_split_fits(K)) {
                GnPart gp{nullptr, 1, 1};
                if (gn_hw > 0 && gn_tiles && gn_fusable(gn_hw, N)) { gp = GnPart{gn_tiles, gn_hw, N / 32}; gn_fused = true; }
                launch_gemm_split(s, ld, w, M, N, K, epi, gp);
                return;
            }
        }
        launch_gemm_big<TB>(s, ld, w, M, N, K, epi);
    }
    template <bool RELU, bool RES, typename TB>
    void group_norm(hipStream_t s, const TB* x, const TB* res, TB* y, const GnW& g, int B, int HW, int C) {
        if (gn_fused) {                                      // the convolution that wrote x left the partial sums per output tile (bk_gemm)
            gn_fused = false;
            hipLaunchKernelGGL(gn_finish_tiles_kernel, dim3(B), dim3(32), 0, s, gn_tiles, gn_stats, HW, (double)HW * (C / 32));
        } else {
            const int chunk_px = std::max(256, (HW + 63) / 64), nchunk = (HW + chunk_px - 1) / chunk_px;
            hipLaunchKernelGGL((gn_partial_kernel<TB>), dim3(nchunk, B), dim3(256), 0, s, x, gn_partial, HW, C, chunk_px);
            hipLaunchKernelGGL(gn_finish_kernel, dim3(B), dim3(32), 0, s, gn_partial, gn_stats, nchunk, (double)HW * (C / 32));
        }
        const size_t nvec = (size_t)B * HW * C / Elem<TB>::PER16;
        hipLaunchKernelGGL((gn_apply_kernel<TB, RELU, RES>), dim3((nvec + 255) / 256), dim3(256), 0, s, x, res, y, gn_stats, g.g,
                           g.b, HW, C, nvec);
    }
    template <typename TB>
    int backbone(Backbone<TB>& k, const float* img, int B, int H, int W, const TB** feat, hipStream_t s) {
        TB* const* act = k.act;
        int h1, w1, pt, pl;
        same_pad(H, 7, 2, &h1, &pt); same_pad(W, 7, 2, &w1, &pl);
        bk_gemm(s, LoadStem<TB>{img, H, W, pt, pl, FastDiv(h1 * w1), FastDiv(w1)}, k.stem_w, B * h1 * w1, 64, 64, EpiStore<TB>{act[1], 64, nullptr}, h1 * w1);
        group_norm<true, false>(s, (const TB*)act[1], (const TB*)nullptr, act[1], k.stem_gn, B, h1 * w1, 64);
        int hc, wc, ppt, ppl;
        same_pad(h1, 3, 2, &hc, &ppt); same_pad(w1, 3, 2, &wc, &ppl);
        {
            const size_t nvec = (size_t)B * hc * wc * 64 / Elem<TB>::PER16;
            hipLaunchKernelGGL((maxpool3x3s2_kernel<TB>), dim3((nvec + 255) / 256), dim3(256), 0, s, act[1], act[0], h1, w1, 64,
                               hc, wc, ppt, ppl, nvec);
        }
        TB* cur = act[0];
        for (const BlockW<TB>& b : k.blocks) {                // Bottleneck.forward (resnet.py:143-149)
            const int ho = (hc + b.stride - 1) / b.stride, wo = (wc + b.stride - 1) / b.stride;
            const TB* res = cur;
            if (b.has_ds) {                                   // DownSample: 1x1 stride-s StdConv + GroupNorm (no act)
                conv<TB>(s, cur, b.ds, act[1], B, hc, wc, b.cin, b.cout, 1, b.stride);
                group_norm<false, false>(s, (const TB*)act[1], (const TB*)nullptr, act[1], b.nds, B, ho * wo, b.cout);
                res = act[1];
            }
            conv<TB>(s, cur, b.c1, act[2], B, hc, wc, b.cin, b.mid, 1, 1);
            group_norm<true, false>(s, (const TB*)act[2], (const TB*)nullptr, act[2], b.n1, B, hc * wc, b.mid);
            conv<TB>(s, act[2], b.c2, act[3], B, hc, wc, b.mid, b.mid, 3, b.stride);
            group_norm<true, false>(s, (const TB*)act[3], (const TB*)nullptr, act[3], b.n2, B, ho * wo, b.mid);
            conv<TB>(s, act[3], b.c3, act[2], B, ho, wo, b.mid, b.cout, 1, 1);
            group_norm<true, true>(s, (const TB*)act[2], res, act[0], b.n3, B, ho * wo, b.cout);   // relu(norm(x) + res)
            cur = act[0]; hc = ho; wc = wo;
        }
        if (hc != H / 16 || wc != W / 16) return fail(TXO_E_INVALID, "backbone output grid does not match H/16 x W/16");
        *feat = cur;
        HIP_TRY(hipGetLastError());
        return 0;
    }

    int encode(const float* img, int B, int C, int H, int W, float* enc_out, hipStream_t s) override {
        if (!ready) return fail(TXO_E_STATE, "weights not finalized");
        if (C != cfg.in_channels) return fail(TXO_E_INVALID, "image channel count does not match in_channels");
        if (H <= 0 || W <= 0 || H % 16 || W % 16) return fail(TXO_E_INVALID, "image height/width must be positive multiples of 16");
        if (H > cfg.canvas_h || W > cfg.canvas_w) return fail(TXO_E_INVALID, "image larger than the position-embedding canvas");
        const int h = H / 16, w = W / 16, hw = h * w, N = hw + 1;
        if (B < 1 || B > Bmax) return fail(TXO_E_INVALID, "batch exceeds engine max_batch");
        if (N > Nmax) return fail(TXO_E_INVALID, "token count exceeds engine max_tokens");
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (prof) { e0 = pool.next(); e1 = pool.next(); (void)hipEventRecord(e0, s); }
        // Image chunks (enc_chunk_images): every row of the stack belongs to ONE image, so the stack may run over a few images at a time
        // through the SAME workspace rows -- a chunk whose intermediates (stream, z, q/k/v, attention output, FFN hidden) fit the 256 MB
        // Infinity Cache keeps every producer -> consumer hand-over on the die instead of through HBM.  Same bits as the whole batch.
        const int bc = enc_chunk_images(B, N);
        for (int b0 = 0; b0 < B; b0 += bc) {
            const int nb = std::min(bc, B - b0);
            if (int r = encode_images(img + (size_t)b0 * C * H * W, nb, C, H, W, enc_out + (size_t)b0 * N * D, s)) return r;
        }
        if (prof) { (void)hipEventRecord(e1, s); ev_enc.push_back({e0, e1}); }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // images per encoder chunk: TXO_ENC_CHUNK=n forces n (0 = whole batch); default by measured working set (profiles/r06_encoder_chunk_sweep.txt)
    int enc_chunk_env = getenv("TXO_ENC_CHUNK") ? atoi(getenv("TXO_ENC_CHUNK")) : -1;
    int enc_chunk_images(int B, int N) const {
        if (enc_chunk_env == 0) return B;
        if (enc_chunk_env > 0) return std::min(B, enc_chunk_env);
        (void)N;
        return B;
    }
    int encode_images(const float* img, int B, int C, int H, int W, float* enc_out, hipStream_t s) {
        const int h = H / 16, w = W / 16, hw = h * w, N = hw + 1;
        const int M = B * N, G = cfg.canvas_w / 16;

        hipLaunchKernelGGL(cls_rows_kernel, dim3((B * D + 255) / 256), dim3(256), 0, s, ex, cls, pos, B, N, D);
        if (!hybrid) {
            launch_gemm_big<T>(s, LoadPatch<T>{img, C, H, W, hw, w}, patch_w, B * hw, D, C * 256,
                               EpiPatch{ex, patch_b, pos, D, hw, w, G});
        } else {
            if (bk_fp32) {
                const float* feat = nullptr;
                if (int r = backbone<float>(bk32, img, B, H, W, &feat, s)) return r;
                bk_gemm(s, LoadPlain<float>{feat, 1024}, (const float*)bk32.proj_w, B * hw, D, 1024, EpiTokens{ex, patch_b, pos, D, hw, w, G});
            } else {
                const T* feat = nullptr;
                if (int r = backbone<T>(bk, img, B, H, W, &feat, s)) return r;
                gemm_plain(s, feat, patch_w, B * hw, D, 1024, EpiTokens{ex, patch_b, pos, D, hw, w, G});
            }
        }
        const size_t hs = (size_t)M * Ie;      // one of q/k/v, head-major [B*heads][N][64]
        // Walk direction (perf mode, enc_walk): successive kernels of the stack walk the rows alternately upwards and downwards, so that each
        // starts on the rows its producer wrote LAST -- still in the 256 MB Infinity Cache -- instead of on the ones it wrote first
        // (every intermediate here is 0.2-0.9 GB at batch 256: with all kernels walking upwards a consumer's first reads are the
        // producer's oldest lines).  Results do not depend on it (every kernel's rows are independent).
        bool down = false;                                    // direction of the NEXT kernel
        auto dir = [&]() -> int { if (!enc_walk || sizeof(T) != 2) return 0; const bool d = down; down = !down; return d ? 1 : 0; };
        for (int l = 0; l < cfg.enc_layers; ++l) {
            // The stream between two sub-layers is x = LN(y) (the residual) and z = LN(x) (the block input), attention.py:242-259.
            // x is never written: the row kernel leaves {mean, rstd} of LN(y) per row (rows.h MODE 3) and the next GEMM epilogue
            // rebuilds its residual from y -- in place, ey is both its residual source and its output -- with the same expression.
            const ResidLN res_x{ey, estats, enc_gb, D}, res_first{ex, nullptr, nullptr, D};
            // outputs far beyond the caches are written non-temporally (they would evict the GEMM's own operand panels from L2)
            const int nt_y = enc_nt((size_t)M * D * 4), nt_qkv = enc_nt((size_t)3 * M * Ie * sizeof(T)), nt_h = enc_nt((size_t)M * Fe * sizeof(T));
            if (l == 0) launch_ln<0, T>(s, ex, nullptr, ez, enc_g, enc_b, M, dir());
            else launch_ln<3, T>(s, ey, estats, ez, enc_g, enc_b, M, dir());
            const dim3 agrid = ea_grid((N + EA_QBLK - 1) / EA_QBLK, B * cfg.enc_heads);   // XCD-aware block order (enc_attn.h: ea_block)
            const int nbh = B * cfg.enc_heads;
            if constexpr (sizeof(T) == 4) {
                gemm_plain(s, ez, enc_attn[l].wqkv, M, 3 * Ie, D,
                                   EpiHeads<float>{eqkv, hs, Ie, cfg.enc_heads, N, nt_qkv});
                hipLaunchKernelGGL((enc_attn_kernel<T>), agrid, dim3(256), 0, s, eqkv, eqkv + hs, eqkv + 2 * hs, eao, N,
                                   cfg.enc_heads, nbh);
            } else {                                              // perf mode: bf16 q/k/v, bf16 MFMA attention
                bf16* qb = reinterpret_cast<bf16*>(eqkv);
                gemm_plain(s, ez, enc_attn[l].wqkv, M, 3 * Ie, D,
                                   EpiHeads<bf16>{qb, hs, Ie, cfg.enc_heads, N, nt_qkv}, dir());
                hipLaunchKernelGGL((enc_attn_bf16_v2_kernel<T>), agrid, dim3(256), 0, s, qb, qb + hs, qb + 2 * hs, eao, N, cfg.enc_heads, nbh, dir());
            }
            gemm_plain(s, eao, enc_attn[l].wo, M, 2 * D, Ie,
                               EpiGluRes<sizeof(T) == 2>{ey, l == 0 ? res_first : res_x, enc_attn[l].bo, nt_y}, dir());
            launch_ln<3, T>(s, ey, estats, ez, enc_g, enc_b, M, dir());
            gemm_plain(s, ez, enc_mlp[l].w1, M, 2 * Fe, D, EpiGeglu<T>{ehid, enc_mlp[l].b1, Fe, nt_h}, dir());
            gemm_plain(s, ehid, enc_mlp[l].w2, M, D, Fe, EpiBiasRes{ey, res_x, enc_mlp[l].b2, nt_y}, dir());
        }
        launch_ln<2, float>(s, ey, nullptr, enc_out, encn_g, encn_b, M, dir());
        return 0;
    }

    // project_kv = false: generate() decides the form of the cross attention once it knows its decode path
    int decode_begin(const float* enc, int B, int N, int eos, hipStream_t s) override { return begin_session(enc, B, N, eos, s, true); }
    int begin_session(const float* enc, int B, int N, int eos, hipStream_t s, bool project_kv) {
        if (!ready) return fail(TXO_E_STATE, "weights not finalized");
        if (B < 1 || B > Bmax) return fail(TXO_E_INVALID, "batch exceeds engine max_batch");
        if (N < 1 || N > Nmax) return fail(TXO_E_INVALID, "token count exceeds engine max_tokens");
        const int M = B * N;
        {   // the session's copy of the encoder rows in the storage type: A operand of the K/V projection; in the latent form what
            // every cross-attention tile of every layer reads (the caller's buffer is not referenced after this call)
            const size_t n4 = (size_t)M * D / 4;
            hipLaunchKernelGGL((cast_rows_kernel<T>), dim3((n4 + 255) / 256), dim3(256), 0, s, enc, enc_t, n4);
        }
        sB = B; sN = N; sImg = B; session = true;
        ckv_valid = false;
        kmask_on = false;
        // a session opened through the C entry point steps with launches: it takes the form generate()'s launches take at this batch size
        // A session opened through txo_decode_begin (project_kv) may be prefilled, and the multi-position forward works on the projected
        // K/V panels: such a session steps in the K/V form too (one-pass and stepwise logits of decoder.net() then come from the same
        // weights), unless TXO_LATENT=1 pins the latent form.  generate() / generate_beam() choose their form themselves.
        use_latent = latent_ok && (lat_mode == 1 || (lat_mode < 0 && !project_kv && auto_latent(B)));
        lat_self = false;
        if (!use_latent && project_kv) ensure_ckv(s);
        set_lanes(1, s);
        reset_lanes(s, eos);
        HIP_TRY(hipGetLastError());
        return 0;
    }

    // K/V of every decoder layer's cross attention in one GEMM: W = [Ld][k|v][Id][D] (attention.py:125-126).  The K/V form's decode
    // steps and the multi-position prefill read these panels; a latent-form session projects them only if a prefill asks.
    void ensure_ckv(hipStream_t s) {
        if (ckv_valid) return;
        const int M = sImg * sN;
        gemm_plain(s, enc_t, wckv, M, cfg.dec_layers * 2 * Id, D, EpiHeads<T>{ckv, (size_t)M * Id, Id, cfg.dec_heads, sN});
        ckv_valid = true;
    }

    // Error exit of a decode loop that forked onto the lanes' own streams: kernels of the failed call may still be running there on this
    // engine's buffers.  The caller's stream is made to wait for every lane, drained, and the engine goes back to one lane; the error
    // code passes through.
    int abandon_lanes(hipStream_t s, int rc) {
        for (int i = 0; i < n_lanes; ++i) {
            if (lanes[i].stream == s) continue;
            if (hipEventRecord(ev_join[i], lanes[i].stream) == hipSuccess) (void)hipStreamWaitEvent(s, ev_join[i], 0);
            else (void)hipStreamSynchronize(lanes[i].stream);
        }
        (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
        set_lanes(1, s);
        stamp_slot = -1;
        return rc;
    }
    // Pick the two streams of a two-range decode: NCAND candidate streams, every pair timed on two chains of 64 launches that hold one wave per
    // CU for 4 us each behind a gate.  Pairs that share a hardware queue take twice as long (0.68 against 0.34 ms: nothing in between) -- a decode
    // on such a pair runs its ranges one after the other (110 ms per generate at batch 256 instead of 66) -- and the first pair that runs side
    // by side becomes lanes[0].own / lanes[1].own.  ~15 ms, once per engine.  (What it does NOT remove: among pairs that do run side by side a
    // generate still takes 66-78 ms by PROCESS, whatever the pair -- profiles/r06_b256_stream_pairs.txt; a trial of real decode positions
    // per pair was built and predicts nothing.)
    int tune_lane_streams() {
        lanes_tuned = true;
        constexpr int NCAND = 5, CHAIN = 64;
        const bool verbose = getenv("TXO_TUNE_LANES_VERBOSE") != nullptr;
        hipStream_t cand[NCAND] = {};
        for (auto& c : cand) HIP_TRY(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&ea)); HIP_TRY(hipEventCreate(&eb));
        auto run_pair = [&](hipStream_t a, hipStream_t b) -> double {
            double best = 1e30;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
                hipLaunchKernelGGL(hold_kernel, dim3(1), dim3(64), 0, a, 100000);          // the gate: the host enqueues both chains behind it
                (void)hipEventRecord(e0, a);
                (void)hipStreamWaitEvent(b, e0, 0);
                for (int i = 0; i < CHAIN; ++i) {
                    hipLaunchKernelGGL(hold_kernel, dim3(n_cus), dim3(64), 0, a, 400);
                    hipLaunchKernelGGL(hold_kernel, dim3(n_cus), dim3(64), 0, b, 400);
                }
                (void)hipEventRecord(ea, a); (void)hipEventRecord(eb, b);
                (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
                float ta = 0, tb = 0;
                (void)hipEventElapsedTime(&ta, e0, ea); (void)hipEventElapsedTime(&tb, e0, eb);
                best = std::min(best, (double)std::max(ta, tb));
            }
            return best;
        };
        (void)run_pair(cand[0], cand[1]);                                  // (code object load, clocks)
        double tp[NCAND][NCAND] = {};
        double bt = 1e30, wt = 0;
        for (int i = 0; i < NCAND; ++i)
            for (int j = i + 1; j < NCAND; ++j) { tp[i][j] = run_pair(cand[i], cand[j]); bt = std::min(bt, tp[i][j]); wt = std::max(wt, tp[i][j]); }
        int bi = 0, bj = 1; double best = 1e30;
        for (int i = 0; i < NCAND; ++i)
            for (int j = i + 1; j < NCAND; ++j) {
                if (tp[i][j] > 1.3 * bt) continue;                         // (one after the other)
                if (verbose) fprintf(stderr, "[txo] streams (%d, %d): side-by-side test %.2f ms\n", i, j, tp[i][j]);
                if (best > 1e29) { best = tp[i][j]; bi = i; bj = j; }      // the first pair that runs side by side
            }
        tune_best_ms = best; tune_worst_ms = wt;
        if (lanes[0].own) (void)hipStreamDestroy(lanes[0].own);
        if (lanes[1].own) (void)hipStreamDestroy(lanes[1].own);
        lanes[0].own = cand[bi]; lanes[1].own = cand[bj];
        for (int i = 0; i < NCAND; ++i) if (i != bi && i != bj) (void)hipStreamDestroy(cand[i]);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
        if (verbose) fprintf(stderr, "[txo] row-range streams: pair (%d, %d) of %d candidates (serialised pairs take %.2f ms)\n", bi, bj, NCAND, wt);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // split the batch into n contiguous row ranges (multiples of 16 rows where possible)
    void set_lanes(int n, hipStream_t s) {
        const int tiles = (sB + 15) / 16;
        if (n > tiles) n = tiles;
        if (n < 1) n = 1;
        n_lanes = n;
        int row = 0;
        for (int i = 0; i < n; ++i) {
            const int tl = tiles / n + (i < tiles % n ? 1 : 0);
            lanes[i].b0 = row;
            lanes[i].nb = std::min(sB - row, tl * 16);
            row += lanes[i].nb;
            lanes[i].stream = i == 0 ? s : lanes[i].own;
        }
        if (n >= 2 && lanes[0].own) lanes[0].stream = lanes[0].own;
    }
    void reset_lanes(hipStream_t s, int eos) {
        for (int i = 0; i < n_lanes; ++i) {
            const Lane& ln = lanes[i];
            const int n = ln.nb > Tmax ? ln.nb : Tmax;
            hipLaunchKernelGGL(reset_state_kernel, dim3((n + 255) / 256), dim3(256), 0, s, st + i, cur_tok + ln.b0,
                               eos_seen + ln.b0, done_flag + (size_t)i * Tmax, ln.nb, Tmax, cfg.bos, eos);
            if (row_stop) hipLaunchKernelGGL(iota_kernel, dim3((ln.nb + 255) / 256), dim3(256), 0, s, row_map + ln.b0, ln.nb, ln.b0);
        }
    }

    // tiled copies of the decode projections' weights for the launch path (dec_gemm.h: w_tiled), keyed by the row-major buffer
    std::map<const void*, T*> wtiled;
    int w_tiled_on = getenv("TXO_W_TILED") ? atoi(getenv("TXO_W_TILED")) : 1;     // TXO_W_TILED=0: every projection reads the row-major buffers (A/B)
    // one allocation for all of them (the arenas' reason: few large mappings); `want` collects (buffer, N, K), make_tiled_all() builds
    struct TileReq { const T* w; int N, K; };
    std::vector<TileReq> tile_reqs;
    void want_tiled(const T* w, int N, int K) {
        if (!w_tiled_on || !w || K % Elem<T>::KCHUNK) return;
        for (auto& r : tile_reqs) if (r.w == w) return;
        tile_reqs.push_back({w, N, K});
    }
    int make_tiled_all() {
        if (tile_reqs.empty()) return 0;
        auto pieces = [](const TileReq& r) { return (size_t)((r.N + 15) / 16) * (r.K / Elem<T>::KCHUNK) * 64; };   // 16-byte pieces
        size_t total = 0;
        for (auto& r : tile_reqs) total += (pieces(r) * 16 + 255) & ~(size_t)255;
        unsigned char* base = nullptr;
        HIP_TRY(hipMalloc(&base, total));
        allocs.push_back(base);
        size_t off = 0;
        for (auto& r : tile_reqs) {
            T* d = reinterpret_cast<T*>(base + off);
            hipLaunchKernelGGL((tile_weights_kernel<T>), dim3((unsigned)((pieces(r) + 255) / 256)), dim3(256), 0, 0, r.w, d, r.N, r.K);
            wtiled[r.w] = d;
            off += (pieces(r) * 16 + 255) & ~(size_t)255;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        tile_reqs.clear();
        return 0;
    }
    void use_tiled(DecGemmArgs<T>& a) const {
        a.w_tiled = 0;
        if (!w_tiled_on) return;
        auto it = wtiled.find(a.W);
        if (it != wtiled.end()) { a.W = it->second; a.w_tiled = 1; }
    }

    template <int PRO, int EPI>
    int launch_dec_gemm(hipStream_t s, DecGemmArgs<T> a) {
        use_tiled(a);
        constexpr bool has_pro = PRO != PRO_NONE;
        // 16-column tiles (half the cold weight bytes per block, twice the blocks) where the block's traffic is its weight
        // slice: the out-projections and FFN-out (K >= 1024).  Launches with an LN prologue keep 32 columns: every block also
        // fetches the 16 rows of y, so more blocks only multiply that traffic (measured: FFN-in 3.9 -> 7 us at 16 columns).
        constexpr bool paired = EPI == EPI_GLU_RES || EPI == EPI_GEGLU;
        const bool half = (paired && !has_pro) || (EPI == EPI_BIAS_RES && a.K >= 1024);
        const int bn = half ? 16 : DG_BN;
        const dim3 grid((a.N + bn - 1) / bn, (a.rows + DG_BM - 1) / DG_BM), blk(256);
        const size_t lds = dec_gemm_lds_bytes<T>(a.K, has_pro);
        a.stamps = (grid.x * grid.y <= (unsigned)STAMP_BLOCKS) ? next_stamp(PRO == PRO_NONE ? (EPI == EPI_GLU_RES ? "gemm out-proj+GLU+res" : "gemm ffn-out+res") : (EPI == EPI_QKV ? "gemm LN+qkv" : (EPI == EPI_GEGLU ? "gemm LN+ffn-in+GeGLU" : (EPI == EPI_STORE_T ? "gemm LN+q' (latent)" : "gemm LN+logits")))) : nullptr;
        // K known at compile time for the shapes of the reference configurations (straight-line code, exact register
        // arrays); any other K takes the run-time form (KW = 0)
        constexpr int KCH = Elem<T>::KCHUNK;
        const int kw = (a.K % (4 * KCH) == 0) ? a.K / (4 * KCH) : 0;
#define TXO_DG(KW_, BN_) hipLaunchKernelGGL((dec_gemm_kernel<T, PRO, EPI, KW_, BN_>), grid, blk, lds, s, a)
        if constexpr ((paired && !has_pro) || EPI == EPI_BIAS_RES) {
            if (half) {
                switch (kw) {
                    case 4: TXO_DG(4, 16); break;    case 6: TXO_DG(6, 16); break;    case 8: TXO_DG(8, 16); break;
                    case 12: TXO_DG(12, 16); break;  case 16: TXO_DG(16, 16); break;
                    case 24:                                 // K = 3072 (FFN-out of a 768-wide decoder): every fragment requested up front, one round trip instead of three
                        if constexpr (EPI == EPI_BIAS_RES && sizeof(T) == 2) { TXO_DG(24, 16); break; }
                        TXO_DG(0, 16); break;
                    default: TXO_DG(0, 16);
                }
            } else {
                TXO_DG(0, 32);
            }
        } else {
            switch (kw) {
                case 2: TXO_DG(2, 32); break;  case 4: TXO_DG(4, 32); break;  case 6: TXO_DG(6, 32); break;
                default: TXO_DG(0, 32);
            }
        }
#undef TXO_DG
        return 0;
    }

    // >= 128 rows of a wide (>= 512) bf16 decoder: the projections without a LayerNorm prologue on blocks of RT x 16 rows that share
    // their weight fragments (dec_gemm.h: RT).  Returns false when the shape has no such instantiation (the caller launches the
    // 16-row kernel).  A row's bits do not depend on RT (same K split over the waves, same reduction order).
    template <int EPI>
    bool launch_dec_gemm_wide(hipStream_t s, DecGemmArgs<T> a) {
        if constexpr (sizeof(T) != 2) { (void)s; (void)a; return false; }
        else {
            if (a.rows < 128 || dec_wide_off) return false;
            use_tiled(a);
            constexpr int KCH = Elem<T>::KCHUNK;
            const int kw = (a.K % (4 * KCH) == 0) ? a.K / (4 * KCH) : 0;
            // narrow decoder: only the folded latent output projection (K = heads * D = 2048) of a beam search's many rows -- 640 rows are
            // 1280 16-row blocks of 128 KB each (19.6 us on one range; on two ranges of 320 rows the 32-row blocks give 1066 -> 1130 img/s).
            // Up to 256 rows per range they change nothing (batch 256: 76.1 vs 76.4 ms on two ranges, 82.0 vs 81.5 on one).  r05: 32 x 16 blocks
            // for 129 .. 256 rows of ONE range (192 rows: 384 blocks of 16 x 16 are one and a half rounds of the CUs, 192 of 32 x 16 one):
            // generate 60.3 -> 59.6 ms at 130 images, 64.7 -> 62.9 at 160, 68.6 -> 66.6 at 192; two ranges of <= 128 rows keep 16 x 16
            // (75.5 ms at 256 against 79.0), profiles/r05_folded_projection_blocks.txt.  A row's bits do not depend on the block.
            if (D < 512 && !(kw == 16 && a.rows >= std::min(wide_min_rows, wide_mid_rows) && EPI == EPI_GLU_RES)) return false;
            a.stamps = nullptr;
#define TXO_DGW(KW_, BN_, RT_)                                                                                              \
            do {                                                                                                            \
                const dim3 grid((a.N + BN_ - 1) / BN_, (a.rows + DG_BM * RT_ - 1) / (DG_BM * RT_)), blk(256);                 \
                hipLaunchKernelGGL((dec_gemm_wide_kernel<T, EPI, KW_, BN_, RT_>), grid, blk, (size_t)RT_ * 8192, s, a); \
                return true;                                                                                                \
            } while (0)
            if (kw == 6) TXO_DGW(6, 32, 4);                  // K = 768: the gated out-projections, FFN-in behind its LayerNorm launch
            if constexpr (EPI == EPI_GLU_RES) {
                if (kw == 16) { if (a.rows < wide_min_rows) TXO_DGW(16, 16, 2); TXO_DGW(16, 32, 2); }   // (at the beam shape, 320 rows per range: 16 columns and / or 64 rows per block 1080-1084 against 1134 img/s)
            }
#undef TXO_DGW
            return false;
        }
    }

    // attention front half: (optional) row prologue + q / qkv projection + cached single-query attention
    struct AttnOpt {
        bool cross = false; int apro = APRO_LN2;
        const T* W = nullptr; T* K = nullptr; T* V = nullptr; int lmax = 0, len = 0;
        float* x_out = nullptr;
        int kv_div = 1; const short* path = nullptr;      // beam search: shared cross K/V, scattered self history
    };
    void launch_dec_attn(hipStream_t s, int li, const AttnOpt& o) {
        const Lane& ln = lanes[li];
        const size_t r0 = ln.b0;
        DecAttnArgs<T> a{};
        a.y = dy + r0 * D; a.tok = cur_tok + r0; a.tok_emb = tok_emb; a.pos_emb = pos_emb; a.x_out = o.x_out;
        a.gamma = dec_g; a.beta = dec_b; a.D = D; a.W = o.W;
        const size_t kr0 = o.cross ? r0 / o.kv_div : r0;      // cross panels are per IMAGE (kv_div beams share one)
        a.K = o.K + kr0 * cfg.dec_heads * o.lmax * DH; a.V = o.V + kr0 * cfg.dec_heads * o.lmax * DH;
        a.out = dao + r0 * Id; a.heads = cfg.dec_heads; a.lmax = o.lmax; a.len = o.len; a.t_ptr = &st[li].t; a.t_host = step_host_t;
        a.qin = dq + r0 * Id; a.kv_div = o.kv_div; a.path = o.path ? o.path + r0 * Tmax : nullptr; a.path_stride = Tmax;   // slots are range-local
        a.kmask = kmask + r0 * Tmax; a.kmask_stride = Tmax;
        a.stamps = (ln.nb * cfg.dec_heads <= STAMP_BLOCKS) ? next_stamp(o.cross ? "attn cross" : "attn self") : nullptr;
        const dim3 grid(ln.nb * cfg.dec_heads), blk(256);
        constexpr int NLS = sizeof(T) == 2 ? 8 : 16;       // self: 256 cached keys per pass
        constexpr int WBS = sizeof(T) == 2 ? 3 : 1;        // self: q,k,v weight rows requested together (bf16) or one by one
        // profiling: hipExtLaunchKernelGGL binds the start/stop events to the dispatch itself (the kernel's own begin /
        // end timestamps, what rocprofv3 reports), not to marker commands around it
        hipEvent_t e0 = nullptr, e1 = nullptr;
        // light mode never creates events inside a timed region: it records until the pre-created pool is used up
        // and instruments every fourth cross-attention launch (an event-carrying launch costs ~2 us of wall time)
        const bool timed = o.cross && (prof || (prof_cross && (cross_seq++ & 3) == 0 && pool.used + 2 <= pool.ev.size()));
        if (timed) { e0 = pool.next(); e1 = pool.next(); }
        const bool narrow = (D & 255) != 0;
#define TXO_DA1(MODE, APRO, NLV, WBV, NARROW)                                                                         \
        do {                                                                                                          \
            if (timed) hipExtLaunchKernelGGL((dec_attn_kernel<T, MODE, APRO, NLV, WBV, NARROW>), grid, blk, 0, s, e0, e1, 0, a); \
            else hipLaunchKernelGGL((dec_attn_kernel<T, MODE, APRO, NLV, WBV, NARROW>), grid, blk, 0, s, a);           \
        } while (0)
#define TXO_DA(MODE, APRO, NLV, WBV)                                                                          \
        do { if (narrow) TXO_DA1(MODE, APRO, NLV, WBV, true); else TXO_DA1(MODE, APRO, NLV, WBV, false); } while (0)
        if (o.cross) TXO_DA(ATT_CROSS, APRO_LN2, DA_NL_CROSS, 1);
        else if (o.apro == APRO_NONE && o.path) {
            if (narrow) hipLaunchKernelGGL((dec_attn_kernel<T, ATT_SELF, APRO_NONE, NLS, 1, true, true>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((dec_attn_kernel<T, ATT_SELF, APRO_NONE, NLS, 1, false, true>), grid, blk, 0, s, a);
        }
        else if (o.apro == APRO_NONE && kmask_on) {               // padding mask over the decoded positions (txo_decode_set_key_mask)
            if (narrow) hipLaunchKernelGGL((dec_attn_kernel<T, ATT_SELF, APRO_NONE, NLS, 1, true, false, true>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((dec_attn_kernel<T, ATT_SELF, APRO_NONE, NLS, 1, false, false, true>), grid, blk, 0, s, a);
        }
        else if (o.apro == APRO_NONE) TXO_DA(ATT_SELF, APRO_NONE, NLS, 1);
        else if (o.apro == APRO_EMBED) TXO_DA(ATT_SELF, APRO_EMBED, NLS, WBS);
        else TXO_DA(ATT_SELF, APRO_LN2, NLS, WBS);
#undef TXO_DA
#undef TXO_DA1
        if (timed) ev_cross.push_back({e0, e1});
    }

    // Where the latent form measured faster on MI355X with launches (config.yml dims, 224x672, bf16): from 129 rows on (batch 256:
    // cross-attention launch 49 -> 32 us, 2700 -> 3290 images/s).  Up to 128 rows the persistent kernel decodes (K/V form: at one
    // (row, head group) tile per CU the latent tile moves 493 KB per CU against 300 KB and is slower, 19 vs 16 us per launch at batch
    // 64); wide decoders (768: the in-tile projections' weights are 3.4 MB per row) and the fp32 parity mode keep the K/V form.
    // the latent form's two per-head projections fold into their neighbouring GEMMs (load_attn) where the folded weights stay small
    bool latent_fold() const { return latent_ok && cfg.dec_heads * DH == 2 * D; }
    bool auto_latent(int rows) const { return sizeof(T) == 2 && D == 256 && rows > PERSIST_MAX_BF16_GREEDY; }
    // heads per latent tile: the smallest group that leaves at most one tile per CU for `rows` rows (fewer tiles = fewer
    // re-reads of an image's encoder rows; more tiles = more CUs pulling).  A head's bits do not depend on it.
    int latent_group(int rows, int slots) const {
        const int H = cfg.dec_heads;
        if (lat_g_env > 0) return std::min(std::min(H, LA_GMAX), lat_g_env);
        for (int g = 2; g < std::min(H, LA_GMAX); g *= 2)
            if (rows * ((H + g - 1) / g) <= slots) return g;
        const int n = (H + LA_GMAX - 1) / LA_GMAX;            // as few tiles per row as the tile allows, evenly filled (12 heads: one tile of 12; 24: 12 + 12)
        return (H + n - 1) / n;
    }
    // cross attention of layer l in latent form (lat_attn.h): [LN sandwich + q] -> [q' per head] -> [scores / values against the raw
    // encoder rows] -> [Wv per head]; the gated output projection follows in enqueue_step as in the K/V form
    template <int KG>
    void launch_grp_gemm(hipStream_t s, const T* A, int lda, const T* W, T* out, int ldo, int rows, int N, int NG) {
        GrpGemmArgs<T> g{A, lda, W, out, ldo, rows, N, NG};
        hipLaunchKernelGGL((grp_gemm_kernel<T, KG>), dim3((N + 63) / 64, (rows + 15) / 16), dim3(256), 0, s, g);
    }
    // the latent core over lane li's rows: q' in dqp, c to dcl.  Cross attention: enc = the session's encoder rows of the lane's first image,
    // len = sN, every beam of an image reads the same rows (kv_div).  Self attention: enc = the lane's rows of the z history of one layer,
    // enc_rows = Tmax, len = position + 1 (host value or *t_ptr), kv_div = 1, path = the beams' slot tables (or null).
    // The latent core's output c feeds the folded output projection as its A operand: written in that GEMM's tiled layout (dec_gemm.h: a_tiled)
    // when the core serves the CROSS attention with folded weights.  A lane's region starts on a 16-row tile boundary and lanes do not overlap.
    int a_tiled_on = getenv("TXO_A_TILED") ? atoi(getenv("TXO_A_TILED")) : 1;
    size_t c_base_row(int li) const { return ((lanes[li].b0 + 15) / 16) * 16 + (size_t)16 * li; }
    bool c_tiled(int l, bool cross) const { return a_tiled_on && w_tiled_on && cross && dec_cross[l].wqp != nullptr && (D * (int)sizeof(T)) % 64 == 0; }
    void launch_lat_core(hipStream_t s, int li, int kv_div, const T* enc, int len, int enc_rows, const int* t_ptr, const short* path,
                         const char* stamp_name, bool cross, int layer = 0) {
        const Lane& ln = lanes[li];
        const size_t r0 = ln.b0;
        const int H = cfg.dec_heads, HD = H * D;
        LatCoreArgs<T> a{};
        a.qp = dqp + r0 * HD; a.c = dcl + r0 * HD; a.enc = enc;
        if (c_tiled(layer, cross)) { a.c = dcl + c_base_row(li) * HD; a.c_hpr = H; }
        a.rows = ln.nb; a.heads = H; a.G = latent_group(sB, n_cus); a.ngrp = (H + a.G - 1) / a.G; a.len = len; a.kv_div = kv_div;
        a.enc_rows = enc_rows; a.t_ptr = t_ptr; a.path = path; a.path_stride = Tmax;
        int nimg = (ln.nb + kv_div - 1) / kv_div;
        if (cross && kv_div > 1 && ln.nb % kv_div == 0 && lat_g_env == 0) {
            // beam search: the k beams of an image read the SAME encoder rows, and their q' / c rows are contiguous ([rows][heads * D]) --
            // one image = ONE row of k * heads heads, LA_GMAX of them per tile (the MFMA tile has 16 head columns whether 8 or 16 are
            // used): 640 tiles of 8 heads become 384 of up to 16.  A head's bits do not depend on the grouping.
            a.rows = nimg; a.heads = H * kv_div; a.G = LA_GMAX; a.ngrp = (a.heads + LA_GMAX - 1) / LA_GMAX; a.kv_div = 1;
        }
        const int nblk = ((nimg + 7) / 8) * 8 * a.kv_div * a.ngrp;   // XCD-aware tile order (lat_core_kernel)
        a.stamps = (nblk <= STAMP_BLOCKS) ? next_stamp(stamp_name) : nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool timed = cross && (prof || (prof_cross && (cross_seq++ & 3) == 0 && pool.used + 2 <= pool.ev.size()));
        if (timed) { e0 = pool.next(); e1 = pool.next(); }
#define TXO_LA(D_)                                                                                                           \
        do {                                                                                                                 \
            if constexpr (la_supported<T, D_>()) {                                                                           \
                auto kern = lat_core_kernel<T, D_>;                                                                          \
                const size_t lds = la_lds_bytes<T, D_>();                                                                    \
                if (timed) hipExtLaunchKernelGGL(kern, dim3(nblk), dim3(la_waves<D_>() * 64), lds, s, e0, e1, 0, a);       \
                else hipLaunchKernelGGL(kern, dim3(nblk), dim3(la_waves<D_>() * 64), lds, s, a);                            \
            }                                                                                                                \
        } while (0)
        if (D == 256 && lat_nw4) {                                // two 4-wave tiles per CU (lat_attn.h: NW_)
            if constexpr (la_supported<T, 256>() && sizeof(T) == 2) {
                auto kern = lat_core_kernel<T, 256, 4>;
                const size_t lds = la_lds_bytes<T, 256, 4>();
                if (timed) hipExtLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, s, e0, e1, 0, a);
                else hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, s, a);
            }
        } else if (D == 64) TXO_LA(64); else if (D == 256) TXO_LA(256); else TXO_LA(768);
#undef TXO_LA
        if (timed) ev_cross.push_back({e0, e1});
    }
    // self attention of layer l in latent form (folded weights): [embedding / LN sandwich + q' = z M^T, z appended to the history] ->
    // [scores / values against the history of z]; the gated output projection (Wo' folded) follows in enqueue_step
    int launch_lat_self(hipStream_t s, int li, int l, const BeamCtx* bm, const DecGemmArgs<T>& base) {
        const Lane& ln = lanes[li];
        const size_t r0 = ln.b0;
        const int HD = cfg.dec_heads * D;
        T* zl = zc + ((size_t)l * sB + r0) * Tmax * D;           // this lane's rows of layer l's history
        DecGemmArgs<T> a = base; a.N = HD; a.K = D; a.W = dec_self[l].wqp; a.y = dy + r0 * D; a.x_out = dx + r0 * D;
        a.tok = cur_tok + r0; a.tok_emb = tok_emb; a.pos_emb = pos_emb;
        a.h_out = dqp + r0 * HD; a.F = HD; a.z_cache = zl;
        if (int r = (l == 0 ? launch_dec_gemm<PRO_EMBED, EPI_STORE_T>(s, a) : launch_dec_gemm<PRO_LN2, EPI_STORE_T>(s, a))) return r;
        launch_lat_core(s, li, 1, zl, step_host_t + 1, Tmax, step_host_t >= 0 ? nullptr : &st[li].t, bm ? bm->path_cur + r0 * Tmax : nullptr,
                        "attn self (latent core)", false);
        return 0;
    }
    int launch_lat_cross(hipStream_t s, int li, int l, int kv_div, const DecGemmArgs<T>& base) {
        const Lane& ln = lanes[li];
        const size_t r0 = ln.b0;
        const int H = cfg.dec_heads, HD = H * D;
        const bool fold = dec_cross[l].wqp != nullptr;
        if (fold) {   // 1+2. x = LN(y) (residual), z = LN(x), q' = z M^T  (M = 0.125 Wk_h^T Wq_h folded at load)
            DecGemmArgs<T> a = base; a.N = HD; a.K = D; a.W = dec_cross[l].wqp; a.y = dy + r0 * D; a.x_out = dx + r0 * D;
            a.h_out = dqp + r0 * HD; a.F = HD;
            if (int r = launch_dec_gemm<PRO_LN2, EPI_STORE_T>(s, a)) return r;
        } else {
            {   // 1. x = LN(y) (residual), z = LN(x), q = z Wq^T
                DecGemmArgs<T> a = base; a.N = Id; a.K = D; a.W = dec_cross[l].wq; a.y = dy + r0 * D; a.x_out = dx + r0 * D;
                a.h_out = dqt + r0 * Id; a.F = Id;
                if (int r = launch_dec_gemm<PRO_LN2, EPI_STORE_T>(s, a)) return r;
            }
            // 2. q'_h = q_h (0.125 Wk_h)
            launch_grp_gemm<DH>(s, dqt + r0 * Id, Id, dec_cross[l].wkT, dqp + r0 * HD, HD, ln.nb, HD, D);
        }
        // 3. c_h = softmax_n(q'_h . enc[n]) enc
        launch_lat_core(s, li, kv_div, enc_t + (r0 / kv_div) * (size_t)sN * D, sN, 0, nullptr, nullptr, "attn cross (latent core)", true, l);
        if (fold) return 0;                                       // 4+5: the gated output projection takes c directly (K = heads*D, Wo' folded at load)
        // 4. o_h = c_h Wv_h^T ; 'b h n d -> b n (h d)'
        if (D == 64) launch_grp_gemm<64>(s, dcl + r0 * HD, HD, dec_cross[l].wv, dao + r0 * Id, Id, ln.nb, Id, DH);
        else if (D == 256) launch_grp_gemm<256>(s, dcl + r0 * HD, HD, dec_cross[l].wv, dao + r0 * Id, Id, ln.nb, Id, DH);
        else launch_grp_gemm<768>(s, dcl + r0 * HD, HD, dec_cross[l].wv, dao + r0 * Id, Id, ln.nb, Id, DH);
        return 0;
    }

    // one decode position of lane `li` on stream s; tokens_out/logits_out are GLOBAL-batch base pointers
    // host_t: the decode position when the caller knows it (every eager loop does); -1 makes the kernels read the
    // device-side counter instead, which is what lets ONE captured graph serve every step
    int enqueue_step(hipStream_t s, int li, int64_t* tokens_out, int out_stride, float* logits_out, int eos,
                     const BeamCtx* bm = nullptr, int host_t = -1) {
        step_host_t = host_t;
        const Lane& ln = lanes[li];
        const int B = sB, N = sN, nb = ln.nb;
        const size_t r0 = ln.b0;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (prof) { e0 = pool.next(); e1 = pool.next(); (void)hipEventRecord(e0, s); }
        DecGemmArgs<T> base{};
        base.rows = nb; base.gamma = dec_g; base.beta = dec_b; base.t_ptr = &st[li].t; base.t_host = host_t; base.D = D;
        base.inner = Id; base.heads = cfg.dec_heads; base.tmax = Tmax;
        float* lx = dx + r0 * D; float* ly = dy + r0 * D; T* lao = dao + r0 * Id; T* lhid = dhid + r0 * Fd;
        float* llog = dlogits + r0 * V;
        const size_t self_stride = (size_t)B * Id * Tmax, cross_stride = (size_t)sImg * N * Id;
        for (int l = 0; l < cfg.dec_layers; ++l) {
            T* kc = skv + (size_t)(2 * l) * self_stride; T* vc = skv + (size_t)(2 * l + 1) * self_stride;
            if (lat_self) {   // causal self attention against the history of normalised block inputs (lat_attn.h)
                if (int r = launch_lat_self(s, li, l, bm, base)) return r;
                dbg(s, "self attn (latent)", l);
                DecGemmArgs<T> g = base; g.N = 2 * D; g.K = cfg.dec_heads * D; g.W = dec_self[l].wo_f; g.bias = dec_self[l].bo;
                g.A = dcl + r0 * cfg.dec_heads * D; g.resid = lx; g.y_out = ly;
                if (!launch_dec_gemm_wide<EPI_GLU_RES>(s, g)) { if (int r = launch_dec_gemm<PRO_NONE, EPI_GLU_RES>(s, g)) return r; }
                dbg(s, "self out", l);
            } else {   // causal self attention
                AttnOpt o; o.W = dec_self[l].wqkv; o.K = kc; o.V = vc; o.lmax = Tmax; o.x_out = lx;
                if (bm) o.path = bm->path_cur;
                if (self_plain || bm || kmask_on) {
                    // default: LN sandwich + QKV GEMM (weights read once per 16 rows; k/v appended to the cache by
                    // its epilogue), then the plain cached attention.  TXO_SELF_FUSED=1 folds the projection into
                    // the attention launch instead (same wall time at B=64; re-reads 96 KB of weights per image).
                    DecGemmArgs<T> a = base; a.N = 3 * Id; a.K = D; a.W = dec_self[l].wqkv; a.y = ly; a.x_out = lx;
                    a.tok = cur_tok + r0; a.tok_emb = tok_emb; a.pos_emb = pos_emb;
                    a.q_out = dq + r0 * Id;
                    a.k_cache = kc + r0 * cfg.dec_heads * Tmax * DH; a.v_cache = vc + r0 * cfg.dec_heads * Tmax * DH;
                    // (one LayerNorm launch + the projection on 64-row blocks, as the out-projections and FFN-in below do at >= 128
                    // rows of a wide decoder, measured the same 17.7 us as this fused launch at 256 x 768: not used here)
                    if (int r = (l == 0 ? launch_dec_gemm<PRO_EMBED, EPI_QKV>(s, a) : launch_dec_gemm<PRO_LN2, EPI_QKV>(s, a))) return r;
                    o.apro = APRO_NONE;
                } else {
                    o.apro = l == 0 ? APRO_EMBED : APRO_LN2;
                }
                launch_dec_attn(s, li, o);
                dbg(s, "self attn", l);
                DecGemmArgs<T> g = base; g.N = 2 * D; g.K = Id; g.W = dec_self[l].wo; g.bias = dec_self[l].bo; g.A = lao;
                g.resid = lx; g.y_out = ly;
                if (!launch_dec_gemm_wide<EPI_GLU_RES>(s, g)) { if (int r = launch_dec_gemm<PRO_NONE, EPI_GLU_RES>(s, g)) return r; }
                dbg(s, "self out", l);
            }
            {   // cross attention (LN sandwich + q projection fused in): against the raw encoder rows (latent form) or over the cached projections
                const bool fold = use_latent && dec_cross[l].wqp != nullptr;
                if (use_latent) { if (int r = launch_lat_cross(s, li, l, bm ? bm->k : 1, base)) return r; }
                else {
                AttnOpt o; o.cross = true; o.W = dec_cross[l].wq; o.K = ckv + (size_t)(2 * l) * cross_stride;
                o.V = ckv + (size_t)(2 * l + 1) * cross_stride; o.lmax = N; o.len = N; o.x_out = lx;
                if (bm) o.kv_div = bm->k;
                launch_dec_attn(s, li, o);
                }
                dbg(s, "cross attn", l);
                DecGemmArgs<T> g = base; g.N = 2 * D; g.K = Id; g.W = dec_cross[l].wo; g.bias = dec_cross[l].bo; g.A = lao;
                g.resid = lx; g.y_out = ly;
                if (fold) {
                    g.K = cfg.dec_heads * D; g.W = dec_cross[l].wo_f; g.A = dcl + r0 * cfg.dec_heads * D;
                    if (c_tiled(l, true)) { g.A = dcl + c_base_row(li) * cfg.dec_heads * D; g.a_tiled = 1; }
                }
                if (!launch_dec_gemm_wide<EPI_GLU_RES>(s, g)) { if (int r = launch_dec_gemm<PRO_NONE, EPI_GLU_RES>(s, g)) return r; }
                dbg(s, "cross out", l);
            }
            {   // GeGLU feed-forward
                if (sizeof(T) == 2 && D >= 512 && nb >= 128 && dec_mlp[l].w1_16) {
                    // perf mode, wide rows (ViT-Base decoder: 768), >= 128 rows: the 16-row launch would be 3072 blocks that each
                    // redo the LayerNorm sandwich of their rows (39 us); one LayerNorm launch + the 128x128 GEMM takes ~15 us.
                    // (fp32 parity mode keeps the 16-row kernel at every batch size: a row's bits never depend on the batch.)
                    launch_ln<1>(s, ly, lx, dz + r0 * D, dec_g, dec_b, nb);
                    DecGemmArgs<T> w = base; w.N = 2 * Fd; w.K = D; w.W = dec_mlp[l].w1; w.bias = dec_mlp[l].b1; w.A = dz + r0 * D;
                    w.h_out = lhid; w.F = Fd;
                    if (!launch_dec_gemm_wide<EPI_GEGLU>(s, w))
                        launch_gemm_big<T>(s, LoadPlain<T>{dz + r0 * D, D}, dec_mlp[l].w1_16, nb, 2 * Fd, D, EpiGeglu<T>{lhid, dec_mlp[l].b1_16, Fd});
                } else {
                    DecGemmArgs<T> a = base; a.N = 2 * Fd; a.K = D; a.W = dec_mlp[l].w1; a.bias = dec_mlp[l].b1; a.y = ly; a.x_out = lx;
                    a.h_out = lhid; a.F = Fd;
                    if (int r = launch_dec_gemm<PRO_LN2, EPI_GEGLU>(s, a)) return r;
                }
                dbg(s, "ffn1", l);
                DecGemmArgs<T> g = base; g.N = D; g.K = Fd; g.W = dec_mlp[l].w2; g.bias = dec_mlp[l].b2; g.A = lhid;
                g.resid = lx; g.y_out = ly;
                if (int r = launch_dec_gemm<PRO_NONE, EPI_BIAS_RES>(s, g)) return r;   // (blocks of 32 / 64 rows x 16 / 32 columns at K = 3072 on two ranges: 840 / 808 / 823 / 784 img/s against 844 -- not used)
            }
        }
        DecGemmArgs<T> f = base; f.N = V; f.K = D; f.W = wlog; f.bias = blog; f.logits = llog;
        f.y = ly; f.gamma = decn_g; f.beta = decn_b;
        if (int r = launch_dec_gemm<PRO_LNF, EPI_LOGITS>(s, f)) return r;
        dbg(s, "logits");
        StepArgs sa{llog, V, nb, cur_tok + r0, tokens_out ? tokens_out + r0 * out_stride : nullptr, out_stride,
                    logits_out ? logits_out + r0 * (size_t)out_stride * V : nullptr, st + li, eos_seen + r0,
                    done_flag + (size_t)li * Tmax, eos, sample_topk, 1.0f / sample_temp, sample_seed, (int)r0,
                    row_stop ? 1 : 0, row_stop ? row_map + r0 : nullptr};
        if (bm) {
            BeamArgs ba{llog, V, bm->k, nb / bm->k, cur_tok + r0, bscore + r0, bfin + r0, bm->path_cur + r0 * Tmax, bm->path_nxt + r0 * Tmax, Tmax,
                        bparent + r0, btok + r0, Bmax, st + li, done_flag + (size_t)li * Tmax, eos, (int)r0};
            hipLaunchKernelGGL(beam_select_kernel, dim3(nb / bm->k), dim3(256), 0, s, ba);
        } else if (sample_mode) launch_sample_step(s, nb, sa);
        else hipLaunchKernelGGL(argmax_step_kernel, dim3(nb), dim3(64), 0, s, sa);
        dbg(s, "argmax");
        if (prof) { (void)hipEventRecord(e1, s); ev_step.push_back({e0, e1}); }
        return 0;
    }

    void dump_stamps(const char* file, hipStream_t s) {
        const int nk = stamp_slot;
        stamp_slot = -1;
        std::vector<unsigned long long> h((size_t)nk * STAMP_BLOCKS * 3);
        if (hipStreamSynchronize(s) != hipSuccess) return;
        if (hipMemcpy(h.data(), stamp_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
        FILE* f = fopen(file, "w");
        if (!f) return;
        unsigned long long t0 = ~0ull;
        for (auto v : h) if (v && v < t0) t0 = v;
        for (int k2 = 0; k2 < nk; ++k2) {
            unsigned long long first = ~0ull, last_in = 0, mid_lo = ~0ull, mid_hi = 0, first_out = ~0ull, last = 0; int nb = 0;
            for (int b = 0; b < STAMP_BLOCKS; ++b) {
                const unsigned long long* d = &h[((size_t)k2 * STAMP_BLOCKS + b) * 3];
                if (!d[0]) continue;
                ++nb; first = std::min(first, d[0]); last_in = std::max(last_in, d[0]); mid_lo = std::min(mid_lo, d[1]); mid_hi = std::max(mid_hi, d[1]);
                first_out = std::min(first_out, d[2]); last = std::max(last, d[2]);
            }
            fprintf(f, "%-24s blocks %4d | first entry %7.2f us, last entry %7.2f | operands/panel done %7.2f .. %7.2f | first exit %7.2f, last exit %7.2f\n",
                    stamp_names[k2].c_str(), nb, (first - t0) / 100.0, (last_in - t0) / 100.0, (mid_lo - t0) / 100.0, (mid_hi - t0) / 100.0,
                    (first_out - t0) / 100.0, (last - t0) / 100.0);
        }
        // TXO_STAMPS_RAW=<substring>: also one line per block of the launches whose name contains it (which CU / tile is the slow one)
        if (const char* raw = getenv("TXO_STAMPS_RAW")) {
            for (int k2 = 0; k2 < nk; ++k2) {
                if (stamp_names[k2].find(raw) == std::string::npos) continue;
                for (int b = 0; b < STAMP_BLOCKS; ++b) {
                    const unsigned long long* d = &h[((size_t)k2 * STAMP_BLOCKS + b) * 3];
                    if (d[0]) fprintf(f, "raw %d %-24s block %4d  %7.2f %7.2f %7.2f\n", k2, stamp_names[k2].c_str(), b, (d[0] - t0) / 100.0, (d[1] - t0) / 100.0, (d[2] - t0) / 100.0);
                }
            }
        }
        fclose(f);
    }

    // capture lane li's step (tokens into the engine-owned tok_buf) as a graph, or reuse the cached one
    int lane_graph(int li, int eos) {
        Lane& ln = lanes[li];
        const std::array<int, 7> key = {ln.b0, ln.nb, sN, eos, sB, sImg, (int)use_latent + 2 * (int)lat_self + 4 * (int)row_stop + 8 * sample_mode};
        auto it = ln.graphs.find(key);
        if (it != ln.graphs.end()) { ln.exec = it->second.second; return 0; }
        if (ln.graphs.size() >= 64) {                              // (shapes keep changing: start over rather than grow without bound)
            for (auto& g : ln.graphs) { (void)hipGraphExecDestroy(g.second.second); (void)hipGraphDestroy(g.second.first); }
            ln.graphs.clear();
        }
        hipStream_t cs = cap_stream;
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        const int r = enqueue_step(cs, li, tok_buf, Tmax, nullptr, eos);
        hipError_t e = hipStreamEndCapture(cs, &graph);
        if (r) return r;
        if (e != hipSuccess) return fail(TXO_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        ln.graphs[key] = {graph, exec};
        ln.exec = exec;
        return 0;
    }

    int decode_step(const int64_t* tok_in, int t, float* logits_out, int64_t* tok_out, hipStream_t s) override {
        if (!session) return fail(TXO_E_STATE, "txo_decode_begin has not been called");
        if (t < 0 || t >= Tmax)
            return fail(TXO_E_INVALID, "position outside the decoder's positional table (the reference would slide its "
                                       "window, decoder.py:99-100; a KV cache cannot reproduce that)");
        if (n_lanes != 1) return fail(TXO_E_STATE, "decode_step needs a session started by txo_decode_begin");
        lanes[0].stream = s;
        if (tok_in) hipLaunchKernelGGL(copy_tokens_kernel, dim3((sB + 255) / 256), dim3(256), 0, s, cur_tok, tok_in, sB, V);
        hipLaunchKernelGGL(set_position_kernel, dim3(1), dim3(1), 0, s, st, t);
        if (int r = enqueue_step(s, 0, nullptr, 0, nullptr, -1, nullptr, t)) return r;
        if (logits_out) HIP_TRY(hipMemcpyAsync(logits_out, dlogits, sizeof(float) * sB * V, hipMemcpyDeviceToDevice, s));
        if (tok_out) HIP_TRY(hipMemcpyAsync(tok_out, cur_tok, sizeof(int64_t) * sB, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipGetLastError());
        return 0;
    }

    // Padding mask of decoder.generate / decoder.net (reference decoder.py:95-101,112; attention.py:130-155): mask [B][cols] bytes,
    // 0 = the position is padding.  Its effect on every position that is NOT itself padding: padded positions are never attended by
    // the causal self attention (energy filled with -FLT_MAX).  Rows of padded positions are computed but unspecified (the reference
    // softmaxes them uniformly over all keys, future ones included; nothing reads them).  Applies to txo_decode_step of this session.
    int decode_set_key_mask(const unsigned char* mask, int cols, hipStream_t s) override {
        if (!session) return fail(TXO_E_STATE, "txo_decode_begin has not been called");
        if (sImg != sB) return fail(TXO_E_STATE, "key masks are not available inside a beam-search session");
        if (!mask) { kmask_on = false; return 0; }
        if (cols < 1 || cols > Tmax) return fail(TXO_E_INVALID, "mask columns must be in [1, max_length]");
        const int n = sB * Tmax;
        hipLaunchKernelGGL(set_key_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mask, kmask, sB, cols, Tmax);
        kmask_on = true;
        HIP_TRY(hipGetLastError());
        return 0;
    }

    int decode_prefill(const int64_t* tokens, int t, float* logits_out, hipStream_t s) override {
        if (n_lanes != 1) return fail(TXO_E_STATE, "decode_prefill needs a session started by txo_decode_begin");
        if (sImg != sB) return fail(TXO_E_STATE, "decode_prefill is not available inside a beam-search session");
        lanes[0].stream = s;
        return prefill(tokens, t, t, logits_out, nullptr, s);
    }

    // ---- multi-position decoder forward (prefill.h): Transformer.forward over t positions at once, filling the self K/V cache ----
    // tokens [B][tok_stride] (the first t of each row); logits_out [B][t][V] or null; last_logits [B][V] or null (final LN + logits of
    // position t-1 only, on the single-position launch).  Runs in the encoder's workspace, image chunks of as many rows as fit.
    template <typename TO>
    void launch_attn_mq(hipStream_t s, bool causal, const T* q, const T* k, const T* v, TO* out, int nb, int nq, int nk, int kv_rows,
                        const unsigned char* km = nullptr) {
        const dim3 grid((nq + EA_QBLK - 1) / EA_QBLK, nb * cfg.dec_heads);
        if (causal && km) hipLaunchKernelGGL((attn_mq_kernel<T, TO, true, true>), grid, dim3(256), 0, s, q, k, v, out, nq, nk, kv_rows, cfg.dec_heads, km, Tmax);
        else if (causal) hipLaunchKernelGGL((attn_mq_kernel<T, TO, true>), grid, dim3(256), 0, s, q, k, v, out, nq, nk, kv_rows, cfg.dec_heads);
        else hipLaunchKernelGGL((attn_mq_kernel<T, TO, false>), grid, dim3(256), 0, s, q, k, v, out, nq, nk, kv_rows, cfg.dec_heads);
    }
    int prefill(const int64_t* tokens, int tok_stride, int t, float* logits_out, float* last_logits, hipStream_t s) {
        if (!session) return fail(TXO_E_STATE, "txo_decode_begin has not been called");
        if (t < 1 || t > Tmax) return fail(TXO_E_INVALID, "prefill length outside the decoder's positional table");
        if (V % 8) return fail(TXO_E_INVALID, "prefill needs a vocabulary size that is a multiple of 8");
        const size_t cap = (size_t)Bmax * Nmax;               // rows of the encoder workspace
        if ((size_t)t > cap) return fail(TXO_E_INVALID, "prefill: one prefix does not fit the engine's workspace (max_batch * max_tokens rows)");
        const int B = sB, N = sN, heads = cfg.dec_heads;
        ensure_ckv(s);
        const int bc = (int)std::min<size_t>(B, cap / t);     // images per chunk
        const size_t self_stride = (size_t)B * Id * Tmax, cross_stride = (size_t)sImg * N * Id;
        T* qbuf = reinterpret_cast<T*>(eqkv);
        for (int b0 = 0; b0 < B; b0 += bc) {
            const int nb = std::min(bc, B - b0), M = nb * t;
            {
                const size_t n4 = (size_t)M * (D / 4);
                hipLaunchKernelGGL(embed_rows_kernel, dim3((n4 + 255) / 256), dim3(256), 0, s, tokens + (size_t)b0 * tok_stride, tok_stride,
                                   tok_emb, pos_emb, ex, M, t, D, V);
            }
            for (int l = 0; l < cfg.dec_layers; ++l) {
                // the shared-LN sandwich exactly as in encode(): x = LN(y) is never written, the GEMM epilogues rebuild it
                const ResidLN res_x{ey, estats, dec_gb, D}, res_first{ex, nullptr, nullptr, D};
                if (l == 0) launch_ln<0, T>(s, ex, nullptr, ez, dec_g, dec_b, M);
                else launch_ln<3, T>(s, ey, estats, ez, dec_g, dec_b, M);
                T* kc = skv + (size_t)(2 * l) * self_stride + (size_t)b0 * heads * Tmax * DH;
                T* vc = skv + (size_t)(2 * l + 1) * self_stride + (size_t)b0 * heads * Tmax * DH;
                gemm_plain(s, ez, dec_self[l].wqkv, M, 3 * Id, D, EpiHeadsKV<T>{qbuf, kc, vc, Id, heads, t, Tmax});
                launch_attn_mq<T>(s, true, qbuf, kc, vc, eao, nb, t, t, Tmax, kmask_on ? kmask + (size_t)b0 * Tmax : nullptr);   // (padding mask of the session, if any)
                gemm_plain(s, eao, dec_self[l].wo16, M, 2 * D, Id, EpiGluRes<sizeof(T) == 2>{ey, l == 0 ? res_first : res_x, dec_self[l].bo16});
                // cross attention over the cached encoder projections (attention.py:114-126: k, v from the raw encoder output)
                launch_ln<3, T>(s, ey, estats, ez, dec_g, dec_b, M);
                gemm_plain(s, ez, dec_cross[l].wq, M, Id, D, EpiHeads<T>{qbuf, 0, Id, heads, t});
                const T* ck = ckv + (size_t)(2 * l) * cross_stride + (size_t)(b0 / std::max(1, sB / sImg)) * heads * N * DH;
                const T* cv = ckv + (size_t)(2 * l + 1) * cross_stride + (size_t)(b0 / std::max(1, sB / sImg)) * heads * N * DH;
                launch_attn_mq<T>(s, false, qbuf, ck, cv, eao, nb, t, N, N);
                gemm_plain(s, eao, dec_cross[l].wo16, M, 2 * D, Id, EpiGluRes<sizeof(T) == 2>{ey, res_x, dec_cross[l].bo16});
                // GeGLU feed-forward
                launch_ln<3, T>(s, ey, estats, ez, dec_g, dec_b, M);
                gemm_plain(s, ez, dec_mlp[l].w1_16, M, 2 * Fd, D, EpiGeglu<T>{ehid, dec_mlp[l].b1_16, Fd});
                gemm_plain(s, ehid, dec_mlp[l].w2, M, D, Fd, EpiBiasRes{ey, res_x, dec_mlp[l].b2});
            }
            if (logits_out) {                                 // decoder.py:57-60 over every position
                launch_ln<2, T>(s, ey, nullptr, ez, decn_g, decn_b, M);
                gemm_plain(s, ez, wlog, M, V, D, EpiStore<float>{logits_out + (size_t)b0 * t * V, V, blog});
            }
            if (last_logits) {                                // the last position only: the step path's final-LN + logits launch on gathered rows
                hipLaunchKernelGGL(gather_last_rows_kernel, dim3((nb * D + 255) / 256), dim3(256), 0, s, ey, dy + (size_t)b0 * D, nb, t, D);
            }
        }
        if (last_logits) {
            DecGemmArgs<T> f{};
            f.rows = B; f.D = D; f.inner = Id; f.heads = heads; f.tmax = Tmax; f.t_host = t - 1; f.t_ptr = &st[0].t;
            f.N = V; f.K = D; f.W = wlog; f.bias = blog; f.logits = last_logits; f.y = dy; f.gamma = decn_g; f.beta = decn_b;
            step_host_t = t - 1;
            if (int r = launch_dec_gemm<PRO_LNF, EPI_LOGITS>(s, f)) return r;
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }

    // ---- the decode loop as ONE persistent launch (persist.h) -------------------------------------------------------
    // Which decode path?  TXO_PERSIST=1 / 0 forces the persistent launch on / off (where it exists: decoder width 256 with 8
    // heads, or 768 with 12 heads in bf16, FFN factor 4).  Default: where it measured faster on MI355X (config.yml dims, 224x672,
    // 256 steps; profiles/r02_persist_ab.txt, last table): width 256, every batch size up to 256 images in both modes -- bf16
    // 22.9 vs 29.1 ms for ONE image, 1998 vs 1522 images/s at batch 64, 2717 vs 2605 at 256; fp32 1094 vs 903 at batch 64.
    static constexpr int PERSIST_MAX_BF16_GREEDY = 128;
    bool persist_usable(int B) const {
        // sampling: the persistent kernel's sampler keeps a row in registers, 16 logits per lane (step.h: sample_row_regs)
        if (sample_mode && V > 64 * SR_PER) return false;
        if (prof || prof_cross || g_dbg || knobs.stamps || knobs.graph >= 0 || knobs.lanes > 0) return false;
        if (cfg.dec_exp != 4 || cfg.dec_layers > PS_MAXLD) return false;
        const bool exists = (D == 256 && cfg.dec_heads == 8) || (D == 768 && cfg.dec_heads == 12 && sizeof(T) == 2);
        if (!exists) return false;
        if (latent_ok && lat_mode == 1) return false;           // latent form forced: the persistent kernel reads projected K/V panels
        if (knobs.persist >= 0) return knobs.persist != 0;
        if (persist_cooldown > 0) return false;                // it gave up twice in a row (not all 256 workgroups co-resident?): not tried for a while
        if (D != 256) return false;                            // the 768-wide variant is opt-in (TXO_PERSIST=1): not measured faster
        // bf16 beyond 128 images: launches with the cross attention in latent form (one row range, two from 224 rows on) are ahead of the
        // persistent launch -- greedy 160: 66.6 vs 72.7 ms, 192: 70.6 vs 79.7, 256: 77.9 vs 99.0; sampled 160: 68.9 vs 73.4, 192: 73.0 vs 80.6, 256: 84.5
        // (one range) vs 99.7; at 128 the persistent launch leads (greedy 54.2 vs 56.7, sampled 55.2 vs 58.4).  fp32: persistent up to 256
        if (sizeof(T) == 2 && B > PERSIST_MAX_BF16_GREEDY) return false;
        return B <= 256;                                       // more rows per team than two 16-row tiles: not measured
    }
    template <int D_, int H_>
    int launch_persist(const PersistArgs<T>& pa, hipStream_t s) {
        const size_t lds = persist_lds_bytes<T, D_, H_>();
        auto kern = sample_mode ? decode_persist_kernel<T, D_, H_, true> : decode_persist_kernel<T, D_, H_, false>;
        // per DEVICE, not per process (one process may drive several GPUs through several engines): set on every launch, it is cheap
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return fail(TXO_E_STATE, "persistent decode: the device refused the launch's dynamic LDS size");   // -> decode with launches
        }
        if (prof_persist && pool.used + 2 <= pool.ev.size()) {      // events bound to the dispatch itself (its begin / end timestamps)
            hipEvent_t e0 = pool.next(), e1 = pool.next();
            hipExtLaunchKernelGGL(kern, dim3(PS_TEAMS * PS_TEAM_BLOCKS), dim3(PS_THREADS), lds, s, e0, e1, 0, pa);
            ev_persist.push_back({e0, e1});
        } else {
            hipLaunchKernelGGL(kern, dim3(PS_TEAMS * PS_TEAM_BLOCKS), dim3(PS_THREADS), lds, s, pa);
        }
        return 0;
    }
    // returns 0 (done), TXO_E_STATE (the launch gave up: redo with launches), or an error
    // n_pos positions are decoded; rows of tokens_out / logits_out are out_stride positions apart
    // *broke: the reference's GLOBAL eos break fired inside these n_pos positions (possibly at the very last one)
    int generate_persist(int B, int N, int n_pos, int out_stride, int eos, int64_t* tokens_out, float* logits_out, int* n_steps, bool* broke, hipStream_t s) {
        const int max_len = n_pos;
        PersistArgs<T> pa{};
        pa.B = B; pa.N = N; pa.V = V; pa.Ld = cfg.dec_layers; pa.Tmax = Tmax; pa.max_len = max_len; pa.eos = eos; pa.bos = cfg.bos;
        for (int l = 0; l < cfg.dec_layers; ++l) {
            PersistLayer<T>& L = pa.L[l];
            L.wqkv = dec_self[l].wqkv; L.wo_s = dec_self[l].wo; L.bo_s = dec_self[l].bo;
            L.wq_c = dec_cross[l].wq; L.wo_c = dec_cross[l].wo; L.bo_c = dec_cross[l].bo;
            L.w1 = dec_mlp[l].w1; L.b1 = dec_mlp[l].b1; L.w2 = dec_mlp[l].w2; L.b2 = dec_mlp[l].b2;
        }
        pa.wlog = wlog;
        {   // the projections' weights as their tiled copies (all of them, or none)
            bool all = w_tiled_on && wtiled.count(wlog);
            for (int l = 0; l < cfg.dec_layers && all; ++l)
                all = wtiled.count(dec_self[l].wqkv) && wtiled.count(dec_self[l].wo) && wtiled.count(dec_cross[l].wo) && wtiled.count(dec_mlp[l].w1) && wtiled.count(dec_mlp[l].w2);
            pa.w_tiled = all ? 1 : 0;
            if (all) {
                for (int l = 0; l < cfg.dec_layers; ++l) {
                    PersistLayer<T>& L = pa.L[l];
                    L.wqkv = wtiled.at(dec_self[l].wqkv); L.wo_s = wtiled.at(dec_self[l].wo); L.wo_c = wtiled.at(dec_cross[l].wo);
                    L.w1 = wtiled.at(dec_mlp[l].w1); L.w2 = wtiled.at(dec_mlp[l].w2);
                }
                pa.wlog = wtiled.at(wlog);
            }
        }
        pa.gamma = dec_g; pa.beta = dec_b; pa.gamma_f = decn_g; pa.beta_f = decn_b; pa.tok_emb = tok_emb; pa.pos_emb = pos_emb;
        pa.blog = blog;
        pa.dx = dx; pa.dy = dy; pa.dq = dq; pa.dlogits = dlogits; pa.dao = dao; pa.dhid = dhid;
        pa.cur_tok = cur_tok; pa.eos_seen = eos_seen; pa.skv = skv; pa.ckv = ckv;
        pa.self_stride = (size_t)sB * Id * Tmax; pa.cross_stride = (size_t)sImg * N * Id;
        pa.tokens_out = tokens_out; pa.out_stride = out_stride; pa.logits_out = logits_out;
        pa.sample = sample_mode; pa.sample_topk = sample_topk; pa.inv_temp = 1.0f / sample_temp; pa.seed = sample_seed;
        pa.ctl = pctl; pa.stamps = pstamps;
        const char* stamp_file = knobs.pstamps ? knobs.pstamps_file.c_str() : nullptr;
        pa.stamp_step = stamp_file ? std::min(max_len - 1, 200) : -1;
        if (knobs.has_stagger) pa.stagger_ticks = knobs.stagger_ticks;
        pa.poll_sleep = 1;
        pa.poll_mode = 3;                           // default 3: scalar polls behind s_dcache_inv (persist.h: TeamSync::poll; -1.3 % per generate against vector polls)
        if (knobs.has_inject) pa.inject_fail = knobs.inject_fail; // tests: the give-up / fall-back path
        HIP_TRY(hipMemsetAsync(pctl, 0, sizeof(PersistCtl), s));
        if (stamp_file) HIP_TRY(hipMemsetAsync(pstamps, 0, sizeof(unsigned long long) * PS_TEAMS * PS_STAMP_RANKS * PS_MAX_STAGES * PS_STAMP_WORDS, s));
        if (D == 256) { if (int r = launch_persist<256, 8>(pa, s)) return r; }
        else if constexpr (sizeof(T) == 2) { if (int r = launch_persist<768, 12>(pa, s)) return r; }
        else return TXO_E_STATE;
        if (hipGetLastError() != hipSuccess) return fail(TXO_E_STATE, "persistent decode: the launch was refused");   // -> decode with launches
        HIP_TRY(hipMemcpyAsync(pctl_host, pctl, sizeof(PersistCtl), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const PersistCtl& c = *pctl_host;
        if (c.fail) { g_err = c.fail & 2u ? "persistent decode: a team spans more than one XCD" : "persistent decode: a hand-off timed out"; return TXO_E_STATE; }
        // GLOBAL eos break (decoder.py:115-116): the loop ends after the first position at which every row contains eos
        const int rpt = (B + PS_TEAMS - 1) / PS_TEAMS, nteams = (B + rpt - 1) / rpt;
        int steps = max_len;
        *broke = false;
        if (eos >= 0) {
            bool all = true; int last = 0;
            for (int k = 0; k < nteams; ++k) {
                const int nr = std::min(rpt, B - k * rpt);
                all = all && (int)c.eos_rows[k] >= nr;
                last = std::max(last, c.last_first_eos[k]);
            }
            if (all) steps = std::min(max_len, last + 1);
            *broke = all && last + 1 <= max_len;
        }
        // every team must have decoded every position that is returned (teams are not synchronised with each other; the kernel
        // lets a team stop only at or beyond the batch's last first-eos position -- checked here, never assumed)
        for (int k = 0; k < nteams; ++k)
            if (c.steps_run[k] < steps) { g_err = "persistent decode: a team stopped before the batch's last position"; return TXO_E_STATE; }
        if (stamp_file) dump_persist_stamps(stamp_file, nteams);
        *n_steps = steps;
        return 0;
    }
    void dump_persist_stamps(const char* file, int nteams) {
        const int ns = 7 * cfg.dec_layers + 2;
        std::vector<unsigned long long> h((size_t)PS_TEAMS * PS_STAMP_RANKS * PS_MAX_STAGES * PS_STAMP_WORDS);
        if (hipMemcpy(h.data(), pstamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
        FILE* f = fopen(file, "w");
        if (!f) return;
        static const char* names[7] = {"LN+qkv gemm", "self attention", "self out-proj+GLU+res", "cross attention (LN+q fused)",
                                       "cross out-proj+GLU+res", "LN+ffn-in+GeGLU", "ffn-out+res"};
        static const int ranks[PS_STAMP_RANKS] = {0, 10, 20, PS_TEAM_BLOCKS - 1};
        fprintf(f, "# persistent decode launch, ONE decode position, workgroups of rank 0 / 10 / 20 / 31 of every team.  Per stage, us:\n"
                   "#   poll  = polling the team counter for the previous stage's arrivals (weights / K panel already requested)\n"
                   "#   work  = barrier + read the previous stage's rows (sc1) + compute + issue the stores\n"
                   "#   drain = s_waitcnt vmcnt(0) of every wave + workgroup barrier\n"
                   "#   pub   = the arrival atomic, issued -> returned\n"
                   "#   end   = time of publication since the position's first stamp\n");
        for (int k = 0; k < nteams; ++k)
            for (int r = 0; r < PS_STAMP_RANKS; ++r) {
                const unsigned long long* d = &h[((size_t)k * PS_STAMP_RANKS + r) * PS_MAX_STAGES * PS_STAMP_WORDS];
                if (!d[2]) continue;
                const unsigned long long t0 = d[2];
                fprintf(f, "team %d rank %d: position span %.2f us\n", k, ranks[r], (d[(ns - 1) * PS_STAMP_WORDS + 4] - t0) / 100.0);
                if (k > 1) continue;                           // the per-stage table for two teams is enough
                for (int i = 0; i < ns; ++i) {
                    const unsigned long long* e = d + i * PS_STAMP_WORDS;
                    const char* nm = i < 7 * cfg.dec_layers ? names[i % 7] : (i == 7 * cfg.dec_layers ? "LNf+logits" : "argmax+append");
                    fprintf(f, "  L%-2d %-30s poll %5.2f  work %5.2f  drain %5.2f  pub %5.2f | end %7.2f", i < 7 * cfg.dec_layers ? i / 7 : -1, nm,
                            e[0] ? (e[1] - e[0]) / 100.0 : 0.0, e[1] ? (e[2] - e[1]) / 100.0 : 0.0, (e[3] - e[2]) / 100.0, (e[4] - e[3]) / 100.0,
                            (e[4] - t0) / 100.0);
                    const bool attn = i < 7 * cfg.dec_layers && (i % 7 == 1 || i % 7 == 3);
                    if (e[5] && e[1] && !attn)   // GEMM tile of the workgroup's first group: rows read + MFMAs | K reduction | epilogue
                        fprintf(f, " | tile: seen->mfma done %5.2f  reduce %5.2f  epilogue+stores %5.2f", (e[6] - e[1]) / 100.0, (e[7] - e[6]) / 100.0, (e[2] - e[7]) / 100.0);
                    if (e[5] && e[1] && attn)    // attention tile of the workgroup's first group: wait end -> last pass's scores and PV done | reductions + store
                        fprintf(f, " | tile: seen->panel consumed %5.2f  reduce+store %5.2f  (other group / barrier %5.2f)", ((long long)e[6] - (long long)e[1]) / 100.0,
                                (e[7] - e[6]) / 100.0, ((long long)e[2] - (long long)e[7]) / 100.0);
                    if (e[5] && e[0] && i > 0)   // before the poll: previous publication -> tile entry (stage set-up) -> poll begin (the tile's weight / bias / gamma requests)
                        fprintf(f, " | pre: setup %5.2f  requests %5.2f", ((long long)e[5] - (long long)(e - PS_STAMP_WORDS)[4]) / 100.0, ((long long)e[0] - (long long)e[5]) / 100.0);
                    fprintf(f, "\n");
                }
            }
        fclose(f);
    }

    // txo_generate / txo_generate_from_enc.  Per-row stop (stop_mode 1, a build extension: the reference has only the global break,
    // decoder.py:115-116): the decode below is the same loop with the same break -- rows are independent, so every row's tokens up to its first eos
    // are what the global-break run gives -- and every token BEHIND a row's first eos becomes cfg.pad (pad_after_eos_kernel); on the launch
    // path the finished rows also stop costing work (compact_lane).
    int generate(const float* img, const float* enc, int B, int C, int H, int W, int N, int max_len, int eos,
                 int64_t* tokens_out, int* n_steps, float* logits_out, hipStream_t s) override {
        int steps = 0;
        row_stop = false; last_compactions = 0;
        const int rc = generate_impl(img, enc, B, C, H, W, N, max_len, eos, tokens_out, &steps, logits_out, s);
        const bool compacted = last_compactions > 0;
        row_stop = false;
        if (rc) return rc;
        if (n_steps) *n_steps = steps;
        if (compacted) session = false;                           // the session's rows are no longer the batch's: a new decode must begin
        if (stop_mode == 1 && eos >= 0 && steps > 0) {
            hipLaunchKernelGGL(pad_after_eos_kernel, dim3((B + 255) / 256), dim3(256), 0, s, tokens_out, max_len, steps, B, eos, cfg.bos, cfg.pad);
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipGetLastError());
        }
        return 0;
    }
    bool row_stop = false;            // this generate compacts the live rows of its row ranges (launch path, stop_mode 1)
    // live rows of range li to its front; the range then launches `new_rows` rows (>= its live rows).  t1 = positions decoded so far.
    int compact_lane(int li, int new_rows, int t1, int eos) {
        Lane& ln = lanes[li];
        hipStream_t s = ln.stream;
        const size_t r0 = ln.b0;
        CompactArgs ca{ln.nb, new_rows, (int)r0, cur_tok + r0, eos_seen + r0, row_map + r0, cur_tok2 + r0, row_map2 + r0, cmoves + 2 * r0,
                       cinfo + 2 * li, st + li, eos};
        hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, ca);
        const int heads = cfg.dec_heads;
        auto move = [&](void* base, size_t outer_stride, size_t inner_stride, size_t row_stride, int outer_n, int inner_n, size_t len_elems) {
            MoveArgs ma{reinterpret_cast<unsigned char*>(base), outer_stride * sizeof(T), inner_stride * sizeof(T), row_stride * sizeof(T), inner_n,
                        (unsigned)(len_elems * sizeof(T) / 16), cmoves + 2 * r0, cinfo + 2 * li};
            if (ma.len16 == 0) return;
            hipLaunchKernelGGL(move_rows_kernel, dim3((ma.len16 + 255) / 256, outer_n * inner_n), dim3(256), 0, s, ma);
        };
        // self-attention history [Ld][2][rows*heads][Tmax][64]: positions 0 .. t1-1 of every head of every moved row
        move(skv + r0 * heads * Tmax * DH, (size_t)sB * Id * Tmax, (size_t)Tmax * DH, (size_t)heads * Tmax * DH, 2 * cfg.dec_layers, heads, (size_t)t1 * DH);
        // the cross attention's operand: the session's encoder rows (latent form) or the projected K/V panels [Ld][2][rows*heads][N][64]
        if (use_latent) move(enc_t + r0 * sN * D, 0, 0, (size_t)sN * D, 1, 1, (size_t)sN * D);
        else move(ckv + r0 * sN * Id, (size_t)sImg * sN * Id, 0, (size_t)sN * Id, 2 * cfg.dec_layers, 1, (size_t)sN * Id);
        ln.nb = new_rows;
        ++last_compactions;
        HIP_TRY(hipGetLastError());
        return 0;
    }
    int generate_impl(const float* img, const float* enc, int B, int C, int H, int W, int N, int max_len, int eos,
                 int64_t* tokens_out, int* n_steps, float* logits_out, hipStream_t s) {
        if (max_len < 1) return fail(TXO_E_INVALID, "max_len must be >= 1");
        // max_len > decoder.max_len: the reference slides its window (decoder.py:99-100).  The first Tmax positions decode with the
        // KV cache as always (n_pos of them; rows of the outputs are max_len apart); every further token re-runs its window of the
        // last Tmax tokens, positions re-indexed from 0, through ONE multi-position forward (prefill) -- generate_window below.
        const int n_pos = std::min(max_len, Tmax);
        // the window's multi-position forward has two preconditions: refuse BEFORE decoding anything
        if (max_len > Tmax && (V % 8 != 0 || (size_t)Tmax > (size_t)Bmax * Nmax))
            return fail(TXO_E_INVALID, "max_len exceeds the decoder's max_length and the sliding window's multi-position forward needs a vocabulary "
                                       "size that is a multiple of 8 and max_length <= max_batch * max_tokens (use decoder.generate's stepwise loop)");
        if (img) {
            if (int r = encode(img, B, C, H, W, eenc, s)) return r;
            enc = eenc; N = 1 + (H / 16) * (W / 16);
        }
        if (int r = begin_session(enc, B, N, eos, s, false)) return r;   // eos also decides whether the BOS column counts
        last_persist = false;
        if (persist_cooldown > 0) --persist_cooldown;
        if (persist_usable(B)) {
            int steps = 0;
            use_latent = false; lat_self = false;
            ensure_ckv(s);
            bool broke = false;
            const int pr = generate_persist(B, N, n_pos, max_len, eos, tokens_out, logits_out, &steps, &broke, s);
            if (pr == 0) {
                last_persist = true; persist_strikes = 0;
                // (an eos break exactly at position n_pos - 1 also leaves steps == n_pos: the window must not start then)
                if (!broke && max_len > n_pos) { if (int r = generate_window(B, n_pos, max_len, eos, tokens_out, logits_out, &steps, s)) return r; }
                if (n_steps) *n_steps = steps;
                return 0;
            }
            if (pr != TXO_E_STATE) return pr;
            ++persist_fallbacks;                                  // placement check or a bounded spin gave up: decode with launches
            if (++persist_strikes >= 2) {
                persist_cooldown = 64; persist_strikes = 0;
                fprintf(stderr, "[txo] persistent decode launch gave up twice in a row (%s): decoding with launches for the next 64 generates\n", g_err.c_str());
            }
            set_lanes(1, s);
            reset_lanes(s, eos);
        }
        // lanes: graphs + extra streams unless per-step logits were asked for or a debug/profiling mode is on
        // Measured on MI355X (B=64, 224x672, T=256): the step is bound by the GPU-side latency chain of its ~26
        // dependent launches, not by the host -- graph replay and 2-4 lanes give the same wall time as eager
        // single-stream launches (57.1 vs 58.0 / 58.2 ms) -- so both stay opt-in: TXO_GRAPH=1, TXO_LANES=n.
        // graph replay by default only for very small batches (B <= 4: there the host's enqueue rate bounds the step --
        // 34.2 vs 37.2 ms per generate at B = 1 -- from B = 8 on it is equal); TXO_GRAPH=1 / 0 forces it on / off
        use_latent = latent_ok && (lat_mode == 1 || (lat_mode < 0 && auto_latent(B)));
        // per-row stop with live-row compaction: inside the positional table, tokens only (a finished row's logits would be unspecified)
        row_stop = stop_mode == 1 && eos >= 0 && logits_out == nullptr && max_len <= Tmax && stop_every > 0 && !prof && !prof_cross;
        lat_self = use_latent && lat_self_ok() && !row_stop;      // (the z history is not moved by compact_lane)
        if (!use_latent) ensure_ckv(s);
        const bool want_graph = knobs.graph >= 0 ? knobs.graph != 0 : B <= 4;
        // per-row stop replays captured steps: as the ranges shrink a position's launches take less time than the host needs to enqueue them
        // (68 launches for two ranges: ~230 us per position on the host against 130 us on the device at 40 % of the rows), and a step
        // per row count (multiples of 16) is captured once and kept (lane_graph).  TXO_STOP_GRAPH=0: eager launches.
        const bool stop_graph = row_stop && !sample_mode && stop_graph_on;
        const bool eager = logits_out != nullptr || g_dbg || sample_mode || !(want_graph || stop_graph);
        // two row ranges on two streams for a WIDE decoder at >= 256 rows (BASELINE cfg 4): one range's latency-bound projection launches
        // run beside the other's HBM-bound attention launches (816 -> 834 images/s; four ranges: 765).  Greedy and sampled alike: a draw is keyed
        // by (seed; row of the batch, position), not by the range (step.h: StepArgs::row0)
        // (also the narrow decoder's bf16 decode beyond the persistent launch's range, see persist_usable)
        int want = (sizeof(T) == 2 && !prof && !prof_cross &&       // (profiling times whole-batch launches)
                    ((D >= 512 && B >= 256) || (D < 512 && B > PERSIST_MAX_BF16_GREEDY))) ? 2 : 1;
        // in latent form the second range pays from ~224 rows on (160: 66.6 ms with one range vs 71.7 with two, 192: 70.6 vs 72.2, 256: 82.0 vs 77.9;
        // K/V form: two ranges from 129 on, 160: 70.9 vs 66.3)
        if (want == 2 && use_latent && D < 512 && B < 224) want = 1;
        if (knobs.lanes > 0) want = std::min(knobs.lanes, max_lanes);
        if (B < 32) want = 1;
        if (want == 2 && tune_lanes && !lanes_tuned) { if (int r = tune_lane_streams()) return r; }
        set_lanes(want, s);
        last_ranges = n_lanes;
        reset_lanes(s, eos);
        bool use_graph = !eager && !prof && !prof_cross;   // event-carrying launches cannot be captured
        if (use_graph) for (int i = 0; i < n_lanes; ++i) if (int r = lane_graph(i, eos)) return r;
        if (n_lanes > 1) {
            HIP_TRY(hipEventRecord(ev_fork, s));
            for (int i = 0; i < n_lanes; ++i) if (lanes[i].stream != s) HIP_TRY(hipStreamWaitEvent(lanes[i].stream, ev_fork, 0));
        }
        int64_t* tdst = use_graph ? tok_buf : tokens_out;
        const int tstride = use_graph ? Tmax : max_len;               // rows of tokens_out are max_len apart (only n_pos positions are decoded here)
        // GLOBAL eos break (decoder.py:115-116): the device records done_flag[t]; the host looks at the flags of every 32 steps.
        // The look must not drain the stream: the flags of a chunk are copied to pinned memory behind the chunk, a few more
        // steps are enqueued, and only then the host waits for that copy -- the GPU keeps running those steps meanwhile (a
        // full stream sync at every chunk left it idle for the host's wake-up + re-enqueue time: 1.2 ms per 256 steps).  After
        // a break at most AHEAD extra steps have run; their tokens lie beyond `steps` and are never returned.
        const int CHUNK = 32, AHEAD = 4;
        int* flags = flags_host;                                   // pinned, allocated in init()
        int steps = n_pos;
        bool broke = false;                                        // the GLOBAL eos break fired inside the positional table
        look_failed = false;
        int pend_lo = -1, pend_hi = -1;                            // chunk whose flags are in flight to the host
        int live_pend = -1;                                        // per-row stop: position behind which the ranges' finished-row counts were requested
        auto look = [&]() -> bool {                                // wait for the pending chunk's flags; true = all rows done
            for (int i = 0; i < n_lanes; ++i) HIP_TRY_B(hipEventSynchronize(ev_flags[i]));
            for (int k = pend_lo; k <= pend_hi; ++k) {
                bool all = true;
                for (int i = 0; i < n_lanes; ++i) all = all && flags[(size_t)i * Tmax + k];
                if (all) { steps = k + 1; broke = true; pend_lo = -1; return true; }
            }
            pend_lo = -1;
            return false;
        };
        const char* stamp_file = knobs.stamps ? knobs.stamps_file.c_str() : nullptr;
        if (stamp_file && !stamp_buf) { if (int r = dalloc(&stamp_buf, (size_t)STAMP_KERNELS * STAMP_BLOCKS * 3)) return r; }
        const int stamp_step = stamp_file ? std::min(n_pos - 1, 200) : -1;
        auto decode_loop = [&]() -> int {
        for (int t = 0; t < n_pos; ++t) {
            if (t == stamp_step) { HIP_TRY(hipMemsetAsync(stamp_buf, 0, sizeof(unsigned long long) * STAMP_KERNELS * STAMP_BLOCKS * 3, s)); stamp_slot = 0; stamp_names.clear(); }
            else if (stamp_slot >= 0) dump_stamps(stamp_file, s);
            for (int i = 0; i < n_lanes; ++i) {
                if (use_graph) HIP_TRY(hipGraphLaunch(lanes[i].exec, lanes[i].stream));
                else if (int r2 = enqueue_step(lanes[i].stream, i, tdst, tstride, logits_out, eos, nullptr, t)) return r2;
            }
            if (eos < 0) continue;
            if (row_stop) {
                // Live-row compaction.  The host only needs an UPPER bound of a range's live rows to size its launches, and live rows only
                // decrease: the finished-row counter of a range is copied to pinned memory behind a step, AHEAD more steps are enqueued,
                // and only then the host reads it (never draining the stream, like the done flags below).  compact_scan_kernel works on the
                // state of the moment it runs; slots between its live count and the host's bound are filler rows that count as finished.
                if (live_pend >= 0 && t == live_pend + AHEAD) {
                    for (int i = 0; i < n_lanes; ++i) HIP_TRY(hipEventSynchronize(ev_live[i]));
                    for (int i = 0; i < n_lanes; ++i) {
                        int bound = lanes[i].nb - live_host[i];
                        if (use_graph) bound = std::min(lanes[i].nb, (bound + 15) & ~15);      // a captured step per multiple of 16 rows
                        if (bound >= 1 && lanes[i].nb - bound >= stop_gain) {
                            if (int r2 = compact_lane(i, bound, t + 1, eos)) return r2;
                            if (use_graph) { if (int r2 = lane_graph(i, eos)) return r2; }
                        }
                    }
                    live_pend = -1;
                }
                if (live_pend < 0 && (t + 1) % stop_every == 0 && t + 1 + AHEAD < n_pos) {
                    for (int i = 0; i < n_lanes; ++i) {
                        HIP_TRY(hipMemcpyAsync(live_host + i, &st[i].rows_with_eos, sizeof(int), hipMemcpyDeviceToHost, lanes[i].stream));
                        HIP_TRY(hipEventRecord(ev_live[i], lanes[i].stream));
                    }
                    live_pend = t;
                }
            }
            if (pend_lo >= 0 && (t == pend_hi + AHEAD || t + 1 == n_pos)) { if (look()) break; if (look_failed) return TXO_E_HIP; }
            if ((t + 1) % CHUNK == 0 || t + 1 == n_pos) {
                const int lo = (t / CHUNK) * CHUNK;
                for (int i = 0; i < n_lanes; ++i) {
                    HIP_TRY(hipMemcpyAsync(flags + (size_t)i * Tmax + lo, done_flag + (size_t)i * Tmax + lo,
                                           sizeof(int) * (t + 1 - lo), hipMemcpyDeviceToHost, lanes[i].stream));
                    HIP_TRY(hipEventRecord(ev_flags[i], lanes[i].stream));
                }
                pend_lo = lo; pend_hi = t;
                if (t + 1 == n_pos) { (void)look(); if (look_failed) return TXO_E_HIP; }
            }
        }
        return 0;
        };
        if (int r = decode_loop()) return abandon_lanes(s, r);
        // join the lanes back into the caller's stream
        for (int i = 0; i < n_lanes; ++i) {
            if (lanes[i].stream == s) continue;
            HIP_TRY(hipEventRecord(ev_join[i], lanes[i].stream));
            HIP_TRY(hipStreamWaitEvent(s, ev_join[i], 0));
        }
        if (use_graph)
            HIP_TRY(hipMemcpy2DAsync(tokens_out, sizeof(int64_t) * max_len, tok_buf, sizeof(int64_t) * Tmax,
                                     sizeof(int64_t) * n_pos, B, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipGetLastError());
        set_lanes(1, s);
        if (!broke && max_len > n_pos) { if (int r = generate_window(B, n_pos, max_len, eos, tokens_out, logits_out, &steps, s)) return r; }
        if (n_steps) *n_steps = steps;
        return 0;
    }

    // The reference beyond its positional table (decoder.py:97-116 with :99-100 active): token i >= Tmax comes from the window
    // output[:, -Tmax:] = tokens i-Tmax .. i-1 at positions 0 .. Tmax-1 (BOS has left the window), the whole window through the
    // decoder, the last position's logits.  Nothing cached survives the shift, so every such token costs one multi-position
    // forward of Tmax rows per image (prefill) + the single-position final LayerNorm / logits / token selection of the step path.
    // The GLOBAL eos test still looks at the whole output (:115), i.e. the per-row "seen" state carries over.
    int generate_window(int B, int n_pos, int max_len, int eos, int64_t* tokens_out, float* logits_out, int* steps, hipStream_t s) {
        set_lanes(1, s);
        lanes[0].stream = s;
        // the launch path's eos bookkeeping, rebuilt from the tokens decoded so far (a persistent launch keeps its own)
        HIP_TRY(hipMemsetAsync(st, 0, sizeof(StepState), s));
        hipLaunchKernelGGL(rebuild_eos_state_kernel, dim3((B + 255) / 256), dim3(256), 0, s, tokens_out, max_len, n_pos, B, eos, cfg.bos,
                           eos_seen, &st[0].rows_with_eos);
        int* flag = done_flag;                                    // ONE flag, rewritten per token (commit_token indexes it with the token index)
        for (int i = n_pos; i < max_len; ++i) {
            if (int r = prefill(tokens_out + (i - Tmax), max_len, Tmax, nullptr, dlogits, s)) return r;
            hipLaunchKernelGGL(set_position_kernel, dim3(1), dim3(1), 0, s, st, i);
            StepArgs sa{dlogits, V, B, cur_tok, tokens_out, max_len, logits_out, st, eos_seen, flag - i, eos, sample_topk, 1.0f / sample_temp, sample_seed, 0};
            if (sample_mode) launch_sample_step(s, B, sa);
            else hipLaunchKernelGGL(argmax_step_kernel, dim3(B), dim3(64), 0, s, sa);
            *steps = i + 1;
            if (eos >= 0) {                                       // slow path: one host look per token
                HIP_TRY(hipMemcpyAsync(flags_host, flag, sizeof(int), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                if (flags_host[0]) break;
            }
        }
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipGetLastError());
        return 0;
    }

    int generate_beam(const float* img, const float* enc, int B, int C, int H, int W, int N, int beams, int max_len, int eos,
                      int64_t* tokens_out, float* scores_out, int64_t* all_tokens_out, int* n_steps, hipStream_t s) override {
        if (beams < 1 || beams > 8) return fail(TXO_E_INVALID, "beams must be in [1, 8]");
        if (max_len < 1 || max_len > Tmax)
            return fail(TXO_E_INVALID, "max_len exceeds the decoder's max_length (a KV cache cannot slide the window)");
        if (Tmax > 1024) return fail(TXO_E_INVALID, "beam search supports max_length <= 1024");
        const int rows = B * beams;
        if (B < 1 || rows > Bmax || rows > 32767) return fail(TXO_E_INVALID, "images * beams exceeds engine max_batch");
        if (img) {
            if (int r = encode(img, B, C, H, W, eenc, s)) return r;
            enc = eenc; N = 1 + (H / 16) * (W / 16);
        }
        if (int r = begin_session(enc, B, N, eos, s, false)) return r;
        use_latent = latent_ok && (lat_mode == 1 || (lat_mode < 0 && auto_latent(rows)));
        lat_self = use_latent && lat_self_ok();
        if (!use_latent) ensure_ckv(s);                               // cross K/V of the B images
        sB = rows; sImg = B;                                          // decode rows are (image, beam) slots
        last_persist = false;
        set_lanes(1, s);
        // Two row ranges on two streams from 256 rows on, as generate() does beyond 128 images: one range's latency-bound projection
        // launches run beside the other's HBM-bound attention launches.  A range is a whole number of IMAGES (beam_select_kernel ranks
        // an image's k beams together; self-attention slots are range-local); every range has its own step state and done flags.
        int want = (rows >= 256 && B >= 2 && !prof && !prof_cross && !g_dbg) ? 2 : 1;
        if (knobs.lanes > 0) want = std::max(1, std::min(std::min(knobs.lanes, max_lanes), B));
        if (want == 2 && tune_lanes && !lanes_tuned) { if (int r = tune_lane_streams()) { sB = B; return r; } }
        if (want > 1) {
            n_lanes = want;
            int img0 = 0;
            for (int i = 0; i < want; ++i) {
                const int ni = B / want + (i < B % want ? 1 : 0);
                lanes[i].b0 = img0 * beams; lanes[i].nb = ni * beams; lanes[i].stream = i == 0 ? s : lanes[i].own;
                img0 += ni;
            }
            if (lanes[0].own) lanes[0].stream = lanes[0].own;          // (the measured pair: tune_lane_streams)
        }
        last_ranges = n_lanes;
        const int n = std::max(rows, Tmax);
        hipLaunchKernelGGL(beam_reset_kernel, dim3((n + 255) / 256), dim3(256), 0, s, st, cur_tok, bscore, bfin, done_flag, rows,
                           beams, Tmax, cfg.bos);
        for (int i = 1; i < n_lanes; ++i)                             // the other ranges' step state and flags (no rows: they were reset above)
            hipLaunchKernelGGL(beam_reset_kernel, dim3((Tmax + 255) / 256), dim3(256), 0, s, st + i, cur_tok, bscore, bfin,
                               done_flag + (size_t)i * Tmax, 0, beams, Tmax, cfg.bos);
        if (n_lanes > 1) {
            HIP_TRY(hipEventRecord(ev_fork, s));
            for (int i = 0; i < n_lanes; ++i) if (lanes[i].stream != s) HIP_TRY(hipStreamWaitEvent(lanes[i].stream, ev_fork, 0));
        }
        // same non-draining eos look as generate(): once every beam is finished further steps only repeat eos at no cost
        // (beam_select_kernel), so the few steps enqueued ahead of the look change neither scores nor slots.  A range's flag stays
        // set once set, so the batch is done at the first position at which every range's flag is set.
        int* flags = flags_host;
        int steps = max_len, cur = 0;
        const int CHUNK = 32, AHEAD = 4;
        int pend_lo = -1, pend_hi = -1;
        bool stop = false, look_err = false;
        auto look = [&]() {
            for (int i = 0; i < n_lanes; ++i) if (hipEventSynchronize(ev_flags[i]) != hipSuccess) { look_err = true; return; }
            for (int k2 = pend_lo; k2 <= pend_hi && !stop; ++k2) {
                bool all = true;
                for (int i = 0; i < n_lanes; ++i) all = all && flags[(size_t)i * Tmax + k2];
                if (all) { steps = k2 + 1; stop = true; }
            }
            pend_lo = -1;
        };
        auto beam_loop = [&]() -> int {
        for (int t = 0; t < max_len && !stop; ++t) {
            BeamCtx bm{beams, bpath[cur], bpath[cur ^ 1]};
            for (int i = 0; i < n_lanes; ++i)
                if (int r2 = enqueue_step(lanes[i].stream, i, nullptr, 0, nullptr, eos, &bm, t)) return r2;
            cur ^= 1;
            if (eos < 0) continue;
            const bool last = t + 1 == max_len;
            if ((t + 1) % CHUNK == 0 || last) {
                if (pend_lo >= 0) {                                   // (only when CHUNK <= AHEAD; kept for safety)
                    look();
                    if (look_err) return fail(TXO_E_HIP, "beam search: waiting for the done flags failed");
                    if (stop) break;
                }
                const int lo = (t / CHUNK) * CHUNK;
                for (int i = 0; i < n_lanes; ++i) {
                    HIP_TRY(hipMemcpyAsync(flags + (size_t)i * Tmax + lo, done_flag + (size_t)i * Tmax + lo, sizeof(int) * (t + 1 - lo),
                                           hipMemcpyDeviceToHost, lanes[i].stream));
                    HIP_TRY(hipEventRecord(ev_flags[i], lanes[i].stream));
                }
                pend_lo = lo; pend_hi = t;
            }
            if (pend_lo >= 0 && (t == pend_hi + AHEAD || last)) {
                look();
                if (look_err) return fail(TXO_E_HIP, "beam search: waiting for the done flags failed");
            }
        }
        return 0;
        };
        if (int r = beam_loop()) { sB = B; return abandon_lanes(s, r); }
        for (int i = 0; i < n_lanes; ++i) {                           // join the ranges back into the caller's stream
            if (lanes[i].stream == s) continue;
            HIP_TRY(hipEventRecord(ev_join[i], lanes[i].stream));
            HIP_TRY(hipStreamWaitEvent(s, ev_join[i], 0));
        }
        set_lanes(1, s);
        // backtrack every beam into tok_buf rows, then hand out the best beam (slot 0: selection order is by score)
        hipLaunchKernelGGL(beam_backtrack_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, bparent, btok, Bmax, steps, rows,
                           tok_buf, Tmax);
        HIP_TRY(hipMemcpy2DAsync(tokens_out, sizeof(int64_t) * max_len, tok_buf, sizeof(int64_t) * Tmax * beams,
                                 sizeof(int64_t) * steps, B, hipMemcpyDeviceToDevice, s));
        if (all_tokens_out)
            HIP_TRY(hipMemcpy2DAsync(all_tokens_out, sizeof(int64_t) * max_len, tok_buf, sizeof(int64_t) * Tmax,
                                     sizeof(int64_t) * steps, rows, hipMemcpyDeviceToDevice, s));
        if (scores_out) HIP_TRY(hipMemcpyAsync(scores_out, bscore, sizeof(float) * rows, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipGetLastError());
        sB = B;
        if (n_steps) *n_steps = steps;
        return 0;
    }

    int query(int what, int64_t* out) override {
        if (what == TXO_Q_LAST_PERSISTENT) *out = last_persist ? 1 : 0;
        else if (what == TXO_Q_PERSIST_FALLBACKS) *out = persist_fallbacks;
        else if (what == TXO_Q_LAST_ROW_RANGES) *out = last_persist ? 1 : last_ranges;
        else if (what == TXO_Q_LAST_LATENT) *out = (!last_persist && use_latent) ? 1 : 0;
        else if (what == TXO_Q_RELOAD_KNOBS) { knobs.read(); *out = 0; }
        else if (what == TXO_Q_LAST_COMPACTIONS) *out = last_compactions;
        else return fail(TXO_E_INVALID, "unknown query");
        return 0;
    }
    int profile_enable(int on) override {
        prof = on == 1; prof_cross = on == 2; prof_persist = on == 3;
        ev_cross.clear(); ev_enc.clear(); ev_step.clear(); ev_persist.clear(); pool.used = 0;
        if (prof_persist) while (pool.ev.size() < 64) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); pool.ev.push_back(e); }
        if (prof_cross) {           // events for 16 generate() calls, created now
            const size_t want = (size_t)16 * 2 * cfg.dec_layers * Tmax / 4;
            while (pool.ev.size() < want) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); pool.ev.push_back(e); }
        }
        return 0;
    }
    int profile_read(int kind, double* avg_ms, int64_t* count) override {
        auto& v = kind == 0 ? ev_cross : (kind == 1 ? ev_enc : (kind == 2 ? ev_step : ev_persist));
        double tot = 0; int64_t n = 0;
        for (auto& p : v) {
            float ms = 0;
            if (hipEventSynchronize(p.second) != hipSuccess) continue;
            if (hipEventElapsedTime(&ms, p.first, p.second) != hipSuccess) continue;
            tot += ms; ++n;
        }
        if (avg_ms) *avg_ms = n ? tot / n : 0.0;
        if (count) *count = n;
        return 0;
    }
};

static int validate(const txo_config& c) {
    if (c.canvas_h <= 0 || c.canvas_h % 16 || c.canvas_w <= 0 || c.canvas_w % 16)
        return fail(TXO_E_INVALID, "canvas height/width must be positive multiples of 16");
    if (c.embed != TXO_EMBED_PATCH && c.embed != TXO_EMBED_HYBRID) return fail(TXO_E_INVALID, "embed must be TXO_EMBED_PATCH or TXO_EMBED_HYBRID");
    if (c.embed == TXO_EMBED_HYBRID && c.in_channels != 1)
        return fail(TXO_E_INVALID, "the hybrid ResNetV2 embedder takes single-channel images");
    if (c.embed_dim < 64 || c.embed_dim % 64 || c.embed_dim > 768)
        return fail(TXO_E_INVALID, "embed_dim must be a multiple of 64 in [64, 768]");
    if (c.enc_heads < 1 || c.dec_heads < 1 || c.enc_layers < 1 || c.dec_layers < 1)
        return fail(TXO_E_INVALID, "heads and layers must be >= 1");
    if (c.enc_exp < 1 || c.dec_exp < 1 || (c.enc_exp * c.embed_dim) % 64 || (c.dec_exp * c.embed_dim) % 64)
        return fail(TXO_E_INVALID, "FFN width must be a multiple of 64");
    if (c.in_channels < 1 || c.vocab < 2 || c.max_len < 1) return fail(TXO_E_INVALID, "bad in_channels / vocab / max_len");
    if (c.bos < 0 || c.bos >= c.vocab) return fail(TXO_E_INVALID, "bos token outside the vocabulary");
    if (c.max_batch < 1 || c.max_batch > 65535) return fail(TXO_E_INVALID, "max_batch must be in [1, 65535]");
    if (c.max_tokens < 0 || c.max_tokens > 1 + (c.canvas_h / 16) * (c.canvas_w / 16))
        return fail(TXO_E_INVALID, "max_tokens exceeds 1 + canvas_h*canvas_w/256");
    if (c.dtype != TXO_F32 && c.dtype != TXO_BF16) return fail(TXO_E_INVALID, "dtype must be TXO_F32 or TXO_BF16");
    return 0;
}

}  // namespace txo

using namespace txo;

struct txo_engine { std::unique_ptr<EngineBase> impl; };

extern "C" {

int txo_engine_create(const txo_config* cfg, txo_engine** out) {
    if (!cfg || !out) return fail(TXO_E_INVALID, "null argument");
    if (int r = validate(*cfg)) return r;
    std::unique_ptr<EngineBase> impl;
    int r;
    if (cfg->dtype == TXO_F32) { auto* e = new Engine<float>(); e->cfg = *cfg; impl.reset(e); r = e->init(); }
    else { auto* e = new Engine<bf16>(); e->cfg = *cfg; impl.reset(e); r = e->init(); }
    if (r) return r;
    *out = new txo_engine{std::move(impl)};
    return 0;
}

void txo_engine_destroy(txo_engine* e) { delete e; }

int txo_engine_set_weight(txo_engine* e, const char* key, const float* data, const int64_t* shape, int32_t ndim) {
    if (!e || !key || !data || !shape || ndim < 1 || ndim > 4) return fail(TXO_E_INVALID, "bad argument");
    if (e->impl->ready) return fail(TXO_E_STATE, "weights already finalized");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { if (shape[i] < 1) return fail(TXO_E_INVALID, "bad shape"); t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(data, data + n);
    e->impl->host[key] = std::move(t);
    return 0;
}

int txo_engine_finalize_weights(txo_engine* e) {
    if (!e) return fail(TXO_E_INVALID, "null engine");
    int r = e->impl->finalize();
    if (r == TXO_E_STATE && g_err.empty()) g_err = "missing weights";
    return r;
}

int txo_encode(txo_engine* e, const float* img, int32_t B, int32_t C, int32_t H, int32_t W, float* enc_out, void* stream) {
    if (!e || !img || !enc_out) return fail(TXO_E_INVALID, "null argument");
    return e->impl->encode(img, B, C, H, W, enc_out, (hipStream_t)stream);
}

int txo_decode_begin(txo_engine* e, const float* enc, int32_t B, int32_t N, void* stream) {
    if (!e || !enc) return fail(TXO_E_INVALID, "null argument");
    return e->impl->decode_begin(enc, B, N, e->impl->cfg.eos, (hipStream_t)stream);
}

int txo_decode_step(txo_engine* e, const int64_t* tok_in, int32_t t, float* logits_out, int64_t* tok_out, void* stream) {
    if (!e) return fail(TXO_E_INVALID, "null engine");
    return e->impl->decode_step(tok_in, t, logits_out, tok_out, (hipStream_t)stream);
}

int txo_decode_prefill(txo_engine* e, const int64_t* tokens, int32_t t, float* logits_out, void* stream) {
    if (!e || !tokens) return fail(TXO_E_INVALID, "null argument");
    return e->impl->decode_prefill(tokens, t, logits_out, (hipStream_t)stream);
}

int txo_decode_set_key_mask(txo_engine* e, const uint8_t* mask, int32_t cols, void* stream) {
    if (!e) return fail(TXO_E_INVALID, "null engine");
    return e->impl->decode_set_key_mask(mask, cols, (hipStream_t)stream);
}

int txo_generate(txo_engine* e, const float* img, int32_t B, int32_t C, int32_t H, int32_t W, int32_t max_len,
                 int32_t eos, int64_t* tokens_out, int32_t* n_steps, float* logits_out, void* stream) {
    if (!e || !img || !tokens_out) return fail(TXO_E_INVALID, "null argument");
    return e->impl->generate(img, nullptr, B, C, H, W, 0, max_len, eos, tokens_out, n_steps, logits_out, (hipStream_t)stream);
}

int txo_generate_from_enc(txo_engine* e, const float* enc, int32_t B, int32_t N, int32_t max_len, int32_t eos,
                          int64_t* tokens_out, int32_t* n_steps, float* logits_out, void* stream) {
    if (!e || !enc || !tokens_out) return fail(TXO_E_INVALID, "null argument");
    return e->impl->generate(nullptr, enc, B, 0, 0, 0, N, max_len, eos, tokens_out, n_steps, logits_out, (hipStream_t)stream);
}

int txo_generate_beam(txo_engine* e, const float* img, int32_t B, int32_t C, int32_t H, int32_t W, int32_t beams, int32_t max_len,
                      int32_t eos, int64_t* tokens_out, float* scores_out, int64_t* all_tokens_out, int32_t* n_steps, void* stream) {
    if (!e || !img || !tokens_out) return fail(TXO_E_INVALID, "null argument");
    return e->impl->generate_beam(img, nullptr, B, C, H, W, 0, beams, max_len, eos, tokens_out, scores_out, all_tokens_out, n_steps,
                                  (hipStream_t)stream);
}

int txo_set_sampling(txo_engine* e, int32_t mode, int32_t topk, float temp, uint64_t seed) {
    if (!e) return fail(TXO_E_INVALID, "null engine");
    if (mode != 0 && mode != 1) return fail(TXO_E_INVALID, "sampling mode must be 0 (greedy) or 1 (top-k / temperature / multinomial)");
    if (mode == 1 && (topk < 1 || !(temp > 0.f))) return fail(TXO_E_INVALID, "sampling needs topk >= 1 and temp > 0");
    e->impl->sample_mode = mode; e->impl->sample_topk = topk; e->impl->sample_temp = mode ? temp : 1.f; e->impl->sample_seed = seed;
    return 0;
}

int txo_set_stop_mode(txo_engine* e, int32_t mode) {
    if (!e) return fail(TXO_E_INVALID, "null engine");
    if (mode != TXO_STOP_GLOBAL && mode != TXO_STOP_ROW) return fail(TXO_E_INVALID, "stop mode must be TXO_STOP_GLOBAL or TXO_STOP_ROW");
    e->impl->stop_mode = mode;
    return 0;
}
int txo_profile_enable(txo_engine* e, int32_t on) { return e ? e->impl->profile_enable(on) : fail(TXO_E_INVALID, "null engine"); }
int txo_profile_read(txo_engine* e, int32_t kind, double* avg_ms, int64_t* count) {
    return e ? e->impl->profile_read(kind, avg_ms, count) : fail(TXO_E_INVALID, "null engine");
}

int txo_engine_query(txo_engine* e, int32_t what, int64_t* out) {
    if (!e || !out) return fail(TXO_E_INVALID, "null argument");
    return e->impl->query(what, out);
}

const char* txo_last_error(void) { return g_err.c_str(); }
const char* txo_version(void) { return "texocr-amd 0.1 (gfx950)"; }

}  // extern "C"
