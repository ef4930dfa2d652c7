// The whole greedy decode loop of AutoRegressiveDecoder.generate (reference model/decoder.py:97-116) as ONE persistent
// launch: every decode position runs the same ~30 stages as the launch-per-stage path (engine.hip: enqueue_step), built
// from the same tile functions (dec_gemm.h, dec_attn.h) and therefore bit-identical to it, but the dependency between two
// stages is an in-launch hand-off instead of a kernel boundary.
//
// Why: at batch 64 a step is a chain of 30 dependent launches whose ~3.5 us each (boundary + launch ramp + cold operand
// fetch) is 105 of its 173 us, while its HBM traffic needs ~58 us (DESIGN_HISTORY.md section 5).  Inside one launch a stage
//   * requests its weights BEFORE it waits for the previous stage (they do not depend on the chain), and
//   * hands its rows over through the L2 it shares with its consumers (probes/team_seam.hip: 1.5 us per stage against
//     3.1 us for an empty dependent launch).
//
// Geometry: 256 workgroups of 512 threads, one per CU, dealt into 8 TEAMS of 32 (team = blockIdx % 8: the dispatcher
// places blocks b and b+8 on one XCD, which the kernel VERIFIES with HW_REG_XCC_ID before it trusts it).  A team owns a
// contiguous range of batch rows for the whole decode; rows of different teams never interact (the reference's only
// cross-row operation is the GLOBAL eos test, decoder.py:115-116, handled below), so there is no chip-wide barrier
// anywhere.  A workgroup is two 256-thread groups; a stage's tiles / (image, head) pairs are dealt over the team's 64
// groups, first groups first (a stage of <= 32 tiles runs one per CU), and a group without a tile only keeps the
// workgroup's barriers.  Resident in LDS for the whole launch: the FFN-out weight fragments of the workgroup's tile
// (64 KB) and the LayerNorm parameters (4 KB).
//
// Hand-off (one per stage, all-to-all inside the team): every storing wave drains its stores (s_waitcnt vmcnt(0)),
// workgroup barrier, ONE plain store of the stage number into the workgroup's word of the team's flag line; a consumer
// polls that line from one wave (scalar loads behind s_dcache_inv by default, or sc1 vector loads: TeamSync), workgroup barrier, then
// reads the rows with sc1 loads (never from its CU's L1, which other CUs' stores do not refresh; the XCD's L2 -- the coherence point
// of its 32 CUs -- serves them).  Stores are plain (they stay
// in that L2).  This is valid only while producer and consumer share an XCD: each workgroup ORs its XCC id into a
// per-team mask, and after the first hand-off every workgroup checks that its team's mask has ONE bit; otherwise the
// launch gives up (ctl.fail) and the engine decodes with launches.  All spins are bounded (4 ms of the real-time counter, fail bit 0;
// the fail word is read with every scalar poll, so one team's give-up ends every team within a hand-off).
//
// GLOBAL eos break: a row handler counts first occurrences of eos per team; a team whose rows all contain eos sets its
// bit in a chip-wide mask (memory-side atomic).  Teams are not synchronised with each other, so once the mask is full a team
// stops only when it has decoded the position at which the LAST row of the batch first produced eos (the maximum over the teams'
// last_first_eos words, final once their bits are set).  The host derives n_steps from those words exactly as the reference's
// (output == eos).any(1).all() would, checks that every team ran that far, and never returns positions beyond it.
// The position's last stage picks the token: arg-max, or the reference's sampler (step.h: sample_row; same draws as the launch path).
#pragma once
#include "dec_attn.h"
#include "dec_gemm.h"
#include "step.h"

// build-time experiment switches (probes/ab_libs.sh): weights requested one stage ahead into registers
#ifndef TXO_PS_AHEAD
#define TXO_PS_AHEAD false
#endif
// per-tile stamps of the attention stages in the TXO_PSTAMPS dump (diagnostic build: -DTXO_PS_ATTN_STAMPS=true)
#ifndef TXO_PS_ATTN_STAMPS
#define TXO_PS_ATTN_STAMPS false
#endif

namespace txo {

constexpr bool PS_ATTN_STAMPS = TXO_PS_ATTN_STAMPS;
constexpr int PS_TEAMS = 8, PS_TEAM_BLOCKS = 32, PS_THREADS = 512, PS_MAXLD = 8;
constexpr int PS_MAX_STAGES = 7 * PS_MAXLD + 2, PS_STAMP_RANKS = 4, PS_STAMP_WORDS = 8;   // 5 hand-off ticks + the tile's {entry, operands consumed, reduced}

struct PersistCtl {                       // zeroed before every launch
    unsigned flags[PS_TEAMS][32];         // arrival flags: one 128-byte line per team, one word per workgroup (TeamSync)
    unsigned xcc_mask[PS_TEAMS];          // OR of (1 << XCC id) over the team's workgroups
    unsigned eos_rows[PS_TEAMS];          // rows of the team that contain eos so far
    int last_first_eos[PS_TEAMS];         // position at which a row of the team FIRST produced eos, maximum over rows
    unsigned stop[PS_TEAMS];              // leave the loop (every active team is done)
    int steps_run[PS_TEAMS];
    unsigned done_mask;                   // teams whose rows all contain eos
    unsigned fail;                        // bit 0: a spin timed out; bit 1: a team spans more than one XCD
    unsigned pad[2];
};
static_assert(sizeof(PersistCtl) % 16 == 0, "memset block");

template <typename T> struct PersistLayer {
    const T* wqkv; const T* wo_s; const float* bo_s;          // self attention
    const T* wq_c; const T* wo_c; const float* bo_c;          // cross attention (K/V projected once per generate)
    const T* w1; const float* b1; const T* w2; const float* b2;
};

template <typename T> struct PersistArgs {
    int B, N, V, Ld, Tmax, max_len, eos, bos;
    PersistLayer<T> L[PS_MAXLD];
    const float *gamma, *beta, *gamma_f, *beta_f, *tok_emb, *pos_emb, *blog;
    const T* wlog;
    float *dx, *dy, *dq, *dlogits; T *dao, *dhid;
    int64_t* cur_tok; int* eos_seen;
    T *skv, *ckv; size_t self_stride, cross_stride;           // per (layer, k|v) plane
    int64_t* tokens_out; int out_stride; float* logits_out;
    PersistCtl* ctl;
    int poll_sleep;                                           // TeamSync: 64-clock sleeps between two polls of the flag line
    int w_tiled;                                              // the projections' weights (wqkv, wo_s, wo_c, w1, w2, wlog) are the TILED copies (dec_gemm.h); wq_c stays row-major (dec_attn.h reads it)
    int sample; int sample_topk; float inv_temp; unsigned long long seed;   // sample != 0: the reference's sampler ends a position (step.h: sample_row) instead of the arg-max
    int early_mask;                                           // bit 0 / 1: the POLLING wave also requests its self-attention history / its cross K panel before the wait (its poll then returns behind them)
    int poll_mode;                                            // TeamSync::poll: 0 vector sc1 loads, 1 scalar glc loads, 2 the same behind s_dcache_inv, 3 (default) s_dcache_inv + plain scalar loads
    int inject_fail;                                          // test hook: position at which team 0 reports a hand-off time-out (0 = never)
    int stagger_ticks;                                        // experiment: team k starts k * this many 10-ns ticks late (desynchronises the teams' HBM phases)
    unsigned long long* stamps; int stamp_step;               // diagnostic: [team][PS_STAMP_RANKS][stage][5] ticks at that position (ranks 0, 10, 20, 31)
};

// The team's arrival flags, seen from one workgroup.  Every workgroup owns ONE word of its team's 128-byte flag line and stores
// the number of the stage it has finished there (a plain store: it lands in the XCD's L2 like the stage's rows, and it is issued
// only after every wave's row stores have been acknowledged).  A consumer's first wave reads the whole line with one sc1 load per
// poll (lane r reads workgroup r's word) until every word has reached the stage it waits for.  No atomic is involved: an
// agent-scope atomic add executes at the memory side, ~0.5-1 us away, and cost more than the stage's arithmetic
// (profiles/r02_persist_v4_stamps.txt: "pub").
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
__device__ inline unsigned min8(const u32x8& v) {
    return min(min(min(v[0], v[1]), min(v[2], v[3])), min(min(v[4], v[5]), min(v[6], v[7])));
}
constexpr unsigned PS_GIVE_UP_TICKS = 400000u;                // bounded spins: 4 ms of the 100 MHz counter (a hand-off takes 1-10 us)
constexpr unsigned PS_GIVE_UP_FIRST_TICKS = 2000000u;         // the first hand-offs also wait for PLACEMENT (all 256 workgroups co-resident: a code-object
                                                              // load, another stream's kernel draining): 20 ms before the launch gives up

struct TeamSync {
    unsigned* flags;                                          // this team's line: PS_TEAM_BLOCKS words
    int rank; unsigned epoch;                                 // stages this workgroup has finished
    unsigned* fail; int* lds_dead; bool armed, dead;
    int poll_sleep;                                           // s_sleep(1) units (64 clocks) between two polls
    // poll: 0 = one vector load of the line per poll (sc1: served by the XCD's L2); 3 (default) = s_dcache_inv + ordinary SCALAR
    // loads (they miss the just-invalidated scalar cache and read the XCD's L2, where the producers' plain stores are); 1 / 2 =
    // scalar loads with glc (correct, but every poll then sees a flag 5-17 us late: profiles/r03_decode_floor_experiments.txt).
    // The scalar path keeps the poll out of the wave's vector-memory queue: vector loads return in order, so a wave that polls
    // with vector loads cannot request its K/V panel or weights BEFORE the wait without the poll's data arriving behind them --
    // with scalar polls EVERY wave requests early (-1.3 ... 1.9 % per generate).
    int poll;
    unsigned long long* stp;                                  // diagnostic: 5 ticks per stage (wait begin / end, drain begin / end, published)
    __device__ inline bool early_all() const { return poll != 0; }
    __device__ inline void operator()() {                     // wait until every workgroup of the team has finished the previous stage
        if (!armed) return;
        armed = false;
        if (threadIdx.x < 64) {
            if (stp && threadIdx.x == 0) stp[0] = __builtin_amdgcn_s_memrealtime();
            const int lane = threadIdx.x;
            unsigned spins = 0;
            const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
            if (poll != 0) {
                // the whole 128-byte line + the launch's fail word per poll, through the scalar data path (lgkmcnt, not vmcnt)
                for (;;) {
                    u32x8 f0, f1, f2, f3; unsigned fl;
                    if (poll == 3) {                          // invalidate the scalar cache, then ordinary scalar loads (they miss it and read the L2)
                        asm volatile("s_dcache_inv\n\t"
                                     "s_load_dwordx8 %0, %5, 0x0\n\t"
                                     "s_load_dwordx8 %1, %5, 0x20\n\t"
                                     "s_load_dwordx8 %2, %5, 0x40\n\t"
                                     "s_load_dwordx8 %3, %5, 0x60\n\t"
                                     "s_load_dword %4, %6, 0x0\n\t"
                                     "s_waitcnt lgkmcnt(0)"
                                     : "=&s"(f0), "=&s"(f1), "=&s"(f2), "=&s"(f3), "=&s"(fl) : "s"(flags), "s"(fail) : "memory");
                    } else {
                    if (poll == 2) asm volatile("s_dcache_inv" ::: "memory");
                    asm volatile("s_load_dwordx8 %0, %5, 0x0 glc\n\t"
                                 "s_load_dwordx8 %1, %5, 0x20 glc\n\t"
                                 "s_load_dwordx8 %2, %5, 0x40 glc\n\t"
                                 "s_load_dwordx8 %3, %5, 0x60 glc\n\t"
                                 "s_load_dword %4, %6, 0x0 glc\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&s"(f0), "=&s"(f1), "=&s"(f2), "=&s"(f3), "=&s"(fl) : "s"(flags), "s"(fail) : "memory");
                    }
                    const unsigned behind = min(min(min8(f0), min8(f1)), min(min8(f2), min8(f3)));
                    if (behind >= epoch) break;
                    // give up when another workgroup has (every hand-off sees the fail word) or after PS_GIVE_UP_TICKS
                    if (fl != 0u || ((++spins & 63u) == 0u && __builtin_amdgcn_s_memrealtime() - t_begin > (epoch <= 2u ? PS_GIVE_UP_FIRST_TICKS : PS_GIVE_UP_TICKS))) {
                        if (lane == 0) { atomicOr(fail, 1u); *lds_dead = 1; }
                        break;
                    }
                    for (int i = 0; i < poll_sleep; ++i) __builtin_amdgcn_s_sleep(1);
                }
            } else {
                // ONE poll in flight.  (Four, a quarter of a round trip apart, to see the last arrival sooner: 37.2 vs 36.5 ms per
                // generate -- the extra reads of the line queue in front of the arrivals' stores at its L2 channel.)
                for (;;) {
                    const unsigned v = lane < PS_TEAM_BLOCKS ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
                    if (__builtin_amdgcn_ballot_w64(v < epoch) == 0ull) break;
                    for (int i = 0; i < poll_sleep; ++i) __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0u) {             // give up after PS_GIVE_UP_TICKS or when another workgroup has
                        if (__builtin_amdgcn_s_memrealtime() - t_begin > (epoch <= 2u ? PS_GIVE_UP_FIRST_TICKS : PS_GIVE_UP_TICKS) ||
                            __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            if (lane == 0) { atomicOr(fail, 1u); *lds_dead = 1; }
                            break;
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (stp && threadIdx.x == 0) stp[1] = __builtin_amdgcn_s_memrealtime();
        }
        __syncthreads();
        dead = dead || *lds_dead != 0;
    }
    __device__ inline void arrive() {                         // this workgroup's part of the stage is in L2
        if (stp && threadIdx.x == 0) stp[2] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains
        __syncthreads();
        ++epoch;
        if (threadIdx.x == 0) {
            if (stp) stp[3] = __builtin_amdgcn_s_memrealtime();
            __hip_atomic_store(flags + rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // a plain global_store_dword
            if (stp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stp[4] = __builtin_amdgcn_s_memrealtime(); stp += PS_STAMP_WORDS; }
        }
        armed = true;
    }
};

// KW (64-byte k-chunks per wave, dec_gemm.h) the launch path picks for a given K: same choice here -> same bits
template <typename T> constexpr int ps_kw_pro(int K) { const int kw = K / (4 * Elem<T>::KCHUNK); return (kw == 2 || kw == 4 || kw == 6) ? kw : 0; }
template <typename T> constexpr int ps_kw_half(int K) {
    const int kw = K / (4 * Elem<T>::KCHUNK); return (kw == 4 || kw == 6 || kw == 8 || kw == 12 || kw == 16) ? kw : 0;
}

template <typename T, int D_, int HEADS_>
constexpr size_t persist_group_lds() {
    size_t g = (size_t)DG_BM * D_ * sizeof(T) + 8192;         // LN-prologue A image (K = D) + the cross-wave K reduction behind it
    if (g < sizeof(DecAttnLds<false>)) g = sizeof(DecAttnLds<false>);
    return (g + 255) & ~(size_t)255;
}

// FFN-out weights resident in LDS (see the kernel): one tile's fragments per 256-thread group, 16 bytes x KW per thread
template <typename T, int D_>
constexpr bool persist_w2_in_lds() {
    return sizeof(T) == 2 && ps_kw_half<T>(4 * D_) > 0 && wfrag_regs(ps_kw_half<T>(4 * D_), 16) <= WBUF_REGS && D_ / 16 == 16;
}
template <typename T, int D_, int HEADS_>
constexpr size_t persist_lds_bytes() {
    return 2 * persist_group_lds<T, D_, HEADS_>() + 16 + (persist_w2_in_lds<T, D_>() ? 2 * (size_t)WBUF_REGS * 256 * 16 : 0) +
           4 * (size_t)D_ * sizeof(float);                    // LayerNorm gamma / beta of the stack and of the final norm (dec_gemm.h: ln_lds)
}

// SAMPLE: the position's last stage is the reference's sampler (step.h) instead of the arg-max.  Two instantiations: the greedy
// kernel carries no sampler code (the sampler inlined into ONE kernel behind a run-time flag cost the greedy decode 40 VGPRs).
template <typename T, int D_, int HEADS_, bool SAMPLE>
__global__ __launch_bounds__(PS_THREADS) void decode_persist_kernel(PersistArgs<T> a) {
    constexpr int D = D_, HEADS = HEADS_, ID = HEADS * DH, F = 4 * D;
    constexpr size_t GLDS = persist_group_lds<T, D_, HEADS_>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int team = blockIdx.x % PS_TEAMS, rank = blockIdx.x / PS_TEAMS;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    int tid = threadIdx.x & 255;
    unsigned char* smem = smem_all + (size_t)grp * GLDS;
    int* lds_dead = reinterpret_cast<int*>(smem_all + 2 * GLDS);
    // a stage's tiles are dealt to the workgroups' FIRST groups before any second group gets one: a stage of 32 tiles (the out
    // projections, the logits; every attention stage up to batch 32) runs one tile on each of the team's 32 CUs instead of two on
    // 16 -- half the weight / panel bytes through each CU's one vector-memory pipeline -- and a group without a tile only keeps
    // the workgroup's barriers (dec_gemm_idle / dec_attn_idle)
    const int sa = grp * PS_TEAM_BLOCKS + rank;
    constexpr int NSB = PS_TEAM_BLOCKS * 2;

    // rows of this team
    const int rpt = (a.B + PS_TEAMS - 1) / PS_TEAMS;
    const int r0 = team * rpt, nr = max(0, min(a.B - r0, rpt));
    if (nr == 0) return;                                      // whole team: nothing to do, nobody waits for it
    const int nteams = (a.B + rpt - 1) / rpt;
    const unsigned full_mask = (1u << nteams) - 1u;
    const int nrt = (nr + DG_BM - 1) / DG_BM;                 // 16-row tiles

    PersistCtl* ctl = a.ctl;
    if (threadIdx.x == 0) {
        *lds_dead = 0;
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        atomicOr(&ctl->xcc_mask[team], 1u << (id & 15u));
        if (rank == 0 && a.eos >= 0 && a.bos == a.eos) {      // BOS already is eos: every row "contains eos" (decoder.py:115)
            __hip_atomic_store(&ctl->eos_rows[team], (unsigned)nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.stagger_ticks > 0 && team > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), until = (unsigned long long)a.stagger_ticks * team;
        while (__builtin_amdgcn_s_memrealtime() - t0 < until) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    TeamSync ts{&ctl->flags[team][0], rank, 0u, &ctl->fail, lds_dead, false, false, a.poll_sleep, a.poll_mode, nullptr};
    bool placement_checked = false;

    // LayerNorm parameters -> LDS once: {gamma, beta} of the stack's shared norm, then of the final norm
    float* ln_lds = reinterpret_cast<float*>(smem_all + 2 * GLDS + 16 + (persist_w2_in_lds<T, D_>() ? 2 * (size_t)WBUF_REGS * 256 * 16 : 0));
    for (int c = threadIdx.x; c < D; c += PS_THREADS) {
        ln_lds[c] = a.gamma[c]; ln_lds[D + c] = a.beta[c]; ln_lds[2 * D + c] = a.gamma_f[c]; ln_lds[3 * D + c] = a.beta_f[c];
    }
    __syncthreads();
    const unsigned ln_lds_addr = (unsigned)(uintptr_t)ln_lds;     // low half of a generic LDS address = the LDS byte address

    DecGemmArgs<T> gb{};
    gb.ln_lds = ln_lds_addr; gb.w_tiled = a.w_tiled;
    gb.rows = nr; gb.gamma = a.gamma; gb.beta = a.beta; gb.t_ptr = nullptr; gb.D = D; gb.inner = ID; gb.heads = HEADS; gb.tmax = a.Tmax;
    float* lx = a.dx + (size_t)r0 * D; float* ly = a.dy + (size_t)r0 * D; float* lq = a.dq + (size_t)r0 * ID;
    T* lao = a.dao + (size_t)r0 * ID; T* lhid = a.dhid + (size_t)r0 * F; float* llog = a.dlogits + (size_t)r0 * a.V;
    const bool poll_wave = threadIdx.x < 64 && !ts.early_all();   // a wave that polls with VECTOR loads requests its panel behind the wait
    const int srank = rank == 0 ? 0 : (rank == 10 ? 1 : (rank == 20 ? 2 : (rank == PS_TEAM_BLOCKS - 1 ? 3 : -1)));
    unsigned long long* stamp_base = (a.stamps && srank >= 0 && threadIdx.x == 0)
        ? a.stamps + ((size_t)team * PS_STAMP_RANKS + srank) * PS_MAX_STAGES * PS_STAMP_WORDS : nullptr;

    // the tile column this group takes in round 0 of a GEMM stage with `ncol` column tiles (what PS_GEMM computes)
    auto bx0 = [&](int ncol) { const int nt = ncol * nrt; return (sa < nt ? sa : nt - 1) % ncol; };
    auto has0 = [&](int ncol) { return rank < ncol * nrt; };            // this workgroup runs a tile in round 0
    constexpr int KWP = ps_kw_pro<T>(D), KWI = ps_kw_half<T>(ID), KWF = ps_kw_half<T>(F);
    constexpr int NC_QKV = (3 * ID + 31) / 32, NC_O = 2 * D / 16, NC_F1 = 2 * F / 32, NC_F2 = D / 16;
    const int nc_log = (a.V + 31) / 32;
    // ONE register buffer for the weight fragments requested a stage ahead (<= 8 x 16 bytes per lane; stages whose tile needs
    // more request their weights themselves, as the launch path does)
    // PS_AHEAD: request a stage's weights one stage ahead into `wbuf`.  Measured (profiles/r02_persist_*): the 32 extra live
    // registers push the attention stages (already at the 256-VGPR limit of an 8-wave workgroup) into scratch, and the
    // arrival's vmcnt(0) drain waits for the early request anyway (vector memory returns in order) -- slower, so off.
    constexpr bool PS_AHEAD = TXO_PS_AHEAD;
    constexpr bool PF_P = PS_AHEAD && KWP > 0 && wfrag_regs(KWP, 32) <= WBUF_REGS, PF_I = PS_AHEAD && KWI > 0 && wfrag_regs(KWI, 16) <= WBUF_REGS,
                   PF_F = PS_AHEAD && KWF > 0 && wfrag_regs(KWF, 16) <= WBUF_REGS;
    WBuf wbuf;
    if constexpr (PF_P) dec_gemm_prefetch<T, KWP, 32>(wbuf, a.L[0].wqkv, 3 * ID, bx0(NC_QKV), tid, has0(NC_QKV), a.w_tiled);

    // FFN-out (K = 4D: the longest weight rows of the step, 32 KB per tile) waited ~1.7 us per stage for its fragments to come
    // from the Infinity Cache (profiles/r02_persist_v6_stamps.txt: "reduce").  Its 16 tiles per layer are therefore pinned to
    // 16 workgroups PER LAYER, which keep their tile's fragments in LDS for the whole decode: every thread parks its own
    // 8 x 16 bytes there once and reads them back each position (thread-private: no synchronisation, same fragments -> same
    // bits as the launch path).
    constexpr bool W2_LDS_OK = persist_w2_in_lds<T, D_>();
    const bool w2_lds = W2_LDS_OK && a.Ld <= PS_TEAM_BLOCKS / 8;
    // group g of workgroup `rank` owns tile column (rank & 15) of layer (rank >> 4) + 2 g: a layer's 16 tiles run on 16 CUs, ONE
    // per CU (two per CU on 8 CUs made each tile wait for the other's rows and fragments: "reduce" 0.7 us)
    const int w2_layer = (rank >> 4) + 2 * grp, w2_tile = rank & 15;             // this group's layer and FFN-out tile column (of 16)
    unsigned char* w2_lds_base = smem_all + 2 * GLDS + 16 + (size_t)grp * (WBUF_REGS * 256 * 16);
    if constexpr (W2_LDS_OK) {
        if (w2_lds && w2_layer < a.Ld) {
            WBuf park;
            dec_gemm_prefetch<T, KWF, 16>(park, a.L[w2_layer].w2, D, w2_tile, tid, true, a.w_tiled);
#pragma unroll
            for (int c = 0; c < KWF; ++c) st16(w2_lds_base + ((size_t)c * 256 + tid) * 16, park.r[c]);
        }
    }

    int t = 0;
    for (; t < a.max_len; ++t) {
        // opaque per position: nothing derived from the thread index is hoisted out of the position loop (hipcc otherwise keeps
        // dozens of per-thread addresses of all stages alive across the whole loop, in a kernel at the 256-VGPR limit)
        asm volatile("" : "+v"(tid));
        gb.t_host = t;
        if (a.inject_fail > 0 && t == a.inject_fail && team == 0 && rank == 0) {   // TXO_PERSIST_INJECT_FAIL (tests): what a time-out does
            if (threadIdx.x == 0) { atomicOr(&ctl->fail, 1u); *lds_dead = 1; }
            __syncthreads();
            ts.dead = true;
            break;
        }
        int stage = 0;
        ts.stp = (stamp_base && t == a.stamp_step) ? stamp_base : nullptr;
        // one GEMM stage: tiles (bx, by) dealt over the team's groups; a workgroup with no tile only synchronises.
        // PRE: the first round's weight fragments when an earlier stage requested them (else nullptr); PF: what this stage
        // requests for a later one once its own loads are out
#define PS_GEMM(PRO, EPI, KW, BN, ARGS, NCOL, PRE, PF)                                                                  \
        do {                                                                                                            \
            const int ncol_ = (NCOL), nt_ = ncol_ * nrt;                                                                \
            for (int base_ = 0; base_ < nt_; base_ += NSB) {                                                            \
                if (base_ > 0) __syncthreads();          /* the previous round's LDS reads are done */                  \
                if (base_ + rank < nt_) {                /* the workgroup's first group has a tile */                   \
                    const int tile_ = base_ + sa;                                                                       \
                    if (tile_ >= nt_) dec_gemm_idle<PRO>(ts);                                                           \
                    else if (base_ == 0) {                                                                              \
                        auto args_ = ARGS;                                                                              \
                        args_.stamps = (ts.stp && grp == 0) ? ts.stp + 5 - 3 * tile_ : nullptr;   /* tile stamps land at stp[5..7] */ \
                        dec_gemm_tile_pf<T, PRO, EPI, KW, BN, true, PRE>(args_, tile_ % ncol_, tile_ / ncol_, tid, smem, true, ts, wbuf, PF); \
                    }                                                                                                   \
                    else dec_gemm_tile<T, PRO, EPI, KW, BN, true>(ARGS, tile_ % ncol_, tile_ / ncol_, tid, smem, true, ts); \
                } else if (base_ == 0) { ts(); PF(); }                                                                  \
            }                                                                                                           \
            ts();                                                                                                       \
            ts.arrive();                                                                                                \
            ++stage;                                                                                                    \
        } while (0)
        // LK: keys of the panel (a group without a tile mirrors the tile's barriers)
#define PS_ATTN(MODE, APRO, NLV, ARGS, PF, LK)                                                                          \
        do {                                                                                                            \
            const int np_ = nr * HEADS;                                                                                 \
            for (int base_ = 0; base_ < np_; base_ += NSB) {                                                            \
                if (base_ > 0) __syncthreads();                                                                         \
                if (base_ + rank < np_) {                                                                               \
                    const int p_ = base_ + sa;                                                                          \
                    if (p_ >= np_) dec_attn_idle<T, MODE, APRO, NLV>((LK), ts);                                         \
                    else if (base_ == 0) {                                                                              \
                        auto args_ = ARGS;                                                                              \
                        args_.stamps = (PS_ATTN_STAMPS && ts.stp && grp == 0) ? ts.stp + 5 - 3 * p_ : nullptr;   /* tile stamps land at stp[5..7] */ \
                        dec_attn_tile<T, MODE, APRO, NLV, 1, false, false, true>(args_, p_, tid,                       \
                            *reinterpret_cast<DecAttnLds<false>*>(smem), true, poll_wave && !((a.early_mask >> (MODE == ATT_SELF ? 0 : 1)) & 1), ts, PF); \
                    }                                                                                                   \
                    else dec_attn_tile<T, MODE, APRO, NLV, 1, false, false, true>(ARGS, p_, tid,          \
                        *reinterpret_cast<DecAttnLds<false>*>(smem), true, poll_wave && !((a.early_mask >> (MODE == ATT_SELF ? 0 : 1)) & 1), ts); \
                } else if (base_ == 0) { ts(); PF(); }                                                                  \
            }                                                                                                           \
            ts();                                                                                                       \
            ts.arrive();                                                                                                \
            ++stage;                                                                                                    \
        } while (0)
        for (int l = 0; l < a.Ld; ++l) {
            asm volatile("" : "+v"(tid));                     // ... nor out of the layer loop
            const PersistLayer<T>& W = a.L[l];
            T* kc = a.skv + (size_t)(2 * l) * a.self_stride + (size_t)r0 * HEADS * a.Tmax * DH;
            T* vc = a.skv + (size_t)(2 * l + 1) * a.self_stride + (size_t)r0 * HEADS * a.Tmax * DH;
            auto pf_os = [&]() { if constexpr (PF_I) dec_gemm_prefetch<T, KWI, 16>(wbuf, W.wo_s, 2 * D, bx0(NC_O), tid, has0(NC_O), a.w_tiled); };
            auto pf_oc = [&]() { if constexpr (PF_I) dec_gemm_prefetch<T, KWI, 16>(wbuf, W.wo_c, 2 * D, bx0(NC_O), tid, has0(NC_O), a.w_tiled); };
            auto pf_f1 = [&]() { if constexpr (PF_P) dec_gemm_prefetch<T, KWP, 32>(wbuf, W.w1, 2 * F, bx0(NC_F1), tid, has0(NC_F1), a.w_tiled); };
            auto pf_f2 = [&]() { if constexpr (PF_F) dec_gemm_prefetch<T, KWF, 16>(wbuf, W.w2, D, bx0(NC_F2), tid, has0(NC_F2), a.w_tiled); };
            auto pf_next = [&]() {                            // behind the layer's last stage: next layer's q,k,v or the logits
                if constexpr (PF_P) {
                    if (l + 1 < a.Ld) dec_gemm_prefetch<T, KWP, 32>(wbuf, a.L[l + 1].wqkv, 3 * ID, bx0(NC_QKV), tid, has0(NC_QKV), a.w_tiled);
                    else dec_gemm_prefetch<T, KWP, 32>(wbuf, a.wlog, a.V, bx0(nc_log), tid, has0(nc_log), a.w_tiled);
                }
            };
            auto none = []() {};
            {   // LN sandwich (or token + position embedding) + q,k,v projection; k,v appended to the cache (attention.py:124-127)
                DecGemmArgs<T> g = gb; g.N = 3 * ID; g.K = D; g.W = W.wqkv; g.y = ly; g.x_out = lx;
                g.tok = a.cur_tok + r0; g.tok_emb = a.tok_emb; g.pos_emb = a.pos_emb; g.q_out = lq; g.k_cache = kc; g.v_cache = vc;
                if (l == 0) PS_GEMM(PRO_EMBED, EPI_QKV, KWP, 32, g, NC_QKV, PF_P, none);
                else PS_GEMM(PRO_LN2, EPI_QKV, KWP, 32, g, NC_QKV, PF_P, none);
            }
            if (!placement_checked) {                         // every workgroup of the team has ORed its XCC id in by now
                placement_checked = true;
                ts();
                if (threadIdx.x == 0) {
                    const unsigned m = __hip_atomic_load(&ctl->xcc_mask[team], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_popcount(m) != 1) { atomicOr(&ctl->fail, 2u); *lds_dead = 1; }
                }
                __syncthreads();
                ts.dead = ts.dead || *lds_dead != 0;
            }
            if (ts.dead) break;
            DecAttnArgs<T> at{};
            at.gamma = a.gamma; at.beta = a.beta; at.D = D; at.heads = HEADS; at.t_host = t; at.kv_div = 1; at.path = nullptr;
            at.y = ly; at.x_out = lx; at.out = lao; at.qin = lq; at.tok = a.cur_tok + r0; at.tok_emb = a.tok_emb; at.pos_emb = a.pos_emb;
            {   // causal self attention over the cache (attention.py:148-173, one query)
                DecAttnArgs<T> s = at; s.W = W.wqkv; s.K = kc; s.V = vc; s.lmax = a.Tmax; s.len = 0;
                PS_ATTN(ATT_SELF, APRO_NONE, (sizeof(T) == 2 ? 8 : 16), s, pf_os, t + 1);
            }
            {   // gated output projection + residual (attention.py:96-99,180)
                DecGemmArgs<T> g = gb; g.N = 2 * D; g.K = ID; g.W = W.wo_s; g.bias = W.bo_s; g.A = lao; g.resid = lx; g.y_out = ly;
                PS_GEMM(PRO_NONE, EPI_GLU_RES, KWI, 16, g, NC_O, PF_I, none);
            }
            if (ts.dead) break;
            {   // cross attention over the cached encoder projections, LN sandwich + q projection fused in front
                DecAttnArgs<T> s = at; s.W = W.wq_c;
                s.K = a.ckv + (size_t)(2 * l) * a.cross_stride + (size_t)r0 * HEADS * a.N * DH;
                s.V = a.ckv + (size_t)(2 * l + 1) * a.cross_stride + (size_t)r0 * HEADS * a.N * DH;
                s.lmax = a.N; s.len = a.N;
                PS_ATTN(ATT_CROSS, APRO_LN2, DA_NL_CROSS, s, pf_oc, a.N);
            }
            {
                DecGemmArgs<T> g = gb; g.N = 2 * D; g.K = ID; g.W = W.wo_c; g.bias = W.bo_c; g.A = lao; g.resid = lx; g.y_out = ly;
                PS_GEMM(PRO_NONE, EPI_GLU_RES, KWI, 16, g, NC_O, PF_I, pf_f1);
            }
            if (ts.dead) break;
            {   // GeGLU feed-forward (attention.py:9-17,41-67)
                DecGemmArgs<T> g = gb; g.N = 2 * F; g.K = D; g.W = W.w1; g.bias = W.b1; g.y = ly; g.x_out = lx; g.h_out = lhid; g.F = F;
                PS_GEMM(PRO_LN2, EPI_GEGLU, KWP, 32, g, NC_F1, PF_P, pf_f2);
                DecGemmArgs<T> h = gb; h.N = D; h.K = F; h.W = W.w2; h.bias = W.b2; h.A = lhid; h.resid = lx; h.y_out = ly;
                bool w2_done = false;
                if constexpr (W2_LDS_OK) {
                    if (w2_lds) {                              // block-uniform
                        w2_done = true;
                        if ((rank >> 4) != (l & 1)) {              // this workgroup owns no tile of the layer
                            ts(); pf_next();
                        } else if (w2_layer != l) {                // the other group's layer: keep the barriers
                            for (int by = 0; by < nrt; ++by) {
                                if (by > 0) __syncthreads();
                                dec_gemm_idle<PRO_NONE>(ts);
                            }
                        } else {
                            WBuf wb;
#pragma unroll
                            for (int c = 0; c < KWF; ++c) wb.r[c] = ld16(w2_lds_base + ((size_t)c * 256 + tid) * 16);
                            for (int by = 0; by < nrt; ++by) {
                                if (by > 0) __syncthreads();
                                auto args_ = h;
                                args_.stamps = (ts.stp && by == 0) ? ts.stp + 5 - 3 * w2_tile : nullptr;
                                if (by == 0) dec_gemm_tile_pf<T, PRO_NONE, EPI_BIAS_RES, KWF, 16, true, true>(args_, w2_tile, by, tid, smem, true, ts, wb, pf_next);
                                else dec_gemm_tile_pf<T, PRO_NONE, EPI_BIAS_RES, KWF, 16, true, true>(args_, w2_tile, by, tid, smem, true, ts, wb, none);
                            }
                        }
                        ts();
                        ts.arrive();
                        ++stage;
                    }
                }
                if (!w2_done) PS_GEMM(PRO_NONE, EPI_BIAS_RES, KWF, 16, h, NC_F2, PF_F, pf_next);
            }
            if (ts.dead) break;
        }
        if (ts.dead) break;
        auto pf_step = [&]() {                                // behind the logits: the next position's first q,k,v projection
            if constexpr (PF_P) dec_gemm_prefetch<T, KWP, 32>(wbuf, a.L[0].wqkv, 3 * ID, bx0(NC_QKV), tid, has0(NC_QKV), a.w_tiled);
        };
        {   // final LayerNorm + logits of this position (decoder.py:57-60)
            DecGemmArgs<T> g = gb; g.N = a.V; g.K = D; g.W = a.wlog; g.bias = a.blog; g.logits = llog; g.y = ly;
            g.gamma = a.gamma_f; g.beta = a.beta_f; g.ln_lds = ln_lds_addr + 2 * D * (unsigned)sizeof(float);
            PS_GEMM(PRO_LNF, EPI_LOGITS, KWP, 32, g, nc_log, PF_P, pf_step);
        }
        {   // greedy token, append, eos bookkeeping (decoder.py:103-116): one wave per row, rows dealt over the team's workgroups
            ts();
            const int lane = threadIdx.x & 63;
            for (int r = (int)(threadIdx.x >> 6) * PS_TEAM_BLOCKS + rank; r < nr && !ts.dead; r += PS_TEAM_BLOCKS * (PS_THREADS / 64)) {
                const int row = r0 + r;
                const float* lg = a.dlogits + (size_t)row * a.V;
                float* lo = a.logits_out ? a.logits_out + ((size_t)row * a.out_stride + t) * a.V : nullptr;
                float best = -3.4e38f; int bi = 0x7fffffff;
                if constexpr (SAMPLE) {
                    // the reference's default decode (decoder.py:104-108): top-k, softmax(/temp), one draw keyed by (seed; row, t) --
                    // the same function and key as the launch path's sample_step_kernel, hence the same draw.  The row lives in
                    // registers (step.h: vocabularies up to 1024 entries; Engine::persist_usable sends larger ones to the launch path).
                    auto ld = [&](int j) { return ldc_f32<true>(lg + j); };
                    auto ld4 = [&](int j) { return ldc16_at<true>(a.dlogits, (size_t)row * a.V + j); };
                    bi = sample_row_regs(ld, ld4, lo, a.V, lane, a.sample_topk, a.inv_temp, a.seed, (unsigned)row, (unsigned)t);
                } else {
                if ((a.V & 3) == 0) {                          // four 16-byte pieces per lane in flight
                    const int n4 = a.V >> 2;
                    for (int base = 0; base < n4; base += 256) {
                        float4 v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int j4 = base + u * 64 + lane;
                            v[u] = ldc_f4_at<true>(a.dlogits, (size_t)row * a.V + (size_t)min(j4, n4 - 1) * 4);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int j4 = base + u * 64 + lane;
                            if (j4 >= n4) continue;
                            if (lo) reinterpret_cast<float4*>(lo)[j4] = v[u];
                            const int j = j4 * 4;               // ascending index: first maximum wins inside a lane
                            if (v[u].x > best) { best = v[u].x; bi = j; }
                            if (v[u].y > best) { best = v[u].y; bi = j + 1; }
                            if (v[u].z > best) { best = v[u].z; bi = j + 2; }
                            if (v[u].w > best) { best = v[u].w; bi = j + 3; }
                        }
                    }
                } else {
                    for (int j = lane; j < a.V; j += 64) {
                        const float v = ldc_f32<true>(lg + j);
                        if (lo) lo[j] = v;
                        if (v > best) { best = v; bi = j; }
                    }
                }
                wave_argmax(best, bi);                                        // ties -> lowest index (torch.argmax)
                }
                if (lane == 0) {
                    bi = in_vocab(bi, a.V);                                       // (non-finite logits: step.h)
                    a.cur_tok[row] = bi;
                    a.tokens_out[(size_t)row * a.out_stride + t] = bi;
                    if (a.eos >= 0 && bi == a.eos &&
                        __hip_atomic_load(reinterpret_cast<unsigned*>(a.eos_seen + row), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                        __hip_atomic_store(reinterpret_cast<unsigned*>(a.eos_seen + row), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // the maximum is PERFORMED (memory-side atomic, its old value back in this lane) before the count that lets the
                        // team declare itself done is issued.  No release / acquire here: an agent-scope fence writes back / invalidates
                        // the XCD's L2 (5-15 us, probes/team_seam.hip), and this runs inside the position.
                        const int prev_max = atomicMax(&ctl->last_first_eos[team], t);
                        asm volatile("s_waitcnt vmcnt(0)" :: "v"(prev_max) : "memory");
                        __hip_atomic_fetch_add(&ctl->eos_rows[team], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (a.eos >= 0 && rank == 0 && threadIdx.x == 0) {
                // rows counted up to the previous position are all visible here (they preceded a hand-off); this position's
                // may or may not be -- the loop ends a few positions late at most, and those positions are never returned.
                // Teams are not synchronised with each other: when the chip-wide mask fills, another team may have produced its
                // last first-eos at a position this team has not decoded yet.  The reference returns every row up to the position at
                // which the LAST row of the batch first produced eos (decoder.py:115-118), so a team only stops once it has decoded
                // that position: t >= max over teams of last_first_eos.  A team's maximum is final once its bit is in the mask (see
                // above), and other teams' words are read with read-modify-write atomics (performed at the memory side: another
                // XCD's L2 never serves them).
                const unsigned n = __hip_atomic_load(&ctl->eos_rows[team], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (n >= (unsigned)nr) {
                    const unsigned bit = 1u << team;
                    const unsigned old = atomicOr(&ctl->done_mask, bit);
                    if (((old | bit) & full_mask) == full_mask) {
                        int need = 0;
                        for (int k = 0; k < nteams; ++k) need = max(need, atomicMax(&ctl->last_first_eos[k], 0));
                        if (t >= need) __hip_atomic_store(&ctl->stop[team], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            ts.arrive();
            ++stage;
        }
        // the next position's first stage reads cur_tok; the stop word is read behind the same hand-off
        ts();
        if (ts.dead) break;
        if (a.eos >= 0 && __hip_atomic_load(&ctl->stop[team], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ++t; break; }
    }
#undef PS_GEMM
#undef PS_ATTN
    if (rank == 0 && threadIdx.x == 0) ctl->steps_run[team] = t;
}

}  // namespace txo
