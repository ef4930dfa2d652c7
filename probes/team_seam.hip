// Probe: what does an in-launch "team seam" cost against a dependent kernel boundary?
//
// A decode step is a chain of ~30 stages; every stage's tiles need ALL columns of the previous stage's rows, so inside one
// persistent launch each stage ends in an all-to-all hand-off among the blocks that share a group of rows (a "team").
// This probe runs that skeleton with an integer recurrence (exact, so a stale or torn read shows up as a wrong word):
//   stage s:  out[r][c] = in[r][(c+1)%D] * 1664525 + in[r][(c*7+3)%D] + s      in, out: [R][D] u32 per team, ping-pong
// in three hand-off forms and as plain launches:
//   F0  write-through: sc1 stores, every wave drains, block barrier, one agent-scope atomic add on the team counter;
//       consumer: one lane polls the counter with sc1 loads, block barrier, sc1 loads         (placement independent;
//       MI355X_MICROARCH.md visibility table, first row)
//   F1  XCD-local: plain stores (stay in the XCD's L2), drain, barrier, atomic add; consumer polls, sc1 loads (L1 bypassed,
//       served by the L2 the producer wrote) -- only meaningful when a team's blocks share an XCD; the probe reports the
//       XCC ids it saw and counts wrong words
//   F2  release/acquire: plain stores, agent release fence, atomic; consumer polls, agent acquire fence, plain loads
//   F3  XCD-local rows as F1, but NO atomic: every block stores the stage number into its word of the team's 128-byte flag line
//       (plain store); the consumer's first wave reads the line with one sc1 load per poll (lane r = block r)
//   F4  as F3, polled with SCALAR loads (s_load_dwordx16 ... glc: lgkmcnt, not vmcnt -- a polling wave could keep vector loads
//       in flight)
//   F5  no flag at all: every word travels as an 8-byte {value, stage} pair (the "LL" idea of the collective libraries);
//       producers just store, no drain, no barrier, no publication; every consumer thread re-reads its own 16-byte pieces
//       (sc1) until both stage tags match
//   L   one launch per stage (grid = all teams), the structure the engine has today
// Optional load: every LOADEVERY-th stage each block also streams STREAM_KB of a large buffer with non-temporal loads
// (the cross-attention stage of the real step).
//
// build: hipcc -O3 --offload-arch=gfx950 probes/team_seam.hip -o /tmp/team_seam ; run: /tmp/team_seam
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int R = 16, D = 256;

struct Args {
    uint32_t* act;            // [teams][2][R][D]
    unsigned* counter;        // [teams] on lines of their own (32 words apart)
    unsigned* flagline;       // [teams][64] words: forms 3 / 4
    unsigned* fail;           // spin time-outs
    unsigned* xcc;            // [blocks] XCC id seen
    const u32x4* big; size_t big_vec;   // streaming buffer
    unsigned long long* ticks;  // [teams] in-kernel duration (100 MHz ticks)
    int teams, S, stages, stream_kb, load_every, first_stage;
};

__device__ inline unsigned poll_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int FORM>
__device__ inline void stage_body(const Args& a, int team, int rank, int s, uint32_t* lds, u32x4& sink) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t* in = a.act + ((size_t)team * 2 + (s & 1)) * R * D;
    uint32_t* out = a.act + ((size_t)team * 2 + ((s + 1) & 1)) * R * D;
    // whole input image -> LDS (what an LN prologue does): 16-byte loads
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(in), 0, R * D * 4, 0x00020000);
    for (int v = tid; v < R * D / 4; v += nthr) {
        u32x4 x;
        if constexpr (FORM == 2) x = reinterpret_cast<const u32x4*>(in)[v];
        else x = __builtin_amdgcn_raw_buffer_load_b128(rin, v * 16, 0, 16);      // aux 16 = sc1
        reinterpret_cast<u32x4*>(lds)[v] = x;
    }
    // optional HBM stream (non-temporal), folded into a sink so that it is not removed
    if (a.stream_kb && a.load_every && (s % a.load_every) == a.load_every / 2) {
        const int nvec = a.stream_kb * 64;                                      // 16-byte vectors
        const size_t base = ((size_t)(blockIdx.x * 131 + s * 7919) * (size_t)nvec) % (a.big_vec - nvec);
        for (int v = tid; v < nvec; v += nthr) {
            const u32x4 x = __builtin_nontemporal_load(a.big + base + v);
            sink.x ^= x.x; sink.y ^= x.y; sink.z ^= x.z; sink.w ^= x.w;
        }
    }
    __syncthreads();
    const int cols = D / a.S;                                                   // columns of this block
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, R * D * 4, 0x00020000);
    for (int e = tid; e < R * cols; e += nthr) {
        const int r = e / cols, c = rank * cols + e % cols;
        const uint32_t v = lds[r * D + (c + 1) % D] * 1664525u + lds[r * D + (c * 7 + 3) % D] + (uint32_t)s;
        if constexpr (FORM == 0) __builtin_amdgcn_raw_buffer_store_b32(v, rout, (r * D + c) * 4, 0, 16);
        else out[r * D + c] = v;
    }
}

template <int FORM>
__global__ __launch_bounds__(512) void team_chain(Args a) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[R * D];
    const int team = blockIdx.x % a.teams, rank = blockIdx.x / a.teams;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        a.xcc[blockIdx.x] = id & 0xf;
    }
    unsigned* cnt = a.counter + team * 32;
    u32x4 sink = {0, 0, 0, 0};
    unsigned long long t0 = 0;
    if (tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    bool dead = false;
    for (int s = 0; s < a.stages; ++s) {
        if (s > 0) {   // wait for every block of the team to have finished stage s-1
            if constexpr (FORM == 3) {
                if (tid < 64) {
                    const unsigned* fl = a.flagline + team * 64;
                    unsigned spins = 0;
                    for (;;) {
                        const unsigned v = tid < a.S ? poll_sc1(fl + tid) : (unsigned)s;
                        if (__builtin_amdgcn_ballot_w64(v < (unsigned)s) == 0ull) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22)) { if (tid == 0) atomicAdd(a.fail, 1u); break; }
                    }
                }
            } else if constexpr (FORM == 4) {
                if (tid < 64) {
                    const unsigned* fl = a.flagline + team * 64;      // 64 words = two 128-byte lines (S <= 64)
                    unsigned spins = 0;
                    for (;;) {
                        typedef unsigned u16v __attribute__((ext_vector_type(16)));
                        u16v w0, w1;
                        asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&s"(w0), "=&s"(w1) : "s"(fl) : "memory");
                        unsigned mn = 0xffffffffu;
                        for (int i = 0; i < 16; ++i) { mn = min(mn, w0[i]); if (a.S > 16) mn = min(mn, w1[i]); }
                        if (mn >= (unsigned)s) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22)) { if (tid == 0) atomicAdd(a.fail, 1u); break; }
                    }
                }
            } else if (tid == 0) {
                const unsigned want = (unsigned)s * (unsigned)a.S;
                unsigned spins = 0;
                while (poll_sc1(cnt) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { atomicAdd(a.fail, 1u); break; }
                }
                if constexpr (FORM == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
        stage_body<(FORM >= 3 ? 1 : FORM)>(a, team, rank, s, lds, sink);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // every storing wave drains
        __syncthreads();
        if (tid == 0) {
            if constexpr (FORM == 2) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if constexpr (FORM >= 3) __hip_atomic_store(a.flagline + team * 64 + rank, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0 && rank == 0) a.ticks[team] = __builtin_amdgcn_s_memrealtime() - t0;
    if (sink.x == 0x12345u && sink.y == 0x777u) a.fail[1] = sink.z + sink.w + dead;   // keep the stream alive
}

// F5: {value, tag} pairs, [teams][2][R][D] x 8 bytes
__global__ __launch_bounds__(512) void team_chain_ll(Args a, uint2* act2) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[R * D];
    const int team = blockIdx.x % a.teams, rank = blockIdx.x / a.teams;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        a.xcc[blockIdx.x] = id & 0xf;
    }
    unsigned long long t0 = 0;
    if (tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    u32x4 sink = {0, 0, 0, 0};
    for (int s = 0; s < a.stages; ++s) {
        const uint2* in = act2 + ((size_t)team * 2 + (s & 1)) * R * D;
        uint2* out = act2 + ((size_t)team * 2 + ((s + 1) & 1)) * R * D;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint2*>(in), 0, R * D * 8, 0x00020000);
        // this thread's pieces: R*D/2 16-byte pieces (two pairs each) over nthr threads
        constexpr int MAXP = 8;
        u32x4 x[MAXP];
        const int np = (R * D / 2 + nthr - 1) / nthr;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < MAXP; ++i) if (i < np) x[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, (tid + i * nthr) * 16, 0, 16);
#pragma unroll
            for (int i = 0; i < MAXP; ++i) if (i < np) ok = ok && x[i].y == (unsigned)s && x[i].w == (unsigned)s;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
            if (++spins > (1u << 22)) { if ((tid & 63) == 0) atomicAdd(a.fail, 1u); break; }
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) if (i < np) { const int v = tid + i * nthr; lds[2 * v] = x[i].x; lds[2 * v + 1] = x[i].z; }
        if (a.stream_kb && a.load_every && (s % a.load_every) == a.load_every / 2) {
            const int nvec = a.stream_kb * 64;
            const size_t base = ((size_t)(blockIdx.x * 131 + s * 7919) * (size_t)nvec) % (a.big_vec - nvec);
            for (int v = tid; v < nvec; v += nthr) {
                const u32x4 y = __builtin_nontemporal_load(a.big + base + v);
                sink.x ^= y.x; sink.y ^= y.y; sink.z ^= y.z; sink.w ^= y.w;
            }
        }
        __syncthreads();
        const int cols = D / a.S;
        for (int e = tid; e < R * cols; e += nthr) {
            const int r = e / cols, c = rank * cols + e % cols;
            const uint32_t v = lds[r * D + (c + 1) % D] * 1664525u + lds[r * D + (c * 7 + 3) % D] + (uint32_t)s;
            out[r * D + c] = make_uint2(v, (unsigned)(s + 1));
        }
        __syncthreads();                                                        // LDS reuse
    }
    if (tid == 0 && rank == 0) a.ticks[team] = __builtin_amdgcn_s_memrealtime() - t0;
    if (sink.x == 0x12345u && sink.y == 0x777u) a.fail[1] = sink.z + sink.w;
}

// the same stage as its own launch: grid = teams * S blocks, in/out by stage parity
__global__ __launch_bounds__(512) void stage_launch(Args a, int s) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[R * D];
    u32x4 sink = {0, 0, 0, 0};
    stage_body<2>(a, blockIdx.x % a.teams, blockIdx.x / a.teams, s, lds, sink);
    if (sink.x == 0x12345u && sink.y == 0x777u) a.fail[1] = sink.z;
}

static void host_ref(std::vector<uint32_t>& x, int stages) {
    std::vector<uint32_t> y(x.size());
    for (int s = 0; s < stages; ++s) {
        for (int r = 0; r < R; ++r)
            for (int c = 0; c < D; ++c) y[r * D + c] = x[r * D + (c + 1) % D] * 1664525u + x[r * D + (c * 7 + 3) % D] + (uint32_t)s;
        x.swap(y);
    }
}

int main(int argc, char** argv) {
    const int stages = 210;
    const size_t big_bytes = (size_t)1 << 30;
    u32x4* big; hipMalloc(&big, big_bytes); hipMemset(big, 1, big_bytes);
    hipStream_t st; hipStreamCreate(&st);
    printf("form teams S thr stream_kb | us/stage (host) | in-kernel us/stage min..max | wrong words | spin fails | xcc spread per team\n");
    for (int thr : {256, 512}) for (int S : {32, 64}) for (int stream_kb : {0, 300}) for (int form : {0, 1, 2, 3, 4, 5, 6}) {
        const int teams = 8, blocks = teams * S;
        if (blocks * thr > 256 * 512 * 2) continue;
        if (thr == 512 && S == 64) continue;                                    // 512 blocks of 512 threads: not co-resident with headroom
        Args a{};
        a.teams = teams; a.S = S; a.stages = stages; a.stream_kb = stream_kb; a.load_every = 7;
        a.big = big; a.big_vec = big_bytes / 16;
        hipMalloc(&a.act, (size_t)teams * 2 * R * D * 4);
        uint2* act2; hipMalloc(&act2, (size_t)teams * 2 * R * D * 8);
        hipMalloc(&a.counter, teams * 32 * 4); hipMalloc(&a.flagline, teams * 64 * 4); hipMalloc(&a.fail, 64); hipMalloc(&a.xcc, blocks * 4); hipMalloc(&a.ticks, teams * 8);
        std::vector<uint32_t> init((size_t)teams * 2 * R * D);
        for (size_t i = 0; i < init.size(); ++i) init[i] = (uint32_t)(i * 2654435761u + 12345u);
        double best = 1e30; long wrong = 0; unsigned fails = 0; double kmin = 1e30, kmax = 0;
        std::vector<unsigned> xcc(blocks);
        for (int rep = 0; rep < 6; ++rep) {
            hipMemcpy(a.act, init.data(), init.size() * 4, hipMemcpyHostToDevice);
            if (form == 6) {
                std::vector<uint2> i2(init.size());
                for (size_t i = 0; i < init.size(); ++i) i2[i] = make_uint2(init[i], 0u);
                for (int t = 0; t < teams; ++t) for (int i = 0; i < R * D; ++i) i2[((size_t)t * 2 + 1) * R * D + i].y = 0xffffffffu;
                hipMemcpy(act2, i2.data(), i2.size() * 8, hipMemcpyHostToDevice);
            }
            hipMemset(a.counter, 0, teams * 32 * 4); hipMemset(a.flagline, 0, teams * 64 * 4); hipMemset(a.fail, 0, 64); hipMemset(a.ticks, 0, teams * 8);
            hipDeviceSynchronize();
            auto t0 = std::chrono::high_resolution_clock::now();
            if (form == 0) hipLaunchKernelGGL(team_chain<0>, dim3(blocks), dim3(thr), 0, st, a);
            else if (form == 1) hipLaunchKernelGGL(team_chain<1>, dim3(blocks), dim3(thr), 0, st, a);
            else if (form == 2) hipLaunchKernelGGL(team_chain<2>, dim3(blocks), dim3(thr), 0, st, a);
            else if (form == 4) hipLaunchKernelGGL(team_chain<3>, dim3(blocks), dim3(thr), 0, st, a);
            else if (form == 5) hipLaunchKernelGGL(team_chain<4>, dim3(blocks), dim3(thr), 0, st, a);
            else if (form == 6) hipLaunchKernelGGL(team_chain_ll, dim3(blocks), dim3(thr), 0, st, a, act2);
            else for (int s = 0; s < stages; ++s) hipLaunchKernelGGL(stage_launch, dim3(blocks), dim3(thr), 0, st, a, s);
            hipError_t e = hipStreamSynchronize(st);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
            if (e != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e)); return 1; }
            if (rep == 0) continue;
            best = std::min(best, us / stages);
            std::vector<uint32_t> got(init.size());
            hipMemcpy(got.data(), a.act, got.size() * 4, hipMemcpyDeviceToHost);
            if (form == 6) {
                std::vector<uint2> g2(init.size());
                hipMemcpy(g2.data(), act2, g2.size() * 8, hipMemcpyDeviceToHost);
                for (size_t i = 0; i < g2.size(); ++i) got[i] = g2[i].x;
            }
            for (int t = 0; t < teams; ++t) {
                std::vector<uint32_t> x(init.begin() + (size_t)t * 2 * R * D, init.begin() + (size_t)t * 2 * R * D + R * D);
                host_ref(x, stages);
                const uint32_t* g = &got[((size_t)t * 2 + (stages & 1)) * R * D];
                for (int i = 0; i < R * D; ++i) wrong += g[i] != x[i];
            }
            unsigned f[2]; hipMemcpy(f, a.fail, 8, hipMemcpyDeviceToHost); fails += f[0];
            if (form != 3) {
                std::vector<unsigned long long> tk(teams); hipMemcpy(tk.data(), a.ticks, teams * 8, hipMemcpyDeviceToHost);
                for (auto v : tk) { kmin = std::min(kmin, v / 100.0 / stages); kmax = std::max(kmax, v / 100.0 / stages); }
                hipMemcpy(xcc.data(), a.xcc, blocks * 4, hipMemcpyDeviceToHost);
            }
        }
        char spread[128] = "-";
        if (form != 3) {   // number of distinct XCC ids inside each team
            int off = 0;
            for (int t = 0; t < teams; ++t) { unsigned m = 0; for (int b = t; b < blocks; b += teams) m |= 1u << xcc[b]; off += snprintf(spread + off, sizeof spread - off, "%d ", __builtin_popcount(m)); }
        }
        printf("%s %d %2d %3d %3d | %7.2f | %6.2f .. %6.2f | %ld | %u | %s\n", form == 3 ? "L " : (form == 0 ? "F0" : form == 1 ? "F1" : form == 2 ? "F2" : form == 4 ? "F3" : form == 5 ? "F4" : "F5"),
               teams, S, thr, stream_kb, best, form != 3 ? kmin : 0.0, form != 3 ? kmax : 0.0, wrong, fails, spread);
        fflush(stdout);
        hipFree(act2); hipFree(a.act); hipFree(a.counter); hipFree(a.flagline); hipFree(a.fail); hipFree(a.xcc); hipFree(a.ticks);
    }
    return 0;
}
