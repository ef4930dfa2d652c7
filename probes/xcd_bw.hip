// Per-XCD streaming rate when all 8 XCDs stream at once (round 5).  256 (or 2048) workgroups each read their own contiguous piece of a buffer with
// 16-byte loads; every workgroup records its physical XCC id and its start / end on the 100 MHz counter.  Modes: static (workgroup b reads piece b)
// and queue (a workgroup takes pieces from an atomic counter until none is left).
//   hipcc -O3 --offload-arch=gfx950 probes/xcd_bw.hip -o probes/xcd_bw && probes/xcd_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Rec { unsigned long long t0, t1; unsigned xcc, pieces; };
__device__ inline unsigned stream_piece(const u32x4* p, size_t n16, int tid, int nthr) {
    unsigned acc = 0;
    size_t i = tid;
    for (; i + 7 * (size_t)nthr < n16; i += 8 * (size_t)nthr) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * (size_t)nthr);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += nthr) { u32x4 v = p[i]; acc += v.x ^ v.w; }
    return acc;
}
// mode 2: the latent tile's request shape -- 4 waves, 16-row tiles of 512-byte rows dealt round-robin, per tile 8 wave-instructions that each
// cover 16 rows x 64 B (lane = row + 16 * piece), three tiles in flight
__device__ inline unsigned stream_frag(const unsigned char* base, int rows, int tid) {
    const int lane = tid & 63, wave = tid >> 6, lc = lane & 15, lg = lane >> 4;
    const int ntiles = (rows + 15) >> 4;
    unsigned acc = 0;
    u32x4 v[3][8];
    auto issue = [&](int t, u32x4 (&d)[8]) {
        const int row = min(t * 16 + lc, rows - 1);
        const unsigned char* p = base + (size_t)row * 512 + lg * 16;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) d[kc] = *reinterpret_cast<const u32x4*>(p + kc * 64);
    };
    auto use = [&](u32x4 (&d)[8]) {
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) acc += d[kc].x ^ d[kc].y ^ d[kc].z ^ d[kc].w;
    };
    int t = wave;
    issue(t, v[0]); issue(t + 4, v[1]);
    for (; t < ntiles; t += 12) {
        issue(t + 8, v[2]); use(v[0]);
        if (t + 4 >= ntiles) break;
        issue(t + 12, v[0]); use(v[1]);
        if (t + 8 >= ntiles) break;
        issue(t + 16, v[1]); use(v[2]);
    }
    return acc;
}
__global__ __launch_bounds__(256) void k_stream(const u32x4* buf, size_t piece16, int n_pieces, int* counter, Rec* rec, unsigned* sink, int queue) {
    __shared__ int s_piece;
    unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0, done = 0;
    if (queue == 2) { acc = stream_frag(reinterpret_cast<const unsigned char*>(buf + (size_t)blockIdx.x * piece16), (int)(piece16 * 16 / 512), threadIdx.x); done = 1; }
    else if (!queue) { acc = stream_piece(buf + (size_t)blockIdx.x * piece16, piece16, threadIdx.x, 256); done = 1; }
    else {
        for (;;) {
            if (threadIdx.x == 0) s_piece = atomicAdd(counter, 1);
            __syncthreads();
            const int pc = s_piece;
            __syncthreads();
            if (pc >= n_pieces) break;
            acc += stream_piece(buf + (size_t)pc * piece16, piece16, threadIdx.x, 256); ++done;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { rec[blockIdx.x].t0 = t0; rec[blockIdx.x].t1 = __builtin_amdgcn_s_memrealtime(); rec[blockIdx.x].xcc = id & 0xf; rec[blockIdx.x].pieces = done; }
}
int main() {
    const size_t piece_bytes = 301568;             // one image's encoder rows at config.yml dims: 589 x 256 bf16
    for (int n_pieces : {256, 2048}) {
        const size_t total = piece_bytes * n_pieces;
        u32x4* buf; hipMalloc(&buf, total); hipMemset(buf, 1, total);
        int* counter; hipMalloc(&counter, 4); Rec* rec; hipMalloc(&rec, sizeof(Rec) * 2048); unsigned* sink; hipMalloc(&sink, 4);
        for (int queue = 0; queue < 3; ++queue) {
            if (queue == 2 && n_pieces != 256) continue;
            const int blocks = queue == 1 ? 256 : n_pieces;
            double xcc_t[16] = {0}, xcc_p[16] = {0}; int xcc_n[16] = {0}; double span = 0; const int reps = 20;
            for (int r = 0; r < reps + 2; ++r) {
                hipMemset(counter, 0, 4);
                hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, buf, piece_bytes / 16, n_pieces, counter, rec, sink, queue);
                hipDeviceSynchronize();
                if (r < 2) continue;
                std::vector<Rec> h(blocks); hipMemcpy(h.data(), rec, sizeof(Rec) * blocks, hipMemcpyDeviceToHost);
                unsigned long long a = ~0ull, b = 0;
                for (auto& x : h) { a = std::min(a, x.t0); b = std::max(b, x.t1); }
                span += (b - a) / 100.0;
                for (auto& x : h) { xcc_t[x.xcc] += (x.t1 - a) / 100.0; xcc_p[x.xcc] += x.pieces; xcc_n[x.xcc]++; }
            }
            printf("%4d pieces of %zu B, %s, %d workgroups: first entry -> last exit %.2f us = %.2f TB/s\n", n_pieces, piece_bytes, queue == 2 ? "fragment-shaped requests" : queue ? "queue " : "static", blocks,
                   span / reps, total / (span / reps) / 1e6);
            for (int x = 0; x < 16; ++x) if (xcc_n[x]) printf("    xcc %d: workgroups %5.1f  mean exit %.2f us  pieces per workgroup %.2f\n", x, xcc_n[x] / (double)reps, xcc_t[x] / xcc_n[x], xcc_p[x] / xcc_n[x]);
        }
        hipFree(buf);
    }
    return 0;
}
