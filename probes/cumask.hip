// Which compute units does a hipExtStreamCreateWithCUMask stream use?  For a few masks: launch 4096 small workgroups, record (XCC_ID, SE, CU) of each.
//   hipcc --offload-arch=gfx950 -O2 probes/cumask.hip -o probes/cumask && probes/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>
__global__ void where(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);            // hold the CU so that the grid spreads
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 4);
    std::vector<unsigned> h(4096);
    auto run = [&](const char* name, std::vector<uint32_t> mask) {
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
        hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per;                       // xcc -> set of (se, cu)
        for (unsigned v : h) per[v >> 16].insert((v >> 8) & 0xff);
        printf("%-28s:", name);
        int tot = 0;
        for (auto& kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
        printf("  (distinct hw ids %d)\n", tot);
        hipStreamDestroy(s);
    };
    run("all 256", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
    run("bits 0-127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
    run("bits 128-255", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
    run("bits 0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    run("every bit with (i%8)<4", {0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu});
    run("every bit with (i%8)>=4", {0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u});
    run("every bit with (i%8)<5", {0x1f1f1f1fu, 0x1f1f1f1fu, 0x1f1f1f1fu, 0x1f1f1f1fu, 0x1f1f1f1fu, 0x1f1f1f1fu, 0x1f1f1f1fu, 0x1f1f1f1fu});
    run("bits 0-63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0});
    run("bits 0-15", {0x0000ffffu, 0, 0, 0, 0, 0, 0, 0});
    run("bits 0-7", {0x000000ffu, 0, 0, 0, 0, 0, 0, 0});
    run("bit 0", {0x1u, 0, 0, 0, 0, 0, 0, 0});
    run("bits 0,8,16,24", {0x01010101u, 0, 0, 0, 0, 0, 0, 0});
    run("even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u});
    return 0;
}
