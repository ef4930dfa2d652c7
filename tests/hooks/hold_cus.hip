// Test helper, NOT part of the product library or of include/texocr.h: a kernel that only holds compute units.
// tests/test_gpu_parity.py::test_persistent_decode_under_real_cu_contention runs the persistent decode launch beside it
// (the launch needs all 256 workgroups co-resident; with CUs held it must give up within its budget and launches take over).
// Built by texocr_amd.build.build_test_hooks() into tests/hooks/libtxo_testhooks.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void hold_cus_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned hold_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    hold_lds[threadIdx.x] = threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (hold_lds[threadIdx.x] == 0xffffffffu) *sink = 1u;       // never true: keeps the LDS allocation alive
}

// occupies `blocks` compute units for `microseconds` (100 MHz real-time counter), each block holding `lds_bytes` of LDS (a whole
// CU's 160 KiB keeps any other workgroup off that CU).  Asynchronous on `stream`.  Returns 0, -1 (bad argument) or -3 (HIP error).
extern "C" int txo_test_hold_cus(int32_t blocks, int32_t lds_bytes, int32_t microseconds, void* stream) {
    if (blocks < 1 || blocks > 4096 || lds_bytes < 1024 || lds_bytes > 160 * 1024 || microseconds < 1 || microseconds > 2000000) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(hold_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -3;
    hipLaunchKernelGGL(hold_cus_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (unsigned long long)microseconds * 100ull,
                       (unsigned*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
