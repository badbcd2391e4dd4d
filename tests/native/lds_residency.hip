// Test helper (never loaded by the product): how many workgroups of T threads with S bytes of dynamic LDS are
// resident on one CU at the same time.  Every workgroup counts itself in, records the peak, waits ~100 us and
// counts itself out; with many more workgroups than the chip holds the peak is CUs x (resident per CU).
// The launch sizing (csrc/sat_capi.hip pick_epw) assumes 128 LDS granules of 1280 bytes per CU.
#include <hip/hip_runtime.h>

__global__ void occupy(int *live, int *peak, unsigned long long ticks)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int me = atomicAdd(live, 1) + 1;
        atomicMax(peak, me);
    }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) atomicSub(live, 1);
    if (lds[(threadIdx.x + 1) % blockDim.x] == -1) *peak = -1;      // keeps the LDS array alive
}

extern "C" int lds_resident_per_cu(int threads, int lds_bytes)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return -1;
    if (hipFuncSetAttribute((const void *)occupy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
    int *d = nullptr;
    if (hipMalloc(&d, 2 * sizeof(int)) != hipSuccess) return -3;
    int result = -4;
    // two launches: the first also loads the code object
    for (int pass = 0; pass < 2; pass++) {
        (void)hipMemset(d, 0, 2 * sizeof(int));
        occupy<<<cus * 40, threads, (size_t)lds_bytes>>>(d, d + 1, 10000ull);   // 100 MHz clock: 100 us
        if (hipDeviceSynchronize() != hipSuccess) { (void)hipFree(d); return -5; }
        int h[2];
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        // the first workgroups may leave while the last CUs still fill: round, and refuse a peak that is
        // more than 3 % short of a whole number per CU
        result = (h[1] + cus / 2) / cus;
        if (result < 1 || h[1] > result * cus || h[1] * 100 < result * cus * 97) result = -1000 - h[1];
    }
    (void)hipFree(d);
    return result;
}
