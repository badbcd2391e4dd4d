// LDS allocation probe: how many workgroups of T threads and S bytes of dynamic LDS run on a CU at once.
// Each workgroup waits a fixed number of clock ticks; the launch time is proportional to
// ceil(blocks_per_cu / resident).  Usage: lds_probe T S0 S1 step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spin(unsigned long long ticks, int* sink) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) % blockDim.x] == -1) *sink = 1;
}
int main(int argc, char** argv) {
    int T = atoi(argv[1]), s0 = atoi(argv[2]), s1 = atoi(argv[3]), step = atoi(argv[4]);
    int* sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int per_cu = 96, cus = 256;
    const unsigned long long ticks = 2000;   // 100 MHz wall clock: 20 us
    int last = -1;
    for (int s = s0; s <= s1; s += step) {
        spin<<<per_cu * cus, T, s>>>(ticks, sink);
        hipDeviceSynchronize();
        hipEventRecord(a);
        spin<<<per_cu * cus, T, s>>>(ticks, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        int resident = (int)(per_cu * 0.020f / ms + 0.5f);
        if (resident != last) printf("T=%d S=%d: %.3f ms -> ~%d resident per CU\n", T, s, ms, resident);
        last = resident;
    }
    return 0;
}
