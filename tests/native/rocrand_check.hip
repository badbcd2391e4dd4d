// tests/native/rocrand_check.hip - TEST CODE: is satk::philox_block(seed, subsequence, block) the block
// that rocRAND's device API returns for rocrand_init(seed, subsequence, 4 * block) + rocrand4()?
// One lane per probe; both sides run on the device, the host compares the words.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <stdint.h>

#include "sat_sa_kernel.hpp"

__global__ void philox_both(int n, const unsigned long long *seed, const unsigned long long *subseq,
                            const uint32_t *block, uint4 *ours, uint4 *theirs, float *uni_ours, float *uni_theirs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ours[i] = satk::philox_block(seed[i], subseq[i], block[i]);
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed[i], subseq[i], 4ull * block[i], &st);
    theirs[i] = rocrand4(&st);
    // the uniform conversion of the first word, library side: a fresh state drawn with rocrand_uniform
    rocrand_init(seed[i], subseq[i], 4ull * block[i], &st);
    uni_theirs[i] = rocrand_uniform(&st);
    uni_ours[i] = satk::to_uniform(ours[i].x);
}

// returns the number of probes whose four words (or whose uniform) differ, or -1 on a HIP error
extern "C" int sat_test_philox_vs_rocrand(int n, const unsigned long long *seed, const unsigned long long *subseq,
                                          const uint32_t *block, uint32_t *first_ours, uint32_t *first_theirs)
{
    unsigned long long *d_seed = nullptr, *d_sub = nullptr;
    uint32_t *d_block = nullptr;
    uint4 *d_a = nullptr, *d_b = nullptr;
    float *d_ua = nullptr, *d_ub = nullptr;
    int bad = -1;
    if (hipMalloc(&d_seed, n * 8) == hipSuccess && hipMalloc(&d_sub, n * 8) == hipSuccess &&
        hipMalloc(&d_block, n * 4) == hipSuccess && hipMalloc(&d_a, n * 16) == hipSuccess &&
        hipMalloc(&d_b, n * 16) == hipSuccess && hipMalloc(&d_ua, n * 4) == hipSuccess && hipMalloc(&d_ub, n * 4) == hipSuccess &&
        hipMemcpy(d_seed, seed, n * 8, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(d_sub, subseq, n * 8, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(d_block, block, n * 4, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(philox_both, dim3((n + 255) / 256), dim3(256), 0, 0, n, d_seed, d_sub, d_block, d_a, d_b, d_ua, d_ub);
        uint4 *a = new uint4[n], *b = new uint4[n];
        float *ua = new float[n], *ub = new float[n];
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(a, d_a, n * 16, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(b, d_b, n * 16, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(ua, d_ua, n * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(ub, d_ub, n * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            bad = 0;
            for (int i = 0; i < n; i++) {
                const bool same = a[i].x == b[i].x && a[i].y == b[i].y && a[i].z == b[i].z && a[i].w == b[i].w && ua[i] == ub[i];
                if (!same && bad++ == 0) {
                    first_ours[0] = a[i].x; first_ours[1] = a[i].y; first_ours[2] = a[i].z; first_ours[3] = a[i].w;
                    first_theirs[0] = b[i].x; first_theirs[1] = b[i].y; first_theirs[2] = b[i].z; first_theirs[3] = b[i].w;
                }
            }
        }
        delete[] a; delete[] b; delete[] ua; delete[] ub;
    }
    (void)hipFree(d_seed); (void)hipFree(d_sub); (void)hipFree(d_block); (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_ua); (void)hipFree(d_ub);
    return bad;
}
