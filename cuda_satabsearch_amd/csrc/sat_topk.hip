// sat_topk.hip - best-k hits of a finished search, selected on the device.
//
// Users of the reference sort the full "name score ..." listing by raw score and keep
// the head (README_example_usage.txt:100, 256: `sort -k 2,2nr | head`); norm2 / z / p are
// pure functions of (score, n1, n2) (gumbelstats.c).  So only k (index, score) pairs need
// to leave the GPU: one pass packs (score, entry index) into 64-bit keys, rocPRIM's radix
// sort (through hipCUB) orders them, the first k come back.  Ties keep database order.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <vector>

#include "sat_ctx.hpp"

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t err__ = (expr);                                                          \
        if (err__ != hipSuccess)                                                            \
            return sat_fail(err__ == hipErrorOutOfMemory ? SAT_ENOMEM : SAT_EDEVICE,        \
                            "%s failed: %s", #expr, hipGetErrorString(err__));              \
    } while (0)

namespace {

// key = biased score in the high word, inverted entry index in the low word: a descending
// sort lists higher scores first and, among equal scores, lower entry indices first
__global__ void pack_keys(const int32_t *scores, int n, unsigned long long *keys)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        keys[i] = ((unsigned long long)(uint32_t)(scores[i] + 0x40000000) << 32) | (0xFFFFFFFFu - (uint32_t)i);
}

}  // namespace

extern "C" int sat_topk(sat_ctx *ctx, int query, int k, int32_t *entry_index, int32_t *scores_out)
{
    if (!ctx) return sat_fail(SAT_EINVAL, "null context");
    if (!entry_index || !scores_out || k < 1) return sat_fail(SAT_EINVAL, "bad top-k arguments");
    if (ctx->n_entries <= 0 || ctx->queries.empty() || !ctx->d_scores || ctx->searched_nq != ctx->queries.size())
        return sat_fail(SAT_ESTATE, "no search has run since the last database upload / query change");
    if (query < 0 || query >= (int)ctx->queries.size()) return sat_fail(SAT_EINVAL, "query %d out of range", query);
    const int n = ctx->n_entries;
    if (k > n) k = n;
    HIP_TRY(hipSetDevice(ctx->device));

    unsigned long long *keys = nullptr, *sorted = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    int rc = SAT_OK;
    auto run = [&]() -> int {
        HIP_TRY(hipMalloc(&keys, (size_t)n * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc(&sorted, (size_t)n * sizeof(unsigned long long)));
        hipLaunchKernelGGL(pack_keys, dim3((n + 255) / 256), dim3(256), 0, ctx->stream,
                           ctx->d_scores + (size_t)query * n, n, keys);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipcub::DeviceRadixSort::SortKeysDescending(nullptr, temp_bytes, keys, sorted, n, 0, 64, ctx->stream));
        HIP_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 1));
        HIP_TRY(hipcub::DeviceRadixSort::SortKeysDescending(temp, temp_bytes, keys, sorted, n, 0, 64, ctx->stream));
        std::vector<unsigned long long> head((size_t)k);
        HIP_TRY(hipMemcpyAsync(head.data(), sorted, (size_t)k * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < k; i++) {
            entry_index[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)(head[(size_t)i] & 0xFFFFFFFFu));
            scores_out[i] = (int32_t)(uint32_t)(head[(size_t)i] >> 32) - 0x40000000;
        }
        return SAT_OK;
    };
    rc = run();
    if (keys) (void)hipFree(keys);
    if (sorted) (void)hipFree(sorted);
    if (temp) (void)hipFree(temp);
    return rc == SAT_OK ? k : rc;
}
