#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = (size_t)66000000 * 4;   // dist_tri of the bench shard
    char *h = (char *)malloc(n); memset(h, 1, n);
    void *d; hipMalloc(&d, n); hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); printf("pageable hipMemcpy: %.1f ms (%.1f GB/s)\n", now() - t, n / (now() - t) / 1e6);
    }
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); hipHostRegister(h, n, hipHostRegisterDefault); double t1 = now();
        hipMemcpy(d, h, n, hipMemcpyHostToDevice); double t2 = now(); hipHostUnregister(h); double t3 = now();
        printf("register %.1f + copy %.1f + unregister %.1f = %.1f ms\n", t1 - t, t2 - t1, t3 - t2, t3 - t);
    }
    // threaded staging through pinned buffers
    const size_t chunk = (size_t)32 << 20; char *p[2]; hipHostMalloc((void **)&p[0], chunk); hipHostMalloc((void **)&p[1], chunk);
    hipStream_t s; hipStreamCreate(&s); hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
    for (int threads : {1, 4, 8}) {
        double t = now(); int k = 0;
        for (size_t off = 0; off < n; off += chunk, k ^= 1) {
            size_t len = n - off < chunk ? n - off : chunk;
            hipEventSynchronize(ev[k]);
            std::vector<std::thread> th; size_t per = (len + threads - 1) / threads;
            for (int i = 0; i < threads; i++) { size_t a = i * per, b = a + per > len ? len : a + per; if (a < b) th.emplace_back([=] { memcpy(p[k] + a, h + off + a, b - a); }); }
            for (auto &x : th) x.join();
            hipMemcpyAsync((char *)d + off, p[k], len, hipMemcpyHostToDevice, s); hipEventRecord(ev[k], s);
        }
        hipStreamSynchronize(s); printf("staged, %d threads: %.1f ms (%.1f GB/s)\n", threads, now() - t, n / (now() - t) / 1e6);
    }
    return 0;
}
