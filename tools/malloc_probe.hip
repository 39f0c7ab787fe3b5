// hipMalloc cost for tens of GB (why: a 3.1 Gbp index build spent 9 of 10.7 s in hipMalloc calls).
// Pass 1: 12 x 20 GiB on a fresh process; pass 2: the same right after dirtying and freeing them; pass 3: after a pause.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    CK(hipFree(0));
    const size_t chunk = 20ull << 30;
    const int count = 12;
    for (int pass = 1; pass <= 3; pass++) {
        std::vector<void*> ps;
        printf("pass %d:", pass);
        double total = 0;
        for (int i = 0; i < count; i++) {
            void* p = nullptr;
            double t0 = now();
            CK(hipMalloc(&p, chunk));
            double dt = now() - t0;
            total += dt;
            printf(" %.0f", dt);
            ps.push_back(p);
        }
        printf("  ms  (sum %.0f ms for %d x 20 GiB)\n", total, count);
        double t0 = now();
        for (void* p : ps) CK(hipMemsetAsync(p, 0xAB, chunk, 0));
        CK(hipDeviceSynchronize());
        double t1 = now();
        for (void* p : ps) CK(hipFree(p));
        double t2 = now();
        printf("        memset all %.0f ms, free all %.0f ms\n", t1 - t0, t2 - t1);
        if (pass == 2) std::this_thread::sleep_for(std::chrono::seconds(8));
    }
    // one big block, then the same bytes as four blocks
    for (int rep = 0; rep < 2; rep++) {
        void* p = nullptr;
        double t0 = now();
        CK(hipMalloc(&p, 80ull << 30));
        double t1 = now();
        CK(hipFree(p));
        void* q[4];
        double t2 = now();
        for (int i = 0; i < 4; i++) CK(hipMalloc(&q[i], 20ull << 30));
        double t3 = now();
        for (int i = 0; i < 4; i++) CK(hipFree(q[i]));
        printf("80 GiB in one block: %.0f ms; as 4 x 20 GiB: %.0f ms\n", t1 - t0, t3 - t2);
    }
    return 0;
}
