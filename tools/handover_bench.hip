// tools/handover_bench.hip -- what a cross-stream hand-over costs between two kernels of one stream:
//   hipcc -O2 --offload-arch=gfx950 tools/handover_bench.hip -o tools/handover_bench && tools/handover_bench
// MI355X, ROCm 7.2: two ~29 us kernels back to back 58.4 us; with an event record / wait to a second stream, a small kernel there
// and an event record / wait back 69.9 us; with a flag kernel + hipStreamWaitValue64 instead of the first event 66.7 us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void work(float *p, int n, int iters) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { float v = p[i]; for (int k = 0; k < iters; k++) v = v * 1.0001f + 0.5f; p[i] = v; } }
__global__ void flag(unsigned long long *f, unsigned long long v) { __threadfence(); *f = v; }
int main() {
    const int n = 1 << 24;   // ~50 us kernel
    float *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t ev, ev2, t0, t1; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    unsigned long long *sig = nullptr;
    hipError_t se = hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory);
    if (se != hipSuccess) { printf("signal memory: %s\n", hipGetErrorString(se)); sig = nullptr; } else CK(hipMemset(sig, 0, 8));
    const int reps = 200;
    auto run = [&](int mode, const char *name) -> int {
        unsigned long long seq = 0;
        for (int w = 0; w < 2; w++) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, s0));
            for (int r = 0; r < reps; r++) {
                work<<<n / 256, 256, 0, s0>>>(a, n, 8);
                if (mode == 1) {                         // event hand-over to s1 (small kernel there), s0 waits for it two kernels later
                    CK(hipEventRecord(ev, s0)); CK(hipStreamWaitEvent(s1, ev, 0));
                    work<<<64, 256, 0, s1>>>(b, 64 * 256, 8);
                    CK(hipEventRecord(ev2, s1));
                } else if (mode == 2 && sig) {           // flag kernel + stream wait value
                    seq++;
                    flag<<<1, 1, 0, s0>>>(sig, seq);
                    CK(hipStreamWaitValue64(s1, sig, seq, hipStreamWaitValueGte, ~0ull));
                    work<<<64, 256, 0, s1>>>(b, 64 * 256, 8);
                    CK(hipEventRecord(ev2, s1));
                }
                work<<<n / 256, 256, 0, s0>>>(a, n, 8);
                if (mode == 1 || (mode == 2 && sig)) CK(hipStreamWaitEvent(s0, ev2, 0));
            }
            CK(hipEventRecord(t1, s0));
            CK(hipEventSynchronize(t1));
        }
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        printf("%-44s %8.2f us per pair of kernels\n", name, 1e3 * ms / reps);
        return 0;
    };
    if (run(0, "two kernels back to back")) return 1;
    if (run(1, "event record / wait both ways")) return 1;
    if (run(2, "flag kernel + hipStreamWaitValue64, event back")) return 1;
    if (run(0, "two kernels back to back")) return 1;
    return 0;
}
