// GPU box: what the HIP runtime's calls cost the calling thread (us per call, 2000 calls each, the streams kept busy enough that nothing
// completes early) - the budget a frame's ~45 calls are made of.   hipcc --offload-arch=gfx950 -O2 tools/hip_api_cost.cpp -o /tmp/hip_api_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
struct Big { void* a[12]; unsigned b[8]; };
__global__ void k_big(Big b) { if (b.a[0] && threadIdx.x == 9999) *(int*)b.a[0] = 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s[4];
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    hipEvent_t ev[8];
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int N = 2000;
    hipFunction_t fn = nullptr;
    hipGetFuncBySymbol(&fn, (const void*)k_empty);
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s[0], nullptr);
        double t1 = now();
        hipStreamSynchronize(s[0]);
        Big b{};
        double t2 = now();
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s[0], b);
        double t3 = now();
        hipStreamSynchronize(s[0]);
        void* arg = nullptr; size_t bytes = sizeof(arg);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &arg, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
        double t4 = now();
        if (fn) for (int i = 0; i < N; i++) hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, s[0], nullptr, extra);
        double t5 = now();
        hipStreamSynchronize(s[0]);
        double t6 = now();
        for (int i = 0; i < N; i++) hipEventRecord(ev[i & 7], s[i & 3]);
        double t7 = now();
        for (int i = 0; i < N; i++) hipStreamWaitEvent(s[(i + 1) & 3], ev[i & 7], 0);
        double t8 = now();
        for (int i = 0; i < N; i++) (void)hipEventQuery(ev[i & 7]);
        double t9 = now();
        for (auto& x : s) hipStreamSynchronize(x);
        double t10 = now();
        for (int i = 0; i < N; i++) hipEventSynchronize(ev[i & 7]);
        double t11 = now();
        for (int i = 0; i < N; i++) hipSetDevice(0);
        double t12 = now();
        // a frame-like pattern: launches alternating over four streams with an event record + wait between them
        double t13 = now();
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s[i & 3], nullptr);
            if ((i & 3) == 3) { hipEventRecord(ev[i & 7], s[i & 3]); hipStreamWaitEvent(s[(i + 1) & 3], ev[i & 7], 0); }
        }
        double t14 = now();
        for (auto& x : s) hipStreamSynchronize(x);
        if (rep) printf("us per call: launch(GGL, 1 arg) %.2f  launch(GGL, 128-byte struct) %.2f  hipModuleLaunchKernel %.2f (fn %p)  eventRecord %.2f  streamWaitEvent %.2f  "
                        "eventQuery %.2f  eventSynchronize(done) %.2f  setDevice %.2f  launches over 4 streams with record+wait every 4th %.2f per launch\n",
                        (t1 - t0) / N, (t3 - t2) / N, (t5 - t4) / N, (void*)fn, (t7 - t6) / N, (t8 - t7) / N, (t9 - t8) / N, (t11 - t10) / N, (t12 - t11) / N, (t14 - t13) / N);
    }
    return 0;
}
