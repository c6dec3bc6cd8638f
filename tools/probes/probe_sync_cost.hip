// tools/probes/probe_sync_cost.hip — what asking an IDLE stream costs the host: hipStreamSynchronize, hipStreamQuery, hipEventQuery / hipEventSynchronize on a
// finished event, hipEventRecord, an empty kernel launch; and the same with a second thread launching kernels on another stream (the block chain's case)
// hipcc -O2 --offload-arch=gfx950 -o /tmp/probe_sync_cost tools/probes/probe_sync_cost.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void k_nop() {}
template <class F> static double per_call_us(F f, int n = 20000)
{
  for (int i = 0; i < 100; i++) f();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++) f();
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main()
{
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e;
  hipEventCreateWithFlags(&e, hipEventDisableTiming);
  hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s);
  hipEventRecord(e, s);
  hipStreamSynchronize(s);
  for (int busy = 0; busy < 2; busy++) {
    std::atomic<bool> stop{false};
    std::thread other;
    if (busy) other = std::thread([&] { while (!stop) { hipLaunchKernelGGL(k_nop, dim3(64), dim3(256), 0, s2); if (hipStreamQuery(s2) != hipSuccess) (void)hipGetLastError(); } hipStreamSynchronize(s2); });
    printf("%s\n", busy ? "-- a second thread launching on another stream --" : "-- alone --");
    printf("hipStreamSynchronize (idle stream)  %.2f us\n", per_call_us([&] { hipStreamSynchronize(s); }));
    printf("hipStreamQuery       (idle stream)  %.2f us\n", per_call_us([&] { (void)hipStreamQuery(s); }));
    printf("hipEventQuery        (finished)     %.2f us\n", per_call_us([&] { (void)hipEventQuery(e); }));
    printf("hipEventSynchronize  (finished)     %.2f us\n", per_call_us([&] { hipEventSynchronize(e); }));
    printf("hipSetDevice                        %.2f us\n", per_call_us([&] { hipSetDevice(0); }));
    printf("launch + hipStreamSynchronize       %.2f us\n", per_call_us([&] { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s); hipStreamSynchronize(s); }, 2000));
    printf("launch (enqueue only, 64 then sync) %.2f us\n", per_call_us([&] { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s); }, 64));
    hipStreamSynchronize(s);
    printf("hipEventRecord (then idle)          %.2f us\n", per_call_us([&] { hipEventRecord(e, s); }, 2000));
    hipStreamSynchronize(s);
    if (busy) { stop = true; other.join(); }
  }
  return 0;
}
