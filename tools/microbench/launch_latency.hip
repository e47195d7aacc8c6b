// Host-side cost of one tiny kernel call: launch + hipStreamSynchronize vs launch + spinning on a flag the kernel writes
// into pinned host memory (what the zero-copy small-batch path could do instead of a stream synchronisation).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_flag(volatile uint32_t* flag, uint32_t v, uint32_t* sink) {
  if (threadIdx.x == 0) { sink[0] = v; __threadfence_system(); *flag = v; }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint32_t* flag; CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
  uint32_t* dflag; CK(hipHostGetDevicePointer((void**)&dflag, flag, 0));
  uint32_t* sink; CK(hipMalloc(&sink, 64));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  *flag = 0;
  const int R = 2000;
  std::vector<double> a, b, c, d;
  for (int i = 1; i <= R; ++i) {
    double t0 = now_us();
    hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, dflag, (uint32_t)i, sink);
    double t1 = now_us();
    CK(hipStreamSynchronize(st));
    double t2 = now_us();
    a.push_back(t1 - t0); b.push_back(t2 - t0);
  }
  for (int i = R + 1; i <= 2 * R; ++i) {
    double t0 = now_us();
    hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, dflag, (uint32_t)i, sink);
    while (*(volatile uint32_t*)flag != (uint32_t)i) { }
    double t2 = now_us();
    c.push_back(t2 - t0);
  }
  CK(hipStreamSynchronize(st));
  for (int i = 2 * R + 1; i <= 3 * R; ++i) {
    double t0 = now_us();
    hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, dflag, (uint32_t)i, sink);
    CK(hipEventRecord(ev, st));
    while (hipEventQuery(ev) == hipErrorNotReady) { }
    double t2 = now_us();
    d.push_back(t2 - t0);
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("{\"launch_call_us\": %.2f, \"launch_plus_stream_sync_us\": %.2f, \"launch_plus_flag_spin_us\": %.2f, \"launch_plus_event_query_spin_us\": %.2f}\n", med(a), med(b), med(c), med(d));
  return 0;
}
