// Host cost of three dependent kernel launches: plain hipLaunchKernelGGL x 3 against hipGraphLaunch of the captured
// chain (is a HIP graph worth it for the tracked frame's detect chain?).   hipcc --offload-arch=gfx950 -O2 graph_launch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(int* p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += v; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int* d;
  hipMalloc(&d, 64);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto chain = [&]() {
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, d, 1);
    hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, s, d, 2);
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, d, 3);
  };
  for (int i = 0; i < 200; i++) chain();
  hipStreamSynchronize(s);
  const int N = 2000;
  double t0 = now_us();
  for (int i = 0; i < N; i++) chain();
  double t1 = now_us();
  hipStreamSynchronize(s);
  double t2 = now_us();
  printf("3 plain launches: host %.2f us per chain (enqueue only), %.2f us per chain incl. drain\n", (t1 - t0) / N, (t2 - t0) / N);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  chain();
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 200; i++) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  t0 = now_us();
  for (int i = 0; i < N; i++) hipGraphLaunch(ge, s);
  t1 = now_us();
  hipStreamSynchronize(s);
  t2 = now_us();
  printf("graph of the 3:   host %.2f us per chain (enqueue only), %.2f us per chain incl. drain\n", (t1 - t0) / N, (t2 - t0) / N);
  // latency of one chain + sync (what a caller that needs the result waits for)
  t0 = now_us();
  for (int i = 0; i < N; i++) { chain(); hipStreamSynchronize(s); }
  t1 = now_us();
  for (int i = 0; i < N; i++) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
  t2 = now_us();
  printf("chain + sync: plain %.2f us, graph %.2f us\n", (t1 - t0) / N, (t2 - t1) / N);
  return 0;
}
