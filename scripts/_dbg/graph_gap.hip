// Micro-benchmark: N dependent tiny kernels per "iteration", launched (a) one by one on a stream, (b) as one captured
// hipGraph. Prints microseconds per kernel for both. Build: hipcc --offload-arch=gfx950 -O2 graph_gap.hip -o graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(double* p, int n) { const int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) p[t] = p[t] * 1.0000001 + 1e-9; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int N = 24, ITER = 200, n = 4096;
  double* d; hipMalloc(&d, n * 8); hipMemset(d, 0, n * 8);
  hipStream_t st; hipStreamCreate(&st);
  for (int w = 0; w < 3; ++w) for (int k = 0; k < N; ++k) hipLaunchKernelGGL(k_tiny, dim3(16), dim3(256), 0, st, d, n);
  hipStreamSynchronize(st);
  double t0 = now();
  for (int it = 0; it < ITER; ++it) for (int k = 0; k < N; ++k) hipLaunchKernelGGL(k_tiny, dim3(16), dim3(256), 0, st, d, n);
  hipStreamSynchronize(st);
  double t1 = now();
  std::printf("stream: %.2f us per kernel (%d kernels x %d)\n", 1e6 * (t1 - t0) / (N * ITER), N, ITER);
  // with a host sync per iteration (like the LM loop's read-back)
  t0 = now();
  for (int it = 0; it < ITER; ++it) { for (int k = 0; k < N; ++k) hipLaunchKernelGGL(k_tiny, dim3(16), dim3(256), 0, st, d, n); hipStreamSynchronize(st); }
  t1 = now();
  std::printf("stream + sync per iteration: %.2f us per kernel, %.1f us per iteration\n", 1e6 * (t1 - t0) / (N * ITER), 1e6 * (t1 - t0) / ITER);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int k = 0; k < N; ++k) hipLaunchKernelGGL(k_tiny, dim3(16), dim3(256), 0, st, d, n);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  t0 = now();
  for (int it = 0; it < ITER; ++it) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  t1 = now();
  std::printf("graph: %.2f us per kernel\n", 1e6 * (t1 - t0) / (N * ITER));
  t0 = now();
  for (int it = 0; it < ITER; ++it) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
  t1 = now();
  std::printf("graph + sync per iteration: %.2f us per kernel, %.1f us per iteration\n", 1e6 * (t1 - t0) / (N * ITER), 1e6 * (t1 - t0) / ITER);
  return 0;
}
