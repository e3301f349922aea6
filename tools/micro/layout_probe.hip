// Micro-benchmark: does the instance-fastest SoA layout ([t][component][Bp], component stride = 8*Bp bytes) cost the per-instance sweep
// kernels (k_step: one lane per instance marching over the knots) throughput against a wave-tiled layout ([b/64][t][component][64])?
// Same loads/stores/flop shape as k_step's backward sweep: per knot 30 loads, ~300 dependent FMAs, 20 stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <bool TILED>
__global__ __launch_bounds__(64, 2) void sweep(const double* __restrict__ in, double* __restrict__ out, int B, int Bp, int T) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  constexpr int KI = 30, KO = 20;
  auto idx = [&](int t, int K, int i) -> size_t {
    if (TILED) return ((((size_t)(b >> 6) * T + t) * K + i) << 6) + (b & 63);
    return ((size_t)t * K + i) * Bp + b;
  };
  double acc[4] = {1.0, 0.5, 0.25, 0.125};
  double nx[KI];
#pragma unroll
  for (int i = 0; i < KI; ++i) nx[i] = in[idx(T - 1, KI, i)];
  for (int t = T - 1; t >= 0; --t) {
    double v[KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) v[i] = nx[i];
    if (t > 0) {
#pragma unroll
      for (int i = 0; i < KI; ++i) nx[i] = in[idx(t - 1, KI, i)];
    }
#pragma unroll
    for (int r = 0; r < 10; ++r)
#pragma unroll
      for (int i = 0; i < KI; ++i) acc[i & 3] = acc[i & 3] * 0.999 + v[i] * acc[(i + 1) & 3];
#pragma unroll
    for (int i = 0; i < KO; ++i) out[idx(t, KO, i)] = acc[i & 3] + v[i];
  }
}
int main() {
  const int T = 48;
  for (int B : {131072, 32768}) {
    const int Bp = B;
    double *in, *out;
    hipMalloc(&in, sizeof(double) * (size_t)Bp * T * 30);
    hipMalloc(&out, sizeof(double) * (size_t)Bp * T * 20);
    hipMemset(in, 0, sizeof(double) * (size_t)Bp * T * 30);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int tiled = 0; tiled < 2; ++tiled) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (tiled) hipLaunchKernelGGL(sweep<true>, dim3(B / 64), dim3(64), 0, 0, in, out, B, Bp, T);
        else hipLaunchKernelGGL(sweep<false>, dim3(B / 64), dim3(64), 0, 0, in, out, B, Bp, T);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double gb = (double)B * T * 50 * 8 / 1e9;
      printf("B=%d %s: %.3f ms  %.2f TB/s\n", B, tiled ? "wave-tiled [b/64][t][k][64]" : "instance-fastest [t][k][Bp]", best, gb / best);
    }
    hipFree(in); hipFree(out);
  }
  return 0;
}
