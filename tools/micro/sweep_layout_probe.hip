// Stand-alone probe (development, not part of the library): the memory access pattern of k_step_zc -- one lane per instance, a backward pass over the knots that
// reads NR rows per knot (one knot prefetched) and writes NG rows of gains, then a forward pass that reads the gains back and writes NZ rows -- under two layouts of
// the stage arrays:
//   0  row-major   a[(t * K + k) * Bp + b]                       (what the library uses: a wavefront touches K pieces of 512 B, 8 Bp bytes apart)
//   1  tile-major  a[((b / 64) * T * K + t * K + k) * 64 + b % 64]  (a wavefront's rows of one knot are one contiguous run of K * 512 B, its knots follow each other)
// One wavefront per SIMD (40 KB of dynamic LDS per 64-thread block), FLOPS dependent multiply-adds per knot stand in for the Riccati step.
//   hipcc --offload-arch=gfx950 -O3 -o sweep_layout_probe sweep_layout_probe.hip && ./sweep_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LAYOUT>
__device__ inline size_t at(const int t, const int K, const int k, const int T, const size_t Bp, const int b) {
  if (LAYOUT == 0) return ((size_t)t * K + k) * Bp + b;
  return (((size_t)(b >> 6) * T + t) * K + k) * 64 + (b & 63);
}

template <int LAYOUT, int NR, int NG, int NZ, int FLOPS>
__global__ __launch_bounds__(64) void k_sweep(const double* __restrict__ a, double* __restrict__ g, double* __restrict__ z, const int T, const int B, const size_t Bp) {
  extern __shared__ double pad[];
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double cur[NR], nxt[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) nxt[k] = a[at<LAYOUT>(T - 1, NR, k, T, Bp, b)];
  double s = 1.0;
  for (int t = T - 1; t >= 0; --t) {
#pragma unroll
    for (int k = 0; k < NR; ++k) cur[k] = nxt[k];
    if (t > 0) {
#pragma unroll
      for (int k = 0; k < NR; ++k) nxt[k] = a[at<LAYOUT>(t - 1, NR, k, T, Bp, b)];
    }
    double acc = s;
#pragma unroll
    for (int k = 0; k < NR; ++k) acc = fma(cur[k], 1e-3, acc);
#pragma unroll 8
    for (int i = 0; i < FLOPS; ++i) acc = fma(acc, 0.999999, 1e-9);
    s = acc;
#pragma unroll
    for (int k = 0; k < NG; ++k) g[at<LAYOUT>(t, NG, k, T, Bp, b)] = acc + k;
  }
  double gn[NG];
#pragma unroll
  for (int k = 0; k < NG; ++k) gn[k] = g[at<LAYOUT>(0, NG, k, T, Bp, b)];
  for (int t = 0; t < T; ++t) {
    double gc[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) gc[k] = gn[k];
    if (t + 1 < T) {
#pragma unroll
      for (int k = 0; k < NG; ++k) gn[k] = g[at<LAYOUT>(t + 1, NG, k, T, Bp, b)];
    }
    double acc = s;
#pragma unroll
    for (int k = 0; k < NG; ++k) acc = fma(gc[k], 1e-3, acc);
#pragma unroll 8
    for (int i = 0; i < FLOPS / 6; ++i) acc = fma(acc, 0.999999, 1e-9);
    s = acc;
#pragma unroll
    for (int k = 0; k < NZ; ++k) z[at<LAYOUT>(t, NZ, k, T, Bp, b)] = acc + k;
  }
}

template <int LAYOUT, int FLOPS>
double run(const int B, const int T, const int reps) {
  constexpr int NR = 35, NG = 20, NZ = 4;
  const size_t Bp = (size_t)B + 13 * 64;
  double *a, *g, *z;
  hipMalloc(&a, sizeof(double) * Bp * T * NR);
  hipMalloc(&g, sizeof(double) * Bp * T * NG);
  hipMalloc(&z, sizeof(double) * Bp * T * NZ);
  hipMemset(a, 0, sizeof(double) * Bp * T * NR);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&]() { hipLaunchKernelGGL((k_sweep<LAYOUT, NR, NG, NZ, FLOPS>), dim3((B + 63) / 64), dim3(64), 40 * 1024, 0, a, g, z, T, B, Bp); };
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(a); hipFree(g); hipFree(z);
  const double bytes = (double)B * T * (NR + 2 * NG + NZ) * 8.0;
  return bytes / (ms / reps * 1e-3) / 1e12;
}

int main() {
  const int T = 48;
  for (int B : {262144, 131072}) {
    printf("B %d  flops/knot 250: row-major %.2f TB/s  tile-major %.2f TB/s\n", B, run<0, 250>(B, T, 5), run<1, 250>(B, T, 5));
    printf("B %d  flops/knot 0  : row-major %.2f TB/s  tile-major %.2f TB/s\n", B, run<0, 0>(B, T, 5), run<1, 0>(B, T, 5));
  }
  return 0;
}
