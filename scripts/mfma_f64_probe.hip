// mfma_f64_probe.hip -- checks the operand / result lane maps of v_mfma_f64_16x16x4_f64 on gfx950 with random data and
// times it (the RTS smoother's two n x n products run on it, rbis_smooth.hpp step 5).
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_f64_probe.hip -o mfma_f64_probe && ./mfma_f64_probe
// Maps checked: A: lane l holds A[l & 15][l >> 4]; B: lane l holds B[l >> 4][l & 15];
//               C/D: lane l, register v holds D[(l >> 4) + 4 v][l & 15].
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4_t __attribute__((ext_vector_type(4)));

__global__ void k_one(const double *a, const double *b, const double *c, double *d)
{
  const int l = threadIdx.x;
  d4_t acc = { c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3] };
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
  for (int v = 0; v < 4; v++) d[4 * l + v] = acc[v];
}

template <int CHAINS>
__global__ void k_time(double *out, int iters, long long *cyc)
{
  const int l = threadIdx.x;
  d4_t acc[CHAINS];
  for (int i = 0; i < CHAINS; i++) acc[i] = d4_t{ 0.0, 0.0, 0.0, 0.0 };
  const double a = 1e-3 * l, b = 1.0 - 1e-3 * l;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < CHAINS; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + l] = s;
  if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
  std::vector<double> A(64), Bm(64), C(256), D(256);
  srand(7);
  for (auto &x : A) x = rand() / (double) RAND_MAX - 0.5;
  for (auto &x : Bm) x = rand() / (double) RAND_MAX - 0.5;
  for (auto &x : C) x = rand() / (double) RAND_MAX - 0.5;
  double *da, *db, *dc, *dd;
  hipMalloc(&da, 64 * 8); hipMalloc(&db, 64 * 8); hipMalloc(&dc, 256 * 8); hipMalloc(&dd, 256 * 8);
  hipMemcpy(da, A.data(), 64 * 8, hipMemcpyHostToDevice);
  hipMemcpy(db, Bm.data(), 64 * 8, hipMemcpyHostToDevice);
  hipMemcpy(dc, C.data(), 256 * 8, hipMemcpyHostToDevice);
  k_one<<<1, 64>>>(da, db, dc, dd);
  hipMemcpy(D.data(), dd, 256 * 8, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int l = 0; l < 64; l++)
    for (int v = 0; v < 4; v++) {
      const int i = (l >> 4) + 4 * v, j = l & 15;
      double ref = C[4 * l + v];
      for (int k = 0; k < 4; k++) ref += A[k * 16 + i] * Bm[k * 16 + j];  // A[i][k] sits in lane k*16+i, B[k][j] in lane k*16+j
      worst = fmax(worst, fabs(ref - D[4 * l + v]));
    }
  printf("v_mfma_f64_16x16x4_f64 lane maps: max |D - (C + A B)| = %.3g  (%s)\n", worst, worst < 1e-14 ? "OK" : "MISMATCH");
  long long *dcyc, cyc;
  double *dout;
  hipMalloc(&dcyc, 8); hipMalloc(&dout, 1024 * 256 * 8);
  const int iters = 4096;
  k_time<1><<<1, 64>>>(dout, iters, dcyc);
  hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  printf("one wave, dependent chain: %.1f clocks (s_memtime units) per MFMA\n", (double) cyc / iters);
  k_time<4><<<1, 64>>>(dout, iters, dcyc);
  hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  printf("one wave, 4 independent chains: %.1f per MFMA\n", (double) cyc / (4.0 * iters));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k_time<4><<<1024, 256>>>(dout, iters, dcyc);
  hipEventRecord(e0);
  k_time<4><<<1024, 256>>>(dout, iters, dcyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 1024.0 * 4 * iters * 4 * 2.0 * 16 * 16 * 4;
  printf("full chip (1024 x 4 waves x 4 chains): %.2f TFLOP/s fp64 on the matrix pipe\n", flops / (ms * 1e-3) / 1e12);
  return worst < 1e-14 ? 0 : 1;
}
