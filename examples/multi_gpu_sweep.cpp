// multi_gpu_sweep.cpp -- the filter-range split of BASELINE config 4 as a plain C++ host program on the C ABI: one
// pb_ctx per shard, one host thread per shard, NO data-path exchange, and ONE RCCL all-reduce (called directly, over xGMI
// between the GPUs of a node) of the 4-double end-of-run summary.  This is the C++ shape of the reference's own
// many-independent-filters workloads (state-estimator/python/param_sweep.py:39-52 runs 8 000 estimators one after the
// other; motion_estimate/scripts/se-batch-process.sh:58-59 replays 8 logs one after the other).
//
//   g++ -std=c++17 -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/multi_gpu_sweep.cpp \
//       -Lpronto_amd/lib -lpronto_batch -L/opt/rocm/lib -lrccl -lamdhip64 -lpthread -o multi_gpu_sweep
//   ./multi_gpu_sweep [filters_per_device=32768] [steps=200] [shards_per_device=1]
//
// Runs on every visible device (1 is fine: a one-rank communicator).  With shards_per_device > 1 several contexts share a
// device, each driven by its own thread (the ABI's threading contract: one context = one thread, distinct contexts may
// run on distinct threads); a device's shards are reduced on the host first, then the device leaders meet in RCCL.
// Check built in: the same job run as ONE context on device 0 (when it fits) must give the same summary.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "pronto_batch.h"

namespace {

// counter-based generator: every sample is a pure function of (global filter id, step, channel), so a shard is a slice
inline uint64_t mix(uint64_t x)
{
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
inline double uni(uint64_t b, uint64_t k, uint64_t ch)
{
  const uint64_t u = mix(mix(0x50524F4E544Full ^ b) + k * 0x9E3779B97F4A7C15ull + ch * 0xD1B54A32D192ED03ull);
  return ((u >> 11) + 0.5) / 9007199254740992.0;
}
inline double nrm(uint64_t b, uint64_t k, uint64_t ch)
{
  return std::sqrt(-2.0 * std::log(uni(b, k, 2 * ch))) * std::cos(2.0 * M_PI * uni(b, k, 2 * ch + 1));
}

struct Shard {
  int device = 0, b0 = 0, B = 0;
  double summary[4] = { 0, 0, 0, 0 };
  float ms = 0;
  int rc = 0;
  char err[512] = "";
};

// a standing robot swaying in yaw: gyro_z = per-filter sinusoid, specific force = g up, leg odometry = zero velocity
void fill_inputs(int b0, int B, int T, std::vector<double> &imu, std::vector<double> &lo, std::vector<uint8_t> &mask)
{
  const double g = 9.80665, dt = 1e-3;
  imu.resize((size_t) T * 7 * B);
  lo.resize((size_t) T * 6 * B);
  mask.resize((size_t) T * B);
  for (int k = 0; k < T; k++)
    for (int b = 0; b < B; b++) {
      const uint64_t gb = (uint64_t) (b0 + b);
      const double f = 0.2 + 1.8 * uni(gb, 1ull << 40, 0);
      double *im = &imu[(size_t) k * 7 * B], *l = &lo[(size_t) k * 6 * B];
      im[0 * (size_t) B + b] = 0.0087 * nrm(gb, k, 0);
      im[1 * (size_t) B + b] = 0.0087 * nrm(gb, k, 1);
      im[2 * (size_t) B + b] = 0.3 * std::sin(2 * M_PI * f * k * dt) + 0.0087 * nrm(gb, k, 2);
      im[3 * (size_t) B + b] = 0.1 * nrm(gb, k, 3);
      im[4 * (size_t) B + b] = 0.1 * nrm(gb, k, 4);
      im[5 * (size_t) B + b] = g + 0.1 * nrm(gb, k, 5);
      im[6 * (size_t) B + b] = dt;
      for (int i = 0; i < 3; i++) {
        l[(size_t) i * B + b] = 0.1 * nrm(gb, k, 10 + i);
        l[(size_t) (3 + i) * B + b] = 0.01;
      }
      mask[(size_t) k * B + b] = uni(gb, k, 40) > 0.09;  // ~9 % of the leg-odometry messages return NULL
    }
}

// one shard, start to finish, on the calling thread
void run_shard(Shard &s, int T)
{
  pb_ctx *ctx = nullptr;
  auto fail = [&](const char *what) {
    s.rc = 1;
    std::snprintf(s.err, sizeof s.err, "%s: %s", what, pb_last_error(ctx));
    if (ctx) pb_destroy(ctx);
  };
  if (pb_create(&ctx, 15, s.B, s.device, 0) != PB_OK) return fail("pb_create");
  double x0[15] = { 0 }, q0[4] = { 1, 0, 0, 0 }, P0[225] = { 0 };
  for (int i = 3; i < 12; i++) P0[i * 15 + i] = (i < 6) ? 0.0225 : (i < 9 ? 0.0027 : 0.25);
  if (pb_reset(ctx, x0, q0, P0, 1, PB_HOST) != PB_OK) return fail("pb_reset");
  std::vector<double> imu, lo;
  std::vector<uint8_t> mask;
  fill_inputs(s.b0, s.B, T, imu, lo, mask);
  void *d_imu = nullptr, *d_lo = nullptr, *d_mask = nullptr;
  if (pb_malloc(ctx, imu.size() * 8, &d_imu) || pb_malloc(ctx, lo.size() * 8, &d_lo) || pb_malloc(ctx, mask.size(), &d_mask))
    return fail("pb_malloc");
  if (pb_memcpy_h2d(ctx, d_imu, imu.data(), imu.size() * 8) || pb_memcpy_h2d(ctx, d_lo, lo.data(), lo.size() * 8) ||
      pb_memcpy_h2d(ctx, d_mask, mask.data(), mask.size()))
    return fail("pb_memcpy_h2d");
  const double q[4] = { 7.6e-5, 0.01, 0, 0 };
  if (pb_run_legodo(ctx, T, (const double *) d_imu, (const double *) d_lo, (const uint8_t *) d_mask, q, &s.ms) != PB_OK)
    return fail("pb_run_legodo");
  if (pb_summary(ctx, s.summary) != PB_OK) return fail("pb_summary");
  pb_free(ctx, d_imu); pb_free(ctx, d_lo); pb_free(ctx, d_mask);
  pb_destroy(ctx);
}

#define HIPOK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)
#define NCCLOK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(r_)); return 3; } } while (0)

}  // namespace

int main(int argc, char **argv)
{
  const int per_dev = argc > 1 ? std::atoi(argv[1]) : 32768;
  const int T = argc > 2 ? std::atoi(argv[2]) : 200;
  const int spd = argc > 3 ? std::atoi(argv[3]) : 1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    std::fprintf(stderr, "multi_gpu_sweep: no HIP device visible (the library has no CPU path)\n");
    return 2;
  }
  if (per_dev < spd || spd < 1 || T < 1) return 2;
  const int total = per_dev * ndev, nshard = ndev * spd;

  // ---- the split: contiguous filter ranges, remainders to the lowest shards (pronto_amd/shard.py: shard_range) ----
  std::vector<Shard> shards(nshard);
  for (int r = 0; r < nshard; r++) {
    const int base = total / nshard, rem = total % nshard;
    shards[r].b0 = r * base + std::min(r, rem);
    shards[r].B = base + (r < rem ? 1 : 0);
    shards[r].device = r / spd;
  }
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (auto &s : shards) th.emplace_back(run_shard, std::ref(s), T);
  for (auto &t : th) t.join();
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (auto &s : shards)
    if (s.rc) { std::fprintf(stderr, "shard at filter %d on device %d failed: %s\n", s.b0, s.device, s.err); return 1; }

  // ---- per device: host reduction of its shards (rank order), then ONE RCCL all-reduce between the devices ----
  std::vector<ncclComm_t> comm(ndev);
  std::vector<int> devs(ndev);
  for (int d = 0; d < ndev; d++) devs[d] = d;
  NCCLOK(ncclCommInitAll(comm.data(), ndev, devs.data()));
  std::vector<hipStream_t> st(ndev);
  std::vector<double *> dsum(ndev), dmax(ndev);
  for (int d = 0; d < ndev; d++) {
    double sums[3] = { 0, 0, 0 }, mx = 0;
    for (int r = d * spd; r < (d + 1) * spd; r++) {
      sums[0] += shards[r].summary[0]; sums[1] += shards[r].summary[1]; sums[2] += shards[r].summary[3];
      mx = std::max(mx, shards[r].summary[2]);
    }
    HIPOK(hipSetDevice(d));
    HIPOK(hipStreamCreate(&st[d]));
    HIPOK(hipMalloc((void **) &dsum[d], 3 * sizeof(double)));
    HIPOK(hipMalloc((void **) &dmax[d], sizeof(double)));
    HIPOK(hipMemcpyAsync(dsum[d], sums, sizeof sums, hipMemcpyHostToDevice, st[d]));
    HIPOK(hipMemcpyAsync(dmax[d], &mx, sizeof mx, hipMemcpyHostToDevice, st[d]));
    HIPOK(hipStreamSynchronize(st[d]));  // sums / mx are stack variables
  }
  NCCLOK(ncclGroupStart());
  for (int d = 0; d < ndev; d++) {
    NCCLOK(ncclAllReduce(dsum[d], dsum[d], 3, ncclDouble, ncclSum, comm[d], st[d]));
    NCCLOK(ncclAllReduce(dmax[d], dmax[d], 1, ncclDouble, ncclMax, comm[d], st[d]));
  }
  NCCLOK(ncclGroupEnd());
  double job[4] = { 0, 0, 0, 0 };
  for (int d = 0; d < ndev; d++) {
    double sums[3], mx;
    HIPOK(hipSetDevice(d));
    HIPOK(hipMemcpyAsync(sums, dsum[d], sizeof sums, hipMemcpyDeviceToHost, st[d]));
    HIPOK(hipMemcpyAsync(&mx, dmax[d], sizeof mx, hipMemcpyDeviceToHost, st[d]));
    HIPOK(hipStreamSynchronize(st[d]));
    const double got[4] = { sums[0], sums[1], mx, sums[2] };
    if (d == 0) std::copy(got, got + 4, job);
    else if (!std::equal(got, got + 4, job)) { std::fprintf(stderr, "device %d disagrees after the all-reduce\n", d); return 1; }
    HIPOK(hipFree(dsum[d])); HIPOK(hipFree(dmax[d])); HIPOK(hipStreamDestroy(st[d]));
    ncclCommDestroy(comm[d]);
  }
  float ms = 0;
  for (auto &s : shards) ms = std::max(ms, s.ms);
  std::printf("%d filters = %d device(s) x %d shard(s) x %d steps: kernels %.2f ms (max over shards) = %.3g steps/s; "
              "end to end incl. input generation %.2f s\n", total, ndev, spd, T, ms, (double) total * T / (ms * 1e-3), wall);
  std::printf("job summary (RCCL all-reduce): sum_loglik %.17g checksum %.17g max|q^2-1| %.3g nonfinite %.0f\n", job[0],
              job[1], job[2], job[3]);

  // ---- check: the same filters as ONE context on device 0 ----
  bool ok = job[3] == 0 && job[2] < 1e-12;
  if ((long) total * 1120 < (3L << 30)) {
    Shard whole;
    whole.B = total;
    run_shard(whole, T);
    if (whole.rc) { std::fprintf(stderr, "whole-batch run failed: %s\n", whole.err); return 1; }
    const double e0 = std::fabs(whole.summary[0] - job[0]) / std::max(1.0, std::fabs(whole.summary[0]));
    const double e1 = std::fabs(whole.summary[1] - job[1]) / std::max(1.0, std::fabs(whole.summary[1]));
    std::printf("whole batch on device 0: sum_loglik %.17g checksum %.17g (rel diff %.2g, %.2g)\n", whole.summary[0],
                whole.summary[1], e0, e1);
    ok = ok && e0 < 1e-12 && e1 < 1e-12 && whole.summary[3] == 0;
  }
  std::printf(ok ? "PASS\n" : "FAIL\n");
  return ok ? 0 : 1;
}
