// pronto_batch.hip -- host side of the C ABI declared in include/pronto_batch.h: context, staging, launches.
// The kernels live in rbis_kernels.hpp, the per-filter arithmetic in rbis_device.hpp.
// There is no CPU path here: without a gfx950 device pb_create fails with PB_ERR_NO_DEVICE.
#include <algorithm>
#include <new>

#include "pb_ctx.hpp"
#include "rbis_frontend.hpp"

#define PB_VERSION_STR "pronto_batch 0.3 gfx950"

extern "C" const char *pb_version(void) { return PB_VERSION_STR; }

extern "C" const char *pb_last_error(const pb_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

extern "C" int pb_create(pb_ctx **out, int n_states, int batch, int device, int n_snapshots)
{
  if (!out) return fail(nullptr, PB_ERR_ARG, "pb_create: out is NULL");
  *out = nullptr;
  if (n_states != 15 && n_states != 21) return fail(nullptr, PB_ERR_ARG, "pb_create: n_states must be 15 or 21");
  if (batch <= 0 || n_snapshots < 0) return fail(nullptr, PB_ERR_ARG, "pb_create: bad batch / n_snapshots");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, PB_ERR_NO_DEVICE, "pb_create: no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(nullptr, PB_ERR_ARG, "pb_create: device %d out of range", device);
  hipDeviceProp_t prop;
  HIPCHK(nullptr, hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, PB_ERR_NO_DEVICE, "pb_create: device %d is %s; kernels are built for gfx950 only", device,
                prop.gcnArchName);
  pb_ctx *c = new (std::nothrow) pb_ctx();
  if (!c) return fail(nullptr, PB_ERR_ARG, "pb_create: out of host memory");
  c->ns = n_states;
  c->B = batch;
  c->dev = device;
  c->nsnap = n_snapshots;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  {
    // 15-state hot kernel: the two-wave cooperative mapping (2 waves/SIMD, reads and writes interleave) measured
    // 3-20 % faster up to 256k filters, the one-lane-per-filter k_step 3-5 % faster beyond (profiles/).
    // PRONTO_BATCH_COOP15=0/1 forces one of them (A/B runs and tests).
    const char *e = getenv("PRONTO_BATCH_COOP15");
    c->coop15 = e ? (e[0] == '1') : (batch <= 393216);
    // Below ~48k filters the two-wave kernel's 64-filter tiles leave workgroup slots empty (1 024 slots: 256 CUs x 4 workgroups of two
    // waves at two waves per SIMD; 32 768 filters = 512 tiles).  Two workgroups per tile, 32 filters each, fill them -- and change
    // nothing: 13.12 against 13.14 us at 32 768 filters, slower everywhere else (profiles/r05_half_tile.txt).  The step time is
    // ~5 us of one tile's dependent chain + bytes / 9.4 TB/s at every size; more workgroups do not shorten the chain.  Kept as an A/B
    // switch (PRONTO_BATCH_HALF=1), off by default.
    const char *eh = getenv("PRONTO_BATCH_HALF");
    c->half15 = c->coop15 && n_states == 15 && eh && eh[0] == '1';
  }
  c->stride = ((long) batch + 63) / 64 * 64;
  c->nc = (n_states == 15) ? Lay<15>::NC : Lay<21>::NC;
  c->state_doubles = (size_t) c->stride * (size_t) ((n_states == 15) ? Slots<15>::NSLOT : Slots<21>::NSLOT);
  {
    // XCD-contiguous tile order: each of the 8 XCDs walks one contiguous range of tiles instead of every 8th tile.  With
    // the tiled layout it measured equal or faster for both state sizes at every batch size (n = 21 at 192k filters:
    // 162 -> 146 us).  PRONTO_BATCH_XCD=0/1 forces it either way for A/B runs.
    const char *e = getenv("PRONTO_BATCH_XCD");
    c->k.xcd_remap = e ? (e[0] == '1') : 1;
    // Cache policy of the state round trip (rbis_kernels.hpp MemHint), measured on both step kernels: a state that
    // fits the XCDs' L2s (< ~48 MB) wants the default policy (sc1 stores 7 % slower at 32k x 15 states); up to ~1.3x
    // the 256 MB memory-side cache sc1 stores are 1-4 % faster; beyond, non-temporal loads+stores are 7-15 % faster
    // (1M filters: 469 -> 417 us) and 10-40 % SLOWER if used on a cache-sized state.  PRONTO_BATCH_MEMHINT=0/1/2 forces.
    const long state_bytes = (long) c->state_doubles * 8;
    const char *gu = getenv("PRONTO_BATCH_GENERIC_UPDATE");
    c->generic_update = gu && gu[0] == '1';
    const char *q21 = getenv("PRONTO_BATCH_QUAD21");
    c->quad21 = !(q21 && q21[0] == '0');
    const char *h = getenv("PRONTO_BATCH_MEMHINT");
    c->mem_hint = h ? (h[0] - '0')
                    : (state_bytes < (48L << 20)    ? MH_DEFAULT
                       : state_bytes < (310L << 20) ? MH_STORE_SC1
                       : state_bytes < (350L << 20) ? MH_DEFAULT   // (160k 21-state filters, 336 MB: 112.6 us against 124.8 with sc1 stores, 118.7 non-temporal)
                                                    : MH_STREAM_NT);
    if (c->mem_hint < 0 || c->mem_hint > 2) c->mem_hint = MH_DEFAULT;
    // Bulk replays of a state that does not fit the memory-side cache (pb_run_legodo): filter range outer, time inner, over blocks
    // of whole tiles whose state stays cache-resident from step to step -- the filters are independent and the streams are known
    // up front (the reference's own many-runs workload replays one log 8 000 times, state-estimator/python/param_sweep.py:39-52).
    // Same T = 1 accounting: every step still loads and stores every posterior once, the round trip just ends in the cache.
    // Block size: the largest that measured at the cache-resident rate (profiles/r05_batch_sweep.txt), the blocks made equal;
    // each block runs the kernel and the cache policy of ITS size.  PRONTO_BATCH_BLOCKED=0 / 1 switches it off / on for any
    // size, PRONTO_BATCH_BLOCK_FILTERS=<n> names the block size.
    {
      const char *eb = getenv("PRONTO_BATCH_BLOCKED"), *ef = getenv("PRONTO_BATCH_BLOCK_FILTERS");
      const long per_filter = state_bytes / c->stride;
      // (15 states: 224k filters = 257 MB of state per block measured best, 0.87 of the roofline at 512k / 1 M filters with 64-step
      // streams against 0.74 step by step; 21 states: 112k = 237 MB, 0.84 against 0.67 -- profiles/r05_batch_sweep.txt)
      long want = ef ? atol(ef) : (n_states == 15 ? 229376 : 114688);
      want = (want + 63) / 64 * 64;
      const bool on = eb ? (eb[0] == '1') : (state_bytes > (256L << 20));
      if (on && want >= 64 && want < batch && (batch & 63) == 0) {
        const long nblocks = (batch + want - 1) / want;
        c->run_block = (int) (((batch + nblocks - 1) / nblocks + 63) / 64 * 64);
        const long blk_bytes = (long) c->run_block * per_filter;
        c->run_block_hint = h ? c->mem_hint : (blk_bytes < (48L << 20) ? MH_DEFAULT : MH_STORE_SC1);
        const char *e15 = getenv("PRONTO_BATCH_COOP15");
        c->run_block_coop15 = e15 ? (e15[0] == '1') : true;
      }
    }
  }
  // The kernels address the STATE through one buffer descriptor per 64-filter tile (64-bit tile base), so its size is
  // bounded by HBM only; the per-message INPUT blocks ([rows][B], at most 36 rows) go through one 32-bit-ranged
  // descriptor each, which bounds the batch of one context at 2^32 / (36 * 8) filters.
  if ((unsigned long long) batch * 36ull * 8ull >= (1ull << 32)) {
    delete c;
    return fail(nullptr, PB_ERR_ARG, "pb_create: batch %d too large for one context (input blocks must stay below 4 GiB; "
                "split the batch over several contexts)", batch);
  }
#define CRCHK(call)                                                                                 \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      fail(nullptr, PB_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));                      \
      pb_destroy(c);                                                                                \
      return PB_ERR_HIP;                                                                            \
    }                                                                                               \
  } while (0)
  CRCHK(hipSetDevice(device));
  CRCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  CRCHK(hipMalloc((void **) &c->st_base, sizeof(double) * c->state_doubles));
  c->st = c->st_base;
  CRCHK(hipMemsetAsync(c->st, 0, sizeof(double) * c->state_doubles, c->stream));
  if (n_snapshots > 0) {
    CRCHK(hipMalloc((void **) &c->snaps, sizeof(double) * 7 * c->stride * n_snapshots));
    CRCHK(hipMemsetAsync(c->snaps, 0, sizeof(double) * 7 * c->stride * n_snapshots, c->stream));
  }
  CRCHK(hipMalloc((void **) &c->d_small, sizeof(double) * 1024));
  CRCHK(hipEventCreate(&c->ev0));
  CRCHK(hipEventCreate(&c->ev1));
  CRCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  CRCHK(hipEventCreateWithFlags(&c->ev_consumed[0], hipEventDisableTiming));
  CRCHK(hipEventCreateWithFlags(&c->ev_consumed[1], hipEventDisableTiming));
  CRCHK(hipEventCreateWithFlags(&c->ev_copied, hipEventDisableTiming));
  CRCHK(hipStreamSynchronize(c->stream));
#undef CRCHK
  *out = c;
  return PB_OK;
}

extern "C" int pb_destroy(pb_ctx *c)
{
  if (!c) return PB_OK;
  (void) hipSetDevice(c->dev);
  if (c->stream) (void) hipStreamSynchronize(c->stream);
  if (c->st_base) (void) hipFree(c->st_base);
  if (c->snaps) (void) hipFree(c->snaps);
  if (c->hist) (void) hipFree(c->hist);
  if (c->notch) (void) hipFree(c->notch);
  if (c->ins_last) (void) hipFree(c->ins_last);
  if (c->ins_prev_ut) (void) hipFree(c->ins_prev_ut);
  if (c->imu_keep) (void) hipFree(c->imu_keep);
  for (int i = 0; i < c->n_fences; i++)
    if (c->fence[i]) (void) hipEventDestroy(c->fence[i]);
  if (c->ev_upload) (void) hipEventDestroy(c->ev_upload);
  if (c->legd) (void) hipFree(c->legd);
  if (c->legi) (void) hipFree(c->legi);
  if (c->leg_chain) (void) hipFree(c->leg_chain);
  if (c->leg_ut) (void) hipFree(c->leg_ut);
  if (c->leg_valid) (void) hipFree(c->leg_valid);
  if (c->leg_nc) (void) hipFree(c->leg_nc);
  if (c->leg_lo) (void) hipFree(c->leg_lo);
  if (c->jf_ring) (void) hipFree(c->jf_ring);
  if (c->jf_kst) (void) hipFree(c->jf_kst);
  if (c->d_small) (void) hipFree(c->d_small);
  if (c->stage) (void) hipFree(c->stage);
  if (c->copy_stream) (void) hipStreamSynchronize(c->copy_stream);
  for (int i = 0; i < 2; i++) {
    if (c->in_stage[i]) (void) hipFree(c->in_stage[i]);
    if (c->ev_consumed[i]) (void) hipEventDestroy(c->ev_consumed[i]);
  }
  if (c->ev_copied) (void) hipEventDestroy(c->ev_copied);
  if (c->copy_stream) (void) hipStreamDestroy(c->copy_stream);
  if (c->ev0) (void) hipEventDestroy(c->ev0);
  if (c->ev1) (void) hipEventDestroy(c->ev1);
  if (c->own_stream) (void) hipStreamDestroy(c->own_stream);
  delete c;
  return PB_OK;
}

extern "C" int pb_set_stream(pb_ctx *c, void *s)
{
  if (!c) return PB_ERR_ARG;
  if ((hipStream_t) s != c->stream) HIPCHK(c, hipStreamSynchronize(c->stream));  // staging buffers in flight belong to the old stream
  c->stream = (hipStream_t) s;  // literal handle: NULL is the (legacy) null stream, which is torch's default stream
  return PB_OK;
}

extern "C" int pb_use_own_stream(pb_ctx *c)
{
  if (!c) return PB_ERR_ARG;
  if (c->stream != c->own_stream) HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = c->own_stream;
  return PB_OK;
}

extern "C" int pb_set_constants(pb_ctx *c, double g, double chi_tol)
{
  if (!c) return PB_ERR_ARG;
  if (!(g > 0) || !(chi_tol >= 0)) return fail(c, PB_ERR_ARG, "pb_set_constants: g must be > 0 and chi_tol >= 0");
  c->k.g = g;
  c->k.chi_tol = chi_tol;
  return PB_OK;
}

extern "C" int pb_sync(pb_ctx *c)
{
  if (!c) return PB_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PB_OK;
}

extern "C" int pb_run_block(const pb_ctx *c) { return c ? c->run_block : 0; }

extern "C" const char *pb_hot_kernel(const pb_ctx *c)
{
  if (!c) return "";
  static const char *const names[4][3] = { { "k_step_quad<true,0>", "k_step_quad<true,1>", "k_step_quad<true,2>" },
                                           { "k_step_coop<21,true,0>", "k_step_coop<21,true,1>", "k_step_coop<21,true,2>" },
                                           { "k_step_coop<15,true,0>", "k_step_coop<15,true,1>", "k_step_coop<15,true,2>" },
                                           { "k_step<15,true,0>", "k_step<15,true,1>", "k_step<15,true,2>" } };
  return names[c->ns == 21 ? (c->quad21 ? 0 : 1) : (c->coop15 ? 2 : 3)][c->mem_hint];
}
extern "C" int pb_batch(const pb_ctx *c) { return c ? c->B : -1; }
extern "C" int pb_n_states(const pb_ctx *c) { return c ? c->ns : -1; }

extern "C" int pb_malloc(pb_ctx *c, uint64_t bytes, void **p)
{
  if (!c || !p) return PB_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipMalloc(p, bytes));
  return PB_OK;
}
extern "C" int pb_free(pb_ctx *c, void *p)
{
  if (!c) return PB_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipFree(p));
  return PB_OK;
}
extern "C" int pb_memcpy_h2d(pb_ctx *c, void *d, const void *h, uint64_t bytes)
{
  if (!c || (!d && bytes) || (!h && bytes)) return PB_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PB_OK;
}
extern "C" int pb_memcpy_d2h(pb_ctx *c, void *h, const void *d, uint64_t bytes)
{
  if (!c || (!d && bytes) || (!h && bytes)) return PB_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PB_OK;
}

extern "C" int pb_host_alloc(pb_ctx *c, uint64_t bytes, void **host_ptr)
{
  if (!c || !host_ptr) return PB_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipHostMalloc(host_ptr, bytes ? (size_t) bytes : 8, hipHostMallocDefault));
  return PB_OK;
}

extern "C" int pb_host_free(pb_ctx *c, void *host_ptr)
{
  if (!c) return PB_ERR_ARG;
  if (host_ptr) HIPCHK(c, hipHostFree(host_ptr));
  return PB_OK;
}

// ---- chunked uploads: a block of pre-staged inputs goes to HBM on the copy stream while the kernels of the previous block run ----
extern "C" int pb_fence_create(pb_ctx *c, int *fence_out)
{
  if (!c || !fence_out) return PB_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->dev));
  if (c->n_fences >= PB_MAX_FENCES) return fail(c, PB_ERR_STATE, "pb_fence_create: at most %d fences per context", PB_MAX_FENCES);
  HIPCHK(c, hipEventCreateWithFlags(&c->fence[c->n_fences], hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->fence[c->n_fences], c->stream));   // (a fence that was never recorded would not be waitable)
  *fence_out = c->n_fences++;
  return PB_OK;
}
extern "C" int pb_fence_record(pb_ctx *c, int fence)
{
  if (!c) return PB_ERR_ARG;
  if (fence < 0 || fence >= c->n_fences) return fail(c, PB_ERR_ARG, "pb_fence_record: no fence %d", fence);
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipEventRecord(c->fence[fence], c->stream));
  return PB_OK;
}
extern "C" int pb_fence_wait(pb_ctx *c, int fence)
{
  if (!c) return PB_ERR_ARG;
  if (fence < 0 || fence >= c->n_fences) return fail(c, PB_ERR_ARG, "pb_fence_wait: no fence %d", fence);
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipEventSynchronize(c->fence[fence]));
  return PB_OK;
}
extern "C" int pb_upload_async(pb_ctx *c, void *d, const void *h, uint64_t bytes, int after_fence)
{
  if (!c || (!d && bytes) || (!h && bytes)) return PB_ERR_ARG;
  if (after_fence >= c->n_fences) return fail(c, PB_ERR_ARG, "pb_upload_async: no fence %d", after_fence);
  HIPCHK(c, hipSetDevice(c->dev));
  if (after_fence >= 0) HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->fence[after_fence], 0));
  if (bytes) HIPCHK(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->copy_stream));
  return PB_OK;
}
extern "C" int pb_upload_join(pb_ctx *c)
{
  if (!c) return PB_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->dev));
  if (!c->ev_upload) HIPCHK(c, hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->ev_upload, c->copy_stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_upload, 0));
  return PB_OK;
}
extern "C" int pb_upload_sync(pb_ctx *c)
{
  if (!c) return PB_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipStreamSynchronize(c->copy_stream));
  return PB_OK;
}

// staging area for PB_HOST inputs/outputs: a device buffer the host blocks are copied into
static int stage_reserve(pb_ctx *c, size_t bytes)
{
  if (bytes <= c->stage_bytes) return PB_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->stage) HIPCHK(c, hipFree(c->stage));
  c->stage = nullptr;
  c->stage_bytes = 0;
  HIPCHK(c, hipMalloc(&c->stage, bytes));
  c->stage_bytes = bytes;
  return PB_OK;
}

// Resolve up to 4 caller buffers: device pointers pass through; host buffers are packed into the staging area.
struct Part {
  const void *src;
  size_t bytes;
  const void *dev;
};
static int stage_in(pb_ctx *c, int mem, Part *parts, int n)
{
  if (mem == PB_DEVICE) {
    for (int i = 0; i < n; i++) parts[i].dev = parts[i].src;
    return PB_OK;
  }
  if (mem != PB_HOST && mem != PB_HOST_BROADCAST) return fail(c, PB_ERR_ARG, "mem must be PB_HOST, PB_DEVICE or PB_HOST_BROADCAST");
  size_t tot = 0;
  for (int i = 0; i < n; i++) tot += (parts[i].bytes + 255) / 256 * 256;
  int rc = (mem == PB_HOST_BROADCAST) ? stage_reserve(c, tot) : PB_OK;
  if (rc) return rc;
  void *in_buf = nullptr;
  bool copied = false;
  if (mem == PB_HOST) {
    // Double-buffered H2D staging on the copy stream.  Everything enqueued on the main stream so far includes the
    // consumer of the buffer used by the previous call; the buffer used now was consumed two calls ago.
    const int prev = c->in_idx, cur = prev ^ 1;
    HIPCHK(c, hipEventRecord(c->ev_consumed[prev], c->stream));
    c->in_idx = cur;
    if (tot > c->in_stage_bytes[cur]) {
      HIPCHK(c, hipEventSynchronize(c->ev_consumed[cur]));
      if (c->in_stage[cur]) HIPCHK(c, hipFree(c->in_stage[cur]));
      c->in_stage[cur] = nullptr;
      c->in_stage_bytes[cur] = 0;
      HIPCHK(c, hipMalloc(&c->in_stage[cur], tot));
      c->in_stage_bytes[cur] = tot;
    }
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_consumed[cur], 0));
    in_buf = c->in_stage[cur];
  }
  size_t off = 0;
  for (int i = 0; i < n; i++) {
    if (parts[i].src && mem == PB_HOST_BROADCAST) {
      // One value per ROW, the same for every filter (one robot's message feeding a whole parameter sweep): the rows are
      // expanded on the device, nothing of batch size crosses PCIe.  A mask (bytes == B) cannot be broadcast.
      const size_t rows = parts[i].bytes / (sizeof(double) * (size_t) c->B);
      if (rows * sizeof(double) * (size_t) c->B != parts[i].bytes || rows == 0 || rows > (size_t) RowVals::MAX)
        return fail(c, PB_ERR_ARG, "PB_HOST_BROADCAST: only blocks of 1..%d double rows can be broadcast (pass mask = NULL)",
                    RowVals::MAX);
      RowVals v;
      memcpy(v.v, parts[i].src, rows * sizeof(double));
      k_fill_rows<<<nblk(c->B), 64, 0, c->stream>>>((double *) ((char *) c->stage + off), (int) rows, c->B, v);
      HIPCHK(c, hipGetLastError());
      parts[i].dev = (char *) c->stage + off;
    } else if (parts[i].src) {
      HIPCHK(c, hipMemcpyAsync((char *) in_buf + off, parts[i].src, parts[i].bytes, hipMemcpyHostToDevice, c->copy_stream));
      parts[i].dev = (char *) in_buf + off;
      copied = true;
    } else {
      parts[i].dev = nullptr;
    }
    off += (parts[i].bytes + 255) / 256 * 256;
  }
  if (copied) {
    // The caller may reuse its buffers as soon as this returns, so wait for the copy -- NOT for the main stream: the
    // kernels of the previous message keep running underneath.
    HIPCHK(c, hipEventRecord(c->ev_copied, c->copy_stream));
    HIPCHK(c, hipEventSynchronize(c->ev_copied));
  }
  return PB_OK;
}

#define ENTER(c)                                       \
  if (!(c)) return PB_ERR_ARG;                         \
  HIPCHK((c), hipSetDevice((c)->dev))

#define NEED_STATE(c) \
  if (!(c)->have_state) return fail((c), PB_ERR_STATE, "%s before pb_reset", __func__)


extern "C" int pb_reset(pb_ctx *c, const double *vec, const double *quat, const double *cov, int broadcast, int mem)
{
  ENTER(c);
  if (!vec || !quat || !cov) return fail(c, PB_ERR_ARG, "pb_reset: NULL input");
  c->st = c->st_base;  // a reset always lands in the context's own array (a checkpoint the head lived in stays intact)
  c->out_slot = -1;
  const int n = c->ns, B = c->B;
  if (broadcast) {
    if (mem != PB_HOST) return fail(c, PB_ERR_ARG, "pb_reset: broadcast inputs must be host memory");
    double comp[Lay<21>::NC];
    const int off_q = n, off_ll = n + 4, off_p = n + 5;
    for (int i = 0; i < n; i++) comp[i] = vec[i];
    for (int i = 0; i < 4; i++) comp[off_q + i] = quat[i];
    comp[off_ll] = 0.0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j <= i; j++) comp[off_p + pk(i, j)] = cov[j * n + i];
    HIPCHK(c, hipMemcpyAsync(c->d_small, comp, sizeof(double) * c->nc, hipMemcpyHostToDevice, c->stream));
    if (n == 15) k_reset_bcast<15><<<nblk(B), 64, 0, c->stream>>>(c->st, B, c->d_small);
    else k_reset_bcast<21><<<nblk(B), 64, 0, c->stream>>>(c->st, B, c->d_small);
    LAUNCHCHK(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));  // comp is a stack buffer
  } else {
    Part p[3] = { { vec, sizeof(double) * n * B, 0 }, { quat, sizeof(double) * 4 * B, 0 },
                  { cov, sizeof(double) * n * n * B, 0 } };
    int rc = stage_in(c, mem, p, 3);
    if (rc) return rc;
    if (n == 15)
      k_reset<15><<<nblk(B), 64, 0, c->stream>>>(c->st, B, (const double *) p[0].dev, (const double *) p[1].dev, (const double *) p[2].dev);
    else
      k_reset<21><<<nblk(B), 64, 0, c->stream>>>(c->st, B, (const double *) p[0].dev, (const double *) p[1].dev, (const double *) p[2].dev);
    LAUNCHCHK(c);
  }
  c->have_state = true;
  return PB_OK;
}

// The posterior of an update that was computed OUTSIDE the library (the shim's RBISHostUpdate: user code written against the
// reference's updateFilter(prior_state, prior_cov, prior_loglikelihood) contract, rbis_update_interface.hpp:14-35) becomes the head.
extern "C" int pb_set_head(pb_ctx *c, const double *vec, const double *quat, const double *cov, const double *loglik, int mem)
{
  ENTER(c);
  NEED_STATE(c);
  if (!vec || !quat || !cov) return fail(c, PB_ERR_ARG, "pb_set_head: NULL input");
  if (mem != PB_HOST && mem != PB_DEVICE) return fail(c, PB_ERR_ARG, "pb_set_head: mem must be PB_HOST or PB_DEVICE");
  const int n = c->ns, B = c->B;
  Part p[4] = { { vec, sizeof(double) * n * B, 0 }, { quat, sizeof(double) * 4 * B, 0 }, { cov, sizeof(double) * n * n * B, 0 },
                { loglik, loglik ? sizeof(double) * B : 0, 0 } };
  int rc = stage_in(c, mem, p, 4);
  if (rc) return rc;
  double *target = update_target(c);   // like every update: in place, or into the checkpoint slot named by pb_set_output_slot
  if (n == 15)
    k_reset<15><<<nblk(B), 64, 0, c->stream>>>(target, B, (const double *) p[0].dev, (const double *) p[1].dev, (const double *) p[2].dev,
                                               (const double *) p[3].dev);
  else
    k_reset<21><<<nblk(B), 64, 0, c->stream>>>(target, B, (const double *) p[0].dev, (const double *) p[1].dev, (const double *) p[2].dev,
                                               (const double *) p[3].dev);
  LAUNCHCHK(c);
  update_done(c, target);
  return PB_OK;
}

// ---- filters without an IMU message in a batched message (independent log segments) ----
extern "C" int pb_set_imu_valid(pb_ctx *c, const uint8_t *valid_dev)
{
  if (!c) return PB_ERR_ARG;
  c->imu_valid_next = valid_dev;
  return PB_OK;
}
// The mask belongs to the NEXT call that takes an IMU step, whatever becomes of it: such an entry point moves it from imu_valid_next to
// imu_valid_cur first thing (ImuIdleTake) and forgets it when it returns; the launchers of the step kernels (pb_step.hip) pass
// their IMU block through pbk_idle_prepare right in front of their ONE launch (rbis_frontend.hpp, k_imu_idle_prepare).
struct ImuIdleTake {
  pb_ctx *c;
  explicit ImuIdleTake(pb_ctx *ctx) : c(ctx)
  {
    if (c) {
      c->imu_valid_cur = c->imu_valid_next;
      c->imu_valid_next = nullptr;
    }
  }
  ~ImuIdleTake()
  {
    if (c) c->imu_valid_cur = nullptr;
  }
};
const double *pbk_idle_prepare(pb_ctx *c, const double *imu_dev, int *rc_out)
{
  *rc_out = PB_OK;
  const uint8_t *valid = c->imu_valid_cur;
  if (!valid || !imu_dev) return imu_dev;
  c->imu_valid_cur = nullptr;   // (one step launch per call)
  if (!c->imu_keep) {
    hipError_t e = hipMalloc((void **) &c->imu_keep, sizeof(double) * 7 * (size_t) c->stride);
    if (e != hipSuccess) {
      *rc_out = fail(c, PB_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
      return imu_dev;
    }
  }
  if (c->ns == 15) k_imu_idle_prepare<15><<<(c->B + 255) / 256, 256, 0, c->stream>>>(c->st, valid, imu_dev, c->imu_keep, c->B);
  else k_imu_idle_prepare<21><<<(c->B + 255) / 256, 256, 0, c->stream>>>(c->st, valid, imu_dev, c->imu_keep, c->B);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) *rc_out = fail(c, PB_ERR_HIP, "k_imu_idle_prepare: %s", hipGetErrorString(e));
  return c->imu_keep;
}

extern "C" int pb_predict(pb_ctx *c, const double *imu_block, const double q[4], int mem)
{
  ImuIdleTake idle(c);
  ENTER(c);
  NEED_STATE(c);
  if (!imu_block || !q) return fail(c, PB_ERR_ARG, "pb_predict: NULL input");
  if (mem == PB_HOST_BROADCAST) {  // one message for every filter: the 7 values are kernel arguments
    StepBcast bc;
    memcpy(bc.imu, imu_block, sizeof(bc.imu));
    bc.on = 1;
    return pbk_step(c, false, nullptr, nullptr, nullptr, q, &bc);
  }
  Part p[1] = { { imu_block, sizeof(double) * 7 * c->B, 0 } };
  int rc = stage_in(c, mem, p, 1);
  if (rc) return rc;
  return pbk_step(c, false, (const double *) p[0].dev, nullptr, nullptr, q);
}

extern "C" int pb_step_legodo(pb_ctx *c, const double *imu_block, const double *lo_block, const uint8_t *mask,
                              const double q[4], int mem)
{
  ImuIdleTake idle(c);
  ENTER(c);
  NEED_STATE(c);
  if (!imu_block || !lo_block || !q) return fail(c, PB_ERR_ARG, "pb_step_legodo: NULL input");
  if (mem == PB_HOST_BROADCAST) {
    // one message for every filter (a parameter sweep replaying one robot's log): the 13 values are kernel arguments --
    // no device block, no fill launch, no input traffic.  A mask cannot be broadcast.
    if (mask) return fail(c, PB_ERR_ARG, "PB_HOST_BROADCAST: a mask cannot be broadcast (pass mask = NULL)");
    StepBcast bc;
    memcpy(bc.imu, imu_block, sizeof(bc.imu));
    memcpy(bc.lo, lo_block, sizeof(bc.lo));
    bc.on = 3;
    return pbk_step(c, true, nullptr, nullptr, nullptr, q, &bc);
  }
  Part p[3] = { { imu_block, sizeof(double) * 7 * c->B, 0 }, { lo_block, sizeof(double) * 6 * c->B, 0 },
                { mask, (size_t) c->B, 0 } };
  int rc = stage_in(c, mem, p, 3);
  if (rc) return rc;
  return pbk_step(c, true, (const double *) p[0].dev, (const double *) p[1].dev, (const uint8_t *) p[2].dev, q);
}

extern "C" int pb_step_legodo_split(pb_ctx *c, const double *imu_block, int imu_mem, const double *lo_block,
                                    const uint8_t *mask, int lo_mem, const double q[4])
{
  ImuIdleTake idle(c);
  if (imu_mem == lo_mem) return pb_step_legodo(c, imu_block, lo_block, mask, q, imu_mem);
  ENTER(c);
  NEED_STATE(c);
  if (!imu_block || !lo_block || !q) return fail(c, PB_ERR_ARG, "pb_step_legodo_split: NULL input");
  if (imu_mem == PB_HOST && lo_mem == PB_HOST) return fail(c, PB_ERR_ARG, "pb_step_legodo_split: unreachable");
  StepBcast bc;
  const double *d_imu = nullptr, *d_lo = nullptr;
  const uint8_t *d_mask = nullptr;
  // at most one of the two groups goes through the double-buffered host staging (PB_HOST); a broadcast group travels as
  // kernel arguments, a device group is read in place
  if (imu_mem == PB_HOST_BROADCAST) {
    memcpy(bc.imu, imu_block, sizeof(bc.imu));
    bc.on |= 1;
  } else {
    Part p[1] = { { imu_block, sizeof(double) * 7 * c->B, 0 } };
    int rc = stage_in(c, imu_mem, p, 1);
    if (rc) return rc;
    d_imu = (const double *) p[0].dev;
  }
  if (lo_mem == PB_HOST_BROADCAST) {
    if (mask) return fail(c, PB_ERR_ARG, "PB_HOST_BROADCAST: a mask cannot be broadcast (pass mask = NULL)");
    memcpy(bc.lo, lo_block, sizeof(bc.lo));
    bc.on |= 2;
  } else {
    Part p[2] = { { lo_block, sizeof(double) * 6 * c->B, 0 }, { mask, (size_t) c->B, 0 } };
    int rc = stage_in(c, lo_mem, p, 2);
    if (rc) return rc;
    d_lo = (const double *) p[0].dev;
    d_mask = (const uint8_t *) p[1].dev;
  }
  return pbk_step(c, true, d_imu, d_lo, d_mask, q, &bc);
}

extern "C" int pb_step_legodo_correct(pb_ctx *c, const double *imu_block, const double *lo_block, const uint8_t *mask,
                                      const double q[4], int mem, int corr_kind, const double *z2, const double *R2,
                                      int r_kind2, const double *quat_meas2, const uint8_t *mask2, int mem2)
{
  ImuIdleTake idle(c);
  ENTER(c);
  NEED_STATE(c);
  if (!imu_block || !lo_block || !q || !z2 || !R2 || !quat_meas2) return fail(c, PB_ERR_ARG, "pb_step_legodo_correct: NULL input");
  if (corr_kind != PB_CORR_POS_ORIENT && corr_kind != PB_CORR_POS_YAW) return fail(c, PB_ERR_ARG, "pb_step_legodo_correct: bad corr_kind %d", corr_kind);
  if (mem2 == PB_HOST_BROADCAST && r_kind2 == PB_R_DIAG) r_kind2 = PB_R_DIAG_BROADCAST;
  if (r_kind2 != PB_R_DIAG && r_kind2 != PB_R_DIAG_BROADCAST) return fail(c, PB_ERR_ARG, "pb_step_legodo_correct: R2 must be diagonal (PB_R_DIAG or PB_R_DIAG_BROADCAST)");
  const int m2 = (corr_kind == PB_CORR_POS_ORIENT) ? 6 : 4;
  const size_t B = (size_t) c->B;
  const bool rbc = r_kind2 == PB_R_DIAG_BROADCAST;
  // (the two-wave 21-state mapping, PRONTO_BATCH_QUAD21=0, has no m = 6 kernel that takes z / quat_meas as arguments:
  // its correction falls back to the generic update, which needs the replicated blocks staged below)
  const bool arg_kernel = c->ns == 15 || c->quad21 || m2 <= 4;
  if (mem == PB_HOST_BROADCAST && mem2 == PB_HOST_BROADCAST && rbc && !mask && !mask2 && arg_kernel) {
    // one robot's three messages for every filter: everything travels as kernel arguments
    StepBcast bc;
    memcpy(bc.imu, imu_block, sizeof(bc.imu));
    memcpy(bc.lo, lo_block, sizeof(bc.lo));
    bc.on = 3;
    return pbk_step_correct(c, corr_kind, nullptr, nullptr, nullptr, q, nullptr, nullptr, R2, nullptr, nullptr, &bc, z2, quat_meas2);
  }
  Part p[7] = { { imu_block, sizeof(double) * 7 * B, 0 }, { lo_block, sizeof(double) * 6 * B, 0 }, { mask, B, 0 },
                { z2, sizeof(double) * m2 * B, 0 }, { rbc ? nullptr : R2, rbc ? 0 : sizeof(double) * m2 * B, 0 },
                { quat_meas2, sizeof(double) * 4 * B, 0 }, { mask2, B, 0 } };
  // one staging call when both groups live in the same space (the double-buffered host staging must not be flipped twice
  // before its consumer is enqueued); otherwise at most one of the two groups is PB_HOST
  int rc;
  if (mem == mem2) rc = stage_in(c, mem, p, 7);
  else {
    rc = stage_in(c, mem, p, 3);
    if (!rc) rc = stage_in(c, mem2, p + 3, 4);
  }
  if (rc) return rc;
  return pbk_step_correct(c, corr_kind, (const double *) p[0].dev, (const double *) p[1].dev, (const uint8_t *) p[2].dev, q,
                          (const double *) p[3].dev, (const double *) p[4].dev, rbc ? R2 : nullptr, (const double *) p[5].dev,
                          (const uint8_t *) p[6].dev);
}

extern "C" int pb_run_legodo(pb_ctx *c, int n_steps, const double *imu_stream, const double *lo_stream,
                             const uint8_t *mask_stream, const double q[4], float *elapsed_ms)
{
  ENTER(c);
  NEED_STATE(c);
  if (n_steps < 0 || !imu_stream || !lo_stream || !q) return fail(c, PB_ERR_ARG, "pb_run_legodo: bad argument");
  const size_t B = (size_t) c->B;
  // Experiment switch: PRONTO_BATCH_GRAPH=1 captures the n_steps launches into a HIP graph and times
  // ONE replay of it (capture and instantiation excluded).  Needs a capturable stream (pb_use_own_stream).
  static const bool want_graph = getenv("PRONTO_BATCH_GRAPH") && getenv("PRONTO_BATCH_GRAPH")[0] == '1';
  if (want_graph && c->stream != nullptr && n_steps > 1) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < n_steps; s++) {
      int rc = pbk_step(c, true, imu_stream + (size_t) s * 7 * B, lo_stream + (size_t) s * 6 * B,
                        mask_stream ? mask_stream + (size_t) s * B : nullptr, q);
      if (rc) return rc;
    }
    HIPCHK(c, hipStreamEndCapture(c->stream, &graph));
    HIPCHK(c, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    HIPCHK(c, hipGraphLaunch(exec, c->stream));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    if (elapsed_ms) HIPCHK(c, hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
    (void) hipGraphExecDestroy(exec);
    (void) hipGraphDestroy(graph);
    return PB_OK;
  }
  if (elapsed_ms) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  if (c->run_block > 0 && n_steps > 1 && c->st == c->st_base && c->out_slot < 0) {
    // cache-blocked order (pb_create): every block of filters runs the whole stream before the next block starts
    for (long b0 = 0; b0 < (long) B; b0 += c->run_block) {
      const int nb = (int) std::min<long>(c->run_block, (long) B - b0);
      for (int s = 0; s < n_steps; s++) {
        int rc = pbk_step_range(c, imu_stream + (size_t) s * 7 * B, lo_stream + (size_t) s * 6 * B,
                                mask_stream ? mask_stream + (size_t) s * B : nullptr, q, b0, nb, c->run_block_coop15, c->run_block_hint);
        if (rc) return rc;
      }
    }
  } else {
    for (int s = 0; s < n_steps; s++) {
      int rc = pbk_step(c, true, imu_stream + (size_t) s * 7 * B, lo_stream + (size_t) s * 6 * B,
                        mask_stream ? mask_stream + (size_t) s * B : nullptr, q);
      if (rc) return rc;
    }
  }
  if (elapsed_ms) {
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
  }
  return PB_OK;
}

extern "C" int pb_replay_legodo_fused(pb_ctx *c, int n_steps, int steps_per_launch, const double *imu_stream,
                                      const double *lo_stream, const uint8_t *mask_stream, const double q[4],
                                      float *elapsed_ms)
{
  ENTER(c);
  NEED_STATE(c);
  if (n_steps < 0 || steps_per_launch < 1 || !imu_stream || !lo_stream || !q)
    return fail(c, PB_ERR_ARG, "pb_replay_legodo_fused: bad argument");
  const size_t B = (size_t) c->B;
  {
    int rc = detach_head(c, true);  // this kernel works in place: never on a checkpoint slot
    if (rc) return rc;
  }
  if (elapsed_ms) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  for (int s = 0; s < n_steps; s += steps_per_launch) {
    const int T = (n_steps - s < steps_per_launch) ? n_steps - s : steps_per_launch;
    int rc = pbk_replay_fused(c, T, imu_stream + (size_t) s * 7 * B, lo_stream + (size_t) s * 6 * B,
                              mask_stream ? mask_stream + (size_t) s * B : nullptr, q);
    if (rc) return rc;
  }
  if (elapsed_ms) {
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
  }
  return PB_OK;
}

extern "C" int pb_replay_legodo_checkpointed(pb_ctx *c, int n_steps, int steps_per_launch, const double *imu_stream,
                                             const double *lo_stream, const uint8_t *mask_stream, const double q[4], int first_slot,
                                             float *elapsed_ms)
{
  ENTER(c);
  NEED_STATE(c);
  if (n_steps < 0 || steps_per_launch < 1 || !imu_stream || !lo_stream || !q)
    return fail(c, PB_ERR_ARG, "pb_replay_legodo_checkpointed: bad argument");
  if (first_slot < 0 || first_slot + n_steps > c->nhist)
    return fail(c, PB_ERR_ARG, "pb_replay_legodo_checkpointed: slots %d..%d of %d (pb_history_reserve)", first_slot, first_slot + n_steps - 1, c->nhist);
  const size_t B = (size_t) c->B;
  {
    int rc = detach_head(c, true);  // the state rides in registers from the context's own array; the slots only receive copies
    if (rc) return rc;
  }
  if (elapsed_ms) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  for (int s = 0; s < n_steps; s += steps_per_launch) {
    const int T = (n_steps - s < steps_per_launch) ? n_steps - s : steps_per_launch;
    int rc = pbk_replay_fused(c, T, imu_stream + (size_t) s * 7 * B, lo_stream + (size_t) s * 6 * B,
                              mask_stream ? mask_stream + (size_t) s * B : nullptr, q, first_slot + s);
    if (rc) return rc;
  }
  if (elapsed_ms) {
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
  }
  return PB_OK;
}

static int update_common(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int rkind,
                         const double *qm, bool orient, const uint8_t *mask, int mem)
{
  ENTER(c);
  NEED_STATE(c);
  if (m < 1 || m > 6) return fail(c, PB_ERR_ARG, "update: m must be 1..6 (got %d)", m);
  if (!idx || !z || !R) return fail(c, PB_ERR_ARG, "update: NULL input");
  if (orient && !qm) return fail(c, PB_ERR_ARG, "update: quat_meas is NULL");
  for (int i = 0; i < m; i++) {
    if (idx[i] < 0 || idx[i] >= c->ns) return fail(c, PB_ERR_ARG, "update: index %d out of range for n_states=%d", idx[i], c->ns);
    for (int j = 0; j < i; j++)
      if (idx[i] == idx[j]) return fail(c, PB_ERR_ARG, "update: duplicate index %d", idx[i]);
  }
  const size_t B = (size_t) c->B;
  size_t rbytes;
  const double *rb = nullptr;
  if (mem == PB_HOST_BROADCAST && rkind == PB_R_DIAG) rkind = PB_R_DIAG_BROADCAST;  // the same thing, without a fill
  if (rkind == PB_R_DIAG_BROADCAST) { rb = R; rbytes = 0; }
  else if (rkind == PB_R_DIAG) rbytes = sizeof(double) * m * B;
  else if (rkind == PB_R_FULL) rbytes = sizeof(double) * m * m * B;
  else return fail(c, PB_ERR_ARG, "update: bad r_kind %d", rkind);
  int rc;
  if (mem == PB_HOST_BROADCAST && rb && !mask && !c->generic_update) {
    // one measurement for every filter on a compile-time-index kernel: z, R and the quaternion are kernel arguments
    rc = pbk_update_ct(c, m, idx, nullptr, nullptr, rb, nullptr, nullptr, z, orient ? qm : nullptr);
    if (rc >= 0) return rc;
  }
  Part p[4] = { { z, sizeof(double) * m * B, 0 }, { rb ? nullptr : R, rbytes, 0 },
                { orient ? qm : nullptr, sizeof(double) * 4 * B, 0 }, { mask, B, 0 } };
  rc = stage_in(c, mem, p, 4);
  if (rc) return rc;
  // the handlers' own index lists run on the compile-time-index kernels (two-role for 15 states, four-wave for 21; no column gather);
  // PRONTO_BATCH_GENERIC_UPDATE=1 forces the run-time-index kernel for A/B runs and tests
  if (!c->generic_update && (rkind == PB_R_DIAG || rkind == PB_R_DIAG_BROADCAST)) {
    rc = pbk_update_ct(c, m, idx, (const double *) p[0].dev, rb ? nullptr : (const double *) p[1].dev, rb,
                       (const double *) p[2].dev, (const uint8_t *) p[3].dev);
    if (rc >= 0) return rc;
  }
  if (!c->generic_update && rkind == PB_R_FULL) {  // a full per-filter R on the handlers' index lists: same kernels
    rc = pbk_update_ct(c, m, idx, (const double *) p[0].dev, nullptr, nullptr, (const double *) p[2].dev, (const uint8_t *) p[3].dev,
                       nullptr, nullptr, (const double *) p[1].dev);
    if (rc >= 0) return rc;
  }
  if (c->ns == 15)
    return pbk_update15(c, m, idx, (const double *) p[0].dev, (const double *) p[1].dev, rkind, rb,
                        (const double *) p[2].dev, (const uint8_t *) p[3].dev);
  return pbk_update21(c, m, idx, (const double *) p[0].dev, (const double *) p[1].dev, rkind, rb,
                      (const double *) p[2].dev, (const uint8_t *) p[3].dev);
}

extern "C" int pb_update_indexed(pb_ctx *c, int m, const int *idx, const double *z, const double *R, int r_kind,
                                 const uint8_t *mask, int mem)
{
  return update_common(c, m, idx, z, R, r_kind, nullptr, false, mask, mem);
}

extern "C" int pb_update_indexed_orient(pb_ctx *c, int m, const int *idx, const double *z, const double *R,
                                        int r_kind, const double *quat_meas, const uint8_t *mask, int mem)
{
  return update_common(c, m, idx, z, R, r_kind, quat_meas, true, mask, mem);
}

extern "C" int pb_snapshot(pb_ctx *c, int slot)
{
  ENTER(c);
  NEED_STATE(c);
  if (slot < 0 || slot >= c->nsnap) return fail(c, PB_ERR_STATE, "pb_snapshot: slot %d of %d", slot, c->nsnap);
  double *snap = c->snaps + (size_t) slot * 7 * c->stride;
  if (c->ns == 15) k_snapshot<15><<<nblk(c->B), 64, 0, c->stream>>>(c->st, c->stride, c->B, snap);
  else k_snapshot<21><<<nblk(c->B), 64, 0, c->stream>>>(c->st, c->stride, c->B, snap);
  LAUNCHCHK(c);
  return PB_OK;
}

extern "C" int pb_snapshot_from_slot(pb_ctx *c, int slot, int checkpoint_slot)
{
  ENTER(c);
  if (slot < 0 || slot >= c->nsnap) return fail(c, PB_ERR_STATE, "pb_snapshot_from_slot: slot %d of %d", slot, c->nsnap);
  if (checkpoint_slot < 0 || checkpoint_slot >= c->nhist)
    return fail(c, PB_ERR_STATE, "pb_snapshot_from_slot: checkpoint slot %d of %d", checkpoint_slot, c->nhist);
  double *snap = c->snaps + (size_t) slot * 7 * c->stride;
  const double *src = c->hist + (size_t) checkpoint_slot * c->state_doubles;
  if (c->ns == 15) k_snapshot<15><<<nblk(c->B), 64, 0, c->stream>>>(src, c->stride, c->B, snap);
  else k_snapshot<21><<<nblk(c->B), 64, 0, c->stream>>>(src, c->stride, c->B, snap);
  LAUNCHCHK(c);
  return PB_OK;
}

extern "C" int pb_compose_delta(pb_ctx *c, int slot, const double *t, const double *q, double *z_out,
                                double *quat_out, int mem)
{
  ENTER(c);
  if (slot < 0 || slot >= c->nsnap) return fail(c, PB_ERR_STATE, "pb_compose_delta: slot %d of %d", slot, c->nsnap);
  if (!t || !q || !z_out || !quat_out) return fail(c, PB_ERR_ARG, "pb_compose_delta: NULL argument");
  Part p[2] = { { t, sizeof(double) * 3 * c->B, 0 }, { q, sizeof(double) * 4 * c->B, 0 } };
  int rc = stage_in(c, mem, p, 2);
  if (rc) return rc;
  const double *snap = c->snaps + (size_t) slot * 7 * c->stride;
  k_compose<<<nblk(c->B), 64, 0, c->stream>>>(snap, c->stride, c->B, (const double *) p[0].dev, (const double *) p[1].dev, z_out, quat_out);
  LAUNCHCHK(c);
  return PB_OK;
}

static int get_state_impl(pb_ctx *c, const double *st, int first, int count, double *vec_out, double *quat_out, double *cov_out, double *ll_out, int mem);

extern "C" int pb_get_head(pb_ctx *c, int first, int count, double *vec_out, double *quat_out, double *cov_out,
                           double *ll_out, int mem)
{
  ENTER(c);
  NEED_STATE(c);
  return get_state_impl(c, c->st, first, count, vec_out, quat_out, cov_out, ll_out, mem);
}

// the same read of a posterior that lives in a checkpoint slot (the head is not touched)
extern "C" int pb_get_slot(pb_ctx *c, int slot, int first, int count, double *vec_out, double *quat_out, double *cov_out, double *ll_out, int mem)
{
  ENTER(c);
  if (slot < 0 || slot >= c->nhist) return fail(c, PB_ERR_STATE, "pb_get_slot: checkpoint slot %d of %d", slot, c->nhist);
  return get_state_impl(c, c->hist + (size_t) slot * c->state_doubles, first, count, vec_out, quat_out, cov_out, ll_out, mem);
}

static int get_state_impl(pb_ctx *c, const double *st, int first, int count, double *vec_out, double *quat_out, double *cov_out, double *ll_out, int mem)
{
  if (first < 0 || count < 0 || (long) first + count > c->B) return fail(c, PB_ERR_ARG, "pb_get_head: range [%d,+%d) outside batch %d", first, count, c->B);
  if (count == 0) return PB_OK;
  const int n = c->ns;
  double *dv = vec_out, *dq = quat_out, *dc = cov_out, *dl = ll_out;
  size_t o_v = 0, o_q = 0, o_c = 0, o_l = 0;
  if (mem == PB_HOST) {
    size_t tot = 0;
    o_v = tot; tot += vec_out ? sizeof(double) * n * count : 0;
    o_q = tot; tot += quat_out ? sizeof(double) * 4 * count : 0;
    o_c = tot; tot += cov_out ? sizeof(double) * n * n * count : 0;
    o_l = tot; tot += ll_out ? sizeof(double) * count : 0;
    int rc = stage_reserve(c, tot ? tot : 8);
    if (rc) return rc;
    char *s = (char *) c->stage;
    dv = vec_out ? (double *) (s + o_v) : nullptr;
    dq = quat_out ? (double *) (s + o_q) : nullptr;
    dc = cov_out ? (double *) (s + o_c) : nullptr;
    dl = ll_out ? (double *) (s + o_l) : nullptr;
  } else if (mem != PB_DEVICE) {
    return fail(c, PB_ERR_ARG, "mem must be PB_HOST or PB_DEVICE");
  }
  if (n == 15) k_get_head<15><<<nblk(count), 64, 0, c->stream>>>(st, first, count, dv, dq, dc, dl);
  else k_get_head<21><<<nblk(count), 64, 0, c->stream>>>(st, first, count, dv, dq, dc, dl);
  LAUNCHCHK(c);
  if (mem == PB_HOST) {
    if (vec_out) HIPCHK(c, hipMemcpyAsync(vec_out, dv, sizeof(double) * n * count, hipMemcpyDeviceToHost, c->stream));
    if (quat_out) HIPCHK(c, hipMemcpyAsync(quat_out, dq, sizeof(double) * 4 * count, hipMemcpyDeviceToHost, c->stream));
    if (cov_out) HIPCHK(c, hipMemcpyAsync(cov_out, dc, sizeof(double) * n * n * count, hipMemcpyDeviceToHost, c->stream));
    if (ll_out) HIPCHK(c, hipMemcpyAsync(ll_out, dl, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return PB_OK;
}

extern "C" int pb_get_filter_state(pb_ctx *c, int filter, double quat[4], double state[21], double cov[441])
{
  if (!c || !quat || !state || !cov) return PB_ERR_ARG;
  const int n = c->ns;
  double v[21], P[441];
  int rc = pb_get_head(c, filter, 1, v, quat, P, nullptr, PB_HOST);
  if (rc) return rc;
  memset(state, 0, sizeof(double) * 21);
  memset(cov, 0, sizeof(double) * 441);
  for (int i = 0; i < n; i++) state[i] = v[i];
  for (int col = 0; col < n; col++)
    for (int r = 0; r < n; r++) cov[col * 21 + r] = P[col * n + r];
  return PB_OK;
}

// bit-level checksum of a whole state array: wrapping sum and xor of every 64-bit word after a position-dependent rotation,
// combined with integer atomics, hence independent of the order in which waves finish
static __global__ __launch_bounds__(256) void k_state_checksum(const uint64_t *__restrict__ w, size_t n, unsigned long long *__restrict__ out)
{
  unsigned long long sum = 0, x = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
    const uint64_t v = w[i];
    const unsigned r = (unsigned) (i % 63) + 1;
    sum += v * (2 * (uint64_t) (i % 1021) + 1);
    x ^= (v << r) | (v >> (64 - r));
  }
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o);
    x ^= __shfl_xor(x, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, sum);
    atomicXor(out + 1, x);
  }
}

extern "C" int pb_state_checksum(pb_ctx *c, int slot, uint64_t out[2])
{
  ENTER(c);
  NEED_STATE(c);
  if (!out) return PB_ERR_ARG;
  if (slot >= c->nhist) return fail(c, PB_ERR_ARG, "pb_state_checksum: slot %d of %d", slot, c->nhist);
  const double *src = slot < 0 ? c->st : c->hist + (size_t) slot * c->state_doubles;
  int rc = stage_reserve(c, 2 * sizeof(uint64_t));
  if (rc) return rc;
  HIPCHK(c, hipMemsetAsync(c->stage, 0, 2 * sizeof(uint64_t), c->stream));
  k_state_checksum<<<2048, 256, 0, c->stream>>>((const uint64_t *) src, c->state_doubles, (unsigned long long *) c->stage);
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(out, c->stage, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PB_OK;
}

extern "C" int pb_summary(pb_ctx *c, double out[4])
{
  ENTER(c);
  NEED_STATE(c);
  if (!out) return PB_ERR_ARG;
  const int nb = nblk(c->B);
  int rc = stage_reserve(c, sizeof(double) * 4 * (size_t) nb);
  if (rc) return rc;
  double *part = (double *) c->stage;
  if (c->ns == 15) k_summary<15><<<nb, 64, 0, c->stream>>>(c->st, c->B, part);
  else k_summary<21><<<nb, 64, 0, c->stream>>>(c->st, c->B, part);
  LAUNCHCHK(c);
  double *h = (double *) malloc(sizeof(double) * 4 * (size_t) nb);
  if (!h) return fail(c, PB_ERR_ARG, "pb_summary: out of host memory");
  hipError_t e = hipMemcpyAsync(h, part, sizeof(double) * 4 * (size_t) nb, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    free(h);
    return fail(c, PB_ERR_HIP, "pb_summary copy failed: %s", hipGetErrorString(e));
  }
  out[0] = out[1] = out[2] = out[3] = 0.0;
  for (int i = 0; i < nb; i++) {
    out[0] += h[4 * i];
    out[1] += h[4 * i + 1];
    if (h[4 * i + 2] > out[2]) out[2] = h[4 * i + 2];
    out[3] += h[4 * i + 3];
  }
  free(h);
  return PB_OK;
}

// number of filters a device-resident update mask [B] lets through (one wave-level popcount + atomic per 256 filters)
static __global__ __launch_bounds__(256) void k_mask_count(const uint8_t *__restrict__ mask, int B, unsigned *__restrict__ out)
{
  const int b = blockIdx.x * 256 + threadIdx.x;
  const bool on = b < B && mask[b] != 0;
  const unsigned n = (unsigned) __popcll(__ballot(on));
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, n);
}

extern "C" int pb_mask_count(pb_ctx *c, const uint8_t *mask_dev, int *count_out)
{
  ENTER(c);
  if (!mask_dev || !count_out) return fail(c, PB_ERR_ARG, "pb_mask_count: NULL argument");
  int rc = stage_reserve(c, sizeof(unsigned));
  if (rc) return rc;
  HIPCHK(c, hipMemsetAsync(c->stage, 0, sizeof(unsigned), c->stream));
  k_mask_count<<<(c->B + 255) / 256, 256, 0, c->stream>>>(mask_dev, c->B, (unsigned *) c->stage);
  LAUNCHCHK(c);
  unsigned n = 0;
  HIPCHK(c, hipMemcpyAsync(&n, c->stage, sizeof n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *count_out = (int) n;
  return PB_OK;
}

extern "C" int pb_set_process_noise_block(pb_ctx *c, const double *q_block_dev)
{
  if (!c) return PB_ERR_ARG;
  c->k.qblk = q_block_dev;  // [4][B] device memory, or NULL to go back to the scalar q of each call
  return PB_OK;
}

template <int NS>
static int launch_nll(pb_ctx *c, int m, const int *idx, const double *tv, const double *tq, double *out, double *err)
{
#define NLL_CASE(M)                                                                                         \
  case M: {                                                                                                 \
    IdxArg<M> ia;                                                                                           \
    for (int i = 0; i < M; i++) ia.v[i] = idx[i];                                                           \
    k_window_nll<NS, M><<<nblk(c->B), 64, 0, c->stream>>>(c->st, c->B, ia, tv, tq, out, err);    \
  } break;
  switch (m) {
    NLL_CASE(1) NLL_CASE(2) NLL_CASE(3) NLL_CASE(4) NLL_CASE(5) NLL_CASE(6) NLL_CASE(7) NLL_CASE(8) NLL_CASE(9)
    default: return fail(c, PB_ERR_ARG, "pb_window_nll: m must be 1..9");
  }
#undef NLL_CASE
  LAUNCHCHK(c);
  return PB_OK;
}

extern "C" int pb_window_nll(pb_ctx *c, int m, const int *idx, const double *truth_vec, const double *truth_quat,
                             double *out3, double *err_out, int mem)
{
  ENTER(c);
  NEED_STATE(c);
  if (m < 1 || m > 9 || !idx || !truth_vec || !truth_quat || !out3) return fail(c, PB_ERR_ARG, "pb_window_nll: bad argument");
  for (int i = 0; i < m; i++) {
    if (idx[i] < 0 || idx[i] >= c->ns) return fail(c, PB_ERR_ARG, "pb_window_nll: index %d out of range", idx[i]);
    for (int j = 0; j < i; j++)
      if (idx[i] == idx[j]) return fail(c, PB_ERR_ARG, "pb_window_nll: duplicate index %d", idx[i]);
  }
  const size_t B = (size_t) c->B, n = (size_t) c->ns;
  const double *tv = truth_vec, *tq = truth_quat;
  double *d_out = out3, *d_err = err_out;
  if (mem == PB_HOST) {
    const size_t o1 = sizeof(double) * n * B, o2 = o1 + sizeof(double) * 4 * B, o3 = o2 + sizeof(double) * 3 * B;
    int rc = stage_reserve(c, o3 + sizeof(double) * n * B);
    if (rc) return rc;
    char *s = (char *) c->stage;
    HIPCHK(c, hipMemcpyAsync(s, truth_vec, o1, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s + o1, truth_quat, sizeof(double) * 4 * B, hipMemcpyHostToDevice, c->stream));
    tv = (const double *) s;
    tq = (const double *) (s + o1);
    d_out = (double *) (s + o2);
    d_err = err_out ? (double *) (s + o3) : nullptr;
  } else if (mem != PB_DEVICE) {
    return fail(c, PB_ERR_ARG, "mem must be PB_HOST or PB_DEVICE");
  }
  int rc = (c->ns == 15) ? launch_nll<15>(c, m, idx, tv, tq, d_out, d_err) : launch_nll<21>(c, m, idx, tv, tq, d_out, d_err);
  if (rc) return rc;
  if (mem == PB_HOST) {
    HIPCHK(c, hipMemcpyAsync(out3, d_out, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, c->stream));
    if (err_out) HIPCHK(c, hipMemcpyAsync(err_out, d_err, sizeof(double) * n * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return PB_OK;
}

extern "C" int pb_legodo_init(pb_ctx *c, double lt, double ht, int64_t low_delay, int64_t high_delay, int filter_contact_events)
{
  ENTER(c);
  if (!(ht >= lt) || low_delay < 0 || high_delay < 0) return fail(c, PB_ERR_ARG, "pb_legodo_init: need high >= low threshold, delays >= 0");
  if (low_delay > 2000000000LL || high_delay > 2000000000LL) return fail(c, PB_ERR_ARG, "pb_legodo_init: delays must be below 2e9 us");
  if (!c->legd) HIPCHK(c, hipMalloc((void **) &c->legd, sizeof(double) * (NLD + NLD_WC) * c->stride));
  if (!c->legi) HIPCHK(c, hipMalloc((void **) &c->legi, sizeof(int64_t) * NLI * c->stride));
  // the thresholds pass through `float` variables in the reference (leg_estimate.cpp:103-104, FootContactAlt.cpp:5)
  // a (re-)initialised context starts like leg_estimate's constructor: FootContactAlt, no controller input, no world
  // constraint, controller contact counts -1 (leg_estimate.cpp:93-142, rbis_legodo_update.cpp:100-101)
  c->leg_par = LegPar{};
  c->leg_meas = LegMeasPar{};
  c->leg_nc_h[0] = c->leg_nc_h[1] = -1;
  c->leg_nc_dev = false;
  c->leg_par.alt = SchmittPar{ (double) (float) lt, (double) (float) ht, low_delay, high_delay };
  c->leg_par.filter_contact_events = filter_contact_events ? 1 : 0;
  k_legodo_reset<<<nblk(c->B), 64, 0, c->stream>>>(c->legd, c->legi, c->stride, c->B, -1);
  LAUNCHCHK(c);
  return PB_OK;
}

extern "C" int pb_legodo_set_contact_mode(pb_ctx *c, int standing, double total_force, double standing_schmitt_level,
                                          int use_controller_input)
{
  ENTER(c);
  if (!c->legd) return fail(c, PB_ERR_STATE, "pb_legodo_set_contact_mode before pb_legodo_init");
  c->leg_par.standing = standing ? 1 : 0;
  c->leg_par.total_force = (float) total_force;                        // float members (FootContact.h:24-28)
  c->leg_par.standing_schmitt_level = (float) standing_schmitt_level;
  c->leg_par.use_controller_input = use_controller_input ? 1 : 0;
  return PB_OK;
}

extern "C" int pb_legodo_set_message_times(pb_ctx *c, const int64_t *utimes, const uint8_t *valid, int mem)
{
  ENTER(c);
  if (!c->legd) return fail(c, PB_ERR_STATE, "pb_legodo_set_message_times before pb_legodo_init");
  if (mem != PB_HOST && mem != PB_DEVICE) return fail(c, PB_ERR_ARG, "pb_legodo_set_message_times: mem must be PB_HOST or PB_DEVICE");
  c->leg_ut_on = c->leg_valid_on = false;
  c->leg_ut_ext = nullptr;
  c->leg_valid_ext = nullptr;
  if (mem == PB_DEVICE) {   // read in place by the consuming launch (no copy): the arrays stay the caller's until that launch has run
    c->leg_ut_ext = utimes;
    c->leg_valid_ext = valid;
    c->leg_ut_on = utimes != nullptr;
    c->leg_valid_on = valid != nullptr;
    return PB_OK;
  }
  if (utimes) {
    if (!c->leg_ut) HIPCHK(c, hipMalloc((void **) &c->leg_ut, sizeof(int64_t) * (size_t) c->stride));
    HIPCHK(c, hipMemcpyAsync(c->leg_ut, utimes, sizeof(int64_t) * (size_t) c->B, hipMemcpyHostToDevice, c->stream));
    c->leg_ut_on = true;
  }
  if (valid) {
    if (!c->leg_valid) HIPCHK(c, hipMalloc((void **) &c->leg_valid, (size_t) c->stride));
    HIPCHK(c, hipMemcpyAsync(c->leg_valid, valid, (size_t) c->B, hipMemcpyHostToDevice, c->stream));
    c->leg_valid_on = true;
  }
  if (utimes || valid) HIPCHK(c, hipStreamSynchronize(c->stream));  // the caller's arrays are free again
  return PB_OK;
}
// The one-shot per-filter times / validity belong to the NEXT odometry or pair call, whatever becomes of it: every such entry point
// takes them FIRST, before it validates anything, so that a call that fails early cannot leave them behind for an unrelated later
// call.
struct LegMsgTimes {
  const int64_t *utimes = nullptr;
  const uint8_t *valid = nullptr;
};
static LegMsgTimes leg_take_message_times(pb_ctx *c)
{
  LegMsgTimes t;
  if (c == nullptr) return t;
  if (c->leg_ut_on) t.utimes = c->leg_ut_ext ? c->leg_ut_ext : c->leg_ut;
  if (c->leg_valid_on) t.valid = c->leg_valid_ext ? c->leg_valid_ext : c->leg_valid;
  c->leg_ut_on = c->leg_valid_on = false;
  c->leg_ut_ext = nullptr;
  c->leg_valid_ext = nullptr;
  return t;
}

extern "C" int pb_legodo_set_measurement_mode(pb_ctx *c, int mode, double r_xyz, double r_vang, double r_vang_uncertain)
{
  ENTER(c);
  if (!c->legd) return fail(c, PB_ERR_STATE, "pb_legodo_set_measurement_mode before pb_legodo_init");
  if (mode < 0 || mode > 2) return fail(c, PB_ERR_ARG, "pb_legodo_set_measurement_mode: mode 0 (lin_rate), 1 (lin_rot_rate) or 2 (pos_and_lin_rate)");
  c->leg_meas = LegMeasPar{};
  c->leg_meas.mode = mode;
  c->leg_meas.r_xyz2 = r_xyz * r_xyz;                    // bot_sq (rbis_legodo_common.cpp:38-44)
  c->leg_meas.r_a2 = r_vang * r_vang;
  c->leg_meas.r_a2_uncertain = r_vang_uncertain * r_vang_uncertain;
  if (mode == 2) c->leg_par.world_constraint = 1;        // the position it measures is leg_estimate's world constraint
  return PB_OK;
}

extern "C" int pb_legodo_set_zero_initial_velocity(pb_ctx *c, int ticks)
{
  ENTER(c);
  if (!c->legd) return fail(c, PB_ERR_STATE, "pb_legodo_set_zero_initial_velocity before pb_legodo_init");
  if (ticks > 65535) return fail(c, PB_ERR_ARG, "pb_legodo_set_zero_initial_velocity: at most 65535 ticks (16-bit per-robot counter)");
  k_legodo_reset<<<nblk(c->B), 64, 0, c->stream>>>(c->legd, c->legi, c->stride, c->B, ticks < 0 ? 0 : ticks);
  LAUNCHCHK(c);
  return PB_OK;
}

extern "C" int pb_legodo_set_control_contacts(pb_ctx *c, const int32_t *n_contacts, int mem)
{
  ENTER(c);
  if (!c->legd) return fail(c, PB_ERR_STATE, "pb_legodo_set_control_contacts before pb_legodo_init");
  if (!n_contacts) return fail(c, PB_ERR_ARG, "pb_legodo_set_control_contacts: NULL input");
  if (mem == PB_HOST_BROADCAST) {
    c->leg_nc_h[0] = n_contacts[0];
    c->leg_nc_h[1] = n_contacts[1];
    c->leg_nc_dev = false;
    return PB_OK;
  }
  if (mem != PB_HOST && mem != PB_DEVICE) return fail(c, PB_ERR_ARG, "mem must be PB_HOST, PB_DEVICE or PB_HOST_BROADCAST");
  if (!c->leg_nc) HIPCHK(c, hipMalloc((void **) &c->leg_nc, sizeof(int32_t) * 2 * (size_t) c->B));
  // kept by the context until the next call, like the handler keeps the last CONTROLLER_FOOT_CONTACT message
  HIPCHK(c, hipMemcpyAsync(c->leg_nc, n_contacts, sizeof(int32_t) * 2 * (size_t) c->B,
                           mem == PB_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
  if (mem == PB_HOST) HIPCHK(c, hipStreamSynchronize(c->stream));
  c->leg_nc_dev = true;
  return PB_OK;
}

extern "C" int pb_legodo_set_chain(pb_ctx *c, int n_left, int n_right, const int *joint_type, const int *joint_row,
                                   const double *origin_xyz_rpy, const double *axis, const float *adjustment_gain)
{
  ENTER(c);
  if (n_left < 1 || n_right < 1 || n_left > LEG_MAXJ || n_right > LEG_MAXJ)
    return fail(c, PB_ERR_ARG, "pb_legodo_set_chain: 1..%d joints per leg", LEG_MAXJ);
  if (!joint_type || !joint_row || !origin_xyz_rpy || !axis) return fail(c, PB_ERR_ARG, "pb_legodo_set_chain: NULL input");
  LegChain ch;
  memset(&ch, 0, sizeof ch);
  ch.n[0] = n_left;
  ch.n[1] = n_right;
  int max_row = -1;
  for (int side = 0, k = 0; side < 2; side++) {
    for (int j = 0; j < ch.n[side]; j++, k++) {
      const int ty = joint_type[k];
      if (ty != LJ_FIXED && ty != LJ_REVOLUTE && ty != LJ_PRISMATIC) return fail(c, PB_ERR_ARG, "pb_legodo_set_chain: joint %d: bad type %d", k, ty);
      if (ty != LJ_FIXED && joint_row[k] < 0) return fail(c, PB_ERR_ARG, "pb_legodo_set_chain: joint %d: negative row", k);
      if (ty != LJ_FIXED && joint_row[k] > max_row) max_row = joint_row[k];
      if (!leg_chain_entry(ch, side, j, ty, joint_row[k], origin_xyz_rpy + 6 * k, axis + 3 * k, adjustment_gain ? adjustment_gain[k] : 0.0f))
        return fail(c, PB_ERR_ARG, "pb_legodo_set_chain: joint %d: zero axis", k);
    }
  }
  if (!c->leg_chain) HIPCHK(c, hipMalloc((void **) &c->leg_chain, sizeof(LegChain)));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // kernels in flight may still read the old table
  HIPCHK(c, hipMemcpy(c->leg_chain, &ch, sizeof ch, hipMemcpyHostToDevice));
  c->leg_chain_h = ch;
  c->leg_chain_rows = max_row + 1;
  c->jf_ready = false;  // the filters' row list came from the old chain
  return PB_OK;
}

// ---- joint-position filters in front of the kinematics (leg_estimate.cpp:411-428) ----------------------------------
extern "C" int pb_joint_filter_init(pb_ctx *c, int mode, double process_noise_pos, double process_noise_vel, double observation_noise)
{
  ENTER(c);
  if (mode != JF_LOWPASS && mode != JF_KALMAN) return fail(c, PB_ERR_ARG, "pb_joint_filter_init: mode must be 1 (lowpass) or 2 (kalman)");
  if (!c->leg_chain) return fail(c, PB_ERR_STATE, "pb_joint_filter_init before pb_legodo_set_chain");
  JfPar par;
  memset(&par, 0, sizeof par);
  par.mode = mode;
  const LegChain &ch = c->leg_chain_h;
  for (int side = 0; side < 2; side++) {
    for (int j = 0; j < ch.n[side]; j++) {
      if ((ch.code[side][j] & LC_TYPE) == LJ_FIXED) continue;
      const int row = ch.row[side][j];
      bool seen = false;
      for (int f = 0; f < par.nf; f++) seen = seen || par.row[f] == row;
      if (!seen && row < JF_NUM_FILT_JOINTS) par.row[par.nf++] = row;  // leg_estimate.cpp:415,419: i < NUM_FILT_JOINTS
      if (ch.gain[side][j] != 0.0f) {
        bool have = false;
        for (int a = 0; a < par.nadj; a++) have = have || par.adj_row[a] == row;
        if (!have) { par.adj_row[par.nadj] = row; par.adj_gain[par.nadj++] = ch.gain[side][j]; }
      }
    }
  }
  jf_lowpass_coeffs(par.coef);
  par.pn_pos = (float) process_noise_pos;   // float members (simple_kalman_filter.hpp:39-40)
  par.pn_vel = (float) process_noise_vel;
  par.r = (float) observation_noise;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->jf_ring) { (void) hipFree(c->jf_ring); c->jf_ring = nullptr; }
  if (c->jf_kst) { (void) hipFree(c->jf_kst); c->jf_kst = nullptr; }
  c->jf_ring_h.clear();
  c->jf_kst_h.clear();
  c->jf_par = par;
  c->jf_first = true;
  c->jf_input = -1;
  c->jf_head = 0;
  c->jf_tlast = 0;
  c->jf_ready = true;
  return PB_OK;
}

extern "C" int pb_joint_filter(pb_ctx *c, int64_t utime, int n_rows, const float *joint_position, const float *joint_velocity,
                               const float *joint_effort, int mem, float *joint_position_out)
{
  ENTER(c);
  if (!c->jf_ready) return fail(c, PB_ERR_STATE, "pb_joint_filter before pb_joint_filter_init (or the chain changed since)");
  if (!joint_position || !joint_position_out) return fail(c, PB_ERR_ARG, "pb_joint_filter: NULL input");
  if (n_rows < c->leg_chain_rows) return fail(c, PB_ERR_ARG, "pb_joint_filter: the chain reads joint row %d, the block has %d rows", c->leg_chain_rows - 1, n_rows);
  JfPar &par = c->jf_par;
  if (par.mode == JF_KALMAN && !joint_velocity) return fail(c, PB_ERR_ARG, "pb_joint_filter: the Kalman filter starts from joint_velocity");
  const int input = (mem == PB_HOST_BROADCAST) ? 1 : 0;
  if (c->jf_input >= 0 && c->jf_input != input)
    return fail(c, PB_ERR_STATE, "pb_joint_filter: per-robot and one-robot messages cannot be mixed (pb_joint_filter_init starts over)");
  const double t = (double) utime * 1E-6;  // leg_estimate.cpp:422
  const double dt = t - c->jf_tlast;
  const int first = c->jf_first ? 1 : 0;
  const size_t B = (size_t) c->B, nf = (size_t) par.nf;
  auto adjusted = [&](const float *pos, const float *eff, int row, size_t at) {
    float g = 0.0f;
    for (int a = 0; a < par.nadj; a++) g = (par.adj_row[a] == row) ? par.adj_gain[a] : g;
    return eff ? torque_adjust(pos[at], eff[at], g) : pos[at];
  };
  if (input == 1) {
    // one robot's joints for every filter of the batch: a per-MESSAGE computation, done once on the host with the functions
    // the kernel runs per robot; the output is a host array the caller passes on as PB_HOST_BROADCAST
    if (c->jf_input < 0) {
      c->jf_ring_h.assign(JF_TAPS * nf, 0.0f);
      c->jf_kst_h.assign(JF_KSTATE * nf, 0.0);
    }
    for (int row = 0; row < n_rows; row++) joint_position_out[row] = adjusted(joint_position, joint_effort, row, (size_t) row);
    for (size_t f = 0; f < nf; f++) {
      const int row = par.row[f];
      const float x = joint_position_out[row];
      if (par.mode == JF_LOWPASS) {
        float *ring = c->jf_ring_h.data();
        if (first) for (int s = 0; s < JF_TAPS; s++) ring[s * nf + f] = x;
        else ring[c->jf_head * nf + f] = x;
        const int head = c->jf_head;
        joint_position_out[row] = jf_lowpass(par.coef, [&](int i) { return first ? x : ring[((head + 1 + i) % JF_TAPS) * nf + f]; });
      } else {
        double s[JF_KSTATE];
        if (first) {
          s[0] = (double) x; s[1] = (double) joint_velocity[row];
          s[2] = 1.0; s[3] = 0.0; s[4] = 0.0; s[5] = 1.0;
        } else {
          for (int i = 0; i < JF_KSTATE; i++) s[i] = c->jf_kst_h[i * nf + f];
          joint_position_out[row] = jf_kalman(s, dt, x, par.pn_pos, par.pn_vel, par.r);
        }
        for (int i = 0; i < JF_KSTATE; i++) c->jf_kst_h[i * nf + f] = s[i];
      }
    }
  } else {
    if (par.mode == JF_LOWPASS && !c->jf_ring) HIPCHK(c, hipMalloc((void **) &c->jf_ring, sizeof(float) * JF_TAPS * (nf ? nf : 1) * B));
    if (par.mode == JF_KALMAN && !c->jf_kst) HIPCHK(c, hipMalloc((void **) &c->jf_kst, sizeof(double) * JF_KSTATE * (nf ? nf : 1) * B));
    const size_t blk = sizeof(float) * (size_t) n_rows * B;
    Part p[3] = { { joint_position, blk, 0 }, { joint_velocity, joint_velocity ? blk : 0, 0 }, { joint_effort, joint_effort ? blk : 0, 0 } };
    int rc = stage_in(c, mem, p, 3);
    if (rc) return rc;
    // four robots per lane (16-byte accesses) where the batch and every block's address allow it
    const bool v4 = c->B % 4 == 0 && ((uintptr_t) p[0].dev | (uintptr_t) p[1].dev | (uintptr_t) p[2].dev | (uintptr_t) joint_position_out) % 16 == 0;
    // (64k robots, 12 chain rows, one box: low-pass 11.7 / 12.8 / 9.9 us for 1 / 2 / 4 robots per lane, Kalman 15.4 / 14.6 / 16.0 us)
    const int V = !v4 ? 1 : par.mode == JF_KALMAN ? 2 : 4, jfb = 256;
    if (V == 4)
      k_joint_filter<4><<<dim3((unsigned) ((c->B / 4 + jfb - 1) / jfb), (unsigned) n_rows), jfb, 0, c->stream>>>(
          par, c->B, (const float *) p[0].dev, (const float *) p[1].dev, (const float *) p[2].dev, joint_position_out, c->jf_ring, c->jf_kst,
          c->jf_head, first, dt);
    else if (V == 2)
      k_joint_filter<2><<<dim3((unsigned) ((c->B / 2 + jfb - 1) / jfb), (unsigned) n_rows), jfb, 0, c->stream>>>(
          par, c->B, (const float *) p[0].dev, (const float *) p[1].dev, (const float *) p[2].dev, joint_position_out, c->jf_ring, c->jf_kst,
          c->jf_head, first, dt);
    else
      k_joint_filter<1><<<dim3((unsigned) ((c->B + 255) / 256), (unsigned) n_rows), 256, 0, c->stream>>>(
          par, c->B, (const float *) p[0].dev, (const float *) p[1].dev, (const float *) p[2].dev, joint_position_out, c->jf_ring, c->jf_kst,
          c->jf_head, first, dt);
    LAUNCHCHK(c);
  }
  c->jf_input = input;
  if (!first && par.mode == JF_LOWPASS) c->jf_head = (c->jf_head + 1) % JF_TAPS;
  c->jf_first = false;
  c->jf_tlast = t;
  return PB_OK;
}

// the joint-state inputs of one message as the kernels take them (LegIn kind 1); forces may be NULL (forward kinematics only)
static int leg_in_joints(pb_ctx *c, const char *who, int n_rows, const float *jpos, const float *jeff, const float *forces, int mem,
                         LegIn &in)
{
  if (!c->leg_chain) return fail(c, PB_ERR_STATE, "%s before pb_legodo_set_chain", who);
  if (!jpos) return fail(c, PB_ERR_ARG, "%s: NULL input", who);
  if (n_rows < c->leg_chain_rows) return fail(c, PB_ERR_ARG, "%s: the chain reads joint row %d, the block has %d rows", who, c->leg_chain_rows - 1, n_rows);
  in.kind = 1;
  if (mem == PB_HOST_BROADCAST) {
    // ONE robot's joint state for every filter of the batch: its two body-to-foot transforms are a per-MESSAGE quantity, the
    // same for all filters, so they are formed once, here, with the very leg_fk the kernels run per filter for per-filter
    // joint blocks (rbis_legodo.hpp), and travel as 14 kernel arguments -- not recomputed 65 536 times on the device.
    const LegChain &ch = c->leg_chain_h;
    Pose feet[2];
    for (int side = 0; side < 2; side++) {
      double ang[LEG_MAXJ];
      leg_angles(ch, side, [&](int j) {
        const int r = ch.row[side][j];  // (0 for the slots the chain does not use: leg_fk skips them)
        return (double) (jeff ? torque_adjust(jpos[r], jeff[r], ch.gain[side][j]) : jpos[r]);
      }, ang);
      leg_fk(ch, side, ang, [&](int j, int f) { return ch.rec[side][j][f]; }, feet[side]);
    }
    for (int side = 0; side < 2; side++) {
      for (int i = 0; i < 3; i++) in.v[7 * side + i] = feet[side].t[i];
      for (int i = 0; i < 4; i++) in.v[7 * side + 3 + i] = feet[side].q[i];
    }
    if (forces) { in.v[14] = forces[0]; in.v[15] = forces[1]; }
    in.kind = 0;
    in.bcast = 1;
    return PB_OK;
  }
  const size_t blk = sizeof(float) * (size_t) n_rows * c->B;
  Part p[3] = { { jpos, blk, 0 }, { jeff, jeff ? blk : 0, 0 }, { forces, forces ? sizeof(float) * 2 * (size_t) c->B : 0, 0 } };
  int rc = stage_in(c, mem, p, 3);
  if (rc) return rc;
  in.jpos = (const float *) p[0].dev;
  in.jeff = (const float *) p[1].dev;
  in.jforces = (const float *) p[2].dev;
  return PB_OK;
}

static int legodo_launch(pb_ctx *c, LegIn &in, const LegMsgTimes &mt, const double *imu_block, int imu_mem, int64_t utime, int zero_delta, double r_vxyz,
                         double r_vxyz_uncertain, double *delta_out, double *status_out, double *lo_out, uint8_t *mask_out,
                         double *pos_out = nullptr, uint8_t *pos_ok_out = nullptr)
{
  LegAhead ah;
  if (imu_block) {
    ah.on = 1;
    if (imu_mem == PB_HOST_BROADCAST) {
      memcpy(ah.v, imu_block, sizeof(ah.v));
      ah.bcast = 1;
    } else {
      Part pi[1] = { { imu_block, sizeof(double) * 7 * c->B, 0 } };
      int rc = stage_in(c, imu_mem, pi, 1);
      if (rc) return rc;
      ah.imu = (const double *) pi[0].dev;
    }
  }
  if (c->leg_nc_dev) in.ncontacts = c->leg_nc;
  in.nc[0] = c->leg_nc_h[0];
  in.nc[1] = c->leg_nc_h[1];
  in.utimes = mt.utimes;
  in.valid = mt.valid;
  LegMeasPar mp = c->leg_meas;
  mp.r_v2 = r_vxyz * r_vxyz;                            // bot_sq (rbis_legodo_common.cpp:40-43)
  mp.r_v2_uncertain = r_vxyz_uncertain * r_vxyz_uncertain;
  // the world constraint (the transition foot's world position) is tracked from the first call that asks for the position
  if (pos_out != nullptr) c->leg_par.world_constraint = 1;
  // per-filter joint blocks: two waves per 64 robots, one leg's forward kinematics each
#define LEGODO_ARGS c->st, c->legd, c->legi, c->stride, c->B, utime, c->leg_par, in, c->leg_chain, ah, zero_delta, mp, delta_out, status_out, lo_out, mask_out, pos_out, pos_ok_out, c->k
  if (in.kind == 1) {
    if (c->ns == 15) k_legodo<15, true><<<nblk(c->B), 128, 0, c->stream>>>(LEGODO_ARGS);
    else k_legodo<21, true><<<nblk(c->B), 128, 0, c->stream>>>(LEGODO_ARGS);
  } else {
    if (c->ns == 15) k_legodo<15><<<nblk(c->B), 64, 0, c->stream>>>(LEGODO_ARGS);
    else k_legodo<21><<<nblk(c->B), 64, 0, c->stream>>>(LEGODO_ARGS);
  }
#undef LEGODO_ARGS
  LAUNCHCHK(c);
  return PB_OK;
}

static int legodo_update_impl(pb_ctx *c, const double *imu_block, int imu_mem, bool ahead, int64_t utime, const double *feet,
                              const double *forces, int mem, int zero_delta, double r_vxyz, double r_vxyz_uncertain,
                              double *delta_out, double *status_out, double *lo_out, uint8_t *mask_out)
{
  const LegMsgTimes mt = leg_take_message_times(c);
  ENTER(c);
  NEED_STATE(c);
  if (!c->legd) return fail(c, PB_ERR_STATE, "pb_legodo_update before pb_legodo_init");
  if (!feet || !forces || (ahead && !imu_block)) return fail(c, PB_ERR_ARG, "pb_legodo_update: NULL input");
  if (ahead && imu_mem == PB_HOST && mem == PB_HOST)
    return fail(c, PB_ERR_ARG, "pb_legodo_update_after_predict: the IMU block and the foot blocks cannot both be PB_HOST");
  LegIn in;
  if (mem == PB_HOST_BROADCAST) {  // one robot's foot poses for every filter: kernel arguments, no device block
    memcpy(in.v, feet, sizeof(double) * 14);
    in.v[14] = forces[0];
    in.v[15] = forces[1];
    in.bcast = 1;
  } else {
    Part p[2] = { { feet, sizeof(double) * 14 * c->B, 0 }, { forces, sizeof(double) * 2 * c->B, 0 } };
    int rc = stage_in(c, mem, p, 2);
    if (rc) return rc;
    in.feet = (const double *) p[0].dev;
    in.forces = (const double *) p[1].dev;
  }
  return legodo_launch(c, in, mt, ahead ? imu_block : nullptr, imu_mem, utime, zero_delta, r_vxyz, r_vxyz_uncertain, delta_out, status_out,
                       lo_out, mask_out);
}

extern "C" int pb_legodo_update(pb_ctx *c, int64_t utime, const double *feet, const double *forces, int mem, int zero_delta,
                                double r_vxyz, double r_vxyz_uncertain, double *delta_out, double *status_out, double *lo_out,
                                uint8_t *mask_out)
{
  return legodo_update_impl(c, nullptr, PB_DEVICE, false, utime, feet, forces, mem, zero_delta, r_vxyz, r_vxyz_uncertain, delta_out,
                            status_out, lo_out, mask_out);
}

extern "C" int pb_legodo_update_after_predict(pb_ctx *c, const double *imu_block, int imu_mem, int64_t utime, const double *feet,
                                              const double *forces, int mem, int zero_delta, double r_vxyz, double r_vxyz_uncertain,
                                              double *delta_out, double *status_out, double *lo_out, uint8_t *mask_out)
{
  return legodo_update_impl(c, imu_block, imu_mem, true, utime, feet, forces, mem, zero_delta, r_vxyz, r_vxyz_uncertain, delta_out,
                            status_out, lo_out, mask_out);
}

extern "C" int pb_legodo_update_joints(pb_ctx *c, const double *imu_block, int imu_mem, int64_t utime, int n_rows,
                                       const float *joint_position, const float *joint_effort, const float *forces, int mem,
                                       int zero_delta, double r_vxyz, double r_vxyz_uncertain, double *delta_out, double *status_out,
                                       double *lo_out, uint8_t *mask_out, double *position_out, uint8_t *position_status_out)
{
  const LegMsgTimes mt = leg_take_message_times(c);
  ENTER(c);
  NEED_STATE(c);
  if (!c->legd) return fail(c, PB_ERR_STATE, "pb_legodo_update_joints before pb_legodo_init");
  if (!forces) return fail(c, PB_ERR_ARG, "pb_legodo_update_joints: NULL input");
  if (imu_block && imu_mem == PB_HOST && mem == PB_HOST)
    return fail(c, PB_ERR_ARG, "pb_legodo_update_joints: the IMU block and the joint blocks cannot both be PB_HOST");
  LegIn in;
  int rc = leg_in_joints(c, "pb_legodo_update_joints", n_rows, joint_position, joint_effort, forces, mem, in);
  if (rc) return rc;
  return legodo_launch(c, in, mt, imu_block, imu_mem, utime, zero_delta, r_vxyz, r_vxyz_uncertain, delta_out, status_out, lo_out, mask_out,
                       position_out, position_status_out);
}

// IMU step + leg odometry + its update (LegOdoCommon's mode, pb_legodo_set_measurement_mode) for one message pair: one kernel where
// the context has it (pbk_step_leg), else the odometry kernel slaved to the state after the IMU step followed by the fused step
// (lin_rate: two launches) or by the process step and the indexed update(s) (the six-row modes); same results to rounding
static int step_leg_impl(pb_ctx *c, LegIn &in, const LegMsgTimes &mt, const double *imu_block, int imu_mem, const double q[4], int64_t utime, double r_vxyz,
                         double r_vxyz_uncertain, double *lo_out, uint8_t *mask_out)
{
  StepBcast bc;
  const double *d_imu = nullptr;
  if (imu_mem == PB_HOST_BROADCAST) {
    memcpy(bc.imu, imu_block, sizeof(bc.imu));
    bc.on = 1;
  } else {
    Part pi[1] = { { imu_block, sizeof(double) * 7 * c->B, 0 } };
    int rc = stage_in(c, imu_mem, pi, 1);
    if (rc) return rc;
    d_imu = (const double *) pi[0].dev;
  }
  if (c->leg_nc_dev) in.ncontacts = c->leg_nc;
  in.nc[0] = c->leg_nc_h[0];
  in.nc[1] = c->leg_nc_h[1];
  in.utimes = mt.utimes;
  in.valid = mt.valid;
  LegMeasPar mp = c->leg_meas;
  mp.r_v2 = r_vxyz * r_vxyz;                            // bot_sq (rbis_legodo_common.cpp:40-43)
  mp.r_v2_uncertain = r_vxyz_uncertain * r_vxyz_uncertain;
  if (mp.mode == 2) c->leg_par.world_constraint = 1;    // the measured position IS leg_estimate's world constraint, tracked from here on
  int rc = pbk_step_leg(c, d_imu, &bc, q, in, utime, mp, lo_out, mask_out);
  if (rc >= 0) return rc;
  const int rows = mp.mode == 0 ? 6 : 12;
  if (lo_out == nullptr) {  // the measurement has to pass through memory between the kernels
    const size_t bytes = sizeof(double) * 12 * (size_t) c->B + 2 * (size_t) c->B;
    if (!c->leg_lo) HIPCHK(c, hipMalloc((void **) &c->leg_lo, bytes));
    lo_out = c->leg_lo;
    mask_out = (uint8_t *) (c->leg_lo + (size_t) rows * c->B);
  }
  LegAhead ah;
  ah.on = 1;
  ah.bcast = bc.on & 1;
  memcpy(ah.v, bc.imu, sizeof(ah.v));
  ah.imu = d_imu;
  if (c->ns == 15)
    k_legodo<15><<<nblk(c->B), 64, 0, c->stream>>>(c->st, c->legd, c->legi, c->stride, c->B, utime, c->leg_par, in, c->leg_chain, ah, 0, mp, nullptr,
                                                   nullptr, lo_out, mask_out, nullptr, nullptr, c->k);
  else
    k_legodo<21><<<nblk(c->B), 64, 0, c->stream>>>(c->st, c->legd, c->legi, c->stride, c->B, utime, c->leg_par, in, c->leg_chain, ah, 0, mp, nullptr,
                                                   nullptr, lo_out, mask_out, nullptr, nullptr, c->k);
  LAUNCHCHK(c);
  if (mp.mode == 0) return pbk_step(c, true, d_imu, lo_out, mask_out, q, &bc);
  rc = pbk_step(c, false, d_imu, nullptr, nullptr, q, &bc);
  if (rc) return rc;
  static const int idx_lr[6] = { 3, 4, 5, 0, 1, 2 }, idx_pv[6] = { 9, 10, 11, 3, 4, 5 }, idx_v[3] = { 3, 4, 5 };
  const size_t B = (size_t) c->B;
  const int slot = pb_head_slot(c);  // a checkpointed step: the update(s) land in the same slot
  if (slot >= 0) c->out_slot = slot;
  rc = update_common(c, 6, mp.mode == 1 ? idx_lr : idx_pv, lo_out, lo_out + 6 * B, PB_R_DIAG, nullptr, false, mask_out, PB_DEVICE);
  if (rc || mp.mode == 1) return rc;
  const int slot2 = pb_head_slot(c);
  if (slot2 >= 0) c->out_slot = slot2;
  return update_common(c, 3, idx_v, lo_out + 3 * B, lo_out + 9 * B, PB_R_DIAG, nullptr, false, mask_out + B, PB_DEVICE);
}

extern "C" int pb_step_legodo_joints(pb_ctx *c, const double *imu_block, int imu_mem, const double q[4], int64_t utime, int n_rows,
                                     const float *joint_position, const float *joint_effort, const float *forces, int mem,
                                     double r_vxyz, double r_vxyz_uncertain, double *lo_block_out, uint8_t *mask_out)
{
  ImuIdleTake idle(c);
  const LegMsgTimes mt = leg_take_message_times(c);
  ENTER(c);
  NEED_STATE(c);
  if (!c->legd) return fail(c, PB_ERR_STATE, "pb_step_legodo_joints before pb_legodo_init");
  if (!imu_block || !q || !forces || (lo_block_out && !mask_out)) return fail(c, PB_ERR_ARG, "pb_step_legodo_joints: NULL input");
  if (imu_mem == PB_HOST && mem == PB_HOST)
    return fail(c, PB_ERR_ARG, "pb_step_legodo_joints: the IMU block and the joint blocks cannot both be PB_HOST");
  LegIn in;
  int rc = leg_in_joints(c, "pb_step_legodo_joints", n_rows, joint_position, joint_effort, forces, mem, in);
  if (rc) return rc;
  return step_leg_impl(c, in, mt, imu_block, imu_mem, q, utime, r_vxyz, r_vxyz_uncertain, lo_block_out, mask_out);
}

extern "C" int pb_step_legodo_feet(pb_ctx *c, const double *imu_block, int imu_mem, const double q[4], int64_t utime, const double *feet,
                                   const double *forces, int mem, double r_vxyz, double r_vxyz_uncertain, double *lo_block_out,
                                   uint8_t *mask_out)
{
  ImuIdleTake idle(c);
  const LegMsgTimes mt = leg_take_message_times(c);
  ENTER(c);
  NEED_STATE(c);
  if (!c->legd) return fail(c, PB_ERR_STATE, "pb_step_legodo_feet before pb_legodo_init");
  if (!imu_block || !q || !feet || !forces || (lo_block_out && !mask_out)) return fail(c, PB_ERR_ARG, "pb_step_legodo_feet: NULL input");
  if (imu_mem == PB_HOST && mem == PB_HOST)
    return fail(c, PB_ERR_ARG, "pb_step_legodo_feet: the IMU block and the foot blocks cannot both be PB_HOST");
  LegIn in;
  if (mem == PB_HOST_BROADCAST) {
    memcpy(in.v, feet, sizeof(double) * 14);
    in.v[14] = forces[0];
    in.v[15] = forces[1];
    in.bcast = 1;
  } else {
    Part p[2] = { { feet, sizeof(double) * 14 * c->B, 0 }, { forces, sizeof(double) * 2 * c->B, 0 } };
    int rc = stage_in(c, mem, p, 2);
    if (rc) return rc;
    in.feet = (const double *) p[0].dev;
    in.forces = (const double *) p[1].dev;
  }
  return step_leg_impl(c, in, mt, imu_block, imu_mem, q, utime, r_vxyz, r_vxyz_uncertain, lo_block_out, mask_out);
}

extern "C" int pb_legodo_fk(pb_ctx *c, int n_rows, const float *joint_position, const float *joint_effort, int mem, double *feet_out)
{
  ENTER(c);
  if (!feet_out) return fail(c, PB_ERR_ARG, "pb_legodo_fk: NULL output");
  LegIn in;
  int rc = leg_in_joints(c, "pb_legodo_fk", n_rows, joint_position, joint_effort, nullptr, mem, in);
  if (rc) return rc;
  k_leg_fk<<<nblk(c->B), 64, 0, c->stream>>>(in, c->leg_chain, c->B, feet_out);
  LAUNCHCHK(c);
  return PB_OK;
}

extern "C" int pb_legodo_get(pb_ctx *c, int filter, double odom_to_body[7], int64_t info[4])
{
  ENTER(c);
  if (!c->legd) return fail(c, PB_ERR_STATE, "pb_legodo_get before pb_legodo_init");
  if (filter < 0 || filter >= c->B || !odom_to_body || !info) return fail(c, PB_ERR_ARG, "pb_legodo_get: bad argument");
  int rc = stage_reserve(c, 256);
  if (rc) return rc;
  double *dp = (double *) c->stage;
  int64_t *di = (int64_t *) (dp + 8);
  k_legodo_get<<<1, 1, 0, c->stream>>>(c->legd, c->legi, c->stride, filter, dp, di);
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(odom_to_body, dp, sizeof(double) * 7, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(info, di, sizeof(int64_t) * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PB_OK;
}

extern "C" int pb_imu_notch_init(pb_ctx *c, double notch_freq, double fs)
{
  ENTER(c);
  if (!(notch_freq > 0) || !(fs > 0) || notch_freq * 4 >= fs / 2)
    return fail(c, PB_ERR_ARG, "pb_imu_notch_init: need 0 < 4*notch_freq < fs/2 (got %g, %g)", notch_freq, fs);
  if (!c->notch) HIPCHK(c, hipMalloc((void **) &c->notch, sizeof(double) * 36 * c->stride));
  HIPCHK(c, hipMemsetAsync(c->notch, 0, sizeof(double) * 36 * c->stride, c->stream));
  for (int i = 0; i < 3; i++) {
    // IIRNotch::IIRNotch + secondOrderNotch (iir_notch.cpp:3-32), notch_freq * 2^i (sensor_handlers.cpp:33-41)
    double Wo = (notch_freq * pow(2, i)) / (fs / 2);
    double BW = Wo;
    const double Ab = fabs(10 * log10(.5));
    BW = BW * M_PI;
    Wo = Wo * M_PI;
    const double Gb = pow(10, -Ab / 20.);
    const double beta = (sqrt(1.0 - Gb * Gb) / Gb) * tan(BW / 2.0);
    const double gain = 1 / (1 + beta);
    c->notch_coef.b[i][0] = gain * 1.0;
    c->notch_coef.b[i][1] = gain * (-2.0 * cos(Wo));
    c->notch_coef.b[i][2] = gain * 1;
    c->notch_coef.a[i][0] = 1.0;
    c->notch_coef.a[i][1] = -2 * gain * cos(Wo);
    c->notch_coef.a[i][2] = 2 * gain - 1;
  }
  c->notch_ready = true;
  return PB_OK;
}

static int imu_notch_impl(pb_ctx *c, const char *who, int n_packets, const int32_t *counts, const double *accel_packets, double *accel_out, int mem)
{
  ENTER(c);
  if (!c->notch_ready) return fail(c, PB_ERR_STATE, "%s before pb_imu_notch_init", who);
  if (n_packets < 0 || (n_packets > 0 && (!accel_packets || !accel_out))) return fail(c, PB_ERR_ARG, "%s: bad argument", who);
  if (n_packets == 0) return PB_OK;
  const size_t B = (size_t) c->B;
  const size_t pk_bytes = sizeof(double) * 3 * B * n_packets, pk_pad = (pk_bytes + 255) / 256 * 256;
  const size_t cn_bytes = counts ? sizeof(int32_t) * B : 0, cn_pad = (cn_bytes + 255) / 256 * 256;
  const double *d_in = accel_packets;
  const int32_t *d_counts = counts;
  double *d_out = accel_out;
  if (mem == PB_HOST) {
    int rc = stage_reserve(c, pk_pad + cn_pad + sizeof(double) * 3 * B);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->stage, accel_packets, pk_bytes, hipMemcpyHostToDevice, c->stream));
    d_in = (const double *) c->stage;
    if (counts) {
      HIPCHK(c, hipMemcpyAsync((char *) c->stage + pk_pad, counts, cn_bytes, hipMemcpyHostToDevice, c->stream));
      d_counts = (const int32_t *) ((char *) c->stage + pk_pad);
    }
    d_out = (double *) ((char *) c->stage + pk_pad + cn_pad);
    if (counts) HIPCHK(c, hipMemsetAsync(d_out, 0, sizeof(double) * 3 * B, c->stream));   // (filters without a packet: a defined 0 comes back)
  } else if (mem != PB_DEVICE) {
    return fail(c, PB_ERR_ARG, "mem must be PB_HOST or PB_DEVICE");
  }
  k_notch_counts<<<dim3((unsigned) nblk(c->B), 3u), 64, 0, c->stream>>>(c->notch, c->stride, c->B, n_packets, d_counts, d_in, d_out, c->notch_coef);
  LAUNCHCHK(c);
  if (mem == PB_HOST) {
    HIPCHK(c, hipMemcpyAsync(accel_out, d_out, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return PB_OK;
}

extern "C" int pb_imu_notch(pb_ctx *c, int n_packets, const double *accel_packets, double *accel_out, int mem)
{
  return imu_notch_impl(c, "pb_imu_notch", n_packets, nullptr, accel_packets, accel_out, mem);
}

extern "C" int pb_imu_notch_counts(pb_ctx *c, int max_packets, const int32_t *counts, const double *accel_packets, double *accel_out, int mem)
{
  if (c && !counts) return fail(c, PB_ERR_ARG, "pb_imu_notch_counts: NULL counts");
  return imu_notch_impl(c, "pb_imu_notch_counts", max_packets, counts, accel_packets, accel_out, mem);
}

extern "C" int pb_ins_body_reset(pb_ctx *c)
{
  ENTER(c);
  if (!c->ins_last) {
    HIPCHK(c, hipMalloc((void **) &c->ins_last, sizeof(double) * 6 * (size_t) c->stride));
    HIPCHK(c, hipMalloc((void **) &c->ins_prev_ut, sizeof(int64_t) * (size_t) c->stride));
  }
  HIPCHK(c, hipMemsetAsync(c->ins_last, 0, sizeof(double) * 6 * (size_t) c->stride, c->stream));
  HIPCHK(c, hipMemsetAsync(c->ins_prev_ut, 0, sizeof(int64_t) * (size_t) c->stride, c->stream));
  return PB_OK;
}

extern "C" int pb_ins_body_block(pb_ctx *c, const double *gyro, const double *accel, const double *raw_dt, const int64_t *utimes, int64_t utime,
                                 const uint8_t *valid, const double rot_quat[4], const double trans_vec[3], double dt_default, int dt_from_utimes,
                                 int mem, double *imu_block_out, uint8_t *valid_out)
{
  ENTER(c);
  if (!gyro || !accel || !rot_quat || !imu_block_out) return fail(c, PB_ERR_ARG, "pb_ins_body_block: NULL argument");
  if (mem != PB_HOST && mem != PB_DEVICE) return fail(c, PB_ERR_ARG, "pb_ins_body_block: mem must be PB_HOST or PB_DEVICE");
  if (!c->ins_last) {
    int rc = pb_ins_body_reset(c);
    if (rc) return rc;
  }
  const size_t B = (size_t) c->B;
  Part p[5] = { { gyro, sizeof(double) * 3 * B, 0 }, { accel, sizeof(double) * 3 * B, 0 }, { raw_dt, raw_dt ? sizeof(double) * B : 0, 0 },
                { utimes, utimes ? sizeof(int64_t) * B : 0, 0 }, { valid, valid ? B : 0, 0 } };
  int rc = stage_in(c, mem, p, 5);
  if (rc) return rc;
  InsFrame f;
  for (int i = 0; i < 4; i++) f.rot[i] = rot_quat[i];
  for (int i = 0; i < 3; i++) f.trans[i] = trans_vec ? trans_vec[i] : 0.0;
  f.translate = trans_vec != nullptr;
  f.dt_from_utimes = dt_from_utimes ? 1 : 0;
  f.dt_default = dt_default;
  k_ins_body<<<(unsigned) ((c->B + 255) / 256), 256, 0, c->stream>>>(c->B, c->stride, (const double *) p[0].dev, (const double *) p[1].dev,
                                                                      (const double *) p[2].dev, (const int64_t *) p[3].dev, utime,
                                                                      (const uint8_t *) p[4].dev, f, c->ins_last, c->ins_prev_ut, imu_block_out, valid_out);
  LAUNCHCHK(c);
  return PB_OK;
}

// the head goes back to the context's own array (copying it there if it currently lives in a checkpoint slot)
int detach_head(pb_ctx *c, bool keep_contents)
{
  if (c->st != c->st_base) {
    if (keep_contents)
      HIPCHK(c, hipMemcpyAsync(c->st_base, c->st, sizeof(double) * c->state_doubles, hipMemcpyDeviceToDevice, c->stream));
    c->st = c->st_base;
  }
  c->out_slot = -1;
  return PB_OK;
}

extern "C" int pb_history_reserve(pb_ctx *c, int n_slots)
{
  ENTER(c);
  if (n_slots < 0) return fail(c, PB_ERR_ARG, "pb_history_reserve: n_slots < 0");
  int rc = detach_head(c, true);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->hist) HIPCHK(c, hipFree(c->hist));
  c->hist = nullptr;
  c->nhist = 0;
  if (n_slots == 0) return PB_OK;
  const size_t bytes = sizeof(double) * c->state_doubles;
  hipError_t e = hipMalloc((void **) &c->hist, bytes * n_slots);
  if (e != hipSuccess)
    return fail(c, PB_ERR_HIP, "pb_history_reserve: %d slots x %zu bytes: %s", n_slots, bytes, hipGetErrorString(e));
  // the padding columns (batch rounded up to 64) of a slot are read by the cooperative kernel's idle lanes
  HIPCHK(c, hipMemsetAsync(c->hist, 0, bytes * n_slots, c->stream));
  c->nhist = n_slots;
  return PB_OK;
}

extern "C" int pb_set_output_slot(pb_ctx *c, int slot)
{
  ENTER(c);
  if (slot < -1 || slot >= c->nhist) return fail(c, PB_ERR_STATE, "pb_set_output_slot: checkpoint slot %d of %d", slot, c->nhist);
  c->out_slot = slot;
  return PB_OK;
}

extern "C" int pb_head_slot(const pb_ctx *c)
{
  if (!c || c->st == c->st_base || !c->hist) return -1;
  return (int) ((size_t) (c->st - c->hist) / c->state_doubles);
}

extern "C" int pb_state_save(pb_ctx *c, int slot)
{
  ENTER(c);
  NEED_STATE(c);
  if (slot < 0 || slot >= c->nhist) return fail(c, PB_ERR_STATE, "checkpoint slot %d of %d", slot, c->nhist);
  const size_t n = c->state_doubles;
  double *h = c->hist + (size_t) slot * n;
  if (h == c->st) return PB_OK;  // the head was written straight into this slot (pb_set_output_slot)
  HIPCHK(c, hipMemcpyAsync(h, c->st, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
  return PB_OK;
}

extern "C" int pb_state_restore(pb_ctx *c, int slot)
{
  ENTER(c);
  NEED_STATE(c);
  if (slot < 0 || slot >= c->nhist) return fail(c, PB_ERR_STATE, "checkpoint slot %d of %d", slot, c->nhist);
  const size_t n = c->state_doubles;
  // always into the context's own array: the slot the head may currently live in stays what it is
  HIPCHK(c, hipMemcpyAsync(c->st_base, c->hist + (size_t) slot * n, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
  c->st = c->st_base;
  c->out_slot = -1;
  return PB_OK;
}

extern "C" int pb_smooth_step(pb_ctx *c, int slot_next_pred, int slot_next, int slot_cur, int slot_out, double dt)
{
  ENTER(c);
  const int s[4] = { slot_next_pred, slot_next, slot_cur, slot_out };
  for (int i = 0; i < 4; i++)
    if (s[i] < 0 || s[i] >= c->nhist) return fail(c, PB_ERR_STATE, "pb_smooth_step: checkpoint slot %d of %d", s[i], c->nhist);
  if (slot_out == slot_next_pred || slot_out == slot_next)
    return fail(c, PB_ERR_ARG, "pb_smooth_step: slot_out may alias slot_cur only");
  const size_t n = c->state_doubles;
  const double *np_ = c->hist + (size_t) slot_next_pred * n, *ns_ = c->hist + (size_t) slot_next * n;
  const double *cu = c->hist + (size_t) slot_cur * n;
  double *out = c->hist + (size_t) slot_out * n;
  return pbk_smooth_step(c, np_, ns_, cu, out, dt);
}

// ---- whole-log RTS smoothing with bounded memory: checkpoint and recompute ----
// EKFSmoothBackwardsPass (mav_state_est.cpp:98-189) walks the WHOLE history backwards and reads, at every INS update, three
// posteriors the reference keeps by value in its update objects.  For a batch that is 2 T slots of the whole state (64k 21-state
// filters: 135 MB each -- one second of a 1 kHz log fills 288 GB).  Here the forward pass keeps only every `stride`-th posterior;
// the backward pass takes the log stretch by stretch, newest first: it re-runs the stretch's steps from its checkpoint into a
// window of 2 * stride slots (the posterior of every process step AND of the update behind it, with the very kernels the
// per-message path runs: pb_predict, pb_update_indexed) and smooths it with the smoother step.  Slots: T / stride + 2 stride + 4.
extern "C" int pb_smooth_log_slots(int n_steps, int stride)
{
  if (n_steps < 1 || stride < 1) return -1;
  return (n_steps + stride - 1) / stride + 2 * stride + 4;
}

extern "C" int pb_smooth_log(pb_ctx *c, int n_steps, int stride, const double *imu_stream, const double *lo_stream, const uint8_t *mask_stream,
                             const double q[4], double dt, int first_slot, pb_smooth_sink sink, void *user, float *elapsed_ms)
{
  ENTER(c);
  NEED_STATE(c);
  if (n_steps < 1 || stride < 1 || !imu_stream || !lo_stream || !q) return fail(c, PB_ERR_ARG, "pb_smooth_log: bad argument");
  const int K = stride, T = n_steps, M = (T + K - 1) / K, need = pb_smooth_log_slots(T, K);
  if (first_slot < 0 || first_slot + need > c->nhist)
    return fail(c, PB_ERR_STATE, "pb_smooth_log: needs checkpoint slots [%d, %d), %d are reserved (pb_history_reserve)", first_slot, first_slot + need, c->nhist);
  const size_t B = (size_t) c->B, n = c->state_doubles;
  const int CK = first_slot, WP = CK + M, WF = WP + K, PC = WF + K, FIN = PC + 1, SP = FIN + 1;   // checkpoints | window | carry | final | ping-pong
  auto slot_ptr = [&](int sl) { return c->hist + (size_t) sl * n; };
  static const int idx_v[3] = { 3, 4, 5 };
  // one step of the log exactly as the per-message path applies it: the process step, then LegOdoCommon's lin_rate update
  // (pred_slot / filt_slot < 0: in place)
  auto step = [&](int j, int pred_slot, int filt_slot) -> int {
    c->out_slot = pred_slot;
    int rc = pbk_step(c, false, imu_stream + (size_t) j * 7 * B, nullptr, nullptr, q);
    if (rc) return rc;
    c->out_slot = filt_slot;
    return update_common(c, 3, idx_v, lo_stream + (size_t) j * 6 * B, lo_stream + (size_t) j * 6 * B + 3 * B, PB_R_DIAG, nullptr, false,
                         mask_stream ? mask_stream + (size_t) j * B : nullptr, PB_DEVICE);
  };
  if (elapsed_ms) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  int rc = detach_head(c, true);
  if (rc) return rc;
  // ---- forward: the filter, a checkpoint in front of every stretch ----
  for (int j = 0; j < T; j++) {
    if (j % K == 0) {
      rc = pb_state_save(c, CK + j / K);
      if (rc) return rc;
    }
    rc = step(j, -1, -1);
    if (rc) return rc;
  }
  rc = pb_state_save(c, FIN);   // the newest posterior: its own smoothed value (and the head again when the pass is over)
  if (rc) return rc;
  // ---- backward: stretch by stretch ----
  int next_sm = FIN, toggle = 0;
  for (int m = M - 1; m >= 0; m--) {
    const int s0 = m * K, s1 = std::min(T, s0 + K) - 1;
    c->st = slot_ptr(CK + m);   // the head lives in the checkpoint: the first process step reads it there and writes into the window
    c->out_slot = -1;
    for (int j = s0; j <= s1; j++) {
      rc = step(j, WP + (j - s0), WF + (j - s0));
      if (rc) return rc;
    }
    for (int j = s1; j >= s0; j--) {
      if (j == T - 1) continue;   // (the newest step is not smoothed: mav_state_est.cpp:120-131 starts one step behind it)
      const int np = (j == s1) ? PC : WP + (j + 1 - s0);
      const int out = SP + toggle;
      rc = pbk_smooth_step(c, slot_ptr(np), slot_ptr(next_sm), slot_ptr(WF + (j - s0)), slot_ptr(out), dt);
      if (rc) return rc;
      if (sink) sink(user, j, out);
      next_sm = out;
      toggle ^= 1;
    }
    // the earlier stretch's last step needs the process-step posterior of THIS stretch's first step
    HIPCHK(c, hipMemcpyAsync(slot_ptr(PC), slot_ptr(WP), sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
  }
  rc = pb_state_restore(c, FIN);
  if (rc) return rc;
  if (elapsed_ms) {
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
  }
  return PB_OK;
}

static int calib_copy_impl(pb_ctx *c, int reps, float *elapsed_ms, uint64_t *checksum)
{
  if (reps < 1) return fail(c, PB_ERR_ARG, "pb_calib_copy: reps must be >= 1");
  double *dst = nullptr;
  const size_t bytes = sizeof(double) * c->state_doubles;
  if (checksum) {
    int rc = stage_reserve(c, 2 * sizeof(uint64_t));
    if (rc) return rc;
  }
  HIPCHK(c, hipMalloc((void **) &dst, bytes));
  hipError_t e = hipEventRecord(c->ev0, c->stream);
  for (int r = 0; r < reps && e == hipSuccess; r++) {
    k_calib_copy<<<nblk(c->B), 64, 0, c->stream>>>(c->st, dst, c->B, (int) (c->state_doubles / (size_t) c->stride / 2));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
  if (e == hipSuccess && checksum) {
    e = hipMemsetAsync(c->stage, 0, 2 * sizeof(uint64_t), c->stream);
    if (e == hipSuccess) {
      k_state_checksum<<<2048, 256, 0, c->stream>>>((const uint64_t *) dst, c->state_doubles, (unsigned long long *) c->stage);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(checksum, c->stage, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
  float ms = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
  (void) hipFree(dst);
  if (e != hipSuccess) return fail(c, PB_ERR_HIP, "pb_calib_copy failed: %s", hipGetErrorString(e));
  if (elapsed_ms) *elapsed_ms = ms;
  return PB_OK;
}

extern "C" int pb_calib_copy(pb_ctx *c, int reps, float *elapsed_ms)
{
  ENTER(c);
  return calib_copy_impl(c, reps, elapsed_ms, nullptr);
}

extern "C" int pb_calib_copy_checksum(pb_ctx *c, int reps, uint64_t out[2])
{
  ENTER(c);
  NEED_STATE(c);
  if (!out) return PB_ERR_ARG;
  return calib_copy_impl(c, reps, nullptr, out);
}

extern "C" int pb_set_utime(pb_ctx *c, int64_t utime)
{
  if (!c) return PB_ERR_ARG;
  c->utime = utime;
  return PB_OK;
}
extern "C" int64_t pb_get_utime(const pb_ctx *c) { return c ? c->utime : 0; }
