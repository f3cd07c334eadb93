// lib.hip -- library-level entry points of the C ABI (include/vvcgpu.h).
#include "common.h"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void vvcgpu_set_error(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

// ---- per-(device, stream) resources: scratch buffer + two persistent zeroed counter sets.  One table behind one mutex; slots are created on
// first use, released by vvcgpu_stream_release (a host that makes streams per job calls it before destroying the stream) or all at once by
// vvcgpu_shutdown.  A buffer that has been outgrown is NOT freed on the spot -- work queued on the stream may still read it, and hipFree is a
// device-wide synchronisation -- but parked in the slot's retired list and freed with the slot.  Capacity grows geometrically (at least doubling),
// so a stream whose batches grow slowly re-allocates O(log n) times and the parked buffers sum to less than the live one: a slot never holds more
// than about four times its largest request.  No entry point synchronises the device or another thread's stream while it holds the table's mutex.
// (A stream is driven by one host thread at a time: per-thread streams, include/vvcgpu.h.)
#include <vector>
namespace {
struct StreamSlot
{
  int device; hipStream_t stream;
  void* ptr[VVC_SCRATCH_REGIONS]; size_t cap[VVC_SCRATCH_REGIONS];   // scratch, one buffer per region (common.h)
  struct Retired { void* ptr; hipEvent_t done; };
  std::vector<Retired> retired;           // outgrown scratch buffers: freed with the slot
  int* counters; int cur; bool dirty;     // int[2][16]; dirty: a launch that owned a set failed -- both sets are cleared before the next use
  void* iotaPtr; int iotaN;               // the identity array of vvcgpu_iota: valid entries [0, iotaN) of the buffer at iotaPtr (region VVC_SCRATCH_IOTA)
};
std::vector<StreamSlot> g_slots;
std::mutex g_slotMutex;

StreamSlot* find_slot(int dev, hipStream_t stream, bool create)
{
  for (auto& s : g_slots)
    if (s.device == dev && s.stream == stream) return &s;
  if (!create) return nullptr;
  g_slots.push_back(StreamSlot{ dev, stream, {}, {}, {}, nullptr, 0, false, nullptr, 0 });
  return &g_slots.back();
}
void free_slot(StreamSlot& s)             // the slot is out of the table (or the caller holds the mutex at shutdown), its device is current, its stream is idle
{
  for (int r = 0; r < VVC_SCRATCH_REGIONS; r++) if (s.ptr[r]) (void)hipFree(s.ptr[r]);
  for (auto& q : s.retired) { (void)hipFree(q.ptr); if (q.done) (void)hipEventDestroy(q.done); }
  s.retired.clear();
  if (s.counters) (void)hipFree(s.counters);
}
}

void* vvcgpu_scratch(hipStream_t stream, size_t bytes) { return vvcgpu_scratch_region(stream, VVC_SCRATCH_ENTRY, bytes); }

void* vvcgpu_scratch_region(hipStream_t stream, int region, size_t bytes)
{
  if (region < 0 || region >= VVC_SCRATCH_REGIONS) { vvcgpu_set_error("scratch: region %d", region); return nullptr; }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { vvcgpu_set_error("hipGetDevice failed"); return nullptr; }
  {
    std::lock_guard<std::mutex> lock(g_slotMutex);
    StreamSlot* slot = find_slot(dev, stream, true);
    if (slot->cap[region] >= bytes) return slot->ptr[region];               // the hot path: no allocation, no event, no free
  }
  // Growth (O(log n) times per stream): allocate OUTSIDE the lock -- hipMalloc / hipFree may synchronise the device, and no other thread's entry
  // point should wait for that behind the table's mutex.  (A stream is driven by one host thread at a time, so the slot's region does not change
  // under us; the slot is looked up again because the table may have been re-allocated.)
  size_t have = 0;
  { std::lock_guard<std::mutex> lock(g_slotMutex); have = find_slot(dev, stream, true)->cap[region]; }
  size_t cap = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
  if (cap < 2 * have) cap = 2 * have;                                       // geometric growth
  void* p = nullptr;
  if (hipMalloc(&p, cap) != hipSuccess)
  {
    (void)hipGetLastError();                                                // the failed attempt must not surface at the caller's next launch check
    cap = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);             // the doubled size did not fit: the request itself
    if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); vvcgpu_set_error("scratch: hipMalloc(%zu) failed", cap); return nullptr; }
  }
  // The outgrown buffer: queued work on the stream may still read it, so it is parked with an event recorded behind that work; parked buffers
  // whose event has completed are freed HERE, on the (rare) growth path and outside the lock, the rest with the slot (vvcgpu_stream_release /
  // vvcgpu_shutdown).  Capacities at least double, so the parked buffers of a slot sum to less than its current capacity.
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, stream) != hipSuccess)
  { (void)hipGetLastError(); if (ev) (void)hipEventDestroy(ev); ev = nullptr; }   // no event: the buffer stays parked until the slot goes
  std::vector<StreamSlot::Retired> done;
  {
    std::lock_guard<std::mutex> lock(g_slotMutex);
    StreamSlot* slot = find_slot(dev, stream, true);
    for (size_t i = 0; i < slot->retired.size();)
    {
      if (slot->retired[i].done && hipEventQuery(slot->retired[i].done) == hipSuccess) { done.push_back(slot->retired[i]); slot->retired.erase(slot->retired.begin() + (ptrdiff_t)i); }
      else i++;
    }
    (void)hipGetLastError();                                                // hipErrorNotReady of a query is not an error of the caller
    if (slot->ptr[region]) slot->retired.push_back(StreamSlot::Retired{ slot->ptr[region], ev });
    else if (ev) { (void)hipEventDestroy(ev); }
    slot->ptr[region] = p; slot->cap[region] = cap;
  }
  for (auto& q : done) { (void)hipFree(q.ptr); (void)hipEventDestroy(q.done); }
  return p;
}

// identity array 0, 1, 2, ... of at least n ints, persistent per (device, stream): written by a launch on that stream when it is first needed or has to
// grow (geometrically, like the scratch), read by every later call (vvcgpu_resi_chain_runs_batch: class lists as ranges of it)
namespace { __global__ void iota_kernel(int* p, int first, int n) { const int i = first + blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = i; } }
int* vvcgpu_iota(hipStream_t stream, int n)
{
  int* p = static_cast<int*>(vvcgpu_scratch_region(stream, VVC_SCRATCH_IOTA, (size_t)(n > 0 ? n : 1) * sizeof(int)));
  if (!p) return nullptr;
  int dev = 0, first = 0, cap = 0;
  if (hipGetDevice(&dev) != hipSuccess) { vvcgpu_set_error("hipGetDevice failed"); return nullptr; }
  {
    std::lock_guard<std::mutex> lock(g_slotMutex);
    StreamSlot* slot = find_slot(dev, stream, true);
    if (slot->iotaPtr != p) { slot->iotaPtr = p; slot->iotaN = 0; }
    first = slot->iotaN;
    cap = (int)std::min<size_t>(slot->cap[VVC_SCRATCH_IOTA] / sizeof(int), (size_t)0x7FFFFFFF);
    if (first < n) slot->iotaN = cap;                                     // the whole buffer is filled below
  }
  if (first < n)
  {
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((cap - first + 255) / 256)), dim3(256), 0, stream, p, first, cap);
    if (hipGetLastError() != hipSuccess) { vvcgpu_set_error("iota: kernel launch failed"); return nullptr; }
  }
  return p;
}

// Two persistent work counters per (device, stream), zero when handed out: a launch that needs a zeroed counter takes counter `cur` and clears
// counter `cur ^ 1` for the next call inside its own kernel (the previous user of that one has finished: same stream), so no fill launch is
// needed in front of it.  EVERY user clears all VVC_CTR_INTS ints of the other set, whatever it uses of its own.  Returns a device pointer to
// int[2][VVC_CTR_INTS] and the index to use; nullptr on failure.  A caller whose launches fail after this call reports it with
// vvcgpu_counters_failed: the sets are then cleared by a memset on the stream in front of the next user.
int* vvcgpu_counters(hipStream_t stream, int* cur)
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { vvcgpu_set_error("hipGetDevice failed"); return nullptr; }
  std::lock_guard<std::mutex> lock(g_slotMutex);
  StreamSlot* slot = find_slot(dev, stream, true);
  if (!slot->counters)
  {
    void* p = nullptr;
    if (hipMalloc(&p, 2 * VVC_CTR_INTS * sizeof(int)) != hipSuccess) { vvcgpu_set_error("counters: hipMalloc failed"); return nullptr; }
    if (hipMemset(p, 0, 2 * VVC_CTR_INTS * sizeof(int)) != hipSuccess) { (void)hipFree(p); vvcgpu_set_error("counters: hipMemset failed"); return nullptr; }
    slot->counters = static_cast<int*>(p); slot->cur = 0; slot->dirty = false;
  }
  if (slot->dirty)
  {
    if (hipMemsetAsync(slot->counters, 0, 2 * VVC_CTR_INTS * sizeof(int), stream) != hipSuccess) { vvcgpu_set_error("counters: hipMemsetAsync failed"); return nullptr; }
    slot->dirty = false; slot->cur = 0;
  }
  *cur = slot->cur;
  slot->cur ^= 1;
  return slot->counters;
}
void vvcgpu_counters_failed(hipStream_t stream)
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_slotMutex);
  if (StreamSlot* slot = find_slot(dev, stream, false)) slot->dirty = true;
}

int vvcgpu_no_mfma(void)
{
  const char* e = getenv("VVCGPU_NO_MFMA");                                 // read per call (a few hundred ns): tools toggle it inside one process
  return e && e[0] == '1';
}

// compute units of the current device, cached per device (persistent kernels size their grids with it on every call)
int vvcgpu_cu_count(void)
{
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  cached[dev].store(cus, std::memory_order_relaxed);
  return cus;
}

extern "C" {
int vvcgpu_version(void) { return 1; }
const char* vvcgpu_last_error(void) { return g_err; }
int vvcgpu_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int vvcgpu_malloc(void** dev_ptr, size_t bytes)
{
  VVC_CHECK_ARG(dev_ptr, "malloc: null out pointer");
  VVC_HIP(hipMalloc(dev_ptr, bytes ? bytes : 1));
  return VVCGPU_OK;
}
int vvcgpu_free(void* p) { if (p) VVC_HIP(hipFree(p)); return VVCGPU_OK; }
int vvcgpu_memcpy_h2d(void* d, const void* s, size_t n, void* st) { VVC_HIP(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st)); return VVCGPU_OK; }
int vvcgpu_memcpy_d2h(void* d, const void* s, size_t n, void* st) { VVC_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st)); return VVCGPU_OK; }
int vvcgpu_memcpy2d_h2d(void* d, size_t dp, const void* s, size_t sp, size_t wb, size_t h, void* st)
{ VVC_HIP(hipMemcpy2DAsync(d, dp, s, sp, wb, h, hipMemcpyHostToDevice, (hipStream_t)st)); return VVCGPU_OK; }
int vvcgpu_memcpy2d_d2h(void* d, size_t dp, const void* s, size_t sp, size_t wb, size_t h, void* st)
{ VVC_HIP(hipMemcpy2DAsync(d, dp, s, sp, wb, h, hipMemcpyDeviceToHost, (hipStream_t)st)); return VVCGPU_OK; }
int vvcgpu_memcpy2d_d2d(void* d, size_t dp, const void* s, size_t sp, size_t wb, size_t h, void* st)
{ VVC_HIP(hipMemcpy2DAsync(d, dp, s, sp, wb, h, hipMemcpyDeviceToDevice, (hipStream_t)st)); return VVCGPU_OK; }
int vvcgpu_stream_sync(void* st) { VVC_HIP(hipStreamSynchronize((hipStream_t)st)); return VVCGPU_OK; }
int vvcgpu_sizeof(int id)
{
  switch (id)
  {
  case 0: return (int)sizeof(vvcgpu_sao_ctu);
  case 1: return (int)sizeof(vvcgpu_deblock_cfg);
  case 2: return (int)sizeof(vvcgpu_dist_desc);
  case 3: return (int)sizeof(vvcgpu_search_blk);
  case 4: return (int)sizeof(vvcgpu_mvcost);
  case 5: return (int)sizeof(vvcgpu_search_best);
  case 6: return (int)sizeof(vvcgpu_if_desc);
  case 7: return (int)sizeof(vvcgpu_mc_desc);
  case 8: return (int)sizeof(vvcgpu_pelop_desc);
  case 9: return (int)sizeof(vvcgpu_pelop_cfg);
  case 10: return (int)sizeof(vvcgpu_tr_desc);
  case 11: return (int)sizeof(vvcgpu_frac_blk);
  case 12: return (int)sizeof(vvcgpu_frac_result);
  case 13: return (int)sizeof(vvcgpu_dqtr_desc);
  case 14: return (int)sizeof(vvcgpu_afg_desc);
  case 15: return (int)sizeof(vvcgpu_afe_desc);
  case 16: return (int)sizeof(vvcgpu_tz_pu);
  case 17: return (int)sizeof(vvcgpu_tz_cfg);
  case 18: return (int)sizeof(vvcgpu_intra_desc);
  case 19: return (int)sizeof(vvcgpu_cclm_desc);
  case 20: return (int)sizeof(vvcgpu_intra_fill_desc);
  case 21: return (int)sizeof(vvcgpu_imv_pu);
  case 22: return (int)sizeof(vvcgpu_imv_result);
  case 23: return (int)sizeof(vvcgpu_quant_desc);
  case 24: return (int)sizeof(vvcgpu_dq_rates);
  case 25: return (int)sizeof(vvcgpu_depquant_desc);
  case 26: return (int)sizeof(vvcgpu_rdoq_rates);
  case 27: return (int)sizeof(vvcgpu_rdoq_desc);
  case 28: return (int)sizeof(vvcgpu_intra_satd_desc);
  case 29: return (int)sizeof(vvcgpu_affine_iter);
  case 30: return (int)sizeof(vvcgpu_me_hier_cfg);
  default: return -1;
  }
}
int vvcgpu_warmup(int bit_depth)
{
  if (bit_depth < 8 || bit_depth > 10) { vvcgpu_set_error("warmup: bit depth %d outside 8..10", bit_depth); return VVCGPU_E_UNSUPPORTED; }
  int rc = vvcgpu_tr_image_build();
  if (rc == VVCGPU_OK) rc = vvcgpu_mc_image_build(bit_depth);
  if (rc == VVCGPU_OK) rc = vvcgpu_frac_image_build(bit_depth);
  return rc;
}
int vvcgpu_set_device(int device)
{
  VVC_HIP(hipSetDevice(device));
  return VVCGPU_OK;
}
int vvcgpu_stream_release(void* stream)
{
  int dev = 0;
  VVC_HIP(hipGetDevice(&dev));
  // the slot of this stream handle -- on the current device first, else on whichever device holds one (a stream belongs to one device; the caller
  // may have switched devices since it used the stream).  Looked up under the lock; the drain of the stream runs WITHOUT it (other threads' entry
  // points keep going); the slot is taken out of the table under the lock again and freed outside it.
  int sdev = -1;
  {
    std::lock_guard<std::mutex> lock(g_slotMutex);
    for (auto& s : g_slots)
      if (s.stream == (hipStream_t)stream && (s.device == dev || sdev < 0)) { sdev = s.device; if (s.device == dev) break; }
  }
  if (sdev < 0) return VVCGPU_OK;                                           // nothing held for this stream
  if (sdev != dev) VVC_HIP(hipSetDevice(sdev));
  const hipError_t e = hipStreamSynchronize((hipStream_t)stream);           // queued work may still read the buffers
  if (e == hipSuccess)
  {
    StreamSlot taken{ sdev, nullptr, {}, {}, {}, nullptr, 0, false, nullptr, 0 };
    bool found = false;
    {
      std::lock_guard<std::mutex> lock(g_slotMutex);
      for (size_t i = 0; i < g_slots.size(); i++)
        if (g_slots[i].stream == (hipStream_t)stream && g_slots[i].device == sdev) { taken = g_slots[i]; g_slots.erase(g_slots.begin() + (ptrdiff_t)i); found = true; break; }
    }
    if (found) free_slot(taken);
  }
  if (sdev != dev) (void)hipSetDevice(dev);
  if (e != hipSuccess) { vvcgpu_set_error("stream_release: hipStreamSynchronize failed on device %d: %s (resources kept)", sdev, hipGetErrorString(e)); return VVCGPU_E_DEVICE; }
  return VVCGPU_OK;
}
int vvcgpu_shutdown(void)
{
  int dev0 = 0;
  VVC_HIP(hipGetDevice(&dev0));
  std::lock_guard<std::mutex> lock(g_slotMutex);
  std::vector<StreamSlot> kept;                                             // slots whose device could not be reached: kept, and reported
  for (auto& s : g_slots)
  {
    if (hipSetDevice(s.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { kept.push_back(s); continue; }
    free_slot(s);
  }
  const size_t nkept = kept.size();
  g_slots.swap(kept);
  VVC_HIP(hipSetDevice(dev0));
  if (nkept) { vvcgpu_set_error("shutdown: %zu stream slot(s) on unreachable devices were kept", nkept); return VVCGPU_E_DEVICE; }
  return VVCGPU_OK;
}
}
